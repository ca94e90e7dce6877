#!/bin/bash
# development aid: `[SR_VIEWS=1] bash tools/ab_plan.sh VAR v1 v2 ...` -> eager / hipGraph time of the UNet step plan per value (one box)
VAR=$1; shift
export SR_AUTOTUNE_CACHE=${SR_AUTOTUNE_CACHE:-gpurun_out/ab_tune.json}
for v in "$@"; do
  echo "== SR_VIEWS=${SR_VIEWS:-8} $VAR=$v"
  env $VAR=$v python tools/profile_plan.py unet f16 2>/dev/null | head -3
done
