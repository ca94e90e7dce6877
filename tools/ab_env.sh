#!/bin/bash
# development aid: `[BENCH_ARGS="--views 1"] bash tools/ab_env.sh VAR v1 v2 [v1 v2 ...]` -> igemm TF/s and UNet evaluation ms of `bench.py --roofline-only` per value (one box)
VAR=$1; shift
export SR_AUTOTUNE_CACHE=${SR_AUTOTUNE_CACHE:-gpurun_out/ab_tune.json}
for v in "$@"; do
  env $VAR=$v python bench.py --roofline-only $BENCH_ARGS 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read())['roofline']; print('$BENCH_ARGS $VAR=$v', r['achieved'], r['unet_eval_ms'])"
done
