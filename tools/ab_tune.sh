#!/bin/bash
# development aid: UNet evaluation (bench.py --roofline-only) with back-to-back vs cold-cache tile tuning, per view count (one box)
for views in "$@"; do
  for cold in 0 1; do
    SR_TUNE_COLD=$cold SR_AUTOTUNE_CACHE=gpurun_out/tune_v${views}_c${cold}.json python bench.py --roofline-only --views $views 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read())['roofline']; print('views $views cold $cold', r['achieved'], r['unet_eval_ms'])"
  done
done
