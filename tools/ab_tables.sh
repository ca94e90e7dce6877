#!/bin/bash
# development: bench.py --roofline-only under two tuner tables on the same box, alternating (A B A B)
for i in 1 2; do
  for t in "$@"; do
    SR_BENCH_TUNE=free SR_AUTOTUNE_TABLES=$t python bench.py --roofline-only $ARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read())['roofline']; print('$t', 'igemm', d['achieved'], 'TF/s, UNet eval', d['unet_eval_ms'], 'ms')"
  done
done
