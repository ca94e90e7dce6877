#!/bin/bash
# development aid: the plain (unsharded) frame loop at small view counts against the number of calls in flight
for v in "$@"; do
  for fl in ${FLS:-1 2 3 4 6}; do
    python bench.py --views $v --steps 12 --warmup 1 --no-cpu-baseline --inflight $fl 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('views $v, $fl call(s) in flight:', d['ms_per_step'], 'ms per call,', round(d['value'],2), 'frames/s; UNet eval', d['roofline']['unet_eval_ms'], 'ms')"
  done
done
