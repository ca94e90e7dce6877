#!/bin/bash
# development: does preferring the columns-first tile order where it is time-neutral alone (less fabric traffic) help once calls are in flight?
for m in 0.97 1.02; do
  for cfg in "--views 1 --inflight 4 --steps 12" "--views 8 --inflight 3 --steps 6"; do
    SR_BENCH_TUNE=free SR_TUNE_ORDER_MARGIN=$m python bench.py $cfg --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('margin $m  $cfg:', d['ms_per_step'], 'ms per call,', round(d['value'],2), 'frames/s; one at a time', d['value_1_in_flight'], '; eval', d['roofline']['unet_eval_ms'], 'ms, igemm', d['roofline']['achieved'])"
  done
done
