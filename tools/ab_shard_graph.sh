#!/bin/bash
# development aid: one rank of a view shard (world-1 RCCL group) with its plan segments replayed as hipGraphs vs launched eagerly
export SR_SHARD_FORCE=1 SR_AUTOTUNE_CACHE=gpurun_out/ab_shard_tune.json
for v in "$@"; do
  for g in "" "--no-graph"; do
    python bench.py --mode shard --views $v --steps 4 --warmup 1 --no-cpu-baseline $g 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('views $v graph=$g', d['ms_per_step'], round(d['value'],2), d['exposed_comm_ms_per_denoise_step'])"
  done
done
