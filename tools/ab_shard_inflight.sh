#!/bin/bash
# development aid: one rank of a view shard (world-1 RCCL group): plan segments eager / as hipGraphs x calls in flight
export SR_SHARD_FORCE=1
for v in "$@"; do
  for seg in ${SEGS:-0 1}; do
    for fl in "--no-shard-inflight" "--inflight 3" "--inflight 5"; do
      SR_SHARD_GRAPH_SEGMENTS=${seg/auto/} python bench.py --mode shard --views $v --steps 6 --warmup 1 --no-cpu-baseline $fl 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('views/rank $v segments-as-graphs=$seg $fl:', d['ms_per_step'], 'ms per call', round(d['value'],2), 'frames/s; exposed comm', d['exposed_comm_ms_per_denoise_step'])"
    done
  done
done
