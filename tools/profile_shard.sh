#!/bin/bash
# Round profile of what ONE rank of a view-sharded 8-view group executes (development tool; GPU box, repo root):
#   bash tools/profile_shard.sh  ->  one line per (views per rank, calls in flight): ms per call, frames/s of the rank, exposed comm, UNet eval ms, igemm TF/s
# SR_SHARD_FORCE=1 keeps every collective of the sharded path in a ONE-rank RCCL group (see profiles/r03_shard_rank_profile.txt).
export SR_SHARD_FORCE=1
R=$GRAFT_REPO_ROOT
for V in 8 4 2 1; do
  for FL in "--no-shard-inflight" "--inflight 3"; do
    python3 $R/bench.py --mode shard --views $V --steps 6 --warmup 1 --no-cpu-baseline $FL 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('views/rank $V  $FL:', d['ms_per_step'], 'ms per call,', round(d['value'], 2), 'frames/s of the rank, exposed comm', d['exposed_comm_ms_per_denoise_step'], 'ms/step, UNet eval', r['unet_eval_ms'], 'ms (', r['unet_eval_launches'], 'ops ), igemm', r['achieved'], 'TF/s')"
  done
done
echo "plain (unsharded) path, one call at a time and three in flight:"
python3 $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms per call (3 in flight) =', d['value'], 'frames/s; one at a time', d['value_1_in_flight'], 'frames/s; UNet eval', d['roofline']['unet_eval_ms'], 'ms')"
