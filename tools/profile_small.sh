#!/bin/bash
# Small-batch profile (development tool; GPU box, repo root: `bash tools/profile_small.sh r04`): what a one-view rank of the 8-GPU
# shard (B = 2) and a three-frame rank of config 4 (B = 6) execute per UNet evaluation.
#   per-op in-sequence table (tools/profile_plan.py seq) + rocprofv3 --kernel-trace --stats of `bench.py --roofline-only --views V`
set -e
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/small_$TAG
mkdir -p $OUT
export SR_AUTOTUNE_CACHE=$OUT/tune.json                  # (starts from the pinned table; the B = 6 shapes it lacks are tuned once, here)
cp $R/tests/golden/tune_table.json $SR_AUTOTUNE_CACHE
cd /tmp && export TMPDIR=/tmp
for V in 1 3; do
  SR_VIEWS=$V python3 $R/tools/profile_plan.py unet f16 seq > $OUT/plan_seq_views$V.txt 2> $OUT/plan_seq_views$V.err
  echo "views $V: $(head -3 $OUT/plan_seq_views$V.txt | tr '\n' ' ')"
  rocprofv3 --kernel-trace --stats -d $OUT/roofline_v$V -o out --output-format csv -- python3 $R/bench.py --roofline-only --views $V > $OUT/roofline_only_views$V.json 2> $OUT/roofline_only_views$V.err
  cat $OUT/roofline_only_views$V.json
done
find $OUT -name "*kernel_stats.csv"
