#!/bin/bash
# Round profile (development tool; run on the GPU box from the repo root: `bash tools/profile_round.sh r03`).
#   1. tile-tuner table written once (un-profiled) so the profiled runs make no tuning launches
#   2. rocprofv3 --kernel-trace --stats of the bench command and of `bench.py --roofline-only`
#   3. two --pmc passes (FETCH_SIZE, WRITE_SIZE; separate: together they exceed the TCC slots) of `bench.py --roofline-only`
#      -> HBM-side bytes of the igemm family per UNet evaluation (tools/summarize_traffic.py)
set -e
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
# (bench.py pins the tuner table itself since round 4: tests/golden/tune_table.json -- the profiled runs make no tuning launches)
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --roofline-only > $OUT/roofline_only_unprofiled.json 2> $OUT/roofline_only_unprofiled.err
rocprofv3 --kernel-trace --stats -d $OUT/roofline -o out --output-format csv -- python3 $R/bench.py --roofline-only > $OUT/roofline_only.json 2> $OUT/roofline_only.err
rocprofv3 --kernel-trace --stats -d $OUT/bench -o out --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o out --output-format csv -- python3 $R/bench.py --roofline-only > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o out --output-format csv -- python3 $R/bench.py --roofline-only > /dev/null 2>&1
find $OUT -name "*kernel_stats.csv" -o -name "*counter_collection.csv" | head -20
python3 $R/tools/summarize_traffic.py $OUT > $OUT/traffic.json
cat $OUT/traffic.json
