#!/bin/bash
# HBM-side traffic of the igemm family at another batch size (development tool; GPU box, repo root): bash tools/profile_traffic.sh r04 1
#   -> gpurun_out/prof_${TAG}_views${V}/traffic.json (copy to profiles/${TAG}_igemm_traffic_views${V}.json) + the kernel summary of the same replay
set -e
TAG=${1:-r04}
V=${2:-1}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_${TAG}_views$V
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# shapes the pinned table does not hold are tuned ONCE here, un-profiled; the counted runs read that table and make no tuning launches
rm -f $OUT/tuned.*.json
SR_AUTOTUNE_DUMP=$OUT/tuned python3 $R/bench.py --roofline-only --views $V > $OUT/roofline_only_unprofiled.json 2> $OUT/roofline_only_unprofiled.err
export SR_AUTOTUNE_TABLES=$(ls $OUT/tuned.*.json | head -1)
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o out --output-format csv -- python3 $R/bench.py --roofline-only --views $V > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o out --output-format csv -- python3 $R/bench.py --roofline-only --views $V > /dev/null 2>&1
python3 $R/tools/summarize_traffic.py $OUT "bench.py --roofline-only --views $V (sd15-512, $V view(s), f16, B=$((2*V)) UNet evaluation)" > $OUT/traffic.json
cat $OUT/traffic.json
tail -1 $OUT/roofline_only_unprofiled.json | python3 -c "import sys,json; r=json.loads(sys.stdin.read())['roofline']; print('algorithmic bytes', r['algorithmic_bytes'], 'igemm', r['achieved'], 'TF/s, eval', r['unet_eval_ms'], 'ms')"
