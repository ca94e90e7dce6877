#!/bin/bash
# Kernel-level summaries of the OTHER workloads bench.py measures (VERDICT r3 missing #7): the ControlNet pair (BASELINE config 4's
# composition) and the SDXL 1024^2 workload (config 5), plus the VAE decoder alone -- rocprofv3 --kernel-trace --stats of
# `bench.py --roofline-only ...`, and their un-profiled bench lines.  bash tools/profile_other.sh r04
set -e
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/other_$TAG
mkdir -p $OUT
export SR_AUTOTUNE_CACHE=$OUT/tune.json
cp $R/tests/golden/tune_table.json $SR_AUTOTUNE_CACHE
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --controlnets --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_controlnets.json 2> $OUT/bench_controlnets.err
echo "controlnets: $(python3 -c "import json; d=json.load(open('$OUT/bench_controlnets.json')); print(d['value'], 'frames/s, igemm', d['roofline']['achieved'], 'TF/s, eval', d['roofline']['unet_eval_ms'], 'ms')")"
rocprofv3 --kernel-trace --stats -d $OUT/controlnets -o out --output-format csv -- python3 $R/bench.py --controlnets --roofline-only > $OUT/roofline_controlnets.json 2> $OUT/roofline_controlnets.err
python3 $R/bench.py --workload sdxl-1024 --steps 2 --warmup 1 > $OUT/bench_sdxl.json 2> $OUT/bench_sdxl.err
echo "sdxl-1024: $(python3 -c "import json; d=json.load(open('$OUT/bench_sdxl.json')); print(d['value'], 'frames/s, igemm', d['roofline']['achieved'], 'TF/s, eval', d['roofline']['unet_eval_ms'], 'ms')")"
rocprofv3 --kernel-trace --stats -d $OUT/sdxl -o out --output-format csv -- python3 $R/bench.py --workload sdxl-1024 --roofline-only > $OUT/roofline_sdxl.json 2> $OUT/roofline_sdxl.err
find $OUT -name "*kernel_stats.csv"
