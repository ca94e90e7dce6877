#!/bin/bash
# development: re-time the pinned tuner tables from nothing on the GPU box (run from the repo root; results under gpurun_out/regen/).
#   1. tests/golden/tune_table.json: every shape the full-size parity tests and bench.py meet, tuned by ONE process each
#   2. tests/golden/bench_check.json: bench.py --record-check on the new table
#   3. tests/golden/tune_table_ranks.json: what the multi-process tests meet beyond that
set -e
OUT=gpurun_out/regen
mkdir -p $OUT/a $OUT/r
mv tests/golden/tune_table.json $OUT/old_tune_table.json
mv tests/golden/tune_table_ranks.json $OUT/old_tune_table_ranks.json
SR_AUTOTUNE_DUMP=$OUT/a/t python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_config1_psnr.py tests/test_gpu_e2e.py tests/test_gpu_models.py -m gpu -x -q > $OUT/tests_a.txt 2>&1 || { tail -30 $OUT/tests_a.txt; exit 1; }
tail -3 $OUT/tests_a.txt
python - <<'PY'
import glob, json
out = {}
for p in sorted(glob.glob("gpurun_out/regen/a/t.*.json")):
    for k, v in json.load(open(p)).items():
        out.setdefault(k, v)
json.dump(dict(sorted(out.items())), open("tests/golden/tune_table.json", "w"), indent=0)
print(len(out), "shapes in tune_table.json;", sum(1 for v in out.values() if len(v) > 2), "columns first")
PY
SR_BENCH_TUNE=free SR_AUTOTUNE_TABLES=tests/golden/tune_table.json SR_AUTOTUNE_DUMP=$OUT/a/b python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_free.json 2> $OUT/bench_free.err
python - <<'PY'
import glob, json
t = json.load(open("tests/golden/tune_table.json"))
n0 = len(t)
for p in sorted(glob.glob("gpurun_out/regen/a/b.*.json")):
    for k, v in json.load(open(p)).items():
        t.setdefault(k, v)
json.dump(dict(sorted(t.items())), open("tests/golden/tune_table.json", "w"), indent=0)
print("bench.py added", len(t) - n0, "shapes")
PY
cp tests/golden/tune_table.json $OUT/tune_table.json
python bench.py --record-check --no-cpu-baseline --steps 6 > $OUT/bench_record.json 2> $OUT/bench_record.err
cp tests/golden/bench_check.json $OUT/bench_check.json
python bench.py --no-cpu-baseline --steps 6 > $OUT/bench_again.json 2> $OUT/bench_again.err
python -c "
import json
d = json.loads(open('$OUT/bench_again.json').read().strip().splitlines()[-1])
print('bench on the new table:', d['value'], 'frames/s, check', d['check']['matches_recorded'], 'igemm', d['roofline']['achieved'], 'eval', d['roofline']['unet_eval_ms'])"
SR_AUTOTUNE_DUMP=$OUT/r/t python -m pytest tests/test_gpu_sharded.py tests/test_bench_launcher.py -m gpu -x -q > $OUT/tests_r.txt 2>&1 || { tail -30 $OUT/tests_r.txt; exit 1; }
tail -3 $OUT/tests_r.txt
python tools/merge_tune_dumps.py $OUT/r/t
cp tests/golden/tune_table_ranks.json $OUT/tune_table_ranks.json
