"""Merge the per-process tuner dumps of a GPU run (SR_AUTOTUNE_DUMP=<prefix>) into tests/golden/tune_table_ranks.json: the shapes
the multi-process tests meet that the pinned single-process table (tests/golden/tune_table.json) does not hold.  First dump wins."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
main = json.load(open(os.path.join(ROOT, "tests", "golden", "tune_table.json")))
out = {}
for path in sorted(glob.glob(sys.argv[1] + ".*.json")):
    for k, v in json.load(open(path)).items():
        if k not in main:
            out.setdefault(k, v)
dst = os.path.join(ROOT, "tests", "golden", "tune_table_ranks.json")
json.dump(dict(sorted(out.items())), open(dst, "w"), indent=0)
print(len(out), "shapes ->", dst)
