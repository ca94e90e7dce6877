"""HBM-side traffic of the igemm family per UNet evaluation from the two PMC passes of tools/profile_round.sh (development tool).
FETCH_SIZE / WRITE_SIZE are reported in KB per dispatch; on gfx950 FETCH_SIZE counts a wide coalesced read at half its bytes
(MI355X_MICROARCH.md, HBM section): doubled here.  `bench.py --roofline-only` makes 7 replays of the igemm subset + 5 of the whole
step plan + 1 run of the whole plan on a real latent beforehand = 13 evaluations' worth of igemm launches (the few extra launches of the same kernels in the prologue plan — cross-attention
K/V and time-embedding GEMMs, run once per sampling run — are counted in: < 0.5 % of the bytes)."""
import csv
import glob
import importlib.util
import json
import os
import sys

out = sys.argv[1]
FAMILY = ("igemm_kernel", "igemm_group_kernel", "conv3p_kernel", "splitk_reduce_kernel")


def total(kind):
    s, n = 0.0, 0
    for f in glob.glob(f"{out}/{kind}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if any(k in r["Kernel_Name"] for k in FAMILY):
                s += float(r["Counter_Value"])
                n += 1
    return s, n


fetch_kb, nf = total("fetch")
write_kb, nw = total("write")
evals = 13 if nf else 0          # 1 run of the whole plan on a real latent + 7 replays of the igemm subset + 5 of the whole step plan (bench.py --roofline-only)
res = {"fetch_size_kb_sum": fetch_kb, "write_size_kb_sum": write_kb, "dispatches_fetch_pass": nf, "dispatches_write_pass": nw,
       "evaluations_in_pass": evals,
       "fetch_bytes_per_eval_corrected": None if not evals else 2.0 * fetch_kb * 1024 / evals,
       "write_bytes_per_eval": None if not evals else write_kb * 1024 / evals,
       "note": "FETCH_SIZE x2 (gfx950 counts 16-B-per-lane reads at half); igemm family = igemm_kernel / igemm_group_kernel tiles + conv3p_kernel + splitk_reduce_kernel"}
if evals:
    res["hbm_bytes_per_eval"] = res["fetch_bytes_per_eval_corrected"] + res["write_bytes_per_eval"]
# identity of the kernels that were counted: bench.py quotes the figure only while the loaded library has this source hash
_bp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "stable-renderer_amd", "csrc", "build.py")
_spec = importlib.util.spec_from_file_location("sr_build", _bp)
_bm = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_bm)
res["source_hash"] = _bm.source_hash()
res["workload"] = sys.argv[2] if len(sys.argv) > 2 else "bench.py --roofline-only (sd15-512, 8 views, f16, B=16 UNet evaluation)"
print(json.dumps(res, indent=1))
