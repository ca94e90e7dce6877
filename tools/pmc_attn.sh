#!/bin/bash
# PMC passes over the d=40 attention micro-benchmark (development tool; run on the GPU box from the repo root)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_LEVEL_WAVES"; do
  n=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp -d $R/gpurun_out/pmc_attn/$n -o out --output-format csv -- python3 $R/tools/bench_attn.py 16 4096 4096 8 40 1 > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$R/gpurun_out/pmc_attn/*/*counter_collection.csv") + glob.glob("$R/gpurun_out/pmc_attn/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "attn_pipe_kernel" in r["Kernel_Name"] or "attn32_kernel" in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (v, n) in sorted(agg.items()):
    print(f"{k:32s} {v/n:16.0f}  (n={n})")
PY
