"""profiles helper: assemble profiles/rNN_small_batch_profile.txt from the outputs of tools/profile_small.sh (per-op in-sequence tables,
rocprofv3 kernel stats of `bench.py --roofline-only --views V`) and tools/trace_small.sh (workgroup timelines).
python tools/make_small_profile.py gpurun_out/small_r04 [gpurun_out/trace_small.txt] > profiles/r04_small_batch_profile.txt"""
import csv
import json
import os
import sys

d = sys.argv[1]
print("# Small-batch UNet evaluation: what a one-view rank of the 8-GPU shard (B = 2) and a three-frame rank of BASELINE config 4 (B = 6)")
print("# execute.  tools/profile_small.sh: (a) tools/profile_plan.py unet f16 seq -- every op of the step plan timed in sequence with an event")
print("# between ops (event overhead included: the serialised sum reads ~15 % above the plan), the plan eagerly and as a hipGraph;")
print("# (b) rocprofv3 --kernel-trace --stats of `bench.py --roofline-only --views V`: 1 run of the whole plan + 7 replays of the igemm ops + 5 of")
print("# the whole plan (6 whole evaluations + 7 igemm-only ones in the file).")
for v in (1, 3):
    print("\n" + "=" * 120)
    print(f"== views {v} (B = {2 * v})")
    with open(os.path.join(d, f"roofline_only_views{v}.json")) as f:
        r = json.loads(f.read().strip().splitlines()[-1])["roofline"]
    print(f"bench.py --roofline-only --views {v} (under the profiler): UNet evaluation {r['unet_eval_ms']} ms over {r.get('unet_eval_launches')} ops, "
          f"igemm family {r['achieved']} TF/s = {r['frac']} of the dense fp16 peak ({r['launches']} igemm ops, {r['avg_launch_us']} us each)")
    rows = list(csv.DictReader(open(os.path.join(d, f"roofline_v{v}", "out_kernel_stats.csv"))))
    ours = [x for x in rows if "at::native" not in x["Name"] and "rocclr" not in x["Name"]]
    fam = [x for x in ours if any(k in x["Name"] for k in ("igemm_kernel", "igemm_group_kernel", "conv3p_kernel", "splitk_reduce"))]
    other = [x for x in ours if x not in fam]
    calls_f, ns_f = sum(int(x["Calls"]) for x in fam), sum(int(x["TotalDurationNs"]) for x in fam)
    calls_o, ns_o = sum(int(x["Calls"]) for x in other), sum(int(x["TotalDurationNs"]) for x in other)
    print(f"kernel launches per evaluation: igemm family {calls_f / 13:.0f} ({ns_f / 13 / 1e6:.3f} ms of kernel time), everything else "
          f"{calls_o / 6:.0f} ({ns_o / 6 / 1e6:.3f} ms) -> {calls_f / 13 + calls_o / 6:.0f} launches, {ns_f / 13 / 1e6 + ns_o / 6 / 1e6:.3f} ms of summed kernel durations")
    print("kernel                                                                                      launches/eval   avg us   ms/eval")
    for x in sorted(ours, key=lambda x: -int(x["TotalDurationNs"]) / (13 if x in fam else 6))[:28]:
        n = 13 if x in fam else 6
        print(f"  {x['Name'][:92]:92s} {int(x['Calls']) / n:8.1f} {float(x['AverageNs']) / 1e3:9.1f} {int(x['TotalDurationNs']) / n / 1e6:8.3f}")
    print(f"\n-- per-op table (tools/profile_plan.py unet f16 seq, SR_VIEWS={v}):")
    with open(os.path.join(d, f"plan_seq_views{v}.txt")) as f:
        txt = [ln.rstrip() for ln in f if "amdgpu.ids" not in ln]
    start = next(i for i, ln in enumerate(txt) if ln.startswith("total "))
    print("\n".join(txt[start:start + 60]))
if len(sys.argv) > 2 and os.path.exists(sys.argv[2]):
    print("\n" + "=" * 120)
    print("== workgroup timelines of the small-grid igemm shapes (tools/trace_small.sh: -DSR_IGEMM_TRACE=1 build; wave 0 of every workgroup stamps")
    print("== the 100 MHz wall clock at kernel entry, first stage issued, first stage landed, K loop done, epilogue issued, stores retired; warm caches)")
    print(open(sys.argv[2]).read())
