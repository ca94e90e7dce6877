"""profiles helper: from a rocprofv3 kernel_stats.csv of `bench.py --roofline-only`, the summed / average duration of the
igemm family (igemm_kernel / igemm_group_kernel instantiations + conv3p_kernel + splitk_reduce_kernel).  python tools/summarize_igemm_trace.py stats.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
fam = [r for r in rows if "igemm_kernel" in r["Name"] or "igemm_group_kernel" in r["Name"] or "splitk_reduce" in r["Name"] or "conv3p_kernel" in r["Name"]]
calls = sum(int(r["Calls"]) for r in fam)
tot = sum(int(r["TotalDurationNs"]) for r in fam)
print(f"igemm family: {calls} kernel launches, {tot/1e6:.3f} ms total, {tot/calls/1e3:.2f} us average")
for r in sorted(fam, key=lambda r: -int(r["TotalDurationNs"])):
    print(f"  {int(r['Calls']):6d} x {float(r['AverageNs'])/1e3:9.2f} us  {r['Name'][:110]}")
