"""Experiment (development tool): how much of the in-sequence time of the UNet plan's igemm ops is cold weight fetch?
Per-op windows (event before / after each op) of the whole plan run in order, (a) as is, (b) with the weights of op i+LEAD
read once (torch sum) after op i's window closes, i.e. resident in the Infinity Cache when their op starts."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stable_renderer_amd import _lib as L  # noqa: E402
from stable_renderer_amd import ops as O  # noqa: E402
from stable_renderer_amd.pipeline import build_sd15_pipeline  # noqa: E402

LEAD = int(sys.argv[1]) if len(sys.argv) > 1 else 1
pipe = build_sd15_pipeline(dtype=torch.float16, use_graph=False)
p = pipe.runner._ensure_plan([3])
pipe.runner._load_ctx(p)
plan = p["step"]
lib = L.lib()
# weight tensors by data pointer
byptr = {}
for t in plan._keep:
    if isinstance(t, torch.Tensor):
        byptr[t.data_ptr()] = t
ops = [plan.ops[i] for i in range(plan.n)]
wt = []
for op in ops:
    w = None
    if op.kind == L.OP_IGEMM:
        w = byptr.get(op.u.igemm.w)
    wt.append(w)
ones = []
for i in range(plan.n):
    o = (L.Op * 1)(plan.ops[i])
    o[0].lane = 0
    ones.append(o)


def run(prefetch):
    tot = [0.0] * plan.n
    reps = 3
    for _ in range(reps):
        ea = [torch.cuda.Event(enable_timing=True) for _ in range(plan.n)]
        eb = [torch.cuda.Event(enable_timing=True) for _ in range(plan.n)]
        torch.cuda.synchronize()
        for i in range(plan.n):
            if ops[i].kind in (L.OP_FORK, L.OP_JOIN):
                continue
            ea[i].record()
            lib.sr_plan_run(ones[i], 1, O.stream_ptr())
            eb[i].record()
            if prefetch:
                j = i + LEAD
                while j < plan.n and wt[j] is None:
                    j += 1
                if j < plan.n:
                    wt[j].view(torch.int16).sum(dtype=torch.int32)
        torch.cuda.synchronize()
        for i in range(plan.n):
            if ops[i].kind not in (L.OP_FORK, L.OP_JOIN):
                tot[i] += ea[i].elapsed_time(eb[i]) * 1e3 / reps
    return tot


plan.run()
torch.cuda.synchronize()
a = run(False)
b = run(True)
a2 = run(False)
ig = [i for i in range(plan.n) if ops[i].kind == L.OP_IGEMM]
print(f"igemm ops in sequence: as is {sum(a[i] for i in ig)/1e3:.2f} ms (repeat {sum(a2[i] for i in ig)/1e3:.2f}); weights pre-read {sum(b[i] for i in ig)/1e3:.2f} ms")
print(f"all ops: {sum(a)/1e3:.2f} -> {sum(b)/1e3:.2f} ms")
rows = []
for i in ig:
    g = ops[i].u.igemm
    rows.append((a[i] - b[i], a[i], b[i], f"B{g.B} {g.H}x{g.W} C{g.C1}+{g.C2} N{g.N} k{g.KH} s{g.stride} u{g.upsample} act{g.act}"))
rows.sort(reverse=True)
for d, x, y, s in rows[:25]:
    print(f"{d:7.1f} us saved  {x:7.1f} -> {y:7.1f}  {s}")
