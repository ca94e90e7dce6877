"""Per-op timing of a launch plan on the GPU (development tool): [SR_VIEWS=n] python tools/profile_plan.py [unet|vae] [f16|f32] [seq]"""
import ctypes as C
import os
import sys
from collections import defaultdict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stable_renderer_amd import _lib as L  # noqa: E402
from stable_renderer_amd import ops as O  # noqa: E402
from stable_renderer_amd.pipeline import build_sd15_pipeline  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "unet"
dtype = torch.float16 if (len(sys.argv) < 3 or sys.argv[2] == "f16") else torch.float32
VIEWS = int(os.environ.get("SR_VIEWS", "8"))
_tt = os.path.join(ROOT, "tests", "golden", "tune_table.json")
if os.path.exists(_tt) and os.environ.get("SR_BENCH_TUNE", "pinned") == "pinned":
    O.load_tune_table(_tt)                                   # the table bench.py runs on
pipe = build_sd15_pipeline(dtype=dtype, use_graph=False, n_views=VIEWS)
if which == "unet":
    p = pipe.runner._ensure_plan([min(3, 2 * VIEWS - 1)])
    pipe.runner._load_ctx(p)
    plan = p["step"]
else:
    plan = pipe.vplan["plan"]
lib = L.lib()
names = {1: "igemm", 2: "groupnorm", 3: "layernorm", 4: "attention", 5: "nchw2nhwc", 6: "nhwc2nchw", 7: "temb", 8: "silu", 9: "softmax"}
agg = defaultdict(lambda: [0.0, 0, 0.0])
plan.run()
torch.cuda.synchronize()
reps = 3
SEQ = "seq" in sys.argv          # in-sequence: the whole plan back to back, one event between ops (cold-ish caches, real order)
seq_us = None
if SEQ:
    seq_us = [0.0] * plan.n
    for _ in range(reps):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(plan.n + 1)]
        ones = []
        for i in range(plan.n):
            o = (L.Op * 1)(plan.ops[i])
            o[0].lane = 0                                       # per-op timing: everything on the measured stream
            ones.append(o)
        torch.cuda.synchronize()
        evs[0].record()
        for i in range(plan.n):
            if plan.ops[i].kind not in (L.OP_FORK, L.OP_JOIN):
                lib.sr_plan_run(ones[i], 1, O.stream_ptr())
            evs[i + 1].record()
        torch.cuda.synchronize()
        for i in range(plan.n):
            seq_us[i] += evs[i].elapsed_time(evs[i + 1]) * 1e3 / reps
for i in range(plan.n):
    op = plan.ops[i]
    if SEQ:
        us = seq_us[i]
    else:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        one = (L.Op * 1)(op)
        one[0].lane = 0
        if op.kind in (L.OP_FORK, L.OP_JOIN):
            continue
        lib.sr_plan_run(one, 1, O.stream_ptr())
        e0.record()
        for _ in range(reps):
            lib.sr_plan_run(one, 1, O.stream_ptr())
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
    k = op.kind
    if k == 1:
        a = op.u.igemm
        sig = f"igemm B{a.B} {a.H}x{a.W} C{a.C1}+{a.C2} N{a.N} k{a.KH} s{a.stride} u{a.upsample} act{a.act} t{a.transpose_out}"
    elif k == 2:
        a = op.u.gn
        sig = f"groupnorm B{a.B} HW{a.HW} C{a.C1}+{a.C2} silu{a.silu}"
    elif k == 4:
        a = op.u.attn
        sig = f"attention B{a.B} Bk{a.Bk} Tq{a.Tq} Tk{a.Tk} h{a.heads} d{a.d}"
    elif k == 3:
        a = op.u.ln
        sig = f"layernorm rows{a.rows} C{a.C}"
    else:
        sig = names.get(k, str(k))
    agg[sig][0] += us
    agg[sig][1] += 1
    agg[sig][2] += plan.op_flops[i]
tot = sum(v[0] for v in agg.values())
print(f"total {tot/1e3:.2f} ms over {plan.n} ops (serialised per-op sum)")
# whole plan as it really runs (both lanes), eager
for _ in range(2):
    plan.run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    plan.run()
e1.record()
torch.cuda.synchronize()
print(f"whole plan, two lanes, eager: {e0.elapsed_time(e1) / 5:.2f} ms")
st = torch.cuda.Stream()
plan.capture(st)
torch.cuda.synchronize()
with torch.cuda.stream(st):
    for _ in range(2):
        plan.launch()
    e0.record()
    for _ in range(5):
        plan.launch()
    e1.record()
torch.cuda.synchronize()
print(f"whole plan, hipGraph: {e0.elapsed_time(e1) / 5:.2f} ms")
for sig, (us, n, fl) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:70]:
    tf = fl / (us * 1e-6) / 1e12 if us > 0 else 0
    print(f"{us/1e3:8.3f} ms {100*us/tot:5.1f}%  n={n:3d} avg {us/n:8.1f} us {tf:7.1f} TF/s  {sig}")
bykind = defaultdict(float)
for sig, (us, n, fl) in agg.items():
    bykind[sig.split()[0]] += us
print({k: round(v / 1e3, 2) for k, v in bykind.items()})
