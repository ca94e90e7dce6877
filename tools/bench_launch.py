"""launch-gap probe (development tool): N trivial dependent kernels in one plan, eager vs hipGraph"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stable_renderer_amd.plan import PlanBuilder

dev = torch.device("cuda")
for n_el in (256, 1 << 20):
    pb = PlanBuilder(dev, torch.float16)
    a = pb.buf(n_el, zero=True); b = pb.buf(n_el, zero=True)
    N = 200
    for i in range(N):
        pb.silu(a if i % 2 == 0 else b, b if i % 2 == 0 else a)
    plan = pb.take()
    for mode in ("eager", "graph"):
        if mode == "graph":
            st = torch.cuda.Stream(); plan.capture(st); torch.cuda.synchronize()
        for _ in range(3):
            plan.launch() if mode == "graph" else plan.run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            plan.launch() if mode == "graph" else plan.run()
        e1.record(); torch.cuda.synchronize()
        print(f"n_el={n_el} {mode}: {e0.elapsed_time(e1) / 5 / N * 1e3:.2f} us per kernel")
