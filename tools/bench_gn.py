"""GroupNorm micro-benchmark (development tool): python tools/bench_gn.py B HW C"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stable_renderer_amd import ops as O
B, HW, C = [int(v) for v in sys.argv[1:4]]
x = torch.randn(B, HW, C, dtype=torch.float16, device="cuda")
g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
for _ in range(3):
    O.groupnorm(x, g, b, B, HW, C, silu=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    O.groupnorm(x, g, b, B, HW, C, silu=True)
e1.record(); torch.cuda.synchronize()
print(f"B{B} HW{HW} C{C}: {e0.elapsed_time(e1)/20*1e3:.1f} us")
