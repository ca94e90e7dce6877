"""LayerNorm micro-benchmark (development tool): python tools/bench_ln.py [rows C]..."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stable_renderer_amd import ops as O
shapes = [(65536, 320), (16384, 640), (4096, 1280), (1024, 1280)]
if len(sys.argv) > 2:
    shapes = [(int(sys.argv[1]), int(sys.argv[2]))]
for rows, C in shapes:
    x = torch.randn(rows, C, dtype=torch.float16, device="cuda")
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    for _ in range(3):
        O.layernorm(x, g, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        O.layernorm(x, g, b)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"rows{rows} C{C}: {us:.1f} us  {rows * C * 4 / us / 1e6:.2f} TB/s")
