"""micro-benchmark of one igemm shape: python tools/bench_igemm.py B H W C N KH [up] [reps]   (BENCH_ACT=0|1|2|3: epilogue activation,
2 = GEGLU halves the output width; SR_IGEMM_TILE / SR_IGEMM_SPLIT force a tile form)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stable_renderer_amd import ops as O
B, H, W, C, N, KH = [int(x) for x in sys.argv[1:7]]
up = int(sys.argv[7]) if len(sys.argv) > 7 else 0
reps = int(sys.argv[8]) if len(sys.argv) > 8 else 20
dt = torch.float16
act = int(os.environ.get("BENCH_ACT", "0"))
x = torch.randn(B, H, W, C, dtype=dt, device="cuda")
w = O.pack_conv_weight(torch.randn(N, C, KH, KH) * (C * KH * KH) ** -0.5, dt).cuda()
Ho, Wo = (2 * H, 2 * W) if up else (H, W)
out = torch.empty(B * Ho * Wo, N // 2 if act == 2 else N, dtype=dt, device="cuda")
bias = torch.zeros(N, device="cuda")
res = torch.randn_like(out) if os.environ.get("BENCH_RES") == "1" else None
for _ in range(3):
    O.igemm(x, w, out, B, H, W, C, N, KH=KH, upsample=up, bias=bias, act=act, residual=res)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    O.igemm(x, w, out, B, H, W, C, N, KH=KH, upsample=up, bias=bias, act=act, residual=res)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
fl = 2.0 * B * Ho * Wo * N * KH * KH * C
print(f"B{B} {H}x{W} C{C} N{N} k{KH} up{up} act{act} res{int(res is not None)}: {ms*1e3:.1f} us  {fl/ms/1e9:.1f} TF/s")
