"""development aid: the small-grid igemm shapes of a one-view (B = 2) UNet evaluation under every candidate (tile, split) --
back to back (warm caches) and `cold` (a 600 MB buffer rewritten between launches, so weights and activations come from HBM as
inside a plan): python tools/bench_small.py [cold]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stable_renderer_amd import ops as O
import ctypes as C
from stable_renderer_amd import _lib as L
cold = "cold" in sys.argv
dt = torch.float16
shapes = [(512, 1, 1, 1280, 1280, 1), (2048, 1, 1, 640, 640, 1), (8192, 1, 1, 320, 320, 1), (128, 1, 1, 1280, 1280, 1),
          (2, 8, 8, 1280, 1280, 3), (2, 16, 16, 1280, 1280, 3), (2, 32, 32, 640, 640, 3), (2, 64, 64, 320, 320, 3),
          (2048, 1, 1, 2560, 640, 1), (512, 1, 1, 5120, 1280, 1), (4096, 1, 1, 1280, 1280, 1), (16, 16, 16, 1280, 1280, 3), (16, 8, 8, 1280, 1280, 3)]
cands = [(0, 0), (2, -1), (2, 0), (2, 2), (2, 4), (2, 8), (3, -1), (3, 0), (3, 4), (3, 8), (4, -1), (13, -1), (14, -1), (14, 2), (14, 3), (14, 4), (14, 5), (14, 8), (14, 12), (14, 16),
         (15, -1), (15, 2), (15, 4), (15, 8), (15, 12)]
flush = torch.empty(600 << 20, dtype=torch.uint8, device="cuda") if cold else None
lib = L.lib()
for (B, H, W, Cc, N, KH) in shapes:
    x = torch.randn(B, H, W, Cc, dtype=dt, device="cuda")
    w = O.pack_conv_weight(torch.randn(N, Cc, KH, KH) * (Cc * KH * KH) ** -0.5, dt).cuda()
    out = torch.empty(B * H * W, N, dtype=dt, device="cuda")
    bias = torch.zeros(N, device="cuda")
    res = torch.randn_like(out)
    row = []
    for tile, split in cands:
        ar = O.igemm_args(x, w, out, B, H, W, Cc, N, KH=KH, bias=bias, residual=res, tile=tile, split=split)
        if lib.sr_igemm(C.byref(ar), O.stream_ptr()) != 0:
            continue
        torch.cuda.synchronize()
        reps = 10 if cold else 40
        tot = 0.0
        if cold:
            for _ in range(reps):
                flush.add_(1)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                lib.sr_igemm(C.byref(ar), O.stream_ptr())
                e1.record()
                torch.cuda.synchronize()
                tot += e0.elapsed_time(e1)
        else:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                lib.sr_igemm(C.byref(ar), O.stream_ptr())
            e1.record()
            torch.cuda.synchronize()
            tot = e0.elapsed_time(e1)
        row.append((tot / reps * 1e3, tile, split))
    best = min(row)
    print(f"B{B} {H}x{W} C{Cc} N{N} k{KH}: " + "  ".join(f"{t}/{s}:{us:.1f}" for us, t, s in row) + f"   best {best[1]}/{best[2]} {best[0]:.1f} us", flush=True)
