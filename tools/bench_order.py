"""development aid: sr_igemm_args.tile_order 0 (rows first) against 1 (columns first) on the weight-heavy layer shapes of a B = 16
evaluation, cold (weights flushed out of the Infinity Cache, activations just written), under the pinned tuner table's tile / split.
python tools/bench_order.py"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stable_renderer_amd import _lib as L                                      # noqa: E402
from stable_renderer_amd import ops as O                                      # noqa: E402

O.load_tune_table(os.path.join(ROOT, "tests", "golden", "tune_table.json"))
lib = L.lib()
flush = torch.empty(320 << 20, dtype=torch.uint8, device="cuda")
shapes = [  # B, H, W, C, N, KH, act
    (16, 8, 8, 1280, 1280, 3, 0), (16, 8, 8, 2560, 1280, 3, 0), (16, 16, 16, 1280, 1280, 3, 0), (16, 16, 16, 2560, 1280, 3, 0),
    (16, 16, 16, 1920, 1280, 3, 0), (16, 16, 16, 640, 1280, 3, 0), (4096, 1, 1, 1280, 10240, 1, 2), (4096, 1, 1, 5120, 1280, 1, 0),
    (4096, 1, 1, 1280, 1280, 1, 0), (1024, 1, 1, 1280, 1280, 1, 0), (1024, 1, 1, 1280, 10240, 1, 2), (1024, 1, 1, 5120, 1280, 1, 0),
    (16, 32, 32, 640, 640, 3, 0), (16, 32, 32, 1280, 640, 3, 0), (16384, 1, 1, 640, 5120, 1, 2), (16384, 1, 1, 640, 640, 1, 0),
    (16, 16, 16, 1280, 1280, 3, 0),
]
cands = [tuple(int(v) for v in c.split("/")) for c in os.environ.get("CANDS", "").split(",") if c]
for (B, H, W, Cin, N, KH, act) in shapes:
    x = torch.randn(B, H, W, Cin, dtype=torch.float16, device="cuda")
    w = O.pack_conv_weight(torch.randn(N, Cin, KH, KH) * (Cin * KH * KH) ** -0.5, torch.float16).cuda()
    out = torch.empty(B * H * W, N // 2 if act == 2 else N, dtype=torch.float16, device="cuda")
    bias = torch.zeros(N, device="cuda")
    ar = O.igemm_args(x, w, out, B, H, W, Cin, N, KH=KH, bias=bias, act=act)
    O.tune_igemm(ar)
    st = O.stream_ptr()

    def timed(n=12):
        best = []
        for _ in range(n):
            flush.add_(1)
            lib.sr_cache_touch(ar.a, B * H * W * Cin * 2, st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            assert lib.sr_igemm(C.byref(ar), st) == 0
            e1.record()
            e1.synchronize()
            best.append(e0.elapsed_time(e1) * 1e3)
        best.sort()
        return sum(best[:n // 2]) / (n // 2)
    row = []
    for tile, split in [(ar.tile, ar.split)] + cands:
        ar.tile, ar.split = tile, split
        ar.tile_order = 0
        if lib.sr_igemm(C.byref(ar), st) != 0:
            continue
        t0 = timed()
        ar.tile_order = 1
        ref = out.clone()
        t1 = timed()
        ar.tile_order = 0
        lib.sr_igemm(C.byref(ar), st)
        same = bool((out == ref).all())
        row.append("tile %d/%d: %.1f | %.1f us%s" % (tile, split, t0, t1, "" if same else " DIFFERENT VALUES"))
    fl = 2.0 * B * H * W * N * KH * KH * Cin
    print("B%d %dx%d C%d N%d k%d act%d (W %.1f MB, A %.1f MB)  rows first | columns first:  %s" %
          (B, H, W, Cin, N, KH, act, N * Cin * KH * KH * 2 / 1e6, B * H * W * Cin * 2 / 1e6, "   ".join(row)), flush=True)
