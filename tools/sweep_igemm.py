"""tile / K sweep of igemm shapes (development tool): python tools/sweep_igemm.py
each line: M N K kh -> us and TF/s for the auto dispatch and every forced tile (SR_IGEMM_TILE is read per call)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stable_renderer_amd import ops as O

def run(B, H, W, C, N, KH, act=0, reps=20):
    dt = torch.float16
    x = torch.randn(B, H, W, C, dtype=dt, device="cuda")
    w = O.pack_conv_weight(torch.randn(N, C, KH, KH) * (C * KH * KH) ** -0.5, dt).cuda()
    out = torch.empty(B * H * W, N // 2 if act == 2 else N, dtype=dt, device="cuda")
    bias = torch.zeros(N, device="cuda")
    for _ in range(3):
        O.igemm(x, w, out, B, H, W, C, N, KH=KH, bias=bias, act=act)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        O.igemm(x, w, out, B, H, W, C, N, KH=KH, bias=bias, act=act)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return ms * 1e3, 2.0 * B * H * W * N * KH * KH * C / ms / 1e9

TILES = tuple(int(v) for v in os.environ.get("SWEEP_TILES", "0,1,2,3,4").split(","))
shapes = [
    (65536, 1, 1, 64, 320, 1, 0), (65536, 1, 1, 320, 320, 1, 0), (65536, 1, 1, 640, 320, 1, 0), (65536, 1, 1, 1280, 320, 1, 0),
    (16384, 1, 1, 640, 640, 1, 0), (4096, 1, 1, 1280, 1280, 1, 0),
    (65536, 1, 1, 320, 2560, 1, 2), (16384, 1, 1, 640, 5120, 1, 2), (4096, 1, 1, 1280, 10240, 1, 2),
    (65536, 1, 1, 320, 2560, 1, 0),
    (16, 8, 8, 1280, 1280, 3, 0), (16, 16, 16, 1280, 1280, 3, 0), (16, 32, 32, 640, 640, 3, 0), (16, 64, 64, 320, 320, 3, 0),
]
if len(sys.argv) > 1 and sys.argv[1] not in ("splitk", "t5", "t78", "vae", "splits", "conv3p"):
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
if len(sys.argv) > 1 and sys.argv[1] == "t5":
    TILES = (0, 1, 2, 3, 5, 7)
    shapes = [(65536, 1, 1, 320, 320, 1, 0), (65536, 1, 1, 1280, 320, 1, 0), (16, 64, 64, 320, 320, 3, 0), (16, 64, 64, 640, 320, 3, 0),
              (65536, 1, 1, 320, 2560, 1, 2), (16384, 1, 1, 640, 5120, 1, 2), (4096, 1, 1, 1280, 10240, 1, 2),
              (16384, 1, 1, 640, 640, 1, 0), (16384, 1, 1, 2560, 640, 1, 0), (16, 32, 32, 640, 640, 3, 0), (16, 32, 32, 1280, 640, 3, 0),
              (4096, 1, 1, 1280, 1280, 1, 0), (4096, 1, 1, 5120, 1280, 1, 0), (16, 16, 16, 1280, 1280, 3, 0),
              (16, 64, 64, 1280, 1280, 3, 0), (16, 64, 64, 640, 640, 3, 0), (16, 16, 16, 2560, 1280, 3, 0), (16, 8, 8, 1280, 1280, 3, 0),
              (1024, 1, 1, 1280, 1280, 1, 0), (1024, 1, 1, 1280, 10240, 1, 2), (16, 32, 32, 1920, 640, 3, 0)]
if len(sys.argv) > 1 and sys.argv[1] == "t78":
    TILES = (0, 2, 3, 7, 8)
    shapes = [(16384, 1, 1, 640, 640, 1, 0), (4096, 1, 1, 1280, 1280, 1, 0), (4096, 1, 1, 5120, 1280, 1, 0), (16384, 1, 1, 2560, 640, 1, 0),
              (16, 32, 32, 640, 640, 3, 0), (16, 32, 32, 1280, 640, 3, 0), (16, 16, 16, 1280, 1280, 3, 0), (16, 8, 8, 1280, 1280, 3, 0),
              (65536, 1, 1, 320, 320, 1, 0), (16384, 1, 1, 640, 5120, 1, 2), (4096, 1, 1, 1280, 10240, 1, 2)]
if len(sys.argv) > 1 and sys.argv[1] == "conv3p":          # patch-stationary 3x3 (tile 8) against the re-staging tiles
    TILES = (5, 8, 6, 7, 1, 8, 5)
    shapes = [(16, 64, 64, 320, 320, 3, 0), (16, 64, 64, 640, 320, 3, 0), (16, 64, 64, 960, 320, 3, 0), (16, 32, 32, 320, 640, 3, 0),
              (16, 32, 32, 640, 640, 3, 0), (16, 32, 32, 1280, 640, 3, 0), (16, 32, 32, 960, 640, 3, 0), (16, 16, 16, 640, 1280, 3, 0),
              (16, 16, 16, 1280, 1280, 3, 0), (16, 16, 16, 2560, 1280, 3, 0), (16, 8, 8, 1280, 1280, 3, 0), (16, 8, 8, 2560, 1280, 3, 0),
              (16, 64, 64, 1280, 1280, 3, 0)]
if len(sys.argv) > 1 and sys.argv[1] == "vae":
    TILES = (0, 1, 2, 8)
    shapes = [(8, 512, 512, 128, 128, 3, 0), (8, 256, 256, 256, 256, 3, 0), (8, 128, 128, 512, 512, 3, 0), (8, 512, 512, 256, 128, 3, 0),
              (16, 64, 64, 1280, 1280, 3, 0), (65536, 1, 1, 320, 2560, 1, 2)]
if len(sys.argv) > 1 and sys.argv[1] == "splits":
    os.environ.pop("SR_IGEMM_TILE", None)
    for sh in [(16, 8, 8, 1280, 1280, 3, 0), (16, 8, 8, 2560, 1280, 3, 0), (16, 16, 16, 1280, 1280, 3, 0), (16, 16, 16, 2560, 1280, 3, 0),
               (16, 16, 16, 1920, 1280, 3, 0), (16, 16, 16, 640, 1280, 3, 0)]:
        row = []
        for S in (0, 1, 2, 3, 4, 5, 6, 8, 10, 12):
            if S:
                os.environ["SR_SPLIT_S"] = str(S)
            else:
                os.environ.pop("SR_SPLIT_S", None)
            best = 1e9
            for _ in range(2):
                us, tf = run(*sh)
                best = min(best, us)
            row.append(f"S{S}:{best:6.1f}")
        print("B%d %dx%d C%d N%d k%d act%d | " % sh + " ".join(row), flush=True)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "splitk":
    os.environ.pop("SR_IGEMM_TILE", None)
    for sh in [(16, 8, 8, 1280, 1280, 3, 0), (16, 8, 8, 2560, 1280, 3, 0), (16, 16, 16, 1280, 1280, 3, 0), (16, 16, 16, 2560, 1280, 3, 0),
               (16, 16, 16, 1920, 1280, 3, 0), (16, 32, 32, 640, 640, 3, 0), (16, 32, 32, 1280, 640, 3, 0), (16, 32, 32, 1920, 640, 3, 0)]:
        row = []
        for target in (0, 384, 512, 768, 1024, 1536, 2048):
            os.environ["SR_SPLITK"] = str(target)
            us, tf = run(*sh)
            row.append(f"s{target}:{us:6.1f}us {tf:5.0f}TF")
        print("B%d %dx%d C%d N%d k%d act%d | " % sh + " | ".join(row), flush=True)
    sys.exit(0)
for sh in shapes:
    row = []
    for tile in TILES:
        if tile:
            os.environ["SR_IGEMM_TILE"] = str(tile)
        else:
            os.environ.pop("SR_IGEMM_TILE", None)
        try:
            us, tf = run(*sh)
            row.append(f"t{tile}:{us:7.1f}us {tf:6.0f}TF")
        except Exception:
            row.append(f"t{tile}:     n/a         ")
    print("B%d %dx%d C%d N%d k%d act%d | " % sh + " | ".join(row), flush=True)
