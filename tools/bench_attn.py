"""micro-benchmark of sr_attention: python tools/bench_attn.py [B Tq Tk heads d Bk] (development tool)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stable_renderer_amd import ops as O

def run(B, Tq, Tk, heads, d, Bk, reps=20):
    dt = torch.float16
    Cc = heads * d
    q = torch.randn(B, Tq, Cc, dtype=dt, device="cuda")
    k = torch.randn(Bk, Tk, Cc, dtype=dt, device="cuda")
    ldt = (Tk + 7) // 8 * 8
    vt = torch.randn(Bk, Cc, ldt, dtype=dt, device="cuda")
    vt = vt.view(Bk, heads, d, ldt)
    for _ in range(3):
        O.attention(q, k, vt, heads, Tk=Tk)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        O.attention(q, k, vt, heads, Tk=Tk)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return ms * 1e3, 4.0 * B * heads * Tq * Tk * d / ms / 1e9

if __name__ == "__main__":
    shapes = [(16, 4096, 4096, 8, 40, 1), (16, 1024, 1024, 8, 80, 1), (16, 4096, 4096, 8, 40, 16), (16, 4096, 77, 8, 40, 16), (16, 1024, 77, 8, 80, 16), (16, 256, 256, 8, 160, 1)]
    if len(sys.argv) > 6:
        shapes = [tuple(int(v) for v in sys.argv[1:7])]
    for sh in shapes:
        row = []
        us, tf = run(*sh)
        row.append(f"{us:7.1f}us {tf:6.0f}TF")
        print("B%d Tq%d Tk%d h%d d%d Bk%d | " % sh + " | ".join(row), flush=True)
