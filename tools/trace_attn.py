"""Where a wave of the d = 40 attention kernel spends its iterations (development tool; needs a library built with -DSR_ATTN_TRACE=1,
loaded through SR_DEV_LIB): python tools/trace_attn.py
parts of an iteration (shader clock, summed over the tiles of one wave): hand-off (barrier + LDS store + next global load),
phase A (QK^T of tile t+1 interleaved with exp / pack of tile t), phase B (PV of tile t interleaved with the row max of tile t+1)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stable_renderer_amd import ops as O
B, Tq, Tk, heads, d = 16, 4096, 4096, 8, 40
Cc = heads * d
dt = torch.float16
q = torch.randn(B, Tq, Cc, dtype=dt, device="cuda")
kbig = torch.randn(1, Tk + 64, Cc, dtype=dt, device="cuda")        # 64 spare rows behind K: the kernel's trace area
k = kbig[:, :Tk]
ldt = (Tk + 7) // 8 * 8
vt = torch.randn(1, Cc, ldt, dtype=dt, device="cuda").view(1, heads, d, ldt)
for _ in range(3):
    O.attention(q, k, vt, heads, Tk=Tk)
torch.cuda.synchronize()
a32 = os.environ.get("SR_ATTN_32", "0") != "0"
ncol = 5 if a32 else 4
raw = kbig[0, Tk:].contiguous().view(torch.int64)
tr = raw[: (raw.numel() // ncol) * ncol].reshape(-1, ncol).cpu().numpy()
tr = tr[tr[:, ncol - 1] == (Tk + 63) // 64]
nt = tr[0, ncol - 1]
print("waves traced: %d, tiles per wave: %d" % (len(tr), nt))
names = ["DMA issue + K reads + QK^T", "row max / lazy shift", "exp2 / pack / PV", "vmcnt wait + barrier"] if a32 else ["hand-off", "phase A (QK^T | exp)", "phase B (PV | row max)"]
for i, nme in enumerate(names):
    print("  %-28s %7.0f cycles per tile (min %6.0f max %6.0f over waves)" % (nme, tr[:, i].mean() / nt, tr[:, i].min() / nt, tr[:, i].max() / nt))
print("  total                        %7.0f cycles per tile" % (tr[:, :ncol - 1].sum(1).mean() / nt))
