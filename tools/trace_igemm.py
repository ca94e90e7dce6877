"""Phase timeline of the workgroups of ONE igemm launch (development tool; needs a library built with -DSR_IGEMM_TRACE=1, loaded through
SR_DEV_LIB): python tools/trace_igemm.py B H W C N KH [act]   (SR_IGEMM_TILE picks the tile form)
stamps (100 MHz wall clock): 0 kernel entry, 1 before the first stage is issued, 2 first stage landed (first barrier), 3 K loop done,
4 epilogue instructions done, 5 stores retired"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stable_renderer_amd import ops as O
B, H, W, C, N, KH = [int(x) for x in sys.argv[1:7]]
act = int(sys.argv[7]) if len(sys.argv) > 7 else 0
dt = torch.float16
x = torch.randn(B, H, W, C, dtype=dt, device="cuda")
w = O.pack_conv_weight(torch.randn(N, C, KH, KH) * (C * KH * KH) ** -0.5, dt).cuda()
out = torch.empty(B * H * W, N // 2 if act == 2 else N, dtype=dt, device="cuda")
bias = torch.zeros(N, device="cuda")
ws = O.workspace(x.device)
for _ in range(3):
    O.igemm(x, w, out, B, H, W, C, N, KH=KH, bias=bias, act=act, split=-1)
torch.cuda.synchronize()
ws.zero_()
torch.cuda.synchronize()
O.igemm(x, w, out, B, H, W, C, N, KH=KH, bias=bias, act=act, split=-1)
torch.cuda.synchronize()
t = ws.view(torch.int64)[: 8 * 65536].reshape(-1, 8).cpu().numpy()
t = t[t[:, 0] != 0]
t0 = t[:, 0].min()
rel = (t[:, :6] - t0) * 0.01                                   # us since the first workgroup started
d = np.diff(t[:, :6], axis=1) * 0.01
print("workgroups traced: %d; launch span %.1f us (first entry -> last store retired)" % (len(t), rel[:, 5].max()))
names = ["entry->issue", "issue->landed", "K loop", "epilogue", "store drain"]
for i, nme in enumerate(names):
    print("  %-14s mean %6.2f us   p10 %6.2f   p90 %6.2f" % (nme, d[:, i].mean(), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
print("  workgroup lifetime mean %.2f us; entry times: p10 %.1f p50 %.1f p90 %.1f us" % ((rel[:, 5] - rel[:, 0]).mean(), *np.percentile(rel[:, 0], [10, 50, 90])))
cu = t[:, 6]
per = {}
for i in range(len(t)):
    per.setdefault(int(cu[i]), []).append((rel[i, 0], rel[i, 5]))
k = sorted(per)[0]
print("  CU id %d ran %d workgroups: " % (k, len(per[k])) + " ".join("[%.1f-%.1f]" % ab for ab in sorted(per[k])))
