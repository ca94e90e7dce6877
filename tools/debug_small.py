"""diagnostic: run the tiny fp32 UNet plan at an 8x8 latent op by op with a sync after each launch"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stable_renderer_amd import _lib as L, ops as O, synth
from stable_renderer_amd.model_shapes import unet_names_shapes
from stable_renderer_amd.unet import UNet, SD15_CFG
cfg = dict(SD15_CFG, model_channels=64, context_dim=64)
ns, norms = unet_names_shapes(cfg)
net = UNet(synth.synth_state_dict(ns, seed=1, norm_names=norms), cfg, dtype=torch.float32)
hw = int(sys.argv[1]) if len(sys.argv) > 1 else 8
p = net.build(8, hw, hw, inject_idx=[5])
p["x"].normal_(); p["t"].fill_(500.0); p["ctx"].normal_()
lib = L.lib()
for name in ("prologue", "step"):
    plan = p[name]
    for i in range(plan.n):
        op = plan.ops[i]
        one = (L.Op * 1)(op)
        desc = ""
        if op.kind == 1:
            a = op.u.igemm; desc = f"igemm B{a.B} {a.H}x{a.W} C{a.C1}+{a.C2} N{a.N} k{a.KH} s{a.stride} u{a.upsample} t{a.transpose_out} ldt{a.ldt} f32{a.out_f32}"
        elif op.kind == 4:
            a = op.u.attn; desc = f"attn B{a.B} Bk{a.Bk} Tq{a.Tq} Tk{a.Tk} d{a.d} ldt{a.ldt}"
        elif op.kind == 2:
            a = op.u.gn; desc = f"gn B{a.B} HW{a.HW} C{a.C1}+{a.C2}"
        print(name, i, op.kind, desc, flush=True)
        rc = lib.sr_plan_run(one, 1, O.stream_ptr())
        assert rc == 0, lib.sr_last_error()
        torch.cuda.synchronize()
print("done, out finite:", bool(torch.isfinite(p["out"]).all()), flush=True)
