"""ControlNet encoder on the HIP launch plan: ``ControlNet.get_control`` (comfyUI/comfy/controlnet.py:180-214) ->
``cldm.ControlNet.forward`` (comfy/cldm/cldm.py:284-311) -> ``control_merge`` (controlnet.py:95-141), consumed by
``apply_control`` in the UNet (openaimodel.py:374-386).  Same kernels and block lowering as the UNet encoder; the 8-conv hint
encoder (16/32/96/256 channels) runs with channels zero-padded to the GEMM K-step."""
import torch

from . import ops as O
from .plan import PlanBuilder
from .unet import BlockLowering, SD15_CFG, _cdiv, pack_weights

HINT_CONVS = [(0, 1), (2, 1), (4, 2), (6, 1), (8, 2), (10, 1), (12, 2), (14, 1)]      # (index in input_hint_block, stride)


class ControlNet:
    def __init__(self, state_dict, cfg=None, dtype=torch.float16, device="cuda", strength=1.0):
        self.cfg = dict(SD15_CFG if cfg is None else cfg)
        self.dtype, self.device, self.strength = dtype, torch.device(device), float(strength)
        # ControlBase.timestep_percent_range (comfy/controlnet.py:41-62, set by ControlNetApplyAdvanced): outside the sigma window
        # [percent_to_sigma(end), percent_to_sigma(start)] get_control returns only the previous nets' residuals (:184-189)
        self.timestep_percent_range = (0.0, 1.0)
        self.ke = O.kelems(dtype)
        hint_layers = tuple(f"input_hint_block.{i}" for i, _ in HINT_CONVS)
        self.w, self.shapes = pack_weights(state_dict, dtype, self.device, pad_cin=("input_blocks.0.0",), pad_cout=hint_layers[:-1])
        # the last hint conv (256 -> model_channels) keeps its true N but its input may be padded
        if hint_layers[-1] + ".b" in self.w:
            w, _ = pack_weights({hint_layers[-1] + ".weight": state_dict[hint_layers[-1] + ".weight"],
                                 hint_layers[-1] + ".bias": state_dict[hint_layers[-1] + ".bias"]}, dtype, self.device,
                                pad_cin=(hint_layers[-1],))
            self.w.update(w)

    def build(self, B, h, w, x_in, t_in, ctx, n_ctx=77):
        """x_in (B,4,h,w) fp32 / t_in (B,) / ctx (B,n_ctx,C): the UNet plan's own input buffers (same xc, timestep and
        context, controlnet.py:205-212).  -> dict(prologue, step, hint=(B,3,8h,8w) fp32 buffer, output=[12], middle)"""
        cfg, dt, dev, W = self.cfg, self.dtype, self.device, self.w
        pb, pro = PlanBuilder(dev, dt), PlanBuilder(dev, dt)
        mc = cfg["model_channels"]
        ke = self.ke
        hint = pb.buf(B, 3, 8 * h, 8 * w, dtype=torch.float32, zero=True)
        # time embedding (own weights)
        temb = pb.buf(B, mc)
        pb.timestep_embedding(t_in, temb, B, mc)
        e1 = pb.buf(B, 4 * mc)
        pb.igemm(temb, W["time_embed.0"], e1, B, 1, 1, mc, 4 * mc, bias=W["time_embed.0.b"], act=1)
        e2 = pb.buf(B, 4 * mc)
        pb.igemm(e1, W["time_embed.2"], e2, B, 1, 1, 4 * mc, 4 * mc, bias=W["time_embed.2.b"])
        emb_s = pb.buf(B, 4 * mc)
        pb.silu(e2, emb_s)
        low = BlockLowering(pb, pro, W, self.shapes, B, cfg, emb_s, ctx, n_ctx)
        # ---- hint encoder: conv3x3 (+SiLU) x7, zero conv -> guided hint at latent resolution; depends only on the hint
        hh, ww = 8 * h, 8 * w
        cpad = _cdiv(3, ke) * ke
        cur = pro.buf(B, hh * ww, cpad)
        pro.nchw_to_nhwc(hint, cur, B, 3, hh * ww, cpad)
        cin = cpad
        for li, (idx, stride) in enumerate(HINT_CONVS):
            name = f"input_hint_block.{idx}"
            last = li == len(HINT_CONVS) - 1
            cout_true = self.shapes[name][0]
            cout = cout_true if last else _cdiv(cout_true, ke) * ke
            ho, wo = (hh + stride - 1) // stride, (ww + stride - 1) // stride
            nxt = pro.buf(B, ho * wo, cout)
            pro.igemm(cur, W[name], nxt, B, hh, ww, cin, cout, KH=3, stride=stride, bias=W[name + ".b"], act=0 if last else 1)
            cur, cin, hh, ww = nxt, cout, ho, wo
        guided = cur
        assert (hh, ww) == (h, w)
        # ---- encoder copy with zero convs
        outs = []

        def zero_conv(i, x, C, HW, hh_, ww_):
            o = pb.buf(B, HW, C)
            pb.igemm(x, W[f"zero_convs.{i}.0"], o, B, hh_, ww_, C, C, bias=W[f"zero_convs.{i}.0.b"], scale=1.0)
            if self.strength != 1.0:
                o2 = pb.buf(B, HW, C)
                z = pb.buf(B, HW, C, zero=True)
                pb.add(z, o, o2, s=self.strength)
                o = o2
            return o
        cin_pad = _cdiv(cfg["in_channels"], ke) * ke
        xh = pb.buf(B, h * w, cin_pad)
        pb.nchw_to_nhwc(x_in, xh, B, cfg["in_channels"], h * w, cin_pad)
        c0 = pb.buf(B, h * w, mc)
        pb.igemm(xh, W["input_blocks.0.0"], c0, B, h, w, cin_pad, mc, KH=3, bias=W["input_blocks.0.0.b"])
        cur = pb.buf(B, h * w, mc)
        pb.add(c0, guided, cur)                       # h += guided_hint after the first block (cldm.py:297-300)
        ch, hh, ww = mc, h, w
        outs.append(zero_conv(0, cur, ch, hh * ww, hh, ww))
        td = list(cfg["transformer_depth"])
        nlev = len(cfg["channel_mult"])
        bi = 1
        for lev in range(nlev):
            cout = mc * cfg["channel_mult"][lev]
            for _ in range(cfg["num_res_blocks"][lev]):
                cur = low.resblock(f"input_blocks.{bi}.0", cur, ch, None, 0, cout, hh * ww, hh, ww)
                ch = cout
                depth = td.pop(0)
                if depth > 0:
                    cur = low.stransformer(f"input_blocks.{bi}.1", cur, ch, hh * ww, hh, ww, depth)
                outs.append(zero_conv(bi, cur, ch, hh * ww, hh, ww))
                bi += 1
            if lev != nlev - 1:
                ho, wo = (hh + 1) // 2, (ww + 1) // 2
                dn = pb.buf(B, ho * wo, ch)
                pb.igemm(cur, W[f"input_blocks.{bi}.0.op"], dn, B, hh, ww, ch, ch, KH=3, stride=2, bias=W[f"input_blocks.{bi}.0.op.b"])
                cur, hh, ww = dn, ho, wo
                outs.append(zero_conv(bi, cur, ch, hh * ww, hh, ww))
                bi += 1
        cur = low.resblock("middle_block.0", cur, ch, None, 0, ch, hh * ww, hh, ww)
        cur = low.stransformer("middle_block.1", cur, ch, hh * ww, hh, ww, cfg["transformer_depth_middle"])
        cur = low.resblock("middle_block.2", cur, ch, None, 0, ch, hh * ww, hh, ww)
        mid = pb.buf(B, hh * ww, ch)
        pb.igemm(cur, W["middle_block_out.0"], mid, B, hh, ww, ch, ch, bias=W["middle_block_out.0.b"])
        if self.strength != 1.0:
            m2, z = pb.buf(B, hh * ww, ch), pb.buf(B, hh * ww, ch, zero=True)
            pb.add(z, mid, m2, s=self.strength)
            mid = m2
        return dict(prologue=pro.take(), step=pb.take(), hint=hint, output=outs, middle=mid)
