"""SD-family UNet lowered onto the HIP launch plan.

Mirrors ``UNetModel.forward`` (comfyUI/comfy/ldm/modules/diffusionmodules/openaimodel.py:841-946),
``ResBlock._forward`` (:253-281), ``Upsample/Downsample`` (:82-148), ``SpatialTransformer.forward`` and
``BasicTransformerBlock._forward`` (comfy/ldm/modules/attention.py:495-726) including the call site of
``corresponder.pre_atten_inject`` (attention.py:583-589).  The module tree is only walked at plan-build time; the
per-step path is ``Plan.launch()`` (native).  Checkpoint keys are the reference's (``input_blocks.1.0.in_layers.0.weight``
...), so a real SD1.5 state dict loads unchanged.

MI355X layout decisions: activations NHWC so every conv / linear is one K-contiguous implicit GEMM
(``b c h w <-> b (h w) c`` rearranges vanish); GroupNorm+SiLU is one streaming pass that also consumes the decoder's
channel concat without materialising it; time-embedding add, bias, residual and GEGLU are GEMM epilogues; V is
written transposed by its projection so attention needs no transpose; cross-attention K/V depend only on the prompt and
are projected once per sampling run (prologue plan), not once per step.
"""
import os

import torch

from . import ops as O
from .plan import PlanBuilder

SD15_CFG = dict(in_channels=4, out_channels=4, model_channels=320, num_res_blocks=[2, 2, 2, 2],
                channel_mult=[1, 2, 4, 4], transformer_depth=[1, 1, 1, 1, 1, 1, 0, 0], transformer_depth_middle=1,
                transformer_depth_output=[1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0], context_dim=768, num_heads=8)


# comfy/supported_models.py:153-160 (SDXL base): no attention at the first level, 2 / 10 transformer blocks below, 64-wide heads,
# linear proj_in / proj_out, 2816-wide vector conditioning through label_emb, 2048-wide context
SDXL_CFG = dict(in_channels=4, out_channels=4, model_channels=320, num_res_blocks=[2, 2, 2], channel_mult=[1, 2, 4],
                transformer_depth=[0, 0, 2, 2, 10, 10], transformer_depth_middle=10,
                transformer_depth_output=[0, 0, 0, 2, 2, 2, 10, 10, 10], context_dim=2048, num_heads=-1, num_head_channels=64,
                use_linear_in_transformer=True, adm_in_channels=2816)


def _cdiv(a, b):
    return (a + b - 1) // b


def pack_weights(sd, dtype, device, pad_cin=(), pad_cout=()):
    """checkpoint tensors -> packed device tensors.  pad_cin: layer names whose input channels are zero-padded to the
    K-step (conv_in); pad_cout: layers whose OUTPUT channels are padded too (ControlNet hint encoder, 16/32/96 channels),
    so that the next layer sees a K-step-aligned channel count filled with exact zeros."""
    ke = O.kelems(dtype)
    w = {}
    for k, v in sd.items():
        if not k.endswith(".weight"):
            continue
        base = k[:-7]
        b = sd.get(base + ".bias")
        if v.dim() >= 2:
            geglu = base.endswith("ff.net.0.proj")
            cin_pad = _cdiv(v.shape[1], ke) * ke if (base in pad_cin or base in pad_cout) else None
            if base in pad_cout and v.shape[0] % ke:
                npad = _cdiv(v.shape[0], ke) * ke
                v = torch.cat([v, v.new_zeros((npad - v.shape[0],) + tuple(v.shape[1:]))], 0)
                if b is not None:
                    b = torch.cat([b, b.new_zeros(npad - b.shape[0])], 0)
            w[base] = O.pack_conv_weight(v, dtype, cin_pad=cin_pad, geglu=geglu).to(device)
            if b is not None:
                w[base + ".b"] = O.pack_bias(b, geglu=geglu).to(device)
        else:                                                   # norm scale/shift stay fp32
            w[base + ".g"] = v.float().contiguous().to(device)
            w[base + ".beta"] = b.float().contiguous().to(device)
    shapes = {k[:-7]: tuple(v.shape) for k, v in sd.items() if k.endswith(".weight")}
    # LayerNorm folded into its consumers (sr_igemm_args.row_stats): norm1 -> attn1.to_q/k/v, norm2 -> attn2.to_q,
    # norm3 -> ff.net.0.proj.  W' = W*gamma, colsum(W'), bias' = bias + W.beta are made once here; the step plan then reads the
    # un-normalised residual stream and never materialises LN(x) (attention.py:521-654)
    for k in list(sd):
        if not k.endswith(".norm1.weight"):
            continue
        tb = k[:-len(".norm1.weight")]
        for norm, lins in (("norm1", ("attn1.to_q", "attn1.to_k", "attn1.to_v")), ("norm2", ("attn2.to_q",)), ("norm3", ("ff.net.0.proj",))):
            g, bt = sd[f"{tb}.{norm}.weight"], sd[f"{tb}.{norm}.bias"]
            for lin in lins:
                base = f"{tb}.{lin}"
                wf, cs, b2 = O.fold_layernorm(sd[base + ".weight"], sd.get(base + ".bias"), g, bt, dtype, geglu=lin.endswith("ff.net.0.proj"))
                w[base + ".f"], w[base + ".f.cs"], w[base + ".f.b"] = wf.to(device), cs.to(device), b2.to(device)
    # every ResBlock's emb_layers.1 Linear reads the same SiLU(time embedding): one batched projection per step instead
    # of 22 GEMV-sized launches that each wait on a cold 3 MB weight read (openaimodel.py:262-267)
    embs = [k[:-7] for k in sd if k.endswith(".emb_layers.1.weight")]
    if embs:
        wcat = torch.cat([sd[e + ".weight"] for e in embs], 0)
        bcat = torch.cat([sd[e + ".bias"] for e in embs], 0)
        w["_emb_all"] = O.pack_conv_weight(wcat, dtype).to(device)
        w["_emb_all.b"] = O.pack_bias(bcat).to(device)
        off = 0
        for e in embs:
            shapes["_emb_off." + e[:-len(".emb_layers.1")]] = (off, sd[e + ".weight"].shape[0])
            off += sd[e + ".weight"].shape[0]
        shapes["_emb_all"] = (off, wcat.shape[1])
    return w, shapes


class BlockLowering:
    """ResBlock / SpatialTransformer / BasicTransformerBlock -> plan ops; shared by the UNet and the ControlNet encoder."""

    def __init__(self, pb, pro, W, shapes, B, cfg, emb_s, ctx, n_ctx, inject_idx=None, sel=None, external=False, sel_err=None):
        self.pb, self.pro, self.W, self.shapes, self.B, self.cfg = pb, pro, W, shapes, B, cfg
        self.sel_err = sel_err                         # device int32 raised by sr_gather_rows on an out-of-range injected index
        self.emb_s, self.ctx, self.n_ctx, self.inject_idx, self.sel = emb_s, ctx, n_ctx, inject_idx, sel
        self.ldt_ctx = _cdiv(n_ctx, 8) * 8
        # external=True (view-sharded multi-GPU): the injected frame's tokens come from another rank.  The plan is CUT
        # twice per transformer block and ``schedule`` lists what the host does in order:
        #   ("run", Plan)      ... up to and including norm1
        #   ("bcast", pairs)   owner: src <- ln[local index]; START the broadcast of src (async on the RCCL stream)
        #   ("run", Plan)      the B-frame Q projection: needs only the local ln, runs while the rows are on the wire
        #   ("wait",)          the compute stream waits for the transfer
        #   ("run", Plan)      K / V^T projections of the received rows, attention, ... next norm1
        self.external, self.schedule = external, []
        # LayerNorm folded into its consumer GEMMs (pack_weights' ".f" tensors, sr_igemm_args.row_stats): parity-clean, but
        # measured neutral on the SD1.5 step (22.33 vs 22.27 ms: 0.81 ms of LayerNorm kernels become 0.25 ms of row statistics,
        # 0.1 ms of extra gathers and +0.3 ms of GEMM epilogue), so the explicit kernels stay the default; SR_FOLD_LN=1 enables
        # it (not in the view-sharded mode)
        self.fold_ln = os.environ.get("SR_FOLD_LN", "0") == "1" and not external
        # ... or folded with the statistics taken INSIDE the consumer GEMMs (sr_igemm_args.ln_inline): no LayerNorm pass, no
        # statistics pass, no statistics tensor to gather or broadcast -- also in the view-sharded mode.  The default since round 4
        # (same box, B = 16 evaluation 18.72 -> 18.44 ms, B = 6 9.70 -> 9.52, B = 2 5.92 -> 5.87; 390 -> 358 ops); SR_LN_INLINE=0 disables
        self.ln_inline = os.environ.get("SR_LN_INLINE", "1") == "1"
        # (SR_LN_INLINE_ROWS = n keeps the LayerNorm kernels where a layer has fewer than n rows: there the consumers lose their
        #  split-K forms to the statistics, which need the whole row in one workgroup -- measured a wash at B = 2, a gain above)
        self.ln_inline_rows = int(os.environ.get("SR_LN_INLINE_ROWS", "0"))
        self.emb_all = None
        if "_emb_all" in W:
            ntot, kin = shapes["_emb_all"]
            self.emb_all = pb.buf(B, ntot, dtype=torch.float32)
            pb.igemm(emb_s, W["_emb_all"], self.emb_all, B, 1, 1, kin, ntot, bias=W["_emb_all.b"], out_f32=1)

    def resblock(self, p, x1, C1, x2, C2, Cout, HW, hh, ww):
        pb, pro, W, B, cfg = self.pb, self.pro, self.W, self.B, self.cfg
        mc, heads, emb_s, ctx, n_ctx, ldt_ctx = cfg["model_channels"], cfg["num_heads"], self.emb_s, self.ctx, self.n_ctx, self.ldt_ctx
        inject_idx, sel = self.inject_idx, self.sel
        cin = C1 + C2
        if self.emb_all is not None:
            off, n_e = self.shapes["_emb_off." + p]
            assert n_e == Cout
            er, er_ld = self.emb_all[:, off:off + Cout], self.emb_all.shape[1]
        else:
            er, er_ld = pb.buf(B, Cout, dtype=torch.float32), 0
            pb.igemm(emb_s, W[p + ".emb_layers.1"], er, B, 1, 1, 4 * mc, Cout, bias=W[p + ".emb_layers.1.b"], out_f32=1)
        has_skip = (p + ".skip_connection") in W
        skip_call = None
        if has_skip:
            # the 1x1 skip convolution only needs the block input: it rides in the launch of the first 3x3 convolution when the
            # tuner finds that faster (PlanBuilder.igemm_group) -- or on the side lane (SR_TWO_LANES=1), beside the GroupNorm / conv path
            skip = pb.buf(B, HW, Cout)
            skip_call = ((x1, W[p + ".skip_connection"], skip, B, hh, ww, C1, Cout), dict(a2=x2, C2=C2, bias=W[p + ".skip_connection.b"]))
            if pb.two_lanes:
                pb.fork()
                with pb.side():
                    pb.igemm(*skip_call[0], **skip_call[1])
                skip_call = None
        else:
            assert x2 is None and C1 == Cout
            skip = x1
        n1 = pb.buf(B, HW, cin)
        pb.groupnorm(x1, W[p + ".in_layers.0.g"], W[p + ".in_layers.0.beta"], n1, B, HW, C1, x2=x2, C2=C2, eps=1e-5, silu=True)
        hmid = pb.buf(B, HW, Cout)
        conv1 = ((n1, W[p + ".in_layers.2"], hmid, B, hh, ww, cin, Cout), dict(KH=3, bias=W[p + ".in_layers.2.b"], rowvec=er, rowvec_ld=er_ld))
        if skip_call is not None:
            pb.igemm_group([conv1, skip_call])
        else:
            pb.igemm(*conv1[0], **conv1[1])
        n2 = pb.buf(B, HW, Cout)
        pb.groupnorm(hmid, W[p + ".out_layers.0.g"], W[p + ".out_layers.0.beta"], n2, B, HW, Cout, eps=1e-5, silu=True)
        if has_skip:
            pb.join()
        out = pb.buf(B, HW, Cout)
        pb.igemm(n2, W[p + ".out_layers.3"], out, B, hh, ww, Cout, Cout, KH=3, bias=W[p + ".out_layers.3.b"], residual=skip)
        return out

    def tblock(self, p, hcur, Cc, HW):
        pb, pro, W, B, cfg = self.pb, self.pro, self.W, self.B, self.cfg
        mc, heads, emb_s, ctx, n_ctx, ldt_ctx = cfg["model_channels"], cfg["num_heads"], self.emb_s, self.ctx, self.n_ctx, self.ldt_ctx
        nhc = cfg.get("num_head_channels") or -1
        if nhc > 0:                                    # SDXL family: fixed head width, head count per level (openaimodel.py:601-608)
            heads = Cc // nhc
        inject_idx, sel = self.inject_idx, self.sel
        d = Cc // heads
        fold = self.fold_ln and (inject_idx is None or (HW * 8) % 16 == 0)     # (the statistics rows are gathered in 16-byte units)

        INLINE = "inline"

        def normed(x, norm):
            """-> (tensor the consumers read, folded-LN kwargs factory).  fold: the raw rows + their (rstd, -rstd*mean)"""
            if self.ln_inline and B * HW >= self.ln_inline_rows:
                return x, INLINE
            if not fold:
                y = pb.buf(B, HW, Cc)
                pb.layernorm(x, W[f"{p}.{norm}.g"], W[f"{p}.{norm}.beta"], y, B * HW, Cc)
                return y, None
            st = pb.buf(B, HW, 2, dtype=torch.float32)
            pb.row_stats(x, st, B * HW, Cc)
            return x, st

        def lin(name, st):
            """weight + epilogue kwargs of a Linear fed by a (possibly folded) LayerNorm"""
            if st is None:
                kw = {}
                if (f"{p}.{name}.b") in W:
                    kw["bias"] = W[f"{p}.{name}.b"]
                return W[f"{p}.{name}"], kw
            if st is INLINE:
                return W[f"{p}.{name}.f"], dict(bias=W[f"{p}.{name}.f.b"], ln_inline=True, colsum=W[f"{p}.{name}.f.cs"])
            return W[f"{p}.{name}.f"], dict(bias=W[f"{p}.{name}.f.b"], row_stats=st, colsum=W[f"{p}.{name}.f.cs"])
        ln, st1 = normed(hcur, "norm1")
        q = pb.buf(B, HW, Cc)
        if inject_idx is None:
            wq, kq = lin("attn1.to_q", st1)
            pb.igemm(ln, wq, q, B * HW, 1, 1, Cc, Cc, **kq)
            Bk, Tk, ldt = B, HW, _cdiv(HW, 8) * 8     # V^T rows padded to 16 B (pad columns stay zero)
            k = pb.buf(B, HW, Cc)
            vt = pb.buf(B, Cc, ldt, zero=True)
            wk, kk = lin("attn1.to_k", st1)
            wv, kv = lin("attn1.to_v", st1)
            pb.igemm(ln, wk, k, B * HW, 1, 1, Cc, Cc, **kk)
            pb.igemm(ln, wv, vt, B, HW, 1, Cc, Cc, transpose_out=1, ldt=ldt, **kv)
        else:
            nr = len(inject_idx)
            Bk, Tk, ldt = 1, nr * HW, _cdiv(nr * HW, 8) * 8
            k = pb.buf(1, Tk, Cc)
            vt = pb.buf(1, Cc, ldt, zero=True)
            # K/V of the injected frame(s) only (B-fold fewer projection FLOPs); the frame is picked on the device
            # (rows padded to a whole 128-row tile: as the swapped V^T problem's "weights" the tokens are read tile-wise)
            src = pb.buf(_cdiv(nr * HW, 128) * 128, Cc, zero=True)[:nr * HW].view(nr, HW, Cc)
            # ln_inline: `ln` is the raw residual stream (Q takes its statistics inside its GEMM); the ONE injected frame's raw rows
            # are picked / received into src_raw and normalised into src by a LayerNorm over nr * HW rows instead of B * HW
            inline1 = st1 is INLINE
            src_raw = pb.buf(nr, HW, Cc) if inline1 else src
            src_st = pb.buf(nr, HW, 2, dtype=torch.float32) if torch.is_tensor(st1) else None
            if self.external:
                self.schedule.append(("run", pb.take()))
                self.schedule.append(("bcast", [(ln, src_raw)] + ([(st1, src_st)] if torch.is_tensor(st1) else [])))
                wq, kq = lin("attn1.to_q", st1)
                pb.igemm(ln, wq, q, B * HW, 1, 1, Cc, Cc, **kq)
                self.schedule.append(("run", pb.take()))
                self.schedule.append(("wait",))
            # the injected frame's K / V^T are one-frame GEMMs (latency bound) and independent of the B-frame Q projection: the
            # three go out as ONE grouped launch when the tuner finds a tile under which that wins (PlanBuilder.igemm_group).
            # V^T = Wv . src^T is written as the row-major GEMM with the operands' roles swapped -- the packed weight rows
            # [C, K] are the "pixels", the gathered tokens [Tk, K] the "weights", out[c][t] with row stride Tk -- so that all
            # three are row-major problems of one kernel (needs ldt == Tk and no folded LayerNorm; else the transposed-output form)
            swap_v = ldt == Tk and src_st is None and not pb.two_lanes and (p + ".attn1.to_v.b") not in W
            pb.fork()
            with pb.side():
                if inline1 and not self.external:            # the frame picked and normalised in ONE launch
                    pb.layernorm_gather(ln, sel, nr, HW, B, W[f"{p}.norm1.g"], W[f"{p}.norm1.beta"], src, Cc, err_flag=self.sel_err)
                elif not self.external:
                    pb.gather_rows(ln, sel, src_raw, nr, HW * Cc * ln.element_size(), B, self.sel_err)
                    if torch.is_tensor(st1):
                        pb.gather_rows(st1, sel, src_st, nr, HW * 2 * 4, B, self.sel_err)
                if inline1 and self.external:                # the rows another rank sent
                    pb.layernorm(src_raw, W[f"{p}.norm1.g"], W[f"{p}.norm1.beta"], src, nr * HW, Cc)
                wk, kk = lin("attn1.to_k", src_st)
                wv, kv = lin("attn1.to_v", src_st)
                calls = [((src, wk, k, Tk, 1, 1, Cc, Cc), kk)]
                if swap_v:
                    calls.append(((wv, src, vt, Cc, 1, 1, Cc, Tk), {}))
                else:
                    calls.append(((src, wv, vt, 1, Tk, 1, Cc, Cc), dict(transpose_out=1, ldt=ldt, **kv)))
                if not self.external and not pb.two_lanes:
                    wq, kq = lin("attn1.to_q", st1)
                    calls.insert(0, ((ln, wq, q, B * HW, 1, 1, Cc, Cc), kq))
                pb.igemm_group(calls)
            if not self.external and pb.two_lanes:
                wq, kq = lin("attn1.to_q", st1)
                pb.igemm(ln, wq, q, B * HW, 1, 1, Cc, Cc, **kq)
            pb.join()
        a = pb.buf(B, HW, Cc)
        pb.attention(q, k, vt, a, B, Bk, HW, Tk, heads, d, ldt)
        h1 = pb.buf(B, HW, Cc)
        pb.igemm(a, W[p + ".attn1.to_out.0"], h1, B * HW, 1, 1, Cc, Cc, bias=W[p + ".attn1.to_out.0.b"], residual=hcur)
        # cross attention: K/V from the prompt, projected once in the prologue plan
        ln2, st2 = normed(h1, "norm2")
        q2 = pb.buf(B, HW, Cc)
        wq2, kq2 = lin("attn2.to_q", st2)
        pb.igemm(ln2, wq2, q2, B * HW, 1, 1, Cc, Cc, **kq2)
        k2 = pro.buf(B, n_ctx, Cc)
        vt2 = pro.buf(B, Cc, ldt_ctx, zero=True)
        cd = cfg["context_dim"]
        pro.igemm(ctx, W[p + ".attn2.to_k"], k2, B * n_ctx, 1, 1, cd, Cc)
        pro.igemm(ctx, W[p + ".attn2.to_v"], vt2, B, n_ctx, 1, cd, Cc, transpose_out=1, ldt=ldt_ctx)
        a2 = pb.buf(B, HW, Cc)
        pb.attention(q2, k2, vt2, a2, B, B, HW, n_ctx, heads, d, ldt_ctx)
        h2 = pb.buf(B, HW, Cc)
        pb.igemm(a2, W[p + ".attn2.to_out.0"], h2, B * HW, 1, 1, Cc, Cc, bias=W[p + ".attn2.to_out.0.b"], residual=h1)
        ln3, st3 = normed(h2, "norm3")
        inner = self.shapes[p + ".ff.net.0.proj"][0] // 2
        ff = pb.buf(B, HW, inner)
        wff, kff = lin("ff.net.0.proj", st3)
        pb.igemm(ln3, wff, ff, B * HW, 1, 1, Cc, 2 * inner, act=2, **kff)
        h3 = pb.buf(B, HW, Cc)
        pb.igemm(ff, W[p + ".ff.net.2"], h3, B * HW, 1, 1, inner, Cc, bias=W[p + ".ff.net.2.b"], residual=h2)
        return h3

    def stransformer(self, p, x, Cc, HW, hh, ww, depth):
        pb, pro, W, B, cfg = self.pb, self.pro, self.W, self.B, self.cfg
        mc, heads, emb_s, ctx, n_ctx, ldt_ctx = cfg["model_channels"], cfg["num_heads"], self.emb_s, self.ctx, self.n_ctx, self.ldt_ctx
        inject_idx, sel = self.inject_idx, self.sel
        n = pb.buf(B, HW, Cc)
        pb.groupnorm(x, W[p + ".norm.g"], W[p + ".norm.beta"], n, B, HW, Cc, eps=1e-6, silu=False)
        hcur = pb.buf(B, HW, Cc)
        pb.igemm(n, W[p + ".proj_in"], hcur, B, hh, ww, Cc, Cc, bias=W[p + ".proj_in.b"])
        for i in range(depth):
            hcur = self.tblock(f"{p}.transformer_blocks.{i}", hcur, Cc, HW)
        out = pb.buf(B, HW, Cc)
        pb.igemm(hcur, W[p + ".proj_out"], out, B, hh, ww, Cc, Cc, bias=W[p + ".proj_out.b"], residual=x)
        return out



class UNet:
    def __init__(self, state_dict, cfg=None, dtype=torch.float16, device="cuda"):
        self.cfg = dict(SD15_CFG if cfg is None else cfg)
        self.dtype, self.device = dtype, torch.device(device)
        self.ke = O.kelems(dtype)
        self.w, self.shapes = pack_weights(state_dict, dtype, self.device, pad_cin=("input_blocks.0.0",))

    # ------------------------------------------------------------------------------------------------
    def build(self, B, h, w, inject_idx=None, n_ctx=77, control=None, inject_external=False, inputs=None):
        """-> dict(prologue=Plan, step=Plan, x=(B,4,h,w) fp32 input buffer, t=(B,) fp32, ctx=(B,n_ctx,ctx_dim),
        out=(B,4,h,w) fp32, inject=(n_rand,) int32 device tensor or None).  inject_idx: list of batch indices whose
        post-LayerNorm tokens every batch entry attends to in self-attention (OverlapCorresponder.pre_atten_inject) or
        None.  The indices live in a device tensor read at run time, so the plan (and its captured graph) is reused when
        the random frame changes between sampling runs: only ``inject`` is rewritten.
        control: optional dict(output=[12 NHWC tensors], middle=NHWC tensor) produced by ControlNet.build for the same
        (B,h,w): added to the skips / middle output as apply_control does (openaimodel.py:374-386, :899, :907)."""
        cfg, dt, dev = self.cfg, self.dtype, self.device
        pb = PlanBuilder(dev, dt)
        pro = PlanBuilder(dev, dt)                     # prompt-only work (cross-attention K/V)
        W = self.w
        mc, heads = cfg["model_channels"], cfg["num_heads"]
        if inputs is not None:                         # (x, t, ctx) buffers shared with ControlNet plans built on them
            x_in, t_in, ctx = inputs
            pb.hold(x_in, t_in, ctx)
        else:
            x_in = pb.buf(B, cfg["in_channels"], h, w, dtype=torch.float32, zero=True)
            t_in = pb.buf(B, dtype=torch.float32, zero=True)
            ctx = pb.buf(B, n_ctx, cfg["context_dim"], zero=True)
        ldt_ctx = _cdiv(n_ctx, 8) * 8
        sel, sel_err = None, None
        if inject_idx is not None:
            if not inject_external and any(int(i) < 0 or int(i) >= B for i in inject_idx):
                raise IndexError(f"injected frame index out of range for batch {B}: {list(inject_idx)}")
            sel = pb.buf(len(inject_idx), dtype=torch.int32)
            sel.copy_(torch.tensor([int(i) if 0 <= int(i) < B else 0 for i in inject_idx], dtype=torch.int32))
            sel_err = pb.buf(1, dtype=torch.int32, zero=True)

        # ---- time embedding ------------------------------------------------------------------------
        temb = pb.buf(B, mc)
        pb.timestep_embedding(t_in, temb, B, mc)
        e1 = pb.buf(B, 4 * mc)
        pb.igemm(temb, W["time_embed.0"], e1, B, 1, 1, mc, 4 * mc, bias=W["time_embed.0.b"], act=1)
        e2 = pb.buf(B, 4 * mc)
        pb.igemm(e1, W["time_embed.2"], e2, B, 1, 1, 4 * mc, 4 * mc, bias=W["time_embed.2.b"])
        y_in = None
        if cfg.get("adm_in_channels"):
            # vector conditioning (SDXL: pooled text + size / crop embeddings): emb = time_embed(t) + label_emb(y)
            # (openaimodel.py:862-864).  y is constant over a sampling run -> label_emb runs in the prologue plan.
            adm = cfg["adm_in_channels"]
            adm_pad = _cdiv(adm, self.ke) * self.ke
            y_in = pro.buf(B, adm_pad, zero=True)
            l1 = pro.buf(B, 4 * mc)
            pro.igemm(y_in, W["label_emb.0.0"], l1, B, 1, 1, adm_pad, 4 * mc, bias=W["label_emb.0.0.b"], act=1)
            lab = pro.buf(B, 4 * mc)
            pro.igemm(l1, W["label_emb.0.2"], lab, B, 1, 1, 4 * mc, 4 * mc, bias=W["label_emb.0.2.b"])
            e3 = pb.buf(B, 4 * mc)
            pb.add(e2, lab, e3)
            e2 = e3
        emb_s = pb.buf(B, 4 * mc)
        pb.silu(e2, emb_s)                             # every ResBlock applies SiLU first (emb_layers.0)

        low = BlockLowering(pb, pro, W, self.shapes, B, cfg, emb_s, ctx, n_ctx, inject_idx, sel, external=inject_external, sel_err=sel_err)
        resblock, stransformer = low.resblock, low.stransformer

        # ---- encoder ---------------------------------------------------------------------------------
        cin_pad = _cdiv(cfg["in_channels"], self.ke) * self.ke
        xh = pb.buf(B, h * w, cin_pad)
        pb.nchw_to_nhwc(x_in, xh, B, cfg["in_channels"], h * w, cin_pad)
        cur = pb.buf(B, h * w, mc)
        pb.igemm(xh, W["input_blocks.0.0"], cur, B, h, w, cin_pad, mc, KH=3, bias=W["input_blocks.0.0.b"])
        hs = [(cur, mc, h, w)]
        ch, hh, ww = mc, h, w
        td = list(cfg["transformer_depth"])
        nlev = len(cfg["channel_mult"])
        bi = 1
        for lev in range(nlev):
            cout = mc * cfg["channel_mult"][lev]
            for _ in range(cfg["num_res_blocks"][lev]):
                cur = resblock(f"input_blocks.{bi}.0", cur, ch, None, 0, cout, hh * ww, hh, ww)
                ch = cout
                depth = td.pop(0)
                if depth > 0:
                    cur = stransformer(f"input_blocks.{bi}.1", cur, ch, hh * ww, hh, ww, depth)
                hs.append((cur, ch, hh, ww))
                bi += 1
            if lev != nlev - 1:
                ho, wo = (hh + 1) // 2, (ww + 1) // 2
                dn = pb.buf(B, ho * wo, ch)
                pb.igemm(cur, W[f"input_blocks.{bi}.0.op"], dn, B, hh, ww, ch, ch, KH=3, stride=2, bias=W[f"input_blocks.{bi}.0.op.b"])
                cur, hh, ww = dn, ho, wo
                hs.append((cur, ch, hh, ww))
                bi += 1
        # ---- middle ----------------------------------------------------------------------------------
        cur = resblock("middle_block.0", cur, ch, None, 0, ch, hh * ww, hh, ww)
        cur = stransformer("middle_block.1", cur, ch, hh * ww, hh, ww, cfg["transformer_depth_middle"])
        cur = resblock("middle_block.2", cur, ch, None, 0, ch, hh * ww, hh, ww)

        def add_control(x, c):
            y = pb.buf(*x.shape)
            pb.add(x, c, y)
            return y
        ctrl_out = list(control["output"]) if control is not None else None
        if control is not None and control.get("middle") is not None:
            cur = add_control(cur, control["middle"])
        # ---- decoder ---------------------------------------------------------------------------------
        tdo = list(cfg["transformer_depth_output"])
        bo = 0
        for lev in reversed(range(nlev)):
            cout = mc * cfg["channel_mult"][lev]
            for i in range(cfg["num_res_blocks"][lev] + 1):
                skip, cs, sh, sw = hs.pop()
                assert (sh, sw) == (hh, ww)
                if ctrl_out:
                    c_ = ctrl_out.pop()
                    if c_ is not None:
                        skip = add_control(skip, c_)
                cur = resblock(f"output_blocks.{bo}.0", cur, ch, skip, cs, cout, hh * ww, hh, ww)
                ch = cout
                depth = tdo.pop()                      # the reference pops from the END (openaimodel.py:737)
                j = 1
                if depth > 0:
                    cur = stransformer(f"output_blocks.{bo}.1", cur, ch, hh * ww, hh, ww, depth)
                    j = 2
                if lev > 0 and i == cfg["num_res_blocks"][lev]:
                    # Upsample.forward(x, output_shape = hs[-1].shape): nearest to the NEXT skip's size, which is 2x only for
                    # even sizes (openaimodel.py:109-121; forward_timestep_embed passes output_shape, :59-60)
                    uh, uw = hs[-1][2], hs[-1][3]
                    up = pb.buf(B, uh * uw, ch)
                    pb.igemm(cur, W[f"output_blocks.{bo}.{j}.conv"], up, B, hh, ww, ch, ch, KH=3, upsample=1, up_hw=(uh, uw),
                             bias=W[f"output_blocks.{bo}.{j}.conv.b"])
                    cur, hh, ww = up, uh, uw
                bo += 1
        # ---- out -------------------------------------------------------------------------------------
        n = pb.buf(B, hh * ww, ch)
        pb.groupnorm(cur, W["out.0.g"], W["out.0.beta"], n, B, hh * ww, ch, eps=1e-5, silu=True)
        oc = cfg["out_channels"]
        o_nhwc = pb.buf(B, hh * ww, oc, dtype=torch.float32)
        pb.igemm(n, W["out.2"], o_nhwc, B, hh, ww, ch, oc, KH=3, bias=W["out.2.b"], out_f32=1)
        out = pb.buf(B, oc, hh, ww, dtype=torch.float32)
        pb.nhwc_to_nchw(o_nhwc, out, B, oc, hh * ww, oc)
        flops = pb.flops
        step = pb.take()
        return dict(prologue=pro.take(), step=step, x=x_in, t=t_in, ctx=ctx, y=y_in, out=out, flops=flops, inject=sel, inject_err=sel_err,
                    schedule=(low.schedule + [("run", step)]) if low.schedule else [])
