"""VAE decoder lowered onto the HIP launch plan.

Mirrors ``VAE.decode`` (comfyUI/comfy/sd.py:329-346): ``post_quant_conv`` -> ``Decoder.forward``
(comfy/ldm/modules/diffusionmodules/model.py:617-650; ResnetBlock :117-170, AttnBlock :173-268, Upsample :54-74) ->
``clamp((y+1)/2, 0, 1)`` -> NHWC.  Same kernels as the UNet: NHWC implicit-GEMM convs with the nearest-x2 upsample
fused into the following conv's gather, GroupNorm+SiLU streaming passes, residual adds in GEMM epilogues.  The single
512-wide mid attention (4096 tokens at 512^2) is two GEMMs + a row softmax per image.
"""
import torch

from . import ops as O
from .plan import PlanBuilder


def _cdiv(a, b):
    return (a + b - 1) // b


class _Lowering:
    """ResnetBlock (model.py:117-170) and the single-head AttnBlock (:173-268) -> plan ops; shared by decoder and encoder"""

    def __init__(self, pb, W, B, ke):
        self.pb, self.W, self.B, self.ke = pb, W, B, ke

    def res(self, p, x, cin, cout, hh, ww):
        pb, W, B = self.pb, self.W, self.B
        HW = hh * ww
        n1 = pb.buf(B, HW, cin)
        pb.groupnorm(x, W[p + ".norm1.g"], W[p + ".norm1.beta"], n1, B, HW, cin, eps=1e-6, silu=True)
        h1 = pb.buf(B, HW, cout)
        pb.igemm(n1, W[p + ".conv1"], h1, B, hh, ww, cin, cout, KH=3, bias=W[p + ".conv1.b"])
        n2 = pb.buf(B, HW, cout)
        pb.groupnorm(h1, W[p + ".norm2.g"], W[p + ".norm2.beta"], n2, B, HW, cout, eps=1e-6, silu=True)
        if (p + ".nin_shortcut") in W:
            sk = pb.buf(B, HW, cout)
            pb.igemm(x, W[p + ".nin_shortcut"], sk, B, hh, ww, cin, cout, bias=W[p + ".nin_shortcut.b"])
        else:
            sk = x
        out = pb.buf(B, HW, cout)
        pb.igemm(n2, W[p + ".conv2"], out, B, hh, ww, cout, cout, KH=3, bias=W[p + ".conv2.b"], residual=sk)
        return out

    def attn(self, p, x, Cc, hh, ww):
        pb, W, B = self.pb, self.W, self.B
        HW = hh * ww
        if HW % self.ke:
            raise ValueError("VAE mid attention needs h*w to be a multiple of %d" % self.ke)
        rows = _cdiv(HW, 128) * 128                 # K is the weight operand of the score GEMM: pad rows
        n = pb.buf(B, HW, Cc)
        pb.groupnorm(x, W[p + ".norm.g"], W[p + ".norm.beta"], n, B, HW, Cc, eps=1e-6, silu=False)
        q = pb.buf(B, HW, Cc)
        k = pb.buf(B, rows, Cc, zero=True)
        vt = pb.buf(B, _cdiv(Cc, 128) * 128, HW, zero=True)
        pb.igemm(n, W[p + ".q"], q, B, hh, ww, Cc, Cc, bias=W[p + ".q.b"])
        for b in range(B):
            pb.igemm(n[b], W[p + ".k"], k[b], 1, hh, ww, Cc, Cc, bias=W[p + ".k.b"])
            pb.igemm(n[b], W[p + ".v"], vt[b], 1, hh, ww, Cc, Cc, bias=W[p + ".v.b"], transpose_out=1, ldt=HW)
        s = pb.buf(HW, HW)                          # one image at a time: 32 MiB of scores at 512^2 (fp16)
        o = pb.buf(B, HW, Cc)
        for b in range(B):
            pb.igemm(q[b], k[b], s, HW, 1, 1, Cc, HW, scale=float(Cc) ** -0.5)
            pb.softmax_rows(s, HW, HW)
            pb.igemm(s, vt[b], o[b], HW, 1, 1, HW, Cc)
        out = pb.buf(B, HW, Cc)
        pb.igemm(o, W[p + ".proj_out"], out, B, hh, ww, Cc, Cc, bias=W[p + ".proj_out.b"], residual=x)
        return out



class VAEDecoder:
    def __init__(self, state_dict, ch_mult=(1, 2, 4, 4), num_res_blocks=2, dtype=torch.float16, device="cuda", prefix="",
                 extra=("post_quant_conv",)):
        self.dtype, self.device = dtype, torch.device(device)
        self.ke = O.kelems(dtype)
        self.ch_mult, self.nrb = tuple(ch_mult), num_res_blocks
        self.w = {}
        self.shapes = {}
        for k, v in state_dict.items():
            if prefix and not (k.startswith(prefix) or any(k.startswith(e) for e in extra)):
                continue
            if not k.endswith(".weight"):
                continue
            name = k[len(prefix):] if prefix and k.startswith(prefix) else k
            base = name[:-7]
            b = state_dict.get(k[:-7] + ".bias")
            self.shapes[base] = tuple(v.shape)
            if v.dim() >= 2:
                cin_pad = _cdiv(v.shape[1], self.ke) * self.ke
                self.w[base] = O.pack_conv_weight(v, dtype, cin_pad=cin_pad).to(self.device)
                if b is not None:
                    self.w[base + ".b"] = O.pack_bias(b).to(self.device)
            else:
                self.w[base + ".g"] = v.float().contiguous().to(self.device)
                self.w[base + ".beta"] = b.float().contiguous().to(self.device)

    def build(self, B, h, w, clamp=True):
        """-> dict(plan, z=(B,4,h,w) fp32 input buffer, img=(B,8h,8w,3) fp32 NHWC output: clamp((y+1)/2,0,1) fused
        into the last conv's epilogue, or the raw decoder output y when clamp=False)"""
        dt, dev, W = self.dtype, self.device, self.w
        pb = PlanBuilder(dev, dt)
        zc = self.shapes["conv_in"][1]
        z_in = pb.buf(B, zc, h, w, dtype=torch.float32, zero=True)
        cpad = _cdiv(zc, self.ke) * self.ke
        zh = pb.buf(B, h * w, cpad)
        pb.nchw_to_nhwc(z_in, zh, B, zc, h * w, cpad)
        if "post_quant_conv" in W:
            zq = pb.buf(B, h * w, cpad, zero=True)
            # N = zc valid channels written into a cpad-wide buffer is not expressible (row stride = N), so the
            # 1x1 post-quant conv output goes to a compact buffer and is re-padded by the layout kernel
            zq_c = pb.buf(B, h * w, zc, dtype=torch.float32)
            pb.igemm(zh, W["post_quant_conv"], zq_c, B, h, w, cpad, zc, bias=W["post_quant_conv.b"], out_f32=1)
            tmp = pb.buf(B, zc, h, w, dtype=torch.float32)
            pb.nhwc_to_nchw(zq_c, tmp, B, zc, h * w, zc)
            pb.nchw_to_nhwc(tmp, zq, B, zc, h * w, cpad)
            zh = zq

        low = _Lowering(pb, W, B, self.ke)
        res, attn = low.res, low.attn
        cin = self.shapes["conv_in"][0]
        cur = pb.buf(B, h * w, cin)
        pb.igemm(zh, W["conv_in"], cur, B, h, w, cpad, cin, KH=3, bias=W["conv_in.b"])
        cur = res("mid.block_1", cur, cin, cin, h, w)
        cur = attn("mid.attn_1", cur, cin, h, w)
        cur = res("mid.block_2", cur, cin, cin, h, w)
        hh, ww, ch = h, w, cin
        for lev in reversed(range(len(self.ch_mult))):
            cout = self.shapes[f"up.{lev}.block.0.conv1"][0]
            for i in range(self.nrb + 1):
                cur = res(f"up.{lev}.block.{i}", cur, ch, cout, hh, ww)
                ch = cout
            if lev != 0:
                up = pb.buf(B, 4 * hh * ww, ch)
                pb.igemm(cur, W[f"up.{lev}.upsample.conv"], up, B, hh, ww, ch, ch, KH=3, upsample=1, bias=W[f"up.{lev}.upsample.conv.b"])
                cur, hh, ww = up, 2 * hh, 2 * ww
        n = pb.buf(B, hh * ww, ch)
        pb.groupnorm(cur, W["norm_out.g"], W["norm_out.beta"], n, B, hh * ww, ch, eps=1e-6, silu=True)
        oc = self.shapes["conv_out"][0]
        raw_nhwc = pb.buf(B, hh, ww, oc, dtype=torch.float32)
        pb.igemm(n, W["conv_out"], raw_nhwc, B, hh, ww, ch, oc, KH=3, bias=W["conv_out.b"], out_f32=1, act=4 if clamp else 0)
        flops = pb.flops
        return dict(plan=pb.take(), z=z_in, img=raw_nhwc, flops=flops, out_hw=(hh, ww))


class VAEEncoder(VAEDecoder):
    """``VAE.encode`` (comfyUI/comfy/sd.py:353-371): pixels (N,H,W,3) in [0,1] -> ``2x - 1`` -> ``Encoder.forward``
    (comfy/ldm/modules/diffusionmodules/model.py:441-520: conv_in, per level 2 ResnetBlocks + Downsample = pad bottom/right by
    one + 3x3 stride-2 conv (:77-95), mid block with the single-head attention, norm_out / SiLU / conv_out -> 2*z channels)
    -> ``quant_conv`` -> ``DiagonalGaussianRegularizer`` with sample=True (comfy/ldm/models/autoencoder.py:13-31, :175-190):
    z = mean + exp(0.5*clamp(logvar, -30, 20)) * randn.  The noise is drawn by the caller from the global CPU generator
    (``torch.randn(mean.shape)``, distributions.py:35-37) and handed in, so the draw order of a graph stays the reference's.
    Checkpoint keys: ``encoder.*`` + ``quant_conv`` (prefix "encoder."), or the bare Encoder names + ``quant_conv``."""

    def __init__(self, state_dict, ch_mult=(1, 2, 4, 4), num_res_blocks=2, dtype=torch.float16, device="cuda", prefix=""):
        super().__init__(state_dict, ch_mult, num_res_blocks, dtype, device, prefix, extra=("quant_conv",))

    def build(self, B, H, W_px):
        """-> dict(plan, pixels=(B,3,H,W) fp32 NCHW input buffer in [0,1], noise=(B,z,h,w) fp32, z=(B,z,h,w) fp32 output,
        moments=(B,h*w,2z) fp32 [mean | logvar])"""
        dt, dev, W = self.dtype, self.device, self.w
        if H % 8 or W_px % 8:
            raise ValueError("VAE encode wants H and W to be multiples of 8 (vae_encode_crop_pixels, sd.py:292-299, crops first)")
        pb = PlanBuilder(dev, dt)
        pix = pb.buf(B, 3, H, W_px, dtype=torch.float32, zero=True)
        minus1 = pb.buf(B, 3, H, W_px, dtype=torch.float32)
        minus1.fill_(-1.0)
        xs = pb.buf(B, 3, H, W_px, dtype=torch.float32)
        pb.add(minus1, pix, xs, s=2.0)                                  # process_input: 2x - 1
        cpad = _cdiv(3, self.ke) * self.ke
        xh = pb.buf(B, H * W_px, cpad)
        pb.nchw_to_nhwc(xs, xh, B, 3, H * W_px, cpad)
        low = _Lowering(pb, W, B, self.ke)
        ch = self.shapes["conv_in"][0]
        cur = pb.buf(B, H * W_px, ch)
        pb.igemm(xh, W["conv_in"], cur, B, H, W_px, cpad, ch, KH=3, bias=W["conv_in.b"])
        hh, ww = H, W_px
        for lev in range(len(self.ch_mult)):
            cout = self.shapes[f"down.{lev}.block.0.conv1"][0]
            for i in range(self.nrb):
                cur = low.res(f"down.{lev}.block.{i}", cur, ch, cout, hh, ww)
                ch = cout
            if lev != len(self.ch_mult) - 1:
                ho, wo = hh // 2, ww // 2
                dn = pb.buf(B, ho * wo, ch)
                pb.igemm(cur, W[f"down.{lev}.downsample.conv"], dn, B, hh, ww, ch, ch, KH=3, stride=2, pad_br=1,
                         bias=W[f"down.{lev}.downsample.conv.b"])
                cur, hh, ww = dn, ho, wo
        cur = low.res("mid.block_1", cur, ch, ch, hh, ww)
        cur = low.attn("mid.attn_1", cur, ch, hh, ww)
        cur = low.res("mid.block_2", cur, ch, ch, hh, ww)
        n = pb.buf(B, hh * ww, ch)
        pb.groupnorm(cur, W["norm_out.g"], W["norm_out.beta"], n, B, hh * ww, ch, eps=1e-6, silu=True)
        z2 = self.shapes["conv_out"][0]                                 # 2 * z_channels
        m0 = pb.buf(B, hh * ww, z2, dtype=torch.float32)
        pb.igemm(n, W["conv_out"], m0, B, hh, ww, ch, z2, KH=3, bias=W["conv_out.b"], out_f32=1)
        # quant_conv (1x1, 2z -> 2z): its K must be a K-step multiple -> re-pad the 8 channels through the layout kernels
        t1 = pb.buf(B, z2, hh, ww, dtype=torch.float32)
        pb.nhwc_to_nchw(m0, t1, B, z2, hh * ww, z2)
        zpad = _cdiv(z2, self.ke) * self.ke
        t2 = pb.buf(B, hh * ww, zpad, zero=True)
        pb.nchw_to_nhwc(t1, t2, B, z2, hh * ww, zpad)
        mom = pb.buf(B, hh * ww, z2, dtype=torch.float32)
        pb.igemm(t2, W["quant_conv"], mom, B, hh, ww, zpad, z2, bias=W["quant_conv.b"], out_f32=1)
        zc = z2 // 2
        noise = pb.buf(B, zc, hh, ww, dtype=torch.float32, zero=True)
        z = pb.buf(B, zc, hh, ww, dtype=torch.float32)
        flops = pb.flops
        return dict(plan=pb.take(), pixels=pix, noise=noise, z=z, moments=mom, flops=flops, latent_hw=(hh, ww))

    def encode(self, built, pixels_nhwc, noise=None):
        """pixels (B,H,W,>=3) in [0,1] -> z (B,zc,h,w) fp32 (a fresh tensor).  noise None: drawn here from the global CPU generator
        exactly as DiagonalGaussianDistribution.sample does"""
        b = built
        b["pixels"].copy_(pixels_nhwc[..., :3].movedim(-1, 1))          # layout only (sd.py:355)
        b["plan"].run()
        if noise is None:
            noise = torch.randn(tuple(b["noise"].shape))
        b["noise"].copy_(noise)
        O.vae_sample(b["moments"], b["noise"], b["z"])
        return b["z"].clone()
