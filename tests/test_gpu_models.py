"""GPU parity of the lowered UNet / VAE decoder against the golden vectors produced by the reference modules
(tests/golden/unet_*.npz, vae_dec.npz).  fp32 path: exact-fp32 MFMA, tolerance 2e-3 abs on O(1) outputs (the
accumulation order differs from the CPU conv kernels); fp16 path: the reference-on-ROCm precision, tolerance 6e-2."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _sd(name, seed):
    from stable_renderer_amd import synth
    with open(os.path.join(GOLD, name)) as f:
        k = json.load(f)
    return synth.synth_state_dict([(n, tuple(s)) for n, s in k["names_shapes"]], seed=seed, norm_names=k["norm_names"])


def T(a):
    return torch.from_numpy(np.asarray(a))


def run_unet(sd, cfg, dtype, x, t, ctx, inject=None, yvec=None):
    from stable_renderer_amd.unet import UNet
    net = UNet(sd, cfg, dtype=dtype)
    B = x.shape[0]
    p = net.build(B, x.shape[2], x.shape[3], inject_idx=inject, n_ctx=ctx.shape[1])
    p["x"].copy_(x)
    p["t"].copy_(t)
    p["ctx"].copy_(ctx.to(dtype))
    if yvec is not None:
        p["y"][:, :yvec.shape[1]].copy_(yvec.to(dtype))
    p["prologue"].run()
    p["step"].run()
    torch.cuda.synchronize()
    return p["out"].cpu(), p


@pytest.mark.parametrize("dtype,atol", [(torch.float32, 2e-3), (torch.float16, 6e-2)])
def test_unet_tiny(dtype, atol):
    from stable_renderer_amd.unet import SD15_CFG
    cfg = dict(SD15_CFG, model_channels=64, context_dim=64)
    d = np.load(os.path.join(GOLD, "unet_tiny.npz"))
    sd = _sd("unet_tiny_keys.json", 1)
    y, _ = run_unet(sd, cfg, dtype, T(d["x"]), T(d["t"]), T(d["ctx"]))
    ref = T(d["y"])
    assert (y - ref).abs().max().item() < atol * max(1.0, ref.abs().max().item()), (y - ref).abs().max()
    yi, _ = run_unet(sd, cfg, dtype, T(d["x"]), T(d["t"]), T(d["ctx"]), inject=d["inj_idx"].tolist())
    refi = T(d["y_inj"])
    assert (yi - refi).abs().max().item() < atol * max(1.0, refi.abs().max().item()), (yi - refi).abs().max()
    # a latent that does not halve evenly (10x12 -> 5x6 -> 3x3 -> 2x2): the upsample fused into the decoder's convs targets the
    # skip tensor's size (sr_igemm_args.up_h / up_w), as Upsample.forward(x, output_shape) does
    yo, _ = run_unet(sd, cfg, dtype, T(d["x_odd"]), T(d["t"])[:2], T(d["ctx"])[:2])
    refo = T(d["y_odd"])
    assert (yo - refo).abs().max().item() < atol * max(1.0, refo.abs().max().item()), (yo - refo).abs().max()


@pytest.mark.parametrize("dtype,atol", [(torch.float32, 2e-3), (torch.float16, 6e-2)])
def test_unet_sd15_shapes(dtype, atol):
    from stable_renderer_amd.unet import SD15_CFG
    d = np.load(os.path.join(GOLD, "unet_sd15_16.npz"))
    sd = _sd("unet_sd15_keys.json", 0)
    y, p = run_unet(sd, SD15_CFG, dtype, T(d["x"]), T(d["t"]), T(d["ctx"]))
    ref = T(d["y"])
    err = (y - ref).abs().max().item()
    assert err < atol * max(1.0, ref.abs().max().item()), err
    # graph replay gives the same bits as the eager plan
    s = torch.cuda.Stream()
    p["step"].capture(s)
    with torch.cuda.stream(s):
        p["step"].launch()
    s.synchronize()
    assert torch.equal(p["out"].cpu(), y)


SDXL_TINY = dict(in_channels=4, out_channels=4, model_channels=64, num_res_blocks=[2, 2, 2], channel_mult=[1, 2, 4],
                 transformer_depth=[0, 0, 2, 2, 3, 3], transformer_depth_middle=3, transformer_depth_output=[0, 0, 0, 2, 2, 2, 3, 3, 3],
                 context_dim=128, num_heads=-1, num_head_channels=32, use_linear_in_transformer=True, adm_in_channels=192)


@pytest.mark.parametrize("dtype,atol", [(torch.float32, 2e-3), (torch.float16, 6e-2)])
def test_unet_sdxl_family(dtype, atol):
    """SDXL topology (BASELINE config 5; comfy/supported_models.py:153-160) at 1/5 width against the reference UNetModel: no
    attention at the first level, 2 / 3 transformer blocks per SpatialTransformer below, 32-wide heads (2 / 4 / 8 per level),
    linear proj_in / proj_out, vector conditioning through label_emb"""
    from stable_renderer_amd.model_shapes import unet_names_shapes
    from stable_renderer_amd.unet import SDXL_CFG
    d = np.load(os.path.join(GOLD, "unet_sdxl_tiny.npz"))
    sd = _sd("unet_sdxl_tiny_keys.json", 4)
    y, p = run_unet(sd, SDXL_TINY, dtype, T(d["x"]), T(d["t"]), T(d["ctx"]), yvec=T(d["yvec"]))
    ref = T(d["y"])
    err = (y - ref).abs().max().item()
    assert err < atol * max(1.0, ref.abs().max().item()), err
    ns, _ = unet_names_shapes(SDXL_CFG)                                # the full-size table is the 2.57 B-parameter SDXL base UNet
    assert sum(int(np.prod(s)) for _, s in ns) == 2567463684


@pytest.mark.parametrize("dtype,atol", [(torch.float32, 2e-3), (torch.float16, 6e-2)])
def test_vae_decoder(dtype, atol):
    from stable_renderer_amd.vae import VAEDecoder
    d = np.load(os.path.join(GOLD, "vae_dec.npz"))
    sd = _sd("vae_dec_keys.json", 2)
    dec = VAEDecoder(sd, dtype=dtype)
    z = T(d["z"])
    p = dec.build(z.shape[0], z.shape[2], z.shape[3], clamp=False)
    p["z"].copy_(z)
    p["plan"].run()
    torch.cuda.synchronize()
    raw = p["img"].cpu().permute(0, 3, 1, 2)
    ref = T(d["y"])
    err = (raw - ref).abs().max().item()
    assert err < atol * max(1.0, ref.abs().max().item()), err
    p2 = dec.build(z.shape[0], z.shape[2], z.shape[3], clamp=True)
    p2["z"].copy_(z)
    p2["plan"].run()
    torch.cuda.synchronize()
    img = p2["img"].cpu()
    assert (img - T(d["img"])).abs().max().item() < atol
    assert float(img.min()) >= 0.0 and float(img.max()) <= 1.0


@pytest.mark.parametrize("dtype,atol", [(torch.float32, 2e-3), (torch.float16, 6e-2)])
def test_vae_encoder(dtype, atol):
    """VAE.encode on the HIP plan (Encoder with the bottom/right-padded stride-2 Downsample convs, quant_conv, posterior sample)
    vs the reference AutoencoderKL's moments and its sampled z (same global-generator noise)"""
    from stable_renderer_amd.vae import VAEEncoder
    d = np.load(os.path.join(GOLD, "vae_enc.npz"))
    enc = VAEEncoder(_sd("vae_enc_keys.json", 3), dtype=dtype)
    px = T(d["pixels"])
    b = enc.build(px.shape[0], px.shape[1], px.shape[2])
    assert b["latent_hw"] == (8, 8)
    torch.manual_seed(31)
    z = enc.encode(b, px.cuda())                                   # draws the posterior noise from the global CPU generator
    torch.cuda.synchronize()
    mom = b["moments"].cpu().reshape(px.shape[0], 8, 8, 8).permute(0, 3, 1, 2)
    ref_m, ref_z = T(d["moments"]), T(d["z"])
    assert (mom - ref_m).abs().max().item() < atol * max(1.0, ref_m.abs().max().item())
    assert (z.cpu() - ref_z).abs().max().item() < atol * max(1.0, ref_z.abs().max().item())
    z2 = enc.encode(b, px.cuda(), noise=T(d["noise"]))
    assert torch.equal(z2, z)


@pytest.mark.parametrize("dtype,atol", [(torch.float32, 2e-3), (torch.float16, 6e-2)])
def test_controlnet_into_unet(dtype, atol):
    """ControlNet encoder plan -> residuals added inside the UNet plan, vs the reference's cldm.ControlNet + UNetModel"""
    from stable_renderer_amd.unet import UNet, SD15_CFG
    from stable_renderer_amd.controlnet import ControlNet
    cfg = dict(SD15_CFG, model_channels=64, context_dim=64)
    d = np.load(os.path.join(GOLD, "controlnet_tiny.npz"))
    net = UNet(_sd("unet_tiny_keys.json", 1), cfg, dtype=dtype)
    cn = ControlNet(_sd("controlnet_tiny_keys.json", 5), cfg, dtype=dtype, strength=0.8)
    x, t, ctx, hint = T(d["x"]), T(d["t"]), T(d["ctx"]), T(d["hint"])
    B = x.shape[0]
    # the UNet plan owns x/t/ctx; the ControlNet plan reads the same buffers, so build the UNet inputs first
    from stable_renderer_amd.plan import PlanBuilder
    pb = PlanBuilder(net.device, dtype)
    x_in = pb.buf(B, 4, 16, 16, dtype=torch.float32, zero=True)
    t_in = pb.buf(B, dtype=torch.float32, zero=True)
    ctx_in = pb.buf(B, 77, 64, zero=True)
    cp = cn.build(B, 16, 16, x_in, t_in, ctx_in)
    up = net.build(B, 16, 16, control=dict(output=cp["output"], middle=cp["middle"]))
    x_in.copy_(x); t_in.copy_(t); ctx_in.copy_(ctx.to(dtype)); cp["hint"].copy_(hint)
    up["x"].copy_(x); up["t"].copy_(t); up["ctx"].copy_(ctx.to(dtype))
    cp["prologue"].run(); up["prologue"].run()
    cp["step"].run(); up["step"].run()
    torch.cuda.synchronize()
    mid = cp["middle"].float().cpu().reshape(B, 2, 2, 256).permute(0, 3, 1, 2) / 0.8
    assert (mid - T(d["mid"])).abs().max().item() < atol * max(1.0, float(T(d["mid"]).abs().max()))
    y = up["out"].cpu()
    ref = T(d["y"])
    assert (y - ref).abs().max().item() < atol * max(1.0, ref.abs().max().item()), (y - ref).abs().max()


def test_plan_side_lane_fork_join():
    """SR_OP_FORK / side-lane ops / SR_OP_JOIN: eager and captured replays give the single-lane result"""
    from stable_renderer_amd.plan import PlanBuilder
    dev = torch.device("cuda")
    pb = PlanBuilder(dev, torch.float32)
    pb.two_lanes = True
    n = 1 << 16
    g = torch.Generator().manual_seed(5)
    a, c = pb.buf(n), pb.buf(n)
    a.copy_(torch.randn(n, generator=g)); c.copy_(torch.randn(n, generator=g))
    b, d, e = pb.buf(n), pb.buf(n), pb.buf(n)
    pb.fork()
    with pb.side():
        pb.silu(a, b)
    pb.silu(c, d)
    pb.join()
    pb.add(b, d, e, s=2.0)
    plan = pb.take()
    assert [plan.ops[i].lane for i in range(plan.n)] == [0, 1, 0, 0, 0]
    ref = torch.nn.functional.silu(a) + 2.0 * torch.nn.functional.silu(c)
    plan.run()
    torch.cuda.synchronize()
    assert torch.allclose(e, ref, atol=1e-6)
    e.zero_()
    st = torch.cuda.Stream()
    plan.capture(st)
    with torch.cuda.stream(st):
        plan.launch()
    torch.cuda.synchronize()
    assert torch.allclose(e, ref, atol=1e-6)


def test_unet_tiny_with_folded_layernorm(monkeypatch):
    """SR_FOLD_LN=1: norm1/2/3 folded into their consumer GEMMs (row statistics + colsum epilogue), incl. the injected-frame
    gather of raw rows + statistics; same goldens as the explicit-LayerNorm plan"""
    from stable_renderer_amd.unet import SD15_CFG
    monkeypatch.setenv("SR_FOLD_LN", "1")
    monkeypatch.setenv("SR_LN_INLINE", "0")                    # (the row-statistics form; the in-GEMM form is the default: next test)
    cfg = dict(SD15_CFG, model_channels=64, context_dim=64)
    d = np.load(os.path.join(GOLD, "unet_tiny.npz"))
    sd = _sd("unet_tiny_keys.json", 1)
    for key, inject in (("y", None), ("y_inj", [int(v) for v in np.atleast_1d(d["inj_idx"])])):
        y, p = run_unet(sd, cfg, torch.float32, T(d["x"]), T(d["t"]), T(d["ctx"]), inject=inject)
        assert any(p["step"].ops[i].kind == 14 for i in range(p["step"].n)), "row-stats ops expected in the folded plan"
        ref = T(d[key])
        assert (y - ref).abs().max().item() < 2e-3 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dtype,atol", [(torch.float32, 2e-3), (torch.float16, 6e-2)])
def test_unet_tiny_layernorm_inside_the_consumer_gemms(dtype, atol, monkeypatch):
    """the default lowering: norm1 / norm2 / norm3 folded into to_q / attn2.to_q / ff.net.0 with the row statistics taken INSIDE
    those GEMMs (sr_igemm_args.ln_inline) -- no LayerNorm launch over the B frames (the ONE injected frame's rows are normalised by
    a LayerNorm over HW rows for its K / V projections), against the same goldens; SR_LN_INLINE=0 brings the launches back"""
    from stable_renderer_amd import _lib as L
    from stable_renderer_amd.unet import SD15_CFG
    cfg = dict(SD15_CFG, model_channels=64, context_dim=64)
    d = np.load(os.path.join(GOLD, "unet_tiny.npz"))
    sd = _sd("unet_tiny_keys.json", 1)

    def count(p, kind):
        return sum(1 for i in range(p["step"].n) if p["step"].ops[i].kind == kind)
    for key, inject in (("y", None), ("y_inj", [int(v) for v in np.atleast_1d(d["inj_idx"])])):
        monkeypatch.setenv("SR_LN_INLINE", "0")
        _, base = run_unet(sd, cfg, dtype, T(d["x"]), T(d["t"]), T(d["ctx"]), inject=inject)
        monkeypatch.setenv("SR_LN_INLINE", "1")
        y, p = run_unet(sd, cfg, dtype, T(d["x"]), T(d["t"]), T(d["ctx"]), inject=inject)
        n_blocks = count(base, L.OP_LAYERNORM) // 3
        assert count(p, L.OP_LAYERNORM) == 0
        assert count(p, L.OP_LAYERNORM_GATHER) == (n_blocks if inject is not None else 0)   # the injected frame: picked + normalised
        assert count(p, L.OP_GATHER_ROWS) == 0
        n_inline = sum(1 for i in range(p["step"].n) if p["step"].ops[i].kind == L.OP_IGEMM and p["step"].ops[i].u.igemm.ln_inline)
        assert n_inline == (3 if inject is not None else 5) * n_blocks                  # q, q2, ff1 (+ k, v without injection)
        ref = T(d[key])
        assert (y - ref).abs().max().item() < atol * max(1.0, ref.abs().max().item())
