"""GPU parity tests of the individual HIP kernels through the C ABI.  Floating-point kernels are compared with a
plain torch fp32 reference of the same op (tolerance stated per test); index/integer kernels with the oracle,
bit exact."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from stable_renderer_amd import ops as o
    return o


def rnd(seed, *shape):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g)


def tol(dtype):
    # fp16 path: fp16 operands, fp32 accumulate -> relative error ~ 2^-11 * sqrt(K) growth; fp32 path ~1e-5
    return (3e-2, 2e-2) if dtype == torch.float16 else (2e-4, 2e-4)


def close(a, b, dtype, scale=1.0):
    atol, rtol = tol(dtype)
    a, b = a.float().cpu(), b.float().cpu()
    err = (a - b).abs()
    lim = atol * scale + rtol * b.abs()
    assert bool((err <= lim).all()), f"max err {err.max().item():.4g} (ref max {b.abs().max().item():.4g})"


DTYPES = [torch.float16, torch.float32]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", ["linear_small", "linear_big", "conv3", "conv3_stride2", "conv3_up", "conv3_cat",
                                  "conv_cin4", "conv_n4", "geglu", "residual_rowvec", "transposed", "transposed_77",
                                  "splitk_residual_rowvec", "splitk_cat_up"])
def test_igemm(ops, dtype, case):
    dev = "cuda"
    ke = ops.kelems(dtype)
    B, H, W, C1, C2, N, KH, stride, up, act = 2, 8, 8, 2 * ke, 0, 96, 1, 1, 0, 0
    trans = False
    if case == "linear_small":
        B, H, W, N = 3, 1, 1, 200                          # time-embedding sized GEMM (M=3)
    elif case == "linear_big":
        B, H, W, C1, N = 2, 40, 36, 5 * ke, 320            # M=2880 -> 128x64 / 128x128 tiles, N=320
    elif case == "conv3":
        KH, N = 3, 128
    elif case == "conv3_stride2":
        KH, stride, H, W = 3, 2, 10, 12
    elif case == "conv3_up":
        KH, up, H, W = 3, 1, 6, 5
    elif case == "conv3_cat":
        KH, C2, N = 3, ke, 64
    elif case == "conv_cin4":
        KH, C1, N, H, W = 3, 4, 64, 16, 16                  # conv_in: 4 channels zero-padded to one K-step
    elif case == "conv_n4":
        KH, N, H, W = 3, 4, 16, 16                          # conv_out: 4 valid output channels
    elif case == "geglu":
        N, act = 256, 2
    elif case == "residual_rowvec":
        KH, N = 3, 160
    elif case == "splitk_residual_rowvec":                  # long K, few output tiles -> split-K + reduce epilogue
        KH, N, C1 = 3, 160, 8 * ke
    elif case == "splitk_cat_up":
        KH, N, C1, C2, up, H, W = 3, 128, 6 * ke, 2 * ke, 1, 6, 5
    elif case == "transposed":
        trans, B, H, W, N = True, 2, 8, 8, 80
    elif case == "transposed_77":
        trans, B, H, W, N = True, 2, 77, 1, 80              # cross-attention V^T from the 77-token context

    cin = C1 + C2
    x = rnd(1, B, cin, H, W)
    w = rnd(2, N, cin, KH, KH) * (cin * KH * KH) ** -0.5
    bias = rnd(3, N) * 0.1
    # reference (fp32, on the dtype-rounded operands)
    xr, wr = x.to(dtype).float(), w.to(dtype).float()
    xi = F.interpolate(xr, scale_factor=2, mode="nearest") if up else xr
    ref = F.conv2d(xi, wr, bias, stride=stride, padding=KH // 2)
    Ho, Wo = ref.shape[2:]
    rowvec = resid = None
    rv_ld = 0
    if case in ("residual_rowvec", "splitk_residual_rowvec"):
        rowvec = rnd(4, B, N)
        resid = rnd(5, B, N, Ho, Wo)
        ref = ref + rowvec[:, :, None, None] + resid.to(dtype).float()
    if act == 2:
        a, g = ref.chunk(2, dim=1)
        ref = a * F.gelu(g)
    # device inputs (NHWC, channel padded)
    c1p = (C1 + ke - 1) // ke * ke
    xa = torch.zeros(B, H, W, c1p, dtype=dtype)
    xa[..., :C1] = x[:, :C1].permute(0, 2, 3, 1).to(dtype)
    xa = xa.to(dev)
    xb = x[:, C1:].permute(0, 2, 3, 1).contiguous().to(dtype).to(dev) if C2 else None
    if C2:
        wp = ops.pack_conv_weight(w, dtype)
    else:
        wp = ops.pack_conv_weight(w, dtype, cin_pad=c1p, geglu=(act == 2))
    wp = wp.to(dev)
    bp = ops.pack_bias(bias, geglu=(act == 2)).to(dev)
    nout = N // 2 if act == 2 else N
    M = B * Ho * Wo
    if trans:
        ldt = (Ho * Wo + 7) // 8 * 8
        out = torch.zeros(B, N, ldt, dtype=dtype, device=dev)
        ops.igemm(xa, wp, out, B, H, W, c1p, N, KH=KH, stride=stride, upsample=up, bias=bp, transpose_out=1, ldt=ldt)
        torch.cuda.synchronize()
        got = out[:, :, :Ho * Wo].float().cpu()
        close(got, ref.reshape(B, N, Ho * Wo), dtype, scale=ref.abs().max().item())
        assert float(out[:, :, Ho * Wo:].abs().max().item() if ldt > Ho * Wo else 0.0) == 0.0
        return
    out = torch.zeros(M, nout, dtype=dtype, device=dev)
    rv = rowvec.to(dev) if rowvec is not None else None
    if rv is not None:                                      # a slice of a wider (batched time-embedding) buffer
        wide = torch.full((B, N + 24), 7.0, device=dev)
        wide[:, 8:8 + N] = rv
        rv, rv_ld = wide[:, 8:8 + N], N + 24
    rs = resid.permute(0, 2, 3, 1).reshape(M, N).contiguous().to(dtype).to(dev) if resid is not None else None
    ops.igemm(xa, wp, out, B, H, W, c1p, N, KH=KH, stride=stride, upsample=up, a2=xb, C2=C2, bias=bp, rowvec=rv,
              residual=rs, act=act, rowvec_ld=rv_ld)
    torch.cuda.synchronize()
    got = out.float().cpu().reshape(B, Ho, Wo, nout).permute(0, 3, 1, 2)
    close(got, ref, dtype, scale=ref.abs().max().item())


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 12, 13, 14, 15])
@pytest.mark.parametrize("act", [0, 2])
def test_igemm_forced_tiles(ops, tile, act):
    """every tile configuration the tuner may pin (sr_igemm_args.tile), incl. the 256x320 tiles with 2 x 128-byte and
    4 x 64-byte LDS stages, on a 3x3 conv with concat source, bias, time-embedding slice and residual / GEGLU"""
    dtype, dev = torch.float16, "cuda"
    B, H, W, C1, C2, N, KH = 2, 24, 20, 128, 64, 640, 3
    x = rnd(1, B, C1 + C2, H, W)
    w = rnd(2, N, C1 + C2, KH, KH) * ((C1 + C2) * 9) ** -0.5
    bias, rowvec = rnd(3, N) * 0.1, rnd(4, B, N)
    ref = F.conv2d(x.to(dtype).float(), w.to(dtype).float(), bias, padding=1) + rowvec[:, :, None, None]
    nout = N
    resid = None
    if act == 2:
        a, g = ref.chunk(2, dim=1)
        ref = a * F.gelu(g)
        nout = N // 2
    else:
        resid = rnd(5, B, N, H, W)
        ref = ref + resid.to(dtype).float()
    xa = x[:, :C1].permute(0, 2, 3, 1).contiguous().to(dtype).to(dev)
    xb = x[:, C1:].permute(0, 2, 3, 1).contiguous().to(dtype).to(dev)
    wp = ops.pack_conv_weight(w, dtype, geglu=(act == 2)).to(dev)
    bp = ops.pack_bias(bias, geglu=(act == 2)).to(dev)
    rvp = ops.pack_bias(rowvec.reshape(-1), geglu=False).reshape(B, N)
    if act == 2:                                            # the row vector follows the interleaved (value, gate) order too
        rvp = torch.stack([rowvec[:, :N // 2], rowvec[:, N // 2:]], dim=2).reshape(B, N).contiguous()
    M = B * H * W
    out = torch.zeros(M, nout, dtype=dtype, device=dev)
    rs = resid.permute(0, 2, 3, 1).reshape(M, N).contiguous().to(dtype).to(dev) if resid is not None else None
    ops.igemm(xa, wp, out, B, H, W, C1, N, KH=KH, a2=xb, C2=C2, bias=bp, rowvec=rvp.to(dev), residual=rs, act=act, tile=tile, split=-1)
    torch.cuda.synchronize()
    got = out.float().cpu().reshape(B, H, W, nout).permute(0, 3, 1, 2)
    close(got, ref, dtype, scale=ref.abs().max().item())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(13, -1, 2, 1, False), (13, -1, 7, 1, False), (13, -1, 3, 3, False), (13, -1, 9, 1, True),
                                  (14, -1, 5, 1, False), (14, 2, 8, 1, False), (14, 4, 2, 3, False), (14, 5, 3, 3, False),
                                  (15, -1, 3, 1, False), (15, 2, 10, 1, False), (15, 3, 4, 3, False)])
def test_igemm_deep_ring_tiles(ops, dtype, case):
    """tiles 13 / 14 / 15 (8 / 6 / 4 LDS stages, the whole ring requested up front): K loops shorter than the ring, exactly as
    long, and longer (steady state + counted drain), ragged M and N, bias + time-embedding slice + residual, forced split-K from
    8 K-steps on (partials + fixed-order reduce), transposed output"""
    tile, split, ksteps, KH, trans = case
    dev = "cuda"
    ke = ops.kelems(dtype)
    B, H, W, C1, N = 3, 7, 9, ksteps * ke, 200
    x = rnd(1, B, C1, H, W)
    w = rnd(2, N, C1, KH, KH) * (C1 * KH * KH) ** -0.5
    bias, rowvec, resid = rnd(3, N) * 0.1, rnd(4, B, N), rnd(5, B, N, H, W)
    ref = F.conv2d(x.to(dtype).float(), w.to(dtype).float(), bias, padding=KH // 2)
    xa = x.permute(0, 2, 3, 1).contiguous().to(dtype).to(dev)
    wp, bp = ops.pack_conv_weight(w, dtype).to(dev), ops.pack_bias(bias).to(dev)
    M = B * H * W
    if trans:
        ldt = (H * W + 7) // 8 * 8
        out = torch.zeros(B, N, ldt, dtype=dtype, device=dev)
        ops.igemm(xa, wp, out, B, H, W, C1, N, KH=KH, bias=bp, transpose_out=1, ldt=ldt, tile=tile, split=split)
        torch.cuda.synchronize()
        close(out[:, :, :H * W].float().cpu(), ref.reshape(B, N, H * W), dtype, scale=ref.abs().max().item())
        return
    ref = ref + rowvec[:, :, None, None] + resid.to(dtype).float()
    out = torch.zeros(M, N, dtype=dtype, device=dev)
    rs = resid.permute(0, 2, 3, 1).reshape(M, N).contiguous().to(dtype).to(dev)
    ops.igemm(xa, wp, out, B, H, W, C1, N, KH=KH, bias=bp, rowvec=rowvec.to(dev), residual=rs, tile=tile, split=split)
    torch.cuda.synchronize()
    close(out.float().cpu().reshape(B, H, W, N).permute(0, 3, 1, 2), ref, dtype, scale=ref.abs().max().item())
    if split > 1:                                           # the split form is bit-reproducible (fixed-order reduce, no atomics)
        again = torch.zeros_like(out)
        ops.igemm(xa, wp, again, B, H, W, C1, N, KH=KH, bias=bp, rowvec=rowvec.to(dev), residual=rs, tile=tile, split=split)
        torch.cuda.synchronize()
        assert torch.equal(out, again)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(2, 4, 40, 3, 0), (2, 8, 36, 1, 1), (3, 3, 36, 3, 3), (14, 5, 10, 3, 0), (15, 3, 12, 1, 1), (15, 3, 9, 3, 0)])
def test_igemm_split_fixup_inside_the_gemm(ops, dtype, case, monkeypatch):
    """split-K finished by the LAST workgroup of a tile to arrive (sr_igemm_args.split_counters): equal to the two-launch form
    (partials + splitk_reduce_kernel) up to the rounding of the fp16 residual add, bit-equal between launches whichever workgroup
    arrives last (200 launches, all z-slices of a tile racing), counters back at zero, and against the fp32 convolution"""
    tile, split, ksteps, KH, act = case
    dev = "cuda"
    ke = ops.kelems(dtype)
    B, H, W, C1, N = 2, 9, 11, ksteps * ke, 328
    x = rnd(1, B, C1, H, W)
    w = rnd(2, N, C1, KH, KH) * (C1 * KH * KH) ** -0.5
    bias, rowvec, resid = rnd(3, N) * 0.1, rnd(4, B, N), rnd(5, B, N, H, W)
    ref = F.conv2d(x.to(dtype).float(), w.to(dtype).float(), bias, padding=KH // 2) + rowvec[:, :, None, None]
    ref = {0: lambda v: v, 1: F.silu, 3: F.gelu}[act](ref) + resid.to(dtype).float()
    xa = x.permute(0, 2, 3, 1).contiguous().to(dtype).to(dev)
    wp, bp = ops.pack_conv_weight(w, dtype).to(dev), ops.pack_bias(bias).to(dev)
    M = B * H * W
    rs = resid.permute(0, 2, 3, 1).reshape(M, N).contiguous().to(dtype).to(dev)
    kw = dict(KH=KH, bias=bp, rowvec=rowvec.to(dev), residual=rs, act=act, tile=tile, split=split)
    cnt = ops.split_counters(xa.device)
    assert int(cnt.abs().sum()) == 0
    two = torch.zeros(M, N, dtype=dtype, device=dev)
    ops.igemm(xa, wp, two, B, H, W, C1, N, **kw)                # default: partials + splitk_reduce_kernel
    torch.cuda.synchronize()
    monkeypatch.setenv("SR_SPLIT_FIXUP", "1")
    fused = torch.zeros(M, N, dtype=dtype, device=dev)
    ops.igemm(xa, wp, fused, B, H, W, C1, N, **kw)
    torch.cuda.synchronize()
    assert int(cnt.abs().sum()) == 0, "a tile counter was left non-zero"
    close(fused.float().cpu().reshape(B, H, W, N).permute(0, 3, 1, 2), ref, dtype, scale=ref.abs().max().item())
    if dtype == torch.float32:
        assert torch.equal(fused, two)                         # same z order, same epilogue arithmetic, no intermediate rounding
    else:                                                      # fp16(fp16(v) + r) against fp16(v + r): half an ulp of v, one of the result
        d = (fused.float() - two.float()).abs()
        assert bool((d <= 2.0 ** -10 * (two.float().abs() + rs.float().abs()) + 1e-6).all())
    for _ in range(200):
        again = torch.empty_like(fused)
        ops.igemm(xa, wp, again, B, H, W, C1, N, **kw)
        assert torch.equal(again, fused)
    assert int(cnt.abs().sum()) == 0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tile", [2, 3, 4, 13, 14, 15, 9, 10])
def test_igemm_group_one_launch_for_independent_problems(ops, dtype, tile):
    """sr_igemm_group: the Q projection of B frames, the K projection of one injected frame, its V^T projection written as the
    operand-swapped row-major problem, and a 3x3 convolution with a concat source run as ONE launch -- bit-equal to the four
    launches one after another under the same tile, and right against fp32 torch; a group whose members cannot share a kernel
    (mixed tiles) falls back to single launches with the same results"""
    import ctypes as C
    from stable_renderer_amd import _lib as L
    if tile in (9, 10) and dtype != torch.float16:
        pytest.skip("fp16 tile")
    dev = "cuda"
    ke = ops.kelems(dtype)
    Cc, HW, B = 320, 192, 3                                   # (N = 320: legal for the 160- / 320-wide tiles too)
    Tk = 320 if tile in (9, 10) else HW
    xq, xs = rnd(1, B * HW, Cc), rnd(2, 384, Cc)              # tokens of B frames; the injected frame's tokens (rows padded to a tile)
    xs[Tk:] = 0
    wq, wk, wv = (rnd(3 + i, Cc, Cc) * Cc ** -0.5 for i in range(3))
    xc1, xc2, wc = rnd(7, 2, 64, 6, 5), rnd(8, 2, 128, 6, 5), rnd(9, Cc, 192, 3, 3) * (192 * 9) ** -0.5
    bias = rnd(10, Cc) * 0.1
    dq, ds = xq.to(dtype).to(dev), xs.to(dtype).to(dev)
    pq, pk, pv = (ops.pack_conv_weight(w, dtype).to(dev) for w in (wq, wk, wv))
    pc, pbias = ops.pack_conv_weight(wc, dtype).to(dev), ops.pack_bias(bias).to(dev)
    a1 = xc1.permute(0, 2, 3, 1).contiguous().to(dtype).to(dev)
    a2 = xc2.permute(0, 2, 3, 1).contiguous().to(dtype).to(dev)

    def problems():
        outs = [torch.zeros(B * HW, Cc, dtype=dtype, device=dev), torch.zeros(Tk, Cc, dtype=dtype, device=dev),
                torch.zeros(Cc, Tk, dtype=dtype, device=dev), torch.zeros(2 * 30, Cc, dtype=dtype, device=dev)]
        ars = [ops.igemm_args(dq, pq, outs[0], B * HW, 1, 1, Cc, Cc, tile=tile, split=-1),
               ops.igemm_args(ds, pk, outs[1], Tk, 1, 1, Cc, Cc, tile=tile, split=-1),
               ops.igemm_args(pv, ds, outs[2], Cc, 1, 1, Cc, Tk, tile=tile, split=-1),           # V^T[c][t] = Wv[c,:] . tokens[t,:]
               ops.igemm_args(a1, pc, outs[3], 2, 6, 5, 64, Cc, KH=3, a2=a2, C2=128, bias=pbias, act=1, tile=tile, split=-1)]
        return ars, outs
    ars, single = problems()
    for ar in ars:
        L.check(L.lib().sr_igemm(C.byref(ar), ops.stream_ptr()))
    ars_g, grouped = problems()
    ops.igemm_group(ars_g)
    torch.cuda.synchronize()
    for s_, g_ in zip(single, grouped):
        assert torch.equal(s_, g_)
    h = lambda t: t.to(dtype).float()
    close(grouped[0], h(xq) @ h(wq).T, dtype, scale=3.0)
    close(grouped[1], h(xs[:Tk]) @ h(wk).T, dtype, scale=3.0)
    close(grouped[2], h(wv) @ h(xs[:Tk]).T, dtype, scale=3.0)
    refc = F.silu(F.conv2d(torch.cat([h(xc1), h(xc2)], 1), h(wc), bias, padding=1))
    close(grouped[3].float().cpu().reshape(2, 6, 5, Cc).permute(0, 3, 1, 2), refc, dtype, scale=refc.abs().max().item())
    ars_m, mixed = problems()                                  # members that cannot share a kernel: launched one by one
    ars_m[1].tile = 4 if tile != 4 else 3
    ops.igemm_group(ars_m[:3])
    torch.cuda.synchronize()
    assert torch.equal(mixed[0], single[0]) and torch.equal(mixed[2], single[2])
    close(mixed[1], h(xs[:Tk]) @ h(wk).T, dtype, scale=3.0)


def test_cold_state_tuner_and_cache_touch(ops, monkeypatch):
    """the tile tuner's measurement state (cache flush + sr_cache_touch of the activations / residual): sr_cache_touch accepts any
    16-byte-aligned range and rejects null / unaligned pointers; a freshly tuned shape (not in any table) ends on a legal
    (tile, split) whose result equals the heuristic tile's within the GEMM tolerance"""
    import ctypes as C
    lib, st = ops.L.lib(), ops.stream_ptr()
    buf = torch.randn(1 << 20, dtype=torch.float16, device="cuda")
    assert lib.sr_cache_touch(buf.data_ptr(), buf.numel() * 2, st) == 0
    assert lib.sr_cache_touch(buf.data_ptr(), 0, st) == 0
    assert lib.sr_cache_touch(None, 64, st) != 0
    assert lib.sr_cache_touch(buf.data_ptr() + 2, 64, st) != 0
    dtype, dev = torch.float16, "cuda"
    B, H, W, C1, N = 1, 24, 24, 1280, 1280                   # M = 576: less than one workgroup per CU on every tile
    x, w = rnd(1, B, C1, H, W), rnd(2, N, C1, 3, 3) * (C1 * 9) ** -0.5
    resid = rnd(5, B, N, H, W)
    ref = F.conv2d(x.to(dtype).float(), w.to(dtype).float(), None, padding=1) + resid.to(dtype).float()
    xa = x.permute(0, 2, 3, 1).contiguous().to(dtype).to(dev)
    wp = ops.pack_conv_weight(w, dtype).to(dev)
    rs = resid.permute(0, 2, 3, 1).reshape(H * W, N).contiguous().to(dtype).to(dev)
    out = torch.zeros(H * W, N, dtype=dtype, device=dev)
    ar = ops.igemm_args(xa, wp, out, B, H, W, C1, N, KH=3, residual=rs)
    monkeypatch.setattr(ops, "_TUNE_COLD", True)
    ops.tune_igemm(ar)
    assert (ar.tile, ar.split) in ops._CANDIDATES
    assert lib.sr_igemm(C.byref(ar), st) == 0
    torch.cuda.synchronize()
    close(out.float().cpu().reshape(B, H, W, N).permute(0, 3, 1, 2), ref, dtype, scale=ref.abs().max().item())


@pytest.mark.parametrize("shape", [(2, 64, 64, 128, 320), (4, 32, 32, 192, 640), (5, 16, 16, 64, 320), (8, 8, 8, 256, 320),
                                   (1, 64, 64, 64, 320), (2, 16, 32, 128, 320)])
@pytest.mark.parametrize("act", [0, 1])
def test_igemm_patch_stationary_conv3(ops, shape, act):
    """tile 8: the patch-stationary 3x3 convolution (halo staged once per 32-channel chunk, nine taps as LDS offsets) on every
    map width it serves -- 64 (4 image rows per tile), 32 (8 rows), 16 (one image), 8 (four images) -- image borders, tile
    borders inside an image, bias + time-embedding slice + residual / SiLU, and its rejection of shapes it cannot tile"""
    dtype, dev = torch.float16, "cuda"
    B, H, W, C1, N = shape
    x = rnd(1, B, C1, H, W)
    w = rnd(2, N, C1, 3, 3) * (C1 * 9) ** -0.5
    bias, rowvec, resid = rnd(3, N) * 0.1, rnd(4, B, N), rnd(5, B, N, H, W)
    ref = F.conv2d(x.to(dtype).float(), w.to(dtype).float(), bias, padding=1) + rowvec[:, :, None, None]
    if act == 1:
        ref = F.silu(ref)
    ref = ref + resid.to(dtype).float()
    M = B * H * W
    xa = x.permute(0, 2, 3, 1).contiguous().to(dtype).to(dev)
    wp, bp = ops.pack_conv_weight(w, dtype).to(dev), ops.pack_bias(bias).to(dev)
    rs = resid.permute(0, 2, 3, 1).reshape(M, N).contiguous().to(dtype).to(dev)
    out = torch.zeros(M, N, dtype=dtype, device=dev)
    ops.igemm(xa, wp, out, B, H, W, C1, N, KH=3, bias=bp, rowvec=rowvec.contiguous().to(dev), residual=rs, act=act, tile=8, split=-1)
    torch.cuda.synchronize()
    got = out.float().cpu().reshape(B, H, W, N).permute(0, 3, 1, 2)
    close(got, ref, dtype, scale=ref.abs().max().item())
    if shape == (2, 64, 64, 128, 320) and act == 0:
        base = torch.zeros_like(out)                            # same operands through the re-staging 256x320 tile
        ops.igemm(xa, wp, base, B, H, W, C1, N, KH=3, bias=bp, rowvec=rowvec.contiguous().to(dev), residual=rs, act=act, tile=5, split=-1)
        assert float((base.float() - out.float()).abs().max()) <= 2e-2 * float(ref.abs().max())
        from stable_renderer_amd import _lib as L
        for bad in (dict(N=640, H=24, W=20), dict(N=256, H=64, W=64)):      # ragged map / N not a multiple of 320
            with pytest.raises(L.SrHipError):
                o2 = torch.zeros(2 * bad["H"] * bad["W"], bad["N"], dtype=dtype, device=dev)
                x2 = torch.zeros(2, bad["H"], bad["W"], C1, dtype=dtype, device=dev)
                w2 = ops.pack_conv_weight(torch.zeros(bad["N"], C1, 3, 3), dtype).to(dev)
                ops.igemm(x2, w2, o2, 2, bad["H"], bad["W"], C1, bad["N"], KH=3, tile=8, split=-1)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("inline", [False, True])
@pytest.mark.parametrize("mode", ["plain", "geglu", "transposed", "residual_big"])
def test_igemm_folded_layernorm(ops, dtype, mode, inline):
    """Linear(LayerNorm(x)) with the normalisation folded into the GEMM (sr_row_stats + sr_igemm_args.row_stats/colsum, or
    ln_inline: the statistics taken inside the launch): the GEMM reads the raw rows; rows with a large mean relative to their
    spread exercise the cancellation"""
    dev = "cuda"
    ke = ops.kelems(dtype)
    M, K, N = (4096 + 13, 320, 640) if mode == "residual_big" else (333, 5 * ke, 256)
    x = rnd(1, M, K) * 1.7 + rnd(2, M, 1) * 3.0              # per-row offsets up to ~2 sigma
    gamma, beta = 1 + 0.2 * rnd(3, K), 0.1 * rnd(4, K)
    w = rnd(5, N, K) * K ** -0.5
    bias = rnd(6, N) * 0.1
    xd = x.to(dtype)
    ln = F.layer_norm(xd.float(), (K,), gamma, beta, 1e-5)
    ref = ln @ w.to(dtype).float().t() + bias
    geglu = mode == "geglu"
    wp, cs, b2 = ops.fold_layernorm(w, bias, gamma, beta, dtype, geglu=geglu)
    st = ops.row_stats(xd.to(dev))
    mean, var = xd.float().mean(1), xd.float().var(1, unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    assert torch.allclose(st[:, 0].cpu(), rstd, rtol=2e-4) and torch.allclose(st[:, 1].cpu(), -rstd * mean, rtol=2e-4, atol=1e-4)
    kw = dict(bias=b2.to(dev), row_stats=st, colsum=cs.to(dev))
    if inline:
        kw = dict(bias=b2.to(dev), ln_inline=True, ln_eps=1e-5, colsum=cs.to(dev))
    if mode == "transposed":
        ldt = (M + 7) // 8 * 8
        out = torch.zeros(1, N, ldt, dtype=dtype, device=dev)
        ops.igemm(xd.to(dev), wp.to(dev), out, 1, M, 1, K, N, transpose_out=1, ldt=ldt, **kw)
        got = out[0, :, :M].t()
    else:
        nout = N // 2 if geglu else N
        if geglu:
            a, g = ref.chunk(2, dim=1)
            ref = a * F.gelu(g)
        rs = None
        if mode == "residual_big":
            resid = rnd(7, M, N)
            ref = ref + resid.to(dtype).float()
            rs = resid.to(dtype).to(dev)
        out = torch.zeros(M, nout, dtype=dtype, device=dev)
        ops.igemm(xd.to(dev), wp.to(dev), out, M, 1, 1, K, N, act=2 if geglu else 0, residual=rs, **kw)
        got = out
    torch.cuda.synchronize()
    # the reference rounds LN(x) to `dtype` before the GEMM, the folded form does not: same tolerance class as the GEMM itself
    close(got, ref, dtype, scale=ref.abs().max().item())


@pytest.mark.parametrize("tile", [2, 3, 4, 5, 7, 9, 10, 11, 12, 13, 14, 15])
@pytest.mark.parametrize("geglu", [False, True])
def test_igemm_inline_layernorm_tiles(ops, tile, geglu):
    """every tile that has a variant taking the LayerNorm statistics inside the launch (ln_inline), at the UNet's three widths
    (K = 320 / 640 / 1280: 5, 10, 20 K-steps), ragged M, residual or GEGLU; the tiles without one refuse the flag"""
    dtype, dev = torch.float16, "cuda"
    for K in (320, 640, 1280):
        M, N = 1000, 640
        x = rnd(1, M, K) * 1.3 + rnd(2, M, 1) * 2.0
        gamma, beta = 1 + 0.2 * rnd(3, K), 0.1 * rnd(4, K)
        w, bias = rnd(5, N, K) * K ** -0.5, rnd(6, N) * 0.1
        xd = x.to(dtype)
        ref = F.layer_norm(xd.float(), (K,), gamma, beta, 1e-5) @ w.to(dtype).float().t() + bias
        wp, cs, b2 = ops.fold_layernorm(w, bias, gamma, beta, dtype, geglu=geglu)
        rs = None
        if geglu:
            a, g = ref.chunk(2, dim=1)
            ref = a * F.gelu(g)
        else:
            resid = rnd(7, M, N)
            ref = ref + resid.to(dtype).float()
            rs = resid.to(dtype).to(dev)
        out = torch.zeros(M, N // 2 if geglu else N, dtype=dtype, device=dev)
        ops.igemm(xd.to(dev), wp.to(dev), out, M, 1, 1, K, N, act=2 if geglu else 0, residual=rs, bias=b2.to(dev), colsum=cs.to(dev),
                  ln_inline=True, tile=tile, split=-1)
        torch.cuda.synchronize()
        close(out, ref, dtype, scale=ref.abs().max().item())
    from stable_renderer_amd import _lib as L
    for bad in (1, 6):
        with pytest.raises(L.SrHipError):
            ops.igemm(xd.to(dev), wp.to(dev), out, M, 1, 1, K, N, act=2 if geglu else 0, bias=b2.to(dev), colsum=cs.to(dev), ln_inline=True, tile=bad, split=-1)


@pytest.mark.parametrize("dtype", DTYPES)
def test_igemm_out_f32_and_scale(ops, dtype):
    ke = ops.kelems(dtype)
    B, T, Cc, N = 2, 50, 2 * ke, 72
    x = rnd(1, B * T, Cc)
    w = rnd(2, N, Cc) * Cc ** -0.5
    out = torch.zeros(B * T, N, dtype=torch.float32, device="cuda")
    ops.igemm(x.to(dtype).cuda(), ops.pack_conv_weight(w, dtype).cuda(), out, B * T, 1, 1, Cc, N, out_f32=1, scale=0.5)
    torch.cuda.synchronize()
    ref = 0.5 * (x.to(dtype).float() @ w.to(dtype).float().t())
    close(out, ref, torch.float32 if dtype == torch.float32 else torch.float16, scale=ref.abs().max().item())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [(2, 70, 320, 0, True), (1, 256, 64, 64, False), (2, 33, 2560, 0, True), (1, 4096, 128, 0, True),
                                 # UNet shapes of the single-launch path (register-resident slab), incl. concat sources whose
                                 # boundary falls inside a group, and the two-pass fallback (64x64 maps)
                                 (2, 64, 1280, 0, True), (2, 64, 1280, 1280, True), (1, 256, 1280, 640, True),
                                 (1, 1024, 640, 0, False), (1, 1024, 1280, 640, True), (1, 1024, 640, 320, True),
                                 (1, 4096, 320, 0, True), (1, 4096, 640, 320, True), (1, 100, 320, 0, False)])
def test_groupnorm(ops, dtype, cfg):
    B, HW, C1, C2, silu = cfg
    x = rnd(1, B, HW, C1 + C2) * 1.5 + 0.3
    gamma, beta = 1 + 0.1 * rnd(2, C1 + C2), 0.1 * rnd(3, C1 + C2)
    xd = x.to(dtype)
    ref = F.group_norm(xd.float().permute(0, 2, 1), 32, gamma, beta, 1e-5).permute(0, 2, 1)
    if silu:
        ref = F.silu(ref)
    x1 = xd[..., :C1].contiguous().cuda()
    x2 = xd[..., C1:].contiguous().cuda() if C2 else None
    y = ops.groupnorm(x1, gamma.cuda(), beta.cuda(), B, HW, C1, x2=x2, C2=C2, eps=1e-5, silu=silu)
    torch.cuda.synchronize()
    a, r = (4e-3, 4e-3) if dtype == torch.float16 else (1e-5, 1e-5)
    assert torch.allclose(y.float().cpu(), ref, atol=a, rtol=r), (y.float().cpu() - ref).abs().max()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("Cc", [64, 128, 320, 640, 1280])
def test_layernorm(ops, dtype, Cc):
    x = rnd(1, 3, 37, Cc) * 2 + 0.5
    gamma, beta = 1 + 0.1 * rnd(2, Cc), 0.1 * rnd(3, Cc)
    xd = x.to(dtype)
    ref = F.layer_norm(xd.float(), (Cc,), gamma, beta, 1e-5)
    y = ops.layernorm(xd.cuda(), gamma.cuda(), beta.cuda())
    torch.cuda.synchronize()
    a, r = (4e-3, 4e-3) if dtype == torch.float16 else (1e-5, 1e-5)
    assert torch.allclose(y.float().cpu(), ref, atol=a, rtol=r)


def _attn_ref(q, k, v, heads):
    B, Tq, Cc = q.shape
    d = Cc // heads
    qh = q.view(B, Tq, heads, d).transpose(1, 2)
    kh = k.view(k.shape[0], -1, heads, d).transpose(1, 2)
    vh = v.view(v.shape[0], -1, heads, d).transpose(1, 2)
    s = (qh @ kh.transpose(-1, -2)) * d ** -0.5
    return (s.softmax(-1) @ vh).transpose(1, 2).reshape(B, Tq, Cc)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [
    # B, Bk, Tq, Tk, heads, d
    (2, 2, 256, 256, 8, 40),      # SD1.5 64x64-level shape class (d=40)
    (2, 2, 100, 77, 8, 40),       # cross attention: ragged Tq, 77 keys (masked tail)
    (3, 1, 128, 192, 8, 40),      # shared K/V (OverlapCorresponder injection)
    (1, 1, 64, 64, 8, 80),
    (1, 1, 64, 130, 8, 160),
    (2, 2, 96, 64, 8, 8),
    (1, 1, 80, 64, 4, 16),
    (1, 1, 80, 70, 2, 32),
    (1, 1, 64, 64, 2, 64),
    # the resident-K/V walk over several query blocks per workgroup (fp16, Tq >= 512, Tk <= 128):
    (2, 2, 4096, 77, 8, 40),      # 64x64-level prompt attention: 32 query blocks, 8 per workgroup
    (3, 3, 1000, 77, 4, 40),      # ragged Tq: the last block is partial, the last walk shorter
    (2, 1, 640, 128, 4, 80),      # two full key tiles, shared K/V, d = 80
    (2, 2, 1024, 64, 2, 48),      # a single key tile, d a multiple of 16 (no ones row)
    (1, 1, 512, 100, 2, 160),     # d = 160
    # long key sequences at d = 40 (fp16: the 32x32-MFMA occupancy kernel; fp32: the simple loop): ragged queries AND a ragged
    # last key tile (1000 = 15 x 64 + 40), shared and per-entry K/V
    (2, 1, 600, 1000, 8, 40),
    (2, 2, 320, 1000, 4, 40),
    (1, 1, 256, 4160, 2, 40),     # 65 key tiles: odd tile count through the double buffer
    (2, 1, 600, 1000, 4, 80),     # wider heads, long ragged key sequences (d = 80: SD1.5 32x32 level; d = 64: SDXL)
    (2, 2, 320, 1024, 4, 64),
    (1, 1, 200, 1500, 2, 64),
])
def test_attention(ops, dtype, cfg):
    B, Bk, Tq, Tk, heads, d = cfg
    Cc = heads * d
    q, k, v = rnd(1, B, Tq, Cc), rnd(2, Bk, Tk, Cc), rnd(3, Bk, Tk, Cc)
    qd, kd, vd = q.to(dtype), k.to(dtype), v.to(dtype)
    ref = _attn_ref(qd.float(), kd.float().expand(B, -1, -1) if Bk == 1 else kd.float(),
                    vd.float().expand(B, -1, -1) if Bk == 1 else vd.float(), heads)
    ldt = (Tk + 7) // 8 * 8
    vt = torch.zeros(Bk, heads, d, ldt, dtype=dtype)
    vt[..., :Tk] = vd.view(Bk, Tk, heads, d).permute(0, 2, 3, 1)
    o = ops.attention(qd.cuda(), kd.cuda(), vt.cuda(), heads, Tk=Tk)
    torch.cuda.synchronize()
    a, r = (6e-3, 2e-2) if dtype == torch.float16 else (2e-5, 1e-4)
    err = (o.float().cpu() - ref).abs().max().item()
    assert torch.allclose(o.float().cpu(), ref, atol=a, rtol=r), err


def test_attention_softmax_spike(ops):
    """online-softmax rescale path: one key dominates late in the sequence (rule: force the rare branch)."""
    dtype = torch.float32
    B, T, heads, d = 1, 256, 8, 40
    q, k, v = rnd(1, B, T, heads * d), rnd(2, B, T, heads * d), rnd(3, B, T, heads * d)
    k[:, 200] = q[:, 17] * 6.0
    ref = _attn_ref(q, k, v, heads)
    vt = v.view(B, T, heads, d).permute(0, 2, 3, 1).contiguous()
    o = ops.attention(q.cuda(), k.cuda(), vt.cuda(), heads)
    assert torch.allclose(o.cpu(), ref, atol=5e-5, rtol=1e-4)


@pytest.mark.parametrize("case", ["late_spike", "low_first_tile", "rising", "injected_bk1", "d48"])
def test_attention_lazy_shift_paths(ops, case):
    """the pipelined fp16 kernel keeps the softmax shift inside the QK^T MFMA and moves it lazily: force every branch -- a key that
    tops the running shift by far more than 2^8 late in the sequence, a first tile far BELOW the later ones, maxima that creep up
    tile after tile, one K/V for the whole batch (Bk = 1), and d = 48 (no spare V^T row: VALU row sum)"""
    dtype = torch.float16
    B, T, heads, d = 2, 1024, 8, 48 if case == "d48" else 40
    q, k, v = rnd(1, B, T, heads * d), rnd(2, B, T, heads * d), rnd(3, B, T, heads * d)
    if case == "late_spike":
        k[:, 900] = q[:, 17] * 4.0
        k[:, 333] = q[:, 600] * 3.0
    elif case == "low_first_tile":
        k[:, :64] = -q[:, :64] * 2.0                       # the first tile's scores sit far below the rest for these queries
    elif case == "rising":
        k = k * torch.linspace(0.2, 3.0, T).view(1, T, 1)
    Bk = 1 if case == "injected_bk1" else B
    qd, kd, vd = q.to(dtype), k[:Bk].to(dtype), v[:Bk].to(dtype)
    ref = _attn_ref(qd.float(), kd.float().expand(B, -1, -1), vd.float().expand(B, -1, -1), heads)
    vt = vd.view(Bk, T, heads, d).permute(0, 2, 3, 1).contiguous()
    o = ops.attention(qd.cuda(), kd.cuda(), vt.cuda(), heads)
    torch.cuda.synchronize()
    assert torch.isfinite(o).all()
    err = (o.float().cpu() - ref).abs().max().item()
    assert torch.allclose(o.float().cpu(), ref, atol=6e-3, rtol=2e-2), (case, err)


@pytest.mark.parametrize("dtype", DTYPES)
def test_layout_and_embedding(ops, dtype):
    x = rnd(1, 2, 4, 6, 5)
    pbs = torch.tensor([0.5, 2.0])
    y = ops.nchw_to_nhwc(x.cuda(), dtype, cpad=8, scale=3.0, per_batch_scale=pbs.cuda())
    ref = torch.zeros(2, 30, 8)
    ref[..., :4] = (x * 3.0 * pbs[:, None, None, None]).permute(0, 2, 3, 1).reshape(2, 30, 4)
    assert torch.allclose(y.float().cpu(), ref.to(dtype).float(), atol=1e-6)
    back = ops.nhwc_to_nchw(y, 2, 4, 6, 5, ldc=8)
    assert torch.allclose(back.cpu(), ref[..., :4].to(dtype).float().reshape(2, 6, 5, 4).permute(0, 3, 1, 2))
    import sr_oracle as O
    t = torch.tensor([0.0, 17.0, 500.0, 999.0])
    e = ops.timestep_embedding(t.cuda(), 320, dtype)
    a = 2e-3 if dtype == torch.float16 else 2e-4      # GPU sinf/cosf of arguments up to ~1e3
    assert torch.allclose(e.float().cpu(), O.timestep_embedding(t, 320), atol=a)


# ------------------------------------------------------------------------------------------------------
def test_idmap_masks_and_overlap_vs_oracle_and_golden(ops):
    import sr_oracle as O
    d = np.load(os.path.join(GOLD, "overlap_step.npz"))
    meta = json.loads(bytes(d["meta"]).decode())
    for name, m in meta.items():
        ids = torch.from_numpy(d[f"{name}_ids"]).cuda()
        x = torch.from_numpy(d[f"{name}_x"])
        assert np.array_equal(ops.idmap_masks(ids).cpu().numpy(), O.idmap_masks(d[f"{name}_ids"]))   # bit exact
        if m["timestep"] < m["stop"]:
            continue
        idx = ops.OverlapIndex(ids, x.shape[2], x.shape[3])
        # integer structure: bit exact against the oracle's vertex_screen_info
        vsi = O.vertex_screen_info(d[f"{name}_ids"])
        assert idx.n_valid == len(vsi)
        n, c, h, w = x.shape
        sx = (vsi[:, 4] * np.float32(w)).astype(np.int32)
        sy = (vsi[:, 5] * np.float32(h)).astype(np.int32)
        cell = (vsi[:, 6].astype(np.int32) * h + sy) * w + sx
        exp_vid = np.full(n * h * w, -1, np.int32)
        exp_vid[cell] = vsi[:, 3].astype(np.int32)               # sequential: last row wins
        assert np.array_equal(idx.cell_vid.cpu().numpy(), exp_vid), name
        xg = x.clone().cuda()
        idx.step(xg, m["ratio"])
        out = xg.cpu()
        ref = torch.from_numpy(d[f"{name}_out"])                 # the reference's own output
        assert torch.allclose(out, ref, atol=1e-5, rtol=1e-5), (name, (out - ref).abs().max())
        assert torch.allclose(out, O.overlap_step(x, d[f"{name}_ids"], m["ratio"]), atol=1e-5, rtol=1e-5)


def test_idmap_and_overlap_index_on_reference_sphere_ids(ops):
    """IDMap masks + the overlap index on two frames of the id maps the reference itself dumped
    (resources/example-sphere-and-object-views/sphere/id; golden = the reference's IDMap.masks / create_vertex_screen_info)"""
    d = np.load(os.path.join(GOLD, "idmap.npz"))
    ids = torch.from_numpy(d["sphere_ids"]).cuda()
    assert np.array_equal(ops.idmap_masks(ids).cpu().numpy(), d["sphere_masks"])
    vsi = d["sphere_vsi"]
    n, H, W = ids.shape[:3]
    h, w = H // 8, W // 8
    idx = ops.OverlapIndex(ids, h, w)
    assert idx.n_valid == len(vsi)
    sx = (vsi[:, 4] * np.float32(w)).astype(np.int32)
    sy = (vsi[:, 5] * np.float32(h)).astype(np.int32)
    cell = (vsi[:, 6].astype(np.int32) * h + sy) * w + sx
    exp_vid = np.full(n * h * w, -1, np.int32)
    exp_vid[cell] = vsi[:, 3].astype(np.int32)                   # sequential: last row wins
    assert np.array_equal(idx.cell_vid.cpu().numpy(), exp_vid)
    # CSR: segment of vertex v = the cells of the vsi rows carrying v (as a multiset)
    off, ent = idx.vid_off.cpu().numpy(), idx.entries.cpu().numpy()
    vids = vsi[:, 3].astype(np.int64)
    assert np.array_equal(np.diff(off), np.bincount(vids, minlength=idx.cap))
    order = np.argsort(vids, kind="stable")
    exp_sorted = np.concatenate([np.sort(c) for c in np.split(cell[order], np.cumsum(np.bincount(vids, minlength=idx.cap))[:-1])])
    got_sorted = np.concatenate([np.sort(ent[off[v]:off[v + 1]]) for v in range(idx.cap)]) if idx.cap < 5000 else None
    if got_sorted is not None:
        assert np.array_equal(got_sorted, exp_sorted)
    else:                                                        # large vertex ranges: check per-vertex sums of cells instead
        s_exp = np.bincount(vids, weights=cell.astype(np.float64), minlength=idx.cap)
        seg = np.repeat(np.arange(idx.cap), np.diff(off))
        s_got = np.bincount(seg, weights=ent[:off[-1]].astype(np.float64), minlength=idx.cap)
        assert np.array_equal(s_exp, s_got)


def test_overlap_step_propagates_non_finite_latents(ops):
    """a NaN / infinity in one view must reach the mean of every view sharing the vertex (the reference's float mean does,
    corresponder.py:339-369) instead of being clamped into a finite value by the fixed-point sum"""
    d = np.load(os.path.join(GOLD, "overlap_step.npz"))
    meta = json.loads(bytes(d["meta"]).decode())
    name = next(n for n, m in meta.items() if m["timestep"] >= m["stop"] and d[f"{n}_x"].shape[0] >= 2)
    ids = torch.from_numpy(d[f"{name}_ids"]).cuda()
    x = torch.from_numpy(d[f"{name}_x"]).clone()
    idx = ops.OverlapIndex(ids, x.shape[2], x.shape[3])
    off, ent, cv = idx.vid_off.cpu().numpy(), idx.entries.cpu().numpy(), idx.cell_vid.cpu().numpy()
    lhw = x.shape[2] * x.shape[3]
    # a vertex seen from two different views whose cells both have it as their winning vertex
    v = next(v for v in range(idx.cap) if off[v + 1] - off[v] >= 2 and len({c // lhw for c in ent[off[v]:off[v + 1]]}) >= 2
             and sum(cv[c] == v for c in set(ent[off[v]:off[v + 1]])) >= 2)
    cells = sorted({int(c) for c in ent[off[v]:off[v + 1]] if cv[c] == v})
    src = cells[0]
    for bad in (float("nan"), float("inf")):
        xb = x.clone()
        xb.view(x.shape[0], x.shape[1], lhw)[src // lhw, 1, src % lhw] = bad
        bl = torch.empty_like(xb).cuda()
        idx.step(xb.cuda(), meta[name]["ratio"], blended_out=bl)
        b = bl.cpu().view(x.shape[0], x.shape[1], lhw)
        for c in cells:                                      # channel 1 of every cell of that vertex, in every view
            assert not np.isfinite(float(b[c // lhw, 1, c % lhw]))
            assert np.isfinite(float(b[c // lhw, 0, c % lhw]))   # other channels untouched


def test_overlap_nonsquare_raises(ops):
    ids = torch.ones(1, 48, 32, 4, dtype=torch.int32).cuda()
    with pytest.raises(IndexError):
        ops.OverlapIndex(ids, 6, 4)


def test_adain_and_noise_pool_vs_golden(ops):
    d = np.load(os.path.join(GOLD, "adain.npz"))
    for s in range(2):
        o = ops.adain_nchw(torch.from_numpy(d[f"nchw_c{s}"]).cuda(), torch.from_numpy(d[f"nchw_s{s}"]).cuda())
        assert torch.allclose(o.cpu(), torch.from_numpy(d[f"nchw_o{s}"]), atol=2e-6, rtol=1e-5)
    d = np.load(os.path.join(GOLD, "noise_pool.npz"))
    for i in range(2):
        pooled, out = ops.noise_pool(torch.from_numpy(d[f"n{i}_noise"]).cuda(), torch.from_numpy(d[f"n{i}_alpha"]).cuda(),
                                     torch.from_numpy(d[f"n{i}_bg"]).cuda())
        assert torch.allclose(pooled.cpu(), torch.from_numpy(d[f"n{i}_pooled"]), atol=1e-6, rtol=1e-6)
        # style statistics are fp16-rounded in the reference: one fp16 ulp of std/mean = 1e-3 relative
        assert torch.allclose(out.cpu(), torch.from_numpy(d[f"n{i}_out"]), atol=3e-3, rtol=2e-3)


def test_sampler_math(ops):
    import sr_oracle as O
    x = rnd(1, 3, 4, 8, 8).cuda()
    n = x.numel()
    xin = torch.empty(2 * n, device="cuda")
    ops.eps_scale_input(x, xin, 2, 3.0)
    ref = O.eps_input(x.cpu(), torch.full((3,), 3.0))
    assert torch.allclose(xin[:n].cpu().view_as(ref), ref, rtol=1e-6) and torch.equal(xin[:n], xin[n:])
    eps = rnd(2, 2 * n).cuda()
    den, dd = torch.empty_like(x), torch.empty_like(x)
    ops.cfg_denoise(x, eps, den, dd, 2, 3.0, 7.5)
    xc = x.cpu().reshape(-1)
    u, c = xc - eps[:n].cpu() * 3.0, xc - eps[n:].cpu() * 3.0
    rden = u + (c - u) * 7.5
    assert torch.allclose(den.cpu().reshape(-1), rden, rtol=1e-5, atol=1e-5)
    assert torch.allclose(dd.cpu().reshape(-1), (xc - rden) / 3.0, rtol=1e-5, atol=1e-5)
    # ddpm / euler / lcm single steps against the oracle loop with a frozen "model"
    sig = torch.tensor([3.0, 1.2, 0.0])
    for sampler in ("euler", "ddpm", "lcm"):
        x0 = rnd(5, 2, 4, 8, 8)
        fixed_den = rnd(6, 2, 4, 8, 8) * 0.3
        torch.manual_seed(9)
        ref = O.sample_loop(lambda xx, s: fixed_den, x0.clone(), sig, sampler)
        torch.manual_seed(9)
        xg, dg = x0.clone().cuda(), fixed_den.cuda()
        for i in range(2):
            s, sn = float(sig[i]), float(sig[i + 1])
            if sampler == "euler":
                dd = (xg - dg) / s
                ops.euler_step(xg, dd, sn - s)
            elif sampler == "ddpm":
                nz = torch.randn_like(x0).cuda() if sn > 0 else None
                ops.ddpm_step(xg, dg, nz, s, sn)
            else:
                nz = torch.randn_like(x0).cuda() if sn > 0 else None
                ops.lcm_step(xg, dg, nz, sn)
        assert torch.allclose(xg.cpu(), ref, atol=2e-5, rtol=1e-5), sampler


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("Cc", [320, 640, 1280, 96])
def test_layernorm_gather_picks_and_normalises_in_one_launch(ops, dtype, Cc):
    """sr_layernorm_gather = sr_gather_rows + sr_layernorm bit for bit (two injected frames, device indices), and an index outside
    the batch gives zero rows and raises the device flag"""
    import ctypes as C
    from stable_renderer_amd import _lib as L
    B, HW = 5, 37
    x = (rnd(1, B, HW, Cc) * 1.4 + rnd(2, B, HW, 1)).to(dtype).cuda()
    gamma, beta = (1 + 0.1 * rnd(3, Cc)).cuda(), (0.1 * rnd(4, Cc)).cuda()
    sel = torch.tensor([3, 1], dtype=torch.int32, device="cuda")
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    y = torch.empty(2, HW, Cc, dtype=dtype, device="cuda")
    L.check(L.lib().sr_layernorm_gather(ops._p(x), ops._p(sel), 2, HW, B, ops._p(err), ops._p(gamma), ops._p(beta), ops._p(y), Cc, 1e-5,
                                        ops.DT[dtype], ops.stream_ptr()))
    ref = ops.layernorm(x[[3, 1]].contiguous(), gamma, beta)
    torch.cuda.synchronize()
    assert torch.equal(y, ref) and int(err.item()) == 0
    close(y, F.layer_norm(x[[3, 1]].float().cpu(), (Cc,), gamma.cpu(), beta.cpu(), 1e-5), dtype, scale=3.0)
    sel.copy_(torch.tensor([3, 7], dtype=torch.int32))
    L.check(L.lib().sr_layernorm_gather(ops._p(x), ops._p(sel), 2, HW, B, ops._p(err), ops._p(gamma), ops._p(beta), ops._p(y), Cc, 1e-5,
                                        ops.DT[dtype], ops.stream_ptr()))
    torch.cuda.synchronize()
    assert int(err.item()) == 1 and float(y[1].abs().max()) == 0.0 and torch.equal(y[0], ref[0])


def test_gather_rows_rejects_out_of_range_index(ops):
    """sr_gather_rows never dereferences a device index outside [0, n_rows): the row is zero-filled and the sticky device flag is
    raised (round 1 recorded a GPU memory fault from exactly this: a global batch index selecting a row of a rank-local batch)"""
    import ctypes as C
    from stable_renderer_amd import _lib as L
    x = rnd(1, 4, 64).cuda()
    sel = torch.tensor([2, 7, -1, 0, 1 << 30], dtype=torch.int32).cuda()
    y = torch.full((5, 64), 9.0).cuda()
    err = torch.zeros(1, dtype=torch.int32).cuda()
    p = lambda t: C.c_void_p(t.data_ptr())
    L.check(L.lib().sr_gather_rows(p(x), p(sel), p(y), 5, 4, 64 * 4, p(err), ops.stream_ptr()))
    torch.cuda.synchronize()
    assert int(err.item()) == 1
    assert torch.equal(y[0], x[2]) and torch.equal(y[3], x[0])
    assert not y[1].any() and not y[2].any() and not y[4].any()
    err.zero_()
    L.check(L.lib().sr_gather_rows(p(x), p(sel[3:4]), p(y), 1, 4, 64 * 4, p(err), ops.stream_ptr()))
    assert int(err.item()) == 0
    assert L.lib().sr_gather_rows(p(x), p(sel), p(y), 5, 0, 64 * 4, p(err), ops.stream_ptr()) != 0       # n_rows is mandatory


def test_sampler_raises_when_the_device_side_injected_index_is_bad():
    """a bad index written to the plan's DEVICE selector (what the host range check cannot see) surfaces as IndexError, no fault"""
    from stable_renderer_amd import synth
    from stable_renderer_amd.model_shapes import unet_names_shapes
    from stable_renderer_amd.sampling import DiffusionRunner
    from stable_renderer_amd.unet import UNet, SD15_CFG
    cfg = dict(SD15_CFG, model_channels=64, context_dim=64)
    ns, norms = unet_names_shapes(cfg)
    net = UNet(synth.synth_state_dict(ns, seed=1, norm_names=norms), cfg, dtype=torch.float32)
    r = DiffusionRunner(net, 2, 8, 8, 5.0, use_graph=False)
    g = torch.Generator().manual_seed(0)
    r.set_conditioning(torch.randn(1, 77, 64, generator=g), torch.randn(1, 77, 64, generator=g))
    noise = torch.randn(2, 4, 8, 8, generator=g)
    out, inj = r.sample(noise, 2, "ddim", "normal", inject_n_rand=1)
    assert torch.isfinite(out).all()
    orig = r._ensure_plan

    def poisoned(inject):
        p = orig(inject)
        p["inject"].fill_(1000)                      # behind the host check's back
        return p
    r._ensure_plan = poisoned
    with pytest.raises(IndexError):
        r.sample(noise, 2, "ddim", "normal", inject_n_rand=1)
    r._ensure_plan = orig
    out2, _ = r.sample(noise, 2, "ddim", "normal", inject_n_rand=1)       # the flag was cleared: the runner stays usable
    assert torch.isfinite(out2).all()


@pytest.mark.gpu
def test_stream_ptr_follows_the_current_stream_of_the_thread(ops):
    """ops.stream_ptr() (torch's raw-stream binding) is the stream `with torch.cuda.stream(...)` made current, per thread"""
    import threading
    O = ops
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    assert (O.stream_ptr().value or 0) == torch.cuda.current_stream().cuda_stream
    with torch.cuda.stream(s1):
        assert O.stream_ptr().value == s1.cuda_stream
        seen = []

        def other():
            with torch.cuda.stream(s2):
                seen.append(O.stream_ptr().value)
        t = threading.Thread(target=other)
        t.start()
        t.join()
        assert seen == [s2.cuda_stream]
        assert O.stream_ptr().value == s1.cuda_stream


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(2, 0, 40, 3), (2, 4, 40, 3), (3, -1, 6, 1), (4, -1, 5, 3), (13, -1, 9, 1), (14, 3, 12, 3), (15, 3, 12, 1),
                                  (9, -1, 5, 1), (7, -1, 5, 1), (0, 0, 10, 3)])
def test_igemm_tile_order_columns_first_is_the_same_gemm(ops, dtype, case):
    """sr_igemm_args.tile_order = 1 only changes which tile a workgroup takes (bands of N-tiles per XCD instead of bands of M-tiles):
    bit-equal outputs for unsplit, split-K (partials + reduce) and the wide tiles, ragged M and N; any other value is refused"""
    tile, split, ksteps, KH = case
    if dtype == torch.float32 and tile in (7, 9):
        pytest.skip("fp16-only tile")
    dev = "cuda"
    ke = ops.kelems(dtype)
    B, H, W, C1, N = 3, 13, 11, ksteps * ke, 640 if tile in (7, 9) else 328
    x = rnd(11, B, C1, H, W)
    w = rnd(12, N, C1, KH, KH) * (C1 * KH * KH) ** -0.5
    bias, resid = rnd(13, N) * 0.1, rnd(15, B, N, H, W)
    ref = F.silu(F.conv2d(x.to(dtype).float(), w.to(dtype).float(), bias, padding=KH // 2)) + resid.to(dtype).float()
    xa = x.permute(0, 2, 3, 1).contiguous().to(dtype).to(dev)
    wp, bp = ops.pack_conv_weight(w, dtype).to(dev), ops.pack_bias(bias).to(dev)
    M = B * H * W
    rs = resid.permute(0, 2, 3, 1).reshape(M, N).contiguous().to(dtype).to(dev)
    kw = dict(KH=KH, bias=bp, residual=rs, act=1, tile=tile, split=split)
    rows = torch.zeros(M, N, dtype=dtype, device=dev)
    ops.igemm(xa, wp, rows, B, H, W, C1, N, **kw)
    cols = torch.zeros(M, N, dtype=dtype, device=dev)
    ops.igemm(xa, wp, cols, B, H, W, C1, N, tile_order=1, **kw)
    torch.cuda.synchronize()
    close(cols.float().cpu().reshape(B, H, W, N).permute(0, 3, 1, 2), ref, dtype, scale=ref.abs().max().item())
    assert torch.equal(rows, cols)
    with pytest.raises(Exception, match="tile_order"):
        ops.igemm(xa, wp, cols, B, H, W, C1, N, tile_order=2, **kw)
