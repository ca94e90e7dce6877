"""TEST INFRASTRUCTURE (container only): generate golden vectors by running the *reference* code on CPU.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py [section ...]

Outputs small ``.npz`` fixtures under ``tests/golden/``.  The reference sources are imported from
``/root/reference`` through ``oracle/_ref_import.py`` (stubbed non-arithmetic deps); the
arithmetic (torch / einops / scipy) is real, so these vectors are authoritative reference outputs.
Nothing here runs on the GPU box.  Every fixture stores inputs (or the seeds that make them) + outputs.
"""
import ast
import contextlib
import io
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)

import _ref_import as R  # noqa: E402

_ARGV = sys.argv[1:]
R.install()
from stable_renderer_amd import synth  # noqa: E402


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    p = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(p, **out)
    print("wrote", p, {k: (v.shape, str(v.dtype)) for k, v in out.items()})


def _jdump(obj, path):
    with open(path, "w") as f:
        json.dump(obj, f)


@contextlib.contextmanager
def quiet():
    """the reference prints whole tensors from step_finished / create_vertex_screen_info."""
    with contextlib.redirect_stdout(io.StringIO()):
        yield


@contextlib.contextmanager
def one_thread():
    """torch's CPU index_put_ with duplicate targets is racy across threads (a chunk boundary row wins);
    single-threaded it is sequential = LAST ROW WINS.  Parity is defined on the sequential semantics, so
    every reference call that scatters with duplicates runs under this guard."""
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        yield
    finally:
        torch.set_num_threads(n)


def rnd(seed, *shape, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float32).to(dtype)


def synth_ids(seed, n, h, w, n_vertex=40, sprite=1, material=1, k=3, frac_bg=0.25, frac_nonai=0.1):
    """synthetic id maps (N,H,W,4) int32: (sprite, material, map_index, vertexID); background all-zero,
    some non-AI pixels with map_index 2048."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.zeros(n, h, w, 4, dtype=torch.int32)
    ids[..., 0] = sprite
    ids[..., 1] = material
    ids[..., 2] = torch.randint(0, k * k, (n, h, w), generator=g, dtype=torch.int32)
    ids[..., 3] = torch.randint(0, n_vertex, (n, h, w), generator=g, dtype=torch.int32)
    u = torch.rand(n, h, w, generator=g)
    bg = u < frac_bg
    nonai = (u >= frac_bg) & (u < frac_bg + frac_nonai)
    ids[bg] = 0
    ids[..., 2][nonai] = 2048
    return ids


# ------------------------------------------------------------------------------------------------
def sec_math():
    import common_utils.math_utils as mu
    out = {}
    for seed in range(2):
        c = rnd(seed, 2, 4, 16, 16) * 1.7 + 0.3
        s = rnd(seed + 10, 2, 4, 16, 16) * 0.6 - 1.1
        out[f"nchw_c{seed}"] = c
        out[f"nchw_s{seed}"] = s
        out[f"nchw_o{seed}"] = mu.adaptive_instance_normalization(c, s)
    c = rnd(5, 1, 8, 8, 4)
    s = rnd(6, 1, 64, 64, 4)
    out["nhwc_c"] = c
    out["nhwc_s"] = s
    out["nhwc_o"] = mu.adaptive_instance_normalization(c, s, mode="NHWC")
    ch = rnd(7, 1, 4, 8, 8).half()
    sh = rnd(8, 1, 4, 32, 32).half()
    out["half_c"] = ch
    out["half_s"] = sh
    out["half_o"] = mu.adaptive_instance_normalization(ch, sh)
    save("adain", **out)

    t = torch.tensor([[2, 1, 4], [2, 9, 12], [6, 4, 4], [7, 3, 99], [8, 1, 3]])
    a0 = mu.tensor_group_by_then_average(t, index_column=0, value_columns=[1, 2])[0]
    a1, u1 = mu.tensor_group_by_then_average(t, index_column=1, value_columns=[0], return_unique=True)
    g = torch.Generator().manual_seed(3)
    tr = torch.cat([torch.randn(5000, 4, generator=g), torch.randint(0, 300, (5000, 1), generator=g).float()], 1)
    ar, ur = mu.tensor_group_by_then_average(tr, index_column=-1, value_columns=[0, 1, 2, 3], return_unique=True)
    save("groupby", doc_t=t, doc_a0=a0, doc_a1=a1, doc_u1=u1, rnd_t=tr, rnd_a=ar, rnd_u=ur)


def sec_idmap():
    cm = R.import_corrmap()
    ids = synth_ids(11, 3, 16, 24, n_vertex=30)      # non-square on purpose: x/H, y/W quirk
    with quiet():
        m = cm.IDMap(tensor=ids.clone())
        vsi = m.create_vertex_screen_info()
    # SURVEY 8c golden (3), second half: two frames of the reference's own dumped sphere id maps
    # (resources/example-sphere-and-object-views/sphere/id, legacy int16 layout (obj, mat, texX, texY)), remapped to the current
    # layout as (obj, mat, 0, texY*1024 + texX) and cropped to a 128x128 window that holds background, silhouette and interior
    real = []
    for fr in (11, 25):
        a = np.load(f"/root/reference/resources/example-sphere-and-object-views/sphere/id/id_{fr}.npy").astype(np.int32)
        a = a[32:160, 32:160]
        real.append(np.stack([a[..., 0], a[..., 1], np.zeros_like(a[..., 0]), a[..., 3] * 1024 + a[..., 2]], -1))
    real = torch.from_numpy(np.stack(real))
    with quiet():
        mr = cm.IDMap(tensor=real.clone())
        vsir = mr.create_vertex_screen_info()
    save("idmap", ids=ids, masks=m.masks, vsi=vsi, height_prop=m.height, width_prop=m.width,
         sphere_ids=real, sphere_masks=mr.masks, sphere_vsi=vsir)


def sec_overlap():
    """OverlapCorresponder.step_finished (corresponder.py:298-376) on synthetic id maps."""
    cm = R.import_corrmap()
    import common_utils.stable_render_utils.corresponder as co
    from comfyUI.types import SamplingCallbackContext

    class ED:  # duck-typed EngineData: only .id_maps is read
        pass

    cases = [
        # name, N, H, W (ids), latent h,w, n_vertex, ratio, stop, timestep
        ("a", 2, 64, 64, 8, 8, 50, 0.5, 500, 800),
        ("b", 4, 64, 64, 8, 8, 400, 0.1, 500, 500),
        ("c", 3, 48, 48, 6, 6, 60, 0.7, 0, 10),          # non power of two: fp32 x/H*w rounding
        ("e", 2, 56, 56, 7, 7, 90, 0.3, 0, 10),          # (non-square latents raise IndexError in the reference's
                                                         #  own debug f-string, corresponder.py:321: not a valid case)
        ("skip", 2, 32, 32, 4, 4, 20, 0.5, 500, 499),    # timestep < stop -> untouched
        ("d", 8, 128, 128, 16, 16, 3000, 0.5, 500, 999),
    ]
    out = {}
    meta = {}
    for name, n, H, W, h, w, nv, ratio, stop, ts in cases:
        ids = synth_ids(100 + len(out), n, H, W, n_vertex=nv)
        x = rnd(200 + len(out), n, 4, h, w)
        ed = ED()
        with quiet(), one_thread():
            ed.id_maps = cm.IDMap(tensor=ids.clone())
            oc = co.OverlapCorresponder(step_finished_inject_ratio=ratio, step_finished_stop_inject_timestep=stop)
            ctx = SamplingCallbackContext(noise=x.clone(), step_index=0, denoised=x.clone(), total_steps=1,
                                          timesteps=[ts], sigmas=[1.0])
            oc.step_finished(ed, ctx)
        out[f"{name}_ids"] = ids
        out[f"{name}_x"] = x
        out[f"{name}_out"] = ctx.noise
        meta[name] = dict(ratio=ratio, stop=stop, timestep=ts)
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    save("overlap_step", **out)

    # pre_atten_inject with preset indices (corresponder.py:188-220)
    oc = co.OverlapCorresponder(pre_attn_inject_num_random_frames=2)
    oc._random_frame_indices = torch.tensor([3, 1])
    nctx = rnd(9, 4, 6, 8)
    q, k, v = oc.pre_atten_inject(None, None, nctx, nctx, nctx, 0)
    save("pre_attn_inject", n=nctx, idx=oc._random_frame_indices, q=q, k=k.contiguous(), v=v.contiguous())


def sec_corrmap():
    """CorrespondMap.update/_update (corrmap.py:578-736)."""
    cm = R.import_corrmap()
    out = {}
    meta = {}

    def run(name, k, mh, mw, frames, ids, mode, masks, sprite, material, inverse, ignore, pre=None):
        m = cm.CorrespondMap(k=k, height=mh, width=mw, immediate_load=False) if False else None
        try:
            m = cm.CorrespondMap(k=k, height=mh, width=mw)
        except Exception as e:  # ResourcesObj may want a name
            m = cm.CorrespondMap(name=f"g_{name}", k=k, height=mh, width=mw)
        if pre is not None:
            pre(m)
        err = ""
        try:
            with quiet(), one_thread():
                m.update(color_frames=frames.clone(), id_maps=ids.clone(), spriteID=sprite, materialID=material,
                         mode=mode, masks=None if masks is None else masks.clone(), inverse_masks=inverse,
                         ignore_obj_mat_id=ignore)
        except Exception as e:
            err = type(e).__name__
        out[f"{name}_frames"] = frames
        out[f"{name}_ids"] = ids
        if masks is not None:
            out[f"{name}_masks"] = masks
        out[f"{name}_values"] = m._values.clone()
        out[f"{name}_writtens"] = m._writtens.clone()
        meta[name] = dict(k=k, mh=mh, mw=mw, mode=mode, sprite=sprite, material=material, inverse=inverse,
                          ignore=ignore, has_masks=masks is not None, err=err)
        return m

    # (1) the reference's own update_test known answer (corrmap.py:905-914): all-red 512^2 frame, map 4 full
    H = W = 64
    ids = torch.zeros(H, W, 4, dtype=torch.int32)
    ids[..., 2] = 4
    ids[..., 3] = torch.arange(H * W, dtype=torch.int32).view(H, W)
    red = torch.zeros(H, W, 4)
    red[..., 0] = 1
    red[..., 3] = 1
    # NB a 3-D (H,W,4) id map never returns in the reference (corrmap.py:629 appends to the list it iterates);
    # the known answer is therefore pinned through the 4-D (1,H,W,4) form.
    run("kat", 3, H, W, red[None], ids[None], "first_avg", None, None, None, False, False)

    # (2) random ids, N=3 frames, full coverage (no mask compaction), dup targets -> last writer wins
    g = torch.Generator().manual_seed(5)
    N, H, W = 3, 32, 32
    ids = torch.zeros(N, H, W, 4, dtype=torch.int32)
    ids[..., 0] = 2
    ids[..., 1] = 7
    ids[..., 2] = torch.randint(0, 9, (N, H, W), generator=g, dtype=torch.int32)
    ids[..., 3] = torch.randint(0, 200, (N, H, W), generator=g, dtype=torch.int32)
    frames = torch.rand(N, H, W, 3, generator=g)
    for mode in ("first", "replace", "first_avg", "replace_avg"):
        run(f"rnd_{mode}", 3, 16, 16, frames, ids, mode, None, 2, 7, False, False)
    # some rows of another sprite: filtered (no mask => single gather, no double-gather quirk)
    ids2 = ids.clone()
    ids2[..., 0][ids2[..., 3] % 3 == 0] = 5
    run("rnd_sprite", 3, 16, 16, frames, ids2, "first", None, 2, 7, False, False)
    run("rnd_ignore", 3, 16, 16, frames, ids2, "first", None, 2, 7, False, True)

    # (3) masks given (DefaultCorresponder.finished passes id-masks with inverse_masks=True).
    #  all-ones coverage: mask path is the identity
    masks0 = torch.zeros(N, H, W)
    run("mask_full", 3, 16, 16, frames, ids, "first", masks0, 2, 7, True, False)
    #  partial coverage: reference re-indexes the compacted colour rows with ORIGINAL pixel indices
    #  (corrmap.py:703 + :710) -> wrong rows or IndexError; pinned here as-is.
    masksp = (torch.rand(N, H, W, generator=g) < 0.3).float()
    masksp[:, -4:, :] = 0.0          # keep the tail valid so max index >= M -> IndexError
    run("mask_partial_err", 3, 16, 16, frames, ids, "first", masksp, 2, 7, True, False)
    masksq = torch.zeros(N, H, W)
    masksq[:, -8:, :] = 1.0          # tail masked out: indices < M for many rows? (first 24 rows valid => i<M)
    run("mask_partial_ok", 3, 16, 16, frames, ids, "first", masksq, 2, 7, True, False)
    run("mask_partial_ignore", 3, 16, 16, frames, ids, "first", masksp, 2, 7, True, True)

    # (4) second update onto a pre-written map ('first' keeps, 'replace' overwrites)
    def pre(m):
        with quiet(), one_thread():
            m.update(color_frames=frames[:1].clone(), id_maps=ids[:1].clone(), spriteID=2, materialID=7, mode="first")
    frames2 = torch.rand(2, H, W, 4, generator=g)
    run("second_first", 3, 16, 16, frames2, ids[1:], "first", None, 2, 7, False, False, pre=pre)
    run("second_replace", 3, 16, 16, frames2, ids[1:], "replace", None, 2, 7, False, False, pre=pre)

    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    save("corrmap_update", **out)


def sec_noisepool():
    """renderManager.py:926-936 is not importable (GL); the two arithmetic lines are restated verbatim on
    torch tensors and AdaIN comes from the reference (math_utils)."""
    import common_utils.math_utils as mu
    out = {}
    for i, (H, W) in enumerate([(64, 64), (128, 64), (512, 512)]):
        noise = rnd(20 + i, 1, H, W, 4).half()                 # noise FBO texture is RGBA16F
        alpha = (torch.rand(1, H, W, generator=torch.Generator().manual_seed(30 + i)) < 0.6).half()
        mask_data = 1.0 - alpha                                 # fp16 (colour FBO is RGBA16F)
        bg = rnd(40 + i, 1, H, W, 4)                            # GlobalBGNoise fp32
        mask = mask_data.unsqueeze(-1).expand_as(noise)
        n = noise * (1.0 - mask) + bg * mask
        n = n.view(-1, 8, 8, 4).mean(dim=(1, 2)).view(H // 8, W // 8, 4)
        o = mu.adaptive_instance_normalization(n.unsqueeze(0), noise, mode="NHWC").contiguous()
        if H <= 128:
            out[f"n{i}_noise"] = noise
            out[f"n{i}_alpha"] = alpha
            out[f"n{i}_bg"] = bg
        out[f"n{i}_pooled"] = n
        out[f"n{i}_out"] = o
    save("noise_pool", **out)


def sec_sched():
    import comfy.samplers as cs
    import comfy.model_sampling as ms
    import comfy.k_diffusion.sampling as ks

    class M:
        pass
    m = M()

    class MS(ms.ModelSamplingDiscrete, ms.EPS):
        pass
    m.model_sampling = MS()
    out = {"sigmas_table": m.model_sampling.sigmas, "log_sigmas": m.model_sampling.log_sigmas}
    for sch in ["normal", "sgm_uniform", "karras", "simple", "ddim_uniform", "exponential"]:
        for steps in (4, 20):
            out[f"{sch}_{steps}"] = cs.calculate_sigmas_scheduler(m, sch, steps)
    # KSampler.set_steps with denoise < 1 (samplers.py:996-1003) + timesteps list
    for sch, steps, den in [("normal", 20, 1.0), ("normal", 20, 0.55), ("sgm_uniform", 4, 0.55), ("karras", 20, 0.7)]:
        k = cs.KSampler(m, steps=steps, device="cpu", sampler="euler", scheduler=sch, denoise=den)
        out[f"ks_{sch}_{steps}_{int(den*100)}_sigmas"] = k.sigmas
        out[f"ks_{sch}_{steps}_{int(den*100)}_timesteps"] = torch.stack([t.reshape(()) for t in k.timesteps])
    sig = torch.tensor([14.6146, 3.2, 0.9, 0.0292, 0.5])
    out["ts_in"] = sig
    out["ts_out"] = m.model_sampling.timestep(sig)
    tq = torch.tensor([0.0, 10.5, 999.0, 512.25])
    out["sg_in"] = tq
    out["sg_out"] = m.model_sampling.sigma(tq)
    x = rnd(1, 2, 4, 8, 8)
    s2 = torch.tensor([3.0, 3.0])
    out["eps_x"] = x
    out["eps_in"] = m.model_sampling.calculate_input(s2, x)
    mo = rnd(2, 2, 4, 8, 8)
    out["eps_mo"] = mo
    out["eps_den"] = m.model_sampling.calculate_denoised(s2, mo, x)

    # toy-model trajectories: denoised = tanh(x) * 0.5 / (1 + sigma)
    def toy(x, sigma, **kw):
        return torch.tanh(x) * 0.5 / (1 + sigma.view(-1, 1, 1, 1))
    sigmas = cs.calculate_sigmas_scheduler(m, "normal", 6)
    x0 = rnd(3, 2, 4, 8, 8) * sigmas[0]
    out["traj_sigmas"] = sigmas
    out["traj_x0"] = x0
    out["traj_euler"] = ks.sample_euler(toy, x0.clone(), sigmas, disable=True)
    torch.manual_seed(77)
    out["traj_ddpm"] = ks.sample_ddpm(toy, x0.clone(), sigmas, disable=True)
    torch.manual_seed(78)
    out["traj_lcm"] = ks.sample_lcm(toy, x0.clone(), sigmas, disable=True)
    save("sampling", **out)


# ------------------------------------------------------------------------------------------------
SD15 = {'use_checkpoint': False, 'image_size': 32, 'out_channels': 4, 'use_spatial_transformer': True, 'legacy': False,
        'adm_in_channels': None, 'dtype': torch.float32, 'in_channels': 4, 'model_channels': 320,
        'num_res_blocks': [2, 2, 2, 2], 'transformer_depth': [1, 1, 1, 1, 1, 1, 0, 0], 'channel_mult': [1, 2, 4, 4],
        'transformer_depth_middle': 1, 'use_linear_in_transformer': False, 'context_dim': 768, 'num_heads': 8,
        'transformer_depth_output': [1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0],
        'use_temporal_attention': False, 'use_temporal_resblock': False}
TINY = dict(SD15, model_channels=64, context_dim=64)      # same topology, 1/5 width: CPU-fast


def build_unet(cfg, seed):
    from comfy.ldm.modules.diffusionmodules.openaimodel import UNetModel
    import comfy.ops
    m = UNetModel(**cfg, operations=comfy.ops.disable_weight_init)
    m.eval()
    ns, norm = synth.fill_module_(m, seed=seed)
    return m, ns, norm


def sec_unet():
    import comfy.ldm.modules.attention as att
    att.optimized_attention = att.attention_basic
    with torch.no_grad():
        # tiny config, with and without the OverlapCorresponder K/V injection
        m, ns, norm = build_unet(TINY, seed=1)
        _jdump({"names_shapes": ns, "norm_names": norm}, os.path.join(GOLD, "unet_tiny_keys.json"))
        x = rnd(1, 4, 4, 16, 16)
        t = torch.tensor([981.0, 981.0, 981.0, 981.0])
        ctx = rnd(2, 4, 77, 64)
        y = m(x, t, context=ctx, transformer_options={})
        import common_utils.stable_render_utils.corresponder as co
        oc = co.OverlapCorresponder(pre_attn_inject_num_random_frames=1)
        oc._random_frame_indices = torch.tensor([2])
        y_inj = m(x, t, context=ctx, transformer_options={"positive_cond_indices": [2, 3]},
                  engine_data=object(), corresponder=oc)
        # a latent whose sizes do not halve evenly (what a conditioning AREA crops out): 10x12 -> 5x6 -> 3x3 -> 2x2; the decoder's
        # Upsample then interpolates to the skip tensor's size instead of x2 (openaimodel.py:100-117)
        x_odd = rnd(9, 2, 4, 10, 12)
        y_odd = m(x_odd, t[:2], context=ctx[:2], transformer_options={})
        save("unet_tiny", x=x, t=t, ctx=ctx, y=y, y_inj=y_inj, inj_idx=oc._random_frame_indices, x_odd=x_odd, y_odd=y_odd)
        del m
        # real SD1.5 shapes, 16x16 latent, B=2
        m, ns, norm = build_unet(SD15, seed=0)
        _jdump({"names_shapes": ns, "norm_names": norm}, os.path.join(GOLD, "unet_sd15_keys.json"))
        x = rnd(3, 2, 4, 16, 16)
        t = torch.tensor([500.0, 500.0])
        ctx = rnd(4, 2, 77, 768)
        y = m(x, t, context=ctx, transformer_options={})
        save("unet_sd15_16", x=x, t=t, ctx=ctx, y=y)


# SDXL-family topology (comfy/supported_models.py:153-160) at 1/5 width: no attention at level 0, deeper transformers below,
# fixed head width, linear proj_in / proj_out, vector conditioning through label_emb
SDXL_TINY = {'use_checkpoint': False, 'image_size': 32, 'out_channels': 4, 'use_spatial_transformer': True, 'legacy': False,
             'num_classes': 'sequential', 'adm_in_channels': 192, 'dtype': torch.float32, 'in_channels': 4, 'model_channels': 64,
             'num_res_blocks': [2, 2, 2], 'transformer_depth': [0, 0, 2, 2, 3, 3], 'channel_mult': [1, 2, 4],
             'transformer_depth_middle': 3, 'use_linear_in_transformer': True, 'context_dim': 128, 'num_head_channels': 32,
             'num_heads': -1, 'transformer_depth_output': [0, 0, 0, 2, 2, 2, 3, 3, 3],
             'use_temporal_attention': False, 'use_temporal_resblock': False}


def sec_sdxl():
    import comfy.ldm.modules.attention as att
    att.optimized_attention = att.attention_basic
    with torch.no_grad():
        m, ns, norm = build_unet(SDXL_TINY, seed=4)
        _jdump({"names_shapes": ns, "norm_names": norm}, os.path.join(GOLD, "unet_sdxl_tiny_keys.json"))
        x = rnd(5, 2, 4, 16, 16)
        t = torch.tensor([731.0, 731.0])
        ctx = rnd(6, 2, 77, 128)
        yv = rnd(7, 2, 192)
        y = m(x, t, context=ctx, y=yv, transformer_options={})
        save("unet_sdxl_tiny", x=x, t=t, ctx=ctx, yvec=yv, y=y)
        # SDXL.encode_adm itself (comfy/model_base.py:352-369) on a stub that only carries the 256-wide Timestep embedder
        import types
        import comfy.model_base as mb
        from comfy.ldm.modules.diffusionmodules.openaimodel import Timestep
        stub = types.SimpleNamespace(embedder=Timestep(256), noise_augmentor=None)
        pooled = rnd(8, 2, 1280)
        a1 = mb.SDXL.encode_adm(stub, pooled_output=pooled, width=1024, height=1024)
        a2 = mb.SDXL.encode_adm(stub, pooled_output=pooled, width=832, height=1216, crop_w=8, crop_h=16, target_width=1024, target_height=1024)
        # latent scaling of the two model families (comfy/latent_formats.py SD15 / SDXL process_in / process_out), picked by
        # comfy/supported_models.py (SD15.latent_format = SD15, SDXL.latent_format = SDXL)
        import comfy.latent_formats as lf
        import comfy.supported_models as sm
        lat = rnd(9, 1, 4, 8, 8)
        save("sdxl_adm", pooled=pooled, adm_default=a1, adm_custom=a2, lat=lat,
             sdxl_in=sm.SDXL.latent_format().process_in(lat), sdxl_out=sm.SDXL.latent_format().process_out(lat),
             sd15_in=sm.SD15.latent_format().process_in(lat), sd15_out=sm.SD15.latent_format().process_out(lat),
             sdxl_scale=np.float64(lf.SDXL().scale_factor), sd15_scale=np.float64(lf.SD15().scale_factor))


def sec_vae():
    from comfy.ldm.modules.diffusionmodules.model import Decoder
    import comfy.ldm.modules.diffusionmodules.model as mm
    import comfy.ops
    dd = {'double_z': True, 'z_channels': 4, 'resolution': 256, 'in_channels': 3, 'out_ch': 3, 'ch': 128,
          'ch_mult': [1, 2, 4, 4], 'num_res_blocks': 2, 'attn_resolutions': [], 'dropout': 0.0}
    with torch.no_grad():
        d = Decoder(**dd)
        d.eval()
        ns, norm = synth.fill_module_(d, seed=2)
        _jdump({"names_shapes": ns, "norm_names": norm}, os.path.join(GOLD, "vae_dec_keys.json"))
        z = rnd(5, 2, 4, 8, 8)
        y = d(z)
        # VAE.decode post-processing (sd.py:329-346): post_quant_conv is part of AutoencoderKL; here only the
        # Decoder + clamp((y+1)/2) + NHWC
        img = torch.clamp((y + 1.0) / 2.0, min=0.0, max=1.0).movedim(1, -1)
        save("vae_dec", z=z, y=y, img=img)


def sec_vaeenc():
    """VAE.encode's arithmetic from the reference classes: Encoder (model.py:441-520) + AutoencoderKL.quant_conv +
    DiagonalGaussianRegularizer(sample=True) (autoencoder.py:13-31, 175-190), full SD1.x width, 64x64 pixels"""
    from comfy.ldm.models.autoencoder import AutoencoderKL
    dd = {'double_z': True, 'z_channels': 4, 'resolution': 256, 'in_channels': 3, 'out_ch': 3, 'ch': 128,
          'ch_mult': [1, 2, 4, 4], 'num_res_blocks': 2, 'attn_resolutions': [], 'dropout': 0.0}
    with torch.no_grad():
        ae = AutoencoderKL(ddconfig=dd, embed_dim=4)
        ae.eval()

        class Enc(torch.nn.Module):                 # state-dict order = Encoder names, then quant_conv
            def __init__(self):
                super().__init__()
                for n, m in ae.encoder.named_children():
                    setattr(self, n, m)
                self.quant_conv = ae.quant_conv
        ns, norm = synth.fill_module_(Enc(), seed=3)
        _jdump({"names_shapes": ns, "norm_names": norm}, os.path.join(GOLD, "vae_enc_keys.json"))
        pixels = torch.rand(2, 64, 64, 3, generator=torch.Generator().manual_seed(9))
        x = pixels.movedim(-1, 1) * 2.0 - 1.0       # VAE.process_input (sd.py:222)
        mom = ae.quant_conv(ae.encoder(x))
        torch.manual_seed(31)
        z = ae.encode(x)                            # draws torch.randn(mean.shape) from the global generator
        torch.manual_seed(31)
        noise = torch.randn(2, 4, 8, 8)
        save("vae_enc", pixels=pixels, moments=mom, z=z, noise=noise)


def sec_e2e():
    """End-to-end reference sampling stack on a tiny UNet: custom_ksampler -> comfy.sample.sample -> KSampler
    -> calc_cond_uncond_batch -> BaseModel.apply_model -> UNetModel, with OverlapCorresponder (step_finished
    callback + K/V injection) exactly as CorrespondSampler wires it (_nodes/samplers.py:163-201)."""
    cm = R.import_corrmap()
    import comfy.ldm.modules.attention as att
    att.optimized_attention = att.attention_basic
    import comfy.supported_models
    import comfy.model_patcher
    import comfy.model_base
    import comfy.sample
    import common_utils.stable_render_utils.corresponder as co
    from functools import partial
    unet_config = {k: v for k, v in TINY.items()}
    mc = comfy.supported_models.SD15(unet_config)
    mc.unet_config = unet_config
    mc.set_inference_dtype(torch.float32, None)
    bm = comfy.model_base.BaseModel(mc, model_type=comfy.model_base.ModelType.EPS, device="cpu")
    bm.eval()
    synth.fill_module_(bm.diffusion_model, seed=1)
    mp = comfy.model_patcher.ModelPatcher(bm, load_device=torch.device("cpu"), offload_device=torch.device("cpu"))

    class ED:
        pass
    out = {}
    meta = {}
    N, H, W = 3, 128, 128
    h, w = H // 8, W // 8
    ids = synth_ids(300, N, H, W, n_vertex=500)
    pos = [[rnd(11, 1, 77, 64), {}]]
    neg = [[rnd(12, 1, 77, 64), {}]]
    noise = rnd(13, N, 4, h, w)
    out.update(ids=ids, pos=pos[0][0], neg=neg[0][0], noise=noise)
    for name, sampler, sched, steps, cfg, use_overlap in [
        ("euler_plain", "euler", "normal", 3, 5.0, False),
        ("ddim_overlap", "ddim", "normal", 4, 7.5, True),
        ("ddpm_overlap", "ddpm", "sgm_uniform", 3, 2.0, True),
        ("euler_cfg1", "euler", "karras", 2, 1.0, False),
    ]:
        ed = ED()
        with quiet():
            ed.id_maps = cm.IDMap(tensor=ids.clone())
        kwargs = {}
        callbacks = []
        if use_overlap:
            oc = co.OverlapCorresponder(step_finished_inject_ratio=0.5, step_finished_stop_inject_timestep=500)

            def make_cb(oc_):
                def on_step(engine_data, context):
                    with one_thread():
                        oc_.step_finished(engine_data, context)
                return on_step
            callbacks = [partial(make_cb(oc), ed)]
            kwargs = dict(engine_data=ed, corresponder=oc)
        torch.manual_seed(4242)
        # exactly CorrespondSampler's call (_nodes/samplers.py:187-201): zero latent + incoming engine noise,
        # seed=None (custom_ksampler draws it from the global RNG)
        latent = {"samples": torch.zeros(N, 4, h, w), "noise": noise.clone()}
        import nodes as ref_nodes
        with quiet(), torch.no_grad():
            s = ref_nodes.custom_ksampler(model=mp, seed=None, steps=steps, cfg=cfg, sampler_name=sampler,
                                          scheduler=sched, positive=pos, negative=neg, latent=latent, denoise=1.0,
                                          noise_option='incoming', callbacks=list(callbacks), **kwargs)[0]["samples"]
        out[f"{name}_samples"] = s
        meta[name] = dict(sampler=sampler, scheduler=sched, steps=steps, cfg=cfg, overlap=use_overlap,
                          rng_seed=4242,
                          inj_idx=(oc._random_frame_indices.tolist() if use_overlap else None))
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    save("e2e_tiny", **out)

def sec_controlnet():
    """cldm.ControlNet.forward + control_merge + UNetModel.forward(control=...) on the tiny topology"""
    import comfy.ldm.modules.attention as att
    att.optimized_attention = att.attention_basic
    import comfy.cldm.cldm as cldm
    import comfy.ops
    cfg = {k: v for k, v in TINY.items() if k not in ("out_channels", "transformer_depth_output")}
    with torch.no_grad():
        cn = cldm.ControlNet(hint_channels=3, operations=comfy.ops.disable_weight_init, **cfg)
        cn.eval()
        ns, norm = synth.fill_module_(cn, seed=5)
        _jdump({"names_shapes": ns, "norm_names": norm}, os.path.join(GOLD, "controlnet_tiny_keys.json"))
        unet, _, _ = build_unet(TINY, seed=1)
        x = rnd(1, 2, 4, 16, 16)
        t = torch.tensor([500.0, 500.0])
        ctx = rnd(2, 2, 77, 64)
        hint = torch.rand(2, 3, 128, 128, generator=torch.Generator().manual_seed(3))
        outs = cn(x=x, hint=hint, timesteps=t, context=ctx)
        strength = 0.8
        control = {"input": [], "middle": [outs[-1].clone() * strength], "output": [o.clone() * strength for o in outs[:-1]]}
        y = unet(x, t, context=ctx, control=control, transformer_options={})
        save("controlnet_tiny", x=x, t=t, ctx=ctx, hint=hint, y=y, mid=outs[-1], out0=outs[0], out11=outs[11])


def sec_legacy():
    """legacy Overlap / ResizeOverlap with the four algorithms (legacy_codes/stable_rendering_algo/overlap/*)"""
    sys.path.insert(0, os.path.join(R.REF, "legacy_codes"))
    R.import_corrmap()            # seeds a bare ``engine.static`` package (the real __init__ pulls in OpenGL)
    sys.modules["engine.static"].Color = type("Color", (), {})       # only imported by name, never used on this path
    import importlib
    ov = importlib.import_module("stable_rendering_algo.overlap.overlap")
    alg = importlib.import_module("stable_rendering_algo.overlap.algorithms")
    sch = importlib.import_module("stable_rendering_algo.overlap.overlap_scheduler")
    dc = importlib.import_module("stable_rendering_algo.data_classes.correspondence_map")
    g = torch.Generator().manual_seed(8)
    T, H, W, C, h, w = 3, 16, 16, 4, 4, 4
    ids = torch.zeros(T, H, W, 4, dtype=torch.int32)
    ids[..., 0] = 1
    ids[..., 2] = torch.randint(0, 12, (T, H, W), generator=g, dtype=torch.int32)   # >= 100 vertices: the reference
    ids[..., 3] = torch.randint(0, 12, (T, H, W), generator=g, dtype=torch.int32)   # divides by len(corr_map)//100
    ids[torch.rand(T, H, W, generator=g) < 0.3] = 0
    cmap = {}
    for f in range(T):
        for i in range(H):
            for j in range(W):
                k = tuple(int(v) for v in ids[f, i, j])
                if k == (0, 0, 0, 0):
                    continue
                cmap.setdefault(k, []).append(([i, j], f))
    cm = dc.CorrespondenceMap(cmap, W, H, T)
    frames = [rnd(20 + f, 1, C, H, W) for f in range(T)]
    lat = [rnd(30 + f, 1, C, h, w) for f in range(T)]
    vn = torch.rand(T, H, W, generator=g)
    out = dict(ids=ids, frames=torch.stack(frames), latents=torch.stack(lat), view_normal=vn)
    algos = dict(average=alg.AverageDistance(), frame=alg.FrameDistance(), pixel=alg.PixelDistance(), view_normal=alg.PerpendicularViewNormal())
    for name, a in algos.items():
        for radius in (0, 1):
            o = ov.Overlap(alpha_scheduler=sch.Scheduler(interpolate_begin=0.6), kernel_radius_scheduler=sch.Scheduler(interpolate_begin=float(radius)),
                           algorithm=a, verbose=False)
            with quiet(), one_thread():
                res = o([f.clone() for f in frames], cm, step=1, timestep=500, view_normal_map=vn.unsqueeze(-1))
            out[f"full_{name}_r{radius}"] = res
            ro = ov.ResizeOverlap(alpha_scheduler=sch.Scheduler(interpolate_begin=0.6), kernel_radius_scheduler=sch.Scheduler(interpolate_begin=float(radius)),
                                  algorithm=a, verbose=False)
            with quiet(), one_thread():
                res2 = ro([l.clone() for l in lat], cm, step=1, timestep=500, view_normal_map=vn.unsqueeze(-1))
            out[f"resize_{name}_r{radius}"] = torch.stack(res2)
    save("legacy_overlap", **out)


def sec_dump():
    """CorrespondMap.dump / Load round trip by the reference (corrmap.py:738-872): the dumped directory itself is the
    fixture (tests/golden/corrmap_dump/*), plus the tensors that went in and came back."""
    import shutil
    cm = R.import_corrmap()
    m = cm.CorrespondMap(name="gold", k=2, height=8, width=8)
    g = torch.Generator().manual_seed(4)
    m._values = torch.rand(4, 64, 4, generator=g).half()
    m._writtens = torch.rand(4, 64, generator=g) > 0.5
    out = os.path.join(GOLD, "corrmap_dump")
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out, exist_ok=True)
    with quiet():
        p = m.dump(out, name="gold")       # (force=True leaves real_path unset in the reference -> TypeError)
        m2 = cm.CorrespondMap.Load(p)
    save("corrmap_dump_io", values_in=m._values, writtens_in=m._writtens, values_back=m2._values, writtens_back=m2._writtens)


def ref_noise_sequence_loader():
    """-> the reference's NoiseSequenceLoader.__call__ (_nodes/loaders.py:79-152) as a plain function(self, directory, frame_start,
    num_frames, sd_version): compiled from the reference's own source text at generation time (the _nodes package itself drags in
    the GL engine), node-type annotations dropped, with the reference's adaptive_instance_normalization"""
    lsrc = open(os.path.join(R.REF, "source/comfyUI/stable_rendering/_nodes/loaders.py")).read()
    ltree = ast.parse(lsrc)
    lcls = next(n for n in ltree.body if isinstance(n, ast.ClassDef) and n.name == "NoiseSequenceLoader")
    call = next(n for n in lcls.body if isinstance(n, ast.FunctionDef) and n.name == "__call__")
    call.args.defaults = []
    for a in call.args.args:                                                   # drop the node-type annotations
        a.annotation = None
    call.returns = None
    ast.fix_missing_locations(call)
    mu = R.import_math_utils() if hasattr(R, "import_math_utils") else __import__("common_utils.math_utils", fromlist=["x"])
    import re as _re

    def extract_index(name, default):
        m = _re.findall(r"\d+", os.path.splitext(name)[0])
        return int(m[-1]) if m else default

    class _L:
        @staticmethod
        def debug(*a, **k):
            pass
    lns = dict(os=os, np=np, torch=torch, extract_index=extract_index, ComfyUILogger=_L,
               adaptive_instance_normalization=mu.adaptive_instance_normalization,
               LATENT=lambda **kw: dict(kw))
    exec(compile(ast.Module(body=[call], type_ignores=[]), "loaders.py[NoiseSequenceLoader.__call__]", "exec"), lns)
    return lns["__call__"]


def sec_gbufdump():
    """G-buffer dump layout (DiffusionManager._outputMap/_outputNumpyData/_outputDepthMap, diffusionManager.py:160-259).
    The manager module cannot be imported here (its package pulls the GL managers), so the three methods are taken from the
    reference FILE at generation time (ast -> exec with numpy/PIL) and run on seeded planes; the PNG/NPY files they write
    are read back and stored as arrays (tests/golden/gbuffer_dump.npz)."""
    import ast
    import queue
    import shutil
    import tempfile
    from PIL import Image
    src = open(os.path.join(R.REF, "source/engine/managers/diffusionManager.py")).read()
    tree = ast.parse(src)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "DiffusionManager")
    want = {"_outputNumpyData", "_outputMap", "_outputDepthMap"}
    fns = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in want]
    assert len(fns) == 3

    class _Log:
        @staticmethod
        def error(msg):
            raise RuntimeError(msg)
    ns = dict(np=np, os=os, Image=Image, Optional=__import__("typing").Optional, EngineLogger=_Log)
    exec(compile(ast.Module(body=fns, type_ignores=[]), "diffusionManager.py[methods]", "exec"), ns)
    tmp = tempfile.mkdtemp()

    class Fake:
        _outputPath = tmp
        _unfinished_queue = queue.Queue()
    fake = Fake()
    H, W = 24, 20
    g = np.random.default_rng(11)
    color = g.random((H, W, 4), dtype=np.float32)
    color[..., 3] = (g.random((H, W)) > 0.3).astype(np.float32)
    normal = g.random((H, W, 3), dtype=np.float32)
    canny = (g.random((H, W, 3)) > 0.8).astype(np.float32)
    ids = g.integers(0, 500, (H, W, 4)).astype(np.int32)
    pos = g.standard_normal((H, W, 3)).astype(np.float32)
    noise = g.standard_normal((H, W, 4)).astype(np.float16)
    depth = g.random((H, W), dtype=np.float32) * (g.random((H, W)) > 0.4)
    gray = g.random((H, W), dtype=np.float32)                 # 2-D map path of _outputMap
    frame = 7
    ns["_outputMap"](fake, "Color", color, True, np.uint8, frame)
    ns["_outputMap"](fake, "normal", normal, True, np.uint8, frame)
    ns["_outputMap"](fake, "canny", canny, True, np.uint8, frame)
    ns["_outputMap"](fake, "gray", gray, True, np.uint8, None)
    ns["_outputNumpyData"](fake, "id", ids, frame)
    ns["_outputNumpyData"](fake, "pos", pos, frame)
    ns["_outputNumpyData"](fake, "noise", noise, frame)
    ns["_outputDepthMap"](fake, depth, frame)
    back = {}
    for name in ("color", "normal", "canny"):
        back[name + "_png"] = np.asarray(Image.open(os.path.join(tmp, name, f"{name}_{frame}.png")))
    back["gray_png"] = np.asarray(Image.open(os.path.join(tmp, "gray", "gray.png")))
    back["depth_png"] = np.asarray(Image.open(os.path.join(tmp, "depth", f"depth_{frame}.png")))
    for name in ("id", "pos", "noise"):
        back[name + "_npy"] = np.load(os.path.join(tmp, name, f"{name}_{frame}.npy"))
    shutil.rmtree(tmp)
    # NoiseSequenceLoader.__call__ (_nodes/loaders.py:79-152) on two seeded 512^2 fp16 noise dumps (inputs are re-made from
    # the seeds in the test; only the loader's output is stored)
    lns = {"__call__": ref_noise_sequence_loader()}
    tmp2 = tempfile.mkdtemp()
    for i, seed in enumerate((21, 22)):
        np.save(os.path.join(tmp2, f"noise_{i}.npy"), np.random.default_rng(seed).standard_normal((512, 512, 4)).astype(np.float16))
    with one_thread():
        lat = lns["__call__"](None, tmp2, 0, 2, "SD15")
    shutil.rmtree(tmp2)
    back["loader_noise"] = lat["noise"].float().numpy()
    back["loader_seeds"] = np.array([21, 22])
    save("gbuffer_dump", color=color, normal=normal, canny=canny, ids=ids, pos=pos, noise=noise, depth=depth, gray=gray,
         frame=np.int64(frame), **back)


def cond_cases():
    """inputs of the conditioning-composition golden (SURVEY 8c golden 9), shared with the tests: name -> (pos, neg, cfg scale);
    an entry = [tensor (1,T,C), extras dict] in ComfyUI's CONDITIONING format"""
    def c(seed):
        return rnd(seed, 1, 5, 6)

    def blob(seed, thr):
        m = torch.nn.functional.avg_pool2d(rnd(seed, 1, 1, 128, 192).abs(), 31, 1, 15)
        return (m[0] > m.median() * thr).float()                       # (1,128,192) 0/1 blobs
    box = torch.zeros(1, 128, 192)
    box[:, 24:100, 40:150] = 1.0
    return {
        "multi": ([[c(1), {"strength": 1.3}], [c(2), {"mask": blob(3, 1.0), "mask_strength": 0.7, "set_area_to_bounds": False}]],
                  [[c(4), {}]], 7.5),
        "area": ([[c(5), {}], [c(6), {"area": ("percentage", 0.5, 0.5, 0.25, 0.25), "strength": 0.9}]], [[c(7), {}]], 4.0),
        "bounds": ([[c(8), {}], [c(9), {"mask": box, "mask_strength": 1.0, "set_area_to_bounds": True}]],
                   [[c(10), {}], [c(11), {"strength": 0.5}]], 6.0),
        "cfg1": ([[c(12), {}], [c(13), {"mask": blob(14, 0.8), "mask_strength": 1.0, "set_area_to_bounds": False}]], [[c(15), {}]], 1.0),
    }


def sec_conds():
    """calc_cond_uncond_batch / get_area_and_mult / cond_cat / sampling_function (comfy/samplers.py:50-358) and the list
    preparation of samplers.sample() (:887-912) with a TOY model: the mult / area / mask / batching arithmetic without a UNet.
    The toy's output depends on the batch row, so the golden also pins the ORDER in which entries are batched."""
    import comfy.samplers as cs
    import comfy.sample as csm
    calls = []

    class Toy:
        def memory_required(self, shape):
            return 0

        def apply_model(self, x, t, c_crossattn=None, transformer_options=None, **kw):
            b = x.shape[0]
            calls.append((list(x.shape), list(transformer_options["cond_or_uncond"]), list(transformer_options["positive_cond_indices"])))
            return (x * 0.5 + c_crossattn.mean(dim=(1, 2)).view(-1, 1, 1, 1) + 0.01 * t.view(-1, 1, 1, 1)
                    + 0.001 * torch.arange(b, dtype=torch.float32).view(-1, 1, 1, 1))
    out, meta = {}, {}
    x = rnd(20, 2, 4, 16, 24)
    sigma = torch.tensor([3.7, 3.7])
    out["x"], out["sigma"] = x, sigma
    for name, (pos, neg, scale) in cond_cases().items():
        p, n = csm.convert_cond(pos), csm.convert_cond(neg)
        cs.resolve_areas_and_cond_masks(p, 16, 24, "cpu")
        cs.resolve_areas_and_cond_masks(n, 16, 24, "cpu")
        for c_ in p:
            cs.create_cond_with_same_area_if_none(n, c_)
        for c_ in n:
            cs.create_cond_with_same_area_if_none(p, c_)
        calls.clear()
        with quiet():
            cp, up = cs.calc_cond_uncond_batch(Toy(), p, n, x, sigma, {})
        meta[name] = dict(calls=list(calls), n_pos=len(p), n_neg=len(n), scale=scale,
                          areas_pos=[list(map(int, e["area"])) if "area" in e else None for e in p],
                          areas_neg=[list(map(int, e["area"])) if "area" in e else None for e in n])
        with quiet():
            res = cs.sampling_function(Toy(), x, sigma, n, p, scale, {})
        out[f"{name}_cond"], out[f"{name}_uncond"], out[f"{name}_cfg"] = cp, up, res
        ent = {"pos": [], "neg": []}                              # the INPUT entries, so tests need not import this module
        for kind, lst in (("pos", pos), ("neg", neg)):
            for i, (t, ex) in enumerate(lst):
                out[f"{name}_{kind}{i}_c"] = t
                e = {k: (list(v) if isinstance(v, tuple) else v) for k, v in ex.items() if k != "mask"}
                if "mask" in ex:
                    out[f"{name}_{kind}{i}_mask"] = ex["mask"]
                    e["has_mask"] = True
                ent[kind].append(e)
        meta[name]["entries"] = ent
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    save("cond_compose", **out)


def cond_range_case():
    """entries with ConditioningSetTimestepRange windows (start_percent / end_percent, comfyUI/nodes.py:270-285): the second
    positive prompt acts in the middle of the schedule only, the negative prompt in its first half only"""
    def c(seed):
        return rnd(seed, 1, 5, 6)
    pos = [[c(31), {}], [c(32), {"start_percent": 0.3, "end_percent": 0.7, "strength": 0.8}]]
    neg = [[c(33), {"start_percent": 0.0, "end_percent": 0.5}]]
    return pos, neg, 5.0, [14.0, 6.0, 2.5, 1.2, 0.6, 0.2]


def sec_cond_ranges():
    """get_area_and_mult's sigma windows (samplers.py:60-67) through calculate_start_end_timesteps (:578-602) with the reference's
    ModelSamplingDiscrete.percent_to_sigma, sampling_function evaluated at sigmas inside and outside the windows (toy model)"""
    import comfy.model_sampling as cms
    import comfy.samplers as cs
    import comfy.sample as csm

    class Toy:
        model_sampling = cms.ModelSamplingDiscrete(None)

        def memory_required(self, shape):
            return 0

        def apply_model(self, x, t, c_crossattn=None, transformer_options=None, **kw):
            b = x.shape[0]
            return (x * 0.5 + c_crossattn.mean(dim=(1, 2)).view(-1, 1, 1, 1) + 0.01 * t.view(-1, 1, 1, 1)
                    + 0.001 * torch.arange(b, dtype=torch.float32).view(-1, 1, 1, 1))
    pos, neg, scale, sigmas = cond_range_case()
    x = rnd(40, 2, 4, 16, 24)
    p, n = csm.convert_cond(pos), csm.convert_cond(neg)
    cs.resolve_areas_and_cond_masks(p, 16, 24, "cpu")
    cs.resolve_areas_and_cond_masks(n, 16, 24, "cpu")
    toy = Toy()
    cs.calculate_start_end_timesteps(toy, n)
    cs.calculate_start_end_timesteps(toy, p)
    out = {"x": x, "sigmas": np.array(sigmas, np.float32), "scale": np.float32(scale),
           "windows": np.array([[e.get("timestep_start", -1.0), e.get("timestep_end", -1.0)] for e in p + n], np.float64)}
    for i, sg in enumerate(sigmas):
        with quiet():
            out[f"cfg_{i}"] = cs.sampling_function(toy, x, torch.tensor([sg, sg]), n, p, scale, {})
    for kind, lst in (("pos", pos), ("neg", neg)):
        for i, (t, _) in enumerate(lst):
            out[f"{kind}{i}_c"] = t
    save("cond_ranges", **out)


def sec_workflow():
    """(1) the reference's own Workflow.Load + build_prompt (engine/static/workflow.py) on every shipped example graph: the
    prompt dict, the output-node ids, or the exception type the reference raises for that file; (2) comfy's LoRA key map
    (comfy/utils.py unet_to_diffusers + comfy/lora.py model_lora_keys_unet's spelling) and merge arithmetic
    (comfy/lora.py load_lora + model_patcher.calculate_weight) on synthetic LoRA tensors."""
    import glob
    import tempfile
    import types
    cm = R.import_corrmap()
    pkg = sys.modules["engine.static"]
    for n in dir(cm):
        if not n.startswith("_") and not hasattr(pkg, n):
            setattr(pkg, n, getattr(cm, n))
    import folder_paths
    folder_paths.folder_names_and_paths["custom_nodes"] = ([tempfile.mkdtemp()], set())      # git-ignored dir of the reference
    with quiet():
        import engine.static.workflow as W
    out = {}
    for path in sorted(glob.glob(os.path.join(R.REF, "resources", "example-workflows", "*.json"))):
        name = os.path.basename(path)[:-5]
        try:
            with quiet():
                wf = W.Workflow.Load(path)
                prompt, ids, extra = wf.build_prompt()
            out[name] = dict(prompt={k: {"inputs": {a: (list(b) if isinstance(b, list) else b) for a, b in v["inputs"].items()},
                                         "class_type": v["class_type"]} for k, v in prompt.items()},
                             node_ids_to_be_ran=list(ids), has_output_node=bool(wf.has_output_node))
        except Exception as e:                                    # noqa: BLE001 - the reference's own failure is the datum
            out[name] = dict(error=type(e).__name__, message=str(e))
        print("  workflow", name, "->", "ok" if "prompt" in out[name] else out[name])
    _jdump(out, os.path.join(GOLD, "workflow_prompts.json"))

    # ---- LoRA
    import comfy.utils
    import comfy.lora
    import comfy.model_patcher
    from stable_renderer_amd.unet import SD15_CFG
    from stable_renderer_amd.model_shapes import unet_names_shapes
    cfg = dict(SD15_CFG)
    ucfg = dict(num_res_blocks=list(cfg["num_res_blocks"]), channel_mult=list(cfg["channel_mult"]),
                transformer_depth=list(cfg["transformer_depth"]), transformer_depth_output=list(cfg["transformer_depth_output"]),
                transformer_depth_middle=cfg["transformer_depth_middle"])
    dk = comfy.utils.unet_to_diffusers(ucfg)
    names, _ = unet_names_shapes(cfg)
    have = {n for n, _ in names}
    km = {}
    for n, _ in names:                                            # model_lora_keys_unet, without the "diffusion_model." prefix
        if n.endswith(".weight"):
            km["lora_unet_" + n[:-7].replace(".", "_")] = n
    for k, v in dk.items():
        if k.endswith(".weight") and v in have:
            km["lora_unet_" + k[:-7].replace(".", "_")] = v
    _jdump(km, os.path.join(GOLD, "lora_key_map_sd15.json"))
    g = torch.Generator().manual_seed(31)
    cases = {"linear": (64, 48), "conv3": (32, 16, 3, 3), "conv1": (24, 40, 1, 1)}
    arrs = {}
    mp = types.SimpleNamespace()
    for name, shp in cases.items():
        r = 4
        w = torch.randn(shp, generator=g)
        up = torch.randn((shp[0], r) + ((1, 1) if len(shp) == 4 else ()), generator=g)
        down = torch.randn((r, shp[1]) + (tuple(shp[2:]) if len(shp) == 4 else ()), generator=g)
        for tag, alpha, strength in (("a", None, 1.0), ("b", 2.0, 0.75)):
            lora = {"m.lora_up.weight": up, "m.lora_down.weight": down}
            if alpha is not None:
                lora["m.alpha"] = torch.tensor(alpha)
            patch = comfy.lora.load_lora(lora, {"m": "w"})["w"]
            out_w = comfy.model_patcher.ModelPatcher.calculate_weight(mp, [(strength, patch, 1.0)], w.clone(), "w")
            arrs[f"{name}_{tag}_out"] = out_w.numpy()
        arrs[f"{name}_w"], arrs[f"{name}_up"], arrs[f"{name}_down"] = w.numpy(), up.numpy(), down.numpy()
    save("lora_merge", **arrs)


SECTIONS = dict(math=sec_math, idmap=sec_idmap, overlap=sec_overlap, corrmap=sec_corrmap, noisepool=sec_noisepool,
                sched=sec_sched, unet=sec_unet, vae=sec_vae, e2e=sec_e2e, dump=sec_dump, controlnet=sec_controlnet, legacy=sec_legacy,
                gbufdump=sec_gbufdump, workflow=sec_workflow, sdxl=sec_sdxl, conds=sec_conds, cond_ranges=sec_cond_ranges, vaeenc=sec_vaeenc)

if __name__ == "__main__":
    todo = _ARGV or list(SECTIONS)
    torch.set_num_threads(8)
    for s in todo:
        print("==", s)
        SECTIONS[s]()
