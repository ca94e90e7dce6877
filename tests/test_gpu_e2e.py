"""GPU end-to-end parity: the device-resident sampling loop (UNet plan + CFG + sampler + latent overlap) against
samples produced by the reference's own custom_ksampler stack (tests/golden/e2e_tiny.npz), and CorrespondMap.update
against the reference's CorrespondMap (tests/golden/corrmap_update.npz, bit exact)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def T(a):
    return torch.from_numpy(np.asarray(a))


def _sd(name, seed):
    from stable_renderer_amd import synth
    with open(os.path.join(GOLD, name)) as f:
        k = json.load(f)
    return synth.synth_state_dict([(n, tuple(s)) for n, s in k["names_shapes"]], seed=seed, norm_names=k["norm_names"])


@pytest.mark.parametrize("use_graph", [False, True])
def test_sampling_vs_reference(use_graph):
    from stable_renderer_amd.unet import UNet, SD15_CFG
    from stable_renderer_amd.sampling import DiffusionRunner
    from stable_renderer_amd.corresponder import OverlapCorresponder
    from stable_renderer_amd.corrmap import IDMap
    from stable_renderer_amd.types import EngineData
    d = np.load(os.path.join(GOLD, "e2e_tiny.npz"))
    meta = json.loads(bytes(d["meta"]).decode())
    cfg = dict(SD15_CFG, model_channels=64, context_dim=64)
    net = UNet(_sd("unet_tiny_keys.json", 1), cfg, dtype=torch.float32)
    noise = T(d["noise"])
    N, _, h, w = noise.shape
    ids = T(d["ids"]).cuda()
    for name, m in meta.items():
        ed = EngineData(frame_indices=list(range(N)), id_maps=IDMap(ids))
        run = DiffusionRunner(net, N, h, w, m["cfg"], n_ctx=77, use_graph=use_graph)
        run.set_conditioning(T(d["pos"]), T(d["neg"]))
        cb, n_rand = None, None
        if m["overlap"]:
            oc = OverlapCorresponder(step_finished_inject_ratio=0.5, step_finished_stop_inject_timestep=500)
            n_rand = oc.pre_attn_inject_num_random_frames

            def cb(ctx, oc=oc, ed=ed):
                oc.step_finished(ed, ctx)
        torch.manual_seed(m["rng_seed"])
        out, inj = run.sample(noise, m["steps"], m["sampler"], m["scheduler"], inject_n_rand=n_rand, step_callback=cb)
        torch.cuda.synchronize()
        if m["overlap"]:
            assert inj == m["inj_idx"], name
        ref = T(d[f"{name}_samples"])
        err = (out.cpu() - ref).abs().max().item()
        assert err < 3e-3 * max(1.0, ref.abs().max().item()), (name, err, ref.abs().max().item())


def test_corrmap_update_vs_reference():
    from stable_renderer_amd.corrmap import CorrespondMap
    d = np.load(os.path.join(GOLD, "corrmap_update.npz"))
    meta = json.loads(bytes(d["meta"]).decode())
    for name, m in meta.items():
        cm = CorrespondMap(k=m["k"], height=m["mh"], width=m["mw"])
        if name.startswith("second_"):
            cm.update(T(d["rnd_first_frames"][:1]).cuda(), T(d["rnd_first_ids"][:1]).cuda(), 2, 7, "first")
        masks = T(d[f"{name}_masks"]).cuda() if m["has_masks"] else None
        err = ""
        try:
            cm.update(T(d[f"{name}_frames"]).cuda(), T(d[f"{name}_ids"]).cuda(), m["sprite"], m["material"], m["mode"],
                      masks, m["inverse"], m["ignore"])
        except IndexError:
            err = "IndexError"
        torch.cuda.synchronize()
        assert err == m["err"], name
        assert np.array_equal(cm.writtens.cpu().numpy(), d[f"{name}_writtens"]), name
        assert np.array_equal(cm._values.cpu().numpy(), d[f"{name}_values"]), name          # fp16 bits


def test_corrmap_errors():
    from stable_renderer_amd.corrmap import CorrespondMap
    cm = CorrespondMap(k=3, height=8, width=8)
    ids = torch.zeros(1, 4, 4, 4, dtype=torch.int32).cuda()
    ids[..., 2] = 2048                      # non-AI map index is out of range for k*k maps -> IndexError (reference)
    ids[..., 0] = 1
    with pytest.raises(IndexError):
        cm.update(torch.rand(1, 4, 4, 3).cuda(), ids)
    assert int(cm.writtens.sum()) == 0
    with pytest.raises(ValueError):
        cm.update(torch.rand(4, 4, 3).cuda(), ids[0])      # 3-D id map: hangs forever in the reference


def test_corrmap_dump_load_interchange(tmp_path):
    """on-disk format of CorrespondMap (corrmap.py:738-872): load a directory dumped BY THE REFERENCE, and dump one the
    reference's Load would read identically (PNG bytes decode to the same arrays)."""
    from PIL import Image
    from stable_renderer_amd.corrmap import CorrespondMap
    d = np.load(os.path.join(GOLD, "corrmap_dump_io.npz"))
    m = CorrespondMap.Load(os.path.join(GOLD, "corrmap_dump", "gold"))
    assert (m.k, m.height, m.width) == (2, 8, 8)
    assert np.array_equal(m._values.cpu().numpy(), d["values_back"])            # what the reference's Load returns
    assert np.array_equal(m.writtens.cpu().numpy(), d["writtens_back"])
    m2 = CorrespondMap(k=2, height=8, width=8, name="gold")
    m2._values.copy_(T(d["values_in"]))
    m2._writtens.copy_(T(d["writtens_in"]).to(torch.uint8))
    p = m2.dump(str(tmp_path))
    for i in range(4):
        for fn in (f"{i}.png", f"{i}_written.png"):
            a = np.array(Image.open(os.path.join(p, fn)))
            b = np.array(Image.open(os.path.join(GOLD, "corrmap_dump", "gold", fn)))
            assert np.array_equal(a, b), fn
    z = m2.dump(str(tmp_path), name="z", zip=True)
    m3 = CorrespondMap.Load(z)
    assert np.array_equal(m3._values.cpu().numpy(), d["values_back"])


def test_legacy_overlap_vs_reference():
    """legacy Overlap / ResizeOverlap against the REFERENCE's outputs for all four algorithms: kernel radius 0, and radius 1 where
    the reference's in-place update order matters (replayed in conflict-free levels of its dict order)"""
    import sr_oracle as ORC
    from stable_renderer_amd import legacy_overlap as LO
    d = np.load(os.path.join(GOLD, "legacy_overlap.npz"))
    ids = T(d["ids"]).cuda()
    cm = LO.CorrespondenceMap(ids)
    ref_map = ORC.legacy_corr_map(d["ids"])
    assert len(cm) == len(ref_map)
    frames = [T(d["frames"][i]).cuda() for i in range(3)]
    lat = [T(d["latents"][i]).cuda() for i in range(3)]
    vn = T(d["view_normal"]).cuda()
    algos = dict(average=LO.AverageDistance(), frame=LO.FrameDistance(), pixel=LO.PixelDistance(), view_normal=LO.PerpendicularViewNormal())
    for name, a in algos.items():
        for r in (0, 1):
            kw = dict(alpha_scheduler=LO.Scheduler(interpolate_begin=0.6), kernel_radius_scheduler=LO.Scheduler(interpolate_begin=float(r)), algorithm=a)
            full = LO.Overlap(**kw)(frames, cm, step=1, timestep=500, view_normal_map=vn).cpu().numpy()
            rs = torch.stack(LO.ResizeOverlap(**kw)(lat, cm, step=1, timestep=500, view_normal_map=vn)).cpu().numpy()
            assert np.allclose(full, d[f"full_{name}_r{r}"], atol=2e-5, rtol=1e-5), (name, r, np.abs(full - d[f"full_{name}_r{r}"]).max())
            assert np.allclose(rs, d[f"resize_{name}_r{r}"], atol=2e-5, rtol=1e-5), (name, r)
            if r == 1:                                        # the in-place order is observable: the Jacobi form differs
                jac = ORC.legacy_overlap(d["frames"], d["ids"], 0.6, 1, name, d["view_normal"], sequential=False)
                assert not np.allclose(jac, d[f"full_{name}_r1"], atol=1e-4), name
    # a larger radius: in-place order vs the oracle's sequential loop (itself pinned to the reference at r = 0, 1)
    kw = dict(alpha_scheduler=LO.Scheduler(interpolate_begin=0.45), kernel_radius_scheduler=LO.Scheduler(interpolate_begin=3.0), algorithm=algos["pixel"])
    full = LO.Overlap(**kw)(frames, cm, step=1, timestep=500).cpu().numpy()
    assert np.allclose(full, ORC.legacy_overlap(d["frames"], d["ids"], 0.45, 3, "pixel", None, sequential=True), atol=2e-5)
    lv, off, nl = cm.levels(3)
    assert nl > 1 and int(off[-1]) == len(lv)
    # alpha == 0 returns the input list untouched (overlap.py:203-204)
    z = LO.ResizeOverlap(LO.Scheduler(interpolate_begin=0.0), LO.Scheduler(), LO.AverageDistance())
    assert z(lat, cm, step=1, timestep=500) is lat


def test_sampling_with_two_controlnets_vs_oracle():
    """config-4 shape of the path: two ControlNets (depth + normal style hints, different strengths) chained as
    control_merge does, inside the device sampling loop, vs the oracle composition controlnet_forward x2 -> summed residuals
    -> unet_forward(control=...) under the same euler loop and CFG (each piece pinned to the reference by its own golden)."""
    import sr_oracle as ORC
    from stable_renderer_amd.unet import UNet, SD15_CFG
    from stable_renderer_amd.controlnet import ControlNet
    from stable_renderer_amd.sampling import DiffusionRunner
    cfg = dict(SD15_CFG, model_channels=64, context_dim=64)
    sd_u = _sd("unet_tiny_keys.json", 1)
    sd_c1, sd_c2 = _sd("controlnet_tiny_keys.json", 5), _sd("controlnet_tiny_keys.json", 6)
    g = torch.Generator().manual_seed(8)
    N, h, w, steps, cfg_scale = 2, 16, 16, 3, 3.0
    noise = torch.randn(N, 4, h, w, generator=g)
    pos, neg = torch.randn(1, 77, 64, generator=g), torch.randn(1, 77, 64, generator=g)
    hints = [torch.rand(N, 3, 8 * h, 8 * w, generator=g), torch.rand(N, 3, 8 * h, 8 * w, generator=g)]
    strengths = (0.8, 0.5)

    # ---- oracle
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ms = ORC.ModelSampling()
    sig, _ = ORC.ksampler_sigmas(ms, "normal", steps, None)
    x0 = noise * torch.sqrt(1.0 + sig[0] ** 2.0)

    def denoise_fn(xx, sigma):
        xin, s2 = torch.cat([xx, xx]), torch.cat([sigma, sigma])
        c = torch.cat([neg.expand(N, -1, -1), pos.expand(N, -1, -1)])
        t = ms.timestep(s2).float()
        xc = ORC.eps_input(xin, s2)
        ctrls = [ORC.controlnet_forward(sd, cfg, xc, torch.cat([hh, hh]), t, c, strength=st)
                 for sd, hh, st in zip((sd_c1, sd_c2), hints, strengths)]
        merged = {"output": [a + b for a, b in zip(ctrls[0]["output"], ctrls[1]["output"])],
                  "middle": [ctrls[0]["middle"][0] + ctrls[1]["middle"][0]]}
        out = ORC.unet_forward(sd_u, cfg, xc, t, c, control=merged)
        den = ORC.eps_denoised(xin, out, s2)
        return den[:N] + (den[N:] - den[:N]) * cfg_scale
    ref = ORC.sample_loop(denoise_fn, x0.clone(), sig, "euler", None) / 0.18215

    # ---- HIP path
    net = UNet(sd_u, cfg, dtype=torch.float32)
    cns = [ControlNet(sd_c1, cfg, dtype=torch.float32, strength=strengths[0]), ControlNet(sd_c2, cfg, dtype=torch.float32, strength=strengths[1])]
    for use_graph in (False, True):
        run = DiffusionRunner(net, N, h, w, cfg_scale, use_graph=use_graph, controlnets=cns)
        run.set_conditioning(pos, neg)
        with pytest.raises(ValueError):
            run.sample(noise, steps, "euler", "normal", seed=0)          # hints not set
        run.set_control_hints(hints)
        out, _ = run.sample(noise, steps, "euler", "normal", seed=0)
        torch.cuda.synchronize()
        err = (out.cpu() - ref).abs().max().item()
        assert err < 2e-3 * max(1.0, ref.abs().max().item()), (use_graph, err)
    # and the control really matters
    plain = DiffusionRunner(net, N, h, w, cfg_scale, use_graph=False)
    plain.set_conditioning(pos, neg)
    o2, _ = plain.sample(noise, steps, "euler", "normal", seed=0)
    assert (o2.cpu() - ref).abs().max().item() > 1e-2


def test_controlnets_with_sigma_windows_vs_oracle():
    """ControlNetApplyAdvanced's start / end percent (ControlBase.timestep_percent_range -> timestep_range, comfy/controlnet.py:
    41-62): a net outside its sigma window contributes nothing at that step -- get_control returns the previous nets' residuals
    alone (:184-189).  Net 1 acts from 35 % of the schedule on, net 2 until 55 %: over the 8 steps the UNet sees net 2 alone, both,
    then net 1 alone.  Oracle: the same composition from its pinned pieces, each net's residuals dropped outside its window."""
    import sr_oracle as ORC
    from stable_renderer_amd.unet import UNet, SD15_CFG
    from stable_renderer_amd.controlnet import ControlNet
    from stable_renderer_amd.sampling import DiffusionRunner
    cfg = dict(SD15_CFG, model_channels=64, context_dim=64)
    sd_u = _sd("unet_tiny_keys.json", 1)
    sds = (_sd("controlnet_tiny_keys.json", 5), _sd("controlnet_tiny_keys.json", 6))
    g = torch.Generator().manual_seed(18)
    N, h, w, steps, cfg_scale = 2, 16, 16, 8, 3.0
    noise = torch.randn(N, 4, h, w, generator=g)
    pos, neg = torch.randn(1, 77, 64, generator=g), torch.randn(1, 77, 64, generator=g)
    hints = [torch.rand(N, 3, 8 * h, 8 * w, generator=g), torch.rand(N, 3, 8 * h, 8 * w, generator=g)]
    strengths, ranges = (0.8, 0.6), ((0.35, 1.0), (0.0, 0.55))
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ms = ORC.ModelSampling()
    sig, _ = ORC.ksampler_sigmas(ms, "normal", steps, None)
    x0 = noise * torch.sqrt(1.0 + sig[0] ** 2.0)
    windows = [(ORC.percent_to_sigma(ms, a), ORC.percent_to_sigma(ms, b)) for a, b in ranges]
    seen = []

    def denoise_fn(xx, sigma):
        xin, s2 = torch.cat([xx, xx]), torch.cat([sigma, sigma])
        c = torch.cat([neg.expand(N, -1, -1), pos.expand(N, -1, -1)])
        t = ms.timestep(s2).float()
        xc = ORC.eps_input(xin, s2)
        on = [not (float(sigma[0]) > st or float(sigma[0]) < en) for st, en in windows]      # controlnet.py:184-185
        seen.append(tuple(on))
        ctrls = [ORC.controlnet_forward(sd, cfg, xc, torch.cat([hh, hh]), t, c, strength=sg)
                 for sd, hh, sg, a in zip(sds, hints, strengths, on) if a]
        merged = None
        if ctrls:
            merged = {"output": [sum(parts) for parts in zip(*[cc["output"] for cc in ctrls])],
                      "middle": [sum(cc["middle"][0] for cc in ctrls)]}
        den = ORC.eps_denoised(xin, ORC.unet_forward(sd_u, cfg, xc, t, c, control=merged), s2)
        return den[:N] + (den[N:] - den[:N]) * cfg_scale
    with torch.no_grad():
        ref = ORC.sample_loop(denoise_fn, x0.clone(), sig, "euler", None) / 0.18215
    assert len(set(seen)) == 3 and (True, True) in seen                   # net 2 alone, both, net 1 alone
    net = UNet(sd_u, cfg, dtype=torch.float32)
    cns = [ControlNet(sd, cfg, dtype=torch.float32, strength=sg) for sd, sg in zip(sds, strengths)]
    for cn, r in zip(cns, ranges):
        cn.timestep_percent_range = r
    for use_graph in (False, True):
        run = DiffusionRunner(net, N, h, w, cfg_scale, use_graph=use_graph, controlnets=cns)
        run.set_conditioning(pos, neg)
        run.set_control_hints(hints)
        out, _ = run.sample(noise, steps, "euler", "normal", seed=0)
        torch.cuda.synchronize()
        err = (out.cpu() - ref).abs().max().item()
        assert err < 2e-3 * max(1.0, ref.abs().max().item()), (use_graph, err)
    for cn in cns:                                                          # without the windows the result is a different one
        cn.timestep_percent_range = (0.0, 1.0)
    run = DiffusionRunner(net, N, h, w, cfg_scale, use_graph=False, controlnets=cns)
    run.set_conditioning(pos, neg)
    run.set_control_hints(hints)
    o2, _ = run.sample(noise, steps, "euler", "normal", seed=0)
    assert (o2.cpu() - ref).abs().max().item() > 1e-2


def test_pipeline_with_gbuffer_driven_controlnets():
    """config 4 wiring: depth + normal ControlNets fed by the G-buffer planes of the same views (reference
    resources/example-workflows/miku-control.json: EngineData.depth_maps / normal_maps -> ControlNetApply x2)"""
    from stable_renderer_amd.pipeline import build_sd15_pipeline
    from stable_renderer_amd.unet import SD15_CFG
    cfg = dict(SD15_CFG, model_channels=64, context_dim=64)
    kw = dict(dtype=torch.float32, n_views=2, steps=2, cfg=3.0, W=128, H=128, unet_cfg=cfg, use_graph=False, vae_ch=32)
    torch.manual_seed(3)
    base = build_sd15_pipeline(**kw).call().clone()
    torch.manual_seed(3)
    pipe = build_sd15_pipeline(controls=[("depth", 1.0), ("normal", 0.7)], **kw)
    img = pipe.call().clone()
    torch.cuda.synchronize()
    assert img.shape == (2, 128, 128, 3) and bool(torch.isfinite(img).all())
    hints = pipe.runner._hints
    assert len(hints) == 2 and hints[0].shape == (2, 3, 128, 128)
    assert float(hints[0].max()) > 0.0 and float((hints[0][:, 0] - hints[0][:, 1]).abs().max()) == 0.0   # depth repeated to 3 channels
    assert float((img - base).abs().max()) > 1e-4            # the residuals reach the UNet
    assert int(pipe.scene.corrmap._writtens.sum()) > 0


def test_calls_in_flight_equal_the_sequential_loop(monkeypatch):
    """pipeline.InflightCalls: two calls in flight (thread + stream + plans each, shared weights / scene / corr-map) bake the
    same corr-map, frame for frame, as the plain loop: RNG draws and corr-map updates are taken in call order"""
    from stable_renderer_amd.pipeline import build_sd15_pipeline, InflightCalls
    from stable_renderer_amd.unet import SD15_CFG
    monkeypatch.setenv("SR_AUTOTUNE", "0")
    cfg = dict(SD15_CFG, model_channels=64, context_dim=64)
    kw = dict(dtype=torch.float32, n_views=4, steps=3, cfg=5.0, W=128, H=128, unet_cfg=cfg, vae_ch=32)

    def bake(inflight):
        pipe = build_sd15_pipeline(**kw)
        torch.manual_seed(77)
        if inflight == 1:
            for _ in range(5):
                pipe.call()
        else:
            fl = InflightCalls(pipe, inflight)
            assert fl.pipes[1].unet is pipe.unet and fl.pipes[1].scene is pipe.scene and fl.pipes[1].runner is not pipe.runner
            fl.run(5)
            assert all(p.frame0 == 20 for p in fl.pipes)
        torch.cuda.synchronize()
        cm = pipe.scene.corrmap
        return cm._values.clone(), cm._writtens.clone(), torch.get_rng_state()
    v1, w1, r1 = bake(1)
    v2, w2, r2 = bake(2)

    def same(va, vb):
        # nothing on the path is order dependent (no float atomics; per-slot scratch): in-flight calls are BIT equal to the loop
        return torch.equal(va, vb)
    assert int(w1.sum()) > 0 and torch.equal(w1, w2) and same(v1, v2)
    assert torch.equal(r1, r2)                                  # the global generator ends in the same state
    v3, w3, r3 = bake(3)
    assert torch.equal(w1, w3) and same(v1, v3) and torch.equal(r1, r3)


def test_calls_in_flight_with_a_sampler_that_draws_noise_every_step(monkeypatch):
    """ddpm draws one noise tensor per step from the global generator: with calls in flight the draws of a call are taken inside
    its RNG turn, so the generator sees the sequential loop's sequence (was NotImplementedError until round 3)"""
    from stable_renderer_amd.pipeline import build_sd15_pipeline, InflightCalls
    from stable_renderer_amd.unet import SD15_CFG
    monkeypatch.setenv("SR_AUTOTUNE", "0")
    cfg = dict(SD15_CFG, model_channels=64, context_dim=64)
    kw = dict(dtype=torch.float32, n_views=2, steps=3, cfg=5.0, W=128, H=128, unet_cfg=cfg, vae_ch=32)

    def bake(inflight):
        pipe = build_sd15_pipeline(**kw)
        pipe.sampler = "ddpm"
        torch.manual_seed(78)
        if inflight == 1:
            for _ in range(4):
                pipe.call()
        else:
            InflightCalls(pipe, inflight).run(4)
        torch.cuda.synchronize()
        cm = pipe.scene.corrmap
        return cm._values.clone(), cm._writtens.clone(), torch.get_rng_state()
    v1, w1, r1 = bake(1)
    v2, w2, r2 = bake(2)
    assert int(w1.sum()) > 0 and torch.equal(w1, w2) and torch.equal(v1, v2) and torch.equal(r1, r2)


def test_sampling_with_an_sdxl_family_unet_needs_and_uses_vector_conditioning():
    """DiffusionRunner over the SDXL-family lowering (label_emb): y is required, reaches both CFG halves, changes the result"""
    from stable_renderer_amd import synth
    from stable_renderer_amd.model_shapes import unet_names_shapes
    from stable_renderer_amd.sampling import DiffusionRunner
    from stable_renderer_amd.unet import UNet
    cfg = dict(in_channels=4, out_channels=4, model_channels=64, num_res_blocks=[2, 2, 2], channel_mult=[1, 2, 4],
               transformer_depth=[0, 0, 1, 1, 2, 2], transformer_depth_middle=2, transformer_depth_output=[0, 0, 0, 1, 1, 1, 2, 2, 2],
               context_dim=128, num_heads=-1, num_head_channels=32, use_linear_in_transformer=True, adm_in_channels=192)
    ns, norms = unet_names_shapes(cfg)
    net = UNet(synth.synth_state_dict(ns, seed=6, norm_names=norms), cfg, dtype=torch.float32)
    g = torch.Generator().manual_seed(2)
    noise, pos, neg = torch.randn(2, 4, 16, 16, generator=g), torch.randn(1, 77, 128, generator=g), torch.randn(1, 77, 128, generator=g)
    r = DiffusionRunner(net, 2, 16, 16, 5.0, use_graph=False)
    r.set_conditioning(pos, neg)
    with pytest.raises(ValueError, match="vector conditioning"):
        r.sample(noise, 2, "euler", "normal", seed=1)
    outs = []
    for s in (0, 0, 1):
        r.set_vector_conditioning(torch.randn(1, 192, generator=torch.Generator().manual_seed(s)))
        out, _ = r.sample(noise, 2, "euler", "normal", seed=1)
        outs.append(out.clone())
    assert bool(torch.isfinite(outs[0]).all()) and torch.equal(outs[0], outs[1]) and not torch.equal(outs[0], outs[2])


def _cond_lists(ctx_dim, H, W):
    """positive / negative conditioning LISTS in ComfyUI's format: strengths, a blob mask, a box mask turned into an area,
    a percentage area"""
    def c(seed):
        return torch.randn(1, 77, ctx_dim, generator=torch.Generator().manual_seed(seed))
    g = torch.Generator().manual_seed(5)
    blob = torch.nn.functional.avg_pool2d(torch.randn(1, 1, H, W, generator=g).abs(), 31, 1, 15)[0]
    blob = (blob > blob.median()).float()
    box = torch.zeros(1, H, W)
    box[:, H // 5: H * 3 // 4, W // 4: W * 4 // 5] = 1.0
    return {
        "masks_and_strengths": ([[c(1), {"strength": 1.3}], [c(2), {"mask": blob, "mask_strength": 0.7, "set_area_to_bounds": False}]],
                                [[c(3), {}]], 6.0),
        "areas": ([[c(4), {}], [c(5), {"area": ("percentage", 0.5, 0.5, 0.25, 0.25), "strength": 0.9}],
                   [c(6), {"mask": box, "mask_strength": 1.0, "set_area_to_bounds": True}]], [[c(7), {}], [c(8), {"strength": 0.5}]], 4.0),
        "cfg1": ([[c(9), {}], [c(10), {"mask": blob, "mask_strength": 1.0, "set_area_to_bounds": False}]], [[c(11), {}]], 1.0),
        # ConditioningSetTimestepRange (comfyUI/nodes.py:270-285): the second prompt acts in the middle of the schedule only, the
        # negative prompt until 60 % of it -- entries outside their sigma window are not run (samplers.py:60-67), so the model
        # calls change from step to step (B = 2N, 3N, 2N, N over the 8 steps)
        "ranges": ([[c(12), {}], [c(13), {"start_percent": 0.3, "end_percent": 0.7, "strength": 0.8}]], [[c(14), {"end_percent": 0.6}]], 5.0),
    }


@pytest.mark.parametrize("case", ["masks_and_strengths", "areas", "cfg1", "ranges"])
def test_conditioning_lists_with_masks_and_areas_vs_oracle(case):
    """calc_cond_uncond_batch's composition on the HIP path (sr_cond_crop_scale / sr_cond_accumulate / sr_cfg_combine + one UNet
    plan per model-call shape) against the oracle, whose composition arithmetic is pinned to the reference by the toy-model golden
    (tests/test_oracle_golden.py::test_cond_composition_vs_reference)"""
    import sr_oracle as ORC
    from stable_renderer_amd.conditioning import entries_of
    from stable_renderer_amd.sampling import DiffusionRunner
    from stable_renderer_amd.unet import UNet, SD15_CFG
    cfg = dict(SD15_CFG, model_channels=64, context_dim=64)
    sd = _sd("unet_tiny_keys.json", 1)
    net = UNet(sd, cfg, dtype=torch.float32)
    N, h, w = 2, 16, 24
    pos, neg, scale = _cond_lists(64, h * 8, w * 8)[case]
    noise = torch.randn(N, 4, h, w, generator=torch.Generator().manual_seed(21))
    run = DiffusionRunner(net, N, h, w, scale, n_ctx=77, use_graph=False)
    run.set_cond_entries(entries_of(pos), entries_of(neg))
    steps = 8 if case == "ranges" else 3
    torch.manual_seed(3)
    out, _ = run.sample(noise, steps, "euler", "normal")
    torch.cuda.synchronize()
    if case == "ranges":                                       # the sigma windows really cut the schedule into different model calls
        assert len({k for k in run._general["variants"]}) >= 3
    torch.manual_seed(3)
    with torch.no_grad():
        ref, _ = ORC.sample_frames(sd, cfg, noise, None, None, None, steps, scale, "euler", "normal",
                                   cond_entries=(entries_of(pos), entries_of(neg)))
    err = (out.cpu() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    assert err < 2e-3, (case, err)
    # the composition matters: the plain [neg | pos] run of the first entries gives a different latent
    run.set_conditioning(pos[0][0], neg[0][0])
    torch.manual_seed(3)
    plain, _ = run.sample(noise, steps, "euler", "normal")
    assert (plain.cpu() - ref).abs().max().item() > 1e-2


def test_masked_text_nodes_feed_the_sampler():
    """MaskedTextEncode / SceneTextEncode(merge=False, idmap) -> custom_ksampler: per-sprite prompts act through id-map masks"""
    from stable_renderer_amd import graph_nodes as GN, nodes as N
    from stable_renderer_amd.corrmap import IDMap
    from stable_renderer_amd.types import Sprite, SpriteInfos, EnvPrompt, LATENT
    from stable_renderer_amd.unet import UNet, SD15_CFG
    cfg = dict(SD15_CFG, model_channels=64, context_dim=64)
    model = N.MODEL(UNet(_sd("unet_tiny_keys.json", 1), cfg, dtype=torch.float32))
    clip = GN.SyntheticCLIP(ctx_dim=64)
    ids = torch.zeros(1, 128, 128, 4, dtype=torch.int32)
    ids[:, 20:90, 10:70, 0] = 1
    ids[:, 60:120, 60:125, 0] = 2
    sprites = SpriteInfos({1: Sprite(1, "a red ball", 1.0, "blurry", 0.5), 2: Sprite(2, "a blue cube", 0.8)})
    pos, neg = GN.SceneTextEncode()(clip, sprites, [EnvPrompt("studio light", "noise")], merge=False, idmap=IDMap(ids.cuda()))
    assert len(pos) == 4 and len(neg) == 1                              # 3 sprite prompts (both signs, as the reference) + env
    assert [("mask" in e[1]) for e in pos] == [True, True, True, False]
    assert pos[1][1]["mask_strength"] == 0.5 and pos[2][1]["mask_strength"] == 0.8
    assert float(pos[0][1]["mask"].sum()) == 70 * 60 - 30 * 10            # sprite 2 covers a corner of sprite 1
    one = GN.MaskedTextEncode()(clip, "a cat", mask=torch.ones(128, 128), inverse_mask=True, strength=0.3, mode="set_cond_area")
    assert one[0][1]["set_area_to_bounds"] and float(one[0][1]["mask"].abs().sum()) == 0 and one[0][1]["mask"].dim() == 3
    lat = LATENT(samples=torch.zeros(1, 4, 16, 16))
    torch.manual_seed(0)
    a = N.custom_ksampler(model, 5, 2, 4.0, "euler", "normal", pos, neg, lat)[0]["samples"]
    b = N.custom_ksampler(model, 5, 2, 4.0, "euler", "normal", [pos[3]], neg, lat)[0]["samples"]
    assert torch.isfinite(a).all() and (a - b).abs().max().item() > 1e-3


def _first_injected_index(rng_seed, noise_shape, B0):
    """the index pre_atten_inject draws on the run's first UNet call, from the global generator in the reference's order:
    custom_ksampler's seed, 'ddim' reseeding with seed + 1 and drawing one noise tensor, then randint(1, B0)"""
    torch.manual_seed(rng_seed)
    seed = int(torch.randint(0, 2 ** 32, (1,)).item())
    g = torch.manual_seed(seed + 1)
    torch.randn(noise_shape, generator=g, device="cpu")
    return int(torch.randint(1, B0, (1,)).item())


def test_kv_injection_with_several_model_calls_vs_oracle():
    """OverlapCorresponder + conditioning AREAS: calc_cond_uncond_batch makes several model calls per step, every one of them
    runs the UNet with the corresponder, and the indices drawn on the FIRST call (randint(1, batch of that call),
    corresponder.py:204-205) are reused by the others -- k_context[idx] raises IndexError in a smaller batch (:207-214).
    Both outcomes against the oracle (was NotImplementedError until round 3)"""
    import sr_oracle as ORC
    from stable_renderer_amd import ops as O
    from stable_renderer_amd.conditioning import entries_of
    from stable_renderer_amd.sampling import DiffusionRunner
    from stable_renderer_amd.unet import UNet, SD15_CFG
    cfg = dict(SD15_CFG, model_channels=64, context_dim=64)
    sd = _sd("unet_tiny_keys.json", 1)
    net = UNet(sd, cfg, dtype=torch.float32)
    N, h, w = 2, 16, 16                                   # (square: the overlap's x/H, y/W index rule needs it, corresponder.py:321-330)
    pos, neg, scale = _cond_lists(64, h * 8, w * 8)["areas"]
    g = torch.Generator().manual_seed(21)
    noise = torch.randn(N, 4, h, w, generator=g)
    ids = torch.zeros(N, h * 8, w * 8, 4, dtype=torch.int32)
    ids[..., 0] = 1
    ids[..., 3] = torch.randint(0, 400, (N, h * 8, w * 8), generator=g, dtype=torch.int32)
    ids[torch.rand(N, h * 8, w * 8, generator=g) < 0.2] = 0
    idx = O.OverlapIndex(ids.cuda(), h, w)
    run = DiffusionRunner(net, N, h, w, scale, n_ctx=77, use_graph=False)
    run.set_cond_entries(entries_of(pos), entries_of(neg))
    B0 = N * run._build_general(1)["groups"][0]["chunks"]
    Bmin = N * min(gr["chunks"] for gr in run._general["groups"])
    assert len(run._general["groups"]) > 1 and Bmin < B0
    ok_seed = next(s_ for s_ in range(100) if _first_injected_index(s_, tuple(noise.shape), B0) < Bmin)
    bad_seed = next(s_ for s_ in range(100) if _first_injected_index(s_, tuple(noise.shape), B0) >= Bmin)

    def cb(ctx):
        if ctx.timestep >= 500:
            idx.step(ctx.noise, 0.5)
    torch.manual_seed(ok_seed)
    out, inj = run.sample(noise, 3, "ddim", "normal", inject_n_rand=1, step_callback=cb)
    torch.cuda.synchronize()
    torch.manual_seed(ok_seed)
    with torch.no_grad():
        ref, rinj = ORC.sample_frames(sd, cfg, noise, None, None, ids.numpy(), 3, scale, "ddim", "normal",
                                      overlap=dict(ratio=0.5, stop=500, n_rand=1), cond_entries=(entries_of(pos), entries_of(neg)))
    assert inj == [int(i) for i in rinj]
    err = (out.cpu() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    assert err < 2e-3, err
    torch.manual_seed(bad_seed)                                 # an index the smaller model calls do not have
    with pytest.raises(IndexError):
        run.sample(noise, 3, "ddim", "normal", inject_n_rand=1, step_callback=cb)
    torch.manual_seed(bad_seed)
    with pytest.raises(IndexError), torch.no_grad():
        ORC.sample_frames(sd, cfg, noise, None, None, ids.numpy(), 3, scale, "ddim", "normal",
                          overlap=dict(ratio=0.5, stop=500, n_rand=1), cond_entries=(entries_of(pos), entries_of(neg)))


def test_controlnets_with_conditioning_areas_vs_oracle():
    """ControlNets + conditioning AREAS: the control net of a model call on a crop sees the cropped latent and the WHOLE hint
    resized to 8x the crop (ControlNet.get_control -> common_upscale(..., 'nearest-exact', 'center'), comfy/controlnet.py:193-201,
    samplers.py:271-276).  Oracle: the same composition from its pinned pieces (was NotImplementedError until round 3)"""
    import sr_oracle as ORC
    from stable_renderer_amd.conditioning import entries_of
    from stable_renderer_amd.controlnet import ControlNet
    from stable_renderer_amd.sampling import DiffusionRunner
    from stable_renderer_amd.unet import UNet, SD15_CFG
    cfg = dict(SD15_CFG, model_channels=64, context_dim=64)
    sd_u, sd_c = _sd("unet_tiny_keys.json", 1), _sd("controlnet_tiny_keys.json", 5)
    N, h, w, steps, scale, strength = 2, 16, 24, 3, 4.0, 0.8
    g = torch.Generator().manual_seed(31)
    noise = torch.randn(N, 4, h, w, generator=g)
    hint = torch.rand(N, 3, 8 * h, 8 * w, generator=g)

    def c(seed):
        return torch.randn(1, 77, 64, generator=torch.Generator().manual_seed(seed))
    pos = [[c(4), {}], [c(5), {"area": ("percentage", 0.5, 0.5, 0.25, 0.25), "strength": 0.9}]]
    neg = [[c(7), {}]]
    net = UNet(sd_u, cfg, dtype=torch.float32)
    run = DiffusionRunner(net, N, h, w, scale, n_ctx=77, use_graph=False, controlnets=[ControlNet(sd_c, cfg, dtype=torch.float32, strength=strength)])
    run.set_cond_entries(entries_of(pos), entries_of(neg))
    run.set_control_hints([hint])
    torch.manual_seed(3)
    out, _ = run.sample(noise, steps, "euler", "normal")
    torch.cuda.synchronize()
    assert len(run._general["groups"]) == 2

    ms = ORC.ModelSampling()
    sig, _ = ORC.ksampler_sigmas(ms, "normal", steps, None)
    entries = ORC.prepare_cond_entries(entries_of(pos), entries_of(neg), h, w)

    def model_fn(xin, s2, ctx):
        hh = hint if xin.shape[2:] == (h, w) else ORC.common_upscale_center(hint, xin.shape[3] * 8, xin.shape[2] * 8)
        hh = torch.cat([hh] * (xin.shape[0] // N))
        t = ms.timestep(s2).float()
        xc = ORC.eps_input(xin, s2)
        ctrl = ORC.controlnet_forward(sd_c, cfg, xc, hh, t, ctx, strength=strength)
        return ORC.eps_denoised(xin, ORC.unet_forward(sd_u, cfg, xc, t, ctx, control=ctrl), s2)
    torch.manual_seed(3)
    x0 = noise * torch.sqrt(1.0 + sig[0] ** 2.0)
    with torch.no_grad():
        ref = ORC.sample_loop(lambda xx, sigma: ORC.sampling_function(model_fn, xx, sigma, entries[1], entries[0], scale), x0.clone(), sig,
                              "euler", None) / 0.18215
    err = (out.cpu() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    assert err < 2e-3, err
