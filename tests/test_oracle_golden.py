"""Pins the CPU oracle (oracle/sr_oracle.py) against golden vectors produced by the reference itself
(oracle/gen_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

import sr_oracle as O
from stable_renderer_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def T(a):
    return torch.from_numpy(np.asarray(a))


def meta_of(d):
    return json.loads(bytes(d["meta"]).decode())


def test_adain(gold):
    d = gold("adain")
    for s in range(2):
        o = O.adain(T(d[f"nchw_c{s}"]), T(d[f"nchw_s{s}"]))
        assert torch.allclose(o, T(d[f"nchw_o{s}"]), atol=1e-6, rtol=1e-6)
    o = O.adain(T(d["nhwc_c"]), T(d["nhwc_s"]), mode="NHWC")
    assert torch.allclose(o, T(d["nhwc_o"]), atol=1e-6)
    o = O.adain(T(d["half_c"]), T(d["half_s"]))
    assert o.dtype == torch.float16 and torch.equal(o, T(d["half_o"]))


def test_groupby(gold):
    d = gold("groupby")
    t = d["doc_t"]
    a0, _ = O.group_by_average(t[:, 0], t[:, 1:3])
    assert np.array_equal(a0, d["doc_a0"])
    a1, u1 = O.group_by_average(t[:, 1], t[:, 0:1])
    assert np.array_equal(a1, d["doc_a1"]) and np.array_equal(u1, d["doc_u1"])
    tr = d["rnd_t"]
    ar, ur = O.group_by_average(tr[:, 4], tr[:, :4])
    assert np.array_equal(ur, d["rnd_u"])
    assert np.array_equal(ar, d["rnd_a"])          # sequential fp32 accumulation: bit exact


def test_idmap(gold):
    d = gold("idmap")
    assert np.array_equal(O.idmap_masks(d["ids"]), d["masks"])
    assert np.array_equal(O.vertex_screen_info(d["ids"]), d["vsi"])
    # two frames of the reference's own dumped sphere id maps (SURVEY 8c golden 3)
    assert np.array_equal(O.idmap_masks(d["sphere_ids"]), d["sphere_masks"])
    assert np.array_equal(O.vertex_screen_info(d["sphere_ids"]), d["sphere_vsi"])


def test_overlap_step(gold):
    d = gold("overlap_step")
    for name, m in meta_of(d).items():
        x = T(d[f"{name}_x"])
        if m["timestep"] < m["stop"]:
            out = x
        else:
            out = O.overlap_step(x, d[f"{name}_ids"], m["ratio"])
        assert torch.allclose(out, T(d[f"{name}_out"]), atol=2e-6, rtol=1e-6), name


def test_pre_attn_inject(gold):
    d = gold("pre_attn_inject")
    q, k, v = O.pre_atten_inject(T(d["n"]), d["idx"])
    assert torch.equal(q, T(d["q"])) and torch.equal(k, T(d["k"])) and torch.equal(v, T(d["v"]))


def test_corrmap_update(gold):
    d = gold("corrmap_update")
    meta = meta_of(d)
    for name, m in meta.items():
        V = m["mh"] * m["mw"]
        values = np.zeros((m["k"] ** 2, V, 4), np.float16)
        writtens = np.zeros((m["k"] ** 2, V), bool)
        frames, ids = d[f"{name}_frames"], d[f"{name}_ids"]
        if name.startswith("second_"):
            O.corrmap_update(values, writtens, d["rnd_first_frames"][:1], d["rnd_first_ids"][:1], 2, 7, "first")
        masks = d[f"{name}_masks"] if m["has_masks"] else None
        err = ""
        try:
            O.corrmap_update(values, writtens, frames, ids, m["sprite"], m["material"], m["mode"], masks,
                             m["inverse"], m["ignore"])
        except IndexError:
            err = "IndexError"
        assert err == m["err"], name
        assert np.array_equal(writtens, d[f"{name}_writtens"]), name
        assert np.array_equal(values, d[f"{name}_values"]), name


def test_noise_pool(gold):
    d = gold("noise_pool")
    for i in range(2):
        pooled, out = O.noise_pool(T(d[f"n{i}_noise"]), T(d[f"n{i}_alpha"]), T(d[f"n{i}_bg"]))
        assert torch.equal(pooled, T(d[f"n{i}_pooled"]))
        assert torch.allclose(out, T(d[f"n{i}_out"]), atol=1e-6)


def test_schedules(gold):
    d = gold("sampling")
    ms = O.ModelSampling()
    assert torch.allclose(ms.sigmas, T(d["sigmas_table"]), rtol=1e-6)
    for sch in ["normal", "sgm_uniform", "karras", "simple", "ddim_uniform", "exponential"]:
        for steps in (4, 20):
            s = O.scheduler_sigmas(ms, sch, steps)
            assert torch.allclose(s, T(d[f"{sch}_{steps}"]), rtol=2e-6, atol=1e-7), (sch, steps)
    for sch, steps, den in [("normal", 20, 1.0), ("normal", 20, 0.55), ("sgm_uniform", 4, 0.55), ("karras", 20, 0.7)]:
        s, ts = O.ksampler_sigmas(ms, sch, steps, den)
        assert torch.allclose(s, T(d[f"ks_{sch}_{steps}_{int(den*100)}_sigmas"]), rtol=2e-6, atol=1e-7)
        assert ts == d[f"ks_{sch}_{steps}_{int(den*100)}_timesteps"].tolist()
    assert torch.equal(ms.timestep(T(d["ts_in"])), T(d["ts_out"]))
    assert torch.allclose(ms.sigma(T(d["sg_in"])), T(d["sg_out"]), rtol=1e-6)
    s2 = torch.tensor([3.0, 3.0])
    assert torch.allclose(O.eps_input(T(d["eps_x"]), s2), T(d["eps_in"]), rtol=1e-6)
    assert torch.allclose(O.eps_denoised(T(d["eps_x"]), T(d["eps_mo"]), s2), T(d["eps_den"]), rtol=1e-6)


def test_sampler_trajectories(gold):
    d = gold("sampling")

    def toy(x, sigma):
        return torch.tanh(x) * 0.5 / (1 + sigma.view(-1, 1, 1, 1))
    sig, x0 = T(d["traj_sigmas"]), T(d["traj_x0"])
    assert torch.allclose(O.sample_loop(toy, x0.clone(), sig, "euler"), T(d["traj_euler"]), atol=1e-5)
    torch.manual_seed(77)
    assert torch.allclose(O.sample_loop(toy, x0.clone(), sig, "ddpm"), T(d["traj_ddpm"]), atol=1e-5)
    torch.manual_seed(78)
    assert torch.allclose(O.sample_loop(toy, x0.clone(), sig, "lcm"), T(d["traj_lcm"]), atol=1e-5)


def _sd_from_keys(name, seed):
    with open(os.path.join(GOLD, name)) as f:
        k = json.load(f)
    return synth.synth_state_dict([(n, tuple(s)) for n, s in k["names_shapes"]], seed=seed, norm_names=k["norm_names"])


TINY = dict(O.SD15_CFG, model_channels=64, context_dim=64)


def test_unet_tiny(gold):
    d = gold("unet_tiny")
    sd = _sd_from_keys("unet_tiny_keys.json", 1)
    with torch.no_grad():
        y = O.unet_forward(sd, TINY, T(d["x"]), T(d["t"]), T(d["ctx"]))
        assert torch.allclose(y, T(d["y"]), atol=2e-4, rtol=1e-4), (y - T(d["y"])).abs().max()
        yi = O.unet_forward(sd, TINY, T(d["x"]), T(d["t"]), T(d["ctx"]), inject_idx=d["inj_idx"])
        assert torch.allclose(yi, T(d["y_inj"]), atol=2e-4, rtol=1e-4)
        assert not torch.allclose(yi, y, atol=1e-3)
        yo = O.unet_forward(sd, TINY, T(d["x_odd"]), T(d["t"])[:2], T(d["ctx"])[:2])       # 10x12: Upsample targets the skip's size
        assert torch.allclose(yo, T(d["y_odd"]), atol=2e-4, rtol=1e-4), (yo - T(d["y_odd"])).abs().max()


@pytest.mark.slow
def test_unet_sd15(gold):
    d = gold("unet_sd15_16")
    sd = _sd_from_keys("unet_sd15_keys.json", 0)
    with torch.no_grad():
        y = O.unet_forward(sd, O.SD15_CFG, T(d["x"]), T(d["t"]), T(d["ctx"]))
    assert torch.allclose(y, T(d["y"]), atol=5e-4, rtol=1e-3), (y - T(d["y"])).abs().max()


def test_vae_decoder(gold):
    d = gold("vae_dec")
    sd = _sd_from_keys("vae_dec_keys.json", 2)
    with torch.no_grad():
        y = O.vae_decoder(sd, T(d["z"]))
        img = O.vae_decode_image(sd, T(d["z"]))
    assert torch.allclose(y, T(d["y"]), atol=5e-4, rtol=1e-3), (y - T(d["y"])).abs().max()
    assert torch.allclose(img, T(d["img"]), atol=5e-4)


def test_e2e_sampling(gold):
    p = os.path.join(GOLD, "e2e_tiny.npz")
    if not os.path.exists(p):
        pytest.skip("e2e golden not generated")
    d = gold("e2e_tiny")
    sd = _sd_from_keys("unet_tiny_keys.json", 1)
    for name, m in meta_of(d).items():
        torch.manual_seed(m["rng_seed"])
        ov = dict(ratio=0.5, stop=500, n_rand=1) if m["overlap"] else None
        with torch.no_grad():
            s, idx = O.sample_frames(sd, TINY, T(d["noise"]), T(d["pos"]), T(d["neg"]), d["ids"], m["steps"], m["cfg"],
                                     m["sampler"], m["scheduler"], overlap=ov)
        if m["overlap"]:
            assert idx.tolist() == m["inj_idx"], name
        ref = T(d[f"{name}_samples"])
        err = (s - ref).abs().max().item()
        assert err < 2e-3 * max(1.0, ref.abs().max().item()), (name, err)


def test_controlnet(gold):
    d = gold("controlnet_tiny")
    sd_c = _sd_from_keys("controlnet_tiny_keys.json", 5)
    sd_u = _sd_from_keys("unet_tiny_keys.json", 1)
    with torch.no_grad():
        c = O.controlnet_forward(sd_c, TINY, T(d["x"]), T(d["hint"]), T(d["t"]), T(d["ctx"]), strength=0.8)
        assert torch.allclose(c["middle"][0] / 0.8, T(d["mid"]), atol=2e-4, rtol=1e-4)
        assert torch.allclose(c["output"][0] / 0.8, T(d["out0"]), atol=2e-4, rtol=1e-4)
        assert torch.allclose(c["output"][11] / 0.8, T(d["out11"]), atol=2e-4, rtol=1e-4)
        y = O.unet_forward(sd_u, TINY, T(d["x"]), T(d["t"]), T(d["ctx"]), control=c)
    assert torch.allclose(y, T(d["y"]), atol=3e-4, rtol=1e-4), (y - T(d["y"])).abs().max()


def test_legacy_overlap(gold):
    d = gold("legacy_overlap")
    ids, frames, lat, vn = d["ids"], d["frames"], d["latents"], d["view_normal"]
    for algo in ("average", "frame", "pixel", "view_normal"):
        for r in (0, 1):
            o = O.legacy_overlap(frames, ids, 0.6, r, algo, vn, sequential=True)
            assert np.allclose(o, d[f"full_{algo}_r{r}"], atol=2e-5, rtol=1e-5), (algo, r)
            o2 = O.legacy_resize_overlap(lat, ids, 0.6, r, algo, vn, sequential=True)
            assert np.allclose(o2, d[f"resize_{algo}_r{r}"], atol=2e-5, rtol=1e-5), (algo, r)
        # radius 0 is order independent: the parallel (Jacobi) form is identical
        assert np.allclose(O.legacy_overlap(frames, ids, 0.6, 0, algo, vn, sequential=False), d[f"full_{algo}_r0"], atol=2e-5)


def load_cond_case(d, meta, name):
    """-> (pos entries, neg entries) in the oracle's / product's entry format from the cond_compose golden"""
    out = {}
    for kind in ("pos", "neg"):
        lst = []
        for i, ex in enumerate(meta[name]["entries"][kind]):
            e = {k: (tuple(v) if isinstance(v, list) else v) for k, v in ex.items() if k != "has_mask"}
            e["cond"] = T(d[f"{name}_{kind}{i}_c"])
            if ex.get("has_mask"):
                e["mask"] = T(d[f"{name}_{kind}{i}_mask"])
            lst.append(e)
        out[kind] = lst
    return out["pos"], out["neg"]


def toy_model(xin, sg, ctx):
    b = xin.shape[0]
    return (xin * 0.5 + ctx.mean(dim=(1, 2)).view(-1, 1, 1, 1) + 0.01 * sg.view(-1, 1, 1, 1)
            + 0.001 * torch.arange(b, dtype=torch.float32).view(-1, 1, 1, 1))


def test_cond_composition_vs_reference(gold):
    """masks / strengths / areas / opposite-area entries / batching order / CFG: the oracle's restatement against the reference's
    calc_cond_uncond_batch + sampling_function driven by the same toy model (golden cond_compose.npz)"""
    d = gold("cond_compose")
    meta = meta_of(d)
    x, sigma = T(d["x"]), T(d["sigma"])
    for name, m in meta.items():
        pos, neg = load_cond_case(d, meta, name)
        p, n = O.prepare_cond_entries(pos, neg, 16, 24)
        assert len(p) == m["n_pos"] and len(n) == m["n_neg"], name
        assert [None if e.get("area") is None else list(e["area"]) for e in p] == m["areas_pos"], name
        assert [None if e.get("area") is None else list(e["area"]) for e in n] == m["areas_neg"], name
        c, u, batches = O.calc_cond_uncond_batch(toy_model, p, n, x, sigma)
        assert [[1 if k == "neg" else 0 for k, _ in b] for b in batches] == [cl[1] for cl in m["calls"]], name
        assert torch.allclose(c, T(d[f"{name}_cond"]), atol=1e-6, rtol=1e-6), name
        assert torch.allclose(u, T(d[f"{name}_uncond"]), atol=1e-6, rtol=1e-6), name
        r = O.sampling_function(toy_model, x, sigma, n, p, m["scale"])
        assert torch.allclose(r, T(d[f"{name}_cfg"]), atol=1e-5, rtol=1e-6), name


def test_cond_timestep_ranges_vs_reference(gold):
    """ConditioningSetTimestepRange windows: calculate_start_end_timesteps + get_area_and_mult's sigma test (comfy/samplers.py:60-67,
    578-602) -- the reference's sampling_function at sigmas inside / outside the windows (golden cond_ranges.npz), the oracle's
    restatement and the product's host-side window arithmetic (conditioning.with_timestep_ranges / entry_active)"""
    from stable_renderer_amd import conditioning as CD
    from stable_renderer_amd.sampling import ModelSamplingDiscrete
    d = gold("cond_ranges")
    x, scale = T(d["x"]), float(d["scale"])
    pos = [dict(cond=T(d["pos0_c"])), dict(cond=T(d["pos1_c"]), start_percent=0.3, end_percent=0.7, strength=0.8)]
    neg = [dict(cond=T(d["neg0_c"]), start_percent=0.0, end_percent=0.5)]
    ms = O.ModelSampling()
    p, n = O.prepare_cond_entries(pos, neg, 16, 24, ms)
    win = [[e.get("timestep_start", -1.0), e.get("timestep_end", -1.0)] for e in p + n]
    assert np.allclose(np.array(win), d["windows"], rtol=1e-6)
    p2, n2 = CD.prepare(pos, neg, 16, 24, ModelSamplingDiscrete())                  # the product's host arithmetic: same windows
    assert np.allclose(np.array([[e.get("timestep_start", -1.0), e.get("timestep_end", -1.0)] for e in p2 + n2]), d["windows"], rtol=1e-6)
    seen = set()
    for i, sg in enumerate(d["sigmas"].tolist()):
        r = O.sampling_function(toy_model, x, torch.tensor([sg, sg]), n, p, scale)
        assert torch.allclose(r, T(d[f"cfg_{i}"]), atol=1e-5, rtol=1e-6), sg
        act = tuple(O.entry_active(e, sg) for e in p + n)
        assert act == tuple(CD.entry_active(e, sg) for e in p2 + n2)
        seen.add(act)
    assert len(seen) >= 4                                                          # the sigmas exercise four different active sets
    assert not CD.is_plain([dict(cond=x, end_percent=0.5)]) and CD.is_plain([dict(cond=x, start_percent=0.0, end_percent=1.0)])


def test_vae_encoder_vs_reference(gold):
    """Encoder + quant_conv + posterior sample (VAE.encode, sd.py:353-371) against the reference AutoencoderKL"""
    d = gold("vae_enc")
    with open(os.path.join(GOLD, "vae_enc_keys.json")) as f:
        k = json.load(f)
    sd = synth.synth_state_dict([(n, tuple(s)) for n, s in k["names_shapes"]], seed=3, norm_names=k["norm_names"])
    with torch.no_grad():
        mom = O.vae_encoder_moments(sd, T(d["pixels"]).movedim(-1, 1) * 2.0 - 1.0)
        assert torch.allclose(mom, T(d["moments"]), atol=2e-4, rtol=1e-4)
        torch.manual_seed(31)
        z = O.vae_encode(sd, T(d["pixels"]))                       # same global-generator draw as the reference
        assert torch.allclose(z, T(d["z"]), atol=2e-4, rtol=1e-4)
        assert torch.allclose(O.vae_encode(sd, T(d["pixels"]), noise=T(d["noise"])), z, atol=1e-6)
