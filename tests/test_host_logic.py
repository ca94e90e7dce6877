"""CPU tests of the host logic: schedules vs the reference goldens, checkpoint key enumeration, scene maths, the C-ABI
library (loads, exports every symbol of include/sr_hip.h; no compute without a GPU)."""
import json
import os
import re

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_model_key_enumeration_matches_reference_state_dicts():
    from stable_renderer_amd.model_shapes import unet_names_shapes, vae_decoder_names_shapes
    from stable_renderer_amd.unet import SD15_CFG
    for fn, cfg in (("unet_sd15_keys.json", SD15_CFG), ("unet_tiny_keys.json", dict(SD15_CFG, model_channels=64, context_dim=64))):
        with open(os.path.join(GOLD, fn)) as f:
            k = json.load(f)
        ns, norms = unet_names_shapes(cfg)
        assert [(n, list(s)) for n, s in ns] == [(n, list(s)) for n, s in k["names_shapes"]]
        assert sorted(norms) == sorted(k["norm_names"])
    with open(os.path.join(GOLD, "vae_dec_keys.json")) as f:
        k = json.load(f)
    ns, norms = vae_decoder_names_shapes()
    assert [(n, list(s)) for n, s in ns] == [(n, list(s)) for n, s in k["names_shapes"]]
    assert sorted(norms) == sorted(k["norm_names"])
    from stable_renderer_amd.model_shapes import controlnet_names_shapes
    with open(os.path.join(GOLD, "controlnet_tiny_keys.json")) as f:          # cldm.ControlNet.state_dict() order (seed positions)
        k = json.load(f)
    ns, norms = controlnet_names_shapes(dict(SD15_CFG, model_channels=64, context_dim=64))
    assert [(n, list(s)) for n, s in ns] == [(n, list(s)) for n, s in k["names_shapes"]]
    assert sorted(norms) == sorted(k["norm_names"])
    from stable_renderer_amd.model_shapes import vae_encoder_names_shapes
    with open(os.path.join(GOLD, "vae_enc_keys.json")) as f:
        k = json.load(f)
    ns, norms = vae_encoder_names_shapes()
    assert [(n, list(s)) for n, s in ns] == [(n, list(s)) for n, s in k["names_shapes"]]
    assert sorted(norms) == sorted(k["norm_names"])


def test_schedules_match_reference():
    from stable_renderer_amd import sampling as S
    d = np.load(os.path.join(GOLD, "sampling.npz"))
    ms = S.ModelSamplingDiscrete()
    assert torch.allclose(ms.sigmas, torch.from_numpy(d["sigmas_table"]), rtol=1e-6)
    for sch in S.SCHEDULER_NAMES:
        for steps in (4, 20):
            assert torch.allclose(S.calculate_sigmas_scheduler(ms, sch, steps), torch.from_numpy(d[f"{sch}_{steps}"]), rtol=2e-6, atol=1e-7)
    for sch, steps, den in [("normal", 20, 1.0), ("normal", 20, 0.55), ("sgm_uniform", 4, 0.55), ("karras", 20, 0.7)]:
        ks = S.KSampler(steps, "euler", sch, den)
        assert torch.allclose(ks.sigmas, torch.from_numpy(d[f"ks_{sch}_{steps}_{int(den*100)}_sigmas"]), rtol=2e-6, atol=1e-7)
        assert ks.timesteps == d[f"ks_{sch}_{steps}_{int(den*100)}_timesteps"].tolist()


def test_library_exports_every_declared_symbol():
    from stable_renderer_amd import _lib
    L = _lib.lib()                       # raises if the .so is missing or a symbol of SYMBOLS is absent
    hdr = open(os.path.join(ROOT, "include", "sr_hip.h")).read()
    declared = set(re.findall(r"\b(sr_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"sr_op", "sr_draw", "sr_gbuffer"}
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(L, name), name
        assert name in _lib.SYMBOLS, f"{name} declared in the header but not bound"
    assert L.sr_version() >= 100
    assert L.sr_groupnorm_scratch_floats(2, 4096) == 2 * 64 * 64 * 2


def test_no_cpu_fallback_without_library(monkeypatch):
    from stable_renderer_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libsr_hip.so")
    monkeypatch.setenv("SR_NO_REBUILD", "1")
    try:
        _lib.lib()
        assert False, "must fail loudly"
    except _lib.SrHipError as e:
        assert "no CPU fallback" in str(e)


def test_stale_library_is_refused(monkeypatch, tmp_path):
    """a libsr_hip.so that was not built from the sources next to it is never loaded silently (sr_source_hash vs build.source_hash)"""
    from stable_renderer_amd import _lib
    bm = _lib._build_module()
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setenv("SR_NO_REBUILD", "1")
    real = bm.source_hash()
    monkeypatch.setattr(_lib, "_build_module", lambda: type("B", (), {"source_hash": staticmethod(lambda: "deadbeef" + real[8:]),
                                                                      "built_hash": staticmethod(bm.built_hash),
                                                                      "build": staticmethod(lambda: None)}))
    with pytest.raises(_lib.SrHipError, match="stale or missing"):
        _lib.lib()


def test_scene_math():
    from stable_renderer_amd import scene as S
    v = S.look_at((0, 0.68, 2.3), (0, 0.68, 0), (0, 1, 0))
    p = S.perspective(np.radians(45.0), 1.0, 0.1, 100.0)
    # a point 1.6 in front of the camera on the axis lands at NDC (0,0) with w = 1.6
    pt = np.array([0, 0.68, 0.7, 1], np.float32)
    eye = (v.T @ pt)
    clip = (p.T @ eye)
    assert np.allclose(eye[:3], [0, 0, -1.6], atol=1e-6) and abs(clip[3] - 1.6) < 1e-6 and abs(clip[0]) < 1e-6
    m = S.matmul(S.translate((1, 2, 3)), S.scale(2.0))
    assert np.allclose((m.T @ np.array([1, 1, 1, 1], np.float32))[:3], [3, 4, 5])
    it = S.inverse_transpose(m)
    assert np.allclose(it.T @ m.T.T, np.eye(4), atol=1e-6) or np.allclose(np.linalg.inv(m.T).T, it.T, atol=1e-6)
    sp = S.Mesh.Sphere(32)
    assert sp.positions.shape == (1089, 3) and sp.tris.shape == (2 * 32 * 32 - 2, 3)
    assert sp.tris.max() < 1089 and (sp.tris % 33 != 32).all()          # seam column never referenced (mesh.py:555-558)
    t = S.strip_to_triangles([0, 1, 2, 3])
    assert t.tolist() == [[0, 1, 2], [2, 1, 3]]


def test_gbuffer_dump_layout_matches_reference(tmp_path):
    """dumps.GBufferDump writes what DiffusionManager._outputMap/_outputNumpyData/_outputDepthMap write
    (diffusionManager.py:160-259; golden produced by those reference methods, oracle/gen_golden.py gbufdump)."""
    import numpy as np
    from PIL import Image
    from stable_renderer_amd.dumps import GBufferDump
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "gbuffer_dump.npz"))
    d = GBufferDump(str(tmp_path))
    f = int(g["frame"])
    d.output_map("Color", g["color"], frame_num=f)
    d.output_map("normal", g["normal"], frame_num=f)
    d.output_map("canny", g["canny"], frame_num=f)
    d.output_map("gray", g["gray"])
    d.output_numpy("id", g["ids"], f)
    d.output_numpy("pos", g["pos"], f)
    d.output_numpy("noise", g["noise"], f)
    d.output_depth(g["depth"], f)
    for name in ("color", "normal", "canny"):
        got = np.asarray(Image.open(tmp_path / name / f"{name}_{f}.png"))
        assert got.shape == g[name + "_png"].shape and (got == g[name + "_png"]).all(), name
    assert (np.asarray(Image.open(tmp_path / "gray" / "gray.png")) == g["gray_png"]).all()
    assert (np.asarray(Image.open(tmp_path / "depth" / f"depth_{f}.png")) == g["depth_png"]).all()
    for name in ("id", "pos", "noise"):
        got = np.load(tmp_path / name / f"{name}_{f}.npy")
        assert got.dtype == g[name + "_npy"].dtype and (got == g[name + "_npy"]).all(), name


def test_call_order_tickets_serialize_sections_in_call_order():
    """pipeline.CallOrder: calls in flight on several threads take their RNG-draw and corr-map-update sections in call order,
    and a failure inside a section releases the waiters instead of deadlocking them"""
    import threading
    import time
    from stable_renderer_amd.pipeline import CallOrder
    order, log = CallOrder(), []

    def work(i, k):
        for c in range(i, 9, k):
            with order.turn("rng", c):
                log.append(("rng", c))
                time.sleep(0.002 * ((c * 7) % 3))
            time.sleep(0.001 * ((c * 5) % 4))            # "sampling": runs concurrently with the other threads
            with order.turn("bake", c):
                log.append(("bake", c))
    ts = [threading.Thread(target=work, args=(i, 3)) for i in range(3)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=30)
        assert not t.is_alive()
    assert [c for k, c in log if k == "rng"] == list(range(9))
    assert [c for k, c in log if k == "bake"] == list(range(9))
    assert all(log.index(("rng", c)) < log.index(("bake", c)) for c in range(9))

    order2, seen = CallOrder(), []

    def boom():
        try:
            with order2.turn("rng", 0):
                raise RuntimeError("x")
        except RuntimeError:
            seen.append("raised")

    def waiter():
        try:
            with order2.turn("rng", 1):
                seen.append("entered")
        except RuntimeError as e:
            seen.append(str(e))
    tw = threading.Thread(target=waiter)
    tw.start()
    time.sleep(0.05)
    boom()
    tw.join(timeout=10)
    assert not tw.is_alive() and seen == ["raised", "another in-flight call failed"]


def test_sdxl_encode_adm_matches_the_reference():
    """sampling.encode_adm_sdxl vs comfy SDXL.encode_adm itself (golden sdxl_adm.npz, made by oracle/gen_golden.py sdxl)"""
    from stable_renderer_amd.sampling import encode_adm_sdxl
    d = np.load(os.path.join(GOLD, "sdxl_adm.npz"))
    pooled = torch.from_numpy(d["pooled"])
    assert torch.equal(encode_adm_sdxl(pooled, 1024, 1024), torch.from_numpy(d["adm_default"]))
    got = encode_adm_sdxl(pooled, 832, 1216, crop_w=8, crop_h=16, target_width=1024, target_height=1024)
    assert torch.equal(got, torch.from_numpy(d["adm_custom"])) and tuple(got.shape) == (2, 2816)


def test_latent_scale_follows_the_model_family():
    """process_latent_in / out scale: 0.18215 for SD1.x, 0.13025 for the SDXL family (golden: comfy's own latent_formats picked
    through supported_models; the sampler multiplies the latent image by it going in and divides coming out)"""
    from stable_renderer_amd.sampling import latent_scale_of
    from stable_renderer_amd.unet import SD15_CFG, SDXL_CFG
    d = np.load(os.path.join(GOLD, "sdxl_adm.npz"))
    assert latent_scale_of(SD15_CFG) == float(d["sd15_scale"]) and latent_scale_of(SDXL_CFG) == float(d["sdxl_scale"])
    lat = torch.from_numpy(d["lat"])
    assert torch.equal(lat * latent_scale_of(SDXL_CFG), torch.from_numpy(d["sdxl_in"]))
    assert torch.equal(lat * latent_scale_of(SD15_CFG), torch.from_numpy(d["sd15_in"]))
    assert torch.allclose(lat / latent_scale_of(SDXL_CFG), torch.from_numpy(d["sdxl_out"]), rtol=1e-6)


def test_conditioning_list_preparation_matches_the_reference():
    """conditioning.prepare / groups_of (host side of calc_cond_uncond_batch): resolved areas, opposite-area entries and the
    batch composition of every model call equal what the reference's samplers.sample() + calc_cond_uncond_batch did (golden
    cond_compose.npz); the weights tensors equal the oracle's get_area_and_mult restatement"""
    import sr_oracle as ORC
    from stable_renderer_amd import conditioning as CD
    from test_oracle_golden import load_cond_case
    d = np.load(os.path.join(GOLD, "cond_compose.npz"))
    meta = json.loads(bytes(d["meta"]).decode())
    x = torch.from_numpy(d["x"])
    for name, m in meta.items():
        pos, neg = load_cond_case(d, meta, name)
        p, n = CD.prepare(pos, neg, 16, 24)
        assert [None if e.get("area") is None else list(e["area"]) for e in p] == m["areas_pos"], name
        assert [None if e.get("area") is None else list(e["area"]) for e in n] == m["areas_neg"], name
        groups = CD.groups_of(p, n, 2, 4, 16, 24)
        assert [[k for k, _, _ in g["members"]] for g in groups] == [cl[1] for cl in m["calls"]], name
        assert [[2 * len(g["members"]), 4, g["area"][0], g["area"][1]] for g in groups] == [cl[0] for cl in m["calls"]], name
        for e in p + n:
            assert torch.equal(CD.mult_of(e, 2, 4, 16, 24)[0], ORC.area_and_mult(e, x)[1]), name
    assert CD.is_plain(CD.entries_of(torch.zeros(1, 77, 8))) and not CD.is_plain(CD.entries_of([[torch.zeros(1, 77, 8), {"strength": 0.5}]]))
    with pytest.raises(TypeError):
        CD.entries_of([torch.zeros(1, 77, 8), {}])           # the flattened pair SceneTextEncode(merge=False, idmap=None) builds in the reference
