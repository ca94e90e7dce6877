"""BASELINE config[0] as a parity case: single sphere, 256x256, SD1.5-shaped UNet + VAE (real layer shapes, seeded
synthetic weights), 4 steps euler / sgm_uniform, cfg 2, 1 view.  The HIP fp32 path must match the oracle (CPU torch fp32
restatement, itself pinned to the reference) with decoded-frame PSNR >= 40 dB (north_star criterion); the fp16 path (what
the reference runs on ROCm) is reported and must stay above 58 dB (measured 64.6 dB; the floor sits 6 dB below)."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def psnr(a, b):
    mse = float(((a.double() - b.double()) ** 2).mean())
    return 99.0 if mse == 0 else 10.0 * math.log10(1.0 / mse)


@pytest.mark.timeout(1500)
def test_config1_psnr_fp32_and_fp16():
    import sr_oracle as ORC
    from stable_renderer_amd import synth
    from stable_renderer_amd.model_shapes import unet_names_shapes, vae_decoder_names_shapes
    from stable_renderer_amd.pipeline import BakeBallScene, FramePipeline
    from stable_renderer_amd.corresponder import DefaultCorresponder
    from stable_renderer_amd.unet import SD15_CFG, UNet
    from stable_renderer_amd.vae import VAEDecoder
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ns, norms = unet_names_shapes(SD15_CFG)
    sd_u = synth.synth_state_dict(ns, seed=0, norm_names=norms)
    vns, vnorms = vae_decoder_names_shapes()
    sd_v = synth.synth_state_dict(vns, seed=2, norm_names=vnorms)
    g = torch.Generator().manual_seed(1)
    pos, neg = torch.randn(1, 77, 768, generator=g), torch.randn(1, 77, 768, generator=torch.Generator().manual_seed(2))
    results = {}
    noise_ref = None
    for dtype in (torch.float32, torch.float16):
        unet = UNet(sd_u, SD15_CFG, dtype=dtype)
        vae = VAEDecoder(sd_v, dtype=dtype)
        scene = BakeBallScene(256, 256, k=3)
        pipe = FramePipeline(unet, vae, scene, n_views=1, steps=4, cfg=2.0, sampler="euler", scheduler="sgm_uniform",
                             corresponder=DefaultCorresponder(), use_graph=False)
        pipe.set_prompt(pos, neg)
        torch.manual_seed(7)
        ed = pipe.render_views()
        noise = ed.noise_maps["noise"].cpu()
        if noise_ref is None:
            noise_ref = noise
        assert torch.equal(noise, noise_ref)                      # raster + noise pooling are dtype independent
        samples = pipe.diffuse(ed)
        img = pipe.decode(samples).cpu()
        results[dtype] = (samples.cpu(), img)
        del pipe, unet, vae
        torch.cuda.empty_cache()
    torch.manual_seed(7)
    with torch.no_grad():
        o_s, _ = ORC.sample_frames(sd_u, SD15_CFG, noise_ref, pos, neg, None, 4, 2.0, "euler", "sgm_uniform")
        o_img = ORC.vae_decode_image(sd_v, o_s)
    p32, p16 = psnr(results[torch.float32][1], o_img), psnr(results[torch.float16][1], o_img)
    print(f"config1 PSNR vs oracle: fp32 {p32:.1f} dB, fp16 {p16:.1f} dB")
    assert p32 >= 40.0, p32
    assert p16 >= 58.0, p16
    rel = (results[torch.float32][0] - o_s).abs().max().item() / o_s.abs().max().item()
    assert rel < 5e-3, rel


@pytest.mark.timeout(900)
def test_config1_from_the_references_own_dumps_through_the_loader_nodes(tmp_path):
    """BASELINE config 1 as the reference states it -- "pre-dumped G-buffers, 256x256, 4 steps, 1 view" -- through this package's node
    chain IDSequenceLoader -> NoiseSequenceLoader -> VirtualEngineData -> DefaultCorresponder -> CorrespondSampler -> VAEDecode on a
    256x256 dump directory made of every second pixel of one frame of the reference's shipped dumps (tests/golden/
    full_config1_dumps.npz carries that subsample, the reference loaders' outputs and the reference stack's samples / frames:
    oracle/gen_golden_full.py config1_dumps).  NoiseSequenceLoader pools with reshape_magnitude = 256 // 64 = 4 (means of 16
    consecutive pixels, viewed 64 x 64: _nodes/loaders.py:131-146) -- the general form sr_noise_pool_strips runs."""
    import json
    from stable_renderer_amd import nodes as N
    from stable_renderer_amd import synth
    from stable_renderer_amd.model_shapes import unet_names_shapes, vae_decoder_names_shapes
    from stable_renderer_amd.unet import SD15_CFG, UNet
    from stable_renderer_amd.vae import VAEDecoder
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "full_config1_dumps.npz"))
    m = json.loads(bytes(g["meta"]).decode())
    os.makedirs(tmp_path / "id"); os.makedirs(tmp_path / "noise")
    np.save(tmp_path / "id" / "id_0.npy", g["id_dump"])
    np.save(tmp_path / "noise" / "noise_0.npy", g["noise_dump"])
    ns, norms = unet_names_shapes(SD15_CFG)
    sd_u = synth.synth_state_dict(ns, seed=m["unet_seed"], norm_names=norms)
    vns, vnorms = vae_decoder_names_shapes()
    sd_v = synth.synth_state_dict(vns, seed=m["vae_seed"], norm_names=vnorms)
    ctx = lambda seed: torch.randn(1, 77, 768, generator=torch.Generator().manual_seed(seed))
    ref_s, ref_img = torch.from_numpy(g["samples"]), torch.from_numpy(g["img_sub"]).float()
    res = {}
    for dtype in (torch.float32, torch.float16):
        idmap = N.IDSequenceLoader()(str(tmp_path / "id"), 0, 1)
        assert np.array_equal(idmap.tensor.cpu().numpy().astype(np.int64), g["loader_ids"].astype(np.int64))      # bit exact
        lat = N.NoiseSequenceLoader()(str(tmp_path / "noise"), 0, 1, "SD15")
        assert tuple(lat["noise"].shape) == (1, 4, 64, 64)
        # the reference rounds the 16-pixel means to the dump's fp16 before AdaIN, sr_noise_pool_strips keeps fp32
        assert torch.allclose(lat["noise"].float().cpu(), torch.from_numpy(g["loader_noise"]), atol=4e-3, rtol=2e-3)
        lat["noise"] = torch.from_numpy(g["loader_noise"]).to(lat["noise"].device)          # identical sampler input from here on
        lat["samples"] = torch.zeros_like(lat["noise"])
        ed = N.VirtualEngineDataNode()(id_maps=idmap, noise_maps=lat)
        corr, vae_cb = N.DefaultCorresponder()(ed, update_corrmap=False)
        model = N.MODEL(UNet(sd_u, SD15_CFG, dtype=dtype))
        torch.manual_seed(m["rng_seed"])
        out = N.CorrespondSampler()(model, ctx(m["pos_seed"]), ctx(m["neg_seed"]), corr, ed, latent=None, steps=m["steps"], cfg=m["cfg"],
                                    sampler_name=m["sampler"], scheduler=m["scheduler"])
        img = N.VAEDecode().decode(VAEDecoder(sd_v, dtype=dtype), out)
        img = img[0] if isinstance(img, tuple) else img
        res[dtype] = (out["samples"].float().cpu(), img.float().cpu())
        del model
        torch.cuda.empty_cache()
    p32, p16 = psnr(res[torch.float32][1][:, ::4, ::4], ref_img), psnr(res[torch.float16][1][:, ::4, ::4], ref_img)
    rel32 = (res[torch.float32][0] - ref_s).abs().max().item() / ref_s.abs().max().item()
    print(f"config 1 from the reference's dumps: PSNR vs the reference fp32 {p32:.1f} dB / fp16 {p16:.1f} dB, latent rel err fp32 {rel32:.2e}")
    assert p32 >= 40.0 and rel32 < 1e-3, (p32, rel32)
    assert p16 >= 58.0, p16                                    # measured 64.8 dB
