"""BASELINE config[0] as a parity case: single sphere, 256x256, SD1.5-shaped UNet + VAE (real layer shapes, seeded
synthetic weights), 4 steps euler / sgm_uniform, cfg 2, 1 view.  The HIP fp32 path must match the oracle (CPU torch fp32
restatement, itself pinned to the reference) with decoded-frame PSNR >= 40 dB (north_star criterion); the fp16 path (what
the reference runs on ROCm) is reported and must stay above 30 dB."""
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
    assert p16 >= 30.0, p16
    rel = (results[torch.float32][0] - o_s).abs().max().item() / o_s.abs().max().item()
    assert rel < 5e-3, rel
