"""TEST INFRASTRUCTURE (outside the product package: it imports the oracle).  One small invocation of the whole hot path on cuda:0, checked against the oracle (driver's smoke()).

2 views of the bake_ball scene at 128x128 -> raster (bit-exact ids vs oracle/raster_ref.c) -> pooled noise ->
tiny SD-topology UNet (fp32 MFMA) 2 ddim steps with latent overlap + K/V injection -> tiny VAE decoder -> corr-map
update; latents and decoded frames are compared with oracle/sr_oracle.py (torch CPU fp32)."""
import numpy as np
import torch


def run():
    import raster_ref as R                      # oracle (checker only)
    import sr_oracle as ORC
    from stable_renderer_amd import scene as S
    from stable_renderer_amd import synth
    from stable_renderer_amd.model_shapes import unet_names_shapes, vae_decoder_names_shapes
    from stable_renderer_amd.pipeline import BakeBallScene, FramePipeline
    from stable_renderer_amd.unet import SD15_CFG, UNet
    from stable_renderer_amd.vae import VAEDecoder
    dev = "cuda:0"
    torch.cuda.set_device(0)
    W = H = 128
    cfg = dict(SD15_CFG, model_channels=64, context_dim=64)
    ns, norms = unet_names_shapes(cfg)
    sd_u = synth.synth_state_dict(ns, seed=1, norm_names=norms)
    vns, vnorms = vae_decoder_names_shapes(ch=32)
    sd_v = synth.synth_state_dict(vns, seed=2, norm_names=vnorms)
    unet = UNet(sd_u, cfg, dtype=torch.float32, device=dev)
    vae = VAEDecoder(sd_v, dtype=torch.float32, device=dev)
    scene = BakeBallScene(W, H, k=3, device=dev)
    pipe = FramePipeline(unet, vae, scene, n_views=2, steps=2, cfg=3.0, sampler="ddim", scheduler="normal", use_graph=False)
    g = torch.Generator().manual_seed(5)
    pos, neg = torch.randn(1, 77, 64, generator=g), torch.randn(1, 77, 64, generator=g)
    pipe.set_prompt(pos, neg)
    torch.manual_seed(123)
    ed = pipe.render_views()
    # --- raster parity (ids bit exact)
    ref = R.GBufferRef(W, H)
    view, proj = scene.camera.view(), scene.camera.projection(1.0)
    for f in range(2):
        ref.clear()
        for t in sorted(scene.tasks(f), key=lambda t: t.order):
            ref.draw(t, S.draw_params(t, view, proj),
                     noise_tex=None if t.noise_tex is None else t.noise_tex.cpu().numpy().view(np.uint16),
                     diffuse_tex=None if t.diffuse_tex is None else t.diffuse_tex.cpu().numpy())
        assert np.array_equal(ed.id_maps.tensor[f].cpu().numpy(), ref.id), "raster ids differ from the oracle"
    noise = ed.noise_maps["noise"].cpu()
    samples = pipe.diffuse(ed)
    images = pipe.decode(samples)
    pipe.baker.finished(ed, images)
    torch.cuda.synchronize()
    # --- oracle for the diffusion part, same RNG stream
    torch.manual_seed(123)
    with torch.no_grad():
        o_s, _ = ORC.sample_frames(sd_u, cfg, noise, pos, neg, ed.id_maps.tensor.cpu().numpy(), 2, 3.0, "ddim", "normal",
                                   overlap=dict(ratio=0.5, stop=500, n_rand=1))
        o_img = ORC.vae_decode_image(sd_v, o_s)
    e1 = (samples.cpu() - o_s).abs().max().item() / max(1.0, o_s.abs().max().item())
    e2 = (images.cpu() - o_img).abs().max().item()
    assert e1 < 5e-3, f"latents differ from the oracle: {e1}"
    assert e2 < 2e-2, f"decoded frames differ from the oracle: {e2}"
    assert int(scene.corrmap.writtens.sum()) > 0
    print(f"smoke ok: latent rel err {e1:.2e}, image abs err {e2:.2e}, corr-map texels written {int(scene.corrmap.writtens.sum())}")
