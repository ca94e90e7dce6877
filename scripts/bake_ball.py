"""Port of the reference's scripts/bake_ball.py onto the HIP path: same scene construction through the same script API
(Engine subclass, beforePrepare, GameObject / addComponent, Sample.Run(...)).  With ``--diffuse`` a random-init SD1.5-shaped
UNet + VAE bake every ``baking_interval`` frames through the node surface (CorrespondSampler -> VAEDecode ->
DefaultCorresponder.finished); without it the script rasterises G-buffers only (``disableComfyUI=True`` in the reference)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))

import torch  # noqa: E402

from stable_renderer_amd.engine import (Camera, CorrMapRenderer, DefaultTextureType, Engine, EngineMode,  # noqa: E402
                                        EqualIntervalRotation, GameObject, Material, MeshRenderer, SpriteInfo, Texture)
from stable_renderer_amd.corrmap import CorrespondMap  # noqa: E402
from stable_renderer_amd.scene import Mesh  # noqa: E402


def make_bake_pipeline(dtype=torch.float16, steps=20, cfg=8.0, tiny=False):
    """node graph of resources/example-workflows/no-mask-prompt-bake.json, minus the loaders (random-init weights)"""
    from stable_renderer_amd import nodes as N, synth
    from stable_renderer_amd.model_shapes import unet_names_shapes, vae_decoder_names_shapes
    from stable_renderer_amd.unet import SD15_CFG, UNet
    from stable_renderer_amd.vae import VAEDecoder
    cfgu = dict(SD15_CFG, model_channels=64, context_dim=64) if tiny else dict(SD15_CFG)
    ns, norms = unet_names_shapes(cfgu)
    model = N.MODEL(UNet(synth.synth_state_dict(ns, seed=0, norm_names=norms), cfgu, dtype=dtype))
    vns, vnorms = vae_decoder_names_shapes(ch=32 if tiny else 128)
    vae = VAEDecoder(synth.synth_state_dict(vns, seed=2, norm_names=vnorms), dtype=dtype)
    g = torch.Generator().manual_seed(7)
    pos, neg = torch.randn(1, 77, cfgu["context_dim"], generator=g), torch.randn(1, 77, cfgu["context_dim"], generator=g)
    sampler, decode = N.CorrespondSampler(), N.VAEDecode()

    def pipeline(engine_data):
        corr, vae_cb = N.DefaultCorresponder()(engine_data, update_corrmap=True, update_mode="first")
        corr.ignore_obj_mat_id_when_update = True
        latent = sampler(model, pos, neg, corr, engine_data, steps=steps, cfg=cfg, sampler_name="euler", scheduler="normal")
        (images,) = decode.decode(vae, latent, callback=vae_cb)
        return N.InferenceOutputNode()(images).frame_color
    return pipeline


class Sample(Engine):
    def beforePrepare(self):
        sphere_mesh = Mesh.Sphere()
        mat = Material.DefaultOpaqueMaterial()
        tex = torch.rand(64, 64, 4, generator=torch.Generator().manual_seed(1))
        tex[..., 3] = 1.0
        mat.addDefaultTexture(tex, DefaultTextureType.DiffuseTex)

        camera = GameObject('Camera', position=[0, 0.68, 2.3])
        camera.addComponent(Camera, bgPrompt='no background')
        camera.transform.lookAt([0, 0.68, 0])

        ball = GameObject('ball', position=[0, 0.68, 0], scale=0.70)
        meshRenderer = ball.addComponent(MeshRenderer, mesh=sphere_mesh)
        meshRenderer.addMaterial(mat)
        rotation_interval = 360
        ball.addComponent(EqualIntervalRotation, interval=rotation_interval)

        win_width, win_height = self.WindowManager.WindowSize
        self.corrmap = CorrespondMap(k=6, width=win_width, height=win_height)
        mat = Material.DefaultTransparentMaterial()
        mat.addDefaultTexture(Texture.CreateNoiseTex(win_width, win_height, seed=3), DefaultTextureType.NoiseTex)
        corrmap_obj = GameObject('corrmap', position=[0, 0.68, 0], scale=0.85)
        corrmap_obj.addComponent(SpriteInfo, auto_spriteID=True, prompt='')
        corrmap_obj.addComponent(CorrMapRenderer, corrmaps=self.corrmap, materials=[mat, ], use_texcoord_id=True)
        corrmap_obj.addComponent(EqualIntervalRotation, interval=rotation_interval)

    def beforeFrameBegin(self):
        if self.RuntimeManager.FrameCount == self.stop_at:
            self.Exit()


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=17)
    ap.add_argument("--diffuse", action="store_true")
    ap.add_argument("--tiny", action="store_true")
    a = ap.parse_args()
    Sample.stop_at = a.frames
    pipe = make_bake_pipeline(tiny=a.tiny) if a.diffuse else None
    e = Sample.Run(winSize=(512, 512), mode=EngineMode.BAKE, mapSavingInterval=1, baking_interval=8, needOutputMaps=True,
                   disableComfyUI=not a.diffuse, pipeline=pipe)
    torch.cuda.synchronize()
    print("frames rendered:", e.RuntimeManager.FrameCount, "bake calls:", len(e.outputs),
          "corr-map texels written:", int(e.corrmap.writtens.sum()))
