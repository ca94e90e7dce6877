"""The drop-in surface: node classes with the reference's signatures and the Engine script API run the HIP path."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bake_ball_script_api_and_nodes():
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import bake_ball as B
    from stable_renderer_amd.engine import EngineMode
    from stable_renderer_amd import nodes as N
    B.Sample.stop_at = 17
    pipe = B.make_bake_pipeline(dtype=torch.float32, steps=2, cfg=2.0, tiny=True)
    e = B.Sample.Run(winSize=(128, 128), mode=EngineMode.BAKE, baking_interval=8, pipeline=pipe)
    torch.cuda.synchronize()
    # bake cadence of the reference: first call carries frames 0..8 (9 frames), then 8 (diffusionManager.py:96-102)
    assert e.RuntimeManager.FrameCount == 17 and len(e.outputs) == 2
    assert tuple(e.outputs[0].shape) == (9, 128, 128, 3) and tuple(e.outputs[1].shape) == (8, 128, 128, 3)
    assert torch.isfinite(e.outputs[0]).all() and float(e.outputs[0].min()) >= 0 and float(e.outputs[0].max()) <= 1
    assert int(e.corrmap.writtens.sum()) > 1000
    # OverlapCorresponder is rejected with a non ddim/ddpm sampler, as in the reference (_nodes/samplers.py:163-164)
    corr, cb = N.OverlapCorresponder()(None)
    with pytest.raises(ValueError):
        N.CorrespondSampler()(None, None, None, corr, None, latent={"samples": torch.zeros(1, 4, 8, 8)}, sampler_name="euler")
    assert cb() is None


def test_raster_only_engine_matches_oracle_ids():
    """disableComfyUI=True flavour: the engine only rasterises; ids of frame 8 match the C oracle bit for bit."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import bake_ball as B
    import raster_ref as R
    from stable_renderer_amd import scene as S
    from stable_renderer_amd.engine import EngineMode, GameObject, Camera, MeshRenderer, CorrMapRenderer
    B.Sample.stop_at = 9
    e = B.Sample.Run(winSize=(160, 160), mode=EngineMode.BAKE, baking_interval=8, pipeline=None)
    ed = e.outputs[0]
    assert ed.id_maps.tensor.shape[0] == 9
    cam = next(c for o in GameObject._all for c in o.components if isinstance(c, Camera)).to_scene_camera()
    view, proj = cam.view(), cam.projection(1.0)
    tasks = []
    for o in GameObject._all:
        for c in o.components:
            if isinstance(c, CorrMapRenderer):
                tasks += c.tasks(view, EngineMode.BAKE)
            elif isinstance(c, MeshRenderer):
                tasks += c.tasks(view)
    ref = R.GBufferRef(160, 160)
    ref.clear()
    for t in sorted(tasks, key=lambda t: t.order):            # scene state = after the last frame's update (frame 8)
        ref.draw(t, S.draw_params(t, view, proj),
                 noise_tex=None if t.noise_tex is None else t.noise_tex.cpu().numpy().view(np.uint16),
                 diffuse_tex=None if t.diffuse_tex is None else t.diffuse_tex.cpu().float().numpy())
    assert np.array_equal(ed.id_maps.tensor[8].cpu().numpy(), ref.id)


def test_sequence_loaders_read_a_dump(tmp_path):
    """IDSequenceLoader / NoiseSequenceLoader / ImageSequenceLoader (_nodes/loaders.py) on a dump directory written by
    dumps.GBufferDump; the noise latent is compared with the reference loader's output (golden, fp16-resolution tolerance:
    the reference rounds the strip mean to fp16 before AdaIN, sr_noise_pool keeps fp32)."""
    import numpy as np
    from stable_renderer_amd.dumps import GBufferDump
    from stable_renderer_amd import nodes as N
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "gbuffer_dump.npz"))
    d = GBufferDump(str(tmp_path))
    for i, seed in enumerate(g["loader_seeds"].tolist()):
        d.output_numpy("noise", np.random.default_rng(seed).standard_normal((512, 512, 4)).astype(np.float16), i)
        ids = np.zeros((512, 512, 4), np.int16)
        ids[..., 3] = i + 1
        d.output_numpy("id", ids, i)
        d.output_map("color", np.full((512, 512, 3), 0.25 * (i + 1), np.float32), frame_num=i)
    lat = N.NoiseSequenceLoader()(str(tmp_path / "noise"), 0, 2, "SD15")
    ref = torch.from_numpy(g["loader_noise"])
    got = lat["noise"].float().cpu()
    assert got.shape == ref.shape == (2, 4, 64, 64)
    assert float((got - ref).abs().max()) < 4e-3 * float(ref.abs().max()), float((got - ref).abs().max())
    assert float(lat["samples"].abs().max()) == 0.0
    idm = N.IDSequenceLoader()(str(tmp_path / "id"), 0, 2)
    assert idm.tensor.shape == (2, 512, 512, 4) and idm.frame_indices == [0, 1]
    assert int(idm.tensor[1, 0, 0, 3]) == 2
    img = N.ImageSequenceLoader()(str(tmp_path / "color"), 0, 2, "SD15")
    assert img.shape == (2, 512, 512, 3)
    assert abs(float(img[1, 5, 5, 0]) - int(0.5 * 255) / 255.0) < 1e-6
    with pytest.raises(FileNotFoundError):
        N.NoiseSequenceLoader()(str(tmp_path / "nope"))


def test_engine_runs_a_shipped_workflow_graph(monkeypatch):
    """Engine.Run(diffuse_workflow=<shipped JSON>) — the reference's script API (engine.py:97, diffusionManager.py:36-77):
    every submitted EngineData goes through workflow.PromptExecutor; sprites / camera bgPrompt reach SceneTextEncode; the
    DefaultCorresponder VAE callback bakes the corr-map; loaders run once for the whole session."""
    from stable_renderer_amd import synth, weights as WT
    from stable_renderer_amd.corrmap import CorrespondMap
    from stable_renderer_amd.engine import (Camera, CorrMapRenderer, DefaultTextureType, Engine, EngineMode, GameObject, Material,
                                            SpriteInfo, Texture)
    from stable_renderer_amd.graph_nodes import SyntheticCLIP
    from stable_renderer_amd.model_shapes import unet_names_shapes, vae_decoder_names_shapes
    from stable_renderer_amd.scene import Mesh
    from stable_renderer_amd.unet import SD15_CFG
    monkeypatch.setenv("SR_DTYPE", "fp32")
    monkeypatch.setenv("SR_AUTOTUNE", "0")
    cfg = dict(SD15_CFG, model_channels=64, context_dim=64)
    ns, norms = unet_names_shapes(cfg)
    vns, vnorms = vae_decoder_names_shapes(ch=32)
    WT.clear_registry()
    WT.register_checkpoint("dreamshaper_8.safetensors", lambda: dict(
        unet=synth.synth_state_dict(ns, seed=1, norm_names=norms), vae=synth.synth_state_dict(vns, seed=3, norm_names=vnorms),
        clip=SyntheticCLIP(ctx_dim=64), unet_cfg=cfg))
    WT.register_lora("lcm/SD1.5/pytorch_lora_weights.safetensors", lambda: {})          # an empty LoRA: nothing to merge
    seen = []

    class Sample(Engine):
        def beforePrepare(self):
            cam = GameObject('Camera', position=[0, 1.0, 0.6])
            cam.addComponent(Camera, bgPrompt='misty forest')
            cam.transform.lookAt([0, 0, 0])
            w, h = self.WindowManager.WindowSize
            self.corrmap = CorrespondMap(k=3, width=w, height=h)
            mat = Material.DefaultTransparentMaterial()
            mat.addDefaultTexture(Texture.CreateNoiseTex(w, h, seed=3), DefaultTextureType.NoiseTex)
            floor = GameObject('floor', position=[0, 0, 0], scale=20.0)              # fills the frame: every pixel carries an id
            floor.addComponent(SpriteInfo, auto_spriteID=True, prompt='mossy stone floor')
            floor.addComponent(CorrMapRenderer, corrmaps=self.corrmap, materials=[mat], use_texcoord_id=True, mesh=Mesh.Plane(4))

        def beforeFrameEnd(self):
            seen.append(len(self.outputs))

    e = Sample.Run(winSize=(128, 128), mode=EngineMode.BAKE, baking_interval=2, max_frames=5,
                   diffuse_workflow=os.path.join(ROOT, "tests", "golden", "workflows", "no-control-bake.json"))
    torch.cuda.synchronize()
    assert len(e.outputs) == 2 and seen == [0, 0, 1, 1, 2]                            # submits after frames 2 and 4
    assert tuple(e.outputs[0].shape) == (3, 128, 128, 3) and tuple(e.outputs[1].shape) == (2, 128, 128, 3)
    assert all(bool(torch.isfinite(o).all()) and float(o.min()) >= 0 and float(o.max()) <= 1 for o in e.outputs)
    assert int(e.corrmap.writtens.sum()) > 0
    ex = e.DiffusionManager.Executor
    ctx = ex.latest_context
    assert ctx.success and not ({"4", "11"} & ctx.executed_node_ids)                  # checkpoint + LoRA cached from the first call
    assert ctx.engine_data.env_prompts[0].prompt == 'misty forest'
    assert [s.prompt for s in ctx.engine_data.sprite_infos.values()] == ['mossy stone floor']
    WT.clear_registry()
    with pytest.raises(ValueError, match="Prompt execution failed"):                 # renderManager.py:1014-1015
        Sample.Run(winSize=(128, 128), mode=EngineMode.BAKE, baking_interval=2, max_frames=3,
                   diffuse_workflow=os.path.join(ROOT, "tests", "golden", "workflows", "no-control-bake.json"))
