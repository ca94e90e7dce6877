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
