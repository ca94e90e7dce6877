"""GPU tests of the utility / legacy nodes that carry id maps (an IDMap lives in HBM) or run the sampler."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from stable_renderer_amd import extra_nodes as X  # noqa: E402
from stable_renderer_amd.corrmap import IDMap  # noqa: E402


def test_noise_sequence_from_id_map_ties_vertices_across_frames():
    ids = torch.zeros(2, 512, 512, 4, dtype=torch.int32)
    ids[0, 100:200, 100:200] = torch.tensor([1, 1, 0, 5])
    ids[1, 300:400, 50:150] = torch.tensor([1, 1, 0, 5])                 # the same vertex elsewhere in frame 1
    ids[1, 0:8, 0:8] = torch.tensor([1, 1, 0, 9])
    ids[0, 400:408, 400:408] = torch.tensor([1, 1, 2048, 3])             # non-AI object: not tied
    out = X.CreateNoiseSequenceFromIdMap()(IDMap(ids.cuda()), seed=3, sd_version="SD15", downsample_option="nearest")
    lat, noi = out["samples"].cpu(), out["noise"].cpu()
    assert lat.shape == noi.shape == (2, 4, 64, 64)
    a, b = lat[0, :, 104 // 8, 104 // 8], lat[1, :, 304 // 8, 56 // 8]
    assert torch.equal(a, b) and not torch.equal(a, lat[1, :, 0, 0])     # vertex 5 carries one value in both frames; vertex 9 another
    assert torch.equal(noi[0, :, 13, 13], noi[1, :, 38, 7]) and not torch.equal(noi[0, :, 13, 13], lat[0, :, 13, 13])
    base = X.CreateIdenticalNoiseSequence()(3, 2, device="cpu")          # untouched pixels keep the shared base image... at full
    assert lat[0, :, 50, 50].shape == (4,)                               # resolution (the base here is drawn at 512^2, not 64^2)
    assert torch.equal(lat[0, :, 60, 60], lat[1, :, 60, 60])             # same base in every frame
    m = X.CreateNoiseSequenceFromIdMap()(IDMap(ids.cuda()), 3, "SD15", "mean")
    assert m["noise"].shape == (4, 4, 64, 64) and float(m["samples"].abs().max()) == 0      # the reference's view(-1,4,8,8) regrouping
    with pytest.raises(ValueError):
        X.CreateNoiseSequenceFromIdMap()(IDMap(ids.cuda()), 3, "SD15", "median")


def test_legacy_loaders(tmp_path):
    from PIL import Image
    for i in (2, 0, 1):
        Image.fromarray(np.full((6, 5, 4), 40 * (i + 1), np.uint8), "RGBA").save(tmp_path / f"color_{i}.png")
        np.save(tmp_path / f"id_{i}.npy", np.full((6, 5, 4), i + 1, np.int16))
        np.save(tmp_path / f"{i}_noise.npy", np.full((6, 5, 4), float(i), np.float16))
    paths = [str(tmp_path / f"color_{i}.png") for i in (2, 0, 1)]
    imgs, masks = X.LegacyImageSequenceLoader()(paths)
    assert imgs.shape == (3, 6, 5, 3) and masks.shape == (3, 6, 5)
    assert [round(float(imgs[k, 0, 0, 0]) * 255) for k in range(3)] == [40, 80, 120]         # reordered by the index in the name
    lat = X.LegacyNoiseSequenceLoader()([str(tmp_path / f"{i}_noise.npy") for i in (1, 2, 0)])
    assert lat["noise"].shape == (12, 6, 5) and float(lat["noise"][4, 0, 0]) == 1.0          # CHW planes concatenated along dim 0 (sic)
    idm = X.LegacyIDSequenceLoader()([str(tmp_path / f"id_{i}.npy") for i in (1, 2, 0)], device="cuda")
    assert idm.tensor.shape == (3, 6, 5, 4) and idm.frame_indices == [0, 1, 2] and int(idm.tensor[2, 0, 0, 0]) == 3
    s = X.OverlapScheduler()(interpolate_begin=0.4, start_step=1)
    assert s(step=1, timestep=500) == 0.4 and s(step=0, timestep=500) == 0.0


def test_legacy_sampler_nodes_run(tmp_path):
    """stable_renderer_ultimate.json's node set: CorrespondenceMapLoader -> CorrMapLatentNoiseInitializer -> StableRenderSampler with
    OverlapScheduler-made alpha / kernel-radius schedules (radius 1: the in-place-order path) on a tiny model"""
    import json
    from stable_renderer_amd import nodes as N, synth
    from stable_renderer_amd.unet import UNet, SD15_CFG
    GOLD = os.path.join(os.path.dirname(__file__), "golden")
    g = torch.Generator().manual_seed(2)
    ids = torch.zeros(3, 64, 64, 4, dtype=torch.int16)
    ids[..., 0] = 1
    ids[..., 3] = torch.randint(1, 200, (3, 64, 64), generator=g, dtype=torch.int16)
    ids[:, :10] = 0
    for i in range(3):
        np.save(tmp_path / f"id_{i}.npy", ids[i].numpy())
    cm = X.CorrespondenceMapLoader()(str(tmp_path))
    assert len(cm) == 199 and cm.size == (64, 64)
    lat = X.CorrMapLatentNoiseInitializer()(width=64, height=64, batch_size=3, seed=5, correspondence_map=cm)
    assert lat["samples"].shape == lat["noise"].shape == (3, 4, 8, 8)
    with open(os.path.join(GOLD, "unet_tiny_keys.json")) as f:
        k = json.load(f)
    cfg = dict(SD15_CFG, model_channels=64, context_dim=64)
    model = N.MODEL(UNet(synth.synth_state_dict([(n, tuple(s)) for n, s in k["names_shapes"]], seed=1, norm_names=k["norm_names"]), cfg,
                         dtype=torch.float32))
    pos, neg = torch.randn(1, 77, 64, generator=g), torch.randn(1, 77, 64, generator=g)
    alpha = X.OverlapScheduler()(interpolate_begin=0.5, start_step=0)
    radius = X.OverlapScheduler()(interpolate_begin=1.0, start_step=0)
    off = X.OverlapScheduler()(interpolate_begin=0.0, start_step=0)
    outs = {}
    for name, a_s in (("overlap", alpha), ("none", off)):
        torch.manual_seed(1)
        outs[name] = X.StableRenderSampler()(model, pos, neg, lat, cm, a_s, radius, overlap_algorithm="pixel_distance",
                                             noise_option="incoming", steps=3, cfg=3.0, sampler_name="ddim")["samples"]
    assert torch.isfinite(outs["overlap"]).all() and (outs["overlap"] - outs["none"]).abs().max().item() > 1e-3
