"""CPU tests of the utility nodes (stable_renderer_amd/extra_nodes.py): no kernel is involved in these."""
import os

import numpy as np
import pytest
import torch

from stable_renderer_amd import extra_nodes as X
from stable_renderer_amd.corrmap import IDMap


def test_identical_noise_sequence_draw_order():
    """both 'generators' of the reference are the global one: seeded seed, reseeded seed+1, latent drawn first, noise second"""
    out = X.CreateIdenticalNoiseSequence()(seed=7, num_frames=3, sd_version="SD15", device="cpu")
    g = torch.manual_seed(8)
    lat = torch.randn([1, 4, 64, 64], generator=g)
    noi = torch.randn([1, 4, 64, 64], generator=g)
    assert torch.equal(out["samples"], lat.repeat(3, 1, 1, 1)) and torch.equal(out["noise"], noi.repeat(3, 1, 1, 1))
    assert X.CreateIdenticalNoiseSequence()(1, 2, "SDXL", device="cpu")["noise"].shape == (2, 4, 128, 128)
    with pytest.raises(ValueError):
        X.CreateIdenticalNoiseSequence()(1, 0)
    with pytest.raises(ValueError):
        X.CreateIdenticalNoiseSequence()(1, 2, "SD3")


def test_processing_nodes(tmp_path, monkeypatch):
    img = torch.rand(2, 8, 8, 4)
    rgb = X.RGBAToRGB()(img, "ff8000")
    want = (1 - img[..., 3:]) * torch.tensor([255, 128, 0]) + img[..., 3:] * img[..., :3]
    assert torch.allclose(rgb, want)
    with pytest.raises(ValueError):
        X.RGBAToRGB()(img, "gggggg")
    th = X.RGBAThreshold()(img, 0.5)
    assert th.shape == img.shape and set(th[..., 3].unique().tolist()) <= {0.0, 1.0} and torch.equal(th[..., :3], img[..., :3])
    assert X.TextConcat()("a ", "b") == "a b" and X.TextReplace()("a cat", "cat", "dog") == "a dog"
    with pytest.raises(RuntimeError, match="anime-seg"):
        X.RemoveBGNode()(img[0, ..., :3])
    X.RemoveBGNode.set_segmenter(lambda im: (im.mean(-1) > 0.5).float())
    try:
        out = X.RemoveBGNode()(img[..., :3])
        assert out.shape == (2, 8, 8, 4) and set(out[..., 3].unique().tolist()) <= {0.0, 1.0}
        assert bool((out[..., :3][out[..., 3] == 0] == 1.0).all())       # background turns white
    finally:
        X.RemoveBGNode.set_segmenter(None)
    monkeypatch.setenv("SR_OUTPUT_DIR", str(tmp_path))
    path = X.SimpleVideoCombine()(list(torch.rand(3, 16, 16, 4)), frame_rate=4, filename_prefix="t_", pingpong=True)
    from PIL import Image
    g = Image.open(path)
    assert g.is_animated and g.n_frames == 4 and os.path.dirname(path) == str(tmp_path)
