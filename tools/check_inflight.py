"""diagnostic: corr-map after 5 calls, sequential loop vs calls in flight, several repetitions each"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SR_AUTOTUNE", "0")
from stable_renderer_amd.pipeline import build_sd15_pipeline, InflightCalls  # noqa: E402
from stable_renderer_amd.unet import SD15_CFG  # noqa: E402

cfg = dict(SD15_CFG, model_channels=64, context_dim=64)
kw = dict(dtype=torch.float32, n_views=4, steps=3, cfg=5.0, W=128, H=128, unet_cfg=cfg, vae_ch=32)


def bake(inflight, graph=True):
    pipe = build_sd15_pipeline(use_graph=graph, **kw)
    torch.manual_seed(77)
    if inflight == 1:
        for _ in range(5):
            pipe.call()
    else:
        InflightCalls(pipe, inflight).run(5)
    torch.cuda.synchronize()
    cm = pipe.scene.corrmap
    return cm._values.float().clone(), cm._writtens.clone()


runs = [("seq", 1, True), ("seq", 1, True), ("fl2", 2, True), ("fl2", 2, True), ("fl2-nograph", 2, False), ("seq-nograph", 1, False)]
res = [(n, bake(k, g)) for n, k, g in runs]
v0, w0 = res[0][1]
for n, (v, w) in res:
    d = (v - v0).abs()
    print(n, "writtens equal", bool(torch.equal(w, w0)), "values: max diff %.3e, differing %d of %d" % (float(d.max()), int((d > 0).sum()), int((w0 > 0).sum()) * 4))
