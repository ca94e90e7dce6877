"""SDXL-family data point (BASELINE config 5: 1024x1024, 8 views): one bake call through FramePipeline with the SDXL base UNet
(2.57 B parameters, random init), the SD VAE decoder at 1024^2 and the same raster / overlap / corr-map path.
usage: python tools/bench_sdxl.py [steps] [views]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stable_renderer_amd.pipeline import build_sd15_pipeline  # noqa: E402
from stable_renderer_amd.unet import SDXL_CFG  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
views = int(sys.argv[2]) if len(sys.argv) > 2 else 8
t0 = time.perf_counter()
pipe = build_sd15_pipeline(dtype=torch.float16, n_views=views, steps=steps, cfg=8.0, W=1024, H=1024, unet_cfg=dict(SDXL_CFG))
pipe.runner.set_vector_conditioning(torch.randn(1, SDXL_CFG["adm_in_channels"], generator=torch.Generator().manual_seed(3)))
torch.manual_seed(0)
pipe.call()                                                # builds / tunes / captures
torch.cuda.synchronize()
print("build + first call: %.1f s" % (time.perf_counter() - t0), flush=True)
tm = {}
t1 = time.perf_counter()
pipe.call(timings=tm)
torch.cuda.synchronize()
dt = time.perf_counter() - t1
print("SDXL 1024^2, %d views, %d steps: %.2f s per call = %.3f frames/s; stages (ms): %s; UNet plan %.1f TFLOP per eval; peak mem %.1f GB"
      % (views, steps, dt, views / dt, {k: round(v, 1) for k, v in tm.items()}, pipe.runner._plan["flops"] / 1e12,
         torch.cuda.max_memory_allocated() / 2 ** 30))
