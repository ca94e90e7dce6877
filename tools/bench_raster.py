"""Time sr_raster_draw on the bake_ball sphere at 512x512 (HIP events on torch's current stream) and state the G-buffer
write rate against the HBM peak.  usage: python tools/bench_raster.py [segments]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stable_renderer_amd import scene as S          # noqa: E402


def main():
    seg = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    W = H = 512
    mesh = S.Mesh.Sphere(seg)
    cam = S.Camera((0, 0, 4.0), (0, 0, 0))
    g = torch.Generator().manual_seed(0)
    noise = torch.randn(64, 64, 4, generator=g).half().cuda()
    task = S.DrawTask(mesh, S.scale((1.5,) * 3) if callable(getattr(S, "scale", None)) else np.eye(4, dtype=np.float32),
                      use_texcoord_id=True, id_size=(64, 64), noise_tex=noise, corrmap_k=6)
    gb = S.GBuffer(W, H)
    view, proj = cam.view(), cam.projection(1.0)
    for _ in range(5):
        gb.clear(); gb.draw(task, view, proj)
    torch.cuda.synchronize()
    cov = float((gb.id[..., 0] != 0).float().mean())
    n = 200
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        gb.draw(task, view, proj)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    byts = 68 * W * H
    print("sphere seg=%d nt=%d coverage=%.2f: %.1f us per draw (setup + tiles), %.2f TB/s of G-buffer writes (%.1f%% of 8 TB/s)"
          % (seg, mesh.tris.shape[0], cov, us, byts / us / 1e6, byts / us / 1e6 / 8 * 100))


if __name__ == "__main__":
    main()
