"""micro-benchmark of the per-step latent overlap (sr_overlap_step) on the bench workload's id maps: the bake_ball scene, 8 views at
512x512, 64x64 latents (development tool): python tools/bench_overlap.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stable_renderer_amd import ops as O
from stable_renderer_amd.pipeline import BakeBallScene
from stable_renderer_amd import scene as S


def main(views=8, reps=50):
    sc = BakeBallScene(device="cuda")
    gb = S.GBuffer(sc.W, sc.H, device="cuda")
    ids = []
    for f in range(views):
        gb.render(sc.tasks(f), sc.camera)
        ids.append(gb.id.clone())
    ids = torch.stack(ids).contiguous()
    idx = O.OverlapIndex(ids, 64, 64)
    x = torch.randn(views, 4, 64, 64, device="cuda")
    seg = (idx.vid_off[1:] - idx.vid_off[:-1])
    cv = idx.cell_vid[idx.cell_vid >= 0].long()
    print("valid pixels %d, vertices with entries %d, covered cells %d, mean segment of a covered cell's vertex %.1f (max %d)" % (
        idx.n_valid, int((seg > 0).sum()), cv.numel(), float(seg[cv].float().mean()), int(seg.max())))
    for _ in range(5):
        idx.step(x.clone(), 0.5)
    torch.cuda.synchronize()
    xs = [x.clone() for _ in range(reps)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        idx.step(xs[i], 0.5)
    e1.record()
    torch.cuda.synchronize()
    print("sr_overlap_step: %.1f us per step (%d views, 512x512 ids, 64x64 latents)" % (e0.elapsed_time(e1) / reps * 1e3, views))


if __name__ == "__main__":
    main()
