"""world_size-2 gloo test of the view-sharded exchange (parallel.py): sharded overlap step == single-process result."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sr_oracle as O
        from stable_renderer_amd.parallel import ViewShard, timed_max_over_ranks
        d = np.load(os.path.join(GOLD, "overlap_step.npz"))
        ids, x = d["b_ids"], torch.from_numpy(d["b_x"])             # 4 views
        sh = ViewShard(4)
        assert sh.n_local == 2
        x_local = x[sh.slice].clone()

        def step(full):
            full.copy_(O.overlap_step(full, ids, 0.1))              # oracle as the compute stand-in on CPU
        sh.overlap_step(x_local, step)
        frames = torch.full((2, 3, 3, 3), float(rank))
        g = sh.gather_frames_to_rank0(frames)
        tmax = timed_max_over_ranks(1.0 + rank, "cpu")
        q.put((rank, x_local.numpy(), None if g is None else g.numpy(), tmax))
    finally:
        dist.destroy_process_group()


def test_view_sharded_overlap_matches_single_process():
    port = 29500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    d = np.load(os.path.join(GOLD, "overlap_step.npz"))
    ref = d["b_out"]                                               # the reference's own output for all 4 views
    for rank, xl, g, tmax in res:
        assert np.allclose(xl, ref[rank * 2:(rank + 1) * 2], atol=2e-6, rtol=1e-6)
        assert tmax == 2.0
        if rank == 0:
            assert g.shape == (4, 3, 3, 3) and (g[:2] == 0).all() and (g[2:] == 1).all()
        else:
            assert g is None
