"""diagnostic: the two-rank sharded flow with stage markers (run: python tools/debug_shard.py)"""
import os, sys, torch
import torch.distributed as dist
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def worker(rank, world, port):
    sys.path.insert(0, ROOT)
    log = open(os.path.join(ROOT, "gpurun_out", f"shard_rank{rank}.log"), "w")
    def mark(s):
        log.write(s + "\n"); log.flush()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mark("pg up")
    from stable_renderer_amd import synth, ops as O
    from stable_renderer_amd.model_shapes import unet_names_shapes
    from stable_renderer_amd.unet import UNet, SD15_CFG
    from stable_renderer_amd.sampling import DiffusionRunner
    from stable_renderer_amd.parallel import ViewShard
    torch.cuda.set_device(0)
    cfg = dict(SD15_CFG, model_channels=64, context_dim=64)
    ns, norms = unet_names_shapes(cfg)
    net = UNet(synth.synth_state_dict(ns, seed=1, norm_names=norms), cfg, dtype=torch.float32)
    mark("unet built")
    N, h, w, H, W = 4, 8, 8, 64, 64
    g = torch.Generator().manual_seed(3)
    ids = torch.zeros(N, H, W, 4, dtype=torch.int32); ids[..., 0] = 1
    ids[..., 3] = torch.randint(0, 300, (N, H, W), generator=g, dtype=torch.int32)
    ids[torch.rand(N, H, W, generator=g) < 0.2] = 0
    ids = ids.cuda()
    noise = torch.randn(N, 4, h, w, generator=g)
    pos, neg = torch.randn(1, 77, 64, generator=g), torch.randn(1, 77, 64, generator=g)
    idx_all = O.OverlapIndex(ids, h, w)
    torch.cuda.synchronize(); mark("overlap index built")
    def run(shard):
        n_loc = N if shard is None else shard.n_local
        r = DiffusionRunner(net, n_loc, h, w, 5.0, use_graph=False, shard=shard)
        r.set_conditioning(pos, neg)
        def cb(ctx):
            torch.cuda.synchronize(); mark(f"  step {ctx.step_index} model done t={ctx.timestep}")
            if ctx.timestep < 500: return
            if shard is None: idx_all.step(ctx.noise, 0.5)
            else: shard.overlap_step(ctx.noise, lambda full: idx_all.step(full, 0.5))
            torch.cuda.synchronize(); mark("  overlap done")
        torch.manual_seed(99)
        nz = noise if shard is None else noise[shard.slice]
        out, inj = r.sample(nz, 3, "ddim", "normal", inject_n_rand=1, step_callback=cb)
        torch.cuda.synchronize()
        return out, inj
    base, inj0 = run(None); mark(f"baseline done inj {inj0}")
    sh = ViewShard(N)
    mine, inj1 = run(sh); mark(f"sharded done inj {inj1}")
    full = sh.gather_latents(mine)
    mark(f"err {(full - base).abs().max().item()}")
    dist.destroy_process_group()

if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=worker, args=(r, 2, 29811)) for r in range(2)]
    [p.start() for p in ps]
    [p.join(timeout=150) for p in ps]
    for p in ps:
        if p.is_alive(): p.kill()
    print([p.exitcode for p in ps])
