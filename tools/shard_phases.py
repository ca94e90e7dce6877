"""development aid: stage timings of one bake call as ONE rank of a view shard executes it (world-1 RCCL group, SR_SHARD_FORCE=1):
python tools/shard_phases.py VIEWS   -> per-stage wall ms (synchronised), the sampling loop against 20 x the UNet evaluation"""
import os, sys, time, torch
os.environ["SR_SHARD_FORCE"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch.distributed as dist
views = int(sys.argv[1]) if len(sys.argv) > 1 else 1
plain = "plain" in sys.argv
torch.cuda.set_device(0)
shard = None
if not plain:
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from stable_renderer_amd.parallel import ViewShard
    shard = ViewShard(views)
from stable_renderer_amd.pipeline import build_sd15_pipeline
pipe = build_sd15_pipeline(n_views=views, steps=20, cfg=8.0, shard=shard)
for _ in range(2):
    pipe.call()
torch.cuda.synchronize()
acc = {}
n = 3
for _ in range(n):
    t = {}
    pipe.call(timings=t)
    for k, v in t.items():
        acc[k] = acc.get(k, 0.0) + v / n
print("views", views, "plain" if plain else "shard", {k: round(v, 2) for k, v in acc.items()}, "sum", round(sum(acc.values()), 2))
# host-side cost of the sampling loop alone: the same call without synchronising inside (host enqueue time vs wall)
torch.cuda.synchronize()
t0 = time.perf_counter()
pipe.call()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("one call: host returned after %.1f ms, GPU done after %.1f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
if dist.is_initialized():
    dist.destroy_process_group()
