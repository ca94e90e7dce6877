"""development aid: host-side (Python) cost of a view-sharded bake call as one rank executes it (world-1 RCCL group): cProfile of
three calls WITHOUT device synchronisation inside, against the wall time of the same calls.  python tools/profile_shard_host.py VIEWS"""
import cProfile
import os
import pstats
import sys
import time

import torch

os.environ["SR_SHARD_FORCE"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29545")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch.distributed as dist                                              # noqa: E402

views = int(sys.argv[1]) if len(sys.argv) > 1 else 1
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from stable_renderer_amd import ops as O                                      # noqa: E402
from stable_renderer_amd.parallel import ViewShard                            # noqa: E402
from stable_renderer_amd.pipeline import build_sd15_pipeline                  # noqa: E402

O.load_tune_table(os.path.join(ROOT, "tests", "golden", "tune_table.json"))
pipe = build_sd15_pipeline(n_views=views, steps=20, cfg=8.0, shard=ViewShard(views))
for _ in range(2):
    pipe.call()
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(3):
    pipe.call()
t_host = time.perf_counter() - t0
pr.disable()
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("views/rank %d: host returned after %.1f ms per call, device finished after %.1f ms per call" % (views, t_host / 3 * 1e3, t_all / 3 * 1e3))
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
dist.destroy_process_group()
