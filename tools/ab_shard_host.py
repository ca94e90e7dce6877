"""development aid: what the K/V-source broadcasts cost one rank of a view shard with calls in flight, beyond the cut of the plan into
segments: the one-rank rehearsal (world-1 RCCL group) as it is, and with broadcast_start replaced by a no-op (WRONG results on a real
group: a measurement only).  python tools/ab_shard_host.py VIEWS INFLIGHT"""
import os
import sys
import time

import torch

os.environ["SR_SHARD_FORCE"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29546")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch.distributed as dist                                              # noqa: E402

views = int(sys.argv[1]) if len(sys.argv) > 1 else 1
K = int(sys.argv[2]) if len(sys.argv) > 2 else 3
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from stable_renderer_amd import ops as O                                      # noqa: E402
from stable_renderer_amd import parallel as PAR                               # noqa: E402
from stable_renderer_amd.pipeline import InflightCalls, build_sd15_pipeline   # noqa: E402

O.load_tune_table(os.path.join(ROOT, "tests", "golden", "tune_table.json"))
real = PAR.broadcast_start
for name, fn in (("as it is", real), ("broadcasts skipped", lambda t, src, group=None: PAR._Done())):
    PAR.broadcast_start = fn
    pipe = build_sd15_pipeline(n_views=views, steps=20, cfg=8.0, shard=PAR.ViewShard(views))
    fl = InflightCalls(pipe, K)
    fl.warm(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fl.run(3 * K)
    torch.cuda.synchronize()
    print("views/rank %d, %d in flight, %s: %.1f ms per call" % (views, K, name, (time.perf_counter() - t0) / (3 * K) * 1e3), flush=True)
    del fl, pipe
dist.destroy_process_group()
