"""Do two UNet evaluations on two HIP streams overlap on the GPU?  (development: SR_VIEWS=n python tools/check_overlap2.py)
Two pipeline slots (own plans, buffers, graphs; shared weights), N evaluations each: one slot alone, the two one after another,
and the two side by side on their streams -- as hipGraph launches and as eager launches."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stable_renderer_amd import ops as O                                      # noqa: E402
from stable_renderer_amd.pipeline import InflightCalls, build_sd15_pipeline   # noqa: E402

V = int(os.environ.get("SR_VIEWS", "1"))
K = int(os.environ.get("SR_SLOTS", "2"))
O.load_tune_table(os.path.join(ROOT, "tests", "golden", "tune_table.json"))
pipe = build_sd15_pipeline(dtype=torch.float16, n_views=V, steps=3, cfg=8.0, use_graph=True)
fl = InflightCalls(pipe, K)
fl.warm(1)
plans = [p.runner._plan["step"] for p in fl.pipes]
streams = fl.streams
N = 20


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


def alone(graph):
    with torch.cuda.stream(streams[0]):
        for _ in range(N):
            plans[0].launch() if graph else plans[0].run()


def side_by_side(graph):
    for _ in range(N):
        for pl, st in zip(plans, streams):
            with torch.cuda.stream(st):
                pl.launch() if graph else pl.run()


def one_after_another(graph):
    with torch.cuda.stream(streams[0]):
        for _ in range(N):
            for pl in plans:
                pl.launch() if graph else pl.run()


for graph in (True, False):
    for f in (alone, one_after_another, side_by_side):
        f(graph)
        ms = min(timed(lambda: f(graph)) for _ in range(3))
        per = ms / N / (1 if f is alone else K)
        print("views %d  %-6s %-18s %7.2f ms for %d x %d evaluations = %.3f ms per evaluation" %
              (V, "graph" if graph else "eager", f.__name__, ms, N, 1 if f is alone else K, per), flush=True)
