"""Where the start-up seconds of a pipeline go (development): cProfile over build_sd15_pipeline + the first call, twice."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stable_renderer_amd import ops as O                                      # noqa: E402
from stable_renderer_amd.pipeline import build_sd15_pipeline                  # noqa: E402

O.load_tune_table(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "tune_table.json"))
for rep in range(2):
    pr = cProfile.Profile()
    t0 = time.time()
    pr.enable()
    pipe = build_sd15_pipeline(dtype=torch.float16, n_views=8, steps=4, cfg=8.0, use_graph=True)
    torch.cuda.synchronize()
    t1 = time.time()
    pipe.call()
    torch.cuda.synchronize()
    pr.disable()
    print("rep %d: build %.1f s, first call %.1f s" % (rep, t1 - t0, time.time() - t1))
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
    del pipe
