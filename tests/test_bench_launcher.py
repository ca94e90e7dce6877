"""bench.py as its own launcher (`python bench.py --gpus N` with no WORLD_SIZE in the environment): the parent never touches the
GPU, starts N rank processes, relays rank 0's line and fails loudly."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(kw)
    return env


def test_launcher_refuses_more_gpus_than_visible():
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(max(n, 2)), "--steps", "1", "--warmup", "0"],
                       env=_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 2, r.stderr
    assert "GPU(s) visible" in r.stderr
    assert r.stdout.strip() == ""


def test_launcher_refuses_views_that_do_not_split():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--views", "8"],
                       env=_env(SR_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=120)
    if torch.cuda.device_count() == 0:                        # gloo rehearsal needs one GPU to share: refused earlier, same code
        assert r.returncode == 2
        return
    assert r.returncode == 2 and "do not split" in r.stderr


def test_region_guard_leaves_the_process():
    """a region that never returns (a rank stuck in a collective) ends the process from the timer thread with the guard's code,
    after on_expire() has printed what was measured"""
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "def last_words():\n    print('{\"fallback\": true}', flush=True); return 0\n"
            "with bench.RegionGuard('test region', 0.5, on_expire=last_words):\n    time.sleep(60)\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and json.loads(r.stdout) == {"fallback": True}
    assert "exceeded its" in r.stderr
    code = code.replace("on_expire=last_words", "code=4")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 4 and r.stdout == ""


@pytest.mark.gpu
def test_bench_gpus_2_self_launch_over_gloo_on_one_gpu():
    """`python bench.py --gpus 2`: two fresh rank processes sharing the test GPU (gloo, host-staged collectives), the 8-view group
    sharded 4 + 4, the replica figure measured first and printed beside the sharded one, then the same sharding with two calls in flight"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
                        "--denoise-steps", "4", "--inflight", "2"],
                       env=_env(SR_DIST_BACKEND="gloo", SR_BENCH_LIMIT_S="900",
                                SR_AUTOTUNE_TABLES=os.path.join(ROOT, "tests", "golden", "tune_table_ranks.json")), capture_output=True, text=True, timeout=1000)
    assert r.returncode == 0, r.stderr[-4000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["shard_error"] is None
    assert d["config"]["parallelism"] == "one group view-sharded x2"
    assert d["replicas"] is not None and d["replicas"]["parallelism"] == "view-group replicas x2" and d["replicas"]["value"] > 0
    assert d["check"]["frames_finite"] and d["value"] > 0
    # the second sharded phase (two sharded calls in flight per rank, every slot its own process group) ran after the plain one
    assert d.get("shard_inflight_error") is None and d["shard_calls_in_flight"]["calls_in_flight_per_gpu"] == 2
    assert d["shard_calls_in_flight"]["value"] > 0
