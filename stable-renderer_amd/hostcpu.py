"""CPU share of this process: min(affinity mask, cgroup CPU quota).  A GPU box shows every host core (256) to a container that
may run on 16 of them; torch sizes its CPU thread pool from the former, and 128 threads on a 16-CPU quota spend their time being
throttled (weight synthesis + packing at start-up: 14 s alone, 46 s in each of two such processes side by side)."""
import math
import os


def cpu_share():
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", None)):
        try:
            with open(path) as f:
                txt = f.read().strip()
            if parse is not None:
                quota, period = parse(txt)
            else:
                quota = txt
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = f.read().strip()
            if quota not in ("max", "-1"):
                n = min(n, max(1, math.ceil(int(quota) / int(period))))
            break
        except (OSError, ValueError):
            continue
    return max(1, n)


def limit_torch_threads(parts=1):
    """size torch's intra-op pool (and OMP_NUM_THREADS for child processes) to this process's share / parts; returns the count"""
    import torch
    n = max(1, cpu_share() // max(1, parts))
    if torch.get_num_threads() > n:
        torch.set_num_threads(n)
    os.environ["OMP_NUM_THREADS"] = str(n)
    return n
