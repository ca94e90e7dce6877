"""Host-side helpers that need no GPU: the CPU share a process may use (cgroup quota x affinity) and the tuner-table files."""
import json
import os

import pytest


def test_cpu_share_reads_the_cgroup_quota(monkeypatch, tmp_path):
    from stable_renderer_amd import hostcpu
    real_open = open
    files = {}

    def fake_open(path, *a, **k):
        if path in files:
            if files[path] is None:
                raise FileNotFoundError(path)
            p = tmp_path / ("f%d" % abs(hash(path)))
            p.write_text(files[path])
            return real_open(p, *a, **k)
        return real_open(path, *a, **k)
    monkeypatch.setattr(hostcpu.os, "sched_getaffinity", lambda pid: set(range(256)), raising=False)
    monkeypatch.setattr("builtins.open", fake_open)
    files["/sys/fs/cgroup/cpu.max"] = "1600000 100000\n"                       # cgroup v2: what a GPU box shows
    assert hostcpu.cpu_share() == 16
    files["/sys/fs/cgroup/cpu.max"] = "max 100000\n"                           # no quota: the affinity mask decides
    assert hostcpu.cpu_share() == 256
    files["/sys/fs/cgroup/cpu.max"] = "150000 100000\n"                        # 1.5 CPUs -> 2 threads
    assert hostcpu.cpu_share() == 2
    files["/sys/fs/cgroup/cpu.max"] = None                                     # cgroup v1
    files["/sys/fs/cgroup/cpu/cpu.cfs_quota_us"] = "800000\n"
    files["/sys/fs/cgroup/cpu/cpu.cfs_period_us"] = "100000\n"
    assert hostcpu.cpu_share() == 8
    files["/sys/fs/cgroup/cpu/cpu.cfs_quota_us"] = "-1\n"
    assert hostcpu.cpu_share() == 256
    files["/sys/fs/cgroup/cpu/cpu.cfs_quota_us"] = None                        # neither: affinity only
    assert hostcpu.cpu_share() == 256


def test_limit_torch_threads_only_lowers(monkeypatch):
    import torch
    from stable_renderer_amd import hostcpu
    before = torch.get_num_threads()
    monkeypatch.setattr(hostcpu, "cpu_share", lambda: 4096)
    monkeypatch.setenv("OMP_NUM_THREADS", os.environ.get("OMP_NUM_THREADS", ""))
    try:
        assert hostcpu.limit_torch_threads() == 4096 and torch.get_num_threads() == before       # never raised
        monkeypatch.setattr(hostcpu, "cpu_share", lambda: 2)
        assert hostcpu.limit_torch_threads(parts=2) == 1 and torch.get_num_threads() == 1
        assert os.environ["OMP_NUM_THREADS"] == "1"                                                 # children inherit it
    finally:
        torch.set_num_threads(before)


def test_tuner_table_round_trip_keeps_the_tile_order(tmp_path, monkeypatch):
    from stable_renderer_amd import ops as O
    monkeypatch.setattr(O, "_TUNED", {})
    O._TUNED[(1, 16, 8, 8, 2560, 0, 1280, 3, 1, 0, 0, 0, 0, False, False, True, 0, 0, 0, 0)] = (15, 3, 1)     # columns first
    O._TUNED[(1, 16, 64, 64, 320, 0, 320, 3, 1, 0, 0, 0, 0, False, False, True, 0, 0, 0, 0)] = (8, -1)
    p = tmp_path / "t.json"
    O.save_tune_table(str(p))
    raw = json.load(open(p))
    assert sorted(len(v) for v in raw.values()) == [2, 3]
    saved = dict(O._TUNED)
    O._TUNED.clear()
    O.load_tune_table(str(p))
    assert O._TUNED == saved


def test_pinned_tables_are_well_formed():
    gold = os.path.join(os.path.dirname(__file__), "golden")
    main = json.load(open(os.path.join(gold, "tune_table.json")))
    ranks = json.load(open(os.path.join(gold, "tune_table_ranks.json")))
    assert len(main) > 1000 and ranks and not (set(main) & set(ranks))          # the ranks table only adds shapes
    for tab in (main, ranks):
        for k, v in tab.items():
            key = json.loads(k)
            assert isinstance(key, list) and all(isinstance(x, int) for x in key)
            assert len(v) in (2, 3) and 0 <= v[0] <= 15 and -1 <= v[1] <= 16 and (len(v) == 2 or v[2] in (0, 1)), (k, v)
    rec = json.load(open(os.path.join(gold, "bench_check.json")))
    assert "sd15-512/f16/views8/steps20" in rec and len(rec["sd15-512/f16/views8/steps20"]["source_hash"]) == 32


def test_recorded_bench_check_and_traffic_belong_to_these_kernels():
    """`check.matches_recorded` and `roofline.traffic` are quoted only while the records carry the source hash of the loaded kernels:
    after a kernel edit they must be re-recorded (tools/regen_tune_tables.sh, tools/profile_round.sh) -- skipped, loudly, until then"""
    import glob
    import importlib.util
    root = os.path.dirname(os.path.dirname(__file__))
    spec = importlib.util.spec_from_file_location("sr_build", os.path.join(root, "stable-renderer_amd", "csrc", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    h = mod.source_hash()
    rec = json.load(open(os.path.join(root, "tests", "golden", "bench_check.json")))["sd15-512/f16/views8/steps20"]
    traffic = json.load(open(sorted(glob.glob(os.path.join(root, "profiles", "r*_igemm_traffic.json")))[-1]))
    stale = [n for n, r in (("tests/golden/bench_check.json", rec), ("profiles/*_igemm_traffic.json", traffic)) if r["source_hash"] != h]
    if stale:
        pytest.skip("recorded on other kernel sources, re-record before the round ends: " + ", ".join(stale))
