"""bench.py — frames/sec of the render-then-diffuse frame loop (BASELINE.json metric) on N MI355X of one node.

A "step" = one bake call = 8 views: rasterise 8 frames -> EngineData -> 20 denoise steps (UNet cond+uncond, B=16, with
per-step latent overlap and K/V injection) -> VAE decode 8 x 512^2 -> corr-map update.  Inputs (meshes, textures,
weights) are resident in HBM before the timed region.  N>1 (one process per GPU): by default ONE 8-view group is view-sharded over the
ranks (--mode shard: per-step latent all-gather + per-block K/V-source broadcast over RCCL; strong scaling, value = the group's
frames / max-over-ranks time).  The replica figure (every rank its own 8-view group, weak scaling, no data-path collective) is
measured FIRST and printed beside it as `replicas`; the sharded phase then runs under a wall-clock guard, and if it fails or stalls
in a collective the line is still printed -- with the replica figure as `value`, "scaling": "weak" and the failure in `shard_error`.
--mode replica makes the replicas the headline without running the sharded phase at all.

Launch forms: `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (one rank per GPU: RANK / LOCAL_RANK /
WORLD_SIZE from the environment), or plain `python bench.py --gpus N`: with no WORLD_SIZE in the environment the process becomes a
launcher that never touches the GPU, starts N fresh rank processes, relays rank 0's JSON line and exits non-zero if a rank fails,
stalls past the guard, or the line does not say n_gpus == N.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def cpu_baseline():
    """oracle (torch-CPU fp32 restatement, kind 'port') on a bounded sample of the same workload"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import sr_oracle as ORC
    from stable_renderer_amd import synth
    from stable_renderer_amd.model_shapes import unet_names_shapes, vae_decoder_names_shapes
    from stable_renderer_amd.unet import SD15_CFG
    from stable_renderer_amd.hostcpu import cpu_share
    cores = max(1, min(cpu_share(), 16))      # affinity mask AND cgroup quota (the GPU box: 256 visible, 16 granted per GPU)
    torch.set_num_threads(cores)
    ns, norms = unet_names_shapes(SD15_CFG)
    sd = synth.synth_state_dict(ns, seed=0, norm_names=norms)
    vns, vnorms = vae_decoder_names_shapes()
    sdv = synth.synth_state_dict(vns, seed=2, norm_names=vnorms)
    g = torch.Generator().manual_seed(0)
    x, ctx = torch.randn(2, 4, 64, 64, generator=g), torch.randn(2, 77, 768, generator=g)
    with torch.no_grad():
        t0 = time.time()
        ORC.unet_forward(sd, SD15_CFG, x, torch.tensor([500.0, 500.0]), ctx)
        t_unet = time.time() - t0
        t0 = time.time()
        ORC.vae_decoder(sdv, torch.randn(1, 4, 32, 32, generator=g))
        t_vae256 = time.time() - t0
    t_frame = 20 * t_unet + t_vae256 * (2548.9 / 624.3)          # VAE cost scaled 256^2 -> 512^2 by FLOPs
    return {"value": 1.0 / t_frame, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "1 UNet eval (cond+uncond, 1 view, 64x64 latent, fp32) = %.1fs x20 steps + 1 VAE decode at 256^2 = %.1fs "
                      "scaled x4.08 to 512^2; extrapolated per view, overlap step / raster excluded (<0.1%%)" % (t_unet, t_vae256)}


def recorded_igemm_traffic(lib_hash, views=8):
    """HBM-side bytes of the igemm family per UNet evaluation: (2 x FETCH_SIZE + WRITE_SIZE) from the two --pmc passes of
    `bench.py --roofline-only` (tools/profile_round.sh -> profiles/rNN_igemm_traffic.json).  PMC counters cannot be read from
    inside the process, so this is a RECORDED figure: it is quoted only when the newest record was taken on kernels with the
    source hash of the library loaded now, otherwise (None, file)."""
    import glob
    # (one record per batch size: rNN_igemm_traffic.json is the headline's B = 16, rNN_igemm_traffic_viewsV.json a V-view call's B = 2V)
    recs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_igemm_traffic%s.json" % ("" if views == 8 else "_views%d" % views))))
    if not recs:
        return None, None
    with open(recs[-1]) as f:
        rec = json.load(f)
    name = os.path.relpath(recs[-1], ROOT)
    if rec.get("source_hash") != lib_hash or rec.get("hbm_bytes_per_eval") is None:
        return None, name
    return float(rec["hbm_bytes_per_eval"]), name


class RegionGuard:
    """wall-clock bound on a region that contains collectives: a rank stuck in `Work.wait()` / a barrier past `limit_s` runs
    `on_expire()` (rank 0: print what has been measured) and leaves the process with `code` -- from a timer thread, through
    os._exit: the main thread may be blocked inside the runtime and no re-exec of a process that touched the GPU ever happens"""

    def __init__(self, what, limit_s, on_expire=None, code=4):
        self.what, self.limit_s, self.on_expire, self.code = what, limit_s, on_expire, code
        self._t = None

    def _fire(self):
        print("bench.py: %s exceeded its %.0f s guard (a rank stuck in a collective?)" % (self.what, self.limit_s), file=sys.stderr, flush=True)
        code = self.code
        try:
            if self.on_expire is not None:
                code = self.on_expire()
        finally:
            sys.stdout.flush()
            os._exit(self.code if code is None else code)

    def __enter__(self):
        self._t = threading.Timer(self.limit_s, self._fire)
        self._t.daemon = True
        self._t.start()
        return self

    def __exit__(self, *exc):
        self._t.cancel()
        return False


def launch_ranks(a, argv):
    """`python bench.py --gpus N` without WORLD_SIZE: start N rank processes (fresh children, started BEFORE this process makes
    any GPU call -- it never makes one; torch.cuda.device_count() does not initialise the runtime), relay rank 0's JSON line, bound
    the whole run, and fail loudly: fewer than N GPUs, a failing rank, a stall, or a line whose n_gpus is not N -> rc != 0"""
    n = a.gpus
    backend = os.environ.get("SR_DIST_BACKEND", "nccl")
    have = torch.cuda.device_count()
    if have < (n if backend == "nccl" else 1):
        print("bench.py: --gpus %d but %d GPU(s) visible" % (n, have), file=sys.stderr)
        return 2
    if a.mode != "replica" and a.views % n:
        print("bench.py: %d views do not split over %d ranks (--mode shard needs views %% gpus == 0)" % (a.views, n), file=sys.stderr)
        return 2
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    from stable_renderer_amd.hostcpu import cpu_share
    threads = os.environ.get("OMP_NUM_THREADS") or str(max(1, cpu_share() // n))      # the ranks share this process's CPU quota
    procs = []
    for r in range(n):
        env = dict(os.environ, OMP_NUM_THREADS=threads, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True if r == 0 else None))
    limit = float(os.environ.get("SR_BENCH_LIMIT_S", "2400"))
    out0 = []
    rd = threading.Thread(target=lambda: out0.extend(procs[0].stdout.readlines()), daemon=True)
    rd.start()
    t0, rc, why = time.time(), 0, None
    while True:
        codes = [p_.poll() for p_ in procs]
        if any(c not in (None, 0) for c in codes):
            rc, why = 1, "rank %d exited with code %d" % next((i, c) for i, c in enumerate(codes) if c not in (None, 0))
            break
        if all(c == 0 for c in codes):
            break
        if time.time() - t0 > limit:
            rc, why = 3, "no result after %.0f s (SR_BENCH_LIMIT_S)" % limit
            break
        time.sleep(0.2)
    for p_ in procs:                                          # exactly the processes started above, by handle
        if p_.poll() is None:
            p_.terminate()
    for p_ in procs:
        try:
            p_.wait(timeout=20)
        except subprocess.TimeoutExpired:
            p_.kill()
    rd.join(timeout=5)
    line = next((ln for ln in reversed(out0) if ln.lstrip().startswith("{")), None)
    if why is not None:
        print("bench.py launcher: " + why, file=sys.stderr)
    if line is not None:
        sys.stdout.write(line if line.endswith("\n") else line + "\n")
        try:
            if rc == 0 and json.loads(line).get("n_gpus") != n:
                print("bench.py launcher: the line says n_gpus=%r, asked for %d" % (json.loads(line).get("n_gpus"), n), file=sys.stderr)
                rc = 5
        except ValueError:
            rc = rc or 5
    elif rc == 0:
        print("bench.py launcher: rank 0 printed no JSON line", file=sys.stderr)
        rc = 5
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--dtype", default="f16", choices=["f16", "f32"])
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--denoise-steps", type=int, default=20)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="print per-stage wall times of one extra (synchronised) call to stderr")
    ap.add_argument("--roofline-only", action="store_true",
                    help="build the pipeline, run ONE call, then only the igemm-subset replays of the roofline object (the command "
                         "profiled with rocprofv3 for profiles/: its kernel stats then cover exactly those launches)")
    ap.add_argument("--inflight", type=int, default=3,
                    help="bake calls in flight on each GPU (pipeline.InflightCalls: one host thread + HIP stream + launch plans per "
                         "slot, RNG draws and corr-map updates kept in call order -> same results as 1); 1 = the plain call loop")
    ap.add_argument("--mode", default=None, choices=["replica", "shard"],
                    help="shard (default when launched with more than one rank): ONE 8-view group split over the GPUs -- one view "
                         "per GPU at N = 8 -- with the per-step latent all-gather and the per-block K/V-source broadcast over RCCL "
                         "(strong scaling; the replica figure is measured after it and printed beside it as `replicas`); "
                         "replica (default on one GPU): every GPU bakes its own 8-view group (weak scaling, no collective)")
    ap.add_argument("--no-replicas-beside", action="store_true", help="shard mode: skip the replica measurement that follows it")
    ap.add_argument("--shard-inflight", action="store_true", default=None,
                    help="shard mode: --inflight K sharded calls in flight per rank, every slot with its own process group.  Default "
                         "ON in the forced one-rank rehearsal (SR_SHARD_FORCE=1 with one rank: what it exercises -- several RCCL "
                         "communicators in flight on one GPU -- is what 2-rank gloo and one-rank RCCL tests cover), opt-in with more "
                         "than one rank: whether RCCL's kernels of several communicators co-schedule on every GPU of a node is "
                         "something only a node can show")
    ap.add_argument("--no-shard-inflight", dest="shard_inflight", action="store_false")
    ap.add_argument("--workload", default="sd15-512", choices=["sd15-512", "sdxl-1024"],
                    help="sd15-512: the configuration BASELINE.json's metric is quoted on (default); sdxl-1024: BASELINE config 5, "
                         "the SDXL base UNet (2.57 B parameters) at 1024x1024 through the same raster / overlap / K-V injection / "
                         "VAE / corr-map path (one call at a time: a slot holds ~120 GB of plans)")
    ap.add_argument("--record-check", action="store_true",
                    help="write the check call's checksum to tests/golden/bench_check.json (done once per kernel change, on a GPU box, "
                         "with the pinned tuner table in force)")
    ap.add_argument("--controlnets", action="store_true",
                    help="attach the depth + normal ControlNet pair driven by the G-buffers (BASELINE config 4's composition)")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ and os.environ.get("SR_SHARD_FORCE") != "1":
        sys.exit(launch_ranks(a, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if "WORLD_SIZE" in os.environ and a.gpus != world and rank == 0:
        print("bench.py: --gpus %d ignored, the launcher set WORLD_SIZE=%d" % (a.gpus, world), file=sys.stderr)
    if a.mode is None:
        a.mode = "shard" if world > 1 else "replica"
    if a.shard_inflight is None:
        a.shard_inflight = world == 1 and os.environ.get("SR_SHARD_FORCE") == "1"
    if a.mode == "shard" and a.views % world:
        sys.exit("bench.py: %d views do not split over %d ranks (--mode shard needs views %% ranks == 0; --mode replica has no such "
                 "constraint)" % (a.views, world))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # one process per GPU; SR_DIST_BACKEND=gloo lets the multi-process path be rehearsed on a box with fewer GPUs than ranks
    # (ranks then share devices, collectives are staged through the host)
    backend = os.environ.get("SR_DIST_BACKEND", "nccl")
    local = local % max(torch.cuda.device_count(), 1)
    guard_s = float(os.environ.get("SR_BENCH_GUARD_S", "600"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with RegionGuard("process-group rendezvous", guard_s):
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            else:
                dist.init_process_group(backend)
    torch.cuda.set_device(local)
    from stable_renderer_amd.hostcpu import limit_torch_threads
    limit_torch_threads(world)                                # start-up (weight synthesis, packing) is CPU work: fit the CPU quota
    from stable_renderer_amd import _lib as L
    from stable_renderer_amd import ops as O
    from stable_renderer_amd.pipeline import InflightCalls, build_sd15_pipeline
    # the tile tuner's table is pinned (tests/golden/tune_table.json, the table the full-size parity tests run on) unless the caller
    # brings a cache of its own or asks for free tuning: no tuning launches at start-up, and a tie between two tiles cannot flip a
    # last fp16 bit between processes -- which is what lets `check.frames_checksum` be compared with a recorded value
    tt = os.path.join(ROOT, "tests", "golden", "tune_table.json")
    pinned = os.path.exists(tt) and not os.environ.get("SR_AUTOTUNE_CACHE") and os.environ.get("SR_BENCH_TUNE", "pinned") == "pinned"
    if pinned:
        O.load_tune_table(tt)
    dtype = torch.float16 if a.dtype == "f16" else torch.float32
    want_shard = a.mode == "shard" and (world > 1 or os.environ.get("SR_SHARD_FORCE") == "1")
    if want_shard and world == 1 and dist is None:            # forced one-rank rehearsal: the collectives need a group to run in
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=0, world_size=1)
    controls = [("depth", 1.0), ("normal", 1.0)] if a.controlnets else None     # BASELINE config 4's pair (miku-control.json)
    sdxl = a.workload == "sdxl-1024"
    res = 1024 if sdxl else 512
    extra = {}
    if sdxl:
        from stable_renderer_amd.unet import SDXL_CFG
        extra = dict(W=1024, H=1024, unet_cfg=dict(SDXL_CFG))
        a.inflight, a.no_cpu_baseline = 1, True               # the CPU port of this size takes hours: not a bounded sample
    cdev = "cuda" if backend == "nccl" else "cpu"
    t_start = time.perf_counter()

    def note(msg):
        """progress on stderr (rank 0): a default run takes minutes, most of it building weights and the CPU baseline"""
        if rank == 0:
            print("[bench %6.1f s] %s" % (time.perf_counter() - t_start, msg), file=sys.stderr, flush=True)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(x):
        if dist is None:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    def build(shard):
        note("building the %s pipeline (random-init weights, launch plans)" % ("view-sharded" if shard is not None else "one-GPU"))
        pipe_ = build_sd15_pipeline(dtype=dtype, n_views=a.views, steps=a.denoise_steps, cfg=8.0, use_graph=not a.no_graph,
                                    device="cuda:%d" % local, shard=shard, controls=controls, **extra)
        if sdxl:
            pipe_.runner.set_vector_conditioning(torch.randn(1, SDXL_CFG["adm_in_channels"], generator=torch.Generator().manual_seed(3)))
        pipe_.runner.time_comm = shard is not None
        return pipe_

    def roofline(pipe_, shard):
        """roofline of the dominant kernel family (implicit-GEMM conv/linear, MFMA bound): algorithmic FLOPs of every igemm launch
        of one UNet evaluation / their summed duration, measured with events on the stream they are launched on"""
        class _Seq:                                           # a view-sharded step plan is a sequence of cut segments
            def __init__(self, plans):
                self.plans = [p_ for p_ in plans if p_.n > 0]
                self.n = sum(p_.n for p_ in self.plans)
                self.op_flops = [f for p_ in self.plans for f in p_.op_flops]

            def igemm_bytes(self):
                return sum(p_.igemm_bytes() for p_ in self.plans)

            def run(self):
                for p_ in self.plans:
                    p_.run()
        note("roofline: igemm launches of one UNet evaluation replayed under events")
        sched = pipe_.runner._plan.get("schedule") or []
        plans = [r[1] for r in sched if r[0] == "run"] or [pipe_.runner._plan["step"]]
        plan = _Seq(plans)
        sub = _Seq([p_.subset(L.OP_IGEMM) for p_ in plans])
        flops = float(sum(sub.op_flops))
        for _ in range(2):
            sub.run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = int(os.environ.get("SR_ROOFLINE_REPS", "5"))
        e0.record()
        for _ in range(reps):
            sub.run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        peak = 2500.0 if a.dtype == "f16" else 157.3
        traffic, traffic_file = None, None
        if a.dtype == "f16" and shard is None and not a.controlnets and not sdxl:
            traffic, traffic_file = recorded_igemm_traffic(L.lib().sr_source_hash().decode(), a.views)
        ach = flops / (ms * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": "igemm family: igemm_kernel / igemm_group_kernel tiles + conv3p_kernel (implicit-GEMM conv/linear, all %d ops of one UNet eval, B=%d)" % (sub.n, (a.views // world if shard is not None else a.views) * 2),
                "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                # HBM-side bytes of these launches per UNet evaluation from the FETCH_SIZE (x2, gfx950) + WRITE_SIZE PMC passes of
                # this very replay: recorded_igemm_traffic() -- null unless the record matches the loaded kernels
                "traffic": traffic, "traffic_recorded_in": traffic_file,
                "algorithmic_bytes": sub.igemm_bytes(),
                "launches": sub.n, "avg_launch_us": round(ms * 1e3 / max(sub.n, 1), 2), "flops_per_eval": flops}
        full = pipe_.runner._plan["flops"]
        e0.record()
        for _ in range(reps):
            plan.run()
        e1.record()
        torch.cuda.synchronize()
        ms_full = e0.elapsed_time(e1) / reps
        roof["unet_eval_ms"] = round(ms_full, 3)
        roof["unet_eval_launches"] = plan.n
        roof["unet_eval_tflops"] = round(full / (ms_full * 1e-3) / 1e12, 2)
        return roof

    def check_frames(pipe_, shard):
        """outside the timed region: the decoded frames of one more call must be finite; their checksum is compared with the value
        recorded beside the pinned tuner table (tests/golden/bench_check.json) when this run used that table"""
        note("result check: one more call, frames finite + checksum")
        pipe_.frame0 = 0                                      # the check call is a function of (scene, seed, kernels) alone:
        torch.manual_seed(4321)                               # frames 0..N-1, a fixed RNG state (sampling never reads the corr-map)
        img = pipe_.call()
        torch.cuda.synchronize()
        finite = bool(torch.isfinite(img).all())
        chk = {"frames_finite": finite, "frames_shape": list(img.shape), "frames_mean": round(float(img.double().mean()), 6),
               "frames_checksum": round(float(img.double().sum()), 3),
               "corrmap_texels_written": int(pipe_.scene.corrmap._writtens.sum()) if (shard is None or rank == 0) else None}
        assert finite, "decoded frames are not finite"
        key = "%s/%s/views%d/steps%d%s%s" % (a.workload, a.dtype, a.views, a.denoise_steps, "/controlnets" if a.controlnets else "",
                                              "" if shard is None else "/shard%d.%d" % (world, rank))
        rec_path = os.path.join(ROOT, "tests", "golden", "bench_check.json")
        recs = {}
        if os.path.exists(rec_path):
            with open(rec_path) as f:
                recs = json.load(f)
        h_ = L.lib().sr_source_hash().decode()
        if a.record_check and pinned:
            recs[key] = {"frames_checksum": chk["frames_checksum"], "frames_mean": chk["frames_mean"], "source_hash": h_}
            with open(rec_path, "w") as f:
                json.dump(recs, f, indent=1, sort_keys=True)
        rec = recs.get(key)
        # matches_recorded: bit-for-bit the frames recorded for THESE kernels (null: free tuning, no record, or a record made by
        # other kernel sources); mean_close_to_recorded survives a kernel change: the picture is the same one
        chk["tuner_table"] = "pinned" if pinned else "free"
        chk["matches_recorded"] = (chk["frames_checksum"] == rec["frames_checksum"]) if (pinned and rec and rec["source_hash"] == h_) else None
        chk["mean_close_to_recorded"] = (abs(chk["frames_mean"] - rec["frames_mean"]) < 2e-3) if rec else None
        return chk

    def line(value, dt, steps, shard, inflight, comm_ms, one_at_a_time, replicas, check, roof, cpu, shard_error=None):
        return {"metric": "frames/sec @1024^2, SDXL 20-step img2img, 8-view overlap" if sdxl else "frames/sec @512^2, SD1.5 20-step img2img, 8-view overlap", "value": round(value, 4), "unit": "frames/s",
                "n_gpus": world, "steps": steps, "warmup": a.warmup, "ms_per_step": round(dt / max(steps, 1) * 1e3, 2),
                "higher_is_better": True, "scaling": "strong" if shard is not None else "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
                "config": {"workload": "bake_ball.py sphere scene %dx%d (HIP raster, corr-map proxy k=6, texcoord ids) -> %s UNet "
                                       "(%s params, random init) %d denoise steps ddim/normal cfg 8, %d views per call with "
                                       "OverlapCorresponder (per-step latent overlap + K/V injection) -> VAE decode %dx%d^2 -> corr-map "
                                       "update; zero latent + engine noise as the reference bake workflows; one call per step"
                                       % (res, res, "SDXL-base-shaped" if sdxl else "SD1.5-shaped", "2.57B" if sdxl else "859.5M",
                                          a.denoise_steps, a.views, a.views, res),
                           "views_per_call": a.views, "denoise_steps": a.denoise_steps, "resolution": res, "parallelism": ("one group view-sharded x%d" if shard is not None else "view-group replicas x%d") % world,
                           "calls_in_flight_per_gpu": inflight, "controlnets": ["depth", "normal"] if a.controlnets else []},
                "exposed_comm_ms_per_denoise_step": None if comm_ms is None else round(comm_ms / max(steps * a.denoise_steps, 1), 4),
                "value_1_in_flight": None if one_at_a_time is None else round(one_at_a_time * (1 if shard is not None else world), 4),
                "replicas": replicas, "shard_error": shard_error,
                "check": check, "roofline": roof, "cpu_baseline": cpu}

    torch.manual_seed(1234 + rank)
    # ---- N > 1, shard mode: the replica figure FIRST (every GPU bakes its own 8-view group, calls in flight, no data-path
    # collective: the only things that can stall are the barriers, and those are guarded) -- the same code as the one-GPU headline,
    # so value(N=1) x N is what perfect weak scaling would read.  Should the sharded phase then fail, this is what gets printed.
    replicas, rep = None, None
    if want_shard and world > 1 and not a.no_replicas_beside and not a.roofline_only:
        with RegionGuard("replica phase", guard_s):
            pipe_r = build(None)
            nfl = 1 if sdxl else max(1, a.inflight)
            nr = max(nfl, min(a.steps, 2 * nfl))
            fl = InflightCalls(pipe_r, nfl)
            note("replica phase: warm-up, %d calls in flight" % nfl)
            fl.warm(1)
            sync()
            note("replica phase: %d timed calls" % nr)
            t1 = time.perf_counter()
            fl.run(nr)
            sync()
            dtr = max_over_ranks(time.perf_counter() - t1)
            replicas = {"value": round(a.views * nr * world / max(dtr, 1e-9), 4), "unit": "frames/s", "scaling": "weak",
                        "steps": nr, "calls_in_flight_per_gpu": nfl, "parallelism": "view-group replicas x%d" % world}
            rep = dict(dt=dtr, steps=nr, inflight=nfl, check=check_frames(pipe_r, None), roof=roofline(pipe_r, None) if rank == 0 else None)
            del fl

    s1 = {}                                                   # the finished one-call-at-a-time sharded measurement, once there is one

    def fallback(err):
        """the sharded phase failed or stalled: rank 0 prints what HAS been measured -- the sharded line without calls in flight when
        only the in-flight phase went wrong, else the replica measurement (weak scaling, no collective)"""
        if s1:
            if rank == 0:
                out_ = dict(s1["line"])
                out_["shard_inflight_error"] = err
                print(json.dumps(out_), flush=True)
            return 0
        if rank == 0 and rep is not None:
            print(json.dumps(line(replicas["value"], rep["dt"], rep["steps"], None, rep["inflight"], None, None, replicas, rep["check"],
                                  rep["roof"], None, shard_error=err)), flush=True)
            return 0
        return 0 if rep is not None else 4

    shard = None
    try:
        with RegionGuard("sharded phase" if want_shard else "bench", guard_s if want_shard else 10 * guard_s,
                         on_expire=(lambda: fallback("the sharded phase did not finish within %.0f s" % guard_s)) if want_shard else None):
            if want_shard:
                from stable_renderer_amd.parallel import ViewShard
                shard = ViewShard(a.views)
            pipe = build(shard)
            if a.roofline_only:                               # no calls: just the UNet step plan (as a sampling run builds it)
                a.warmup, a.steps, a.no_cpu_baseline = 0, 0, True
                if pipe.controls:                             # (the ControlNet hint encoders run in _load_ctx: give them the G-buffer planes)
                    pipe.render_views()
                    pipe.runner.set_control_hints(pipe.control_hints())
                p_ = pipe.runner._ensure_plan([min(3, 2 * a.views - 1)])
                pipe.runner._load_ctx(p_)
                # real activations in every buffer the igemm replay reads: a mid-schedule latent through the WHOLE plan once (a replay
                # over the zero-filled buffers of a fresh plan multiplies zeros: ~5 % faster at the clocks idle data lines allow)
                p_["x"].copy_(torch.randn(p_["x"].shape, generator=torch.Generator().manual_seed(5)).to(p_["x"].device) * 0.8)
                p_["t"].fill_(500.0)
                for r_ in (p_.get("schedule") or [("run", p_["step"])]):
                    if r_[0] == "run":
                        r_[1].run()
                torch.cuda.synchronize()
            inflight = max(1, a.inflight) if ((shard is None or a.shard_inflight) and not a.roofline_only) else 1
            if inflight > 1:
                fl = InflightCalls(pipe, inflight)
                note("warm-up: %d call(s) on each of %d slots (plans built and captured here)" % (a.warmup, inflight))
                fl.warm(a.warmup)                             # every slot builds / tunes / captures alone, W calls each
                sync()
                note("timed region: %d calls, %d in flight" % (a.steps, inflight))
                t0 = time.perf_counter()
                fl.run(a.steps)                               # exactly K calls, call c on slot c % inflight
                sync()
            else:
                note("warm-up: %d call(s) (plans built and captured here)" % a.warmup)
                for _ in range(a.warmup):
                    pipe.call()
                sync()
                if shard is not None:
                    pipe.runner.exposed_comm_ms()             # drop the warm-up's records
                note("timed region: %d calls, one at a time" % a.steps)
                t0 = time.perf_counter()
                for _ in range(a.steps):
                    pipe.call()
                sync()
            dt = max(time.perf_counter() - t0, 1e-9)
            comm_ms = pipe.runner.exposed_comm_ms() if shard is not None else None     # compute-stream stalls in the K/V-source waits
            dt = max_over_ranks(dt)
            frames = a.views * a.steps * (1 if shard is not None else world)
            # ---- outside the timed region: the same loop one call at a time (the headline keeps `inflight` calls in flight on the
            # GPU), and a result check -- the decoded frames of one more call must be finite, their checksum goes into the line
            one_at_a_time, check = None, None
            if not a.roofline_only:
                if inflight > 1:
                    n1 = max(1, min(a.steps, 4))
                    note("outside the timed region: %d calls one at a time" % n1)
                    sync()
                    t1 = time.perf_counter()
                    for _ in range(n1):
                        pipe.call()
                    sync()
                    one_at_a_time = a.views * n1 / max(time.perf_counter() - t1, 1e-9)
                check = check_frames(pipe, shard)
            if a.breakdown and rank == 0:
                tm = {}
                pipe.call(timings=tm)
                print("stage breakdown (ms, synchronised): " + json.dumps({k: round(v, 2) for k, v in tm.items()}), file=sys.stderr)
            roof = roofline(pipe, shard) if rank == 0 else None
            shard_inflight = None
            if (shard is not None and world > 1 and inflight == 1 and a.inflight > 1 and not a.roofline_only
                    and os.environ.get("SR_BENCH_SHARD_INFLIGHT_PHASE", "1") == "1"):
                # ---- second sharded phase: the SAME group sharding with `--inflight` calls in flight per rank, every slot on its own
                # communicator -- what the one-GPU headline does too (3 calls in flight), so the strong-scaling curve compares like with
                # like.  Rehearsed over 2-rank gloo and a one-rank RCCL group only (several RCCL communicators in flight on a node is
                # what a node has to show), hence AFTER the plain sharded measurement is complete: if this phase raises or stalls, the
                # line printed is the plain sharded one with the failure in `shard_inflight_error`.
                sync()
                s1["line"] = line(frames / dt, dt, a.steps, shard, 1, comm_ms, None, replicas, check, roof, None)
                fl2 = InflightCalls(pipe, a.inflight)
                note("second sharded phase: %d calls in flight per rank" % a.inflight)
                fl2.warm(max(1, a.warmup))
                sync()
                t2 = time.perf_counter()
                fl2.run(a.steps)
                sync()
                dt2 = max_over_ranks(max(time.perf_counter() - t2, 1e-9))
                shard_inflight = {"value": round(frames / dt2, 4), "ms_per_step": round(dt2 / max(a.steps, 1) * 1e3, 2),
                                  "calls_in_flight_per_gpu": a.inflight}
                if dt2 < dt:                                  # the headline keeps calls in flight, as at N = 1
                    one_at_a_time, dt, inflight, comm_ms = frames / dt, dt2, a.inflight, None
            if dist is not None:
                sync()                                        # every rank has finished its sharded phase before the line goes out
    except Exception as e:                                    # noqa: BLE001 -- a failing sharded phase must not cost the line
        if not (want_shard and (rep is not None or s1)):
            raise
        import traceback
        traceback.print_exc()
        code = fallback("%s: %s" % (type(e).__name__, e))
        sys.stdout.flush()
        os._exit(code)                                        # the other ranks may sit in a collective this rank will never join
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        note("CPU baseline: the oracle on a bounded sample of the same workload")
        cpu = cpu_baseline()
    note("done")
    if rank == 0:
        out = line(frames / dt, dt, a.steps, shard, inflight, comm_ms, one_at_a_time, replicas, check, roof, cpu)
        if shard_inflight is not None:
            out["shard_calls_in_flight"] = shard_inflight
        print(json.dumps(out), flush=True)
    if dist is not None:
        with RegionGuard("process-group teardown", 60, code=0):
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
