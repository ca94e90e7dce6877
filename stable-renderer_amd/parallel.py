"""Multi-GPU sharding of ONE overlapped view group: one process per GPU, views split contiguously across
ranks, ``torch.distributed`` over RCCL/xGMI on GPUs (backend "nccl") or gloo in the CPU tests.

Exchange points of the path (they exist only with OverlapCorresponder):
  * per denoise step — the latent overlap needs every view's latent: all-gather of (n_local,4,h,w) fp32 (64 KiB per view at
    512^2), then every rank runs the identical overlap step on the full batch (id maps are replicated, built once) and keeps
    its own slice.  An all-gather of latents moves 80x fewer bytes than all-reducing a dense per-vertex sum table.
  * end of call — corr-map 'first' priority is frame order, so decoded frames are gathered to rank 0 and applied in order.
Independent view groups (bench.py --gpus N, weak scaling) need none of this: no data-path collective."""
import os

import torch
import torch.distributed as dist


def _single(group=None):
    """True when there is nobody to exchange with.  SR_SHARD_FORCE=1 keeps every collective of the sharded path in the run even
    in a one-rank group, so that the RCCL calls (async broadcast + wait, all_gather_into_tensor, gather) and the segment /
    collective interleave execute on a single-GPU box (tests/test_gpu_sharded.py, world-size-1 nccl group)."""
    if not dist.is_initialized():
        return True
    return dist.get_world_size(group) == 1 and os.environ.get("SR_SHARD_FORCE", "0") != "1"


def _staged(t):
    """gloo has no device collectives for every op: stage CUDA tensors through the host (tests on one GPU); RCCL runs
    them on the device directly."""
    return t.is_cuda and dist.get_backend() == "gloo"


def broadcast(t, src, group=None):
    if _single(group):
        return t
    if _staged(t):
        c = t.cpu()
        dist.broadcast(c, src=src, group=group)
        t.copy_(c)
    elif not t.is_cuda and dist.get_backend(group) == "nccl":      # RCCL moves device memory only (host-side bookkeeping tensors)
        c = t.cuda()
        dist.broadcast(c, src=src, group=group)
        t.copy_(c.cpu())
    else:
        dist.broadcast(t, src=src, group=group)
    return t


class _Done:
    """handle of a transfer that has already completed (gloo staging, one-rank groups)"""
    is_async = False

    def __init__(self, result=None, cached=False):
        self.result, self.cached = result, cached      # cached: `result` is a buffer the next gather of the same shape overwrites

    def wait(self):
        return True


class _Pending:
    """handle of a collective running on the process group's own stream; wait() orders the CURRENT stream after it"""
    is_async = True

    def __init__(self, work, result=None, cached=False):
        self.work, self.result, self.cached = work, result, cached

    def wait(self):
        self.work.wait()
        return True


def broadcast_start(t, src, group=None):
    """start a broadcast and return a handle whose ``wait()`` orders the CURRENT stream after the transfer.  RCCL: the
    collective is enqueued on the process group's own stream (it first waits for the work already on the current stream), so
    kernels launched on the current stream between this call and ``wait()`` overlap the transfer.  gloo (tests): staged and
    synchronous."""
    if _single(group):
        return _Done()
    if _staged(t) or not t.is_cuda:
        broadcast(t, src, group)
        return _Done()
    return _Pending(dist.broadcast(t, src=src, group=group, async_op=True))


class ViewShard:
    def __init__(self, n_views, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if n_views % self.world:
            raise ValueError(f"{n_views} views do not split over {self.world} ranks")
        self.n_views, self.n_local = n_views, n_views // self.world
        self.slice = slice(self.rank * self.n_local, (self.rank + 1) * self.n_local)
        self._full = {}                                  # (shape, dtype, device) -> reusable all-gather destination

    @property
    def active(self):
        """the sharded code path (collectives, external K/V source) is in use: more than one rank, or forced (SR_SHARD_FORCE)"""
        return not _single(self.group)

    def gather_latents_start(self, x_local):
        """start the all-gather of (n_local,C,h,w) -> (N,C,h,w) (rank order = view order) and return a handle: ``wait()``
        orders the current stream after the transfer, ``result`` is the full tensor.  RCCL: asynchronous on the process
        group's stream (it first waits for what is already queued on the current stream, so the caller may go on launching
        kernels that only READ x_local); gloo: staged through the host, complete on return."""
        if not self.active:
            return _Done(x_local)
        x_local = x_local.contiguous()
        if _staged(x_local):
            c = x_local.cpu()
            parts = [torch.empty_like(c) for _ in range(self.world)]
            dist.all_gather(parts, c, group=self.group)
            return _Done(torch.cat(parts, 0).to(x_local.device))
        key = (tuple(x_local.shape), x_local.dtype, str(x_local.device))
        full = self._full.get(key)
        if full is None:
            full = self._full[key] = torch.empty((self.n_views,) + tuple(x_local.shape[1:]), dtype=x_local.dtype, device=x_local.device)
        if x_local.is_cuda:
            return _Pending(dist.all_gather_into_tensor(full, x_local, group=self.group, async_op=True), full, cached=True)
        dist.all_gather(list(full.split(self.n_local)), x_local, group=self.group)
        return _Done(full, cached=True)

    def gather_latents(self, x_local):
        """blocking form of gather_latents_start (a fresh tensor: callers keep it, e.g. the id maps of a call)"""
        h = self.gather_latents_start(x_local)
        h.wait()
        return h.result.clone() if h.cached else h.result

    def overlap_step(self, x_local, step_fn, handle=None):
        """step_fn(full) mutates the full (N,C,h,w) latent in place (OverlapIndex.step on GPUs); every rank computes the
        same result and writes back its own views.  handle: a gather_latents_start() of the SAME x_local issued earlier (the
        sampler starts it before the UNet evaluation of the step: x is only read until this point, so the transfer hides
        behind the evaluation)."""
        h = handle if handle is not None else self.gather_latents_start(x_local)
        h.wait()
        full = h.result
        step_fn(full)
        if full is not x_local:
            x_local.copy_(full[self.slice])
        return x_local

    def gather_frames_to_rank0(self, frames_local):
        """decoded frames (n_local,H,W,C) -> rank 0 gets (N,H,W,C) in frame order (ordered 'first' corr-map merge)"""
        if not self.active:
            return frames_local
        frames_local = frames_local.contiguous()
        dev = frames_local.device
        if _staged(frames_local):
            frames_local = frames_local.cpu()
        out = [torch.empty_like(frames_local) for _ in range(self.world)] if self.rank == 0 else None
        dist.gather(frames_local, out, dst=0, group=self.group)
        return torch.cat(out, 0).to(dev) if self.rank == 0 else None

    def owner_of(self, global_batch_index, chunks=None):
        """global batch entry of a model call -- calc_cond_uncond_batch lays its chunks out one after the other, every chunk
        holding all views: [uncond views | cond views] in the plain case, `chunks` of them for conditioning lists -> (owner
        rank, index in that rank's local batch, which has the same chunk order over its own views)"""
        g = int(global_batch_index)
        chunk, view = g // self.n_views, g % self.n_views
        if chunks is not None and chunk >= chunks:
            raise IndexError(f"batch entry {g} outside a model call of {chunks} x {self.n_views} entries")
        return view // self.n_local, chunk * self.n_local + view % self.n_local


def timed_max_over_ranks(seconds, device):
    """bench helper: MAX over ranks of a wall time"""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
