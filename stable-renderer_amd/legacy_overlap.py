"""Legacy "latent overlapping" (the dict-based CorrespondenceMap path of legacy_codes/stable_rendering_algo), same class
names and call contract: ``Scheduler`` (overlap/overlap_scheduler.py:8-106), the four ``OverlapAlgorithm``s
(overlap/algorithms.py:34-118), ``Overlap`` / ``ResizeOverlap`` (overlap/overlap.py:18-222).  The per-vertex Python loop of the
reference becomes one HIP kernel (``sr_legacy_overlap``) over a CSR of vertex traces built once per id-map batch; with a kernel
radius > 0, where the reference's in-place update order is observable, the vertices are replayed in conflict-free levels of that
order (``sr_legacy_levels`` + ``sr_legacy_overlap_seq``) on a corr-map-resolution working tensor."""
import math

import torch

from . import _lib as L
from . import ops as O


def value_interpolation(x, start, end, power=1.0, interpolate_function="constant"):      # overlap/utils.py:24-53
    assert 0 <= x <= 1 and power >= 0
    if interpolate_function == "constant":
        return start
    if interpolate_function == "linear":
        return start + (end - start) * x ** power
    if interpolate_function == "cosine":
        return start + (end - start) * (1 + math.cos(x ** power * math.pi)) / 2
    if interpolate_function == "exponential":
        return start * (end / start) ** (x ** power)
    raise NotImplementedError


class Scheduler:
    def __init__(self, every_step=1, start_step=0, end_step=1000, start_timestep=0, end_timestep=1000, interpolate_begin=0.0,
                 interpolate_end=1.0, power=1.0, interpolate_type='constant', no_interpolate_return=0.0):
        self._every_step, self._start_step, self._end_step = every_step, start_step, end_step
        self._start_timestep, self._end_timestep = start_timestep, end_timestep
        self._interpolate_start, self._interpolate_end, self._power = interpolate_begin, interpolate_end, power
        self._interpolate_type, self._no_interpolate_return = interpolate_type, no_interpolate_return

    def __call__(self, step=None, timestep=None, **kwargs):
        """overlap_scheduler.py:89-107"""
        if step < self._start_step or step > self._end_step or step % self._every_step != 0 \
                or timestep < self._start_timestep or timestep > self._end_timestep:
            return self._no_interpolate_return
        t = 1 - (timestep / 1000)                          # increases from 0 to 1 while denoising
        return value_interpolation(t, self._interpolate_start, self._interpolate_end, self._power, self._interpolate_type)


class AverageDistance:
    algo = 0


class FrameDistance:
    algo = 1


class PixelDistance:
    algo = 2


class PerpendicularViewNormal:
    algo = 3


class CorrespondenceMap:
    """CSR form of the legacy ``{id_tuple: [([y, x], frame), ...]}`` dict, built on the device from (T,H,W,4) id maps
    (data_classes/correspondence_map.py:150-166: all-zero ids are skipped, entries in (frame, y, x) order)."""

    def __init__(self, ids: torch.Tensor):
        assert ids.dim() == 4 and ids.shape[-1] == 4 and ids.is_cuda
        T, H, W, _ = ids.shape
        self.num_frames, self.height, self.width = T, H, W
        flat = ids.reshape(-1, 4).to(torch.int64)
        valid = (flat != 0).any(dim=1)
        uniq, inv = torch.unique(flat, dim=0, return_inverse=True)           # sort-based grouping (index plumbing, once per call)
        zero_row = (uniq == 0).all(dim=1)
        remap = torch.cumsum((~zero_row).to(torch.int64), 0) - 1
        vert = torch.where(valid, remap[inv], torch.full_like(inv, -1))
        self.n_vertices = int((~zero_row).sum())
        self.pix_vert = vert.to(torch.int32).contiguous()
        pix = torch.nonzero(valid).reshape(-1)
        order = torch.argsort(vert[pix], stable=True)                        # stable: (frame,y,x) order inside a trace
        pix = pix[order]
        counts = torch.bincount(vert[pix], minlength=self.n_vertices)
        self.offsets = torch.cat([counts.new_zeros(1), torch.cumsum(counts, 0)]).to(torch.int32).contiguous()
        self.tr_f = (pix // (H * W)).to(torch.int32).contiguous()
        self.tr_y = ((pix // W) % H).to(torch.int32).contiguous()
        self.tr_x = (pix % W).to(torch.int32).contiguous()
        # the reference's dict order = order of first appearance in the (frame, y, x) scan (correspondence_map.py:150-166):
        # what the in-place update order of Overlap.__call__ follows (kernel radius > 0)
        first = torch.full((max(self.n_vertices, 1),), T * H * W, dtype=torch.int64, device=ids.device)
        first.scatter_reduce_(0, vert[pix], pix, reduce="amin")
        self.order = torch.argsort(first[:self.n_vertices], stable=True).to(torch.int32)
        self.max_len = int(counts.max()) if self.n_vertices else 1
        self._levels = {}

    def levels(self, radius):
        """-> (lvl_vert device int32: multi-pixel vertices grouped by level, lvl_off host int32 prefix offsets, n_levels) for the
        in-place order at this kernel radius (sr_legacy_levels; built once per radius)"""
        import ctypes as C
        import numpy as np
        if radius not in self._levels:
            off, f, y, x, order = (t.cpu().numpy() for t in (self.offsets, self.tr_f, self.tr_y, self.tr_x, self.order))
            lvl = np.empty(max(self.n_vertices, 1), np.int32)
            nl = C.c_int32(0)
            ptr = lambda a: a.ctypes.data_as(C.c_void_p)
            L.check(L.lib().sr_legacy_levels(ptr(off), ptr(f), ptr(y), ptr(x), ptr(order), self.n_vertices, self.num_frames, self.height,
                                             self.width, int(radius), ptr(lvl), C.byref(nl)))
            lvl = lvl[:self.n_vertices]
            active = np.nonzero(lvl >= 0)[0]
            by_level = active[np.argsort(lvl[active], kind="stable")].astype(np.int32)
            lvl_off = np.concatenate([[0], np.cumsum(np.bincount(lvl[active], minlength=nl.value))]).astype(np.int32)
            self._levels[radius] = (torch.from_numpy(by_level).to(self.pix_vert.device), lvl_off, int(nl.value))
        return self._levels[radius]

    @property
    def size(self):
        return (self.width, self.height)

    def __len__(self):
        return self.n_vertices


class Overlap:
    """frames at corr-map resolution (overlap.py:83-152)."""
    keep_nonzero = 0

    def __init__(self, alpha_scheduler, kernel_radius_scheduler, algorithm, verbose=True):
        self.alpha_scheduler, self.kernel_radius_scheduler, self.algorithm, self.verbose = alpha_scheduler, kernel_radius_scheduler, algorithm, verbose

    def _run(self, x, corr_map, alpha, radius, view_normal_map):
        T, B, C, h, w = x.shape
        assert B == 1, "the legacy path handles one latent per frame"
        xin = x.reshape(T, C, h, w).contiguous().float()
        if radius > 0:
            return self._run_in_place_order(xin, corr_map, alpha, radius, view_normal_map).reshape(T, 1, C, h, w)
        y = torch.empty_like(xin)
        vn = None
        if self.algorithm.algo == 3:
            if view_normal_map is None:
                raise TypeError("overlap() missing 1 required positional argument: 'view_normal_map'")
            vn = view_normal_map.reshape(T, corr_map.height, corr_map.width).contiguous().float().to(xin.device)
        L.check(L.lib().sr_legacy_overlap(O._p(xin), O._p(y), O._p(corr_map.pix_vert), O._p(corr_map.offsets), O._p(corr_map.tr_f),
                                          O._p(corr_map.tr_y), O._p(corr_map.tr_x), O._p(vn), T, C, h, w, corr_map.height, corr_map.width,
                                          float(alpha), int(radius), self.algorithm.algo, self.keep_nonzero, O.stream_ptr()))
        return y.reshape(T, 1, C, h, w)

    def _run_in_place_order(self, xin, corr_map, alpha, radius, view_normal_map):
        """kernel radius > 0: the reference's in-place update order (overlap.py:103-146), replayed level by level on a
        corr-map-resolution working tensor (sr_legacy_overlap_seq).  ResizeOverlap: nearest up, overlap, nearest down + where."""
        import ctypes as C
        T, Cc, h, w = xin.shape
        H, W = corr_map.height, corr_map.width
        lib, st = L.lib(), O.stream_ptr()
        vn = None
        if self.algorithm.algo == 3:
            if view_normal_map is None:
                raise TypeError("overlap() missing 1 required positional argument: 'view_normal_map'")
            vn = view_normal_map.reshape(T, H, W).contiguous().float().to(xin.device)
        if (h, w) == (H, W):
            U = xin.clone()
        else:
            U = torch.empty(T, Cc, H, W, dtype=torch.float32, device=xin.device)
            L.check(lib.sr_nearest_resize(O._p(xin), O._p(U), T * Cc, h, w, H, W, None, st))
        lvl_vert, lvl_off, n_levels = corr_map.levels(radius)
        if n_levels > 0:
            newval = torch.empty(int(corr_map.offsets[-1]) * Cc, dtype=torch.float32, device=xin.device)
            L.check(lib.sr_legacy_overlap_seq(O._p(U), O._p(newval), O._p(lvl_vert), lvl_off.ctypes.data_as(C.c_void_p), n_levels,
                                              corr_map.max_len, O._p(corr_map.offsets), O._p(corr_map.tr_f), O._p(corr_map.tr_y),
                                              O._p(corr_map.tr_x), O._p(vn), Cc, H, W, float(alpha), int(radius), self.algorithm.algo, st))
        if (h, w) == (H, W):
            return U
        y = torch.empty_like(xin)
        L.check(lib.sr_nearest_resize(O._p(U), O._p(y), T * Cc, H, W, h, w, O._p(xin) if self.keep_nonzero else None, st))
        return y

    def __call__(self, frame_seq, corr_map, step=None, timestep=None, view_normal_map=None, **kwargs):
        assert tuple(frame_seq[0].shape[2:]) == (corr_map.height, corr_map.width), "frame shape does not match corr_map shape"
        x = torch.stack(list(frame_seq), 0)
        return self._run(x, corr_map, self.alpha_scheduler(step, timestep), int(self.kernel_radius_scheduler(step, timestep)), view_normal_map)


class ResizeOverlap(Overlap):
    """latents are (conceptually) nearest-resized to the corr-map size, overlapped and resized back (overlap.py:180-222); the
    kernel does all three at latent resolution."""
    keep_nonzero = 1

    def __init__(self, alpha_scheduler, kernel_radius_scheduler, algorithm, verbose=True, interpolate_mode='nearest'):
        super().__init__(alpha_scheduler, kernel_radius_scheduler, algorithm, verbose)
        if interpolate_mode != 'nearest':
            raise NotImplementedError("only nearest interpolation is fused")
        self.interpolate_mode = interpolate_mode

    def __call__(self, frame_seq, corr_map, step=None, timestep=None, view_normal_map=None, **kwargs):
        alpha = self.alpha_scheduler(step, timestep)
        if alpha == 0:
            return frame_seq
        out = self._run(torch.stack(list(frame_seq), 0), corr_map, alpha, int(self.kernel_radius_scheduler(step, timestep)), view_normal_map)
        return [out[i] for i in range(out.shape[0])]
