"""Launch plans: a model forward lowered once to a flat array of C-ABI ops (include/sr_hip.h: sr_op) over
pre-allocated HBM buffers, then replayed natively (eagerly or as a hipGraph) every denoise step."""
import ctypes as C
import os

import torch

from . import _lib as L
from . import ops as O


def igemm_algorithmic_bytes(ar):
    """HBM bytes one igemm launch has to move once: source pixels (both concat sources), packed weights, output
    (+ residual read); the figure the measured FETCH_SIZE / WRITE_SIZE traffic of a launch is compared with"""
    es = 2 if ar.dtype == L.SR_F16 else 4
    st, up = max(ar.stride, 1), ar.upsample
    Ho, Wo = ((ar.up_h, ar.up_w) if ar.up_h else (2 * ar.H, 2 * ar.W)) if up else ((ar.H + st - 1) // st, (ar.W + st - 1) // st)
    M, C = ar.B * Ho * Wo, ar.C1 + ar.C2
    n_out = ar.N // 2 if ar.act == 2 else ar.N
    return (ar.B * ar.H * ar.W * C * es + ar.N * ar.KH * ar.KH * C * es + M * n_out * (4 if ar.out_f32 else es)
            + (M * n_out * es if ar.residual else 0))


class Plan:
    def __init__(self, op_list, keep, op_flops=None):
        self.op_flops = list(op_flops) if op_flops is not None else [0] * len(op_list)
        self.n = len(op_list)
        self.ops = (L.Op * max(self.n, 1))(*op_list)
        self._keep = keep                      # tensors referenced by raw pointers inside the ops
        self._graph = None
        self._graph_stream = None

    def run(self):
        L.check(L.lib().sr_plan_run(self.ops, self.n, O.stream_ptr()))

    def igemm_bytes(self):
        """sum of igemm_algorithmic_bytes over the plan's igemm ops"""
        return sum(igemm_algorithmic_bytes(self.ops[i].u.igemm) for i in range(self.n) if self.ops[i].kind == L.OP_IGEMM)

    def subset(self, kind):
        """-> Plan holding only the ops of one kind (same buffers), all on the main lane: used to time one kernel family
        in isolation"""
        idx = [i for i in range(self.n) if self.ops[i].kind == kind]
        sel = []
        for i in idx:
            op = L.Op()
            C.memmove(C.byref(op), C.byref(self.ops[i]), C.sizeof(L.Op))
            op.lane = 0
            sel.append(op)
        return Plan(sel, self._keep, [self.op_flops[i] for i in idx])

    def capture(self, stream):
        """capture into a hipGraph on `stream` (a torch.cuda.Stream, not the default one)"""
        ge = C.c_void_p()
        L.check(L.lib().sr_plan_capture(self.ops, self.n, C.c_void_p(stream.cuda_stream), C.byref(ge)))
        self._graph, self._graph_stream = ge, stream

    def launch(self):
        if self._graph is None:
            return self.run()
        L.check(L.lib().sr_graph_launch(self._graph, O.stream_ptr()))

    def __del__(self):
        try:
            if self._graph is not None:
                L.lib().sr_graph_destroy(self._graph)
        except Exception:
            pass


class PlanBuilder:
    """Collects ops + owns the activation buffers they point at."""

    def __init__(self, device, dtype):
        self.device, self.dtype = device, dtype
        self.ops = []
        self.op_flops = []
        self.keep = []
        self._gn_scratch = None
        self.flops = 0
        self._lane = 0
        # Side lane off by default: measured on the SD1.5 UNet step (skip convolutions and the injected frame's K/V
        # projections beside the main chain) it is neutral under a hipGraph (24.15 vs 24.00 ms) and costs 1 ms eagerly
        # (60 event record/wait pairs); SR_TWO_LANES=1 turns it on.
        self.two_lanes = os.environ.get("SR_TWO_LANES", "0") == "1"
        # every igemm warms the Infinity Cache with the packed weights of the NEXT igemm of the plan (sr_igemm_args.prefetch): one
        # UNet evaluation streams 1.7 GB of weights, so none of them survives in the 256 MB cache from one evaluation to the next
        # and each layer would fetch its own from HBM on its critical path.  Opt-in (SR_PREFETCH=1): measured -0.15...-0.27 ms per
        # UNet evaluation (1 %), but every touched sector is one more request at the fabric-side counters (FETCH_SIZE: +2.6 GB per
        # evaluation, 2.52x instead of 2.35x the algorithmic bytes), so the default keeps the traffic figure honest
        self.prefetch = os.environ.get("SR_PREFETCH", "0") == "1"
        self._prev_igemm = None

    # ---- side lane (include/sr_hip.h: SR_OP_FORK / SR_OP_JOIN) -----------------------------------------
    def fork(self):
        if self.two_lanes:
            self._emit(L.OP_FORK, None, None)

    def join(self):
        if self.two_lanes:
            self._emit(L.OP_JOIN, None, None)

    def side(self):
        """context manager: ops emitted inside run on the side lane (between fork() and join())"""
        pb = self

        class _Side:
            def __enter__(self_inner):
                pb._lane = 1 if pb.two_lanes else 0

            def __exit__(self_inner, *a):
                pb._lane = 0
        return _Side()

    def buf(self, *shape, dtype=None, zero=False):
        t = (torch.zeros if zero else torch.empty)(*shape, dtype=dtype or self.dtype, device=self.device)
        self.keep.append(t)
        return t

    def hold(self, *ts):
        self.keep.extend(t for t in ts if t is not None)

    def _emit(self, kind, field, args):
        op = L.Op()
        op.kind = kind
        op.lane = self._lane if field is not None else 0
        if field is not None:
            setattr(op.u, field, args)
        self.ops.append(op)
        self.op_flops.append(0)

    # ---- ops ------------------------------------------------------------------------------------------
    def igemm(self, a, w, out, B, H, W, C1, N, **kw):
        self.hold(a, w, out, kw.get("a2"), kw.get("bias"), kw.get("rowvec"), kw.get("residual"), kw.get("row_stats"), kw.get("colsum"))
        if (kw.get("KH", 1) == 1 and kw.get("stride", 1) == 1 and not kw.get("upsample", 0) and kw.get("rowvec") is None
                and not kw.get("transpose_out", 0)):
            B, H, W = B * H * W, 1, 1          # a 1x1 convolution IS the linear layer over all pixels: no per-row (b, y, x) split
        ar = O.igemm_args(a, w, out, B, H, W, C1, N, dtype=self.dtype, **kw)
        if self._lane == 1:
            ar.split = -1                      # the split-K workspace belongs to the main lane
        if O.autotune_enabled():
            O.tune_igemm(ar, allow_split=self._lane == 0)
        self._emit(L.OP_IGEMM, "igemm", ar)
        if self.prefetch and w.is_cuda:
            prev, self._prev_igemm = self._prev_igemm, self.ops[-1]
            if prev is not None and prev.u.igemm.w != w.data_ptr():
                prev.u.igemm.prefetch = w.data_ptr()
                prev.u.igemm.prefetch_bytes = min(w.numel() * w.element_size(), 48 << 20)
        KH, st, up = kw.get("KH", 1), kw.get("stride", 1), kw.get("upsample", 0)
        Ho, Wo = (tuple(kw["up_hw"]) if kw.get("up_hw") else (2 * H, 2 * W)) if up else ((H + st - 1) // st, (W + st - 1) // st)
        f = 2 * B * Ho * Wo * N * KH * KH * (C1 + kw.get("C2", 0))
        self.flops += f
        self.op_flops[-1] = f

    def igemm_group(self, calls):
        """calls: [(positional args, kwargs)] of igemm ops that do NOT depend on each other.  They are emitted back to back (so
        every per-op tool still sees plain igemm ops); when ops.tune_group finds one tile under which a single grouped launch beats
        the launches one after another, the first op carries ``group = n`` and sr_plan_run hands them to sr_igemm_group."""
        first = len(self.ops)
        for args, kw in calls:
            self.igemm(*args, **kw)
        members = [op.u.igemm for op in self.ops[first:]]
        if self._lane == 0 and len(members) == len(calls):
            O.tune_group(members)

    def groupnorm(self, x, gamma, beta, y, B, HW, C1, x2=None, C2=0, eps=1e-5, silu=False):
        need = L.lib().sr_groupnorm_scratch_floats(B, HW)
        if self._gn_scratch is None or self._gn_scratch.numel() < need:
            self._gn_scratch = self.buf(need, dtype=torch.float32)
        self.hold(x, x2, gamma, beta, y)
        self._emit(L.OP_GROUPNORM, "gn", O.groupnorm_args(x, gamma, beta, y, B, HW, C1, self._gn_scratch, x2, C2, 32, eps, silu))

    def layernorm(self, x, gamma, beta, y, rows, Cc, eps=1e-5):
        self.hold(x, gamma, beta, y)
        a = L._Ln()
        a.x, a.gamma, a.beta, a.y = O._p(x), O._p(gamma), O._p(beta), O._p(y)
        a.rows, a.C, a.dtype, a.eps = rows, Cc, O.DT[self.dtype], eps
        self._emit(L.OP_LAYERNORM, "ln", a)

    def layernorm_gather(self, x, sel, nsel, frame_rows, n_frames, gamma, beta, y, Cc, err_flag=None, eps=1e-5):
        """y[j * frame_rows + r] = LayerNorm(x[sel[j], r]) (sr_layernorm_gather): the injected frame picked and normalised at once"""
        self.hold(x, sel, gamma, beta, y, err_flag)
        a = L._Ln()
        a.x, a.gamma, a.beta, a.y = O._p(x), O._p(gamma), O._p(beta), O._p(y)
        a.rows, a.C, a.dtype, a.eps = nsel * frame_rows, Cc, O.DT[self.dtype], eps
        a.sel, a.err_flag, a.frame_rows, a.n_frames = O._p(sel), O._p(err_flag), frame_rows, n_frames
        self._emit(L.OP_LAYERNORM_GATHER, "ln", a)

    def row_stats(self, x, stats, rows, Cc, eps=1e-5):
        self.hold(x, stats)
        a = L._Ln()
        a.x, a.gamma, a.beta, a.y = O._p(x), None, None, O._p(stats)
        a.rows, a.C, a.dtype, a.eps = rows, Cc, O.DT[x.dtype], eps
        self._emit(L.OP_ROW_STATS, "ln", a)

    def attention(self, q, k, vt, o, B, Bk, Tq, Tk, heads, d, ldt):
        self.hold(q, k, vt, o)
        self._emit(L.OP_ATTENTION, "attn", O.attention_args(q, k, vt, o, B, Bk, Tq, Tk, heads, d, ldt))
        self.flops += 4 * B * heads * Tq * Tk * d
        self.op_flops[-1] = 4 * B * heads * Tq * Tk * d

    def nchw_to_nhwc(self, x, y, B, Cc, HW, Cpad, scale=1.0):
        self.hold(x, y)
        a = L._Cvt()
        a.x, a.y, a.per_batch_scale = O._p(x), O._p(y), None
        a.B, a.C, a.HW, a.Cpad, a.dtype, a.ldc, a.scale = B, Cc, HW, Cpad, O.DT[y.dtype], Cpad, scale
        self._emit(L.OP_NCHW_TO_NHWC, "cvt", a)

    def nhwc_to_nchw(self, x, y, B, Cc, HW, ldc):
        self.hold(x, y)
        a = L._Cvt()
        a.x, a.y, a.per_batch_scale = O._p(x), O._p(y), None
        a.B, a.C, a.HW, a.Cpad, a.dtype, a.ldc, a.scale = B, Cc, HW, ldc, O.DT[x.dtype], ldc, 1.0
        self._emit(L.OP_NHWC_TO_NCHW, "cvt", a)

    def timestep_embedding(self, t, y, B, dim):
        self.hold(t, y)
        a = L._Temb()
        a.t, a.y, a.B, a.dim, a.dtype = O._p(t), O._p(y), B, dim, O.DT[y.dtype]
        self._emit(L.OP_TIMESTEP_EMBED, "temb", a)

    def silu(self, x, y):
        self.hold(x, y)
        a = L._Ew()
        a.x, a.y, a.n, a.dtype = O._p(x), O._p(y), x.numel(), O.DT[x.dtype]
        self._emit(L.OP_SILU, "ew", a)

    def softmax_rows(self, x, rows, cols):
        self.hold(x)
        a = L._Ew()
        a.x, a.y, a.n, a.dtype, a.rows, a.cols = O._p(x), O._p(x), x.numel(), O.DT[x.dtype], rows, cols
        self._emit(L.OP_SOFTMAX_ROWS, "ew", a)

    def add(self, a, b, y, s=1.0):
        """y = a + s*b (same shape / dtype)"""
        self.hold(a, b, y)
        ar = L._Add()
        ar.a, ar.b, ar.y, ar.n, ar.s, ar.dtype = O._p(a), O._p(b), O._p(y), a.numel(), s, O.DT[a.dtype]
        self._emit(L.OP_ADD_SCALED, "add", ar)

    def gather_rows(self, x, sel, y, nsel, row_bytes, n_rows, err_flag=None):
        """y[j] = x[sel[j]] (rows of row_bytes bytes, x holds n_rows of them); an out-of-range device index zero-fills its row and
        sets err_flag (device int32) instead of reading memory"""
        self.hold(x, sel, y, err_flag)
        a = L._Gather()
        a.x, a.y, a.sel, a.row_bytes, a.nsel, a.n_rows, a.err_flag = O._p(x), O._p(y), O._p(sel), row_bytes, nsel, n_rows, O._p(err_flag)
        self._emit(L.OP_GATHER_ROWS, "gather", a)

    def take(self):
        """-> Plan of the ops emitted so far (the builder keeps collecting into a fresh list)"""
        p = Plan(self.ops, list(self.keep), self.op_flops)
        self.ops = []
        self.op_flops = []
        self._prev_igemm = None
        return p
