// Launch-plan executor: the Python host lowers a model (UNet / VAE decoder / ControlNet encoder) to a flat
// array of sr_op once per (weights, batch, resolution); every denoise step then replays it from native code —
// either eagerly (sr_plan_run) or as a captured hipGraph (sr_plan_capture / sr_graph_launch), so the per-step
// hot loop has no Python, no allocation and no host synchronisation in it.
#include "sr_common.h"

static int run_op(const sr_op& op, void* stream) {
  switch (op.kind) {
    case SR_OP_IGEMM: return sr_igemm(&op.u.igemm, stream);
    case SR_OP_GROUPNORM: return sr_groupnorm(&op.u.gn, stream);
    case SR_OP_ATTENTION: return sr_attention(&op.u.attn, stream);
    case SR_OP_LAYERNORM:
      return sr_layernorm(op.u.ln.x, op.u.ln.gamma, op.u.ln.beta, op.u.ln.y, op.u.ln.rows, op.u.ln.C, op.u.ln.eps, op.u.ln.dtype, stream);
    case SR_OP_LAYERNORM_GATHER:
      if (op.u.ln.frame_rows < 1 || op.u.ln.rows % op.u.ln.frame_rows) { sr_set_error("sr_plan_run: layernorm_gather: rows %% frame_rows"); return SR_ERR_INVALID; }
      return sr_layernorm_gather(op.u.ln.x, op.u.ln.sel, op.u.ln.rows / op.u.ln.frame_rows, op.u.ln.frame_rows, op.u.ln.n_frames, op.u.ln.err_flag,
                                 op.u.ln.gamma, op.u.ln.beta, op.u.ln.y, op.u.ln.C, op.u.ln.eps, op.u.ln.dtype, stream);
    case SR_OP_ROW_STATS:
      return sr_row_stats(op.u.ln.x, (float*)op.u.ln.y, op.u.ln.rows, op.u.ln.C, op.u.ln.eps, op.u.ln.dtype, stream);
    case SR_OP_NCHW_TO_NHWC:
      return sr_nchw_to_nhwc((const float*)op.u.cvt.x, op.u.cvt.y, op.u.cvt.B, op.u.cvt.C, op.u.cvt.HW, op.u.cvt.Cpad, op.u.cvt.scale,
                             op.u.cvt.per_batch_scale, op.u.cvt.dtype, stream);
    case SR_OP_NHWC_TO_NCHW:
      return sr_nhwc_to_nchw(op.u.cvt.x, (float*)op.u.cvt.y, op.u.cvt.B, op.u.cvt.C, op.u.cvt.HW, op.u.cvt.ldc, op.u.cvt.dtype, stream);
    case SR_OP_TIMESTEP_EMBED:
      return sr_timestep_embedding(op.u.temb.t, op.u.temb.y, op.u.temb.B, op.u.temb.dim, op.u.temb.dtype, stream);
    case SR_OP_SILU: return sr_silu(op.u.ew.x, op.u.ew.y, op.u.ew.n, op.u.ew.dtype, stream);
    case SR_OP_SOFTMAX_ROWS: return sr_softmax_rows(op.u.ew.y, op.u.ew.rows, op.u.ew.cols, op.u.ew.dtype, stream);
    case SR_OP_ADD_SCALED: return sr_add_scaled(op.u.add.a, op.u.add.b, op.u.add.y, op.u.add.n, op.u.add.s, op.u.add.dtype, stream);
    case SR_OP_GATHER_ROWS: return sr_gather_rows(op.u.gather.x, op.u.gather.sel, op.u.gather.y, op.u.gather.nsel, op.u.gather.n_rows, op.u.gather.row_bytes,
                                                   op.u.gather.err_flag, stream);
    default: sr_set_error("sr_plan_run: unknown op kind %d", op.kind); return SR_ERR_INVALID;
  }
}

// side lane: one extra stream + two events (fork / join) PER MAIN STREAM (calls in flight on several streams each get their
// own, created under a mutex on the device that is current when the main stream first forks).  Re-recording an event is
// safe here: a wait captures the record that precedes it in program order, eagerly and under stream capture alike.
// The map key is (device, stream handle): a hipStream_t value can come back for a NEW stream once the old one was destroyed,
// possibly on another device; keyed like this a recycled handle either finds a side stream of its own device (still a valid
// partner: the lane is nothing but "some other stream of this device" plus two events) or gets a fresh one.
#include <map>
#include <mutex>
#include <utility>
struct sr_side { hipStream_t s = nullptr; hipEvent_t fork = nullptr, join = nullptr; };
static std::mutex g_side_mu;
static std::map<std::pair<int, void*>, sr_side> g_sides;
static int side_get(void* main_stream, sr_side* out) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) SR_FAIL(SR_ERR_LAUNCH, "side lane: hipGetDevice");
  std::lock_guard<std::mutex> lk(g_side_mu);
  const auto key = std::make_pair(dev, main_stream);
  auto it = g_sides.find(key);
  if (it == g_sides.end()) {
    sr_side sd;
    if (hipStreamCreateWithFlags(&sd.s, hipStreamNonBlocking) != hipSuccess) SR_FAIL(SR_ERR_LAUNCH, "side stream");
    if (hipEventCreateWithFlags(&sd.fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&sd.join, hipEventDisableTiming) != hipSuccess)
      SR_FAIL(SR_ERR_LAUNCH, "side events");
    it = g_sides.emplace(key, sd).first;
  }
  *out = it->second;
  return SR_OK;
}

extern "C" int sr_plan_run(const sr_op* ops, int32_t n, void* stream) {
  if (!ops || n < 0) SR_FAIL(SR_ERR_INVALID, "sr_plan_run: bad args");
  bool side_open = false;                                    // side-lane work issued since the last JOIN
  sr_side sd;
  for (int i = 0; i < n; ++i) {
    if (ops[i].kind == SR_OP_FORK || ops[i].kind == SR_OP_JOIN || ops[i].lane == 1) {
      if (!sd.s && side_get(stream, &sd) != SR_OK) return SR_ERR_LAUNCH;
    }
    if (ops[i].kind == SR_OP_FORK) {
      if (hipEventRecord(sd.fork, sr_stream(stream)) != hipSuccess || hipStreamWaitEvent(sd.s, sd.fork, 0) != hipSuccess)
        SR_FAIL(SR_ERR_LAUNCH, "sr_plan_run: fork at op %d", i);
      side_open = true;
      continue;
    }
    if (ops[i].kind == SR_OP_JOIN) {
      if (side_open && (hipEventRecord(sd.join, sd.s) != hipSuccess || hipStreamWaitEvent(sr_stream(stream), sd.join, 0) != hipSuccess))
        SR_FAIL(SR_ERR_LAUNCH, "sr_plan_run: join at op %d", i);
      side_open = false;
      continue;
    }
    if (ops[i].lane == 1 && !side_open) SR_FAIL(SR_ERR_INVALID, "sr_plan_run: side-lane op %d outside FORK..JOIN", i);
    int rc;
    const int grp = ops[i].kind == SR_OP_IGEMM ? ops[i].u.igemm.group : 0;
    if (grp > 1) {
      // sr_igemm_args.group: this op and the next grp-1 are independent igemm ops -> one grouped launch (sr_igemm_group)
      if (grp > SR_IGEMM_GROUP_MAX || i + grp > n) SR_FAIL(SR_ERR_INVALID, "sr_plan_run: op %d: group of %d", i, grp);
      const sr_igemm_args* ptr[SR_IGEMM_GROUP_MAX];
      for (int j = 0; j < grp; ++j) {
        if (ops[i + j].kind != SR_OP_IGEMM || ops[i + j].lane != ops[i].lane) SR_FAIL(SR_ERR_INVALID, "sr_plan_run: op %d: group member %d is not an igemm op of the same lane", i, j);
        ptr[j] = &ops[i + j].u.igemm;
      }
      rc = sr_igemm_group(ptr, grp, ops[i].lane == 1 ? (void*)sd.s : stream);
      if (rc == SR_OK) { i += grp - 1; continue; }
    } else
      rc = run_op(ops[i], ops[i].lane == 1 ? (void*)sd.s : stream);
    if (rc != SR_OK) {
      char buf[400];
      snprintf(buf, sizeof(buf), "%s", sr_last_error());
      sr_set_error("plan op %d (kind %d): %s", i, ops[i].kind, buf);
      return rc;
    }
  }
  if (side_open) {                                           // a plan must not end with the side lane detached
    if (hipEventRecord(sd.join, sd.s) != hipSuccess || hipStreamWaitEvent(sr_stream(stream), sd.join, 0) != hipSuccess)
      SR_FAIL(SR_ERR_LAUNCH, "sr_plan_run: final join");
  }
  return SR_OK;
}

extern "C" int sr_plan_capture(const sr_op* ops, int32_t n, void* stream, void** graph_exec) {
  if (!ops || !graph_exec || !stream) SR_FAIL(SR_ERR_INVALID, "sr_plan_capture: needs a non-default stream");
  hipStream_t st = sr_stream(stream);
  // warm every kernel once outside the capture (function attributes are set lazily on first launch)
  int rc = sr_plan_run(ops, n, stream);
  if (rc != SR_OK) return rc;
  if (hipStreamSynchronize(st) != hipSuccess) SR_FAIL(SR_ERR_LAUNCH, "sr_plan_capture: warm-up failed: %s", hipGetErrorString(hipGetLastError()));
  if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) SR_FAIL(SR_ERR_LAUNCH, "sr_plan_capture: begin capture failed");
  rc = sr_plan_run(ops, n, stream);
  hipGraph_t graph = nullptr;
  const hipError_t e = hipStreamEndCapture(st, &graph);
  if (rc != SR_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
  if (e != hipSuccess || !graph) SR_FAIL(SR_ERR_LAUNCH, "sr_plan_capture: end capture: %s", hipGetErrorString(e));
  hipGraphExec_t ex = nullptr;
  const hipError_t e2 = hipGraphInstantiate(&ex, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e2 != hipSuccess) SR_FAIL(SR_ERR_LAUNCH, "sr_plan_capture: instantiate: %s", hipGetErrorString(e2));
  *graph_exec = (void*)ex;
  return SR_OK;
}

extern "C" int sr_graph_launch(void* graph_exec, void* stream) {
  if (!graph_exec) SR_FAIL(SR_ERR_INVALID, "sr_graph_launch: null");
  const hipError_t e = hipGraphLaunch((hipGraphExec_t)graph_exec, sr_stream(stream));
  if (e != hipSuccess) SR_FAIL(SR_ERR_LAUNCH, "sr_graph_launch: %s", hipGetErrorString(e));
  return SR_OK;
}

extern "C" int sr_graph_destroy(void* graph_exec) {
  if (graph_exec) (void)hipGraphExecDestroy((hipGraphExec_t)graph_exec);
  return SR_OK;
}
