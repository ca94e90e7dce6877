// Operator-level entry points of the C ABI (include/sr_hip.h: sr_model_*, sr_unet_forward, sr_vae_decode).
//
// A *model bundle* is a launch plan made relocatable: the flat sr_op arrays a Python host lowered a UNet / VAE to (unet.py,
// vae.py, plan.py -- once per (weights, batch, resolution), tiles already chosen by the tuner), every device pointer in them
// replaced by (tensor id, byte offset), plus the tensor table (size, initial contents: packed weights, norm parameters, zero
// pages; activations are zero-filled) and the named input / output windows.  sr_model_load allocates the tensors on the current
// device, uploads their contents and patches the pointers back in; from then on a host written in any language runs
//     sr_unet_forward(m, x, t, ctx, out, stream)          = UNetModel.forward       (openaimodel.py:841-946)
//     sr_vae_decode(m, z, img, stream)                    = VAE.decode              (comfy/sd.py:329-346)
// through this library alone -- no Python, no torch in the process (tests/test_gpu_model_bundle.py does exactly that from a
// child process that imports neither).  The file is produced by stable-renderer_amd/bundle.py:export_bundle.
//
// File layout (little endian):  "SRMODEL1" | u32 version=1 | u32 n_tensors | u32 n_io | u32 n_plans
//   tensors[n_tensors] : u64 nbytes, u64 data_offset (0 = zero-filled, else absolute file offset of nbytes bytes)
//   io[n_io]           : char name[32], u32 tensor, u32 is_output, u64 offset, u64 nbytes
//   plans[n_plans]     : char name[16], u32 n_ops, u32 n_reloc, u32 sizeof_op, u32 pad,
//                        ops[n_ops * sizeof_op], reloc[n_reloc] = { u32 op, u32 field_offset, u32 tensor, u32 pad, u64 offset }
#include "sr_common.h"
#include <stddef.h>
#include <stdio.h>
#include <string.h>
#include <new>
#include <string>
#include <vector>

namespace {
struct Io { char name[32]; uint32_t tensor, is_output; uint64_t offset, nbytes; };
struct Reloc { uint32_t op, field, tensor, pad; uint64_t offset; };
struct PlanRec { char name[16]; std::vector<sr_op> ops; };
}  // namespace

struct sr_model {
  std::vector<void*> tensors;
  std::vector<uint64_t> sizes;
  std::vector<Io> io;
  std::vector<PlanRec> plans;
  int device = -1;
  bool prologue_ran = false;                                 // sr_unet_forward(ctx = NULL) needs a projected prompt
};

static bool rd(FILE* f, void* p, size_t n) { return fread(p, 1, n, f) == n; }

// byte offsets (inside sr_op) of every device-pointer field of an op of this kind
static std::vector<size_t> pointer_fields(int kind) {
#define F(member, field) (offsetof(sr_op, u) + offsetof(decltype(sr_op::u), member) + offsetof(decltype(decltype(sr_op::u)::member), field))
  switch (kind) {
    case SR_OP_IGEMM: return {F(igemm, a), F(igemm, a2), F(igemm, w), F(igemm, bias), F(igemm, rowvec), F(igemm, residual), F(igemm, out),
                              F(igemm, zero_page), F(igemm, workspace), F(igemm, row_stats), F(igemm, colsum), F(igemm, prefetch), F(igemm, split_counters)};
    case SR_OP_GROUPNORM: return {F(gn, x), F(gn, x2), F(gn, gamma), F(gn, beta), F(gn, y), F(gn, partials)};
    case SR_OP_ATTENTION: return {F(attn, q), F(attn, k), F(attn, vt), F(attn, o)};
    case SR_OP_LAYERNORM: case SR_OP_ROW_STATS: return {F(ln, x), F(ln, gamma), F(ln, beta), F(ln, y)};
    case SR_OP_LAYERNORM_GATHER: return {F(ln, x), F(ln, gamma), F(ln, beta), F(ln, y), F(ln, sel), F(ln, err_flag)};
    case SR_OP_NCHW_TO_NHWC: case SR_OP_NHWC_TO_NCHW: return {F(cvt, x), F(cvt, y), F(cvt, per_batch_scale)};
    case SR_OP_TIMESTEP_EMBED: return {F(temb, t), F(temb, y)};
    case SR_OP_SILU: case SR_OP_SOFTMAX_ROWS: return {F(ew, x), F(ew, y)};
    case SR_OP_GATHER_ROWS: return {F(gather, x), F(gather, y), F(gather, sel), F(gather, err_flag)};
    case SR_OP_ADD_SCALED: return {F(add, a), F(add, b), F(add, y)};
    default: return {};
  }
#undef F
}

static void free_model(sr_model* m) {
  if (!m) return;
  for (void* p : m->tensors) if (p) (void)hipFree(p);
  delete m;
}

extern "C" int sr_model_load(const char* path, sr_model** out) {
  if (!path || !out) SR_FAIL(SR_ERR_INVALID, "sr_model_load: null argument");
  *out = nullptr;
  FILE* f = fopen(path, "rb");
  if (!f) SR_FAIL(SR_ERR_INVALID, "sr_model_load: cannot open %s", path);
  char magic[8];
  uint32_t hdr[4];
  if (!rd(f, magic, 8) || memcmp(magic, "SRMODEL1", 8) != 0 || !rd(f, hdr, sizeof(hdr)) || hdr[0] != 1) {
    fclose(f);
    SR_FAIL(SR_ERR_INVALID, "sr_model_load: %s is not a version-1 model bundle", path);
  }
  const uint32_t nt = hdr[1], nio = hdr[2], np = hdr[3];
  // the file is not trusted: every count is bounded by what a file of this size can hold before anything is allocated
  long fsize = 0;
  if (fseek(f, 0, SEEK_END) != 0 || (fsize = ftell(f)) < 24 || fseek(f, 24, SEEK_SET) != 0 ||
      (uint64_t)nt * 16 + (uint64_t)nio * sizeof(Io) + (uint64_t)np * 32 > (uint64_t)fsize) {
    fclose(f);
    SR_FAIL(SR_ERR_INVALID, "sr_model_load: %s: header counts (%u tensors, %u io, %u plans) do not fit the file", path, nt, nio, np);
  }
  sr_model* m = nullptr;
  try {
  m = new sr_model();
  (void)hipGetDevice(&m->device);
  std::vector<uint64_t> data_off(nt);
  m->sizes.resize(nt);
  m->tensors.assign(nt, nullptr);
  bool ok = true;
  for (uint32_t i = 0; i < nt && ok; ++i) { uint64_t v[2]; ok = rd(f, v, sizeof(v)); m->sizes[i] = v[0]; data_off[i] = v[1]; }
  m->io.resize(nio);
  for (uint32_t i = 0; i < nio && ok; ++i) ok = rd(f, &m->io[i], sizeof(Io));
  std::vector<std::vector<Reloc>> relocs(np);
  m->plans.resize(np);
  for (uint32_t i = 0; i < np && ok; ++i) {
    uint32_t ph[4];
    ok = rd(f, m->plans[i].name, 16) && rd(f, ph, sizeof(ph));
    if (ok && ph[2] != sizeof(sr_op)) {
      fclose(f); free_model(m);
      SR_FAIL(SR_ERR_INVALID, "sr_model_load: bundle built for sizeof(sr_op) = %u, this library has %zu (re-export it)", ph[2], sizeof(sr_op));
    }
    if (!ok) break;
    if ((uint64_t)ph[0] * sizeof(sr_op) + (uint64_t)ph[1] * sizeof(Reloc) > (uint64_t)fsize) { ok = false; break; }
    m->plans[i].ops.resize(ph[0]);
    relocs[i].resize(ph[1]);
    ok = (ph[0] == 0 || rd(f, m->plans[i].ops.data(), (size_t)ph[0] * sizeof(sr_op))) && (ph[1] == 0 || rd(f, relocs[i].data(), (size_t)ph[1] * sizeof(Reloc)));
  }
  if (!ok) { fclose(f); free_model(m); SR_FAIL(SR_ERR_INVALID, "sr_model_load: %s is truncated", path); }
  // ---- tensors: allocate, zero or upload
  std::vector<char> stage;
  for (uint32_t i = 0; i < nt; ++i) {
    const uint64_t nb = m->sizes[i] ? m->sizes[i] : 16;
    if (hipMalloc(&m->tensors[i], nb) != hipSuccess) { fclose(f); free_model(m); SR_FAIL(SR_ERR_LAUNCH, "sr_model_load: hipMalloc of %llu bytes failed", (unsigned long long)nb); }
    if (data_off[i] == 0) {
      if (hipMemset(m->tensors[i], 0, nb) != hipSuccess) { fclose(f); free_model(m); SR_FAIL(SR_ERR_LAUNCH, "sr_model_load: hipMemset failed"); }
    } else {
      if (data_off[i] > (uint64_t)fsize || m->sizes[i] > (uint64_t)fsize - data_off[i]) {
        fclose(f); free_model(m);
        SR_FAIL(SR_ERR_INVALID, "sr_model_load: tensor %u of %s lies outside the file", i, path);
      }
      stage.resize(m->sizes[i]);
      if (fseek(f, (long)data_off[i], SEEK_SET) != 0 || !rd(f, stage.data(), m->sizes[i]) ||
          hipMemcpy(m->tensors[i], stage.data(), m->sizes[i], hipMemcpyHostToDevice) != hipSuccess) {
        fclose(f); free_model(m);
        SR_FAIL(SR_ERR_INVALID, "sr_model_load: cannot read / upload tensor %u of %s", i, path);
      }
    }
  }
  fclose(f);
  f = nullptr;
  // ---- relocations: (tensor, offset) -> device pointer, written into the pointer field of the op.  A relocation must name a
  // pointer field of that op's kind (pointer_fields: the table of them, from the struct definitions), and every pointer field
  // that NO relocation names must be NULL in the file: no address from the file ever reaches a kernel.
  for (uint32_t i = 0; i < np; ++i) {
    std::vector<std::vector<bool>> hit(m->plans[i].ops.size());
    for (const Reloc& r : relocs[i]) {
      if (r.op >= m->plans[i].ops.size() || r.tensor >= nt || r.field + sizeof(void*) > sizeof(sr_op) || r.offset > m->sizes[r.tensor]) {
        free_model(m);
        SR_FAIL(SR_ERR_INVALID, "sr_model_load: bad relocation in plan %u", i);
      }
      const std::vector<size_t> pf = pointer_fields(m->plans[i].ops[r.op].kind);
      size_t which = pf.size();
      for (size_t k = 0; k < pf.size(); ++k) if (pf[k] == r.field) which = k;
      if (which == pf.size()) { free_model(m); SR_FAIL(SR_ERR_INVALID, "sr_model_load: plan %u op %u: relocation of a non-pointer field", i, r.op); }
      if (hit[r.op].empty()) hit[r.op].assign(pf.size(), false);
      hit[r.op][which] = true;
      char* field = (char*)&m->plans[i].ops[r.op] + r.field;
      void* ptr = (char*)m->tensors[r.tensor] + r.offset;
      memcpy(field, &ptr, sizeof(void*));
    }
    for (size_t o = 0; o < m->plans[i].ops.size(); ++o) {
      const std::vector<size_t> pf = pointer_fields(m->plans[i].ops[o].kind);
      for (size_t k = 0; k < pf.size(); ++k) {
        if (!hit[o].empty() && hit[o][k]) continue;
        void* v = nullptr;
        memcpy(&v, (const char*)&m->plans[i].ops[o] + pf[k], sizeof(void*));
        if (v) { free_model(m); SR_FAIL(SR_ERR_INVALID, "sr_model_load: plan %u op %zu carries a raw address in an unrelocated pointer field", i, o); }
      }
    }
  }
  for (const Io& e : m->io)
    if (e.tensor >= nt || e.offset > m->sizes[e.tensor] || e.nbytes > m->sizes[e.tensor] - e.offset) { free_model(m); SR_FAIL(SR_ERR_INVALID, "sr_model_load: bad io window"); }
  } catch (const std::bad_alloc&) {                          // (never through the extern "C" frame)
    if (f) fclose(f);
    free_model(m);
    SR_FAIL(SR_ERR_INVALID, "sr_model_load: %s: out of host memory for its tables", path);
  }
  *out = m;
  return SR_OK;
}

// the handle belongs to the device it was loaded on: its tensors live there
static int on_device(const sr_model* m, const char* who) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev != m->device)
    SR_FAIL(SR_ERR_INVALID, "%s: the model was loaded on device %d, the current device is %d", who, m->device, dev);
  return SR_OK;
}

extern "C" int sr_model_free(sr_model* m) {
  free_model(m);
  return SR_OK;
}

static const Io* find_io(const sr_model* m, const char* name) {
  for (const Io& e : m->io)
    if (strncmp(e.name, name, sizeof(e.name)) == 0) return &e;
  return nullptr;
}

extern "C" int sr_model_io(sr_model* m, const char* name, void** dev_ptr, int64_t* nbytes) {
  if (!m || !name) SR_FAIL(SR_ERR_INVALID, "sr_model_io: null argument");
  const Io* e = find_io(m, name);
  if (!e) SR_FAIL(SR_ERR_INVALID, "sr_model_io: the bundle has no input / output named '%s'", name);
  if (dev_ptr) *dev_ptr = (char*)m->tensors[e->tensor] + e->offset;
  if (nbytes) *nbytes = (int64_t)e->nbytes;
  return SR_OK;
}

extern "C" int sr_model_run(sr_model* m, const char* plan, void* stream) {
  if (!m || !plan) SR_FAIL(SR_ERR_INVALID, "sr_model_run: null argument");
  if (on_device(m, "sr_model_run") != SR_OK) return SR_ERR_INVALID;
  for (const PlanRec& p : m->plans)
    if (strncmp(p.name, plan, sizeof(p.name)) == 0) {
      const int rc = sr_plan_run(p.ops.data(), (int32_t)p.ops.size(), stream);
      if (rc == SR_OK && strncmp(plan, "prologue", 9) == 0) m->prologue_ran = true;
      return rc;
    }
  SR_FAIL(SR_ERR_INVALID, "sr_model_run: the bundle has no plan named '%s'", plan);
}

// src / dst may be host or device memory (hipMemcpyDefault resolves it): a host without any HIP binding feeds host buffers
static int io_copy(sr_model* m, const char* name, const void* src, void* dst, bool write, void* stream, const char* who) {
  const Io* e = find_io(m, name);
  if (!e) SR_FAIL(SR_ERR_INVALID, "%s: the bundle has no input / output named '%s'", who, name);
  if (on_device(m, who) != SR_OK) return SR_ERR_INVALID;
  void* dev = (char*)m->tensors[e->tensor] + e->offset;
  const hipError_t rc = write ? hipMemcpyAsync(dev, src, e->nbytes, hipMemcpyDefault, sr_stream(stream))
                              : hipMemcpyAsync(dst, dev, e->nbytes, hipMemcpyDefault, sr_stream(stream));
  if (rc != hipSuccess) SR_FAIL(SR_ERR_LAUNCH, "%s: copy of '%s' failed: %s", who, name, hipGetErrorString(rc));
  return SR_OK;
}

extern "C" int sr_model_write(sr_model* m, const char* name, const void* src, void* stream) {
  if (!m || !name || !src) SR_FAIL(SR_ERR_INVALID, "sr_model_write: null argument");
  return io_copy(m, name, src, nullptr, true, stream, "sr_model_write");
}

extern "C" int sr_model_read(sr_model* m, const char* name, void* dst, void* stream) {
  if (!m || !name || !dst) SR_FAIL(SR_ERR_INVALID, "sr_model_read: null argument");
  const int rc = io_copy(m, name, nullptr, dst, false, stream, "sr_model_read");
  if (rc != SR_OK) return rc;
  if (hipStreamSynchronize(sr_stream(stream)) != hipSuccess) SR_FAIL(SR_ERR_LAUNCH, "sr_model_read: stream synchronisation failed");
  return SR_OK;
}

extern "C" int sr_unet_forward(sr_model* m, const float* x, const float* t, const void* ctx, float* out, void* stream) {
  if (!m || !x || !t || !out) SR_FAIL(SR_ERR_INVALID, "sr_unet_forward: null argument");
  int rc;
  if (ctx) {                                                 // a new prompt: cross-attention K / V are projected once (prologue plan)
    if ((rc = io_copy(m, "ctx", ctx, nullptr, true, stream, "sr_unet_forward")) != SR_OK) return rc;
    if ((rc = sr_model_run(m, "prologue", stream)) != SR_OK) return rc;
  } else if (!m->prologue_ran)
    SR_FAIL(SR_ERR_INVALID, "sr_unet_forward: ctx == NULL keeps the previous call's prompt, and there has been none (the cross-attention K / V are not projected yet)");
  if ((rc = io_copy(m, "x", x, nullptr, true, stream, "sr_unet_forward")) != SR_OK) return rc;
  if ((rc = io_copy(m, "t", t, nullptr, true, stream, "sr_unet_forward")) != SR_OK) return rc;
  if ((rc = sr_model_run(m, "step", stream)) != SR_OK) return rc;
  if ((rc = io_copy(m, "out", nullptr, out, false, stream, "sr_unet_forward")) != SR_OK) return rc;
  // a bundled plan with K/V injection carries the device flag sr_gather_rows raises on an out-of-range injected index (io window
  // "inject_err"): surfaced here, at the cost of one 4-byte read behind the evaluation
  if (find_io(m, "inject_err")) {
    int32_t flag = 0;
    if ((rc = sr_model_read(m, "inject_err", &flag, stream)) != SR_OK) return rc;
    if (flag != 0) {
      const int32_t zero = 0;
      (void)sr_model_write(m, "inject_err", &zero, stream);
      SR_FAIL(SR_ERR_INVALID, "sr_unet_forward: an injected frame index ('inject') lies outside the batch");
    }
  }
  return SR_OK;
}

extern "C" int sr_vae_decode(sr_model* m, const float* z, float* img, void* stream) {
  if (!m || !z || !img) SR_FAIL(SR_ERR_INVALID, "sr_vae_decode: null argument");
  int rc;
  if ((rc = io_copy(m, "z", z, nullptr, true, stream, "sr_vae_decode")) != SR_OK) return rc;
  if ((rc = sr_model_run(m, "step", stream)) != SR_OK) return rc;
  return io_copy(m, "img", nullptr, img, false, stream, "sr_vae_decode");
}
