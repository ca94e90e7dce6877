/* sr_hip.h — C ABI of libsr_hip.so: the MI355X (gfx950) hot path of the render-then-diffuse frame loop.
 *
 * The reference (92MING/Stable-Renderer) has no FFI for this path: everything is Python + GLSL + torch ops
 * (SURVEY.md §8b).  These entry points are what a binding *beneath* the reference's Python operators calls;
 * each one cites the reference code it replaces (paths relative to <reference>/source).  INTEGRATION.md
 * shows the ctypes stubs a maintainer of the reference would add.
 *
 * Conventions: extern "C"; every pointer is a caller-owned DEVICE pointer unless named host_*; sizes are
 * plain ints; `stream` is a hipStream_t passed as void* (NULL = default stream); return 0 on success or a
 * negative sr_status.  No hidden global state, no allocation, no host<->device sync inside any call unless
 * the comment says so (graph-capture safe).  Activations are NHWC ("pixels x channels"), dtype tag SR_F16 or
 * SR_F32 selects the arithmetic path (fp16 MFMA with fp32 accumulate / exact fp32 MFMA).
 */
#ifndef SR_HIP_H
#define SR_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  SR_OK = 0,
  SR_ERR_INVALID = -1,     /* bad argument / unsupported shape (message via sr_last_error) */
  SR_ERR_LAUNCH = -2,      /* HIP launch/runtime error */
  SR_ERR_UNSUPPORTED = -3
} sr_status;

typedef enum { SR_F16 = 0, SR_F32 = 1 } sr_dtype;

const char* sr_last_error(void);          /* thread-local text of the last failure */
int sr_version(void);
const char* sr_source_hash(void);          /* sha256 of the sources / headers / build flags the binary was made from */
int sr_device_sync(void);                  /* hipDeviceSynchronize (tests only) */

/* ---------------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution / linear layer on MFMA.
 * Replaces torch conv2d / Linear inside ResBlock._forward (comfyUI/comfy/ldm/modules/diffusionmodules/
 * openaimodel.py:253-281), Upsample/Downsample (:82-148), SpatialTransformer proj_in/out and the q/k/v/out/
 * FF Linear layers of BasicTransformerBlock (comfy/ldm/modules/attention.py:495-726), and the VAE decoder
 * convs (comfy/ldm/modules/diffusionmodules/model.py:541-650).
 * out[m, n] = act( scale * sum_k A[m,k] * Wt[n,k] + bias[n] + rowvec[b(m), n] ) + residual[m, n]
 *   m = (b, oy, ox) output pixel, k = (ky, kx, c) with c running over the channel concat [a | a2].
 */
typedef struct {
  const void* a;          /* [B, H, W, C1] activations (dtype)                                        */
  const void* a2;         /* optional [B, H, W, C2] second source (decoder skip concat) or NULL        */
  const void* w;          /* packed weights [Npad, KH*KH*(C1+C2)] dtype, K contiguous, Npad%128==0     */
  const float* bias;      /* [N] fp32 or NULL                                                          */
  const float* rowvec;    /* [B, N] fp32 added per batch entry (time embedding) or NULL                */
  const void* residual;   /* [M, N] dtype or NULL (ignored when transpose_out)                         */
  void* out;              /* [M, N] (or [B, N, ldt] when transpose_out) dtype, fp32 when out_f32       */
  const void* zero_page;  /* >= 16 bytes of zeros (source for padding taps / rows >= M)                 */
  int32_t B, H, W;        /* input batch and spatial size                                              */
  int32_t C1, C2;         /* channels of a / a2 (multiples of 64 for fp16, 32 for fp32)                */
  int32_t N;              /* valid output channels (N of the GEMM; for GEGLU the 2*inner interleaved)  */
  int32_t KH;             /* 1 or 3 (square kernel, pad = KH/2)                                        */
  int32_t stride;         /* 1 or 2                                                                    */
  int32_t upsample;       /* 1: nearest x2 upsample fused in front of the conv (Upsample, :82-119)     */
  int32_t act;            /* 0 none, 1 SiLU, 2 GEGLU ((value,gate) pairs interleaved along n), 3 GELU,
                             4 clamp((v+1)/2, 0, 1) (VAE.decode image range, sd.py:335)                */
  int32_t transpose_out;  /* 1: write out[b][n][t] (t = pixel in batch, row stride ldt) — V^T for attn */
  int32_t ldt;            /* row stride of the transposed output (>= pixels per batch)                 */
  int32_t out_f32;        /* 1: store fp32 regardless of dtype                                         */
  int32_t dtype;          /* sr_dtype                                                                  */
  float scale;            /* multiplies the accumulator (1.0 normally)                                 */
  int32_t rowvec_ld;      /* row stride of rowvec in floats (0 = N): lets every ResBlock read its slice of ONE
                             batched time-embedding projection                                         */
  void* workspace;        /* optional fp32 scratch for split-K (small-M convs of the 8x8 / 16x16 levels: too few
                             output tiles to fill 256 CUs).  NULL = never split.  May be shared by all ops
                             of a stream; contents are dead after the call.                            */
  int64_t workspace_bytes;
  const float* row_stats; /* optional [M, 2] fp32 (rstd, -rstd*mean) of the INPUT rows: LayerNorm folded into the GEMM --
                             out = rstd[m] * (x[m,:] . W'[n,:]) + (-rstd[m]*mean[m]) * colsum[n] + bias'[n] with
                             W' = W * gamma (per k), colsum[n] = sum_k W'[n,k], bias' = bias + W . beta: the GEMM reads
                             the un-normalised rows and the normalised tensor is never materialised
                             (BasicTransformerBlock norm1/2/3 -> to_q/k/v, ff.net.0, attention.py:521-654)          */
  const float* colsum;    /* [N] fp32, required with row_stats                                                     */
  int32_t tile;           /* 0 = library heuristic; 1 = 256x128 (8 waves, 3-stage), 2 = 128x128, 3 = 128x64, 4 = 64x64,
                             5 = 256x320 (fp16, N % 320 == 0; 2 LDS stages of 128-byte K-steps), 6 = 256x320 with a 4-deep ring of
                             64-byte K-steps, 7 = 128x320 (8 waves, for half as many pixels), 8 = patch-stationary 3x3 (256x320; fp16, KH 3,
                             stride 1, one source, N % 320 == 0, M % 256 == 0, 256-pixel tiles = whole image rows or whole 8x8
                             images: the activation halo is staged once per 32-channel chunk and the nine taps read it at nine LDS
                             offsets; SR_ERR_INVALID otherwise), 9 = 128x160 and 10 = 128x320 with 64-byte K-steps (fp16; two co-resident
                             workgroups per CU for the K-short linear layers), 11 = 128x128 as 8 waves and 12 = 256x128 with 64-byte K-steps
                             (fp16; the same for widths that are multiples of 128 only), 13 = 64x64 / 14 = 128x64 / 15 = 128x128 with a
                             deep LDS ring (8 / 6 / 4 stages, one workgroup per CU: for grids of less than one workgroup per CU -- small
                             batches, the 16x16 / 8x8 levels -- where nothing else hides the latency of a K-step).  Set by the host-side
                             per-shape tuner (ops.tune_igemm)                                                                 */
  int32_t split;          /* 0 = split-K decided by the library's cost model, -1 = never split, 2..16 = split the
                             partly empty last round of workgroups this many ways over K (tiles 2 / 3 from 32 K-steps of 128
                             bytes on, tiles 14 / 15 from 8 on; what the per-shape tuner measures)                 */
  int32_t pad_br;         /* 1: a 3x3 conv pads ONLY the bottom / right border (window of output (y,x) starts at input
                             (y*stride, x*stride)): the VAE encoder's Downsample = F.pad(x, (0,1,0,1)) + conv(stride 2,
                             padding 0) (comfy/ldm/modules/diffusionmodules/model.py:77-95).  0: symmetric KH/2 padding          */
  const void* prefetch;   /* optional: bytes another launch will stream soon (the packed weights of the NEXT igemm of a plan).  Every
                             workgroup touches its share of them at kernel start (one 4-byte LDS-DMA read per 64 bytes, the data
                             is discarded), so they sit in the Infinity Cache when their layer starts instead of coming from HBM
                             on that layer's critical path (a UNet evaluation streams 1.7 GB of weights, far more than the
                             256 MB cache keeps between evaluations).  NULL = none.  Never changes a result.                    */
  int64_t prefetch_bytes;
  int32_t up_h, up_w;     /* with upsample = 1: output size of the fused nearest upsample when it is not exactly 2H x 2W (0 = x2);
                             src = floor(dst * in/out) as F.interpolate(mode="nearest"): odd-sized latents (conditioning areas)
                             where Upsample.forward targets the skip tensor's size (openaimodel.py:109-121)                      */
  int32_t* split_counters;/* optional: SR_IGEMM_SPLIT_COUNTERS int32 tile counters, ZERO before the first launch that sees them and left
                             zero by every launch (one array per workspace: launches that may overlap in time need their own).  With
                             them a split-K launch finishes inside the GEMM kernel: a workgroup publishes its fp32 partial, counts
                             itself in, and the LAST of a tile's S workgroups to arrive sums the S partials in fixed z order and runs
                             the ordinary epilogue -- same bits whichever workgroup that is, no float atomics, nobody waits, and no
                             second launch.  NULL: the partials are reduced by a separate kernel (two launches).                  */
  int32_t group;          /* plans only (sr_plan_run / sr_plan_capture; sr_igemm ignores it): this op and the next group-1 ops are igemm
                             ops that do not depend on each other (the Q, K and V^T projections of a transformer block; a ResBlock's
                             skip convolution and its first 3x3 convolution) and are handed to sr_igemm_group together.  0 / 1 = alone */
  int32_t ln_inline;      /* 1: the folded LayerNorm of `row_stats`, with the statistics taken INSIDE this launch: a K-short linear layer
                             stages every complete input row (K = C1 = the normalised width) through its workgroup anyway, so the waves
                             sum x and x^2 of the fragments they feed to the MFMAs (v_dot2_f32_f16, fp32 accumulation) and the epilogue
                             applies out = rstd * (x . W'^T) - rstd * mean * colsum + bias' -- no LayerNorm pass, no statistics pass, no
                             statistics tensor.  Needs colsum, row_stats == NULL, KH 1, stride 1, one source; never split over K        */
  float ln_eps;           /* epsilon of that LayerNorm (1e-5 in BasicTransformerBlock)                                                  */
  int32_t tile_order;     /* which operand an XCD keeps in ITS L2 (workgroups are dealt to the 8 XCDs round-robin; each XCD gets a contiguous
                             run of the tile sequence): 0 = rows first -- an XCD works through a band of M-tiles and every N-tile of them:
                             the activations cross the fabric once, the packed weights once per XCD (right for the 64x64 / 32x32 levels,
                             whose weights are small); 1 = columns first -- an XCD takes a band of N-tiles and every M-tile of them: the
                             weights cross once, the activations once per XCD (the 16x16 / 8x8 levels: 29.5-59 MB of weights against
                             2.6-21 MB of activations per layer).  Same values either way; tile 8 ignores it.                          */
} sr_igemm_args;
#define SR_IGEMM_SPLIT_COUNTERS 4096
#define SR_IGEMM_GROUP_MAX 4
int sr_igemm(const sr_igemm_args* args, void* stream);
/* n <= SR_IGEMM_GROUP_MAX INDEPENDENT problems (no output of one is an input of another).  When they can share a kernel -- same
 * dtype, the same pinned `tile` (2, 3, 4, 13, 14, 15; fp16 also 9, 10), split = -1, row-major outputs -- they run as ONE launch
 * whose workgroups are divided among the problems: at small batches every such kernel is a latency chain on a fraction of the
 * CUs, and kernels of one stream otherwise run strictly one after another.  Anything else is launched one by one: same results. */
int sr_igemm_group(const sr_igemm_args* const* args, int32_t n, void* stream);

/* GroupNorm(32 groups) [+SiLU] over NHWC, optional channel concat of two sources (th.cat([h, hsp]) in
 * UNetModel.forward, openaimodel.py:921).  Replaces GroupNorm32 + SiLU in ResBlock.in_layers/out_layers,
 * SpatialTransformer.norm (eps 1e-6), VAE Normalize.  `partials` scratch: sr_groupnorm_scratch_floats(B, HW) floats -- the largest
 * need of any batch up to B (small batches use more, smaller pixel chunks), so a buffer sized for a host's largest batch serves
 * every smaller one. */
typedef struct {
  const void* x; const void* x2;      /* [B, HW, C1], [B, HW, C2] or NULL */
  const float* gamma; const float* beta;  /* [C1+C2] fp32 */
  void* y;                             /* [B, HW, C1+C2] dtype */
  float* partials;                     /* scratch */
  int32_t B, HW, C1, C2, groups, silu, dtype;
  float eps;
} sr_groupnorm_args;
int sr_groupnorm(const sr_groupnorm_args* args, void* stream);
int64_t sr_groupnorm_scratch_floats(int32_t B, int32_t HW);

/* Per-row LayerNorm statistics for the folded form (sr_igemm_args.row_stats): stats[r] = (rstd, -rstd*mean), fp32,
 * biased variance as torch.nn.LayerNorm. */
int sr_row_stats(const void* x, float* stats, int32_t rows, int32_t C, float eps, int32_t dtype, void* stream);
/* LayerNorm over the last dim (BasicTransformerBlock.norm1/2/3, attention.py:521,613,648). rows x C. */
int sr_layernorm(const void* x, const float* gamma, const float* beta, void* y, int32_t rows, int32_t C,
                 float eps, int32_t dtype, void* stream);
/* The same over GATHERED rows: output row r = LayerNorm of row sel[r / frame_rows] * frame_rows + r % frame_rows of x ([n_frames,
 * frame_rows, C]); sel is a DEVICE array of nsel frame indices.  With sr_igemm_args.ln_inline the B-frame LayerNorm in front of the
 * self-attention disappears into the Q projection and only the K/V-injected frame's tokens need normalising
 * (OverlapCorresponder.pre_atten_inject, corresponder.py:204-214): this is that LayerNorm and the pick of the frame in one launch.
 * An index outside [0, n_frames) gives zero rows and raises *err_flag (as sr_gather_rows). */
int sr_layernorm_gather(const void* x, const int32_t* sel, int32_t nsel, int32_t frame_rows, int32_t n_frames, int32_t* err_flag,
                        const float* gamma, const float* beta, void* y, int32_t C, float eps, int32_t dtype, void* stream);

/* Fused softmax(Q K^T / sqrt(d)) V (optimized_attention, attention.py:92-387), fp32 softmax statistics.
 *  q  [B, Tq, heads*d]   k [Bk, Tk, heads*d]   vt [Bk, heads, d, ldt] (V transposed, written by sr_igemm
 *  transpose_out)   o [B, Tq, heads*d].  Bk == B, or Bk == 1: every batch entry attends to the same K/V
 *  (OverlapCorresponder.pre_atten_inject, common_utils/stable_render_utils/corresponder.py:188-220). */
typedef struct {
  const void* q; const void* k; const void* vt; void* o;
  int32_t B, Bk, Tq, Tk, heads, d, ldt, dtype;
  int32_t q_stride, k_stride;   /* row strides in elements (heads*d when packed) */
  float scale;                  /* d^-0.5 */
} sr_attention_args;
int sr_attention(const sr_attention_args* args, void* stream);

/* small element-wise pieces of UNetModel.forward / BaseModel.apply_model */
int sr_nchw_to_nhwc(const float* x, void* y, int32_t B, int32_t C, int32_t HW, int32_t Cpad, float scale_mul,
                    const float* per_batch_scale, int32_t dtype, void* stream);   /* y[b,p,c] = x[b,c,p]*s */
int sr_nhwc_to_nchw(const void* x, float* y, int32_t B, int32_t C, int32_t HW, int32_t ldc, int32_t dtype,
                    void* stream);
int sr_timestep_embedding(const float* t, void* y, int32_t B, int32_t dim, int32_t dtype, void* stream);
                                                    /* util.py:241-261: cat(cos, sin)(t * 10000^(-i/half)) */
int sr_silu(const void* x, void* y, int64_t n, int32_t dtype, void* stream);
int sr_cast(const void* x, int32_t src_dtype, void* y, int32_t dst_dtype, int64_t n, void* stream);
/* tuner aid: read [p, p + bytes) (16-byte aligned) so that it sits in L2 / Infinity Cache like a tensor the previous kernel of a plan
 * has just produced; no output */
int sr_cache_touch(const void* p, int64_t bytes, void* stream);
int sr_softmax_rows(void* x, int32_t rows, int32_t cols, int32_t dtype, void* stream);   /* in place; VAE mid attention */
/* y[j] = x[sel[j]] for j < nsel, rows of row_bytes bytes (multiple of 16); `sel` is a DEVICE int32 array read at run
 * time, so a captured plan stays valid when the injected frame changes: random_k = k_context[_random_frame_indices]
 * (OverlapCorresponder.pre_atten_inject, corresponder.py:207-214).  x holds n_rows rows: an index outside [0, n_rows) is
 * never dereferenced -- its output row is zero-filled and *err_flag (optional DEVICE int32, sticky) is set to 1, which the
 * host turns into the IndexError the reference's k_context[idx] raises. */
int sr_gather_rows(const void* x, const int32_t* sel, void* y, int32_t nsel, int32_t n_rows, int64_t row_bytes,
                   int32_t* err_flag, void* stream);
/* y = a + s*b (dtype tensors): apply_control (openaimodel.py:374-386), guided-hint add (cldm.py:297-300) */
int sr_add_scaled(const void* a, const void* b, void* y, int64_t n, float s, int32_t dtype, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Launch plan: a flat array of ops executed back-to-back on one stream by native code (the Python host
 * builds it once per (model, batch, resolution); no Python in the per-step path).  */
typedef enum {
  SR_OP_IGEMM = 1, SR_OP_GROUPNORM = 2, SR_OP_LAYERNORM = 3, SR_OP_ATTENTION = 4, SR_OP_NCHW_TO_NHWC = 5,
  SR_OP_NHWC_TO_NCHW = 6, SR_OP_TIMESTEP_EMBED = 7, SR_OP_SILU = 8, SR_OP_SOFTMAX_ROWS = 9, SR_OP_GATHER_ROWS = 10, SR_OP_ADD_SCALED = 11,
  SR_OP_FORK = 12,   /* side lane may start: it waits for everything issued on the main lane so far                       */
  SR_OP_JOIN = 13,   /* main lane waits for everything issued on the side lane so far                                     */
  SR_OP_ROW_STATS = 14, /* sr_row_stats; uses the `ln` member: x, y = stats, rows, C, dtype, eps                          */
  SR_OP_LAYERNORM_GATHER = 15 /* sr_layernorm_gather; the `ln` member with sel / err_flag / frame_rows / n_frames, rows = nsel * frame_rows */
} sr_op_kind;
/* lane: 0 = the caller's stream, 1 = the executor's side stream.  Independent branches of the graph (a ResBlock's 1x1
 * skip convolution beside its GroupNorm/conv path; the injected frame's K/V projections beside the Q projection) are
 * emitted as FORK, side-lane ops, main-lane ops ..., JOIN: many UNet kernels are single-round or latency-bound, so a second
 * lane fills their ramps and tails.  In a captured plan the two lanes become parallel branches of the hipGraph. */
typedef struct {
  int32_t kind; int32_t lane;
  union {
    sr_igemm_args igemm;
    sr_groupnorm_args gn;
    sr_attention_args attn;
    struct { const void* x; const float* gamma; const float* beta; void* y; int32_t rows, C, dtype; float eps;
             const int32_t* sel; int32_t* err_flag; int32_t frame_rows, n_frames; } ln;   /* (the last four: SR_OP_LAYERNORM_GATHER) */
    struct { const void* x; void* y; const float* per_batch_scale; int32_t B, C, HW, Cpad, dtype, ldc; float scale; } cvt;
    struct { const float* t; void* y; int32_t B, dim, dtype; } temb;
    struct { const void* x; void* y; int64_t n; int32_t dtype; int32_t rows, cols; } ew;
    struct { const void* x; void* y; const int32_t* sel; int64_t row_bytes; int32_t nsel; int32_t n_rows; int32_t* err_flag; } gather;
    struct { const void* a; const void* b; void* y; int64_t n; float s; int32_t dtype; } add;
  } u;
} sr_op;
int sr_plan_run(const sr_op* host_ops, int32_t n_ops, void* stream);
/* Capture the plan into a hipGraph on `stream` (must be a non-default stream); *graph_exec receives an opaque
 * handle for sr_graph_launch / sr_graph_destroy. */
int sr_plan_capture(const sr_op* host_ops, int32_t n_ops, void* stream, void** graph_exec);
int sr_graph_launch(void* graph_exec, void* stream);
int sr_graph_destroy(void* graph_exec);

/* ---------------------------------------------------------------------------------------------------
 * Operator-level entry points: what the reference's model calls look like to a host written in any language.
 *   sr_unet_forward  replaces  BaseModel.apply_model -> self.diffusion_model(xc, t, context=...)   (comfy/model_base.py:93-127,
 *                              UNetModel.forward, comfy/ldm/modules/diffusionmodules/openaimodel.py:841-946)
 *   sr_vae_decode    replaces  VAE.decode -> self.first_stage_model.decode(samples)                  (comfy/sd.py:329-346,
 *                              Decoder.forward, comfy/ldm/modules/diffusionmodules/model.py:541-650)
 * A *model bundle* file holds a lowered launch plan in relocatable form -- the sr_op arrays, (tensor, offset) in place of every
 * device pointer, the tensors' sizes and initial contents (packed weights), the named input / output windows -- for ONE
 * (weights, batch, resolution, injected-frame count); stable-renderer_amd/bundle.py writes it from a built plan (tiles already chosen by the
 * tuner).  sr_model_load allocates and uploads on the CURRENT device; the handle is then independent of Python and of torch.
 * Plans: "prologue" (prompt-only work: cross-attention K / V, label_emb; run when ctx changes) and "step" (one evaluation).
 * Inputs / outputs by name: UNet "x" (B,4,h,w) fp32, "t" (B,) fp32, "ctx" (B,n_ctx,ctx_dim) in the model dtype, optional "y",
 * "inject" (n_rand int32 batch indices of the K/V-injected frames), "out" (B,4,h,w) fp32; VAE "z" (N,4,h,w) fp32, "img" (N,H,W,3) fp32.
 * sr_model_write / sr_model_read / sr_unet_forward / sr_vae_decode accept host OR device pointers (hipMemcpyDefault). */
typedef struct sr_model sr_model;
int sr_model_load(const char* path, sr_model** model);
int sr_model_free(sr_model* model);
int sr_model_io(sr_model* model, const char* name, void** dev_ptr, int64_t* nbytes);
int sr_model_run(sr_model* model, const char* plan, void* stream);
int sr_model_write(sr_model* model, const char* name, const void* src, void* stream);          /* asynchronous on `stream` */
int sr_model_read(sr_model* model, const char* name, void* dst, void* stream);                 /* synchronises `stream`    */
/* ctx == NULL: keep the prompt of the previous call (its K / V stay projected).  `out` is complete when `stream` has drained. */
int sr_unet_forward(sr_model* unet, const float* x, const float* t, const void* ctx, float* out, void* stream);
int sr_vae_decode(sr_model* vae, const float* z, float* img, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Sampler arithmetic (comfy/model_sampling.py:7-29 EPS; comfy/samplers.py:323-358 CFG;
 * comfy/k_diffusion/sampling.py:129-149 euler, :749-776 ddpm, :779-793 lcm).  All fp32, x is (N,4,h,w). */
/* xin[0:N] = xin[N:2N] = x / sqrt(sigma^2+1)  (uncond chunk first, then cond; `copies` = 1 or 2) */
int sr_eps_scale_input(const float* x, float* xin, int64_t n_per_copy, int32_t copies, float sigma, void* stream);
/* denoised = x - eps*sigma per chunk; CFG: u + (c-u)*cfg (copies==2, eps = [uncond | cond]);
 * d = (x - denoised)/sigma (euler derivative, computed BEFORE callbacks as the reference does) */
int sr_cfg_denoise(const float* x, const float* eps, float* denoised, float* d, int64_t n, int32_t copies,
                   float sigma, float cfg, void* stream);
/* Conditioning composition: several positive / negative conditionings with masks, strengths and areas
 * (comfy/samplers.py:50-127 get_area_and_mult, :176-320 calc_cond_uncond_batch).  One GROUP = the entries that run as one
 * model call: same area (ah, aw, y0, x0) of the (N,C,h,w) latent, `chunks` entries in batch order.
 *   sr_cond_crop_scale: xin[j*N+n] = x[n, :, y0:y0+ah, x0:x0+aw] / sqrt(sigma^2+1) for every chunk j
 *   sr_cond_accumulate: for j in batch order: out_kind[area] += (x_crop - eps[j]*sigma) * mult[j]; cnt_kind[area] += mult[j]
 *                       (kinds[j] 0 = cond, 1 = uncond, DEVICE int32; mult (chunks,N,C,ah,aw) = mask*mask_strength*strength
 *                       with the 8-cell feathering of mask-less areas; out_* start at 0, cnt_* at 1e-37)
 *   sr_cfg_combine:     c = out_c/cnt_c, u = out_u/cnt_u, denoised = u + (c-u)*cfg (samplers.py:314-351), d = (x-denoised)/sigma */
int sr_cond_crop_scale(const float* x, float* xin, int32_t N, int32_t C, int32_t h, int32_t w, int32_t ah, int32_t aw, int32_t y0,
                       int32_t x0, int32_t chunks, float sigma, void* stream);
int sr_cond_accumulate(const float* x, const float* eps, const float* mult, const int32_t* kinds, float* out_c, float* cnt_c,
                       float* out_u, float* cnt_u, int32_t N, int32_t C, int32_t h, int32_t w, int32_t ah, int32_t aw, int32_t y0,
                       int32_t x0, int32_t chunks, float sigma, void* stream);
int sr_cfg_combine(const float* x, const float* out_c, const float* cnt_c, const float* out_u, const float* cnt_u, float* denoised,
                   float* d, int64_t n, float sigma, float cfg, void* stream);
int sr_euler_step(float* x, const float* d, int64_t n, float dt, void* stream);           /* x += d*dt */
/* VAE.encode's posterior sample (comfy/ldm/modules/distributions/distributions.py:24-37 via DiagonalGaussianRegularizer,
 * comfy/ldm/models/autoencoder.py:13-31): moments (B, HW, 2*zc) fp32 NHWC = [mean | logvar] from quant_conv;
 * z[b,c,p] = mean + exp(0.5*clamp(logvar,-30,20)) * noise[b,c,p]; noise / z (B, zc, HW) fp32 NCHW */
int sr_vae_sample(const float* moments, const float* noise, float* z, int32_t B, int32_t zc, int32_t HW, void* stream);
/* DDPMSampler_step + rescale; noise = host-drawn randn (may be NULL when sigma_next == 0) */
int sr_ddpm_step(float* x, const float* denoised, const float* noise, int64_t n, float sigma, float sigma_next,
                 void* stream);
int sr_lcm_step(float* x, const float* denoised, const float* noise, int64_t n, float sigma_next, void* stream);
int sr_axpby(float* y, const float* x, int64_t n, float a, float b, void* stream);         /* y = a*x + b*y */

/* ---------------------------------------------------------------------------------------------------
 * Stable-rendering kernels */
/* IDMap masks (engine/static/corrmap.py:119-126): mask = (map_index==2048) | all-zero, as fp32 0/1. */
int sr_idmap_masks(const int32_t* ids, float* masks, int64_t n_pixels, void* stream);

/* Build the per-call overlap structure from id maps (replaces IDMap.create_vertex_screen_info,
 * corrmap.py:220-280 + the per-step unique() of tensor_group_by_then_average, math_utils.py:86-161).
 * For every valid pixel (frame f, y, x): latent cell = (f, int(fp32(y/W)*lh), int(fp32(x/H)*lw)).
 *   cell_vid[cell]  = vertexID of the LAST valid pixel (f,y,x order) mapping to the cell, -1 if none
 *   pix_cell[pixel] = cell index or -1, pix_vid[pixel] = vertexID (valid pixels only)
 * vid_capacity = 1 + max vertexID the dense per-vertex tables must hold (returned through *max_vid, device).
 * Two-pass, no atomics on the winner: deterministic = sequential "last writer wins". */
int sr_overlap_build(const int32_t* ids, int32_t N, int32_t H, int32_t W, int32_t lh, int32_t lw,
                     int32_t* pix_cell, int32_t* cell_vid, int32_t* max_vid, void* stream);
/* Second build phase, after the caller has read max_vid back: CSR of vertexID -> the latent cells of EVERY valid pixel that
 * carries it (one entry per pixel, so a cell seen through 64 pixels counts 64 times, exactly as the rows of
 * create_vertex_screen_info do).  vid_off: vid_capacity+1 ints (exclusive prefix sum of the per-vertex pixel counts),
 * entries: one int per valid pixel (info[2] of sr_overlap_build), scratch: sr_overlap_csr_scratch_ints(vid_capacity) ints.
 * Integer atomics only (counts, cursors): the order of the entries inside a segment may differ between runs, nothing the
 * step computes depends on it. */
int64_t sr_overlap_csr_scratch_ints(int32_t vid_capacity);
int sr_overlap_csr(const int32_t* ids, const int32_t* pix_cell, int32_t N, int32_t H, int32_t W, int32_t vid_capacity,
                   int32_t* vid_off, int32_t* entries, int32_t* scratch, void* stream);
/* One OverlapCorresponder.step_finished (corresponder.py:298-376) on x (N,C,lh,lw) fp32, in place:
 * per latent cell, the mean over all pixels carrying the cell's winning vertexID of the latent they gather (each id pixel counts
 * once; tensor_group_by_then_average, math_utils.py:86-161), blend (1-r)*v + r*mean, then AdaIN(content = x, style = blended)
 * per (n,c) (math_utils.py:55-80).  No floating-point atomics: the segment sum is exact 2^-28 fixed point in int64 and the
 * statistics are fixed-order reductions, so the result is bit-reproducible.  Range contract of that sum: a finite latent value
 * saturates at +-4096 (the conversion rounds to nearest 2^-28); a NaN or an infinity in ANY pixel of a vertex makes that
 * vertex's mean of that channel NaN in every view sharing the vertex -- as the reference's float mean does -- so a diverged
 * view stays visible to a finiteness check downstream.  blended: (N,C,lh,lw) fp32, written by the call (the style tensor of
 * the AdaIN; also what tests read). */
int sr_overlap_step(float* x, const int32_t* cell_vid, const int32_t* vid_off, const int32_t* entries, int32_t N, int32_t C,
                    int32_t lh, int32_t lw, int32_t vid_capacity, float ratio, float* blended, void* stream);

/* adaptive_instance_normalization NCHW fp32 (math_utils.py:27-80): out = (c-mean_c)/std_c*std_s+mean_s,
 * std = sqrt(unbiased var + eps).  content (N,C,HWc), style (N,C,HWs) with element strides so NHWC inputs
 * work too (chan stride / pixel stride).  stats scratch N*C*4 floats. */
int sr_adain(const float* content, int64_t c_ps, int64_t c_cs, int64_t c_ns, int32_t HWc,
             const void* style, int32_t style_dtype, int64_t s_ps, int64_t s_cs, int64_t s_ns, int32_t HWs,
             float* out, int32_t N, int32_t C, float eps, float* stats, void* stream);

/* Engine noise -> latent noise (RenderManager._save_frame_data, engine/managers/renderManager.py:926-936):
 * n = noise*(1-mask) + bg*mask (fp16 product, fp32 sum), 64-pixel row-strip means ("view(-1,8,8,4)"),
 * then AdaIN against the full-res fp16 noise -> out (1,4,H/8,W/8) fp32.  pooled: (H/8*W/8, 4) fp32.
 * stats: optional scratch of 2048 floats -- with it the style statistics are reduced by 256 workgroups instead of 4. */
int sr_noise_pool(const void* noise_f16, const void* alpha_f16, const float* bg, float* pooled, float* out,
                  int32_t H, int32_t W, float* stats, void* stream);
/* The same with `strip` consecutive pixels per mean instead of 64: NoiseSequenceLoader pools with reshape_magnitude m = H // 64
 * (SD15) or H // 128 (SDXL), i.e. view(-1, m, m, 4).mean((1, 2)) = means of m*m consecutive pixels, viewed (H/m, W/m)
 * (comfyUI/stable_rendering/_nodes/loaders.py:131-146).  pooled: (H*W/strip, 4) fp32, out: (1, 4, H*W/strip) fp32. */
int sr_noise_pool_strips(const void* noise_f16, const void* alpha_f16, const float* bg, float* pooled, float* out,
                         int32_t H, int32_t W, int32_t strip, float* stats, void* stream);

/* CorrespondMap._update (engine/static/corrmap.py:672-736) for ONE frame, deterministic last-writer-wins.
 * frame (H*W, Cf) fp32 (Cf = 3: alpha 1 appended, corrmap.py:684), ids (H*W,4) int32, mask (H*W) fp32 or NULL
 * (rows with mask>0 kept), values (k*k, V, 4) fp16, writtens (k*k, V) uint8.  src_index (H*W) int32 or NULL:
 * colour row used for pixel i (the host reproduces the reference's double-gather quirk by passing it).
 * winner scratch: k*k*V int32.  mode_first: 1 = skip already-written cells.  Out-of-range (map_index, vid)
 * -> *err_flag (device int) set to 1 and the row is skipped (the reference raises IndexError). */
int sr_corrmap_update(const float* frame, int32_t Cf, const int32_t* ids, const float* mask,
                      const int32_t* src_index, int32_t n_pixels, int32_t sprite, int32_t material,
                      int32_t check_sprite, int32_t check_material, int32_t mode_first, void* values,
                      uint8_t* writtens, int32_t kk, int32_t V, int32_t* winner, int32_t* err_flag, void* stream);

/* Legacy "latent overlapping" (legacy_codes/stable_rendering_algo/overlap/overlap.py:83-152 Overlap.__call__ and :180-222
 * ResizeOverlap.__call__ with nearest interpolation; algorithms.py:34-118), fused at latent resolution:
 * every output cell (f,i,j) looks up the corr-map pixel (f, i*H/h, j*W/w) that the nearest down-sampling would keep, walks
 * that vertex's trace (CSR: offsets / tr_f / tr_y / tr_x in (frame,y,x) order), gathers the latent cells under the trace
 * pixels ((2r+1)-pixel clamped DIAGONAL pooling, overlap.py:69-76), forms the weighted mean (algo 0 average, 1 frame
 * distance, 2 pixel distance, 3 view normal incl. the reference's column-sum-per-row normalisation) and blends with alpha;
 * keep_nonzero: ResizeOverlap's where(ovlp != 0, ovlp, orig).  Jacobi form: reads x, writes y (the reference updates in
 * place in dict order, which only differs for radius > 0).  x,y: (T,C,h,w) fp32; pix_vert: (T*H*W) vertex index or -1. */
int sr_legacy_overlap(const float* x, float* y, const int32_t* pix_vert, const int32_t* offsets, const int32_t* tr_f,
                      const int32_t* tr_y, const int32_t* tr_x, const float* view_normal, int32_t T, int32_t C, int32_t h,
                      int32_t w, int32_t H, int32_t W, float alpha, int32_t radius, int32_t algo, int32_t keep_nonzero,
                      void* stream);

/* The same legacy Overlap in the REFERENCE'S IN-PLACE ORDER, needed when the kernel radius is > 0: the reference writes into a tensor
 * that aliases its source (overlap.py:103 `.detach()`), so a vertex sees the updates of the vertices before it (dict order)
 * through its diagonal pooling windows.
 *   sr_legacy_levels (HOST arrays in and out): sorts the vertices into conflict-free levels; `order` = vertex indices in dict order
 *     (first occurrence in (frame, y, x) scan order); level_of[v] = -1 for vertices seen once (they never write).
 *   sr_legacy_overlap_seq: U (T,C,H,W) fp32 at corr-map resolution is updated in place, level by level (lvl_vert = DEVICE vertex
 *     indices grouped by level, lvl_off_host = HOST prefix offsets of the groups, max_len = longest trace); newval: scratch of
 *     (#trace entries x C) floats.
 *   sr_nearest_resize: F.interpolate(mode="nearest") between latent and corr-map size; with keep_if_zero (same shape as dst) a zero
 *     result keeps that tensor's value: ResizeOverlap's where(ovlp != 0, ovlp, orig) (overlap.py:180-222). */
int sr_legacy_levels(const int32_t* offsets, const int32_t* tr_f, const int32_t* tr_y, const int32_t* tr_x, const int32_t* order,
                     int32_t n_vertices, int32_t T, int32_t H, int32_t W, int32_t radius, int32_t* level_of, int32_t* n_levels);
int sr_legacy_overlap_seq(float* U, float* newval, const int32_t* lvl_vert, const int32_t* lvl_off_host, int32_t n_levels, int32_t max_len,
                          const int32_t* offsets, const int32_t* tr_f, const int32_t* tr_y, const int32_t* tr_x, const float* view_normal,
                          int32_t C, int32_t H, int32_t W, float alpha, int32_t radius, int32_t algo, void* stream);
int sr_nearest_resize(const float* src, float* dst, int64_t planes, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo,
                      const float* keep_if_zero, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Software rasterizer for the G-buffer pass (engine/shaders/default_Gbuffer.{vert,frag}.glsl,
 * engine/managers/renderManager.py:499-571).  One call = one draw task (mesh x material). */
typedef struct {
  const float* pos; const float* normal; const float* uv;   /* per-vertex: [nv,3] [nv,3] [nv,2] fp32 */
  const float* color;                                       /* [nv,3] or NULL */
  const int32_t* vertex_id;                                 /* [nv] flat ids (mesh.py:279-281) or NULL */
  const int32_t* tris;                                      /* [nt,3] */
  int32_t nv, nt;
  float MV[16], MV_IT[16], P[16];                           /* column-major as GLM (runtimeManager.py:165-206) */
  int32_t sprite_id, material_id, corrmap_k, use_texcoord_id, render_mode;  /* 0 normal, 1 baked, 2 baking */
  int32_t has_vertex_color, depth_test, cull_back;
  int32_t id_w, id_h;                                       /* texcoord-id grid (diffuse or corr-map size) */
  const void* noise_tex; int32_t noise_w, noise_h;          /* RGBA16F NEAREST or NULL */
  const void* diffuse_tex; int32_t diffuse_w, diffuse_h;    /* RGBA32F or NULL; NEAREST unless diffuse_levels >= 2 (below) */
  const void* corrmap_tex; int32_t corr_w, corr_h;          /* fp16 (k*k, h, w, 4) for render_mode 1 or NULL */
  /* tangent-space normal map (default_Gbuffer.frag.glsl:114-123, vert.glsl:52-53): all three or none.  view normal =
   * normalize(MV_IT * normalize(TBN * normalize(tex.rgb*2-1))), TBN columns = interpolated normalize(tangent), normalize(bitangent),
   * raw normal */
  const float* tangent; const float* bitangent;             /* per-vertex [nv,3] fp32 or NULL */
  const void* normal_tex; int32_t normal_w, normal_h;       /* RGBA32F NEAREST or NULL */
  int32_t diffuse_levels;                                   /* 0 / 1: diffuse_tex is one level, sampled NEAREST.  >= 2: the level-0 image is
                                                               followed by its mip chain (level k = max(1, w >> k) x max(1, h >> k) texels,
                                                               RGBA32F, scene.build_mip_chain) and sampled TRILINEAR with REPEAT -- the
                                                               reference's default for file textures (GL_LINEAR_MIPMAP_LINEAR / GL_LINEAR,
                                                               engine/static/texture/texture.py:57-60, 276-289; anisotropy not restated).
                                                               Level of detail from the perspective-correct uv of the same triangle one
                                                               pixel right / down; fixed fp32 order = oracle/raster_ref.c tex_trilinear */
} sr_draw;
typedef struct {
  void* color;      /* [H,W,4] fp16 */
  int32_t* id;      /* [H,W,4] int32 */
  float* pos;       /* [H,W,3] fp32 */
  void* normal_depth; /* [H,W,4] fp16 */
  void* noise;      /* [H,W,4] fp16 */
  float* canny;     /* [H,W,3] fp32 */
  float* zbuf;      /* [H,W] fp32 window-space depth (1.0 = far) */
  int32_t W, H;
} sr_gbuffer;
int sr_gbuffer_clear(const sr_gbuffer* g, void* stream);
/* identical-G-buffer tasks (RenderManager.AddIdenticalGBufferTask / _wrapIdenticalGBufferTask save_to_temp,
 * engine/managers/renderManager.py:95-133, 709-733): an object drawn ALONE into `src` is merged into the accumulated planes `acc`
 * wherever its depth (normal_depth.a, closer = larger) is greater; all seven planes of the winning pixel move together. */
int sr_gbuffer_depth_merge(const sr_gbuffer* acc, const sr_gbuffer* src, void* stream);
/* the display image: defer pass (engine/shaders/default_defer_render.frag.glsl:20-59: baking-mode rainbow tint of AI-object ids)
 * followed by the post process (default_post_process.frag.glsl:21-39: gamma, exposure, saturation, brightness, contrast, HDR).
 * color RGBA16F (H,W,4), ids (H,W,4) int32 (may be NULL unless is_baking), out RGBA fp32 (H,W,4). */
int sr_defer_post(const void* color_rgba16f, const int32_t* ids, float* out_rgba, int32_t W, int32_t H, int32_t is_baking,
                  int32_t enable_gamma, int32_t enable_hdr, float gamma, float exposure, float saturation, float brightness,
                  float contrast, void* stream);
/* scratch: see sr_raster_scratch_bytes(nt, W, H) */
int sr_raster_draw(const sr_draw* d, const sr_gbuffer* g, void* scratch, int64_t scratch_bytes, void* stream);
int64_t sr_raster_scratch_bytes(int32_t nt, int32_t W, int32_t H);

#ifdef __cplusplus
}
#endif
#endif
