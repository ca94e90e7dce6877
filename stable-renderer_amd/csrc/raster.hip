// Tiled software rasterizer for the G-buffer pass (see include/sr_hip.h: sr_raster_draw) — replaces the OpenGL
// draw + six glCopyTexSubImage2D snapshots per task (engine/managers/renderManager.py:499-571) and the GL<->CUDA
// texture copies (engine/static/texture/texture.py:166-254): the six G-buffer planes are plain HBM tensors.
//
//   raster_setup : one thread per triangle — vertex stage (default_Gbuffer.vert.glsl:40-57) for its three vertices,
//                  28.4 fixed-point snapping, integer edge setup, pixel bbox; writes a 256-byte TriRec.
//   raster_tiles : one 256-thread workgroup per 16x16 pixel tile, one thread per pixel.  The workgroup walks the
//                  draw's triangles in index order, bins the ones whose bbox touches the tile into LDS (ballot +
//                  prefix compaction keeps primitive order; the bounding boxes come from a dense 16-byte-per-triangle array
//                  behind the TriRecs), stages their TriRecs in LDS, and every thread evaluates
//                  coverage / depth / the fragment shader (default_Gbuffer.frag.glsl:100-257) for its own pixel with the
//                  pixel's state in registers.  Each pixel is read at most once (lazily, the pre-draw "snapshot") and
//                  written at most once per draw: 68 B/pixel of HBM traffic, no atomics, no inter-workgroup traffic,
//                  and results independent of scheduling (bit-identical to oracle/raster_ref.c, which defines the rule).
//                  (Measured and not kept: every lane searching its own next covering fragment and all lanes then shading
//                  together -- one shader pass per fragment layer instead of one per triangle touching the wave: the
//                  per-lane record reads cost more than the passes saved, 24.4 -> 28.3 us on the bench sphere.)
// Built with -ffp-contract=off: the fp32 evaluation order below IS the specification.
#include "sr_common.h"

namespace {

constexpr int TILE = 16;
constexpr int STAGE = 32;                 // TriRecs staged in LDS at a time
constexpr int BIN_CH = 8;       // 256-triangle chunks binned per pass (2048 triangles: the sphere proxy in one pass)
constexpr float PI_F = 3.14159265359f;
constexpr float CANNY_THRESHOLD = 0.17364817766693041f;   // cos(PI*4/9)
constexpr int NON_AI_OBJ_MAP_INDEX = 2048;

// valid = 2: a triangle with one or two vertices at clip w <= 0, rasterised in homogeneous coordinates (oracle/raster_ref.c header):
// fx, fy, z then hold the nine inverse-matrix coefficients E (as float bits), iw the clip z and pad[0..2] the clip w of the vertices
struct __attribute__((aligned(16))) TriRec {
  int x0, x1, y0, y1;                     // pixel bbox (inclusive), first so the binning pass reads one int4
  int fx[3], fy[3];                       // 28.4 fixed-point window coordinates
  int sgn, tl, vid, valid;
  float farea;
  float z[3], iw[3];
  float vp[9], vn[9], uv[6], col[9];
  int pad[10];
};
static_assert(sizeof(TriRec) == 256, "TriRec must be 256 bytes");

// fp16 <-> fp32, round to nearest even, identical bit routine to the oracle
__device__ __forceinline__ uint16_t f2h(float f) {
  uint32_t x = __float_as_uint(f);
  const uint32_t sign = (x >> 16) & 0x8000u;
  x &= 0x7fffffffu;
  if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((x > 0x7f800000u) ? 0x200u : 0u));
  if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);
  if (x < 0x33000001u) return (uint16_t)sign;
  if (x < 0x38800000u) {
    const uint32_t mant = (x & 0x7fffffu) | 0x800000u;
    const int shift = 126 - (int)(x >> 23);
    uint32_t h = mant >> shift;
    const uint32_t rem = mant & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (h & 1u))) h++;
    return (uint16_t)(sign | h);
  }
  uint32_t h = (x - 0x38000000u) >> 13;
  const uint32_t rem = x & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
  return (uint16_t)(sign | h);
}
__device__ __forceinline__ float h2f(uint16_t h) {
  const uint32_t sign = ((uint32_t)h & 0x8000u) << 16, e = (h >> 10) & 0x1fu;
  uint32_t m = h & 0x3ffu, x;
  if (e == 0) {
    if (m == 0) x = sign;
    else { int s = 0; while (!(m & 0x400u)) { m <<= 1; s++; } m &= 0x3ffu; x = sign | ((uint32_t)(113 - s) << 23) | (m << 13); }
  } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
  else x = sign | ((e + 112u) << 23) | (m << 13);
  return __uint_as_float(x);
}

__device__ __forceinline__ void mat_vec(const float* M, float x, float y, float z, float w, float* o) {
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = ((M[i] * x + M[4 + i] * y) + M[8 + i] * z) + M[12 + i] * w;
}
__device__ __forceinline__ void normalize3(float* v) {
  const float l = sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
  v[0] = v[0] / l; v[1] = v[1] / l; v[2] = v[2] / l;
}
__device__ __forceinline__ int nearest_index(float t, int n) {
  const float f = t - floorf(t);
  int i = (int)(f * (float)n);
  if (i >= n) i = n - 1;
  if (i < 0) i = 0;
  return i;
}
// ---- trilinear sampling of a mip-mapped RGBA32F texture (sr_draw.diffuse_levels >= 2): statement for statement
// oracle/raster_ref.c tex_trilinear (OpenGL 4.6 section 8.14 in fp32 with a fixed operation order; lambda from the exponent of
// rho^2 and a cubic for the mantissa, so that C and HIP produce the same bits)
__device__ __forceinline__ int tex_wrap(int i, int n) { i %= n; return i < 0 ? i + n : i; }
__device__ __forceinline__ void tex_bilinear(const float* lvl, int w, int h, float s, float t, float* o) {
  const float u = s * (float)w - 0.5f, v = t * (float)h - 0.5f;
  const float fu = floorf(u), fv = floorf(v);
  const float a = u - fu, b = v - fv;
  const int i0 = tex_wrap((int)fu, w), i1 = tex_wrap(i0 + 1, w), j0 = tex_wrap((int)fv, h), j1 = tex_wrap(j0 + 1, h);
  const float4 t00 = ((const float4*)lvl)[(size_t)j0 * w + i0], t10 = ((const float4*)lvl)[(size_t)j0 * w + i1];
  const float4 t01 = ((const float4*)lvl)[(size_t)j1 * w + i0], t11 = ((const float4*)lvl)[(size_t)j1 * w + i1];
  const float c00[4] = {t00.x, t00.y, t00.z, t00.w}, c10[4] = {t10.x, t10.y, t10.z, t10.w};
  const float c01[4] = {t01.x, t01.y, t01.z, t01.w}, c11[4] = {t11.x, t11.y, t11.z, t11.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float top = c00[k] * (1.0f - a) + c10[k] * a, bot = c01[k] * (1.0f - a) + c11[k] * a;
    o[k] = top * (1.0f - b) + bot * b;
  }
}
__device__ __forceinline__ const float* tex_level(const float* tex, int w, int h, int level, int* lw, int* lh) {
  size_t off = 0;
  for (int k = 0; k < level; ++k) { off += (size_t)w * h * 4; w = w > 1 ? w >> 1 : 1; h = h > 1 ? h >> 1 : 1; }
  *lw = w; *lh = h;
  return tex + off;
}
__device__ __noinline__ void tex_trilinear(const float* tex, int w, int h, int levels, float s, float t, float rho2, float* o) {
  int lw, lh;
  s = s - floorf(s); t = t - floorf(t);
  if (!(rho2 > 1.0f) || levels <= 1) { tex_bilinear(tex, w, h, s, t, o); return; }       // magnification (and NaN): level 0
  unsigned bits = __float_as_uint(rho2);
  const int e = (int)(bits >> 23) - 127;
  bits = (bits & 0x7fffffu) | 0x3f800000u;
  const float m = __uint_as_float(bits);
  const float z = m - 1.0f;
  const float l2m = z * (1.4380732774734497f + z * (-0.6747666597366333f + z * (0.31700071692466736f + z * -0.08030730485916138f)));
  const float lam = 0.5f * ((float)e + l2m);
  const int maxl = levels - 1;
  if (!(lam < (float)maxl)) { const float* l = tex_level(tex, w, h, maxl, &lw, &lh); tex_bilinear(l, lw, lh, s, t, o); return; }
  const int d1 = (int)lam;
  const float fr = lam - (float)d1;
  float c1[4], c2[4];
  const float* l1 = tex_level(tex, w, h, d1, &lw, &lh); tex_bilinear(l1, lw, lh, s, t, c1);
  const float* l2 = tex_level(tex, w, h, d1 + 1, &lw, &lh); tex_bilinear(l2, lw, lh, s, t, c2);
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = c1[k] * (1.0f - fr) + c2[k] * fr;
}
__device__ __forceinline__ int to_fixed(float v) { return (int)floorf(v * 16.0f + 0.5f); }
__device__ __forceinline__ long long edge_fn(int ax, int ay, int bx, int by, int px, int py) {
  return (long long)(bx - ax) * (long long)(py - ay) - (long long)(by - ay) * (long long)(px - ax);
}
__device__ __forceinline__ int top_left(int dx, int dy) { return (dy < 0) || (dy == 0 && dx > 0); }

__global__ void raster_setup(const sr_draw d, TriRec* __restrict__ recs, int W, int H) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= d.nt) return;
  TriRec r;
  r.valid = 0;
  float cx[3], cy[3], cz[3], cw[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int idx = d.tris[3 * t + k];
    const float* p = d.pos + 3 * idx;
    const float* n = d.normal + 3 * idx;
    float tv[4], c[4], vn[4];
    mat_vec(d.MV, p[0], p[1], p[2], 1.0f, tv);
    r.vp[3 * k] = tv[0]; r.vp[3 * k + 1] = tv[1]; r.vp[3 * k + 2] = tv[2];
    mat_vec(d.P, tv[0], tv[1], tv[2], 1.0f, c);
    cx[k] = c[0]; cy[k] = c[1]; cz[k] = c[2]; cw[k] = c[3];
    mat_vec(d.MV_IT, n[0], n[1], n[2], 0.0f, vn);
    float v3[3] = {vn[0], vn[1], vn[2]};
    normalize3(v3);
    r.vn[3 * k] = v3[0]; r.vn[3 * k + 1] = v3[1]; r.vn[3 * k + 2] = v3[2];
    r.uv[2 * k] = d.uv ? d.uv[2 * idx] : 0.0f; r.uv[2 * k + 1] = d.uv ? d.uv[2 * idx + 1] : 0.0f;
#pragma unroll
    for (int j = 0; j < 3; ++j) r.col[3 * k + j] = d.color ? d.color[3 * idx + j] : 0.0f;
    if (k == 2) r.vid = d.vertex_id ? d.vertex_id[idx] : idx;         // flat: provoking (last) vertex
  }
  const int nfront = (cw[0] > 0.0f) + (cw[1] > 0.0f) + (cw[2] > 0.0f);
  if (nfront == 3) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      r.iw[k] = 1.0f / cw[k];
      const float nx = cx[k] * r.iw[k], ny = cy[k] * r.iw[k], nz = cz[k] * r.iw[k];
      const float sx = (nx * 0.5f + 0.5f) * (float)W;
      const float sy = (1.0f - (ny * 0.5f + 0.5f)) * (float)H;
      r.z[k] = nz * 0.5f + 0.5f;
      r.fx[k] = to_fixed(sx); r.fy[k] = to_fixed(sy);
    }
    const long long area = edge_fn(r.fx[0], r.fy[0], r.fx[1], r.fy[1], r.fx[2], r.fy[2]);
    // GL front face = visually counter-clockwise = NEGATIVE area in these y-down window coordinates
    if (area != 0 && !(area > 0 && d.cull_back)) {
      const int sgn = area > 0 ? 1 : -1;
      int minx = min(r.fx[0], min(r.fx[1], r.fx[2])), maxx = max(r.fx[0], max(r.fx[1], r.fx[2]));
      int miny = min(r.fy[0], min(r.fy[1], r.fy[2])), maxy = max(r.fy[0], max(r.fy[1], r.fy[2]));
      r.x0 = max((minx - 8 + 15) >> 4, 0); r.x1 = min((maxx - 8) >> 4, W - 1);
      r.y0 = max((miny - 8 + 15) >> 4, 0); r.y1 = min((maxy - 8) >> 4, H - 1);
      r.sgn = sgn;
      r.tl = top_left(sgn * (r.fx[2] - r.fx[1]), sgn * (r.fy[2] - r.fy[1])) |
             (top_left(sgn * (r.fx[0] - r.fx[2]), sgn * (r.fy[0] - r.fy[2])) << 1) |
             (top_left(sgn * (r.fx[1] - r.fx[0]), sgn * (r.fy[1] - r.fy[0])) << 2);
      r.farea = (float)((long long)sgn * area);
      r.valid = (r.x0 <= r.x1 && r.y0 <= r.y1) ? 1 : 0;
    }
  } else if (nfront > 0) {
    float cof[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int j = (i + 1) % 3, k = (i + 2) % 3;
      cof[3 * i] = cy[j] * cw[k] - cy[k] * cw[j];
      cof[3 * i + 1] = cx[k] * cw[j] - cx[j] * cw[k];
      cof[3 * i + 2] = cx[j] * cy[k] - cx[k] * cy[j];
    }
    const float det = (cx[0] * cof[0] + cy[0] * cof[1]) + cw[0] * cof[2];
    if (det != 0.0f && !(det < 0.0f && d.cull_back)) {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        r.fx[i] = __float_as_int(cof[i] / det);
        r.fy[i] = __float_as_int(cof[3 + i] / det);
        r.z[i] = cof[6 + i] / det;
        r.iw[i] = cz[i];
        r.pad[i] = __float_as_int(cw[i]);
      }
      r.x0 = 0; r.x1 = W - 1; r.y0 = 0; r.y1 = H - 1;
      r.sgn = 1; r.tl = 0; r.farea = 1.0f;
      r.valid = 2;
    }
  }
  recs[t] = r;
  // binning reads ONLY this: a 16-byte lane-contiguous record instead of 16 bytes out of every 256-byte TriRec (one 128-byte
  // line per lane); a culled / degenerate triangle gets a box no tile meets
  ((int4*)(recs + d.nt))[t] = r.valid ? make_int4(r.x0, r.x1, r.y0, r.y1) : make_int4(0x7fffffff, -1, 0x7fffffff, -1);
}

// TRILINEAR: the instantiation for a mip-mapped diffuse texture (sr_draw.diffuse_levels >= 2): uv derivatives + tex_trilinear cost
// 30 more VGPRs, which the common instantiation (3 waves per SIMD at 170) must not pay
template <bool TRILINEAR>
__global__ __launch_bounds__(256) void raster_tiles(const sr_draw d, const sr_gbuffer g, const TriRec* __restrict__ recs) {
  __shared__ TriRec srec[STAGE];
  __shared__ int sbin[BIN_CH * 256];
  __shared__ int swcnt[BIN_CH * 4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int W = g.W, H = g.H;
  const int tiles_x = (W + TILE - 1) / TILE;
  const int tile_x = blockIdx.x % tiles_x, tile_y = blockIdx.x / tiles_x;
  const int tx0 = tile_x * TILE, ty0 = tile_y * TILE;
  const int x = tx0 + (tid & 15), y = ty0 + (tid >> 4);
  const bool inimg = x < W && y < H;
  const size_t pi = (size_t)y * W + x;
  const int px = x * 16 + 8, py = y * 16 + 8;

  // per-pixel state
  float zcur = 1.0f;
  if (inimg && d.depth_test) zcur = g.zbuf[pi];
  bool touched = false;
  float curColor[4], curND[4], curNoise[4], curPos[3], curCanny[3];
  int curID[4];
  float oColor[4], oND[4], oNoise[4], oPos[3], oCanny[3];
  int oID[4];

  // pre-draw snapshot of this pixel (what the fragment shader blends against): loaded up front so its latency overlaps the
  // binning pass instead of stalling the first covering triangle
  if (inimg) {
    const ushort4 c = ((const ushort4*)g.color)[pi], nd = ((const ushort4*)g.normal_depth)[pi], nz = ((const ushort4*)g.noise)[pi];
    curColor[0] = h2f(c.x); curColor[1] = h2f(c.y); curColor[2] = h2f(c.z); curColor[3] = h2f(c.w);
    curND[0] = h2f(nd.x); curND[1] = h2f(nd.y); curND[2] = h2f(nd.z); curND[3] = h2f(nd.w);
    curNoise[0] = h2f(nz.x); curNoise[1] = h2f(nz.y); curNoise[2] = h2f(nz.z); curNoise[3] = h2f(nz.w);
    const int4 iv = ((const int4*)g.id)[pi];
    curID[0] = iv.x; curID[1] = iv.y; curID[2] = iv.z; curID[3] = iv.w;
#pragma unroll
    for (int k = 0; k < 3; ++k) { curPos[k] = g.pos[pi * 3 + k]; curCanny[k] = g.canny[pi * 3 + k]; }
  }
  // ---- bin: which triangles touch this tile.  Up to BIN_CH x 256 triangles per pass: ALL their bounding-box loads are issued
  // before the first is used (one memory latency per pass instead of one per 256 triangles), then an order-preserving
  // compaction over the whole pass (primitive order is the blend order of the TRANSPARENT queue) with two barriers in all.
  for (int c0 = 0; c0 < d.nt; c0 += BIN_CH * 256) {
    bool hit[BIN_CH];
    int4 bb[BIN_CH];
    int valid[BIN_CH];
#pragma unroll
    for (int c = 0; c < BIN_CH; ++c) {
      const int ti = c0 + c * 256 + tid;
      bb[c] = make_int4(1, 0, 1, 0);
      valid[c] = 0;
      if (ti < d.nt) {
        bb[c] = ((const int4*)(recs + d.nt))[ti];           // x0,x1,y0,y1 (raster_setup's dense copy)
        valid[c] = 1;
      }
    }
    unsigned long long m[BIN_CH];
#pragma unroll
    for (int c = 0; c < BIN_CH; ++c) {
      hit[c] = valid[c] && bb[c].x <= tx0 + TILE - 1 && bb[c].y >= tx0 && bb[c].z <= ty0 + TILE - 1 && bb[c].w >= ty0;
      m[c] = __ballot(hit[c]);
      if (lane == 0) swcnt[c * 4 + wv] = __popcll(m[c]);
    }
    __syncthreads();
    int total = 0, run = 0;
#pragma unroll
    for (int c = 0; c < BIN_CH; ++c)
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const int n = swcnt[c * 4 + w];
        if (hit[c] && w == wv) sbin[run + __popcll(m[c] & ((1ull << lane) - 1ull))] = c0 + c * 256 + tid;
        run += n;
      }
    total = run;
    __syncthreads();
    // ---- process binned triangles in order, STAGE records at a time through LDS
    for (int s0 = 0; s0 < total; s0 += STAGE) {
      const int ns = min(STAGE, total - s0);
      for (int i = tid; i < ns * 64; i += 256) {
        const int r = i >> 6, wd = i & 63;
        ((int*)&srec[r])[wd] = ((const int*)&recs[sbin[s0 + r]])[wd];
      }
      __syncthreads();
      if (inimg) {
        for (int j = 0; j < ns; ++j) {
          const TriRec& T = srec[j];
          if (x < T.x0 || x > T.x1 || y < T.y0 || y > T.y1) continue;
          float f0, f1, f2, zf;
          if (T.valid != 2) {
            const long long w0 = (long long)T.sgn * edge_fn(T.fx[1], T.fy[1], T.fx[2], T.fy[2], px, py);
            const long long w1 = (long long)T.sgn * edge_fn(T.fx[2], T.fy[2], T.fx[0], T.fy[0], px, py);
            const long long w2 = (long long)T.sgn * edge_fn(T.fx[0], T.fy[0], T.fx[1], T.fy[1], px, py);
            if (w0 < 0 || w1 < 0 || w2 < 0) continue;
            if ((w0 == 0 && !(T.tl & 1)) || (w1 == 0 && !(T.tl & 2)) || (w2 == 0 && !(T.tl & 4))) continue;
            const float b0 = (float)w0 / T.farea, b1 = (float)w1 / T.farea, b2 = (float)w2 / T.farea;
            zf = (T.z[0] * b0 + T.z[1] * b1) + T.z[2] * b2;
            f0 = b0 * T.iw[0]; f1 = b1 * T.iw[1]; f2 = b2 * T.iw[2];
          } else {
            const float X = (((float)x + 0.5f) / (float)W) * 2.0f - 1.0f;
            const float Y = 1.0f - (((float)y + 0.5f) / (float)H) * 2.0f;
            f0 = (__int_as_float(T.fx[0]) * X + __int_as_float(T.fx[1]) * Y) + __int_as_float(T.fx[2]);
            f1 = (__int_as_float(T.fy[0]) * X + __int_as_float(T.fy[1]) * Y) + __int_as_float(T.fy[2]);
            f2 = (T.z[0] * X + T.z[1] * Y) + T.z[2];
            if (!(f0 >= 0.0f && f1 >= 0.0f && f2 >= 0.0f)) continue;
            const float es = (f0 + f1) + f2;
            if (!(es > 0.0f)) continue;
            const float zc = (T.iw[0] * f0 + T.iw[1] * f1) + T.iw[2] * f2;
            const float wc = (__int_as_float(T.pad[0]) * f0 + __int_as_float(T.pad[1]) * f1) + __int_as_float(T.pad[2]) * f2;
            zf = (zc / wc) * 0.5f + 0.5f;
          }
          if (!(zf >= 0.0f && zf <= 1.0f)) continue;               // near / far clip
          if (d.depth_test) { if (!(zf < zcur)) continue; }
          const float fs = (f0 + f1) + f2;
#define INTERP(a0, a1, a2) ((((a0) * f0 + (a1) * f1) + (a2) * f2) / fs)
          float vp[3], vn[3], uv[2], vc[3];
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            vp[k] = INTERP(T.vp[k], T.vp[3 + k], T.vp[6 + k]);
            vn[k] = INTERP(T.vn[k], T.vn[3 + k], T.vn[6 + k]);
            vc[k] = INTERP(T.col[k], T.col[3 + k], T.col[6 + k]);
          }
#pragma unroll
          for (int k = 0; k < 2; ++k) uv[k] = INTERP(T.uv[k], T.uv[2 + k], T.uv[4 + k]);
#undef INTERP
          // screen-space derivatives of uv for the mip level (oracle/raster_ref.c: the same triangle's perspective-correct uv one
          // pixel right and one pixel down); only for a mip-mapped diffuse texture
          float rho2 = 0.0f;
          if constexpr (TRILINEAR) {
            float duv[2][2];
#pragma unroll
            for (int ax = 0; ax < 2; ++ax) {
              const int xx = x + (ax == 0), yy = y + (ax == 1);
              float g0, g1, g2;
              if (T.valid != 2) {
                const int qx = xx * 16 + 8, qy = yy * 16 + 8;
                const long long u0 = (long long)T.sgn * edge_fn(T.fx[1], T.fy[1], T.fx[2], T.fy[2], qx, qy);
                const long long u1 = (long long)T.sgn * edge_fn(T.fx[2], T.fy[2], T.fx[0], T.fy[0], qx, qy);
                const long long u2 = (long long)T.sgn * edge_fn(T.fx[0], T.fy[0], T.fx[1], T.fy[1], qx, qy);
                g0 = ((float)u0 / T.farea) * T.iw[0]; g1 = ((float)u1 / T.farea) * T.iw[1]; g2 = ((float)u2 / T.farea) * T.iw[2];
              } else {
                const float X = (((float)xx + 0.5f) / (float)W) * 2.0f - 1.0f;
                const float Y = 1.0f - (((float)yy + 0.5f) / (float)H) * 2.0f;
                g0 = (__int_as_float(T.fx[0]) * X + __int_as_float(T.fx[1]) * Y) + __int_as_float(T.fx[2]);
                g1 = (__int_as_float(T.fy[0]) * X + __int_as_float(T.fy[1]) * Y) + __int_as_float(T.fy[2]);
                g2 = (T.z[0] * X + T.z[1] * Y) + T.z[2];
              }
              const float gs = (g0 + g1) + g2;
#pragma unroll
              for (int k = 0; k < 2; ++k) duv[ax][k] = (((T.uv[k] * g0 + T.uv[2 + k] * g1) + T.uv[4 + k] * g2) / gs) - uv[k];
            }
            const float ux = duv[0][0] * (float)d.diffuse_w, vx = duv[0][1] * (float)d.diffuse_h;
            const float uy = duv[1][0] * (float)d.diffuse_w, vy = duv[1][1] * (float)d.diffuse_h;
            const float rx = ux * ux + vx * vx, ry = uy * uy + vy * vy;
            rho2 = rx > ry ? rx : ry;
          }
          // ---------------- fragment shader ----------------
          float outNoise[4] = {0.f, 0.f, 0.f, 0.f};
          if (d.noise_tex) {
            const int tx = nearest_index(uv[0], d.noise_w), ty = nearest_index(uv[1], d.noise_h);
            const ushort4 nv = ((const ushort4*)d.noise_tex)[(size_t)ty * d.noise_w + tx];
            outNoise[0] = h2f(nv.x); outNoise[1] = h2f(nv.y); outNoise[2] = h2f(nv.z); outNoise[3] = h2f(nv.w);
          }
          const float depth = 1.0f - zf;
          float n[3] = {vn[0], vn[1], vn[2]};
          normalize3(n);
          if (d.normal_tex && d.tangent && d.bitangent) {
            // TBN normal-map branch (frag.glsl:118-122): the three extra varyings are fetched per covering fragment from the
            // vertex arrays (L2 resident) instead of widening every TriRec for a branch few materials take
            const int tri = sbin[s0 + j];
            float mt[3], mb[3], mn[3];
            float T3[3][3], B3[3][3], N3[3][3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              const int idx = d.tris[3 * tri + k];
#pragma unroll
              for (int c = 0; c < 3; ++c) { T3[k][c] = d.tangent[3 * idx + c]; B3[k][c] = d.bitangent[3 * idx + c]; N3[k][c] = d.normal[3 * idx + c]; }
              normalize3(T3[k]); normalize3(B3[k]);
            }
#define INTERP2(a0, a1, a2) ((((a0) * f0 + (a1) * f1) + (a2) * f2) / fs)
#pragma unroll
            for (int c = 0; c < 3; ++c) { mt[c] = INTERP2(T3[0][c], T3[1][c], T3[2][c]); mb[c] = INTERP2(B3[0][c], B3[1][c], B3[2][c]); mn[c] = INTERP2(N3[0][c], N3[1][c], N3[2][c]); }
#undef INTERP2
            const int ntx = nearest_index(uv[0], d.normal_w), nty = nearest_index(uv[1], d.normal_h);
            const float4 tp = ((const float4*)d.normal_tex)[(size_t)nty * d.normal_w + ntx];
            float c3[3] = {tp.x * 2.0f - 1.0f, tp.y * 2.0f - 1.0f, tp.z * 2.0f - 1.0f};
            normalize3(c3);
            float m3[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) m3[c] = (mt[c] * c3[0] + mb[c] * c3[1]) + mn[c] * c3[2];
            normalize3(m3);
            float v4[4];
            mat_vec(d.MV_IT, m3[0], m3[1], m3[2], 0.0f, v4);
            n[0] = v4[0]; n[1] = v4[1]; n[2] = v4[2];
            normalize3(n);
          }
          float outND[4] = {n[0] * 0.5f + 0.5f, n[1] * 0.5f + 0.5f, n[2] * 0.5f + 0.5f, depth};
          int real_vid;
          if (!d.use_texcoord_id) real_vid = T.vid;
          else real_vid = (int)((uv[1] * (float)d.id_h) * (float)d.id_w + uv[0] * (float)d.id_w);
          int outID[4];
          int map_index = NON_AI_OBJ_MAP_INDEX;
          if (d.render_mode == 0) { outID[0] = d.sprite_id; outID[1] = d.material_id; outID[2] = NON_AI_OBJ_MAP_INDEX; outID[3] = real_vid; }
          else {
            const int k = d.corrmap_k;
            const float l1 = sqrtf((0.0f * 0.0f + n[1] * n[1]) + n[2] * n[2]);
            float theta = (l1 == 0.0f) ? 0.0f : n[1] / l1;
            theta = PI_F / 2.0f - theta;
            const float l2 = sqrtf((n[0] * n[0] + 0.0f * 0.0f) + n[2] * n[2]);
            float phi = (l2 == 0.0f) ? 0.0f : n[0] / l2;
            phi = PI_F / 2.0f - phi;
            const float step = PI_F / (float)k;
            int xi = (int)(theta / step), yi = (int)(phi / step);
            xi = min(max(xi, 0), k - 1); yi = min(max(yi, 0), k - 1);
            map_index = xi + (k - 1 - yi) * k;
            outID[0] = d.sprite_id; outID[1] = d.material_id; outID[2] = map_index; outID[3] = real_vid;
          }
          float outColor[4];
          auto sample_diffuse = [&]() {
            if constexpr (TRILINEAR) { tex_trilinear((const float*)d.diffuse_tex, d.diffuse_w, d.diffuse_h, d.diffuse_levels, uv[0], uv[1], rho2, outColor); return; }
            const int tx = nearest_index(uv[0], d.diffuse_w), ty = nearest_index(uv[1], d.diffuse_h);
            const float4 t4 = ((const float4*)d.diffuse_tex)[(size_t)ty * d.diffuse_w + tx];
            outColor[0] = t4.x; outColor[1] = t4.y; outColor[2] = t4.z; outColor[3] = t4.w;
          };
          if (d.render_mode == 0) {
            if (!d.diffuse_tex) {
              if (d.has_vertex_color) { outColor[0] = vc[0]; outColor[1] = vc[1]; outColor[2] = vc[2]; outColor[3] = 1.0f; }
              else { outColor[0] = outColor[1] = outColor[2] = outColor[3] = 0.0f; }
            } else sample_diffuse();
          } else if (d.render_mode == 2) { outColor[0] = outColor[1] = outColor[2] = outColor[3] = 0.0f; }
          else {
            if (d.corrmap_tex) {
              const int tx = nearest_index(uv[1], d.corr_w), ty = nearest_index(uv[0], d.corr_h);
              const ushort4 cv = ((const ushort4*)d.corrmap_tex)[((size_t)map_index * d.corr_h + ty) * d.corr_w + tx];
              outColor[0] = h2f(cv.x); outColor[1] = h2f(cv.y); outColor[2] = h2f(cv.z); outColor[3] = h2f(cv.w);
            } else if (!d.diffuse_tex) {
              if (d.has_vertex_color) { outColor[0] = vc[0]; outColor[1] = vc[1]; outColor[2] = vc[2]; outColor[3] = 1.0f; }
              else { outColor[0] = 1.0f; outColor[1] = 0.0f; outColor[2] = 1.0f; outColor[3] = 1.0f; }
            } else sample_diffuse();
          }
          const float cn = (n[2] < CANNY_THRESHOLD && n[2] > 0.0f) ? 1.0f : 0.0f;
          float outPos[3] = {vp[0], vp[1], vp[2]}, outCanny[3] = {cn, cn, cn};
          if (d.render_mode == 2 || (outColor[3] == 0.0f && d.render_mode == 1)) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { outColor[k] = curColor[k]; outND[k] = curND[k]; }
            if (d.render_mode == 1) { outID[0] = curID[0]; outID[1] = curID[1]; outID[2] = curID[2]; outID[3] = curID[3]; }
#pragma unroll
            for (int k = 0; k < 3; ++k) { outPos[k] = curPos[k]; outCanny[k] = curCanny[k]; }
          } else if (outColor[3] < 1.0f) {
            const float latest_depth = curND[3];
            const float nsum = ((curNoise[0] + curNoise[1]) + curNoise[2]) + curNoise[3];
            const float a = outColor[3];
            if (latest_depth < depth) {
#pragma unroll
              for (int k = 0; k < 3; ++k) outColor[k] = outColor[k] * a + curColor[k] * (1.0f - a);
              if (nsum > 0.001f) {
#pragma unroll
                for (int k = 0; k < 4; ++k) outNoise[k] = outNoise[k] * a + curNoise[k] * (1.0f - a);
              }
            } else {
              const float ca = curColor[3];
#pragma unroll
              for (int k = 0; k < 3; ++k) outColor[k] = curColor[k] * ca + outColor[k] * (1.0f - ca);
              outColor[3] = ca;
              if (nsum > 0.001f) {
#pragma unroll
                for (int k = 0; k < 4; ++k) outNoise[k] = curNoise[k] * ca + outNoise[k] * (1.0f - ca);
              }
              outND[3] = latest_depth;
            }
          }
          if (d.depth_test) zcur = zf;
          touched = true;
#pragma unroll
          for (int k = 0; k < 4; ++k) { oColor[k] = outColor[k]; oND[k] = outND[k]; oNoise[k] = outNoise[k]; oID[k] = outID[k]; }
#pragma unroll
          for (int k = 0; k < 3; ++k) { oPos[k] = outPos[k]; oCanny[k] = outCanny[k]; }
        }
      }
      __syncthreads();
    }
  }
  if (inimg && touched) {
    if (d.depth_test) g.zbuf[pi] = zcur;
    ((ushort4*)g.color)[pi] = make_ushort4(f2h(oColor[0]), f2h(oColor[1]), f2h(oColor[2]), f2h(oColor[3]));
    ((ushort4*)g.normal_depth)[pi] = make_ushort4(f2h(oND[0]), f2h(oND[1]), f2h(oND[2]), f2h(oND[3]));
    ((ushort4*)g.noise)[pi] = make_ushort4(f2h(oNoise[0]), f2h(oNoise[1]), f2h(oNoise[2]), f2h(oNoise[3]));
    ((int4*)g.id)[pi] = make_int4(oID[0], oID[1], oID[2], oID[3]);
#pragma unroll
    for (int k = 0; k < 3; ++k) { g.pos[pi * 3 + k] = oPos[k]; g.canny[pi * 3 + k] = oCanny[k]; }
  }
}

__global__ void gbuffer_clear_kernel(const sr_gbuffer g) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n = (size_t)g.W * g.H;
  if (i >= n) return;
  ((ushort4*)g.color)[i] = make_ushort4(0, 0, 0, 0);
  ((ushort4*)g.normal_depth)[i] = make_ushort4(0, 0, 0, 0);
  ((ushort4*)g.noise)[i] = make_ushort4(0, 0, 0, 0);
  ((int4*)g.id)[i] = make_int4(0, 0, 0, 0);
  for (int k = 0; k < 3; ++k) { g.pos[i * 3 + k] = 0.f; g.canny[i * 3 + k] = 0.f; }
  g.zbuf[i] = 1.0f;
}

}  // namespace

extern "C" int64_t sr_raster_scratch_bytes(int32_t nt, int32_t, int32_t) { return (int64_t)nt * (int64_t)(sizeof(TriRec) + sizeof(int4)); }

// ---- identical-G-buffer merge (renderManager.py:118-133): an object drawn ALONE into a cleared G-buffer is folded into the
// accumulated planes wherever its depth (normal_depth.a = 1 - window z: closer = larger, compared in fp16 as stored) beats theirs
__global__ void gbuffer_depth_merge_kernel(sr_gbuffer acc, const sr_gbuffer src, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const ushort4 s_nd = ((const ushort4*)src.normal_depth)[i];
  const ushort4 a_nd = ((const ushort4*)acc.normal_depth)[i];
  if (!(h2f(s_nd.w) > h2f(a_nd.w))) return;
  ((ushort4*)acc.normal_depth)[i] = s_nd;
  ((ushort4*)acc.color)[i] = ((const ushort4*)src.color)[i];
  ((int4*)acc.id)[i] = ((const int4*)src.id)[i];
  ((ushort4*)acc.noise)[i] = ((const ushort4*)src.noise)[i];
#pragma unroll
  for (int k = 0; k < 3; ++k) { acc.pos[(size_t)i * 3 + k] = src.pos[(size_t)i * 3 + k]; acc.canny[(size_t)i * 3 + k] = src.canny[(size_t)i * 3 + k]; }
  if (acc.zbuf && src.zbuf) acc.zbuf[i] = src.zbuf[i];
}

// ---- defer pass + post process (default_defer_render.frag.glsl:20-59, default_post_process.frag.glsl:21-39): the display image.
// Baking: AI-object pixels (any id set, map_index != 2048) are tinted 10 % with a rainbow of the vertex id; then gamma, exposure,
// saturation, brightness, contrast, optional HDR tone map.  out RGBA fp32.
__global__ void defer_post_kernel(const uint16_t* __restrict__ color, const int32_t* __restrict__ ids, float* __restrict__ out, int n,
                                  int is_baking, int gamma_on, int hdr_on, float gamma, float exposure, float saturation,
                                  float brightness, float contrast) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const ushort4 c = ((const ushort4*)color)[i];
  float rgb[3] = {h2f(c.x), h2f(c.y), h2f(c.z)};
  float a = h2f(c.w);
  if (is_baking) {
    const int4 id = ((const int4*)ids)[i];
    if (id.x + id.y + id.z + id.w > 0 && id.z != NON_AI_OBJ_MAP_INDEX) {
      float ratio = (float)id.w / (float)(512 * 512);
      ratio = 1.0f - fminf(fmaxf(ratio, 0.0f), 1.0f);
      float col[3] = {1.0f, 1.0f, 1.0f};
      if (ratio < 1.0f / 6.0f) { col[0] = 1.0f; col[1] = ratio * 6.0f; col[2] = 0.0f; }
      else if (ratio < 2.0f / 6.0f) { col[0] = 1.0f - (ratio - 1.0f / 6.0f) * 6.0f; col[1] = 1.0f; col[2] = 0.0f; }
      else if (ratio < 3.0f / 6.0f) { col[0] = 0.0f; col[1] = 1.0f; col[2] = (ratio - 2.0f / 6.0f) * 6.0f; }
      else if (ratio < 4.0f / 6.0f) { col[0] = 0.0f; col[1] = 1.0f - (ratio - 3.0f / 6.0f) * 6.0f; col[2] = 1.0f; }
      else if (ratio < 5.0f / 6.0f) { col[0] = (ratio - 4.0f / 6.0f) * 6.0f; col[1] = 0.0f; col[2] = 1.0f; }
      else { col[0] = 1.0f; col[1] = 0.0f; col[2] = 1.0f - (ratio - 5.0f / 6.0f) * 6.0f; }
#pragma unroll
      for (int k = 0; k < 3; ++k) rgb[k] = rgb[k] * (1.0f - 0.1f) + col[k] * 0.1f;      // mix(FragColor.rgb, col, 0.1)
      a = 1.0f;
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float v = rgb[k];
    if (gamma_on) v = powf(v, 1.0f / gamma);
    v = v * exposure;
    v = 0.5f * (1.0f - saturation) + v * saturation;                                     // mix(vec3(0.5), rgb, saturation)
    v = v * brightness;
    v = (v - 0.5f) * contrast + 0.5f;
    if (hdr_on) v = v / (v + 1.0f);
    out[(size_t)i * 4 + k] = v;
  }
  out[(size_t)i * 4 + 3] = a;
}

extern "C" int sr_gbuffer_depth_merge(const sr_gbuffer* acc, const sr_gbuffer* src, void* stream) {
  if (!acc || !src || acc->W != src->W || acc->H != src->H) SR_FAIL(SR_ERR_INVALID, "sr_gbuffer_depth_merge: bad args");
  const int n = acc->W * acc->H;
  hipLaunchKernelGGL(gbuffer_depth_merge_kernel, dim3((n + 255) / 256), dim3(256), 0, sr_stream(stream), *acc, *src, n);
  SR_CHECK_LAUNCH("sr_gbuffer_depth_merge");
  return SR_OK;
}

extern "C" int sr_defer_post(const void* color_rgba16f, const int32_t* ids, float* out_rgba, int32_t W, int32_t H, int32_t is_baking,
                             int32_t enable_gamma, int32_t enable_hdr, float gamma, float exposure, float saturation, float brightness,
                             float contrast, void* stream) {
  if (!color_rgba16f || !out_rgba || (is_baking && !ids) || W < 1 || H < 1) SR_FAIL(SR_ERR_INVALID, "sr_defer_post: bad args");
  const int n = W * H;
  hipLaunchKernelGGL(defer_post_kernel, dim3((n + 255) / 256), dim3(256), 0, sr_stream(stream), (const uint16_t*)color_rgba16f, ids, out_rgba,
                     n, is_baking, enable_gamma, enable_hdr, gamma, exposure, saturation, brightness, contrast);
  SR_CHECK_LAUNCH("sr_defer_post");
  return SR_OK;
}

extern "C" int sr_gbuffer_clear(const sr_gbuffer* g, void* stream) {
  if (!g || !g->color || !g->id || !g->pos || !g->normal_depth || !g->noise || !g->canny || !g->zbuf) SR_FAIL(SR_ERR_INVALID, "sr_gbuffer_clear: null plane");
  const size_t n = (size_t)g->W * g->H;
  hipLaunchKernelGGL(gbuffer_clear_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, sr_stream(stream), *g);
  SR_CHECK_LAUNCH("sr_gbuffer_clear");
  return SR_OK;
}

extern "C" int sr_raster_draw(const sr_draw* d, const sr_gbuffer* g, void* scratch, int64_t scratch_bytes, void* stream) {
  if (!d || !g || !scratch) SR_FAIL(SR_ERR_INVALID, "sr_raster_draw: null");
  if (!d->pos || !d->normal || !d->tris || d->nt <= 0 || d->nv <= 0) SR_FAIL(SR_ERR_INVALID, "sr_raster_draw: mesh arrays missing");
  if (scratch_bytes < sr_raster_scratch_bytes(d->nt, g->W, g->H)) SR_FAIL(SR_ERR_INVALID, "sr_raster_draw: scratch too small");
  if (d->render_mode != 0 && d->corrmap_k <= 0) SR_FAIL(SR_ERR_INVALID, "sr_raster_draw: corrmap_k");
  if (d->use_texcoord_id && (d->id_w <= 0 || d->id_h < 0)) SR_FAIL(SR_ERR_INVALID, "sr_raster_draw: id grid size");
  if (d->normal_tex && (!d->tangent || !d->bitangent || !d->uv || d->normal_w <= 0 || d->normal_h <= 0))
    SR_FAIL(SR_ERR_INVALID, "sr_raster_draw: a normal map needs uvs, tangents and bitangents");
  hipStream_t st = sr_stream(stream);
  TriRec* recs = (TriRec*)scratch;
  hipLaunchKernelGGL(raster_setup, dim3((d->nt + 255) / 256), dim3(256), 0, st, *d, recs, g->W, g->H);
  const int tiles = ((g->W + TILE - 1) / TILE) * ((g->H + TILE - 1) / TILE);
  if (d->diffuse_tex && d->diffuse_levels >= 2) {
    if (d->diffuse_levels > 31 || d->diffuse_w <= 0 || d->diffuse_h <= 0) SR_FAIL(SR_ERR_INVALID, "sr_raster_draw: diffuse mip chain");
    hipLaunchKernelGGL(raster_tiles<true>, dim3(tiles), dim3(256), 0, st, *d, *g, recs);
  } else hipLaunchKernelGGL(raster_tiles<false>, dim3(tiles), dim3(256), 0, st, *d, *g, recs);
  SR_CHECK_LAUNCH("sr_raster_draw");
  return SR_OK;
}
