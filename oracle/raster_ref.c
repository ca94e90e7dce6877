/* ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar C restatement of the reference's G-buffer pass:
 *   vertex stage   engine/shaders/default_Gbuffer.vert.glsl:40-57
 *   fragment stage engine/shaders/default_Gbuffer.frag.glsl:100-257
 *   per-draw snapshot semantics ("current*" samplers) engine/managers/renderManager.py:544-571
 *   depth test on for OPAQUE / off for TRANSPARENT tasks renderManager.py:508-513
 *
 * PARITY UNPINNED vs a real OpenGL driver: the reference rasterises with the GL driver of the author's GPU, which
 * cannot run here (no GL context, no GPU) and whose coverage/interpolation is implementation defined anyway; no
 * pre-dumped G-buffer in /root/reference carries the scene that produced it.  This file therefore
 * DEFINES the rasterisation rule the HIP rasterizer must reproduce bit for bit:
 *   - vertices to 28.4 fixed point (round half away from zero via floorf(x*16+0.5)), integer edge functions, top-left
 *     fill rule, pixel centres at (x+0.5, y+0.5), row 0 = top of the image (the reference flips GL rows on read-back);
 *   - near / far clipping per fragment: a fragment whose window depth leaves [0, 1] is discarded (what GL's clip against the
 *     near and far planes leaves of a triangle); a triangle with one or two vertices at clip w <= 0 (behind the eye, where the
 *     perspective divide is meaningless) is rasterised in homogeneous coordinates -- edge functions from the inverse of the
 *     clip-space (x, y, w) matrix evaluated at the pixel's NDC centre (Olano & Greer), inclusive edges, the same
 *     perspective-correct interpolation -- over the whole viewport; all three at w <= 0: dropped.  No GL output exists for
 *     such triangles in the reference (its scenes keep geometry in front of the camera): this path is defined here;
 *   - barycentrics b_i = (float)E_i / (float)area, perspective-correct attributes sum(a_i b_i/w_i) / sum(b_i/w_i)
 *     in the association order written below, window depth z = sum(z_i b_i), GL_LESS against a 1.0-cleared buffer;
 *   - flat vertexID = last vertex of the triangle (GL provoking vertex), textures sampled NEAREST with REPEAT (a diffuse
 *     texture handed over with its mip chain, diffuse_levels >= 2, is sampled trilinear: tex_trilinear below);
 *   - fp32 only, no fused multiply-add (built with -ffp-contract=off), IEEE sqrt/div, fp16 stores round-to-nearest-even.
 * Triangles are processed in index order; every fragment of a draw sees the G-buffer as it was BEFORE the draw
 * (the reference copies all six targets after each draw task), and the last passing fragment of a pixel wins.
 *
 * Build: make -C oracle   (gcc, outputs oracle/_build/libraster_ref.so)
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define PI_F 3.14159265359f
#define NON_AI_OBJ_MAP_INDEX 2048
#define CANNY_THRESHOLD 0.17364817766693041f /* cos(PI*4/9) */

typedef struct {
  const float* pos; const float* normal; const float* uv; const float* color; const int32_t* vertex_id;
  const int32_t* tris; int32_t nv, nt;
  float MV[16], MV_IT[16], P[16];
  int32_t sprite_id, material_id, corrmap_k, use_texcoord_id, render_mode, has_vertex_color, depth_test, cull_back;
  int32_t id_w, id_h;
  const uint16_t* noise_tex; int32_t noise_w, noise_h;
  const float* diffuse_tex; int32_t diffuse_w, diffuse_h;
  const uint16_t* corrmap_tex; int32_t corr_w, corr_h;
  const float* tangent; const float* bitangent;              /* per-vertex, for the TBN normal-map branch (frag:118-122) */
  const float* normal_tex; int32_t normal_w, normal_h;       /* RGBA32F */
  int32_t diffuse_levels;                                    /* >= 2: mip chain behind level 0, sampled trilinear (see tex_trilinear) */
} ref_draw;

typedef struct {
  uint16_t* color; int32_t* id; float* pos; uint16_t* normal_depth; uint16_t* noise; float* canny; float* zbuf;
  int32_t W, H;
} ref_gbuffer;

/* ---- fp16 <-> fp32 (round to nearest even), bit exact on every platform ---- */
static uint16_t f2h(float f) {
  uint32_t x; memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u;
  x &= 0x7fffffffu;
  if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((x > 0x7f800000u) ? 0x200u : 0u));
  if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                 /* rounds to inf */
  if (x < 0x33000001u) return (uint16_t)sign;                               /* rounds to zero */
  if (x < 0x38800000u) {                                                    /* subnormal half */
    uint32_t mant = (x & 0x7fffffu) | 0x800000u;
    int shift = 126 - (int)(x >> 23);                                       /* 14..24 */
    uint32_t h = mant >> shift;
    uint32_t rem = mant & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (h & 1u))) h++;
    return (uint16_t)(sign | h);
  }
  uint32_t h = ((x - 0x38000000u) >> 13);
  uint32_t rem = x & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
  return (uint16_t)(sign | h);
}
static float h2f(uint16_t h) {
  uint32_t sign = ((uint32_t)h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu, x;
  if (e == 0) {
    if (m == 0) x = sign;
    else { int s = 0; while (!(m & 0x400u)) { m <<= 1; s++; } m &= 0x3ffu; x = sign | ((uint32_t)(113 - s) << 23) | (m << 13); }
  } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
  else x = sign | ((e + 112u) << 23) | (m << 13);
  float f; memcpy(&f, &x, 4); return f;
}

static void mat_vec(const float* M, float x, float y, float z, float w, float* o) { /* column-major, fixed order */
  for (int i = 0; i < 4; ++i) o[i] = ((M[i] * x + M[4 + i] * y) + M[8 + i] * z) + M[12 + i] * w;
}
static void normalize3(float* v) {
  float l = sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
  v[0] = v[0] / l; v[1] = v[1] / l; v[2] = v[2] / l;
}
static int nearest_index(float t, int n) {        /* REPEAT + NEAREST */
  float f = t - floorf(t);
  int i = (int)(f * (float)n);
  if (i >= n) i = n - 1;
  if (i < 0) i = 0;
  return i;
}

/* ---- trilinear sampling of a mip-mapped RGBA32F texture (the reference's file textures: GL_LINEAR_MIPMAP_LINEAR / GL_LINEAR,
 * GL_REPEAT, engine/static/texture/texture.py:57-60, 276-289; anisotropy not restated).  OpenGL 4.6 section 8.14 in fp32 with a
 * fixed operation order: levels are stored one behind the other (level k = max(1, w >> k) x max(1, h >> k) texels, box-filtered
 * by the host), rho^2 = max(|d(u,v)/dx|^2, |d(u,v)/dy|^2) in level-0 texels, lambda = log2(rho) from the exponent of rho^2 and
 * a cubic for the mantissa (|err| < 1.5e-4: far below the 8-bit LOD fraction of GL hardware, and the same bits in C and HIP,
 * which libm's log2f would not give). */
static int tex_wrap(int i, int n) { i %= n; return i < 0 ? i + n : i; }
static void tex_bilinear(const float* lvl, int w, int h, float s, float t, float* o) {
  const float u = s * (float)w - 0.5f, v = t * (float)h - 0.5f;
  const float fu = floorf(u), fv = floorf(v);
  const float a = u - fu, b = v - fv;
  const int i0 = tex_wrap((int)fu, w), i1 = tex_wrap(i0 + 1, w), j0 = tex_wrap((int)fv, h), j1 = tex_wrap(j0 + 1, h);
  const float* t00 = lvl + ((size_t)j0 * w + i0) * 4; const float* t10 = lvl + ((size_t)j0 * w + i1) * 4;
  const float* t01 = lvl + ((size_t)j1 * w + i0) * 4; const float* t11 = lvl + ((size_t)j1 * w + i1) * 4;
  for (int k = 0; k < 4; ++k) {
    const float top = t00[k] * (1.0f - a) + t10[k] * a, bot = t01[k] * (1.0f - a) + t11[k] * a;
    o[k] = top * (1.0f - b) + bot * b;
  }
}
static const float* tex_level(const float* tex, int w, int h, int level, int* lw, int* lh) {
  size_t off = 0;
  for (int k = 0; k < level; ++k) { off += (size_t)w * h * 4; w = w > 1 ? w >> 1 : 1; h = h > 1 ? h >> 1 : 1; }
  *lw = w; *lh = h;
  return tex + off;
}
static void tex_trilinear(const float* tex, int w, int h, int levels, float s, float t, float rho2, float* o) {
  int lw, lh;
  s = s - floorf(s); t = t - floorf(t);
  if (!(rho2 > 1.0f) || levels <= 1) { tex_bilinear(tex, w, h, s, t, o); return; }       /* magnification (and NaN): level 0 */
  uint32_t bits; memcpy(&bits, &rho2, 4);
  const int e = (int)(bits >> 23) - 127;
  bits = (bits & 0x7fffffu) | 0x3f800000u;
  float m; memcpy(&m, &bits, 4);
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
  for (int k = 0; k < 4; ++k) o[k] = c1[k] * (1.0f - fr) + c2[k] * fr;
}

void ref_gbuffer_clear(ref_gbuffer* g) {
  size_t n = (size_t)g->W * g->H;
  memset(g->color, 0, n * 8); memset(g->id, 0, n * 16); memset(g->pos, 0, n * 12);
  memset(g->normal_depth, 0, n * 8); memset(g->noise, 0, n * 8); memset(g->canny, 0, n * 12);
  for (size_t i = 0; i < n; ++i) g->zbuf[i] = 1.0f;
}

typedef struct { float vp[3], vn[3], uv[2], col[3]; float cx, cy, cz, cw; int32_t vid; } vtx;

static void run_vertex(const ref_draw* d, int idx, vtx* o) {
  const float* p = d->pos + 3 * idx; const float* n = d->normal + 3 * idx;
  float t[4];
  mat_vec(d->MV, p[0], p[1], p[2], 1.0f, t);                       /* worldPos = (MV*pos).xyz  (vert:46) */
  o->vp[0] = t[0]; o->vp[1] = t[1]; o->vp[2] = t[2];
  float c[4];
  mat_vec(d->P, t[0], t[1], t[2], 1.0f, c);                        /* gl_Position = projection*vec4(worldPos,1) */
  o->cx = c[0]; o->cy = c[1]; o->cz = c[2]; o->cw = c[3];
  float vn[4];
  mat_vec(d->MV_IT, n[0], n[1], n[2], 0.0f, vn);                   /* viewNormal = normalize((MV_IT*n).xyz) */
  o->vn[0] = vn[0]; o->vn[1] = vn[1]; o->vn[2] = vn[2];
  normalize3(o->vn);
  o->uv[0] = d->uv ? d->uv[2 * idx] : 0.0f; o->uv[1] = d->uv ? d->uv[2 * idx + 1] : 0.0f;
  for (int k = 0; k < 3; ++k) o->col[k] = d->color ? d->color[3 * idx + k] : 0.0f;
  o->vid = d->vertex_id ? d->vertex_id[idx] : idx;
}

static int to_fixed(float v) { return (int)floorf(v * 16.0f + 0.5f); }
static int64_t edge(int ax, int ay, int bx, int by, int px, int py) {
  return (int64_t)(bx - ax) * (int64_t)(py - ay) - (int64_t)(by - ay) * (int64_t)(px - ax);
}
static int top_left(int dx, int dy) { return (dy < 0) || (dy == 0 && dx > 0); }

void ref_raster_draw(const ref_draw* d, ref_gbuffer* g) {
  const int W = g->W, H = g->H;
  size_t npx = (size_t)W * H;
  /* snapshot of the targets before this draw ("current*" samplers) */
  uint16_t* s_color = g->color; uint16_t* s_nd = g->normal_depth; uint16_t* s_noise = g->noise;
  /* we need the pre-draw values while overwriting: keep copies */
  static uint16_t *c_color = 0, *c_nd = 0, *c_noise = 0; static int32_t* c_id = 0; static float *c_pos = 0, *c_canny = 0;
  static size_t cap = 0;
  if (cap < npx) {
    extern void* realloc(void*, size_t);
    c_color = realloc(c_color, npx * 8); c_nd = realloc(c_nd, npx * 8); c_noise = realloc(c_noise, npx * 8);
    c_id = realloc(c_id, npx * 16); c_pos = realloc(c_pos, npx * 12); c_canny = realloc(c_canny, npx * 12); cap = npx;
  }
  memcpy(c_color, s_color, npx * 8); memcpy(c_nd, s_nd, npx * 8); memcpy(c_noise, s_noise, npx * 8);
  memcpy(c_id, g->id, npx * 16); memcpy(c_pos, g->pos, npx * 12); memcpy(c_canny, g->canny, npx * 12);

  for (int t = 0; t < d->nt; ++t) {
    vtx v[3];
    for (int k = 0; k < 3; ++k) run_vertex(d, d->tris[3 * t + k], &v[k]);
    const int nfront = (v[0].cw > 0.0f) + (v[1].cw > 0.0f) + (v[2].cw > 0.0f);
    if (nfront == 0) continue;
    const int homog = nfront < 3;
    int fx[3] = {0, 0, 0}, fy[3] = {0, 0, 0}; float z[3] = {0, 0, 0}, iw[3] = {0, 0, 0};
    int sgn = 1, tl[3] = {0, 0, 0}; float farea = 1.0f;
    int x0 = 0, x1 = W - 1, y0 = 0, y1 = H - 1;
    float E[9];
    if (!homog) {
      for (int k = 0; k < 3; ++k) {
        iw[k] = 1.0f / v[k].cw;
        float nx = v[k].cx * iw[k], ny = v[k].cy * iw[k], nz = v[k].cz * iw[k];
        float sx = (nx * 0.5f + 0.5f) * (float)W;
        float sy = (1.0f - (ny * 0.5f + 0.5f)) * (float)H;
        z[k] = nz * 0.5f + 0.5f;
        fx[k] = to_fixed(sx); fy[k] = to_fixed(sy);
      }
      int64_t area = edge(fx[0], fy[0], fx[1], fy[1], fx[2], fy[2]);
      if (area == 0) continue;
      /* GL front face = counter-clockwise as seen on screen; with y pointing DOWN in these window coordinates a visually
         CCW triangle has a NEGATIVE edge-function area */
      if (area > 0 && d->cull_back) continue;
      sgn = area > 0 ? 1 : -1;
      int minx = fx[0], maxx = fx[0], miny = fy[0], maxy = fy[0];
      for (int k = 1; k < 3; ++k) { if (fx[k] < minx) minx = fx[k]; if (fx[k] > maxx) maxx = fx[k]; if (fy[k] < miny) miny = fy[k]; if (fy[k] > maxy) maxy = fy[k]; }
      x0 = (minx - 8 + 15) >> 4; x1 = (maxx - 8) >> 4; y0 = (miny - 8 + 15) >> 4; y1 = (maxy - 8) >> 4;
      if (x0 < 0) x0 = 0; if (y0 < 0) y0 = 0; if (x1 > W - 1) x1 = W - 1; if (y1 > H - 1) y1 = H - 1;
      /* edge i is opposite vertex i: e0 = v1->v2, e1 = v2->v0, e2 = v0->v1 */
      tl[0] = top_left(sgn * (fx[2] - fx[1]), sgn * (fy[2] - fy[1]));
      tl[1] = top_left(sgn * (fx[0] - fx[2]), sgn * (fy[0] - fy[2]));
      tl[2] = top_left(sgn * (fx[1] - fx[0]), sgn * (fy[1] - fy[0]));
      farea = (float)(sgn * area);
    } else {
      /* rows of the inverse of M = [(cx, cy, cw)_i]: e_i(X, Y) = (E[3i] X + E[3i+1] Y) + E[3i+2] = lambda_i / w at the pixel */
      float cof[9];
      for (int i = 0; i < 3; ++i) {
        const int j = (i + 1) % 3, k = (i + 2) % 3;
        cof[3 * i] = v[j].cy * v[k].cw - v[k].cy * v[j].cw;
        cof[3 * i + 1] = v[k].cx * v[j].cw - v[j].cx * v[k].cw;
        cof[3 * i + 2] = v[j].cx * v[k].cy - v[k].cx * v[j].cy;
      }
      const float det = (v[0].cx * cof[0] + v[0].cy * cof[1]) + v[0].cw * cof[2];
      if (det == 0.0f) continue;
      if (det < 0.0f && d->cull_back) continue;                  /* front faces have det > 0 (= negative window area above) */
      for (int i = 0; i < 9; ++i) E[i] = cof[i] / det;
    }
    for (int y = y0; y <= y1; ++y) for (int x = x0; x <= x1; ++x) {
      float f0, f1, f2, fs, zf;
      if (!homog) {
        int px = x * 16 + 8, py = y * 16 + 8;
        int64_t w0 = sgn * edge(fx[1], fy[1], fx[2], fy[2], px, py);
        int64_t w1 = sgn * edge(fx[2], fy[2], fx[0], fy[0], px, py);
        int64_t w2 = sgn * edge(fx[0], fy[0], fx[1], fy[1], px, py);
        if (w0 < 0 || w1 < 0 || w2 < 0) continue;
        if ((w0 == 0 && !tl[0]) || (w1 == 0 && !tl[1]) || (w2 == 0 && !tl[2])) continue;
        const float b0 = (float)w0 / farea, b1 = (float)w1 / farea, b2 = (float)w2 / farea;
        zf = (z[0] * b0 + z[1] * b1) + z[2] * b2;                     /* gl_FragCoord.z */
        f0 = b0 * iw[0]; f1 = b1 * iw[1]; f2 = b2 * iw[2];
      } else {
        const float X = (((float)x + 0.5f) / (float)W) * 2.0f - 1.0f;
        const float Y = 1.0f - (((float)y + 0.5f) / (float)H) * 2.0f;
        f0 = (E[0] * X + E[1] * Y) + E[2];
        f1 = (E[3] * X + E[4] * Y) + E[5];
        f2 = (E[6] * X + E[7] * Y) + E[8];
        if (!(f0 >= 0.0f && f1 >= 0.0f && f2 >= 0.0f)) continue;
        const float es = (f0 + f1) + f2;
        if (!(es > 0.0f)) continue;
        const float zc = (v[0].cz * f0 + v[1].cz * f1) + v[2].cz * f2;
        const float wc = (v[0].cw * f0 + v[1].cw * f1) + v[2].cw * f2;
        zf = (zc / wc) * 0.5f + 0.5f;
      }
      if (!(zf >= 0.0f && zf <= 1.0f)) continue;                     /* near / far clip */
      const size_t pi = (size_t)y * W + x;
      if (d->depth_test) { if (!(zf < g->zbuf[pi])) continue; }
      fs = (f0 + f1) + f2;
#define INTERP(a0, a1, a2) ((((a0) * f0 + (a1) * f1) + (a2) * f2) / fs)
      float vp[3], vn[3], uv[2], vc[3];
      for (int k = 0; k < 3; ++k) { vp[k] = INTERP(v[0].vp[k], v[1].vp[k], v[2].vp[k]); vn[k] = INTERP(v[0].vn[k], v[1].vn[k], v[2].vn[k]); vc[k] = INTERP(v[0].col[k], v[1].col[k], v[2].col[k]); }
      for (int k = 0; k < 2; ++k) uv[k] = INTERP(v[0].uv[k], v[1].uv[k], v[2].uv[k]);
      /* screen-space derivatives of uv for the mip level: the SAME triangle's perspective-correct uv one pixel to the right and
         one pixel down (forward differences of the exact attribute plane; GL hardware differences the quad's four invocations) */
      float rho2 = 0.0f;
      if (d->diffuse_tex && d->diffuse_levels >= 2) {
        float duv[2][2];
        for (int ax = 0; ax < 2; ++ax) {
          const int xx = x + (ax == 0), yy = y + (ax == 1);
          float g0, g1, g2;
          if (!homog) {
            const int qx = xx * 16 + 8, qy = yy * 16 + 8;
            const int64_t u0 = sgn * edge(fx[1], fy[1], fx[2], fy[2], qx, qy);
            const int64_t u1 = sgn * edge(fx[2], fy[2], fx[0], fy[0], qx, qy);
            const int64_t u2 = sgn * edge(fx[0], fy[0], fx[1], fy[1], qx, qy);
            g0 = ((float)u0 / farea) * iw[0]; g1 = ((float)u1 / farea) * iw[1]; g2 = ((float)u2 / farea) * iw[2];
          } else {
            const float X = (((float)xx + 0.5f) / (float)W) * 2.0f - 1.0f;
            const float Y = 1.0f - (((float)yy + 0.5f) / (float)H) * 2.0f;
            g0 = (E[0] * X + E[1] * Y) + E[2]; g1 = (E[3] * X + E[4] * Y) + E[5]; g2 = (E[6] * X + E[7] * Y) + E[8];
          }
          const float gs = (g0 + g1) + g2;
          for (int k = 0; k < 2; ++k) duv[ax][k] = (((v[0].uv[k] * g0 + v[1].uv[k] * g1) + v[2].uv[k] * g2) / gs) - uv[k];
        }
        const float ux = duv[0][0] * (float)d->diffuse_w, vx = duv[0][1] * (float)d->diffuse_h;
        const float uy = duv[1][0] * (float)d->diffuse_w, vy = duv[1][1] * (float)d->diffuse_h;
        const float rx = ux * ux + vx * vx, ry = uy * uy + vy * vy;
        rho2 = rx > ry ? rx : ry;
      }
      /* ---------------- fragment shader (frag.glsl:100-257) ---------------- */
      float outNoise[4] = {0, 0, 0, 0};
      if (d->noise_tex) {
        const int tx = nearest_index(uv[0], d->noise_w), ty = nearest_index(uv[1], d->noise_h);
        for (int k = 0; k < 4; ++k) outNoise[k] = h2f(d->noise_tex[((size_t)ty * d->noise_w + tx) * 4 + k]);
      }
      float outPos[3] = {vp[0], vp[1], vp[2]};
      const float depth = 1.0f - zf;
      float n[3] = {vn[0], vn[1], vn[2]};
      normalize3(n);
      if (d->normal_tex && d->tangent && d->bitangent) {
        /* frag:118-122 with the varyings of vert:49-53: modelTangent / modelBitangent are normalised per vertex, modelNormal
           is the raw attribute; all three are interpolated perspective-correct like every other varying */
        float T3[3][3], B3[3][3], N3[3][3];
        for (int k = 0; k < 3; ++k) {
          const int idx = d->tris[3 * t + k];
          for (int c = 0; c < 3; ++c) { T3[k][c] = d->tangent[3 * idx + c]; B3[k][c] = d->bitangent[3 * idx + c]; N3[k][c] = d->normal[3 * idx + c]; }
          normalize3(T3[k]); normalize3(B3[k]);
        }
        float mt[3], mb[3], mn[3];
        for (int c = 0; c < 3; ++c) { mt[c] = INTERP(T3[0][c], T3[1][c], T3[2][c]); mb[c] = INTERP(B3[0][c], B3[1][c], B3[2][c]); mn[c] = INTERP(N3[0][c], N3[1][c], N3[2][c]); }
        const int ntx = nearest_index(uv[0], d->normal_w), nty = nearest_index(uv[1], d->normal_h);
        const float* tp = d->normal_tex + ((size_t)nty * d->normal_w + ntx) * 4;
        float c3[3] = {tp[0] * 2.0f - 1.0f, tp[1] * 2.0f - 1.0f, tp[2] * 2.0f - 1.0f};
        normalize3(c3);
        float m3[3];
        for (int c = 0; c < 3; ++c) m3[c] = (mt[c] * c3[0] + mb[c] * c3[1]) + mn[c] * c3[2];
        normalize3(m3);
        float v4[4];
        mat_vec(d->MV_IT, m3[0], m3[1], m3[2], 0.0f, v4);
        n[0] = v4[0]; n[1] = v4[1]; n[2] = v4[2];
        normalize3(n);
      }
      float outND[4] = {n[0] * 0.5f + 0.5f, n[1] * 0.5f + 0.5f, n[2] * 0.5f + 0.5f, depth};
      int32_t real_vid;
      if (!d->use_texcoord_id) real_vid = v[2].vid;                      /* flat: provoking (last) vertex */
      else real_vid = (int32_t)((uv[1] * (float)d->id_h) * (float)d->id_w + uv[0] * (float)d->id_w);
      int32_t outID[4]; int32_t map_index = NON_AI_OBJ_MAP_INDEX;
      if (d->render_mode == 0) { outID[0] = d->sprite_id; outID[1] = d->material_id; outID[2] = NON_AI_OBJ_MAP_INDEX; outID[3] = real_vid; }
      else {
        const int k = d->corrmap_k;
        float l1 = sqrtf((0.0f * 0.0f + n[1] * n[1]) + n[2] * n[2]);
        float theta = (l1 == 0.0f) ? 0.0f : n[1] / l1;                   /* dot(normalize(0,ny,nz),(0,1,0)); 0/0 is UB in GLSL */
        theta = PI_F / 2.0f - theta;
        float l2 = sqrtf((n[0] * n[0] + 0.0f * 0.0f) + n[2] * n[2]);
        float phi = (l2 == 0.0f) ? 0.0f : n[0] / l2;
        phi = PI_F / 2.0f - phi;
        const float step = PI_F / (float)k;
        int xi = (int)(theta / step), yi = (int)(phi / step);
        if (xi < 0) xi = 0; if (xi > k - 1) xi = k - 1; if (yi < 0) yi = 0; if (yi > k - 1) yi = k - 1;
        map_index = xi + (k - 1 - yi) * k;
        outID[0] = d->sprite_id; outID[1] = d->material_id; outID[2] = map_index; outID[3] = real_vid;
      }
      float outColor[4];
      if (d->render_mode == 0) {
        if (!d->diffuse_tex) {
          if (d->has_vertex_color) { outColor[0] = vc[0]; outColor[1] = vc[1]; outColor[2] = vc[2]; outColor[3] = 1.0f; }
          else { outColor[0] = outColor[1] = outColor[2] = outColor[3] = 0.0f; }
        } else {
          if (d->diffuse_levels >= 2) tex_trilinear(d->diffuse_tex, d->diffuse_w, d->diffuse_h, d->diffuse_levels, uv[0], uv[1], rho2, outColor);
          else {
            const int tx = nearest_index(uv[0], d->diffuse_w), ty = nearest_index(uv[1], d->diffuse_h);
            for (int k = 0; k < 4; ++k) outColor[k] = d->diffuse_tex[((size_t)ty * d->diffuse_w + tx) * 4 + k];
          }
        }
      } else if (d->render_mode == 2) { outColor[0] = outColor[1] = outColor[2] = outColor[3] = 0.0f; }
      else {
        if (d->corrmap_tex) {                                               /* corrmap_uv = (uv.y, uv.x, map_index) (sic) */
          const int tx = nearest_index(uv[1], d->corr_w), ty = nearest_index(uv[0], d->corr_h);
          const size_t base = (((size_t)map_index * d->corr_h + ty) * d->corr_w + tx) * 4;
          for (int k = 0; k < 4; ++k) outColor[k] = h2f(d->corrmap_tex[base + k]);
        } else if (!d->diffuse_tex) {
          if (d->has_vertex_color) { outColor[0] = vc[0]; outColor[1] = vc[1]; outColor[2] = vc[2]; outColor[3] = 1.0f; }
          else { outColor[0] = 1.0f; outColor[1] = 0.0f; outColor[2] = 1.0f; outColor[3] = 1.0f; }
        } else {
          if (d->diffuse_levels >= 2) tex_trilinear(d->diffuse_tex, d->diffuse_w, d->diffuse_h, d->diffuse_levels, uv[0], uv[1], rho2, outColor);
          else {
            const int tx = nearest_index(uv[0], d->diffuse_w), ty = nearest_index(uv[1], d->diffuse_h);
            for (int k = 0; k < 4; ++k) outColor[k] = d->diffuse_tex[((size_t)ty * d->diffuse_w + tx) * 4 + k];
          }
        }
      }
      float outCanny = (n[2] < CANNY_THRESHOLD && n[2] > 0.0f) ? 1.0f : 0.0f;
      /* blend against the pre-draw snapshot */
      float curColor[4], curND[4], curNoise[4];
      for (int k = 0; k < 4; ++k) { curColor[k] = h2f(c_color[pi * 4 + k]); curND[k] = h2f(c_nd[pi * 4 + k]); curNoise[k] = h2f(c_noise[pi * 4 + k]); }
      float outPos3[3] = {outPos[0], outPos[1], outPos[2]}, outCanny3[3] = {outCanny, outCanny, outCanny};
      if (d->render_mode == 2 || (outColor[3] == 0.0f && d->render_mode == 1)) {
        for (int k = 0; k < 4; ++k) outColor[k] = curColor[k];
        if (d->render_mode == 1) for (int k = 0; k < 4; ++k) outID[k] = c_id[pi * 4 + k];
        for (int k = 0; k < 3; ++k) { outPos3[k] = c_pos[pi * 3 + k]; outCanny3[k] = c_canny[pi * 3 + k]; }
        for (int k = 0; k < 4; ++k) outND[k] = curND[k];
      } else if (outColor[3] < 1.0f) {
        const float latest_depth = curND[3];
        const float nsum = ((curNoise[0] + curNoise[1]) + curNoise[2]) + curNoise[3];
        const float a = outColor[3];
        if (latest_depth < depth) {
          for (int k = 0; k < 3; ++k) outColor[k] = outColor[k] * a + curColor[k] * (1.0f - a);
          if (nsum > 0.001f) for (int k = 0; k < 4; ++k) outNoise[k] = outNoise[k] * a + curNoise[k] * (1.0f - a);
        } else {
          const float ca = curColor[3];
          for (int k = 0; k < 3; ++k) outColor[k] = curColor[k] * ca + outColor[k] * (1.0f - ca);
          outColor[3] = ca;
          if (nsum > 0.001f) for (int k = 0; k < 4; ++k) outNoise[k] = curNoise[k] * ca + outNoise[k] * (1.0f - ca);
          outND[3] = latest_depth;
        }
      }
      /* write */
      if (d->depth_test) g->zbuf[pi] = zf;
      for (int k = 0; k < 4; ++k) { g->color[pi * 4 + k] = f2h(outColor[k]); g->normal_depth[pi * 4 + k] = f2h(outND[k]); g->noise[pi * 4 + k] = f2h(outNoise[k]); g->id[pi * 4 + k] = outID[k]; }
      for (int k = 0; k < 3; ++k) { g->pos[pi * 3 + k] = outPos3[k]; g->canny[pi * 3 + k] = outCanny3[k]; }
    }
  }
}
