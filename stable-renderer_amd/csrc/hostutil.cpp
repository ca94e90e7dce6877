// Host-side (CPU) helpers of the C ABI: index bookkeeping that is sequential by nature and runs once per call, never per step.
#include "sr_common.h"
#include <vector>
#include <algorithm>

// Level schedule of the legacy in-place Overlap (see overlap.hip: legacy_seq_compute).  Vertices are visited in the reference's
// dict order (`order`); per pixel the level of its last writer and the highest level of its readers so far are tracked; a vertex
// goes one level above everything it conflicts with:  level(j) = 1 + max( max_{p in read(j)} wlevel[p], max_{p in write(j)} rlevel[p] ).
// read(j) = the clamped (2r+1)-pixel diagonals around its trace pixels (overlap.py:61-81), write(j) = its trace pixels.
// Vertices seen once never write (overlap.py:122-126): they get level -1.  All pointers are HOST memory.
extern "C" int sr_legacy_levels(const int32_t* offsets, const int32_t* tr_f, const int32_t* tr_y, const int32_t* tr_x, const int32_t* order,
                                int32_t n_vertices, int32_t T, int32_t H, int32_t W, int32_t radius, int32_t* level_of, int32_t* n_levels) {
  if (!offsets || !tr_f || !tr_y || !tr_x || !order || !level_of || !n_levels || radius < 0) SR_FAIL(SR_ERR_INVALID, "sr_legacy_levels: bad args");
  const size_t npx = (size_t)T * H * W;
  std::vector<int32_t> wl(npx, -1), rl(npx, -1);
  int32_t top = -1;
  for (int32_t oi = 0; oi < n_vertices; ++oi) {
    const int32_t v = order[oi];
    if (v < 0 || v >= n_vertices) SR_FAIL(SR_ERR_INVALID, "sr_legacy_levels: vertex out of range");
    const int32_t b = offsets[v], e = offsets[v + 1];
    if (e - b < 2) { level_of[v] = -1; continue; }
    int32_t lv = -1;
    for (int32_t s = b; s < e; ++s) {
      const size_t base = (size_t)tr_f[s] * H * W;
      lv = std::max(lv, rl[base + (size_t)tr_y[s] * W + tr_x[s]]);
      for (int k = -radius; k <= radius; ++k) {
        const int yy = std::min(std::max(tr_y[s] + k, 0), H - 1), xx = std::min(std::max(tr_x[s] + k, 0), W - 1);
        lv = std::max(lv, wl[base + (size_t)yy * W + xx]);
      }
    }
    lv += 1;
    level_of[v] = lv;
    top = std::max(top, lv);
    for (int32_t s = b; s < e; ++s) {
      const size_t base = (size_t)tr_f[s] * H * W;
      wl[base + (size_t)tr_y[s] * W + tr_x[s]] = lv;
      for (int k = -radius; k <= radius; ++k) {
        const int yy = std::min(std::max(tr_y[s] + k, 0), H - 1), xx = std::min(std::max(tr_x[s] + k, 0), W - 1);
        int32_t& r = rl[base + (size_t)yy * W + xx];
        r = std::max(r, lv);
      }
    }
  }
  *n_levels = top + 1;
  return SR_OK;
}
