// labels.hip — training-label creation of the distance method on the device (second part).
// Reference: src/training/train_data_representations.py, distance_label (:261-361) and bottom_hat_closing (:40-72): per
// cell a Euclidean distance transform inside a search window around the rounded centroid (cell distances, normalised per
// cell), a second one towards the other cells of the window (neighbour distances), gaps between close cells found by a
// bottom-hat transform with a radius-3 disk, artefact gaps removed by the neighbour distances on their rim, an exponential
// rescaling and a 3 x 3 grey closing.  The reference loops over the cells in Python (scipy EDT per crop, two binary
// closings per cell over the whole image); here every step is one pass over the pixel batch:
//
//   runs      per row: for every pixel the columns where its constant-label run ends on either side
//   props     per cell id: pixel count and coordinate sums (integer atomics) -> rounded centroid -> search window
//   edt       per cell pixel: exact squared Euclidean distances (integers) by a row sweep outwards from the pixel; row y'
//             contributes dy^2 + g^2 with g read off the run ends (own-label run for the cell distance; walking over the
//             alternating background / own-cell runs for the neighbour distance), clipped to the cell's window; the sweep
//             stops when dy^2 reaches the best value.  Per-cell maxima by integer atomicMax; sqrt / divisions in fp64 as
//             scipy does, so the values are the reference's.
//   closing   label_bin = OR over the cells of binary_closing(cell, disk(3)) (scipy's border_value 0: pixels less than 3
//             from the frame never survive the erosion), then closing(label_bin) & ~label_bin = the gaps
//   gaps      8-connected components (union-find), per component area + second moments (integer atomics) -> minor axis
//             length, rim sum of the neighbour distances (fp64 atomics) -> artefact test; rim of wide gaps weighs 0.8
//   final     max(neighbour, gap weight, border) -> 1/sqrt(0.65 + 0.5 exp(-11 (v - 0.75))) - 0.19 -> clip -> 3x3 max, 3x3 min
//
// Exactness: distances, windows, closings, components are integer work (bit-exact).  The fp64 sums that feed the two
// thresholds (minor axis >= 3, rim sum < th) are accumulated in a different order than numpy's, so a decision can differ
// only when a sum lies within an ulp of its threshold; exp() of the rescaling is the device's fp64 exp (results are rounded
// to fp32 afterwards).  Tests compare with the oracle at 1e-6 absolute.
// Not reproduced: a search window that contains no pixel outside the cell (scipy's EDT is undefined without a background
// pixel): such a cell gets 0 in both maps.
#include "common.h"

#define LB_BLOCK 256
#define LB_IDS 65536          // uint16 instance ids: the per-cell tables are addressed by the id itself
#define LB_INF 0x7fffffff

static inline unsigned lb_blocks(size_t n) {
  size_t b = (n + LB_BLOCK - 1) / LB_BLOCK;
  return (unsigned)(b < 1 ? 1 : (b > 65535u * 16u ? 65535u * 16u : b));
}

struct LbCell {               // per (image, id)
  unsigned long long sy, sx;
  unsigned cnt;
  int maxc, maxn;             // largest squared cell / neighbour distance inside the window
  int y0, y1, x0, x1;         // search window [y0, y1) x [x0, x1)
  int pad;
};

struct LbGap {                // per (image, root pixel of a gap component)
  unsigned long long sy, sx, syy, sxx, sxy;
  double ring;                // sum of the neighbour distances over the 8-neighbour rim
  unsigned cnt;
  unsigned pad;
};

// ---- runs ---------------------------------------------------------------------------------------------------------------
__global__ void lb_runs_kernel(const uint16_t* __restrict__ mask, int N, int H, int W, int32_t* __restrict__ prevd,
                               int32_t* __restrict__ nextd) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= N * H) return;
  const uint16_t* m = mask + (size_t)row * W;
  int32_t* pd = prevd + (size_t)row * W;
  int32_t* nd = nextd + (size_t)row * W;
  int last = -1;                                   // column of the last pixel whose label differs from the current run
  for (int x = 0; x < W; ++x) {
    if (x > 0 && m[x] != m[x - 1]) last = x - 1;
    pd[x] = last;
  }
  last = W;
  for (int x = W - 1; x >= 0; --x) {
    if (x + 1 < W && m[x] != m[x + 1]) last = x + 1;
    nd[x] = last;
  }
}

// ---- cell properties ------------------------------------------------------------------------------------------------------
__global__ void lb_props_kernel(const uint16_t* __restrict__ mask, int N, int H, int W, LbCell* __restrict__ cells) {
  const size_t hw = (size_t)H * W, n = (size_t)N * hw;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
    const int k = mask[t];
    if (!k) continue;
    const size_t s = t / hw;
    const int r = (int)(t - s * hw);
    const int y = r / W, x = r - y * W;
    LbCell* c = cells + s * LB_IDS + k;
    atomicAdd(&c->cnt, 1u);
    atomicAdd(&c->sy, (unsigned long long)y);
    atomicAdd(&c->sx, (unsigned long long)x);
  }
}

__global__ void lb_window_kernel(LbCell* __restrict__ cells, int N, int H, int W, int sr) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)N * LB_IDS) return;
  LbCell* c = cells + i;
  if (!c->cnt) return;
  // regionprops centroid = mean of the pixel coordinates (fp64), np.round = round half to even
  const double cy = rint((double)c->sy / (double)c->cnt), cx = rint((double)c->sx / (double)c->cnt);
  c->y0 = (int)fmax(cy - sr, 0.0);
  c->y1 = (int)fmin(cy + sr, (double)H);
  c->x0 = (int)fmax(cx - sr, 0.0);
  c->x1 = (int)fmin(cx + sr, (double)W);
}

// ---- windowed exact Euclidean distances -------------------------------------------------------------------------------------
// horizontal distance in row `m` (labels), from column x, to the nearest pixel that is not cell k, inside [x0, x1)
__device__ __forceinline__ int lb_gap_cell(const uint16_t* m, const int32_t* pd, const int32_t* nd, int x, int k, int x0,
                                           int x1) {
  if (m[x] != k) return 0;
  const int l = pd[x], r = nd[x];
  int g = LB_INF;
  if (l >= x0) g = x - l;
  if (r < x1) g = min(g, r - x);
  return g;
}

// ... to the nearest pixel of ANOTHER cell (label neither 0 nor k); `lim` = largest useful distance (exclusive)
__device__ __forceinline__ int lb_gap_other(const uint16_t* m, const int32_t* pd, const int32_t* nd, int x, int k, int x0,
                                            int x1, int lim) {
  const int v = m[x];
  if (v != 0 && v != k) return 0;
  int g = LB_INF;
  int c = pd[x];
  while (c >= x0 && x - c < lim) {
    const int lc = m[c];
    if (lc != 0 && lc != k) { g = x - c; break; }
    c = pd[c];
  }
  if (g < lim) lim = g;
  c = nd[x];
  while (c < x1 && c - x < lim) {
    const int lc = m[c];
    if (lc != 0 && lc != k) { g = c - x; break; }
    c = nd[c];
  }
  return g;
}

__device__ __forceinline__ int lb_isqrt_ceil(int v) {       // smallest r with r*r >= v
  int r = (int)sqrtf((float)v);
  while (r * r < v) ++r;
  while (r > 0 && (r - 1) * (r - 1) >= v) --r;
  return r;
}

__global__ void lb_edt_kernel(const uint16_t* __restrict__ mask, int N, int H, int W, const int32_t* __restrict__ prevd,
                              const int32_t* __restrict__ nextd, LbCell* __restrict__ cells, int32_t* __restrict__ d2c,
                              int32_t* __restrict__ d2n) {
  const size_t hw = (size_t)H * W, n = (size_t)N * hw;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
    const int k = mask[t];
    int bc = -1, bn = -1;
    if (k) {
      const size_t s = t / hw;
      const int r = (int)(t - s * hw);
      const int y = r / W, x = r - y * W;
      LbCell* c = cells + s * LB_IDS + k;
      const int y0 = c->y0, y1 = c->y1, x0 = c->x0, x1 = c->x1;
      if (y >= y0 && y < y1 && x >= x0 && x < x1) {
        const uint16_t* m = mask + s * hw;
        const int32_t* pd = prevd + s * hw;
        const int32_t* nd = nextd + s * hw;
        const int dymax = max(y - y0, y1 - 1 - y);
        int best = LB_INF;
        for (int dy = 0; dy <= dymax && dy * dy < best; ++dy) {
#pragma unroll
          for (int sgn = 0; sgn < 2; ++sgn) {
            if (sgn && !dy) continue;
            const int yy = sgn ? y + dy : y - dy;
            if (yy < y0 || yy >= y1) continue;
            const size_t o = (size_t)yy * W;
            const int g = lb_gap_cell(m + o, pd + o, nd + o, x, k, x0, x1);
            if (g < 46341) best = min(best, dy * dy + g * g);
          }
        }
        if (best != LB_INF) bc = best;
        best = LB_INF;
        for (int dy = 0; dy <= dymax && dy * dy < best; ++dy) {
          const int lim = best == LB_INF ? 46341 : lb_isqrt_ceil(best - dy * dy);
#pragma unroll
          for (int sgn = 0; sgn < 2; ++sgn) {
            if (sgn && !dy) continue;
            const int yy = sgn ? y + dy : y - dy;
            if (yy < y0 || yy >= y1) continue;
            const size_t o = (size_t)yy * W;
            const int g = lb_gap_other(m + o, pd + o, nd + o, x, k, x0, x1, lim);
            if (g < 46341) best = min(best, dy * dy + g * g);
          }
        }
        if (best != LB_INF) bn = best;
        if (bc > 0) atomicMax(&c->maxc, bc);
        if (bn > 0) atomicMax(&c->maxn, bn);
      }
    }
    d2c[t] = bc;
    d2n[t] = bn;
  }
}

// clip > 0: cell_distance_label(..., apply_clipping=True, clip_val=clip): min(distance, clip) / clip instead of the
// per-cell normalisation (train_data_representations.py:245-256); nb may be NULL (cell distances only)
__global__ void lb_value_kernel(const uint16_t* __restrict__ mask, int N, int H, int W, const LbCell* __restrict__ cells,
                                const int32_t* __restrict__ d2c, const int32_t* __restrict__ d2n, float* __restrict__ cell,
                                double* __restrict__ nb, double clip) {
  const size_t hw = (size_t)H * W, n = (size_t)N * hw;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
    const int k = mask[t];
    float cv = 0.f;
    double nv = 0.0;
    if (k) {
      const LbCell* c = cells + (t / hw) * LB_IDS + k;
      const int bc = d2c[t], bn = d2n[t];
      if (bc > 0 && c->maxc > 0) {
        const double dmax = sqrt((double)c->maxc);
        cv = clip > 0.0 ? (float)(fmin(sqrt((double)bc), clip) / clip) : (float)(sqrt((double)bc) / dmax);
        if (bn > 0) {
          const double den = fmin(dmax + 3.0, sqrt((double)c->maxn));
          nv = 1.0 - fmin(sqrt((double)bn) / den, 1.0);
        }
      }
    }
    cell[t] = cv;
    if (nb) nb[t] = nv;
  }
}

// ---- binary closings with disk(3) --------------------------------------------------------------------------------------------
__device__ __forceinline__ bool lb_in_disk(int dy, int dx) { return dy * dy + dx * dx <= 9; }

// is pixel (y, x) in dilate(cell k, disk(3))?  (pixels outside the image carry nothing)
__device__ __forceinline__ bool lb_dilated(const uint16_t* m, int H, int W, int y, int x, int k) {
  for (int dy = -3; dy <= 3; ++dy) {
    const int yy = y + dy;
    if (yy < 0 || yy >= H) continue;
    for (int dx = -3; dx <= 3; ++dx) {
      const int xx = x + dx;
      if (xx < 0 || xx >= W || !lb_in_disk(dy, dx)) continue;
      if (m[(size_t)yy * W + xx] == k) return true;
    }
  }
  return false;
}

// is (y, x) in binary_closing(cell k, disk(3)) (dilation, then erosion with border_value 0)?
__device__ __forceinline__ bool lb_closed(const uint16_t* m, int H, int W, int y, int x, int k) {
  if (y < 3 || y + 3 >= H || x < 3 || x + 3 >= W) return false;   // the erosion sees the frame as empty
  for (int dy = -3; dy <= 3; ++dy)
    for (int dx = -3; dx <= 3; ++dx) {
      if (!lb_in_disk(dy, dx)) continue;
      if (!lb_dilated(m, H, W, y + dy, x + dx, k)) return false;
    }
  return true;
}

__global__ void lb_close_cells_kernel(const uint16_t* __restrict__ mask, int N, int H, int W, uint8_t* __restrict__ bin) {
  const size_t hw = (size_t)H * W, n = (size_t)N * hw;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
    const size_t s = t / hw;
    const int r = (int)(t - s * hw);
    const int y = r / W, x = r - y * W;
    const uint16_t* m = mask + s * hw;
    bool in = false;
    if (y >= 3 && y + 3 < H && x >= 3 && x + 3 < W) {
      if (m[r]) {
        in = true;                                   // closing is extensive away from the frame
      } else {
        int tried[6];
        int nt = 0;
        for (int dy = -3; dy <= 3 && !in; ++dy)
          for (int dx = -3; dx <= 3 && !in; ++dx) {
            if (!lb_in_disk(dy, dx)) continue;
            const int k = m[(size_t)(y + dy) * W + x + dx];
            if (!k) continue;
            bool seen = false;
            for (int j = 0; j < nt; ++j) seen |= tried[j] == k;
            if (seen) continue;
            if (nt < 6) tried[nt++] = k;
            in = lb_closed(m, H, W, y, x, k);
          }
      }
    }
    bin[t] = in ? 1 : 0;
  }
}

template <bool ERODE>
__global__ void lb_disk_kernel(const uint8_t* __restrict__ in, const uint8_t* __restrict__ bin, int N, int H, int W,
                               uint8_t* __restrict__ out) {
  const size_t hw = (size_t)H * W, n = (size_t)N * hw;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
    const size_t s = t / hw;
    const int r = (int)(t - s * hw);
    const int y = r / W, x = r - y * W;
    const uint8_t* p = in + s * hw;
    bool v = ERODE;
    for (int dy = -3; dy <= 3; ++dy)
      for (int dx = -3; dx <= 3; ++dx) {
        if (!lb_in_disk(dy, dx)) continue;
        const int yy = y + dy, xx = x + dx;
        const bool inside = yy >= 0 && yy < H && xx >= 0 && xx < W;
        const bool q = inside && p[(size_t)yy * W + xx];
        if (ERODE) v &= q; else v |= q;
      }
    out[t] = ERODE ? (uint8_t)(v && !bin[t]) : (uint8_t)v;     // erosion pass: closing & ~label_bin = the gaps
  }
}

// ---- gap components -------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lb_find(const int32_t* L, int i) {
  int p = __hip_atomic_load(&L[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while (p != i) {
    i = p;
    p = __hip_atomic_load(&L[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return i;
}

__device__ __forceinline__ void lb_union(int32_t* L, int a, int b) {
  for (;;) {
    a = lb_find(L, a);
    b = lb_find(L, b);
    if (a == b) return;
    if (a > b) { const int t = a; a = b; b = t; }
    const int old = atomicMin(&L[b], a);
    if (old == b) return;
    b = old;
  }
}

__global__ void lb_ccl_init_kernel(const uint8_t* __restrict__ gap, int N, size_t hw, int32_t* __restrict__ L) {
  const size_t n = (size_t)N * hw;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x)
    L[t] = gap[t] ? (int32_t)(t % hw) : -1;
}

__global__ void lb_ccl_merge_kernel(const uint8_t* __restrict__ gap, int N, int H, int W, int32_t* __restrict__ L) {
  const size_t hw = (size_t)H * W, n = (size_t)N * hw;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
    if (!gap[t]) continue;
    const size_t s = t / hw;
    const int r = (int)(t - s * hw);
    const int y = r / W, x = r - y * W;
    const uint8_t* g = gap + s * hw;
    int32_t* l = L + s * hw;
    if (x > 0 && g[r - 1]) lb_union(l, r, r - 1);
    if (y > 0) {
      if (g[r - W]) lb_union(l, r, r - W);
      if (x > 0 && g[r - W - 1]) lb_union(l, r, r - W - 1);
      if (x + 1 < W && g[r - W + 1]) lb_union(l, r, r - W + 1);
    }
  }
}

__global__ void lb_ccl_flatten_kernel(int N, size_t hw, int32_t* __restrict__ L) {
  const size_t n = (size_t)N * hw;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x)
    if (L[t] >= 0) L[t] = lb_find(L + (t / hw) * hw, (int)(t % hw));
}

__global__ void lb_gap_props_kernel(const int32_t* __restrict__ L, const double* __restrict__ nb, int N, int H, int W,
                                    LbGap* __restrict__ gaps) {
  const size_t hw = (size_t)H * W, n = (size_t)N * hw;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
    const size_t s = t / hw;
    const int r = (int)(t - s * hw);
    const int y = r / W, x = r - y * W;
    const int32_t* l = L + s * hw;
    LbGap* gp = gaps + s * hw;
    const int root = l[r];
    if (root >= 0) {
      LbGap* g = gp + root;
      const unsigned long long uy = (unsigned long long)y, ux = (unsigned long long)x;
      atomicAdd(&g->cnt, 1u);
      atomicAdd(&g->sy, uy);
      atomicAdd(&g->sx, ux);
      atomicAdd(&g->syy, uy * uy);
      atomicAdd(&g->sxx, ux * ux);
      atomicAdd(&g->sxy, ux * uy);
    } else {
      const double v = nb[t];
      if (v == 0.0) continue;
      int seen[8];
      int ns = 0;
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
          if (!dy && !dx) continue;
          const int yy = y + dy, xx = x + dx;
          if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
          const int q = l[(size_t)yy * W + xx];
          if (q < 0) continue;
          bool dup = false;
          for (int j = 0; j < ns; ++j) dup |= seen[j] == q;
          if (dup) continue;
          seen[ns++] = q;
          atomicAdd(&gp[q].ring, v);                 // this pixel lies on the 3x3 dilation rim of component q
        }
    }
  }
}

// ---- final: gap weights, border, rescaling --------------------------------------------------------------------------------------
__global__ void lb_final_kernel(const uint16_t* __restrict__ mask, const int32_t* __restrict__ L,
                                const LbGap* __restrict__ gaps, const double* __restrict__ nb, int N, int H, int W,
                                float* __restrict__ out) {
  const size_t hw = (size_t)H * W, n = (size_t)N * hw;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
    const size_t s = t / hw;
    const int r = (int)(t - s * hw);
    const int y = r / W, x = r - y * W;
    const uint16_t* m = mask + s * hw;
    const int32_t* l = L + s * hw;
    double v = nb[t];
    const int root = l[r];
    if (root >= 0) {
      const LbGap g = gaps[s * hw + root];
      const double cnt = (double)g.cnt;
      const int area = (int)g.cnt;
      const double th = area <= 20 ? 5.0 : area <= 30 ? 8.0 : area <= 50 ? 10.0 : 20.0;
      if (!(g.ring < th)) {
        // normalised second central moments from the exact integer sums: (n * Syy - Sy^2) / n^2
        const double a = (double)(long long)(g.cnt * g.syy - g.sy * g.sy) / (cnt * cnt);
        const double c = (double)(long long)(g.cnt * g.sxx - g.sx * g.sx) / (cnt * cnt);
        const double b = (double)((long long)(g.cnt * g.sxy) - (long long)(g.sx * g.sy)) / (cnt * cnt);
        const double l2 = 0.5 * (a + c) - 0.5 * sqrt(4.0 * b * b + (a - c) * (a - c));
        const bool wide = 4.0 * sqrt(fmax(l2, 0.0)) >= 3.0;
        bool rim = false;
        if (wide) {
          rim = y == 0 || y == H - 1 || x == 0 || x == W - 1;
          if (!rim) rim = l[r - 1] != root || l[r + 1] != root || l[r - W] != root || l[r + W] != root;
        }
        v = fmax(v, rim ? (double)0.8f : 1.0);
      }
    }
    const int k = m[r];
    if (k) {                                          // border_label == 2: a cell pixel that touches another cell
      bool other = false;
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
          const int yy = y + dy, xx = x + dx;
          if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
          const int q = m[(size_t)yy * W + xx];
          other |= (q != 0) & (q != k);
        }
      if (other) v = fmax(v, 1.0);
    }
    double sc = 1.0 / sqrt(0.65 + 0.5 * exp(-11.0 * (v - 0.75))) - 0.19;
    sc = fmin(fmax(sc, 0.0), 1.0);
    out[t] = (float)sc;
  }
}

// 3x3 flat grey dilation (MAXF) / erosion with scipy's 'reflect' border (= clamped indices for a radius of 1)
template <bool MAXF>
__global__ void lb_grey3_kernel(const float* __restrict__ in, int N, int H, int W, float* __restrict__ out) {
  const size_t hw = (size_t)H * W, n = (size_t)N * hw;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
    const size_t s = t / hw;
    const int r = (int)(t - s * hw);
    const int y = r / W, x = r - y * W;
    const float* p = in + s * hw;
    float v = p[r];
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int yy = min(max(y + dy, 0), H - 1), xx = min(max(x + dx, 0), W - 1);
        const float q = p[(size_t)yy * W + xx];
        v = MAXF ? fmaxf(v, q) : fminf(v, q);
      }
    out[t] = v;
  }
}

// ---- C ABI ------------------------------------------------------------------------------------------------------------------------
static inline size_t lb_align(size_t v) { return (v + 255) & ~(size_t)255; }

struct LbLayout {
  size_t cells, gaps, prevd, nextd, d2c, d2n, nb, bin, tmp, gap, L, total, zero_bytes;
};

static LbLayout lb_layout(int N, int H, int W) {
  LbLayout l;
  const size_t px = (size_t)N * H * W;
  size_t o = 0;
  l.cells = o; o += lb_align((size_t)N * LB_IDS * sizeof(LbCell));
  l.gaps = o;  o += lb_align(px * sizeof(LbGap));
  l.zero_bytes = o;                                  // the two accumulator tables are cleared by every call
  l.prevd = o; o += lb_align(px * 4);
  l.nextd = o; o += lb_align(px * 4);
  l.d2c = o;   o += lb_align(px * 4);
  l.d2n = o;   o += lb_align(px * 4);
  l.nb = o;    o += lb_align(px * 8);
  l.bin = o;   o += lb_align(px);
  l.tmp = o;   o += lb_align(px);
  l.gap = o;   o += lb_align(px);
  l.L = o;     o += lb_align(px * 4);
  l.total = o;
  return l;
}

extern "C" size_t mseg_label_distance_workspace_bytes(int N, int H, int W) {
  if (N <= 0 || H <= 0 || W <= 0 || (size_t)H * W >= 0x7fffffffull || H > 32767 || W > 32767) return 0;
  return lb_layout(N, H, W).total;
}

extern "C" int mseg_label_distance(const uint16_t* mask, int N, int H, int W, int search_radius, float* cell_out,
                                   float* neighbor_out, void* ws, size_t ws_bytes, void* stream) {
  if (!mask || !cell_out || !neighbor_out || !ws || search_radius <= 0) return MSEG_EINVAL;
  const size_t need = mseg_label_distance_workspace_bytes(N, H, W);
  if (need == 0) return MSEG_EINVAL;
  if (ws_bytes < need) return MSEG_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const LbLayout l = lb_layout(N, H, W);
  char* base = (char*)ws;
  LbCell* cells = (LbCell*)(base + l.cells);
  LbGap* gaps = (LbGap*)(base + l.gaps);
  int32_t* prevd = (int32_t*)(base + l.prevd);
  int32_t* nextd = (int32_t*)(base + l.nextd);
  int32_t* d2c = (int32_t*)(base + l.d2c);
  int32_t* d2n = (int32_t*)(base + l.d2n);
  double* nb = (double*)(base + l.nb);
  uint8_t* bin = (uint8_t*)(base + l.bin);
  uint8_t* tmp = (uint8_t*)(base + l.tmp);
  uint8_t* gap = (uint8_t*)(base + l.gap);
  int32_t* L = (int32_t*)(base + l.L);
  const size_t hw = (size_t)H * W, px = (size_t)N * hw;
  const unsigned nbk = lb_blocks(px);
  if (hipMemsetAsync(ws, 0, l.zero_bytes, st) != hipSuccess) return MSEG_ELAUNCH;
  hipLaunchKernelGGL(lb_runs_kernel, dim3((N * H + 63) / 64), dim3(64), 0, st, mask, N, H, W, prevd, nextd);
  hipLaunchKernelGGL(lb_props_kernel, dim3(nbk), dim3(LB_BLOCK), 0, st, mask, N, H, W, cells);
  hipLaunchKernelGGL(lb_window_kernel, dim3(lb_blocks((size_t)N * LB_IDS)), dim3(LB_BLOCK), 0, st, cells, N, H, W,
                     search_radius);
  hipLaunchKernelGGL(lb_edt_kernel, dim3(nbk), dim3(LB_BLOCK), 0, st, mask, N, H, W, prevd, nextd, cells, d2c, d2n);
  hipLaunchKernelGGL(lb_value_kernel, dim3(nbk), dim3(LB_BLOCK), 0, st, mask, N, H, W, cells, d2c, d2n, cell_out, nb, 0.0);
  hipLaunchKernelGGL(lb_close_cells_kernel, dim3(nbk), dim3(LB_BLOCK), 0, st, mask, N, H, W, bin);
  hipLaunchKernelGGL(lb_disk_kernel<false>, dim3(nbk), dim3(LB_BLOCK), 0, st, bin, bin, N, H, W, tmp);
  hipLaunchKernelGGL(lb_disk_kernel<true>, dim3(nbk), dim3(LB_BLOCK), 0, st, tmp, bin, N, H, W, gap);
  hipLaunchKernelGGL(lb_ccl_init_kernel, dim3(nbk), dim3(LB_BLOCK), 0, st, gap, N, hw, L);
  hipLaunchKernelGGL(lb_ccl_merge_kernel, dim3(nbk), dim3(LB_BLOCK), 0, st, gap, N, H, W, L);
  hipLaunchKernelGGL(lb_ccl_flatten_kernel, dim3(nbk), dim3(LB_BLOCK), 0, st, N, hw, L);
  hipLaunchKernelGGL(lb_gap_props_kernel, dim3(nbk), dim3(LB_BLOCK), 0, st, L, nb, N, H, W, gaps);
  float* f0 = (float*)d2c;                           // the distance planes are free again: scratch of the grey closing
  float* f1 = (float*)d2n;
  hipLaunchKernelGGL(lb_final_kernel, dim3(nbk), dim3(LB_BLOCK), 0, st, mask, L, gaps, nb, N, H, W, f0);
  hipLaunchKernelGGL(lb_grey3_kernel<true>, dim3(nbk), dim3(LB_BLOCK), 0, st, f0, N, H, W, f1);
  hipLaunchKernelGGL(lb_grey3_kernel<false>, dim3(nbk), dim3(LB_BLOCK), 0, st, f1, N, H, W, neighbor_out);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

// bottom_hat_closing alone (train_data_representations.py:40-72): the gap components between the per-cell closings and
// their weights.  root[t] = raster index (inside its image) of the component's first pixel, -1 outside the gaps — ranking the
// distinct roots gives measure.label's numbering; corr[t] = 0 outside, 1 in a gap, 0.8 on the 4-neighbour rim of a gap whose
// minor axis length is >= 3.  Same passes as inside mseg_label_distance.
__global__ void lb_bottom_hat_out_kernel(const int32_t* __restrict__ L, const LbGap* __restrict__ gaps, int N, int H, int W,
                                         int32_t* __restrict__ root_out, float* __restrict__ corr) {
  const size_t hw = (size_t)H * W, n = (size_t)N * hw;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
    const size_t s = t / hw;
    const int r = (int)(t - s * hw);
    const int y = r / W, x = r - y * W;
    const int32_t* l = L + s * hw;
    const int root = l[r];
    float v = 0.f;
    if (root >= 0) {
      const LbGap g = gaps[s * hw + root];
      const double cnt = (double)g.cnt;
      const double a = (double)(long long)(g.cnt * g.syy - g.sy * g.sy) / (cnt * cnt);
      const double c = (double)(long long)(g.cnt * g.sxx - g.sx * g.sx) / (cnt * cnt);
      const double b = (double)((long long)(g.cnt * g.sxy) - (long long)(g.sx * g.sy)) / (cnt * cnt);
      const double l2 = 0.5 * (a + c) - 0.5 * sqrt(4.0 * b * b + (a - c) * (a - c));
      const bool wide = 4.0 * sqrt(fmax(l2, 0.0)) >= 3.0;
      bool rim = false;
      if (wide) {
        rim = y == 0 || y == H - 1 || x == 0 || x == W - 1;
        if (!rim) rim = l[r - 1] != root || l[r + 1] != root || l[r - W] != root || l[r + W] != root;
      }
      v = rim ? 0.8f : 1.f;
    }
    root_out[t] = root;
    corr[t] = v;
  }
}

extern "C" int mseg_label_bottom_hat(const uint16_t* mask, int N, int H, int W, int32_t* root_out, float* corr_out, void* ws,
                                     size_t ws_bytes, void* stream) {
  if (!mask || !root_out || !corr_out || !ws) return MSEG_EINVAL;
  const size_t need = mseg_label_distance_workspace_bytes(N, H, W);
  if (need == 0) return MSEG_EINVAL;
  if (ws_bytes < need) return MSEG_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const LbLayout l = lb_layout(N, H, W);
  char* base = (char*)ws;
  LbGap* gaps = (LbGap*)(base + l.gaps);
  double* nb = (double*)(base + l.nb);
  uint8_t* bin = (uint8_t*)(base + l.bin);
  uint8_t* tmp = (uint8_t*)(base + l.tmp);
  uint8_t* gap = (uint8_t*)(base + l.gap);
  int32_t* L = (int32_t*)(base + l.L);
  const size_t hw = (size_t)H * W, px = (size_t)N * hw;
  const unsigned nbk = lb_blocks(px);
  if (hipMemsetAsync(ws, 0, l.zero_bytes, st) != hipSuccess) return MSEG_ELAUNCH;
  if (hipMemsetAsync(nb, 0, px * sizeof(double), st) != hipSuccess) return MSEG_ELAUNCH;    // no neighbour distances here
  hipLaunchKernelGGL(lb_close_cells_kernel, dim3(nbk), dim3(LB_BLOCK), 0, st, mask, N, H, W, bin);
  hipLaunchKernelGGL(lb_disk_kernel<false>, dim3(nbk), dim3(LB_BLOCK), 0, st, bin, bin, N, H, W, tmp);
  hipLaunchKernelGGL(lb_disk_kernel<true>, dim3(nbk), dim3(LB_BLOCK), 0, st, tmp, bin, N, H, W, gap);
  hipLaunchKernelGGL(lb_ccl_init_kernel, dim3(nbk), dim3(LB_BLOCK), 0, st, gap, N, hw, L);
  hipLaunchKernelGGL(lb_ccl_merge_kernel, dim3(nbk), dim3(LB_BLOCK), 0, st, gap, N, H, W, L);
  hipLaunchKernelGGL(lb_ccl_flatten_kernel, dim3(nbk), dim3(LB_BLOCK), 0, st, N, hw, L);
  hipLaunchKernelGGL(lb_gap_props_kernel, dim3(nbk), dim3(LB_BLOCK), 0, st, L, nb, N, H, W, gaps);
  hipLaunchKernelGGL(lb_bottom_hat_out_kernel, dim3(nbk), dim3(LB_BLOCK), 0, st, L, gaps, N, H, W, root_out, corr_out);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

// ---- j4_label (train_data_representations.py:157-188; Pena et al., ISBI 2020): background / cell / touching / gap -------------
// gap = bottom-hat of the binary mask with disk(se_radius) (closing ^ mask; scipy's border_value 0), touching = cell pixels
// whose (2k+1)^2 window holds more than one instance id (compute_neighbor_instances :191-216).  0 background, 1 cell,
// 2 touching, 3 gap.  Two passes: dilation of the binary mask, then erosion + neighbour count + classification.
__global__ void lb_j4_dilate_kernel(const uint16_t* __restrict__ mask, int N, int H, int W, int R, uint8_t* __restrict__ dil) {
  const size_t hw = (size_t)H * W, n = (size_t)N * hw;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
    const size_t s = t / hw;
    const int r = (int)(t - s * hw);
    const int y = r / W, x = r - y * W;
    const uint16_t* m = mask + s * hw;
    bool v = false;
    for (int dy = -R; dy <= R && !v; ++dy)
      for (int dx = -R; dx <= R; ++dx) {
        if (dy * dy + dx * dx > R * R) continue;
        const int yy = y + dy, xx = x + dx;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W && m[(size_t)yy * W + xx]) { v = true; break; }
      }
    dil[t] = v ? 1 : 0;
  }
}

__global__ void lb_j4_final_kernel(const uint16_t* __restrict__ mask, const uint8_t* __restrict__ dil, int N, int H, int W,
                                   int R, int K, uint8_t* __restrict__ out) {
  const size_t hw = (size_t)H * W, n = (size_t)N * hw;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
    const size_t s = t / hw;
    const int r = (int)(t - s * hw);
    const int y = r / W, x = r - y * W;
    const uint16_t* m = mask + s * hw;
    const uint8_t* d = dil + s * hw;
    const int lab = m[r];
    uint8_t cls;
    if (lab) {
      // distinct ids in the (2K+1)^2 window (zero padded): more than one -> touching
      bool other = false;
      for (int dy = -K; dy <= K && !other; ++dy)
        for (int dx = -K; dx <= K; ++dx) {
          const int yy = y + dy, xx = x + dx;
          if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
          const int q = m[(size_t)yy * W + xx];
          if (q && q != lab) { other = true; break; }
        }
      cls = other ? 2 : 1;
    } else {
      bool closed = true;                              // erosion of the dilated mask; outside the image counts as empty
      for (int dy = -R; dy <= R && closed; ++dy)
        for (int dx = -R; dx <= R; ++dx) {
          if (dy * dy + dx * dx > R * R) continue;
          const int yy = y + dy, xx = x + dx;
          if (yy < 0 || yy >= H || xx < 0 || xx >= W || !d[(size_t)yy * W + xx]) { closed = false; break; }
        }
      cls = closed ? 3 : 0;
    }
    out[t] = cls;
  }
}

extern "C" int mseg_label_j4(const uint16_t* mask, int N, int H, int W, int k_neighbors, int se_radius, uint8_t* tmp,
                             uint8_t* out, void* stream) {
  if (!mask || !tmp || !out || N <= 0 || H <= 0 || W <= 0 || k_neighbors < 0 || k_neighbors > 16 || se_radius < 1 ||
      se_radius > 16)
    return MSEG_EINVAL;
  const unsigned nbk = lb_blocks((size_t)N * H * W);
  hipLaunchKernelGGL(lb_j4_dilate_kernel, dim3(nbk), dim3(LB_BLOCK), 0, (hipStream_t)stream, mask, N, H, W, se_radius, tmp);
  hipLaunchKernelGGL(lb_j4_final_kernel, dim3(nbk), dim3(LB_BLOCK), 0, (hipStream_t)stream, mask, (const uint8_t*)tmp, N, H,
                     W, se_radius, k_neighbors, out);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

// cell_distance_label (train_data_representations.py:219-258): the cell-distance half alone, optionally clipped
extern "C" int mseg_label_cell_distance(const uint16_t* mask, int N, int H, int W, int search_radius, float clip_val,
                                        float* cell_out, void* ws, size_t ws_bytes, void* stream) {
  if (!mask || !cell_out || !ws || search_radius <= 0 || clip_val < 0.f) return MSEG_EINVAL;
  const size_t need = mseg_label_distance_workspace_bytes(N, H, W);
  if (need == 0) return MSEG_EINVAL;
  if (ws_bytes < need) return MSEG_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const LbLayout l = lb_layout(N, H, W);
  char* base = (char*)ws;
  LbCell* cells = (LbCell*)(base + l.cells);
  int32_t* prevd = (int32_t*)(base + l.prevd);
  int32_t* nextd = (int32_t*)(base + l.nextd);
  int32_t* d2c = (int32_t*)(base + l.d2c);
  int32_t* d2n = (int32_t*)(base + l.d2n);
  const unsigned nbk = lb_blocks((size_t)N * H * W);
  if (hipMemsetAsync(cells, 0, (size_t)N * LB_IDS * sizeof(LbCell), st) != hipSuccess) return MSEG_ELAUNCH;
  hipLaunchKernelGGL(lb_runs_kernel, dim3((N * H + 63) / 64), dim3(64), 0, st, mask, N, H, W, prevd, nextd);
  hipLaunchKernelGGL(lb_props_kernel, dim3(nbk), dim3(LB_BLOCK), 0, st, mask, N, H, W, cells);
  hipLaunchKernelGGL(lb_window_kernel, dim3(lb_blocks((size_t)N * LB_IDS)), dim3(LB_BLOCK), 0, st, cells, N, H, W,
                     search_radius);
  hipLaunchKernelGGL(lb_edt_kernel, dim3(nbk), dim3(LB_BLOCK), 0, st, mask, N, H, W, prevd, nextd, cells, d2c, d2n);
  hipLaunchKernelGGL(lb_value_kernel, dim3(nbk), dim3(LB_BLOCK), 0, st, mask, N, H, W, cells, d2c, d2n, cell_out,
                     (double*)nullptr, (double)clip_val);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

// ---- largest major axis length of a mask (CreateLabelsWorker.create_labels, src/training/train.py:73-78) -------------------------
// regionprops(mask)[i].major_axis_length = 4 * sqrt(larger eigenvalue of the normalised second central moments); the label
// creation derives the search radius of distance_label from its maximum over the cells.  Integer moment sums by atomics,
// eigenvalue in fp64; out_dev[n] = max over the cells of image n (0 for an empty mask).
struct LbMom {
  unsigned long long sy, sx, syy, sxx, sxy;
  unsigned cnt, pad;
};

__global__ void lb_cell_moments_kernel(const uint16_t* __restrict__ mask, int N, int H, int W, LbMom* __restrict__ mom) {
  const size_t hw = (size_t)H * W, n = (size_t)N * hw;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
    const int k = mask[t];
    if (!k) continue;
    const size_t s = t / hw;
    const int r = (int)(t - s * hw);
    const unsigned long long y = (unsigned long long)(r / W), x = (unsigned long long)(r % W);
    LbMom* c = mom + s * LB_IDS + k;
    atomicAdd(&c->cnt, 1u);
    atomicAdd(&c->sy, y);
    atomicAdd(&c->sx, x);
    atomicAdd(&c->syy, y * y);
    atomicAdd(&c->sxx, x * x);
    atomicAdd(&c->sxy, x * y);
  }
}

__global__ void lb_major_kernel(const LbMom* __restrict__ mom, int N, double* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)N * LB_IDS) return;
  const LbMom g = mom[i];
  if (!g.cnt) return;
  const double cnt = (double)g.cnt;
  const double a = (double)(long long)(g.cnt * g.syy - g.sy * g.sy) / (cnt * cnt);
  const double c = (double)(long long)(g.cnt * g.sxx - g.sx * g.sx) / (cnt * cnt);
  const double b = (double)((long long)(g.cnt * g.sxy) - (long long)(g.sx * g.sy)) / (cnt * cnt);
  const double l1 = 0.5 * (a + c) + 0.5 * sqrt(4.0 * b * b + (a - c) * (a - c));
  const double major = 4.0 * sqrt(fmax(l1, 0.0));
  // non-negative doubles order like their bit patterns
  atomicMax((unsigned long long*)(out + i / LB_IDS), (unsigned long long)__double_as_longlong(major));
}

extern "C" size_t mseg_label_major_axis_workspace_bytes(int N) {
  return N > 0 ? (size_t)N * LB_IDS * sizeof(LbMom) : 0;
}

extern "C" int mseg_label_max_major_axis(const uint16_t* mask, int N, int H, int W, double* out_dev, void* ws,
                                         size_t ws_bytes, void* stream) {
  if (!mask || !out_dev || !ws || N <= 0 || H <= 0 || W <= 0 || H > 32767 || W > 32767) return MSEG_EINVAL;
  if (ws_bytes < mseg_label_major_axis_workspace_bytes(N)) return MSEG_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(ws, 0, mseg_label_major_axis_workspace_bytes(N), st) != hipSuccess) return MSEG_ELAUNCH;
  if (hipMemsetAsync(out_dev, 0, (size_t)N * sizeof(double), st) != hipSuccess) return MSEG_ELAUNCH;
  hipLaunchKernelGGL(lb_cell_moments_kernel, dim3(lb_blocks((size_t)N * H * W)), dim3(LB_BLOCK), 0, st, mask, N, H, W,
                     (LbMom*)ws);
  hipLaunchKernelGGL(lb_major_kernel, dim3(lb_blocks((size_t)N * LB_IDS)), dim3(LB_BLOCK), 0, st, (const LbMom*)ws, N,
                     out_dev);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}
