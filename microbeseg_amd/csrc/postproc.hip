// postproc.hip — prediction -> instance labels on the GPU, bit-exact with the reference's CPU chain
//   distance_postprocessing  (src/inference/postprocessing.py:7-59)
//   boundary_postprocessing  (src/inference/postprocessing.py:62-90)
// i.e. scipy.ndimage.gaussian_filter(sigma 0.5) -> thresholds / tan -> skimage.measure.label (8-conn) -> small-seed
// removal -> relabel -> skimage.segmentation.watershed(markers, mask) (4-conn priority flood), semantics per
// SURVEY.md Appendix B (recovered from scipy 1.7.1 / scikit-image 0.18.3).
//
// HBM-bound integer/byte work — no MFMA.  Stages:
//   gaussian (2 passes, fp64 accumulate in scipy's order, no FMA contraction) + thresholds: streaming kernels;
//   connected components: lock-free union-find (atomicMin), root = smallest raster index of the component;
//   ordering of instance ids: prefix sum over "first pixel" flags (raster or column-major key space);
//   watershed: the reference is a strictly serial priority flood keyed (value, age).  Within one 4-connected
//     component of the mask the flood is independent of all other components as long as no two *initial* marker
//     pixels of that component carry exactly equal values (age-0 ties are broken by the global heap layout).  So:
//     fast path = one GPU thread per mask component, each running the exact textbook heap on its own segment
//     (64 floods per wavefront in lock step); the (rare) constellations in which such a tie can change a label
//     raise a taint flag (rules at pp_flood), and a tainted frame (always: the constant image of the boundary
//     method) is re-done by the exact global serial flood, on the device, without a host round trip.
#include "common.h"
#include <math.h>

#define PP_BLOCK 256
#define SCAN_ITEMS 8
#define SCAN_TILE (PP_BLOCK * SCAN_ITEMS)

// counters (int32 slots in workspace)
enum { C_NCOMP = 0, C_TOTAL = 1, C_KEPT = 2, C_TAINT = 3, C_NMCOMP = 4, C_SERIAL = 5, C_SCRATCH = 6, C_CONST_M = 7,
       C_CONST_STREAM = 8,   // 1: every marker key is equal (constant image): the marker phase runs as pp_flood_const_stream
       C_CONST_Q = 12, C_WORK_S = 16, C_WORK_L = 18, C_PART = 20, C_COUNT = 84 };   // C_WORK_*: two slots each; C_PART: 2 x 32

struct PPWs {
  float* tmp; float* cs;
  uint8_t* mask; uint8_t* seedb; uint8_t* hasm;
  int32_t* slab; int32_t* area; int32_t* ckey; int32_t* flag; int32_t* scan; int32_t* bsum;
  int32_t* markers; int32_t* mlab; int32_t* out; int32_t* carea; int32_t* hoff; int32_t* clist;
  int32_t* bymin; int32_t* bymax; int32_t* bxmin; int32_t* bxmax;
  unsigned long long* hkey; uint32_t* hidx;
  int32_t* counters;
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static size_t pp_carve(PPWs* w, void* base, int H, int W) {
  const size_t n = (size_t)H * W;
  const size_t nb = (n + SCAN_TILE - 1) / SCAN_TILE + 1;
  size_t off = 0;
  char* b = (char*)base;
#define PP_TAKE(field, type, count)                         \
  do {                                                      \
    off = align_up(off, 256);                               \
    if (w) w->field = (type*)(b + off);                     \
    off += sizeof(type) * (size_t)(count);                  \
  } while (0)
  PP_TAKE(tmp, float, n); PP_TAKE(cs, float, n);
  PP_TAKE(mask, uint8_t, n); PP_TAKE(seedb, uint8_t, n); PP_TAKE(hasm, uint8_t, n);
  PP_TAKE(slab, int32_t, n); PP_TAKE(area, int32_t, n); PP_TAKE(ckey, int32_t, n); PP_TAKE(flag, int32_t, n);
  PP_TAKE(scan, int32_t, n); PP_TAKE(bsum, int32_t, 2 * nb);
  PP_TAKE(markers, int32_t, n); PP_TAKE(mlab, int32_t, n); PP_TAKE(out, int32_t, n); PP_TAKE(carea, int32_t, n);
  PP_TAKE(hoff, int32_t, n); PP_TAKE(clist, int32_t, n);
  PP_TAKE(bymin, int32_t, n); PP_TAKE(bymax, int32_t, n); PP_TAKE(bxmin, int32_t, n); PP_TAKE(bxmax, int32_t, n);
  PP_TAKE(hkey, unsigned long long, n + 1); PP_TAKE(hidx, uint32_t, n + 1);
  PP_TAKE(counters, int32_t, C_COUNT);
#undef PP_TAKE
  return align_up(off, 256);
}

extern "C" size_t mseg_postproc_workspace_bytes(int H, int W) {
  if (H <= 0 || W <= 0 || (long long)H * W > 0x7fffffffLL) return 0;
  return pp_carve(nullptr, nullptr, H, W);
}

static inline unsigned pp_blocks(size_t n) {
  size_t b = (n + PP_BLOCK - 1) / PP_BLOCK;
  return (unsigned)(b < 1 ? 1 : b);
}

// ---- gaussian_filter(sigma 0.5), one axis: tmp = c*w2; tmp += (m2+p2)*w0; tmp += (m1+p1)*w1 in fp64, round to fp32 ----
__device__ __forceinline__ int pp_reflect(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) {
    if (i < 0) i = -i - 1;
    if (i >= n) i = 2 * n - 1 - i;
  }
  return i;
}

__global__ void pp_gauss_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W, int axis,
                                double w0, double w1, double w2) {
  const size_t n = (size_t)H * W;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int y = (int)(i / W), x = (int)(i - (size_t)y * W);
  double c, m1, p1, m2, p2;
  if (axis == 0) {
    c = in[i];
    m2 = in[(size_t)pp_reflect(y - 2, H) * W + x]; p2 = in[(size_t)pp_reflect(y + 2, H) * W + x];
    m1 = in[(size_t)pp_reflect(y - 1, H) * W + x]; p1 = in[(size_t)pp_reflect(y + 1, H) * W + x];
  } else {
    const float* row = in + (size_t)y * W;
    c = row[x];
    m2 = row[pp_reflect(x - 2, W)]; p2 = row[pp_reflect(x + 2, W)];
    m1 = row[pp_reflect(x - 1, W)]; p1 = row[pp_reflect(x + 1, W)];
  }
  // explicit rounding at every step (scipy's correlate1d is compiled without FMA)
  double t = __dmul_rn(c, w2);
  t = __dadd_rn(t, __dmul_rn(__dadd_rn(m2, p2), w0));
  t = __dadd_rn(t, __dmul_rn(__dadd_rn(m1, p1), w1));
  out[i] = (float)t;
}

// ---- thresholds -----------------------------------------------------------------------------------------------
__global__ void pp_distance_thresh_kernel(const float* __restrict__ border, const float* __restrict__ cs, size_t n,
                                          float th_cell, float th_seed, uint8_t* __restrict__ mask,
                                          uint8_t* __restrict__ seedb) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float b = border[i];
  b = b < 0.f ? 0.f : (b > 1.f ? 1.f : b);       // np.clip(border, 0, 1)
  const float c = cs[i];
  mask[i] = c > th_cell;
  const float b2 = __fmul_rn(b, b);              // border ** 2 in float32
  float t = (float)tan((double)b2);              // np.tan(float32): evaluated in fp64, rounded once
  if (t < 0.05f) t = 0.f;
  t = t < 0.f ? 0.f : (t > 1.f ? 1.f : t);
  seedb[i] = __fsub_rn(c, t) > th_seed;
}

__global__ void pp_boundary_thresh_kernel(const float* __restrict__ p, size_t n, uint8_t* __restrict__ mask,
                                          uint8_t* __restrict__ seedb, float* __restrict__ img) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float p0 = p[3 * i], p1 = p[3 * i + 1], p2 = p[3 * i + 2];
  int arg = 0;                                   // np.argmax: first maximum wins
  float best = p0;
  if (p1 > best) { best = p1; arg = 1; }
  if (p2 > best) { best = p2; arg = 2; }
  const bool m = (arg == 1);
  mask[i] = m;
  seedb[i] = __fmul_rn(p1, __fsub_rn(1.f, p2)) > 0.5f;
  img[i] = m ? 1.f : 0.f;                        // watershed(image=mask): constant inside the mask
}

__global__ void pp_negate_kernel(const float* __restrict__ in, float* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = -in[i];
}

// ---- union-find connected components ------------------------------------------------------------------------------
__device__ __forceinline__ int uf_find(const int32_t* L, int i) {
  int p = __hip_atomic_load(&L[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while (p != i) {
    i = p;
    p = __hip_atomic_load(&L[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return i;
}

__device__ __forceinline__ void uf_union(int32_t* L, int a, int b) {
  for (;;) {
    a = uf_find(L, a);
    b = uf_find(L, b);
    if (a == b) return;
    if (a > b) { const int t = a; a = b; b = t; }   // a < b: hang b under a
    const int old = atomicMin(&L[b], a);
    if (old == b) return;
    b = old;                                        // somebody else re-rooted b meanwhile: retry with its new parent
  }
}

__global__ void pp_ccl_init_kernel(const uint8_t* __restrict__ bin, int32_t* __restrict__ L, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) L[i] = bin[i] ? (int32_t)i : -1;
}

template <bool EIGHT>
__global__ void pp_ccl_merge_kernel(const uint8_t* __restrict__ bin, int32_t* __restrict__ L, int H, int W) {
  const size_t n = (size_t)H * W;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !bin[i]) return;
  const int y = (int)(i / W), x = (int)(i - (size_t)y * W);
  if (x > 0 && bin[i - 1]) uf_union(L, (int)i, (int)i - 1);
  if (y > 0) {
    if (bin[i - W]) uf_union(L, (int)i, (int)i - W);
    if (EIGHT) {
      if (x > 0 && bin[i - W - 1]) uf_union(L, (int)i, (int)i - W - 1);
      if (x + 1 < W && bin[i - W + 1]) uf_union(L, (int)i, (int)i - W + 1);
    }
  }
}

__global__ void pp_ccl_flatten_kernel(int32_t* __restrict__ L, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (L[i] >= 0) L[i] = uf_find(L, (int)i);       // roots are fixed points, so concurrent flattening is benign
}

// ---- seed statistics, filtering, ordering ----------------------------------------------------------------------------
// Per-pixel statistics of labelled components are atomics onto the component's root — hundreds of pixels per root.  Lanes
// of a wavefront hold 64 consecutive pixels, so a component's pixels arrive in horizontal RUNS: only the first lane of a
// run (same root as its left neighbour, same image row) issues the atomics, with the run's length.
// Returns the run length for the run's first lane, 0 for every other lane.
__device__ __forceinline__ int pp_run_length(int r, int x) {
  const int lane = threadIdx.x & 63;
  const int rp = __shfl_up(r, 1, 64);
  const bool valid = r >= 0;
  const bool head = valid && !(lane > 0 && rp == r && x > 0);
  const unsigned long long cont = __ballot(valid && !head);          // lanes that continue their left neighbour's run
  if (!head) return 0;
  const unsigned long long m = lane == 63 ? 0ull : (cont >> (lane + 1));
  return 1 + (int)__builtin_ctzll(~m);                                // trailing ones of m (m has at most 63 - lane bits)
}


__global__ void pp_fill_kernel(int32_t* __restrict__ a, int32_t v, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = v;
}

__global__ void pp_seed_stats_kernel(const int32_t* __restrict__ L, int H, int W, int col_major,
                                     int32_t* __restrict__ area, int32_t* __restrict__ ckey,
                                     int32_t* __restrict__ counters) {
  const size_t n = (size_t)H * W;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int r = i < n ? L[i] : -1;
  const int y = i < n ? (int)(i / W) : 0, x = i < n ? (int)(i - (size_t)y * W) : 0;
  const int run = pp_run_length(r, x);
  if (run > 0) {                                            // first pixel of a horizontal run: the run's smallest key
    atomicAdd(&area[r], run);
    atomicMin(&ckey[r], col_major ? x * H + y : (int)i);
  }
  // the two frame-wide counts.  Atomics onto ONE address are served at ~10^8 per second: one per seed pixel, or even one
  // per wavefront (30 k), is what this kernel's time consisted of.  A workgroup combines its four waves in LDS and adds to
  // one of 32 partial counters; pp_seed_totals_kernel sums those.
  __shared__ int sh_cnt[2];
  if (threadIdx.x < 2) sh_cnt[threadIdx.x] = 0;
  __syncthreads();
  const unsigned long long seeds = __ballot(r >= 0), roots = __ballot(r >= 0 && r == (int)i);
  if ((threadIdx.x & 63) == 0) {
    if (seeds) atomicAdd(&sh_cnt[0], __popcll(seeds));
    if (roots) atomicAdd(&sh_cnt[1], __popcll(roots));
  }
  __syncthreads();
  if (threadIdx.x < 2 && sh_cnt[threadIdx.x])
    atomicAdd(&counters[C_PART + 32 * threadIdx.x + (blockIdx.x & 31)], sh_cnt[threadIdx.x]);
}

__global__ void pp_seed_totals_kernel(int32_t* __restrict__ counters) {
  const int k = threadIdx.x >> 5, j = threadIdx.x & 31;           // 64 threads: [total | components] x 32 partials
  int v = counters[C_PART + 32 * k + j];
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if (j == 0) counters[k == 0 ? C_TOTAL : C_NCOMP] = v;
}

__device__ __forceinline__ bool pp_keep(int area, const int32_t* counters, int distance_rule) {
  double min_area = 4.0;                                      // boundary: area <= 4 removed
  if (distance_rule) {
    const int nc = counters[C_NCOMP];
    double m = nc > 0 ? 0.10 * ((double)counters[C_TOTAL] / (double)nc) : 0.0;   // 0.10 * np.mean(areas)
    min_area = m > 4.0 ? m : 4.0;                             // np.maximum(min_area, 4)
  }
  return !((double)area <= min_area);
}

__global__ void pp_seed_select_kernel(const int32_t* __restrict__ L, size_t n, const int32_t* __restrict__ area,
                                      const int32_t* __restrict__ ckey, int distance_rule,
                                      int32_t* __restrict__ flag, int32_t* __restrict__ counters) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || L[i] != (int)i) return;
  if (pp_keep(area[i], counters, distance_rule)) {
    flag[ckey[i]] = 1;
    atomicAdd(&counters[C_KEPT], 1);
  }
}

__global__ void pp_markers_kernel(const int32_t* __restrict__ L, size_t n, const int32_t* __restrict__ area,
                                  const int32_t* __restrict__ ckey, const int32_t* __restrict__ scan,
                                  const uint8_t* __restrict__ mask, int distance_rule,
                                  const int32_t* __restrict__ counters, int32_t* __restrict__ markers,
                                  int32_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int id = 0;
  const int r = L[i];
  if (r >= 0 && mask[i] && pp_keep(area[r], counters, distance_rule)) id = scan[ckey[r]] + 1;   // markers * mask
  markers[i] = id;
  out[i] = id;
}

// ---- exclusive prefix sum (3 kernels) ---------------------------------------------------------------------------------
__global__ __launch_bounds__(PP_BLOCK) void pp_scan_tile_kernel(const int32_t* __restrict__ in,
                                                                int32_t* __restrict__ out,
                                                                int32_t* __restrict__ bsum, size_t n) {
  __shared__ int32_t sh[PP_BLOCK];
  const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
  int32_t v[SCAN_ITEMS];
  int32_t s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    v[k] = (base + k < n) ? in[base + k] : 0;
    s += v[k];
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 1; o < PP_BLOCK; o <<= 1) {
    const int32_t t = threadIdx.x >= (unsigned)o ? sh[threadIdx.x - o] : 0;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  int32_t run = sh[threadIdx.x] - s;   // exclusive offset of this thread within the tile
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    if (base + k < n) out[base + k] = run;
    run += v[k];
  }
  if (threadIdx.x == PP_BLOCK - 1) bsum[blockIdx.x] = sh[PP_BLOCK - 1];
}

__global__ __launch_bounds__(PP_BLOCK) void pp_scan_sums_kernel(int32_t* __restrict__ bsum, int nb,
                                                                int32_t* __restrict__ total) {
  __shared__ int32_t sh[PP_BLOCK];
  __shared__ int32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nb; base += PP_BLOCK) {
    const int i = base + threadIdx.x;
    const int32_t v = i < nb ? bsum[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < PP_BLOCK; o <<= 1) {
      const int32_t t = threadIdx.x >= (unsigned)o ? sh[threadIdx.x - o] : 0;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nb) bsum[i] = carry + sh[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 0) carry += sh[PP_BLOCK - 1];
    __syncthreads();
  }
  if (threadIdx.x == 0 && total) *total = carry;
}

__global__ void pp_scan_add_kernel(int32_t* __restrict__ out, const int32_t* __restrict__ bsum, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] += bsum[i / SCAN_TILE];
}

static int pp_exclusive_scan(const int32_t* in, int32_t* out, int32_t* bsum, size_t n, int32_t* total_dev,
                             hipStream_t st) {
  const int nb = (int)((n + SCAN_TILE - 1) / SCAN_TILE);
  hipLaunchKernelGGL(pp_scan_tile_kernel, dim3(nb), dim3(PP_BLOCK), 0, st, in, out, bsum, n);
  hipLaunchKernelGGL(pp_scan_sums_kernel, dim3(1), dim3(PP_BLOCK), 0, st, bsum, nb, total_dev);
  hipLaunchKernelGGL(pp_scan_add_kernel, dim3(pp_blocks(n)), dim3(PP_BLOCK), 0, st, out, (const int32_t*)bsum, n);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

// ---- watershed ----------------------------------------------------------------------------------------------------------
// heap element = 64-bit key (order-preserving float bits << 32 | age) + pixel index; smaller key pops first.
__device__ __forceinline__ unsigned long long pp_key(float v, unsigned age) {
  if (v == 0.f) v = 0.f;                               // -0.0 == +0.0 for the reference's float compare
  unsigned u = __float_as_uint(v);
  u ^= (u >> 31) ? 0xffffffffu : 0x80000000u;          // monotone map float -> uint
  return ((unsigned long long)u << 32) | age;
}

struct PPHeap {
  unsigned long long* key;
  uint32_t* idx;
  int n;
  int taint;
};

__device__ __forceinline__ bool pp_less(PPHeap& h, unsigned long long a, unsigned long long b) {
  (void)h;
  return a < b;
}

__device__ __forceinline__ void pp_push(PPHeap& h, unsigned long long k, uint32_t ix) {
  int child = h.n++;
  h.key[child] = k; h.idx[child] = ix;
  while (child > 0) {
    const int parent = (child + 1) / 2 - 1;
    const unsigned long long kp = h.key[parent];
    if (pp_less(h, k, kp)) {
      h.key[child] = kp; h.idx[child] = h.idx[parent];
      h.key[parent] = k; h.idx[parent] = ix;
      child = parent;
    } else break;
  }
}

__device__ __forceinline__ void pp_pop(PPHeap& h, unsigned long long& tk, uint32_t& ti) {
  tk = h.key[0]; ti = h.idx[0];
  const int n = --h.n;
  if (n == 0) return;
  const unsigned long long lk = h.key[n];
  const uint32_t li = h.idx[n];
  h.key[0] = lk; h.idx[0] = li;
  int i = 0;
  for (;;) {
    const int l = 2 * i + 1, r = 2 * i + 2;
    if (l >= n) break;
    int smallest = i;
    unsigned long long ks = lk;                       // key currently at position i is always `last`
    const unsigned long long kl = h.key[l];
    if (pp_less(h, kl, ks)) { smallest = l; ks = kl; }
    if (r < n) {
      const unsigned long long kr = h.key[r];
      if (pp_less(h, kr, ks)) { smallest = r; ks = kr; }
    }
    if (smallest == i) break;
    h.key[i] = ks; h.idx[i] = h.idx[smallest];
    h.key[smallest] = lk; h.idx[smallest] = li;
    i = smallest;
  }
}

// Flood from a filled heap.
//
// TRACK = true (per-component fast path) watches for the only situations in which the result can depend on how the
// reference's single global heap orders two *initial* markers X, Y of equal value v in one component (all other keys
// are unique: every later push gets a fresh age).  Popped here in the order X, Y; the reference may use Y, X.
// What the swap changes: (a) who claims a pixel adjacent to both; (b) who claims the pixels of X's / Y's *cascade*
// (pixels with value < v that pop before the other marker); (c) the relative AGES of {pushes of X and its cascade}
// versus {pushes of Y and its cascade} — which decides a later pop order only between entries of equal VALUE.
// (a), (b) can change a label only if X and Y carry different labels.  Conservative rules:
//   1   different labels, Y meets a neighbour that X has just labelled                           -> taint
//   2a  different labels, something popped between X and Y (X's cascade)                          -> taint
//   2b  different labels, a pop with value < v after Y (Y's cascade)                              -> taint
//   3a  no cascade: a direct push of X and a direct push of Y have equal values                  -> taint (diff. labels)
//                                                                                                   / sticky (same label)
//   3b  same label with a cascade on either side, or a chain of >= 3 tied markers                -> sticky
//   3   sticky set: two consecutive pops with equal value anywhere later in this component       -> taint
// A taint sends the whole frame to the exact serial flood.  Tie-free float data never taints; the randomised
// tie-stress test checks frames that contain ties and stay on this path against the oracle.
template <bool TRACK>
__device__ __forceinline__ void pp_flood(PPHeap& h, const float* __restrict__ img, const uint8_t* __restrict__ mask,
                                         int32_t* __restrict__ out, int H, int W, unsigned age) {
  unsigned long long prev0_key = ~0ull;   // last popped initial marker
  int prev0_label = 0;
  int since_prev0 = 0;                    // pops since that marker
  bool prev0_tied = false;                // that marker was itself the second of a tie
  uint32_t pushed_prev[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
  unsigned pushed_prev_val[4] = {0u, 0u, 0u, 0u};
  bool sticky = false;
  unsigned prev_val = 0xffffffffu;        // value bits of the previous pop
  unsigned watch_val = 0u;                // value of the last marker tie (0 = none) ...
  bool watch_diff = false;                // ... and whether its labels differed
  while (h.n > 0) {
    unsigned long long k; uint32_t e;
    pp_pop(h, k, e);
    const int y = (int)(e / (unsigned)W), x = (int)(e - (unsigned)y * W);
    const int lab = out[e];
    bool tie = false, tie_diff = false, had_cascade = false;
    uint32_t pushed_now[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    unsigned pushed_now_val[4] = {0u, 0u, 0u, 0u};
    if (TRACK) {
      const unsigned val = (unsigned)(k >> 32), a = (unsigned)k;
      if (sticky && val == prev_val) h.taint |= 16;                              // rule 3
      if (watch_val && val < watch_val) {                                        // Y's cascade
        if (watch_diff) h.taint |= 4;                                            // rule 2b
        else sticky = true;                                                      // rule 3b
      }
      if (a == 0u) {
        if (k == prev0_key) {
          tie = true;
          tie_diff = (lab != prev0_label);
          if (prev0_tied) sticky = true;                                         // rule 3b (chain of ties)
          had_cascade = since_prev0 > 0;
          if (since_prev0 > 0) {
            if (tie_diff) h.taint |= 2;                                          // rule 2a
            else sticky = true;                                                  // rule 3b
          }
          watch_val = val; watch_diff = tie_diff;
        } else {
          watch_val = 0u;
        }
        prev0_tied = tie;
        prev0_key = k; prev0_label = lab; since_prev0 = 0;
      } else {
        ++since_prev0;
      }
      prev_val = val;
    }
    // neighbour order of skimage's raveled offsets: up, left, right, down
#define PP_VISIT(COND, J, SLOT)                                                              \
    if (COND) {                                                                              \
      const uint32_t j = (J);                                                                \
      if (mask[j]) {                                                                         \
        if (out[j] == 0) {                                                                   \
          const unsigned long long nk = pp_key(img[j], ++age);                               \
          out[j] = lab; pp_push(h, nk, j);                                                   \
          if (TRACK) { pushed_now[SLOT] = j; pushed_now_val[SLOT] = (unsigned)(nk >> 32); }  \
        } else if (TRACK && tie_diff &&                                                      \
                   (j == pushed_prev[0] || j == pushed_prev[1] || j == pushed_prev[2] || j == pushed_prev[3])) { \
          h.taint |= 1;                                                     /* rule 1 */     \
        }                                                                                    \
      }                                                                                      \
    }
    PP_VISIT(y > 0, e - W, 0)
    PP_VISIT(x > 0, e - 1, 1)
    PP_VISIT(x + 1 < W, e + 1, 2)
    PP_VISIT(y + 1 < H, e + W, 3)
#undef PP_VISIT
    if (TRACK && (unsigned)k == 0u) {
      if (tie && !had_cascade) {                                                 // rule 3a
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (pushed_now[p] != 0xffffffffu && pushed_prev[q] != 0xffffffffu &&
                pushed_now_val[p] == pushed_prev_val[q]) {
              if (tie_diff) h.taint |= 8; else sticky = true;
            }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) { pushed_prev[q] = pushed_now[q]; pushed_prev_val[q] = pushed_now_val[q]; }
    }
  }
}

// mask-component statistics: pixel count, bounding box, "has a marker"
__global__ void pp_mcomp_stats_kernel(const int32_t* __restrict__ mlab, const int32_t* __restrict__ markers, int H,
                                      int W, int32_t* __restrict__ carea, int32_t* __restrict__ bymin,
                                      int32_t* __restrict__ bymax, int32_t* __restrict__ bxmin,
                                      int32_t* __restrict__ bxmax, uint8_t* __restrict__ hasm) {
  const size_t n = (size_t)H * W;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int r = i < n ? mlab[i] : -1;
  const int y = i < n ? (int)(i / W) : 0, x = i < n ? (int)(i - (size_t)y * W) : 0;
  const int run = pp_run_length(r, x);                      // all 64 lanes take part
  if (run > 0) {
    atomicAdd(&carea[r], run);
    atomicMin(&bymin[r], y); atomicMax(&bymax[r], y);
    atomicMin(&bxmin[r], x); atomicMax(&bxmax[r], x + run - 1);
  }
  if (r >= 0 && markers[i] != 0) hasm[r] = 1;
}

// flag = 1 at roots of components that contain a marker; carea zeroed elsewhere (it becomes the heap-size scan input)
__global__ void pp_mcomp_flag_kernel(const int32_t* __restrict__ mlab, const uint8_t* __restrict__ hasm, size_t n,
                                     int32_t* __restrict__ flag, int32_t* __restrict__ carea) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool on = (mlab[i] == (int)i) && hasm[i];
  flag[i] = on ? 1 : 0;
  if (!on) carea[i] = 0;
}

__global__ void pp_mcomp_list_kernel(const int32_t* __restrict__ flag, const int32_t* __restrict__ scan, size_t n,
                                     int32_t* __restrict__ clist) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flag[i]) clist[scan[i]] = (int32_t)i;
}

// ---- fast path: one WAVEFRONT per mask component ---------------------------------------------------------------------------
// Every key (value, age) of a component's flood is unique except between tied initial markers (the taint rules above), so
// ANY priority queue pops the same sequence as the reference's binary heap.  A binary heap is the worst shape for a GPU
// (a sift is ~20 dependent memory round trips for one lane); this queue is built for a 64-lane wave instead:
//   * entries live in LDS as a 64-column table, lane l owns column l (slot = row * 64 + l) and caches its column's minimum
//     in registers; pop = a wave-wide DPP min-reduction over the 64 cached minima (value word, then age word), the owner
//     lane fills the hole with its last row and rescans its few rows; push = append to the column that was just popped or
//     to the shortest column (so a column never holds more than total / 64 + 1 rows);
//   * the component's bounding box (+ 1 px rim) of the image is staged in LDS once, coalesced; a tile word is the
//     pixel's value while it can still be claimed and a sentinel afterwards (not in the mask / labelled / rim), so the
//     four neighbour probes of a pop are four lanes reading LDS — no global round trip inside the loop, labels are
//     write-only stores;
//   * rows beyond the LDS table spill into the component's segment of the global heap arrays (owner lane only), and a
//     component whose box exceeds the tile probes global memory (workgroup-scope acquire / release on the labels): both
//     paths are exact, only slower.
// Two launches share the component list: small boxes (3 workgroups per CU) and large ones (one per CU, 120 KB tile).
#define PPW_ROWS 16
#define PPW_CAP (PPW_ROWS * 64)
#define PPW_SENT 0xffffffffu
#define PPW_TILE_S 8192
#define PPW_TILE_L 30720

static int g_ppw_rows = PPW_ROWS;         // test hooks (mseg_postproc_tuning): force the spill / global-probe paths
static int g_ppw_tile_s = PPW_TILE_S;
static int g_ppw_tile_l = PPW_TILE_L;

__device__ __forceinline__ unsigned ppw_wave_min(unsigned v) {
  // min over the 64 lanes in six DPP steps, each ONE instruction (v_min_u32 with a DPP-permuted first operand, in place;
  // the compiler's own lowering of update_dpp + min is a copy, a DPP move and the min per step, and the pop of the flood is
  // a chain of dependent instructions).  s_nop 1: two wait states between a VALU write and its DPP read.
  asm volatile(
      "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v));
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// One component.  LDS: the box is staged in s_tile and a queue entry names its pixel by tile coordinates (row << 16 | col);
// otherwise entries carry the raster index and the probes go to global memory.  SPILL: queue rows >= rows_lds exist.
template <bool LDS, bool SPILL>
__device__ __forceinline__ int ppw_component(
    unsigned long long* s_key, uint32_t* s_idx, int32_t* s_lab, uint32_t* s_tile, const float* __restrict__ img,
    const uint8_t* __restrict__ mask, const int32_t* __restrict__ mlab, unsigned long long* __restrict__ gkey,
    uint32_t* __restrict__ gidx, int32_t* __restrict__ glab, int32_t* out, int H, int W, int root, int y0, int x0, int th,
    int tw, int rows_lds) {
  const int lane = threadIdx.x;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const long long px = (long long)th * tw;
  int taint = 0;
  // column `lane` of the queue: rows < rows_lds in LDS, the rest in the component's global segment
  int cnt = 0, mrow = 0;
  unsigned long long mk = ~0ull;
  auto q_key = [&](int r) { return (!SPILL || r < rows_lds) ? s_key[r * 64 + lane] : gkey[(r - rows_lds) * 64 + lane]; };
  auto q_idx = [&](int r) { return (!SPILL || r < rows_lds) ? s_idx[r * 64 + lane] : gidx[(r - rows_lds) * 64 + lane]; };
  auto q_lab = [&](int r) { return (!SPILL || r < rows_lds) ? s_lab[r * 64 + lane] : glab[(r - rows_lds) * 64 + lane]; };
  auto q_set_at = [&](int r, int col, unsigned long long k, uint32_t ix, int lb) {
    if (!SPILL || r < rows_lds) { s_key[r * 64 + col] = k; s_idx[r * 64 + col] = ix; s_lab[r * 64 + col] = lb; }
    else { const int g = (r - rows_lds) * 64 + col; gkey[g] = k; gidx[g] = ix; glab[g] = lb; }
  };
  auto q_rescan = [&]() {
    mk = ~0ull; mrow = 0;
#pragma unroll 4
    for (int r = 0; r < cnt; ++r) {
      const unsigned long long k = q_key(r);
      if (k < mk) { mk = k; mrow = r; }
    }
  };

  // ---- stage the box, collect the initial markers (all with age 0) ----
  int total = 0;
  int ty = 0, tx = lane;                                  // tile coordinates of p = p0 + lane, kept without divisions
  while (tx >= tw) { tx -= tw; ++ty; }
  const int sy = 64 / tw, sx = 64 - sy * tw;              // 64 = sy * tw + sx
  for (long long p0 = 0; p0 < px; p0 += 64) {
    const long long p = p0 + lane;
    bool is_marker = false;
    unsigned long long k0 = 0; uint32_t j0 = 0; int l0 = 0;
    if (p < px) {
      uint32_t tv = PPW_SENT;
      if (ty > 0 && ty < th - 1 && tx > 0 && tx < tw - 1) {
        const uint32_t j = (uint32_t)(y0 - 1 + ty) * (uint32_t)W + (uint32_t)(x0 - 1 + tx);
        if (mask[j]) {
          const int m = out[j];                           // = markers[j] (pp_markers_kernel)
          const float v = img[j];
          if (m != 0) {
            if (mlab[j] == root) { is_marker = true; k0 = pp_key(v, 0u); j0 = LDS ? ((uint32_t)ty << 16 | (uint32_t)tx) : j; l0 = m; }
          } else {
            tv = __float_as_uint(v);
          }
        }
      }
      if (LDS) s_tile[p] = tv;
    }
    const unsigned long long bal = __ballot(is_marker);
    if (is_marker) {
      const int slot = total + __popcll(bal & lt_mask);
      q_set_at(slot >> 6, slot & 63, k0, j0, l0);
    }
    total += __popcll(bal);
    ty += sy; tx += sx;
    if (tx >= tw) { tx -= tw; ++ty; }
  }
  if (SPILL) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // spilled rows were written by other lanes
  __syncthreads();
  if (SPILL) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  cnt = total > lane ? (total - lane + 63) >> 6 : 0;
  q_rescan();

  // ---- flood ----
  const int dy = lane == 0 ? -1 : (lane == 3 ? 1 : 0), dx = lane == 1 ? -1 : (lane == 2 ? 1 : 0);
  const int dt = dy * tw + dx, dj = dy * W + dx;
  const int jbase = (y0 - 1) * W + (x0 - 1);
  unsigned age = 0u;
  unsigned long long prev0_key = ~0ull;   // tracking state of pp_flood<true>, uniform over the wave
  int prev0_label = 0, since_prev0 = 0;
  bool prev0_tied = false, sticky = false, watch_diff = false;
  uint32_t pushed_prev[4] = {PPW_SENT, PPW_SENT, PPW_SENT, PPW_SENT};
  unsigned pushed_prev_val[4] = {0u, 0u, 0u, 0u};
  unsigned prev_val = 0xffffffffu, watch_val = 0u;
  while (total > 0) {
    // pop: smallest (value, age) over the cached column minima
    const unsigned hi = cnt > 0 ? (unsigned)(mk >> 32) : 0xffffffffu;
    const unsigned mh = ppw_wave_min(hi);
    const bool cand = cnt > 0 && hi == mh;
    const unsigned long long cands = __ballot(cand);
    unsigned ml;
    int winner;
    if ((cands & (cands - 1ull)) == 0ull) {                // one column holds the smallest value: no age comparison needed
      winner = __builtin_amdgcn_readfirstlane(__ffsll((long long)cands) - 1);
      ml = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)mk, winner);
    } else {                                               // equal values in several columns: the smallest age wins
      ml = ppw_wave_min(cand ? (unsigned)mk : 0xffffffffu);
      winner = __builtin_amdgcn_readfirstlane(__ffsll((long long)__ballot(cand && (unsigned)mk == ml)) - 1);
    }
    const unsigned long long k = ((unsigned long long)mh << 32) | ml;
    uint32_t e = 0; int lab = 0;
    if (lane == winner) {
      e = q_idx(mrow); lab = q_lab(mrow);
      const int last = cnt - 1;
      if (mrow != last) q_set_at(mrow, lane, q_key(last), q_idx(last), q_lab(last));
      cnt = last;
      q_rescan();
    }
    e = (uint32_t)__builtin_amdgcn_readlane((int)e, winner);
    lab = __builtin_amdgcn_readlane(lab, winner);
    --total;

    bool tie = false, tie_diff = false, had_cascade = false;
    {
      const unsigned val = mh, a = ml;
      if (sticky && val == prev_val) taint |= 16;                              // rule 3
      if (watch_val && val < watch_val) {                                      // Y's cascade
        if (watch_diff) taint |= 4;                                            // rule 2b
        else sticky = true;                                                    // rule 3b
      }
      if (a == 0u) {
        if (k == prev0_key) {
          tie = true;
          tie_diff = (lab != prev0_label);
          if (prev0_tied) sticky = true;                                       // rule 3b (chain of ties)
          had_cascade = since_prev0 > 0;
          if (since_prev0 > 0) {
            if (tie_diff) taint |= 2;                                          // rule 2a
            else sticky = true;                                                // rule 3b
          }
          watch_val = val; watch_diff = tie_diff;
        } else {
          watch_val = 0u;
        }
        prev0_tied = tie;
        prev0_key = k; prev0_label = lab; since_prev0 = 0;
      } else {
        ++since_prev0;
      }
      prev_val = val;
    }

    // neighbours in skimage's order up, left, right, down: lanes 0..3
    bool push = false;
    uint32_t jn = 0;                                       // the neighbour's name in this mode (tile coordinates / raster index)
    unsigned long long nk = 0;
    float v = 0.f;
    if (lane < 4) {
      bool valid;
      if (LDS) {
        const int ey = (int)(e >> 16), ex = (int)(e & 0xffffu);
        const int t = ey * tw + ex + dt;
        const uint32_t tv = s_tile[t];
        jn = (uint32_t)(ey + dy) << 16 | (uint32_t)(ex + dx);
        valid = true;                                      // a rim pixel is never in pushed_prev
        push = tv != PPW_SENT;
        v = __uint_as_float(tv);
        if (push) { s_tile[t] = PPW_SENT; out[jbase + ey * W + ex + dj] = lab; }
      } else {
        const int y = (int)(e / (unsigned)W), x = (int)(e - (unsigned)y * W);
        valid = (unsigned)(y + dy) < (unsigned)H && (unsigned)(x + dx) < (unsigned)W;
        jn = (uint32_t)((int)e + dj);
        if (valid && mask[jn]) {
          push = __hip_atomic_load(&out[jn], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0;
          if (push) {
            v = img[jn];
            __hip_atomic_store(&out[jn], lab, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
      if (!push && tie_diff && valid &&
          (jn == pushed_prev[0] || jn == pushed_prev[1] || jn == pushed_prev[2] || jn == pushed_prev[3]))
        taint |= 1;                                                            // rule 1
    }
    const unsigned pb = (unsigned)__ballot(push) & 0xfu;
    if (push) nk = pp_key(v, age + 1u + (unsigned)__popc(pb & ((1u << lane) - 1u)));
    age += (unsigned)__popc(pb);
    uint32_t pushed_now[4] = {PPW_SENT, PPW_SENT, PPW_SENT, PPW_SENT};
    unsigned pushed_now_val[4] = {0u, 0u, 0u, 0u};
    bool first = true;
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
      if (!((pb >> sl) & 1u)) continue;                                        // uniform
      const unsigned khi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(nk >> 32), sl);
      const unsigned klo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)nk, sl);
      const uint32_t js = (uint32_t)__builtin_amdgcn_readlane((int)jn, sl);
      const unsigned long long ks = ((unsigned long long)khi << 32) | klo;
      pushed_now[sl] = js; pushed_now_val[sl] = khi;
      int t = winner;                                     // the column that just lost a row ...
      if (!first) t = (int)(ppw_wave_min(((unsigned)cnt << 6) | (unsigned)lane) & 63u);   // ... else the shortest one
      first = false;
      if (lane == t) {
        q_set_at(cnt, lane, ks, js, lab);
        if (ks < mk) { mk = ks; mrow = cnt; }
        ++cnt;
      }
      ++total;
    }
    if (ml == 0u) {
      if (tie && !had_cascade) {                                               // rule 3a
#pragma unroll
        for (int pn = 0; pn < 4; ++pn)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (pushed_now[pn] != PPW_SENT && pushed_prev[q] != PPW_SENT && pushed_now_val[pn] == pushed_prev_val[q]) {
              if (tie_diff) taint |= 8; else sticky = true;
            }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) { pushed_prev[q] = pushed_now[q]; pushed_prev_val[q] = pushed_now_val[q]; }
    }
  }
  __syncthreads();                                        // the next component reuses the tile
  return taint;
}

#define PPW_BIG_AREA 512      // components at least this large are started first (the longest one bounds the launch)

template <int TILE_PX>
__global__ __launch_bounds__(64) void pp_flood_wave_kernel(
    const float* __restrict__ img, const uint8_t* __restrict__ mask, const int32_t* __restrict__ mlab,
    const int32_t* __restrict__ clist, const int32_t* __restrict__ hoff, const int32_t* __restrict__ carea,
    const int32_t* __restrict__ bymin, const int32_t* __restrict__ bymax, const int32_t* __restrict__ bxmin,
    const int32_t* __restrict__ bxmax, unsigned long long* __restrict__ hkey, uint32_t* __restrict__ hidx,
    int32_t* __restrict__ hlab, int32_t* out, int H, int W, int32_t* __restrict__ counters, int work_slot,
    long long px_lo, long long px_hi, int tile_px, int rows_lds) {
  __shared__ unsigned long long s_key[PPW_CAP];
  __shared__ uint32_t s_idx[PPW_CAP];
  __shared__ int32_t s_lab[PPW_CAP];
  __shared__ uint32_t s_tile[TILE_PX];
  if (counters[C_SERIAL]) return;                        // caller forces the exact serial path (boundary method)
  const int lane = threadIdx.x;
  const int ncomp = counters[C_NMCOMP];
  int taint = 0;
  for (int pass = 0; pass < 2; ++pass) {
    for (;;) {
      int c = 0;
      if (lane == 0) c = atomicAdd(&counters[work_slot + pass], 1);
      c = __builtin_amdgcn_readfirstlane(c);
      if (c >= ncomp) break;
      const int root = clist[c];
      const int area = carea[root];
      if ((area >= PPW_BIG_AREA) != (pass == 0)) continue;
      const int y0 = bymin[root], y1 = bymax[root], x0 = bxmin[root], x1 = bxmax[root];
      const int th = y1 - y0 + 3, tw = x1 - x0 + 3;       // bounding box + 1 px rim
      const long long px = (long long)th * tw;
      if (px <= px_lo || px > px_hi) continue;            // the other launch's class
      unsigned long long* gkey = hkey + hoff[root];
      uint32_t* gidx = hidx + hoff[root];
      int32_t* glab = hlab + hoff[root];
#define PPW_RUN(L, S) ppw_component<L, S>(s_key, s_idx, s_lab, s_tile, img, mask, mlab, gkey, gidx, glab, out, H, W, root, \
                                          y0, x0, th, tw, rows_lds)
      if (px > (long long)tile_px || tw > 64 * 1024 || th > 32 * 1024) taint |= PPW_RUN(false, true);
      else if (area > rows_lds * 64) taint |= PPW_RUN(true, true);
      else taint |= PPW_RUN(true, false);                 // the queue can never outgrow its LDS rows
#undef PPW_RUN
    }
  }
  if (taint) atomicOr(&counters[C_TAINT], taint);
}

// exact path: the whole image through ONE heap, exactly like the reference (single thread; correctness anchor)
__global__ void pp_flood_serial_kernel(const float* __restrict__ img, const uint8_t* __restrict__ mask,
                                       const int32_t* __restrict__ markers, unsigned long long* __restrict__ hkey,
                                       uint32_t* __restrict__ hidx, int32_t* __restrict__ out, int H, int W,
                                       int32_t* __restrict__ counters) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  if (!counters[C_SERIAL] && !counters[C_TAINT]) return;
  const size_t n = (size_t)H * W;
  for (size_t i = 0; i < n; ++i) out[i] = markers[i];    // discard whatever the fast path wrote
  PPHeap h;
  h.key = hkey; h.idx = hidx; h.n = 0; h.taint = 0;
  for (size_t i = 0; i < n; ++i)
    if (markers[i] != 0) pp_push(h, pp_key(img[i], 0u), (uint32_t)i);
  pp_flood<false>(h, img, mask, out, H, W, 0u);
  counters[C_SCRATCH] = 1;                               // status bit 0: exact serial path was used
}

// Exact flood of a CONSTANT image (boundary method: watershed(image=mask, ...), every key ties on the value).
// The reference's heap then orders entries by age alone:
//   phase 1  all initial markers carry age 0; they pop first, in an order that only the binary heap's internal layout
//            defines (equal keys are neither swapped on push nor on pop, pushed neighbours with larger keys sink through
//            them).  That order — and with it the order in which the first ring of neighbours is pushed — has to be
//            replayed on the very same heap: one lane, exactly the code of the generic flood;
//   phase 2  every later entry has a unique, increasing age: the heap degenerates to a FIFO queue and the flood is a
//            breadth-first search in queue order.  It is replayed level by level by the whole workgroup: a level's
//            entries claim their unlabelled neighbours with atomicMin((position in level) * 4 + neighbour slot) — the
//            smallest key is the entry the serial flood would have reached first — and the winners are appended to the
//            queue in (position, slot) order through a block-wide prefix sum, which is the serial push order.
// One workgroup of 1024 lanes (a frame's frontier is some 10^4 pixels per level); phase 1 is the serial remainder.
// Heap of the serial marker phase with its top levels in LDS: a sift walks one node per level, and the first
// PPL_TOP = 8191 nodes (13 levels) cost an LDS access instead of a dependent global-memory round trip.
// Same array layout and the same compare / swap sequence as PPHeap (pp_push / pp_pop above).
#define PPL_TOP 8192
struct PPHeapL {
  unsigned long long* gkey; uint32_t* gidx;     // nodes >= PPL_TOP
  unsigned long long* lkey; uint32_t* lidx;     // nodes <  PPL_TOP (LDS)
  int n;
  __device__ __forceinline__ unsigned long long key(int i) const { return i < PPL_TOP ? lkey[i] : gkey[i]; }
  __device__ __forceinline__ uint32_t idx(int i) const { return i < PPL_TOP ? lidx[i] : gidx[i]; }
  __device__ __forceinline__ void set(int i, unsigned long long k, uint32_t ix) {
    if (i < PPL_TOP) { lkey[i] = k; lidx[i] = ix; } else { gkey[i] = k; gidx[i] = ix; }
  }
};

__device__ __forceinline__ void ppl_push(PPHeapL& h, unsigned long long k, uint32_t ix) {
  int child = h.n++;
  h.set(child, k, ix);
  while (child > 0) {
    const int parent = (child + 1) / 2 - 1;
    const unsigned long long kp = h.key(parent);
    if (k < kp) {
      h.set(child, kp, h.idx(parent));
      h.set(parent, k, ix);
      child = parent;
    } else break;
  }
}

// last_k / last_i: the array's last entry (position n - 1), fetched by the caller ahead of time
__device__ __forceinline__ void ppl_pop(PPHeapL& h, unsigned long long& tk, uint32_t& ti, unsigned long long last_k,
                                        uint32_t last_i) {
  tk = h.key(0); ti = h.idx(0);
  const int n = --h.n;
  if (n == 0) return;
  const unsigned long long lk = last_k;
  const uint32_t li = last_i;
  h.set(0, lk, li);
  int i = 0;
  for (;;) {
    const int l = 2 * i + 1, r = 2 * i + 2;
    if (l >= n) break;
    int smallest = i;
    unsigned long long ks = lk;                       // key currently at position i is always `last`
    const unsigned long long kl = h.key(l);
    if (kl < ks) { smallest = l; ks = kl; }
    if (r < n) {
      const unsigned long long kr = h.key(r);
      if (kr < ks) { smallest = r; ks = kr; }
    }
    if (smallest == i) break;
    h.set(i, ks, h.idx(smallest));
    h.set(smallest, lk, li);
    i = smallest;
  }
}

#define PPC_THREADS 1024
// test / ablation hook (mseg_postproc_tuning): 0 = the marker phase of a constant-image flood replays the textbook heap
// (pp_flood_const_serial_kernel) instead of the closed form (pp_flood_const_stream_kernel); same labels either way
static int g_ppc_stream = 1;
__device__ __forceinline__ int ppc_block_scan(int* sh, int tid, int v, int* total) {   // exclusive prefix of v over lanes
  sh[tid] = v;
  __syncthreads();
  for (int o = 1; o < PPC_THREADS; o <<= 1) {
    const int t = tid >= o ? sh[tid - o] : 0;
    __syncthreads();
    sh[tid] += t;
    __syncthreads();
  }
  const int incl = sh[tid];
  *total = sh[PPC_THREADS - 1];
  __syncthreads();
  return incl - v;
}

// kernel 1/3: labels start as the markers; marker pixels listed in raster order at the END of the queue buffer (the queue
// itself never holds a marker, so it stays below n - M).  Three small launches over all pixels: flags, their prefix sum
// (pp_exclusive_scan, total = M), the list.  (One 1024-thread workgroup walking the 4 M pixels of a 2048^2 frame took 6.8 ms.)
__global__ void pp_flood_const_flag_kernel(const int32_t* __restrict__ markers, size_t n, int32_t* __restrict__ flag,
                                           uint32_t* __restrict__ claim, int32_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int m = markers[i];
  out[i] = m;
  claim[i] = 0xffffffffu;
  flag[i] = m != 0;
}
__global__ void pp_flood_const_list_kernel(const int32_t* __restrict__ flag, const int32_t* __restrict__ scan, size_t n,
                                           uint32_t* __restrict__ queue, int32_t* __restrict__ counters, int stream_mode) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) counters[C_CONST_STREAM] = stream_mode ? 1 : 0;
  if (i >= n || !flag[i]) return;
  const size_t M = (size_t)counters[C_CONST_M];
  queue[n - M + (size_t)scan[i]] = (uint32_t)i;
}

// kernel 2/3: the age-0 markers through the reference heap — one wavefront.
// One pop = take the root, move the array's last entry to the root, sift it down.  While that last entry is an age-0
// marker it ties with everything below it and stays at the root, so a run of pops simply walks the array backwards:
// root, a[n-1], a[n-2], ...  Most markers are interior pixels of a seed whose neighbours are all labelled already: their
// pops push nothing and change nothing but the root.  The wavefront therefore looks 64 pops ahead at once — lane u
// fetches the entry and the neighbour state of the u-th predicted pop — retires the leading pops that neither push a
// neighbour nor move a pushed (age > 0) entry to the root in one step, and lets lane 0 execute the first pop that does
// with the reference code.  Same pop / push sequence as the serial flood, ~1/50 of its dependent memory round trips.
// (bodies as device functions: each has a one-frame launch and a launch with one workgroup per FRAME of a batch, below)
__device__ __forceinline__ void pp_flood_const_serial_body(
    const float* __restrict__ img, const uint8_t* __restrict__ mask, unsigned long long* __restrict__ hkey,
    uint32_t* __restrict__ hidx, uint32_t* __restrict__ queue, int32_t* __restrict__ out, int H, int W,
    int32_t* __restrict__ counters) {
  __shared__ unsigned long long sh_hkey[PPL_TOP];
  __shared__ uint32_t sh_hidx[PPL_TOP];
  __shared__ int sh_n, sh_qn, sh_mleft;
  __shared__ unsigned sh_age;
  if (!counters[C_SERIAL] || counters[C_CONST_STREAM]) return;       // (constant image: pp_flood_const_stream_kernel did it)
  const size_t n = (size_t)H * W;
  const int lane = threadIdx.x;
  const int M = counters[C_CONST_M];
  const uint32_t* mlist = queue + (n - (size_t)M);
  PPHeapL h;
  h.gkey = hkey; h.gidx = hidx; h.lkey = sh_hkey; h.lidx = sh_hidx; h.n = 0;
#define PPC_LD(ptr) __hip_atomic_load((ptr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
  // ---- initial heap: equal keys are never swapped on push, so the array is the marker list in raster order -------------
  const unsigned long long key0 = M > 0 ? pp_key(img[mlist[0]], 0u) : 0ull;
  bool all_equal = true;
  for (int i = lane; i < M; i += 64) {
    const uint32_t j = mlist[i];
    const unsigned long long k = pp_key(img[j], 0u);
    all_equal &= (k == key0);
    h.set(i, k, j);
  }
  all_equal = __all(all_equal);
  __syncthreads();
  if (!all_equal && lane == 0) {                         // not a constant image after all: build it the reference way
    h.n = 0;
    for (int i = 0; i < M; ++i) { const uint32_t j = mlist[i]; ppl_push(h, pp_key(img[j], 0u), j); }
  }
  if (lane == 0) { sh_n = M; sh_qn = 0; sh_mleft = M; sh_age = 0u; }
  __syncthreads();
  // ---- pops -------------------------------------------------------------------------------------------------------------
  for (;;) {
    const int n0 = sh_n, m_left = sh_mleft;
    if (m_left <= 0) break;                              // uniform
    int B = m_left < 64 ? m_left : 64;
    if (B > n0) B = n0;
    // lane u: predicted pop u = root (u = 0) or entry n0 - u; the entry that pop moves to the root is entry n0 - 1 - u
    const bool act = lane < B;
    const int ppos = (lane == 0) ? 0 : n0 - lane;
    const int tpos = n0 - 1 - lane;
    bool event = false;
    if (act) {
      const uint32_t e = h.idx(ppos > 0 ? ppos : 0);
      // a pushed entry about to reach the root (or already predicted as a pop): the prediction ends here
      const bool tail_big = (tpos >= 1) && ((unsigned)h.key(tpos) != 0u);
      const bool self_big = (unsigned)h.key(ppos > 0 ? ppos : 0) != 0u;
      const int y = (int)(e / (unsigned)W), x = (int)(e - (unsigned)y * W);
      bool push = false;
      if (y > 0) push |= mask[e - W] && PPC_LD(&out[e - W]) == 0;
      if (x > 0) push |= mask[e - 1] && PPC_LD(&out[e - 1]) == 0;
      if (x + 1 < W) push |= mask[e + 1] && PPC_LD(&out[e + 1]) == 0;
      if (y + 1 < H) push |= mask[e + W] && PPC_LD(&out[e + W]) == 0;
      event = push | tail_big | self_big;
    }
    const unsigned long long evmask = __ballot(act && event);
    const int c = evmask ? (__ffsll((long long)evmask) - 1) : B;      // pops 0 .. c-1 are plain root replacements
    __syncthreads();
    if (lane == 0) {
      int nn = n0, ml = m_left;
      if (c > 0) {                                       // retire them: only the root and the size change
        nn = n0 - c;
        if (nn > 0) h.set(0, h.key(nn), h.idx(nn));
        ml -= c;
      }
      h.n = nn;
      if (c < B) {                                       // the pop that pushes / sinks a pushed entry: reference code
        unsigned age = sh_age;
        int qn = sh_qn;
        unsigned long long k; uint32_t e;
        ppl_pop(h, k, e, h.n >= 2 ? h.key(h.n - 1) : 0ull, h.n >= 2 ? h.idx(h.n - 1) : 0u);
        --ml;
        const int y = (int)(e / (unsigned)W), x = (int)(e - (unsigned)y * W);
        const bool nv[4] = {y > 0, x > 0, x + 1 < W, y + 1 < H};
        const uint32_t nj[4] = {nv[0] ? e - W : e, nv[1] ? e - 1 : e, nv[2] ? e + 1 : e, nv[3] ? e + W : e};
        const int lab = PPC_LD(&out[e]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (nv[q] && mask[nj[q]] && PPC_LD(&out[nj[q]]) == 0) {
            const uint32_t j = nj[q];
            out[j] = lab;
            ppl_push(h, pp_key(img[j], ++age), j);
            queue[qn++] = j;
          }
        }
        sh_age = age; sh_qn = qn;
      }
      sh_n = h.n; sh_mleft = ml;
    }
    __threadfence();
    __syncthreads();
  }
  if (lane == 0) {
    counters[C_CONST_Q] = sh_qn;
    counters[C_SCRATCH] = 1;                             // status bit 0: the exact serial phase was used
  }
#undef PPC_LD
}

// ---- the marker phase WITHOUT the heap (round 3) ------------------------------------------------------------------------------
// On a constant image every key ties on the value and all M markers carry age 0.  What the textbook heap then does has a
// closed form (derived from heap_push / heap_pop of the reference, checked against it pop for pop — tools/flood_stream_model.py):
//   * the heap array starts as the markers in raster order (equal keys are never swapped on push), i.e. an implicit complete
//     binary tree over positions 0 .. M-1; pushed entries (age > 0) are larger than every marker, so all M markers pop
//     before any of them, and a pushed entry never sits above a marker;
//   * a pop moves the array's LAST entry to the root.  If that entry is a pushed one it sinks to the bottom along
//     "left child if it holds a marker, else right child if it holds a marker" and every marker on the path moves up one
//     level: the markers therefore surface in the PREORDER of the tree, and the positions turn into pushed entries in its
//     POSTORDER (a position converts once neither child holds a marker);
//   * if the last entry is a marker still in its original place (the array has shrunk back into the original region, the
//     position has not converted yet) it ties with the root's children, stays at the root and is the NEXT pop: it jumps the
//     queue and leaves the tree (always the tree's last leaf).
// So with count = heap size, zn = size of the original region still intact, cursor = postorder conversions done:
//   pop: output the root's marker; count -= 1; tail = count;
//        tail >= zn                  -> a pushed entry sinks: cursor advances over the next position < zn, next = preorder-next;
//        tail == zn-1, converted     -> the same, zn = tail;           (post_rank[tail] < cursor)
//        tail == zn-1, still marker  -> zn = tail, the marker at `tail` jumps: next = it, and the preorder skips it later;
//        then the popped marker's free neighbours are labelled and pushed (count += k), ages in push order.
// O(1) per pop instead of a ~18-level sift, and — because both candidate streams are known in advance — 64 pops per
// window: lanes fetch the next 64 preorder and 64 tail candidates with their free neighbours, a wave-uniform (scalar) loop
// runs the little automaton above over them, the lanes then claim their pops' neighbours (atomicMin of pop index * 4 +
// slot) and the window is cut before the first pop that lost a neighbour to an earlier pop of the same window (its push
// count was overestimated); everything before it is committed, the rest redone.  Bit-identical pop and push order.

// positions -> preorder / postorder ranks of the implicit complete tree of M nodes
__device__ __forceinline__ unsigned ppc_subtree_size(unsigned v, unsigned M) {
  unsigned long long lo = v, cnt = 1, size = 0;
  while (lo < M) {
    size += (lo + cnt <= M) ? cnt : (M - lo);
    lo = 2 * lo + 1;
    cnt *= 2;
  }
  return (unsigned)size;
}

// per marker position p (heap array index): preorder rank r of p; PE[r] = pixel, PPOS[r] = p, PFL[r] = free-neighbour mask
// of the pixel at the start (bits 0-3: up, left, right, down; bit 31 is set later when the marker jumps the queue);
// TPK[p] = r << 4 | the same mask; MPOS[pixel] = p.  The masks are kept CURRENT: whoever labels a pixel clears its bit in
// the masks of all marker neighbours (a few thousand pushes per frame), so a window reads what a pop will push instead of
// polling the labels of four neighbours per candidate (a second memory round trip per window).
__global__ void pp_flood_const_orders_kernel(const float* __restrict__ img, const uint8_t* __restrict__ mask,
                                             const int32_t* __restrict__ out, const uint32_t* __restrict__ queue, int H, int W,
                                             uint32_t* __restrict__ PE, uint32_t* __restrict__ PPOS,
                                             uint32_t* __restrict__ PFL, uint32_t* __restrict__ TPK,
                                             uint32_t* __restrict__ MPOS, uint32_t* __restrict__ JB,
                                             int32_t* __restrict__ counters) {
  if (!counters[C_SERIAL]) return;
  const size_t n = (size_t)H * W;
  const unsigned M = (unsigned)counters[C_CONST_M];
  // JB: one bit per preorder rank, set when that marker has jumped the queue (the preorder stream skips it)
  for (unsigned wd = blockIdx.x * blockDim.x + threadIdx.x; wd < (M + 31u) / 32u; wd += gridDim.x * blockDim.x) JB[wd] = 0u;
  const uint32_t* mlist = queue + (n - (size_t)M);
  const float v0 = M ? img[mlist[0]] : 0.f;
  bool equal = true;
  for (unsigned p = blockIdx.x * blockDim.x + threadIdx.x; p < M; p += gridDim.x * blockDim.x) {
    const uint32_t e = mlist[p];
    equal &= (img[e] == v0);
    unsigned pre = 0;
    for (unsigned v = p; v > 0; v = (v - 1) >> 1) {
      pre += 1u;
      if ((v & 1u) == 0u) pre += ppc_subtree_size(v - 1, M);    // right child: the whole left sibling subtree comes before
    }
    const int y = (int)(e / (unsigned)W), x = (int)(e - (unsigned)y * W);
    unsigned f0 = 0;
    if (y > 0 && mask[e - W] && out[e - W] == 0) f0 |= 1u;
    if (x > 0 && mask[e - 1] && out[e - 1] == 0) f0 |= 2u;
    if (x + 1 < W && mask[e + 1] && out[e + 1] == 0) f0 |= 4u;
    if (y + 1 < H && mask[e + W] && out[e + W] == 0) f0 |= 8u;
    PE[pre] = e; PPOS[pre] = p; PFL[pre] = f0;
    TPK[p] = (pre << 4) | f0;
    MPOS[e] = p;                                       // marker pixel -> its position
  }
  if (!equal) atomicAnd((unsigned*)&counters[C_CONST_STREAM], 0u);
}

// "position q comes strictly before position c in the postorder of the implicit tree" from the two root paths (index + 1 in
// binary = 1, then left / right bits): a descendant comes before its ancestor; otherwise the one whose ancestor at the common
// depth lies further left.  c < 0: the cursor is past the root — everything is before it.
__device__ __forceinline__ bool ppc_post_before(int q, int c) {
  if (c < 0) return true;
  const unsigned a = (unsigned)q + 1u, b = (unsigned)c + 1u;
  const int da = 31 - __clz(a), db = 31 - __clz(b);
  const int d = da < db ? da : db;
  const unsigned aa = a >> (da - d), bb = b >> (db - d);
  if (aa == bb) return da > db;
  return aa < bb;
}
// postorder successor of c in the tree of the first zn positions (-1 after the root)
__device__ __forceinline__ int ppc_post_next(int c, int zn) {
  if (c <= 0) return -1;
  if ((c & 1) && c + 1 < zn) {
    int v = c + 1;
    while (2 * v + 1 < zn) v = 2 * v + 1;
    return v;
  }
  return (c - 1) >> 1;
}

#define PPS_W 64
// lane `lane` of `old` replaced by the wave-uniform `val`
__device__ __forceinline__ int pps_writelane(int val, int lane, int old) {
  return (int)(threadIdx.x & 63u) == lane ? val : old;
}
__device__ __forceinline__ void pp_flood_const_stream_body(
    const uint8_t* __restrict__ mask, uint32_t* __restrict__ queue, uint32_t* __restrict__ claim,
    const uint32_t* __restrict__ PE, const uint32_t* __restrict__ PPOS, uint32_t* __restrict__ PFL,
    uint32_t* __restrict__ TPK, const uint32_t* __restrict__ MPOS, const int32_t* __restrict__ isMarker,
    uint32_t* __restrict__ JB, int32_t* __restrict__ out, int H, int W, int32_t* __restrict__ counters) {
  if (!counters[C_SERIAL] || !counters[C_CONST_STREAM]) return;
  const size_t n = (size_t)H * W;
  const int lane = threadIdx.x;
  const int M = counters[C_CONST_M];
  const uint32_t* mlist = queue + (n - (size_t)M);
#define PPC_LD(ptr) __hip_atomic_load((ptr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define RL(v, i) __builtin_amdgcn_readlane((int)(v), (i))
  auto nbr = [&](uint32_t e, int s) -> uint32_t { return s == 0 ? e - W : s == 1 ? e - 1 : s == 2 ? e + 1 : e + W; };

  // wave-uniform state: pops done, heap size, size of the original region still intact, postorder cursor (next position to
  // turn into a pushed entry), preorder rank of the next candidate, the marker at the root (pixel, initial free mask, position)
  int pops = 0, count = M, zn = M, ip = 1, qn = 0, cur_pos = 0;
  int c = 0;
  while (2 * c + 1 < M) c = 2 * c + 1;
  uint32_t cur_e = M ? mlist[0] : 0u;
  // (every window commits a pop or skips >= 64 dead candidates; the bound only guards the device against a defect in this
  // code — an endless loop would take the GPU with it)
  long long guard = 3LL * M + 64;
  while (pops < M && --guard >= 0) {
    // ---- candidates of this window: ONE round trip (+ one more for the few that still had a free neighbour at the start) ----
    const int zn0 = zn, ip0 = ip;
    const int T_pos = zn0 - 1 - lane >= 1 ? zn0 - 1 - lane : -1;
    uint32_t T_e = 0, P_e = 0;
    unsigned T_pk = 0, P_fl = 0u;
    int P_pos = -1, P_rank = M;
    if (T_pos >= 0) { T_e = mlist[T_pos]; T_pk = PPC_LD(&TPK[T_pos]); }
    // the jumped-bitmap words of the next 2048 preorder ranks (used below if the window is not a plain run of jumps)
    const int w0 = ip0 >> 5;
    unsigned jbw = 0xffffffffu;
    if ((long long)(w0 + lane) * 32 < M) jbw = PPC_LD(&JB[w0 + lane]);
    const int C_fm = (int)(PPC_LD(&TPK[cur_pos]) & 15u);                        // (uniform) the root marker's free neighbours
    const int T_fm = (int)(T_pk & 15u);
    // ---- fast path: a run of markers that jump the queue -------------------------------------------------------------------------
    // count == zn: the array's last entry is the original marker at zn - 1.  If the pop pushes nothing and that position has
    // not converted, the marker jumps to the root and is the next pop — and so on down the array: pops cur, T0, T1, ... as long
    // as each of them pushes nothing and the next position is still a marker.
    if (count == zn && C_fm == 0) {
      const bool okj = T_pos >= 0 && !ppc_post_before(T_pos, c);              // candidate `lane` can jump
      const bool quiet = T_fm == 0;                                             // ... and its own pop pushes nothing
      // R pops: cur, T0 .. T(R-2); pop i jumps candidate i.  Candidate i must be able to jump for i < R; pops 1 .. R-1 (= T0 ..
      // T(R-2)) must be quiet.  R is limited by the markers left as well.
      const unsigned long long nj = ~__ballot(okj), nq = ~__ballot(quiet);
      int R = nj ? __ffsll((long long)nj) - 1 : 64;                              // candidates 0 .. R-1 can jump
      const int Q = nq ? __ffsll((long long)nq) - 1 : 64;                        // candidates 0 .. Q-1 are quiet
      if (R > Q + 1) R = Q + 1;
      if (R > M - 1 - pops) R = M - 1 - pops;                                    // the last marker's pop is done below
      if (R >= 1) {
        if (lane < R) atomicOr(&JB[T_pk >> 9], 1u << ((T_pk >> 4) & 31u));       // they leave the preorder
        pops += R; count -= R; zn -= R;
        cur_pos = zn0 - R;                                                        // = position of candidate R-1
        cur_e = (uint32_t)RL(T_e, R - 1);
        if (c >= zn) c = (c - 1) >> 1;                                           // (the cursor's own position left the tree)
        continue;
      }
    }
    // ---- the general case ------------------------------------------------------------------------------------------------------
    // preorder candidates: the next 64 ranks >= ip whose marker has NOT jumped, packed (late in a frame most of the deeper
    // markers have jumped: 64 consecutive ranks would hold only a few live ones and starve the window)
    {
      unsigned valid = ~jbw;
      if (lane == 0) valid &= 0xffffffffu << (ip0 & 31);                         // ranks below ip are done
      const long long base = (long long)(w0 + lane) * 32;
      if (base + 32 > M) valid &= (base >= M) ? 0u : (0xffffffffu >> (32 - (int)(M - base)));
      const int cnt = __builtin_popcount(valid);
      int incl = cnt;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
      }
      const int excl = incl - cnt;
      const int ptotal = __shfl(incl, 63);
      if (ptotal == 0 && ip0 < M) {                                              // 2048 ranks without a live marker: move on
        const long long nip = (long long)(w0 + 64) * 32;
        ip = nip < M ? (int)nip : M;
        continue;
      }
      // lane l takes the l-th live rank: the word whose exclusive count is the last one <= l, then the bit inside it
      int wl = 0;
#pragma unroll
      for (int stp = 32; stp >= 1; stp >>= 1) {
        const int cand = wl + stp;
        const int v = __shfl(excl, cand & 63);
        if (cand < 64 && v <= lane) wl = cand;
      }
      unsigned wv = (unsigned)__shfl((int)valid, wl);
      int kth = lane - __shfl(excl, wl);
      if (lane < ptotal) {
        while (kth-- > 0) wv &= wv - 1u;
        P_rank = (w0 + wl) * 32 + (__ffs((int)wv) - 1);
        P_e = PE[P_rank]; P_pos = (int)PPOS[P_rank]; P_fl = PPC_LD(&PFL[P_rank]);
      }
    }
    const int P_fm = (int)(P_fl & 15u);
    const int jt = P_pos >= 0 ? zn0 - 1 - P_pos : -1;                          // this preorder candidate's index in the tail list
    unsigned long long pvalid = __ballot(P_pos >= 0);                          // live preorder candidates, in rank order
    // per-pop records, lane i = pop i: source (0 cur, 1 preorder, 2 tail) and index in its list, the tail candidate that
    // jumped at the pop (or -1); state BEFORE the pop
    int r_src = 0, r_idx = 0, r_jump = -1, s_pops = 0, s_count = 0, s_zn = 0, s_c = 0, s_ip = 0, s_cur = 0;
    int src = 0, idx = 0, L = 0;
    bool stop = false;
    int e_pops = pops, e_count = count, e_zn = zn, e_c = c, e_ip = ip, e_cur = cur_pos;
    const unsigned long long quietmask = __ballot(T_fm == 0);
    int jev = -1;                                      // lane = tail candidate: the recorded pop in front of which it jumped in a run
    for (int i = 0; i < PPS_W && !stop; ++i) {
      // a run of markers jumping the queue (as in the fast path above), retired in one step inside the window: the root
      // marker pushes nothing and the array is exactly the intact prefix.  Only pops that push, sink or convert go through
      // the scalar code below (~200 instructions each for a lone wavefront).
      {
        const int fm0 = src == 0 ? C_fm : src == 1 ? RL(P_fm, idx) : RL(T_fm, idx);
        const int j0 = zn0 - e_zn;                     // index of the array's last slot in the tail list
        if (fm0 == 0 && e_count == e_zn && j0 < PPS_W && e_pops < M - 1) {
          const unsigned long long okm = __ballot(T_pos >= 0 && !ppc_post_before(T_pos, e_c)) >> j0;
          const unsigned long long qm = quietmask >> j0;
          int R = __ffsll((long long)~okm) - 1;        // (the shift brought zeros in at the top: ~okm is never 0 for j0 > 0)
          if (~okm == 0ull) R = 64;
          int Q = __ffsll((long long)~qm) - 1;
          if (~qm == 0ull) Q = 64;
          if (R > Q + 1) R = Q + 1;
          if (R > PPS_W - j0) R = PPS_W - j0;
          if (R > M - 1 - e_pops) R = M - 1 - e_pops;
          if (R >= 1) {
            if (lane >= j0 && lane < j0 + R) jev = i;
            pvalid &= ~__ballot(jt >= j0 && jt < j0 + R);
            e_pops += R; e_count -= R; e_zn -= R;
            e_cur = zn0 - j0 - R;                      // position of the last one: it sits at the root now
            src = 2; idx = j0 + R - 1;
            if (e_c >= e_zn) e_c = (e_c - 1) >> 1;
          }
        }
      }
      r_src = pps_writelane(src, i, r_src); r_idx = pps_writelane(idx, i, r_idx);
      s_pops = pps_writelane(e_pops, i, s_pops); s_count = pps_writelane(e_count, i, s_count);
      s_zn = pps_writelane(e_zn, i, s_zn); s_c = pps_writelane(e_c, i, s_c);
      s_ip = pps_writelane(e_ip, i, s_ip); s_cur = pps_writelane(e_cur, i, s_cur);
      const int fm = src == 0 ? C_fm : src == 1 ? RL(P_fm, idx) : RL(T_fm, idx);
      const int k = __builtin_popcount((unsigned)fm);
      int n_pops = e_pops + 1, n_count = e_count - 1, n_zn = e_zn, n_c = e_c, n_ip = e_ip, n_cur = e_cur;
      int n_src = 0, n_idx = 0, jumped_j = -1;
      if (n_pops < M) {
        const int tail = n_count;
        bool sink = true;
        if (tail < n_zn) {                             // the array is back inside the original region: tail == zn - 1
          const int j = zn0 - 1 - tail;                // its index in the tail list
          if (j >= PPS_W) { stop = true; break; }      // beyond the candidates of this window: the pop starts the next one
          n_zn = tail;
          if (!ppc_post_before(tail, n_c)) {           // still a marker: it jumps to the root
            sink = false;
            jumped_j = j;
            pvalid &= ~__ballot(jt == j);              // ... and leaves the preorder
            n_src = 2; n_idx = j; n_cur = tail;
            if (n_c >= n_zn) n_c = (n_c - 1) >> 1;
          }
        }
        if (sink) {
          n_c = ppc_post_next(n_c, n_zn);              // one position converts
          const unsigned long long rest = pvalid;
          if (!rest) { stop = true; break; }           // the preorder candidates of this window are used up: redo the pop
          const int q = __ffsll((long long)rest) - 1;  // the preorder-next marker that has not jumped
          pvalid &= ~(1ull << q);
          n_src = 1; n_idx = q; n_cur = RL(P_pos, q);
          n_ip = RL(P_rank, q) + 1;
        }
      }
      n_count += k;
      r_jump = pps_writelane(jumped_j, i, r_jump);
      e_pops = n_pops; e_count = n_count; e_zn = n_zn; e_c = n_c; e_ip = n_ip; e_cur = n_cur;
      src = n_src; idx = n_idx;
      L = i + 1;
      if (n_pops >= M) stop = true;
    }
    // ---- lanes: pop `lane` claims its free neighbours; the window is cut at the first pop that lost one ---------------------
    const bool mine = lane < L;
    const int my_idx = r_idx & 63;
    const uint32_t pe = __shfl(P_e, my_idx), te = __shfl(T_e, my_idx);
    const int pfm = __shfl(P_fm, my_idx), tfm = __shfl(T_fm, my_idx);
    const uint32_t e = r_src == 0 ? cur_e : r_src == 1 ? pe : te;
    const int fm = mine ? (r_src == 0 ? C_fm : r_src == 1 ? pfm : tfm) : 0;
    int T = L;
    int total = 0;
    if (__ballot(fm != 0)) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
        if (fm & (1 << s)) atomicMin(&claim[nbr(e, s)], (unsigned)(lane * 4 + s));
      __threadfence();
      bool lost = false;
#pragma unroll
      for (int s = 0; s < 4; ++s)
        if (fm & (1 << s)) lost |= PPC_LD(&claim[nbr(e, s)]) != (unsigned)(lane * 4 + s);
      const unsigned long long lmask = __ballot(lost);
      if (lmask) T = __ffsll((long long)lmask) - 1;                    // >= 1: pop 0 has nobody before it
      // ---- commit pops 0 .. T-1 ---------------------------------------------------------------------------------------------------
      const int k = (lane < T) ? __builtin_popcount((unsigned)fm) : 0;
      int off = k;                                      // inclusive prefix sum over the lanes
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(off, o);
        if (lane >= o) off += t;
      }
      total = __shfl(off, 63);
      if (lane < T && fm) {
        const int lab = out[e];                        // a marker's own label never changes
        int w = qn + off - k;
#pragma unroll
        for (int s = 0; s < 4; ++s)
          if (fm & (1 << s)) {
            const uint32_t q = nbr(e, s);
            queue[w++] = q; out[q] = lab;
            // q is no longer free for any marker next to it (this one included): clear its bit in their masks
            const int qy = (int)(q / (unsigned)W), qx = (int)(q - (unsigned)qy * W);
            const bool nv[4] = {qy > 0, qx > 0, qx + 1 < W, qy + 1 < H};
            const uint32_t nm[4] = {q - W, q - 1, q + 1, q + W};
            const unsigned bit[4] = {8u, 4u, 2u, 1u};          // q is the DOWN / RIGHT / LEFT / UP neighbour of nm[.]
#pragma unroll
            for (int d = 0; d < 4; ++d)
              if (nv[d] && isMarker[nm[d]]) {
                const unsigned old = atomicAnd(&TPK[MPOS[nm[d]]], ~bit[d]);
                atomicAnd(&PFL[old >> 4], ~bit[d]);
              }
          }
      }
      if (lane >= T && fm) {                           // the pops that are redone take their claims back
#pragma unroll
        for (int s = 0; s < 4; ++s)
          if (fm & (1 << s)) {
            const uint32_t q = nbr(e, s);
            if ((PPC_LD(&claim[q]) >> 2) >= (unsigned)T) claim[q] = 0xffffffffu;
          }
      }
    }
    // markers that jumped at a committed pop stay out of the preorder for good
    const int jrank = __shfl((int)(T_pk >> 4), (r_jump >= 0 ? r_jump : 0) & 63);     // preorder rank of that tail candidate
    if (lane < T && r_jump >= 0) atomicOr(&JB[jrank >> 5], 1u << (jrank & 31));
    if (jev >= 0 && jev <= T) atomicOr(&JB[T_pk >> 9], 1u << ((T_pk >> 4) & 31u));    // runs in front of a committed pop
    __threadfence();
    // ---- state after pop T-1 = state before pop T ------------------------------------------------------------------------------
    qn += total;
    if (T < L) {
      pops = RL(s_pops, T); count = RL(s_count, T); zn = RL(s_zn, T); c = RL(s_c, T); ip = RL(s_ip, T);
      cur_pos = RL(s_cur, T);
      cur_e = (uint32_t)RL(e, T);
    } else {
      pops = e_pops; count = e_count; zn = e_zn; c = e_c; ip = e_ip; cur_pos = e_cur;
      if (pops < M && src != 0) {                      // the marker the last pop (or run) put at the root
        cur_e = src == 1 ? (uint32_t)RL(P_e, idx) : (uint32_t)RL(T_e, idx);
      }
    }
  }
  if (lane == 0) {
    counters[C_CONST_Q] = qn;
    counters[C_SCRATCH] = 1;                             // status bit 0: the exact single-wavefront marker phase was used
    if (pops < M) counters[C_TAINT] = 0x40;              // (never seen: the guard fired)
  }
#undef RL
#undef PPC_LD
}

// kernel 3/3: breadth-first search in queue order, level by level
__device__ __forceinline__ void pp_flood_const_bfs_body(
    const uint8_t* __restrict__ mask, uint32_t* __restrict__ queue, uint32_t* __restrict__ claim,
    int32_t* __restrict__ out, int H, int W, const int32_t* __restrict__ counters) {
  __shared__ int sh_scan[PPC_THREADS];
  __shared__ int sh_head, sh_tail;
  if (!counters[C_SERIAL]) return;
  const int tid = threadIdx.x;
  if (tid == 0) { sh_head = 0; sh_tail = counters[C_CONST_Q]; }
  __syncthreads();
  // values written by other waves of the workgroup are read with agent-scope loads (past the per-CU vector cache)
#define PPC_LD(ptr) __hip_atomic_load((ptr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
  for (;;) {
    const int head = sh_head, tail = sh_tail;
    const int cnt = tail - head;
    if (cnt <= 0) break;                                 // uniform exit: every lane reads the same shared values
    const int per = (cnt + PPC_THREADS - 1) / PPC_THREADS;
    const int b = tid * per < cnt ? tid * per : cnt, e_end = (b + per < cnt) ? b + per : cnt;
    // claim: position in level * 4 + neighbour slot (up, left, right, down = the reference's neighbour order)
    for (int q = b; q < e_end; ++q) {
      const uint32_t e = PPC_LD(&queue[head + q]);
      const int y = (int)(e / (unsigned)W), x = (int)(e - (unsigned)y * W);
      const unsigned key = (unsigned)q * 4u;
      if (y > 0 && mask[e - W] && PPC_LD(&out[e - W]) == 0) atomicMin(&claim[e - W], key);
      if (x > 0 && mask[e - 1] && PPC_LD(&out[e - 1]) == 0) atomicMin(&claim[e - 1], key + 1u);
      if (x + 1 < W && mask[e + 1] && PPC_LD(&out[e + 1]) == 0) atomicMin(&claim[e + 1], key + 2u);
      if (y + 1 < H && mask[e + W] && PPC_LD(&out[e + W]) == 0) atomicMin(&claim[e + W], key + 3u);
    }
    __threadfence();
    __syncthreads();
    // winners of this lane's (contiguous) part of the level.  A claim value equal to this entry's key identifies the
    // unique winner of a pixel; out == 0 rules out stale claims of pixels labelled in earlier levels.
    int wins = 0;
    for (int q = b; q < e_end; ++q) {
      const uint32_t e = PPC_LD(&queue[head + q]);
      const int y = (int)(e / (unsigned)W), x = (int)(e - (unsigned)y * W);
      const unsigned key = (unsigned)q * 4u;
      if (y > 0 && mask[e - W] && PPC_LD(&claim[e - W]) == key && PPC_LD(&out[e - W]) == 0) ++wins;
      if (x > 0 && mask[e - 1] && PPC_LD(&claim[e - 1]) == key + 1u && PPC_LD(&out[e - 1]) == 0) ++wins;
      if (x + 1 < W && mask[e + 1] && PPC_LD(&claim[e + 1]) == key + 2u && PPC_LD(&out[e + 1]) == 0) ++wins;
      if (y + 1 < H && mask[e + W] && PPC_LD(&claim[e + W]) == key + 3u && PPC_LD(&out[e + W]) == 0) ++wins;
    }
    int total = 0;
    int pos = tail + ppc_block_scan(sh_scan, tid, wins, &total);
    // append in (position, slot) order and label: only the winner of a pixel passes its test, so it is the only writer
    for (int q = b; q < e_end; ++q) {
      const uint32_t e = PPC_LD(&queue[head + q]);
      const int y = (int)(e / (unsigned)W), x = (int)(e - (unsigned)y * W);
      const unsigned key = (unsigned)q * 4u;
      const int lab = PPC_LD(&out[e]);
#define PPC_TAKE(COND, J, K)                                                     \
      if (COND) {                                                                \
        const uint32_t j = (J);                                                  \
        if (mask[j] && PPC_LD(&claim[j]) == (K) && PPC_LD(&out[j]) == 0) { queue[pos++] = j; out[j] = lab; }  \
      }
      PPC_TAKE(y > 0, e - W, key)
      PPC_TAKE(x > 0, e - 1, key + 1u)
      PPC_TAKE(x + 1 < W, e + 1, key + 2u)
      PPC_TAKE(y + 1 < H, e + W, key + 3u)
#undef PPC_TAKE
    }
    __threadfence_block();
    __syncthreads();
    if (tid == 0) { sh_head = tail; sh_tail = tail + total; }
    __syncthreads();
  }
#undef PPC_LD
}

// ---- launches of the three bodies: one frame, or one workgroup per FRAME of a batch ------------------------------------------
// The marker phase is one wavefront busy for ~45 ms per 2048^2 frame — a latency, not a load.  Frames of a stack are
// independent, so mseg_boundary_flood_batch puts up to PP_BATCH_MAX of them into ONE launch of each body (blockIdx.x = frame,
// every frame with its own workspace): eight floods take the time of one, whatever HIP does with streams and queues.
#define PP_BATCH_MAX 8
struct PPConstFrame {
  const float* img; const uint8_t* mask; unsigned long long* hkey; uint32_t* hidx; uint32_t* clist; uint32_t* hoff;
  uint32_t *bymin, *bymax, *bxmin, *bxmax, *carea; const int32_t* flag; int32_t* out; int32_t* counters;
};
struct PPConstBatch { PPConstFrame f[PP_BATCH_MAX]; };

__global__ __launch_bounds__(64) void pp_flood_const_stream_kernel(
    const uint8_t* __restrict__ mask, uint32_t* __restrict__ queue, uint32_t* __restrict__ claim,
    const uint32_t* __restrict__ PE, const uint32_t* __restrict__ PPOS, uint32_t* __restrict__ PFL,
    uint32_t* __restrict__ TPK, const uint32_t* __restrict__ MPOS, const int32_t* __restrict__ isMarker,
    uint32_t* __restrict__ JB, int32_t* __restrict__ out, int H, int W, int32_t* __restrict__ counters) {
  pp_flood_const_stream_body(mask, queue, claim, PE, PPOS, PFL, TPK, MPOS, isMarker, JB, out, H, W, counters);
}
__global__ __launch_bounds__(64) void pp_flood_const_serial_kernel(
    const float* __restrict__ img, const uint8_t* __restrict__ mask, unsigned long long* __restrict__ hkey,
    uint32_t* __restrict__ hidx, uint32_t* __restrict__ queue, int32_t* __restrict__ out, int H, int W,
    int32_t* __restrict__ counters) {
  pp_flood_const_serial_body(img, mask, hkey, hidx, queue, out, H, W, counters);
}
__global__ __launch_bounds__(PPC_THREADS) void pp_flood_const_bfs_kernel(
    const uint8_t* __restrict__ mask, uint32_t* __restrict__ queue, uint32_t* __restrict__ claim,
    int32_t* __restrict__ out, int H, int W, const int32_t* __restrict__ counters) {
  pp_flood_const_bfs_body(mask, queue, claim, out, H, W, counters);
}
__global__ __launch_bounds__(64) void pp_flood_const_stream_multi_kernel(const PPConstBatch b, int H, int W) {
  const PPConstFrame& p = b.f[blockIdx.x];
  pp_flood_const_stream_body(p.mask, p.clist, p.hoff, p.bymin, p.bymax, p.bxmin, p.bxmax, p.carea, p.flag, p.hidx, p.out, H, W,
                             p.counters);
}
__global__ __launch_bounds__(64) void pp_flood_const_serial_multi_kernel(const PPConstBatch b, int H, int W) {
  const PPConstFrame& p = b.f[blockIdx.x];
  pp_flood_const_serial_body(p.img, p.mask, p.hkey, p.hidx, p.clist, p.out, H, W, p.counters);
}
__global__ __launch_bounds__(PPC_THREADS) void pp_flood_const_bfs_multi_kernel(const PPConstBatch b, int H, int W) {
  const PPConstFrame& p = b.f[blockIdx.x];
  pp_flood_const_bfs_body(p.mask, p.clist, p.hoff, p.out, H, W, p.counters);
}

__global__ void pp_finalize_kernel(const int32_t* __restrict__ out, uint16_t* __restrict__ labels, size_t n,
                                   const int32_t* __restrict__ counters, int32_t* __restrict__ n_inst,
                                   int32_t* __restrict__ status) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) labels[i] = (uint16_t)out[i];               // astype(np.uint16): wraps above 65535 like the reference
  if (i == 0) {
    if (n_inst) *n_inst = counters[C_KEPT];
    // bit0: exact serial flood used, bit1: tie taint, bits 8..12: which tie rule(s) fired (1, 2a, 2b, 3a, 3)
    if (status) *status = (counters[C_SCRATCH] ? 1 : 0) | (counters[C_TAINT] ? 2 : 0) | (counters[C_TAINT] << 8);
  }
}

static void pp_gauss_weights(double w[3]) {
  // scipy.ndimage._gaussian_kernel1d(sigma=0.5, order=0, radius=2): exp(-0.5/sigma^2 * x^2) / sum, float64 (host libm)
  const double sigma2 = 0.5 * 0.5;
  double phi[5], sum = 0.0;
  for (int i = 0; i < 5; ++i) {
    const double x = (double)(i - 2);
    phi[i] = exp(-0.5 / sigma2 * x * x);
    sum += phi[i];
  }
  w[0] = phi[0] / sum; w[1] = phi[1] / sum; w[2] = phi[2] / sum;
}

// shared tail: seeds (binary) + mask + image -> labels
// phase 0: everything; 1 (constant image only): up to the launches of the three constant-image flood bodies — which
// mseg_boundary_flood_batch then makes for several frames at once; 2: what follows them
static int pp_seeds_to_labels(const PPWs& w, const float* img, int H, int W, int distance_rule, int col_major,
                              int force_serial, uint16_t* labels, int32_t* n_inst, int32_t* status, hipStream_t st,
                              int phase = 0) {
  const size_t n = (size_t)H * W;
  const unsigned nb = pp_blocks(n);
  if (phase == 2) {
    hipLaunchKernelGGL(pp_finalize_kernel, dim3(nb), dim3(PP_BLOCK), 0, st, (const int32_t*)w.out, labels, n,
                       (const int32_t*)w.counters, n_inst, status);
    MSEG_LAUNCH_CHECK();
    return MSEG_OK;
  }
  (void)hipMemsetAsync(w.counters, 0, sizeof(int32_t) * C_COUNT, st);
  if (force_serial) hipLaunchKernelGGL(pp_fill_kernel, dim3(1), dim3(PP_BLOCK), 0, st, w.counters + C_SERIAL, 1, (size_t)1);
  (void)hipMemsetAsync(w.area, 0, sizeof(int32_t) * n, st);
  (void)hipMemsetAsync(w.flag, 0, sizeof(int32_t) * n, st);
  hipLaunchKernelGGL(pp_fill_kernel, dim3(nb), dim3(PP_BLOCK), 0, st, w.ckey, 0x7fffffff, n);
  // 8-connected seed components
  hipLaunchKernelGGL(pp_ccl_init_kernel, dim3(nb), dim3(PP_BLOCK), 0, st, (const uint8_t*)w.seedb, w.slab, n);
  hipLaunchKernelGGL((pp_ccl_merge_kernel<true>), dim3(nb), dim3(PP_BLOCK), 0, st, (const uint8_t*)w.seedb, w.slab, H, W);
  hipLaunchKernelGGL(pp_ccl_flatten_kernel, dim3(nb), dim3(PP_BLOCK), 0, st, w.slab, n);
  hipLaunchKernelGGL(pp_seed_stats_kernel, dim3(nb), dim3(PP_BLOCK), 0, st, (const int32_t*)w.slab, H, W, col_major,
                     w.area, w.ckey, w.counters);
  hipLaunchKernelGGL(pp_seed_totals_kernel, dim3(1), dim3(64), 0, st, w.counters);
  hipLaunchKernelGGL(pp_seed_select_kernel, dim3(nb), dim3(PP_BLOCK), 0, st, (const int32_t*)w.slab, n,
                     (const int32_t*)w.area, (const int32_t*)w.ckey, distance_rule, w.flag, w.counters);
  MSEG_LAUNCH_CHECK();
  if (pp_exclusive_scan(w.flag, w.scan, w.bsum, n, nullptr, st)) return MSEG_ELAUNCH;
  hipLaunchKernelGGL(pp_markers_kernel, dim3(nb), dim3(PP_BLOCK), 0, st, (const int32_t*)w.slab, n,
                     (const int32_t*)w.area, (const int32_t*)w.ckey, (const int32_t*)w.scan, (const uint8_t*)w.mask,
                     distance_rule, (const int32_t*)w.counters, w.markers, w.out);
  // 4-connected mask components -> independent floods
  hipLaunchKernelGGL(pp_ccl_init_kernel, dim3(nb), dim3(PP_BLOCK), 0, st, (const uint8_t*)w.mask, w.mlab, n);
  hipLaunchKernelGGL((pp_ccl_merge_kernel<false>), dim3(nb), dim3(PP_BLOCK), 0, st, (const uint8_t*)w.mask, w.mlab, H, W);
  hipLaunchKernelGGL(pp_ccl_flatten_kernel, dim3(nb), dim3(PP_BLOCK), 0, st, w.mlab, n);
  (void)hipMemsetAsync(w.carea, 0, sizeof(int32_t) * n, st);
  (void)hipMemsetAsync(w.hasm, 0, n, st);
  (void)hipMemsetAsync(w.bymax, 0, sizeof(int32_t) * n, st);
  (void)hipMemsetAsync(w.bxmax, 0, sizeof(int32_t) * n, st);
  hipLaunchKernelGGL(pp_fill_kernel, dim3(nb), dim3(PP_BLOCK), 0, st, w.bymin, 0x7fffffff, n);
  hipLaunchKernelGGL(pp_fill_kernel, dim3(nb), dim3(PP_BLOCK), 0, st, w.bxmin, 0x7fffffff, n);
  hipLaunchKernelGGL(pp_mcomp_stats_kernel, dim3(nb), dim3(PP_BLOCK), 0, st, (const int32_t*)w.mlab,
                     (const int32_t*)w.markers, H, W, w.carea, w.bymin, w.bymax, w.bxmin, w.bxmax, w.hasm);
  hipLaunchKernelGGL(pp_mcomp_flag_kernel, dim3(nb), dim3(PP_BLOCK), 0, st, (const int32_t*)w.mlab,
                     (const uint8_t*)w.hasm, n, w.flag, w.carea);
  MSEG_LAUNCH_CHECK();
  if (pp_exclusive_scan(w.carea, w.hoff, w.bsum, n, nullptr, st)) return MSEG_ELAUNCH;             // heap segments
  if (pp_exclusive_scan(w.flag, w.scan, w.bsum, n, w.counters + C_NMCOMP, st)) return MSEG_ELAUNCH; // compact list
  hipLaunchKernelGGL(pp_mcomp_list_kernel, dim3(nb), dim3(PP_BLOCK), 0, st, (const int32_t*)w.flag,
                     (const int32_t*)w.scan, n, w.clist);
  // one wavefront per component: small boxes first (3 workgroups per CU), then the large ones (1 per CU)
  hipLaunchKernelGGL((pp_flood_wave_kernel<PPW_TILE_S>), dim3(256 * 3), dim3(64), 0, st, img, (const uint8_t*)w.mask,
                     (const int32_t*)w.mlab, (const int32_t*)w.clist, (const int32_t*)w.hoff, (const int32_t*)w.carea,
                     (const int32_t*)w.bymin, (const int32_t*)w.bymax, (const int32_t*)w.bxmin,
                     (const int32_t*)w.bxmax, w.hkey, w.hidx, w.scan,
                     w.out, H, W, w.counters, (int)C_WORK_S, 0LL, (long long)g_ppw_tile_s, g_ppw_tile_s, g_ppw_rows);
  hipLaunchKernelGGL((pp_flood_wave_kernel<PPW_TILE_L>), dim3(256), dim3(64), 0, st, img, (const uint8_t*)w.mask,
                     (const int32_t*)w.mlab, (const int32_t*)w.clist, (const int32_t*)w.hoff, (const int32_t*)w.carea,
                     (const int32_t*)w.bymin, (const int32_t*)w.bymax, (const int32_t*)w.bxmin,
                     (const int32_t*)w.bxmax, w.hkey, w.hidx, w.scan,
                     w.out, H, W, w.counters, (int)C_WORK_L, (long long)g_ppw_tile_s, 0x7fffffffffffffffLL, g_ppw_tile_l,
                     g_ppw_rows);
  if (force_serial) {
    // constant image (boundary method): serial heap phase for the age-0 markers, then an ordered parallel BFS
    hipLaunchKernelGGL(pp_flood_const_flag_kernel, dim3(nb), dim3(PP_BLOCK), 0, st, (const int32_t*)w.markers, n, w.flag,
                       (uint32_t*)w.hoff, w.out);
    if (pp_exclusive_scan(w.flag, w.scan, w.bsum, n, w.counters + C_CONST_M, st)) return MSEG_ELAUNCH;
    hipLaunchKernelGGL(pp_flood_const_list_kernel, dim3(nb), dim3(PP_BLOCK), 0, st, (const int32_t*)w.flag,
                       (const int32_t*)w.scan, n, (uint32_t*)w.clist, w.counters, g_ppc_stream);
    // marker phase: the closed form of the heap's behaviour when every key ties (the flag is cleared if they do not) ...
    hipLaunchKernelGGL(pp_flood_const_orders_kernel, dim3(256 * 4), dim3(256), 0, st, img, (const uint8_t*)w.mask,
                       (const int32_t*)w.out, (const uint32_t*)w.clist, H, W, (uint32_t*)w.bymin, (uint32_t*)w.bymax,
                       (uint32_t*)w.bxmin, (uint32_t*)w.bxmax, (uint32_t*)w.carea, (uint32_t*)w.hidx, w.counters);
    if (phase == 1) {
      MSEG_LAUNCH_CHECK();
      return MSEG_OK;
    }
    hipLaunchKernelGGL(pp_flood_const_stream_kernel, dim3(1), dim3(64), 0, st, (const uint8_t*)w.mask, (uint32_t*)w.clist,
                       (uint32_t*)w.hoff, (const uint32_t*)w.bymin, (const uint32_t*)w.bymax, (uint32_t*)w.bxmin,
                       (uint32_t*)w.bxmax, (const uint32_t*)w.carea, (const int32_t*)w.flag, (uint32_t*)w.hidx, w.out, H, W,
                       w.counters);
    // ... or, for keys that differ, the replay of the heap itself
    hipLaunchKernelGGL(pp_flood_const_serial_kernel, dim3(1), dim3(64), 0, st, img, (const uint8_t*)w.mask, w.hkey,
                       w.hidx, (uint32_t*)w.clist, w.out, H, W, w.counters);
    hipLaunchKernelGGL(pp_flood_const_bfs_kernel, dim3(1), dim3(PPC_THREADS), 0, st, (const uint8_t*)w.mask,
                       (uint32_t*)w.clist, (uint32_t*)w.hoff, w.out, H, W, (const int32_t*)w.counters);
  } else {
    hipLaunchKernelGGL(pp_flood_serial_kernel, dim3(1), dim3(64), 0, st, img, (const uint8_t*)w.mask,
                       (const int32_t*)w.markers, w.hkey, w.hidx, w.out, H, W, w.counters);
  }
  hipLaunchKernelGGL(pp_finalize_kernel, dim3(nb), dim3(PP_BLOCK), 0, st, (const int32_t*)w.out, labels, n,
                     (const int32_t*)w.counters, n_inst, status);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

static int pp_distance_smooth(const PPWs& w, const float* cell, int H, int W, hipStream_t st) {
  const size_t n = (size_t)H * W;
  const unsigned nb = pp_blocks(n);
  double gw[3];
  pp_gauss_weights(gw);
  hipLaunchKernelGGL(pp_gauss_kernel, dim3(nb), dim3(PP_BLOCK), 0, st, cell, w.tmp, H, W, 0, gw[0], gw[1], gw[2]);
  hipLaunchKernelGGL(pp_gauss_kernel, dim3(nb), dim3(PP_BLOCK), 0, st, (const float*)w.tmp, w.cs, H, W, 1, gw[0], gw[1], gw[2]);
  hipLaunchKernelGGL(pp_negate_kernel, dim3(nb), dim3(PP_BLOCK), 0, st, (const float*)w.cs, w.tmp, n);  // image = -cell
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

static int pp_distance_tail(const PPWs& w, const float* border, int H, int W, float th_cell, float th_seed,
                            int col_major_ids, uint16_t* labels, int32_t* n_inst, int32_t* status, hipStream_t st) {
  const size_t n = (size_t)H * W;
  hipLaunchKernelGGL(pp_distance_thresh_kernel, dim3(pp_blocks(n)), dim3(PP_BLOCK), 0, st, border, (const float*)w.cs, n,
                     th_cell, th_seed, w.mask, w.seedb);
  MSEG_LAUNCH_CHECK();
  return pp_seeds_to_labels(w, w.tmp, H, W, 1, col_major_ids, 0, labels, n_inst, status, st);
}

// Test / tuning hook: rows of the per-wave queue kept in LDS (1..16) and the two tile capacities in pixels (small <= 8192,
// large <= 30720; 0 = probe global memory for every component).  Process-wide; negative values restore the defaults.
extern "C" int mseg_postproc_set_const_stream(int on) {
  g_ppc_stream = on ? 1 : 0;
  return MSEG_OK;
}

extern "C" int mseg_postproc_tuning(int heap_rows, int tile_small_px, int tile_large_px) {
  if (heap_rows == 0 || heap_rows > PPW_ROWS || tile_small_px > PPW_TILE_S || tile_large_px > PPW_TILE_L ||
      (tile_small_px >= 0 && tile_large_px >= 0 && tile_large_px < tile_small_px))
    return MSEG_EINVAL;
  g_ppw_rows = heap_rows < 0 ? PPW_ROWS : heap_rows;
  g_ppw_tile_s = tile_small_px < 0 ? PPW_TILE_S : tile_small_px;
  g_ppw_tile_l = tile_large_px < 0 ? PPW_TILE_L : tile_large_px;
  return MSEG_OK;
}

extern "C" int mseg_distance_postprocess(const float* border, const float* cell, int H, int W, float th_cell,
                                         float th_seed, int col_major_ids, uint16_t* labels, int32_t* n_instances_dev,
                                         int32_t* status_dev, void* ws, size_t ws_bytes, void* stream) {
  if (!border || !cell || !labels || !ws || H <= 0 || W <= 0) return MSEG_EINVAL;
  const size_t need = mseg_postproc_workspace_bytes(H, W);
  if (need == 0) return MSEG_EINVAL;
  if (ws_bytes < need) return MSEG_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  PPWs w;
  pp_carve(&w, ws, H, W);
  if (pp_distance_smooth(w, cell, H, W, st)) return MSEG_ELAUNCH;
  return pp_distance_tail(w, border, H, W, th_cell, th_seed, col_major_ids, labels, n_instances_dev, status_dev, st);
}

// Threshold sweep of the evaluation (EvalWorker.inference: src/evaluation/eval.py:127-131,397-409 runs
// distance_postprocessing once per (th_cell, th_seed) pair on the same prediction): the smoothed cell map does not depend
// on the thresholds, so it is computed once; thresholds / seeds / watershed run per pair.  labels: [nth][H][W].
extern "C" int mseg_distance_postprocess_sweep(const float* border, const float* cell, int H, int W,
                                               const float* th_cell, const float* th_seed, int nth, int col_major_ids,
                                               uint16_t* labels, int32_t* n_instances_dev, int32_t* status_dev, void* ws,
                                               size_t ws_bytes, void* stream) {
  if (!border || !cell || !labels || !ws || !th_cell || !th_seed || nth <= 0 || H <= 0 || W <= 0) return MSEG_EINVAL;
  const size_t need = mseg_postproc_workspace_bytes(H, W);
  if (need == 0) return MSEG_EINVAL;
  if (ws_bytes < need) return MSEG_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  PPWs w;
  pp_carve(&w, ws, H, W);
  if (pp_distance_smooth(w, cell, H, W, st)) return MSEG_ELAUNCH;
  const size_t n = (size_t)H * W;
  for (int i = 0; i < nth; ++i) {
    const int rc = pp_distance_tail(w, border, H, W, th_cell[i], th_seed[i], col_major_ids, labels + (size_t)i * n,
                                    n_instances_dev ? n_instances_dev + i : nullptr, status_dev ? status_dev + i : nullptr,
                                    st);
    if (rc) return rc;
  }
  return MSEG_OK;
}

extern "C" int mseg_boundary_postprocess(const float* probs_hwc, int H, int W, uint16_t* labels,
                                         int32_t* n_instances_dev, int32_t* status_dev, void* ws, size_t ws_bytes,
                                         void* stream) {
  if (!probs_hwc || !labels || !ws || H <= 0 || W <= 0) return MSEG_EINVAL;
  const size_t need = mseg_postproc_workspace_bytes(H, W);
  if (need == 0) return MSEG_EINVAL;
  if (ws_bytes < need) return MSEG_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  PPWs w;
  pp_carve(&w, ws, H, W);
  const size_t n = (size_t)H * W;
  hipLaunchKernelGGL(pp_boundary_thresh_kernel, dim3(pp_blocks(n)), dim3(PP_BLOCK), 0, st, probs_hwc, n, w.mask,
                     w.seedb, w.tmp);
  MSEG_LAUNCH_CHECK();
  // constant image: every key ties -> the order is the global heap's; go straight to the exact serial flood
  return pp_seeds_to_labels(w, w.tmp, H, W, 0, 0, 1, labels, n_instances_dev, status_dev, st);
}

// The same in three calls, for several frames in flight (InferWorker.infer_stack): _pre (thresholds, components, marker list,
// preorder ranks) per frame on its own workspace, ONE _flood_batch for up to 8 frames (one workgroup per frame), _post per
// frame.  pre + flood_batch(B = 1) + post == mseg_boundary_postprocess, launch for launch.
extern "C" int mseg_boundary_postprocess_pre(const float* probs_hwc, int H, int W, void* ws, size_t ws_bytes, void* stream) {
  if (!probs_hwc || !ws || H <= 0 || W <= 0) return MSEG_EINVAL;
  const size_t need = mseg_postproc_workspace_bytes(H, W);
  if (need == 0) return MSEG_EINVAL;
  if (ws_bytes < need) return MSEG_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  PPWs w;
  pp_carve(&w, ws, H, W);
  const size_t n = (size_t)H * W;
  hipLaunchKernelGGL(pp_boundary_thresh_kernel, dim3(pp_blocks(n)), dim3(PP_BLOCK), 0, st, probs_hwc, n, w.mask,
                     w.seedb, w.tmp);
  MSEG_LAUNCH_CHECK();
  return pp_seeds_to_labels(w, w.tmp, H, W, 0, 0, 1, nullptr, nullptr, nullptr, st, 1);
}

extern "C" int mseg_boundary_flood_batch(void* const* ws_list, int B, int H, int W, void* stream) {
  if (!ws_list || B <= 0 || B > PP_BATCH_MAX || H <= 0 || W <= 0) return MSEG_EINVAL;
  if (mseg_postproc_workspace_bytes(H, W) == 0) return MSEG_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  PPConstBatch b;
  for (int i = 0; i < PP_BATCH_MAX; ++i) {
    void* base = ws_list[i < B ? i : 0];
    if (!base) return MSEG_EINVAL;
    PPWs w;
    pp_carve(&w, base, H, W);
    PPConstFrame& f = b.f[i];
    f.img = w.tmp; f.mask = (const uint8_t*)w.mask; f.hkey = w.hkey; f.hidx = (uint32_t*)w.hidx; f.clist = (uint32_t*)w.clist;
    f.hoff = (uint32_t*)w.hoff; f.bymin = (uint32_t*)w.bymin; f.bymax = (uint32_t*)w.bymax; f.bxmin = (uint32_t*)w.bxmin;
    f.bxmax = (uint32_t*)w.bxmax; f.carea = (uint32_t*)w.carea; f.flag = (const int32_t*)w.flag; f.out = w.out;
    f.counters = w.counters;
  }
  hipLaunchKernelGGL(pp_flood_const_stream_multi_kernel, dim3(B), dim3(64), 0, st, b, H, W);
  hipLaunchKernelGGL(pp_flood_const_serial_multi_kernel, dim3(B), dim3(64), 0, st, b, H, W);
  hipLaunchKernelGGL(pp_flood_const_bfs_multi_kernel, dim3(B), dim3(PPC_THREADS), 0, st, b, H, W);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

extern "C" int mseg_boundary_postprocess_post(int H, int W, uint16_t* labels, int32_t* n_instances_dev, int32_t* status_dev,
                                              void* ws, size_t ws_bytes, void* stream) {
  if (!labels || !ws || H <= 0 || W <= 0) return MSEG_EINVAL;
  const size_t need = mseg_postproc_workspace_bytes(H, W);
  if (need == 0) return MSEG_EINVAL;
  if (ws_bytes < need) return MSEG_EWORKSPACE;
  PPWs w;
  pp_carve(&w, ws, H, W);
  return pp_seeds_to_labels(w, w.tmp, H, W, 0, 0, 1, labels, n_instances_dev, status_dev, (hipStream_t)stream, 2);
}

// =====================================================================================================================
// Evaluation helpers — EvalWorker.calc_scores (src/evaluation/eval.py:248-256) per test image:
//   border_correction (src/utils/utils.py:25-47) -> skimage.measure.label -> get_fast_aji_plus (stats_utils.py:98-179).
// On the device: the relabelling (instances not visible inside the field of interest dropped; 8-connected components of
// EQUAL value; ids in raster order of the first pixel) and the integer statistics AJI+ is made of (areas, pairwise
// intersections).  The Hungarian pairing on the small IoU matrix stays on the host (the reference calls scipy for it).
struct EvWs {
  int32_t* val; int32_t* L; int32_t* flag; int32_t* scan; int32_t* bsum; uint8_t* seen;
};

static size_t ev_carve(EvWs* w, void* base, int H, int W) {
  const size_t n = (size_t)H * W;
  const size_t nb = (n + SCAN_TILE - 1) / SCAN_TILE + 1;
  size_t off = 0;
  char* b = (char*)base;
#define EV_TAKE(field, type, count)                         \
  do {                                                      \
    off = align_up(off, 256);                               \
    if (w) w->field = (type*)(b + off);                     \
    off += sizeof(type) * (size_t)(count);                  \
  } while (0)
  EV_TAKE(val, int32_t, n); EV_TAKE(L, int32_t, n); EV_TAKE(flag, int32_t, n); EV_TAKE(scan, int32_t, n);
  EV_TAKE(bsum, int32_t, 2 * nb); EV_TAKE(seen, uint8_t, 65536);
#undef EV_TAKE
  return align_up(off, 256);
}

extern "C" size_t mseg_eval_workspace_bytes(int H, int W) {
  if (H <= 0 || W <= 0 || (long long)H * W > 0x7fffffffLL) return 0;
  return ev_carve(nullptr, nullptr, H, W);
}

__global__ void ev_seen_kernel(const uint16_t* __restrict__ mask, int H, int W, int bw, uint8_t* __restrict__ seen) {
  const size_t n = (size_t)H * W;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int y = (int)(i / W), x = (int)(i - (size_t)y * W);
  if (y >= bw && y < H - bw && x >= bw && x < W - bw) seen[mask[i]] = 1;   // ids visible in the field of interest
}

__global__ void ev_filter_kernel(const uint16_t* __restrict__ mask, size_t n, const uint8_t* __restrict__ seen,
                                 int32_t* __restrict__ val, int32_t* __restrict__ L) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int v = (mask[i] != 0 && seen[mask[i]]) ? (int)mask[i] : 0;
  val[i] = v;
  L[i] = v ? (int32_t)i : -1;
}

__global__ void ev_merge_kernel(const int32_t* __restrict__ val, int32_t* __restrict__ L, int H, int W) {
  const size_t n = (size_t)H * W;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int v = val[i];
  if (!v) return;
  const int y = (int)(i / W), x = (int)(i - (size_t)y * W);
  if (x > 0 && val[i - 1] == v) uf_union(L, (int)i, (int)i - 1);
  if (y > 0) {
    if (val[i - W] == v) uf_union(L, (int)i, (int)i - W);
    if (x > 0 && val[i - W - 1] == v) uf_union(L, (int)i, (int)i - W - 1);
    if (x + 1 < W && val[i - W + 1] == v) uf_union(L, (int)i, (int)i - W + 1);
  }
}

__global__ void ev_rootflag_kernel(const int32_t* __restrict__ L, size_t n, int32_t* __restrict__ flag) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[i] = (L[i] == (int32_t)i) ? 1 : 0;   // the root is the component's smallest raster index = first pixel
}

__global__ void ev_assign_kernel(const int32_t* __restrict__ L, const int32_t* __restrict__ scan, size_t n,
                                 int32_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = L[i] >= 0 ? scan[L[i]] + 1 : 0;
}

extern "C" int mseg_eval_relabel(const uint16_t* mask, int H, int W, int border_width, int32_t* lab_out,
                                 int32_t* n_out_dev, void* ws, size_t ws_bytes, void* stream) {
  if (!mask || !lab_out || !n_out_dev || !ws || H <= 0 || W <= 0 || border_width < 0) return MSEG_EINVAL;
  const size_t need = mseg_eval_workspace_bytes(H, W);
  if (need == 0) return MSEG_EINVAL;
  if (ws_bytes < need) return MSEG_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  EvWs w;
  ev_carve(&w, ws, H, W);
  const size_t n = (size_t)H * W;
  const unsigned nb = pp_blocks(n);
  (void)hipMemsetAsync(w.seen, 0, 65536, st);
  hipLaunchKernelGGL(ev_seen_kernel, dim3(nb), dim3(PP_BLOCK), 0, st, mask, H, W, border_width, w.seen);
  hipLaunchKernelGGL(ev_filter_kernel, dim3(nb), dim3(PP_BLOCK), 0, st, mask, n, (const uint8_t*)w.seen, w.val, w.L);
  hipLaunchKernelGGL(ev_merge_kernel, dim3(nb), dim3(PP_BLOCK), 0, st, (const int32_t*)w.val, w.L, H, W);
  hipLaunchKernelGGL(pp_ccl_flatten_kernel, dim3(nb), dim3(PP_BLOCK), 0, st, w.L, n);
  hipLaunchKernelGGL(ev_rootflag_kernel, dim3(nb), dim3(PP_BLOCK), 0, st, (const int32_t*)w.L, n, w.flag);
  MSEG_LAUNCH_CHECK();
  if (pp_exclusive_scan(w.flag, w.scan, w.bsum, n, n_out_dev, st)) return MSEG_ELAUNCH;
  hipLaunchKernelGGL(ev_assign_kernel, dim3(nb), dim3(PP_BLOCK), 0, st, (const int32_t*)w.L, (const int32_t*)w.scan, n,
                     lab_out);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

__global__ void ev_pair_kernel(const int32_t* __restrict__ t, const int32_t* __restrict__ p, size_t n, int nt, int np,
                               int32_t* __restrict__ area_t, int32_t* __restrict__ area_p,
                               int32_t* __restrict__ inter) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int a = t[i], b = p[i];
  if (a < 0 || a > nt || b < 0 || b > np) return;     // ids outside the declared range are ignored
  if (a) atomicAdd(&area_t[a], 1);
  if (b) atomicAdd(&area_p[b], 1);
  if (a && b) atomicAdd(&inter[(size_t)a * (np + 1) + b], 1);
}

extern "C" int mseg_eval_pair_counts(const int32_t* true_lab, const int32_t* pred_lab, int H, int W, int nt, int np,
                                     int32_t* area_t, int32_t* area_p, int32_t* inter, void* stream) {
  if (!true_lab || !pred_lab || !area_t || !area_p || !inter || H <= 0 || W <= 0 || nt < 0 || np < 0) return MSEG_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const size_t n = (size_t)H * W;
  (void)hipMemsetAsync(area_t, 0, sizeof(int32_t) * (size_t)(nt + 1), st);
  (void)hipMemsetAsync(area_p, 0, sizeof(int32_t) * (size_t)(np + 1), st);
  (void)hipMemsetAsync(inter, 0, sizeof(int32_t) * (size_t)(nt + 1) * (size_t)(np + 1), st);
  hipLaunchKernelGGL(ev_pair_kernel, dim3(pp_blocks(n)), dim3(PP_BLOCK), 0, st, true_lab, pred_lab, n, nt, np, area_t,
                     area_p, inter);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}
