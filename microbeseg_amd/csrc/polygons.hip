// polygons.hip — instance label image -> outer contour polygons.
//
// Replaces, for a whole frame at once, the reference's per-instance loop on the OMERO upload route:
//   get_indices_pandas(prediction)            src/utils/hull_polygon.py:8-41   (pandas groupby over all labelled pixels)
//   for every instance: cv2_countour(idx)     src/utils/hull_polygon.py:44-89  (paint bbox mask, cv2.findContours(RETR_TREE,
//                                             CHAIN_APPROX_NONE), keep the outer border)
//   points string "x,y x,y ... "              src/inference/infer.py:283-286
//
// cv2.findContours is Suzuki-Abe border following (OpenCV 4.5 contours.cpp, icvFetchContour; restated in
// oracle/contour_ref.py).  The trace of one border is inherently serial, but a frame holds thousands of instances:
//   polygons_find_kernel    one thread per pixel: START CANDIDATES = pixels whose W, NW, N and NE neighbours carry another
//                           label.  The raster-first pixel of every 8-connected component is one; other candidates (tops of
//                           further convex parts, arm tips inside holes) start a trace of an outer border at a later pixel or
//                           of a hole border.
//   polygons_measure_kernel one lane per candidate walks its border once WITHOUT storing it: length + smallest raster index
//                           on the way.  The candidate is the first pixel of an OUTER border iff it is that minimum itself
//                           (a hole border always has a pixel above the hole, an outer border's minimum is the component's
//                           first pixel) — everything else gets length 0.
//   (host: a few thousand (id, first pixel, length) records -> order by id, then first pixel -> offsets)
//   polygons_trace_kernel   one lane per polygon walks the border again and stores (row, col) per visited pixel.
// All integer work on a uint16 image that sits in L2 / Infinity Cache (8 MB for 2048 x 2048): latency-bound, a few
// hundred dependent 8-neighbourhood probes per instance.
#include "common.h"

namespace {

// direction codes of OpenCV: 0..7 = E, NE, N, NW, W, SW, S, SE (image coordinates, y down)
__device__ __constant__ int c_dy[8] = {0, -1, -1, -1, 0, 1, 1, 1};
__device__ __constant__ int c_dx[8] = {1, 1, 0, -1, -1, -1, 0, 1};

__device__ __forceinline__ bool is_fg(const uint16_t* __restrict__ lab, int H, int W, int y, int x, uint16_t L) {
  return (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W && lab[(size_t)y * W + x] == L;
}

// Border following from start pixel (y0, x0) of label L whose west neighbour is not L.  STORE: write (row, col) pairs to out.
// Returns the number of border points; *min_idx = smallest raster index visited.  Every border is a closed walk over
// (pixel, incoming direction) states, so the loop ends after at most 8 * H * W steps; `cap` is a belt-and-braces bound.
template <bool STORE>
__device__ int follow_border(const uint16_t* __restrict__ lab, int H, int W, int y0, int x0, int* __restrict__ out,
                             int* min_idx, long long cap) {
  const uint16_t L = lab[(size_t)y0 * W + x0];
  int s = 4;
  do {                                                       // clockwise from W: NW, N, NE, E, SE, S, SW
    s = (s - 1) & 7;
  } while (s != 4 && !is_fg(lab, H, W, y0 + c_dy[s], x0 + c_dx[s], L));
  int mn = y0 * W + x0;
  if (s == 4) {                                              // single pixel
    if (STORE) { out[0] = y0; out[1] = x0; }
    if (min_idx) *min_idx = mn;
    return 1;
  }
  const int y1 = y0 + c_dy[s], x1 = x0 + c_dx[s];
  int y = y0, x = x0, n = 0;
  for (long long guard = 0; guard < cap; ++guard) {
    int d = s, ny = y, nx = x;
#pragma unroll 1
    for (int k = 1; k <= 8; ++k) {                           // counter-clockwise from s + 1
      d = (s + k) & 7;
      ny = y + c_dy[d]; nx = x + c_dx[d];
      if (is_fg(lab, H, W, ny, nx, L)) break;
    }
    if (STORE) { out[2 * n] = y; out[2 * n + 1] = x; }
    ++n;
    const int idx = y * W + x;
    mn = idx < mn ? idx : mn;
    if (ny == y0 && nx == x0 && y == y1 && x == x1) break;
    y = ny; x = nx;
    s = (d + 4) & 7;
  }
  if (min_idx) *min_idx = mn;
  return n;
}

__global__ __launch_bounds__(256) void polygons_find_kernel(const uint16_t* __restrict__ lab, int H, int W,
                                                            int* __restrict__ cand_px, int capacity,
                                                            int* __restrict__ n_cand) {
  const size_t total = (size_t)H * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const uint16_t L = lab[i];
    if (L == 0) continue;
    const int y = (int)(i / W), x = (int)(i - (size_t)y * W);
    if (is_fg(lab, H, W, y, x - 1, L) || is_fg(lab, H, W, y - 1, x - 1, L) || is_fg(lab, H, W, y - 1, x, L) ||
        is_fg(lab, H, W, y - 1, x + 1, L))
      continue;
    const int slot = atomicAdd(n_cand, 1);
    if (slot < capacity) cand_px[slot] = (int)i;
  }
}

__global__ __launch_bounds__(64) void polygons_measure_kernel(const uint16_t* __restrict__ lab, int H, int W,
                                                              const int* __restrict__ cand_px,
                                                              const int* __restrict__ n_cand, int capacity,
                                                              int* __restrict__ cand_id, int* __restrict__ cand_len) {
  const int n = min(*n_cand, capacity);
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  const int px = cand_px[c];
  const int y0 = px / W, x0 = px - y0 * W;
  int mn;
  const int len = follow_border<false>(lab, H, W, y0, x0, nullptr, &mn, 8LL * H * W + 8);
  cand_id[c] = lab[px];
  cand_len[c] = (mn == px) ? len : 0;                        // not the first pixel of an outer border: no polygon
}

__global__ __launch_bounds__(64) void polygons_trace_kernel(const uint16_t* __restrict__ lab, int H, int W,
                                                            const int* __restrict__ start_px,
                                                            const long long* __restrict__ offsets, int n_poly,
                                                            int* __restrict__ points_yx) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_poly) return;
  const int px = start_px[k];
  const int y0 = px / W, x0 = px - y0 * W;
  follow_border<true>(lab, H, W, y0, x0, points_yx + 2 * offsets[k], nullptr, offsets[k + 1] - offsets[k]);
}

}  // namespace

extern "C" int mseg_polygons_find(const uint16_t* labels, int H, int W, int32_t* cand_px, int32_t* cand_id,
                                  int32_t* cand_len, int capacity, int32_t* n_cand_dev, void* stream) {
  if (!labels || !cand_px || !cand_id || !cand_len || !n_cand_dev || H <= 0 || W <= 0 || capacity <= 0 ||
      (long long)H * W >= (1LL << 31))
    return MSEG_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(n_cand_dev, 0, sizeof(int32_t), st) != hipSuccess) return MSEG_ELAUNCH;
  const size_t total = (size_t)H * W;
  unsigned blocks = (unsigned)((total + 255) / 256);
  if (blocks > 256u * 16u) blocks = 256u * 16u;
  hipLaunchKernelGGL(polygons_find_kernel, dim3(blocks), dim3(256), 0, st, labels, H, W, cand_px, capacity, n_cand_dev);
  MSEG_LAUNCH_CHECK();
  // the number of candidates is only known on the device: one lane per possible slot, idle lanes leave at once
  hipLaunchKernelGGL(polygons_measure_kernel, dim3((unsigned)((capacity + 63) / 64)), dim3(64), 0, st, labels, H, W,
                     cand_px, n_cand_dev, capacity, cand_id, cand_len);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

extern "C" int mseg_polygons_trace(const uint16_t* labels, int H, int W, const int32_t* start_px, const int64_t* offsets,
                                   int n_poly, int32_t* points_yx, void* stream) {
  if (!labels || !start_px || !offsets || !points_yx || H <= 0 || W <= 0 || n_poly < 0) return MSEG_EINVAL;
  if (n_poly == 0) return MSEG_OK;
  hipLaunchKernelGGL(polygons_trace_kernel, dim3((unsigned)((n_poly + 63) / 64)), dim3(64), 0, (hipStream_t)stream, labels,
                     H, W, start_px, (const long long*)offsets, n_poly, points_yx);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}
