// augment.hip — training augmentation of a crop batch on the device.
// Reference: src/training/mytransforms.py (augmentors :12-35): Flip -> Contrast -> Scaling -> Rotate -> Blur -> Noise ->
// ToTensor, per sample on the CPU (numpy / scipy / scikit-image / imgaug) inside DataLoader workers.  Here the host draws
// the random decisions and parameters of every sample exactly along the reference's decision tree
// (training/device_augment.py) and these kernels apply them to the whole batch in HBM.  The reference pipeline is not
// numerically pinned (unseeded RNG, uint16 round trips between stages); parity is per operation (tests compare each
// kernel with the numpy / scipy formula it replaces) and distributional for the pipeline.
// All images are fp32 planes [N][H][W] (the uint16 range 0..65535 is kept until the final normalisation); HBM-bound.
#include "common.h"

#define AUG_BLOCK 256

static inline unsigned aug_blocks(size_t n) {
  size_t b = (n + AUG_BLOCK - 1) / AUG_BLOCK;
  return (unsigned)(b < 1 ? 1 : (b > 65535u * 16u ? 65535u * 16u : b));
}

// ---- uint16 -> fp32 -----------------------------------------------------------------------------------------------------
__global__ void aug_u16_to_f32_kernel(const uint16_t* __restrict__ in, float* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = (float)in[i];
}

extern "C" int mseg_aug_u16_to_f32(const uint16_t* in, float* out, size_t n, void* stream) {
  if (!in || !out || n == 0) return MSEG_EINVAL;
  hipLaunchKernelGGL(aug_u16_to_f32_kernel, dim3(aug_blocks(n)), dim3(AUG_BLOCK), 0, (hipStream_t)stream, in, out, n);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

// ---- Flip (mytransforms.py:129-232): the eight symmetries of the square, exact --------------------------------------------
// code 0 identity, 1 flip left-right, 2 flip up-down, 3 rot90, 4 rot180, 5 rot270, 6 flip-lr then rot90, 7 flip-ud then rot90
// (np.rot90 rotates counter-clockwise: rot90(a)[i][j] = a[j][W-1-i]).  Square planes only for the codes that transpose.
__device__ __forceinline__ void aug_flip_src(int code, int H, int W, int i, int j, int& si, int& sj) {
  switch (code) {
    case 1: si = i; sj = W - 1 - j; break;
    case 2: si = H - 1 - i; sj = j; break;
    case 3: si = j; sj = W - 1 - i; break;                 // rot90
    case 4: si = H - 1 - i; sj = W - 1 - j; break;         // rot180
    case 5: si = H - 1 - j; sj = i; break;                 // rot270
    case 6: si = j; sj = i; break;                         // rot90(fliplr(a))[i][j] = fliplr(a)[j][W-1-i] = a[j][i]
    case 7: si = H - 1 - j; sj = W - 1 - i; break;         // rot90(flipud(a))[i][j] = flipud(a)[j][W-1-i] = a[H-1-j][W-1-i]
    default: si = i; sj = j; break;
  }
}

__global__ void aug_flip_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int H, int W,
                                const int32_t* __restrict__ codes) {
  const size_t hw = (size_t)H * W, n = (size_t)N * hw;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
    const int s = (int)(t / hw);
    const int r = (int)(t - (size_t)s * hw);
    const int i = r / W, j = r - i * W;
    int si, sj;
    aug_flip_src(codes[s], H, W, i, j, si, sj);
    out[t] = in[(size_t)s * hw + (size_t)si * W + sj];
  }
}

extern "C" int mseg_aug_flip(const float* in, float* out, int N, int H, int W, const int32_t* codes_dev, void* stream) {
  if (!in || !out || !codes_dev || N <= 0 || H <= 0 || W <= 0 || in == out) return MSEG_EINVAL;
  hipLaunchKernelGGL(aug_flip_kernel, dim3(aug_blocks((size_t)N * H * W)), dim3(AUG_BLOCK), 0, (hipStream_t)stream, in,
                     out, N, H, W, codes_dev);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

// ---- Scaling / Rotate (mytransforms.py:259-362: imgaug Affine, order 1 for images and float labels, order 0 for uint8 labels,
// constant border 0).  mat[s] = 6 floats: source (x, y) = (m0*x + m1*y + m2, m3*x + m4*y + m5) of destination pixel (x, y);
// the host builds it about the image centre.  apply[s] == 0 copies the sample unchanged.
__global__ void aug_affine_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int H, int W,
                                  const float* __restrict__ mats, const int32_t* __restrict__ apply, int nearest) {
  const size_t hw = (size_t)H * W, n = (size_t)N * hw;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
    const int s = (int)(t / hw);
    if (!apply[s]) { out[t] = in[t]; continue; }
    const int r = (int)(t - (size_t)s * hw);
    const int y = r / W, x = r - y * W;
    const float* m = mats + 6 * s;
    const float sx = m[0] * x + m[1] * y + m[2], sy = m[3] * x + m[4] * y + m[5];
    const float* p = in + (size_t)s * hw;
    float v = 0.f;
    if (nearest) {
      const int ix = (int)floorf(sx + 0.5f), iy = (int)floorf(sy + 0.5f);
      if (ix >= 0 && ix < W && iy >= 0 && iy < H) v = p[(size_t)iy * W + ix];
    } else {
      const float fx = floorf(sx), fy = floorf(sy);
      const int x0 = (int)fx, y0 = (int)fy;
      const float ax = sx - fx, ay = sy - fy;
      auto at = [&](int yy, int xx) -> float {
        return (xx >= 0 && xx < W && yy >= 0 && yy < H) ? p[(size_t)yy * W + xx] : 0.f;      // constant border 0
      };
      v = (1.f - ay) * ((1.f - ax) * at(y0, x0) + ax * at(y0, x0 + 1)) +
          ay * ((1.f - ax) * at(y0 + 1, x0) + ax * at(y0 + 1, x0 + 1));
    }
    out[t] = v;
  }
}

extern "C" int mseg_aug_affine(const float* in, float* out, int N, int H, int W, const float* mats_dev,
                               const int32_t* apply_dev, int nearest, void* stream) {
  if (!in || !out || !mats_dev || !apply_dev || N <= 0 || H <= 0 || W <= 0 || in == out) return MSEG_EINVAL;
  hipLaunchKernelGGL(aug_affine_kernel, dim3(aug_blocks((size_t)N * H * W)), dim3(AUG_BLOCK), 0, (hipStream_t)stream, in,
                     out, N, H, W, mats_dev, apply_dev, nearest);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

// ---- Blur (mytransforms.py:38-62): scipy.ndimage.gaussian_filter(img, sigma), sigma in [1, 2) per sample: separable, radius
// int(4 sigma + 0.5), 'reflect' borders (d c b a | a b c d | d c b a), one axis per launch; sigma <= 0 copies the sample.
__global__ void aug_blur_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int H, int W, int axis,
                                const float* __restrict__ sigmas) {
  const size_t hw = (size_t)H * W, n = (size_t)N * hw;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
    const int s = (int)(t / hw);
    const float sigma = sigmas[s];
    if (!(sigma > 0.f)) { out[t] = in[t]; continue; }
    const int r = (int)(t - (size_t)s * hw);
    const int y = r / W, x = r - y * W;
    const int L = axis == 0 ? H : W, c = axis == 0 ? y : x;
    const size_t stride = axis == 0 ? (size_t)W : 1;
    const float* line = in + (size_t)s * hw + (axis == 0 ? (size_t)x : (size_t)y * W);
    const int radius = (int)(4.0f * sigma + 0.5f);
    const float inv2s2 = -0.5f / (sigma * sigma);
    float acc = 0.f, wsum = 0.f;
    for (int k = -radius; k <= radius; ++k) {
      int q = c + k;
      // scipy 'reflect': period 2L, mirrored about the half-sample positions -0.5 and L-0.5
      const int period = 2 * L;
      q %= period;
      if (q < 0) q += period;
      if (q >= L) q = period - 1 - q;
      const float wgt = __expf(inv2s2 * (float)(k * k));
      acc += wgt * line[(size_t)q * stride];
      wsum += wgt;
    }
    out[t] = acc / wsum;
  }
}

extern "C" int mseg_aug_blur(const float* in, float* tmp, float* out, int N, int H, int W, const float* sigmas_dev,
                             void* stream) {
  if (!in || !tmp || !out || !sigmas_dev || N <= 0 || H <= 0 || W <= 0 || in == tmp || tmp == out) return MSEG_EINVAL;
  const unsigned nb = aug_blocks((size_t)N * H * W);
  hipLaunchKernelGGL(aug_blur_kernel, dim3(nb), dim3(AUG_BLOCK), 0, (hipStream_t)stream, in, tmp, N, H, W, 0, sigmas_dev);
  hipLaunchKernelGGL(aug_blur_kernel, dim3(nb), dim3(AUG_BLOCK), 0, (hipStream_t)stream, (const float*)tmp, out, N, H, W,
                     1, sigmas_dev);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

// ---- per-sample statistics: min, max, sum (fp64) and a 65536-bin histogram of the rounded uint16 value ---------------------
// stats[s] = {min, max, mean}; hist[s][65536] (uint32, only when hist != nullptr: percentiles of the contrast stretch).
__global__ __launch_bounds__(AUG_BLOCK) void aug_stats_kernel(const float* __restrict__ in, int H, int W,
                                                              float* __restrict__ stats, uint32_t* __restrict__ hist) {
  __shared__ float smin[AUG_BLOCK], smax[AUG_BLOCK];
  __shared__ double ssum[AUG_BLOCK];
  const int s = blockIdx.x;
  const size_t hw = (size_t)H * W;
  const float* p = in + (size_t)s * hw;
  float mn = 3.4e38f, mx = -3.4e38f;
  double sum = 0.0;
  for (size_t i = threadIdx.x; i < hw; i += AUG_BLOCK) {
    const float v = p[i];
    mn = fminf(mn, v); mx = fmaxf(mx, v); sum += v;
    if (hist) {
      int b = (int)floorf(v + 0.5f);
      b = b < 0 ? 0 : (b > 65535 ? 65535 : b);
      atomicAdd(&hist[(size_t)s * 65536 + b], 1u);
    }
  }
  smin[threadIdx.x] = mn; smax[threadIdx.x] = mx; ssum[threadIdx.x] = sum;
  __syncthreads();
  for (int o = AUG_BLOCK / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      smin[threadIdx.x] = fminf(smin[threadIdx.x], smin[threadIdx.x + o]);
      smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + o]);
      ssum[threadIdx.x] += ssum[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    stats[3 * s] = smin[0]; stats[3 * s + 1] = smax[0]; stats[3 * s + 2] = (float)(ssum[0] / (double)hw);
  }
}

extern "C" int mseg_aug_stats(const float* in, int N, int H, int W, float* stats_dev, uint32_t* hist_dev, void* stream) {
  if (!in || !stats_dev || N <= 0 || H <= 0 || W <= 0) return MSEG_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (hist_dev) (void)hipMemsetAsync(hist_dev, 0, sizeof(uint32_t) * 65536 * (size_t)N, st);
  hipLaunchKernelGGL(aug_stats_kernel, dim3(N), dim3(AUG_BLOCK), 0, st, in, H, W, stats_dev, hist_dev);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

// ---- contrast parameters on the device: percentiles of the stretch from the histogram (np.percentile, linear
// interpolation between order statistics), mean / min / max of the contrast + gamma branch from the statistics.
// choice[s] = 4 floats drawn by the host: {mode (0 none, 1 stretch, 2 contrast + gamma), q_lo, q_hi (percent) | factor, gamma}
//   mode 1: {1, q_lo, q_hi, -}      mode 2: {2, factor, gamma, -}
__device__ float aug_order_stat(const uint32_t* __restrict__ h, const uint32_t* __restrict__ part /*[256] exclusive*/,
                                uint32_t rank) {
  int c = 0;
  for (int k = 1; k < 256; ++k) if (part[k] <= rank) c = k;          // chunk whose range holds the rank
  uint32_t cum = part[c];
  for (int b = c * 256; b < c * 256 + 256; ++b) {
    cum += h[b];
    if (cum > rank) return (float)b;
  }
  return 65535.f;
}

__global__ __launch_bounds__(256) void aug_contrast_params_kernel(const float* __restrict__ stats,
                                                                  const uint32_t* __restrict__ hist,
                                                                  const float* __restrict__ choice, int HW,
                                                                  float* __restrict__ par) {
  __shared__ uint32_t part[256];
  const int s = blockIdx.x;
  const float* ch = choice + 4 * s;
  float* q = par + 8 * s;
  const int mode = (int)ch[0];
  if (mode == 1) {
    const uint32_t* h = hist + (size_t)s * 65536;
    uint32_t sum = 0;
    for (int b = 0; b < 256; ++b) sum += h[threadIdx.x * 256 + b];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t run = 0;
      for (int k = 0; k < 256; ++k) { const uint32_t t = part[k]; part[k] = run; run += t; }
      float pv[2];
      for (int e = 0; e < 2; ++e) {
        const double pos = (double)ch[1 + e] / 100.0 * (double)(HW - 1);
        const uint32_t lo = (uint32_t)floor(pos);
        const uint32_t hi = lo + 1 < (uint32_t)HW ? lo + 1 : lo;
        const float vlo = aug_order_stat(h, part, lo), vhi = aug_order_stat(h, part, hi);
        pv[e] = vlo + (float)(pos - (double)lo) * (vhi - vlo);
      }
      q[0] = 1.f; q[1] = pv[0]; q[2] = pv[1];
    }
  } else if (threadIdx.x == 0) {
    if (mode == 2) {
      // statistics of v01 = v / 65535; the contrast step is linear, so min / max / mean of u follow analytically
      const float mn = stats[3 * s] / 65535.f, mx = stats[3 * s + 1] / 65535.f, mean = stats[3 * s + 2] / 65535.f;
      const float f = ch[1];
      const float a = (mn - mean) * f + mean, b = (mx - mean) * f + mean;
      const float umin = fminf(a, b), umax = fmaxf(a, b);
      q[0] = 2.f; q[1] = mean; q[2] = f; q[3] = umin; q[4] = umax - umin; q[5] = ch[2];
    } else {
      q[0] = 0.f;
    }
  }
}

extern "C" int mseg_aug_contrast_params(const float* stats_dev, const uint32_t* hist_dev, const float* choice_dev, int N,
                                        int HW, float* par_dev, void* stream) {
  if (!stats_dev || !hist_dev || !choice_dev || !par_dev || N <= 0 || HW <= 0) return MSEG_EINVAL;
  hipLaunchKernelGGL(aug_contrast_params_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, stats_dev, hist_dev,
                     choice_dev, HW, par_dev);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

// ---- Contrast (mytransforms.py:65-126), pointwise part; the host chooses the mode and finishes the parameters from the
// statistics.  par[s] = 8 floats:
//   mode 0  unchanged
//   mode 1  contrast stretching: rescale_intensity(img, in_range=(p0, p1)) -> [0, 65535]:  clip((v - p0) / (p1 - p0), 0, 1) * 65535
//           par = {1, p0, p1}
//   mode 2  contrast + gamma on v/65535: u = (v01 - mean) * f + mean;  w = ((u - mn) / (rng + 1e-7))^gamma * rng + mn;
//           clip(w, 0, 1) * 65535, truncated like astype(uint16);  par = {2, mean, f, mn, rng, gamma}  (mn, rng of u)
__global__ void aug_contrast_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int H, int W,
                                    const float* __restrict__ par) {
  const size_t hw = (size_t)H * W, n = (size_t)N * hw;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
    const int s = (int)(t / hw);
    const float* q = par + 8 * s;
    const int mode = (int)q[0];
    float v = in[t];
    if (mode == 1) {
      const float d = q[2] - q[1];
      float u = d > 0.f ? (v - q[1]) / d : 0.f;
      u = fminf(fmaxf(u, 0.f), 1.f);
      v = floorf(u * 65535.f + 0.5f);                      // rescale_intensity keeps the uint16 dtype
    } else if (mode == 2) {
      float u = (v * (1.f / 65535.f) - q[1]) * q[2] + q[1];
      const float base = (u - q[3]) / (q[4] + 1e-7f);
      float w2 = powf(fmaxf(base, 0.f), q[5]) * q[4] + q[3];
      w2 = fminf(fmaxf(w2, 0.f), 1.f);
      v = floorf(w2 * 65535.f);                            // astype(uint16) truncates
    }
    out[t] = v;
  }
}

extern "C" int mseg_aug_contrast(const float* in, float* out, int N, int H, int W, const float* par_dev, void* stream) {
  if (!in || !out || !par_dev || N <= 0 || H <= 0 || W <= 0) return MSEG_EINVAL;
  hipLaunchKernelGGL(aug_contrast_kernel, dim3(aug_blocks((size_t)N * H * W)), dim3(AUG_BLOCK), 0, (hipStream_t)stream, in,
                     out, N, H, W, par_dev);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

// ---- Contrast, CLAHE branch (mytransforms.py:92-95: skimage.exposure.equalize_adapthist(img, clip_limit=0.01)) -----------
// Contrast-limited adaptive histogram equalisation after Zuiderveld, with scikit-image's defaults: an 8 x 8 grid of tiles
// (kernel_size = shape / 8), 256 bins over 2^14 grey levels, histogram clipped at max(1, 0.01 * tile pixels) with the
// excess redistributed, cumulative mapping per tile, bilinear interpolation between the mappings of the four nearest tile
// centres.  Output back in the uint16 range (the reference multiplies by 65535 and truncates).  Distributional stand-in
// for the library routine (its padding / rounding details are not reproduced).  mode[s] != 3: sample untouched.
#define CLAHE_T 8
#define CLAHE_BINS 256
#define CLAHE_GRAY 16384

__global__ __launch_bounds__(256) void aug_clahe_maps_kernel(const float* __restrict__ in, int H, int W,
                                                             const float* __restrict__ choice,
                                                             float* __restrict__ maps /*[N][64][256]*/) {
  __shared__ unsigned hist[CLAHE_BINS];
  __shared__ float cum[CLAHE_BINS];
  const int s = blockIdx.y, tile = blockIdx.x;
  if ((int)choice[4 * s] != 3) return;
  const int ty = tile / CLAHE_T, tx = tile - ty * CLAHE_T;
  const int y0 = (int)((long long)ty * H / CLAHE_T), y1 = (int)((long long)(ty + 1) * H / CLAHE_T);
  const int x0 = (int)((long long)tx * W / CLAHE_T), x1 = (int)((long long)(tx + 1) * W / CLAHE_T);
  const int tw = x1 - x0, npx = (y1 - y0) * tw;
  hist[threadIdx.x] = 0u;
  __syncthreads();
  const float* p = in + (size_t)s * H * W;
  for (int i = threadIdx.x; i < npx; i += 256) {
    const int y = y0 + i / tw, x = x0 + i % tw;
    int g = (int)floorf(p[(size_t)y * W + x] * ((CLAHE_GRAY - 1) / 65535.f) + 0.5f);
    g = g < 0 ? 0 : (g > CLAHE_GRAY - 1 ? CLAHE_GRAY - 1 : g);
    atomicAdd(&hist[g / (CLAHE_GRAY / CLAHE_BINS)], 1u);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int clim = (int)(0.01f * (float)npx);
    if (clim < 1) clim = 1;
    int excess = 0;
    for (int b = 0; b < CLAHE_BINS; ++b) if ((int)hist[b] > clim) excess += (int)hist[b] - clim;
    const int incr = excess / CLAHE_BINS, upper = clim - incr;
    for (int b = 0; b < CLAHE_BINS; ++b) {
      const int h = (int)hist[b];
      if (h > clim) hist[b] = (unsigned)clim;
      else if (h > upper) { excess -= clim - h; hist[b] = (unsigned)clim; }
      else { excess -= incr; hist[b] = (unsigned)(h + incr); }
    }
    for (int guard = 0; excess > 0 && guard < 64; ++guard) {            // leftover: one count per bin, cyclically
      for (int b = 0; b < CLAHE_BINS && excess > 0; ++b)
        if ((int)hist[b] < clim) { hist[b] += 1u; --excess; }
    }
    float run = 0.f;
    const float scale = (float)(CLAHE_GRAY - 1) / (float)(npx > 0 ? npx : 1);
    for (int b = 0; b < CLAHE_BINS; ++b) {
      run += (float)hist[b];
      cum[b] = fminf(run * scale, (float)(CLAHE_GRAY - 1));
    }
  }
  __syncthreads();
  maps[((size_t)s * CLAHE_T * CLAHE_T + tile) * CLAHE_BINS + threadIdx.x] = cum[threadIdx.x];
}

__global__ void aug_clahe_apply_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int H, int W,
                                       const float* __restrict__ choice, const float* __restrict__ maps) {
  const size_t hw = (size_t)H * W, n = (size_t)N * hw;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
    const int s = (int)(t / hw);
    const float v = in[t];
    if ((int)choice[4 * s] != 3) { out[t] = v; continue; }
    const int r = (int)(t - (size_t)s * hw);
    const int y = r / W, x = r - y * W;
    int g = (int)floorf(v * ((CLAHE_GRAY - 1) / 65535.f) + 0.5f);
    g = g < 0 ? 0 : (g > CLAHE_GRAY - 1 ? CLAHE_GRAY - 1 : g);
    const int bin = g / (CLAHE_GRAY / CLAHE_BINS);
    // position in units of tiles relative to the tile centres
    const float fy = ((float)y + 0.5f) * CLAHE_T / (float)H - 0.5f, fx = ((float)x + 0.5f) * CLAHE_T / (float)W - 0.5f;
    int ty0 = (int)floorf(fy), tx0 = (int)floorf(fx);
    const float ay = fy - (float)ty0, ax = fx - (float)tx0;
    const int ty1 = ty0 + 1 > CLAHE_T - 1 ? CLAHE_T - 1 : ty0 + 1, tx1 = tx0 + 1 > CLAHE_T - 1 ? CLAHE_T - 1 : tx0 + 1;
    ty0 = ty0 < 0 ? 0 : ty0; tx0 = tx0 < 0 ? 0 : tx0;
    const float* m = maps + (size_t)s * CLAHE_T * CLAHE_T * CLAHE_BINS + bin;
    const float m00 = m[(ty0 * CLAHE_T + tx0) * CLAHE_BINS], m01 = m[(ty0 * CLAHE_T + tx1) * CLAHE_BINS];
    const float m10 = m[(ty1 * CLAHE_T + tx0) * CLAHE_BINS], m11 = m[(ty1 * CLAHE_T + tx1) * CLAHE_BINS];
    const float mapped = (1.f - ay) * ((1.f - ax) * m00 + ax * m01) + ay * ((1.f - ax) * m10 + ax * m11);
    out[t] = floorf(fminf(fmaxf(mapped / (float)(CLAHE_GRAY - 1), 0.f), 1.f) * 65535.f);
  }
}

extern "C" size_t mseg_aug_clahe_workspace_bytes(int N) {
  return N > 0 ? (size_t)N * CLAHE_T * CLAHE_T * CLAHE_BINS * sizeof(float) : 0;
}

extern "C" int mseg_aug_clahe(const float* in, float* out, int N, int H, int W, const float* choice_dev, void* ws,
                              void* stream) {
  if (!in || !out || !choice_dev || !ws || N <= 0 || H < CLAHE_T || W < CLAHE_T || in == out) return MSEG_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(aug_clahe_maps_kernel, dim3(CLAHE_T * CLAHE_T, N), dim3(256), 0, st, in, H, W, choice_dev,
                     (float*)ws);
  hipLaunchKernelGGL(aug_clahe_apply_kernel, dim3(aug_blocks((size_t)N * H * W)), dim3(AUG_BLOCK), 0, st, in, out, N, H,
                     W, choice_dev, (const float*)ws);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

// ---- Noise (mytransforms.py:235-256: additive Gaussian noise, sigma = 1..5 % of the image maximum, result clipped to the
// uint16 range) fused with ToTensor's min-max normalisation (utils.py:50-74: clip to [min, max], 2 (v - min) / (max - min) - 1).
// Counter-based generator: two 32-bit hashes of (seed, sample, pixel) -> Box-Muller.  frac[s] = 0.01 .. 0.05 (0: no noise),
// stats[s] = {min, max, mean} of the image at this stage (mseg_aug_stats).
__device__ __forceinline__ uint32_t aug_hash(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

__global__ void aug_noise_normalize_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int H, int W,
                                           const float* __restrict__ frac, const float* __restrict__ stats,
                                           uint32_t seed, float vmin, float vmax) {
  const size_t hw = (size_t)H * W, n = (size_t)N * hw;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
    const int s = (int)(t / hw);
    float v = in[t];
    const float sigma = frac[s] * stats[3 * s + 1];       // fraction of the image maximum
    if (sigma > 0.f) {
      const uint32_t k = (uint32_t)t * 2654435761u + seed;
      const uint32_t a = aug_hash(k ^ 0x9e3779b9u), b = aug_hash(k + 0x85ebca6bu + (uint32_t)(t >> 32));
      const float u1 = ((float)(a >> 8) + 0.5f) * (1.f / 16777216.f), u2 = ((float)(b >> 8) + 0.5f) * (1.f / 16777216.f);
      const float g = sqrtf(-2.f * __logf(u1)) * __cosf(6.2831853f * u2);
      v = floorf(fminf(fmaxf(v + sigma * g, 0.f), 65535.f) + 0.5f);     // imgaug clips and rounds back to uint16
    }
    v = fminf(fmaxf(v, vmin), vmax);
    out[t] = 2.f * (v - vmin) / (vmax - vmin) - 1.f;
  }
}

extern "C" int mseg_aug_noise_normalize(const float* in, float* out, int N, int H, int W, const float* frac_dev,
                                        const float* stats_dev, uint32_t seed, float vmin, float vmax, void* stream) {
  if (!in || !out || !frac_dev || !stats_dev || N <= 0 || H <= 0 || W <= 0 || !(vmax > vmin)) return MSEG_EINVAL;
  hipLaunchKernelGGL(aug_noise_normalize_kernel, dim3(aug_blocks((size_t)N * H * W)), dim3(AUG_BLOCK), 0,
                     (hipStream_t)stream, in, out, N, H, W, frac_dev, stats_dev, seed, vmin, vmax);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

// =====================================================================================================================
// Label creation for the boundary method (first part): boundary_label / border_label of
// src/training/train_data_representations.py (:75-99, :102-125).  The reference loops over the instances
// (binary_dilation(nucleus, 3x3) ^ nucleus, OR-ed over all instances); per pixel that is a 3x3 neighbourhood rule:
//   boundary(p) = some 8-neighbour of p carries an instance id different from label(p)  (nothing outside the image)
//   outer(p)    = label(p) == 0 and some 8-neighbour is foreground
//   mode 0 (boundary_label): 2 where boundary, else 1 where label > 0, else 0
//   mode 1 (border_label):   2 where boundary XOR outer, else 1 where label > 0, else 0
// One pass, 9 reads + 1 byte written per pixel; exact.
__global__ void label_boundary_kernel(const uint16_t* __restrict__ mask, int N, int H, int W, int mode,
                                      uint8_t* __restrict__ out) {
  const size_t hw = (size_t)H * W, n = (size_t)N * hw;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
    const size_t s = t / hw;
    const int r = (int)(t - s * hw);
    const int y = r / W, x = r - y * W;
    const uint16_t* p = mask + s * hw;
    const int lab = p[r];
    bool other = false, anyfg = false;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        if (dy == 0 && dx == 0) continue;
        const int yy = y + dy, xx = x + dx;
        if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
        const int q = p[(size_t)yy * W + xx];
        other |= (q > 0) & (q != lab);
        anyfg |= q > 0;
      }
    const bool outer = (lab == 0) & anyfg;
    const bool two = mode == 0 ? other : (other != outer);
    out[t] = two ? 2 : (lab > 0 ? 1 : 0);
  }
}

extern "C" int mseg_label_boundary(const uint16_t* mask, int N, int H, int W, int mode, uint8_t* out, void* stream) {
  if (!mask || !out || N <= 0 || H <= 0 || W <= 0 || (mode != 0 && mode != 1)) return MSEG_EINVAL;
  hipLaunchKernelGGL(label_boundary_kernel, dim3(aug_blocks((size_t)N * H * W)), dim3(AUG_BLOCK), 0, (hipStream_t)stream,
                     mask, N, H, W, mode, out);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}
