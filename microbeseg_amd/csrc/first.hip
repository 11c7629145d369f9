// first.hip — the network's first convolution (Conv2d(ch_in, 64, 3, padding=1): src/utils/unets.py:303-304,413-414) and
// its weight gradient.  With ch_in = 1 (4 with the zero padding of the input tensor) the layer has 9..36 multiply-adds per
// output against a 256-byte output row: it is bound by the HBM write of z (forward) / the read of dz (weight gradient),
// and running it through the 32-channel K-steps of the matrix-core kernels wastes 8/9 of their work.  Plain VALU kernels:
//   forward: a thread owns 4 output channels of one pixel (16-byte store), weights of its 4 channels in registers;
//   wgrad:   a thread owns 4 output channels and walks pixels, 9 x 4 running sums; fixed-order two-stage reduction.
// The input x is the raw network input (no pending activation / normalisation), NHWC with C = 4 (zero padded).
#include "common.h"

#define FIRST_MAXC 256   // output channels handled (64 in every reference configuration)

template <int CI>   // input channels actually multiplied: 1 (reference nets) or 4 (the padded tensor)
__global__ __launch_bounds__(256) void first_conv_fwd_kernel(const float* __restrict__ x4, const float* __restrict__ w,
                                                             const float* __restrict__ bias, int N, int H, int W,
                                                             int Cin, int Cout, void* __restrict__ z, int z_dtype) {
  // w: torch layout [Cout][Cin][3][3]; thread (pixel slot, channel quad)
  const int CQ = Cout >> 2;                  // channel quads per pixel
  const int ppb = 256 / CQ;                  // pixels per workgroup pass
  const int cq = threadIdx.x % CQ, ps = threadIdx.x / CQ;
  if (ps >= ppb) return;
  const int co = cq * 4;
  float wr[4][CI][9];                        // [co quad lane][ci][tap]; ci beyond Cin stay zero
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int ci = 0; ci < CI; ++ci)
#pragma unroll
      for (int t = 0; t < 9; ++t) wr[j][ci][t] = (ci < Cin) ? w[((size_t)(co + j) * Cin + ci) * 9 + t] : 0.f;
  float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias) b = *reinterpret_cast<const float4*>(bias + co);
  const long long P = (long long)N * H * W;
  for (long long p0 = (long long)blockIdx.x * ppb; p0 < P; p0 += (long long)gridDim.x * ppb) {
    const long long pix = p0 + ps;
    if (pix >= P) continue;
    const int n = (int)(pix / ((long long)H * W));
    const int rem = (int)(pix - (long long)n * H * W);
    const int y = rem / W, x = rem - y * W;
    float4 acc = b;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = y + ky - 1;
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = x + kx - 1;
        if (ix < 0 || ix >= W) continue;
        const float4 v = *reinterpret_cast<const float4*>(x4 + (((size_t)n * H + iy) * W + ix) * 4);
        const int t = ky * 3 + kx;
        acc.x += v.x * wr[0][0][t]; acc.y += v.x * wr[1][0][t]; acc.z += v.x * wr[2][0][t]; acc.w += v.x * wr[3][0][t];
        if (CI > 1) {
          acc.x += v.y * wr[0][1][t]; acc.y += v.y * wr[1][1][t]; acc.z += v.y * wr[2][1][t]; acc.w += v.y * wr[3][1][t];
          acc.x += v.z * wr[0][2][t]; acc.y += v.z * wr[1][2][t]; acc.z += v.z * wr[2][2][t]; acc.w += v.z * wr[3][2][t];
          acc.x += v.w * wr[0][3][t]; acc.y += v.w * wr[1][3][t]; acc.z += v.w * wr[2][3][t]; acc.w += v.w * wr[3][3][t];
        }
      }
    }
    st_f4_rt(z, (size_t)pix * Cout + co, acc, z_dtype);
  }
}

// ---- row-walking forms for Cin = 1 (every reference configuration) ---------------------------------------------------------
// The per-pixel form above spends ~300 index / bounds instructions (a 64-bit division per pixel) on 36 multiply-adds and
// stores 8 bytes per lane: 1.1 TB/s on a layer that only has to write z.  Here a thread owns 8 output channels (16-byte
// bf16 stores; 8 lanes = one pixel's 128 contiguous bytes) of one image COLUMN and walks FR_ROWS rows with the 3 x 3 input
// window in registers (3 new values per row), so a pixel costs 3 loads, 72 multiply-adds and one store, no divisions.
// Same accumulation order as above (bias, then taps row-major; a tap outside the image adds 0).
#define FR_ROWS 8

// ---- K14: frame normalisation on the device (reference: infer.py:346-348, infer_script_local.py:130-132) --------------------
// The reference normalises a frame on the host, `2 * (img.astype(np.float32) - min) / (max - min) - 1` with min / max the
// frame's extrema as scalars of the image dtype, after padding its top / left edge with `min` (utils.py:124-163) — i.e. pad
// pixels become exactly -1.  Here the raw uint8 / uint16 frame is uploaded as it is, `frame_minmax_kernel` reduces the
// extrema on the device (integer atomics: order-independent) and the first convolution applies the same fp32 operations
// in the same order while it loads its 3 x 3 window (explicit round-to-nearest operations: no fused multiply-add), so the
// host never touches a pixel and no normalised or padded copy of the frame exists.
struct RawFrame {
  const void* raw;               // [H0][W0] uint8 / uint16
  const uint32_t* minmax;        // device: {~min, max} as written by frame_minmax_kernel
  int dtype, H0, W0, pad_top, pad_left;
};

__device__ __forceinline__ float raw_frame_norm(unsigned v, float fmin, float frange) {
  // 2 * (f32(x) - min) / (max - min) - 1 as numpy evaluates it on a float32 array with integer scalars
  return __fsub_rn(__fdiv_rn(__fmul_rn(2.f, __fsub_rn((float)v, fmin)), frange), 1.f);
}
__device__ __forceinline__ float raw_frame_px(const RawFrame& f, int yy, int xx, int H, int W, float fmin, float frange) {
  if ((unsigned)yy >= (unsigned)H || (unsigned)xx >= (unsigned)W) return 0.f;     // the convolution's zero padding
  const int y0 = yy - f.pad_top, x0 = xx - f.pad_left;
  if (y0 < 0 || x0 < 0) return -1.f;                                              // frame padding (value `min`)
  const size_t i = (size_t)y0 * f.W0 + x0;
  const unsigned v = f.dtype == MSEG_PIX_U8 ? reinterpret_cast<const uint8_t*>(f.raw)[i]
                                            : reinterpret_cast<const uint16_t*>(f.raw)[i];
  return raw_frame_norm(v, fmin, frange);
}

__global__ __launch_bounds__(256) void frame_minmax_kernel(const void* __restrict__ raw, int dtype, size_t n,
                                                           uint32_t* __restrict__ minmax) {
  unsigned lo = 0xffffffffu, hi = 0u;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const unsigned v = dtype == MSEG_PIX_U8 ? reinterpret_cast<const uint8_t*>(raw)[i]
                                            : reinterpret_cast<const uint16_t*>(raw)[i];
    lo = v < lo ? v : lo;
    hi = v > hi ? v : hi;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned l2 = __shfl_xor(lo, o, 64), h2 = __shfl_xor(hi, o, 64);
    lo = l2 < lo ? l2 : lo;
    hi = h2 > hi ? h2 : hi;
  }
  if ((threadIdx.x & 63) == 0) {                       // both words start at 0: the minimum is kept inverted
    atomicMax(minmax, ~lo);
    atomicMax(minmax + 1, hi);
  }
}

// the normalised, padded frame as an fp32 tensor [H][W] (networks whose first layer does not take the fused kernel below)
__global__ __launch_bounds__(256) void frame_normalize_kernel(const RawFrame f, int H, int W, float* __restrict__ out) {
  const float fmin = (float)(~f.minmax[0]), frange = (float)(f.minmax[1] - ~f.minmax[0]);
  const size_t n = (size_t)H * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int yy = (int)(i / W), xx = (int)(i - (size_t)yy * W);
    out[i] = raw_frame_px(f, yy, xx, H, W, fmin, frange);
  }
}

// RAW: the input is a raw frame (RawFrame, one image, H x W = its padded size) instead of the 4-channel fp32 tensor
template <bool S16, bool RAW>
__global__ __launch_bounds__(256) void first_conv_fwd_rows_kernel(const float* __restrict__ x4, const float* __restrict__ w,
                                                                  const float* __restrict__ bias, int H, int W, int Cout,
                                                                  void* __restrict__ z, const RawFrame rf) {
  const int CG = Cout >> 3;                  // channel groups of 8
  const int XS = 256 / CG;                   // image columns per workgroup
  const int cg = threadIdx.x % CG, xs = threadIdx.x / CG;
  const int co = cg * 8;
  float wr[8][9];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int t = 0; t < 9; ++t) wr[j][t] = w[(size_t)(co + j) * 9 + t];
  float b[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) b[j] = bias ? bias[co + j] : 0.f;
  const int x = blockIdx.x * XS + xs, y0 = blockIdx.y * FR_ROWS, n = blockIdx.z;
  if (x >= W) return;
  const float* const img = RAW ? nullptr : x4 + (size_t)n * H * W * 4;
  float fmin = 0.f, frange = 1.f;
  if (RAW) { fmin = (float)(~rf.minmax[0]); frange = (float)(rf.minmax[1] - ~rf.minmax[0]); }
  auto px = [&](int yy, int xx) -> float {
    if (RAW) return raw_frame_px(rf, yy, xx, H, W, fmin, frange);
    return ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? img[((size_t)yy * W + xx) * 4] : 0.f;
  };
  float win[3][3];                           // rows y - 1, y, y + 1; columns x - 1, x, x + 1
#pragma unroll
  for (int k = 0; k < 3; ++k) { win[0][k] = px(y0 - 1, x - 1 + k); win[1][k] = px(y0, x - 1 + k); }
  const int yend = y0 + FR_ROWS < H ? y0 + FR_ROWS : H;
  for (int y = y0; y < yend; ++y) {
#pragma unroll
    for (int k = 0; k < 3; ++k) win[2][k] = px(y + 1, x - 1 + k);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = b[j];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += win[ky][kx] * wr[j][ky * 3 + kx];
    const size_t e = (((size_t)n * H + y) * W + x) * Cout + co;
    if (S16) {
      const uint2 lo = f32x4_to_bf16(make_float4(acc[0], acc[1], acc[2], acc[3]));
      const uint2 hi = f32x4_to_bf16(make_float4(acc[4], acc[5], acc[6], acc[7]));
      *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(z) + e) = make_uint4(lo.x, lo.y, hi.x, hi.y);
    } else {
      float* const zp = reinterpret_cast<float*>(z) + e;
      *reinterpret_cast<float4*>(zp) = make_float4(acc[0], acc[1], acc[2], acc[3]);
      *reinterpret_cast<float4*>(zp + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { win[0][k] = win[1][k]; win[1][k] = win[2][k]; }
  }
}

// weight gradient, same walk: a thread accumulates 9 taps x 8 channels over the rows of its units (unit = FR_ROWS rows of
// XS columns of one image; units dealt round-robin to the workgroups), then the workgroup combines its 256 / CG columns in
// fixed order.  part[block][9][Cout] as in the per-pixel form.
template <bool S16>
__global__ __launch_bounds__(256) void first_wgrad_rows_kernel(const float* __restrict__ x4, const void* __restrict__ dz,
                                                               int N, int H, int W, int Cout, float* __restrict__ part) {
  __shared__ float fr_red[9 * 2048];         // [9][XS][Cout]; XS * Cout = 256 threads x 8 channels for every shape
  const int CG = Cout >> 3, XS = 256 / CG;
  const int cg = threadIdx.x % CG, xs = threadIdx.x / CG;
  const int co = cg * 8;
  const int ubx = (W + XS - 1) / XS, uby = (H + FR_ROWS - 1) / FR_ROWS;
  const int units = N * uby * ubx;
  float acc[9][8];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
  for (int u = blockIdx.x; u < units; u += gridDim.x) {
    const int n = u / (uby * ubx);
    const int r = u - n * (uby * ubx);
    const int by = r / ubx, bx = r - by * ubx;
    const int x = bx * XS + xs, y0 = by * FR_ROWS;
    if (x >= W) continue;
    const float* const img = x4 + (size_t)n * H * W * 4;
    auto px = [&](int yy, int xx) -> float {
      return ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? img[((size_t)yy * W + xx) * 4] : 0.f;
    };
    float win[3][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { win[0][k] = px(y0 - 1, x - 1 + k); win[1][k] = px(y0, x - 1 + k); }
    const int yend = y0 + FR_ROWS < H ? y0 + FR_ROWS : H;
    // (measured: fetching the next row's dz one trip ahead made the pass slower, 182 -> 228 us; the hardware's own
    // multi-wave overlap is enough)
    for (int y = y0; y < yend; ++y) {
#pragma unroll
      for (int k = 0; k < 3; ++k) win[2][k] = px(y + 1, x - 1 + k);
      const size_t e = (((size_t)n * H + y) * W + x) * Cout + co;
      float g[8];
      if (S16) {
        const uint4 raw = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(dz) + e);
        g[0] = bf16_lo(raw.x); g[1] = bf16_hi(raw.x); g[2] = bf16_lo(raw.y); g[3] = bf16_hi(raw.y);
        g[4] = bf16_lo(raw.z); g[5] = bf16_hi(raw.z); g[6] = bf16_lo(raw.w); g[7] = bf16_hi(raw.w);
      } else {
        const float* const gp = reinterpret_cast<const float*>(dz) + e;
        const float4 a = *reinterpret_cast<const float4*>(gp), c = *reinterpret_cast<const float4*>(gp + 4);
        g[0] = a.x; g[1] = a.y; g[2] = a.z; g[3] = a.w; g[4] = c.x; g[5] = c.y; g[6] = c.z; g[7] = c.w;
      }
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[ky * 3 + kx][j] += g[j] * win[ky][kx];
#pragma unroll
      for (int k = 0; k < 3; ++k) { win[0][k] = win[1][k]; win[1][k] = win[2][k]; }
    }
  }
  // combine the columns of the workgroup (fixed order), one (tap, channel) per thread
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) fr_red[(t * XS + xs) * Cout + co + j] = acc[t][j];
  __syncthreads();
  for (int e = threadIdx.x; e < 9 * Cout; e += 256) {
    const int t = e / Cout, c = e - t * Cout;
    float s = 0.f;
    for (int k = 0; k < XS; ++k) s += fr_red[(t * XS + k) * Cout + c];
    part[((size_t)blockIdx.x * 9 + t) * Cout + c] = s;
  }
}

// shapes the row-walking kernels take: Cin = 1, 8-channel groups that tile a 256-thread workgroup
static bool first_rows_ok(int Cin, int Cout) { return Cin == 1 && (Cout & 7) == 0 && (256 % (Cout >> 3)) == 0 && Cout <= 256; }

// dW[co][0][t] = sum_p dz[p][co] * x[p + tap t][0]   (Cin = 1).  Stage 1: every workgroup reduces a contiguous pixel range
// into part[block][9][Cout]; stage 2 sums the blocks in fixed order (deterministic, fp64).
__global__ __launch_bounds__(256) void first_wgrad_kernel(const float* __restrict__ x4, const void* __restrict__ dz,
                                                          int dz_dtype, int N, int H, int W, int Cout, int pix_per_block,
                                                          float* __restrict__ part) {
  __shared__ float red[9][256 * 4 / 4 * 4];   // [tap][pixel slot * Cout + channel]  (256 threads x 4 channels)
  const int CQ = Cout >> 2;
  const int ppb = 256 / CQ;
  const int cq = threadIdx.x % CQ, ps = threadIdx.x / CQ;
  const int co = cq * 4;
  const long long P = (long long)N * H * W;
  const long long pbeg = (long long)blockIdx.x * pix_per_block;
  long long pend = pbeg + pix_per_block;
  if (pend > P) pend = P;
  float4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ps < ppb) {
    for (long long pix = pbeg + ps; pix < pend; pix += ppb) {
      const int n = (int)(pix / ((long long)H * W));
      const int rem = (int)(pix - (long long)n * H * W);
      const int y = rem / W, x = rem - y * W;
      const float4 g = ld_f4_rt(dz, (size_t)pix * Cout + co, dz_dtype);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int iy = y + ky - 1;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int ix = x + kx - 1;
          const bool ok = (iy >= 0) & (iy < H) & (ix >= 0) & (ix < W);
          const float v = ok ? x4[(((size_t)n * H + iy) * W + ix) * 4] : 0.f;
          const int t = ky * 3 + kx;
          acc[t].x += g.x * v; acc[t].y += g.y * v; acc[t].z += g.z * v; acc[t].w += g.w * v;
        }
      }
    }
  }
  // combine the pixel slots of the workgroup (fixed order), one (tap, channel) per thread
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    if (ps < ppb) *reinterpret_cast<float4*>(&red[t][ps * Cout + co]) = acc[t];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 9 * Cout; e += 256) {
    const int t = e / Cout, c = e - t * Cout;
    float s = 0.f;
    for (int k = 0; k < ppb; ++k) s += red[t][k * Cout + c];
    part[((size_t)blockIdx.x * 9 + t) * Cout + c] = s;
  }
}

// fixed-order sum over the workgroup partials: 8 outputs x 32 partial groups per workgroup, combined through LDS (a thread
// walks nblocks / 32 partials: with 32 outputs x 8 groups this tiny kernel took as long as the streaming pass it finishes)
__global__ __launch_bounds__(256) void first_wgrad_reduce_kernel(const float* __restrict__ part, int nblocks, int Cout,
                                                                 float* __restrict__ dW) {
  __shared__ double red[32][8];
  const int o = threadIdx.x & 7, kg = threadIdx.x >> 3;
  const int e = blockIdx.x * 8 + o;
  const bool live = e < 9 * Cout;
  const int t = live ? e / Cout : 0, c = live ? e - t * Cout : 0;
  double s = 0.0;
  if (live)
    for (int b = kg; b < nblocks; b += 32) s += (double)part[((size_t)b * 9 + t) * Cout + c];
  red[kg][o] = s;
  __syncthreads();
  if (kg == 0 && live) {
    double v = 0.0;
#pragma unroll
    for (int j = 0; j < 32; ++j) v += red[j][o];
    dW[(size_t)c * 9 + t] = (float)v;          // torch layout (Cout, 1, 3, 3)
  }
}

static int first_blocks(long long P, int Cout, int* pix_per_block) {
  const int ppb = 256 / (Cout >> 2);
  long long blocks = 2048;
  long long per = (P + blocks - 1) / blocks;
  per = (per + ppb - 1) / ppb * ppb;
  if (per < ppb) per = ppb;
  blocks = (P + per - 1) / per;
  *pix_per_block = (int)per;
  return (int)blocks;
}

static bool first_shape_ok(int N, int H, int W, int Cin, int Cout) {
  return N > 0 && H > 0 && W > 0 && Cin >= 1 && Cin <= 4 && Cout >= 4 && Cout <= FIRST_MAXC && (Cout & 3) == 0 &&
         (256 % (Cout >> 2)) == 0;
}

extern "C" int mseg_first_conv_fwd(const float* x4, const float* w, const float* bias, int N, int H, int W, int Cin,
                                   int Cout, void* z, int z_dtype, void* stream) {
  if (!x4 || !w || !z || !first_shape_ok(N, H, W, Cin, Cout)) return MSEG_EINVAL;
  if (z_dtype != MSEG_ST_F32 && z_dtype != MSEG_ST_BF16) return MSEG_EINVAL;
  const long long P = (long long)N * H * W;
  if (first_rows_ok(Cin, Cout) && (H + FR_ROWS - 1) / FR_ROWS <= 65535 && N <= 65535) {
    const int XS = 256 / (Cout >> 3);
    const dim3 grid((unsigned)((W + XS - 1) / XS), (unsigned)((H + FR_ROWS - 1) / FR_ROWS), (unsigned)N);
    const RawFrame none = {};
    if (z_dtype == MSEG_ST_BF16)
      hipLaunchKernelGGL((first_conv_fwd_rows_kernel<true, false>), grid, dim3(256), 0, (hipStream_t)stream, x4, w, bias, H, W, Cout, z, none);
    else
      hipLaunchKernelGGL((first_conv_fwd_rows_kernel<false, false>), grid, dim3(256), 0, (hipStream_t)stream, x4, w, bias, H, W, Cout, z, none);
    MSEG_LAUNCH_CHECK();
    return MSEG_OK;
  }
  const int ppb = 256 / (Cout >> 2);
  long long blocks = (P + ppb - 1) / ppb;
  if (blocks > 8192) blocks = 8192;
  if (Cin == 1)
    hipLaunchKernelGGL((first_conv_fwd_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x4, w, bias,
                       N, H, W, Cin, Cout, z, z_dtype);
  else
    hipLaunchKernelGGL((first_conv_fwd_kernel<4>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x4, w, bias,
                       N, H, W, Cin, Cout, z, z_dtype);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

static int raw_frame_check(const void* raw, int dtype, int H0, int W0, int pad_top, int pad_left, const uint32_t* minmax) {
  if (!raw || !minmax || (dtype != MSEG_PIX_U8 && dtype != MSEG_PIX_U16) || H0 <= 0 || W0 <= 0 || pad_top < 0 || pad_left < 0)
    return MSEG_EINVAL;
  if ((long long)H0 + pad_top > 65535 * (long long)FR_ROWS || (long long)W0 + pad_left > 0x7fffffffLL) return MSEG_EINVAL;
  return MSEG_OK;
}

extern "C" int mseg_frame_minmax(const void* raw, int dtype, size_t npix, uint32_t* minmax, void* stream) {
  if (!raw || !minmax || npix == 0 || (dtype != MSEG_PIX_U8 && dtype != MSEG_PIX_U16)) return MSEG_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(minmax, 0, 2 * sizeof(uint32_t), st) != hipSuccess) return MSEG_ELAUNCH;
  size_t blocks = (npix + 256 * 16 - 1) / (256 * 16);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(frame_minmax_kernel, dim3((unsigned)blocks), dim3(256), 0, st, raw, dtype, npix, minmax);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

extern "C" int mseg_frame_normalize(const void* raw, int dtype, int H0, int W0, int pad_top, int pad_left,
                                    const uint32_t* minmax, float* out, void* stream) {
  if (raw_frame_check(raw, dtype, H0, W0, pad_top, pad_left, minmax) || !out) return MSEG_EINVAL;
  const RawFrame f = {raw, minmax, dtype, H0, W0, pad_top, pad_left};
  const int H = H0 + pad_top, W = W0 + pad_left;
  size_t blocks = ((size_t)H * W + 1023) / 1024;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(frame_normalize_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, f, H, W, out);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

extern "C" int mseg_first_conv_fwd_raw(const void* raw, int dtype, int H0, int W0, int pad_top, int pad_left,
                                       const uint32_t* minmax, const float* w, const float* bias, int Cout, void* z,
                                       int z_dtype, void* stream) {
  if (raw_frame_check(raw, dtype, H0, W0, pad_top, pad_left, minmax) || !w || !z) return MSEG_EINVAL;
  if (z_dtype != MSEG_ST_F32 && z_dtype != MSEG_ST_BF16) return MSEG_EINVAL;
  const int H = H0 + pad_top, W = W0 + pad_left;
  if (!first_shape_ok(1, H, W, 1, Cout) || !first_rows_ok(1, Cout)) return MSEG_EINVAL;
  const RawFrame f = {raw, minmax, dtype, H0, W0, pad_top, pad_left};
  const int XS = 256 / (Cout >> 3);
  const dim3 grid((unsigned)((W + XS - 1) / XS), (unsigned)((H + FR_ROWS - 1) / FR_ROWS), 1u);
  if (z_dtype == MSEG_ST_BF16)
    hipLaunchKernelGGL((first_conv_fwd_rows_kernel<true, true>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)nullptr, w,
                       bias, H, W, Cout, z, f);
  else
    hipLaunchKernelGGL((first_conv_fwd_rows_kernel<false, true>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)nullptr, w,
                       bias, H, W, Cout, z, f);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

extern "C" size_t mseg_first_wgrad_workspace_bytes(int N, int H, int W, int Cout) {
  if (!first_shape_ok(N, H, W, 1, Cout)) return 0;
  int per;
  const int blocks = first_blocks((long long)N * H * W, Cout, &per);
  return (size_t)blocks * 9 * Cout * sizeof(float);
}

extern "C" int mseg_first_wgrad(const float* x4, const void* dz, int dz_dtype, int N, int H, int W, int Cout, float* dW,
                                void* ws, void* stream) {
  if (!x4 || !dz || !dW || !ws || !first_shape_ok(N, H, W, 1, Cout)) return MSEG_EINVAL;
  if (dz_dtype != MSEG_ST_F32 && dz_dtype != MSEG_ST_BF16) return MSEG_EINVAL;
  int per;
  const int blocks = first_blocks((long long)N * H * W, Cout, &per);
  hipStream_t st = (hipStream_t)stream;
  const int XS = 256 / (Cout >> 3 ? Cout >> 3 : 1);
  const long long units = first_rows_ok(1, Cout) ? (long long)N * ((H + FR_ROWS - 1) / FR_ROWS) * ((W + XS - 1) / XS) : 0;
  if (units > 0 && units <= 0x7fffffffLL) {
    const unsigned g = (unsigned)(units < blocks ? units : blocks);   // part[] rows beyond g stay unused: reduce over g
    if (dz_dtype == MSEG_ST_BF16)
      hipLaunchKernelGGL((first_wgrad_rows_kernel<true>), dim3(g), dim3(256), 0, st, x4, dz, N, H, W, Cout, (float*)ws);
    else
      hipLaunchKernelGGL((first_wgrad_rows_kernel<false>), dim3(g), dim3(256), 0, st, x4, dz, N, H, W, Cout, (float*)ws);
    MSEG_LAUNCH_CHECK();
    hipLaunchKernelGGL(first_wgrad_reduce_kernel, dim3((9 * Cout + 7) / 8), dim3(256), 0, st, (const float*)ws, (int)g,
                       Cout, dW);
    MSEG_LAUNCH_CHECK();
    return MSEG_OK;
  }
  hipLaunchKernelGGL(first_wgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x4, dz, dz_dtype, N, H, W, Cout,
                     per, (float*)ws);
  MSEG_LAUNCH_CHECK();
  hipLaunchKernelGGL(first_wgrad_reduce_kernel, dim3((9 * Cout + 7) / 8), dim3(256), 0, st, (const float*)ws, blocks,
                     Cout, dW);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}
