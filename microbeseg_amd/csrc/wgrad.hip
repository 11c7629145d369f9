// wgrad.hip — weight gradients of Conv2d 3x3 (s1/s2) and ConvTranspose2d 2x2 s2 on the fp32 matrix cores.
// Reference call site: loss.backward() of nn.Conv2d / nn.ConvTranspose2d (src/training/train.py:488,
// modules built at src/utils/unets.py:112,137,192,244).
//
//   G[t][mch][nch] = sum over pixels p of  P[p][mch] * Q[gather(p, t)][nch]
//
// GEMM view: M = mch (64 per workgroup), N = nch (64 per workgroup), K = pixels (split across gridDim.y).
// One workgroup keeps the accumulators of ALL taps (<= 9 x 16 registers per wave), so the P slab of a K-step
// (32 pixels x 64 channels) is staged once and its MFMA A-fragments stay in registers across the taps; the
// Q slab of each tap is a shifted re-read of the same neighbourhood (L2 hits).  Q is normalised on load, so the
// normalised activation is never stored.  Partial sums go to a split-K workspace that is reduced in fixed order.
#include "common.h"

#define WG_PIX 32
#define WG_LDS 68

template <int TT>
__global__ __launch_bounds__(256) void wgrad_kernel(const MsegWgrad p, int splits, int steps_per_split) {
  __shared__ __attribute__((aligned(16))) float lds[3 * WG_PIX * WG_LDS];
  float* Ps = lds;
  float* Qs = lds + WG_PIX * WG_LDS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;

  const int Mch = p.P.C, Nch = p.Nch;
  const int ntiles_n = (Nch + 63) / 64;
  const int mt = blockIdx.x / ntiles_n, nt = blockIdx.x - mt * ntiles_n;
  const int split = blockIdx.y;

  const long long Ptot = (long long)p.NB * p.Hp * p.Wp;
  const long long pix_begin = (long long)split * steps_per_split * WG_PIX;
  long long pix_end = pix_begin + (long long)steps_per_split * WG_PIX;
  if (pix_end > Ptot) pix_end = Ptot;

  f32x16 acc[TT];
#pragma unroll
  for (int t = 0; t < TT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int spx = tid >> 4;  // staged pixel within a 16-pixel pass
  const int sc4 = tid & 15;  // float4 column
  const int mc = mt * 64 + sc4 * 4;
  const int qc = nt * 64 + sc4 * 4;
  const int Q0 = p.Q[0].C;
  const int qsi = (p.nq > 1 && qc >= Q0) ? 1 : 0;
  const int qcl = qsi ? qc - Q0 : qc;
  const bool mvalid = mc < Mch, qvalid = qc < Nch;

  for (long long pix0 = pix_begin; pix0 < pix_end; pix0 += WG_PIX) {
    int pn[2], py[2], px[2];
    float4 rp[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long long pp = pix0 + spx + 16 * i;
      rp[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (pp < pix_end) {
        const int n = (int)(pp / (p.Hp * p.Wp));
        const int rem = (int)(pp - (long long)n * (p.Hp * p.Wp));
        pn[i] = n; py[i] = rem / p.Wp; px[i] = rem - py[i] * p.Wp;
        if (mvalid) {
          float4 v = *reinterpret_cast<const float4*>(p.P.ptr + (size_t)pp * Mch + mc);
          rp[i] = src_transform4(v, p.P, n, mc);
        }
      } else {
        pn[i] = -1; py[i] = 0; px[i] = 0;
      }
    }

    auto load_q = [&](int t, float4 (&rq)[2]) {
      const int ky = t / p.KW, kx = t - ky * p.KW;
      const MsegSrc& s = p.Q[qsi];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (qvalid && pn[i] >= 0) {
          const int qy = py[i] * p.stride + ky - p.pad, qx = px[i] * p.stride + kx - p.pad;
          if (qy >= 0 && qy < p.Hq && qx >= 0 && qx < p.Wq) {
            const size_t qp = ((size_t)pn[i] * p.Hq + qy) * p.Wq + qx;
            v = *reinterpret_cast<const float4*>(s.ptr + qp * s.C + qcl);
            v = src_transform4(v, s, pn[i], qcl);
          }
        }
        rq[i] = v;
      }
    };

    float4 rq[2];
    load_q(0, rq);

    // previous step's readers of Ps are done (>= 1 barrier since their last read when TT >= 2)
    if (TT < 2) __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<float4*>(Ps + (spx + 16 * i) * WG_LDS + sc4 * 4) = rp[i];

    float a[16];
#pragma unroll
    for (int t = 0; t < TT; ++t) {
      float* Qb = Qs + (t & 1) * WG_PIX * WG_LDS;
#pragma unroll
      for (int i = 0; i < 2; ++i) *reinterpret_cast<float4*>(Qb + (spx + 16 * i) * WG_LDS + sc4 * 4) = rq[i];
      __syncthreads();
      if (t + 1 < TT) load_q(t + 1, rq);
      if (t == 0) {
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) a[kk] = Ps[(2 * kk + lh) * WG_LDS + wm * 32 + li];
      }
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        const float b = Qb[(2 * kk + lh) * WG_LDS + wn * 32 + li];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b, acc[t], 0, 0, 0);
      }
    }
    if (TT & 1) __syncthreads();  // odd tap count: next step's tap 0 reuses the buffer tap TT-1 just read
  }

  // ---- store partial tiles: ws[((split*T + t)*Mch + m)*Nch + n] ---------------------------------------------
  const int n = nt * 64 + wn * 32 + li;
  if (n < Nch) {
#pragma unroll
    for (int t = 0; t < TT; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mt * 64 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < Mch) p.ws[(((size_t)split * TT + t) * Mch + m) * Nch + n] = acc[t][r];
      }
    }
  }
}

// dst[(m*Nst + n)*T + t] = sum_s ws[((s*T + t)*Mch + m)*Nch + n]   (fixed order -> deterministic)
__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dst, int splits, int T,
                                    int Mch, int Nch, int Nst) {
  const size_t total = (size_t)T * Mch * Nch;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(i % Nch);
    const size_t tm = i / Nch;
    const int m = (int)(tm % Mch);
    const int t = (int)(tm / Mch);
    if (n >= Nst) continue;
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += ws[(size_t)k * total + i];
    dst[((size_t)m * Nst + n) * T + t] = s;
  }
}

static int wgrad_plan(const MsegWgrad& p, int& splits, int& steps_per_split) {
  const long long Ptot = (long long)p.NB * p.Hp * p.Wp;
  if (Ptot <= 0) return MSEG_EINVAL;
  const long long steps_total = (Ptot + WG_PIX - 1) / WG_PIX;
  const int tiles = ((p.P.C + 63) / 64) * ((p.Nch + 63) / 64);
  long long s = p.splits > 0 ? p.splits : (1024 + tiles - 1) / tiles;
  if (s > steps_total) s = steps_total;
  if (s > 2048) s = 2048;
  if (s < 1) s = 1;
  steps_per_split = (int)((steps_total + s - 1) / s);
  splits = (int)((steps_total + steps_per_split - 1) / steps_per_split);
  return MSEG_OK;
}

static int wgrad_check(const MsegWgrad& p) {
  if (!p.P.ptr || p.P.C <= 0 || (p.P.C & 3)) return MSEG_EINVAL;
  if (p.nq < 1 || p.nq > 2) return MSEG_EINVAL;
  int csum = 0;
  for (int i = 0; i < p.nq; ++i) {
    if (!p.Q[i].ptr || p.Q[i].C <= 0 || (p.Q[i].C & 3)) return MSEG_EINVAL;
    csum += p.Q[i].C;
  }
  if (csum != p.Nch || p.Nch_store <= 0 || p.Nch_store > p.Nch) return MSEG_EINVAL;
  const int T = p.KH * p.KW;
  if (T != 9 && T != 4) return MSEG_EINVAL;
  if (p.NB <= 0 || p.Hp <= 0 || p.Wp <= 0 || p.Hq <= 0 || p.Wq <= 0 || p.stride < 1) return MSEG_EINVAL;
  return MSEG_OK;
}

extern "C" size_t mseg_wgrad_workspace_bytes(const MsegWgrad* pp) {
  if (!pp || wgrad_check(*pp)) return 0;
  int splits, sps;
  if (wgrad_plan(*pp, splits, sps)) return 0;
  return (size_t)splits * pp->KH * pp->KW * pp->P.C * pp->Nch * sizeof(float);
}

extern "C" int mseg_wgrad(const MsegWgrad* pp, void* stream) {
  if (!pp) return MSEG_EINVAL;
  const MsegWgrad& p = *pp;
  if (wgrad_check(p) || !p.ws || !p.dst) return MSEG_EINVAL;
  int splits, sps;
  if (wgrad_plan(p, splits, sps)) return MSEG_EINVAL;
  const int T = p.KH * p.KW;
  const int tiles = ((p.P.C + 63) / 64) * ((p.Nch + 63) / 64);
  hipStream_t st = (hipStream_t)stream;
  if (p.phase != 2) {
    if (T == 9)
      hipLaunchKernelGGL((wgrad_kernel<9>), dim3(tiles, splits), dim3(256), 0, st, p, splits, sps);
    else
      hipLaunchKernelGGL((wgrad_kernel<4>), dim3(tiles, splits), dim3(256), 0, st, p, splits, sps);
    MSEG_LAUNCH_CHECK();
  }
  if (p.phase == 1) return MSEG_OK;
  const size_t total = (size_t)T * p.P.C * p.Nch;
  unsigned blocks = (unsigned)((total + 255) / 256);
  if (blocks > 8192u) blocks = 8192u;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)p.ws, p.dst, splits, T, p.P.C,
                     p.Nch, p.Nch_store);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}
