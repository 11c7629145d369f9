// wgrad.hip — weight gradients of Conv2d 3x3 (s1/s2) and ConvTranspose2d 2x2 s2 on the fp32 matrix cores.
// Reference call site: loss.backward() of nn.Conv2d / nn.ConvTranspose2d (src/training/train.py:488,
// modules built at src/utils/unets.py:112,137,192,244).
//
//   G[t][mch][nch] = sum over pixels p of  P[p][mch] * Q[gather(p, t)][nch]
//
// GEMM view: M = mch (64 per workgroup), N = nch (64 per workgroup), K = pixels (split across gridDim.y); a
// workgroup owns one kernel row ky (gridDim.z) and keeps the accumulators of its KW taps (KW x 16 registers per
// wave), so the P slab of a K-step (32 pixels x 64 channels) is staged once and its MFMA A-fragments are reused by
// the KW taps; the Q slabs are x-shifted re-reads of the same neighbourhood (L2 hits).  P and Q are normalised on
// load, so the normalised activation is never stored.  Stages are double buffered in LDS: one barrier per
// KW x 16 MFMAs.  Partial sums go to a split-K workspace that is reduced in fixed order (deterministic).
#include "common.h"

#define WG_PIX 32
#define WG_LDS 68

// KWT = taps handled by one workgroup (= one kernel row ky = blockIdx.z): 3 for the 3x3 convs, 2 for the 2x2 convT.
template <int KWT, bool GENERIC_ACT, bool PER_SAMPLE>
__global__ __launch_bounds__(256) void wgrad_kernel(const MsegWgrad p, int splits, int steps_per_split) {
  constexpr int SLAB = WG_PIX * WG_LDS;              // one [32 px][64 ch] slab (row stride 68)
  constexpr int STAGE = (1 + KWT) * SLAB;            // P slab + KWT tap slabs
  __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;

  const int Mch = p.P.C, Nch = p.Nch;
  const int T = p.KH * p.KW;
  const int ntiles_n = (Nch + 63) / 64;
  const int mt = blockIdx.x / ntiles_n, nt = blockIdx.x - mt * ntiles_n;
  const int split = blockIdx.y;
  const int ky = blockIdx.z;

  const long long Ptot = (long long)p.NB * p.Hp * p.Wp;
  const long long pix_begin = (long long)split * steps_per_split * WG_PIX;
  long long pix_end = pix_begin + (long long)steps_per_split * WG_PIX;
  if (pix_end > Ptot) pix_end = Ptot;
  const int nsteps = pix_end > pix_begin ? (int)((pix_end - pix_begin + WG_PIX - 1) / WG_PIX) : 0;

  f32x16 acc[KWT];
#pragma unroll
  for (int t = 0; t < KWT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int spx = tid >> 4;  // staged pixel within a 16-pixel pass
  const int sc4 = tid & 15;  // float4 column
  const int mc = mt * 64 + sc4 * 4;
  const int qc = nt * 64 + sc4 * 4;
  const int Q0 = p.Q[0].C;
  const bool q1 = (p.nq > 1) && (qc >= Q0);
  const bool mvalid = mc < Mch, qvalid = qc < Nch;
  const int mcl = mvalid ? mc : 0;
  const int qcl = qvalid ? (q1 ? qc - Q0 : qc) : 0;
  const float* qptr = q1 ? p.Q[1].ptr : p.Q[0].ptr;
  const float* qscale = q1 ? p.Q[1].scale : p.Q[0].scale;
  const float* qshift = q1 ? p.Q[1].shift : p.Q[0].shift;
  const int qC = q1 ? p.Q[1].C : p.Q[0].C;
  const int qss = q1 ? p.Q[1].ss : p.Q[0].ss;
  const int qact = q1 ? p.Q[1].act : p.Q[0].act;
  const float qlo = (qact == MSEG_ACT_RELU) ? 0.f : -3.402823466e38f;
  const float plo = (p.P.act == MSEG_ACT_RELU) ? 0.f : -3.402823466e38f;

  // a thread's channel quad is fixed for the whole kernel: BatchNorm-style tables (ss == 0) are loaded once
  constexpr int NS = PER_SAMPLE ? 2 : 1;
  const float4 one4 = make_float4(1.f, 1.f, 1.f, 1.f), zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 psc[NS], psh[NS], qsc[NS], qsh[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    psc[i] = (p.P.scale && !PER_SAMPLE) ? *reinterpret_cast<const float4*>(p.P.scale + mcl) : one4;
    psh[i] = (p.P.scale && !PER_SAMPLE) ? *reinterpret_cast<const float4*>(p.P.shift + mcl) : zero4;
    qsc[i] = (qscale && !PER_SAMPLE) ? *reinterpret_cast<const float4*>(qscale + qcl) : one4;
    qsh[i] = (qscale && !PER_SAMPLE) ? *reinterpret_cast<const float4*>(qshift + qcl) : zero4;
  }

  // ---- staging registers (raw loads are issued before the MFMAs of the current step, consumed after them) ----
  float4 rp[2], rq[KWT][2];
  unsigned pmask = 0u, qmask = 0u;

  auto issue = [&](int step) {
    const long long pix0 = pix_begin + (long long)step * WG_PIX;
    pmask = 0u; qmask = 0u;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long long pp = pix0 + spx + 16 * i;
      const bool inr = pp < pix_end;
      const long long ppc = inr ? pp : pix_begin;
      const int n = (int)(ppc / (p.Hp * p.Wp));
      const int rem = (int)(ppc - (long long)n * (p.Hp * p.Wp));
      const int py = rem / p.Wp;
      const int px = rem - py * p.Wp;
      const bool pok = inr && mvalid;
      rp[i] = *reinterpret_cast<const float4*>(p.P.ptr + (pok ? (size_t)pp * Mch + mcl : 0));
      pmask |= pok ? (1u << i) : 0u;
      if (PER_SAMPLE) {
        if (p.P.scale) {
          psc[i] = *reinterpret_cast<const float4*>(p.P.scale + (size_t)n * p.P.ss + mcl);
          psh[i] = *reinterpret_cast<const float4*>(p.P.shift + (size_t)n * p.P.ss + mcl);
        }
        if (qscale) {
          qsc[i] = *reinterpret_cast<const float4*>(qscale + (size_t)n * qss + qcl);
          qsh[i] = *reinterpret_cast<const float4*>(qshift + (size_t)n * qss + qcl);
        }
      }
      const int qy = py * p.stride + ky - p.pad;
      const bool yok = qvalid && inr && qy >= 0 && qy < p.Hq;
#pragma unroll
      for (int kx = 0; kx < KWT; ++kx) {
        const int qx = px * p.stride + kx - p.pad;
        const bool ok = yok && qx >= 0 && qx < p.Wq;
        const size_t off = ok ? (((size_t)n * p.Hq + qy) * p.Wq + qx) * qC + qcl : 0;
        rq[kx][i] = *reinterpret_cast<const float4*>(qptr + off);
        qmask |= ok ? (1u << (kx * 2 + i)) : 0u;
      }
    }
  };

  auto finish = [&](float4 v, float4 sc, float4 sh, bool ok, int act, float lo) -> float4 {
    if (GENERIC_ACT) {
      v = act_fwd4(v, act);
    } else {
      v.x = fmaxf(v.x, lo); v.y = fmaxf(v.y, lo); v.z = fmaxf(v.z, lo); v.w = fmaxf(v.w, lo);
    }
    v.x = ok ? v.x * sc.x + sh.x : 0.f;
    v.y = ok ? v.y * sc.y + sh.y : 0.f;
    v.z = ok ? v.z * sc.z + sh.z : 0.f;
    v.w = ok ? v.w * sc.w + sh.w : 0.f;
    return v;
  };

  auto commit = [&](float* stage) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int si = PER_SAMPLE ? i : 0;
      float* row = stage + (spx + 16 * i) * WG_LDS + sc4 * 4;
      *reinterpret_cast<float4*>(row) = finish(rp[i], psc[si], psh[si], (pmask >> i) & 1u, p.P.act, plo);
#pragma unroll
      for (int kx = 0; kx < KWT; ++kx)
        *reinterpret_cast<float4*>(row + (1 + kx) * SLAB) =
            finish(rq[kx][i], qsc[si], qsh[si], (qmask >> (kx * 2 + i)) & 1u, qact, qlo);
    }
  };

  // double-buffered stages, one barrier per 32-pixel step (KWT x 16 MFMAs per wave)
  if (nsteps > 0) {
    issue(0);
    commit(lds);
  }
  __syncthreads();
  int buf = 0;
  for (int step = 0; step < nsteps; ++step) {
    issue(step + 1 < nsteps ? step + 1 : step);   // last step: harmless re-read keeps the body branch-free
    const float* st = lds + buf * STAGE;
    float a[16];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) a[kk] = st[(2 * kk + lh) * WG_LDS + wm * 32 + li];
#pragma unroll
    for (int kx = 0; kx < KWT; ++kx) {
      const float* Qb = st + (1 + kx) * SLAB;
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        const float b = Qb[(2 * kk + lh) * WG_LDS + wn * 32 + li];
        acc[kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b, acc[kx], 0, 0, 0);
      }
    }
    commit(lds + (buf ^ 1) * STAGE);
    __syncthreads();
    buf ^= 1;
  }

  // ---- store partial tiles: ws[((split*T + t)*Mch + m)*Nch + n] ---------------------------------------------
  const int n = nt * 64 + wn * 32 + li;
  if (n < Nch) {
#pragma unroll
    for (int kx = 0; kx < KWT; ++kx) {
      const int t = ky * KWT + kx;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mt * 64 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < Mch) p.ws[(((size_t)split * T + t) * Mch + m) * Nch + n] = acc[kx][r];
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
  const int per_split_wgs = tiles * p.KH;
  long long s = p.splits > 0 ? p.splits : (1536 + per_split_wgs - 1) / per_split_wgs;
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
  if (!((p.KH == 3 && p.KW == 3) || (p.KH == 2 && p.KW == 2))) return MSEG_EINVAL;
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
    bool generic = (p.P.act != MSEG_ACT_NONE && p.P.act != MSEG_ACT_RELU);
    bool per_sample = (p.P.scale && p.P.ss != 0);
    for (int i = 0; i < p.nq; ++i) {
      if (p.Q[i].act != MSEG_ACT_NONE && p.Q[i].act != MSEG_ACT_RELU) generic = true;
      if (p.Q[i].scale && p.Q[i].ss != 0) per_sample = true;
    }
    const dim3 grid(tiles, splits, p.KH), block(256);
#define MSEG_WGRAD_LAUNCH(KW_, GA_, PS_) \
  hipLaunchKernelGGL((wgrad_kernel<KW_, GA_, PS_>), grid, block, 0, st, p, splits, sps)
    if (p.KW == 3) {
      if (generic) { if (per_sample) MSEG_WGRAD_LAUNCH(3, true, true); else MSEG_WGRAD_LAUNCH(3, true, false); }
      else         { if (per_sample) MSEG_WGRAD_LAUNCH(3, false, true); else MSEG_WGRAD_LAUNCH(3, false, false); }
    } else {
      if (generic) { if (per_sample) MSEG_WGRAD_LAUNCH(2, true, true); else MSEG_WGRAD_LAUNCH(2, true, false); }
      else         { if (per_sample) MSEG_WGRAD_LAUNCH(2, false, true); else MSEG_WGRAD_LAUNCH(2, false, false); }
    }
#undef MSEG_WGRAD_LAUNCH
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
