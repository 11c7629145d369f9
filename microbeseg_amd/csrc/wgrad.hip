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
#include "bf16_affine.h"

#define WG_PIX 32
#define WG_LDS 68
// halo kernel: rows of exactly 256 bytes, so that two slab rows are one ds_read2st64_b32 with immediate offsets (no
// per-row address arithmetic in the K-loop); b32 reads of 32 consecutive floats per half-wave stay conflict-free
#define WH_LDS 64

// identity affine for operands without tables (filled once per device by the launcher)
__device__ float g_wg_ident_scale[8];
__device__ float g_wg_ident_shift[8];
__global__ void wgrad_init_ident_kernel() {
  if (threadIdx.x < 8) { g_wg_ident_scale[threadIdx.x] = 1.f; g_wg_ident_shift[threadIdx.x] = 0.f; }
}

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
  // 1-D grid; logical order: tile fastest, then kernel row, then split -> all workgroups that read the same pixel
  // range (one split) are neighbours in logical order and run on one XCD (shared L2 for the P / Q slabs)
  const int ntiles = ((Mch + 63) / 64) * ntiles_n;
  const int lid = (int)xcd_logical_id(blockIdx.x, gridDim.x);
  const int tile = lid % ntiles;
  const int ky = (lid / ntiles) % p.KH;
  const int split = lid / (ntiles * p.KH);
  const int mt = tile / ntiles_n, nt = tile - mt * ntiles_n;

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

// =====================================================================================================================
// Fast path (same reasoning as igemm_fast_kernel: fp32 MFMA and VALU share the SIMD's execution time, so the staging
// code is stripped of multiplies, divisions and 64-bit address math):
//   * pixel coordinates and byte offsets of the two staged pixels per thread advance incrementally (+32 pixels per
//     step) with wave-uniform increments and carries; one division per kernel, none per step;
//   * buffer-descriptor loads with 32-bit offsets, out-of-range offset for padding taps / the pixel tail (returns 0);
//   * P (the output gradient dz of a conv, or the convT input) and Q are transformed only if they carry a transform
//     (PTR / QTR: 0 plain, 1 none|ReLU + affine, 2 any activation + affine).
// Preconditions (host): operands < 2 GiB, concat boundary multiple of 64, and for per-sample (Group/InstanceNorm) tables
// an image size that is a multiple of the 32-pixel K-step.
template <int KWT, int PTR, int QTR>
__global__ __launch_bounds__(256) void wgrad_fast_kernel(const MsegWgrad p, int splits, int steps_per_split) {
  constexpr int SLAB = WG_PIX * WG_LDS;
  constexpr int STAGE = (1 + KWT) * SLAB;
  __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int Mch = p.P.C, Nch = p.Nch;
  const int T = p.KH * p.KW;
  const int ntiles_n = (Nch + 63) / 64;
  // 1-D grid; logical order: tile fastest, then kernel row, then split -> all workgroups that read the same pixel
  // range (one split) are neighbours in logical order and run on one XCD (shared L2 for the P / Q slabs)
  const int ntiles = ((Mch + 63) / 64) * ntiles_n;
  const int lid = (int)xcd_logical_id(blockIdx.x, gridDim.x);
  const int tile = lid % ntiles;
  const int ky = (lid / ntiles) % p.KH;
  const int split = lid / (ntiles * p.KH);
  const int mt = tile / ntiles_n, nt = tile - mt * ntiles_n;

  const int Ptot = p.NB * p.Hp * p.Wp;
  const int pix_begin = split * steps_per_split * WG_PIX;
  int pix_end = pix_begin + steps_per_split * WG_PIX;
  if (pix_end > Ptot) pix_end = Ptot;
  const int nsteps = pix_end > pix_begin ? (pix_end - pix_begin + WG_PIX - 1) / WG_PIX : 0;

  f32x16 acc[KWT];
#pragma unroll
  for (int t = 0; t < KWT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int spx = tid >> 4, sc4 = tid & 15;
  const int mc = mt * 64 + sc4 * 4;
  const int qc = nt * 64 + sc4 * 4;
  const bool q1 = (p.nq > 1) && (nt * 64 >= p.Q[0].C);      // wave-uniform (concat boundary is a multiple of 64)
  const MsegSrc& qs = q1 ? p.Q[1] : p.Q[0];
  const bool mvalid = mc < Mch, qvalid = qc < Nch;
  const unsigned qC4 = (unsigned)qs.C * 4u, mC4 = (unsigned)Mch * 4u;
  const unsigned qcl4 = (unsigned)(q1 ? qc - p.Q[0].C : qc) * 4u;
  const unsigned OOB = 0x80000000u;

  const __amdgpu_buffer_rsrc_t rsp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.P.ptr), 0, Ptot * Mch * 4,
                                                                        0x00020000);
  const __amdgpu_buffer_rsrc_t rsq = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(qs.ptr), 0, p.NB * p.Hq * p.Wq * qs.C * 4, 0x00020000);

  // per-channel tables: this thread's channel quad is fixed for the whole kernel
  float4 psc, psh, qsc, qsh;
  {
    const float* a = (PTR && p.P.scale) ? p.P.scale + (mvalid ? mc : 0) : g_wg_ident_scale;
    const float* b = (PTR && p.P.scale) ? p.P.shift + (mvalid ? mc : 0) : g_wg_ident_shift;
    const float* c = (QTR && qs.scale) ? qs.scale + (qvalid ? (int)(qcl4 >> 2) : 0) : g_wg_ident_scale;
    const float* d = (QTR && qs.scale) ? qs.shift + (qvalid ? (int)(qcl4 >> 2) : 0) : g_wg_ident_shift;
    psc = *reinterpret_cast<const float4*>(a); psh = *reinterpret_cast<const float4*>(b);
    qsc = *reinterpret_cast<const float4*>(c); qsh = *reinterpret_cast<const float4*>(d);
  }
  const float plo = (p.P.act == MSEG_ACT_RELU) ? 0.f : -3.402823466e38f;
  const float qlo = (qs.act == MSEG_ACT_RELU) ? 0.f : -3.402823466e38f;

  // ---- incremental pixel state of the two staged pixels (rows spx and spx + 16 of the 32-pixel step) ---------
  // pp: flat pixel index; (py, px): position in its image; qoff: byte offset of the Q pixel
  // (py*stride + ky - pad, px*stride - pad) + this thread's channel quad in the current Q source.
  const int adv_y = WG_PIX / p.Wp, adv_x = WG_PIX - adv_y * p.Wp;              // +32 pixels = adv_y rows + adv_x cols
  const int q_dx = p.stride * (int)qC4;                                        // one column right in P -> bytes in Q
  const int q_dy = p.stride * p.Wq * (int)qC4;                                 // one row down in P
  const int q_wrap_x = -p.Wp * q_dx + q_dy;                                    // column carry
  const int q_wrap_y = -p.Hp * q_dy + p.Hq * p.Wq * (int)qC4;                  // row carry into the next image
  int pp[2], py[2], px[2];
  unsigned qoff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int v = pix_begin + spx + 16 * i;
    if (v >= Ptot) v = Ptot - 1;                      // tail rows: any valid pixel; they are masked by pp >= pix_end
    pp[i] = pix_begin + spx + 16 * i;
    const int n = v / (p.Hp * p.Wp);
    const int rem = v - n * (p.Hp * p.Wp);
    py[i] = rem / p.Wp; px[i] = rem - py[i] * p.Wp;
    qoff[i] = (unsigned)(((n * p.Hq + py[i] * p.stride + ky - p.pad) * p.Wq + px[i] * p.stride - p.pad) * (int)qC4) + qcl4;
  }
  const unsigned pvoff0 = (unsigned)spx * mC4 + (unsigned)(mvalid ? mc : 0) * 4u;

  float4 rp[2], rq[KWT][2];
  float pm[2], qm[KWT][2];

  // per-sample (Group/InstanceNorm) tables: a 32-pixel step lies in ONE image (Hp*Wp % 32 == 0, host-checked), so the
  // tables of the step are wave-uniform in n and are re-read only when the image changes
  const bool p_ps = PTR && p.P.scale && p.P.ss != 0;
  const bool q_ps = QTR && qs.scale && qs.ss != 0;
  const int HWp = p.Hp * p.Wp;
  int sn = pix_begin / HWp, srem = pix_begin - sn * HWp;    // image of the next step to load (scalar)
  int tab_n = -1;

  auto issue = [&](int step) {          // loads of `step`; then advances the pixel state to step + 1
    const unsigned psoff = (unsigned)(pix_begin + step * WG_PIX) * mC4;        // scalar
    if ((p_ps || q_ps) && step < nsteps && sn != tab_n) {
      tab_n = sn;
      if (p_ps) {
        psc = *reinterpret_cast<const float4*>(p.P.scale + (size_t)sn * p.P.ss + (mvalid ? mc : 0));
        psh = *reinterpret_cast<const float4*>(p.P.shift + (size_t)sn * p.P.ss + (mvalid ? mc : 0));
      }
      if (q_ps) {
        qsc = *reinterpret_cast<const float4*>(qs.scale + (size_t)sn * qs.ss + (qvalid ? (int)(qcl4 >> 2) : 0));
        qsh = *reinterpret_cast<const float4*>(qs.shift + (size_t)sn * qs.ss + (qvalid ? (int)(qcl4 >> 2) : 0));
      }
    }
    srem += WG_PIX;
    if (srem >= HWp) { srem -= HWp; sn += 1; }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool inr = pp[i] < pix_end;
      const bool pok = inr & mvalid;
      const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
          rsp, pok ? pvoff0 + (unsigned)(16 * i) * mC4 : OOB, psoff, 0));
      rp[i] = make_float4(v[0], v[1], v[2], v[3]);
      if (PTR) pm[i] = pok ? 1.f : 0.f;
      const int qy = py[i] * p.stride + ky - p.pad;
      const bool yok = inr & qvalid & (qy >= 0) & (qy < p.Hq);
      const int qx0 = px[i] * p.stride - p.pad;
#pragma unroll
      for (int kx = 0; kx < KWT; ++kx) {
        const bool ok = yok & (qx0 + kx >= 0) & (qx0 + kx < p.Wq);
        const f32x4 q = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
            rsq, ok ? qoff[i] + (unsigned)kx * qC4 : OOB, 0, 0));
        rq[kx][i] = make_float4(q[0], q[1], q[2], q[3]);
        if (QTR) qm[kx][i] = ok ? 1.f : 0.f;
      }
      // advance by 32 pixels (wave-uniform increments, per-lane carries)
      pp[i] += WG_PIX;
      px[i] += adv_x; py[i] += adv_y;
      qoff[i] += (unsigned)(adv_x * q_dx + adv_y * q_dy);
      const bool cx = px[i] >= p.Wp;
      px[i] -= cx ? p.Wp : 0; py[i] += cx ? 1 : 0;
      qoff[i] += cx ? (unsigned)q_wrap_x : 0u;
      while (py[i] >= p.Hp) { py[i] -= p.Hp; qoff[i] += (unsigned)q_wrap_y; }
    }
  };

  auto xf = [&](float4 v, const float4& sc, const float4& sh, float m, int act, float lo, int tr) -> float4 {
    if (tr == 0) return v;
    if (tr == 2) v = act_fwd4(v, act);
    else { v.x = clamp_lo(v.x, lo); v.y = clamp_lo(v.y, lo); v.z = clamp_lo(v.z, lo); v.w = clamp_lo(v.w, lo); }
    v.x = (v.x * sc.x + sh.x) * m; v.y = (v.y * sc.y + sh.y) * m;
    v.z = (v.z * sc.z + sh.z) * m; v.w = (v.w * sc.w + sh.w) * m;
    return v;
  };

  auto commit = [&](float* stage) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float* row = stage + (spx + 16 * i) * WG_LDS + sc4 * 4;
      *reinterpret_cast<float4*>(row) = xf(rp[i], psc, psh, PTR ? pm[i] : 1.f, p.P.act, plo, PTR);
#pragma unroll
      for (int kx = 0; kx < KWT; ++kx)
        *reinterpret_cast<float4*>(row + (1 + kx) * SLAB) =
            xf(rq[kx][i], qsc, qsh, QTR ? qm[kx][i] : 1.f, qs.act, qlo, QTR);
    }
  };

  if (nsteps > 0) {
    issue(0);
    commit(lds);
  }
  __syncthreads();
  int buf = 0;
  for (int step = 0; step < nsteps; ++step) {
    issue(step + 1);                                  // beyond the last step every row is >= pix_end -> all zeros
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

// =====================================================================================================================
// Halo variant for the 3x3 stride-1 convolutions.  A K-step is a TH x TW block of 32 pixels of one image (TW = largest
// power of two <= 32 dividing the row length, TH = 32 / TW), so the three x-shifted Q slabs of a kernel row are ONE
// (TH) x (TW + 2) halo slab read at column offsets 0 / 1 / 2: Q is loaded and normalised once instead of three times.
// The block position is wave-uniform (scalar registers) and every staged row has a constant offset relative to it, so
// per staged row the K-loop spends one add and a range check on addressing; the LDS rows of the MFMA B fragments
// (pixel k -> slab row (k / TW) * (TW + 2) + k % TW) are per-lane constants.  Blocks that hang over the bottom of the
// image (H % TH != 0) contribute zeros through the P operand.  LDS per stage: (32 + up to 48) x 68 floats.
// TWL = log2(TW) is a template parameter so that the LDS offsets of the B fragments are immediates (the compiler then
// pairs them into ds_read2); P (the output gradient dz) is always a plain operand here (PTR = 0).
template <int TWL, int QTR>
__global__ __launch_bounds__(256) void wgrad_halo_kernel(const MsegWgrad p, int splits, int steps_per_split) {
  constexpr int tw_log2 = TWL;
  constexpr int PTR = 0;
  constexpr int QROWS_MAX = 48;                       // TW = 4: 8 x 6
  constexpr int STAGE = (WG_PIX + QROWS_MAX) * WH_LDS;
  __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int Mch = p.P.C, Nch = p.Nch;
  const int ntiles_n = (Nch + 63) / 64;
  const int ntiles = ((Mch + 63) / 64) * ntiles_n;
  const int lid = (int)xcd_logical_id(blockIdx.x, gridDim.x);
  // kernel row fastest: the three row-workgroups of a (channel tile, split) have neighbouring logical ids, i.e. run on ONE
  // XCD at about the same time, and fetch the same P slab and two thirds of each other's Q rows out of that XCD's L2
  const int ky = lid % 3;
  const int tile = (lid / 3) % ntiles;
  const int split = lid / (ntiles * 3);
  const int mt = tile / ntiles_n, nt = tile - mt * ntiles_n;

  constexpr int TW = 1 << tw_log2, TH = WG_PIX >> tw_log2, QW = TW + 2;
  constexpr int QROWS = TH * QW;
  const int bx = p.Wp >> tw_log2, by = (p.Hp + TH - 1) / TH;      // blocks per image row / column
  const int steps_img = bx * by;
  const int steps_total = p.NB * steps_img;
  const int step_begin = split * steps_per_split;
  int step_end = step_begin + steps_per_split;
  if (step_end > steps_total) step_end = steps_total;
  const int nsteps = step_end > step_begin ? step_end - step_begin : 0;

  f32x16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int spx = tid >> 4, sc4 = tid & 15;
  const int mc = mt * 64 + sc4 * 4;
  const int qc = nt * 64 + sc4 * 4;
  const bool q1 = (p.nq > 1) && (nt * 64 >= p.Q[0].C);
  const MsegSrc& qs = q1 ? p.Q[1] : p.Q[0];
  const bool mvalid = mc < Mch, qvalid = qc < Nch;
  const unsigned qC4 = (unsigned)qs.C * 4u, mC4 = (unsigned)Mch * 4u;
  const unsigned qcl4 = (unsigned)(q1 ? qc - p.Q[0].C : qc) * 4u;
  const unsigned OOB = 0x80000000u;
  // buffer descriptors of ONE image, re-based whenever the walk enters the next image: 32-bit offsets only have to span
  // an image (< 2 GiB, host-checked), so the batch size is unlimited
  const int HWp = p.Hp * p.Wp;
  __amdgpu_buffer_rsrc_t rsp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.P.ptr), 0, HWp * Mch * 4, 0x00020000);
  __amdgpu_buffer_rsrc_t rsq = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(qs.ptr), 0, HWp * qs.C * 4, 0x00020000);
  int desc_n = 0;
  float4 psc, psh, qsc, qsh;
  {
    const float* a = (PTR && p.P.scale) ? p.P.scale + (mvalid ? mc : 0) : g_wg_ident_scale;
    const float* b = (PTR && p.P.scale) ? p.P.shift + (mvalid ? mc : 0) : g_wg_ident_shift;
    const float* c = (QTR && qs.scale) ? qs.scale + (qvalid ? (int)(qcl4 >> 2) : 0) : g_wg_ident_scale;
    const float* d = (QTR && qs.scale) ? qs.shift + (qvalid ? (int)(qcl4 >> 2) : 0) : g_wg_ident_shift;
    psc = *reinterpret_cast<const float4*>(a); psh = *reinterpret_cast<const float4*>(b);
    qsc = *reinterpret_cast<const float4*>(c); qsh = *reinterpret_cast<const float4*>(d);
  }
  const float plo = (p.P.act == MSEG_ACT_RELU) ? 0.f : -3.402823466e38f;
  const float qlo = (qs.act == MSEG_ACT_RELU) ? 0.f : -3.402823466e38f;

  // wave-uniform position of the NEXT block to load: image sn, first row spy, first column spx0
  int sn = step_begin / steps_img;
  const int srem0 = step_begin - sn * steps_img;
  int spy = (srem0 / bx) * TH, spx0 = (srem0 - (srem0 / bx) * bx) * TW;
  // per-thread constants.  P rows k = spx, spx + 16 (pixel (k / TW, k % TW) of the block); Q slab rows s = spx, spx + 16,
  // spx + 32 (halo position (s / QW, s % QW), column qx = px0 - pad + s % QW)
  int pr[2], pcol[2];
  unsigned pv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int k = spx + 16 * i;
    pr[i] = k >> tw_log2; pcol[i] = k & (TW - 1);
    pv[i] = (unsigned)(pr[i] * p.Wp + pcol[i]) * mC4 + (unsigned)(mvalid ? mc : 0) * 4u;
  }
  int qr[3], qj[3];
  unsigned qv[3];
  bool qlive[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int sidx = spx + 16 * i;
    qlive[i] = sidx < QROWS;
    qr[i] = sidx / QW; qj[i] = sidx - qr[i] * QW;
    qv[i] = (unsigned)(qr[i] * p.Wq + qj[i]) * qC4 + qcl4;
  }
  // LDS float offset of the B fragment of pixel k = 2 kk + lh: slab row (k / TW) * QW + k % TW.  TW >= 2, so k and k - lh
  // lie in the same block row: offset = compile-time part (kk) + lane part (lh, wn, li)
  const int blane = lh * WH_LDS + wn * 32 + li;

  float4 rp[2], rq[3];
  float pm[2], qm[3];

  // per-sample (Group/InstanceNorm) tables: a block lies in ONE image, so the tables of the step are wave-uniform in n
  // and are re-read only when the image changes
  const bool p_ps = PTR && p.P.scale && p.P.ss != 0;
  const bool q_ps = QTR && qs.scale && qs.ss != 0;
  int tab_n = -1;

  auto issue = [&](int step) {
    const bool live = step < nsteps;                                             // scalar
    if ((p_ps || q_ps) && live && sn != tab_n) {
      tab_n = sn;
      if (p_ps) {
        psc = *reinterpret_cast<const float4*>(p.P.scale + (size_t)sn * p.P.ss + (mvalid ? mc : 0));
        psh = *reinterpret_cast<const float4*>(p.P.shift + (size_t)sn * p.P.ss + (mvalid ? mc : 0));
      }
      if (q_ps) {
        qsc = *reinterpret_cast<const float4*>(qs.scale + (size_t)sn * qs.ss + (qvalid ? (int)(qcl4 >> 2) : 0));
        qsh = *reinterpret_cast<const float4*>(qs.shift + (size_t)sn * qs.ss + (qvalid ? (int)(qcl4 >> 2) : 0));
      }
    }
    if (live && sn != desc_n) {                                                  // scalar: next image
      desc_n = sn;
      rsp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.P.ptr + (size_t)sn * HWp * Mch), 0, HWp * Mch * 4,
                                              0x00020000);
      rsq = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(qs.ptr + (size_t)sn * HWp * qs.C), 0, HWp * qs.C * 4,
                                              0x00020000);
    }
    const unsigned psoff = (unsigned)(spy * p.Wp + spx0) * mC4;                  // scalar: first pixel of the block
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool pok = live & mvalid & (spy + pr[i] < p.Hp);                     // rows below the image: zeros
      const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsp, pok ? pv[i] : OOB, psoff, 0));
      rp[i] = make_float4(v[0], v[1], v[2], v[3]);
      if (PTR) pm[i] = pok ? 1.f : 0.f;
    }
    const int qy0 = spy + ky - p.pad, qx0 = spx0 - p.pad;                        // scalar: halo origin
    const unsigned sb = (unsigned)(qy0 * p.Wq + qx0) * qC4;                      // scalar, wraps by design (halo rows < 0)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (i < 2 || qlive[i]) {
        const int qy = qy0 + qr[i], qx = qx0 + qj[i];
        const bool ok = live & qvalid & qlive[i] & (qy >= 0) & (qy < p.Hq) & (qx >= 0) & (qx < p.Wq);
        const f32x4 q = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsq, ok ? sb + qv[i] : OOB, 0, 0));
        rq[i] = make_float4(q[0], q[1], q[2], q[3]);
        if (QTR) qm[i] = ok ? 1.f : 0.f;
      }
    }
    // advance the scalar position by one block
    spx0 += TW;
    if (spx0 >= p.Wp) { spx0 = 0; spy += TH; if (spy >= p.Hp) { spy = 0; sn += 1; } }
  };

  auto xf = [&](float4 v, const float4& sc, const float4& sh, float m, int act, float lo, int tr) -> float4 {
    if (tr == 0) return v;
    if (tr == 2) v = act_fwd4(v, act);
    else { v.x = clamp_lo(v.x, lo); v.y = clamp_lo(v.y, lo); v.z = clamp_lo(v.z, lo); v.w = clamp_lo(v.w, lo); }
    v.x = (v.x * sc.x + sh.x) * m; v.y = (v.y * sc.y + sh.y) * m;
    v.z = (v.z * sc.z + sh.z) * m; v.w = (v.w * sc.w + sh.w) * m;
    return v;
  };

  auto commit = [&](float* stage) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      *reinterpret_cast<float4*>(stage + (spx + 16 * i) * WH_LDS + sc4 * 4) =
          xf(rp[i], psc, psh, PTR ? pm[i] : 1.f, p.P.act, plo, PTR);
    float* qst = stage + WG_PIX * WH_LDS;
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (i < 2 || qlive[i])
        *reinterpret_cast<float4*>(qst + (spx + 16 * i) * WH_LDS + sc4 * 4) =
            xf(rq[i], qsc, qsh, QTR ? qm[i] : 1.f, qs.act, qlo, QTR);
  };

  if (nsteps > 0) {
    issue(0);
    commit(lds);
  }
  __syncthreads();
  int buf = 0;
  for (int step = 0; step < nsteps; ++step) {
    issue(step + 1);
    const float* st = lds + buf * STAGE;
    const float* qst = st + WG_PIX * WH_LDS + blane;
    float a[16];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) a[kk] = st[(2 * kk + lh) * WH_LDS + wm * 32 + li];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        const float b = qst[((((2 * kk) >> tw_log2) * QW + ((2 * kk) & (TW - 1))) + kx) * WH_LDS];   // column + kx of the slab
        acc[kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b, acc[kx], 0, 0, 0);
      }
    }
    commit(lds + (buf ^ 1) * STAGE);
    __syncthreads();
    buf ^= 1;
  }

  const int n = nt * 64 + wn * 32 + li;
  if (n < Nch) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int t = ky * 3 + kx;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mt * 64 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < Mch) p.ws[(((size_t)split * 9 + t) * Mch + m) * Nch + n] = acc[kx][r];
      }
    }
  }
}

// =====================================================================================================================
// All nine taps in one workgroup (3x3 stride-1 layers whose rows are a multiple of 4; the default network: every layer).
// wgrad_halo_kernel gives each kernel ROW its own workgroup, so P and two thirds of Q are fetched three times (measured:
// 1.2 GB of HBM traffic per launch against 0.56 GB of operands).  Here a workgroup owns a 64 x 64 channel tile for ALL taps
// (9 x 16 accumulator registers per lane, two workgroups per CU) and walks 32-pixel blocks of TH x TW = 4 x 8 (or 8 x 4)
// pixels: per block ONE (TH + 2) x (TW + 2) = 60-pixel halo image of Q and the 32-pixel P slab are staged, and every tap
// reads its MFMA B fragments from the halo at a constant row offset (ky * (TW + 2) + kx) — 144 MFMAs per wave and
// barrier against 6 staged float4 per thread (the row kernel: 48 against 5), operands fetched once.
template <int TWL, int QTR>
__global__ __launch_bounds__(256, 2) void wgrad_halo9_kernel(const MsegWgrad p, int splits, int steps_per_split) {
  constexpr int tw_log2 = TWL;
  constexpr int PTR = 0;
  constexpr int QROWS_MAX = 64;                       // (TH + 2) x (TW + 2) = 6 x 10 or 10 x 6 = 60 rows, staged in 4 passes
  constexpr int STAGE = (WG_PIX + QROWS_MAX) * WH_LDS;
  __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int Mch = p.P.C, Nch = p.Nch;
  const int ntiles_n = (Nch + 63) / 64;
  const int ntiles = ((Mch + 63) / 64) * ntiles_n;
  const int lid = (int)xcd_logical_id(blockIdx.x, gridDim.x);
  // channel tiles of one split (= one pixel range) are neighbours: they run on one XCD and share its L2 for P and Q
  const int tile = lid % ntiles;
  const int split = lid / ntiles;
  const int mt = tile / ntiles_n, nt = tile - mt * ntiles_n;

  constexpr int TW = 1 << tw_log2, TH = WG_PIX >> tw_log2, QW = TW + 2;
  constexpr int QROWS = (TH + 2) * QW;                 // ONE halo image serves all nine taps
  const int bx = p.Wp >> tw_log2, by = (p.Hp + TH - 1) / TH;      // blocks per image row / column
  const int steps_img = bx * by;
  const int steps_total = p.NB * steps_img;
  const int step_begin = split * steps_per_split;
  int step_end = step_begin + steps_per_split;
  if (step_end > steps_total) step_end = steps_total;
  const int nsteps = step_end > step_begin ? step_end - step_begin : 0;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int spx = tid >> 4, sc4 = tid & 15;
  const int mc = mt * 64 + sc4 * 4;
  const int qc = nt * 64 + sc4 * 4;
  const bool q1 = (p.nq > 1) && (nt * 64 >= p.Q[0].C);
  const MsegSrc& qs = q1 ? p.Q[1] : p.Q[0];
  const bool mvalid = mc < Mch, qvalid = qc < Nch;
  const unsigned qC4 = (unsigned)qs.C * 4u, mC4 = (unsigned)Mch * 4u;
  const unsigned qcl4 = (unsigned)(q1 ? qc - p.Q[0].C : qc) * 4u;
  const unsigned OOB = 0x80000000u;
  // buffer descriptors of ONE image, re-based whenever the walk enters the next image: 32-bit offsets only have to span
  // an image (< 2 GiB, host-checked), so the batch size is unlimited
  const int HWp = p.Hp * p.Wp;
  __amdgpu_buffer_rsrc_t rsp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.P.ptr), 0, HWp * Mch * 4, 0x00020000);
  __amdgpu_buffer_rsrc_t rsq = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(qs.ptr), 0, HWp * qs.C * 4, 0x00020000);
  int desc_n = 0;
  float4 psc, psh, qsc, qsh;
  {
    const float* a = (PTR && p.P.scale) ? p.P.scale + (mvalid ? mc : 0) : g_wg_ident_scale;
    const float* b = (PTR && p.P.scale) ? p.P.shift + (mvalid ? mc : 0) : g_wg_ident_shift;
    const float* c = (QTR && qs.scale) ? qs.scale + (qvalid ? (int)(qcl4 >> 2) : 0) : g_wg_ident_scale;
    const float* d = (QTR && qs.scale) ? qs.shift + (qvalid ? (int)(qcl4 >> 2) : 0) : g_wg_ident_shift;
    psc = *reinterpret_cast<const float4*>(a); psh = *reinterpret_cast<const float4*>(b);
    qsc = *reinterpret_cast<const float4*>(c); qsh = *reinterpret_cast<const float4*>(d);
  }
  const float plo = (p.P.act == MSEG_ACT_RELU) ? 0.f : -3.402823466e38f;
  const float qlo = (qs.act == MSEG_ACT_RELU) ? 0.f : -3.402823466e38f;

  // wave-uniform position of the NEXT block to load: image sn, first row spy, first column spx0
  int sn = step_begin / steps_img;
  const int srem0 = step_begin - sn * steps_img;
  int spy = (srem0 / bx) * TH, spx0 = (srem0 - (srem0 / bx) * bx) * TW;
  // per-thread constants.  P rows k = spx, spx + 16 (pixel (k / TW, k % TW) of the block); Q slab rows s = spx, spx + 16,
  // spx + 32 (halo position (s / QW, s % QW), column qx = px0 - pad + s % QW)
  int pr[2], pcol[2];
  unsigned pv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int k = spx + 16 * i;
    pr[i] = k >> tw_log2; pcol[i] = k & (TW - 1);
    pv[i] = (unsigned)(pr[i] * p.Wp + pcol[i]) * mC4 + (unsigned)(mvalid ? mc : 0) * 4u;
  }
  int qr[4], qj[4];
  unsigned qv[4];
  bool qlive[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int sidx = spx + 16 * i;
    qlive[i] = sidx < QROWS;
    qr[i] = sidx / QW; qj[i] = sidx - qr[i] * QW;
    qv[i] = (unsigned)(qr[i] * p.Wq + qj[i]) * qC4 + qcl4;
  }
  // LDS float offset of the B fragment of pixel k = 2 kk + lh: slab row (k / TW) * QW + k % TW.  TW >= 2, so k and k - lh
  // lie in the same block row: offset = compile-time part (kk) + lane part (lh, wn, li)
  const int blane = lh * WH_LDS + wn * 32 + li;

  float4 rp[2], rq[4];
  float pm[2], qm[4];

  // per-sample (Group/InstanceNorm) tables: a block lies in ONE image, so the tables of the step are wave-uniform in n
  // and are re-read only when the image changes
  const bool p_ps = PTR && p.P.scale && p.P.ss != 0;
  const bool q_ps = QTR && qs.scale && qs.ss != 0;
  int tab_n = -1;

  auto issue = [&](int step) {
    const bool live = step < nsteps;                                             // scalar
    if ((p_ps || q_ps) && live && sn != tab_n) {
      tab_n = sn;
      if (p_ps) {
        psc = *reinterpret_cast<const float4*>(p.P.scale + (size_t)sn * p.P.ss + (mvalid ? mc : 0));
        psh = *reinterpret_cast<const float4*>(p.P.shift + (size_t)sn * p.P.ss + (mvalid ? mc : 0));
      }
      if (q_ps) {
        qsc = *reinterpret_cast<const float4*>(qs.scale + (size_t)sn * qs.ss + (qvalid ? (int)(qcl4 >> 2) : 0));
        qsh = *reinterpret_cast<const float4*>(qs.shift + (size_t)sn * qs.ss + (qvalid ? (int)(qcl4 >> 2) : 0));
      }
    }
    if (live && sn != desc_n) {                                                  // scalar: next image
      desc_n = sn;
      rsp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.P.ptr + (size_t)sn * HWp * Mch), 0, HWp * Mch * 4,
                                              0x00020000);
      rsq = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(qs.ptr + (size_t)sn * HWp * qs.C), 0, HWp * qs.C * 4,
                                              0x00020000);
    }
    const unsigned psoff = (unsigned)(spy * p.Wp + spx0) * mC4;                  // scalar: first pixel of the block
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool pok = live & mvalid & (spy + pr[i] < p.Hp);                     // rows below the image: zeros
      const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsp, pok ? pv[i] : OOB, psoff, 0));
      rp[i] = make_float4(v[0], v[1], v[2], v[3]);
      if (PTR) pm[i] = pok ? 1.f : 0.f;
    }
    const int qy0 = spy - p.pad, qx0 = spx0 - p.pad;                             // scalar: halo origin
    const unsigned sb = (unsigned)(qy0 * p.Wq + qx0) * qC4;                      // scalar, wraps by design (halo rows < 0)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < 3 || qlive[i]) {
        const int qy = qy0 + qr[i], qx = qx0 + qj[i];
        const bool ok = live & qvalid & qlive[i] & (qy >= 0) & (qy < p.Hq) & (qx >= 0) & (qx < p.Wq);
        const f32x4 q = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsq, ok ? sb + qv[i] : OOB, 0, 0));
        rq[i] = make_float4(q[0], q[1], q[2], q[3]);
        if (QTR) qm[i] = ok ? 1.f : 0.f;
      }
    }
    // advance the scalar position by one block
    spx0 += TW;
    if (spx0 >= p.Wp) { spx0 = 0; spy += TH; if (spy >= p.Hp) { spy = 0; sn += 1; } }
  };

  auto xf = [&](float4 v, const float4& sc, const float4& sh, float m, int act, float lo, int tr) -> float4 {
    if (tr == 0) return v;
    if (tr == 2) v = act_fwd4(v, act);
    else { v.x = clamp_lo(v.x, lo); v.y = clamp_lo(v.y, lo); v.z = clamp_lo(v.z, lo); v.w = clamp_lo(v.w, lo); }
    v.x = (v.x * sc.x + sh.x) * m; v.y = (v.y * sc.y + sh.y) * m;
    v.z = (v.z * sc.z + sh.z) * m; v.w = (v.w * sc.w + sh.w) * m;
    return v;
  };

  auto commit = [&](float* stage) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      *reinterpret_cast<float4*>(stage + (spx + 16 * i) * WH_LDS + sc4 * 4) =
          xf(rp[i], psc, psh, PTR ? pm[i] : 1.f, p.P.act, plo, PTR);
    float* qst = stage + WG_PIX * WH_LDS;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < 3 || qlive[i])
        *reinterpret_cast<float4*>(qst + (spx + 16 * i) * WH_LDS + sc4 * 4) =
            xf(rq[i], qsc, qsh, QTR ? qm[i] : 1.f, qs.act, qlo, QTR);
  };

  if (nsteps > 0) {
    issue(0);
    commit(lds);
  }
  __syncthreads();
  int buf = 0;
  for (int step = 0; step < nsteps; ++step) {
    issue(step + 1);
    const float* st = lds + buf * STAGE;
    const float* qst = st + WG_PIX * WH_LDS + blane;
    float a[16];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) a[kk] = st[(2 * kk + lh) * WH_LDS + wm * 32 + li];
#pragma unroll
    for (int t = 0; t < 9; ++t) {                      // tap (ky, kx) = row + ky, column + kx of the halo image
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        const float b = qst[(((((2 * kk) >> tw_log2) + t / 3) * QW + ((2 * kk) & (TW - 1))) + t % 3) * WH_LDS];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b, acc[t], 0, 0, 0);
      }
    }
    commit(lds + (buf ^ 1) * STAGE);
    __syncthreads();
    buf ^= 1;
  }

  const int n = nt * 64 + wn * 32 + li;
  if (n < Nch) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mt * 64 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < Mch) p.ws[(((size_t)split * 9 + t) * Mch + m) * Nch + n] = acc[t][r];
      }
    }
  }
}

// =====================================================================================================================
// bf16 variant of the halo kernel (BASELINE configs[2]; MsegWgrad.precision == MSEG_PREC_BF16).
// The contraction of a weight gradient runs over PIXELS, while the tensors are channel-contiguous (NHWC): the matrix cores
// want every lane to hold 8 consecutive pixels of one channel.  gfx950's transposing LDS read (ds_read_b64_tr_b16) does
// that for free: P and Q are staged exactly as they arrive — [pixel][32 channels] bf16 rows of 64 B, two such planes per
// 64-channel tile (conflict-free for the transposed reads) — and a lane group of 16 reads a 4-pixel x 16-channel block
// column-major.  A tap shift is a ROW offset of the Q halo image, so one (TH + 2) x (TW + 2) halo serves all nine taps:
// a workgroup (4 waves, 32 x 32 channels each) owns a 64 x 64 channel tile for ALL 9 taps (9 x 16 accumulator registers)
// and walks 64-pixel blocks (8 x 8, or 16 x 4 when the row length is no multiple of 8): 36 v_mfma_f32_32x32x16_bf16 per
// wave and block against 11 staged float4 per thread — three times the arithmetic intensity of the fp32 kernel, which
// the 16x faster matrix pipe needs.  Operands are rounded to bf16 (RNE) after the norm-on-load transform; fp32 accumulate.
typedef __bf16 wbf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 wbf16x8 __attribute__((ext_vector_type(8)));
typedef short ws16x4 __attribute__((ext_vector_type(4)));
typedef short ws16x8 __attribute__((ext_vector_type(8)));
#define WB_PIX 64

__device__ __forceinline__ wbf16x8 wb_tr_read8(const __bf16* lo, const __bf16* hi) {
  const ws16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ws16x4*)lo);
  const ws16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ws16x4*)hi);
  ws16x8 v;
  v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
  return __builtin_bit_cast(wbf16x8, v);
}

// The same kernel covers the strided layers (S = 2): KT = 3 is the weight gradient of a 3x3 stride-2 convolution (ConvPool;
// Q = its input, twice the size of P = dz), KT = 2 that of ConvTranspose2d 2x2 stride 2 (P = its input, with norm-on-load
// transform PTR; Q = the gradient of its output).  The Q image of a block is (S TH + KT - S) x (S TW + KT - S) pixels, a tap
// is still a row offset and consecutive block pixels are S image rows apart — every lane of a transposed read supplies
// its own row address, so the stride costs nothing.  Blocks are 32 pixels for S = 2 (the Q image is four times the block).
// S16: P and Q are bf16 tensors (MSEG_ST_BF16): a staging thread owns 8 channels of a row (one 16-byte load), half the
// passes; a plain operand goes to LDS as it arrives.
template <int TWL, int QTR, int S = 1, int KT = 3, int PTR = 0, bool S16 = false>
__global__ __launch_bounds__(256, 2) void wgrad_halo_bf16_kernel(const MsegWgrad p, int splits, int steps_per_split) {
  constexpr int PIX = S == 1 ? WB_PIX : 32, NT = KT * KT;
  constexpr int TW = 1 << TWL, TH = PIX >> TWL, QW = S * TW + KT - S, QH = S * TH + KT - S;
  constexpr int QROWS = QH * QW;                       // S = 1: 100 (8 x 8 blocks) or 108 (16 x 4); S = 2: 153 or 128
  constexpr int SQN = S16 ? 8 : 16;                    // staging threads per pixel row (8 / 4 channels each)
  constexpr int RP = 256 / SQN;                        // pixel rows per staging pass
  constexpr int CPT = S16 ? 8 : 4, ESZ = S16 ? 2 : 4;
  constexpr int NP = PIX / RP, NQ = (QROWS + RP - 1) / RP;
  constexpr int PEL = 2 * PIX * 32, QEL = 2 * QROWS * 32;
  constexpr int STAGE = PEL + QEL;
  __shared__ __attribute__((aligned(16))) __bf16 lds[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int Mch = p.P.C, Nch = p.Nch;
  const int ntiles_n = (Nch + 63) / 64;
  const int ntiles = ((Mch + 63) / 64) * ntiles_n;
  const int lid = (int)xcd_logical_id(blockIdx.x, gridDim.x);
  const int tile = lid % ntiles;
  const int split = lid / ntiles;
  const int mt = tile / ntiles_n, nt = tile - mt * ntiles_n;

  const int bx = p.Wp >> TWL, by = (p.Hp + TH - 1) / TH;
  const int steps_img = bx * by;
  const int steps_total = p.NB * steps_img;
  const int step_begin = split * steps_per_split;
  int step_end = step_begin + steps_per_split;
  if (step_end > steps_total) step_end = steps_total;
  const int nsteps = step_end > step_begin ? step_end - step_begin : 0;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int spx = tid / SQN, sc4 = tid % SQN;
  const int mc = mt * 64 + sc4 * CPT;
  const int qc = nt * 64 + sc4 * CPT;
  const bool q1 = (p.nq > 1) && (nt * 64 >= p.Q[0].C);
  const MsegSrc& qs = q1 ? p.Q[1] : p.Q[0];
  const bool mvalid = mc < Mch, qvalid = qc < Nch;
  const unsigned qC4 = (unsigned)qs.C * (unsigned)ESZ, mC4 = (unsigned)Mch * (unsigned)ESZ;     // bytes per pixel
  const unsigned qcl4 = (unsigned)(q1 ? qc - p.Q[0].C : qc) * (unsigned)ESZ;                    // byte offset of the channels
  const int qcl = q1 ? qc - p.Q[0].C : qc;
  const unsigned OOB = 0x80000000u;
  const int HWp = p.Hp * p.Wp, HWq = p.Hq * p.Wq;
  __amdgpu_buffer_rsrc_t rsp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.P.ptr), 0, HWp * Mch * ESZ, 0x00020000);
  __amdgpu_buffer_rsrc_t rsq = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(qs.ptr), 0, HWq * qs.C * ESZ, 0x00020000);
  int desc_n = 0;
  float4 qsc, qsh, psc, psh, qsc2, qsh2, psc2, psh2;   // ...2: channels 4..7 of a bf16-source thread
  if (PTR) {
    const float* a = p.P.scale ? p.P.scale + (mvalid ? mc : 0) : g_wg_ident_scale;
    const float* b = p.P.scale ? p.P.shift + (mvalid ? mc : 0) : g_wg_ident_shift;
    psc = *reinterpret_cast<const float4*>(a); psh = *reinterpret_cast<const float4*>(b);
    if (S16) { psc2 = *reinterpret_cast<const float4*>(a + 4); psh2 = *reinterpret_cast<const float4*>(b + 4); }
  }
  const float plo = (p.P.act == MSEG_ACT_RELU) ? 0.f : -3.402823466e38f;
  {
    const float* c = (QTR && qs.scale) ? qs.scale + (qvalid ? qcl : 0) : g_wg_ident_scale;
    const float* d = (QTR && qs.scale) ? qs.shift + (qvalid ? qcl : 0) : g_wg_ident_shift;
    qsc = *reinterpret_cast<const float4*>(c); qsh = *reinterpret_cast<const float4*>(d);
    if (S16) { qsc2 = *reinterpret_cast<const float4*>(c + 4); qsh2 = *reinterpret_cast<const float4*>(d + 4); }
  }
  const float qlo = (qs.act == MSEG_ACT_RELU) ? 0.f : -3.402823466e38f;

  int sn = step_begin / steps_img;
  const int srem0 = step_begin - sn * steps_img;
  int spy = (srem0 / bx) * TH, spx0 = (srem0 - (srem0 / bx) * bx) * TW;
  // per-thread constants: P rows k = spx + 16 i (block pixel (k / TW, k % TW)), Q halo rows s = spx + 16 i
  // (block row of P row k = spx + 16 i: (spx >> TWL) + (16 >> TWL) * i — derived, not stored)
  // P row k = spx + 16 i is block pixel ((spx >> TWL) + (16 >> TWL) i, spx % TW): one per-lane offset, the step between
  // the rows of a thread is wave-uniform.  Address constants are kept few on purpose: the 144 accumulators + 44 staging
  // registers leave no room, and a spilled constant is reloaded in every step.
  const int pr0 = spx >> TWL;
  const unsigned pv0 = (unsigned)(pr0 * p.Wp + (spx & (TW - 1))) * mC4 + (unsigned)(mvalid ? mc : 0) * (unsigned)ESZ;
  const unsigned pvstep = (unsigned)((RP >> TWL) * p.Wp) * mC4;                  // scalar
  int qrj[NQ];                                         // halo row << 8 | halo column
  unsigned qoff[NQ];                                   // byte offset of that halo pixel (this thread's channels) from the
#pragma unroll                                         // block's halo origin: the per-step part is ONE scalar
  for (int i = 0; i < NQ; ++i) {
    const int sidx = spx + RP * i;
    const int r = sidx / QW, c = sidx - r * QW;
    qrj[i] = (r << 8) | c;
    qoff[i] = (unsigned)(r * p.Wq + c) * qC4 + qcl4;
  }
  // LDS element offsets of this thread's staging writes: plane (sc4 >> 3), row, 4 channels at (sc4 & 7) * 4
  const int wplane = (sc4 * CPT) >> 5, wcol = (sc4 * CPT) & 31;
  // transposed-read addresses: lane 4q + pp of a 16-lane group supplies row q, channels 4 pp .. 4 pp + 3 of its block;
  // the group (lane >> 4) & 1 takes channels 16 .. 31 of the wave's 32, the half lane >> 5 the pixels 8 .. 15 of a k-step
  const int tq = (lane >> 2) & 3, tp = lane & 3, tcb = (lane >> 4) & 1;
  const int a_lane = (wm * PIX + 8 * lh + tq) * 32 + tcb * 16 + tp * 4;
  const int b_lane = PEL + (wn * QROWS + (TWL == 3 ? S * lh * QW : 2 * S * lh * QW) + S * tq) * 32 + tcb * 16 + tp * 4;

  f32x4 rp[NP], rq[NQ];                                // 16 raw bytes: 4 fp32 or 8 bf16 channels
  unsigned qmask = 0u;                                 // bit i: halo row i of the step in flight is a real pixel
  unsigned pmask = 0u;                                 // same for the P rows (PTR only)
  const bool q_ps = QTR && qs.scale && qs.ss != 0;
  const bool p_ps = PTR && p.P.scale && p.P.ss != 0;
  int tab_n = -1;

  auto issue = [&](int step) {
    const bool live = step < nsteps;
    if ((q_ps || p_ps) && live && sn != tab_n) {
      tab_n = sn;
      if (q_ps) {
        const size_t o = (size_t)sn * qs.ss + (qvalid ? qcl : 0);
        qsc = *reinterpret_cast<const float4*>(qs.scale + o);
        qsh = *reinterpret_cast<const float4*>(qs.shift + o);
        if (S16) { qsc2 = *reinterpret_cast<const float4*>(qs.scale + o + 4); qsh2 = *reinterpret_cast<const float4*>(qs.shift + o + 4); }
      }
      if (p_ps) {
        const size_t o = (size_t)sn * p.P.ss + (mvalid ? mc : 0);
        psc = *reinterpret_cast<const float4*>(p.P.scale + o);
        psh = *reinterpret_cast<const float4*>(p.P.shift + o);
        if (S16) { psc2 = *reinterpret_cast<const float4*>(p.P.scale + o + 4); psh2 = *reinterpret_cast<const float4*>(p.P.shift + o + 4); }
      }
    }
    if (live && sn != desc_n) {
      desc_n = sn;
      rsp = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>((const char*)p.P.ptr + (size_t)sn * HWp * Mch * ESZ), 0,
                                              HWp * Mch * ESZ, 0x00020000);
      rsq = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>((const char*)qs.ptr + (size_t)sn * HWq * qs.C * ESZ), 0,
                                              HWq * qs.C * ESZ, 0x00020000);
    }
    const unsigned psoff = (unsigned)(spy * p.Wp + spx0) * mC4;
    unsigned pbits = 0u;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const bool pok = live & mvalid & (spy + pr0 + (RP >> TWL) * i < p.Hp);
      rp[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsp, pok ? pv0 + pvstep * i : OOB, psoff, 0));
      pbits |= (unsigned)pok << i;
    }
    pmask = pbits;
    const int qy0 = S * spy - p.pad, qx0 = S * spx0 - p.pad;
    const unsigned qbase = (unsigned)(qy0 * p.Wq + qx0) * qC4;       // scalar; wraps for the rows above the image (never loaded)
    unsigned okbits = 0u;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const bool qlive = spx + RP * i < QROWS;
      const int qy = qy0 + (qrj[i] >> 8), qx = qx0 + (qrj[i] & 255);
      const bool ok = live & qvalid & qlive & ((unsigned)qy < (unsigned)p.Hq) & ((unsigned)qx < (unsigned)p.Wq);
      const unsigned qo = qbase + qoff[i];
      rq[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsq, ok ? qo : OOB, 0, 0));
      okbits |= (unsigned)ok << i;
    }
    qmask = okbits;
    spx0 += TW;
    if (spx0 >= p.Wp) { spx0 = 0; spy += TH; if (spy >= p.Hp) { spy = 0; sn += 1; } }
  };

  auto to_bf = [](const float4& v) -> wbf16x4 {
    wbf16x4 h;
    h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
    return h;
  };

  auto xf = [&](float4 v, const float4& sc, const float4& sh, int act, float lo, float m, int tr) -> float4 {
    if (tr == 2) v = act_fwd4(v, act);
    else { v.x = clamp_lo(v.x, lo); v.y = clamp_lo(v.y, lo); v.z = clamp_lo(v.z, lo); v.w = clamp_lo(v.w, lo); }
    v.x = (v.x * sc.x + sh.x) * m; v.y = (v.y * sc.y + sh.y) * m;
    v.z = (v.z * sc.z + sh.z) * m; v.w = (v.w * sc.w + sh.w) * m;
    return v;
  };
  // one staged row segment (4 fp32 or 8 bf16 channels) -> bf16 in LDS at element offset e
  auto put = [&](__bf16* stage, int e, const f32x4& raw, int tr, const float4& sc, const float4& sh, const float4& sc2,
                 const float4& sh2, int act, float lo, float m) {
    if (S16) {
      const uint4 r = __builtin_bit_cast(uint4, raw);
      if (tr == 0) {
        *reinterpret_cast<uint4*>(stage + e) = r;
      } else if (tr == 1) {
        *reinterpret_cast<uint4*>(stage + e) = mseg_affine8_bf16(r, lo == 0.f ? 0u : 0x80008000u, sc, sh, sc2, sh2, m != 0.f);
      } else {
        const uint2 a = f32x4_to_bf16(xf(bf16x4_to_f32(make_uint2(r.x, r.y)), sc, sh, act, lo, m, tr));
        const uint2 b = f32x4_to_bf16(xf(bf16x4_to_f32(make_uint2(r.z, r.w)), sc2, sh2, act, lo, m, tr));
        *reinterpret_cast<uint4*>(stage + e) = make_uint4(a.x, a.y, b.x, b.y);
      }
    } else {
      float4 v = make_float4(raw[0], raw[1], raw[2], raw[3]);
      if (tr != 0) v = xf(v, sc, sh, act, lo, m, tr);
      *reinterpret_cast<wbf16x4*>(stage + e) = to_bf(v);
    }
  };

  auto commit = [&](__bf16* stage) {
#pragma unroll
    for (int i = 0; i < NP; ++i)
      put(stage, (wplane * PIX + spx + RP * i) * 32 + wcol, rp[i], PTR, psc, psh, psc2, psh2, p.P.act, plo,
          ((pmask >> i) & 1u) ? 1.f : 0.f);
#pragma unroll
    for (int i = 0; i < NQ; ++i)
      if (spx + RP * i < QROWS)
        put(stage, PEL + (wplane * QROWS + spx + RP * i) * 32 + wcol, rq[i], QTR, qsc, qsh, qsc2, qsh2, qs.act, qlo,
            ((qmask >> i) & 1u) ? 1.f : 0.f);
  };

  if (nsteps > 0) {
    issue(0);
    commit(lds);
  }
  __syncthreads();
  int buf = 0;
  for (int step = 0; step < nsteps; ++step) {
    issue(step + 1);
    const __bf16* st = lds + buf * STAGE;
    const __bf16* ap = st + a_lane;
    const __bf16* bp = st + b_lane;
#pragma unroll
    for (int s4 = 0; s4 < PIX / 16; ++s4) {
      // pixels k0 = 16 s4 + 8 lh (+4): P rows are the block pixels in order
      const wbf16x8 a = wb_tr_read8(ap + (16 * s4) * 32, ap + (16 * s4 + 4) * 32);
#pragma unroll
      for (int ky = 0; ky < KT; ++ky)
#pragma unroll
        for (int kx = 0; kx < KT; ++kx) {
          // Q image row of pixel k0 for tap (ky, kx): (S (k0 / TW) + ky) * QW + S (k0 % TW) + kx; the lane part is in b_lane
          const int r0 = TWL == 3 ? (2 * S * s4 + ky) * QW + kx : (4 * S * s4 + ky) * QW + kx;
          const int r1 = TWL == 3 ? r0 + 4 * S : r0 + S * QW;
          const wbf16x8 b = wb_tr_read8(bp + r0 * 32, bp + r1 * 32);
          acc[ky * KT + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[ky * KT + kx], 0, 0, 0);
        }
    }
    commit(lds + (buf ^ 1) * STAGE);
    __syncthreads();
    buf ^= 1;
  }

  const int n = nt * 64 + wn * 32 + li;
  if (n < Nch) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mt * 64 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < Mch) p.ws[(((size_t)split * NT + t) * Mch + m) * Nch + n] = acc[t][r];
      }
    }
  }
}

// dst[(m*Nst + n)*T + t] = sum_s ws[((s*T + t)*Mch + m)*Nch + n]   (fixed order -> deterministic)
// A workgroup owns one m and NL = 256 / KG consecutive n, i.e. a CONTIGUOUS run of NL * T floats of dst: the T taps of
// an (m, n) are summed together (T accumulators per thread, split range strided over KG thread groups), combined through
// LDS in fixed order and written as full lines — instead of one scattered 4-byte store per thread.
template <int KG>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dst,
                                                           int splits, int T, int Mch, int Nch, int Nst) {
  constexpr int NL = 256 / KG;
  constexpr int TMAX = 9;
  __shared__ float red[KG][TMAX][NL];
  const int nl = threadIdx.x % NL, kg = threadIdx.x / NL;
  const int nchunks = (Nch + NL - 1) / NL;
  const int m = blockIdx.x / nchunks;
  const int n0 = (blockIdx.x - m * nchunks) * NL;
  const int n = n0 + nl;
  float acc[TMAX];
#pragma unroll
  for (int t = 0; t < TMAX; ++t) acc[t] = 0.f;
  if (n < Nch) {
    const size_t tstride = (size_t)Mch * Nch;
    const float* base = ws + (size_t)m * Nch + n;
    for (int k = kg; k < splits; k += KG) {
      const float* pk = base + (size_t)k * T * tstride;
#pragma unroll
      for (int t = 0; t < TMAX; ++t)
        if (t < T) acc[t] += pk[(size_t)t * tstride];
    }
  }
#pragma unroll
  for (int t = 0; t < TMAX; ++t) red[kg][t][nl] = acc[t];
  __syncthreads();
  const int nrun = (Nst - n0 < NL ? Nst - n0 : NL) * T;          // floats of dst this workgroup owns
  float* out = dst + ((size_t)m * Nst + n0) * T;
  for (int e = threadIdx.x; e < nrun; e += 256) {
    const int en = e / T, et = e - en * T;
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < KG; ++j) v += red[j][et][en];
    out[e] = v;
  }
}

// The same reduction for MANY splits (few channel tiles: 64 .. 256 channels, 32 .. 512 splits of 75 MB in all): a workgroup
// owns one (m, tap) and 64 consecutive n — 16 lanes of 4 channels (16-byte loads, 256 contiguous bytes per split) x 16
// thread groups striding the split range, combined through LDS in fixed order.  The one-tap-per-workgroup grid is 9 x larger
// than the one above (576 workgroups for a 64 x 64 layer instead of 256 with 64-byte runs), which is what the read
// bandwidth needs; its output, 4 floats per lane at stride T, is 0.2 % of the traffic.
__global__ __launch_bounds__(256) void wgrad_reduce_many_kernel(const float* __restrict__ ws, float* __restrict__ dst,
                                                                int splits, int T, int Mch, int Nch, int Nst) {
  __shared__ float4 red[16][16];
  const int nl = threadIdx.x & 15, kg = threadIdx.x >> 4;
  const int nchunks = (Nch + 63) / 64;
  int idx = blockIdx.x;
  const int t = idx % T; idx /= T;
  const int nc = idx % nchunks, m = idx / nchunks;
  const int n = nc * 64 + nl * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (n < Nch) {
    const size_t sstride = (size_t)T * Mch * Nch;
    const float* base = ws + ((size_t)t * Mch + m) * Nch + n;
#pragma unroll 4
    for (int k = kg; k < splits; k += 16) {
      const float4 v = *reinterpret_cast<const float4*>(base + (size_t)k * sstride);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  red[kg][nl] = acc;
  __syncthreads();
  if (kg == 0 && n < Nch) {
    float4 v = red[0][nl];
#pragma unroll
    for (int j = 1; j < 16; ++j) { const float4 u = red[j][nl]; v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
    float* out = dst + ((size_t)m * Nst + n) * T + t;
    if (n + 0 < Nst) out[0 * T] = v.x;
    if (n + 1 < Nst) out[1 * T] = v.y;
    if (n + 2 < Nst) out[2 * T] = v.z;
    if (n + 3 < Nst) out[3 * T] = v.w;
  }
}

// Which kernel a launch goes to (shared by the split-K plan and the launcher): 2 = halo (pixel blocks), 1 = fast (linear
// 32-pixel steps), 0 = generic.
struct WgradSel { int kind, tw_log2, ptr, qtr; bool generic_act, per_sample; };

static WgradSel wgrad_select(const MsegWgrad& p) {
  WgradSel w;
  auto tr_of = [](const MsegSrc& s) {
    if (s.act == MSEG_ACT_NONE && !s.scale) return 0;
    return (s.act == MSEG_ACT_NONE || s.act == MSEG_ACT_RELU) ? 1 : 2;
  };
  w.ptr = tr_of(p.P);
  w.qtr = 0;
  w.generic_act = (p.P.act != MSEG_ACT_NONE && p.P.act != MSEG_ACT_RELU);
  w.per_sample = (p.P.scale && p.P.ss != 0);
  for (int i = 0; i < p.nq; ++i) {
    const int v = tr_of(p.Q[i]);
    if (v > w.qtr) w.qtr = v;
    if (p.Q[i].act != MSEG_ACT_NONE && p.Q[i].act != MSEG_ACT_RELU) w.generic_act = true;
    if (p.Q[i].scale && p.Q[i].ss != 0) w.per_sample = true;
  }
  // halo shape: 3x3 stride-1 conv, plain P (the output gradient), row length divisible by 4 -> TH x TW pixel blocks with
  // TW = largest power of two <= 32 dividing W; used when at most 20 % of the block rows hang over the image bottom
  w.tw_log2 = 5;
  bool halo = p.KH == 3 && p.KW == 3 && p.stride == 1 && p.Hq == p.Hp && p.Wq == p.Wp && (p.Wp % 4) == 0 && w.ptr == 0;
  if (halo) {
    while ((p.Wp & ((1 << w.tw_log2) - 1)) != 0) --w.tw_log2;
    const int TH = WG_PIX >> w.tw_log2;
    halo = (long long)p.Hp * 5 >= (long long)((p.Hp + TH - 1) / TH) * TH * 4;
  }
  // fast-path preconditions: 32-bit buffer offsets — the whole operand < 2 GiB for the linear-step kernel, ONE IMAGE of it
  // for the halo kernel (its descriptors are re-based per image: any batch size); concat boundary on a 64-channel tile;
  // per-sample (Group/InstanceNorm) tables need a K-step that cannot straddle two images (always true for pixel blocks)
  const long long Ptot64 = (long long)p.NB * p.Hp * p.Wp;
  const bool pix_ok = (Ptot64 + 2048LL * WG_PIX) < 0x7fffffffLL;
  bool whole_fits = Ptot64 * p.P.C * 4 < 0x80000000LL, image_fits = (long long)p.Hp * p.Wp * p.P.C * 4 < 0x80000000LL;
  for (int i = 0; i < p.nq; ++i) {
    if ((long long)p.NB * p.Hq * p.Wq * p.Q[i].C * 4 >= 0x80000000LL) whole_fits = false;
    if ((long long)p.Hq * p.Wq * p.Q[i].C * 4 >= 0x80000000LL) image_fits = false;
  }
  const bool concat_ok = !(p.nq > 1 && (p.Q[0].C % 64));
  if (p.precision == MSEG_PREC_BF16) {
    // bf16 halo kernel: 64-pixel blocks, 8 x 8 or (row length no multiple of 8) 16 x 4; used down to 50 % live block rows
    w.tw_log2 = (p.Wp % 8) == 0 ? 3 : 2;
    const int TH = 64 >> w.tw_log2;
    const bool ok16 = p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && p.Hq == p.Hp && p.Wq == p.Wp &&
                      (p.Wp % 4) == 0 && w.ptr == 0 && image_fits && pix_ok && concat_ok &&
                      (long long)p.Hp * 2 >= (long long)((p.Hp + TH - 1) / TH) * TH;     // >= 50 % of the block rows
    w.kind = ok16 ? 3 : -1;                            // -1: the launch is refused (the engine keeps such layers fp32)
    if (!ok16 && p.stride == 2 && p.Hq == 2 * p.Hp && p.Wq == 2 * p.Wp && (p.Wp % 4) == 0 && image_fits && pix_ok &&
        p.nq == 1) {
      // strided layers: 32-pixel blocks, 4 x 8 or 8 x 4
      const int TH2 = 32 >> w.tw_log2;
      const bool rows_ok = (long long)p.Hp * 2 >= (long long)((p.Hp + TH2 - 1) / TH2) * TH2;
      if (rows_ok && p.KH == 3 && p.KW == 3 && p.pad == 1 && w.ptr == 0) w.kind = 4;        // 3x3 stride-2 conv
      if (rows_ok && p.KH == 2 && p.KW == 2 && p.pad == 0 && w.qtr == 0) w.kind = 5;        // ConvTranspose 2x2
    }
    return w;
  }
  halo = halo && image_fits && pix_ok && concat_ok;
  const bool fast = halo || (whole_fits && pix_ok && concat_ok &&
                             (!w.per_sample || ((long long)p.Hp * p.Wp) % WG_PIX == 0));
  w.kind = fast ? (halo ? 2 : 1) : 0;
  if (halo) {
    // all taps per workgroup on 4 x 8 / 8 x 4 pixel blocks, when at most 20 % of the block rows hang over the image bottom
    const int tw9 = (p.Wp % 8) == 0 ? 3 : 2, TH9 = WG_PIX >> tw9;
    if ((long long)p.Hp * 5 >= (long long)((p.Hp + TH9 - 1) / TH9) * TH9 * 4) { w.kind = 6; w.tw_log2 = tw9; }
  }
  return w;
}

static int wgrad_plan(const MsegWgrad& p, int& splits, int& steps_per_split) {
  const long long Ptot = (long long)p.NB * p.Hp * p.Wp;
  if (Ptot <= 0) return MSEG_EINVAL;
  long long steps_total = (Ptot + WG_PIX - 1) / WG_PIX;
  const WgradSel sel = wgrad_select(p);
  if (sel.kind < 0) return MSEG_EINVAL;
  if (sel.kind == 2 || sel.kind == 6) {
    const int TH = WG_PIX >> sel.tw_log2;
    steps_total = (long long)p.NB * ((p.Hp + TH - 1) / TH) * (p.Wp >> sel.tw_log2);   // pixel blocks (wgrad_halo*_kernel)
  }
  if (sel.kind >= 3) {
    const int TH = (sel.kind == 3 ? 64 : 32) >> sel.tw_log2;
    steps_total = (long long)p.NB * ((p.Hp + TH - 1) / TH) * (p.Wp >> sel.tw_log2);   // pixel blocks, all taps per workgroup
  }
  const int tiles = ((p.P.C + 63) / 64) * ((p.Nch + 63) / 64);
  const int per_split_wgs = sel.kind >= 3 ? tiles : tiles * p.KH;      // kinds >= 3: all taps in one workgroup
  // 768 workgroups = one full round of 3 resident workgroups on each of the 256 CUs (bf16 kernel: 2 resident -> 512)
  const int round_wgs = sel.kind >= 3 ? 512 : 768;
  long long s = p.splits > 0 ? p.splits : (round_wgs + per_split_wgs - 1) / per_split_wgs;
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
  if (p.precision != MSEG_PREC_F32 && p.precision != MSEG_PREC_BF16) return MSEG_EINVAL;
  // tensor storage: P and all Q of one type; bf16 tensors (8 channels per staging load) only with the bf16 kernels
  if (p.P.dtype != MSEG_ST_F32 && p.P.dtype != MSEG_ST_BF16) return MSEG_EINVAL;
  for (int i = 0; i < p.nq; ++i)
    if (p.Q[i].dtype != p.P.dtype) return MSEG_EINVAL;
  if (p.P.dtype == MSEG_ST_BF16) {
    if (p.precision != MSEG_PREC_BF16 || (p.P.C & 7)) return MSEG_EINVAL;
    for (int i = 0; i < p.nq; ++i)
      if (p.Q[i].C & 7) return MSEG_EINVAL;
  }
  return MSEG_OK;
}

extern "C" size_t mseg_wgrad_workspace_bytes(const MsegWgrad* pp) {
  if (!pp || wgrad_check(*pp)) return 0;
  int splits, sps;
  if (wgrad_plan(*pp, splits, sps)) return 0;
  return (size_t)splits * pp->KH * pp->KW * pp->P.C * pp->Nch * sizeof(float);
}

static int wgrad_dispatch(const MsegWgrad* pp, void* stream) {
  if (!pp) return MSEG_EINVAL;
  const MsegWgrad& p = *pp;
  if (wgrad_check(p) || !p.ws || !p.dst) return MSEG_EINVAL;
  int splits, sps;
  if (wgrad_plan(p, splits, sps)) return MSEG_EINVAL;
  const int T = p.KH * p.KW;
  const int tiles = ((p.P.C + 63) / 64) * ((p.Nch + 63) / 64);
  hipStream_t st = (hipStream_t)stream;
  mseg_dispatch_note(p.precision, (size_t)splits * T * p.P.C * p.Nch * sizeof(float));
  if (p.phase != 2) {
    const WgradSel sel = wgrad_select(p);
    const bool generic = sel.generic_act, per_sample = sel.per_sample;
    const int ptr = sel.ptr, qtr = sel.qtr, tw_log2 = sel.tw_log2;
    const bool s16 = p.P.dtype == MSEG_ST_BF16;         // bf16 tensor storage (wgrad_check: P and Q agree, bf16 kernels only)
    const dim3 grid((unsigned)tiles * (unsigned)splits * (unsigned)(sel.kind >= 3 ? 1 : p.KH)), block(256);
    if (sel.kind != 0) {
      static bool ident_ready[64] = {false};
      int devid = 0;
      const bool dry = mseg_dispatch_dry() != 0;      // a query: nothing is launched, no device is needed
      if (!dry && (hipGetDevice(&devid) != hipSuccess || devid < 0 || devid >= 64)) return MSEG_ELAUNCH;
      if (!dry && !ident_ready[devid]) {
        MSEG_KL_AUX(wgrad_init_ident_kernel, dim3(1), dim3(64), 0, st);
        MSEG_LAUNCH_CHECK();
        ident_ready[devid] = true;
      }
      if (sel.kind == 4) {                             // 3x3 stride-2 convolution (plain P)
#define MSEG_WB4(T_, Q_)                                                                                          \
  do {                                                                                                            \
    if (s16) MSEG_KL((wgrad_halo_bf16_kernel<T_, Q_, 2, 3, 0, true>), grid, block, 0, st, p, splits, sps); \
    else MSEG_KL((wgrad_halo_bf16_kernel<T_, Q_, 2, 3, 0, false>), grid, block, 0, st, p, splits, sps);    \
  } while (0)
#define MSEG_WB4_Q(T_) do { if (qtr == 0) MSEG_WB4(T_, 0); else if (qtr == 1) MSEG_WB4(T_, 1); else MSEG_WB4(T_, 2); } while (0)
        if (tw_log2 == 3) MSEG_WB4_Q(3); else MSEG_WB4_Q(2);
#undef MSEG_WB4_Q
#undef MSEG_WB4
        MSEG_LAUNCH_CHECK();
      } else if (sel.kind == 5) {                      // ConvTranspose2d 2x2 stride 2 (plain Q)
#define MSEG_WB5(T_, P_)                                                                                          \
  do {                                                                                                            \
    if (s16) MSEG_KL((wgrad_halo_bf16_kernel<T_, 0, 2, 2, P_, true>), grid, block, 0, st, p, splits, sps); \
    else MSEG_KL((wgrad_halo_bf16_kernel<T_, 0, 2, 2, P_, false>), grid, block, 0, st, p, splits, sps);    \
  } while (0)
#define MSEG_WB5_P(T_) do { if (ptr == 0) MSEG_WB5(T_, 0); else if (ptr == 1) MSEG_WB5(T_, 1); else MSEG_WB5(T_, 2); } while (0)
        if (tw_log2 == 3) MSEG_WB5_P(3); else MSEG_WB5_P(2);
#undef MSEG_WB5_P
#undef MSEG_WB5
        MSEG_LAUNCH_CHECK();
      } else if (sel.kind == 3) {
#define MSEG_WB(T_, Q_)                                                                                           \
  do {                                                                                                            \
    if (s16) MSEG_KL((wgrad_halo_bf16_kernel<T_, Q_, 1, 3, 0, true>), grid, block, 0, st, p, splits, sps); \
    else MSEG_KL((wgrad_halo_bf16_kernel<T_, Q_, 1, 3, 0, false>), grid, block, 0, st, p, splits, sps);    \
  } while (0)
#define MSEG_WB_Q(T_) do { if (qtr == 0) MSEG_WB(T_, 0); else if (qtr == 1) MSEG_WB(T_, 1); else MSEG_WB(T_, 2); } while (0)
        if (tw_log2 == 3) MSEG_WB_Q(3); else MSEG_WB_Q(2);
#undef MSEG_WB_Q
#undef MSEG_WB
        MSEG_LAUNCH_CHECK();
      } else if (sel.kind == 6) {
#define MSEG_W9(T_, Q_) MSEG_KL((wgrad_halo9_kernel<T_, Q_>), grid, block, 0, st, p, splits, sps)
#define MSEG_W9_Q(T_) do { if (qtr == 0) MSEG_W9(T_, 0); else if (qtr == 1) MSEG_W9(T_, 1); else MSEG_W9(T_, 2); } while (0)
        if (tw_log2 == 3) MSEG_W9_Q(3); else MSEG_W9_Q(2);
#undef MSEG_W9_Q
#undef MSEG_W9
        MSEG_LAUNCH_CHECK();
      } else if (sel.kind == 2) {
#define MSEG_WH(T_, Q_) MSEG_KL((wgrad_halo_kernel<T_, Q_>), grid, block, 0, st, p, splits, sps)
#define MSEG_WH_Q(T_) do { if (qtr == 0) MSEG_WH(T_, 0); else if (qtr == 1) MSEG_WH(T_, 1); else MSEG_WH(T_, 2); } while (0)
        if (tw_log2 == 5) MSEG_WH_Q(5); else if (tw_log2 == 4) MSEG_WH_Q(4); else if (tw_log2 == 3) MSEG_WH_Q(3); else MSEG_WH_Q(2);
#undef MSEG_WH_Q
#undef MSEG_WH
        MSEG_LAUNCH_CHECK();
      } else {
#define MSEG_WF(KW_, P_, Q_) MSEG_KL((wgrad_fast_kernel<KW_, P_, Q_>), grid, block, 0, st, p, splits, sps)
#define MSEG_WF_Q(KW_, P_) do { if (qtr == 0) MSEG_WF(KW_, P_, 0); else if (qtr == 1) MSEG_WF(KW_, P_, 1); else MSEG_WF(KW_, P_, 2); } while (0)
#define MSEG_WF_P(KW_) do { if (ptr == 0) MSEG_WF_Q(KW_, 0); else if (ptr == 1) MSEG_WF_Q(KW_, 1); else MSEG_WF_Q(KW_, 2); } while (0)
      if (p.KW == 3) MSEG_WF_P(3); else MSEG_WF_P(2);
#undef MSEG_WF_P
#undef MSEG_WF_Q
#undef MSEG_WF
      }
    } else {
#define MSEG_WGRAD_LAUNCH(KW_, GA_, PS_) \
  MSEG_KL((wgrad_kernel<KW_, GA_, PS_>), grid, block, 0, st, p, splits, sps)
    if (p.KW == 3) {
      if (generic) { if (per_sample) MSEG_WGRAD_LAUNCH(3, true, true); else MSEG_WGRAD_LAUNCH(3, true, false); }
      else         { if (per_sample) MSEG_WGRAD_LAUNCH(3, false, true); else MSEG_WGRAD_LAUNCH(3, false, false); }
    } else {
      if (generic) { if (per_sample) MSEG_WGRAD_LAUNCH(2, true, true); else MSEG_WGRAD_LAUNCH(2, true, false); }
      else         { if (per_sample) MSEG_WGRAD_LAUNCH(2, false, true); else MSEG_WGRAD_LAUNCH(2, false, false); }
    }
#undef MSEG_WGRAD_LAUNCH
    }
    MSEG_LAUNCH_CHECK();
  }
  if (p.phase == 1) return MSEG_OK;
  // many splits (few channel tiles: the wide shallow levels) -> 16 thread groups share the split range
  if (splits >= 32 && (p.Nch & 3) == 0 && (((uintptr_t)p.ws) & 15) == 0 &&
      (long long)p.P.C * ((p.Nch + 63) / 64) * T <= 0x7fffffffLL) {
    const unsigned blocks = (unsigned)p.P.C * (unsigned)((p.Nch + 63) / 64) * (unsigned)T;
    MSEG_KL_AUX(wgrad_reduce_many_kernel, dim3(blocks), dim3(256), 0, st, (const float*)p.ws, p.dst, splits, T,
                       p.P.C, p.Nch, p.Nch_store);
  } else if (splits >= 32) {
    const unsigned blocks = (unsigned)p.P.C * (unsigned)((p.Nch + 15) / 16);
    MSEG_KL_AUX((wgrad_reduce_kernel<16>), dim3(blocks), dim3(256), 0, st, (const float*)p.ws, p.dst, splits, T,
                       p.P.C, p.Nch, p.Nch_store);
  } else {
    const unsigned blocks = (unsigned)p.P.C * (unsigned)((p.Nch + 63) / 64);
    MSEG_KL_AUX((wgrad_reduce_kernel<4>), dim3(blocks), dim3(256), 0, st, (const float*)p.ws, p.dst, splits, T,
                       p.P.C, p.Nch, p.Nch_store);
  }
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

extern "C" int mseg_wgrad(const MsegWgrad* pp, void* stream) {
  mseg_dispatch_begin(0);
  const int rc = wgrad_dispatch(pp, stream);
  mseg_dispatch_end(nullptr);
  return rc;
}

// the same dispatch with the launches switched off (common.h: MSEG_KL)
extern "C" int mseg_wgrad_query(const MsegWgrad* pp, MsegKernelInfo* info) {
  if (!info) return MSEG_EINVAL;
  mseg_dispatch_begin(1);
  const int rc = wgrad_dispatch(pp, nullptr);
  mseg_dispatch_end(info);
  return rc;
}
