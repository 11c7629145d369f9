// igemm.hip — gather implicit-GEMM convolution on the fp32 matrix cores of gfx950
// (v_mfma_f32_32x32x2_f32: exact f32, 64 FLOP/clk/SIMD — MI355X has no xf32/TF32 path).
//
// One kernel family covers (reference call sites in parentheses):
//   * Conv2d 3x3 s1 p1 forward            (src/utils/unets.py:112,137)   mode CONV
//   * Conv2d 3x3 s2 p1 forward (ConvPool) (src/utils/unets.py:192)       mode CONV, stride 2
//   * data gradient of both               (autograd; train.py:488)       mode TCONV (+ parity M-order for s2)
//   * ConvTranspose2d 2x2 s2 forward      (src/utils/unets.py:244)       1 tap, N = 4*Cout, SCATTER2X2 epilogue
//   * its data gradient                                                  mode CONV, 2x2 taps, stride 2, pad 0
//   * torch.cat([up, skip], 1)            (unets.py:373,492,502)         two sources, never materialised
//   * activation + Batch/Group/InstanceNorm of the *producer* layer      applied while staging A (norm-on-load)
//
// Tiling: a workgroup (256 threads = 4 waves) owns BM output pixels x BN output channels; every K-step stages a
// [BM][32] slab of gathered source pixels (one tap, 32 input channels) and a [BN][32] slab of weights in LDS
// (row stride 36 floats -> conflict-free ds_read_b128), double buffered (one barrier per K-step); each wave accumulates
// a 64x64 (BN=128) or 32x64 (BN=64) block as 32x32 MFMA tiles.  The raw global loads of step s+1 are issued before the
// MFMAs of step s and consumed (activation, scale/shift, ds_write) after them.
#include "common.h"

#define KC 32
#define LDS_STRIDE 36

template <int BM, int BN>
struct IgemmCfg {
  static constexpr int WM = (BN >= 128) ? 2 : 4;  // waves along M
  static constexpr int WN = 4 / WM;               // waves along N
  static constexpr int TM = BM / WM;              // per-wave rows   (multiple of 32)
  static constexpr int TN = BN / WN;              // per-wave cols
  static constexpr int MB = TM / 32;
  static constexpr int NB = TN / 32;
  static constexpr int AROWS = BM / 32;           // A rows staged per thread
  static constexpr int BROWS = BN / 32;
};

struct RowInfo {
  int n;    // image index, -1 = row beyond M
  int oy, ox;
};

__device__ __forceinline__ RowInfo decode_row(const MsegIgemm& p, int m, int M) {
  RowInfo r;
  if (m >= M) { r.n = -1; r.oy = 0; r.ox = 0; return r; }
  if (p.morder == MSEG_MORDER_PARITY) {
    const int Hh = p.Ho >> 1, Wh = p.Wo >> 1;
    const int per = p.NB * Hh * Wh;
    const int cls = m / per;
    int rem = m - cls * per;
    r.n = rem / (Hh * Wh);
    rem -= r.n * (Hh * Wh);
    const int y2 = rem / Wh;
    r.oy = 2 * y2 + (cls >> 1);
    r.ox = 2 * (rem - y2 * Wh) + (cls & 1);
  } else {
    r.n = m / (p.Ho * p.Wo);
    int rem = m - r.n * (p.Ho * p.Wo);
    r.oy = rem / p.Wo;
    r.ox = rem - r.oy * p.Wo;
  }
  return r;
}

// source pixel of (row, tap); returns false if the tap falls outside / on a dead phase
__device__ __forceinline__ bool tap_coord(const MsegIgemm& p, const RowInfo& r, int ky, int kx, int& iy, int& ix) {
  if (r.n < 0) return false;
  if (p.mode == MSEG_MODE_CONV) {
    iy = r.oy * p.stride + ky - p.pad;
    ix = r.ox * p.stride + kx - p.pad;
  } else {
    const int ty = r.oy + p.pad - ky, tx = r.ox + p.pad - kx;
    if (ty < 0 || tx < 0) return false;
    if (p.stride == 2) {
      if ((ty | tx) & 1) return false;
      iy = ty >> 1; ix = tx >> 1;
    } else {
      iy = ty; ix = tx;
    }
  }
  return iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
}

template <int BM, int BN, bool PER_SAMPLE, bool GENERIC_ACT>
__global__ __launch_bounds__(256) void igemm_kernel(const MsegIgemm p) {
  using Cfg = IgemmCfg<BM, BN>;
  constexpr int STAGE = (BM + BN) * LDS_STRIDE;
  __shared__ __attribute__((aligned(16))) float lds[2 * STAGE + 4];
  unsigned* tapmask_s = reinterpret_cast<unsigned*>(lds + 2 * STAGE);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;

  const int M = p.NB * p.Ho * p.Wo;
  const int ntiles_n = (p.Ngemm + BN - 1) / BN;
  // consecutive workgroups share the A slab (same M tile, different N tile) -> L2 reuse
  const int tile_m = blockIdx.x / ntiles_n;
  const int tile_n = blockIdx.x - tile_m * ntiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int T = p.KH * p.KW;
  const int srow = tid >> 3;  // staging row within a 32-row pass
  const int scol = tid & 7;   // float4 column (4 channels)

  RowInfo rows[Cfg::AROWS];
#pragma unroll
  for (int i = 0; i < Cfg::AROWS; ++i) rows[i] = decode_row(p, m0 + srow + 32 * i, M);

  // which taps are live for at least one row of this tile: only a stride-2 transposed conv has dead taps (tile-uniform
  // under the parity M-order); everywhere else every tap is live and per-row masks handle the borders
  unsigned tapmask = (T >= 32) ? 0xffffffffu : ((1u << T) - 1u);
  if (p.mode == MSEG_MODE_TCONV && p.stride == 2) {
    if (tid == 0) *tapmask_s = 0u;
    __syncthreads();
    unsigned mine = 0u;
    for (int t = 0; t < T; ++t) {
      const int ky = t / p.KW, kx = t - ky * p.KW;
      bool any = false;
#pragma unroll
      for (int i = 0; i < Cfg::AROWS; ++i) {
        int iy, ix;
        any |= tap_coord(p, rows[i], ky, kx, iy, ix);
      }
      if (any) mine |= (1u << t);
    }
    if (mine) atomicOr(tapmask_s, mine);
    __syncthreads();
    tapmask = *tapmask_s;
    __syncthreads();
  }

  f32x16 acc[Cfg::MB][Cfg::NB];
#pragma unroll
  for (int a = 0; a < Cfg::MB; ++a)
#pragma unroll
    for (int b = 0; b < Cfg::NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int nchunks = (p.Cin + KC - 1) / KC;
  const int C0 = p.src[0].C;

  // ---- staging registers: raw loads are issued before the MFMAs of the current step and only *used* (activation,
  // scale/shift, LDS store) after them, so their latency hides behind the matrix work -------------------------------
  constexpr int NSC = PER_SAMPLE ? Cfg::AROWS : 1;
  float4 ra[Cfg::AROWS], rb[Cfg::BROWS], rsc[NSC], rsh[NSC];
  unsigned amask = 0u;
  int ract = 0;

  auto issue = [&](int chunk, int t, int ky, int kx) {
    const int c = chunk * KC + scol * 4;
    const bool cvalid = c < p.Cin;
    const bool s1 = (p.nsrc > 1) && (c >= C0);
    const float* sptr = s1 ? p.src[1].ptr : p.src[0].ptr;
    const float* sscale = s1 ? p.src[1].scale : p.src[0].scale;
    const float* sshift = s1 ? p.src[1].shift : p.src[0].shift;
    const int sC = s1 ? p.src[1].C : p.src[0].C;
    const int sss = s1 ? p.src[1].ss : p.src[0].ss;
    ract = s1 ? p.src[1].act : p.src[0].act;
    const int cl = cvalid ? (s1 ? c - C0 : c) : 0;
    amask = 0u;
#pragma unroll
    for (int i = 0; i < Cfg::AROWS; ++i) {
      int iy = 0, ix = 0;
      const bool ok = cvalid && tap_coord(p, rows[i], ky, kx, iy, ix);
      const size_t off = ok ? (((size_t)rows[i].n * p.Hi + iy) * p.Wi + ix) * sC + cl : 0;
      ra[i] = *reinterpret_cast<const float4*>(sptr + off);
      amask |= ok ? (1u << i) : 0u;
    }
    if (sscale) {
      if (PER_SAMPLE) {
#pragma unroll
        for (int i = 0; i < NSC; ++i) {
          const int n = rows[i].n < 0 ? 0 : rows[i].n;
          rsc[i] = *reinterpret_cast<const float4*>(sscale + (size_t)n * sss + cl);
          rsh[i] = *reinterpret_cast<const float4*>(sshift + (size_t)n * sss + cl);
        }
      } else {
        rsc[0] = *reinterpret_cast<const float4*>(sscale + cl);
        rsh[0] = *reinterpret_cast<const float4*>(sshift + cl);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NSC; ++i) {
        rsc[i] = make_float4(1.f, 1.f, 1.f, 1.f);
        rsh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    const float* wt = p.w + ((size_t)t * p.Ngemm) * p.Kpad;
#pragma unroll
    for (int i = 0; i < Cfg::BROWS; ++i) {
      const int n = n0 + srow + 32 * i;
      const bool ok = (n < p.Ngemm) && (c < p.Kpad);
      const float4 v = *reinterpret_cast<const float4*>(wt + (ok ? (size_t)n * p.Kpad + c : 0));
      rb[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };

  auto commit = [&](float* As, float* Bs) {
    const float lo = (ract == MSEG_ACT_RELU) ? 0.f : -3.402823466e38f;
#pragma unroll
    for (int i = 0; i < Cfg::AROWS; ++i) {
      float4 v = ra[i];
      if (GENERIC_ACT) {
        v = act_fwd4(v, ract);
      } else {
        v.x = fmaxf(v.x, lo); v.y = fmaxf(v.y, lo); v.z = fmaxf(v.z, lo); v.w = fmaxf(v.w, lo);
      }
      const float4 sc = rsc[PER_SAMPLE ? i : 0], sh = rsh[PER_SAMPLE ? i : 0];
      const bool ok = (amask >> i) & 1u;
      v.x = ok ? v.x * sc.x + sh.x : 0.f;
      v.y = ok ? v.y * sc.y + sh.y : 0.f;
      v.z = ok ? v.z * sc.z + sh.z : 0.f;
      v.w = ok ? v.w * sc.w + sh.w : 0.f;
      *reinterpret_cast<float4*>(As + (srow + 32 * i) * LDS_STRIDE + scol * 4) = v;
    }
#pragma unroll
    for (int i = 0; i < Cfg::BROWS; ++i)
      *reinterpret_cast<float4*>(Bs + (srow + 32 * i) * LDS_STRIDE + scol * 4) = rb[i];
  };

  // K-step iterator (chunk outer, tap inner), by value so that it stays in scalar registers
  struct Pos { int chunk, t, ky, kx; };
  const int KWm1 = p.KW - 1;
  auto advance = [&](Pos q, bool& ok) -> Pos {
    ok = true;
    for (;;) {
      const int wrap = (q.kx == KWm1) ? 1 : 0;
      q.t += 1;
      q.kx = wrap ? 0 : q.kx + 1;
      q.ky += wrap;
      if (q.t >= T) {
        q.t = 0; q.ky = 0; q.kx = 0; q.chunk += 1;
        if (q.chunk >= nchunks) { ok = false; return q; }
      }
      if (tapmask & (1u << q.t)) return q;
    }
  };

  Pos pos = {0, -1, 0, -1};
  bool have = false;
  if (tapmask != 0u) pos = advance(pos, have);
  int cur = 0;
  if (have) {
    issue(pos.chunk, pos.t, pos.ky, pos.kx);
    commit(lds, lds + BM * LDS_STRIDE);
  }
  __syncthreads();

  const int li = lane & 31, lh = lane >> 5;
  while (have) {
    // prefetch the next K-step (on the last step: a harmless re-read of the current one, keeps the body branch-free)
    bool have_next;
    const Pos nxt = advance(pos, have_next);
    if (have_next) pos = nxt;
    issue(pos.chunk, pos.t, pos.ky, pos.kx);

    const float* As = lds + cur * STAGE;
    const float* Bs = As + BM * LDS_STRIDE;
#pragma unroll
    for (int kk = 0; kk < KC / 8; ++kk) {
      float4 af[Cfg::MB], bf[Cfg::NB];
#pragma unroll
      for (int a = 0; a < Cfg::MB; ++a)
        af[a] = *reinterpret_cast<const float4*>(As + (wm * Cfg::TM + a * 32 + li) * LDS_STRIDE + kk * 8 + lh * 4);
#pragma unroll
      for (int b = 0; b < Cfg::NB; ++b)
        bf[b] = *reinterpret_cast<const float4*>(Bs + (wn * Cfg::TN + b * 32 + li) * LDS_STRIDE + kk * 8 + lh * 4);
#pragma unroll
      for (int a = 0; a < Cfg::MB; ++a)
#pragma unroll
        for (int b = 0; b < Cfg::NB; ++b) {
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].x, bf[b].x, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].y, bf[b].y, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].z, bf[b].z, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].w, bf[b].w, acc[a][b], 0, 0, 0);
        }
    }
    float* An = lds + (cur ^ 1) * STAGE;
    commit(An, An + BM * LDS_STRIDE);
    __syncthreads();
    cur ^= 1;
    have = have_next;
  }

  // ---- epilogue: D[row = (r&3) + 8*(r>>2) + 4*(lane>>5)][col = lane&31] ------------------------------------
  // The epilogue-only fields are re-read from kernarg memory here (opaque pointer) so that they do not occupy
  // SGPRs during the K-loop (the whole descriptor live = SGPR spills reloaded every iteration).
  const MsegIgemm* pe = (const MsegIgemm*)__builtin_amdgcn_kernarg_segment_ptr();  // kernel argument 0
  asm volatile("" : "+s"(pe));
  const float* e_bias = pe->bias;
  float* e_dst0 = pe->dst0;
  float* e_dst1 = pe->dst1;
  const int e_epi = pe->epi, e_split = pe->split, e_ld0 = pe->ld0, e_ld1 = pe->ld1, e_acc0 = pe->acc0,
            e_acc1 = pe->acc1, e_Cq = pe->Cq, e_morder = pe->morder, e_Ho = pe->Ho, e_Wo = pe->Wo, e_Ngemm = pe->Ngemm;
#pragma unroll
  for (int b = 0; b < Cfg::NB; ++b) {
    const int n = n0 + wn * Cfg::TN + b * 32 + li;
    const bool nvalid = n < e_Ngemm;
    float bias = 0.f;
    float* dst = e_dst0;
    int ld = e_ld0, noff = n, accf = e_acc0;
    int sa = 0, sb = 0;
    if (e_epi == MSEG_EPI_SCATTER2X2) {
      const int ab = nvalid ? n / e_Cq : 0;
      const int co = n - ab * e_Cq;
      sa = ab >> 1; sb = ab & 1;
      noff = co;
      if (e_bias && nvalid) bias = e_bias[co];
    } else {
      if (n >= e_split) { dst = e_dst1; ld = e_ld1; noff = n - e_split; accf = e_acc1; }
      if (e_bias && nvalid) bias = e_bias[n];
    }
#pragma unroll
    for (int a = 0; a < Cfg::MB; ++a) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int m = m0 + wm * Cfg::TM + a * 32 + row;
        if (!nvalid || m >= M) continue;
        size_t off;
        if (e_epi == MSEG_EPI_SCATTER2X2) {
          const int img = m / (e_Ho * e_Wo);
          const int rem = m - img * (e_Ho * e_Wo);
          const int oy = rem / e_Wo, ox = rem - oy * e_Wo;
          off = (((size_t)img * (2 * e_Ho) + 2 * oy + sa) * (2 * e_Wo) + 2 * ox + sb) * e_Cq + noff;
        } else if (e_morder == MSEG_MORDER_PARITY) {
          const RowInfo ri = decode_row(*pe, m, M);
          off = (((size_t)ri.n * e_Ho + ri.oy) * e_Wo + ri.ox) * ld + noff;
        } else {
          off = (size_t)m * ld + noff;
        }
        float v = acc[a][b][r] + bias;
        if (accf) v += dst[off];
        dst[off] = v;
      }
    }
  }
}

static int check_src(const MsegSrc& s) {
  if (!s.ptr || s.C <= 0 || (s.C & 3)) return MSEG_EINVAL;
  if ((s.scale == nullptr) != (s.shift == nullptr)) return MSEG_EINVAL;
  if (s.scale && (s.ss & 3)) return MSEG_EINVAL;
  return MSEG_OK;
}

extern "C" int mseg_igemm(const MsegIgemm* pp, void* stream) {
  if (!pp) return MSEG_EINVAL;
  const MsegIgemm& p = *pp;
  if (p.nsrc < 1 || p.nsrc > 2) return MSEG_EINVAL;
  int csum = 0;
  for (int i = 0; i < p.nsrc; ++i) {
    if (check_src(p.src[i])) return MSEG_EINVAL;
    csum += p.src[i].C;
  }
  if (csum != p.Cin) return MSEG_EINVAL;
  if (!p.w || !p.dst0 || p.Kpad < p.Cin || (p.Kpad & 3)) return MSEG_EINVAL;
  if (p.NB <= 0 || p.Hi <= 0 || p.Wi <= 0 || p.Ho <= 0 || p.Wo <= 0) return MSEG_EINVAL;
  if (p.KH <= 0 || p.KW <= 0 || p.KH * p.KW > 16) return MSEG_EINVAL;
  if (p.stride != 1 && p.stride != 2) return MSEG_EINVAL;
  if (p.mode != MSEG_MODE_CONV && p.mode != MSEG_MODE_TCONV) return MSEG_EINVAL;
  if (p.morder == MSEG_MORDER_PARITY && ((p.Ho | p.Wo) & 1)) return MSEG_EINVAL;
  if (p.Ngemm <= 0) return MSEG_EINVAL;
  if (p.epi == MSEG_EPI_SCATTER2X2) {
    if (p.Cq <= 0 || p.Ngemm != 4 * p.Cq || p.morder != MSEG_MORDER_LINEAR) return MSEG_EINVAL;
  } else if (p.epi == MSEG_EPI_PLAIN) {
    if (p.split < p.Ngemm && !p.dst1) return MSEG_EINVAL;
  } else {
    return MSEG_EINVAL;
  }
  const long long M = (long long)p.NB * p.Ho * p.Wo;
  if (M <= 0 || M > 0x7fffffffLL) return MSEG_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  bool per_sample = false, generic = false;
  for (int i = 0; i < p.nsrc; ++i) {
    if (p.src[i].scale && p.src[i].ss != 0) per_sample = true;
    if (p.src[i].act != MSEG_ACT_NONE && p.src[i].act != MSEG_ACT_RELU) generic = true;
  }
  const bool wide = p.Ngemm > 64;
  const int BMv = 128, BNv = wide ? 128 : 64;
  const long long tiles = ((M + BMv - 1) / BMv) * ((p.Ngemm + BNv - 1) / BNv);
  if (tiles > 0x7fffffffLL) return MSEG_EINVAL;
  const dim3 grid((unsigned)tiles), block(256);
#define MSEG_IGEMM_LAUNCH(BM_, BN_, PS_, GA_) \
  hipLaunchKernelGGL((igemm_kernel<BM_, BN_, PS_, GA_>), grid, block, 0, st, p)
  if (wide) {
    if (per_sample) { if (generic) MSEG_IGEMM_LAUNCH(128, 128, true, true); else MSEG_IGEMM_LAUNCH(128, 128, true, false); }
    else            { if (generic) MSEG_IGEMM_LAUNCH(128, 128, false, true); else MSEG_IGEMM_LAUNCH(128, 128, false, false); }
  } else {
    if (per_sample) { if (generic) MSEG_IGEMM_LAUNCH(128, 64, true, true); else MSEG_IGEMM_LAUNCH(128, 64, true, false); }
    else            { if (generic) MSEG_IGEMM_LAUNCH(128, 64, false, true); else MSEG_IGEMM_LAUNCH(128, 64, false, false); }
  }
#undef MSEG_IGEMM_LAUNCH
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

// ---- weight repack ------------------------------------------------------------------------------------------
__global__ void pack_weight_kernel(const float* __restrict__ src, float* __restrict__ dst, int T, int R, int C,
                                   int Cpad, int st, int sr, int sc) {
  const size_t total = (size_t)T * R * Cpad;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cpad);
    const size_t tr = i / Cpad;
    const int r = (int)(tr % R);
    const int t = (int)(tr / R);
    dst[i] = c < C ? src[(size_t)t * st + (size_t)r * sr + (size_t)c * sc] : 0.f;
  }
}

extern "C" int mseg_pack_weight(const float* src, float* dst, int T, int R, int C, int Cpad, int st, int sr, int sc,
                                void* stream) {
  if (!src || !dst || T <= 0 || R <= 0 || C <= 0 || Cpad < C || (Cpad & 3)) return MSEG_EINVAL;
  const size_t total = (size_t)T * R * Cpad;
  unsigned blocks = (unsigned)((total + 255) / 256);
  if (blocks > 4096u) blocks = 4096u;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, dst, T, R, C, Cpad, st,
                     sr, sc);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}
