// igemm_common.h — pieces shared by the implicit-GEMM translation units (igemm.hip, igemm_p8.hip): tile configuration,
// GEMM-row decoding, tap geometry and the common epilogue.
#pragma once
#include "common.h"

#define KC 32
#define LDS_STRIDE 36
#define MSEG_MAX_CH 8192

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

// Tap geometry folded into three kernel-uniform integers so that the per-row code of the K-loop is straight-line:
//   t = o * sm + dir * (k - pad);  live iff (t & sh) == 0;  source coordinate = t >> sh
//   CONV : sm = stride, dir = +1, sh = 0          TCONV (transposed conv / dgrad): sm = 1, dir = -1, sh = stride >> 1
struct TapGeom { int sm, dir, sh, pad, Hi, Wi; };

__device__ __forceinline__ TapGeom make_geom(const MsegIgemm& p) {
  TapGeom g;
  const bool conv = p.mode == MSEG_MODE_CONV;
  g.sm = conv ? p.stride : 1;
  g.dir = conv ? 1 : -1;
  g.sh = conv ? 0 : (p.stride >> 1);
  g.pad = p.pad; g.Hi = p.Hi; g.Wi = p.Wi;
  return g;
}

// source pixel of (row, tap); false if the tap falls outside the source / on a dead phase (no early exits)
__device__ __forceinline__ bool tap_coord(const TapGeom& g, const RowInfo& r, int ky, int kx, int& iy, int& ix) {
  const int ty = r.oy * g.sm + g.dir * (ky - g.pad);
  const int tx = r.ox * g.sm + g.dir * (kx - g.pad);
  iy = ty >> g.sh; ix = tx >> g.sh;                     // arithmetic shift: negative stays negative -> rejected below
  return (r.n >= 0) & (((ty | tx) & g.sh) == 0) & (iy >= 0) & (iy < g.Hi) & (ix >= 0) & (ix < g.Wi);
}

// Gather kernels: first source pixel a tile of BM consecutive GEMM rows can touch (64-bit, wave-uniform).  The buffer
// descriptors of a tile are based there, so the 32-bit offsets of its loads only span the few image rows the tile reads —
// operands of any size (a 4096 x 4096 x 128 level-1 tensor is 8.6 GB) stay on the fast kernels.  Source pixels grow
// monotonically with the GEMM row inside a tile (linear M-order, and parity M-order within one parity class), so the
// first row's tap (0, 0), minus the reach of the mirrored taps of a transposed convolution, is a lower bound.
__device__ __forceinline__ long long tile_base_pixel(const MsegIgemm& p, const TapGeom& g, int m0, int M) {
  const RowInfo r = decode_row(p, m0, M);
  const int iy0 = (r.oy * g.sm - g.dir * g.pad) >> g.sh, ix0 = (r.ox * g.sm - g.dir * g.pad) >> g.sh;
  long long px = ((long long)r.n * p.Hi + iy0) * p.Wi + ix0;
  if (g.dir < 0) px -= (long long)((p.KH - 1) >> g.sh) * p.Wi + ((p.KW - 1) >> g.sh);
  return px;
}

// ---- shared epilogue ---------------------------------------------------------------------------------------------
template <typename Cfg, typename AccT>
__device__ __forceinline__ void igemm_epilogue(AccT& acc /* f32x16[MB][NB] */, int m0, int n0, int wm, int wn,
                                               int lane, int M, int tw_log2 = -1, int img2 = 0, int oy0 = 0,
                                               int ox0 = 0) {
  const int li = lane & 31, lh = lane >> 5;
  // The epilogue-only fields are re-read from kernarg memory here (opaque pointer) so that they do not occupy
  // SGPRs during the K-loop (the whole descriptor live = SGPR spills reloaded every iteration).
  const MsegIgemm* pe = (const MsegIgemm*)__builtin_amdgcn_kernarg_segment_ptr();  // kernel argument 0
  asm volatile("" : "+s"(pe));
  const float* e_bias = pe->bias;
  float* e_dst0 = pe->dst0;
  float* e_dst1 = pe->dst1;
  const int e_epi = pe->epi, e_split = pe->split, e_ld0 = pe->ld0, e_ld1 = pe->ld1, e_acc0 = pe->acc0,
            e_acc1 = pe->acc1, e_Cq = pe->Cq, e_morder = pe->morder, e_Ho = pe->Ho, e_Wo = pe->Wo, e_Ngemm = pe->Ngemm;
  const bool e_d16 = pe->dst_dtype == MSEG_ST_BF16;   // destinations stored as bf16 (round to nearest even on store)
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
      // Row addressing.  The 32 rows of one MFMA tile are consecutive GEMM rows; the destination element of row `row` is
      // tbase + rel(row), with everything that needs a division decoded ONCE per tile (wave-uniform) and only adds /
      // compares per row — per-element divisions are VALU work the short-K layers (ConvTranspose: K = Cin) cannot hide:
      //   mode 0  affine: rel = row * step  (one image-row segment; plain linear order; 32 | row length of the scatter /
      //           parity forms)
      //   mode 1  halo kernel with pixel tiles narrower than 32: shifts and masks
      //   mode 2  scatter (ConvTranspose) over rows that are no multiple of 32: column wraps by compare
      //   mode 3  parity order (stride-2 data gradient) likewise, inside one parity class
      //   mode 4  tile straddling two parity classes (generic kernel only): full decode per element
      const int mb = m0 + wm * Cfg::TM + a * 32;
      const int rows_left = (tw_log2 >= 0) ? 32 : M - mb;         // halo tiles never straddle the end of M
      bool tile_ok = mb < M;
      int mode = 0, step = 0;
      size_t tbase = 0;
      int b_img = 0, b_y = 0, b_x = 0, cls = 0;                  // decoded first row of the tile (modes 2, 3)
      if (tw_log2 >= 0) {
        const int i = wm * Cfg::TM + a * 32;
        const int oy = oy0 + (i >> tw_log2), ox = ox0 + (i & ((1 << tw_log2) - 1));
        tile_ok = oy < e_Ho;
        tbase = ((size_t)(img2 * e_Ho + oy) * e_Wo + ox) * ld + noff;
        step = ld;
        mode = tw_log2 >= 5 ? 0 : 1;
      } else if (e_epi == MSEG_EPI_SCATTER2X2) {
        if (tile_ok) {
          b_img = mb / (e_Ho * e_Wo);
          const int rem = mb - b_img * (e_Ho * e_Wo);
          b_y = rem / e_Wo; b_x = rem - b_y * e_Wo;
          tbase = (((size_t)b_img * (2 * e_Ho) + 2 * b_y + sa) * (2 * e_Wo) + 2 * b_x + sb) * e_Cq + noff;
          step = 2 * e_Cq;
          mode = (e_Wo & 31) == 0 ? 0 : 2;
        }
      } else if (e_morder == MSEG_MORDER_PARITY) {
        if (tile_ok) {
          const int Hh = e_Ho >> 1, Wh = e_Wo >> 1, per = (M >> 2);
          cls = mb / per;
          const int rem = mb - cls * per;
          b_img = rem / (Hh * Wh);
          const int r2 = rem - b_img * (Hh * Wh);
          b_y = r2 / Wh; b_x = r2 - b_y * Wh;
          tbase = (((size_t)b_img * e_Ho + 2 * b_y + (cls >> 1)) * e_Wo + 2 * b_x + (cls & 1)) * ld + noff;
          step = 2 * ld;
          mode = (Wh & 31) == 0 ? 0 : (rem + 31 < per ? 3 : 4);
        }
      } else {
        tbase = (size_t)mb * ld + noff;
        step = ld;
      }
      if (!nvalid || !tile_ok) continue;
      const unsigned BAD = 0xffffffffu;
      // element offset of `row` relative to tbase, or BAD — one closure per mode, selected by a wave-uniform branch
      // OUTSIDE the row loops (a mode switch inside them is if-converted into selects that every row pays for)
      auto rel_affine = [&](int row) -> unsigned { return row < rows_left ? (unsigned)(row * step) : BAD; };
      auto rel_narrow = [&](int row) -> unsigned {
        const int i0 = wm * Cfg::TM + a * 32, i = i0 + row;
        const int dy = (i >> tw_log2) - (i0 >> tw_log2), dx = (i & ((1 << tw_log2) - 1)) - (i0 & ((1 << tw_log2) - 1));
        if (oy0 + (i >> tw_log2) >= e_Ho) return BAD;
        return (unsigned)((dy * e_Wo + dx) * ld);
      };
      // modes 2 / 3: walk `row` pixels to the right inside a (rows x W) grid of W = Wo (scatter) or Wo / 2 (parity)
      auto rel_walk = [&](int row) -> unsigned {
        if (row >= rows_left) return BAD;
        const int Wg = mode == 2 ? e_Wo : (e_Wo >> 1), Hg = mode == 2 ? e_Ho : (e_Ho >> 1);
        int x = b_x + row, y = b_y, img = b_img;
        while (x >= Wg) { x -= Wg; ++y; }                         // <= 2 trips for rows of >= 16 pixels
        while (y >= Hg) { y -= Hg; ++img; }
        const long long d = mode == 2
            ? ((((long long)(img - b_img) * (2 * e_Ho) + 2 * (y - b_y)) * (2 * e_Wo)) + 2 * (x - b_x)) * e_Cq
            : (((long long)(img - b_img) * e_Ho + 2 * (y - b_y)) * e_Wo + 2 * (x - b_x)) * ld;
        return (unsigned)d;                                        // a 32-row tile spans a few image rows: fits 32 bits
      };
      // all read-modify-write loads first, then all stores: a load/store pair per element would serialise 16 memory round
      // trips (same pointer, the compiler may not reorder them), which short-K layers cannot hide
      auto emit = [&](auto rel) {
        if (e_d16) {
          // bf16 destination: a lane holds ONE channel of 16 rows, i.e. 2-byte stores.  Lanes 2k / 2k+1 (channels n, n+1;
          // bf16 storage has even channel counts) swap half of their values through a DPP quad permute, so that the even
          // lane stores the channel PAIR of the even accumulator rows and the odd lane that of the odd rows: 8 dword
          // stores (and 8 dword read-modify-write loads) per lane instead of 16 short ones.
          const bool odd = li & 1;
          __bf16* const d = reinterpret_cast<__bf16*>(dst) + (tbase - (odd ? 1 : 0));     // first channel of the pair
          unsigned oldw[8], ro[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int r = 2 * q + (odd ? 1 : 0);
            ro[q] = rel((r & 3) + 8 * (r >> 2) + 4 * lh);
            oldw[q] = 0u;
          }
          if (accf) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
              if (ro[q] != BAD) oldw[q] = *reinterpret_cast<const unsigned*>(d + ro[q]);
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const float mine_e = acc[a][b][2 * q] + bias, mine_o = acc[a][b][2 * q + 1] + bias;
            // the even lane keeps row 2q and hands row 2q+1 to its neighbour; the odd lane the other way round
            const float give = odd ? mine_e : mine_o, keep = odd ? mine_o : mine_e;
            const float got = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, give), 0xB1, 0xf, 0xf, true));
            const float lo = (odd ? got : keep) + bf16_lo(oldw[q]), hi = (odd ? keep : got) + bf16_hi(oldw[q]);
            if (ro[q] != BAD) *reinterpret_cast<unsigned*>(d + ro[q]) = pack_bf16x2(lo, hi);
          }
          return;
        }
        float* const d = dst + tbase;
        float old[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) old[r] = 0.f;
        if (accf) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const unsigned o = rel((r & 3) + 8 * (r >> 2) + 4 * lh);
            if (o != BAD) old[r] = d[o];
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const unsigned o = rel((r & 3) + 8 * (r >> 2) + 4 * lh);
          const float v = acc[a][b][r] + bias + old[r];   // (unconditional use: no read-modify-write load stays pending
          if (o != BAD) d[o] = v;                         //  past the epilogue in the compiler's wait-count bookkeeping)
        }
      };
      if (mode == 0) { emit(rel_affine); continue; }
      if (mode == 1) { emit(rel_narrow); continue; }
      if (mode != 4) { emit(rel_walk); continue; }
#pragma unroll
      for (int r = 0; r < 16; ++r) {                              // mode 4: parity tile across two classes
        const int m = mb + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= M) continue;
        const RowInfo ri = decode_row(*pe, m, M);
        const size_t off = (((size_t)ri.n * e_Ho + ri.oy) * e_Wo + ri.ox) * ld + noff;
        float v = acc[a][b][r] + bias;
        if (e_d16) {
          __bf16* const d = reinterpret_cast<__bf16*>(dst);
          if (accf) v += (float)d[off];
          d[off] = (__bf16)v;
        } else {
          if (accf) v += dst[off];
          dst[off] = v;
        }
      }
    }
  }
}
