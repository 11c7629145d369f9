// igemm_p8.hip — 3x3 stride-1 forward / data-gradient implicit GEMM on bf16 tensors for the layers with >= 128 output
// channels (levels 1..4 of the U-Nets: src/utils/unets.py:112,137 and their autograd, train.py:488): ONE 8-wave workgroup
// per compute unit on a large tile, built around three things the tile-per-workgroup kernels of igemm.hip lack:
//
//   * weights by LDS-DMA (`buffer_load_dwordx4 ... lds`) into a RING of tap slabs that stays in flight across barriers:
//     a stage = one tap x 32 input channels x BN output channels (BN x 64 bytes); the slab of stage s + 2 is issued while
//     stage s computes, a counted `s_waitcnt vmcnt(N)` retires exactly the slab of stage s + 1, raw `s_barrier`s (no
//     `__syncthreads()`: its fence would drain the DMA queue).  No weight byte passes through a register.
//   * conflict-free LDS images: weight rows are 64 bytes, DMA destinations are lane-linear, so the swizzle is applied to
//     the per-lane SOURCE address (16-byte chunk c of row r is stored at position c ^ ((r >> 2) & 3)) and to the fragment
//     read; halo rows are 80 bytes apart and a halo line is 32 rows apart for 16-pixel-wide tiles (34 for 32-pixel tiles),
//     so the 16 rows a `ds_read_b128` lane group touches are distinct modulo 16 for every tap shift
//     (MI355X_MICROARCH.md, LDS: groups {0-3,12-15,20-27} / {4-11,16-19,28-31}).
//   * two wave groups (waves 0-3 / 4-7: the two waves of each SIMD) run half a stage apart: while one group issues its 16
//     MFMAs of a stage, the other reads the next stage's fragments, issues DMA and stages the next chunk's halo.  Two
//     barriers per stage keep that alternation.
//
// Stage order inside a 32-channel chunk: the 9 taps; the halo of chunk c + 1 is fetched into registers at tap 0 of chunk
// c, normalised (act + scale / shift: norm-on-load) and written to the OTHER halo buffer at tap 5.
#include "igemm_common.h"
#include "bf16_affine.h"
#include <stdlib.h>

typedef __bf16 p8_bf16x8 __attribute__((ext_vector_type(8)));

#define P8_AROW 80              // bytes between halo rows in LDS (64 of data + 16: rows distinct mod 16 never share a bank)

// pixel tile of a launch: TH x TW pixels, TW a divisor of the image width.  32- and 16-pixel-wide tiles fill the BM GEMM rows
// exactly; images whose width is no multiple of 16 (the 40- and 20-pixel-wide deep levels of 320 x 320 crops) take tiles as
// wide as the image and floor(BM / W) rows tall (240 of 256 GEMM rows live).  HP: LDS rows between two halo lines — 32 for
// 16-pixel tiles and TW + 16 for image-wide ones (a line wrap inside a 32-pixel MFMA block then moves on by a multiple of
// 16 rows: the 16 rows a ds_read_b128 lane group touches stay distinct modulo 16), TW + 2 for 32-pixel tiles (no wrap).
struct P8Tile { int TW, TH, HP; };

template <int BM_, int BN_>
struct P8Cfg {
  static constexpr int BM = BM_, BN = BN_;
  static constexpr int WN = BN / 64, WM = 8 / WN;       // BN = 256: 2 x 4 waves; BN = 128: 4 x 2 waves
  static constexpr int TM = BM / WM, TN = 64;
  static constexpr int MB = TM / 32, NB = 2;
  static constexpr int NP = BN / 128;                   // DMA instructions (1 KB = 16 weight rows each) per wave and stage
  // real halo rows: (TH + 2) x (TW + 2); BM = 256: 18 x 18 or 10 x 34; BM = 512: 18 x 34 (32-pixel-wide tiles only)
  static constexpr int HREAL = BM == 256 ? 340 : 612;
  static constexpr int HL = (HREAL + 127) / 128;        // 16-byte halo loads per thread and chunk
  // LDS rows of a halo buffer: 16-wide tiles use a line pitch of 32 rows
  static constexpr int HROWS_LDS = BM == 256 ? 17 * 32 + 18 : 17 * 34 + 34;     // 562 / 612 (image-wide tiles: <= 8 x 66 + slack)
  static constexpr int HALO_BYTES = HROWS_LDS * P8_AROW;
  static constexpr int SLOT_BYTES = BN * 64;
  static constexpr int TAB_BYTES = 256;                  // norm-on-load tables of one chunk: scale[32] | shift[32]
  // weight slabs in the ring; a slab is issued P = RING - 2 stages ahead (hazard notes in the kernel).  In-order
  // completion of the vector-memory queue makes P also the number of stages a halo fetch (and an epilogue's stores) may
  // take before somebody waits for them: 128-channel slabs are 8 KB, six of them fit; 256-channel slabs (16 KB) leave
  // room for four.
  static constexpr int RING = BN == 128 ? 6 : 4;
  static constexpr int P = RING - 2;
  static constexpr int BIAS_BYTES = 8 * 256;             // per wave: the bias of its 64 output channels (epilogue)
  static constexpr int LDS_BYTES = 2 * HALO_BYTES + RING * SLOT_BYTES + 2 * TAB_BYTES + BIAS_BYTES;
};

// one 1 KB piece (16 weight rows of 64 bytes) of a weight slab by LDS-DMA
// (a __device__ function, not a lambda of the kernel: the host pass cannot instantiate the LDS-DMA builtin)
__device__ __forceinline__ void p8_issue_w(const void* w, unsigned wbytes, char* dst, unsigned voff, unsigned wso) {
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(w), 0, wbytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)dst, 16, voff, wso, 0, 0);
}

// ---- epilogue ---------------------------------------------------------------------------------------------------------------
// The accumulators are TRANSPOSED (the weights are the A operand of the MFMA): acc[a][b] is a [32 channels][32 pixels]
// block, lane (li, lh) holds pixel li and the channel quads 8 g + 4 lh + {0..3}, g = 0..3.  A lane therefore owns runs of
// 4 consecutive channels of ONE pixel of an NHWC tensor; lanes li and li + 32 exchange quads (v_permlane32_swap) so that
// each stores 8 consecutive bf16 channels = ONE 16-byte store per two quads.  Store instructions are what an epilogue of
// this shape is bound by (~70 cycles per wave-instruction whatever its width): the one-channel-per-lane layout of
// igemm_epilogue needs 64 dword stores per wave for this tile, this one 16 (measured on the 128 -> 128 channel layer at
// 160 x 160: 120 of 346 us were the stores).
// Same semantics as igemm_epilogue for what this kernel is launched on: plain epilogue, linear pixel order, bias, two
// destinations split at a multiple of 64 channels, optional accumulation into the destination, bf16 or fp32 tensors.
// ST: the launch also takes the statistics of the normalisation that follows (MsegIgemm.stats): per wave and tile the sums
// of act(z) and act(z)^2 over the wave's live pixels, z as stored, for its 64 channels -> stats[srow][0 / 1][channel].  A
// lane holds 16 channels of ONE pixel per MFMA block, so the sum over pixels is a sum over lanes: the 32 per-lane values
// (16 channels x 2 statistics, accumulated over the wave's MB blocks first) go through a five-step halving butterfly
// (ds_swizzle, xor 16 .. 1: each step a lane keeps the half of its list its lane bit selects and adds the partner's copy of
// that half) that leaves ONE total per lane — fixed summation tree, no atomics.
template <int MASK>
__device__ __forceinline__ float p8_swizzle_xor(float x) {      // x of lane (l ^ MASK), within groups of 32 lanes
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, x), (MASK << 10) | 0x1f));
}

template <typename Cfg, bool ST>
__device__ __forceinline__ void p8_epilogue(f32x16 (&acc)[Cfg::MB][Cfg::NB], int cw /* first channel of this wave */, int wm,
                                            int lane, int TW, int TH, int img, int oy0, int ox0, int srow,
                                            char* bias_lds /* this wave's 256 bytes */, int& staged_cw) {
  const int li = lane & 31, lh = lane >> 5;
  const MsegIgemm* pe = (const MsegIgemm*)__builtin_amdgcn_kernarg_segment_ptr();  // kernel argument 0 (kept out of the
  asm volatile("" : "+s"(pe));                                                     // K-loop's SGPRs, like igemm_epilogue)
  const float* e_bias = pe->bias;
  const int e_split = pe->split, e_Ngemm = pe->Ngemm, e_H = pe->Ho, e_W = pe->Wo;
  const bool e_d16 = pe->dst_dtype == MSEG_ST_BF16;
  if (cw >= e_Ngemm) return;
  const bool second = cw >= e_split;                   // a wave's 64 channels lie on one side of the split (launcher)
  typedef __attribute__((address_space(1))) char gchar;      // global, not flat: the pointer comes out of an opaque load
  gchar* const dst = (gchar*)(second ? pe->dst1 : pe->dst0);
  const int ld = second ? pe->ld1 : pe->ld0;
  const int accf = second ? pe->acc1 : pe->acc0;
  const int noff = cw - (second ? e_split : 0);
  // The bias of this wave's 64 channels sits in LDS (one memory round trip per channel-tile change of the persistent walk —
  // usually one per kernel — instead of two per tile); written and read by this wave only: no barrier.
  if (e_bias && staged_cw != cw) {
    *reinterpret_cast<float*>(bias_lds + lane * 4) = *(const __attribute__((address_space(1))) float*)(e_bias + cw + lane);
    staged_cw = cw;
  }
  if (ST) {
    // Statistics FIRST, as a pass of its own over the accumulators (nothing of this tile is in the vector-memory queue
    // yet), one channel quad at a time: 8 running sums per lane.  The bias is added to the accumulators IN PLACE (the store
    // pass below then adds none).  ReLU and the masking of dead rows are ONE v_med3 with per-row bounds.
    float* const e_stats = pe->stats;
    const float lo_live = pe->stats_act == MSEG_ACT_RELU ? 0.f : -__builtin_inff();
    float lo_a[Cfg::MB], hi_a[Cfg::MB];
#pragma unroll
    for (int a = 0; a < Cfg::MB; ++a) {
      const int i = wm * Cfg::TM + a * 32 + li;
      const int iy = i / TW;
      const bool ok = (iy < TH) & (oy0 + iy < e_H);
      lo_a[a] = ok ? lo_live : 0.f;
      hi_a[a] = ok ? __builtin_inff() : 0.f;
    }
    const bool u16 = (li & 16) != 0, u8 = (li & 8) != 0, u4 = (li & 4) != 0;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int b = 0; b < Cfg::NB; ++b) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {                    // channel quad g: accumulator registers 4 g .. 4 g + 3
        f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
        if (e_bias) bias4 = *reinterpret_cast<const f32x4*>(bias_lds + (b * 32 + 8 * g + 4 * lh) * 4);
        f32x4 S = {0.f, 0.f, 0.f, 0.f}, Q = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < Cfg::MB; ++a) {
#pragma unroll
          for (int k = 0; k < 4; k += 2) {
            float x0 = acc[a][b][4 * g + k] + bias4[k], x1 = acc[a][b][4 * g + k + 1] + bias4[k + 1];
            acc[a][b][4 * g + k] = x0; acc[a][b][4 * g + k + 1] = x1;
            if (e_d16) {                               // the values as the store below rounds them
              const unsigned d = pack_bf16x2(x0, x1);
              x0 = bf16_lo(d); x1 = bf16_hi(d);
            }
            x0 = __builtin_amdgcn_fmed3f(x0, lo_a[a], hi_a[a]);
            x1 = __builtin_amdgcn_fmed3f(x1, lo_a[a], hi_a[a]);
            S[k] += x0; S[k + 1] += x1;
            Q[k] = fmaf(x0, x0, Q[k]); Q[k + 1] = fmaf(x1, x1, Q[k + 1]);
          }
        }
        // Sum over the 32 lanes that hold the same channels (a lane = one pixel): halving butterfly — in each step a lane
        // keeps the half of its list its lane bit selects and adds the partner's copy of that half — down to one value,
        // then two plain exchange-and-add steps.  Fixed summation tree.
        const f32x4 s4 = u16 ? S : Q, k4 = u16 ? Q : S;
        f32x4 g4;
#pragma unroll
        for (int j = 0; j < 4; ++j) g4[j] = p8_swizzle_xor<16>(s4[j]);
        const f32x4 w4 = k4 + g4;
        const f32x2 lo2 = w4.lo, hi2 = w4.hi;
        const f32x2 s2 = u8 ? lo2 : hi2, k2 = u8 ? hi2 : lo2;
        f32x2 g2;
        g2[0] = p8_swizzle_xor<8>(s2[0]); g2[1] = p8_swizzle_xor<8>(s2[1]);
        const f32x2 w2 = k2 + g2;
        float t = (u4 ? w2[1] : w2[0]) + p8_swizzle_xor<4>(u4 ? w2[0] : w2[1]);
        t += p8_swizzle_xor<2>(t);
        t += p8_swizzle_xor<1>(t);
        // lanes with equal bits 4, 3, 2 now hold statistic (li >> 4) of channel 2 bit3 + bit2 of the quad
        if ((li & 3) == 0)
          e_stats[((size_t)srow * 2 + (li >> 4)) * e_Ngemm + cw + b * 32 + 8 * g + 4 * lh + ((li >> 2) & 3)] = t;
      }
    }
  }
#pragma unroll
  for (int b = 0; b < Cfg::NB; ++b) {
    float4 bq[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
    {
      bq[g] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e_bias && !ST) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(bias_lds + (b * 32 + 8 * g + 4 * lh) * 4);
        bq[g] = make_float4(t[0], t[1], t[2], t[3]);
      }
    }
#pragma unroll
    for (int a = 0; a < Cfg::MB; ++a) {
      const int i = wm * Cfg::TM + a * 32 + li;
      const int iy = i / TW;
      const int oy = oy0 + iy, ox = ox0 + (i - iy * TW);
      const bool ok = (iy < TH) & (oy < e_H);          // dead GEMM rows of a tile, tile rows below the image
      const size_t e0 = ((size_t)(img * e_H + oy) * e_W + ox) * ld + noff + b * 32;   // element offset of channel quad 0
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = acc[a][b][r] + (&bq[r >> 2].x)[r & 3];
      if (e_d16) {
#pragma unroll
        for (int P = 0; P < 2; ++P) {
          // this lane stores channel group 2 P + lh (8 channels, 16 bytes)
          typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
          auto* const q = (__attribute__((address_space(1))) u32x4*)(dst + (e0 + 8 * (2 * P + lh)) * 2);
          if (accf) {
            u32x4 old = {0u, 0u, 0u, 0u};
            if (ok) old = *q;
            // bring the old values to the lanes that hold the matching quads (the inverse of the exchange below)
            auto s0 = __builtin_amdgcn_permlane32_swap(old.x, old.z, false, false);
            auto s1 = __builtin_amdgcn_permlane32_swap(old.y, old.w, false, false);
            const unsigned o0 = s0[0], o2 = s0[1], o1 = s1[0], o3 = s1[1];
            v[8 * P + 0] += bf16_lo(o0); v[8 * P + 1] += bf16_hi(o0); v[8 * P + 2] += bf16_lo(o1); v[8 * P + 3] += bf16_hi(o1);
            v[8 * P + 4] += bf16_lo(o2); v[8 * P + 5] += bf16_hi(o2); v[8 * P + 6] += bf16_lo(o3); v[8 * P + 7] += bf16_hi(o3);
          }
          const unsigned d0 = pack_bf16x2(v[8 * P + 0], v[8 * P + 1]), d1 = pack_bf16x2(v[8 * P + 2], v[8 * P + 3]);
          const unsigned d2 = pack_bf16x2(v[8 * P + 4], v[8 * P + 5]), d3 = pack_bf16x2(v[8 * P + 6], v[8 * P + 7]);
          // lanes 32-63 hand their quad of group 2 P down and take the lower lanes' quad of group 2 P + 1
          auto x0 = __builtin_amdgcn_permlane32_swap(d0, d2, false, false);
          auto x1 = __builtin_amdgcn_permlane32_swap(d1, d3, false, false);
          if (ok) { const u32x4 o = {x0[0], x1[0], x0[1], x1[1]}; *q = o; }
        }
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          auto* const q = (__attribute__((address_space(1))) f32x4*)(dst + (e0 + 8 * g + 4 * lh) * 4);
          f32x4 o = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
          if (ok) {
            if (accf) o += *q;
            *q = o;
          }
        }
      }
    }
  }
}

template <int BM, int BN, int TR, bool ST>
__global__ __launch_bounds__(512, 2) void igemm_p8_kernel(const MsegIgemm p, const P8Tile tg, int m_fastest, int ntiles, int dbg) {
  using Cfg = P8Cfg<BM, BN>;
  constexpr int HL = Cfg::HL, NP = Cfg::NP, RING = Cfg::RING, P = Cfg::P;
  constexpr int CSTAGES = 7 - P;                       // stages P .. 6 of a chunk commit the next chunk's halo rows
  constexpr int CROWS = (HL + CSTAGES - 1) / CSTAGES;  // rows per such stage
  constexpr int HLX = HL + (TR != 0 ? 1 : 0);          // vector-memory operations of one halo fetch (+ 1 table load)
  __shared__ __attribute__((aligned(16))) char lds[Cfg::LDS_BYTES];
  char* const ring = lds + 2 * Cfg::HALO_BYTES;
  char* const tabs = ring + Cfg::RING * Cfg::SLOT_BYTES;
  char* const bias_lds = tabs + 2 * Cfg::TAB_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const int li = lane & 31, lh = lane >> 5;
  const int srow = tid >> 2, scol = tid & 3;           // halo staging: 128 rows x 4 groups of 8 channels per pass
  const int TW = tg.TW, TH = tg.TH, HW2 = TW + 2;       // tile = TH x TW pixels (TH * TW <= BM; the rest are dead GEMM rows)
  const int HP = tg.HP;                                // LDS rows between two halo lines
  const int NREAL = (TH + 2) * HW2;
  const int H = p.Hi, W = p.Wi;
  const int tiles_x = W / TW, tiles_y = (H + TH - 1) / TH;
  const int ntiles_n = (p.Ngemm + BN - 1) / BN;
  const int ntiles_m = ntiles / ntiles_n;
  const bool conv = p.mode == MSEG_MODE_CONV;
  const int nchunks = (p.Cin + KC - 1) / KC;
  const int C0 = p.src[0].C;
  const int C1 = p.nsrc > 1 ? p.src[1].C : p.src[0].C;
  const unsigned OOB = 0x80000000u;
  const unsigned wbytes = 9u * (unsigned)p.Npad * (unsigned)p.Kpad * 2u;
  const unsigned tap_stride = (unsigned)p.Npad * (unsigned)p.Kpad * 2u;     // bytes between the slabs of two taps

  // ---- tile-independent per-thread tables ----------------------------------------------------------------------------------
  // halo staging: (row, column) of this thread's rows inside a tile's halo and their LDS byte offsets
  int hyx[HL];
  unsigned hreal = 0u;
#pragma unroll
  for (int j = 0; j < HL; ++j) {
    const int idx = srow + 128 * j;
    const int hy = idx / HW2, hx = idx - hy * HW2;
    hyx[j] = (hy << 16) | hx;
    hreal |= (unsigned)(idx < NREAL) << j;
  }
  // fragment read addresses (bytes): A = halo row of this lane's pixel (tap offset added per stage), B = swizzled weight row
  int abase[Cfg::MB];
#pragma unroll
  for (int a = 0; a < Cfg::MB; ++a) {
    const int i = wm * Cfg::TM + a * 32 + li;          // GEMM row -> pixel (i / TW, i % TW) of the tile; rows >= TH * TW are
    const int iy = i / TW;                             // dead (their reads land in the halo buffer's slack, their results
    abase[a] = (iy * HP + (i - iy * TW)) * P8_AROW + lh * 16;   // are never stored)
  }
  int bbase[2][2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int r = wn * 64 + b * 32 + li;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) bbase[b][kk] = r * 64 + (((kk * 2 + lh) ^ ((r >> 2) & 3)) << 4);
  }
  // weight DMA: piece q = wave + 8 i covers slab rows 16 q .. 16 q + 15; lane L fills position L & 3 of row 16 q + (L >> 2)
  // with the row's chunk (L & 3) ^ ((row >> 2) & 3)
  unsigned wrow_off[NP];                               // tile independent part: (row, chunk) of this lane
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int row = 16 * (wave + 8 * i) + (lane >> 2);
    const int c = (lane & 3) ^ ((row >> 2) & 3);
    wrow_off[i] = ((unsigned)row * (unsigned)p.Kpad + (unsigned)c * 8u) * 2u;
  }

  // ---- the persistent tile walk ----------------------------------------------------------------------------------------------
  // A workgroup takes the tiles lw, lw + G, lw + 2G, ... (lw = its XCD-aware logical id: in every round the workgroups of
  // one XCD work on neighbouring tiles and share weight / halo lines in that XCD's L2).  Its stage stream runs ACROSS tiles:
  // the halo of the next tile's first chunk and its first two weight slabs are fetched during the last chunk of the
  // current tile, the epilogue of one wave group runs under the other group's last MFMAs.
  const int G = (int)gridDim.x;
  const int lw = (int)xcd_logical_id(blockIdx.x, gridDim.x);
  // geometry of the tile the FETCH side works for (halo loads: from tap 0 of a tile's last chunk on they belong to the next
  // tile; weight slabs: from tap 7 of the last chunk on) and of the tile being COMPUTED (epilogue)
  int f_img = 0, f_oy0 = 0, f_ox0 = 0, f_band0 = 0, f_band_rows = 1;
  bool f_live = false;
  unsigned w_n0off = 0u;
  unsigned hvalid = 0u;
  int c_img = 0, c_oy0 = 0, c_ox0 = 0, c_n0 = 0;
  int staged_cw = -1;                                  // first channel of the bias this wave holds in LDS

  auto tile_m_of = [&](int t) { return m_fastest ? t % ntiles_m : t / ntiles_n; };
  auto tile_coords = [&](int t, int& img, int& oy0, int& ox0, int& n0) {
    const int tile_m = m_fastest ? t % ntiles_m : t / ntiles_n;
    const int tile_n = m_fastest ? t / ntiles_m : t - tile_m * ntiles_n;
    img = tile_m / (tiles_x * tiles_y);
    const int trem = tile_m - img * (tiles_x * tiles_y);
    const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
    oy0 = ty * TH; ox0 = tx * TW; n0 = tile_n * BN;
  };
  auto set_fetch_halo = [&](int t) {                   // t >= ntiles: no further tile, every fetch is dead
    f_live = t < ntiles;
    int n0;
    tile_coords(f_live ? t : 0, f_img, f_oy0, f_ox0, n0);
    f_band0 = f_oy0 > 0 ? f_oy0 - 1 : 0;               // descriptors span the row band of the tile's halo
    f_band_rows = (f_oy0 + TH + 1 < H ? f_oy0 + TH + 1 : H) - f_band0;
    hvalid = 0u;
#pragma unroll
    for (int j = 0; j < HL; ++j) {
      const int iy = f_oy0 - 1 + (hyx[j] >> 16), ix = f_ox0 - 1 + (hyx[j] & 0xffff);
      const bool ok = f_live & ((hreal >> j) & 1u) & (iy >= 0) & (iy < H) & (ix >= 0) & (ix < W);
      hvalid |= (unsigned)ok << j;
    }
  };
  auto set_fetch_w = [&](int t) {                      // t >= ntiles: keep the old slab addresses (read, never consumed)
    if (t < ntiles) {
      int img, oy0, ox0, n0;
      tile_coords(t, img, oy0, ox0, n0);
      w_n0off = (unsigned)n0 * (unsigned)p.Kpad * 2u;
    }
  };

#define P8_ISSUE_W(chunk_, tap_, slot_)                                                                    \
  _Pragma("unroll") for (int i_ = 0; i_ < NP; ++i_)                                                         \
    p8_issue_w(p.w, wbytes, ring + (slot_) * Cfg::SLOT_BYTES + (wave + 8 * i_) * 1024, wrow_off[i_] + w_n0off, \
               (unsigned)(tap_) * tap_stride + (unsigned)(chunk_) * (KC * 2u))

  f32x4 rh[HL];                                        // 16 raw bytes: 8 bf16 source channels
  f32x4 rtab;                                          // lanes 0..15 of a wave: one float4 of the fetched chunk's tables
  float tone = 0.f;                                    // (wave 0 hands them to everybody through LDS: 16 registers less per lane)
  unsigned hlive = 0u;
  int ract = 0;

  auto issue_halo = [&](int chunk) {                   // halo of `chunk` of the fetch tile (dead tile: offsets out of bounds)
    const bool s1 = (p.nsrc > 1) && (chunk * KC >= C0);
    const MsegSrc& s = s1 ? p.src[1] : p.src[0];
    const unsigned sCB = (unsigned)s.C * 2u;                                       // bytes per pixel
    const unsigned soff = (unsigned)(chunk * KC - (s1 ? C0 : 0)) * 2u + scol * 16u;
    if (TR != 0) {
      ract = s.act;
      const bool has_aff = f_live && s.scale != nullptr;
      // ALWAYS issued (one load per lane), so that the counted waits of the stage loop see the same number of operations
      // whether or not the operand carries an affine (then a harmless address is read and zeros are published instead).
      // Issued BEFORE the halo rows: the wait for it in the next stage then does not wait for them.
      // lanes 0..7: scale of channels 4 l .. 4 l + 3 of the chunk, lanes 8..15: shift
      const float* tp = (const float*)p.w;
      if (has_aff && lane < 16)
        tp = ((lane & 8) ? s.shift : s.scale) + (size_t)f_img * (unsigned)s.ss + (unsigned)(chunk * KC - (s1 ? C0 : 0)) + (lane & 7) * 4;
      rtab = *(const __attribute__((address_space(1))) f32x4*)tp;
      tone = has_aff ? 0.f : 1.f;                      // no affine: scale 1 (0 loaded + 1), shift 0 — added at commit time
    }
    const char* const base = (const char*)s.ptr + ((size_t)f_img * H + f_band0) * W * sCB;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0,
                                                                         f_band_rows * W * (int)sCB, 0x00020000);
    hlive = hvalid;
#pragma unroll
    for (int j = 0; j < HL; ++j) {
      const bool ok = (hlive >> j) & 1u;
      const int iy = f_oy0 - 1 + (hyx[j] >> 16), ix = f_ox0 - 1 + (hyx[j] & 0xffff);
      const unsigned vo = ok ? (unsigned)((iy - f_band0) * W + ix) * sCB + soff : OOB;
      rh[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 0, 0));
    }
  };
  // wave 0 publishes the fetched tables (placed BEFORE a stage's fragment reads: the wait for those reads then covers it)
  auto publish_tables = [&](char* tb) {
    if (TR != 0 && wave == 0 && lane < 16) {
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(tb + lane * 16) = tone != 0.f ? zero : rtab;
    }
  };

  auto xform4 = [&](float4 v, const float4& sc, const float4& sh, float lo, float m) -> float4 {
    if (TR == 2) v = act_fwd4(v, ract);
    else { v.x = clamp_lo(v.x, lo); v.y = clamp_lo(v.y, lo); v.z = clamp_lo(v.z, lo); v.w = clamp_lo(v.w, lo); }
    v.x = (v.x * sc.x + sh.x) * m; v.y = (v.y * sc.y + sh.y) * m;
    v.z = (v.z * sc.z + sh.z) * m; v.w = (v.w * sc.w + sh.w) * m;
    return v;
  };

  // row j of the fetched halo: registers -> LDS (norm-on-load in between; tables of the chunk from `tb`)
  auto commit_row = [&](char* hbuf, const char* tb, int j) {
    if (!((hreal >> j) & 1u)) return;
    const uint4 raw = __builtin_bit_cast(uint4, rh[j]);
    char* const dstp = hbuf + ((hyx[j] >> 16) * HP + (hyx[j] & 0xffff)) * P8_AROW + scol * 16;
    if (TR == 0) {                                     // plain bf16 operand: already in LDS format (dead rows arrive as zeros)
      *reinterpret_cast<uint4*>(dstp) = raw;
    } else {
      const float lo = (ract == MSEG_ACT_RELU) ? 0.f : -3.402823466e38f;
      const f32x4 c0 = *reinterpret_cast<const f32x4*>(tb + scol * 32), c1 = *reinterpret_cast<const f32x4*>(tb + scol * 32 + 16);
      const f32x4 h0 = *reinterpret_cast<const f32x4*>(tb + 128 + scol * 32), h1 = *reinterpret_cast<const f32x4*>(tb + 144 + scol * 32);
      const float4 rsc = make_float4(c0[0] + tone, c0[1] + tone, c0[2] + tone, c0[3] + tone);
      const float4 rsc2 = make_float4(c1[0] + tone, c1[1] + tone, c1[2] + tone, c1[3] + tone);
      const float4 rsh = make_float4(h0[0], h0[1], h0[2], h0[3]);
      const float4 rsh2 = make_float4(h1[0], h1[1], h1[2], h1[3]);
      if (TR == 1) {
        *reinterpret_cast<uint4*>(dstp) = mseg_affine8_bf16(raw, ract == MSEG_ACT_RELU ? 0u : 0x80008000u, rsc, rsh, rsc2, rsh2,
                                                            ((hlive >> j) & 1u) != 0u);
        return;
      }
      const float m = ((hlive >> j) & 1u) ? 1.f : 0.f;
      const float4 a = xform4(bf16x4_to_f32(make_uint2(raw.x, raw.y)), rsc, rsh, lo, m);
      const float4 b = xform4(bf16x4_to_f32(make_uint2(raw.z, raw.w)), rsc2, rsh2, lo, m);
      const uint2 pa = f32x4_to_bf16(a), pb = f32x4_to_bf16(b);
      *reinterpret_cast<uint4*>(dstp) = make_uint4(pa.x, pa.y, pb.x, pb.y);
    }
  };

  f32x16 acc[Cfg::MB][Cfg::NB];

  // ---- prologue: halo of the first tile's chunk 0, weight slabs of its stages 0 and 1 -----------------------------------------
  int t = lw;
  set_fetch_halo(t);
  set_fetch_w(t);
  issue_halo(0);
#pragma unroll
  for (int k = 0; k < P; ++k) P8_ISSUE_W(0, k, k);
  publish_tables(tabs);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int j = 0; j < HL; ++j) commit_row(lds, tabs, j);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  // waves 4-7 run half a stage behind waves 0-3: one barrier more here, one more for waves 0-3 after the loop
  if (wave >= 4) __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  // Hazards (interval = code between two barriers; waves 0-3 run L(s) in interval 2s and M(s) in 2s + 1, waves 4-7 one
  // interval later; a wave in interval j knows that every wave has finished interval j - 2 and every LDS read issued in it):
  //   * weights of stage u: DMA issued in L(u - P), retired by the issuing wave's counted vmcnt in L(u - 1), i.e. before
  //     barrier 2u at the latest; first read in interval 2u (one barrier after the wait);
  //   * the slab of stage u replaces that of stage u - RING, last read in interval 2(u - RING) + 1 and complete by the
  //     lgkmcnt(0) that opens interval 2(u - RING) + 2; the earliest DMA for stage u is issued in interval 2(u - P) = that
  //     + 2 (RING = P + 2);
  //   * halo buffer of the next chunk: rows written at taps P .. 6 of this chunk (intervals >= 18c + 4; the previous
  //     chunk's last reads are complete in 18c + 1); a row written in L(6) by waves 4-7 (interval 18c + 13) is complete
  //     when their reads of L(7) have returned (interval 18c + 16); first read at tap 0 of the next chunk (18c + 18);
  //   * across a tile boundary nothing changes for LDS (the stage and chunk counters run on); the vector-memory queue is
  //     drained before an epilogue (its stores are not countable), so the first stage of a tile needs no counted wait.
  int rs = 0, gchunk = 0;                              // ring slot of the current stage, chunk counter (halo / table buffer parity)
  for (; t < ntiles; t += G) {
    tile_coords(t, c_img, c_oy0, c_ox0, c_n0);
    for (int chunk = 0; chunk < nchunks; ++chunk, ++gchunk) {
      const bool last = chunk == nchunks - 1;
      const char* const hb = lds + (gchunk & 1) * Cfg::HALO_BYTES;
      char* const hb_next = lds + ((gchunk & 1) ^ 1) * Cfg::HALO_BYTES;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap, rs = (rs + 1 == RING ? 0 : rs + 1)) {
        const int ky = tap / 3, kx = tap - 3 * ky;
        char* const tb_next = tabs + ((gchunk & 1) ^ 1) * Cfg::TAB_BYTES;
        if (tap == 1) publish_tables(tb_next);           // fetched at tap 0, read by the commits of taps 2..
        // next chunk's halo: one row per stage, registers -> LDS.  Before this stage's fragment reads (the commit's
        // temporaries and the 48 fragment registers are then never live together) and before its DMA is issued (whatever the
        // compiler waits for here is older than the slab of stage s + 1, which has to be complete below anyway).
        if (tap >= P && tap <= 6 && !(dbg & 2)) {
#pragma unroll
          for (int r = 0; r < CROWS; ++r)
            if ((tap - P) * CROWS + r < HL) commit_row(hb_next, tb_next, (tap - P) * CROWS + r);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- L(s): fragments of this stage --------------------------------------------------------------------------
        const int toff = ((conv ? ky : 2 - ky) * HP + (conv ? kx : 2 - kx)) * P8_AROW;          // scalar
        const char* const As = hb + toff;
        const char* const Bs = ring + rs * Cfg::SLOT_BYTES;
        p8_bf16x8 af[Cfg::MB][2], bf[2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) bf[b][kk] = *reinterpret_cast<const p8_bf16x8*>(Bs + bbase[b][kk]);
#pragma unroll
        for (int a = 0; a < Cfg::MB; ++a)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) af[a][kk] = *reinterpret_cast<const p8_bf16x8*>(As + abase[a] + kk * 32);
        // ---- weights of stage s + 2 ---------------------------------------------------------------------------------
        if (tap == 9 - P && last) set_fetch_w(t + G);
        {
          int t2 = tap + P, c2 = chunk;
          if (t2 >= 9) { t2 -= 9; c2 = last ? 0 : chunk + 1; }
          P8_ISSUE_W(c2, t2, rs + P >= RING ? rs + P - RING : rs + P);
        }
        __builtin_amdgcn_sched_barrier(0);               // the counted waits below rely on this issue order
        if (tap == 0) {
          if (last) set_fetch_halo(t + G);
          issue_halo(last ? 0 : chunk + 1);
        }
        // Retire the slab of stage s + 1.  Outstanding, oldest first: slabs s + 1 .. s + P, with the halo fetch of this
        // chunk's tap 0 right behind the slab issued there — younger than slab s + 1 up to tap P - 1.  The first P - 1
        // stages of a tile read slabs that were retired before the previous epilogue (or by the prologue), and what else
        // is outstanding then (the epilogue's stores) cannot be counted: no wait.
        if (tap < P) {
          if (!(tap < P - 1 && chunk == 0)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((P - 1) * NP + HLX) : "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"((P - 1) * NP) : "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- M(s) ---------------------------------------------------------------------------------------------------
        __builtin_amdgcn_s_setprio(1);
        if (tap == 0 && chunk == 0) {                    // first stage of a tile: the accumulators start from zero
          f32x16 zero;
#pragma unroll
          for (int r = 0; r < 16; ++r) zero[r] = 0.f;
#pragma unroll
          for (int a = 0; a < Cfg::MB; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[b][0], af[a][0], zero, 0, 0, 0);
        } else {
#pragma unroll
          for (int a = 0; a < Cfg::MB; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[b][0], af[a][0], acc[a][b], 0, 0, 0);
        }
#pragma unroll
        for (int a = 0; a < Cfg::MB; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[b][1], af[a][1], acc[a][b], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- epilogue of this tile (the other wave group is half a stage away: its MFMAs / fragment reads run meanwhile) --------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // slabs of the next tile's stages 0 and 1 (see above)
    if (dbg & 1) {                                       // timing ablation: no stores (one element keeps the MFMAs alive)
      if (acc[0][0][0] == 12345.678f) p.dst0[0] = acc[1][1][3];
    } else {
      p8_epilogue<Cfg, ST>(acc, c_n0 + wn * 64, wm, lane, TW, TH, c_img, c_oy0, c_ox0, ST ? tile_m_of(t) * Cfg::WM + wm : 0,
                           bias_lds + wave * 256, staged_cw);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (wave < 4) __builtin_amdgcn_s_barrier();
  // no DMA may land in this workgroup's LDS after it has ended
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef P8_ISSUE_W
}

#ifdef P8_INSPECT       // device-only build of one instantiation for reading the assembly
template __global__ void igemm_p8_kernel<P8_INSPECT>(const MsegIgemm, const P8Tile, int, int, int);
#else
// Launches the kernel above when the layer qualifies; returns 1 if it did, 0 if the caller has to pick another kernel,
// a negative MSEG_E* code on a launch error.  `tr`: 0 plain operand, 1 ReLU / no activation + affine, 2 any activation.
static int g_p8_on = -1;                               // -1: not set yet (MSEG_P8 in the environment, else 1)
extern "C" int mseg_igemm_set_p8(int on) {
  g_p8_on = on;                                        // 0 off, 1 default choice, 2 = force the 256-pixel x 128-channel form
  return MSEG_OK;
}

// tile geometry for this layer and BM, or false (tiles that would waste more than ~30 % of their GEMM rows, images narrower
// than 16 pixels, ...); *tiles = pixel tiles of the launch
static bool p8_geometry(const MsegIgemm& p, int BMv, P8Tile* tg, long long* tiles) {
  const int W = p.Wi, H = p.Hi;
  if ((W % 32) == 0) { tg->TW = 32; tg->HP = 34; }
  else if ((W % 16) == 0 && BMv == 256) { tg->TW = 16; tg->HP = 32; }
  else if (BMv == 256 && W >= 16 && W <= 64 && (W % 4) == 0) { tg->TW = W; tg->HP = W + 16; }   // line wrap = 16 rows further
  else return false;
  tg->TH = BMv / tg->TW;
  if ((tg->TH + 2) * (tg->TW + 2) > (BMv == 256 ? 340 : 612)) return false;      // staging loads per thread (P8Cfg::HREAL)
  if ((tg->TH + 1) * tg->HP + tg->TW + 2 > (BMv == 256 ? 562 : 612)) return false;   // LDS rows of a halo buffer
  const long long ty = (H + tg->TH - 1) / tg->TH;
  // live GEMM rows / rows computed: dead rows of a tile and tile rows below the image
  if ((long long)H * W * 10 < ty * (W / tg->TW) * BMv * 7) return false;
  *tiles = (long long)p.NB * ty * (W / tg->TW);
  return true;
}

int igemm_p8_try(const MsegIgemm& p, int tr, int m_fastest, int cus, hipStream_t st) {
  if (g_p8_on < 0) {
    const char* e = getenv("MSEG_P8");                  // ablation from the shell: MSEG_P8=0 python bench.py ...
    g_p8_on = e ? atoi(e) : 1;
  }
  if (!(g_p8_on & 255)) return 0;
  if (p.Ngemm < 128 || (p.Ngemm % 64) != 0 || (p.Cin % KC) != 0) return 0;
  if (p.split < p.Ngemm && (p.split % 64) != 0) return 0;            // a wave's 64 channels go to ONE destination
  if ((p.ld0 % 8) != 0 || (p.split < p.Ngemm && (p.ld1 % 8) != 0)) return 0;   // 16-byte stores
  if (((uintptr_t)p.dst0 | (uintptr_t)p.dst1) & 15) return 0;
  if (p.bias && ((uintptr_t)p.bias & 15)) return 0;
  if (p.nsrc > 1 && (p.src[0].C % KC) != 0) return 0;
  if ((long long)p.Hi * p.Wi < 256) return 0;
  // statistics for the following normalisation (MsegIgemm.stats): plain / ReLU operand forms, one destination written once
  const bool stats = p.stats && tr != 2 && p.split >= p.Ngemm && !p.acc0 &&
                     (p.stats_act == MSEG_ACT_NONE || p.stats_act == MSEG_ACT_RELU);
#define P8_LAUNCH(BM_, BN_, tiles_)                                                                                    \
  do {                                                                                                                 \
    const long long nt_ = (tiles_) * ((p.Ngemm + BN_ - 1) / BN_);                                                      \
    const dim3 grid((unsigned)(nt_ < cus ? nt_ : cus));                                                                \
    if (stats && (tiles_) * P8Cfg<BM_, BN_>::WM > 0x7fffffffLL) return 0;                                               \
    if (stats) mseg_dispatch_note_stats((int)((tiles_) * P8Cfg<BM_, BN_>::WM));                                         \
    if (stats && tr == 0) MSEG_KL((igemm_p8_kernel<BM_, BN_, 0, true>), grid, dim3(512), 0, st, p, tg, m_fastest, (int)nt_, g_p8_on >> 8); \
    else if (stats) MSEG_KL((igemm_p8_kernel<BM_, BN_, 1, true>), grid, dim3(512), 0, st, p, tg, m_fastest, (int)nt_, g_p8_on >> 8); \
    else if (tr == 0) MSEG_KL((igemm_p8_kernel<BM_, BN_, 0, false>), grid, dim3(512), 0, st, p, tg, m_fastest, (int)nt_, g_p8_on >> 8);    \
    else if (tr == 1) MSEG_KL((igemm_p8_kernel<BM_, BN_, 1, false>), grid, dim3(512), 0, st, p, tg, m_fastest, (int)nt_, g_p8_on >> 8); \
    else MSEG_KL((igemm_p8_kernel<BM_, BN_, 2, false>), grid, dim3(512), 0, st, p, tg, m_fastest, (int)nt_, g_p8_on >> 8);            \
    MSEG_LAUNCH_CHECK();                                                                                               \
  } while (0)
  P8Tile tg;
  long long t = 0;
  if (p.Ngemm % 256 == 0) {
    if (!p8_geometry(p, 256, &tg, &t) || t * (p.Ngemm / 256) < cus) return 0;
    P8_LAUNCH(256, 256, t);
    return 1;
  }
  if ((g_p8_on & 255) != 2 && p8_geometry(p, 512, &tg, &t) && t * ((p.Ngemm + 127) / 128) >= cus) {
    P8_LAUNCH(512, 128, t);
    return 1;
  }
  if (!p8_geometry(p, 256, &tg, &t) || t * ((p.Ngemm + 127) / 128) < cus) return 0;
  P8_LAUNCH(256, 128, t);
  return 1;
#undef P8_LAUNCH
}
#endif
