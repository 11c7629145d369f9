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
// Three kernels share one epilogue (dispatch: mseg_igemm / igemm_plan at the end of this file):
//   igemm_halo_kernel  3x3 stride-1 forward + data gradient (the bulk of the FLOPs): 128 pixels x BN channels per
//                      8-wave workgroup, the input halo of a 32-channel chunk staged and normalised ONCE for all 9 taps,
//                      two workgroups per CU;
//   igemm_fast_kernel  everything else that meets the 32-bit-offset preconditions (stride-2 convs and their parity-ordered
//                      data gradients, ConvTranspose as a 1x1 GEMM, its data gradient): per-tap gather with precomputed
//                      row offsets / tap masks, 4 waves, 32x32 MFMA tiles;
//   igemm_kernel       fully general gather (tensors >= 2 GiB, channel counts that are no multiple of 32 at a concat).
// Common tiling: every K-step consumes a [pixels][32] slab of source pixels (one tap, 32 input channels) and a [BN][32]
// slab of packed weights from LDS (row stride 36 floats -> conflict-free ds_read_b128); raw global loads of step s+1 are
// issued before the MFMAs of step s and consumed (activation, scale/shift, ds_write) after them.
#include "common.h"
#include "igemm_common.h"
#include "bf16_affine.h"

// igemm_p8.hip: one 8-wave workgroup per CU with DMA-streamed weights (>= 128 output channels, bf16 tensors)
int igemm_p8_try(const MsegIgemm& p, int tr, int m_fastest, int cus, hipStream_t st);

// identity affine for operands that carry no scale/shift table: lets the K-loop load the tables unconditionally
__device__ float g_ident_scale[MSEG_MAX_CH];
__device__ float g_ident_shift[MSEG_MAX_CH];
__global__ void init_ident_kernel() {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < MSEG_MAX_CH; i += gridDim.x * blockDim.x) {
    g_ident_scale[i] = 1.f;
    g_ident_shift[i] = 0.f;
  }
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
  // consecutive logical ids share the A slab (same M tile, different N tile) and are placed on one XCD -> L2 reuse
  const int lid = (int)xcd_logical_id(blockIdx.x, gridDim.x);
  const int tile_m = lid / ntiles_n;
  const int tile_n = lid - tile_m * ntiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int T = p.KH * p.KW;
  const int srow = tid >> 3;  // staging row within a 32-row pass
  const int scol = tid & 7;   // float4 column (4 channels)

  const TapGeom geom = make_geom(p);
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
        any |= tap_coord(geom, rows[i], ky, kx, iy, ix);
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
    const bool s1 = (p.nsrc > 1) & (c >= C0);
    const float* sptr = s1 ? p.src[1].ptr : p.src[0].ptr;
    const float* sscale = s1 ? p.src[1].scale : p.src[0].scale;
    const float* sshift = s1 ? p.src[1].shift : p.src[0].shift;
    const unsigned sC = (unsigned)(s1 ? p.src[1].C : p.src[0].C);
    const unsigned sss = (unsigned)(s1 ? p.src[1].ss : p.src[0].ss);
    ract = s1 ? p.src[1].act : p.src[0].act;
    const unsigned cl = cvalid ? (unsigned)(s1 ? c - C0 : c) : 0u;
    amask = 0u;
#pragma unroll
    for (int i = 0; i < Cfg::AROWS; ++i) {
      int iy, ix;
      const bool ok = cvalid & tap_coord(geom, rows[i], ky, kx, iy, ix);
      // pixel index fits 32 bits (M <= 2^31 checked on the host); dead rows read element 0 (always mapped) and are
      // zeroed at commit time -> no divergent branch around the load
      const unsigned pix = (unsigned)((rows[i].n * p.Hi + iy) * p.Wi + ix) & (0u - (unsigned)ok);
      const size_t off = (size_t)pix * sC + (cl & (0u - (unsigned)ok));
      ra[i] = *reinterpret_cast<const float4*>(sptr + off);
      amask |= (unsigned)ok << i;
    }
    // scale/shift tables, loaded unconditionally (operands without affine use the identity tables)
    const bool has_aff = sscale != nullptr;
    const float* scp = has_aff ? sscale : g_ident_scale;
    const float* shp = has_aff ? sshift : g_ident_shift;
#pragma unroll
    for (int i = 0; i < NSC; ++i) {
      const unsigned n = PER_SAMPLE ? (unsigned)(rows[i].n < 0 ? 0 : rows[i].n) : 0u;
      const size_t o = (size_t)n * (has_aff ? sss : 0u) + cl;
      rsc[i] = *reinterpret_cast<const float4*>(scp + o);
      rsh[i] = *reinterpret_cast<const float4*>(shp + o);
    }
    // weights are packed [T][Npad][Kpad] with Npad % 128 == 0 and Kpad % 32 == 0 (zero filled): no bounds checks
    const float* wt = p.w + ((size_t)t * p.Npad) * p.Kpad;
#pragma unroll
    for (int i = 0; i < Cfg::BROWS; ++i)
      rb[i] = *reinterpret_cast<const float4*>(wt + (size_t)(n0 + srow + 32 * i) * p.Kpad + c);
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

#ifdef MSEG_ABLATE
  const int abl = p.Cq >> 16;   // bit0: no global loads in the loop, bit1: no barriers, bit2: no MFMA, bit3: no commit
  if ((abl & 16) && ((blockIdx.x >> 8) & 1)) {   // bit4: stagger the second resident workgroup of a CU by ~half a K-step
    __builtin_amdgcn_s_sleep(48);
  }
  long long t_issue = 0, t_mfma = 0, t_commit = 0, t_bar = 0;   // bit5: per-phase cycle stamps (diagnostic build only)
#define STAMP(var) do { if (abl & 32) { __builtin_amdgcn_sched_barrier(0); var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); } } while (0)
  long long s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;
#else
#define STAMP(var)
#endif
  const int li = lane & 31, lh = lane >> 5;
  while (have) {
    // prefetch the next K-step (on the last step: a harmless re-read of the current one, keeps the body branch-free)
    bool have_next;
    const Pos nxt = advance(pos, have_next);
    if (have_next) pos = nxt;
    STAMP(s0);
#ifdef MSEG_ABLATE
    if (!(abl & 1))
#endif
    issue(pos.chunk, pos.t, pos.ky, pos.kx);
    STAMP(s1);

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
#ifdef MSEG_ABLATE
      if (abl & 4) { asm volatile("" :: "v"(af[0].x), "v"(bf[0].x)); continue; }
#endif
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
#ifdef MSEG_ABLATE
    STAMP(s2);
    if (!(abl & 8)) commit(An, An + BM * LDS_STRIDE);
    STAMP(s3);
    if (!(abl & 2)) __syncthreads();
    STAMP(s4);
    t_issue += s1 - s0; t_mfma += s2 - s1; t_commit += s3 - s2; t_bar += s4 - s3;
#else
    commit(An, An + BM * LDS_STRIDE);
    __syncthreads();
#endif
    cur ^= 1;
    have = have_next;
  }

#ifdef MSEG_ABLATE
  if ((abl & 32) && p.dst1 && lane == 0 && blockIdx.x < 1024) {
    float* dbg = p.dst1 + ((size_t)blockIdx.x * 4 + wave) * 4;
    dbg[0] = (float)t_issue; dbg[1] = (float)t_mfma; dbg[2] = (float)t_commit; dbg[3] = (float)t_bar;
  }
#endif
  igemm_epilogue<IgemmCfg<BM, BN>>(acc, m0, n0, wm, wn, lane, M);
}

// =====================================================================================================================
// Fast path.  Measured on MI355X (tools/ubench/mfma_valu.hip): v_mfma_f32_32x32x2_f32 does NOT co-execute with VALU
// work — an fp32-MFMA wave and a VALU wave on one SIMD take the SUM of their times (the f32 matrix op runs at the
// vector rate on shared hardware).  Every VALU instruction in the K-loop is therefore paid in MFMA throughput, and
// the staging code is written to issue as few as possible:
//   * per-row source offsets (bytes, 32 bit) and a per-row tap-validity bit mask are computed once per tile; a K-step
//     needs one v_add (wave-uniform tap delta) + bit test + select per staged row, no multiplies, no 64-bit math;
//   * loads go through buffer descriptors: 32-bit voffset, dead rows use an out-of-range offset and come back as 0;
//   * weight rows need no per-step VALU at all (constant voffset, the step offset is the scalar soffset);
//   * scale/shift tables are fetched once per 32-channel chunk, not once per tap;
//   * operands without transform (all dgrad launches) are committed to LDS untouched.
// Preconditions (checked on the host, otherwise the generic kernel above runs): CONV mode, stride-1 TCONV, or stride-2
// TCONV in parity M-order with parity classes that are whole tiles; a tile's source rows < 2 GiB; the concat boundary C0 a
// multiple of 32 (wave-uniform source selection).
// TR: 0 = plain operand, 1 = none/ReLU + affine, 2 = any activation + affine.
// SB: ONE LDS stage instead of two (an extra barrier per K-step, half the LDS: three workgroups per CU instead of two).
// Chosen for launches of at most 8 K-steps (ConvTranspose as a 1x1 GEMM, the data gradients of the level-0 stride-2 /
// transposed convolutions): such a tile is prologue, first fetch and epilogue more than K-loop, and what hides those is the
// number of co-resident workgroups, not the overlap inside one.
template <int BM, int BN, int TR, bool PER_SAMPLE, bool SB = false>
__global__ __launch_bounds__(256) void igemm_fast_kernel(const MsegIgemm p) {
  using Cfg = IgemmCfg<BM, BN>;
  constexpr int STAGE = (BM + BN) * LDS_STRIDE;
  __shared__ __attribute__((aligned(16))) float lds[(SB ? 1 : 2) * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const int M = p.NB * p.Ho * p.Wo;
  const int ntiles_n = (p.Ngemm + BN - 1) / BN;
  const int lid = (int)xcd_logical_id(blockIdx.x, gridDim.x);   // N tiles of one M tile -> same XCD (share the A slab)
  const int tile_m = lid / ntiles_n;
  const int tile_n = lid - tile_m * ntiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int T = p.KH * p.KW;
  const int srow = tid >> 3, scol = tid & 7;

  // ---- per-row geometry, once per tile ------------------------------------------------------------------------
  const TapGeom geom = make_geom(p);
  const long long tile_px = tile_base_pixel(p, geom, m0, M);
  int pix0[Cfg::AROWS];            // source pixel index of tap (0,0); meaningful only where a tap is live
  int rown[Cfg::AROWS];            // image index (per-sample tables)
  unsigned vmask[Cfg::AROWS];      // bit t set <=> tap t of this row reads a real source pixel
#pragma unroll
  for (int i = 0; i < Cfg::AROWS; ++i) {
    const RowInfo r = decode_row(p, m0 + srow + 32 * i, M);
    // source coordinate of a live tap = (o * sm - dir * pad) >> sh  +  dir * (k >> sh): linear in the tap for every
    // row of the tile (stride-2 TCONV: the parity M-order makes the live-tap set tile-uniform, see `taplist`)
    const int iy0 = (r.oy * geom.sm - geom.dir * geom.pad) >> geom.sh, ix0 = (r.ox * geom.sm - geom.dir * geom.pad) >> geom.sh;
    pix0[i] = (int)(((long long)r.n * p.Hi + iy0) * p.Wi + ix0 - tile_px);   // relative to the tile's descriptor base
    rown[i] = r.n < 0 ? 0 : r.n;
    unsigned mk = 0u;
    for (int t = 0; t < T; ++t) {
      const int ky = t / p.KW, kx = t - ky * p.KW;
      int iy, ix;
      mk |= (unsigned)tap_coord(geom, r, ky, kx, iy, ix) << t;
    }
    vmask[i] = mk;
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
  const unsigned OOB = 0x80000000u;   // >= num_records of every descriptor

  // wave-uniform buffer descriptors (kernarg-derived only)
  // based at the tile's first source pixel; every load is masked by vmask (dead rows use OOB), so the record count only
  // has to exceed the tile's span (host-checked to stay below 2 GiB)
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.src[0].ptr + tile_px * p.src[0].C), 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>((p.nsrc > 1 ? p.src[1].ptr : p.src[0].ptr) + tile_px * (p.nsrc > 1 ? p.src[1].C : p.src[0].C)), 0,
      0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0,
                                                                        T * p.Npad * p.Kpad * 4, 0x00020000);
  unsigned wvoff[Cfg::BROWS];
#pragma unroll
  for (int i = 0; i < Cfg::BROWS; ++i) wvoff[i] = ((unsigned)(n0 + srow + 32 * i) * (unsigned)p.Kpad + scol * 4u) * 4u;

  // ---- staging state ---------------------------------------------------------------------------------------------
  constexpr int NSC = (TR == 0) ? 1 : (PER_SAMPLE ? Cfg::AROWS : 1);
  float4 ra[Cfg::AROWS], rb[Cfg::BROWS], rsc[NSC], rsh[NSC];
  float rm[Cfg::AROWS];              // 1.0 for live rows, 0.0 for padding / tail rows (TR != 0)
  unsigned rowoff[Cfg::AROWS];       // byte offset of (tap (0,0), this thread's channel quad) in the current source
  int ract = 0;
  int cur_chunk = -1;
  bool cur_s1 = false;

  auto issue = [&](int chunk, int t, int ky, int kx) {
    const int c = chunk * KC + scol * 4;
    if (chunk != cur_chunk) {        // wave-uniform: new 32-channel chunk -> source select, row offsets, tables
      cur_chunk = chunk;
      cur_s1 = (p.nsrc > 1) && (chunk * KC >= C0);
      const unsigned sC4 = (unsigned)(cur_s1 ? p.src[1].C : p.src[0].C) * 4u;
      const unsigned cl4 = (unsigned)(cur_s1 ? c - C0 : c) * 4u;
#pragma unroll
      for (int i = 0; i < Cfg::AROWS; ++i) rowoff[i] = (unsigned)pix0[i] * sC4 + cl4;
      if (TR != 0) {
        const MsegSrc& s = cur_s1 ? p.src[1] : p.src[0];
        ract = s.act;
        const bool has_aff = s.scale != nullptr;
        const float* scp = has_aff ? s.scale : g_ident_scale;
        const float* shp = has_aff ? s.shift : g_ident_shift;
        const unsigned cl = (c < p.Cin) ? (cl4 >> 2) : 0u;
#pragma unroll
        for (int i = 0; i < NSC; ++i) {
          const size_t o = (size_t)(PER_SAMPLE ? rown[i] : 0) * (has_aff ? (unsigned)s.ss : 0u) + cl;
          rsc[i] = *reinterpret_cast<const float4*>(scp + o);
          rsh[i] = *reinterpret_cast<const float4*>(shp + o);
        }
      }
    }
    const unsigned cbit = (c < p.Cin) ? (1u << t) : 0u;                      // channel tail of the last chunk
    const unsigned sC4 = (unsigned)(cur_s1 ? p.src[1].C : p.src[0].C) * 4u;   // scalar
    const unsigned delta = (unsigned)(geom.dir * ((ky >> geom.sh) * p.Wi + (kx >> geom.sh))) * sC4;   // scalar, wraps by design
#pragma unroll
    for (int i = 0; i < Cfg::AROWS; ++i) {
      const bool ok = (vmask[i] & cbit) != 0u;
      const unsigned vo = ok ? rowoff[i] + delta : OOB;
      const f32x4 v = cur_s1 ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, vo, 0, 0))
                             : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs0, vo, 0, 0));
      ra[i] = make_float4(v[0], v[1], v[2], v[3]);
      if (TR != 0) rm[i] = ok ? 1.f : 0.f;
    }
    const unsigned wso = ((unsigned)t * (unsigned)p.Npad * (unsigned)p.Kpad + (unsigned)chunk * KC) * 4u;   // scalar
#pragma unroll
    for (int i = 0; i < Cfg::BROWS; ++i) {
      const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, wvoff[i], wso, 0));
      rb[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
  };

  auto commit = [&](float* As, float* Bs) {
    const float lo = (ract == MSEG_ACT_RELU) ? 0.f : -3.402823466e38f;
#pragma unroll
    for (int i = 0; i < Cfg::AROWS; ++i) {
      float4 v = ra[i];
      if (TR != 0) {
        if (TR == 2) {
          v = act_fwd4(v, ract);
        } else {
          v.x = clamp_lo(v.x, lo); v.y = clamp_lo(v.y, lo); v.z = clamp_lo(v.z, lo); v.w = clamp_lo(v.w, lo);
        }
        const float4 sc = rsc[PER_SAMPLE ? i : 0], sh = rsh[PER_SAMPLE ? i : 0];
        const float m = rm[i];
        v.x = (v.x * sc.x + sh.x) * m; v.y = (v.y * sc.y + sh.y) * m;
        v.z = (v.z * sc.z + sh.z) * m; v.w = (v.w * sc.w + sh.w) * m;
      }
      *reinterpret_cast<float4*>(As + (srow + 32 * i) * LDS_STRIDE + scol * 4) = v;
    }
#pragma unroll
    for (int i = 0; i < Cfg::BROWS; ++i)
      *reinterpret_cast<float4*>(Bs + (srow + 32 * i) * LDS_STRIDE + scol * 4) = rb[i];
  };

  // K-step iterator: chunk outer, live tap inner.  Every tap is live for at least one row in CONV / stride-1 TCONV; a
  // stride-2 TCONV tile (parity M-order, class size a multiple of BM: host-checked) has the taps whose phase matches
  // the tile's output parity.  The live taps are kept as a packed scalar list: 8 bits per entry = t | ky << 4 | kx << 6.
  unsigned long long taplist = 0ull, taplist_hi = 0ull;
  int nlive = 0;
  {
    int py = 0, px = 0;
    if (geom.sh) {
      const int cls = m0 / (M >> 2);
      py = cls >> 1; px = cls & 1;
    }
    for (int tt = 0; tt < T; ++tt) {
      const int ky_ = tt / p.KW, kx_ = tt - ky_ * p.KW;
      const bool live = (((py - geom.dir * (ky_ - geom.pad)) | (px - geom.dir * (kx_ - geom.pad))) & geom.sh) == 0;
      if (live) {
        const unsigned long long e = (unsigned long long)(tt | (ky_ << 4) | (kx_ << 6));
        if (nlive < 8) taplist |= e << (8 * nlive); else taplist_hi |= e << (8 * (nlive - 8));
        ++nlive;
      }
    }
  }
  auto tap_at = [&](int j) -> unsigned { return (unsigned)((j < 8 ? taplist >> (8 * j) : taplist_hi >> (8 * (j - 8))) & 0xffull); };
  int chunk = 0, j = 0;
  const int nsteps = nchunks * nlive;
  unsigned e0 = tap_at(0);
  issue(0, (int)(e0 & 15u), (int)((e0 >> 4) & 3u), (int)(e0 >> 6));
  commit(lds, lds + BM * LDS_STRIDE);
  __syncthreads();
  int cur = 0;
  const int li = lane & 31, lh = lane >> 5;
  for (int step = 0; step < nsteps; ++step) {
    // prefetch the next K-step (the last iteration re-reads the current one: keeps the body branch-free)
    if (step + 1 < nsteps) {
      j += 1;
      if (j >= nlive) { j = 0; ++chunk; }
    }
    e0 = tap_at(j);
    issue(chunk, (int)(e0 & 15u), (int)((e0 >> 4) & 3u), (int)(e0 >> 6));

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
    if (SB) {
      __syncthreads();                                   // every wave is done reading the only stage
      commit(lds, lds + BM * LDS_STRIDE);
      __syncthreads();
    } else {
      float* An = lds + (cur ^ 1) * STAGE;
      commit(An, An + BM * LDS_STRIDE);
      __syncthreads();
      cur ^= 1;
    }
  }
  igemm_epilogue<IgemmCfg<BM, BN>>(acc, m0, n0, wm, wn, lane, M);
}

// =====================================================================================================================
// Halo kernel for the 3x3 stride-1 convolutions and their data gradients (the bulk of the FLOPs).  A workgroup owns a
// TH x TW pixel rectangle (TH * TW = 128, TW = largest power of two <= 64 dividing the row length) of one image and
// BN output channels.  Per 32-channel chunk the (TH+2) x (TW+2) input halo is loaded, normalised and written to LDS
// ONCE; the nine taps then read their MFMA A-fragments from shifted rows of that slab (per-lane base + wave-uniform
// tap offset).  Compared with the gather kernels this removes 8/9 of the operand loads and of the norm-on-load VALU
// work — which matters because the fp32 matrix op shares the SIMD with VALU (see igemm_fast_kernel).
// LDS: ONE halo buffer (pixel tiles are at most 64 wide: (2 + 2) x (64 + 2) = 264 rows) + a double-buffered weight slab,
// 36-float rows: 73 KiB (BN = 128) -> TWO workgroups of 8 waves per CU.  The halo of chunk c+1 is fetched into
// registers during chunk c and written after its last tap (one extra barrier per chunk); whatever a workgroup cannot
// overlap itself (first fetch, epilogue stores and read-modify-writes, barrier skew) is covered by the MFMAs of its
// co-resident neighbour.
// A tile never spans two images, so per-sample (Group/InstanceNorm) tables need no special case.
template <int BN>
struct HaloCfg {
  static constexpr int WN = (BN >= 128) ? 4 : 2;
  static constexpr int WM = 8 / WN;
  static constexpr int TM = 128 / WM, TN = BN / WN;
  static constexpr int MB = TM / 32, NB = TN / 32;
  static constexpr int BROWS = BN / 64;
};

// Split-K (small batches / deep levels: fewer than one tile per workgroup slot): ksplit workgroups share a tile, each
// reducing a contiguous range of the 32-channel chunks into its own slice of a workspace [ksplit][M][Ngemm] (the host
// passes a descriptor whose destination is that workspace); igemm_splitk_reduce_kernel sums the slices in fixed order and
// applies bias / accumulate / the split destinations.  ksplit == 1: the plain kernel.
template <int BN, int TR>
__global__ __launch_bounds__(512, 2) void igemm_halo_kernel(const MsegIgemm p, int tw_log2, int ksplit,
                                                            int chunks_per_split) {
  constexpr int BM = 128;
  // 8 waves (two per SIMD, same workgroup): 2 x 4 wave grid for BN = 128 (64 x 32 per wave), 4 x 2 for BN = 64 (32 x 32)
  using Cfg = HaloCfg<BN>;
  constexpr int HMAX = 264;                          // (2 + 2) x (64 + 2), the largest halo (tw_log2 <= 6)
  constexpr int HL = (HMAX * 8 + 511) / 512;         // float4 per thread per chunk (5)
  constexpr int ASTAGE = HMAX * LDS_STRIDE;
  constexpr int BSTAGE = BN * LDS_STRIDE;
  __shared__ __attribute__((aligned(16))) float lds[ASTAGE + 2 * BSTAGE];
  float* const Abuf = lds;
  float* const Bbuf = lds + ASTAGE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const int li = lane & 31, lh = lane >> 5;
  const int srow = tid >> 3, scol = tid & 7;
  const int TW = 1 << tw_log2, TH = BM >> tw_log2, HW2 = TW + 2;
  const int HROWS = (TH + 2) * HW2;
  const int H = p.Hi, W = p.Wi;                      // stride 1, pad 1: output and input have the same size
  const int M = p.NB * H * W;
  const int tiles_x = W >> tw_log2, tiles_y = (H + TH - 1) / TH;
  const int ntiles_n = (p.Ngemm + BN - 1) / BN;
  const int lid_all = (int)xcd_logical_id(blockIdx.x, gridDim.x);
  const int ntiles_all = (int)gridDim.x / ksplit;
  const int kz = lid_all / ntiles_all;                 // split index (slowest: the N tiles of an M tile stay neighbours)
  const int lid = lid_all - kz * ntiles_all;
  const int tile_m = lid / ntiles_n, tile_n = lid - tile_m * ntiles_n;
  const int img = tile_m / (tiles_x * tiles_y);
  const int trem = tile_m - img * (tiles_x * tiles_y);
  const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
  const int oy0 = ty * TH, ox0 = tx * TW, n0 = tile_n * BN;
  // buffer descriptors span only the row band of this tile's halo (image rows band0 .. band0 + band_rows - 1): 32-bit
  // offsets then never see more than (TH + 2) x W pixels, so neither the batch size nor the FRAME size is limited
  // (8192 x 8192 frames, the reference's largest tested shape, stay on this kernel)
  const int band0 = oy0 > 0 ? oy0 - 1 : 0;
  const int band_rows = (oy0 + TH + 1 < H ? oy0 + TH + 1 : H) - band0;
  const bool conv = p.mode == MSEG_MODE_CONV;       // TCONV (data gradient) = the same halo with the taps mirrored

  // ---- halo entries of this thread: rows hrow = srow + 32 j, channel quad scol ----------------------------------
  int hpix[HL];
  unsigned hvalid = 0u;
#pragma unroll
  for (int j = 0; j < HL; ++j) {
    const int hrow = srow + 64 * j;
    const int hy = hrow / HW2, hx = hrow - hy * HW2;
    const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
    const bool ok = (hrow < HROWS) & (iy >= 0) & (iy < H) & (ix >= 0) & (ix < W);
    hpix[j] = (iy - band0) * W + ix;                 // pixel index inside the tile's row band (see the descriptors)
    hvalid |= (unsigned)ok << j;
  }
  // per-lane LDS row of the MFMA A rows (tile pixel -> halo coordinates of tap (0,0))
  int abase[Cfg::MB];
#pragma unroll
  for (int a = 0; a < Cfg::MB; ++a) {
    const int i = wm * Cfg::TM + a * 32 + li;
    abase[a] = ((i >> tw_log2) * HW2 + (i & (TW - 1))) * LDS_STRIDE + lh * 4;
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
  const unsigned OOB = 0x80000000u;
  const int C1 = p.nsrc > 1 ? p.src[1].C : p.src[0].C;
  const float* const base0 = p.src[0].ptr + ((size_t)img * H + band0) * W * p.src[0].C;
  const float* const base1 = (p.nsrc > 1 ? p.src[1].ptr : p.src[0].ptr) + ((size_t)img * H + band0) * W * C1;
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base0), 0,
                                                                        band_rows * W * p.src[0].C * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base1), 0,
                                                                        band_rows * W * C1 * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0,
                                                                        9 * p.Npad * p.Kpad * 4, 0x00020000);
  unsigned wvoff[Cfg::BROWS];
#pragma unroll
  for (int i = 0; i < Cfg::BROWS; ++i) wvoff[i] = ((unsigned)(n0 + srow + 64 * i) * (unsigned)p.Kpad + scol * 4u) * 4u;

  float4 rh[HL], rb[Cfg::BROWS], rsc, rsh;
  unsigned hlive = 0u;       // validity of the halo registers currently in flight (hvalid & channel tail)
  int ract = 0;
  bool cur_s1 = false;

  auto issue_halo = [&](int chunk) {
    const int c = chunk * KC + scol * 4;
    cur_s1 = (p.nsrc > 1) && (chunk * KC >= C0);
    const MsegSrc& s = cur_s1 ? p.src[1] : p.src[0];
    const unsigned sC4 = (unsigned)s.C * 4u;                                     // scalar
    const unsigned soff = (unsigned)(chunk * KC - (cur_s1 ? C0 : 0)) * 4u + scol * 16u;
    hlive = (c < p.Cin) ? hvalid : 0u;
#pragma unroll
    for (int j = 0; j < HL; ++j) {
      const bool ok = (hlive >> j) & 1u;
      const unsigned vo = ok ? (unsigned)hpix[j] * sC4 + soff : OOB;   // one v_mad per load, once per chunk: keeps the
                                                                       // kernel within 128 VGPRs (4 waves per SIMD)
      const f32x4 v = cur_s1 ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, vo, 0, 0))
                             : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs0, vo, 0, 0));
      rh[j] = make_float4(v[0], v[1], v[2], v[3]);
    }
    if (TR != 0) {
      ract = s.act;
      const bool has_aff = s.scale != nullptr;
      const float* scp = has_aff ? s.scale : g_ident_scale;
      const float* shp = has_aff ? s.shift : g_ident_shift;
      const unsigned cl = (c < p.Cin) ? (unsigned)(cur_s1 ? c - C0 : c) : 0u;
      const size_t o = (size_t)img * (has_aff ? (unsigned)s.ss : 0u) + cl;
      rsc = *reinterpret_cast<const float4*>(scp + o);
      rsh = *reinterpret_cast<const float4*>(shp + o);
    }
  };

  auto commit_halo = [&](float* As) {
    const float lo = (ract == MSEG_ACT_RELU) ? 0.f : -3.402823466e38f;
#pragma unroll
    for (int j = 0; j < HL; ++j) {
      const int hrow = srow + 64 * j;
      if (hrow < HMAX) {
        float4 v = rh[j];
        if (TR != 0) {
          if (TR == 2) v = act_fwd4(v, ract);
          else { v.x = clamp_lo(v.x, lo); v.y = clamp_lo(v.y, lo); v.z = clamp_lo(v.z, lo); v.w = clamp_lo(v.w, lo); }
          const float m = ((hlive >> j) & 1u) ? 1.f : 0.f;
          v.x = (v.x * rsc.x + rsh.x) * m; v.y = (v.y * rsc.y + rsh.y) * m;
          v.z = (v.z * rsc.z + rsh.z) * m; v.w = (v.w * rsc.w + rsh.w) * m;
        }
        *reinterpret_cast<float4*>(As + hrow * LDS_STRIDE + scol * 4) = v;
      }
    }
  };

  auto issue_b = [&](int chunk, int t) {
    const unsigned wso = ((unsigned)t * (unsigned)p.Npad * (unsigned)p.Kpad + (unsigned)chunk * KC) * 4u;
#pragma unroll
    for (int i = 0; i < Cfg::BROWS; ++i) {
      const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, wvoff[i], wso, 0));
      rb[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
  };
  auto commit_b = [&](float* Bs) {
#pragma unroll
    for (int i = 0; i < Cfg::BROWS; ++i)
      *reinterpret_cast<float4*>(Bs + (srow + 64 * i) * LDS_STRIDE + scol * 4) = rb[i];
  };

  const int c_begin = kz * chunks_per_split;
  const int c_end = (c_begin + chunks_per_split < nchunks) ? c_begin + chunks_per_split : nchunks;
  issue_halo(c_begin);
  issue_b(c_begin, 0);
  commit_halo(Abuf);
  commit_b(Bbuf);
  __syncthreads();

  int bsel = 0;
  for (int chunk = c_begin; chunk < c_end; ++chunk) {
    const float* As = Abuf;
    const bool more_chunks = chunk + 1 < c_end;
    for (int t = 0; t < 9; ++t) {
      if (t == 0 && more_chunks) issue_halo(chunk + 1);          // a whole chunk of MFMAs hides this fetch
      // weights of the next K-step (the very last step re-reads its own: keeps the body branch-free)
      const bool last = (t == 8) && !more_chunks;
      issue_b(t == 8 ? (more_chunks ? chunk + 1 : chunk) : chunk, last ? 8 : (t == 8 ? 0 : t + 1));

      const int ky = t / 3, kx = t - 3 * ky;
      const int toff = ((conv ? ky : 2 - ky) * HW2 + (conv ? kx : 2 - kx)) * LDS_STRIDE;   // scalar
      const float* Bs = Bbuf + bsel * BSTAGE;
#pragma unroll
      for (int kk = 0; kk < KC / 8; ++kk) {
        float4 af[Cfg::MB], bf[Cfg::NB];
#pragma unroll
        for (int a = 0; a < Cfg::MB; ++a)
          af[a] = *reinterpret_cast<const float4*>(As + abase[a] + toff + kk * 8);
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
      commit_b(Bbuf + (bsel ^ 1) * BSTAGE);
      if (t == 8 && more_chunks) {
        __syncthreads();                 // every wave is done reading this chunk's halo
        commit_halo(Abuf);
      }
      __syncthreads();
      bsel ^= 1;
    }
  }
  // split slices are addressed as further images of the workspace tensor [ksplit * NB][H][W][Ngemm]
  igemm_epilogue<Cfg>(acc, 0, n0, wm, wn, lane, M * ksplit, tw_log2, img + kz * p.NB, oy0, ox0);
}

// ---- bf16 variant of the halo kernel (BASELINE configs[2]: bf16 forward / backward, fp32 accumulate) ------------------
// Same tiling, operands and epilogue as igemm_halo_kernel; the differences:
//   * the (normalised, activated) source pixels are rounded to bf16 (RNE, v_cvt_pk_bf16_f32) while they are staged into
//     LDS, the packed weights arrive as bf16 (mseg_f32_to_bf16 of the packed fp32 tensor); storage in HBM stays fp32;
//   * v_mfma_f32_32x32x16_bf16: one instruction consumes 16 input channels (fp32 accumulate; same C layout as 32x32x2);
//   * LDS rows are 32 bf16 = 64 B + 16 B pad (80 B stride: conflict-free ds_read_b128), tiles are at most 32 pixels wide
//     (halo <= 204 rows), and the weights are staged one kernel ROW (3 taps) at a time, double buffered: 76 KiB for
//     BN = 128 -> two workgroups per CU, one barrier per 12 MFMAs of a wave instead of one per tap.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#define HB_STRIDE 40          // bf16 elements per LDS row

// S16: the sources are bf16 tensors (MSEG_ST_BF16).  A thread then stages 8 channels per 16-byte load (half the loads and LDS
// writes of the fp32-source form); a plain operand (TR = 0) goes to LDS as it arrives.
// BM: pixels per tile.  128 everywhere except the 128 -> 64 channel layers of level 0 (igemm_halo_bf16m512_kernel below).
template <int BN, int BM>
struct HaloCfgM {
  static constexpr int WN = (BN >= 128) ? 4 : 2;
  static constexpr int WM = 8 / WN;
  static constexpr int TM = BM / WM, TN = BN / WN;
  static constexpr int MB = TM / 32, NB = TN / 32;
  static constexpr int BROWS = BN / 64;
};

template <int BN, int TR, bool S16, int BM>
__device__ __forceinline__ void igemm_halo_bf16_body(const MsegIgemm& p, int tw_log2, int ksplit, int chunks_per_split,
                                                     int m_fastest) {
  using Cfg = HaloCfgM<BN, BM>;
  // largest halo: BM = 128: (4 + 2) x (32 + 2) (tw_log2 <= 5); BM = 512: (16 + 2) x (32 + 2) = (32 + 2) x (16 + 2) (tw_log2 4, 5)
  constexpr int HMAX = BM == 128 ? 204 : 612;
  constexpr int SQ = S16 ? 4 : 8;                    // staging threads per halo row (8 / 4 channels each)
  constexpr int SROWS = 512 / SQ;                    // halo rows per staging pass
  constexpr int HL = (HMAX + SROWS - 1) / SROWS;     // 16-byte loads per thread per chunk (4; 2 for bf16 sources)
  constexpr int ASTAGE = HMAX * HB_STRIDE;
  constexpr int BSTAGE = 3 * BN * HB_STRIDE;
  __shared__ __attribute__((aligned(16))) __bf16 lds[ASTAGE + 2 * BSTAGE];
  __bf16* const Abuf = lds;
  __bf16* const Bbuf = lds + ASTAGE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const int li = lane & 31, lh = lane >> 5;
  const int srow = tid / SQ, scol = tid % SQ;        // halo staging: SROWS rows x SQ channel groups per pass
  const int brow = tid >> 2, bcol = tid & 3;         // weight staging: 128 rows x 4 groups of 8 channels
  const int TW = 1 << tw_log2, TH = BM >> tw_log2, HW2 = TW + 2;
  const int HROWS = (TH + 2) * HW2;
  const int H = p.Hi, W = p.Wi;
  const int M = p.NB * H * W;
  const int tiles_x = W >> tw_log2, tiles_y = (H + TH - 1) / TH;
  const int ntiles_n = (p.Ngemm + BN - 1) / BN;
  const int lid_all = (int)xcd_logical_id(blockIdx.x, gridDim.x);
  const int ntiles_all = (int)gridDim.x / ksplit;
  const int kz = lid_all / ntiles_all;
  const int lid = lid_all - kz * ntiles_all;
  // Workgroup order inside an XCD (neighbouring logical ids share an L2).  N tiles fastest: the N tiles of a pixel tile
  // run together and share its halo.  With many input channels the WEIGHTS are the larger stream (9 x Cin x BN bf16 per
  // tile, all of it re-read by every pixel tile): pixel tiles fastest then keeps ONE column of weight tiles (<= 2.4 MB)
  // resident in each XCD's 4 MB L2 instead of cycling the whole weight tensor through it.
  const int ntiles_m = ntiles_all / ntiles_n;
  const int tile_m = m_fastest ? lid % ntiles_m : lid / ntiles_n;
  const int tile_n = m_fastest ? lid / ntiles_m : lid - tile_m * ntiles_n;
  const int img = tile_m / (tiles_x * tiles_y);
  const int trem = tile_m - img * (tiles_x * tiles_y);
  const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
  const int oy0 = ty * TH, ox0 = tx * TW, n0 = tile_n * BN;
  // buffer descriptors span only the row band of this tile's halo (image rows band0 .. band0 + band_rows - 1): 32-bit
  // offsets then never see more than (TH + 2) x W pixels, so neither the batch size nor the FRAME size is limited
  // (8192 x 8192 frames, the reference's largest tested shape, stay on this kernel)
  const int band0 = oy0 > 0 ? oy0 - 1 : 0;
  const int band_rows = (oy0 + TH + 1 < H ? oy0 + TH + 1 : H) - band0;
  const bool conv = p.mode == MSEG_MODE_CONV;

  int hpix[HL];
  unsigned hvalid = 0u;
#pragma unroll
  for (int j = 0; j < HL; ++j) {
    const int hrow = srow + SROWS * j;
    const int hy = hrow / HW2, hx = hrow - hy * HW2;
    const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
    const bool ok = (hrow < HROWS) & (iy >= 0) & (iy < H) & (ix >= 0) & (ix < W);
    hpix[j] = (iy - band0) * W + ix;
    hvalid |= (unsigned)ok << j;
  }
  int abase[Cfg::MB];
#pragma unroll
  for (int a = 0; a < Cfg::MB; ++a) {
    const int i = wm * Cfg::TM + a * 32 + li;
    abase[a] = ((i >> tw_log2) * HW2 + (i & (TW - 1))) * HB_STRIDE + lh * 8;
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
  const unsigned OOB = 0x80000000u;
  const int C1 = p.nsrc > 1 ? p.src[1].C : p.src[0].C;
  constexpr int ESZ = S16 ? 2 : 4;                   // bytes per source element
  const char* const base0 = (const char*)p.src[0].ptr + ((size_t)img * H + band0) * W * p.src[0].C * ESZ;
  const char* const base1 = (const char*)(p.nsrc > 1 ? p.src[1].ptr : p.src[0].ptr) + ((size_t)img * H + band0) * W * C1 * ESZ;
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base0), 0,
                                                                        band_rows * W * p.src[0].C * ESZ, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base1), 0,
                                                                        band_rows * W * C1 * ESZ, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0,
                                                                        9 * p.Npad * p.Kpad * 2, 0x00020000);
  const bool bact = brow < BN;
  const unsigned wvoff = bact ? ((unsigned)(n0 + brow) * (unsigned)p.Kpad + bcol * 8u) * 2u : OOB;

  f32x4 rh[HL];                                      // 16 raw bytes: 4 fp32 or 8 bf16 source channels
  float4 rsc, rsh, rsc2, rsh2;                       // norm-on-load tables of this thread's 4 (8) channels
  f32x4 rb[3];
  unsigned hlive = 0u;
  int ract = 0;
  bool cur_s1 = false;

  auto issue_halo = [&](int chunk) {
    constexpr int CPT = S16 ? 8 : 4;                 // channels per staging thread
    const int c = chunk * KC + scol * CPT;
    cur_s1 = (p.nsrc > 1) && (chunk * KC >= C0);
    const MsegSrc& s = cur_s1 ? p.src[1] : p.src[0];
    const unsigned sCB = (unsigned)s.C * (unsigned)ESZ;                            // bytes per pixel
    const unsigned soff = (unsigned)(chunk * KC - (cur_s1 ? C0 : 0)) * (unsigned)ESZ + scol * 16u;
    hlive = (c < p.Cin) ? hvalid : 0u;
#pragma unroll
    for (int j = 0; j < HL; ++j) {
      const bool ok = (hlive >> j) & 1u;
      const unsigned vo = ok ? (unsigned)hpix[j] * sCB + soff : OOB;
      rh[j] = cur_s1 ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, vo, 0, 0))
                     : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs0, vo, 0, 0));
    }
    if (TR != 0) {
      ract = s.act;
      const bool has_aff = s.scale != nullptr;
      const float* scp = has_aff ? s.scale : g_ident_scale;
      const float* shp = has_aff ? s.shift : g_ident_shift;
      const unsigned cl = (c < p.Cin) ? (unsigned)(cur_s1 ? c - C0 : c) : 0u;
      const size_t o = (size_t)img * (has_aff ? (unsigned)s.ss : 0u) + cl;
      rsc = *reinterpret_cast<const float4*>(scp + o);
      rsh = *reinterpret_cast<const float4*>(shp + o);
      if (S16) {                                     // channels 4..7 of this thread (the identity tables are per channel too)
        rsc2 = *reinterpret_cast<const float4*>(scp + o + 4);
        rsh2 = *reinterpret_cast<const float4*>(shp + o + 4);
      }
    }
  };

  auto xform4 = [&](float4 v, const float4& sc, const float4& sh, float lo, float m) -> float4 {
    if (TR == 2) v = act_fwd4(v, ract);
    else { v.x = clamp_lo(v.x, lo); v.y = clamp_lo(v.y, lo); v.z = clamp_lo(v.z, lo); v.w = clamp_lo(v.w, lo); }
    v.x = (v.x * sc.x + sh.x) * m; v.y = (v.y * sc.y + sh.y) * m;
    v.z = (v.z * sc.z + sh.z) * m; v.w = (v.w * sc.w + sh.w) * m;
    return v;
  };

  auto commit_halo = [&](__bf16* As) {
    const float lo = (ract == MSEG_ACT_RELU) ? 0.f : -3.402823466e38f;
#pragma unroll
    for (int j = 0; j < HL; ++j) {
      const int hrow = srow + SROWS * j;
      if (hrow < HMAX) {
        const float m = ((hlive >> j) & 1u) ? 1.f : 0.f;
        if (S16) {
          const uint4 raw = __builtin_bit_cast(uint4, rh[j]);
          if (TR == 0) {                             // plain bf16 operand: already in LDS format
            *reinterpret_cast<uint4*>(As + hrow * HB_STRIDE + scol * 8) = raw;
          } else if (TR == 1) {                       // ReLU / none + affine: the short form (bf16_affine.h)
            *reinterpret_cast<uint4*>(As + hrow * HB_STRIDE + scol * 8) =
                mseg_affine8_bf16(raw, ract == MSEG_ACT_RELU ? 0u : 0x80008000u, rsc, rsh, rsc2, rsh2, m != 0.f);
          } else {
            const float4 a = xform4(bf16x4_to_f32(make_uint2(raw.x, raw.y)), rsc, rsh, lo, m);
            const float4 b = xform4(bf16x4_to_f32(make_uint2(raw.z, raw.w)), rsc2, rsh2, lo, m);
            const uint2 pa = f32x4_to_bf16(a), pb = f32x4_to_bf16(b);
            *reinterpret_cast<uint4*>(As + hrow * HB_STRIDE + scol * 8) = make_uint4(pa.x, pa.y, pb.x, pb.y);
          }
        } else {
          float4 v = make_float4(rh[j][0], rh[j][1], rh[j][2], rh[j][3]);
          if (TR != 0) v = xform4(v, rsc, rsh, lo, m);
          bf16x4 h;
          h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
          *reinterpret_cast<bf16x4*>(As + hrow * HB_STRIDE + scol * 4) = h;
        }
      }
    }
  };

  // weights of kernel row ky of a chunk: 3 taps x [BN][32] bf16
  auto issue_b = [&](int chunk, int ky) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const unsigned wso = ((unsigned)(ky * 3 + kx) * (unsigned)p.Npad * (unsigned)p.Kpad + (unsigned)chunk * KC) * 2u;
      rb[kx] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, wvoff, wso, 0));
    }
  };
  auto commit_b = [&](__bf16* Bs) {
    if (bact) {
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
        *reinterpret_cast<f32x4*>(Bs + (kx * BN + brow) * HB_STRIDE + bcol * 8) = rb[kx];
    }
  };

  const int c_begin = kz * chunks_per_split;
  const int c_end = (c_begin + chunks_per_split < nchunks) ? c_begin + chunks_per_split : nchunks;
  issue_halo(c_begin);
  issue_b(c_begin, 0);
  commit_halo(Abuf);
  commit_b(Bbuf);
  __syncthreads();

  int bsel = 0;
  for (int chunk = c_begin; chunk < c_end; ++chunk) {
    const __bf16* As = Abuf;
    const bool more_chunks = chunk + 1 < c_end;
    for (int ky = 0; ky < 3; ++ky) {
      if (ky == 0 && more_chunks) issue_halo(chunk + 1);
      // weights of the next stage (the very last stage re-reads its own: keeps the body branch-free)
      issue_b(ky == 2 ? (more_chunks ? chunk + 1 : chunk) : chunk, ky == 2 ? (more_chunks ? 0 : 2) : ky + 1);
      const __bf16* Bs = Bbuf + bsel * BSTAGE;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int toff = ((conv ? ky : 2 - ky) * HW2 + (conv ? kx : 2 - kx)) * HB_STRIDE;   // scalar
#pragma unroll
        for (int kk = 0; kk < KC / 16; ++kk) {
          bf16x8 af[Cfg::MB], bf[Cfg::NB];
#pragma unroll
          for (int a = 0; a < Cfg::MB; ++a)
            af[a] = *reinterpret_cast<const bf16x8*>(As + abase[a] + toff + kk * 16);
#pragma unroll
          for (int b = 0; b < Cfg::NB; ++b)
            bf[b] = *reinterpret_cast<const bf16x8*>(Bs + (kx * BN + wn * Cfg::TN + b * 32 + li) * HB_STRIDE + kk * 16 +
                                                     lh * 8);
#pragma unroll
          for (int a = 0; a < Cfg::MB; ++a)
#pragma unroll
            for (int b = 0; b < Cfg::NB; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
      }
      commit_b(Bbuf + (bsel ^ 1) * BSTAGE);
      if (ky == 2 && more_chunks) {
        __syncthreads();                 // every wave is done reading this chunk's halo
        commit_halo(Abuf);
      }
      __syncthreads();
      bsel ^= 1;
    }
  }
  igemm_epilogue<Cfg>(acc, 0, n0, wm, wn, lane, M * ksplit, tw_log2, img + kz * p.NB, oy0, ox0);
}

template <int BN, int TR, bool S16>
__global__ __launch_bounds__(512, BN == 64 ? 3 : 2) void igemm_halo_bf16_kernel(const MsegIgemm p, int tw_log2, int ksplit,
                                                                 int chunks_per_split, int m_fastest) {
  igemm_halo_bf16_body<BN, TR, S16, 128>(p, tw_log2, ksplit, chunks_per_split, m_fastest);
}

// 512-pixel tiles for layers with few output channels and many input channels (128 -> 64 at level 0: the concat convolutions
// of the decoders).  Their weights (147 KB) cannot stay resident like those of the 64 -> 64 layers, and with 128-pixel tiles
// every tile re-stages all of them — 5.6 x the bytes of its halo.  Four pixel tiles per weight stage: a quarter of the
// weight traffic into LDS, a wave owns 128 pixels x 32 channels (1.25 LDS reads per MFMA instead of 2), the halo shrinks
// from 1.59 to 1.20 input rows per output row.  80 KB of LDS: two workgroups per CU.
template <int TR, bool S16>
__global__ __launch_bounds__(512, 2) void igemm_halo_bf16m512_kernel(const MsegIgemm p, int tw_log2, int ksplit,
                                                                     int chunks_per_split, int m_fastest) {
  igemm_halo_bf16_body<64, TR, S16, 512>(p, tw_log2, ksplit, chunks_per_split, m_fastest);
}

// ---- 4-wave form of the bf16 halo kernel for BN = 128 --------------------------------------------------------------------
// Same tile (128 pixels x 128 channels), LDS images and stage structure as igemm_halo_bf16_kernel, but FOUR waves that own
// 64 x 64 each (2 x 2 MFMA blocks): a stage is 24 MFMAs per wave instead of 12 and needs one LDS read per MFMA instead of
// 1.5; with one wave per SIMD and workgroup a lane may use 256 registers, which pays for fetching the weights TWO stages
// ahead (two register sets; the set is a compile-time choice, so the loop is unrolled over two chunks = six stages): a
// fetch has a full stage of its own wave plus the co-resident workgroup's MFMAs to arrive before it is written to LDS.
struct HaloCfg4 {
  static constexpr int WN = 2, WM = 2;
  static constexpr int TM = 64, TN = 64;
  static constexpr int MB = 2, NB = 2;
};

template <int TR, bool S16>
__global__ __launch_bounds__(256, 2) void igemm_halo_bf16w4_kernel(const MsegIgemm p, int tw_log2, int ksplit,
                                                                   int chunks_per_split, int m_fastest) {
  constexpr int BM = 128, BN = 128;
  using Cfg = HaloCfg4;
  constexpr int HMAX = 204;
  constexpr int SQ = S16 ? 4 : 8;                    // staging threads per halo row (8 / 4 channels each)
  constexpr int SROWS = 256 / SQ;                    // halo rows per staging pass
  constexpr int HL = (HMAX + SROWS - 1) / SROWS;     // 16-byte loads per thread per chunk (7; 4 for bf16 sources)
  constexpr int BP = BN / 64;                        // weight rows per thread and tap (2)
  constexpr int ASTAGE = HMAX * HB_STRIDE;
  constexpr int BSTAGE = 3 * BN * HB_STRIDE;
  __shared__ __attribute__((aligned(16))) __bf16 lds[ASTAGE + 2 * BSTAGE];
  __bf16* const Abuf = lds;
  __bf16* const Bbuf = lds + ASTAGE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int srow = tid / SQ, scol = tid % SQ;        // halo staging: SROWS rows x SQ channel groups per pass
  const int brow = tid >> 2, bcol = tid & 3;         // weight staging: 64 rows x 4 groups of 8 channels per pass
  const int TW = 1 << tw_log2, TH = BM >> tw_log2, HW2 = TW + 2;
  const int HROWS = (TH + 2) * HW2;
  const int H = p.Hi, W = p.Wi;
  const int M = p.NB * H * W;
  const int tiles_x = W >> tw_log2, tiles_y = (H + TH - 1) / TH;
  const int ntiles_n = (p.Ngemm + BN - 1) / BN;
  const int lid_all = (int)xcd_logical_id(blockIdx.x, gridDim.x);
  const int ntiles_all = (int)gridDim.x / ksplit;
  const int kz = lid_all / ntiles_all;
  const int lid = lid_all - kz * ntiles_all;
  const int ntiles_m = ntiles_all / ntiles_n;
  const int tile_m = m_fastest ? lid % ntiles_m : lid / ntiles_n;
  const int tile_n = m_fastest ? lid / ntiles_m : lid - tile_m * ntiles_n;
  const int img = tile_m / (tiles_x * tiles_y);
  const int trem = tile_m - img * (tiles_x * tiles_y);
  const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
  const int oy0 = ty * TH, ox0 = tx * TW, n0 = tile_n * BN;
  // buffer descriptors span only the row band of this tile's halo (image rows band0 .. band0 + band_rows - 1): 32-bit
  // offsets then never see more than (TH + 2) x W pixels, so neither the batch size nor the FRAME size is limited
  // (8192 x 8192 frames, the reference's largest tested shape, stay on this kernel)
  const int band0 = oy0 > 0 ? oy0 - 1 : 0;
  const int band_rows = (oy0 + TH + 1 < H ? oy0 + TH + 1 : H) - band0;
  const bool conv = p.mode == MSEG_MODE_CONV;

  int hpix[HL];
  unsigned hvalid = 0u;
#pragma unroll
  for (int j = 0; j < HL; ++j) {
    const int hrow = srow + SROWS * j;
    const int hy = hrow / HW2, hx = hrow - hy * HW2;
    const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
    const bool ok = (hrow < HROWS) & (iy >= 0) & (iy < H) & (ix >= 0) & (ix < W);
    hpix[j] = (iy - band0) * W + ix;
    hvalid |= (unsigned)ok << j;
  }
  int abase[Cfg::MB];
#pragma unroll
  for (int a = 0; a < Cfg::MB; ++a) {
    const int i = wm * Cfg::TM + a * 32 + li;
    abase[a] = ((i >> tw_log2) * HW2 + (i & (TW - 1))) * HB_STRIDE + lh * 8;
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
  const unsigned OOB = 0x80000000u;
  const int C1 = p.nsrc > 1 ? p.src[1].C : p.src[0].C;
  constexpr int ESZ = S16 ? 2 : 4;                   // bytes per source element
  const char* const base0 = (const char*)p.src[0].ptr + ((size_t)img * H + band0) * W * p.src[0].C * ESZ;
  const char* const base1 = (const char*)(p.nsrc > 1 ? p.src[1].ptr : p.src[0].ptr) + ((size_t)img * H + band0) * W * C1 * ESZ;
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base0), 0,
                                                                        band_rows * W * p.src[0].C * ESZ, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base1), 0,
                                                                        band_rows * W * C1 * ESZ, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0,
                                                                        9 * p.Npad * p.Kpad * 2, 0x00020000);
  const unsigned wvoff = ((unsigned)(n0 + brow) * (unsigned)p.Kpad + bcol * 8u) * 2u;
  const unsigned wvstep = 64u * (unsigned)p.Kpad * 2u;                              // scalar: 64 weight rows further

  f32x4 rh[HL];                                      // 16 raw bytes: 4 fp32 or 8 bf16 source channels
  float4 rsc, rsh, rsc2, rsh2;
  f32x4 rbA[3][BP], rbB[3][BP];
  unsigned hlive = 0u;
  int ract = 0;
  bool cur_s1 = false;

  auto issue_halo = [&](int chunk) {
    constexpr int CPT = S16 ? 8 : 4;                 // channels per staging thread
    const int c = chunk * KC + scol * CPT;
    cur_s1 = (p.nsrc > 1) && (chunk * KC >= C0);
    const MsegSrc& s = cur_s1 ? p.src[1] : p.src[0];
    const unsigned sCB = (unsigned)s.C * (unsigned)ESZ;                            // bytes per pixel
    const unsigned soff = (unsigned)(chunk * KC - (cur_s1 ? C0 : 0)) * (unsigned)ESZ + scol * 16u;
    hlive = (c < p.Cin) ? hvalid : 0u;
#pragma unroll
    for (int j = 0; j < HL; ++j) {
      const bool ok = (hlive >> j) & 1u;
      const unsigned vo = ok ? (unsigned)hpix[j] * sCB + soff : OOB;
      rh[j] = cur_s1 ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, vo, 0, 0))
                     : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs0, vo, 0, 0));
    }
    if (TR != 0) {
      ract = s.act;
      const bool has_aff = s.scale != nullptr;
      const float* scp = has_aff ? s.scale : g_ident_scale;
      const float* shp = has_aff ? s.shift : g_ident_shift;
      const unsigned cl = (c < p.Cin) ? (unsigned)(cur_s1 ? c - C0 : c) : 0u;
      const size_t o = (size_t)img * (has_aff ? (unsigned)s.ss : 0u) + cl;
      rsc = *reinterpret_cast<const float4*>(scp + o);
      rsh = *reinterpret_cast<const float4*>(shp + o);
      if (S16) {                                     // channels 4..7 of this thread (the identity tables are per channel too)
        rsc2 = *reinterpret_cast<const float4*>(scp + o + 4);
        rsh2 = *reinterpret_cast<const float4*>(shp + o + 4);
      }
    }
  };

  auto xform4 = [&](float4 v, const float4& sc, const float4& sh, float lo, float m) -> float4 {
    if (TR == 2) v = act_fwd4(v, ract);
    else { v.x = clamp_lo(v.x, lo); v.y = clamp_lo(v.y, lo); v.z = clamp_lo(v.z, lo); v.w = clamp_lo(v.w, lo); }
    v.x = (v.x * sc.x + sh.x) * m; v.y = (v.y * sc.y + sh.y) * m;
    v.z = (v.z * sc.z + sh.z) * m; v.w = (v.w * sc.w + sh.w) * m;
    return v;
  };

  auto commit_halo = [&](__bf16* As) {
    const float lo = (ract == MSEG_ACT_RELU) ? 0.f : -3.402823466e38f;
#pragma unroll
    for (int j = 0; j < HL; ++j) {
      const int hrow = srow + SROWS * j;
      if (hrow < HMAX) {
        const float m = ((hlive >> j) & 1u) ? 1.f : 0.f;
        if (S16) {
          const uint4 raw = __builtin_bit_cast(uint4, rh[j]);
          if (TR == 0) {                             // plain bf16 operand: already in LDS format
            *reinterpret_cast<uint4*>(As + hrow * HB_STRIDE + scol * 8) = raw;
          } else if (TR == 1) {                       // ReLU / none + affine: the short form (bf16_affine.h)
            *reinterpret_cast<uint4*>(As + hrow * HB_STRIDE + scol * 8) =
                mseg_affine8_bf16(raw, ract == MSEG_ACT_RELU ? 0u : 0x80008000u, rsc, rsh, rsc2, rsh2, m != 0.f);
          } else {
            const float4 a = xform4(bf16x4_to_f32(make_uint2(raw.x, raw.y)), rsc, rsh, lo, m);
            const float4 b = xform4(bf16x4_to_f32(make_uint2(raw.z, raw.w)), rsc2, rsh2, lo, m);
            const uint2 pa = f32x4_to_bf16(a), pb = f32x4_to_bf16(b);
            *reinterpret_cast<uint4*>(As + hrow * HB_STRIDE + scol * 8) = make_uint4(pa.x, pa.y, pb.x, pb.y);
          }
        } else {
          float4 v = make_float4(rh[j][0], rh[j][1], rh[j][2], rh[j][3]);
          if (TR != 0) v = xform4(v, rsc, rsh, lo, m);
          bf16x4 h;
          h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
          *reinterpret_cast<bf16x4*>(As + hrow * HB_STRIDE + scol * 4) = h;
        }
      }
    }
  };

  auto issue_b = [&](f32x4 (&rb)[3][BP], int chunk, int ky) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const unsigned wso = ((unsigned)(ky * 3 + kx) * (unsigned)p.Npad * (unsigned)p.Kpad + (unsigned)chunk * KC) * 2u;
#pragma unroll
      for (int i = 0; i < BP; ++i)
        rb[kx][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, wvoff, wso + wvstep * i, 0));
    }
  };
  auto commit_b = [&](const f32x4 (&rb)[3][BP], __bf16* Bs) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int i = 0; i < BP; ++i)
        *reinterpret_cast<f32x4*>(Bs + (kx * BN + brow + 64 * i) * HB_STRIDE + bcol * 8) = rb[kx][i];
  };

  const int c_begin = kz * chunks_per_split;
  const int c_end = (c_begin + chunks_per_split < nchunks) ? c_begin + chunks_per_split : nchunks;
  // stage s = (chunk, kernel row): weights fetched two stages ahead into register set s & 1, written to LDS buffer s & 1
  // at the end of stage s - 1; stages past the end re-read the last one (branch-free, harmless)
  auto stage_at = [&](int chunk, int ky, int ahead, int& c2, int& k2) {
    k2 = ky + ahead; c2 = chunk;
    if (k2 >= 3) { k2 -= 3; c2 += 1; }
    if (c2 >= c_end) { c2 = c_end - 1; k2 = 2; }
  };
  int bsel = 0;
  {
    int c1, k1;
    issue_halo(c_begin);
    issue_b(rbA, c_begin, 0);
    stage_at(c_begin, 0, 1, c1, k1);
    issue_b(rbB, c1, k1);
    commit_halo(Abuf);
    commit_b(rbA, Bbuf);
    __syncthreads();
  }
#define MSEG_HB4_STAGE(FREE_, HELD_, chunk_, ky_)                                                                  \
  {                                                                                                                \
    const int chunk = (chunk_);                                                                                    \
    constexpr int ky = (ky_);                                                                                      \
    const bool more_chunks = chunk + 1 < c_end;                                                                    \
    if (ky == 0 && more_chunks) issue_halo(chunk + 1);                                                             \
    int c2, k2;                                                                                                    \
    stage_at(chunk, ky, 2, c2, k2);                                                                                \
    issue_b(FREE_, c2, k2);                                                                                        \
    const __bf16* As = Abuf;                                                                                       \
    const __bf16* Bs = Bbuf + bsel * BSTAGE;                                                                       \
    __builtin_amdgcn_iglp_opt(1);                                                                                  \
    _Pragma("unroll") for (int kx = 0; kx < 3; ++kx) {                                                             \
      const int toff = ((conv ? ky : 2 - ky) * HW2 + (conv ? kx : 2 - kx)) * HB_STRIDE;                            \
      _Pragma("unroll") for (int kk = 0; kk < KC / 16; ++kk) {                                                     \
        bf16x8 af[Cfg::MB], bf[Cfg::NB];                                                                           \
        _Pragma("unroll") for (int a = 0; a < Cfg::MB; ++a)                                                        \
          af[a] = *reinterpret_cast<const bf16x8*>(As + abase[a] + toff + kk * 16);                                \
        _Pragma("unroll") for (int b = 0; b < Cfg::NB; ++b)                                                        \
          bf[b] = *reinterpret_cast<const bf16x8*>(Bs + (kx * BN + wn * Cfg::TN + b * 32 + li) * HB_STRIDE +      \
                                                   kk * 16 + lh * 8);                                              \
        _Pragma("unroll") for (int a = 0; a < Cfg::MB; ++a)                                                        \
          _Pragma("unroll") for (int b = 0; b < Cfg::NB; ++b)                                                      \
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);                 \
      }                                                                                                            \
    }                                                                                                              \
    commit_b(HELD_, Bbuf + (bsel ^ 1) * BSTAGE);                                                                   \
    if (ky == 2 && more_chunks) {                                                                                  \
      __syncthreads();                                                                                             \
      commit_halo(Abuf);                                                                                           \
    }                                                                                                              \
    __syncthreads();                                                                                               \
    bsel ^= 1;                                                                                                     \
  }
  int chunk0 = c_begin;
  for (; chunk0 + 1 < c_end; chunk0 += 2) {
    MSEG_HB4_STAGE(rbA, rbB, chunk0, 0)
    MSEG_HB4_STAGE(rbB, rbA, chunk0, 1)
    MSEG_HB4_STAGE(rbA, rbB, chunk0, 2)
    MSEG_HB4_STAGE(rbB, rbA, chunk0 + 1, 0)
    MSEG_HB4_STAGE(rbA, rbB, chunk0 + 1, 1)
    MSEG_HB4_STAGE(rbB, rbA, chunk0 + 1, 2)
  }
  if (chunk0 < c_end) {
    MSEG_HB4_STAGE(rbA, rbB, chunk0, 0)
    MSEG_HB4_STAGE(rbB, rbA, chunk0, 1)
    MSEG_HB4_STAGE(rbA, rbB, chunk0, 2)
  }
#undef MSEG_HB4_STAGE
  igemm_epilogue<Cfg>(acc, 0, n0, wm, wn, lane, M * ksplit, tw_log2, img + kz * p.NB, oy0, ox0);
}

// ---- 256-pixel tiles for the 128-channel-tile layers -----------------------------------------------------------------------
// igemm_halo_bf16w4_kernel is bound by LDS traffic, not by the matrix pipe (counters: its LDS-array cycles equal its MFMA
// cycles once the weight writes and the A-fragment conflicts of narrow tiles are counted).  Same four waves and the same 24
// MFMAs per wave and stage, but a tile of 256 pixels x 128 channels: a wave owns 128 pixels x 64 channels (4 x 2 MFMA
// blocks: 0.75 LDS reads per MFMA instead of 1.0) and a weight stage serves twice the pixels (half the weight bytes
// written to LDS and fetched from L2 per MFMA).  To keep two workgroups per CU a stage holds HALF a 32-channel chunk of
// the three taps of a kernel row (rows of 16 bf16 = 32 B at a 48-byte pitch: conflict-free 16-byte reads): six stages per
// chunk, 64 KB of LDS.  Weights are fetched two stages ahead into two register sets like in the 128-pixel form (six
// stages per chunk: the set of a stage is a compile-time choice without unrolling over two chunks).
struct HaloCfg4M {
  static constexpr int WN = 2, WM = 2;
  static constexpr int TM = 128, TN = 64;
  static constexpr int MB = 4, NB = 2;
};
#define HBM_BSTRIDE 24          // bf16 per weight row of a stage: 16 + 8 pad (48 B)

template <int TR, bool S16>
__global__ __launch_bounds__(256, 2) void igemm_halo_bf16w4m_kernel(const MsegIgemm p, int tw_log2, int ksplit,
                                                                    int chunks_per_split, int m_fastest) {
  constexpr int BM = 256, BN = 128;
  using Cfg = HaloCfg4M;
  constexpr int HMAX = 340;                          // (8 + 2) x (32 + 2) = (32 + 2) x (8 + 2); 16 x 16 tiles: 18 x 18 = 324
  constexpr int SQ = S16 ? 4 : 8;                    // staging threads per halo row (8 / 4 channels each)
  constexpr int SROWS = 256 / SQ;                    // halo rows per staging pass
  constexpr int HL = (HMAX + SROWS - 1) / SROWS;     // 16-byte loads per thread per chunk (11; 6 for bf16 sources)
  constexpr int ASTAGE = HMAX * HB_STRIDE;
  constexpr int BSTAGE = 3 * BN * HBM_BSTRIDE;
  __shared__ __attribute__((aligned(16))) __bf16 lds[ASTAGE + 2 * BSTAGE];
  __shared__ __attribute__((aligned(16))) float tabs[64];   // norm-on-load tables of the chunk being committed: scale[32] | shift[32]
  __bf16* const Abuf = lds;
  __bf16* const Bbuf = lds + ASTAGE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int srow = tid / SQ, scol = tid % SQ;        // halo staging: SROWS rows x SQ channel groups per pass
  const int brow = tid >> 1, bcol = tid & 1;         // weight staging: 128 rows x 2 groups of 8 channels per tap
  const int TW = 1 << tw_log2, TH = BM >> tw_log2, HW2 = TW + 2;
  const int HROWS = (TH + 2) * HW2;
  const int H = p.Hi, W = p.Wi;
  const int M = p.NB * H * W;
  const int tiles_x = W >> tw_log2, tiles_y = (H + TH - 1) / TH;
  const int ntiles_n = (p.Ngemm + BN - 1) / BN;
  const int lid_all = (int)xcd_logical_id(blockIdx.x, gridDim.x);
  const int ntiles_all = (int)gridDim.x / ksplit;
  const int kz = lid_all / ntiles_all;
  const int lid = lid_all - kz * ntiles_all;
  const int ntiles_m = ntiles_all / ntiles_n;
  const int tile_m = m_fastest ? lid % ntiles_m : lid / ntiles_n;
  const int tile_n = m_fastest ? lid / ntiles_m : lid - tile_m * ntiles_n;
  const int img = tile_m / (tiles_x * tiles_y);
  const int trem = tile_m - img * (tiles_x * tiles_y);
  const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
  const int oy0 = ty * TH, ox0 = tx * TW, n0 = tile_n * BN;
  const int band0 = oy0 > 0 ? oy0 - 1 : 0;           // descriptors span the row band of this tile's halo (see the 128-pixel form)
  const int band_rows = (oy0 + TH + 1 < H ? oy0 + TH + 1 : H) - band0;
  const bool conv = p.mode == MSEG_MODE_CONV;

  int hpix[HL];
  unsigned hvalid = 0u;
#pragma unroll
  for (int j = 0; j < HL; ++j) {
    const int hrow = srow + SROWS * j;
    const int hy = hrow / HW2, hx = hrow - hy * HW2;
    const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
    const bool ok = (hrow < HROWS) & (iy >= 0) & (iy < H) & (ix >= 0) & (ix < W);
    hpix[j] = (iy - band0) * W + ix;
    hvalid |= (unsigned)ok << j;
  }
  int abase[Cfg::MB];
#pragma unroll
  for (int a = 0; a < Cfg::MB; ++a) {
    const int i = wm * Cfg::TM + a * 32 + li;
    abase[a] = ((i >> tw_log2) * HW2 + (i & (TW - 1))) * HB_STRIDE + lh * 8;
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
  const unsigned OOB = 0x80000000u;
  const int C1 = p.nsrc > 1 ? p.src[1].C : p.src[0].C;
  constexpr int ESZ = S16 ? 2 : 4;                   // bytes per source element
  const char* const base0 = (const char*)p.src[0].ptr + ((size_t)img * H + band0) * W * p.src[0].C * ESZ;
  const char* const base1 = (const char*)(p.nsrc > 1 ? p.src[1].ptr : p.src[0].ptr) + ((size_t)img * H + band0) * W * C1 * ESZ;
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base0), 0,
                                                                        band_rows * W * p.src[0].C * ESZ, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base1), 0,
                                                                        band_rows * W * C1 * ESZ, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0,
                                                                        9 * p.Npad * p.Kpad * 2, 0x00020000);
  const unsigned wvoff = ((unsigned)(n0 + brow) * (unsigned)p.Kpad + bcol * 8u) * 2u;

  f32x4 rh[HL];                                      // 16 raw bytes: 4 fp32 or 8 bf16 source channels
  float4 rt = make_float4(0.f, 0.f, 0.f, 0.f);       // threads 0..15: one float4 of the next chunk's tables (the accumulators
  f32x4 rbA[3], rbB[3];                              // leave no room for every staging thread's own copy: they go through LDS)
  unsigned hlive = 0u;
  int ract = 0;
  bool cur_s1 = false;

  auto issue_halo = [&](int chunk) {
    constexpr int CPT = S16 ? 8 : 4;                 // channels per staging thread
    const int c = chunk * KC + scol * CPT;
    cur_s1 = (p.nsrc > 1) && (chunk * KC >= C0);
    const MsegSrc& s = cur_s1 ? p.src[1] : p.src[0];
    const unsigned sCB = (unsigned)s.C * (unsigned)ESZ;                            // bytes per pixel
    const unsigned soff = (unsigned)(chunk * KC - (cur_s1 ? C0 : 0)) * (unsigned)ESZ + scol * 16u;
    hlive = (c < p.Cin) ? hvalid : 0u;
#pragma unroll
    for (int j = 0; j < HL; ++j) {
      const bool ok = (hlive >> j) & 1u;
      const unsigned vo = ok ? (unsigned)hpix[j] * sCB + soff : OOB;
      rh[j] = cur_s1 ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, vo, 0, 0))
                     : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs0, vo, 0, 0));
    }
    if (TR != 0) {
      ract = s.act;
      if (tid < 16) {                                  // scale: threads 0..7, shift: threads 8..15, 4 channels each
        const bool has_aff = s.scale != nullptr;
        const float* tp = tid < 8 ? (has_aff ? s.scale : g_ident_scale) : (has_aff ? s.shift : g_ident_shift);
        const int c4 = chunk * KC + (tid & 7) * 4;
        const unsigned cl = (c4 < p.Cin) ? (unsigned)(cur_s1 ? c4 - C0 : c4) : 0u;
        rt = *reinterpret_cast<const float4*>(tp + (size_t)img * (has_aff ? (unsigned)s.ss : 0u) + cl);
      }
    }
  };
  // before the barrier that precedes commit_halo
  auto publish_tables = [&]() {
    if (TR != 0 && tid < 16) *reinterpret_cast<float4*>(tabs + tid * 4) = rt;
  };

  auto xform4 = [&](float4 v, const float4& sc, const float4& sh, float lo, float m) -> float4 {
    if (TR == 2) v = act_fwd4(v, ract);
    else { v.x = clamp_lo(v.x, lo); v.y = clamp_lo(v.y, lo); v.z = clamp_lo(v.z, lo); v.w = clamp_lo(v.w, lo); }
    v.x = (v.x * sc.x + sh.x) * m; v.y = (v.y * sc.y + sh.y) * m;
    v.z = (v.z * sc.z + sh.z) * m; v.w = (v.w * sc.w + sh.w) * m;
    return v;
  };

  auto commit_halo = [&](__bf16* As) {
    const float lo = (ract == MSEG_ACT_RELU) ? 0.f : -3.402823466e38f;
    constexpr int CPT = S16 ? 8 : 4;
    float4 rsc = make_float4(1.f, 1.f, 1.f, 1.f), rsh = make_float4(0.f, 0.f, 0.f, 0.f), rsc2 = rsc, rsh2 = rsh;
    if (TR != 0) {
      rsc = *reinterpret_cast<const float4*>(tabs + scol * CPT);
      rsh = *reinterpret_cast<const float4*>(tabs + 32 + scol * CPT);
      if (S16) {
        rsc2 = *reinterpret_cast<const float4*>(tabs + scol * CPT + 4);
        rsh2 = *reinterpret_cast<const float4*>(tabs + 32 + scol * CPT + 4);
      }
    }
#pragma unroll
    for (int j = 0; j < HL; ++j) {
      const int hrow = srow + SROWS * j;
      if (hrow < HMAX) {
        const float m = ((hlive >> j) & 1u) ? 1.f : 0.f;
        if (S16) {
          const uint4 raw = __builtin_bit_cast(uint4, rh[j]);
          if (TR == 0) {                             // plain bf16 operand: already in LDS format
            *reinterpret_cast<uint4*>(As + hrow * HB_STRIDE + scol * 8) = raw;
          } else if (TR == 1) {                       // ReLU / none + affine: the short form (bf16_affine.h)
            *reinterpret_cast<uint4*>(As + hrow * HB_STRIDE + scol * 8) =
                mseg_affine8_bf16(raw, ract == MSEG_ACT_RELU ? 0u : 0x80008000u, rsc, rsh, rsc2, rsh2, m != 0.f);
          } else {
            const float4 a = xform4(bf16x4_to_f32(make_uint2(raw.x, raw.y)), rsc, rsh, lo, m);
            const float4 b = xform4(bf16x4_to_f32(make_uint2(raw.z, raw.w)), rsc2, rsh2, lo, m);
            const uint2 pa = f32x4_to_bf16(a), pb = f32x4_to_bf16(b);
            *reinterpret_cast<uint4*>(As + hrow * HB_STRIDE + scol * 8) = make_uint4(pa.x, pa.y, pb.x, pb.y);
          }
        } else {
          float4 v = make_float4(rh[j][0], rh[j][1], rh[j][2], rh[j][3]);
          if (TR != 0) v = xform4(v, rsc, rsh, lo, m);
          bf16x4 h;
          h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
          *reinterpret_cast<bf16x4*>(As + hrow * HB_STRIDE + scol * 4) = h;
        }
      }
    }
  };

  // stage = (chunk, kernel row ky, K half kh): weights of 3 taps x [BN][16] bf16
  auto issue_b = [&](f32x4 (&rb)[3], int chunk, int ky, int kh) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const unsigned wso = ((unsigned)(ky * 3 + kx) * (unsigned)p.Npad * (unsigned)p.Kpad + (unsigned)chunk * KC +
                            (unsigned)kh * 16u) * 2u;
      rb[kx] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, wvoff, wso, 0));
    }
  };
  auto commit_b = [&](const f32x4 (&rb)[3], __bf16* Bs) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
      *reinterpret_cast<f32x4*>(Bs + (kx * BN + brow) * HBM_BSTRIDE + bcol * 8) = rb[kx];
  };

  const int c_begin = kz * chunks_per_split;
  const int c_end = (c_begin + chunks_per_split < nchunks) ? c_begin + chunks_per_split : nchunks;
  // stage index within a chunk: st = ky * 2 + kh (0..5); stages past the end re-read the last one (branch-free, harmless)
  auto stage_at = [&](int chunk, int st, int ahead, int& c2, int& s2) {
    s2 = st + ahead; c2 = chunk;
    if (s2 >= 6) { s2 -= 6; c2 += 1; }
    if (c2 >= c_end) { c2 = c_end - 1; s2 = 5; }
  };
  int bsel = 0;
  {
    int c1, s1;
    issue_halo(c_begin);
    issue_b(rbA, c_begin, 0, 0);
    stage_at(c_begin, 0, 1, c1, s1);
    issue_b(rbB, c1, s1 >> 1, s1 & 1);
    publish_tables();
    __syncthreads();
    commit_halo(Abuf);
    commit_b(rbA, Bbuf);
    __syncthreads();
  }
#define MSEG_HBM_STAGE(FREE_, HELD_, chunk_, st_)                                                                  \
  {                                                                                                                \
    const int chunk = (chunk_);                                                                                    \
    constexpr int st = (st_);                                                                                      \
    constexpr int ky = st >> 1, kh = st & 1;                                                                       \
    const bool more_chunks = chunk + 1 < c_end;                                                                    \
    if (st == 0 && more_chunks) issue_halo(chunk + 1);                                                             \
    int c2, s2;                                                                                                    \
    stage_at(chunk, st, 2, c2, s2);                                                                                \
    issue_b(FREE_, c2, s2 >> 1, s2 & 1);                                                                           \
    const __bf16* As = Abuf;                                                                                       \
    const __bf16* Bs = Bbuf + bsel * BSTAGE;                                                                       \
    __builtin_amdgcn_iglp_opt(1);                                                                                  \
    _Pragma("unroll") for (int kx = 0; kx < 3; ++kx) {                                                             \
      const int toff = ((conv ? ky : 2 - ky) * HW2 + (conv ? kx : 2 - kx)) * HB_STRIDE;                            \
      bf16x8 bf[Cfg::NB];                                                                                          \
      _Pragma("unroll") for (int b = 0; b < Cfg::NB; ++b)                                                          \
        bf[b] = *reinterpret_cast<const bf16x8*>(Bs + (kx * BN + wn * Cfg::TN + b * 32 + li) * HBM_BSTRIDE + lh * 8); \
      _Pragma("unroll") for (int a = 0; a < Cfg::MB; ++a) {                                                        \
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(As + abase[a] + toff + kh * 16);                        \
        _Pragma("unroll") for (int b = 0; b < Cfg::NB; ++b)                                                        \
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf[b], acc[a][b], 0, 0, 0);                      \
      }                                                                                                            \
    }                                                                                                              \
    commit_b(HELD_, Bbuf + (bsel ^ 1) * BSTAGE);                                                                   \
    if (st == 5 && more_chunks) {                                                                                  \
      publish_tables();                                                                                            \
      __syncthreads();                                                                                             \
      commit_halo(Abuf);                                                                                           \
    }                                                                                                              \
    __syncthreads();                                                                                               \
    bsel ^= 1;                                                                                                     \
  }
  for (int chunk0 = c_begin; chunk0 < c_end; ++chunk0) {
    MSEG_HBM_STAGE(rbA, rbB, chunk0, 0)
    MSEG_HBM_STAGE(rbB, rbA, chunk0, 1)
    MSEG_HBM_STAGE(rbA, rbB, chunk0, 2)
    MSEG_HBM_STAGE(rbB, rbA, chunk0, 3)
    MSEG_HBM_STAGE(rbA, rbB, chunk0, 4)
    MSEG_HBM_STAGE(rbB, rbA, chunk0, 5)
  }
#undef MSEG_HBM_STAGE
  igemm_epilogue<Cfg>(acc, 0, n0, wm, wn, lane, M * ksplit, tw_log2, img + kz * p.NB, oy0, ox0);
}

// ---- persistent bf16 kernel for the 64 -> 64 channel layers of level 0 ------------------------------------------------------
// Level 0 has the most pixels and the fewest channels: 3 x 3 x 64 x 64 is 241 GFLOP per launch at 32 x 320 x 320 (0.1 ms
// at the bf16 matrix peak) against 0.84 GB of bf16 tensors in and out (0.17 ms at 5 TB/s) — an HBM-bound layer.  The
// tile-per-workgroup kernel above re-stages the layer's 74 KB of weights for every 128-pixel tile (three times the bytes
// of the tile's halo) and runs at a third of that bound.  Here the weights of all nine taps stay in LDS for the lifetime
// of a workgroup (one per CU), which walks pixel tiles: per step TWO independent 128-pixel tiles (one per group of four
// waves, sharing the weights), their halos (64 channels per row) fetched into registers during the previous step's MFMAs
// and written to LDS after them, labels of the step before stored while the next fetch is in flight.  64-bit global
// addresses per halo row: no descriptor limits on frame or batch size.
struct C64Cfg {
  static constexpr int WN = 2, WM = 2;
  static constexpr int TM = 64, TN = 32;
  static constexpr int MB = 2, NB = 1;
};
#define C64_STRIDE 72           // bf16 per LDS row: 64 channels + 8 pad (144 B: conflict-free 16-byte reads of consecutive rows)
#define C64_HPAD 208            // staging rows per tile (204 halo rows rounded up to a multiple of 8: a wave stays in one tile)

// Workgroup barrier that orders LDS only: __syncthreads() also waits for every global store of the epilogue and every halo
// load in flight (vmcnt(0)) — the two latencies this kernel exists to overlap.  Global memory needs no ordering here: the
// workgroups write disjoint outputs and read tensors earlier launches produced.
#define C64_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// Barrier among the four waves of one tile group (there is one hardware barrier per workgroup, and the two groups must NOT
// wait for each other): a monotonically increasing counter in LDS.  A wave's LDS operations execute in order, so whoever
// sees its increment sees its earlier writes.  Every wave of a group passes the same number of these barriers.
__device__ __forceinline__ void c64_group_barrier(unsigned* cnt, unsigned& target) {
  target += 4u;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
}

// Two groups of four waves per workgroup, each walking its own sequence of 128-pixel tiles with its own halo buffer and its
// own barriers; they only share the resident weights.  Running them out of step is the point: with all waves of a CU in the
// same phase the vector instructions of the fetch / commit / store phases are hidden behind nobody's MFMAs (measured with
// one common barrier: those phases cost more time than the 72 MFMAs of a tile), whereas a group's MFMAs now run under the
// other group's loads, LDS writes and stores — what two co-resident workgroups would give, if two copies of the weights
// fitted in LDS.  Everything a thread needs per tile besides the tile's origin is computed ONCE per launch (offsets of its
// halo rows and of its output rows relative to the tile origin).
template <int TR, bool D16>
__global__ __launch_bounds__(512, 1) void igemm_c64p_bf16_kernel(const MsegIgemm p, int tw_log2, int ntiles) {
  using Cfg = C64Cfg;
  constexpr int HL = 7;                               // staging passes of a group: 208 rows x 8 threads over 256 threads
  constexpr int NE = D16 ? 8 : 16;                    // stores per lane and 32-row block (bf16: channel pairs)
  __shared__ __attribute__((aligned(16))) __bf16 lds[9 * 64 * C64_STRIDE + 2 * C64_HPAD * C64_STRIDE];
  __shared__ unsigned gcount[2];
  __bf16* const Wl = lds;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 2;                          // tile group of this wave
  const int gtid = tid & 255;
  __bf16* const Hl = lds + 9 * 64 * C64_STRIDE + grp * C64_HPAD * C64_STRIDE;
  const int wm = (wave >> 1) & 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int srow = gtid >> 3, scol = gtid & 7;        // staging: 32 rows x 8 groups of 8 channels per pass
  const int TW = 1 << tw_log2, TH = 128 >> tw_log2, HW2 = TW + 2;
  const int HROWS = (TH + 2) * HW2;
  const int H = p.Hi, W = p.Wi;
  const int tiles_x = W >> tw_log2, tiles_y = (H + TH - 1) / TH;
  const int tiles_img = tiles_x * tiles_y;
  const bool conv = p.mode == MSEG_MODE_CONV;
  const MsegSrc& s = p.src[0];
  const __bf16* const srcp = reinterpret_cast<const __bf16*>(s.ptr);

  // halo rows of this thread: position inside the halo (hy, hx) and element offset from the halo's first pixel
  // (image position (oy0 - 1, ox0 - 1) — possibly outside the image; such rows are never dereferenced)
  int hyx[HL], roff[HL];
  unsigned hgeo = 0u;
#pragma unroll
  for (int j = 0; j < HL; ++j) {
    const int hrow = srow + 32 * j;
    const int hy = hrow / HW2, hx = hrow - hy * HW2;
    hyx[j] = (hy << 16) | hx;
    roff[j] = (hy * W + hx) * 64 + scol * 8;
    hgeo |= (unsigned)(hrow < HROWS) << j;
  }
  const int roff_safe = (W + 1) * 64 + scol * 8;      // the tile's own first pixel: always inside the image
  int abase[Cfg::MB];
#pragma unroll
  for (int a = 0; a < Cfg::MB; ++a) {
    const int i = wm * Cfg::TM + a * 32 + li;
    abase[a] = ((i >> tw_log2) * HW2 + (i & (TW - 1))) * C64_STRIDE + lh * 8;
  }
  // output rows of this lane: element offset from the tile's first output pixel, and the row's dy (rows past the image's
  // last row — only possible in the bottom tile row — are skipped)
  const int n_out = wn * 32 + li;                     // this lane's output channel
  const bool odd = li & 1;
  const int ld = p.ld0;
  int eoff[Cfg::MB][NE], edy[Cfg::MB][NE];
#pragma unroll
  for (int a = 0; a < Cfg::MB; ++a)
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int r = D16 ? 2 * e + (odd ? 1 : 0) : e;
      const int i = wm * Cfg::TM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      edy[a][e] = i >> tw_log2;
      eoff[a][e] = ((i >> tw_log2) * W + (i & (TW - 1))) * ld + n_out - (D16 && odd ? 1 : 0);
    }
  const bool nvalid = n_out < p.Ngemm;
  const float bias = (p.bias && nvalid) ? p.bias[n_out] : 0.f;
  const int accf = p.acc0;

  // the layer's weights, all nine taps: [tap][n < 64][k < 64] bf16, resident for the whole launch
  {
    const __bf16* const wp = reinterpret_cast<const __bf16*>(p.w);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int idx = tid + 512 * i;                   // 9 x 64 rows x 8 column groups = 4608
      const int row = idx >> 3, c8 = idx & 7;
      const int tap = row >> 6, n = row & 63;
      const uint4 v = *reinterpret_cast<const uint4*>(wp + ((size_t)tap * p.Npad + n) * p.Kpad + c8 * 8);
      *reinterpret_cast<uint4*>(Wl + row * C64_STRIDE + c8 * 8) = v;
    }
    if (tid < 2) gcount[tid] = 0u;
  }
  __syncthreads();                                     // the only workgroup-wide barrier

  uint4 rh[HL];
  unsigned hlive = 0u;
  float4 tsc[2], tsh[2];                               // norm-on-load tables of this thread's 8 channels (per tile: per-sample norms)
  int ract = 0;

  auto decode = [&](int t, int& img, int& oy, int& ox) {
    img = t / tiles_img;
    const int trem = t - img * tiles_img;
    const int ty = trem / tiles_x;
    oy = ty * TH; ox = (trem - ty * tiles_x) * TW;
  };

  auto issue_halo = [&](int t) {
    int img, oy, ox;
    decode(t, img, oy, ox);
    // first pixel of the tile's halo (wave-uniform 64-bit base; the per-row offsets are 32-bit and launch-invariant)
    const __bf16* const hb = srcp + (((long long)img * H + oy - 1) * W + ox - 1) * 64;
    hlive = 0u;
#pragma unroll
    for (int j = 0; j < HL; ++j) {
      const int iy = oy - 1 + (hyx[j] >> 16), ix = ox - 1 + (hyx[j] & 0xffff);
      const bool live = ((hgeo >> j) & 1u) && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      // always load (a dead row reads the tile's first pixel and is zeroed when written to LDS): a conditional load would
      // be merged with its zero right here, i.e. waited for BEFORE the MFMAs it is meant to overlap
      rh[j] = *reinterpret_cast<const uint4*>(hb + (live ? roff[j] : roff_safe));
      hlive |= (unsigned)live << j;
    }
    if (TR != 0) {
      ract = s.act;
      const bool has_aff = s.scale != nullptr;
      const float* scp = has_aff ? s.scale : g_ident_scale;
      const float* shp = has_aff ? s.shift : g_ident_shift;
      const size_t o = (size_t)img * (has_aff ? (unsigned)s.ss : 0u) + scol * 8;
      tsc[0] = *reinterpret_cast<const float4*>(scp + o); tsc[1] = *reinterpret_cast<const float4*>(scp + o + 4);
      tsh[0] = *reinterpret_cast<const float4*>(shp + o); tsh[1] = *reinterpret_cast<const float4*>(shp + o + 4);
    }
  };

  auto xform4 = [&](float4 v, const float4& sc, const float4& sh, float lo) -> float4 {
    if (TR == 2) v = act_fwd4(v, ract);
    else { v.x = clamp_lo(v.x, lo); v.y = clamp_lo(v.y, lo); v.z = clamp_lo(v.z, lo); v.w = clamp_lo(v.w, lo); }
    v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
    return v;
  };

  auto commit_halo = [&]() {
    const float lo = (ract == MSEG_ACT_RELU) ? 0.f : -3.402823466e38f;
#pragma unroll
    for (int j = 0; j < HL; ++j) {
      if (!((hgeo >> j) & 1u)) continue;
      uint4 v = rh[j];
      if (TR != 0) {                                    // (TR == 0: a plain bf16 operand is already in LDS format)
        if (TR == 1) {                                  // ReLU / none + affine: the short form (bf16_affine.h)
          v = mseg_affine8_bf16(v, ract == MSEG_ACT_RELU ? 0u : 0x80008000u, tsc[0], tsh[0], tsc[1], tsh[1], true);
        } else {
          const float4 a = xform4(bf16x4_to_f32(make_uint2(v.x, v.y)), tsc[0], tsh[0], lo);
          const float4 b = xform4(bf16x4_to_f32(make_uint2(v.z, v.w)), tsc[1], tsh[1], lo);
          const uint2 pa = f32x4_to_bf16(a), pb = f32x4_to_bf16(b);
          v = make_uint4(pa.x, pa.y, pb.x, pb.y);
        }
      }
      if (!((hlive >> j) & 1u)) v = make_uint4(0u, 0u, 0u, 0u);     // zero padding / rows outside the image
      *reinterpret_cast<uint4*>(Hl + (srow + 32 * j) * C64_STRIDE + scol * 8) = v;
    }
  };

  // stores of one finished tile: straight-line, every read-modify-write load unconditional (rows past the image re-read the
  // tile's first row) and consumed before the first store — nothing stays pending across the loop's back edge
  auto store_tile = [&](f32x16 (&acc)[Cfg::MB][Cfg::NB], int img, int oy0, int ox0) {
    if (!nvalid) return;
    const int rows_in = H - oy0;                        // image rows left from the tile's first row
    if (D16) {
      __bf16* const d = reinterpret_cast<__bf16*>(p.dst0) + (((long long)img * H + oy0) * W + ox0) * ld;
      unsigned oldw[Cfg::MB][NE];
#pragma unroll
      for (int a = 0; a < Cfg::MB; ++a)
#pragma unroll
        for (int e = 0; e < NE; ++e) oldw[a][e] = 0u;
      if (accf) {
#pragma unroll
        for (int a = 0; a < Cfg::MB; ++a)
#pragma unroll
          for (int e = 0; e < NE; ++e)
            oldw[a][e] = *reinterpret_cast<const unsigned*>(d + (edy[a][e] < rows_in ? eoff[a][e] : eoff[a][0] - edy[a][0] * W * ld));
        // pin the consumption here: the compiler sinks the additions below into the per-lane store branches, which leaves
        // these loads "pending" on the other path — and makes it wait for EVERY outstanding load and store at the top of
        // the next trip
#pragma unroll
        for (int a = 0; a < Cfg::MB; ++a)
#pragma unroll
          for (int e = 0; e < NE; ++e) asm volatile("" : "+v"(oldw[a][e]));
      }
#pragma unroll
      for (int a = 0; a < Cfg::MB; ++a) {
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          // lanes 2k / 2k+1 hold channels n, n+1 of the same rows: they swap halves through a DPP quad permute, the even
          // lane stores the channel PAIR of the even accumulator rows and the odd lane that of the odd rows
          const float mine_e = acc[a][0][2 * e] + bias, mine_o = acc[a][0][2 * e + 1] + bias;
          const float give = odd ? mine_e : mine_o, keep = odd ? mine_o : mine_e;
          const float got = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, give), 0xB1, 0xf, 0xf, true));
          const float lo = (odd ? got : keep) + bf16_lo(oldw[a][e]), hi = (odd ? keep : got) + bf16_hi(oldw[a][e]);
          const unsigned v = pack_bf16x2(lo, hi);
          if (edy[a][e] < rows_in) *reinterpret_cast<unsigned*>(d + eoff[a][e]) = v;
        }
      }
    } else {
      float* const d = reinterpret_cast<float*>(p.dst0) + (((long long)img * H + oy0) * W + ox0) * ld;
#pragma unroll
      for (int a = 0; a < Cfg::MB; ++a) {
        float old[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) old[e] = 0.f;
        if (accf) {
#pragma unroll
          for (int e = 0; e < NE; ++e) old[e] = d[edy[a][e] < rows_in ? eoff[a][e] : eoff[a][0] - edy[a][0] * W * ld];
#pragma unroll
          for (int e = 0; e < NE; ++e) asm volatile("" : "+v"(old[e]));
        }
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          const float v = acc[a][0][e] + bias + old[e];
          if (edy[a][e] < rows_in) d[eoff[a][e]] = v;
        }
      }
    }
  };

  // tiles of this group: 2 * lid + grp, then every 2 * G-th (neighbouring tiles — which share halo rows — run at the same
  // time on one XCD's L2)
  const int stride = 2 * (int)gridDim.x;
  int t = 2 * (int)xcd_logical_id(blockIdx.x, gridDim.x) + grp;
  unsigned gb_target = 0u;
  unsigned* const gcnt = &gcount[grp];
  // Software pipeline: the halo of tile t + 2 * stride is fetched (into registers) right after the one of t + stride went to
  // LDS, i.e. before the stores of tile t; it has those stores, a barrier and the MFMAs of t + stride to arrive.
  if (t < ntiles) {
    issue_halo(t);
    commit_halo();
    if (t + stride < ntiles) issue_halo(t + stride);
    c64_group_barrier(gcnt, gb_target);
  }
  for (; t < ntiles; t += stride) {
    const int next = t + stride;
    f32x16 acc[Cfg::MB][Cfg::NB];
#pragma unroll
    for (int a = 0; a < Cfg::MB; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][0][r] = 0.f;
    {
      const __bf16* Wt = Wl + (wn * Cfg::TN + li) * C64_STRIDE + lh * 8;
      int ky = 0, kx = 0;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int toff = ((conv ? ky : 2 - ky) * HW2 + (conv ? kx : 2 - kx)) * C64_STRIDE;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const bf16x8 bf = *reinterpret_cast<const bf16x8*>(Wt + kk * 16);
#pragma unroll
          for (int a = 0; a < Cfg::MB; ++a) {
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(Hl + abase[a] + toff + kk * 16);
            acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[a][0], 0, 0, 0);
          }
        }
        Wt += 64 * C64_STRIDE;
        if (++kx == 3) { kx = 0; ++ky; }
      }
    }
    c64_group_barrier(gcnt, gb_target);                 // the group is done reading this tile's halo
    if (next < ntiles) {
      commit_halo();
      if (next + stride < ntiles) issue_halo(next + stride);
    }
    int e_img, e_oy, e_ox;
    decode(t, e_img, e_oy, e_ox);
    store_tile(acc, e_img, e_oy, e_ox);
    c64_group_barrier(gcnt, gb_target);                 // the next halo is in LDS
  }
}

// ---- persistent bf16 kernel for the level-0 ConvTranspose2d(128 -> 64, 2, stride 2) ------------------------------------------
// As a GEMM this layer is M = input pixels, K = 128, N = 4 x 64 (the four output positions of a pixel), i.e. 66 kFLOP per
// 768 bytes moved: HBM-bound like the 64 -> 64 convolutions, and like them it ran at a third of that bound on the
// tile-per-workgroup gather kernel (four K-steps per tile: prologue, first fetch and scatter epilogue with nothing to
// overlap them).  Same treatment as igemm_c64p_bf16_kernel: the 64 KB of weights stay in LDS, two groups of four waves walk
// their own 128-pixel tiles out of step (LDS-counter barriers), rows fetched one tile ahead, LDS-only synchronisation.
// No halo here: a tile is 128 consecutive pixels of the NHWC tensor (one contiguous 32 KB read); a wave owns 64 pixels x
// 128 of the 256 GEMM columns; 32-pixel blocks stay inside an image row (W % 32 == 0), so the scatter is affine per block.
struct CtpCfg {
  static constexpr int MB = 2, NB = 4;               // per wave: 64 pixels x 128 columns
};
#define CTP_STRIDE 136          // bf16 per LDS row: 128 channels + 8 pad (272 B: conflict-free 16-byte reads of consecutive rows)

template <int TR, bool D16>
__global__ __launch_bounds__(512, 1) void igemm_ctp_bf16_kernel(const MsegIgemm p, int ntiles) {
  constexpr int HL = 8;                               // staging passes of a group: 128 rows x 16 threads over 256 threads
  constexpr int NE = D16 ? 8 : 16;                    // stores per lane and 32 x 32 block (bf16: channel pairs)
  __shared__ __attribute__((aligned(16))) __bf16 lds[256 * CTP_STRIDE + 2 * 128 * CTP_STRIDE];
  __shared__ unsigned gcount[2];
  __bf16* const Wl = lds;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 2;
  const int gtid = tid & 255;
  __bf16* const Al = lds + 256 * CTP_STRIDE + grp * 128 * CTP_STRIDE;
  const int wm = (wave >> 1) & 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int srow = gtid >> 4, scol = gtid & 15;       // staging: 16 rows x 16 groups of 8 channels per pass
  const int H = p.Hi, W = p.Wi, HW = H * W;
  const int Cq = p.Cq;                                // 64 output channels
  const MsegSrc& s = p.src[0];
  const __bf16* const srcp = reinterpret_cast<const __bf16*>(s.ptr);

  // weights: [256 GEMM columns][128 k] bf16, resident for the whole launch
  {
    const __bf16* const wp = reinterpret_cast<const __bf16*>(p.w);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = tid + 512 * i;                   // 256 rows x 16 column groups = 4096
      const int row = idx >> 4, c8 = idx & 15;
      const uint4 v = *reinterpret_cast<const uint4*>(wp + (size_t)row * p.Kpad + c8 * 8);
      *reinterpret_cast<uint4*>(Wl + row * CTP_STRIDE + c8 * 8) = v;
    }
    if (tid < 2) gcount[tid] = 0u;
  }
  __syncthreads();                                     // the only workgroup-wide barrier

  // output side, launch-invariant per lane: this lane's channel, its bias, the offsets of its rows inside a 32-pixel block
  // (consecutive input pixels of one image row are 2 output pixels = 2 * Cq elements apart)
  const bool odd = li & 1;
  const int co = li;                                   // + 32 * (b & 1): two 32-column blocks per output position
  float bias[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) bias[h] = p.bias ? p.bias[32 * h + co] : 0.f;
  int eoff[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int r = D16 ? 2 * e + (odd ? 1 : 0) : e;
    eoff[e] = ((r & 3) + 8 * (r >> 2) + 4 * lh) * 2 * Cq;
  }

  uint4 rh[HL];
  float4 tsc[2], tsh[2];
  int ract = 0;

  auto issue_rows = [&](int t) {
    const __bf16* const tb = srcp + (size_t)t * 128 * 128 + scol * 8;          // tile t: pixels 128 t .. 128 t + 127
#pragma unroll
    for (int j = 0; j < HL; ++j) rh[j] = *reinterpret_cast<const uint4*>(tb + (size_t)(srow + 16 * j) * 128);
    if (TR != 0) {
      ract = s.act;
      const bool has_aff = s.scale != nullptr;
      const float* scp = has_aff ? s.scale : g_ident_scale;
      const float* shp = has_aff ? s.shift : g_ident_shift;
      const int img = (int)(((long long)t * 128) / HW);                         // a tile never spans two images (HW % 128 == 0)
      const size_t o = (size_t)img * (has_aff ? (unsigned)s.ss : 0u) + scol * 8;
      tsc[0] = *reinterpret_cast<const float4*>(scp + o); tsc[1] = *reinterpret_cast<const float4*>(scp + o + 4);
      tsh[0] = *reinterpret_cast<const float4*>(shp + o); tsh[1] = *reinterpret_cast<const float4*>(shp + o + 4);
    }
  };
  auto xform4 = [&](float4 v, const float4& sc, const float4& sh, float lo) -> float4 {
    if (TR == 2) v = act_fwd4(v, ract);
    else { v.x = clamp_lo(v.x, lo); v.y = clamp_lo(v.y, lo); v.z = clamp_lo(v.z, lo); v.w = clamp_lo(v.w, lo); }
    v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
    return v;
  };
  auto commit_rows = [&]() {
    const float lo = (ract == MSEG_ACT_RELU) ? 0.f : -3.402823466e38f;
#pragma unroll
    for (int j = 0; j < HL; ++j) {
      uint4 v = rh[j];
      if (TR != 0) {
        if (TR == 1) {                                  // ReLU / none + affine: the short form (bf16_affine.h)
          v = mseg_affine8_bf16(v, ract == MSEG_ACT_RELU ? 0u : 0x80008000u, tsc[0], tsh[0], tsc[1], tsh[1], true);
        } else {
          const float4 a = xform4(bf16x4_to_f32(make_uint2(v.x, v.y)), tsc[0], tsh[0], lo);
          const float4 b = xform4(bf16x4_to_f32(make_uint2(v.z, v.w)), tsc[1], tsh[1], lo);
          const uint2 pa = f32x4_to_bf16(a), pb = f32x4_to_bf16(b);
          v = make_uint4(pa.x, pa.y, pb.x, pb.y);
        }
      }
      *reinterpret_cast<uint4*>(Al + (srow + 16 * j) * CTP_STRIDE + scol * 8) = v;
    }
  };

  // scatter of one finished tile: GEMM column n = (a * 2 + b) * Cq + co goes to output pixel (2 y + a, 2 x + b), channel co
  auto store_tile = [&](f32x16 (&acc)[CtpCfg::MB][CtpCfg::NB], int t) {
#pragma unroll
    for (int mb = 0; mb < CtpCfg::MB; ++mb) {
      const int p0 = t * 128 + wm * 64 + mb * 32;       // first input pixel of this 32-row block (wave-uniform)
      const int img = p0 / HW;
      const int rem = p0 - img * HW;
      const int y = rem / W, x0 = rem - y * W;
#pragma unroll
      for (int nb = 0; nb < CtpCfg::NB; ++nb) {
        const int ab = wn * 2 + (nb >> 1);              // output position of this 32-column block
        const long long base = ((((long long)img * 2 * H + 2 * y + (ab >> 1)) * 2 * W) + 2 * x0 + (ab & 1)) * Cq +
                               32 * (nb & 1) + co;
        const float bv = bias[nb & 1];
        if (D16) {
          __bf16* const d = reinterpret_cast<__bf16*>(p.dst0) + base - (odd ? 1 : 0);
#pragma unroll
          for (int e = 0; e < NE; ++e) {
            const float mine_e = acc[mb][nb][2 * e] + bv, mine_o = acc[mb][nb][2 * e + 1] + bv;
            const float give = odd ? mine_e : mine_o, keep = odd ? mine_o : mine_e;
            const float got = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, give), 0xB1, 0xf, 0xf, true));
            // the neighbour lane holds the adjacent channel, whose bias differs: it was added on the giving side
            *reinterpret_cast<unsigned*>(d + eoff[e]) = pack_bf16x2(odd ? got : keep, odd ? keep : got);
          }
        } else {
          float* const d = reinterpret_cast<float*>(p.dst0) + base;
#pragma unroll
          for (int e = 0; e < NE; ++e) d[eoff[e]] = acc[mb][nb][e] + bv;
        }
      }
    }
  };

  const int stride = 2 * (int)gridDim.x;
  int t = 2 * (int)xcd_logical_id(blockIdx.x, gridDim.x) + grp;
  unsigned gb_target = 0u;
  unsigned* const gcnt = &gcount[grp];
  if (t < ntiles) {
    issue_rows(t);
    commit_rows();
    if (t + stride < ntiles) issue_rows(t + stride);
    c64_group_barrier(gcnt, gb_target);
  }
  const int a_off = (wm * 64 + li) * CTP_STRIDE + lh * 8;
  const int b_off = (wn * 128 + li) * CTP_STRIDE + lh * 8;
  for (; t < ntiles; t += stride) {
    const int next = t + stride;
    f32x16 acc[CtpCfg::MB][CtpCfg::NB];
#pragma unroll
    for (int mb = 0; mb < CtpCfg::MB; ++mb)
#pragma unroll
      for (int nb = 0; nb < CtpCfg::NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      bf16x8 af[CtpCfg::MB];
#pragma unroll
      for (int mb = 0; mb < CtpCfg::MB; ++mb)
        af[mb] = *reinterpret_cast<const bf16x8*>(Al + a_off + mb * 32 * CTP_STRIDE + kk * 16);
#pragma unroll
      for (int nb = 0; nb < CtpCfg::NB; ++nb) {
        const bf16x8 bf = *reinterpret_cast<const bf16x8*>(Wl + b_off + nb * 32 * CTP_STRIDE + kk * 16);
#pragma unroll
        for (int mb = 0; mb < CtpCfg::MB; ++mb)
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mb], bf, acc[mb][nb], 0, 0, 0);
      }
    }
    c64_group_barrier(gcnt, gb_target);                 // the group is done reading this tile's rows
    if (next < ntiles) {
      commit_rows();
      if (next + stride < ntiles) issue_rows(next + stride);
    }
    store_tile(acc, t);
    c64_group_barrier(gcnt, gb_target);                 // the next tile is in LDS
  }
}

#define CTD_ASTRIDE 264         // bf16 per LDS row of 256 k + 8 pad (528 B: conflict-free 16-byte reads of consecutive rows)

// ---- ... for the level-1 ConvTranspose2d(256 -> 128): 262 KB of weights, resident per OUTPUT POSITION ------------------------
// A workgroup keeps the 256 x 128 weights of ONE of the four output positions (a, b) (68 KB) and computes that position for
// its tiles; the four workgroups of a tile stream have neighbouring logical ids (one XCD): they walk the same tiles at the
// same time, so the tile (64 pixels x 256 channels = 32 KB contiguous) comes from HBM once and from L2 three times.  A wave
// owns 32 pixels x 64 of the 128 channels.
template <int TR, bool D16>
__global__ __launch_bounds__(512, 1) void igemm_ctp2_bf16_kernel(const MsegIgemm p, int ntiles) {
  constexpr int HL = 8;                               // staging passes of a group: 64 rows x 32 threads over 256 threads
  constexpr int NE = D16 ? 8 : 16;
  __shared__ __attribute__((aligned(16))) __bf16 lds[128 * CTD_ASTRIDE + 2 * 64 * CTD_ASTRIDE];
  __shared__ unsigned gcount[2];
  __bf16* const Wl = lds;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 2;
  const int gtid = tid & 255;
  __bf16* const Al = lds + 128 * CTD_ASTRIDE + grp * 64 * CTD_ASTRIDE;
  const int wm = (wave >> 1) & 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int srow = gtid >> 5, scol = gtid & 31;       // staging: 8 rows x 32 groups of 8 channels per pass
  const int H = p.Hi, W = p.Wi, HW = H * W;
  const int Cq = p.Cq;                                // 128 output channels
  const MsegSrc& s = p.src[0];
  const __bf16* const srcp = reinterpret_cast<const __bf16*>(s.ptr);
  const int lid = (int)xcd_logical_id(blockIdx.x, gridDim.x);
  const int ab = lid & 3;                             // this workgroup's output position
  const int stream = lid >> 2, nstreams = (int)gridDim.x >> 2;

  {
    const __bf16* const wp = reinterpret_cast<const __bf16*>(p.w) + (size_t)ab * Cq * p.Kpad;   // rows ab * 128 ..
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = tid + 512 * i;                   // 128 rows x 32 column groups = 4096
      const int row = idx >> 5, c8 = idx & 31;
      const uint4 v = *reinterpret_cast<const uint4*>(wp + (size_t)row * p.Kpad + c8 * 8);
      *reinterpret_cast<uint4*>(Wl + row * CTD_ASTRIDE + c8 * 8) = v;
    }
    if (tid < 2) gcount[tid] = 0u;
  }
  __syncthreads();

  const bool odd = li & 1;
  float bias[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) bias[h] = p.bias ? p.bias[wn * 64 + 32 * h + li] : 0.f;
  // rows of this lane inside a 32-pixel block.  A block may wrap into the next image row once (W >= 32; W = 80 at 320 x 320
  // crops): the wrapped rows sit 2 W Cq elements further (one output row of the other parity in between)
  int eoff[NE], erow[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int r = D16 ? 2 * e + (odd ? 1 : 0) : e;
    erow[e] = (r & 3) + 8 * (r >> 2) + 4 * lh;
    eoff[e] = erow[e] * 2 * Cq;
  }
  const int wrap = 2 * W * Cq;

  uint4 rh0, rh1, rh2, rh3, rh4, rh5, rh6, rh7;
  float4 tsc[2], tsh[2];
  int ract = 0;
  auto issue_rows = [&](int t) {
    const __bf16* const tb = srcp + ((size_t)t * 64 + srow) * 256 + scol * 8;          // tile t: pixels 64 t .. 64 t + 63
    rh0 = *reinterpret_cast<const uint4*>(tb); rh1 = *reinterpret_cast<const uint4*>(tb + 8 * 256);
    rh2 = *reinterpret_cast<const uint4*>(tb + 16 * 256); rh3 = *reinterpret_cast<const uint4*>(tb + 24 * 256);
    rh4 = *reinterpret_cast<const uint4*>(tb + 32 * 256); rh5 = *reinterpret_cast<const uint4*>(tb + 40 * 256);
    rh6 = *reinterpret_cast<const uint4*>(tb + 48 * 256); rh7 = *reinterpret_cast<const uint4*>(tb + 56 * 256);
    if (TR != 0) {
      ract = s.act;
      const bool has_aff = s.scale != nullptr;
      const float* scp = has_aff ? s.scale : g_ident_scale;
      const float* shp = has_aff ? s.shift : g_ident_shift;
      const int img = (int)(((long long)t * 64) / HW);                         // a tile never spans two images (HW % 64 == 0)
      const size_t o = (size_t)img * (has_aff ? (unsigned)s.ss : 0u) + scol * 8;
      tsc[0] = *reinterpret_cast<const float4*>(scp + o); tsc[1] = *reinterpret_cast<const float4*>(scp + o + 4);
      tsh[0] = *reinterpret_cast<const float4*>(shp + o); tsh[1] = *reinterpret_cast<const float4*>(shp + o + 4);
    }
  };
  auto xf8 = [&](uint4 v) -> uint4 {
    if (TR == 0) return v;
    if (TR == 1) return mseg_affine8_bf16(v, ract == MSEG_ACT_RELU ? 0u : 0x80008000u, tsc[0], tsh[0], tsc[1], tsh[1], true);
    const float lo = (ract == MSEG_ACT_RELU) ? 0.f : -3.402823466e38f;
    float4 a = bf16x4_to_f32(make_uint2(v.x, v.y)), b = bf16x4_to_f32(make_uint2(v.z, v.w));
    if (TR == 2) { a = act_fwd4(a, ract); b = act_fwd4(b, ract); }
    else {
      a.x = clamp_lo(a.x, lo); a.y = clamp_lo(a.y, lo); a.z = clamp_lo(a.z, lo); a.w = clamp_lo(a.w, lo);
      b.x = clamp_lo(b.x, lo); b.y = clamp_lo(b.y, lo); b.z = clamp_lo(b.z, lo); b.w = clamp_lo(b.w, lo);
    }
    a.x = a.x * tsc[0].x + tsh[0].x; a.y = a.y * tsc[0].y + tsh[0].y; a.z = a.z * tsc[0].z + tsh[0].z; a.w = a.w * tsc[0].w + tsh[0].w;
    b.x = b.x * tsc[1].x + tsh[1].x; b.y = b.y * tsc[1].y + tsh[1].y; b.z = b.z * tsc[1].z + tsh[1].z; b.w = b.w * tsc[1].w + tsh[1].w;
    const uint2 pa = f32x4_to_bf16(a), pb = f32x4_to_bf16(b);
    return make_uint4(pa.x, pa.y, pb.x, pb.y);
  };
  auto commit_rows = [&]() {
    __bf16* const ab_ = Al + srow * CTD_ASTRIDE + scol * 8;
    *reinterpret_cast<uint4*>(ab_) = xf8(rh0); *reinterpret_cast<uint4*>(ab_ + 8 * CTD_ASTRIDE) = xf8(rh1);
    *reinterpret_cast<uint4*>(ab_ + 16 * CTD_ASTRIDE) = xf8(rh2); *reinterpret_cast<uint4*>(ab_ + 24 * CTD_ASTRIDE) = xf8(rh3);
    *reinterpret_cast<uint4*>(ab_ + 32 * CTD_ASTRIDE) = xf8(rh4); *reinterpret_cast<uint4*>(ab_ + 40 * CTD_ASTRIDE) = xf8(rh5);
    *reinterpret_cast<uint4*>(ab_ + 48 * CTD_ASTRIDE) = xf8(rh6); *reinterpret_cast<uint4*>(ab_ + 56 * CTD_ASTRIDE) = xf8(rh7);
  };
  auto store_tile = [&](f32x16 (&acc)[2], int t) {
    const int p0 = t * 64 + wm * 32;                    // first input pixel of this wave's 32-row block
    const int img = p0 / HW;
    const int rem = p0 - img * HW;
    const int y = rem / W, x0 = rem - y * W;
    const long long pos = ((((long long)img * 2 * H + 2 * y + (ab >> 1)) * 2 * W) + 2 * x0 + (ab & 1)) * Cq + wn * 64 + li;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const long long base = pos + 32 * nb;
      const float bv = bias[nb];
      if (D16) {
        __bf16* const d = reinterpret_cast<__bf16*>(p.dst0) + base - (odd ? 1 : 0);
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          const float mine_e = acc[nb][2 * e] + bv, mine_o = acc[nb][2 * e + 1] + bv;
          const float give = odd ? mine_e : mine_o, keep = odd ? mine_o : mine_e;
          const float got = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, give), 0xB1, 0xf, 0xf, true));
          *reinterpret_cast<unsigned*>(d + eoff[e] + (x0 + erow[e] >= W ? wrap : 0)) = pack_bf16x2(odd ? got : keep, odd ? keep : got);
        }
      } else {
        float* const d = reinterpret_cast<float*>(p.dst0) + base;
#pragma unroll
        for (int e = 0; e < NE; ++e) d[eoff[e] + (x0 + erow[e] >= W ? wrap : 0)] = acc[nb][e] + bv;
      }
    }
  };

  const int stride = 2 * nstreams;
  int t = 2 * stream + grp;
  unsigned gb_target = 0u;
  unsigned* const gcnt = &gcount[grp];
  if (t < ntiles) {
    issue_rows(t);
    commit_rows();
    if (t + stride < ntiles) issue_rows(t + stride);
    c64_group_barrier(gcnt, gb_target);
  }
  const int a_off = (wm * 32 + li) * CTD_ASTRIDE + lh * 8;
  const int b_off = (wn * 64 + li) * CTD_ASTRIDE + lh * 8;
  for (; t < ntiles; t += stride) {
    const int next = t + stride;
    f32x16 acc[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const bf16x8 af = *reinterpret_cast<const bf16x8*>(Al + a_off + kk * 16);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const bf16x8 bf = *reinterpret_cast<const bf16x8*>(Wl + b_off + nb * 32 * CTD_ASTRIDE + kk * 16);
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[nb], 0, 0, 0);
      }
    }
    c64_group_barrier(gcnt, gb_target);
    if (next < ntiles) {
      commit_rows();
      if (next + stride < ntiles) issue_rows(next + stride);
    }
    store_tile(acc, t);
    c64_group_barrier(gcnt, gb_target);
  }
}

// ---- ... and for its data gradient: Conv2d(64 -> 128, 2, stride 2) over dz ------------------------------------------------
// dx[p][ci] = sum over the 2 x 2 output positions (a, b) and co of dz[2 p + (a, b)][co] * W[ci][co][a][b]: M = low-resolution
// pixels, K = 4 taps x 64 channels, N = 128.  Resident weights (4 x 128 rows of 64 k), tiles of 64 low-resolution pixels =
// two 32-pixel blocks inside an image row, each staged with all four taps ([64 rows][256 k], 528-byte pitch): the two source
// pixels of a row pair are adjacent in memory (256 contiguous bytes per pixel and image row).  A wave owns 32 pixels x 64
// columns.  Plain operand (dz), one destination, no accumulation — anything else stays on the gather kernel.

template <bool D16>
__global__ __launch_bounds__(512, 1) void igemm_ctd_bf16_kernel(const MsegIgemm p, int ntiles) {
  constexpr int HL = 8;                               // staging passes of a group: 64 rows x 4 taps x 8 column groups / 256
  constexpr int NE = D16 ? 8 : 16;
  __shared__ __attribute__((aligned(16))) __bf16 lds[4 * 128 * C64_STRIDE + 2 * 64 * CTD_ASTRIDE];
  __shared__ unsigned gcount[2];
  __bf16* const Wl = lds;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = wave >> 2;
  const int gtid = tid & 255;
  __bf16* const Al = lds + 4 * 128 * C64_STRIDE + grp * 64 * CTD_ASTRIDE;
  const int wm = (wave >> 1) & 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int Hi = p.Hi, Wi = p.Wi, Ho = p.Ho, Wo = p.Wo, HWo = Ho * Wo;
  const int ld = p.ld0;
  const __bf16* const srcp = reinterpret_cast<const __bf16*>(p.src[0].ptr);

  {
    const __bf16* const wp = reinterpret_cast<const __bf16*>(p.w);   // [4 taps][Npad][Kpad = 64]
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = tid + 512 * i;                   // 4 x 128 rows x 8 column groups = 4096
      const int row = idx >> 3, c8 = idx & 7;
      const int tap = row >> 7, n = row & 127;
      const uint4 v = *reinterpret_cast<const uint4*>(wp + ((size_t)tap * p.Npad + n) * p.Kpad + c8 * 8);
      *reinterpret_cast<uint4*>(Wl + row * C64_STRIDE + c8 * 8) = v;
    }
    if (tid < 2) gcount[tid] = 0u;
  }
  __syncthreads();

  // staging: load j of a thread is (row = (gtid >> 5) + 8 j, tap = (gtid >> 3) & 3, 8 channels c8 = gtid & 7); rows 0..31
  // (j < 4) belong to the tile's first 32-pixel block, the others to the second
  const int s_c8 = gtid & 7, s_tap = (gtid >> 3) & 3, s_r0 = gtid >> 5;
  int roff[HL], aoff[HL];
#pragma unroll
  for (int j = 0; j < HL; ++j) {
    const int row = s_r0 + 8 * j;
    roff[j] = (((s_tap >> 1) * Wi) + 2 * (row & 31) + (s_tap & 1)) * 64 + s_c8 * 8;   // from the block's first source pixel
    aoff[j] = row * CTD_ASTRIDE + s_tap * 64 + s_c8 * 8;
  }
  const int n_out = wn * 64 + li;                      // + 32 nb
  const bool odd = li & 1;
  int eoff[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int r = D16 ? 2 * e + (odd ? 1 : 0) : e;
    eoff[e] = ((r & 3) + 8 * (r >> 2) + 4 * lh) * ld;
  }

  uint4 rh0, rh1, rh2, rh3, rh4, rh5, rh6, rh7;       // (scalars: as an array the compiler kept them in scratch memory)
  auto issue_rows = [&](int t) {
    auto block_base = [&](int p0) -> const __bf16* {    // first source pixel of a 32-pixel block (wave-uniform)
      const int img = p0 / HWo;
      const int rem = p0 - img * HWo;
      const int y = rem / Wo, x0 = rem - y * Wo;
      return srcp + (((long long)img * Hi + 2 * y) * Wi + 2 * x0) * 64;
    };
    const __bf16* const bb0 = block_base(t * 64);
    const __bf16* const bb1 = block_base(t * 64 + 32);
    rh0 = *reinterpret_cast<const uint4*>(bb0 + roff[0]); rh1 = *reinterpret_cast<const uint4*>(bb0 + roff[1]);
    rh2 = *reinterpret_cast<const uint4*>(bb0 + roff[2]); rh3 = *reinterpret_cast<const uint4*>(bb0 + roff[3]);
    rh4 = *reinterpret_cast<const uint4*>(bb1 + roff[4]); rh5 = *reinterpret_cast<const uint4*>(bb1 + roff[5]);
    rh6 = *reinterpret_cast<const uint4*>(bb1 + roff[6]); rh7 = *reinterpret_cast<const uint4*>(bb1 + roff[7]);
  };
  auto commit_rows = [&]() {
    *reinterpret_cast<uint4*>(Al + aoff[0]) = rh0; *reinterpret_cast<uint4*>(Al + aoff[1]) = rh1;
    *reinterpret_cast<uint4*>(Al + aoff[2]) = rh2; *reinterpret_cast<uint4*>(Al + aoff[3]) = rh3;
    *reinterpret_cast<uint4*>(Al + aoff[4]) = rh4; *reinterpret_cast<uint4*>(Al + aoff[5]) = rh5;
    *reinterpret_cast<uint4*>(Al + aoff[6]) = rh6; *reinterpret_cast<uint4*>(Al + aoff[7]) = rh7;
  };
  auto store_tile = [&](f32x16 (&acc)[2], int t) {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int n = n_out + 32 * nb;                    // < Ngemm = 128 (launch rule)
      const long long base = (long long)(t * 64 + wm * 32) * ld + n;
      if (D16) {
        __bf16* const d = reinterpret_cast<__bf16*>(p.dst0) + base - (odd ? 1 : 0);
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          const float mine_e = acc[nb][2 * e], mine_o = acc[nb][2 * e + 1];
          const float give = odd ? mine_e : mine_o, keep = odd ? mine_o : mine_e;
          const float got = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, give), 0xB1, 0xf, 0xf, true));
          *reinterpret_cast<unsigned*>(d + eoff[e]) = pack_bf16x2(odd ? got : keep, odd ? keep : got);
        }
      } else {
        float* const d = reinterpret_cast<float*>(p.dst0) + base;
#pragma unroll
        for (int e = 0; e < NE; ++e) d[eoff[e]] = acc[nb][e];
      }
    }
  };

  const int stride = 2 * (int)gridDim.x;
  int t = 2 * (int)xcd_logical_id(blockIdx.x, gridDim.x) + grp;
  unsigned gb_target = 0u;
  unsigned* const gcnt = &gcount[grp];
  if (t < ntiles) {
    issue_rows(t);
    commit_rows();
    if (t + stride < ntiles) issue_rows(t + stride);
    c64_group_barrier(gcnt, gb_target);
  }
  const int a_off = (wm * 32 + li) * CTD_ASTRIDE + lh * 8;
  const int b_off = (wn * 64 + li) * C64_STRIDE + lh * 8;
  for (; t < ntiles; t += stride) {
    const int next = t + stride;
    f32x16 acc[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
#pragma unroll
    for (int tap = 0; tap < 4; ++tap)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(Al + a_off + tap * 64 + kk * 16);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const bf16x8 bf = *reinterpret_cast<const bf16x8*>(Wl + b_off + (tap * 128 + nb * 32) * C64_STRIDE + kk * 16);
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[nb], 0, 0, 0);
        }
      }
    c64_group_barrier(gcnt, gb_target);
    if (next < ntiles) {
      commit_rows();
      if (next + stride < ntiles) issue_rows(next + stride);
    }
    store_tile(acc, t);
    c64_group_barrier(gcnt, gb_target);
  }
}

// ---- bf16 variant of the gather kernel (stride-2 convolutions, ConvTranspose as a 1x1 GEMM, their data gradients) ----
// igemm_fast_kernel with bf16 matrix-core inputs: same per-row offsets / tap masks / live-tap list and epilogue; the staged
// source pixels are rounded to bf16 after the norm-on-load transform, the weights arrive as bf16, LDS rows are 32 bf16 +
// pad (80 B), one K-step (one tap x 32 channels) is two v_mfma_f32_32x32x16_bf16 per 32 x 32 block.
// S16: bf16 source tensors; a staging thread then owns 8 channels of a row (one 16-byte load), half as many passes.
template <int BM, int BN, int TR, bool PER_SAMPLE, bool S16>
__global__ __launch_bounds__(256, PER_SAMPLE ? 2 : 4) void igemm_fast_bf16_kernel(const MsegIgemm p) {
  using Cfg = IgemmCfg<BM, BN>;
  constexpr int STAGE = (BM + BN) * HB_STRIDE;
  constexpr int BPASS = BN / 64;                       // weight staging: 64 rows x 4 groups of 8 channels per pass
  __shared__ __attribute__((aligned(16))) __bf16 lds[2 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const int M = p.NB * p.Ho * p.Wo;
  const int ntiles_n = (p.Ngemm + BN - 1) / BN;
  const int lid = (int)xcd_logical_id(blockIdx.x, gridDim.x);
  const int tile_m = lid / ntiles_n;
  const int tile_n = lid - tile_m * ntiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int T = p.KH * p.KW;
  constexpr int SQ = S16 ? 4 : 8;                      // staging threads per source row (8 / 4 channels each)
  constexpr int RS = 256 / SQ;                         // rows per staging pass
  constexpr int AR = BM / RS;                          // staging passes (4; 2 for bf16 sources)
  constexpr int CPT = S16 ? 8 : 4, ESZ = S16 ? 2 : 4;
  const int srow = tid / SQ, scol = tid % SQ;
  const int brow = tid >> 2, bcol = tid & 3;

  const TapGeom geom = make_geom(p);
  const long long tile_px = tile_base_pixel(p, geom, m0, M);
  int pix0[AR];
  int rown[AR];
  unsigned vmask[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const RowInfo r = decode_row(p, m0 + srow + RS * i, M);
    const int iy0 = (r.oy * geom.sm - geom.dir * geom.pad) >> geom.sh, ix0 = (r.ox * geom.sm - geom.dir * geom.pad) >> geom.sh;
    pix0[i] = (int)(((long long)r.n * p.Hi + iy0) * p.Wi + ix0 - tile_px);   // relative to the tile's descriptor base
    rown[i] = r.n < 0 ? 0 : r.n;
    unsigned mk = 0u;
    for (int t = 0; t < T; ++t) {
      const int ky = t / p.KW, kx = t - ky * p.KW;
      int iy, ix;
      mk |= (unsigned)tap_coord(geom, r, ky, kx, iy, ix) << t;
    }
    vmask[i] = mk;
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
  const unsigned OOB = 0x80000000u;
  // based at the tile's first source pixel; every load is masked by vmask (dead rows use OOB), so the record count only
  // has to exceed the tile's span (host-checked to stay below 2 GiB)
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>((const char*)p.src[0].ptr + tile_px * p.src[0].C * ESZ), 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>((const char*)(p.nsrc > 1 ? p.src[1].ptr : p.src[0].ptr) +
                        tile_px * (p.nsrc > 1 ? p.src[1].C : p.src[0].C) * ESZ), 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0,
                                                                        T * p.Npad * p.Kpad * 2, 0x00020000);
  unsigned wvoff[BPASS];
#pragma unroll
  for (int i = 0; i < BPASS; ++i) wvoff[i] = ((unsigned)(n0 + brow + 64 * i) * (unsigned)p.Kpad + bcol * 8u) * 2u;

  constexpr int NSC = (TR == 0) ? 1 : (PER_SAMPLE ? AR : 1);
  f32x4 ra[AR];                                        // 16 raw bytes: 4 fp32 or 8 bf16 source channels
  float4 rsc[NSC], rsh[NSC], rsc2[NSC], rsh2[NSC];
  f32x4 rb[BPASS];
  float rm[AR];
  unsigned rowoff[AR];
  int ract = 0;
  int cur_chunk = -1;
  bool cur_s1 = false;

  auto issue = [&](int chunk, int t, int ky, int kx) {
    const int c = chunk * KC + scol * CPT;
    if (chunk != cur_chunk) {
      cur_chunk = chunk;
      cur_s1 = (p.nsrc > 1) && (chunk * KC >= C0);
      const unsigned sCB = (unsigned)(cur_s1 ? p.src[1].C : p.src[0].C) * (unsigned)ESZ;      // bytes per pixel
      const unsigned cl = (unsigned)(cur_s1 ? c - C0 : c);
#pragma unroll
      for (int i = 0; i < AR; ++i) rowoff[i] = (unsigned)pix0[i] * sCB + cl * (unsigned)ESZ;
      if (TR != 0) {
        const MsegSrc& s = cur_s1 ? p.src[1] : p.src[0];
        ract = s.act;
        const bool has_aff = s.scale != nullptr;
        const float* scp = has_aff ? s.scale : g_ident_scale;
        const float* shp = has_aff ? s.shift : g_ident_shift;
        const unsigned clv = (c < p.Cin) ? cl : 0u;
#pragma unroll
        for (int i = 0; i < NSC; ++i) {
          const size_t o = (size_t)(PER_SAMPLE ? rown[i] : 0) * (has_aff ? (unsigned)s.ss : 0u) + clv;
          rsc[i] = *reinterpret_cast<const float4*>(scp + o);
          rsh[i] = *reinterpret_cast<const float4*>(shp + o);
          if (S16) {
            rsc2[i] = *reinterpret_cast<const float4*>(scp + o + 4);
            rsh2[i] = *reinterpret_cast<const float4*>(shp + o + 4);
          }
        }
      }
    }
    const unsigned cbit = (c < p.Cin) ? (1u << t) : 0u;
    const unsigned sCB = (unsigned)(cur_s1 ? p.src[1].C : p.src[0].C) * (unsigned)ESZ;
    const unsigned delta = (unsigned)(geom.dir * ((ky >> geom.sh) * p.Wi + (kx >> geom.sh))) * sCB;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const bool ok = (vmask[i] & cbit) != 0u;
      const unsigned vo = ok ? rowoff[i] + delta : OOB;
      ra[i] = cur_s1 ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, vo, 0, 0))
                     : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs0, vo, 0, 0));
      if (TR != 0) rm[i] = ok ? 1.f : 0.f;
    }
    const unsigned wso = ((unsigned)t * (unsigned)p.Npad * (unsigned)p.Kpad + (unsigned)chunk * KC) * 2u;
#pragma unroll
    for (int i = 0; i < BPASS; ++i)
      rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, wvoff[i], wso, 0));
  };

  auto xform4 = [&](float4 v, const float4& sc, const float4& sh, float lo, float m) -> float4 {
    if (TR == 2) v = act_fwd4(v, ract);
    else { v.x = clamp_lo(v.x, lo); v.y = clamp_lo(v.y, lo); v.z = clamp_lo(v.z, lo); v.w = clamp_lo(v.w, lo); }
    v.x = (v.x * sc.x + sh.x) * m; v.y = (v.y * sc.y + sh.y) * m;
    v.z = (v.z * sc.z + sh.z) * m; v.w = (v.w * sc.w + sh.w) * m;
    return v;
  };

  auto commit = [&](__bf16* As, __bf16* Bs) {
    const float lo = (ract == MSEG_ACT_RELU) ? 0.f : -3.402823466e38f;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const int ti = PER_SAMPLE ? i : 0;
      if (S16) {
        const uint4 raw = __builtin_bit_cast(uint4, ra[i]);
        if (TR == 0) {
          *reinterpret_cast<uint4*>(As + (srow + RS * i) * HB_STRIDE + scol * 8) = raw;
        } else if (TR == 1) {                          // ReLU / none + affine: the short form (bf16_affine.h)
          *reinterpret_cast<uint4*>(As + (srow + RS * i) * HB_STRIDE + scol * 8) =
              mseg_affine8_bf16(raw, ract == MSEG_ACT_RELU ? 0u : 0x80008000u, rsc[ti], rsh[ti], rsc2[ti], rsh2[ti], rm[i] != 0.f);
        } else {
          const float4 u = xform4(bf16x4_to_f32(make_uint2(raw.x, raw.y)), rsc[ti], rsh[ti], lo, rm[i]);
          const float4 w = xform4(bf16x4_to_f32(make_uint2(raw.z, raw.w)), rsc2[ti], rsh2[ti], lo, rm[i]);
          const uint2 pu = f32x4_to_bf16(u), pw = f32x4_to_bf16(w);
          *reinterpret_cast<uint4*>(As + (srow + RS * i) * HB_STRIDE + scol * 8) = make_uint4(pu.x, pu.y, pw.x, pw.y);
        }
      } else {
        float4 v = make_float4(ra[i][0], ra[i][1], ra[i][2], ra[i][3]);
        if (TR != 0) v = xform4(v, rsc[ti], rsh[ti], lo, rm[i]);
        bf16x4 h;
        h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
        *reinterpret_cast<bf16x4*>(As + (srow + RS * i) * HB_STRIDE + scol * 4) = h;
      }
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i)
      *reinterpret_cast<f32x4*>(Bs + (brow + 64 * i) * HB_STRIDE + bcol * 8) = rb[i];
  };

  unsigned long long taplist = 0ull, taplist_hi = 0ull;
  int nlive = 0;
  {
    int py = 0, px = 0;
    if (geom.sh) {
      const int cls = m0 / (M >> 2);
      py = cls >> 1; px = cls & 1;
    }
    for (int tt = 0; tt < T; ++tt) {
      const int ky_ = tt / p.KW, kx_ = tt - ky_ * p.KW;
      const bool live = (((py - geom.dir * (ky_ - geom.pad)) | (px - geom.dir * (kx_ - geom.pad))) & geom.sh) == 0;
      if (live) {
        const unsigned long long e = (unsigned long long)(tt | (ky_ << 4) | (kx_ << 6));
        if (nlive < 8) taplist |= e << (8 * nlive); else taplist_hi |= e << (8 * (nlive - 8));
        ++nlive;
      }
    }
  }
  auto tap_at = [&](int j) -> unsigned { return (unsigned)((j < 8 ? taplist >> (8 * j) : taplist_hi >> (8 * (j - 8))) & 0xffull); };
  int chunk = 0, j = 0;
  const int nsteps = nchunks * nlive;
  unsigned e0 = tap_at(0);
  issue(0, (int)(e0 & 15u), (int)((e0 >> 4) & 3u), (int)(e0 >> 6));
  commit(lds, lds + BM * HB_STRIDE);
  __syncthreads();
  int cur = 0;
  const int li = lane & 31, lh = lane >> 5;
  for (int step = 0; step < nsteps; ++step) {
    if (step + 1 < nsteps) {
      j += 1;
      if (j >= nlive) { j = 0; ++chunk; }
    }
    e0 = tap_at(j);
    issue(chunk, (int)(e0 & 15u), (int)((e0 >> 4) & 3u), (int)(e0 >> 6));

    const __bf16* As = lds + cur * STAGE;
    const __bf16* Bs = As + BM * HB_STRIDE;
#pragma unroll
    for (int kk = 0; kk < KC / 16; ++kk) {
      bf16x8 af[Cfg::MB], bf[Cfg::NB];
#pragma unroll
      for (int a = 0; a < Cfg::MB; ++a)
        af[a] = *reinterpret_cast<const bf16x8*>(As + (wm * Cfg::TM + a * 32 + li) * HB_STRIDE + kk * 16 + lh * 8);
#pragma unroll
      for (int b = 0; b < Cfg::NB; ++b)
        bf[b] = *reinterpret_cast<const bf16x8*>(Bs + (wn * Cfg::TN + b * 32 + li) * HB_STRIDE + kk * 16 + lh * 8);
#pragma unroll
      for (int a = 0; a < Cfg::MB; ++a)
#pragma unroll
        for (int b = 0; b < Cfg::NB; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
    }
    __bf16* An = lds + (cur ^ 1) * STAGE;
    commit(An, An + BM * HB_STRIDE);
    __syncthreads();
    cur ^= 1;
  }
  igemm_epilogue<IgemmCfg<BM, BN>>(acc, m0, n0, wm, wn, lane, M);
}

// out[m][n] = bias[n] + sum_k ws[k][m][n]  (fixed order), routed like the PLAIN epilogue: columns < split to dst0, the
// rest to dst1, each with its own leading dimension and accumulate flag.  One thread per 4 columns.
__global__ void igemm_splitk_reduce_kernel(const float* __restrict__ ws, int ksplit, size_t M, int N,
                                           const float* __restrict__ bias, float* __restrict__ dst0, int ld0, int acc0,
                                           float* __restrict__ dst1, int ld1, int acc1, int split, int d16) {
  const int N4 = N >> 2;
  const size_t total = M * (size_t)N4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t m = i / N4;
    const int n = (int)(i - m * N4) * 4;
    float4 s = *reinterpret_cast<const float4*>(ws + m * N + n);
    for (int k = 1; k < ksplit; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(ws + ((size_t)k * M + m) * N + n);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (bias) { s.x += bias[n]; s.y += bias[n + 1]; s.z += bias[n + 2]; s.w += bias[n + 3]; }
    const bool second = n >= split;                    // split % 4 == 0 (channel counts are multiples of 4)
    void* base = second ? (void*)dst1 : (void*)dst0;
    const size_t e = second ? m * ld1 + (n - split) : m * ld0 + n;
    if (second ? acc1 : acc0) {
      const float4 o = ld_f4_rt(base, e, d16);
      s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
    }
    st_f4_rt(base, e, s, d16);
  }
}

static int check_src(const MsegSrc& s) {
  if (!s.ptr || s.C <= 0 || (s.C & 3)) return MSEG_EINVAL;
  if (s.dtype != MSEG_ST_F32 && s.dtype != MSEG_ST_BF16) return MSEG_EINVAL;
  if (s.dtype == MSEG_ST_BF16 && (s.C & 7)) return MSEG_EINVAL;      // 8 bf16 channels per 16-byte staging load
  if ((s.scale == nullptr) != (s.shift == nullptr)) return MSEG_EINVAL;
  if (s.scale && (s.ss & 3)) return MSEG_EINVAL;
  return MSEG_OK;
}

// split-K of a halo launch: only when the tiles leave most workgroup slots (2 per CU) empty and every split keeps at
// least two 32-channel chunks; 1 = no split
static int halo_ksplit(long long htiles, int nchunks, int* chunks_per_split) {
  int ks = 1;
  if (htiles < 256 && nchunks >= 4) {
    long long want = (512 + htiles - 1) / htiles;
    ks = (int)(want < 16 ? want : 16);
    if (ks > nchunks / 2) ks = nchunks / 2;
    if (ks < 1) ks = 1;
  }
  *chunks_per_split = (nchunks + ks - 1) / ks;
  return (nchunks + *chunks_per_split - 1) / *chunks_per_split;     // splits that actually get chunks
}

static int g_c64p_on = 1;
// Test / ablation hook: 0 sends the 64 -> 64 channel bf16 layers back to the tile-per-workgroup kernel.  Process-wide.
extern "C" int mseg_igemm_set_persistent(int on) {
  g_c64p_on = on ? 1 : 0;
  return MSEG_OK;
}

// one persistent workgroup per compute unit of the current device
static int c64p_workgroups() {
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (!cached[dev]) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev] = n;
  }
  return cached[dev];
}

// 512-pixel tiles (igemm_halo_bf16m512_kernel): tile count, or 0 when more than a fifth of the tile rows would fall below
// the image
static long long halo_big_tiles(const MsegIgemm& p, int tw_log2, int BMv) {
  const int TH = BMv >> tw_log2;
  const long long rows = (long long)((p.Hi + TH - 1) / TH) * TH;
  if ((long long)p.Hi * 5 < rows * 4) return 0;
  return (long long)p.NB * ((p.Hi + TH - 1) / TH) * (p.Wi >> tw_log2);
}
static long long halo512_tiles(const MsegIgemm& p, int tw_log2) { return halo_big_tiles(p, tw_log2, 512); }

static int g_w4m_on = 1;
// Test / ablation hook: 0 sends the 128-channel-tile bf16 layers back to 128-pixel tiles.  Process-wide.
extern "C" int mseg_igemm_set_wide_tiles(int on) {
  g_w4m_on = on ? 1 : 0;
  return MSEG_OK;
}

static bool halo_geometry(const MsegIgemm& p, int BNv, int* tw_log2_out, long long* htiles_out) {
  if (!(p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && p.Ho == p.Hi && p.Wo == p.Wi &&
        p.epi == MSEG_EPI_PLAIN && p.morder == MSEG_MORDER_LINEAR && (p.Wi % 4) == 0))
    return false;
  int tw_log2 = p.precision == MSEG_PREC_BF16 ? 5 : 6;             // the bf16 kernel's tiles are at most 32 wide
  while ((p.Wi & ((1 << tw_log2) - 1)) != 0) --tw_log2;           // largest power of two <= 64 dividing W
  const int TH = 128 >> tw_log2;
  const long long mt = (long long)p.NB * ((p.Hi + TH - 1) / TH) * (p.Wi >> tw_log2);
  *htiles_out = mt * ((p.Ngemm + BNv - 1) / BNv);
  *tw_log2_out = tw_log2;
  // rows of the 128-pixel tile that fall below the image are wasted matrix work (e.g. 20 x 20 images: 4 x 32 tiles
  // cover 32 rows for 20); below 80 % the linear-M gather kernel, which wastes nothing, is the faster choice
  // (the bf16 kernel is ~6x faster per tile: it stays the better choice down to 50 %)
  if (p.precision == MSEG_PREC_BF16) return (long long)p.Hi * 2 >= (long long)((p.Hi + TH - 1) / TH) * TH;
  return (long long)p.Hi * 5 >= (long long)((p.Hi + TH - 1) / TH) * TH * 4;
}

extern "C" size_t mseg_igemm_workspace_bytes(const MsegIgemm* pp) {
  if (!pp) return 0;
  const MsegIgemm& p = *pp;
  if (p.Ngemm <= 0 || p.Cin <= 0 || p.NB <= 0 || p.Hi <= 0 || p.Wi <= 0) return 0;
  int tw_log2 = 0, cps = 0;
  long long htiles = 0;
  if (!halo_geometry(p, p.Ngemm > 64 ? 128 : 64, &tw_log2, &htiles)) return 0;
  const int ks = halo_ksplit(htiles, (p.Cin + KC - 1) / KC, &cps);
  if (ks <= 1 || (p.Ngemm & 3)) return 0;
  return (size_t)ks * p.NB * p.Hi * p.Wi * p.Ngemm * sizeof(float);
}

static int igemm_dispatch(const MsegIgemm* pp, void* stream) {
  if (!pp) return MSEG_EINVAL;
  const MsegIgemm& p = *pp;
  if (p.nsrc < 1 || p.nsrc > 2) return MSEG_EINVAL;
  int csum = 0;
  for (int i = 0; i < p.nsrc; ++i) {
    if (check_src(p.src[i])) return MSEG_EINVAL;
    csum += p.src[i].C;
  }
  if (csum != p.Cin) return MSEG_EINVAL;
  if (!p.w || !p.dst0 || p.Kpad < p.Cin || (p.Kpad % KC) || p.Npad < p.Ngemm || (p.Npad % 128)) return MSEG_EINVAL;
  if (p.Cin > MSEG_MAX_CH) return MSEG_EINVAL;
  if (p.NB <= 0 || p.Hi <= 0 || p.Wi <= 0 || p.Ho <= 0 || p.Wo <= 0) return MSEG_EINVAL;
  if (p.KH <= 0 || p.KW <= 0 || p.KH * p.KW > 16) return MSEG_EINVAL;
  if (p.stride != 1 && p.stride != 2) return MSEG_EINVAL;
  if (p.mode != MSEG_MODE_CONV && p.mode != MSEG_MODE_TCONV) return MSEG_EINVAL;
  if (p.morder == MSEG_MORDER_PARITY && ((p.Ho | p.Wo) & 1)) return MSEG_EINVAL;
  if (p.Ngemm <= 0) return MSEG_EINVAL;
  if (p.precision != MSEG_PREC_F32 && p.precision != MSEG_PREC_BF16) return MSEG_EINVAL;
  // bf16 tensor storage (sources and / or destinations) exists for the bf16 matrix-core kernels only
  if (p.dst_dtype != MSEG_ST_F32 && p.dst_dtype != MSEG_ST_BF16) return MSEG_EINVAL;
  const bool s16 = p.src[0].dtype == MSEG_ST_BF16;
  if (p.nsrc > 1 && p.src[1].dtype != p.src[0].dtype) return MSEG_EINVAL;
  if ((s16 || p.dst_dtype == MSEG_ST_BF16) && p.precision != MSEG_PREC_BF16) return MSEG_EINVAL;
  if (p.dst_dtype == MSEG_ST_BF16) {        // the epilogue stores channel PAIRS (4 bytes) of a bf16 destination
    if ((p.Ngemm | p.ld0 | p.ld1 | p.Cq) & 1) return MSEG_EINVAL;
    if (p.split < p.Ngemm && (p.split & 1)) return MSEG_EINVAL;
    if (((uintptr_t)p.dst0 | (uintptr_t)p.dst1) & 3) return MSEG_EINVAL;
  }
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
  mseg_dispatch_note(p.precision, 0);
  // one-off, idempotent fill of the per-device identity tables (concurrent first calls write the same values)
  static bool ident_ready[64] = {false};
  int devid = 0;
  const bool dry = mseg_dispatch_dry() != 0;          // a query: nothing is launched, no device is needed
  if (!dry) {
    if (hipGetDevice(&devid) != hipSuccess || devid < 0 || devid >= 64) return MSEG_ELAUNCH;
    if (!ident_ready[devid]) {
      MSEG_KL_AUX(init_ident_kernel, dim3(8), dim3(256), 0, st);
      MSEG_LAUNCH_CHECK();
      ident_ready[devid] = true;
    }
  }
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
  // fast path preconditions (see igemm_fast_kernel): 32-bit buffer offsets over the source rows of ONE TILE (< 2 GiB)
  bool fast = (p.mode == MSEG_MODE_CONV) || (p.stride == 1) ||
              (p.morder == MSEG_MORDER_PARITY && (M % (4 * BMv)) == 0);   // s2 dgrad: parity-uniform tiles
  bool plain = true, band_fits = true;
  for (int i = 0; i < p.nsrc; ++i) {
    // gather kernels: descriptors per tile; the span of a tile = the source rows 128 consecutive output pixels read
    if (((128LL / p.Wo + 3) * p.stride + p.KH) * p.Wi * p.src[i].C * 4 >= 0x7ffffff0LL) fast = false;
    if (34LL * p.Wi * p.src[i].C * 4 >= 0x80000000LL) band_fits = false;   // halo row band: at most 32 + 2 rows
    if (p.src[i].act != MSEG_ACT_NONE || p.src[i].scale) plain = false;
  }
  bool common = true;                                                // shared by the fast and the halo kernel
  if ((long long)p.KH * p.KW * p.Npad * p.Kpad * 4 >= 0x80000000LL) common = false;
  if (p.KH > 4 || p.KW > 4) common = false;                          // packed tap list: 2 bits per tap coordinate
  if (p.nsrc > 1 && (p.src[0].C % KC)) common = false;
  fast = fast && common;
  // halo kernel: 3x3, stride 1, pad 1 (forward of every ConvBlock conv and its data gradient).  Its descriptors span the
  // row band of a tile's halo, so only (TH + 2) image rows of an operand have to stay below 2 GiB: any batch size and any
  // frame size (the reference pads frames up to 8192 x 8192, utils.py:137-138) keep the fast path.
  int tw_log2 = 6;
  long long htiles = 0;
  if (common && band_fits && halo_geometry(p, BNv, &tw_log2, &htiles) && htiles <= 0x7fffffffLL) {
    const int tr = plain ? 0 : (generic ? 2 : 1);
    // split-K over the input-channel chunks when the tiles would leave most of the chip idle (small batches, deep levels)
    int cps = 0;
    int ks = halo_ksplit(htiles, (p.Cin + KC - 1) / KC, &cps);
    const size_t need = (size_t)ks * (size_t)M * (size_t)p.Ngemm * sizeof(float);
    if (ks > 1 && (!p.ws || p.ws_bytes < need || (p.Ngemm & 3) || (p.split & 3) || htiles * ks > 0x7fffffffLL)) {
      ks = 1;
      cps = (p.Cin + KC - 1) / KC;
    }
    MsegIgemm q = p;                                   // descriptor of the partial launch: plain stores into the scratch
    if (ks > 1) {
      q.dst0 = (float*)p.ws; q.dst1 = nullptr; q.ld0 = p.Ngemm; q.ld1 = 0; q.split = p.Ngemm; q.acc0 = 0; q.acc1 = 0;
      q.bias = nullptr;
      q.dst_dtype = MSEG_ST_F32;                       // partial sums stay fp32; the reduction rounds once
    }
    mseg_dispatch_note(p.precision, ks > 1 ? need : 0);
    const dim3 hgrid((unsigned)(htiles * ks));
#define MSEG_HALO(BN_, TR_) MSEG_KL((igemm_halo_kernel<BN_, TR_>), hgrid, dim3(512), 0, st, q, tw_log2, ks, cps)
    // weight tensor beyond ~2 MB (bf16) and several N tiles: pixel tiles fastest (see the kernel)
    const int m_fastest = ((long long)9 * p.Kpad * p.Ngemm * 2 > (2ll << 20)) && (p.Ngemm > BNv) ? 1 : 0;
#define MSEG_HALO16(BN_, TR_)                                                                                          \
  do {                                                                                                                 \
    if (s16) MSEG_KL((igemm_halo_bf16_kernel<BN_, TR_, true>), hgrid, dim3(512), 0, st, q, tw_log2, ks, cps, m_fastest); \
    else MSEG_KL((igemm_halo_bf16_kernel<BN_, TR_, false>), hgrid, dim3(512), 0, st, q, tw_log2, ks, cps, m_fastest);    \
  } while (0)
#define MSEG_HALO16W4(TR_)                                                                                             \
  do {                                                                                                                 \
    if (s16) MSEG_KL((igemm_halo_bf16w4_kernel<TR_, true>), hgrid, dim3(256), 0, st, q, tw_log2, ks, cps, m_fastest);    \
    else MSEG_KL((igemm_halo_bf16w4_kernel<TR_, false>), hgrid, dim3(256), 0, st, q, tw_log2, ks, cps, m_fastest);       \
  } while (0)
    int p8rc = 0;
    if (p.precision == MSEG_PREC_BF16 && s16 && !wide && g_c64p_on && p.nsrc == 1 && p.Cin == 64 && p.Kpad == 64 &&
        ks == 1 && tw_log2 <= 5 && p.split >= p.Ngemm && htiles >= 4 * (long long)c64p_workgroups()) {
      // 64 -> 64 channels on bf16 tensors (level 0): persistent workgroups that keep the layer's weights in LDS
      const int wgs = c64p_workgroups();
      const dim3 pgrid((unsigned)wgs);                      // htiles >= 4 per workgroup: every group has work
#define MSEG_C64P(TR_)                                                                                                        \
  do {                                                                                                                         \
    if (q.dst_dtype == MSEG_ST_BF16) MSEG_KL((igemm_c64p_bf16_kernel<TR_, true>), pgrid, dim3(512), 0, st, q, tw_log2, (int)htiles); \
    else MSEG_KL((igemm_c64p_bf16_kernel<TR_, false>), pgrid, dim3(512), 0, st, q, tw_log2, (int)htiles);               \
  } while (0)
      if (tr == 0) MSEG_C64P(0); else if (tr == 1) MSEG_C64P(1); else MSEG_C64P(2);
#undef MSEG_C64P
    } else if (p.precision == MSEG_PREC_BF16 && !wide && g_c64p_on && p.Cin >= 128 && ks == 1 && tw_log2 >= 4 &&
               halo512_tiles(p, tw_log2) >= 2 * (long long)c64p_workgroups()) {
      // few output channels, many input channels (the 128 -> 64 concat convolutions of level 0): 512-pixel tiles
      const dim3 mgrid((unsigned)halo512_tiles(p, tw_log2));
#define MSEG_HALO512(TR_)                                                                                                      \
  do {                                                                                                                         \
    if (s16) MSEG_KL((igemm_halo_bf16m512_kernel<TR_, true>), mgrid, dim3(512), 0, st, q, tw_log2, 1, cps, 0);      \
    else MSEG_KL((igemm_halo_bf16m512_kernel<TR_, false>), mgrid, dim3(512), 0, st, q, tw_log2, 1, cps, 0);         \
  } while (0)
      if (tr == 0) MSEG_HALO512(0); else if (tr == 1) MSEG_HALO512(1); else MSEG_HALO512(2);
#undef MSEG_HALO512
    } else if (p.precision == MSEG_PREC_BF16 && wide && s16 && ks == 1 &&
               (p8rc = igemm_p8_try(q, tr, m_fastest, c64p_workgroups(), st)) != 0) {
      // >= 128 output channels on bf16 tensors with at least one 256-pixel tile per CU: igemm_p8.hip
      if (p8rc < 0) return p8rc;
    } else if (p.precision == MSEG_PREC_BF16 && wide && s16 && g_w4m_on && ks == 1 && tw_log2 >= 3 &&
               halo_big_tiles(p, tw_log2, 256) * ((p.Ngemm + 127) / 128) >= 2 * (long long)c64p_workgroups()) {
      // 128-channel tiles on enough pixels: 256-pixel tiles (fewer LDS reads and weight bytes per MFMA)
      const dim3 mgrid((unsigned)(halo_big_tiles(p, tw_log2, 256) * ((p.Ngemm + 127) / 128)));
      // (bf16 tensors only: with fp32 sources the 11 staging loads per thread do not fit beside 128 accumulator registers)
#define MSEG_HALOW4M(TR_) \
  MSEG_KL((igemm_halo_bf16w4m_kernel<TR_, true>), mgrid, dim3(256), 0, st, q, tw_log2, 1, cps, m_fastest)
      if (tr == 0) MSEG_HALOW4M(0); else if (tr == 1) MSEG_HALOW4M(1); else MSEG_HALOW4M(2);
#undef MSEG_HALOW4M
    } else if (p.precision == MSEG_PREC_BF16) {
      if (wide) { if (tr == 0) MSEG_HALO16W4(0); else if (tr == 1) MSEG_HALO16W4(1); else MSEG_HALO16W4(2); }
      else      { if (tr == 0) MSEG_HALO16(64, 0); else if (tr == 1) MSEG_HALO16(64, 1); else MSEG_HALO16(64, 2); }
    } else if (wide) { if (tr == 0) MSEG_HALO(128, 0); else if (tr == 1) MSEG_HALO(128, 1); else MSEG_HALO(128, 2); }
    else      { if (tr == 0) MSEG_HALO(64, 0); else if (tr == 1) MSEG_HALO(64, 1); else MSEG_HALO(64, 2); }
#undef MSEG_HALO
#undef MSEG_HALO16
#undef MSEG_HALO16W4
    MSEG_LAUNCH_CHECK();
    if (ks > 1) {
      const size_t total = (size_t)M * (size_t)(p.Ngemm >> 2);
      size_t blocks = (total + 255) / 256;
      if (blocks > 16384) blocks = 16384;
      MSEG_KL_AUX(igemm_splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)p.ws, ks,
                         (size_t)M, p.Ngemm, p.bias, p.dst0, p.ld0, p.acc0, p.dst1, p.ld1, p.acc1,
                         p.split < p.Ngemm ? p.split : p.Ngemm, p.dst_dtype == MSEG_ST_BF16 ? 1 : 0);
      MSEG_LAUNCH_CHECK();
    }
    return MSEG_OK;
  }
  if (p.precision == MSEG_PREC_BF16 && s16 && g_c64p_on && p.epi == MSEG_EPI_SCATTER2X2 && p.KH == 1 && p.KW == 1 &&
      p.stride == 1 && p.pad == 0 && p.mode == MSEG_MODE_CONV && p.nsrc == 1 && p.Cin == 128 && p.Kpad == 128 &&
      p.Cq == 64 && p.Ngemm == 256 && !p.acc0 && (p.Wi % 32) == 0 && (((long long)p.Hi * p.Wi) % 128) == 0 &&
      M / 128 >= 4 * (long long)c64p_workgroups()) {
    // ConvTranspose2d(128 -> 64, 2, stride 2) on bf16 tensors (level 0): persistent workgroups with resident weights
    const int tr = plain ? 0 : (generic ? 2 : 1);
    const dim3 pgrid((unsigned)c64p_workgroups());
    const int nt = (int)(M / 128);
#define MSEG_CTP(TR_)                                                                                                \
  do {                                                                                                               \
    if (p.dst_dtype == MSEG_ST_BF16) MSEG_KL((igemm_ctp_bf16_kernel<TR_, true>), pgrid, dim3(512), 0, st, p, nt); \
    else MSEG_KL((igemm_ctp_bf16_kernel<TR_, false>), pgrid, dim3(512), 0, st, p, nt);                     \
  } while (0)
    if (tr == 0) MSEG_CTP(0); else if (tr == 1) MSEG_CTP(1); else MSEG_CTP(2);
#undef MSEG_CTP
    MSEG_LAUNCH_CHECK();
    return MSEG_OK;
  }
  if (p.precision == MSEG_PREC_BF16 && s16 && g_c64p_on && p.epi == MSEG_EPI_SCATTER2X2 && p.KH == 1 && p.KW == 1 &&
      p.stride == 1 && p.pad == 0 && p.mode == MSEG_MODE_CONV && p.nsrc == 1 && p.Cin == 256 && p.Kpad == 256 &&
      p.Cq == 128 && p.Ngemm == 512 && !p.acc0 && p.Wi >= 32 && (((long long)p.Hi * p.Wi) % 64) == 0 &&
      (c64p_workgroups() & 3) == 0 && M / 64 >= 4 * (long long)(c64p_workgroups() / 4)) {
    // ConvTranspose2d(256 -> 128, 2, stride 2) on bf16 tensors (level 1): weights resident per output position
    const int tr = plain ? 0 : (generic ? 2 : 1);
    const dim3 pgrid((unsigned)c64p_workgroups());
    const int nt = (int)(M / 64);
#define MSEG_CTP2(TR_)                                                                                               \
  do {                                                                                                               \
    if (p.dst_dtype == MSEG_ST_BF16) MSEG_KL((igemm_ctp2_bf16_kernel<TR_, true>), pgrid, dim3(512), 0, st, p, nt); \
    else MSEG_KL((igemm_ctp2_bf16_kernel<TR_, false>), pgrid, dim3(512), 0, st, p, nt);                    \
  } while (0)
    if (tr == 0) MSEG_CTP2(0); else if (tr == 1) MSEG_CTP2(1); else MSEG_CTP2(2);
#undef MSEG_CTP2
    MSEG_LAUNCH_CHECK();
    return MSEG_OK;
  }
  if (p.precision == MSEG_PREC_BF16 && s16 && g_c64p_on && plain && p.epi == MSEG_EPI_PLAIN && p.KH == 2 && p.KW == 2 &&
      p.stride == 2 && p.pad == 0 && p.mode == MSEG_MODE_CONV && p.morder == MSEG_MORDER_LINEAR && p.nsrc == 1 &&
      p.Cin == 64 && p.Kpad == 64 && p.Ngemm == 128 && p.split >= p.Ngemm && !p.acc0 && !p.bias && (p.Wo % 32) == 0 &&
      p.Hi == 2 * p.Ho && p.Wi == 2 * p.Wo && (M % 64) == 0 && M / 64 >= 4 * (long long)c64p_workgroups()) {
    // data gradient of the level-0 ConvTranspose2d(128 -> 64): persistent workgroups with resident weights
    const dim3 pgrid((unsigned)c64p_workgroups());
    if (p.dst_dtype == MSEG_ST_BF16) MSEG_KL((igemm_ctd_bf16_kernel<true>), pgrid, dim3(512), 0, st, p, (int)(M / 64));
    else MSEG_KL((igemm_ctd_bf16_kernel<false>), pgrid, dim3(512), 0, st, p, (int)(M / 64));
    MSEG_LAUNCH_CHECK();
    return MSEG_OK;
  }
  if (p.precision == MSEG_PREC_BF16) {                             // bf16 inputs: the halo and the gather kernel only
    if (!fast) return MSEG_EINVAL;
    const int tr = plain ? 0 : (generic ? 2 : 1);
#define MSEG_FAST16_LAUNCH(BM_, BN_, TR_, PS_)                                                          \
  do {                                                                                                  \
    if (s16) MSEG_KL((igemm_fast_bf16_kernel<BM_, BN_, TR_, PS_, true>), grid, block, 0, st, p);   \
    else MSEG_KL((igemm_fast_bf16_kernel<BM_, BN_, TR_, PS_, false>), grid, block, 0, st, p);      \
  } while (0)
#define MSEG_FAST16_TILE(BM_, BN_)                                                        \
  do {                                                                                    \
    if (tr == 0) MSEG_FAST16_LAUNCH(BM_, BN_, 0, false);                                  \
    else if (tr == 1) { if (per_sample) MSEG_FAST16_LAUNCH(BM_, BN_, 1, true); else MSEG_FAST16_LAUNCH(BM_, BN_, 1, false); } \
    else { if (per_sample) MSEG_FAST16_LAUNCH(BM_, BN_, 2, true); else MSEG_FAST16_LAUNCH(BM_, BN_, 2, false); }              \
  } while (0)
    if (wide) MSEG_FAST16_TILE(128, 128); else MSEG_FAST16_TILE(128, 64);
#undef MSEG_FAST16_TILE
#undef MSEG_FAST16_LAUNCH
    MSEG_LAUNCH_CHECK();
    return MSEG_OK;
  }
  if (fast) {
    const int tr = plain ? 0 : (generic ? 2 : 1);
    // K-steps of a tile: 32-channel chunks x live taps (a stride-2 transposed convolution in parity order has at most
    // ceil(KH / 2) x ceil(KW / 2) live taps per tile)
    const int live_taps = (p.mode == MSEG_MODE_TCONV && p.stride == 2) ? ((p.KH + 1) / 2) * ((p.KW + 1) / 2) : p.KH * p.KW;
    const bool short_k = ((p.Cin + KC - 1) / KC) * live_taps <= 8;
#define MSEG_FAST_LAUNCH(BM_, BN_, TR_, PS_)                                                              \
  do {                                                                                                    \
    if (short_k) MSEG_KL((igemm_fast_kernel<BM_, BN_, TR_, PS_, true>), grid, block, 0, st, p); \
    else MSEG_KL((igemm_fast_kernel<BM_, BN_, TR_, PS_, false>), grid, block, 0, st, p);        \
  } while (0)
#define MSEG_FAST_TILE(BM_, BN_)                                                          \
  do {                                                                                    \
    if (tr == 0) MSEG_FAST_LAUNCH(BM_, BN_, 0, false);                                    \
    else if (tr == 1) { if (per_sample) MSEG_FAST_LAUNCH(BM_, BN_, 1, true); else MSEG_FAST_LAUNCH(BM_, BN_, 1, false); } \
    else { if (per_sample) MSEG_FAST_LAUNCH(BM_, BN_, 2, true); else MSEG_FAST_LAUNCH(BM_, BN_, 2, false); }              \
  } while (0)
    if (wide) MSEG_FAST_TILE(128, 128); else MSEG_FAST_TILE(128, 64);
#undef MSEG_FAST_TILE
#undef MSEG_FAST_LAUNCH
    MSEG_LAUNCH_CHECK();
    return MSEG_OK;
  }
#define MSEG_IGEMM_LAUNCH(BM_, BN_, PS_, GA_) \
  MSEG_KL((igemm_kernel<BM_, BN_, PS_, GA_>), grid, block, 0, st, p)
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

extern "C" int mseg_igemm(const MsegIgemm* pp, void* stream) {
  mseg_dispatch_begin(0);
  const int rc = igemm_dispatch(pp, stream);
  mseg_dispatch_end(nullptr);
  return rc;
}

// the same dispatch with the launches switched off (common.h: MSEG_KL)
extern "C" int mseg_igemm_query(const MsegIgemm* pp, MsegKernelInfo* info) {
  if (!info) return MSEG_EINVAL;
  mseg_dispatch_begin(1);
  const int rc = igemm_dispatch(pp, nullptr);
  mseg_dispatch_end(info);
  return rc;
}

// ---- fp32 -> bf16 (round to nearest even) ------------------------------------------------------------------------------
__global__ void f32_to_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = (__bf16)src[i];
}

extern "C" int mseg_f32_to_bf16(const float* src, uint16_t* dst, size_t n, void* stream) {
  if (!src || !dst || n == 0) return MSEG_EINVAL;
  size_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, (__bf16*)dst, n);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

// ---- weight repack ------------------------------------------------------------------------------------------
__global__ void pack_weight_kernel(const float* __restrict__ src, float* __restrict__ dst, int T, int R, int Rpad,
                                   int C, int Cpad, int st, int sr, int sc) {
  const size_t total = (size_t)T * Rpad * Cpad;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cpad);
    const size_t tr = i / Cpad;
    const int r = (int)(tr % Rpad);
    const int t = (int)(tr / Rpad);
    dst[i] = (c < C && r < R) ? src[(size_t)t * st + (size_t)r * sr + (size_t)c * sc] : 0.f;
  }
}

extern "C" int mseg_pack_weight(const float* src, float* dst, int T, int R, int Rpad, int C, int Cpad, int st, int sr,
                                int sc, void* stream) {
  if (!src || !dst || T <= 0 || R <= 0 || Rpad < R || C <= 0 || Cpad < C || (Cpad & 3)) return MSEG_EINVAL;
  const size_t total = (size_t)T * Rpad * Cpad;
  unsigned blocks = (unsigned)((total + 255) / 256);
  if (blocks > 4096u) blocks = 4096u;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, dst, T, R, Rpad, C,
                     Cpad, st, sr, sc);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

// ---- all weight repacks of a network in ONE launch (called by the optimizer after it has updated the weights) ----------
// jobs[] lives in device memory (uploaded once: weights are views of the optimizer's flat arena and the packed operands are
// persistent buffers, so the pointers never change); block b serves job j with first_block[j] <= b < first_block[j+1].
// A block owns a 32 x 32 (r, c) tile of ALL T taps.  In torch's layouts the taps are innermost (st = 1) and either c or r
// comes next (stride T), so for a fixed outer index the tile's source is ONE contiguous run of 32 T floats: the block reads
// 32 such runs coalesced into LDS and writes [t][r][32 consecutive c] rows of 128 B — an element-wise gather would fetch
// every source line T times from HBM (measured: 3.1 GB for 0.37 GB of weights).  dst (fp32 operand) and dst16 (bf16
// operand of the MSEG_PREC_BF16 kernels) are both optional.
#define PACK_TILE 32
__global__ __launch_bounds__(256) void pack_weights_multi_kernel(const MsegPackJob* __restrict__ jobs, int njobs) {
  __shared__ float tile[PACK_TILE * (PACK_TILE * 9 + 1)];      // T <= 9 (3x3 convs, 2x2 transposed convs)
  int lo = 0, hi = njobs - 1;
  const unsigned b = blockIdx.x;
  while (lo < hi) {                                  // last job whose first_block <= b
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].first_block <= b) lo = mid; else hi = mid - 1;
  }
  const MsegPackJob j = jobs[lo];
  const int T = j.T;
  const int ctiles = (j.Cpad + PACK_TILE - 1) / PACK_TILE;
  const int tb = (int)(b - j.first_block);
  const int r0 = (tb / ctiles) * PACK_TILE, c0 = (tb % ctiles) * PACK_TILE;
  const bool inner_c = j.sc <= j.sr;                 // which of c / r is contiguous (after the taps) in the source
  const int run = PACK_TILE * T, pitch = run + 1;    // odd pitch: the transposed reads below are bank-conflict free
  const int o0 = inner_c ? r0 : c0, i0 = inner_c ? c0 : r0;
  const int on = inner_c ? j.R : j.C, in = inner_c ? j.C : j.R;
  const long long so = inner_c ? j.sr : j.sc, si = inner_c ? j.sc : j.sr;
  const bool contiguous = j.st == 1 && si == T;
  for (int e = threadIdx.x; e < PACK_TILE * run; e += 256) {
    const int o = e / run, k = e - o * run;          // outer index in the tile, position in its run (= i * T + t)
    const int i = k / T, t = k - i * T;
    float v = 0.f;
    if (o0 + o < on && i0 + i < in)
      v = contiguous ? j.src[(long long)(o0 + o) * so + (long long)i0 * T + k]
                     : j.src[(long long)t * j.st + (long long)(o0 + o) * so + (long long)(i0 + i) * si];
    tile[o * pitch + k] = v;
  }
  __syncthreads();
  const int c = threadIdx.x & 31;
  for (int q = threadIdx.x >> 5; q < PACK_TILE * T; q += 8) {
    const int r = q % PACK_TILE, t = q / PACK_TILE;
    if (r0 + r >= j.Rpad || c0 + c >= j.Cpad) continue;
    const float v = inner_c ? tile[r * pitch + c * T + t] : tile[c * pitch + r * T + t];
    const size_t d = ((size_t)t * j.Rpad + r0 + r) * j.Cpad + c0 + c;
    if (j.dst) j.dst[d] = v;
    if (j.dst16) ((__bf16*)j.dst16)[d] = (__bf16)v;
  }
}

extern "C" unsigned mseg_pack_job_blocks(int T, int Rpad, int Cpad) {
  if (T <= 0 || T > 9 || Rpad <= 0 || Cpad <= 0) return 0;
  return (unsigned)(((Rpad + PACK_TILE - 1) / PACK_TILE) * ((Cpad + PACK_TILE - 1) / PACK_TILE));
}

extern "C" int mseg_pack_weights_multi(const MsegPackJob* jobs_dev, int njobs, unsigned total_blocks, void* stream) {
  if (!jobs_dev || njobs <= 0 || total_blocks == 0) return MSEG_EINVAL;
  hipLaunchKernelGGL(pack_weights_multi_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, njobs);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}
