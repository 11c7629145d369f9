// norm.hip — BatchNorm2d / GroupNorm(8) / InstanceNorm2d applied AFTER the activation, as in the reference blocks
// conv -> act -> norm (src/utils/unets.py:111-134,189-212) and convT -> norm (unets.py:244-262).
//
// HBM-bound kernels.  The forward never writes a normalised tensor: it reduces a = act(z) to per-(sample,channel)
// fp64 sums (one read of z), and a tiny finalize turns them into scale/shift tables that the *consumer* kernels
// apply while loading (MsegSrc, "norm-on-load").  The backward is one reduction pass + one elementwise pass.
// Layout NHWC: a thread owns 4 consecutive channels and walks pixels, so reads are 16 B/lane coalesced and the
// channel reduction needs no shuffles across lanes; partial sums meet in LDS (fp64) and then in a fixed-order
// second stage (deterministic, no atomics).
#include "common.h"
#include <stdlib.h>

struct NormGeom {
  int N, HW, C, chunks, rows_per_chunk;
  int cw, slices;       // channel groups (of V channels) per workgroup, channel slices of a tensor row
};

#define NORM_CTR_BYTES 65536          // head of the workspace: arrival counters of the in-kernel tails (see norm_tail_*)
#define NORM_CTR_MAX (NORM_CTR_BYTES / 4)

static NormGeom norm_geom(int N, int HW, int C, int vec = 4) {
  NormGeom g;
  g.N = N; g.HW = HW; g.C = C;
  // A workgroup = 256 threads = rpi row slots x cw channel groups (a thread owns V channels of its rows), one per
  // (chunk of rows, sample, slice of <= 16 channel groups).  Slicing the channels keeps >= 16 row slots per workgroup at
  // any channel count and gives the reductions over chunks / samples of a slice to the LAST workgroup of that slice (the
  // in-kernel tails below) in small pieces.  Aim at 1024 workgroups (four per CU: all resident at once, each a long
  // streaming loop; measured round 3 against 2048 and 4096: the epilogue and the partial sums weigh less, -0.6 % per bf16
  // step) with at least 8 rows per row slot.
  const int C4 = C / vec;
  g.cw = C4 < 16 ? (C4 > 0 ? C4 : 1) : 16;
  g.slices = (C4 + g.cw - 1) / g.cw;
  const int rpi = 256 / g.cw;
  const int min_rows = 8 * rpi;
  const int target = 1024;
  int want = (target + N * g.slices - 1) / (N * g.slices > 0 ? N * g.slices : 1);
  int chunks = HW / min_rows;
  if (chunks > want) chunks = want;
  if (chunks < 1) chunks = 1;
  g.rows_per_chunk = (HW + chunks - 1) / chunks;
  g.chunks = (HW + g.rows_per_chunk - 1) / g.rows_per_chunk;
  return g;
}

// threads own 4 channels of an fp32 tensor, 8 of a bf16 tensor (16-byte accesses either way; C % 8 == 0 required for bf16)
static inline int norm_vec(int st) { return st == MSEG_ST_BF16 ? 8 : 4; }

extern "C" size_t mseg_norm_workspace_bytes(int N, int HW, int C) {
  if (N <= 0 || HW <= 0 || C <= 0) return 0;
  NormGeom g4 = norm_geom(N, HW, C, 4), g8 = norm_geom(N, HW, C, 8);
  const int chunks = g4.chunks > g8.chunks ? g4.chunks : g8.chunks;
  // arrival counters + fp64 partials [N][chunks][3][C] + per-(n,c) sums [3][N][C] + fp32 k-tables [3][N][C] (2*N*C doubles)
  // (+ [2][16][C] slice sums of mseg_norm_stats_from_conv)
  return NORM_CTR_BYTES + ((size_t)N * chunks * 3 * C + (size_t)3 * N * C + (size_t)2 * N * C + (size_t)32 * C) * sizeof(double);
}

// V consecutive channels (V = 4: fp32 storage, V = 8: bf16 storage) at element offset e.  Loaded values stay in their 16 raw
// bytes (4 registers) until `get` widens them right before use: a thread keeps U x 3 such loads in flight, and widening
// them all at load time would double the registers of the bf16 form (2 waves per SIMD instead of 4).
template <bool S16>
struct NormVec {
  static constexpr int V = S16 ? 8 : 4;
  uint4 raw;
  __device__ __forceinline__ void load(const void* base, size_t e) {
    if (S16) raw = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(base) + e);
    else raw = *reinterpret_cast<const uint4*>(reinterpret_cast<const float*>(base) + e);
  }
  __device__ __forceinline__ void get(float (&v)[V]) const {
    if (S16) {
      v[0] = bf16_lo(raw.x); v[1] = bf16_hi(raw.x); v[2] = bf16_lo(raw.y); v[3] = bf16_hi(raw.y);
      v[V - 4] = bf16_lo(raw.z); v[V - 3] = bf16_hi(raw.z); v[V - 2] = bf16_lo(raw.w); v[V - 1] = bf16_hi(raw.w);
    } else {
      v[0] = __uint_as_float(raw.x); v[1] = __uint_as_float(raw.y); v[2] = __uint_as_float(raw.z); v[3] = __uint_as_float(raw.w);
    }
  }
  // v as a store + load of the tensor would return it
  static __device__ __forceinline__ void round(float (&v)[V]) {
    if (S16) {
#pragma unroll
      for (int j = 0; j < V; j += 2) {
        const uint32_t r = pack_bf16x2(v[j], v[j + 1]);
        v[j] = bf16_lo(r); v[j + 1] = bf16_hi(r);
      }
    }
  }
  // stores v (rounding to bf16 for S16) and returns the values AS STORED in v
  static __device__ __forceinline__ void put(void* base, size_t e, float (&v)[V]) {
    if (S16) {
      uint4 r;
      r.x = pack_bf16x2(v[0], v[1]); r.y = pack_bf16x2(v[2], v[3]);
      r.z = pack_bf16x2(v[V - 4], v[V - 3]); r.w = pack_bf16x2(v[V - 2], v[V - 1]);
      *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(base) + e) = r;
      v[0] = bf16_lo(r.x); v[1] = bf16_hi(r.x); v[2] = bf16_lo(r.y); v[3] = bf16_hi(r.y);
      v[V - 4] = bf16_lo(r.z); v[V - 3] = bf16_hi(r.z); v[V - 2] = bf16_lo(r.w); v[V - 1] = bf16_hi(r.w);
    } else {
      *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + e) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
};

// ---- in-kernel tails: what used to be three more launches per pass ---------------------------------------------------------
// A pass leaves fp64 partial sums per (sample, chunk of rows, channel).  Reducing them over the chunks, then over the
// samples (BatchNorm, bias gradients) or over the channels of a group (Group / InstanceNorm), and turning the sums into
// the tables the next kernels read took `norm_reduce_chunks` + a finalize kernel (+ `norm_colsum`) per pass: six ~5 us
// launches per layer and step, each with its dispatch gap.  Now the LAST workgroup to arrive does it:
//   level 1: the workgroups of one (sample, channel slice) count their arrivals; the last one sums that slice's chunks in
//            a fixed order and publishes nc[s][n][c];
//   level 2: those level-1 finishers count arrivals per channel slice (BatchNorm / column sums: N arrivals) or per sample
//            (per-sample norms: `slices` arrivals); the last one finishes its slice / sample and writes the tables.
// Every summation order is fixed by the indices, never by arrival order: results are bit-reproducible.
// Hand-off (cdna_hip_programming.md, Guideline 16): payload by write-through (sc1) stores, every storing wave drains its
// stores, workgroup barrier, ONE lane's relaxed agent-scope add; the finisher runs ONE agent-scope acquire and then reads
// the payload with plain loads (several in flight).  No release fence: a pass that has just written a tensor must not flush its L2.
// Counters: NORM_CTR_BYTES at the head of the workspace, zero before the first call, left zero by every call.
typedef __attribute__((address_space(1))) unsigned long long norm_gu64;
typedef __attribute__((address_space(1))) unsigned norm_gu32;
__device__ __forceinline__ void norm_st_sc1(double* p, double v) {
  __hip_atomic_store((norm_gu64*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double norm_ld_sc1(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load((norm_gu64*)const_cast<double*>(p), __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT));
}
// true (in every thread) in the workgroup whose arrival completes `expected`; that workgroup leaves the counter zero
__device__ __forceinline__ bool norm_arrive(unsigned* ctr, unsigned expected, int tid, int* s_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave: its write-through stores have left
  __syncthreads();
  if (tid == 0) {
    const unsigned ticket = __hip_atomic_fetch_add((norm_gu32*)ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = ticket == expected - 1u;
    if (last) {
      __hip_atomic_store((norm_gu32*)ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    *s_flag = last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  return *s_flag != 0;
}

#define NORM_TAIL_NONE 0
#define NORM_TAIL_BN_FWD 1     // statistics -> scale / shift / mean / rstd / running statistics
#define NORM_TAIL_PS_FWD 2     // the same per sample (GroupNorm, InstanceNorm)
#define NORM_TAIL_BN_BWD 3     // sums of (gy, gy a) -> k1, k2, k3, dgamma, dbeta
#define NORM_TAIL_COLSUM 4     // sum of dz over samples -> the producing convolution's bias gradient
struct NormTail {
  int kind, norm;
  unsigned* ctr;               // NORM_CTR_MAX counters
  double* nc;                  // [3][N][C] per-(sample, channel) sums
  const float* gamma; const float* beta;
  float eps, momentum;
  float *scale, *shift, *mean_out, *rstd_out, *running_mean, *running_var;   // forward
  const float *mean, *rstd;                                                   // backward
  float *k1, *k2, *k3, *dgamma, *dbeta, *colsum;
};

// MODE 0: forward  sums of (a, a*a)           from z        (+ optional store of a = act(z) into `aio`)
// MODE 1: backward sums of (gy, gy*a)         from (gy, z)  (a read from `aio` when given)
// MODE 2: backward apply: dz = (k1*gy + k2*a + k3) * act'(z), sums of (dz)  (a recomputed from z, never read)
// MODE 3: a = act(z) only (eval-mode BatchNorm with an expensive activation: materialise it once for the consumers)
// `aio`: for expensive activations (Mish / ELU / LeakyReLU path) the activated tensor is materialised once so that the
// conv / wgrad K-loops do not re-evaluate transcendentals for each of the 9 taps and every output tile.
// S16: z, gy, dz and aio are bf16 tensors (the sums are taken over the values as stored, i.e. of the rounded dz).
// AIO: the launch carries `aio` (compile-time, so that the ReLU / no-activation passes keep their exact code)
// grid: (chunks, N, slices)
template <int MODE, bool S16, bool AIO>
__global__ __launch_bounds__(256) void norm_pass_kernel(const void* __restrict__ z, const void* __restrict__ gy,
                                                        void* __restrict__ dz, const float* __restrict__ k1,
                                                        const float* __restrict__ k2, const float* __restrict__ k3,
                                                        int kss, NormGeom g, int act, double* __restrict__ part,
                                                        void* __restrict__ aio, const NormTail t) {
  constexpr int V = S16 ? 8 : 4;
  __shared__ double red[256 * 2 * V];
  __shared__ int s_flag;
  const int tid = threadIdx.x;
  const int chunk = blockIdx.x, n = blockIdx.y, slice = blockIdx.z;
  const int C4 = g.C / V;
  const int CW = g.cw;
  const int rpi = 256 / CW;
  const int cq = tid % CW, r0 = tid / CW;
  const int row_begin = chunk * g.rows_per_chunk;
  int row_end = row_begin + g.rows_per_chunk;
  if (row_end > g.HW) row_end = g.HW;
  constexpr int NS = (MODE == 2) ? 1 : 2;
  double* pout = part + ((size_t)n * g.chunks + chunk) * 3 * g.C;
  const int c4 = slice * CW + cq;
  const bool active = (c4 < C4) && (r0 < rpi);
  double s0[V], s1[V];
#pragma unroll
  for (int j = 0; j < V; ++j) { s0[j] = 0.0; s1[j] = 0.0; }
  if (active) {
    const int c = c4 * V;
    float a1[V], a2[V], a3[V];
    if (MODE == 2) {
#pragma unroll
      for (int j = 0; j < V; ++j) {
        a1[j] = k1[(size_t)n * kss + c + j]; a2[j] = k2[(size_t)n * kss + c + j]; a3[j] = k3[(size_t)n * kss + c + j];
      }
    }
    // U rows per trip, all loads issued before any use: a thread keeps U (x 2-3 operands) 16-byte loads in flight —
    // one load pair per trip left the pass latency-bound at ~3 TB/s
    constexpr int U = (MODE == 0 && !AIO) ? 8 : !S16 ? 4 : MODE == 1 ? (AIO ? 2 : 4) : MODE == 2 ? 2 : 4;   // measured per pass (registers vs loads in flight)
    for (int r = row_begin + r0; r < row_end; r += rpi * U) {
      NormVec<S16> zv[U], gv[U], av[U];
      bool ok[U];
      // the sums of one trip (<= U values per channel) are taken in fp32 and join the fp64 running sums once per trip: the
      // conversions and fp64 additions per ELEMENT were a third of the pass's vector instructions
      float p0[V], p1[V];
#pragma unroll
      for (int j = 0; j < V; ++j) { p0[j] = 0.f; p1[j] = 0.f; }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int ru = r + u * rpi;
        ok[u] = ru < row_end;
        const size_t off = ((size_t)n * g.HW + (ok[u] ? ru : r)) * g.C + c;
        if (MODE != 1 || !AIO) zv[u].load(z, off);            // the sums of MODE 1 need a only: z stays in HBM when a is stored
        if (MODE == 1 && AIO) av[u].load(aio, off);           // MODE 2 recomputes a next to act'(z) instead (act_pair)
        if (MODE == 1 || MODE == 2) gv[u].load(gy, off);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (!ok[u]) continue;
        const size_t off = ((size_t)n * g.HW + r + u * rpi) * g.C + c;
        float zf[V], a4[V];
        if (MODE != 1 || !AIO) zv[u].get(zf);
        if (MODE == 2 && AIO) {                 // expensive activation: a (as the forward stored it) and act'(z) in one go
          float gf[V], d[V];
          gv[u].get(gf);
#pragma unroll
          for (int j = 0; j < V; ++j) act_pair<S16>(zf[j], act, a4[j], d[j]);
          NormVec<S16>::round(a4);
#pragma unroll
          for (int j = 0; j < V; ++j) d[j] = (a1[j] * gf[j] + a2[j] * a4[j] + a3[j]) * d[j];
          NormVec<S16>::put(dz, off, d);
#pragma unroll
          for (int j = 0; j < V; ++j) p0[j] += d[j];
          continue;
        }
        if (MODE == 1 && AIO) av[u].get(a4);
        else {
          if (S16 && act != MSEG_ACT_NONE && act != MSEG_ACT_RELU) {
#pragma unroll
            for (int j = 0; j < V; ++j) a4[j] = act_fwd_fast(zf[j], act);
          } else {
#pragma unroll
            for (int j = 0; j < V; j += 4) {       // act_fwd4: the cheap activations take a wave-uniform fast path
              const float4 t4 = act_fwd4(make_float4(zf[j], zf[j + 1], zf[j + 2], zf[j + 3]), act);
              a4[j] = t4.x; a4[j + 1] = t4.y; a4[j + 2] = t4.z; a4[j + 3] = t4.w;
            }
          }
        }
        if ((MODE == 0 || MODE == 3) && AIO)
          NormVec<S16>::put(aio, off, a4);      // the consumers read the ROUNDED activation: the statistics describe that
        if (MODE == 3) continue;
        if (MODE == 0) {
#pragma unroll
          for (int j = 0; j < V; ++j) { p0[j] += a4[j]; p1[j] = fmaf(a4[j], a4[j], p1[j]); }
        } else {
          float gf[V];
          gv[u].get(gf);
          if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < V; ++j) { p0[j] += gf[j]; p1[j] = fmaf(gf[j], a4[j], p1[j]); }
          } else {
            float d[V];
#pragma unroll
            for (int j = 0; j < V; ++j)
              d[j] = (a1[j] * gf[j] + a2[j] * a4[j] + a3[j]) * (S16 ? act_bwd_fast(zf[j], act) : act_bwd(zf[j], act));
            NormVec<S16>::put(dz, off, d);      // d now holds dz as stored
#pragma unroll
            for (int j = 0; j < V; ++j) p0[j] += d[j];
          }
        }
      }
      if (MODE != 3) {
#pragma unroll
        for (int j = 0; j < V; ++j) { s0[j] += (double)p0[j]; if (NS == 2) s1[j] += (double)p1[j]; }
      }
    }
  }
  if (MODE == 3) return;
  double t0[V], t1[V];
  const bool fold = (CW & (CW - 1)) == 0;               // channel groups per row slot a power of two (<= 16): the common case
  if (fold) {
    // The row slots of a wave that hold the same channels are the lanes l, l ^ CW, l ^ 2 CW, ...: an xor butterfly leaves
    // their sum in every lane (fixed tree), the four waves meet in LDS — 64 values per finishing thread instead of the
    // 2 x 16 x 32 serial LDS reads below, which were ~6 us of every pass.
    for (int m = CW; m < 64; m <<= 1) {
#pragma unroll
      for (int j = 0; j < V; ++j) {
        s0[j] += __shfl_xor(s0[j], m, 64);
        if (NS == 2) s1[j] += __shfl_xor(s1[j], m, 64);
      }
    }
    const int lane = tid & 63, wv = tid >> 6;
    if (lane < CW) {
#pragma unroll
      for (int j = 0; j < V; ++j) {
        red[(wv * CW + lane) * 2 * V + j] = s0[j];
        red[(wv * CW + lane) * 2 * V + V + j] = s1[j];
      }
    }
    __syncthreads();
    if (tid < CW) {
#pragma unroll
      for (int j = 0; j < V; ++j) { t0[j] = red[tid * 2 * V + j]; t1[j] = red[tid * 2 * V + V + j]; }
#pragma unroll
      for (int w = 1; w < 4; ++w)
#pragma unroll
        for (int j = 0; j < V; ++j) {
          t0[j] += red[(w * CW + tid) * 2 * V + j];
          t1[j] += red[(w * CW + tid) * 2 * V + V + j];
        }
    }
  } else {
#pragma unroll
    for (int j = 0; j < V; ++j) {
      red[tid * 2 * V + j] = s0[j];
      red[tid * 2 * V + V + j] = s1[j];
    }
    __syncthreads();
    if (r0 == 0 && c4 < C4) {
#pragma unroll
      for (int j = 0; j < V; ++j) { t0[j] = 0.0; t1[j] = 0.0; }
      for (int k = 0; k < rpi; ++k) {
#pragma unroll
        for (int j = 0; j < V; ++j) {
          t0[j] += red[(k * CW + cq) * 2 * V + j];
          t1[j] += red[(k * CW + cq) * 2 * V + V + j];
        }
      }
    }
  }
  if (r0 == 0 && c4 < C4) {
    if (t.kind == NORM_TAIL_NONE) {
#pragma unroll
      for (int j = 0; j < V; ++j) {
        pout[c4 * V + j] = t0[j];
        if (NS == 2) pout[g.C + c4 * V + j] = t1[j];
      }
    } else {                                           // handed to another workgroup of this launch: write-through
#pragma unroll
      for (int j = 0; j < V; ++j) {
        norm_st_sc1(pout + c4 * V + j, t0[j]);
        if (NS == 2) norm_st_sc1(pout + g.C + c4 * V + j, t1[j]);
      }
    }
  }
  if (t.kind == NORM_TAIL_NONE) return;

  // ---- level 1: the last workgroup of (sample n, this slice) sums the slice's chunks ---------------------------------------
  if (!norm_arrive(t.ctr + n * g.slices + slice, (unsigned)g.chunks, tid, &s_flag)) return;
  const int SC = CW * V;                               // channels of a slice (<= 128)
  const int c_lo = slice * SC;
  const int nout = NS * SC;                            // outputs (s, channel) of the slice: <= 256
  int KG = 256 / nout;                                 // chunk groups summed side by side, then in order
  {
    const int o = tid % nout, kg = tid / nout;
    const int sidx = o / SC, c = c_lo + o % SC;
    double acc = 0.0;
    if (kg < KG && c < g.C) {
      // (plain loads behind the finisher's acquire; unrolled so that eight are in flight — the additions keep their order)
      const double* src = part + ((size_t)n * g.chunks * 3 + sidx) * g.C + c;
      const size_t kstride = (size_t)3 * g.C;
#pragma unroll 8
      for (int k = kg; k < g.chunks; k += KG) acc += src[k * kstride];
    }
    red[tid] = acc;
    __syncthreads();
    if (kg == 0) {
      double tot = 0.0;
      for (int j = 0; j < KG; ++j) tot += red[j * nout + o];
      red[256 + o] = tot;                              // sums[s][channel of the slice]
      if (c < g.C) norm_st_sc1(t.nc + ((size_t)sidx * g.N + n) * g.C + c, tot);
    }
  }
  // ---- level 2 ---------------------------------------------------------------------------------------------------------------
  unsigned* const ctr2 = t.ctr + g.N * g.slices;
  if (t.kind == NORM_TAIL_PS_FWD) {
    if (!norm_arrive(ctr2 + n, (unsigned)g.slices, tid, &s_flag)) return;
    // all slices of sample n are in nc: statistics of its groups (fixed order over the group's channels), then the tables
    const int groups = (t.norm == MSEG_NORM_GN) ? 8 : g.C;
    const int cg = g.C / groups;
    const double* S = t.nc + (size_t)n * g.C;
    const double* Q = t.nc + ((size_t)g.N + n) * g.C;
    for (int base = 0; base < groups; base += 128) {   // 128 groups at a time through LDS: mean | rstd
      const int grp = base + tid;
      if (tid < 128 && grp < groups) {
        double sm = 0.0, q = 0.0;
        for (int k = 0; k < cg; ++k) { sm += S[grp * cg + k]; q += Q[grp * cg + k]; }
        const double cnt = (double)cg * g.HW;
        const double mean = sm / cnt;
        double var = q / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        const double rstd = 1.0 / sqrt(var + (double)t.eps);
        red[tid] = mean; red[128 + tid] = rstd;
        t.mean_out[n * groups + grp] = (float)mean;
        t.rstd_out[n * groups + grp] = (float)rstd;
      }
      __syncthreads();
      const int c_end = (base + 128 < groups ? base + 128 : groups) * cg;
      for (int c = base * cg + tid; c < c_end; c += 256) {
        const int gl = c / cg - base;
        const double mean = red[gl], rstd = red[128 + gl];
        const double ga = (t.norm == MSEG_NORM_GN && t.gamma) ? (double)t.gamma[c] : 1.0;
        const double be = (t.norm == MSEG_NORM_GN && t.beta) ? (double)t.beta[c] : 0.0;
        t.scale[(size_t)n * g.C + c] = (float)(ga * rstd);
        t.shift[(size_t)n * g.C + c] = (float)(be - mean * ga * rstd);
      }
      __syncthreads();
    }
    return;
  }
  if (!norm_arrive(ctr2 + slice, (unsigned)g.N, tid, &s_flag)) return;
  // every sample of this slice is in nc: sums over the samples in index order, then the per-channel results
  {
    const int o = tid % nout;
    const int sidx = o / SC, c = c_lo + o % SC;
    if (tid < nout) {
      double sm = 0.0;
      if (c < g.C) {
        const double* src = t.nc + (size_t)sidx * g.N * g.C + c;
#pragma unroll 8
        for (int i = 0; i < g.N; ++i) sm += src[(size_t)i * g.C];
      }
      red[o] = sm;
    }
    __syncthreads();
    const int c2 = c_lo + tid;
    if (tid < SC && c2 < g.C) {
      const double a0 = red[tid], b0 = NS == 2 ? red[SC + tid] : 0.0;
      const double cnt = (double)g.N * g.HW;
      if (t.kind == NORM_TAIL_COLSUM) {
        t.colsum[c2] = (float)a0;
      } else if (t.kind == NORM_TAIL_BN_FWD) {
        const double mean = a0 / cnt;
        double var = b0 / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        const double rstd = 1.0 / sqrt(var + (double)t.eps);
        const double ga = t.gamma ? (double)t.gamma[c2] : 1.0, be = t.beta ? (double)t.beta[c2] : 0.0;
        t.scale[c2] = (float)(ga * rstd);
        t.shift[c2] = (float)(be - mean * ga * rstd);
        t.mean_out[c2] = (float)mean;
        t.rstd_out[c2] = (float)rstd;
        if (t.running_mean) {
          const double unb = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
          t.running_mean[c2] = (float)((1.0 - (double)t.momentum) * t.running_mean[c2] + (double)t.momentum * mean);
          t.running_var[c2] = (float)((1.0 - (double)t.momentum) * t.running_var[c2] + (double)t.momentum * unb);
        }
      } else {                                         // NORM_TAIL_BN_BWD: a0 = sum gy, b0 = sum gy a
        const double mu = t.mean[c2], r = t.rstd[c2], ga = t.gamma ? (double)t.gamma[c2] : 1.0;
        const double sx = r * (b0 - mu * a0);          // sum gy * xhat
        const double m1 = ga * a0 / cnt, m2 = ga * sx / cnt;
        t.k1[c2] = (float)(ga * r);
        t.k2[c2] = (float)(-r * r * m2);
        t.k3[c2] = (float)(-r * m1 + r * r * m2 * mu);
        if (t.dgamma) t.dgamma[c2] = (float)sx;
        if (t.dbeta) t.dbeta[c2] = (float)a0;
      }
    }
  }
}

// nc[s][n][c] = sum over chunks of part[n][chunk][s][c]; 32 outputs x 8 chunk groups per workgroup, fixed order
__device__ __forceinline__ void norm_reduce_chunks_body(const double* __restrict__ part, double* __restrict__ nc,
                                                        const NormGeom& g, int ns, bool handoff, double (*red)[32]) {
  const int o = threadIdx.x & 31, kg = threadIdx.x >> 5;
  const int total = ns * g.N * g.C;
  for (int base = blockIdx.x * 32; base < total; base += gridDim.x * 32) {
    const int i = base + o;
    double acc = 0.0;
    int c = 0, n = 0, sidx = 0;
    if (i < total) {
      c = i % g.C;
      n = (i / g.C) % g.N;
      sidx = i / (g.C * g.N);
      for (int k = kg; k < g.chunks; k += 8) acc += part[(((size_t)n * g.chunks + k) * 3 + sidx) * g.C + c];
    }
    red[kg][o] = acc;
    __syncthreads();
    if (kg == 0 && i < total) {
      double t = 0.0;
#pragma unroll
      for (int j = 0; j < 8; ++j) t += red[j][o];
      double* dst = nc + ((size_t)sidx * g.N + n) * g.C + c;
      if (handoff) norm_st_sc1(dst, t);                // read by another workgroup of this launch: write-through
      else *dst = t;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void norm_reduce_chunks_kernel(const double* __restrict__ part, double* __restrict__ nc,
                                                                 NormGeom g, int ns) {
  __shared__ double red[8][32];
  norm_reduce_chunks_body(part, nc, g, ns, false, red);
}

// forward finalize: statistics -> scale/shift tables (+ saved mean/rstd, BN running stats)
// (`first`, `stride`: the calling thread's index and the thread count of the launch — or of ONE workgroup, when the last
// workgroup of the reduction launch finishes the job itself, norm_reduce_finish_kernel)
__device__ __forceinline__ void norm_fwd_finalize_body(const double* __restrict__ nc, const NormGeom& g, int norm,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float eps, float* __restrict__ scale, float* __restrict__ shift,
                                                       float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                       float* __restrict__ running_mean, float* __restrict__ running_var,
                                                       float momentum, double count, int first, int stride) {
  const double* S = nc;
  const double* Q = nc + (size_t)g.N * g.C;
  if (norm == MSEG_NORM_BN) {
    for (int c = first; c < g.C; c += stride) {
      double s = 0.0, q = 0.0;
      // (unrolled: the loads of 8 samples are in flight together; rolled, this tiny kernel was 32 dependent memory round
      // trips = 11 us, 114 such launches per step; the order of the additions is unchanged)
#pragma unroll 8
      for (int n = 0; n < g.N; ++n) { s += S[(size_t)n * g.C + c]; q += Q[(size_t)n * g.C + c]; }
      // (count > 0: the g.N "samples" are row slices of a convolution's partial sums, mseg_norm_stats_from_conv)
      const double cnt = count > 0.0 ? count : (double)g.N * g.HW;
      const double mean = s / cnt;
      double var = q / cnt - mean * mean;
      if (var < 0.0) var = 0.0;
      const double rstd = 1.0 / sqrt(var + (double)eps);
      const double ga = gamma ? (double)gamma[c] : 1.0, be = beta ? (double)beta[c] : 0.0;
      scale[c] = (float)(ga * rstd);
      shift[c] = (float)(be - mean * ga * rstd);
      mean_out[c] = (float)mean;
      rstd_out[c] = (float)rstd;
      if (running_mean) {
        const double unb = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
        running_mean[c] = (float)((1.0 - (double)momentum) * running_mean[c] + (double)momentum * mean);
        running_var[c] = (float)((1.0 - (double)momentum) * running_var[c] + (double)momentum * unb);
      }
    }
    return;
  }
  const int groups = (norm == MSEG_NORM_GN) ? 8 : g.C;
  const int cg = g.C / groups;
  const int total = g.N * g.C;
  for (int i = first; i < total; i += stride) {
    const int c = i % g.C, n = i / g.C;
    const int grp = c / cg;
    double s = 0.0, q = 0.0;
    for (int k = 0; k < cg; ++k) {
      s += S[(size_t)n * g.C + grp * cg + k];
      q += Q[(size_t)n * g.C + grp * cg + k];
    }
    const double cnt = (double)cg * g.HW;
    const double mean = s / cnt;
    double var = q / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    const double ga = (norm == MSEG_NORM_GN && gamma) ? (double)gamma[c] : 1.0;
    const double be = (norm == MSEG_NORM_GN && beta) ? (double)beta[c] : 0.0;
    scale[i] = (float)(ga * rstd);
    shift[i] = (float)(be - mean * ga * rstd);
    if (c == grp * cg) {
      mean_out[n * groups + grp] = (float)mean;
      rstd_out[n * groups + grp] = (float)rstd;
    }
  }
}

__global__ void norm_fwd_finalize_kernel(const double* __restrict__ nc, NormGeom g, int norm,
                                         const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                         float* __restrict__ scale, float* __restrict__ shift,
                                         float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                         float* __restrict__ running_mean, float* __restrict__ running_var,
                                         float momentum, double count) {
  norm_fwd_finalize_body(nc, g, norm, gamma, beta, eps, scale, shift, mean_out, rstd_out, running_mean, running_var, momentum,
                         count, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// backward finalize: per-(n,c) sums of gy and gy*a -> k1,k2,k3 tables, dgamma, dbeta
__device__ __forceinline__ void norm_bwd_finalize_body(const double* __restrict__ nc, const NormGeom& g, int norm,
                                                       const float* __restrict__ gamma, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, float* __restrict__ k1,
                                                       float* __restrict__ k2, float* __restrict__ k3,
                                                       float* __restrict__ dgamma, float* __restrict__ dbeta, int first,
                                                       int stride) {
  const double* S1 = nc;                        // sum gy
  const double* S2 = nc + (size_t)g.N * g.C;    // sum gy*a
  if (norm == MSEG_NORM_BN) {
    for (int c = first; c < g.C; c += stride) {
      double s1 = 0.0, s2 = 0.0;
#pragma unroll 8
      for (int n = 0; n < g.N; ++n) { s1 += S1[(size_t)n * g.C + c]; s2 += S2[(size_t)n * g.C + c]; }
      const double cnt = (double)g.N * g.HW;
      const double mu = mean[c], r = rstd[c], ga = gamma ? (double)gamma[c] : 1.0;
      const double sx = r * (s2 - mu * s1);  // sum gy * xhat
      const double m1 = ga * s1 / cnt, m2 = ga * sx / cnt;
      k1[c] = (float)(ga * r);
      k2[c] = (float)(-r * r * m2);
      k3[c] = (float)(-r * m1 + r * r * m2 * mu);
      if (dgamma) dgamma[c] = (float)sx;
      if (dbeta) dbeta[c] = (float)s1;
    }
    return;
  }
  const int groups = (norm == MSEG_NORM_GN) ? 8 : g.C;
  const int cg = g.C / groups;
  const int total = g.N * g.C;
  for (int i = first; i < total; i += stride) {
    const int c = i % g.C, n = i / g.C;
    const int grp = c / cg;
    const double mu = mean[n * groups + grp], r = rstd[n * groups + grp];
    double m1 = 0.0, m2 = 0.0;
    for (int k = 0; k < cg; ++k) {
      const int cc = grp * cg + k;
      const double ga = (norm == MSEG_NORM_GN && gamma) ? (double)gamma[cc] : 1.0;
      const double s1 = S1[(size_t)n * g.C + cc], s2 = S2[(size_t)n * g.C + cc];
      m1 += ga * s1;
      m2 += ga * r * (s2 - mu * s1);
    }
    const double cnt = (double)cg * g.HW;
    m1 /= cnt; m2 /= cnt;
    const double gac = (norm == MSEG_NORM_GN && gamma) ? (double)gamma[c] : 1.0;
    k1[i] = (float)(gac * r);
    k2[i] = (float)(-r * r * m2);
    k3[i] = (float)(-r * m1 + r * r * m2 * mu);
  }
  if (norm == MSEG_NORM_GN && dgamma) {
    // dgamma_c = sum_n rstd*(S2 - mean*S1), dbeta_c = sum_n S1   (one thread per channel, fixed order)
    for (int c = first; c < g.C; c += stride) {
      const int grp = c / cg;
      double dg = 0.0, db = 0.0;
      for (int n = 0; n < g.N; ++n) {
        const double mu = mean[n * groups + grp], r = rstd[n * groups + grp];
        const double s1 = S1[(size_t)n * g.C + c], s2 = S2[(size_t)n * g.C + c];
        dg += r * (s2 - mu * s1);
        db += s1;
      }
      dgamma[c] = (float)dg;
      if (dbeta) dbeta[c] = (float)db;
    }
  }
}

__global__ void norm_bwd_finalize_kernel(const double* __restrict__ nc, NormGeom g, int norm,
                                         const float* __restrict__ gamma, const float* __restrict__ mean,
                                         const float* __restrict__ rstd, float* __restrict__ k1,
                                         float* __restrict__ k2, float* __restrict__ k3, float* __restrict__ dgamma,
                                         float* __restrict__ dbeta) {
  norm_bwd_finalize_body(nc, g, norm, gamma, mean, rstd, k1, k2, k3, dgamma, dbeta, blockIdx.x * blockDim.x + threadIdx.x,
                         gridDim.x * blockDim.x);
}

// dbias[c] = sum_n nc[0][n][c]
__device__ __forceinline__ void norm_colsum_body(const double* __restrict__ nc, const NormGeom& g, float* __restrict__ out,
                                                 int first, int stride) {
  for (int c = first; c < g.C; c += stride) {
    double s = 0.0;
#pragma unroll 8
    for (int n = 0; n < g.N; ++n) s += nc[(size_t)n * g.C + c];
    out[c] = (float)s;
  }
}
__global__ void norm_colsum_kernel(const double* __restrict__ nc, NormGeom g, float* __restrict__ out) {
  norm_colsum_body(nc, g, out, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// The reduction over the chunks AND what used to be the launch after it (finalize / column sums), for per-channel results
// of <= 256 channels (BatchNorm, bias gradients: one channel per thread of ONE workgroup): the last workgroup of the
// reduction to arrive does it — same code, same order, same results as the two launches, one ~5 us launch less per pass.
// Unlike the in-kernel tails of the big pass (slower: every one of its 1024 workgroups pays for the hand-off) the hand-off
// here is between the few workgroups of a 5-us kernel.  Counter: the LAST slot of the workspace's counter block.
__global__ __launch_bounds__(256) void norm_reduce_finish_kernel(const double* __restrict__ part, double* __restrict__ nc,
                                                                 NormGeom g, int ns, const NormTail t) {
  __shared__ double red[8][32];
  __shared__ int s_flag;
  norm_reduce_chunks_body(part, nc, g, ns, true, red);
  if (!norm_arrive(t.ctr + NORM_CTR_MAX - 1, gridDim.x, threadIdx.x, &s_flag)) return;
  if (t.kind == NORM_TAIL_BN_FWD)
    norm_fwd_finalize_body(nc, g, MSEG_NORM_BN, t.gamma, t.beta, t.eps, t.scale, t.shift, t.mean_out, t.rstd_out, t.running_mean,
                           t.running_var, t.momentum, 0.0, threadIdx.x, 256);
  else if (t.kind == NORM_TAIL_BN_BWD)
    norm_bwd_finalize_body(nc, g, MSEG_NORM_BN, t.gamma, t.mean, t.rstd, t.k1, t.k2, t.k3, t.dgamma, t.dbeta, threadIdx.x, 256);
  else
    norm_colsum_body(nc, g, t.colsum, threadIdx.x, 256);
}

// BatchNorm / bias gradients (per-channel results): from a pass's partial sums to the tables in ONE launch WITHOUT a hand-off
// between workgroups.  A workgroup owns FOUR channels and does for them, in the order of the launches it replaces, what
// norm_reduce_chunks_body (per sample: eight strided partial sums over the chunks, combined in order), the finalize bodies
// (the samples summed in order) and norm_colsum_body do — same additions, same bits — with the per-sample sums handed on
// through LDS instead of through `nc` + an arrival counter + an acquire (they are still written to `nc`, as before).  The
// finishing reduction above spends ~10 us per launch on that hand-off (load, store, atomic, load again: four dependent
// memory round trips); this is one round trip of independent loads, three barriers per statistic and N additions.
// Threads: (sample lane nl = tid >> 5, chunk group kg = (tid >> 2) & 7, channel cl = tid & 3); any number of channels;
// 256 threads (8 sample lanes) for up to 8 samples, else 1024 (32 lanes: the load phase is what the launch waits for).
#define NORM_FB_CH 4
#define NORM_FB_NB 64          // samples per round through LDS
__global__ __launch_bounds__(1024) void norm_bn_finish_kernel(const double* __restrict__ part, double* __restrict__ nc,
                                                             NormGeom g, int ns, const NormTail t) {
  __shared__ double red[NORM_FB_NB][8][NORM_FB_CH];
  __shared__ double tn[NORM_FB_NB][NORM_FB_CH];
  __shared__ double tot[3][NORM_FB_CH];
  const int tid = threadIdx.x;
  const int cl = tid & 3, kg = (tid >> 2) & 7, nl = tid >> 5, nlanes = (int)blockDim.x >> 5;
  const int c = blockIdx.x * NORM_FB_CH + cl;
  const bool cok = c < g.C;
  for (int sidx = 0; sidx < ns; ++sidx) {
    double run = 0.0;                                  // threads 0..3: the sum over the samples, in order
    for (int n0 = 0; n0 < g.N; n0 += NORM_FB_NB) {
      const int nb = g.N - n0 < NORM_FB_NB ? g.N - n0 : NORM_FB_NB;
#pragma unroll 2
      for (int nn = nl; nn < nb; nn += nlanes) {
        double acc = 0.0;
        if (cok) {
          const double* src = part + (((size_t)(n0 + nn) * g.chunks) * 3 + sidx) * g.C + c;
          const size_t kstride = (size_t)3 * g.C;
#pragma unroll 4
          for (int k = kg; k < g.chunks; k += 8) acc += src[k * kstride];
        }
        red[nn][kg][cl] = acc;
      }
      __syncthreads();
      {
        const int nn = tid >> 2;                       // 64 samples x 4 channels
        if (nn < nb) {
          double s8 = 0.0;
#pragma unroll
          for (int j = 0; j < 8; ++j) s8 += red[nn][j][cl];
          tn[nn][cl] = s8;
          if (cok) nc[((size_t)sidx * g.N + n0 + nn) * g.C + c] = s8;
        }
      }
      __syncthreads();
      if (tid < NORM_FB_CH)
        for (int nn = 0; nn < nb; ++nn) run += tn[nn][tid];
      __syncthreads();
    }
    if (tid < NORM_FB_CH) tot[sidx][tid] = run;
  }
  if (tid >= NORM_FB_CH || !cok) return;               // (threads 0..3 read what they wrote themselves)
  const double a0 = tot[0][tid], b0 = ns > 1 ? tot[1][tid] : 0.0;
  if (t.kind == NORM_TAIL_BN_FWD) {
    const double cnt = (double)g.N * g.HW;
    const double mean = a0 / cnt;
    double var = b0 / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)t.eps);
    const double ga = t.gamma ? (double)t.gamma[c] : 1.0, be = t.beta ? (double)t.beta[c] : 0.0;
    t.scale[c] = (float)(ga * rstd);
    t.shift[c] = (float)(be - mean * ga * rstd);
    t.mean_out[c] = (float)mean;
    t.rstd_out[c] = (float)rstd;
    if (t.running_mean) {
      const double unb = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
      t.running_mean[c] = (float)((1.0 - (double)t.momentum) * t.running_mean[c] + (double)t.momentum * mean);
      t.running_var[c] = (float)((1.0 - (double)t.momentum) * t.running_var[c] + (double)t.momentum * unb);
    }
  } else if (t.kind == NORM_TAIL_BN_BWD) {             // a0 = sum gy, b0 = sum gy a
    const double cnt = (double)g.N * g.HW;
    const double mu = t.mean[c], r = t.rstd[c], ga = t.gamma ? (double)t.gamma[c] : 1.0;
    const double sx = r * (b0 - mu * a0);              // sum gy * xhat
    const double m1 = ga * a0 / cnt, m2 = ga * sx / cnt;
    t.k1[c] = (float)(ga * r);
    t.k2[c] = (float)(-r * r * m2);
    t.k3[c] = (float)(-r * m1 + r * r * m2 * mu);
    if (t.dgamma) t.dgamma[c] = (float)sx;
    if (t.dbeta) t.dbeta[c] = (float)a0;
  } else {
    t.colsum[c] = (float)a0;
  }
}

__global__ void bn_eval_coeffs_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rm, const float* __restrict__ rv, float eps, int C,
                                      float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  // torch eval-mode batch_norm: (x - running_mean) / sqrt(running_var + eps) * weight + bias
  const float inv = 1.0f / sqrtf(rv[c] + eps);
  const float sc = (gamma ? gamma[c] : 1.f) * inv;
  scale[c] = sc;
  shift[c] = (beta ? beta[c] : 0.f) - rm[c] * sc;
}

static inline unsigned nblocks(size_t n, unsigned cap = 1024u) {
  size_t b = (n + 255) / 256;
  if (b < 1) b = 1;
  return (unsigned)(b > cap ? cap : b);
}

#define NORM_PASS_A(MODE_, S16_, ...)                                                                                \
  do {                                                                                                               \
    if (aio_) hipLaunchKernelGGL((norm_pass_kernel<MODE_, S16_, true>), dim3(g.chunks, N, g.slices), dim3(256), 0, st, __VA_ARGS__); \
    else hipLaunchKernelGGL((norm_pass_kernel<MODE_, S16_, false>), dim3(g.chunks, N, g.slices), dim3(256), 0, st, __VA_ARGS__);     \
  } while (0)
// AIO_ (`aio`: stored activation, nullable) also selects the instantiation; the kernel arguments follow
#define NORM_PASS(MODE_, AIO_, ...)                                                                                  \
  do {                                                                                                               \
    auto aio_ = (AIO_);                                                                                              \
    if (st_ == MSEG_ST_BF16) NORM_PASS_A(MODE_, true, __VA_ARGS__);                                                  \
    else NORM_PASS_A(MODE_, false, __VA_ARGS__);                                                                     \
  } while (0)

// 1 (default): the reduction over the chunks also writes the per-channel results where one workgroup can (BatchNorm tables,
// bias gradients, <= 256 channels) instead of leaving them to a further launch; MSEG_NORM_FINISH=0: always two launches
static int g_norm_finish = -1;
extern "C" int mseg_norm_set_finish(int on) {
  g_norm_finish = on < 0 ? -1 : (on > 2 ? 2 : on);     // 2: norm_bn_finish_kernel (any channel count); -1: the default
  return MSEG_OK;
}
static int norm_finish_mode() {
  if (g_norm_finish < 0) {
    const char* e = getenv("MSEG_NORM_FINISH");
    g_norm_finish = e ? atoi(e) : 2;
    if (g_norm_finish < 0 || g_norm_finish > 2) g_norm_finish = 2;
  }
  return g_norm_finish;
}
static bool norm_finish_fits(int C) { return norm_finish_mode() == 2 || (norm_finish_mode() == 1 && C <= 256); }
// the per-channel results of a BatchNorm pass / a bias gradient from the partial sums: one launch (see the two kernels)
static void norm_launch_finish(const double* part, double* nc, const NormGeom& g, int ns, const NormTail& tf, int N, int C,
                               hipStream_t st) {
  if (norm_finish_mode() == 2)
    hipLaunchKernelGGL(norm_bn_finish_kernel, dim3((unsigned)((C + NORM_FB_CH - 1) / NORM_FB_CH)), dim3(N > 8 ? 1024 : 256), 0,
                       st, part, nc, g, ns, tf);
  else
    hipLaunchKernelGGL(norm_reduce_finish_kernel, dim3(nblocks((size_t)ns * N * C * 8, 4096u)), dim3(256), 0, st, part, nc, g,
                       ns, tf);
}
static int g_norm_tails = -1;          // -1: not set yet (MSEG_NORM_TAILS in the environment, else the default below)
// Test / ablation hook: 1 = the last workgroups of a pass finish the reductions in the kernel, 0 = every pass is followed by
// the separate reduction / finalize launches (same results, bit for bit)
extern "C" int mseg_norm_set_tails(int on) {
  g_norm_tails = on ? 1 : 0;
  return MSEG_OK;
}
static bool norm_tails_fit(const NormGeom& g) {
  if (g_norm_tails < 0) {
    const char* e = getenv("MSEG_NORM_TAILS");
    g_norm_tails = e ? (atoi(e) != 0) : 0;
  }
  return g_norm_tails && (long long)g.N * g.slices + (g.N > g.slices ? g.N : g.slices) <= NORM_CTR_MAX;
}

extern "C" int mseg_norm_stats(const void* z, int N, int HW, int C, int st_, int act, int norm, const float* gamma,
                               const float* beta, float eps, float* scale, float* shift, float* mean, float* rstd,
                               float* running_mean, float* running_var, float momentum, void* act_out, void* ws,
                               void* stream) {
  if (!z || !scale || !shift || !mean || !rstd || !ws || N <= 0 || HW <= 0 || C <= 0 || (C & 3)) return MSEG_EINVAL;
  if (st_ != MSEG_ST_F32 && st_ != MSEG_ST_BF16) return MSEG_EINVAL;
  if (st_ == MSEG_ST_BF16 && (C & 7)) return MSEG_EINVAL;
  if (norm == MSEG_NORM_GN && (C % 8)) return MSEG_EINVAL;
  if (norm < 0 || norm > 2) return MSEG_EINVAL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return MSEG_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  NormGeom g = norm_geom(N, HW, C, norm_vec(st_));
  double* part = (double*)((char*)ws + NORM_CTR_BYTES);
  double* nc = part + (size_t)N * g.chunks * 3 * C;
  NormTail t = {};
  t.kind = norm_tails_fit(g) ? (norm == MSEG_NORM_BN ? NORM_TAIL_BN_FWD : NORM_TAIL_PS_FWD) : NORM_TAIL_NONE;
  t.norm = norm; t.ctr = (unsigned*)ws; t.nc = nc; t.gamma = gamma; t.beta = beta; t.eps = eps; t.momentum = momentum;
  t.scale = scale; t.shift = shift; t.mean_out = mean; t.rstd_out = rstd;
  t.running_mean = running_mean; t.running_var = running_var;
  NORM_PASS(0, act_out, z, (const void*)nullptr, (void*)nullptr, (const float*)nullptr, (const float*)nullptr,
            (const float*)nullptr, 0, g, act, part, aio_, t);
  MSEG_LAUNCH_CHECK();
  if (t.kind != NORM_TAIL_NONE) return MSEG_OK;
  if (norm == MSEG_NORM_BN && norm_finish_fits(C)) {
    t.kind = NORM_TAIL_BN_FWD;
    norm_launch_finish((const double*)part, nc, g, 2, t, N, C, st);
    MSEG_LAUNCH_CHECK();
    return MSEG_OK;
  }
  hipLaunchKernelGGL(norm_reduce_chunks_kernel, dim3(nblocks((size_t)2 * N * C * 8, 4096u)), dim3(256), 0, st,
                     (const double*)part, nc, g, 2);
  MSEG_LAUNCH_CHECK();
  hipLaunchKernelGGL(norm_fwd_finalize_kernel, dim3(nblocks(norm == MSEG_NORM_BN ? C : (size_t)N * C)), dim3(256), 0,
                     st, (const double*)nc, g, norm, gamma, beta, eps, scale, shift, mean, rstd, running_mean,
                     running_var, momentum, 0.0);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

// ---- statistics from a convolution's epilogue (MsegIgemm.stats) ------------------------------------------------------------
// part[rows][2][C] fp32 (one row per pixel tile and wave row of the producing kernel) -> nc[s][slice][c] fp64: the rows are
// cut into CONV_STAT_SLICES contiguous slices, a workgroup sums 32 channels of one slice and one statistic (8 thread groups
// stride the slice, combined through LDS in fixed order); norm_fwd_finalize_kernel then adds the slices like samples.
#define CONV_STAT_SLICES 16
__global__ __launch_bounds__(256) void conv_stats_reduce_kernel(const float* __restrict__ part, double* __restrict__ nc,
                                                                int rows, int C, int rows_per_slice) {
  __shared__ double red[8][32];
  const int o = threadIdx.x & 31, kg = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + o, slice = blockIdx.y, s = blockIdx.z;
  const int r0 = slice * rows_per_slice;
  const int r1 = r0 + rows_per_slice < rows ? r0 + rows_per_slice : rows;
  double acc = 0.0;
  if (c < C) {
    const float* base = part + (size_t)s * C + c;
#pragma unroll 4
    for (int r = r0 + kg; r < r1; r += 8) acc += (double)base[(size_t)r * 2 * C];
  }
  red[kg][o] = acc;
  __syncthreads();
  if (kg == 0 && c < C) {
    double t = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) t += red[j][o];
    nc[((size_t)s * CONV_STAT_SLICES + slice) * C + c] = t;
  }
}

extern "C" int mseg_norm_stats_from_conv(const float* part, int rows, int C, long long count, const float* gamma,
                                         const float* beta, float eps, float* scale, float* shift, float* mean,
                                         float* rstd, float* running_mean, float* running_var, float momentum, void* ws,
                                         void* stream) {
  if (!part || !scale || !shift || !mean || !rstd || !ws || rows <= 0 || C <= 0 || count <= 0) return MSEG_EINVAL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return MSEG_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  double* nc = (double*)((char*)ws + NORM_CTR_BYTES);
  const int rps = (rows + CONV_STAT_SLICES - 1) / CONV_STAT_SLICES;
  hipLaunchKernelGGL(conv_stats_reduce_kernel, dim3((C + 31) / 32, CONV_STAT_SLICES, 2), dim3(256), 0, st, part, nc, rows,
                     C, rps);
  MSEG_LAUNCH_CHECK();
  NormGeom g = {};
  g.N = CONV_STAT_SLICES; g.HW = 1; g.C = C;         // (slices past the last row hold zeros: their loops are empty)
  hipLaunchKernelGGL(norm_fwd_finalize_kernel, dim3(nblocks(C)), dim3(256), 0, st, (const double*)nc, g, MSEG_NORM_BN,
                     gamma, beta, eps, scale, shift, mean, rstd, running_mean, running_var, momentum, (double)count);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

extern "C" int mseg_activation(const void* z, int N, int HW, int C, int st_, int act, void* act_out, void* stream) {
  if (!z || !act_out || N <= 0 || HW <= 0 || C <= 0 || (C & 3)) return MSEG_EINVAL;
  if ((st_ != MSEG_ST_F32 && st_ != MSEG_ST_BF16) || (st_ == MSEG_ST_BF16 && (C & 7))) return MSEG_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  NormGeom g = norm_geom(N, HW, C, norm_vec(st_));
  NormTail t = {};
  NORM_PASS(3, act_out, z, (const void*)nullptr, (void*)nullptr, (const float*)nullptr, (const float*)nullptr,
            (const float*)nullptr, 0, g, act, (double*)nullptr, aio_, t);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

extern "C" int mseg_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                                   const float* running_var, float eps, int C, float* scale, float* shift,
                                   void* stream) {
  if (!running_mean || !running_var || !scale || !shift || C <= 0) return MSEG_EINVAL;
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                     running_mean, running_var, eps, C, scale, shift);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

extern "C" int mseg_norm_bwd(const void* gy, const void* z, int N, int HW, int C, int st_, int act, int norm,
                             const float* gamma, const float* mean, const float* rstd, void* dz, float* dgamma,
                             float* dbeta, float* dbias, const void* act_in, void* ws, void* stream) {
  if (!gy || !z || !mean || !rstd || !dz || !ws || N <= 0 || HW <= 0 || C <= 0 || (C & 3)) return MSEG_EINVAL;
  if ((st_ != MSEG_ST_F32 && st_ != MSEG_ST_BF16) || (st_ == MSEG_ST_BF16 && (C & 7))) return MSEG_EINVAL;
  if (norm < 0 || norm > 2) return MSEG_EINVAL;
  if (norm == MSEG_NORM_GN && (C % 8)) return MSEG_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  NormGeom g = norm_geom(N, HW, C, norm_vec(st_));
  double* part = (double*)((char*)ws + NORM_CTR_BYTES);
  double* nc = part + (size_t)N * g.chunks * 3 * C;
  float* k1 = (float*)(nc + (size_t)3 * N * C);
  float* k2 = k1 + (size_t)N * C;
  float* k3 = k2 + (size_t)N * C;
  const int kss = (norm == MSEG_NORM_BN) ? 0 : C;
  const bool tails = norm_tails_fit(g);
  NormTail t = {};
  // BatchNorm: the last workgroups of the sums pass write k1 / k2 / k3, dgamma, dbeta.  Group / InstanceNorm need the sums of
  // a whole sample AND (dgamma, dbeta) of all samples: they keep the two small launches.
  t.kind = (tails && norm == MSEG_NORM_BN) ? NORM_TAIL_BN_BWD : NORM_TAIL_NONE;
  t.norm = norm; t.ctr = (unsigned*)ws; t.nc = nc; t.gamma = gamma; t.mean = mean; t.rstd = rstd;
  t.k1 = k1; t.k2 = k2; t.k3 = k3; t.dgamma = dgamma; t.dbeta = dbeta;
  NORM_PASS(1, const_cast<void*>(act_in), z, gy, (void*)nullptr, (const float*)nullptr, (const float*)nullptr,
            (const float*)nullptr, 0, g, act, part, aio_, t);
  MSEG_LAUNCH_CHECK();
  if (t.kind == NORM_TAIL_NONE && norm == MSEG_NORM_BN && norm_finish_fits(C)) {
    NormTail tf = t;
    tf.kind = NORM_TAIL_BN_BWD;
    norm_launch_finish((const double*)part, nc, g, 2, tf, N, C, st);
    MSEG_LAUNCH_CHECK();
  } else if (t.kind == NORM_TAIL_NONE) {
    hipLaunchKernelGGL(norm_reduce_chunks_kernel, dim3(nblocks((size_t)2 * N * C * 8, 4096u)), dim3(256), 0, st,
                       (const double*)part, nc, g, 2);
    MSEG_LAUNCH_CHECK();
    hipLaunchKernelGGL(norm_bwd_finalize_kernel, dim3(nblocks(norm == MSEG_NORM_BN ? C : (size_t)N * C)), dim3(256), 0,
                       st, (const double*)nc, g, norm, gamma, mean, rstd, k1, k2, k3, dgamma, dbeta);
    MSEG_LAUNCH_CHECK();
  }
  double* part2 = part;
  NormTail t2 = {};
  t2.kind = (tails && dbias) ? NORM_TAIL_COLSUM : NORM_TAIL_NONE;
  t2.norm = norm; t2.ctr = (unsigned*)ws; t2.nc = nc; t2.colsum = dbias;
  NORM_PASS(2, const_cast<void*>(act_in), z, gy, dz, (const float*)k1, (const float*)k2, (const float*)k3, kss, g, act,
            part2, aio_, t2);
  MSEG_LAUNCH_CHECK();
  if (dbias && t2.kind == NORM_TAIL_NONE && norm_finish_fits(C)) {
    NormTail tf = t2;
    tf.kind = NORM_TAIL_COLSUM;
    norm_launch_finish((const double*)part2, nc, g, 1, tf, N, C, st);
    MSEG_LAUNCH_CHECK();
  } else if (dbias && t2.kind == NORM_TAIL_NONE) {
    hipLaunchKernelGGL(norm_reduce_chunks_kernel, dim3(nblocks((size_t)N * C * 8, 4096u)), dim3(256), 0, st,
                       (const double*)part2, nc, g, 1);
    MSEG_LAUNCH_CHECK();
    hipLaunchKernelGGL(norm_colsum_kernel, dim3(nblocks(C)), dim3(256), 0, st, (const double*)nc, g, dbias);
    MSEG_LAUNCH_CHECK();
  }
  return MSEG_OK;
}

// ---- MaxPool2d(kernel 2, stride 2) on a norm-on-load operand (pool_method='max': src/utils/unets.py:306-307,363-364)
// forward: out[n][y][x][c] = max over the 2x2 window of act(z)*scale+shift   (plain NHWC tensor, no further transform)
// backward: the gradient goes to the FIRST maximum of the window in row-major order (torch's MaxPool2d rule).
__global__ void maxpool_fwd_kernel(const MsegSrc s, int N, int H, int W, float* __restrict__ out) {
  const int Ho = H >> 1, Wo = W >> 1, C4 = s.C >> 2;
  const size_t total = (size_t)N * Ho * Wo * C4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    size_t r = i / C4;
    const int x = (int)(r % Wo); r /= Wo;
    const int y = (int)(r % Ho);
    const int n = (int)(r / Ho);
    float4 best;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const size_t pix = ((size_t)n * H + 2 * y + (k >> 1)) * W + 2 * x + (k & 1);
      float4 v = *reinterpret_cast<const float4*>(s.ptr + pix * s.C + c);
      v = src_transform4(v, s, n, c);
      if (k == 0) best = v;
      else { best.x = v.x > best.x ? v.x : best.x; best.y = v.y > best.y ? v.y : best.y;
             best.z = v.z > best.z ? v.z : best.z; best.w = v.w > best.w ? v.w : best.w; }
    }
    *reinterpret_cast<float4*>(out + i * 4) = best;
  }
}

__global__ void maxpool_bwd_kernel(const MsegSrc s, int N, int H, int W, const float* __restrict__ gout,
                                   float* __restrict__ gin, int accumulate) {
  const int Ho = H >> 1, Wo = W >> 1, C4 = s.C >> 2;
  const size_t total = (size_t)N * Ho * Wo * C4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    size_t r = i / C4;
    const int x = (int)(r % Wo); r /= Wo;
    const int y = (int)(r % Ho);
    const int n = (int)(r / Ho);
    float4 v[4];
    size_t off[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const size_t pix = ((size_t)n * H + 2 * y + (k >> 1)) * W + 2 * x + (k & 1);
      off[k] = pix * s.C + c;
      v[k] = src_transform4(*reinterpret_cast<const float4*>(s.ptr + off[k]), s, n, c);
    }
    const float4 g = *reinterpret_cast<const float4*>(gout + i * 4);
    int ax = 0, ay = 0, az = 0, aw = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      if (v[k].x > v[ax].x) ax = k;
      if (v[k].y > v[ay].y) ay = k;
      if (v[k].z > v[az].z) az = k;
      if (v[k].w > v[aw].w) aw = k;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float4 o = make_float4(ax == k ? g.x : 0.f, ay == k ? g.y : 0.f, az == k ? g.z : 0.f, aw == k ? g.w : 0.f);
      if (accumulate) {
        const float4 old = *reinterpret_cast<const float4*>(gin + off[k]);
        o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
      }
      *reinterpret_cast<float4*>(gin + off[k]) = o;
    }
  }
}

extern "C" int mseg_maxpool2x2_fwd(const MsegSrc* src, int N, int H, int W, float* out, void* stream) {
  if (!src || !src->ptr || !out || N <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || (src->C & 3)) return MSEG_EINVAL;
  const size_t total = (size_t)N * (H / 2) * (W / 2) * (src->C / 4);
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(nblocks(total, 8192u)), dim3(256), 0, (hipStream_t)stream, *src, N, H, W, out);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

extern "C" int mseg_maxpool2x2_bwd(const MsegSrc* src, int N, int H, int W, const float* gout, float* gin,
                                   int accumulate, void* stream) {
  if (!src || !src->ptr || !gout || !gin || N <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || (src->C & 3))
    return MSEG_EINVAL;
  const size_t total = (size_t)N * (H / 2) * (W / 2) * (src->C / 4);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(nblocks(total, 8192u)), dim3(256), 0, (hipStream_t)stream, *src, N, H, W,
                     gout, gin, accumulate);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}
