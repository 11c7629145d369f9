// common.h — shared device helpers for libmseg_hip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mseg_hip.h"

extern "C" void mseg_set_hip_error(int e);
extern "C" int mseg_dispatch_dry(void);

#define MSEG_LAUNCH_CHECK()                           \
  do {                                                \
    if (mseg_dispatch_dry()) break;                   \
    hipError_t e__ = hipGetLastError();               \
    if (e__ != hipSuccess) {                          \
      mseg_set_hip_error((int)e__);                   \
      return MSEG_ELAUNCH;                            \
    }                                                 \
  } while (0)

// Every kernel launch of the dispatchers goes through these: the call records which kernel a launch maps to
// (mseg_last_kernel, MsegKernelInfo) and — in a query (mseg_igemm_query / mseg_wgrad_query) — launches nothing, so the
// query IS the dispatch code, not a copy of its rules.  MSEG_KL: the call's main (matrix) kernel; MSEG_KL_AUX: helpers
// (table initialisation, split-K reductions).  K is the kernel in parentheses, e.g. (igemm_halo_kernel<128, 1>).
extern "C" void mseg_note_launch(const char* kernel, unsigned grid, unsigned block, int aux);
extern "C" void mseg_dispatch_begin(int dry);
extern "C" void mseg_dispatch_end(MsegKernelInfo* info);
extern "C" void mseg_dispatch_note(int precision, size_t workspace);
extern "C" void mseg_dispatch_note_stats(int rows);
#define MSEG_KL(K, grid_, block_, shmem_, st_, ...)                                              \
  do {                                                                                           \
    mseg_note_launch(#K, (unsigned)(grid_).x, (unsigned)(block_).x, 0);                          \
    if (!mseg_dispatch_dry()) hipLaunchKernelGGL(K, grid_, block_, shmem_, st_, ##__VA_ARGS__);    \
  } while (0)
#define MSEG_KL_AUX(K, grid_, block_, shmem_, st_, ...)                                          \
  do {                                                                                           \
    mseg_note_launch(#K, (unsigned)(grid_).x, (unsigned)(block_).x, 1);                          \
    if (!mseg_dispatch_dry()) hipLaunchKernelGGL(K, grid_, block_, shmem_, st_, ##__VA_ARGS__);    \
  } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- activations (reference: src/utils/unets.py:81-89,115-124) -------------------------------------------
__device__ __forceinline__ float mseg_softplus(float x) {
  // torch.nn.functional.softplus(beta=1, threshold=20)
  return x > 20.f ? x : log1pf(expf(x));
}

// Mish (unets.py:87-89: x * tanh(F.softplus(x)), softplus with torch's threshold 20) with ONE exponential instead of
// exp + log1p + tanh:  tanh(log(1 + w)) = ((1 + w)^2 - 1) / ((1 + w)^2 + 1) = n / (n + 2),  n = w (w + 2),  w = e^x.
// Algebraically identical, no cancellation anywhere (n >= 0), a few ulp from the three-call form; the transcendental-heavy
// form made the normalisation passes of a Mish network VALU-bound (10 + 14 ms of a 70 ms bf16 step).  FAST (bf16 tensor
// storage, where 2^-9 rounding follows anyway): hardware exp2 / rcp instead of the correctly rounded library calls.
template <bool FAST>
__device__ __forceinline__ float mish_fwd(float x) {
  if (x > 20.f) return x;
  const float w = FAST ? __expf(x) : expf(x);
  const float n = w * (w + 2.f);
  return FAST ? x * n * __frcp_rn(n + 2.f) : x * (n / (n + 2.f));
}
// d/dx [x tanh(softplus(x))] = t + x (1 - t^2) sigmoid(x) with t = n / (n + 2):  1 - t^2 = 4 (w + 1)^2 / (n + 2)^2 and
// sigmoid = w / (w + 1)  ->  t + 4 x w (w + 1) / (n + 2)^2
template <bool FAST>
__device__ __forceinline__ float mish_bwd(float x) {
  if (x > 20.f) return 1.f;
  const float w = FAST ? __expf(x) : expf(x);
  const float n = w * (w + 2.f);
  const float r = FAST ? __frcp_rn(n + 2.f) : 1.f / (n + 2.f);
  return n * r + 4.f * x * w * (w + 1.f) * r * r;
}

template <int ACT>
__device__ __forceinline__ float act_fwd_t(float x) {
  if (ACT == MSEG_ACT_RELU) return x > 0.f ? x : 0.f;
  if (ACT == MSEG_ACT_LEAKY) return x > 0.f ? x : 0.01f * x;
  if (ACT == MSEG_ACT_ELU) return x > 0.f ? x : expm1f(x);
  if (ACT == MSEG_ACT_MISH) return mish_fwd<false>(x);
  return x;
}

__device__ __forceinline__ float act_fwd(float x, int act) {
  switch (act) {
    case MSEG_ACT_RELU: return act_fwd_t<MSEG_ACT_RELU>(x);
    case MSEG_ACT_LEAKY: return act_fwd_t<MSEG_ACT_LEAKY>(x);
    case MSEG_ACT_ELU: return act_fwd_t<MSEG_ACT_ELU>(x);
    case MSEG_ACT_MISH: return act_fwd_t<MSEG_ACT_MISH>(x);
    default: return x;
  }
}

// d act(z) / dz, following torch autograd of the reference modules
__device__ __forceinline__ float act_bwd(float z, int act) {
  switch (act) {
    case MSEG_ACT_RELU: return z > 0.f ? 1.f : 0.f;
    case MSEG_ACT_LEAKY: return z > 0.f ? 1.f : 0.01f;
    case MSEG_ACT_ELU: return z > 0.f ? 1.f : expf(z);  // alpha * exp(z)
    case MSEG_ACT_MISH: return mish_bwd<false>(z);
    default: return 1.f;
  }
}

// bf16-storage flavours (FAST Mish / ELU: the result is rounded to 8 significant bits right after)
__device__ __forceinline__ float act_fwd_fast(float x, int act) {
  if (act == MSEG_ACT_MISH) return mish_fwd<true>(x);
  if (act == MSEG_ACT_ELU) return x > 0.f ? x : __expf(x) - 1.f;
  return act_fwd(x, act);
}
__device__ __forceinline__ float act_bwd_fast(float z, int act) {
  if (act == MSEG_ACT_MISH) return mish_bwd<true>(z);
  if (act == MSEG_ACT_ELU) return z > 0.f ? 1.f : __expf(z);
  return act_bwd(z, act);
}

// a = act(z) and act'(z) together (the backward norm pass needs both; Mish / ELU share their exponential).  `a` is
// bit-identical to act_fwd / act_fwd_fast of the same z — the forward pass stored exactly that value.
template <bool FAST>
__device__ __forceinline__ void act_pair(float z, int act, float& a, float& d) {
  if (act == MSEG_ACT_MISH) {
    if (z > 20.f) { a = z; d = 1.f; return; }
    const float w = FAST ? __expf(z) : expf(z);
    const float n = w * (w + 2.f);
    if (FAST) {
      const float r = __frcp_rn(n + 2.f);
      a = z * n * r;
      d = n * r + 4.f * z * w * (w + 1.f) * r * r;
    } else {
      const float r = 1.f / (n + 2.f);
      a = z * (n / (n + 2.f));
      d = n * r + 4.f * z * w * (w + 1.f) * r * r;
    }
    return;
  }
  a = FAST ? act_fwd_fast(z, act) : act_fwd(z, act);
  d = FAST ? act_bwd_fast(z, act) : act_bwd(z, act);
}

__device__ __forceinline__ float4 act_fwd4(float4 v, int act) {
  if (act == MSEG_ACT_NONE) return v;
  if (act == MSEG_ACT_RELU) {
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    return v;
  }
  v.x = act_fwd(v.x, act); v.y = act_fwd(v.y, act); v.z = act_fwd(v.z, act); v.w = act_fwd(v.w, act);
  return v;
}

// 4 consecutive channels at element offset `e` of an MsegSrc tensor (fp32 or bf16 storage)
__device__ __forceinline__ float4 src_load4(const MsegSrc& s, size_t e);

// norm-on-load of 4 consecutive channels c..c+3 of sample n from an MsegSrc
__device__ __forceinline__ float4 src_transform4(float4 v, const MsegSrc& s, int n, int c) {
  v = act_fwd4(v, s.act);
  if (s.scale) {
    const float4 sc = *reinterpret_cast<const float4*>(s.scale + (size_t)n * s.ss + c);
    const float4 sh = *reinterpret_cast<const float4*>(s.shift + (size_t)n * s.ss + c);
    v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
  }
  return v;
}

// ---- bf16 tensor storage (MSEG_ST_BF16: BASELINE configs[2], "bf16 forward / backward") -----------------------------------
// Activations (pre-activation conv outputs z, materialised activations) and activation gradients may live in HBM as bf16;
// every kernel computes in fp32: widening is a 16-bit shift, narrowing is round-to-nearest-even (v_cvt_pk_bf16_f32).
typedef __bf16 mseg_bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf16_lo(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bf16_hi(unsigned v) { return __uint_as_float(v & 0xffff0000u); }
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  mseg_bf16x2 v;
  v[0] = (__bf16)a; v[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float4 bf16x4_to_f32(uint2 v) {
  return make_float4(bf16_lo(v.x), bf16_hi(v.x), bf16_lo(v.y), bf16_hi(v.y));
}
__device__ __forceinline__ uint2 f32x4_to_bf16(float4 v) { return make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)); }

// 4 consecutive channels at element offset `e` of a tensor stored as fp32 (S16 = false) or bf16 (true)
template <bool S16>
__device__ __forceinline__ float4 ld_f4(const void* base, size_t e) {
  if (S16) return bf16x4_to_f32(*reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + e));
  return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + e);
}
template <bool S16>
__device__ __forceinline__ void st_f4(void* base, size_t e, float4 v) {
  if (S16) *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(base) + e) = f32x4_to_bf16(v);
  else *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + e) = v;
}
// run-time flavour for the HBM-bound kernels that are not worth two instantiations
__device__ __forceinline__ float4 ld_f4_rt(const void* base, size_t e, int bf16) {
  return bf16 ? ld_f4<true>(base, e) : ld_f4<false>(base, e);
}
__device__ __forceinline__ void st_f4_rt(void* base, size_t e, float4 v, int bf16) {
  if (bf16) st_f4<true>(base, e, v); else st_f4<false>(base, e, v);
}
__device__ __forceinline__ float ld_f1_rt(const void* base, size_t e, int bf16) {
  return bf16 ? __uint_as_float((unsigned)reinterpret_cast<const uint16_t*>(base)[e] << 16)
              : reinterpret_cast<const float*>(base)[e];
}

__device__ __forceinline__ float4 src_load4(const MsegSrc& s, size_t e) { return ld_f4_rt(s.ptr, e, s.dtype); }

// XCD-aware workgroup order (MI355X: 8 XCDs, each with a private L2; hardware deals consecutive workgroup ids
// round-robin over the XCDs).  Maps the hardware id to a logical id such that each XCD works on one CONTIGUOUS range of
// logical ids: workgroups that share an operand slab (neighbouring logical ids) then share an L2.  Bijective for any
// grid size; affects speed only, never correctness.
__device__ __forceinline__ unsigned xcd_logical_id(unsigned hw_id, unsigned nwg) {
  const unsigned q = nwg >> 3, r = nwg & 7u, xcd = hw_id & 7u;
  return (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + (hw_id >> 3);
}

// max(v, lo) as ONE VALU instruction.  fmaxf / fmed3f lower to a canonicalising v_max(v, v) followed by the v_max proper;
// the instruction itself already returns the non-NaN operand, so it is emitted directly.  lo = 0 -> ReLU,
// -FLT_MAX -> no-op.
__device__ __forceinline__ float clamp_lo(float v, float lo) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(lo));
  return r;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
