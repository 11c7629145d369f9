// bf16_affine.h — the norm-on-load transform of 8 bf16 channels with as few vector instructions as it takes.
// The staging code of the bf16 matrix kernels shares the SIMD's issue port with the MFMAs of the partner wave: in the
// weight-gradient kernel the transform of the Q rows was ~130 of the ~190 VALU instructions per 36 MFMAs (measured:
// removing 45 LDS reads per block changed nothing, adding 36 VALU cost 3 %).  Per 8 channels:
//   ReLU (or identity) on the PACKED bf16 pairs: v_pk_max_i16 against 0 (or -32768) — the order of bf16 values of either
//   sign bit against +0 is the order of their bit patterns as int16 — 4 instead of 8 v_max_f32;
//   widen (8), fused multiply-add in fp32 (8), narrow (4 v_cvt_pk_bf16_f32);
//   padding rows become zeros by 4 v_cndmask on the packed result instead of 8 multiplies by a 0 / 1 mask.
// Values are those of  bf16(fma(max(x, lo), scale, shift))  exactly as the float4 form computes them (a dead row's -0
// becomes +0: no difference to a sum).
#pragma once
#include "common.h"

typedef short mseg_s16x2 __attribute__((ext_vector_type(2)));

// lo16: 0 = ReLU, 0x80008000 = identity
__device__ __forceinline__ unsigned mseg_relu_bf16x2(unsigned v, unsigned lo16) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(mseg_s16x2, v),
                                                                __builtin_bit_cast(mseg_s16x2, lo16)));
}

__device__ __forceinline__ uint4 mseg_affine8_bf16(uint4 r, unsigned lo16, const float4& sc, const float4& sh,
                                                   const float4& sc2, const float4& sh2, bool live) {
  r.x = mseg_relu_bf16x2(r.x, lo16); r.y = mseg_relu_bf16x2(r.y, lo16);
  r.z = mseg_relu_bf16x2(r.z, lo16); r.w = mseg_relu_bf16x2(r.w, lo16);
  const unsigned d0 = pack_bf16x2(fmaf(bf16_lo(r.x), sc.x, sh.x), fmaf(bf16_hi(r.x), sc.y, sh.y));
  const unsigned d1 = pack_bf16x2(fmaf(bf16_lo(r.y), sc.z, sh.z), fmaf(bf16_hi(r.y), sc.w, sh.w));
  const unsigned d2 = pack_bf16x2(fmaf(bf16_lo(r.z), sc2.x, sh2.x), fmaf(bf16_hi(r.z), sc2.y, sh2.y));
  const unsigned d3 = pack_bf16x2(fmaf(bf16_lo(r.w), sc2.z, sh2.z), fmaf(bf16_hi(r.w), sc2.w, sh2.w));
  return live ? make_uint4(d0, d1, d2, d3) : make_uint4(0u, 0u, 0u, 0u);
}
