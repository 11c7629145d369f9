// Microbenchmark: does v_mfma_f32_32x32x2_f32 co-execute with VALU work of another wave on the same SIMD?
// 512-thread blocks = 2 waves per SIMD: waves 0-3 run MFMAs, waves 4-7 run fp32 FMAs (mode 2), or only one half runs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void k(float* out, int mode, int iters) {
  const int wave = threadIdx.x >> 6;
  float r = 0.f;
  if (wave < 4) {
    if (mode == 0 || mode == 2 || mode == 3) {
      f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
      float a = threadIdx.x * 1e-3f, b = 1.0001f;
      for (int i = 0; i < iters; ++i) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc3, 0, 0, 0);
      }
      r = acc0[0] + acc1[1] + acc2[2] + acc3[3];
    }
  } else {
    if (mode == 1 || mode == 2) {       // fp32 FMA chain x8 independent
      float x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3, x4 = 4, x5 = 5, x6 = 6, x7 = 7;
      const float m = 1.0000001f, c = 1e-7f;
      for (int i = 0; i < iters * 16; ++i) {
        x0 = fmaf(x0, m, c); x1 = fmaf(x1, m, c); x2 = fmaf(x2, m, c); x3 = fmaf(x3, m, c);
        x4 = fmaf(x4, m, c); x5 = fmaf(x5, m, c); x6 = fmaf(x6, m, c); x7 = fmaf(x7, m, c);
      }
      r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    }
    if (mode == 3) {                     // integer VALU (address-math like)
      unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3, x4 = 4, x5 = 5, x6 = 6, x7 = 7;
      for (int i = 0; i < iters * 16; ++i) {
        x0 = x0 * 3u + 1u; x1 = x1 * 3u + 1u; x2 = x2 * 3u + 1u; x3 = x3 * 3u + 1u;
        x4 = x4 * 3u + 1u; x5 = x5 * 3u + 1u; x6 = x6 * 3u + 1u; x7 = x7 * 3u + 1u;
      }
      r = (float)(x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  const int iters = 20000;
  const char* names[] = {"MFMA only (waves 0-3)", "FMA only (waves 4-7)", "MFMA + FMA", "MFMA + int VALU"};
  for (int mode = 0; mode < 4; ++mode) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, mode, iters); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, mode, iters); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    double mfma_tf = 4.0 * iters * 4096.0 * 4 * 256 / (ms * 1e-3) / 1e12;
    printf("%-26s %8.3f ms   (MFMA-equivalent %.1f TF/s)\n", names[mode], ms, mfma_tf);
  }
  return 0;
}
