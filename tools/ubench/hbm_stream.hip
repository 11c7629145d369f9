// GPU box: what plain streaming kernels reach on this HBM, as a yardstick for the normalisation passes (csrc/norm.hip):
// read-only, read + read, read + read + write of 16-byte vectors; grid size, loads in flight and the walk order varied.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_stream tools/ubench/hbm_stream.hip && /tmp/hbm_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// MODE 0: sum of a; 1: sum of a and b; 2: c = a + b.  U vectors in flight per thread.  CHUNKED: a workgroup owns a
// contiguous range (like norm_pass_kernel); otherwise grid-stride.
template <int MODE, int U, bool CHUNKED>
__global__ __launch_bounds__(256) void stream_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b,
                                                     uint4* __restrict__ c, size_t n, unsigned* __restrict__ out) {
  unsigned acc = 0;
  size_t begin, end, stride;
  if (CHUNKED) {
    const size_t per = (n + gridDim.x - 1) / gridDim.x;
    begin = blockIdx.x * per + threadIdx.x;
    end = (blockIdx.x + 1) * per < n ? (blockIdx.x + 1) * per : n;
    stride = 256;
  } else {
    begin = (size_t)blockIdx.x * 256 + threadIdx.x;
    end = n;
    stride = (size_t)gridDim.x * 256;
  }
  for (size_t i = begin; i < end; i += stride * U) {
    uint4 va[U], vb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t j = i + u * stride;
      const size_t k = j < end ? j : i;
      va[u] = a[k];
      if (MODE >= 1) vb[u] = b[k];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t j = i + u * stride;
      if (MODE == 2) {
        if (j < end) c[j] = make_uint4(va[u].x + vb[u].x, va[u].y + vb[u].y, va[u].z + vb[u].z, va[u].w + vb[u].w);
      } else {
        acc += va[u].x ^ va[u].y ^ va[u].z ^ va[u].w;
        if (MODE == 1) acc += vb[u].x ^ vb[u].y ^ vb[u].z ^ vb[u].w;
      }
    }
  }
  if (MODE != 2 && acc == 0x12345678u) out[0] = acc;
}

template <int MODE, int U, bool CHUNKED>
static void run(const char* name, uint4* a, uint4* b, uint4* c, size_t n, unsigned* out, int wgs) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stream_kernel<MODE, U, CHUNKED>), dim3(wgs), dim3(256), 0, 0, a, b, c, n, out);
  CHECK(hipEventRecord(e0, 0));
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stream_kernel<MODE, U, CHUNKED>), dim3(wgs), dim3(256), 0, 0, a, b, c, n, out);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)n * 16 * (MODE + 1);
  printf("%-22s U=%d %s wgs=%5d: %8.1f us  %5.2f TB/s\n", name, U, CHUNKED ? "chunked" : "strided", wgs, ms / reps * 1e3,
         bytes / (ms / reps * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
  const size_t mb = argc > 1 ? atoi(argv[1]) : 420;
  const size_t n = mb * 1024 * 1024 / 16;
  uint4 *a, *b, *c;
  unsigned* out;
  CHECK(hipMalloc(&a, n * 16)); CHECK(hipMalloc(&b, n * 16)); CHECK(hipMalloc(&c, n * 16)); CHECK(hipMalloc(&out, 64));
  CHECK(hipMemset(a, 1, n * 16)); CHECK(hipMemset(b, 2, n * 16));
  printf("tensors of %zu MB\n", mb);
  for (int wgs : {1024, 2048, 4096, 8192}) {
    run<0, 4, true>("read", a, b, c, n, out, wgs);
    run<0, 8, true>("read", a, b, c, n, out, wgs);
    run<0, 8, false>("read", a, b, c, n, out, wgs);
    run<1, 4, true>("read + read", a, b, c, n, out, wgs);
    run<1, 4, false>("read + read", a, b, c, n, out, wgs);
    run<2, 2, true>("read + read + write", a, b, c, n, out, wgs);
    run<2, 4, true>("read + read + write", a, b, c, n, out, wgs);
    run<2, 4, false>("read + read + write", a, b, c, n, out, wgs);
  }
  return 0;
}
