// head.hip — the 1x1 output convolutions (src/utils/unets.py:347 UNet head, :460-461 DUNet heads).
// HBM-bound: one read of the last decoder tensor (normalised on load), Co <= 4 dot products per pixel.
// Output / incoming gradient are NCHW ([N][Co][HW]) because that is the layout of the Python boundary.
#include "common.h"

#define HEAD_MAXCO 4

// A group of `lpp` lanes owns a pixel: every lane loads 16 bytes of it (4 fp32 / 8 bf16 channels per trip), multiplies by the
// Co weight rows and the group reduces by shuffles.  U pixels are in flight per group (all loads issued before the first
// use): with one load per trip the pass was latency-bound at 0.6 TB/s.
template <bool S16>
__global__ __launch_bounds__(256) void head_fwd_kernel(const MsegSrc s, int N, int HW, const float* __restrict__ w,
                                                       const float* __restrict__ b, int Co, int lpp,
                                                       float* __restrict__ out) {
  constexpr int V = S16 ? 8 : 4, U = 4;
  const int CV = s.C / V;
  const long long total = (long long)N * HW;
  const int sub = threadIdx.x % lpp;
  const long long ppb = blockDim.x / lpp;  // pixels per block and slot
  for (long long base = (long long)blockIdx.x * ppb * U; base < total; base += (long long)gridDim.x * ppb * U) {
    // all lanes of a wave stay in the loop together (shuffles below); out-of-range pixels contribute 0
    float acc[U][HEAD_MAXCO];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int co = 0; co < HEAD_MAXCO; ++co) acc[u][co] = 0.f;
    for (int cv = sub; cv < CV; cv += lpp) {
      uint4 raw[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long pix = base + u * ppb + threadIdx.x / lpp;
        const size_t e = (size_t)(pix < total ? pix : 0) * s.C + cv * V;
        raw[u] = S16 ? *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(s.ptr) + e)
                     : *reinterpret_cast<const uint4*>(s.ptr + e);
      }
      float4 wv[HEAD_MAXCO][V / 4];
#pragma unroll
      for (int co = 0; co < HEAD_MAXCO; ++co)
#pragma unroll
        for (int h = 0; h < V / 4; ++h)
          wv[co][h] = co < Co ? *reinterpret_cast<const float4*>(w + (size_t)co * s.C + cv * V + 4 * h)
                              : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long pix = base + u * ppb + threadIdx.x / lpp;
        if (pix >= total) continue;
        const int n = (int)(pix / HW);
#pragma unroll
        for (int h = 0; h < V / 4; ++h) {
          float4 v;
          if (S16) v = bf16x4_to_f32(h == 0 ? make_uint2(raw[u].x, raw[u].y) : make_uint2(raw[u].z, raw[u].w));
          else v = make_float4(__uint_as_float(raw[u].x), __uint_as_float(raw[u].y), __uint_as_float(raw[u].z),
                               __uint_as_float(raw[u].w));
          v = src_transform4(v, s, n, cv * V + 4 * h);
#pragma unroll
          for (int co = 0; co < HEAD_MAXCO; ++co)
            if (co < Co) acc[u][co] += v.x * wv[co][h].x + v.y * wv[co][h].y + v.z * wv[co][h].z + v.w * wv[co][h].w;
        }
      }
    }
    for (int o = lpp >> 1; o > 0; o >>= 1) {
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int co = 0; co < HEAD_MAXCO; ++co) acc[u][co] += __shfl_xor(acc[u][co], o, 64);
    }
    if (sub == 0) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long pix = base + u * ppb + threadIdx.x / lpp;
        if (pix >= total) continue;
        const int n = (int)(pix / HW);
        const int p = (int)(pix - (long long)n * HW);
        for (int co = 0; co < Co; ++co) out[((size_t)n * Co + co) * HW + p] = acc[u][co] + (b ? b[co] : 0.f);
      }
    }
  }
}

static int head_lpp(int CV) {
  int l = 1;
  while (l * 2 <= CV && l * 2 <= 16) l *= 2;
  return l;
}

extern "C" int mseg_head_fwd(const MsegSrc* src, int N, int HW, const float* w, const float* b, int Co,
                             float* out_nchw, void* stream) {
  if (!src || !src->ptr || !w || !out_nchw || N <= 0 || HW <= 0 || Co <= 0 || Co > HEAD_MAXCO) return MSEG_EINVAL;
  if (src->C <= 0 || (src->C & 3)) return MSEG_EINVAL;
  const bool s16 = src->dtype == MSEG_ST_BF16;
  if (s16 && (src->C & 7)) return MSEG_EINVAL;
  const int lpp = head_lpp(src->C / (s16 ? 8 : 4));
  const long long total = (long long)N * HW;
  const long long ppb = (256 / lpp) * 4;       // pixels per block and trip (U = 4 per group)
  long long blocks = (total + ppb - 1) / ppb;
  if (blocks > 8192) blocks = 8192;
  if (s16)
    hipLaunchKernelGGL((head_fwd_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *src, N, HW, w,
                       b, Co, lpp, out_nchw);
  else
    hipLaunchKernelGGL((head_fwd_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *src, N, HW, w,
                       b, Co, lpp, out_nchw);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

// ---- backward ------------------------------------------------------------------------------------------------
struct HeadGeom {
  int N, HW, C, Co, chunks, rows_per_chunk;
};

static HeadGeom head_geom(int N, int HW, int C, int Co) {
  HeadGeom g;
  g.N = N; g.HW = HW; g.C = C; g.Co = Co;
  int maxc = 1024 / (N > 0 ? N : 1);
  if (maxc < 1) maxc = 1;
  int chunks = HW / 512;
  if (chunks > maxc) chunks = maxc;
  if (chunks < 1) chunks = 1;
  g.rows_per_chunk = (HW + chunks - 1) / chunks;
  g.chunks = (HW + g.rows_per_chunk - 1) / g.rows_per_chunk;
  return g;
}

extern "C" size_t mseg_head_bwd_workspace_bytes(int N, int HW, int C, int Co) {
  if (N <= 0 || HW <= 0 || C <= 0 || Co <= 0 || Co > HEAD_MAXCO) return 0;
  HeadGeom g = head_geom(N, HW, C, Co);
  return (size_t)N * g.chunks * (HEAD_MAXCO * (C + 1)) * sizeof(double);
}

// thread owns 4 channels, walks the pixels of its chunk: gy = sum_co g[co]*W[co][c..c+3]; dW partial sums fp64
__global__ __launch_bounds__(256) void head_bwd_kernel(const MsegSrc s, HeadGeom g, const float* __restrict__ w,
                                                       const float* __restrict__ gout, void* __restrict__ gy,
                                                       int gy_dtype, double* __restrict__ part) {
  __shared__ double red[256 * 4 * HEAD_MAXCO];
  __shared__ double redb[256];
  const int tid = threadIdx.x;
  const int chunk = blockIdx.x, n = blockIdx.y;
  const int C4 = g.C >> 2;
  const int CW = C4 < 256 ? C4 : 256;
  const int rpi = 256 / CW;
  const int cq = tid % CW, r0 = tid / CW;
  const int row_begin = chunk * g.rows_per_chunk;
  int row_end = row_begin + g.rows_per_chunk;
  if (row_end > g.HW) row_end = g.HW;
  double* pout = part + ((size_t)n * g.chunks + chunk) * (HEAD_MAXCO * (g.C + 1));

  for (int cbase = 0; cbase < C4; cbase += CW) {
    const int c4 = cbase + cq;
    const bool active = (c4 < C4) && (r0 < rpi);
    double dw[HEAD_MAXCO][4];
    double dbs[HEAD_MAXCO];
#pragma unroll
    for (int co = 0; co < HEAD_MAXCO; ++co) {
      dbs[co] = 0.0;
#pragma unroll
      for (int j = 0; j < 4; ++j) dw[co][j] = 0.0;
    }
    if (active) {
      const int c = c4 * 4;
      float4 wv[HEAD_MAXCO];
#pragma unroll
      for (int co = 0; co < HEAD_MAXCO; ++co)
        wv[co] = co < g.Co ? *reinterpret_cast<const float4*>(w + (size_t)co * g.C + c) : make_float4(0, 0, 0, 0);
      constexpr int U = 1;                               // measured: U = 4 spills the fp64 sums (349 -> 514 us); kept general
      for (int r = row_begin + r0; r < row_end; r += rpi * U) {
        float4 yraw[U];
        float gv[U][HEAD_MAXCO];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int ru = r + u * rpi;
          ok[u] = ru < row_end;
          const int rr = ok[u] ? ru : r;
          yraw[u] = src_load4(s, ((size_t)n * g.HW + rr) * g.C + c);
#pragma unroll
          for (int co = 0; co < HEAD_MAXCO; ++co)
            gv[u][co] = co < g.Co ? gout[((size_t)n * g.Co + co) * g.HW + rr] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (!ok[u]) continue;
          const size_t off = ((size_t)n * g.HW + r + u * rpi) * g.C + c;
          const float4 yv = src_transform4(yraw[u], s, n, c);
          float4 o = make_float4(0, 0, 0, 0);
#pragma unroll
          for (int co = 0; co < HEAD_MAXCO; ++co) {
            if (co < g.Co) {
              const float gvc = gv[u][co];
              o.x += gvc * wv[co].x; o.y += gvc * wv[co].y; o.z += gvc * wv[co].z; o.w += gvc * wv[co].w;
              dw[co][0] += (double)gvc * yv.x; dw[co][1] += (double)gvc * yv.y;
              dw[co][2] += (double)gvc * yv.z; dw[co][3] += (double)gvc * yv.w;
              if (c4 == 0) dbs[co] += gvc;
            }
          }
          st_f4_rt(gy, off, o, gy_dtype);
        }
      }
    }
    for (int co = 0; co < g.Co; ++co) {
#pragma unroll
      for (int j = 0; j < 4; ++j) red[(tid * HEAD_MAXCO + co) * 4 + j] = dw[co][j];
    }
    __syncthreads();
    if (r0 == 0 && c4 < C4) {
      for (int co = 0; co < g.Co; ++co) {
        double t[4] = {0, 0, 0, 0};
        for (int k = 0; k < rpi; ++k)
#pragma unroll
          for (int j = 0; j < 4; ++j) t[j] += red[((k * CW + cq) * HEAD_MAXCO + co) * 4 + j];
#pragma unroll
        for (int j = 0; j < 4; ++j) pout[(size_t)co * g.C + c4 * 4 + j] = t[j];
      }
    }
    __syncthreads();
    if (cbase == 0) {
      // bias gradient: threads with c4 == 0 (cq == 0) hold the per-row-slice sums
      for (int co = 0; co < g.Co; ++co) {
        redb[tid] = (cq == 0 && r0 < rpi) ? dbs[co] : 0.0;
        __syncthreads();
        if (tid == 0) {
          double t = 0.0;
          for (int k = 0; k < rpi; ++k) t += redb[k * CW];
          pout[(size_t)HEAD_MAXCO * g.C + co] = t;
        }
        __syncthreads();
      }
    }
  }
}

// sums the per-workgroup partials in fixed order: 32 outputs x 8 partial groups per workgroup, combined through LDS
__global__ __launch_bounds__(256) void head_bwd_reduce_kernel(const double* __restrict__ part, HeadGeom g,
                                                              float* __restrict__ dW, float* __restrict__ db) {
  __shared__ double red[8][32];
  const int per = HEAD_MAXCO * (g.C + 1);
  const int blocks = g.N * g.chunks;
  const int total = g.Co * g.C + g.Co;
  const int o = threadIdx.x & 31, kg = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + o;
  double s = 0.0;
  if (i < total) {
    const int slot = (i < g.Co * g.C) ? i : HEAD_MAXCO * g.C + (i - g.Co * g.C);   // [co][c] then the bias sums
    for (int k = kg; k < blocks; k += 8) s += part[(size_t)k * per + slot];
  }
  red[kg][o] = s;
  __syncthreads();
  if (kg == 0 && i < total) {
    double t = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) t += red[j][o];
    if (i < g.Co * g.C) dW[i] = (float)t;
    else if (db) db[i - g.Co * g.C] = (float)t;
  }
}

extern "C" int mseg_head_bwd(const MsegSrc* src, int N, int HW, const float* w, int Co, const float* gout_nchw,
                             void* gy, int gy_dtype, float* dW, float* db, void* ws, void* stream) {
  if (!src || !src->ptr || !w || !gout_nchw || !gy || !dW || !ws) return MSEG_EINVAL;
  if (gy_dtype != MSEG_ST_F32 && gy_dtype != MSEG_ST_BF16) return MSEG_EINVAL;
  if (N <= 0 || HW <= 0 || Co <= 0 || Co > HEAD_MAXCO || src->C <= 0 || (src->C & 3)) return MSEG_EINVAL;
  HeadGeom g = head_geom(N, HW, src->C, Co);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(head_bwd_kernel, dim3(g.chunks, N), dim3(256), 0, st, *src, g, w, gout_nchw, gy, gy_dtype,
                     (double*)ws);
  MSEG_LAUNCH_CHECK();
  hipLaunchKernelGGL(head_bwd_reduce_kernel, dim3((Co * (src->C + 1) + 31) / 32), dim3(256), 0, st,
                     (const double*)ws, g, dW, db);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

// ---- softmax over the 3 boundary classes, NCHW logits -> HWC probabilities with the top/left pad cropped ------------
// Replaces F.softmax(prediction, dim=1)[:, :, pads[0]:, pads[1]:] + transpose to (H, W, 3) (infer.py:371-374,
// infer_script_local.py:155-157) in front of boundary_postprocessing.
__global__ void softmax3_hwc_kernel(const float* __restrict__ logits, int Hp, int Wp, int py, int px, int H, int W,
                                    float* __restrict__ probs) {
  const size_t n = (size_t)H * W;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int y = (int)(i / W), x = (int)(i - (size_t)y * W);
  const size_t src = (size_t)(y + py) * Wp + (x + px);
  const size_t plane = (size_t)Hp * Wp;
  const float l0 = logits[src], l1 = logits[plane + src], l2 = logits[2 * plane + src];
  const float m = fmaxf(l0, fmaxf(l1, l2));
  const float e0 = expf(l0 - m), e1 = expf(l1 - m), e2 = expf(l2 - m);
  const float s = e0 + e1 + e2;
  probs[3 * i] = e0 / s; probs[3 * i + 1] = e1 / s; probs[3 * i + 2] = e2 / s;
}

extern "C" int mseg_softmax3_hwc(const float* logits_chw, int Hp, int Wp, int pad_y, int pad_x, float* probs_hwc,
                                 void* stream) {
  if (!logits_chw || !probs_hwc || Hp <= 0 || Wp <= 0 || pad_y < 0 || pad_x < 0 || pad_y >= Hp || pad_x >= Wp)
    return MSEG_EINVAL;
  const int H = Hp - pad_y, W = Wp - pad_x;
  const size_t n = (size_t)H * W;
  hipLaunchKernelGGL(softmax3_hwc_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     logits_chw, Hp, Wp, pad_y, pad_x, H, W, probs_hwc);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}
