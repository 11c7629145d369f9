// head.hip — the 1x1 output convolutions (src/utils/unets.py:347 UNet head, :460-461 DUNet heads).
// HBM-bound: one read of the last decoder tensor (normalised on load), Co <= 4 dot products per pixel.
// Output / incoming gradient are NCHW ([N][Co][HW]) because that is the layout of the Python boundary.
#include "common.h"

#define HEAD_MAXCO 4

// A group of `lpp` lanes owns a pixel: every lane loads 16 bytes of it (4 fp32 / 8 bf16 channels per trip), multiplies by the
// Co weight rows and the group reduces by shuffles.  U pixels are in flight per group (all loads issued before the first
// use): with one load per trip the pass was latency-bound at 0.6 TB/s.
// CO: compile-time bound of the output count (1 for the DU-Net heads: a quarter of the accumulators and shuffles)
template <bool S16, int CO>
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
    float acc[U][CO];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int co = 0; co < CO; ++co) acc[u][co] = 0.f;
    for (int cv = sub; cv < CV; cv += lpp) {
      uint4 raw[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long pix = base + u * ppb + threadIdx.x / lpp;
        const size_t e = (size_t)(pix < total ? pix : 0) * s.C + cv * V;
        raw[u] = S16 ? *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(s.ptr) + e)
                     : *reinterpret_cast<const uint4*>(s.ptr + e);
      }
      float4 wv[CO][V / 4];
#pragma unroll
      for (int co = 0; co < CO; ++co)
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
          for (int co = 0; co < CO; ++co)
            if (co < Co) acc[u][co] += v.x * wv[co][h].x + v.y * wv[co][h].y + v.z * wv[co][h].z + v.w * wv[co][h].w;
        }
      }
    }
    for (int o = lpp >> 1; o > 0; o >>= 1) {
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int co = 0; co < CO; ++co) acc[u][co] += __shfl_xor(acc[u][co], o, 64);
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
#define HEAD_FWD(S16_, CO_) hipLaunchKernelGGL((head_fwd_kernel<S16_, CO_>), dim3((unsigned)blocks), dim3(256), 0,      \
                                               (hipStream_t)stream, *src, N, HW, w, b, Co, lpp, out_nchw)
  if (s16) { if (Co == 1) HEAD_FWD(true, 1); else if (Co == 2) HEAD_FWD(true, 2); else HEAD_FWD(true, 4); }
  else     { if (Co == 1) HEAD_FWD(false, 1); else if (Co == 2) HEAD_FWD(false, 2); else HEAD_FWD(false, 4); }
#undef HEAD_FWD
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

// thread owns V channels (4 of an fp32 tensor, 8 of a bf16 tensor: 16-byte loads either way) and walks the pixels of its
// chunk: gy = sum_co g[co] * W[co][c ..]; dW partial sums in fp64.  The norm-on-load tables of the thread's channels are
// read once (a workgroup stays inside one sample).
// CO: compile-time bound of the output count (1 for the DU-Net heads: a quarter of the fp64 sums); U rows in flight.
template <int CO, int U, bool S16>
__global__ __launch_bounds__(256) void head_bwd_kernel(const MsegSrc s, HeadGeom g, const float* __restrict__ w,
                                                       const float* __restrict__ gout, void* __restrict__ gy,
                                                       int gy_dtype, double* __restrict__ part) {
  constexpr int V = S16 ? 8 : 4;
  __shared__ double red[256 * V * CO];
  __shared__ double redb[256];
  const int tid = threadIdx.x;
  const int chunk = blockIdx.x, n = blockIdx.y;
  const int CVn = g.C / V;
  const int CW = CVn < 256 ? CVn : 256;
  const int rpi = 256 / CW;
  const int cq = tid % CW, r0 = tid / CW;
  const int row_begin = chunk * g.rows_per_chunk;
  int row_end = row_begin + g.rows_per_chunk;
  if (row_end > g.HW) row_end = g.HW;
  double* pout = part + ((size_t)n * g.chunks + chunk) * (HEAD_MAXCO * (g.C + 1));

  for (int cbase = 0; cbase < CVn; cbase += CW) {
    const int cv = cbase + cq;
    const bool active = (cv < CVn) && (r0 < rpi);
    double dw[CO][V];
    double dbs[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) {
      dbs[co] = 0.0;
#pragma unroll
      for (int j = 0; j < V; ++j) dw[co][j] = 0.0;
    }
    if (active) {
      const int c = cv * V;
      float wv[CO][V], sc[V], sh[V];
#pragma unroll
      for (int co = 0; co < CO; ++co)
#pragma unroll
        for (int j = 0; j < V; ++j) wv[co][j] = co < g.Co ? w[(size_t)co * g.C + c + j] : 0.f;
#pragma unroll
      for (int j = 0; j < V; ++j) {
        sc[j] = s.scale ? s.scale[(size_t)n * s.ss + c + j] : 1.f;
        sh[j] = s.scale ? s.shift[(size_t)n * s.ss + c + j] : 0.f;
      }
      for (int r = row_begin + r0; r < row_end; r += rpi * U) {
        uint4 yraw[U];
        float gv[U][CO];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int ru = r + u * rpi;
          ok[u] = ru < row_end;
          const int rr = ok[u] ? ru : r;
          const size_t e = ((size_t)n * g.HW + rr) * g.C + c;
          yraw[u] = S16 ? *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(s.ptr) + e)
                        : *reinterpret_cast<const uint4*>(s.ptr + e);
#pragma unroll
          for (int co = 0; co < CO; ++co)
            gv[u][co] = co < g.Co ? gout[((size_t)n * g.Co + co) * g.HW + rr] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (!ok[u]) continue;
          const size_t off = ((size_t)n * g.HW + r + u * rpi) * g.C + c;
          float yv[V], o[V];
          if (S16) {
            yv[0] = bf16_lo(yraw[u].x); yv[1] = bf16_hi(yraw[u].x); yv[2] = bf16_lo(yraw[u].y); yv[3] = bf16_hi(yraw[u].y);
            yv[V - 4] = bf16_lo(yraw[u].z); yv[V - 3] = bf16_hi(yraw[u].z); yv[V - 2] = bf16_lo(yraw[u].w); yv[V - 1] = bf16_hi(yraw[u].w);
          } else {
            yv[0] = __uint_as_float(yraw[u].x); yv[1] = __uint_as_float(yraw[u].y);
            yv[2] = __uint_as_float(yraw[u].z); yv[3] = __uint_as_float(yraw[u].w);
          }
#pragma unroll
          for (int j = 0; j < V; j += 4) {                 // the same arithmetic as src_transform4, tables from registers
            const float4 a = act_fwd4(make_float4(yv[j], yv[j + 1], yv[j + 2], yv[j + 3]), s.act);
            yv[j] = a.x; yv[j + 1] = a.y; yv[j + 2] = a.z; yv[j + 3] = a.w;
          }
          if (s.scale) {
#pragma unroll
            for (int j = 0; j < V; ++j) yv[j] = yv[j] * sc[j] + sh[j];
          }
#pragma unroll
          for (int j = 0; j < V; ++j) o[j] = 0.f;
#pragma unroll
          for (int co = 0; co < CO; ++co) {
            if (co < g.Co) {
              const float gvc = gv[u][co];
#pragma unroll
              for (int j = 0; j < V; ++j) {
                o[j] += gvc * wv[co][j];
                dw[co][j] += (double)gvc * yv[j];
              }
              if (cv == 0) dbs[co] += gvc;
            }
          }
          if (V == 8 && gy_dtype == MSEG_ST_BF16) {
            *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(gy) + off) =
                make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[V - 4], o[V - 3]),
                           pack_bf16x2(o[V - 2], o[V - 1]));
          } else {
#pragma unroll
            for (int j = 0; j < V; j += 4) st_f4_rt(gy, off + j, make_float4(o[j], o[j + 1], o[j + 2], o[j + 3]), gy_dtype);
          }
        }
      }
    }
#pragma unroll
    for (int co = 0; co < CO; ++co) {                   // (compile-time trip count: dw[][] must stay in registers)
      if (co < g.Co) {
#pragma unroll
        for (int j = 0; j < V; ++j) red[(tid * CO + co) * V + j] = dw[co][j];
      }
    }
    __syncthreads();
    if (r0 == 0 && cv < CVn) {
      for (int co = 0; co < g.Co && co < CO; ++co) {
        double t[V];
#pragma unroll
        for (int j = 0; j < V; ++j) t[j] = 0.0;
        for (int k = 0; k < rpi; ++k)
#pragma unroll
          for (int j = 0; j < V; ++j) t[j] += red[((k * CW + cq) * CO + co) * V + j];
#pragma unroll
        for (int j = 0; j < V; ++j) pout[(size_t)co * g.C + cv * V + j] = t[j];
      }
    }
    __syncthreads();
    if (cbase == 0) {
      // bias gradient: threads with cv == 0 (cq == 0) hold the per-row-slice sums
#pragma unroll
      for (int co = 0; co < CO; ++co) {
        if (co >= g.Co) break;                          // uniform
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

// sums the per-workgroup partials in fixed order: 8 outputs x 32 partial groups per workgroup, combined through LDS
__global__ __launch_bounds__(256) void head_bwd_reduce_kernel(const double* __restrict__ part, HeadGeom g,
                                                              float* __restrict__ dW, float* __restrict__ db) {
  __shared__ double red[32][8];
  const int per = HEAD_MAXCO * (g.C + 1);
  const int blocks = g.N * g.chunks;
  const int total = g.Co * g.C + g.Co;
  const int o = threadIdx.x & 7, kg = threadIdx.x >> 3;
  const int i = blockIdx.x * 8 + o;
  double s = 0.0;
  if (i < total) {
    const int slot = (i < g.Co * g.C) ? i : HEAD_MAXCO * g.C + (i - g.Co * g.C);   // [co][c] then the bias sums
    for (int k = kg; k < blocks; k += 32) s += part[(size_t)k * per + slot];
  }
  red[kg][o] = s;
  __syncthreads();
  if (kg == 0 && i < total) {
    double t = 0.0;
#pragma unroll
    for (int j = 0; j < 32; ++j) t += red[j][o];
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
#define HEAD_BWD(CO_, U_, S16_) hipLaunchKernelGGL((head_bwd_kernel<CO_, U_, S16_>), dim3(g.chunks, N), dim3(256), 0, st, *src, g, \
                                                   w, gout_nchw, gy, gy_dtype, (double*)ws)
  if (src->dtype == MSEG_ST_BF16) {
    if (src->C & 7) return MSEG_EINVAL;
    if (Co == 1) HEAD_BWD(1, 2, true); else if (Co == 2) HEAD_BWD(2, 2, true); else HEAD_BWD(4, 1, true);
  } else {
    if (Co == 1) HEAD_BWD(1, 4, false); else if (Co == 2) HEAD_BWD(2, 2, false); else HEAD_BWD(4, 1, false);
  }
#undef HEAD_BWD
  MSEG_LAUNCH_CHECK();
  hipLaunchKernelGGL(head_bwd_reduce_kernel, dim3((Co * (src->C + 1) + 7) / 8), dim3(256), 0, st,
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
