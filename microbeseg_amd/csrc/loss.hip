// loss.hip — training losses of src/training/losses.py as fused forward(+backward) HBM-bound kernels.
//   distance method: nn.SmoothL1Loss / L1Loss / MSELoss per head (losses.py:24-32), summed at train.py:480-482
//   boundary method: ce_dice = CrossEntropy + 0.5 * sum_{c=1,2} c * Dice_c (losses.py:71-97, dice_loss :40-68)
// Reductions are fp64 with a fixed two-stage order (deterministic).
#include "common.h"

#define LOSS_BLOCKS 1024

extern "C" size_t mseg_loss_workspace_bytes(size_t n) {
  (void)n;
  return (size_t)LOSS_BLOCKS * 8 * sizeof(double);
}

__device__ __forceinline__ double block_sum(double v, double* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  double t = 0.0;
  const int nw = (blockDim.x + 63) >> 6;
  for (int i = 0; i < nw; ++i) t += sh[i];
  return t;
}

__device__ __forceinline__ float reg_loss(float d, int kind) {
  const float ad = fabsf(d);
  if (kind == 0) return ad < 1.f ? 0.5f * d * d : ad - 0.5f;
  if (kind == 1) return ad;
  return d * d;
}

__device__ __forceinline__ float reg_grad(float d, int kind) {
  if (kind == 0) return fabsf(d) < 1.f ? d : (d > 0.f ? 1.f : -1.f);
  if (kind == 1) return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
  return 2.f * d;
}

__global__ __launch_bounds__(256) void reg_loss_partial_kernel(const float* __restrict__ pred,
                                                               const float* __restrict__ target, size_t n, int kind,
                                                               double* __restrict__ part) {
  __shared__ double sh[4];
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    s += (double)reg_loss(pred[i] - target[i], kind);
  s = block_sum(s, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void reg_loss_final_kernel(const double* __restrict__ part, int nparts, size_t n,
                                                             float* __restrict__ out) {
  __shared__ double sh[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < nparts; i += blockDim.x) s += part[i];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) out[0] = (float)(s / (double)n);
}

__global__ void reg_loss_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ target, size_t n,
                                    int kind, const float* __restrict__ gscale, float* __restrict__ grad) {
  const float gs = (gscale ? gscale[0] : 1.f) / (float)n;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    grad[i] = gs * reg_grad(pred[i] - target[i], kind);
}

static unsigned loss_blocks(size_t n) {
  size_t b = (n + 255) / 256;
  if (b < 1) b = 1;
  return (unsigned)(b > LOSS_BLOCKS ? LOSS_BLOCKS : b);
}

extern "C" int mseg_regression_loss(const float* pred, const float* target, size_t n, int kind, float* loss_out,
                                    void* ws, void* stream) {
  if (!pred || !target || !loss_out || !ws || n == 0 || kind < 0 || kind > 2) return MSEG_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const unsigned nb = loss_blocks(n);
  hipLaunchKernelGGL(reg_loss_partial_kernel, dim3(nb), dim3(256), 0, st, pred, target, n, kind, (double*)ws);
  MSEG_LAUNCH_CHECK();
  hipLaunchKernelGGL(reg_loss_final_kernel, dim3(1), dim3(256), 0, st, (const double*)ws, (int)nb, n, loss_out);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

extern "C" int mseg_regression_loss_bwd(const float* pred, const float* target, size_t n, int kind,
                                        const float* gscale_dev, float* grad, void* stream) {
  if (!pred || !target || !grad || n == 0 || kind < 0 || kind > 2) return MSEG_EINVAL;
  hipLaunchKernelGGL(reg_loss_bwd_kernel, dim3(loss_blocks(n) * 4), dim3(256), 0, (hipStream_t)stream, pred, target,
                     n, kind, gscale_dev, grad);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

// ---- ce_dice ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void softmax3(float l0, float l1, float l2, float& p0, float& p1, float& p2, float& lse) {
  const float m = fmaxf(l0, fmaxf(l1, l2));
  const float e0 = expf(l0 - m), e1 = expf(l1 - m), e2 = expf(l2 - m);
  const float s = e0 + e1 + e2;
  p0 = e0 / s; p1 = e1 / s; p2 = e2 / s;
  lse = m + logf(s);
}

// part[block][7] = {sum g1 p1, sum p1^2, sum g1, sum g2 p2, sum p2^2, sum g2, sum CE}
__global__ __launch_bounds__(256) void ce_dice_partial_kernel(const float* __restrict__ logits,
                                                              const int64_t* __restrict__ labels, int N, int HW,
                                                              double* __restrict__ part) {
  __shared__ double sh[4];
  double acc[7] = {0, 0, 0, 0, 0, 0, 0};
  const size_t total = (size_t)N * HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t n = i / HW, p = i - n * HW;
    const float* l = logits + n * 3 * (size_t)HW + p;
    const float l0 = l[0], l1 = l[HW], l2 = l[2 * (size_t)HW];
    float p0, p1, p2, lse;
    softmax3(l0, l1, l2, p0, p1, p2, lse);
    const int y = (int)labels[i];
    const float ly = y == 0 ? l0 : (y == 1 ? l1 : l2);
    acc[6] += (double)(lse - ly);
    acc[1] += (double)p1 * p1;
    acc[4] += (double)p2 * p2;
    if (y == 1) { acc[0] += p1; acc[2] += 1.0; }
    if (y == 2) { acc[3] += p2; acc[5] += 1.0; }
  }
  for (int k = 0; k < 7; ++k) {
    const double s = block_sum(acc[k], sh);
    if (threadIdx.x == 0) part[(size_t)blockIdx.x * 8 + k] = s;
  }
}

__global__ __launch_bounds__(256) void ce_dice_final_kernel(const double* __restrict__ part, int nparts,
                                                            double* __restrict__ sums6, double* __restrict__ ce_sum) {
  __shared__ double sh[4];
  for (int k = 0; k < 7; ++k) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) s += part[(size_t)i * 8 + k];
    s = block_sum(s, sh);
    if (threadIdx.x == 0) {
      if (k < 6) sums6[k] = s;
      else ce_sum[0] = s;
    }
  }
}

extern "C" int mseg_ce_dice_fwd(const float* logits, const int64_t* labels, int N, int HW, int with_dice,
                                double* sums6, double* ce_sum, void* ws, void* stream) {
  (void)with_dice;
  if (!logits || !labels || !sums6 || !ce_sum || !ws || N <= 0 || HW <= 0) return MSEG_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const unsigned nb = loss_blocks((size_t)N * HW);
  hipLaunchKernelGGL(ce_dice_partial_kernel, dim3(nb), dim3(256), 0, st, logits, labels, N, HW, (double*)ws);
  MSEG_LAUNCH_CHECK();
  hipLaunchKernelGGL(ce_dice_final_kernel, dim3(1), dim3(256), 0, st, (const double*)ws, (int)nb, sums6, ce_sum);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

// d/dlogit_k [ CE_mean + 0.5 * sum_c c * (1 - (2 I_c + 1)/(G_c + P_c + 1)) ]
__global__ void ce_dice_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels, int N,
                                   int HW, int with_dice, const double* __restrict__ sums6, double total_px,
                                   float dice_weight, const float* __restrict__ gscale, float* __restrict__ grad) {
  const float gs = gscale ? gscale[0] : 1.f;
  float A[3] = {0, 0, 0}, B[3] = {1, 1, 1};
  if (with_dice) {
    for (int c = 1; c <= 2; ++c) {
      const double I = sums6[(c - 1) * 3 + 0], P = sums6[(c - 1) * 3 + 1], G = sums6[(c - 1) * 3 + 2];
      A[c] = (float)(2.0 * I + 1.0);
      B[c] = (float)(G + P + 1.0);
    }
  }
  const float inv_total = (float)(1.0 / total_px);
  const size_t total = (size_t)N * HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t n = i / HW, p = i - n * HW;
    const float* l = logits + n * 3 * (size_t)HW + p;
    float pr[3], lse;
    softmax3(l[0], l[HW], l[2 * (size_t)HW], pr[0], pr[1], pr[2], lse);
    const int y = (int)labels[i];
    float g[3];
    for (int k = 0; k < 3; ++k) g[k] = (pr[k] - (y == k ? 1.f : 0.f)) * inv_total;
    if (with_dice) {
      // dDice_c/dp_c = -2 g_c / B_c + 2 A_c p_c / B_c^2 ;  dp_c/dlogit_k = p_c (delta_ck - p_k)
      for (int c = 1; c <= 2; ++c) {
        const float gc = (y == c) ? 1.f : 0.f;
        const float D = dice_weight * 0.5f * (float)c * (-2.f * gc / B[c] + 2.f * A[c] * pr[c] / (B[c] * B[c]));
        for (int k = 0; k < 3; ++k) g[k] += D * pr[c] * ((k == c ? 1.f : 0.f) - pr[k]);
      }
    }
    float* go = grad + n * 3 * (size_t)HW + p;
    go[0] = gs * g[0]; go[HW] = gs * g[1]; go[2 * (size_t)HW] = gs * g[2];
  }
}

extern "C" int mseg_ce_dice_bwd(const float* logits, const int64_t* labels, int N, int HW, int with_dice,
                                const double* sums6, double total_px, double dice_weight, const float* gscale_dev,
                                float* grad, void* stream) {
  if (!logits || !labels || !sums6 || !grad || N <= 0 || HW <= 0 || total_px <= 0.0) return MSEG_EINVAL;
  hipLaunchKernelGGL(ce_dice_bwd_kernel, dim3(loss_blocks((size_t)N * HW) * 4), dim3(256), 0, (hipStream_t)stream,
                     logits, labels, N, HW, with_dice, sums6, total_px, (float)dice_weight, gscale_dev, grad);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

// ---- Adam (amsgrad) — torch.optim.Adam(lr 8e-4, betas (0.9, 0.999), eps 1e-8, amsgrad=True): train.py:380-385 ----
// ONE launch over the flat parameter arena of training/optim.py (FusedAdam): every parameter tensor of the network is a
// view of one fp32 buffer (likewise gradients and the three state tensors), so the update is a single HBM-bound stream:
// reads p, g, m, v, vmax and writes p, m, v, vmax = 36 B / parameter (1.67 GB for the default DU-Net).
__device__ __forceinline__ void adam_amsgrad_one(float& p, float g, float& m, float& v, float& vmax, float w1, float beta2,
                                                 float w2, float eps, float step_size, float bc2_sqrt) {
  m = m + (g - m) * w1;                           // exp_avg.lerp_(grad, 1 - beta1)
  v = v * beta2 + w2 * g * g;                     // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
  vmax = fmaxf(vmax, v);                          // torch.maximum(max_exp_avg_sq, exp_avg_sq)
  const float denom = sqrtf(vmax) / bc2_sqrt + eps;
  p = p - step_size * (m / denom);                // param.addcdiv_(exp_avg, denom, value=-step_size)
}

__global__ __launch_bounds__(256) void adam_amsgrad_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                           float* __restrict__ m, float* __restrict__ v,
                                                           float* __restrict__ vmax, size_t n, float step_size,
                                                           float w1, float beta2, float w2, float eps,
                                                           float bc2_sqrt, const float* __restrict__ derived) {
  if (derived) {                                  // device-side step counter (capturable form): see adam_advance_kernel
    step_size = derived[0];
    bc2_sqrt = derived[1];
  }
  const size_t n4 = n >> 2;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const bool aligned = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v) | ((uintptr_t)vmax)) & 15) == 0;
  size_t tail = 0;
  if (aligned) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
      float4 pp = ((float4*)p)[i], mm = ((float4*)m)[i], vv = ((float4*)v)[i], vx = ((float4*)vmax)[i];
      const float4 gg = ((const float4*)g)[i];
      adam_amsgrad_one(pp.x, gg.x, mm.x, vv.x, vx.x, w1, beta2, w2, eps, step_size, bc2_sqrt);
      adam_amsgrad_one(pp.y, gg.y, mm.y, vv.y, vx.y, w1, beta2, w2, eps, step_size, bc2_sqrt);
      adam_amsgrad_one(pp.z, gg.z, mm.z, vv.z, vx.z, w1, beta2, w2, eps, step_size, bc2_sqrt);
      adam_amsgrad_one(pp.w, gg.w, mm.w, vv.w, vx.w, w1, beta2, w2, eps, step_size, bc2_sqrt);
      ((float4*)p)[i] = pp; ((float4*)m)[i] = mm; ((float4*)v)[i] = vv; ((float4*)vmax)[i] = vx;
    }
    tail = n4 << 2;
  }
  for (size_t i = tail + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float pp = p[i], mm = m[i], vv = v[i], vx = vmax[i];
    adam_amsgrad_one(pp, g[i], mm, vv, vx, w1, beta2, w2, eps, step_size, bc2_sqrt);
    p[i] = pp; m[i] = mm; v[i] = vv; vmax[i] = vx;
  }
}

extern "C" int mseg_adam_amsgrad_step(float* p, const float* g, float* m, float* v, float* vmax, size_t n, double lr,
                                      double beta1, double beta2, double eps, int step, void* stream) {
  if (!p || !g || !m || !v || !vmax || n == 0 || step < 1) return MSEG_EINVAL;
  // python-float hyper-parameters arrive as doubles: torch evaluates 1 - beta, the bias corrections and lr / bc1 in
  // double and rounds each scalar to fp32 once (1.f - 0.999f would be 4.7e-5 off the weight torch uses)
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  size_t blocks = (n / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 256u * 16u) blocks = 256u * 16u;      // 16 workgroups of 4 waves per CU, grid-stride
  hipLaunchKernelGGL(adam_amsgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, vmax,
                     n, (float)(lr / bc1), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,
                     (float)sqrt(bc2), (const float*)nullptr);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

// The same update with the step counter and the learning rate ON THE DEVICE, so that the launch carries no per-step host
// scalar and a hipGraph of the training step can be replayed (torch's `capturable=True` idea, without its per-tensor
// launches).  state = 4 doubles: [0] lr (written by the host when a scheduler changes it), [1] step count (advanced here),
// [2..3] two floats each widened to a double slot: lr / (1 - beta1^step) and sqrt(1 - beta2^step), rounded to fp32 once
// like the host form above.
__global__ void adam_advance_kernel(double* __restrict__ state, double beta1, double beta2) {
  if (threadIdx.x || blockIdx.x) return;
  const double step = state[1] + 1.0;
  state[1] = step;
  const double bc1 = 1.0 - pow(beta1, step);
  const double bc2 = 1.0 - pow(beta2, step);
  float* d = reinterpret_cast<float*>(state + 2);
  d[0] = (float)(state[0] / bc1);
  d[1] = (float)sqrt(bc2);
}

extern "C" int mseg_adam_amsgrad_step_dev(float* p, const float* g, float* m, float* v, float* vmax, size_t n,
                                          double* state, double beta1, double beta2, double eps, void* stream) {
  if (!p || !g || !m || !v || !vmax || !state || n == 0) return MSEG_EINVAL;
  if ((uintptr_t)state & 7) return MSEG_EINVAL;
  size_t blocks = (n / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 256u * 16u) blocks = 256u * 16u;
  hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, beta1, beta2);
  MSEG_LAUNCH_CHECK();
  hipLaunchKernelGGL(adam_amsgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, vmax,
                     n, 0.f, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, 1.f,
                     (const float*)(state + 2));
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

// ---- Ranger (RAdam + lookahead + gradient centralisation), one launch per parameter tensor -----------------------
// Reference: src/training/ranger2020.py:101-208.  One workgroup per centralisation row (dim 0 of a conv / convT weight;
// 1-D parameters are processed in rows of `cols` elements without centralisation).  HBM-bound: reads p, g, m, v (+slow),
// writes p, m, v (+slow) — 28..36 B/param — instead of ~10 separate elementwise launches per tensor.
__global__ __launch_bounds__(256) void ranger_step_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                          float* __restrict__ m, float* __restrict__ v,
                                                          float* __restrict__ slow, size_t n, int cols, float beta1,
                                                          float w1, float beta2, float w2, float eps, float step_lr,
                                                          int rectified, int do_gc, int lookahead, float alpha) {
  __shared__ double sh[4];
  const size_t base = (size_t)blockIdx.x * cols;
  size_t len = cols;
  if (base + len > n) len = n - base;
  float mean = 0.f;
  if (do_gc) {   // x.add_(-x.mean(dim=1.., keepdim=True))
    double s = 0.0;
    for (size_t i = threadIdx.x; i < len; i += blockDim.x) s += (double)g[base + i];
    s = block_sum(s, sh);
    mean = (float)(s / (double)len);
  }
  for (size_t i = threadIdx.x; i < len; i += blockDim.x) {
    const size_t j = base + i;
    const float gi = g[j] - mean;
    const float vi = v[j] * beta2 + w2 * (gi * gi);             // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1 - beta2)
    const float mi = m[j] * beta1 + w1 * gi;                    // exp_avg.mul_(beta1).add_(g, alpha=1 - beta1)
    v[j] = vi; m[j] = mi;
    const float upd = rectified ? mi / (sqrtf(vi) + eps) : mi;   // N_sma > threshold ? adaptive : plain momentum
    float pi = p[j] - step_lr * upd;
    if (lookahead) {                                             // every k steps: slow += alpha (p - slow); p = slow
      const float sl = slow[j] + alpha * (pi - slow[j]);
      slow[j] = sl;
      pi = sl;
    }
    p[j] = pi;
  }
}

extern "C" int mseg_ranger_step(float* p, const float* g, float* m, float* v, float* slow, size_t n, int rows,
                                double beta1, double beta2, double eps, double step_lr, int rectified, int do_gc,
                                int lookahead, double alpha, void* stream) {
  if (!p || !g || !m || !v || !slow || n == 0 || rows <= 0) return MSEG_EINVAL;
  int cols;
  unsigned blocks;
  if (do_gc) {
    if (n % (size_t)rows) return MSEG_EINVAL;
    cols = (int)(n / rows);
    blocks = (unsigned)rows;
  } else {
    cols = 4096;
    blocks = (unsigned)((n + cols - 1) / cols);
  }
  hipLaunchKernelGGL(ranger_step_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, slow, n, cols,
                     (float)beta1, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)step_lr,
                     rectified, do_gc, lookahead, (float)alpha);
  MSEG_LAUNCH_CHECK();
  return MSEG_OK;
}

// ---- the same update for MANY parameter tensors in one launch --------------------------------------------------
// A network has ~250 parameter tensors, most of them tiny (biases, norm weights): one launch each costs more than the
// 28..36 B/param of traffic.  Up to RANGER_BATCH job records travel in the kernel arguments (no device table to keep
// in step with autograd's fresh gradient tensors); a workgroup finds its job by its first block and then runs
// ranger_step_kernel's row update.
#define RANGER_BATCH 48
struct RangerBatch {
  MsegRangerJob job[RANGER_BATCH];
  unsigned first[RANGER_BATCH + 1];
  int njobs;
};

__global__ __launch_bounds__(256) void ranger_step_multi_kernel(const RangerBatch b, float beta1, float w1, float beta2,
                                                                float w2, float eps, float alpha) {
  __shared__ double sh[4];
  int lo = 0, hi = b.njobs - 1;              // uniform: scalar loads from the kernel-argument segment
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (b.first[mid] <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const MsegRangerJob& J = b.job[lo];
  const unsigned row = blockIdx.x - b.first[lo];
  const bool do_gc = J.rows > 0;
  const size_t n = J.n;
  const size_t cols = do_gc ? n / (size_t)J.rows : 4096;
  const size_t base = (size_t)row * cols;
  size_t len = cols;
  if (base + len > n) len = n - base;
  float* __restrict__ p = J.p;
  const float* __restrict__ g = J.g;
  float* __restrict__ m = J.m;
  float* __restrict__ v = J.v;
  float* __restrict__ slow = J.slow;
  const float step_lr = J.step_lr;
  const bool rectified = J.flags & 1, lookahead = J.flags & 2;
  float mean = 0.f;
  if (do_gc) {
    double s = 0.0;
    for (size_t i = threadIdx.x; i < len; i += blockDim.x) s += (double)g[base + i];
    s = block_sum(s, sh);
    mean = (float)(s / (double)len);
  }
  for (size_t i = threadIdx.x; i < len; i += blockDim.x) {
    const size_t j = base + i;
    const float gi = g[j] - mean;
    const float vi = v[j] * beta2 + w2 * (gi * gi);
    const float mi = m[j] * beta1 + w1 * gi;
    v[j] = vi; m[j] = mi;
    const float upd = rectified ? mi / (sqrtf(vi) + eps) : mi;
    float pi = p[j] - step_lr * upd;
    if (lookahead) {
      const float sl = slow[j] + alpha * (pi - slow[j]);
      slow[j] = sl;
      pi = sl;
    }
    p[j] = pi;
  }
}

extern "C" int mseg_ranger_step_multi(const MsegRangerJob* jobs, int njobs, double beta1, double beta2, double eps,
                                      double alpha, void* stream) {
  if (!jobs || njobs <= 0) return MSEG_EINVAL;
  for (int i = 0; i < njobs; ++i) {
    const MsegRangerJob& J = jobs[i];
    if (!J.p || !J.g || !J.m || !J.v || !J.slow || J.n == 0 || J.rows < 0) return MSEG_EINVAL;
    if (J.rows > 0 && J.n % (uint64_t)J.rows) return MSEG_EINVAL;
    if ((J.rows > 0 ? (uint64_t)J.rows : (J.n + 4095) / 4096) > 0x7fffffffull) return MSEG_EINVAL;
  }
  for (int at = 0; at < njobs; at += RANGER_BATCH) {
    RangerBatch b;
    b.njobs = njobs - at < RANGER_BATCH ? njobs - at : RANGER_BATCH;
    uint64_t blocks = 0;
    for (int i = 0; i < b.njobs; ++i) {
      b.job[i] = jobs[at + i];
      b.first[i] = (unsigned)blocks;
      blocks += b.job[i].rows > 0 ? (uint64_t)b.job[i].rows : (b.job[i].n + 4095) / 4096;
      if (blocks > 0x7fffffffull) return MSEG_EINVAL;
    }
    for (int i = b.njobs; i < RANGER_BATCH; ++i) { b.job[i] = b.job[0]; b.first[i] = (unsigned)blocks; }
    b.first[RANGER_BATCH] = (unsigned)blocks;
    hipLaunchKernelGGL(ranger_step_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, b,
                       (float)beta1, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)alpha);
    MSEG_LAUNCH_CHECK();
  }
  return MSEG_OK;
}
