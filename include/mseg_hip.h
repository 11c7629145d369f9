/*
 * mseg_hip.h — C ABI of libmseg_hip.so, the gfx950 (MI355X / CDNA4) kernel library under the
 * microbeSEG U-Net train / infer / watershed hot path.
 *
 * The reference (hip-satomi/microbeSEG) is pure Python on stock torch.nn / scipy / scikit-image and defines
 * no FFI of its own.  Each entry point below therefore cites the reference *call site* whose
 * library op it replaces.  All pointers are DEVICE pointers unless a name ends in `_host`; the library never
 * allocates user-visible memory, never synchronises the stream on the compute path, and returns 0 on success
 * or a negative MSEG_E* code (never throws across the boundary).  `stream` is a hipStream_t passed as void*.
 *
 * Activation tensors are NHWC fp32 ("pixel-major"): element (n, y, x, c) at ((n*H + y)*W + x)*C + c.
 * A network input/output with one channel is bit-identical in NCHW and NHWC, which is how the Python boundary
 * (NCHW, src/utils/unets.py:349,463) maps onto this layout without a copy.
 */
#ifndef MSEG_HIP_H
#define MSEG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSEG_OK 0
#define MSEG_EINVAL (-1)   /* bad argument / unsupported shape */
#define MSEG_ELAUNCH (-2)  /* hipLaunch / runtime error (mseg_last_hip_error() has the hipError_t) */
#define MSEG_EWORKSPACE (-3)

/* activation ids — reference: src/utils/unets.py:115-124 (relu / leakyrelu / elu / mish), Mish at :81-89 */
#define MSEG_ACT_NONE 0
#define MSEG_ACT_RELU 1
#define MSEG_ACT_LEAKY 2 /* negative_slope 0.01 (torch default) */
#define MSEG_ACT_ELU 3   /* alpha 1.0 */
#define MSEG_ACT_MISH 4  /* x * tanh(softplus(x)), softplus threshold 20 */

/* normalisation ids — reference: src/utils/unets.py:127-134 */
#define MSEG_NORM_BN 0 /* BatchNorm2d, stats over (N,H,W) */
#define MSEG_NORM_GN 1 /* GroupNorm(8, C), stats over (C/8,H,W) per sample */
#define MSEG_NORM_IN 2 /* InstanceNorm2d, stats over (H,W) per (sample, channel), no affine */

/* A tensor operand that is *normalised on load*: value(n,p,c) = act(ptr[...]) * scale[n*ss + c] + shift[n*ss + c].
 * scale == NULL means "no affine" (identity), act == MSEG_ACT_NONE means no activation.
 * This is how conv -> activation -> norm (src/utils/unets.py:163-173: ConvBlock.forward) is executed without ever
 * materialising the normalised tensor: the producing conv stores z = conv(x)+b, the consumer applies act+norm. */
#define MSEG_ST_F32 0
#define MSEG_ST_BF16 1 /* tensor stored as bfloat16 in HBM (BASELINE configs[2]); all arithmetic stays fp32 */
typedef struct MsegSrc {
  const float* ptr;   /* [N][H][W][C]; holds bfloat16 elements when dtype == MSEG_ST_BF16 */
  const float* scale; /* [N][C] (ss == C) or [C] (ss == 0), or NULL; always fp32 */
  const float* shift;
  int32_t C;
  int32_t act;
  int32_t ss;
  int32_t dtype;      /* MSEG_ST_F32 / MSEG_ST_BF16 */
} MsegSrc;

/* ---- implicit-GEMM convolution family (fp32 MFMA v_mfma_f32_32x32x2_f32) ---------------------------------
 * Replaces nn.Conv2d 3x3 s1/s2 (unets.py:112,137,192), nn.ConvTranspose2d 2x2 s2 (unets.py:244) and the
 * data-gradients of both (autograd of the same modules, train.py:488).
 *   out[m][n] = bias[n] + sum_t sum_c  A_t[m][c] * w[t][n][c]
 * m indexes the "M-space" pixels (NB, Ho, Wo); A_t[m] is the source pixel selected by tap t:
 *   mode CONV : (iy, ix) = (oy*stride + ky - pad, ox*stride + kx - pad)
 *   mode TCONV: ty = oy + pad - ky must be divisible by stride, iy = ty / stride   (transposed conv = dgrad)
 * the source is the channel-concatenation of src[0..nsrc) (torch.cat([up, skip], 1): unets.py:373,492,502).
 * Epilogue PLAIN: dst = (n < split ? dst0[m*ld0 + n] : dst1[m*ld1 + n - split]) (+= if accN)   (split >= Ngemm: single dst)
 * Epilogue SCATTER2X2 (ConvTranspose2d as a 1x1 GEMM with N = 4*Cq): n = (a*2+b)*Cq + co,
 *   dst0[((img*2Ho + 2oy+a)*2Wo + 2ox+b)*Cq + co], bias index co.                                          */
#define MSEG_MODE_CONV 0
#define MSEG_MODE_TCONV 1
#define MSEG_EPI_PLAIN 0
#define MSEG_EPI_SCATTER2X2 1
#define MSEG_MORDER_LINEAR 0
#define MSEG_MORDER_PARITY 1 /* M ordered by (oy&1, ox&1) class first: lets stride-2 TCONV tiles skip dead taps */
#define MSEG_PREC_F32 0
#define MSEG_PREC_BF16 1

typedef struct MsegIgemm {
  MsegSrc src[2];
  const float* w;    /* packed [T][Npad][Kpad], Npad % 128 == 0, Kpad % 32 == 0, zero filled beyond Ngemm / Cin */
  const float* bias; /* [Ngemm] (PLAIN) / [Cq] (SCATTER2X2) or NULL */
  float* dst0;
  float* dst1;
  int32_t nsrc, Cin, Kpad, Npad;
  int32_t NB, Hi, Wi, Ho, Wo;
  int32_t KH, KW, stride, pad, mode, morder;
  int32_t Ngemm, epi, split, ld0, ld1, acc0, acc1, Cq;
  int32_t precision; /* MSEG_PREC_F32, or MSEG_PREC_BF16: bf16 matrix-core inputs, fp32 accumulate (BASELINE configs[2]) —
                      * 3x3 stride-1 launches only (MSEG_EINVAL otherwise); `w` then points at the bf16 copy of the
                      * packed weights (mseg_f32_to_bf16); bias stays fp32 */
  int32_t dst_dtype; /* MSEG_ST_F32 / MSEG_ST_BF16: element type of dst0 / dst1.  bf16 sources (MsegSrc.dtype) and bf16
                      * destinations need precision == MSEG_PREC_BF16 (MSEG_EINVAL otherwise) */
  /* optional split-K scratch (small batches: fewer 3x3 stride-1 tiles than workgroup slots).  ws == NULL or too small: the
   * launch simply is not split.  Size: mseg_igemm_workspace_bytes(). */
  void* ws;
  size_t ws_bytes;
  /* optional: statistics of the output for the normalisation that follows it, taken in the epilogue instead of by a pass
   * over the stored tensor (unets.py:127-134: conv -> act -> norm).  stats != NULL asks for per-tile partial sums
   * stats[row][0 / 1][Ngemm] = sum / sum of squares of act(z) over the pixels of partial row `row`, z as stored (i.e. of the
   * bf16-rounded value for a bf16 destination), act = stats_act (MSEG_ACT_NONE or MSEG_ACT_RELU).  Only some kernels can:
   * mseg_igemm_query() on the same descriptor reports in MsegKernelInfo.stats_rows how many rows the call will write
   * (0: none — the buffer is left untouched and the caller runs mseg_norm_stats); size the buffer for that.  Plain epilogue,
   * one destination, no accumulation.  Reduced by mseg_norm_stats_from_conv().                                           */
  float* stats;
  int32_t stats_act;
  int32_t reserved0;
} MsegIgemm;

int mseg_igemm(const MsegIgemm* p, void* stream);

/* Which kernel a call maps to.  The query runs the SAME dispatch code as the call it describes and launches nothing
 * (no device work, no stream): same argument checks, same return code (MSEG_EINVAL: no kernel for this descriptor, e.g.
 * MSEG_PREC_BF16 on a shape that has no bf16 kernel), so host code never has to restate a dispatch rule.
 * `name` is the instantiation as rocprofv3's kernel trace prints it, without "void " and the argument list. */
typedef struct MsegKernelInfo {
  char name[120];
  int32_t precision; /* MSEG_PREC_* the kernel computes in */
  int32_t launches;  /* kernel launches of the call (split-K partial + reduction = 2; one-off table initialisation counted) */
  uint32_t grid, block;
  size_t workspace;  /* split-K scratch bytes the call uses (0: none) */
  int32_t stats_rows; /* partial rows the call writes to MsegIgemm.stats (0: the kernel takes no statistics) */
  int32_t reserved0;
} MsegKernelInfo;
int mseg_igemm_query(const MsegIgemm* p, MsegKernelInfo* info);
/* name of the main kernel of this thread's last mseg_igemm / mseg_wgrad call ("" before the first) */
const char* mseg_last_kernel(void);
/* round-to-nearest-even conversion of n floats to bf16 (the packed weights of a MSEG_PREC_BF16 launch) */
int mseg_f32_to_bf16(const float* src, uint16_t* dst, size_t n, void* stream);
/* bytes of split-K scratch this launch would use (0: it would not be split) */
size_t mseg_igemm_workspace_bytes(const MsegIgemm* p);
/* Test / ablation hook: on = 0 sends the 64 -> 64 channel bf16-storage layers (level 0 of the U-Nets) back to the
 * tile-per-workgroup kernel instead of the persistent one that keeps the layer's weights in LDS.  Same results either
 * way (the tests compare both).  Process-wide; default 1. */
int mseg_igemm_set_persistent(int on);
/* Test / ablation hook: on = 0 sends the 128-channel-tile bf16 layers back from 256-pixel to 128-pixel tiles
 * (igemm_halo_bf16w4m_kernel -> igemm_halo_bf16w4_kernel).  Same results up to the fp32 accumulation order.  Default 1. */
int mseg_igemm_set_wide_tiles(int on);
/* Test / ablation hook for the one-workgroup-per-CU kernel with DMA-streamed weights (igemm_p8.hip: 3x3 stride-1 layers
 * with >= 128 output channels on bf16 tensors): 0 = off (the tile-per-workgroup kernels above), 1 = on (default),
 * 2 = on, 128-channel layers on 256-pixel instead of 512-pixel tiles.  Same results up to the fp32 accumulation order. */
int mseg_igemm_set_p8(int mode);

/* weight gradient: G[t][mch][nch] = sum_p P[p][mch] * Q[gather(p, t)][nch], written to dst[(mch*Nch + nch)*T + t]
 * which *is* torch's layout for both Conv2d.weight (Cout,Cin,KH,KW) [P = dz, Q = conv input] and
 * ConvTranspose2d.weight (Cin,Cout,2,2) [P = convT input, Q = d(convT output), KH=KW=2, stride 2, pad 0].
 * gather: (qy, qx) = (py*stride + ky - pad, px*stride + kx - pad).  Split-K over pixels into `ws`, then a
 * fixed-order reduction (deterministic).  Nch_store <= Nch lets a zero-padded Q (first layer, Cin=1->4) drop pads. */
typedef struct MsegWgrad {
  MsegSrc P;
  MsegSrc Q[2];
  float* ws;  /* >= mseg_wgrad_workspace_bytes() */
  float* dst; /* [Mch][Nch_store][T] */
  int32_t nq, Nch, Nch_store;
  int32_t NB, Hp, Wp, Hq, Wq;
  int32_t KH, KW, stride, pad;
  int32_t splits; /* 0 = choose */
  int32_t phase;  /* 0 = partial + reduce, 1 = split-K partial kernel only, 2 = reduction only (profiling) */
  int32_t precision; /* MSEG_PREC_F32, or MSEG_PREC_BF16: P and Q rounded to bf16 for the matrix cores, fp32 accumulate —
                      * 3x3 stride-1 weight gradients with a plain P only (MSEG_EINVAL otherwise) */
  int32_t reserved;
} MsegWgrad;

size_t mseg_wgrad_workspace_bytes(const MsegWgrad* p);
int mseg_wgrad(const MsegWgrad* p, void* stream);
int mseg_wgrad_query(const MsegWgrad* p, MsegKernelInfo* info);

/* The network's first convolution, Conv2d(ch_in <= 4, Cout, 3, padding=1) (unets.py:303-304,413-414), and its weight
 * gradient for ch_in == 1: HBM-bound VALU kernels (9..36 multiply-adds per output), not worth a 32-channel matrix-core
 * K-step.  x4: raw network input, NHWC with 4 channels (zero padded); w / dW: torch layout (Cout, Cin, 3, 3);
 * z, dz: [N][H][W][Cout].  Cout % 4 == 0, Cout <= 256, 256 % (Cout / 4) == 0 — otherwise MSEG_EINVAL (use mseg_igemm). */
/* z_dtype / dz_dtype: MSEG_ST_F32 or MSEG_ST_BF16 — how z is stored / dz is read (the network input x4 is always fp32) */
int mseg_first_conv_fwd(const float* x4, const float* w, const float* bias, int N, int H, int W, int Cin, int Cout,
                        void* z, int z_dtype, void* stream);
/* K14 on the device — the frame normalisation of the inference loop (reference: infer.py:346-348,
 * infer_script_local.py:130-132: min / max of the frame, top / left padding with min (utils.py:124-163),
 * 2 * (f32(x) - min) / (max - min) - 1) without the host touching a pixel:
 *   mseg_frame_minmax        raw: [npix] uint8 (MSEG_PIX_U8) / uint16 (MSEG_PIX_U16) -> minmax[2] on the device ({~min, max})
 *   mseg_first_conv_fwd_raw  the first convolution Conv2d(1, Cout, 3, padding=1) of ONE frame, reading the raw frame and
 *                            normalising / padding while it loads: the same fp32 operations in the same order as the host
 *                            formula (bit-identical input values); z: [H0 + pad_top][W0 + pad_left][Cout]
 *   mseg_frame_normalize     the normalised, padded frame itself (fp32 [H0 + pad_top][W0 + pad_left]) for networks whose
 *                            first layer does not take the kernel above                                                   */
#define MSEG_PIX_U8 0
#define MSEG_PIX_U16 1
int mseg_frame_minmax(const void* raw, int dtype, size_t npix, uint32_t* minmax, void* stream);
int mseg_first_conv_fwd_raw(const void* raw, int dtype, int H0, int W0, int pad_top, int pad_left, const uint32_t* minmax,
                            const float* w, const float* bias, int Cout, void* z, int z_dtype, void* stream);
int mseg_frame_normalize(const void* raw, int dtype, int H0, int W0, int pad_top, int pad_left, const uint32_t* minmax,
                         float* out, void* stream);
size_t mseg_first_wgrad_workspace_bytes(int N, int H, int W, int Cout);
int mseg_first_wgrad(const float* x4, const void* dz, int dz_dtype, int N, int H, int W, int Cout, float* dW, void* ws,
                     void* stream);

/* strided repack of a weight tensor into the zero-padded [T][Rpad][Cpad] GEMM operand:
 *   dst[(t*Rpad + r)*Cpad + c] = (r < R && c < C) ? src[t*st + r*sr + c*sc] : 0                             */
int mseg_pack_weight(const float* src, float* dst, int T, int R, int Rpad, int C, int Cpad, int st, int sr, int sc,
                     void* stream);

/* Every repack of a network in one launch, after an optimizer step (replaces the per-layer repacks of the conv
 * weights the forward / data-gradient kernels consume: unets.py:112,137,192,244 hold them in torch's layout).
 * `jobs_dev`: njobs records in DEVICE memory, sorted by first_block; job j owns blocks [first_block[j],
 * first_block[j] + mseg_pack_job_blocks(T, Rpad, Cpad)), total_blocks in all.  dst / dst16 (bf16 operand) may be NULL. */
typedef struct MsegPackJob {
  const float* src;
  float* dst;
  uint16_t* dst16;
  int32_t T, R, Rpad, C, Cpad, st, sr, sc;
  uint32_t first_block;
  uint32_t reserved;
} MsegPackJob;
unsigned mseg_pack_job_blocks(int T, int Rpad, int Cpad);
int mseg_pack_weights_multi(const MsegPackJob* jobs_dev, int njobs, unsigned total_blocks, void* stream);

/* ---- normalisation (BatchNorm2d / GroupNorm(8) / InstanceNorm2d applied AFTER the activation) --------------
 * forward statistics of a = act(z) over an NHWC tensor, fp64 accumulation:
 *   mseg_norm_stats: fills scale/shift ([N][C], ss = C for GN/IN; [C], ss = 0 for BN) such that
 *   y = a*scale + shift, and mean/rstd ([C] BN, [N][8] GN, [N][C] IN) for the backward.
 *   BN training additionally updates running_mean/var with `momentum` (unbiased var), torch semantics
 *   (unets.py:127-128; torch BatchNorm2d defaults eps 1e-5, momentum 0.1).
 * ws: scratch of mseg_norm_workspace_bytes(N, HW, C) bytes.  Its first 64 KiB hold the arrival counters of the in-kernel
 * reductions (the last workgroups of a pass finish the sums and write the tables: no separate reduction launches):
 * ZERO them once after allocating ws (hipMemset); every call leaves them zero.  One ws per concurrent stream.          */
size_t mseg_norm_workspace_bytes(int N, int HW, int C);
/* 1 = the last workgroups of a pass finish the reductions inside the pass kernel, 0 = a pass is followed by separate
 * reduction / finalize launches (same results, bit for bit).  Default 0 (measured faster at batch 32); MSEG_NORM_TAILS in
 * the environment sets it for a process. */
int mseg_norm_set_tails(int on);
/* How a BatchNorm pass's partial sums (and a bias gradient's) become the per-channel results.  2 (default): ONE launch
 * without a hand-off between workgroups — a workgroup owns four channels and does the reduction over the chunks, the sum
 * over the samples and the finalize arithmetic itself, in the order of the launches it replaces (any channel count);
 * 1: the reduction launch whose last workgroup to arrive also finalizes (<= 256 channels, else two launches); 0: always
 * the separate reduction and finalize / column-sum launches; -1: back to the default.  Same bits in every mode.
 * MSEG_NORM_FINISH in the environment sets it for a process. */
int mseg_norm_set_finish(int on);
/* st: MSEG_ST_F32 / MSEG_ST_BF16 = storage of the activation tensors of the call (z, act_out; gy, dz, act_in below).  With
 * bf16 storage a thread owns 8 channels (C % 8 == 0) and the statistics are those of the values as stored.           */
int mseg_norm_stats(const void* z, int N, int HW, int C, int st, int act, int norm, const float* gamma,
                    const float* beta, float eps, float* scale, float* shift, float* mean, float* rstd,
                    float* running_mean, float* running_var, float momentum, void* act_out, void* ws, void* stream);
/* BatchNorm statistics from the partial sums a convolution's epilogue left (MsegIgemm.stats: part[rows][2][C], C = the
 * launch's Ngemm): the same tables, running statistics and saved mean / rstd as mseg_norm_stats(norm = MSEG_NORM_BN) on the
 * stored tensor, without reading it.  count = N * H * W values per channel.  Fixed summation order (rows in 16 slices, then
 * the slices).  ws: as for mseg_norm_stats.                                                                             */
int mseg_norm_stats_from_conv(const float* part, int rows, int C, long long count, const float* gamma, const float* beta,
                              float eps, float* scale, float* shift, float* mean, float* rstd, float* running_mean,
                              float* running_var, float momentum, void* ws, void* stream);
/* act_out (nullable): also store a = act(z).  Used for the expensive activations (mish / elu / leakyrelu): consumers then
 * read `a` with MSEG_ACT_NONE instead of re-evaluating the activation for each of the 9 taps in their K-loops.
 * mseg_activation: a = act(z) alone (eval-mode BatchNorm has no statistics pass).                              */
int mseg_activation(const void* z, int N, int HW, int C, int st, int act, void* act_out, void* stream);
/* eval-mode BatchNorm: scale = gamma / sqrt(running_var + eps), shift = beta - running_mean*scale */
int mseg_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                        const float* running_var, float eps, int C, float* scale, float* shift, void* stream);
/* backward through y = norm(act(z)): given gy = dL/dy writes dz = dL/dz (may alias gy), dgamma, dbeta
 * (NULL for IN) and optionally dbias_prev[c] = sum_p dz (the producing conv's bias gradient).               */
int mseg_norm_bwd(const void* gy, const void* z, int N, int HW, int C, int st, int act, int norm, const float* gamma,
                  const float* mean, const float* rstd, void* dz, float* dgamma, float* dbeta, float* dbias,
                  const void* act_in /* nullable: a = act(z) stored by the forward */, void* ws, void* stream);

/* MaxPool2d(2, 2) of a norm-on-load operand (pool_method = 'max': unets.py:306-307,363-364).  Forward writes the plain
 * pooled tensor [N][H/2][W/2][C]; backward routes gout to the first maximum of each window (torch's rule) and writes
 * (or accumulates into) gin = dL/d(normalised operand), [N][H][W][C].                                          */
int mseg_maxpool2x2_fwd(const MsegSrc* src, int N, int H, int W, float* out, void* stream);
int mseg_maxpool2x2_bwd(const MsegSrc* src, int N, int H, int W, const float* gout, float* gin, int accumulate,
                        void* stream);

/* ---- 1x1 output heads (unets.py:347,460-461) ------------------------------------------------------------
 * out (NCHW, [N][Co][HW]) = W[Co][C] . norm-on-load(src) + b ; Co <= 4.                                      */
int mseg_head_fwd(const MsegSrc* src, int N, int HW, const float* w, const float* b, int Co, float* out_nchw,
                  void* stream);
/* gy[N][HW][C] = sum_co gout[n][co][p] * W[co][c];  dW[Co][C], db[Co] (fp64 accumulation, deterministic) */
size_t mseg_head_bwd_workspace_bytes(int N, int HW, int C, int Co);
int mseg_head_bwd(const MsegSrc* src, int N, int HW, const float* w, int Co, const float* gout_nchw, void* gy,
                  int gy_dtype /* MSEG_ST_*: storage of the gradient written to gy */, float* dW, float* db, void* ws,
                  void* stream);

/* softmax over the 3 boundary classes of logits [3][Hp][Wp] -> probabilities [Hp-pad_y][Wp-pad_x][3] (HWC), top/left
 * padding cropped: F.softmax(dim=1) + slice + transpose in front of boundary_postprocessing (infer.py:371-374). */
int mseg_softmax3_hwc(const float* logits_chw, int Hp, int Wp, int pad_y, int pad_x, float* probs_hwc, void* stream);

/* ---- losses (src/training/losses.py) -----------------------------------------------------------------------
 * smooth-L1 (beta 1, mean) / L1 / MSE of one head, forward value and gradient in one pass (losses.py:24-32).
 * kind: 0 smooth_l1, 1 l1, 2 l2.  loss_out[0] = mean loss; grad = d(loss)/d(pred) * gscale_dev[0] (or 1).    */
size_t mseg_loss_workspace_bytes(size_t n);
int mseg_regression_loss(const float* pred, const float* target, size_t n, int kind, float* loss_out,
                         void* ws, void* stream);
int mseg_regression_loss_bwd(const float* pred, const float* target, size_t n, int kind, const float* gscale_dev,
                             float* grad, void* stream);
/* ce_dice (losses.py:71-97): CE(logits, y) + 0.5 * sum_{c=1,2} c * (1 - (2 sum g p + 1)/(sum g^2 + sum p^2 + 1)),
 * logits NCHW [N][3][HW], labels int64 [N][HW].  sums[6] = {sum g1 p1, sum p1^2, sum g1, sum g2 p2, sum p2^2, sum g2}
 * are exposed so a data-parallel caller can all-reduce them between the two calls.        */
int mseg_ce_dice_fwd(const float* logits, const int64_t* labels, int N, int HW, int with_dice, double* sums6,
                     double* ce_sum, void* ws, void* stream);
/* dice_weight: 1 for single-process training; the world size under data parallelism (the Dice term is a function
 * of GLOBAL sums, so its per-rank gradient must survive the 1/world averaging of the gradient all-reduce).   */
int mseg_ce_dice_bwd(const float* logits, const int64_t* labels, int N, int HW, int with_dice, const double* sums6,
                     double total_px, double dice_weight, const float* gscale_dev, float* grad, void* stream);

/* ---- instance masks -> polygon ROIs --------------------------------------------------------
 * Replaces get_indices_pandas + the per-instance cv2_countour loop (src/utils/hull_polygon.py:8-89, called from
 * src/inference/infer.py:274-287) for a whole uint16 label image [H][W] (0 = background) on the device.
 * mseg_polygons_find: start candidates (pixels whose W / NW / N / NE neighbours carry another label) into cand_px, and
 *   per candidate its label (cand_id) and the length of the OUTER border that starts there (cand_len; 0 when the
 *   candidate is not the raster-first pixel of an outer border).  *n_cand_dev = candidates found; if it exceeds
 *   `capacity` the surplus was dropped and the caller repeats the call with larger arrays.
 * mseg_polygons_trace: polygon k starts at pixel start_px[k] and owns points offsets[k] .. offsets[k+1]-1 of points_yx
 *   ((row, col) int32 pairs, in the order cv2.findContours(..., CHAIN_APPROX_NONE) lists an outer border: from the
 *   top-left-most pixel DOWN the left side, every visited pixel, 8-connected foreground).                              */
int mseg_polygons_find(const uint16_t* labels, int H, int W, int32_t* cand_px, int32_t* cand_id, int32_t* cand_len,
                       int capacity, int32_t* n_cand_dev, void* stream);
int mseg_polygons_trace(const uint16_t* labels, int H, int W, const int32_t* start_px, const int64_t* offsets, int n_poly,
                        int32_t* points_yx, void* stream);

/* ---- fused optimizers (train.py:379-428, ranger2020.py:101-208) --------------------------------------------- */
/* torch.optim.Adam(amsgrad=True, weight_decay=0) on ONE flat range (train.py:380-385): p, g, exp_avg, exp_avg_sq,
 * max_exp_avg_sq of n elements; `step` = 1-based step count.  Hyper-parameters are doubles (python floats): 1 - beta,
 * the bias corrections and lr / bc1 are evaluated in fp64 and rounded to fp32 once, as torch does.                   */
int mseg_adam_amsgrad_step(float* p, const float* g, float* m, float* v, float* vmax, size_t n, double lr,
                           double beta1, double beta2, double eps, int step, void* stream);
/* The same update with learning rate and step count in DEVICE memory (no per-step host scalar: the launch can be recorded
 * in a hipGraph and replayed).  `state` = 4 doubles on the device: [0] learning rate (the host rewrites it when a scheduler
 * changes it), [1] steps done so far (this call adds 1 before it updates), [2..3] scratch of the call.  Two launches: a
 * one-thread kernel that advances the counter and derives lr / (1 - beta1^step), sqrt(1 - beta2^step) in fp64, and the
 * update.                                                                                                               */
int mseg_adam_amsgrad_step_dev(float* p, const float* g, float* m, float* v, float* vmax, size_t n, double* state,
                               double beta1, double beta2, double eps, void* stream);
/* One Ranger update of one parameter tensor (ranger2020.py:142-206): gradient centralisation over dims 1.. (do_gc, rows =
 * shape[0]), moments, rectified or plain-momentum update with step_lr = step_size * lr (host-side RAdam buffer,
 * ranger2020.py:160-176), and the lookahead blend every k-th step (lookahead = 1).                              */
int mseg_ranger_step(float* p, const float* g, float* m, float* v, float* slow, size_t n, int rows, double beta1,
                     double beta2, double eps, double step_lr, int rectified, int do_gc, int lookahead, double alpha,
                     void* stream);
/* The same update for many parameter tensors (ranger2020.py:117-206, the loop over group['params']): `jobs` are njobs
 * records in HOST memory; they travel in kernel arguments, 48 per launch.  rows > 0: gradient centralisation with rows =
 * shape[0]; rows = 0: none.  step_lr / flags are per tensor because the reference keeps `step` per parameter.         */
typedef struct MsegRangerJob {
  float* p;
  const float* g;
  float* m;        /* exp_avg     */
  float* v;        /* exp_avg_sq  */
  float* slow;     /* slow_buffer */
  uint64_t n;
  int32_t rows;
  float step_lr;   /* step_size * lr */
  uint32_t flags;  /* bit 0: rectified (N_sma > threshold), bit 1: lookahead blend this step */
  uint32_t reserved;
} MsegRangerJob;
int mseg_ranger_step_multi(const MsegRangerJob* jobs, int njobs, double beta1, double beta2, double eps, double alpha,
                           void* stream);

/* ---- inference post-processing (src/inference/postprocessing.py) --------------------------------------------
 * distance_postprocessing(border, cell, th_seed, th_cell) (postprocessing.py:7-59) and
 * boundary_postprocessing(softmax probs HWC) (postprocessing.py:62-90): gaussian(0.5) -> thresholds -> 8-conn CCL
 * -> small-seed removal -> relabel -> marker watershed (4-conn), labels bit-exact.  Inputs: device fp32 [H][W]
 * (distance) / [H][W][3] (boundary); output: device uint16 [H][W].  `col_major_ids` selects the id numbering the
 * reference gets for (H,W,1) inputs.  status_dev[0] receives flags (bit0: exact
 * serial path was used).                                                                                        */
size_t mseg_postproc_workspace_bytes(int H, int W);
/* Test / tuning hook of the per-component watershed flood (one wavefront per mask component, csrc/postproc.hip):
 * heap_rows = rows of the wave's queue kept in LDS (1..16, the rest spills to the workspace); tile_small_px /
 * tile_large_px = bounding-box capacities (pixels incl. a 1 px rim) of the two LDS-staged launches (<= 8192 / <= 30720;
 * 0 = probe global memory instead).  Negative values restore the defaults.  Results never depend on these values; the
 * tests use them to drive the spill and global-probe paths.  Process-wide. */
int mseg_postproc_tuning(int heap_rows, int tile_small_px, int tile_large_px);
/* Test / ablation hook: the marker phase of a constant-image flood (the boundary method: watershed(image = mask), every key
 * ties) — 1 (default) = the closed form of what the reference's binary heap does with equal keys (markers surface in the
 * preorder of the implicit heap tree, array slots turn into pushed entries in its postorder, a marker still in the array's
 * last slot jumps the queue; 64 pops per step), 0 = the replay of the heap itself.  Same pop and push order, same labels. */
int mseg_postproc_set_const_stream(int on);
int mseg_distance_postprocess(const float* border, const float* cell, int H, int W, float th_cell, float th_seed,
                              int col_major_ids, uint16_t* labels, int32_t* n_instances_dev, int32_t* status_dev,
                              void* ws, size_t ws_bytes, void* stream);
int mseg_boundary_postprocess(const float* probs_hwc, int H, int W, uint16_t* labels, int32_t* n_instances_dev,
                              int32_t* status_dev, void* ws, size_t ws_bytes, void* stream);
/* The same in three calls, so that the frames of a stack (infer.py:250-259: a 2D+t stack, frame by frame) can be in flight
 * together: _pre per frame on its OWN workspace (thresholds, components, marker list), ONE _flood_batch for up to 8 frames
 * — the boundary method's flood is one wavefront busy for ~45 ms per 2048 x 2048 frame, a latency and not a load, and the
 * batch launch runs one workgroup per frame — then _post per frame.  ws_list: B host pointers to the frames' device
 * workspaces.  pre + flood_batch(B = 1) + post makes the launches of mseg_boundary_postprocess: same labels bit for bit. */
int mseg_boundary_postprocess_pre(const float* probs_hwc, int H, int W, void* ws, size_t ws_bytes, void* stream);
int mseg_boundary_flood_batch(void* const* ws_list, int B, int H, int W, void* stream);
int mseg_boundary_postprocess_post(int H, int W, uint16_t* labels, int32_t* n_instances_dev, int32_t* status_dev, void* ws,
                                   size_t ws_bytes, void* stream);

/* Threshold sweep of the evaluation (EvalWorker.inference, src/evaluation/eval.py:127-131,397-409: one
 * distance_postprocessing per (th_cell, th_seed) pair on the same prediction).  The smoothed cell map is computed once;
 * th_cell / th_seed are HOST arrays of nth floats; labels [nth][H][W], n_instances_dev / status_dev [nth] (nullable). */
int mseg_distance_postprocess_sweep(const float* border, const float* cell, int H, int W, const float* th_cell,
                                    const float* th_seed, int nth, int col_major_ids, uint16_t* labels,
                                    int32_t* n_instances_dev, int32_t* status_dev, void* ws, size_t ws_bytes,
                                    void* stream);

/* ---- training augmentation on the device (src/training/mytransforms.py:12-406) ------------------
 * The reference pipeline Flip -> Contrast -> Scaling -> Rotate -> Blur -> Noise -> ToTensor runs per sample on the CPU; here
 * the host draws every sample's random decisions along the same decision tree and these calls apply them to a batch of
 * fp32 planes [N][H][W] (uint16 value range until the final normalisation).  All *_dev arguments are device arrays.
 *   flip: codes[N] in 0..7 = identity, fliplr, flipud, rot90, rot180, rot270, fliplr+rot90, flipud+rot90 (np.rot90: ccw).
 *   affine: mats[N][6], source (x, y) = (m0 x + m1 y + m2, m3 x + m4 y + m5) of destination pixel (x, y); bilinear
 *           (nearest != 0: order 0 for uint8 labels), constant border 0; apply[N] == 0 copies the sample.
 *   blur: scipy.ndimage.gaussian_filter semantics (radius int(4 sigma + 0.5), 'reflect'); sigma <= 0 copies.
 *   stats: {min, max, mean}[N] (+ 65536-bin histogram per sample when hist != NULL).
 *   contrast_params / contrast: choice[N][4] = {mode, a, b, -}: mode 1 stretch to the (a, b) percentiles
 *           (np.percentile + rescale_intensity), mode 2 contrast factor a and gamma b (mytransforms.py:103-122).
 *   clahe: choice[s][0] == 3: contrast-limited adaptive histogram equalisation (equalize_adapthist defaults: 8 x 8 tiles,
 *           256 bins, clip limit 0.01; mytransforms.py:92-95), other samples copied; ws >= mseg_aug_clahe_workspace_bytes(N).
 *   noise_normalize: additive Gaussian noise of sigma = frac[N] * max (0: none), clip to uint16, then ToTensor's
 *           min_max_normalization to [-1, 1] with (vmin, vmax).                                                     */
int mseg_aug_u16_to_f32(const uint16_t* in, float* out, size_t n, void* stream);
int mseg_aug_flip(const float* in, float* out, int N, int H, int W, const int32_t* codes_dev, void* stream);
int mseg_aug_affine(const float* in, float* out, int N, int H, int W, const float* mats_dev, const int32_t* apply_dev,
                    int nearest, void* stream);
int mseg_aug_blur(const float* in, float* tmp, float* out, int N, int H, int W, const float* sigmas_dev, void* stream);
int mseg_aug_stats(const float* in, int N, int H, int W, float* stats_dev, uint32_t* hist_dev, void* stream);
int mseg_aug_contrast_params(const float* stats_dev, const uint32_t* hist_dev, const float* choice_dev, int N, int HW,
                             float* par_dev, void* stream);
int mseg_aug_contrast(const float* in, float* out, int N, int H, int W, const float* par_dev, void* stream);
size_t mseg_aug_clahe_workspace_bytes(int N);
int mseg_aug_clahe(const float* in, float* out, int N, int H, int W, const float* choice_dev, void* ws, void* stream);
int mseg_aug_noise_normalize(const float* in, float* out, int N, int H, int W, const float* frac_dev,
                             const float* stats_dev, uint32_t seed, float vmin, float vmax, void* stream);

/* ---- label creation for the boundary method (first part) --------------------------------------------
 * boundary_label (mode 0) / border_label (mode 1) of src/training/train_data_representations.py:75-125 for a batch of
 * instance masks [N][H][W] (uint16): 2 = boundary / touching border, 1 = cell interior, 0 = background; exact.       */
int mseg_label_boundary(const uint16_t* mask, int N, int H, int W, int mode, uint8_t* out, void* stream);

/* ---- label creation for the distance method (second part) ---------------------------------------------
 * distance_label(label, search_radius) of src/training/train_data_representations.py:261-361 (with bottom_hat_closing
 * :40-72) for a batch of instance masks [N][H][W] (uint16): cell_out = per-cell normalised Euclidean distance transform
 * inside the search window around the rounded centroid; neighbor_out = neighbour distances (inverse normalised distance
 * to the other cells of the window, gaps between close cells from the disk(3) bottom-hat transform, borders of touching
 * cells), rescaled and closed with a 3x3 grey closing.  Both float32 [N][H][W].  ws: device scratch of at least
 * mseg_label_distance_workspace_bytes(N, H, W) bytes (0 = unsupported shape; H, W <= 32767).                          */
size_t mseg_label_distance_workspace_bytes(int N, int H, int W);
int mseg_label_distance(const uint16_t* mask, int N, int H, int W, int search_radius, float* cell_out,
                        float* neighbor_out, void* ws, size_t ws_bytes, void* stream);
/* bottom_hat_closing(label) of train_data_representations.py:40-72 alone: root_out[N][H][W] = raster index (inside its image)
 * of the first pixel of the gap component a pixel belongs to, -1 outside the gaps (the rank of a root among the distinct
 * roots of an image + 1 is measure.label's id); corr_out = 0 outside, 1 inside a gap, 0.8 on the 4-neighbour rim of a gap
 * with minor_axis_length >= 3.  Same workspace as mseg_label_distance.                                                */
int mseg_label_bottom_hat(const uint16_t* mask, int N, int H, int W, int32_t* root_out, float* corr_out, void* ws,
                          size_t ws_bytes, void* stream);
/* j4_label(label, k_neighbors, se_radius) of train_data_representations.py:157-216 (Pena et al. 2020): 0 background,
 * 1 cell, 2 touching (another instance inside the (2k+1)^2 window), 3 gap (bottom-hat with disk(se_radius)); uint8.
 * tmp: device scratch of N*H*W bytes.                                                                                 */
int mseg_label_j4(const uint16_t* mask, int N, int H, int W, int k_neighbors, int se_radius, uint8_t* tmp, uint8_t* out,
                  void* stream);
/* cell_distance_label(label, search_radius, apply_clipping, clip_val) of train_data_representations.py:219-258: the cell
 * distances alone; clip_val == 0: normalised per cell (label types 'cell_dist'), > 0: min(d, clip_val) / clip_val
 * ('cell_dist_clipped', clip_val 5).  Same workspace as mseg_label_distance.                                          */
int mseg_label_cell_distance(const uint16_t* mask, int N, int H, int W, int search_radius, float clip_val,
                             float* cell_out, void* ws, size_t ws_bytes, void* stream);
/* Largest skimage regionprops major_axis_length over the cells of each mask (CreateLabelsWorker.create_labels,
 * src/training/train.py:73-78, which sets search_radius = ceil(0.75 * ceil(max major axis))): out_dev double [N].    */
size_t mseg_label_major_axis_workspace_bytes(int N);
int mseg_label_max_major_axis(const uint16_t* mask, int N, int H, int W, double* out_dev, void* ws, size_t ws_bytes,
                              void* stream);

/* ---- evaluation helpers (EvalWorker.calc_scores, src/evaluation/eval.py:248-256) -----------------
 * mseg_eval_relabel: border_correction(mask, border_width) (src/utils/utils.py:25-47: instances not visible inside the
 *   frame minus its border are deleted) followed by skimage.measure.label (8-neighbours of EQUAL value connect; new ids
 *   1..K in raster order of each component's first pixel) -> lab_out int32 [H][W], *n_out_dev = K.
 * mseg_eval_pair_counts: the integer statistics of get_fast_aji_plus (src/evaluation/stats_utils.py:98-179) for two
 *   contiguous label images: area_t[nt+1], area_p[np+1], inter[nt+1][np+1] (device arrays, zeroed by the call).     */
size_t mseg_eval_workspace_bytes(int H, int W);
int mseg_eval_relabel(const uint16_t* mask, int H, int W, int border_width, int32_t* lab_out, int32_t* n_out_dev,
                      void* ws, size_t ws_bytes, void* stream);
int mseg_eval_pair_counts(const int32_t* true_lab, const int32_t* pred_lab, int H, int W, int nt, int np,
                          int32_t* area_t, int32_t* area_p, int32_t* inter, void* stream);

/* ---- misc ---------------------------------------------------------------------------------------------------- */
int mseg_version(void);
const char* mseg_strerror(int code);
int mseg_last_hip_error(void);

#ifdef __cplusplus
}
#endif
#endif /* MSEG_HIP_H */
