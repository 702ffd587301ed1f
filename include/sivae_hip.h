/* libsivae_hip — C ABI of the MI355X (gfx950 / CDNA4) Soft-IntroVAE training kernels.
 *
 * The reference (taldatech/soft-intro-vae-pytorch) has NO native layer: its hot path is a chain of
 * ATen ops issued from soft_intro_vae/train_soft_intro_vae.py.  This header is the drop-in boundary a
 * replacement shared library must export; each entry point names the reference op (file:line, all
 * relative to the reference checkout) whose arithmetic it reproduces.
 *
 * Conventions (all functions):
 *   - return int: 0 = ok; >0 = hipError_t of the launch; <0 = argument error (SIVAE_ERR_*).
 *   - never allocate, never synchronise, never touch global state: re-entrant, one stream per call.
 *   - all tensors are caller-owned device buffers, fp32, contiguous NCHW (or [rows][cols] where said).
 *   - scratch memory is passed as (workspace, workspace_bytes); sizes come from *_workspace_bytes().
 *   - `stream` is a hipStream_t (void* here so that the header needs no HIP include).
 */
#ifndef SIVAE_HIP_H
#define SIVAE_HIP_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SIVAE_OK 0
#define SIVAE_ERR_NULL -1      /* a required pointer is null */
#define SIVAE_ERR_SHAPE -2     /* unsupported / inconsistent shape */
#define SIVAE_ERR_KSIZE -3     /* kernel size not in {1,3,5} */
#define SIVAE_ERR_WORKSPACE -4 /* workspace missing or too small */
#define SIVAE_ERR_RANGE -5     /* tensor too large for the kernels' 32-bit addressing */
#define SIVAE_ERR_MODE -6      /* unknown mode / flag value */

typedef void* sivae_stream_t;

/* ---- probes ---------------------------------------------------------------------------------- */
int sivae_abi_version(void);
const char* sivae_arch(void); /* "gfx950" */
int sivae_device_count(void);

/* ---- convolution (stride 1, padding k/2, k in {1,3,5}) ------------------------------------------
 * nn.Conv2d call sites: ResidualBlock conv_expand/conv1/conv2 train_soft_intro_vae.py:51-61, Encoder
 * stem :89, Decoder predict :159; nn.Linear :109,:146 run through ks = 1 with H = W = 1.
 * Weights are consumed in a packed GEMM layout produced by sivae_pack_conv_weight:
 *   mode 0 = forward operand, mode 1 = data-gradient operand (transposed + 180-degree flipped). */
int sivae_conv_ck(int ks);
int sivae_conv_ci_pad(int ks, int ci);
int sivae_conv_co_pad(int co);
size_t sivae_pack_conv_weight_bytes(int Co, int Ci, int ks, int mode);
int sivae_pack_conv_weight(const float* w /*[Co][Ci][ks][ks]*/, float* wp, int Co, int Ci, int ks, int mode,
                           sivae_stream_t stream);
/* Batched form (round 4): ONE launch rebuilds one operand form of EVERY weight of a network after its optimizer step
 * (torch.optim.Adam.step of train_soft_intro_vae.py:588,622 changes all of them; 35 per-weight launches of 5-11 us became
 * 5).  A job table lives in device memory and is reused from step to step (the packed buffers do not move):
 *   sivae_pack_job_fill   writes job `index` of a HOST table (sivae_pack_job_bytes() bytes per job) for operand form
 *                         0 direct (ks, mode) / 1 Winograd F(2x2,3x3) (mode) / 2 Winograd F(4x4,3x3) (mode) /
 *                         3 upsample-phase forward / 4 upsample-phase data gradient / 5 Winograd F(4x4,3x3) pre-split
 *                         into three bf16 pieces (mode; sivae_pack_wino4_b6_weight) / 6 bf16 operand slabs of the bf16
 *                         mode (ks incl. the code 51, mode; sivae_bf16_pack_conv_weight); dst = the buffer the per-weight
 *                         sivae_pack_* call of that form writes; returns the job's block count (its blocks are
 *                         [first_block, first_block + count) of the launch) or an error code (< 0)
 *   sivae_pack_batch      the launch: jobs_dev = the uploaded table, block_job_dev[b] = job index of block b (uint16) */
int sivae_pack_job_bytes(void);
int sivae_pack_job_fill(void* jobs_host, int index, int form, const float* w, float* dst, int Co, int Ci, int ks, int mode,
                        int first_block);
int sivae_pack_batch(int form, const void* jobs_dev, const unsigned short* block_job_dev, int n_blocks,
                     sivae_stream_t stream);

/* y[B][Co][H][W] (+)= conv(x', wp) + bias.
 *   x' = x, or LeakyReLU((x-pro_mean[c])*pro_invstd[c]*pro_gamma[c]+pro_beta[c], pro_slope) when
 *        pro_mean != NULL (producer BatchNorm2d + LeakyReLU fused into the load, :58-59,:90-91);
 *   upsample != 0: x is [B][Ci][H/2][W/2] and is read through nn.Upsample(2,'nearest') (:155);
 *   stats_partial != NULL: per-pixel-tile per-channel {sum, sumsq} of y, [n_px_tiles][Co][2], for the
 *        consumer BatchNorm2d (see sivae_bn_stats_from_conv);
 *   accumulate != 0: y += result (used to sum the two branches of the residual data gradient).
 * The data gradient of the same conv is this function on dy with the mode-1 pack (Ci/Co swapped):
 * aten::convolution_backward (input half). */
int sivae_conv2d_fwd_num_px_tiles(int B, int Co, int H, int W);
int sivae_conv2d_fwd(const float* x, const float* wp, float* y, const float* bias, const float* pro_mean,
                     const float* pro_invstd, const float* pro_gamma, const float* pro_beta, float pro_slope,
                     float* stats_partial, int B, int Ci, int Co, int H, int W, int ks, int upsample,
                     int accumulate, sivae_stream_t stream);

/* Winograd F(2x2,3x3) version of sivae_conv2d_fwd for ks == 3 (same nn.Conv2d(k=3,s=1,p=1) of
 * soft_intro_vae/train_soft_intro_vae.py:56-61; same fused prologue / upsample / epilogues): 2.25x fewer
 * multiplies on the fp32 matrix pipe.  `up` is the transformed filter U = G g G^T from
 * sivae_pack_wino_weight (mode 0 forward, mode 1 data gradient).  Handles even H >= 8 and even W >= 16, and the 8x8 / 4x4 maps
 * (sivae_conv2d_wino_supported); stats_partial has sivae_conv2d_wino_num_px_tiles(B, H, W) rows.  bias must be
 * NULL (SIVAE_ERR_MODE otherwise): no 3x3 conv of the model has one. */
size_t sivae_pack_wino_weight_bytes(int Co, int Ci, int mode);
int sivae_pack_wino_weight(const float* w /*[Co][Ci][3][3]*/, float* up, int Co, int Ci, int mode,
                           sivae_stream_t stream);
int sivae_conv2d_wino_supported(int H, int W);
int sivae_conv2d_wino_num_px_tiles(int B, int H, int W);
int sivae_conv2d_wino_fwd(const float* x, const float* up, float* y, const float* bias, const float* pro_mean,
                          const float* pro_invstd, const float* pro_gamma, const float* pro_beta, float pro_slope,
                          float* stats_partial, int B, int Ci, int Co, int H, int W, int upsample, int accumulate,
                          sivae_stream_t stream);

/* Split-K form of sivae_conv2d_wino_fwd for launches that would leave most of the chip idle (the deep 4x4 / 8x8 layers
 * at small batch: the 16-image per-GPU shard of config 4): the input-channel range is cut into
 * sivae_conv2d_wino_splitk(...) slices computed by separate work items into `workspace`, and a fixed-order reduce kernel
 * sums them into y (+= with accumulate) and writes per-image {sum, sumsq} rows.  With 1 slice it IS
 * sivae_conv2d_wino_fwd.  stats_partial has sivae_conv2d_wino_splitk_stats_rows(...) rows. */
int sivae_conv2d_wino_splitk(int B, int Ci, int Co, int H, int W);
size_t sivae_conv2d_wino_splitk_workspace_bytes(int B, int Ci, int Co, int H, int W);
int sivae_conv2d_wino_splitk_stats_rows(int B, int Ci, int Co, int H, int W);
int sivae_conv2d_wino_fwd_splitk(const float* x, const float* up, float* y, const float* pro_mean,
                                 const float* pro_invstd, const float* pro_gamma, const float* pro_beta, float pro_slope,
                                 float* stats_partial, int B, int Ci, int Co, int H, int W, int upsample, int accumulate,
                                 void* workspace, size_t workspace_bytes, sivae_stream_t stream);

/* 3x3 conv of a nearest-2x-upsampled input (nn.Upsample :155 -> ResidualBlock.conv1 :56), computed on the low-resolution
 * tensor: four output-parity phases, each a 2x2 conv run as Winograd F(2x2,2x2) — 36 multiplies per 4x4 output pixels
 * instead of 64 (sivae_conv2d_wino_fwd with upsample) or 144 (direct).  x_half is [B][Ci][H/2][W/2], y is
 * [B][Co][H][W]; same BatchNorm+LeakyReLU prologue and BatchNorm-partials epilogue (stats_partial has
 * sivae_conv2d_wino_up_num_px_tiles(B, H, W) rows).  Forward only (the data gradient keeps the F(2x2,3x3) kernel).
 * Supported maps: H >= 16 even, W >= 32 with W % 4 == 0 (the kernel writes four output pixels per 16-byte store; y must be
 * 16-byte aligned). */
size_t sivae_pack_wino_up_weight_bytes(int Co, int Ci);
int sivae_pack_wino_up_weight(const float* w /*[Co][Ci][3][3]*/, float* up, int Co, int Ci, sivae_stream_t stream);
int sivae_conv2d_wino_up_supported(int H, int W);
int sivae_conv2d_wino_up_num_px_tiles(int B, int H, int W);
int sivae_conv2d_wino_up_fwd(const float* x_half, const float* up, float* y, const float* pro_mean,
                             const float* pro_invstd, const float* pro_gamma, const float* pro_beta, float pro_slope,
                             float* stats_partial, int B, int Ci, int Co, int H, int W, sivae_stream_t stream);

/* Data gradient of the same op with respect to the low-resolution input (replaces the full-resolution F(2x2,3x3) data
 * gradient followed by nn.Upsample's 2x2 block sum): one 2x2 conv over the 4*C parity planes dy[2i+p][2j+q] of
 * dy [B][C][2Hs][2Ws] (read in place with stride-2 addressing) as Winograd F(2x2,2x2) with the phases folded into K.
 * dx is [B][N][Hs][Ws] (N = the conv's input channels); accumulate != 0: dx += result.
 * sivae_space_to_depth2 materialises the parity planes ([B][4][C][Hs][Ws]); the kernels do not need it. */
int sivae_space_to_depth2(const float* in /*[B][C][2Hs][2Ws]*/, float* out /*[B][4][C][Hs][Ws]*/, int B, int C, int Hs,
                          int Ws, sivae_stream_t stream);
size_t sivae_pack_wino_up_dgrad_weight_bytes(int Co, int Ci);
int sivae_pack_wino_up_dgrad_weight(const float* w /*[Co][Ci][3][3]*/, float* ud, int Co, int Ci,
                                    sivae_stream_t stream);
int sivae_conv2d_wino_up_dgrad_supported(int Hs, int Ws);
int sivae_conv2d_wino_up_dgrad(const float* dy, const float* ud, float* dx, int B, int C, int N, int Hs, int Ws,
                               int accumulate, sivae_stream_t stream);
/* Split-K form for small shards (16 images per GPU: 512 -> 512 @16x16 is 128 work items each walking 4 x 512 planes):
 * _splitk = number of K slices the run would use (1: the plain kernel), _splitk_workspace_bytes = size of the partial
 * outputs (0 when 1 slice), _splitk_run = sliced kernel + fixed-order reduce (same numbers as the plain entry up to the
 * fp32 summation order of the slices). */
int sivae_conv2d_wino_up_dgrad_splitk(int B, int C, int N, int Hs, int Ws);
size_t sivae_conv2d_wino_up_dgrad_splitk_workspace_bytes(int B, int C, int N, int Hs, int Ws);
int sivae_conv2d_wino_up_dgrad_splitk_run(const float* dy, const float* ud, float* dx, int B, int C, int N, int Hs,
                                          int Ws, int accumulate, void* workspace, size_t workspace_bytes,
                                          sivae_stream_t stream);

/* Weight gradient of the same op in the phase form (dU_pq = sum_tiles (A dY_pq A^T).(B^T d_pq B), F(2x2,2x2);
 * dW folded back from the four 2x2 phase filters): 36 multiplies per 4x4 dy pixels instead of 64.  x_half is
 * [B][Ci][Hs][Ws], dy [B][Co][2Hs][2Ws], dw [Co][Ci][3][3]; deterministic two-pass reduction. */
int sivae_conv2d_wino_up_wgrad_supported(int Hs, int Ws);
size_t sivae_conv2d_wino_up_wgrad_workspace_bytes(int B, int Ci, int Co, int Hs, int Ws);
int sivae_conv2d_wino_up_wgrad(const float* x_half, const float* dy, float* dw, int B, int Ci, int Co, int Hs, int Ws,
                               void* workspace, size_t workspace_bytes, sivae_stream_t stream);

/* sivae_conv2d_wino_fwd used as a DATA GRADIENT whose output y = dL/dh feeds the backward of
 * h = LeakyReLU(BatchNorm(bn_x)) (:58-59): the epilogue also reads bn_x (same shape as y) and leaves per pixel tile
 * {sum g, sum g*xhat}, g = y * LeakyReLU'(z), in bnbwd_partial [sivae_conv2d_wino_num_px_tiles(B,H,W)][Co][2] —
 * the first reduction pass of the BatchNorm backward; sivae_bn_bwd_from_partials finishes it. */
int sivae_conv2d_wino_dgrad_bnbwd(const float* dy, const float* up, float* y, const float* bn_x, const float* bn_mean,
                                  const float* bn_invstd, const float* bn_gamma, const float* bn_beta, float slope,
                                  float* bnbwd_partial, int B, int Ci, int Co, int H, int W, sivae_stream_t stream);
int sivae_bn_bwd_from_partials(const float* dy, const float* x, const float* mean, const float* invstd,
                               const float* gamma, const float* beta, float slope, const float* partials, int n_tiles,
                               float* dx, float* dgamma, float* dbeta, int B, int C, int HW, void* workspace,
                               size_t workspace_bytes, sivae_stream_t stream);

/* Winograd-domain weight gradient for ks == 3 (dU = sum_tiles (A dY A^T) . (B^T d B), dW = G^T dU G): the
 * weight half of aten::convolution_backward for the nn.Conv2d(k=3) layers (:56-61) with 2.25x fewer multiplies;
 * same prologue / upsample options and the same deterministic two-pass reduction as sivae_conv2d_wgrad.
 * Maps: sivae_conv2d_wino_wgrad_supported(H, W) (even H >= 8, even W >= 16). */
int sivae_conv2d_wino_wgrad_supported(int H, int W);
size_t sivae_conv2d_wino_wgrad_workspace_bytes(int B, int Ci, int Co, int H, int W);
int sivae_conv2d_wino_wgrad(const float* x, const float* dy, float* dw, const float* pro_mean,
                            const float* pro_invstd, const float* pro_gamma, const float* pro_beta, float pro_slope,
                            int B, int Ci, int Co, int H, int W, int upsample, void* workspace,
                            size_t workspace_bytes, sivae_stream_t stream);

/* dw[Co][Ci][ks][ks] = weight gradient (aten::convolution_backward, weight half); x is read with the
 * same optional prologue / upsample addressing as the forward. Deterministic split-K. */
size_t sivae_conv2d_wgrad_workspace_bytes(int B, int Ci, int Co, int H, int W, int ks);
int sivae_conv2d_wgrad(const float* x, const float* dy, float* dw, const float* pro_mean,
                       const float* pro_invstd, const float* pro_gamma, const float* pro_beta, float pro_slope,
                       int B, int Ci, int Co, int H, int W, int ks, int upsample, void* workspace,
                       size_t workspace_bytes, sivae_stream_t stream);

/* ---- 1x1 convolution as a streaming kernel (ResidualBlock.conv_expand :50-54, forward and data gradient) ---------
 * y[b][co][p] (+)= sum_ci W[co][ci] x[b][ci][p] over planes of HW pixels; wp is the ks = 1 direct pack of
 * sivae_pack_conv_weight (mode 0 forward, mode 1 data gradient).  x goes straight from 16-byte global loads into the MFMA
 * B operand (a lane's four consecutive pixels serve four column tiles), the 64-channel weight tile lives in LDS, outputs
 * leave as 16-byte stores.  _supported: HW % 4 == 0, Ci even and <= 256. */
int sivae_conv1x1_stream_supported(int B, int Ci, int Co, int HW);
int sivae_conv1x1_stream(const float* x, const float* wp, float* y, int B, int Ci, int Co, int HW, int accumulate,
                         sivae_stream_t stream);

/* ---- 5x5 convolution FROM <= 3 channels INTO <= 64 with the whole contraction merged (K = 3*25 = 75) -----------
 * The encoder stem's forward (:88-89: nn.Conv2d(cdim, 64, 5, 1, 2)) and the data gradient of Decoder.predict (:159):
 * y[co][px] = sum_k W[co][k] X[k][px], k = (ci, kh, kw); 38 MFMA k-steps per 32x32 outputs instead of 50, weights in
 * registers.  pack mode 0: w [n_big][n_small][5][5] (forward); mode 1: w [n_small][n_big][5][5] (data gradient:
 * taps flipped, channels transposed).  stats_partial (optional): [sivae_conv5_k75_num_px_tiles(B, H, W)][Co][2]
 * {sum, sumsq} partials for the BatchNorm that follows the stem. */
size_t sivae_pack_conv5_k75_bytes(void);
int sivae_pack_conv5_k75(const float* w, float* wq, int n_small, int n_big, int mode, sivae_stream_t stream);
int sivae_conv5_k75_supported(int Ci, int Co);
int sivae_conv5_k75_num_px_tiles(int B, int H, int W);
int sivae_conv5_k75_fwd(const float* x, const float* wq, float* y, const float* bias, float* stats_partial, int B,
                        int Ci, int Co, int H, int W, sivae_stream_t stream);

/* ---- 5x5 convolutions with <= 3 channels on one side (Decoder.predict :159, Encoder stem :89) -------------
 * The small channel count is merged with the kernel column into one 16-wide MFMA dimension.
 *   pack: n_small = the <=3 side, n_big = the other; mode 0 = forward operand of a [n_small][n_big][5][5]
 *         weight (predict), mode 1 = data-gradient operand of a [n_big][n_small][5][5] weight (stem).
 *   sivae_conv5_smallco_fwd: y[B][Co<=3][H][W] = conv5x5(x[B][Ci][H][W]) + bias.
 *   sivae_conv5_edge_wgrad:  dw[Co][Ci][5][5] with min(Ci, Co) <= 3 (deterministic split over pixels). */
size_t sivae_pack_conv5_smallco_bytes(int n_small, int n_big);
int sivae_pack_conv5_smallco(const float* w, float* wq, int n_small, int n_big, int mode, sivae_stream_t stream);
int sivae_conv5_smallco_fwd(const float* x, const float* wq, float* y, const float* bias, int B, int Ci, int Co,
                            int H, int W, sivae_stream_t stream);
size_t sivae_conv5_edge_wgrad_workspace_bytes(int B, int Ci, int Co, int H, int W);
int sivae_conv5_edge_wgrad(const float* x, const float* dy, float* dw, int B, int Ci, int Co, int H, int W,
                           void* workspace, size_t workspace_bytes, sivae_stream_t stream);

/* ---- BatchNorm2d (training mode) + LeakyReLU + residual add ----------------------------------------
 * nn.BatchNorm2d(eps 1e-5, momentum 0.1) :58,:62,:90 ; nn.LeakyReLU(0.2) :59,:63,:91 ; torch.add :74.
 * running_var gets the unbiased variance, num_batches_tracked (int64, device) is incremented. */
size_t sivae_bn_workspace_bytes(int B, int C, int HW);
int sivae_bn_stats(const float* x, int B, int C, int HW, float eps, float momentum, float* running_mean,
                   float* running_var, long long* num_batches_tracked, float* mean_out, float* invstd_out,
                   void* workspace, size_t workspace_bytes, sivae_stream_t stream);
int sivae_bn_stats_from_conv(const float* partials, int n_tiles, int B, int C, int HW, float eps, float momentum,
                             float* running_mean, float* running_var, long long* num_batches_tracked,
                             float* mean_out, float* invstd_out, sivae_stream_t stream);
/* ---- synchronised BatchNorm for data-parallel runs (opt-in): the shard's per-channel {sum, sumsq}
 * (fp64 [C][2]) from the conv-epilogue partials — the caller all-reduces it — and the finalize from the (global) sums
 * and the global element count.  Backward: `reduce` leaves the local {sum dz, sum dz*xhat} (fp64 [C][2]), `apply`
 * takes the local sums (-> dgamma, dbeta) and the all-reduced sums + global count (-> dx, dz). */
int sivae_bn_sums_from_conv(const float* partials, int n_tiles, int C, double* sums, sivae_stream_t stream);
int sivae_bn_finalize_sums(const double* sums, int C, double count, float eps, float momentum, float* running_mean,
                           float* running_var, long long* num_batches_tracked, float* mean_out, float* invstd_out,
                           sivae_stream_t stream);
int sivae_bn_bwd_reduce(const float* dy, const float* y, const float* x, const float* mean, const float* invstd,
                        const float* gamma, const float* beta, int act_mode, float slope, double* sums, int B, int C,
                        int HW, void* workspace, size_t workspace_bytes, sivae_stream_t stream);
int sivae_bn_bwd_apply(const float* dy, const float* y, const float* x, const float* mean, const float* invstd,
                       const float* gamma, const float* beta, int act_mode, float slope, const double* sums_local,
                       const double* sums_global, double count_global, float* dx, float* dz_out, float* dgamma,
                       float* dbeta, int B, int C, int HW, void* workspace, size_t workspace_bytes,
                       sivae_stream_t stream);
/* one more running-stat update from saved batch statistics (a forward pass replayed from cached
 * activations still counts as one BatchNorm call of the reference); count = B*H*W. */
int sivae_bn_update_running(const float* mean, const float* invstd, int C, double count, float eps, float momentum,
                            float* running_mean, float* running_var, long long* num_batches_tracked,
                            sivae_stream_t stream);
/* y = LeakyReLU((x-mean[c])*invstd[c]*gamma[c]+beta[c] (+ res), slope); slope = 1 -> identity act. */
int sivae_bn_apply_act(const float* x, const float* res, const float* mean, const float* invstd,
                       const float* gamma, const float* beta, float slope, float* y, int B, int C, int HW,
                       sivae_stream_t stream);
/* same op, also writing AvgPool2d(2) (:92,:98) of the result in the same pass: y [B][C][H][W] (kept for backward)
 * and y_pooled [B][C][H/2][W/2]; res may be NULL; y may be NULL (pooled output only: the stem, whose backward
 * recomputes the activation from the conv output); H even, W % 4 == 0. */
int sivae_bn_apply_act_pool(const float* x, const float* res, const float* mean, const float* invstd,
                            const float* gamma, const float* beta, float slope, float* y, float* y_pooled, int B, int C,
                            int H, int W, sivae_stream_t stream);
/* same, with the residual stored at half resolution [B][C][H/2][W/2] and read through nn.Upsample(2,'nearest')
 * (:155) addressing — the upsampled tensor is never written; H even, W % 4 == 0. */
int sivae_bn_apply_act_resup(const float* x, const float* res_half, const float* mean, const float* invstd,
                             const float* gamma, const float* beta, float slope, float* y, int B, int C, int H, int W,
                             sivae_stream_t stream);
/* LeakyReLU sign mask (1 bit per element: pre-activation > 0; element e -> bit e&7 of byte e>>3).  The apply pass
 * of "LeakyReLU(BN(x) + res)" (ResidualBlock output, train_soft_intro_vae.py:71-74) writes it next to its output and
 * the backward reads it instead of the saved output: 1/32 of a tensor per backward pass, and an encoder block
 * (output consumed only through the AvgPool2d behind it, :95-99) never writes its full-resolution output.
 *   y_pooled != NULL: + AvgPool2d(2), y may be NULL; res_up != 0: res is [B][C][H/2][W/2] read through
 *   Upsample(2,'nearest') addressing (:155).  mask: sivae_bn_signmask_bytes() bytes.  H even, W % 8 == 0. */
size_t sivae_bn_signmask_bytes(int B, int C, int HW);
int sivae_bn_apply_act_signmask(const float* x, const float* res, int res_up, const float* mean, const float* invstd,
                                const float* gamma, const float* beta, float slope, float* y, float* y_pooled,
                                unsigned char* mask, int B, int C, int H, int W, sivae_stream_t stream);
/* backward with the sign from that mask.  dy_pooled != 0: dy is the gradient of AvgPool2d(2)(output) at half
 * resolution; dz_sum != 0: dz_out = 2x2 block sums of the residual-branch gradient [B][C][H/2][W/2] (not both). */
int sivae_bn_bwd_signmask(const float* dy, const unsigned char* mask, const float* x, const float* mean,
                          const float* invstd, const float* gamma, float slope, float* dx, float* dz_out,
                          float* dgamma, float* dbeta, int B, int C, int H, int W, int dy_pooled, int dz_sum,
                          void* workspace, size_t workspace_bytes, sivae_stream_t stream);
/* backward of the above: dz = dy*(s>0?1:slope); dx = BN backward of dz; dz_out (optional) = gradient
 * of the residual branch; dgamma/dbeta optional.  act_mode selects where the LeakyReLU sign s comes from:
 *   0 no activation, 1 the saved OUTPUT y (valid since slope > 0), 2 recomputed from x (needs beta; used
 *   when the BatchNorm output was never stored because it was fused into the next conv's load). */
int sivae_bn_bwd(const float* dy, const float* y, const float* x, const float* mean, const float* invstd,
                 const float* gamma, const float* beta, int act_mode, float slope, float* dx, float* dz_out,
                 float* dgamma, float* dbeta, int B, int C, int HW, void* workspace, size_t workspace_bytes,
                 sivae_stream_t stream);
/* act_mode-1 backward (LeakyReLU sign from the saved output y) that returns the residual-branch gradient as its 2x2
 * block sum dz_half [B][C][H/2][W/2] — the adjoint of the nn.Upsample (:155) in front of a decoder block — instead of
 * the full-resolution dz; H even, W % 4 == 0. */
int sivae_bn_bwd_dzsum(const float* dy, const float* y, const float* x, const float* mean, const float* invstd,
                       const float* gamma, float slope, float* dx, float* dz_half, float* dgamma, float* dbeta, int B,
                       int C, int H, int W, void* workspace, size_t workspace_bytes, sivae_stream_t stream);
/* same with dy given as the gradient of AvgPool2d(2)(y) at half resolution [B][C][H/2][W/2] (:92,:98): the pool's
 * adjoint (0.25 * dy_half[h>>1][w>>1]) is applied on load; H even, W % 4 == 0. */
int sivae_bn_bwd_pooled_dy(const float* dy_half, const float* y, const float* x, const float* mean,
                           const float* invstd, const float* gamma, const float* beta, int act_mode, float slope,
                           float* dx, float* dz_out, float* dgamma, float* dbeta, int B, int C, int H, int W,
                           void* workspace, size_t workspace_bytes, sivae_stream_t stream);
/* out[c] = sum over (B, HW) — bias gradient of Decoder.predict (:159). Workspace as sivae_bn_workspace_bytes. */
int sivae_channel_sum(const float* x, float* out, int B, int C, int HW, void* workspace, size_t workspace_bytes,
                      sivae_stream_t stream);

/* ---- pooling / upsampling / ReLU --------------------------------------------------------------------
 * nn.AvgPool2d(2) :92,:98 (floor semantics for odd sizes); nn.Upsample(scale_factor=2, 'nearest') :155;
 * nn.ReLU(True) :147. rows = B*C. */
int sivae_avgpool2_fwd(const float* x, float* y, int rows, int Hin, int Win, sivae_stream_t stream);
int sivae_avgpool2_bwd(const float* dy, float* dx, int rows, int Hin, int Win, sivae_stream_t stream);
int sivae_upsample2_fwd(const float* x, float* y, int rows, int H, int W, sivae_stream_t stream);
int sivae_upsample2_bwd(const float* dy, float* dx, int rows, int H, int W, sivae_stream_t stream);
int sivae_relu_fwd(const float* x, float* y, size_t n, sivae_stream_t stream);
int sivae_relu_bwd(const float* dy, const float* y, float* dx, size_t n, sivae_stream_t stream);
int sivae_add_inplace(float* y, const float* x, size_t n, sivae_stream_t stream);

/* ---- sampler and losses -------------------------------------------------------------------------------
 * reparameterize :254-265 ; calc_kl :231-251 ; calc_reconstruction_loss :268-294 ; expELBO :580-581.
 * mu / logvar are [B][Z] with leading dimension ld (the two halves of the encoder fc output). */
int sivae_reparam_fwd(const float* mu, const float* logvar, int ld, const float* eps, float* z, int B, int Z,
                      sivae_stream_t stream);
int sivae_reparam_bwd(const float* dz, const float* logvar, int ld, const float* eps, float* dmu, float* dlogvar,
                      int ldg, int B, int Z, sivae_stream_t stream);
int sivae_kl_fwd(const float* logvar, const float* mu, int ld, float mu_o, float logvar_o, float* out /*[B]*/,
                 int B, int Z, sivae_stream_t stream);
int sivae_kl_bwd(const float* g, int g_per_sample, float g_scale, const float* logvar, const float* mu, int ld,
                 float mu_o, float logvar_o, float* dlogvar, float* dmu, int ldg, int B, int Z,
                 sivae_stream_t stream);
/* the same with TENSOR priors (calc_kl's mu_o / logvar_o may be tensors, :237-243): device pointers read through
 * broadcast strides in elements (row stride, column stride; 0 = broadcast), i.e. 0-d, [Z], [1][Z], [B][1] or [B][Z]
 * priors without a host round trip.  Gradients are those with respect to logvar / mu. */
int sivae_kl_fwd_t(const float* logvar, const float* mu, int ld, const float* mu_o, int mu_o_rs, int mu_o_cs,
                   const float* logvar_o, int lv_o_rs, int lv_o_cs, float* out /*[B]*/, int B, int Z,
                   sivae_stream_t stream);
int sivae_kl_bwd_t(const float* g, int g_per_sample, float g_scale, const float* logvar, const float* mu, int ld,
                   const float* mu_o, int mu_o_rs, int mu_o_cs, const float* logvar_o, int lv_o_rs, int lv_o_cs,
                   float* dlogvar, float* dmu, int ldg, int B, int Z, sivae_stream_t stream);
/* loss_type: 0 mse, 1 l1, 2 bce.  rowsum: out[b] = sum_i term(x[b,i], recon[b,i]). */
size_t sivae_recon_workspace_bytes(int B, int D);
int sivae_recon_rowsum_fwd(const float* x, const float* recon, int loss_type, float* out, int B, int D,
                           void* workspace, size_t workspace_bytes, sivae_stream_t stream);
/* g_mode 0: g[B] per sample, 1: scalar g[0], 2: per element g[B*D]; effective grad = g * g_scale. */
int sivae_recon_bwd(const float* x, const float* recon, int loss_type, const float* g, int g_mode, float g_scale,
                    float* d_recon, float* d_x, int B, int D, sivae_stream_t stream);
int sivae_recon_elem_fwd(const float* x, const float* recon, int loss_type, float* out, size_t numel,
                         sivae_stream_t stream);
int sivae_vec_sum(const float* v, int n, float scale, float* out, sivae_stream_t stream);
/* e[b] = exp(-2*scale*(beta_rec*L[b] + beta_neg*KL[b])), out[0] = mean_b e[b]. */
int sivae_expelbo_fwd(const float* L, const float* KL, float scale, float beta_rec, float beta_neg, int B,
                      float* e, float* out, sivae_stream_t stream);
int sivae_expelbo_bwd(const float* gout, const float* e, float scale, float beta_rec, float beta_neg, int B,
                      float* dL, float* dKL, sivae_stream_t stream);
/* Weighted sum of up to six device scalars in one launch, terms added in index order: out[0] = sum_i w_i * p_i[0], i < n.
 * Replaces the 0-dim torch arithmetic of lossE / lossD (train_soft_intro_vae.py:583-586, :618-620). */
int sivae_lincomb(const float* p0, const float* p1, const float* p2, const float* p3, const float* p4, const float* p5,
                  float w0, float w1, float w2, float w3, float w4, float w5, int n, float* out, sivae_stream_t stream);
/* its gradient: out[i] = g[0] * w_i, i < n. */
int sivae_lincomb_bwd(const float* g, float w0, float w1, float w2, float w3, float w4, float w5, int n, float* out,
                      sivae_stream_t stream);
/* Philox4x32-10 standard normals (replaces torch.randn / randn_like :264,:547 in fast mode). */
int sivae_randn(float* out, size_t n, unsigned long long seed, unsigned long long offset, sivae_stream_t stream);
/* same stream with its position kept in device memory (*offset_dev is read, then advanced by ceil(n/4)): nothing
 * that a captured HIP graph bakes in changes between replays. */
int sivae_randn_dev(float* out, size_t n, unsigned long long seed, unsigned long long* offset_dev,
                    sivae_stream_t stream);

/* ---- optimizer ------------------------------------------------------------------------------------------
 * torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8) :450-451, one launch over a flat parameter buffer. */
int sivae_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                    float step_size /* lr / (1 - beta1^t) */, float beta1, float beta2, float eps,
                    float bias_correction2_sqrt /* sqrt(1 - beta2^t) */, float grad_scale,
                    sivae_stream_t stream);

/* the same Adam step with its state on the device (for whole-iteration HIP graphs): state = double[4]
 * {t, lr, lr/(1-beta1^t), sqrt(1-beta2^t)}; the call advances t and refreshes the two factors before the update. */
int sivae_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, double* state,
                        double beta1, double beta2, float eps, float grad_scale, sivae_stream_t stream);

/* grad[i] += s0[i] + s1[i] + s2[i] + s3[i] (s1 .. s3 may be NULL): folds the per-use parameter-gradient slabs of one backward pass
 * into the flat gradient buffer — the reference's `loss.backward()` accumulation `p.grad += g` (:571, :619) for
 * parameters used by several passes, as ONE launch per network instead of one torch add per tensor and use. */
int sivae_sum_slabs(float* grad, const float* s0, const float* s1, const float* s2, const float* s3, size_t n,
                    sivae_stream_t stream);

/* ---- input side ------------------------------------------------------------------------------
 * uint8 image batch [B][C][H][W] (nhwc == 0) or [B][H][W][C] (nhwc != 0) -> fp32 NCHW * scale, sample b mirrored
 * horizontally when flip[b] != 0 (flip may be NULL): the reference's random mirror + transforms.ToTensor()
 * (dataset.py:27-28,46,68-70; train_soft_intro_vae.py:379) done on the device, on the prefetch stream. */
int sivae_u8_to_f32(const unsigned char* src, float* dst, const int* flip, int B, int C, int H, int W, int nhwc,
                    float scale, sivae_stream_t stream);

/* ---- output side -----------------------------------------------------------------------------
 * generated images fp32 -> uint8 with the reference's quantisation for the FID network
 * (metrics/fid_score.py:247-249: np.clip(images * 255, 0, 255).astype(np.uint8)): dst = trunc(clamp(src*scale, 0, 255)).
 * src and dst 16-byte aligned. */
int sivae_f32_to_u8(const float* src, unsigned char* dst, size_t numel, float scale, sivae_stream_t stream);

/* ---- nn.Linear (Encoder.fc :109,:121 / Decoder.fc :146,:166) as small-M GEMMs -------------------------------------
 * x [B][K], W [N][K] (nn.Linear.weight), y [B][N], all fp32 row-major; B <= 256, K % 4 == 0, N % 4 == 0
 * (sivae_linear_supported; other shapes run as 1x1 convolutions through sivae_conv2d_fwd / _wgrad).
 * One wave per 32 output columns x a slice of the contraction (~1000 waves per call instead of cdiv(N,128) blocks),
 * fixed-order reduction of the slices -> deterministic.  relu != 0 fuses the nn.ReLU after Decoder.fc (:147). */
int sivae_linear_supported(int B, int K, int N);
size_t sivae_linear_workspace_bytes(int B, int K, int N);
int sivae_linear_fwd(const float* x, const float* w, const float* bias, float* y, int relu, int B, int K, int N,
                     void* workspace, size_t workspace_bytes, sivae_stream_t stream);
int sivae_linear_dgrad(const float* dy, const float* w, float* dx, int B, int K, int N, void* workspace,
                       size_t workspace_bytes, sivae_stream_t stream);
int sivae_linear_wgrad(const float* dy, const float* x, float* dw, int B, int K, int N, sivae_stream_t stream);

/* ---- bf16 mode (config 3 of BASELINE.json: "CelebA 128x128 ... bf16") -----------------------------------------
 * The reference trains in fp32 only (soft_intro_vae/train_soft_intro_vae.py:376-440 has no autocast / GradScaler), so
 * this mode is build-defined: activations and activation gradients are stored in bf16, the convolutions run on
 * v_mfma_f32_32x32x16_bf16 with fp32 accumulation, BatchNorm statistics / parameter gradients / master weights / Adam
 * stay fp32, the loss kernels above stay fp32 (Decoder.predict writes fp32 NCHW).
 * Activation layout ("blocked NCHW"): x[b][c/8][h][w][c%8] bf16, channel count padded to a multiple of 16 with zeros
 * (sivae_bf16_cblocks(C) 8-channel blocks).  Same nn.Conv2d / nn.BatchNorm2d / nn.LeakyReLU / nn.AvgPool2d /
 * nn.Upsample call sites as the fp32 entry points above (:51-75, :88-99, :153-159). */
int sivae_bf16_cblocks(int C);
/* fp32 NCHW <-> blocked bf16 (padded channels written as 0; `scale` multiplies on the way in) */
int sivae_bf16_from_f32_nchw(const float* src, void* dst, int B, int C, int H, int W, float scale,
                             sivae_stream_t stream);
int sivae_bf16_to_f32_nchw(const void* src, float* dst, int B, int C, int H, int W, sivae_stream_t stream);
/* kw-packed form of the RGB-side 5x5 layers (Encoder stem train_soft_intro_vae.py:88, Decoder.predict :159; C <= 3):
 * the 5 kernel columns move into the channel dimension, so the 5x5 conv over / into C channels runs as a 5-tap conv
 * (ks code 51 of the conv / weight-gradient entry points: 5 rows x 1 column) over / into 5C <= 15 channels.
 *   im2col: dst (blocked bf16, 16 channels) [kw*C + c][h][w] = src (fp32 NCHW) [c][h][w + sgn*(kw-2)], 0 outside
 *   fold:   dst (fp32 NCHW, C channels) [c][h][w] = bias[c] + sum_kw g (fp32 NCHW, 5C channels) [kw*C + c][h][w + sgn*(kw-2)]
 * sgn = +1 / -1.  (stem: im2col +1 of the image, fold -1 for its input gradient; predict: fold +1 for its output,
 * im2col -1 of its output gradient.) */
int sivae_bf16_im2col_kw5(const float* src, void* dst, int B, int C, int H, int W, int sgn, sivae_stream_t stream);
int sivae_bf16_fold_kw5(const float* g, const float* bias, float* dst, int B, int C, int H, int W, int sgn,
                        sivae_stream_t stream);
/* fp32 master weight [Co][Ci][ks][ks] -> bf16 MFMA-operand slabs; mode 0 forward, mode 1 data gradient.
 * ks is 1, 3, 5 or the code 51 = 5 rows x 1 column (weight [Co][Ci][5][1]) in all bf16 conv entry points */
size_t sivae_bf16_pack_conv_weight_bytes(int Co, int Ci, int ks, int mode);
int sivae_bf16_pack_conv_weight(const float* w, void* wp, int Co, int Ci, int ks, int mode, sivae_stream_t stream);
/* y (+)= conv(x', wp) + bias with the fusions of sivae_conv2d_fwd (producer BatchNorm+LeakyReLU prologue — 3x3 only —,
 * upsample addressing, {sum, sumsq} partials of the rounded output, accumulate).  out_f32_nchw != 0: y is float
 * [B][Co][H][W] (Co <= 32, no stats: Decoder.predict :159).  stats_partial has sivae_bf16_conv2d_num_px_tiles(B, Co, H, W, ks) rows and
 * feeds sivae_bn_stats_from_conv unchanged.  The data gradient is this function on dy with the mode-1 pack. */
int sivae_bf16_conv2d_num_px_tiles(int B, int Co, int H, int W, int ks);
int sivae_bf16_conv2d_fwd(const void* x, const void* wp, void* y, const float* bias, const float* pro_mean,
                          const float* pro_invstd, const float* pro_gamma, const float* pro_beta, float pro_slope,
                          float* stats_partial, int B, int Ci, int Co, int H, int W, int ks, int upsample,
                          int accumulate, int out_f32_nchw, sivae_stream_t stream);
/* split-K form for small grids (the 512-channel 8x8 / 4x4 layers: 64-256 blocks each walking 32 chunks): the
 * input-channel range is cut into sivae_bf16_conv2d_splitk(...) slices (1: not split) written as fp32 partials into
 * `workspace` and summed in a fixed order by a reduce kernel that rounds to bf16 and leaves per-image {sum, sumsq} rows
 * (stats_partial has sivae_bf16_conv2d_splitk_stats_rows(...) rows).  3x3, no bias, bf16 output only. */
int sivae_bf16_conv2d_splitk(int B, int Ci, int Co, int H, int W, int ks);
size_t sivae_bf16_conv2d_splitk_workspace_bytes(int B, int Ci, int Co, int H, int W, int ks);
int sivae_bf16_conv2d_splitk_stats_rows(int B, int Ci, int Co, int H, int W, int ks);
int sivae_bf16_conv2d_fwd_splitk(const void* x, const void* wp, void* y, const float* pro_mean,
                                 const float* pro_invstd, const float* pro_gamma, const float* pro_beta, float pro_slope,
                                 float* stats_partial, int B, int Ci, int Co, int H, int W, int ks, int upsample,
                                 int accumulate, void* workspace, size_t workspace_bytes, sivae_stream_t stream);
/* y_half [B][Co][H/2][W/2] (blocked bf16) (+)= the 2x2 block sums of conv3x3(x): the data gradient of a 3x3 conv that read
 * its input through nearest-2x upsample addressing, the adjoint of the nn.Upsample (train_soft_intro_vae.py:155) folded
 * into the epilogue (four fp32 accumulators summed, one rounding; the full-resolution gradient is never written).  wp: the
 * mode-1 pack; H, W even; never split-K (the caller keeps the two-launch form where sivae_bf16_conv2d_splitk(...) > 1). */
int sivae_bf16_conv2d_fwd_pool(const void* x, const void* wp, void* y_half, int B, int Ci, int Co, int H, int W,
                               int accumulate, sivae_stream_t stream);
/* dw [Co][Ci][ks][ks] fp32 = weight gradient (aten::convolution_backward, weight half); x' as above (prologue 3x3
 * only, upsample addressing); deterministic two-pass reduction over pixel slices */
size_t sivae_bf16_conv2d_wgrad_workspace_bytes(int B, int Ci, int Co, int H, int W, int ks);
int sivae_bf16_conv2d_wgrad(const void* x, const void* dy, float* dw, const float* pro_mean, const float* pro_invstd,
                            const float* pro_gamma, const float* pro_beta, float pro_slope, int B, int Ci, int Co,
                            int H, int W, int ks, int upsample, void* workspace, size_t workspace_bytes,
                            sivae_stream_t stream);
/* y = LeakyReLU((x-mean)*invstd*gamma+beta + res): res NULL, same shape, or (res_up) [B][C][H/2][W/2] read through
 * nearest-2x addressing; y and/or y_pool = AvgPool2d(2)(y) are written (either may be NULL, not both).  sign_mask (may be
 * NULL): sivae_bf16_bn_signmask_bytes(...) bytes, one per 8-channel pixel vector, bit e = (output e > 0) — the backward
 * reads it (1/16 of the tensor) instead of the saved output, and a pooled block need not write its full-resolution y. */
size_t sivae_bf16_bn_signmask_bytes(int B, int C, int H, int W);
int sivae_bf16_bn_apply_act(const void* x, const void* res, int res_up, const float* mean, const float* invstd,
                            const float* gamma, const float* beta, float slope, void* y, void* y_pool,
                            unsigned char* sign_mask, int B, int C, int H, int W, sivae_stream_t stream);
/* backward of the above: dy (or, dy_pooled, the gradient of y_pool); activation sign from sign_mask, else from y, else —
 * both NULL — recomputed from x (no residual; needs beta); dx, dz = gradient of the residual branch (NULL to skip;
 * dz_sum: its 2x2 block sums [B][C][H/2][W/2]), dgamma / dbeta (NULL to skip) */
size_t sivae_bf16_bn_bwd_workspace_bytes(int B, int C, int H, int W);
int sivae_bf16_bn_bwd(const void* dy, int dy_pooled, const void* y, const unsigned char* sign_mask, const void* x,
                      const float* mean, const float* invstd, const float* gamma, const float* beta, float slope,
                      void* dx, void* dz, int dz_sum, float* dgamma, float* dbeta, int B, int C, int H, int W,
                      void* workspace, size_t workspace_bytes, sivae_stream_t stream);
/* The same backward as ONE launch that reads dy and x once (bf16_bn_fused.hip, round 4; the bf16 counterpart of
 * sivae_bn_bwd_fused): raw vectors held in registers across a grid barrier, or a barrier-free launch where a channel
 * block's plane set fits one block.  Power-of-two maps (sivae_bf16_bn_bwd_fused_supported); `state` = the barrier state
 * of sivae_bn_bwd_fused (sivae_bn_bwd_fused_state_uints() unsigned ints, zeroed once, one buffer per stream). */
int sivae_bf16_bn_bwd_fused_supported(int B, int C, int H, int W);
size_t sivae_bf16_bn_bwd_fused_workspace_bytes(int B, int C, int H, int W);
int sivae_bf16_bn_bwd_fused(const void* dy, int dy_pooled, const void* y, const unsigned char* sign_mask, const void* x,
                            const float* mean, const float* invstd, const float* gamma, const float* beta, float slope,
                            void* dx, void* dz, int dz_sum, float* dgamma, float* dbeta, int B, int C, int H, int W,
                            unsigned int* state, void* workspace, size_t workspace_bytes, sivae_stream_t stream);
/* SEGMENTED batches of the bf16 mode (see "segmented batches" below: B = nseg * seg_images images laid end to end, mean /
 * invstd [nseg][C], gamma / beta [C], dgamma / dbeta summed over the passes in pass order).  The bf16 convolutions have
 * no BatchNorm prologue in the default configuration (functional16.MATERIALIZE_H), their statistics rows are in image
 * order, and the weight gradients sum over the whole batch — so the two BatchNorm kernels are all that needs a _seg form
 * for the pass pairs of train_soft_intro_vae.py:567-568, :601-608 to run as one launch per layer.  seg_images == B is the
 * plain form (the entry points above forward to these). */
int sivae_bf16_bn_apply_act_seg(const void* x, const void* res, int res_up, const float* mean, const float* invstd,
                                const float* gamma, const float* beta, float slope, void* y, void* y_pool,
                                unsigned char* sign_mask, int B, int C, int H, int W, int seg_images,
                                sivae_stream_t stream);
int sivae_bf16_bn_bwd_fused_seg_supported(int B, int C, int H, int W, int seg_images);
size_t sivae_bf16_bn_bwd_fused_seg_workspace_bytes(int B, int C, int H, int W, int seg_images);
int sivae_bf16_bn_bwd_fused_seg(const void* dy, int dy_pooled, const void* y, const unsigned char* sign_mask,
                                const void* x, const float* mean, const float* invstd, const float* gamma,
                                const float* beta, float slope, void* dx, void* dz, int dz_sum, float* dgamma,
                                float* dbeta, int B, int C, int H, int W, int seg_images, unsigned int* state,
                                void* workspace, size_t workspace_bytes, sivae_stream_t stream);
int sivae_bf16_upsample2_fwd(const void* x, void* y, int B, int C, int Hs, int Ws, sivae_stream_t stream);
int sivae_bf16_upsample2_bwd(const void* dy, void* dx, int B, int C, int Hs, int Ws, sivae_stream_t stream);
int sivae_bf16_add_inplace(void* y, const void* x, size_t nvec, sivae_stream_t stream);

/* ---- segmented batches: several passes of a network through the SAME weights as one launch ---------
 * The reference's iteration runs pairs of independent passes through unchanged weights — model(rec.detach()) /
 * model(fake.detach()) train_soft_intro_vae.py:567-568, encode(rec) / encode(fake) :601-605, decode(z_rec) /
 * decode(z_fake) :607-608, bootstrap decode_target x2 soft_intro_vae_bootstrap/train_soft_intro_vae_bootstrap.py:635-636.
 * The engine lays such passes end to end in ONE batch of B = nseg * seg_images images ("segments", pass g = images
 * [g*seg_images, (g+1)*seg_images)).  Convolutions do not care; training-mode nn.BatchNorm2d (:58,:62,:90) must keep
 * ONE set of batch statistics PER PASS, so every BatchNorm kernel and every fused BatchNorm prologue has a _seg form:
 * mean / invstd are [nseg][C], gamma / beta / running buffers [C]; dgamma / dbeta are summed over the passes; the
 * running buffers receive one momentum update per pass in pass order.  nseg = 1 is the unsegmented op bit for bit. */
int sivae_bn_stats_from_conv_seg(const float* partials, int n_tiles, int nseg, int seg_rev, int B_seg, int C, int HW,
                                 float eps, float momentum, float* running_mean, float* running_var,
                                 long long* num_batches_tracked, float* mean_out, float* invstd_out,
                                 sivae_stream_t stream);
int sivae_bn_update_running_seg(const float* mean, const float* invstd, int nseg, int seg_rev, int C, double count,
                                float eps, float momentum, float* running_mean, float* running_var,
                                long long* num_batches_tracked, sivae_stream_t stream);
/* sivae_bn_stats_from_conv_seg with a scratch buffer: from 2048 partial rows per pass on (the 256x256 / 128x128 layers at
 * batch 128) the rows are folded in two coalesced stages instead of one strided walk per channel; same statistics, same
 * running-buffer updates (nn.BatchNorm2d training-mode forward, train_soft_intro_vae.py:57,62,90).  The workspace size is 0
 * (and the pointer may be NULL) where the one-stage form is used. */
size_t sivae_bn_stats_from_conv_workspace_bytes(int n_tiles, int nseg, int C);
int sivae_bn_stats_from_conv_ws(const float* partials, int n_tiles, int nseg, int seg_rev, int B_seg, int C, int HW,
                                float eps, float momentum, float* running_mean, float* running_var,
                                long long* num_batches_tracked, float* mean_out, float* invstd_out, void* workspace,
                                size_t workspace_bytes, sivae_stream_t stream);
int sivae_bn_apply_act_seg(const float* x, const float* res, int res_up, const float* mean, const float* invstd,
                           const float* gamma, const float* beta, float slope, float* y, float* y_pooled, int B, int C,
                           int H, int W, int seg_images, sivae_stream_t stream);
int sivae_bn_apply_act_signmask_seg(const float* x, const float* res, int res_up, const float* mean,
                                    const float* invstd, const float* gamma, const float* beta, float slope, float* y,
                                    float* y_pooled, unsigned char* mask, int B, int C, int H, int W, int seg_images,
                                    sivae_stream_t stream);
/* every backward variant in one entry: act_mode 0 none / 1 sign from y / 2 recomputed from x (beta) / 3 from mask;
 * dy_pooled, dz_sum as in sivae_bn_bwd_signmask; workspace: sivae_bn_workspace_bytes(seg_images, nseg * C, H * W).
 * counters: NULL, or >= C zero-initialised unsigned ints the call leaves zero (caller-owned, one stream at a time): the
 * per-channel finalize then runs inside the reduction kernel instead of as its own launch — same fixed summation order */
int sivae_bn_bwd_seg(const float* dy, const float* y, const unsigned char* mask, const float* x, const float* mean,
                     const float* invstd, const float* gamma, const float* beta, int act_mode, float slope, float* dx,
                     float* dz_out, float* dgamma, float* dbeta, int B, int C, int H, int W, int dy_pooled, int dz_sum,
                     int seg_images, unsigned int* counters, void* workspace, size_t workspace_bytes,
                     sivae_stream_t stream);
/* The same backward as ONE persistent launch that reads dy and x once and writes dx once (bn_fused.hip, round 4): the
 * chip's register files hold the activations between the reduction phase and the dx phase, (segment, channel) plane sets are
 * walked in groups separated by a grid barrier, per-channel sums are folded in a fixed order (deterministic).  Replaces
 * the backward of nn.BatchNorm2d + nn.LeakyReLU (+ torch.add) of train_soft_intro_vae.py:57-63,71-74,90-91 like
 * sivae_bn_bwd_seg, with the same argument meaning.  `state`: sivae_bn_bwd_fused_state_uints() unsigned ints, zeroed ONCE
 * by the caller and then left consistent by every call (one buffer per stream).  Power-of-two maps only
 * (sivae_bn_bwd_fused_supported); the grid (2 blocks per CU) must be fully resident: not to be run concurrently with
 * another persistent kernel on the device. */
int sivae_bn_bwd_fused_supported(int B, int C, int H, int W, int seg_images);
size_t sivae_bn_bwd_fused_workspace_bytes(int B, int C, int H, int W, int seg_images);
int sivae_bn_bwd_fused_state_uints(void);
/* Grid-barrier timeout (round 5): a persistent launch whose grid is not fully resident no longer traps — after
 * SIVAE_BN_FUSED_SPIN_LIMIT polls (default 2^24, tens of seconds) the waiting block sets word
 * sivae_bn_bwd_fused_poison_word() of `state` and every block leaves the kernel (that launch's outputs are garbage, later
 * launches on the same state return at once).  The host reads that word where it reads results back anyway, raises, and
 * zeroes the whole state (sivae_hip.ops.bn_fused_check).  Plans also refuse the persistent form under a CU mask
 * (HSA_CU_MASK / ROC_GLOBAL_CU_MASK), with SIVAE_BN_FUSED_PERSISTENT=0, or when the runtime reports fewer than two
 * resident blocks per CU: sivae_bn_bwd_fused_supported() then returns 0 for plane sets that need the barrier and the
 * callers keep the three-launch form.  (The timeout path is exercised without a hook in this library: the test
 * corrupts the caller-owned `state` so that the arrivals never add up and shortens the spin limit for that call; the
 * resource-holding "squatter" kernel of the co-residency tests lives in tests/support/, not here.) */
int sivae_bn_bwd_fused_poison_word(void);
int sivae_bn_bwd_fused(const float* dy, const float* y, const unsigned char* mask, const float* x, const float* mean,
                       const float* invstd, const float* gamma, const float* beta, int act_mode, float slope, float* dx,
                       float* dz_out, float* dgamma, float* dbeta, int B, int C, int H, int W, int dy_pooled, int dz_sum,
                       int seg_images, unsigned int* state, void* workspace, size_t workspace_bytes,
                       sivae_stream_t stream);
/* Winograd 3x3 forward / data gradient and weight gradient with a segmented BatchNorm prologue (pro_* may be NULL: then
 * only the row order of stats_partial matters — image order, so rows [g*n/nseg, (g+1)*n/nseg) are pass g).  On 8x8 /
 * 4x4 maps seg_images must be a multiple of 2 / 4 (a tile block holds that many images). */
int sivae_conv2d_wino_fwd_seg(const float* x, const float* up, float* y, const float* pro_mean,
                              const float* pro_invstd, const float* pro_gamma, const float* pro_beta, float pro_slope,
                              float* stats_partial, int B, int Ci, int Co, int H, int W, int upsample, int accumulate,
                              int seg_images, sivae_stream_t stream);
int sivae_conv2d_wino_fwd_splitk_seg(const float* x, const float* up, float* y, const float* pro_mean,
                                     const float* pro_invstd, const float* pro_gamma, const float* pro_beta,
                                     float pro_slope, float* stats_partial, int B, int Ci, int Co, int H, int W,
                                     int upsample, int accumulate, int seg_images, void* workspace,
                                     size_t workspace_bytes, sivae_stream_t stream);
int sivae_conv2d_wino_wgrad_seg(const float* x, const float* dy, float* dw, const float* pro_mean,
                                const float* pro_invstd, const float* pro_gamma, const float* pro_beta, float pro_slope,
                                int B, int Ci, int Co, int H, int W, int upsample, int seg_images, void* workspace,
                                size_t workspace_bytes, sivae_stream_t stream);

/* ---- Winograd F(4x4,3x3) for the large-map 3x3 convs (nn.Conv2d k=3, train_soft_intro_vae.py:56-61) ----
 * 36 multiplies per 4x4 output tile: 1.78x fewer matrix-pipe passes than F(2x2,3x3) (sivae_conv2d_wino_fwd) at 1.2e-5
 * relative error per layer in fp32 (3.3e-5 on the reconstruction of the six-level network end to end).  Maps: H % 16 == 0,
 * W % 32 == 0.  up: [6][Ci_pad][Co_pad][6] from sivae_pack_wino4_weight (mode 0 forward, 1 data gradient).
 * stats_partial: [sivae_conv2d_wino4_num_px_tiles][Co][2] rows in image order (sivae_bn_stats_from_conv[_seg]). */
size_t sivae_pack_wino4_weight_bytes(int Co, int Ci, int mode);
int sivae_pack_wino4_weight(const float* w, float* up, int Co, int Ci, int mode, sivae_stream_t stream);
int sivae_conv2d_wino4_supported(int H, int W); /* 1: H % 16 == 0, W % 32 == 0; 2 / 3 / 4: 16 x 16 / 8 x 8 / 4 x 4 maps — a
                                                     work item (32 x 16 pixels) is a grid of 2 x 1 / 4 x 2 / 8 x 4 whole
                                                     images (round 6: the 8 x 8 and 4 x 4 maps of the deep 512-channel
                                                     blocks, res_in_8 / res_in_4 of train_soft_intro_vae.py:100-103,
                                                     153-158), every seam zero padding: B and seg_images must be multiples
                                                     of sivae_conv2d_wino4_images_per_item; stats rows per item; 0 otherwise */
int sivae_conv2d_wino4_images_per_item(int H, int W); /* 1, 2, 8, 32 for modes 1..4; 0: unsupported map */
/* Round 6: the data gradient of conv3x3(Upsample2(x)) with respect to the low-resolution x (conv1 behind nn.Upsample,
 * train_soft_intro_vae.py:155,56) in one F(4x4,3x3) pass: dx[B][N][H/2][W/2] (+)= the 2x2 block sums of conv3x3^T(dy[B][C][H][W]),
 * the block sum folded into the output transform.  up: sivae_pack_wino4_weight(w[C][N][3][3], mode 1).  H % 16 == 0, W % 32 == 0.
 * `_pays`: N <= 64 and a work item for every CU (there the phase-folded sivae_conv2d_wino_up_dgrad splits K over wave pairs). */
int sivae_conv2d_wino4_dgrad_pool_pays(int B, int C, int N, int H, int W);
int sivae_conv2d_wino4_dgrad_pool(const float* dy, const float* up, float* dx, int B, int C, int N, int H, int W,
                                  int accumulate, sivae_stream_t stream);
/* modes 3 / 4 only: 1 when the image-grid launch (with its split-K plan) is expected to beat F(2x2,3x3) — enough
 * (slices x items) for every CU and a long enough K slice (measured: profiles/r6_wino4_small_maps_vs_f23.txt) */
int sivae_conv2d_wino4_small_pays(int B, int Ci, int Co, int H, int W);
int sivae_conv2d_wino4_pays(int B, int Ci, int Co, int H, int W); /* supported AND >= one work item per CU */
int sivae_conv2d_wino4_num_px_tiles(int B, int H, int W);
int sivae_conv2d_wino4_fwd(const float* x, const float* up, float* y, float* stats_partial, int B, int Ci, int Co, int H,
                           int W, int accumulate, sivae_stream_t stream);
/* the same with the producer BatchNorm2d + LeakyReLU fused into the input read (:58-59; pro_* as in sivae_conv2d_fwd);
 * seg_images > 0: segmented batch with pro_mean / pro_invstd [B / seg_images][Ci]; segments * padded Ci <= 1024 */
int sivae_conv2d_wino4_fwd_pro(const float* x, const float* up, float* y, const float* pro_mean, const float* pro_invstd,
                               const float* pro_gamma, const float* pro_beta, float pro_slope, float* stats_partial,
                               int B, int Ci, int Co, int H, int W, int accumulate, int seg_images,
                               sivae_stream_t stream);

/* split-K form of the F(4x4,3x3) forward / data gradient for launches with fewer work items than CUs (the deep layers of
 * the per-GPU shards): sivae_conv2d_wino4_splitk -> number of K slices S (1 = the plain kernel); for S > 1 the partial
 * outputs go through `workspace` ([S][B][Co][H][W]), are summed in a fixed order, and stats_partial is [B][Co][2] (per
 * image) instead of per pixel tile.  pro_mean may be NULL (no prologue); seg_images as in sivae_conv2d_wino4_fwd_pro. */
int sivae_conv2d_wino4_splitk(int B, int Ci, int Co, int H, int W);
size_t sivae_conv2d_wino4_splitk_workspace_bytes(int B, int Ci, int Co, int H, int W);
int sivae_conv2d_wino4_fwd_splitk(const float* x, const float* up, float* y, const float* pro_mean, const float* pro_invstd,
                                  const float* pro_gamma, const float* pro_beta, float pro_slope, float* stats_partial, int B,
                                  int Ci, int Co, int H, int W, int accumulate, int seg_images, void* workspace,
                                  size_t workspace_bytes, sivae_stream_t stream);

/* The same F(4x4,3x3) forward / data gradient with its 36 frequency GEMMs on the BF16 matrix pipe and FP32-EXACT products
 * (conv_wino4_b6.hip, round 5): every fp32 operand is split into three bf16 pieces by truncation (exact), a product is the
 * fp32 sum of six of the nine piece products (the dropped terms are <= 2^-24 of it) — the accuracy of v_mfma_f32_32x32x2_f32
 * (profiles/r4_probe_bf16x6_accuracy.txt), at 1/2.7 of its matrix cycles.  Replaces the same nn.Conv2d(k=3, s=1, p=1)
 * (soft_intro_vae/train_soft_intro_vae.py:56-61) with the same argument meaning as sivae_conv2d_wino4_fwd_pro /
 * sivae_conv2d_wino4_fwd_splitk (pro_mean may be NULL; no limit on segments x channels: the prologue parameters are read
 * through scalar loads); `up` = the pre-split operand of sivae_pack_wino4_b6_weight ([j][Ci_pad/16][Co_pad/32][i][piece]
 * blocks of 64 lanes x 8 bf16, 1.5x the bytes of the fp32 pack); maps: modes 1 and 2 of sivae_conv2d_wino4_supported; the split-K plan
 * and workspace are those of sivae_conv2d_wino4_splitk / _splitk_workspace_bytes. */
size_t sivae_pack_wino4_b6_weight_bytes(int Co, int Ci, int mode);
int sivae_pack_wino4_b6_weight(const float* w, void* up, int Co, int Ci, int mode, sivae_stream_t stream);
int sivae_conv2d_wino4_b6_fwd(const float* x, const void* up, float* y, const float* pro_mean, const float* pro_invstd,
                              const float* pro_gamma, const float* pro_beta, float pro_slope, float* stats_partial, int B,
                              int Ci, int Co, int H, int W, int accumulate, int seg_images, sivae_stream_t stream);
int sivae_conv2d_wino4_b6_fwd_splitk(const float* x, const void* up, float* y, const float* pro_mean,
                                     const float* pro_invstd, const float* pro_gamma, const float* pro_beta,
                                     float pro_slope, float* stats_partial, int B, int Ci, int Co, int H, int W,
                                     int accumulate, int seg_images, void* workspace, size_t workspace_bytes,
                                     sivae_stream_t stream);

/* Winograd F(4x4,3x3) weight gradient (conv_wino4_wgrad.hip) — the weight half of aten::convolution_backward of the
 * nn.Conv2d(k=3) layers (soft_intro_vae/train_soft_intro_vae.py:56-61) on maps with H % 4 == 0, W % 16 == 0:
 * dw[Co][Ci][3][3] from x [B][Ci][H][W] (or, pro_mean != NULL, LeakyReLU(BatchNorm(x)) recomputed on load; per-segment
 * statistics [nseg][Ci] when seg_images > 0, nseg = B / seg_images <= 2) and dy [B][Co][H][W]; x / dy 16-byte aligned.
 * `pays`: supported AND enough stages for one block per CU.  Round 6: also the 8 x 8 and 4 x 4 maps (the deep 512-channel
 * blocks): a stage's 4 x 16 pixel strip is then 2 / 4 whole images side by side (every seam zero padding) and B and
 * seg_images must be multiples of sivae_conv2d_wino4_wgrad_images_per_stage (1 for W >= 16). */
int sivae_conv2d_wino4_wgrad_supported(int H, int W);
int sivae_conv2d_wino4_wgrad_images_per_stage(int H, int W);
int sivae_conv2d_wino4_wgrad_pays(int B, int Ci, int Co, int H, int W);
size_t sivae_conv2d_wino4_wgrad_workspace_bytes(int B, int Ci, int Co, int H, int W);
int sivae_conv2d_wino4_wgrad(const float* x, const float* dy, float* dw, const float* pro_mean, const float* pro_invstd,
                             const float* pro_gamma, const float* pro_beta, float pro_slope, int B, int Ci, int Co, int H,
                             int W, int seg_images, void* workspace, size_t workspace_bytes, sivae_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SIVAE_HIP_H */
