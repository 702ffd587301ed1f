// Training-mode BatchNorm2d (+ LeakyReLU, + residual add) for NCHW fp32 on gfx950.
// All kernels are HBM-bound streaming kernels: float4 loads along the contiguous H*W axis, per-thread
// fp64 accumulation, wave64 butterfly (`__shfl_xor`) + LDS block reduction, one partial per block,
// and a tiny per-channel finalize that combines partials in a FIXED order in fp64 (reproducible).
//
// Reference ops being replaced: nn.BatchNorm2d (eps 1e-5, momentum 0.1, affine) + nn.LeakyReLU(0.2)
// + torch.add as used in ResidualBlock / Encoder stem,
// soft_intro_vae/train_soft_intro_vae.py:58-63,71-74,90-91.
#include "common.h"

namespace {

// One channel's data is B runs of HW contiguous floats.  A "slice" is a contiguous range of the
// channel-local index n = b*HW + i.
struct SlicePlan {
  int S;          // slices per channel
  long long len;  // elements per slice (multiple of 4)
};
static SlicePlan plan_slices(long long n_per_ch, int C) {
  SlicePlan p;
  long long s1 = (n_per_ch + 8191) / 8192;
  long long s2 = 2048 / C;
  if (s2 < 1) s2 = 1;
  long long S = s1 < s2 ? s1 : s2;
  if (S < 1) S = 1;
  long long len = (n_per_ch + S - 1) / S;
  len = (len + 3) & ~3LL;
  p.S = (int)((n_per_ch + len - 1) / len);
  p.len = len;
  return p;
}

}  // namespace

// ---- LeakyReLU sign mask: 1 bit per element (set = pre-activation > 0), element e -> bit (e & 7) of byte e >> 3.
// A thread owns one float4 = one nibble; the even lane of each lane pair writes the byte (its nibble | the odd
// lane's << 4, fetched with one DPP quad_perm move — the whole wave must be active at the call).  The backward of
// "LeakyReLU(BN(c) + skip)" then reads 1/32 of a tensor for the sign instead of the saved output (and an encoder
// block, whose output is only consumed through the AvgPool2d behind it, never writes that output at all).
__device__ __forceinline__ unsigned sign_nibble(const float4 v) {
  return (v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u);
}
__device__ __forceinline__ unsigned pair_nibbles(unsigned nib) {
  const unsigned other = (unsigned)__builtin_amdgcn_update_dpp(0, (int)nib, 0xB1, 0xf, 0xf, true);  // lane ^ 1
  return nib | (other << 4);
}
__device__ __forceinline__ unsigned load_nibble(const unsigned char* __restrict__ mask, size_t i4) {
  return ((unsigned)mask[i4 >> 1] >> (unsigned)((i4 & 1) * 4)) & 0xfu;
}
__device__ __forceinline__ void apply_nibble(float* gz, unsigned nib, float slope) {
  gz[0] = (nib & 1u) ? gz[0] : gz[0] * slope;
  gz[1] = (nib & 2u) ? gz[1] : gz[1] * slope;
  gz[2] = (nib & 4u) ? gz[2] : gz[2] * slope;
  gz[3] = (nib & 8u) ? gz[3] : gz[3] * slope;
}

// ------------------------------------------------------------------------------------------------
// forward statistics
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) bn_stats_partial_kernel(const float* __restrict__ x, double* __restrict__ part,
                                                               int C, int HW, long long n_per_ch,
                                                               long long slice_len, int S) {
  __shared__ double red[4];
  const int c = blockIdx.x, s = blockIdx.y;
  const long long n0 = (long long)s * slice_len;
  long long n1 = n0 + slice_len;
  if (n1 > n_per_ch) n1 = n_per_ch;
  double sum = 0.0, sq = 0.0;
  if ((HW & 3) == 0) {
    for (long long n = n0 + (long long)threadIdx.x * 4; n < n1; n += 1024) {
      const long long b = n / HW;
      const int i = (int)(n - b * HW);
      const float4 v = *reinterpret_cast<const float4*>(x + ((size_t)b * C + c) * HW + i);
      sum += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
      sq += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
  } else {
    for (long long n = n0 + threadIdx.x; n < n1; n += 256) {
      const long long b = n / HW;
      const int i = (int)(n - b * HW);
      const float v = x[((size_t)b * C + c) * HW + i];
      sum += (double)v;
      sq += (double)v * v;
    }
  }
  sum = block_sum<256>(sum, red);
  sq = block_sum<256>(sq, red);
  if (threadIdx.x == 0) {
    part[((size_t)c * S + s) * 2 + 0] = sum;
    part[((size_t)c * S + s) * 2 + 1] = sq;
  }
}

// partials from bn_stats_partial_kernel, layout [C][S][2] doubles (S <= 2048 / C)
__global__ void __launch_bounds__(64) bn_finalize_kernel(const double* __restrict__ part, int S, int C,
                                                         double count, float eps,
                                                         float momentum, float* running_mean,
                                                         float* running_var, long long* num_batches_tracked,
                                                         float* __restrict__ mean_out,
                                                         float* __restrict__ invstd_out) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
  if (c >= C) return;
  double sum = 0.0, sq = 0.0;
  for (int s = 0; s < S; ++s) {
    const size_t o = ((size_t)c * S + s) * 2;
    sum += part[o];
    sq += part[o + 1];
  }
  const double mean = sum / count;
  double var = sq / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  mean_out[c] = (float)mean;
  invstd_out[c] = invstd;
  if (running_mean) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
    running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
  }
}

// Finalize for the conv-epilogue partials ([S][C][2] floats, S = number of pixel tiles, can be 10^4+):
// one block per channel (NT = 1024 threads when there are thousands of tiles, so a channel's strided walk is
// spread over 16 waves), threads stride over tiles, fixed-shape fp64 tree -> deterministic.
template <int NT>
__global__ void __launch_bounds__(NT) bn_finalize_conv_kernel(const float* __restrict__ part, int S, int C,
                                                               double count, float eps, float momentum,
                                                               float* running_mean, float* running_var,
                                                               long long* num_batches_tracked,
                                                               float* __restrict__ mean_out,
                                                               float* __restrict__ invstd_out, int nseg, int seg_rev) {
  // nseg > 1: the batch is nseg independent passes of the network laid end to end (same weights, one launch — see
  // functional.py "segments"); rows [g*S, (g+1)*S) of `part` belong to pass g, every pass gets its own statistics
  // (mean_out / invstd_out are [nseg][C]) and the running buffers receive one momentum update per pass, in pass order
  // (seg_rev: last segment first — the order in which the reference would have run the passes).
  __shared__ double red[NT / 64];
  const int c = blockIdx.x;
  for (int gi = 0; gi < nseg; ++gi) {
    const int g = seg_rev ? nseg - 1 - gi : gi;
    double sum = 0.0, sq = 0.0;
    for (int s = threadIdx.x; s < S; s += NT) {
      const float2 v = *reinterpret_cast<const float2*>(part + ((size_t)(g * S + s) * C + c) * 2);
      sum += (double)v.x;
      sq += (double)v.y;
    }
    sum = block_sum<NT>(sum, red);
    sq = block_sum<NT>(sq, red);
    if (threadIdx.x == 0) {
      if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
      const double mean = sum / count;
      double var = sq / count - mean * mean;
      if (var < 0.0) var = 0.0;
      mean_out[g * C + c] = (float)mean;
      invstd_out[g * C + c] = (float)(1.0 / sqrt(var + (double)eps));
      if (running_mean) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
      }
    }
    __syncthreads();
  }
}

extern "C" size_t sivae_bn_workspace_bytes(int B, int C, int HW) {
  if (B <= 0 || C <= 0 || HW <= 0) return 0;
  SlicePlan p = plan_slices((long long)B * HW, C);
  // forward: [C][S][2] doubles; backward needs the same + [C][2] coefficients
  return ((size_t)C * p.S * 2 + (size_t)C * 2) * sizeof(double);
}

extern "C" int sivae_bn_stats(const float* x, int B, int C, int HW, float eps, float momentum,
                              float* running_mean, float* running_var, long long* num_batches_tracked,
                              float* mean_out, float* invstd_out, void* workspace, size_t workspace_bytes,
                              hipStream_t stream) {
  if (!x || !mean_out || !invstd_out) return SIVAE_ERR_NULL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return SIVAE_ERR_NULL;
  if (B <= 0 || C <= 0 || HW <= 0) return SIVAE_ERR_SHAPE;
  if (!workspace || workspace_bytes < sivae_bn_workspace_bytes(B, C, HW)) return SIVAE_ERR_WORKSPACE;
  const long long n = (long long)B * HW;
  SlicePlan p = plan_slices(n, C);
  double* part = (double*)workspace;
  hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(C, p.S), dim3(256), 0, stream, x, part, C, HW, n, p.len, p.S);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 64)), dim3(64), 0, stream, (const double*)part, p.S, C,
                     (double)n, eps, momentum, running_mean, running_var, num_batches_tracked, mean_out,
                     invstd_out);
  return sivae_launch_status();
}

// Statistics from the per-pixel-tile partial sums the conv forward epilogue wrote ([n_tiles][C][2] floats).
static int bn_stats_from_conv_impl(const float* partials, int n_tiles, int nseg, int seg_rev, int B, int C, int HW,
                                   float eps, float momentum, float* running_mean, float* running_var,
                                   long long* num_batches_tracked, float* mean_out, float* invstd_out,
                                   hipStream_t stream) {
  if (!partials || !mean_out || !invstd_out) return SIVAE_ERR_NULL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return SIVAE_ERR_NULL;
  if (B <= 0 || C <= 0 || HW <= 0 || n_tiles <= 0 || nseg <= 0 || (n_tiles % nseg) != 0) return SIVAE_ERR_SHAPE;
  const int S = n_tiles / nseg;
  if (S >= 4096)
    hipLaunchKernelGGL((bn_finalize_conv_kernel<1024>), dim3(C), dim3(1024), 0, stream, partials, S, C,
                       (double)B * HW, eps, momentum, running_mean, running_var, num_batches_tracked, mean_out,
                       invstd_out, nseg, seg_rev);
  else
    hipLaunchKernelGGL((bn_finalize_conv_kernel<256>), dim3(C), dim3(256), 0, stream, partials, S, C,
                       (double)B * HW, eps, momentum, running_mean, running_var, num_batches_tracked, mean_out,
                       invstd_out, nseg, seg_rev);
  return sivae_launch_status();
}

// ---- two-stage form for MANY partial rows (thousands of pixel tiles: the 256x256 / 128x128 layers at batch 128): the
// one-block-per-channel walk above reads 8 of every 64 bytes it touches (a row holds all channels) and runs C blocks —
// 81 us per call on [32768][64][2], 1.9 ms per headline iteration.  Stage 1 reads whole rows (consecutive threads =
// consecutive channels) in chunks of BNC_ROWS rows, fp64 per thread, fixed-order fold over the row lanes through LDS ->
// [nseg][C][nchunks][2] doubles; stage 2 is one thread per channel over the chunks (+ the running-buffer updates in
// pass order, as above).
#define BNC_ROWS 128
__global__ void __launch_bounds__(256) bn_conv_rows_partial_kernel(const float* __restrict__ part, int S, int C,
                                                                   int nchunks, double* __restrict__ out) {
  __shared__ double red[2][256];
  const int chunk = blockIdx.x, g = blockIdx.y, t = threadIdx.x;
  int cw = 1;
  while (cw < C && cw < 256) cw <<= 1;  // channel lanes (power of two <= 256), the other 256 / cw lanes walk rows
  const int rp = 256 / cw, tx = t & (cw - 1), ty = t / cw;
  const int r0 = chunk * BNC_ROWS;
  const int r1 = r0 + BNC_ROWS < S ? r0 + BNC_ROWS : S;
  for (int c0 = 0; c0 < C; c0 += cw) {
    const int c = c0 + tx;
    double sum = 0.0, sq = 0.0;
    if (c < C)
      for (int r = r0 + ty; r < r1; r += rp) {
        const float2 v = *reinterpret_cast<const float2*>(part + ((size_t)(g * S + r) * C + c) * 2);
        sum += (double)v.x;
        sq += (double)v.y;
      }
    red[0][t] = sum;
    red[1][t] = sq;
    __syncthreads();
    if (ty == 0 && c < C) {
      double a = 0.0, b = 0.0;
      for (int k = 0; k < rp; ++k) {
        a += red[0][k * cw + tx];
        b += red[1][k * cw + tx];
      }
      double* o = out + (((size_t)g * C + c) * nchunks + chunk) * 2;
      o[0] = a;
      o[1] = b;
    }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(64) bn_conv_rows_finalize_kernel(const double* __restrict__ part2, int nchunks, int C,
                                                                   double count, float eps, float momentum,
                                                                   float* running_mean, float* running_var,
                                                                   long long* num_batches_tracked,
                                                                   float* __restrict__ mean_out,
                                                                   float* __restrict__ invstd_out, int nseg,
                                                                   int seg_rev) {
  // one wave per channel: lanes stride over the chunks, then the fixed butterfly
  const int c = blockIdx.x, lane = threadIdx.x;
  if (c == 0 && lane == 0 && num_batches_tracked) *num_batches_tracked += nseg;
  for (int gi = 0; gi < nseg; ++gi) {
    const int g = seg_rev ? nseg - 1 - gi : gi;
    const double* p = part2 + ((size_t)g * C + c) * nchunks * 2;
    double sum = 0.0, sq = 0.0;
    for (int k = lane; k < nchunks; k += 64) {
      sum += p[2 * k];
      sq += p[2 * k + 1];
    }
    sum = wave_sum(sum);
    sq = wave_sum(sq);
    if (lane == 0) {
      const double mean = sum / count;
      double var = sq / count - mean * mean;
      if (var < 0.0) var = 0.0;
      mean_out[g * C + c] = (float)mean;
      invstd_out[g * C + c] = (float)(1.0 / sqrt(var + (double)eps));
      if (running_mean) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
      }
    }
  }
}

// workspace of sivae_bn_stats_from_conv_ws (0: the call does not use one for this shape)
extern "C" size_t sivae_bn_stats_from_conv_workspace_bytes(int n_tiles, int nseg, int C) {
  if (n_tiles <= 0 || nseg <= 0 || C <= 0 || (n_tiles % nseg) != 0) return 0;
  const int S = n_tiles / nseg;
  if (S < 2048) return 0;
  return (size_t)nseg * C * cdiv(S, BNC_ROWS) * 2 * sizeof(double);
}

// sivae_bn_stats_from_conv_seg with a scratch buffer: from 2048 partial rows per pass on, the statistics are folded in
// two coalesced stages (above); below that this IS sivae_bn_stats_from_conv_seg (the workspace may then be NULL)
extern "C" int sivae_bn_stats_from_conv_ws(const float* partials, int n_tiles, int nseg, int seg_rev, int B_seg, int C,
                                           int HW, float eps, float momentum, float* running_mean, float* running_var,
                                           long long* num_batches_tracked, float* mean_out, float* invstd_out,
                                           void* workspace, size_t workspace_bytes, hipStream_t stream) {
  const size_t need = sivae_bn_stats_from_conv_workspace_bytes(n_tiles, nseg, C);
  if (need == 0)
    return bn_stats_from_conv_impl(partials, n_tiles, nseg, seg_rev, B_seg, C, HW, eps, momentum, running_mean,
                                   running_var, num_batches_tracked, mean_out, invstd_out, stream);
  if (!partials || !mean_out || !invstd_out) return SIVAE_ERR_NULL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return SIVAE_ERR_NULL;
  if (B_seg <= 0 || HW <= 0) return SIVAE_ERR_SHAPE;
  if (!workspace || workspace_bytes < need) return SIVAE_ERR_WORKSPACE;
  const int S = n_tiles / nseg, nchunks = cdiv(S, BNC_ROWS);
  double* part2 = (double*)workspace;
  hipLaunchKernelGGL(bn_conv_rows_partial_kernel, dim3(nchunks, nseg), dim3(256), 0, stream, partials, S, C, nchunks,
                     part2);
  hipLaunchKernelGGL(bn_conv_rows_finalize_kernel, dim3(C), dim3(64), 0, stream, (const double*)part2, nchunks,
                     C, (double)B_seg * HW, eps, momentum, running_mean, running_var, num_batches_tracked, mean_out,
                     invstd_out, nseg, seg_rev);
  return sivae_launch_status();
}

extern "C" int sivae_bn_stats_from_conv(const float* partials, int n_tiles, int B, int C, int HW, float eps,
                                        float momentum, float* running_mean, float* running_var,
                                        long long* num_batches_tracked, float* mean_out, float* invstd_out,
                                        hipStream_t stream) {
  return bn_stats_from_conv_impl(partials, n_tiles, 1, 0, B, C, HW, eps, momentum, running_mean, running_var,
                                 num_batches_tracked, mean_out, invstd_out, stream);
}

// Segmented form: the batch is `nseg` passes of `B_seg` images each (n_tiles rows in pass order, n_tiles % nseg == 0);
// mean_out / invstd_out are [nseg][C]; the running buffers get one update per pass (seg_rev: last pass first).
extern "C" int sivae_bn_stats_from_conv_seg(const float* partials, int n_tiles, int nseg, int seg_rev, int B_seg, int C,
                                            int HW, float eps, float momentum, float* running_mean, float* running_var,
                                            long long* num_batches_tracked, float* mean_out, float* invstd_out,
                                            hipStream_t stream) {
  return bn_stats_from_conv_impl(partials, n_tiles, nseg, seg_rev, B_seg, C, HW, eps, momentum, running_mean,
                                 running_var, num_batches_tracked, mean_out, invstd_out, stream);
}

// ---- synchronised BatchNorm (opt-in, data-parallel runs): the per-channel {sum, sumsq} of the local
// shard in fp64, to be all-reduced by the caller, and the finalize from (global) sums.
__global__ void __launch_bounds__(256) bn_sums_conv_kernel(const float* __restrict__ part, int S, int C,
                                                           double* __restrict__ sums) {
  __shared__ double red[4];
  const int c = blockIdx.x;
  double sum = 0.0, sq = 0.0;
  for (int s = threadIdx.x; s < S; s += 256) {
    const float2 v = *reinterpret_cast<const float2*>(part + ((size_t)s * C + c) * 2);
    sum += (double)v.x;
    sq += (double)v.y;
  }
  sum = block_sum<256>(sum, red);
  sq = block_sum<256>(sq, red);
  if (threadIdx.x == 0) {
    sums[c * 2 + 0] = sum;
    sums[c * 2 + 1] = sq;
  }
}

__global__ void __launch_bounds__(64) bn_finalize_sums_kernel(const double* __restrict__ sums, int C, double count,
                                                              float eps, float momentum, float* running_mean,
                                                              float* running_var, long long* num_batches_tracked,
                                                              float* __restrict__ mean_out,
                                                              float* __restrict__ invstd_out) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
  const double mean = sums[c * 2] / count;
  double var = sums[c * 2 + 1] / count - mean * mean;
  if (var < 0.0) var = 0.0;
  mean_out[c] = (float)mean;
  invstd_out[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
    running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
  }
}

extern "C" int sivae_bn_sums_from_conv(const float* partials, int n_tiles, int C, double* sums, hipStream_t stream) {
  if (!partials || !sums) return SIVAE_ERR_NULL;
  if (C <= 0 || n_tiles <= 0) return SIVAE_ERR_SHAPE;
  hipLaunchKernelGGL(bn_sums_conv_kernel, dim3(C), dim3(256), 0, stream, partials, n_tiles, C, sums);
  return sivae_launch_status();
}

extern "C" int sivae_bn_finalize_sums(const double* sums, int C, double count, float eps, float momentum,
                                      float* running_mean, float* running_var, long long* num_batches_tracked,
                                      float* mean_out, float* invstd_out, hipStream_t stream) {
  if (!sums || !mean_out || !invstd_out) return SIVAE_ERR_NULL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return SIVAE_ERR_NULL;
  if (C <= 0 || !(count > 0.0)) return SIVAE_ERR_SHAPE;
  hipLaunchKernelGGL(bn_finalize_sums_kernel, dim3(cdiv(C, 64)), dim3(64), 0, stream, sums, C, count, eps, momentum,
                     running_mean, running_var, num_batches_tracked, mean_out, invstd_out);
  return sivae_launch_status();
}

// Re-apply a running-statistics update from SAVED batch statistics (mean, invstd) without touching the
// activations: used when a forward pass is replayed from cached activations (the decoder passes that the
// reference recomputes with unchanged weights, train_soft_intro_vae.py:557 vs :597 and :561 vs :598) so that
// running_mean / running_var / num_batches_tracked still receive exactly one update per reference pass.
__global__ void __launch_bounds__(64) bn_update_running_kernel(const float* __restrict__ mean,
                                                               const float* __restrict__ invstd, int C, double count,
                                                               float eps, float momentum, float* running_mean,
                                                               float* running_var, long long* num_batches_tracked) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
  if (c >= C) return;
  const double is = (double)invstd[c];
  double var = 1.0 / (is * is) - (double)eps;
  if (var < 0.0) var = 0.0;
  const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
  running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * (double)mean[c]);
  running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
}

// segmented replay: mean / invstd [nseg][C], one momentum update per pass (seg_rev: last pass first)
__global__ void __launch_bounds__(64) bn_update_running_seg_kernel(const float* __restrict__ mean,
                                                                   const float* __restrict__ invstd, int nseg,
                                                                   int seg_rev, int C, double count, float eps,
                                                                   float momentum, float* running_mean,
                                                                   float* running_var, long long* num_batches_tracked) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c == 0 && num_batches_tracked) *num_batches_tracked += nseg;
  if (c >= C) return;
  float rm = running_mean[c], rv = running_var[c];
  for (int gi = 0; gi < nseg; ++gi) {
    const int g = seg_rev ? nseg - 1 - gi : gi;
    const double is = (double)invstd[g * C + c];
    double var = 1.0 / (is * is) - (double)eps;
    if (var < 0.0) var = 0.0;
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    rm = (float)((1.0 - momentum) * rm + momentum * (double)mean[g * C + c]);
    rv = (float)((1.0 - momentum) * rv + momentum * unbiased);
  }
  running_mean[c] = rm;
  running_var[c] = rv;
}

extern "C" int sivae_bn_update_running_seg(const float* mean, const float* invstd, int nseg, int seg_rev, int C,
                                           double count, float eps, float momentum, float* running_mean,
                                           float* running_var, long long* num_batches_tracked, hipStream_t stream) {
  if (!mean || !invstd || !running_mean || !running_var) return SIVAE_ERR_NULL;
  if (C <= 0 || nseg <= 0 || count <= 0.0) return SIVAE_ERR_SHAPE;
  hipLaunchKernelGGL(bn_update_running_seg_kernel, dim3(cdiv(C, 64)), dim3(64), 0, stream, mean, invstd, nseg, seg_rev,
                     C, count, eps, momentum, running_mean, running_var, num_batches_tracked);
  return sivae_launch_status();
}

extern "C" int sivae_bn_update_running(const float* mean, const float* invstd, int C, double count, float eps,
                                       float momentum, float* running_mean, float* running_var,
                                       long long* num_batches_tracked, hipStream_t stream) {
  if (!mean || !invstd || !running_mean || !running_var) return SIVAE_ERR_NULL;
  if (C <= 0 || count <= 0.0) return SIVAE_ERR_SHAPE;
  hipLaunchKernelGGL(bn_update_running_kernel, dim3(cdiv(C, 64)), dim3(64), 0, stream, mean, invstd, C, count, eps,
                     momentum, running_mean, running_var, num_batches_tracked);
  return sivae_launch_status();
}

// ------------------------------------------------------------------------------------------------
// apply:  y = LeakyReLU( (x - mean[c]) * invstd[c]*gamma[c] + beta[c]  (+ res) )
// slope == 1 -> no activation.
// ------------------------------------------------------------------------------------------------
template <bool HAS_RES, bool VEC>
__global__ void __launch_bounds__(256) bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                       const float* __restrict__ mean,
                                                       const float* __restrict__ invstd,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float slope,
                                                       float* __restrict__ y, unsigned char* __restrict__ mask,
                                                       int C, int HW, size_t numel, int segB) {
  // segB > 0: images [g*segB, (g+1)*segB) form pass g with its own statistics mean/invstd[g*C + c] (gamma/beta shared)
  const size_t stride = (size_t)gridDim.x * 256;
  if (VEC) {
    const size_t n4 = numel >> 2;
    // (block-uniform trip count: the sign-mask lane exchange needs the whole wave)
    for (size_t base = (size_t)blockIdx.x * 256; base < n4; base += stride) {
      const size_t i = base + threadIdx.x;
      const bool ok = i < n4;
      unsigned nib = 0;
      if (ok) {
        const size_t e = i << 2;
        const unsigned pl = (unsigned)(e / HW);
        const int c = (int)(pl % (unsigned)C);
        const int sc = segB > 0 ? c + (int)((pl / (unsigned)C) / (unsigned)segB) * C : c;
        const float m = mean[sc], g = invstd[sc] * gamma[c], bt = beta[c];
        float4 v = reinterpret_cast<const float4*>(x)[i];
        v.x = (v.x - m) * g + bt;
        v.y = (v.y - m) * g + bt;
        v.z = (v.z - m) * g + bt;
        v.w = (v.w - m) * g + bt;
        if (HAS_RES) {
          const float4 r = reinterpret_cast<const float4*>(res)[i];
          v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        nib = sign_nibble(v);
        v.x = lrelu(v.x, slope); v.y = lrelu(v.y, slope); v.z = lrelu(v.z, slope); v.w = lrelu(v.w, slope);
        reinterpret_cast<float4*>(y)[i] = v;
      }
      if (mask != nullptr) {
        const unsigned byte = pair_nibbles(nib);
        if (ok && !(threadIdx.x & 1)) mask[i >> 1] = (unsigned char)byte;
      }
    }
  } else {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < numel; e += stride) {
      const unsigned pl = (unsigned)(e / HW);
      const int c = (int)(pl % (unsigned)C);
      const int sc = segB > 0 ? c + (int)((pl / (unsigned)C) / (unsigned)segB) * C : c;
      float v = (x[e] - mean[sc]) * (invstd[sc] * gamma[c]) + beta[c];
      if (HAS_RES) v += res[e];
      y[e] = lrelu(v, slope);
    }
  }
}

static int bn_apply_impl(const float* x, const float* res, const float* mean, const float* invstd,
                         const float* gamma, const float* beta, float slope, float* y, unsigned char* mask, int B,
                         int C, int HW, hipStream_t stream, int segB = 0) {
  if (!x || !mean || !invstd || !gamma || !beta || !y) return SIVAE_ERR_NULL;
  if (B <= 0 || C <= 0 || HW <= 0) return SIVAE_ERR_SHAPE;
  if (segB < 0 || (segB > 0 && B % segB != 0)) return SIVAE_ERR_SHAPE;
  const size_t numel = (size_t)B * C * HW;
  const bool vec = (HW & 3) == 0;
  if (mask && !vec) return SIVAE_ERR_SHAPE;
  long long work = vec ? (long long)(numel >> 2) : (long long)numel;
  int nb = cdiv(work, 256);
  if (nb > 8192) nb = 8192;
  if (nb < 1) nb = 1;
#define LAUNCH(R, V) \
  hipLaunchKernelGGL((bn_apply_kernel<R, V>), dim3(nb), dim3(256), 0, stream, x, res, mean, invstd, gamma, beta, \
                     slope, y, mask, C, HW, numel, segB)
  if (res) {
    if (vec) LAUNCH(true, true); else LAUNCH(true, false);
  } else {
    if (vec) LAUNCH(false, true); else LAUNCH(false, false);
  }
#undef LAUNCH
  return sivae_launch_status();
}

extern "C" int sivae_bn_apply_act(const float* x, const float* res, const float* mean, const float* invstd,
                                  const float* gamma, const float* beta, float slope, float* y, int B, int C,
                                  int HW, hipStream_t stream) {
  return bn_apply_impl(x, res, mean, invstd, gamma, beta, slope, y, nullptr, B, C, HW, stream);
}

// Same op with the residual stored at HALF resolution and read through nn.Upsample(2,'nearest') addressing
// (res[b][c][h>>1][w>>1]): the decoder's upsampled block inputs (train_soft_intro_vae.py:155) are never
// materialised — the convs read them the same way (upsample flag of the conv kernels).  W % 4 == 0.
__global__ void __launch_bounds__(256) bn_apply_resup_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ invstd,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float slope,
                                                             float* __restrict__ y, unsigned char* __restrict__ mask,
                                                             int C, int H, int W, size_t numel, int segB) {
  const size_t stride = (size_t)gridDim.x * 256;
  const int W4 = W >> 2, Hs = H >> 1, Ws = W >> 1;
  const size_t n4 = numel >> 2;
  for (size_t base = (size_t)blockIdx.x * 256; base < n4; base += stride) {
    const size_t i = base + threadIdx.x;
    const bool ok = i < n4;
    unsigned nib = 0;
    if (ok) {
      const int w4 = (int)(i % W4);
      size_t t = i / W4;
      const int h = (int)(t % H);
      t /= H;  // b*C + c
      const int c = (int)(t % C);
      const int sc = segB > 0 ? c + (int)(((unsigned)t / (unsigned)C) / (unsigned)segB) * C : c;
      const float m = mean[sc], g = invstd[sc] * gamma[c], bt = beta[c];
      float4 v = reinterpret_cast<const float4*>(x)[i];
      const float2 r = *reinterpret_cast<const float2*>(res + (t * Hs + (h >> 1)) * Ws + 2 * w4);
      v.x = (v.x - m) * g + bt + r.x;
      v.y = (v.y - m) * g + bt + r.x;
      v.z = (v.z - m) * g + bt + r.y;
      v.w = (v.w - m) * g + bt + r.y;
      nib = sign_nibble(v);
      v.x = lrelu(v.x, slope); v.y = lrelu(v.y, slope); v.z = lrelu(v.z, slope); v.w = lrelu(v.w, slope);
      reinterpret_cast<float4*>(y)[i] = v;
    }
    if (mask != nullptr) {
      const unsigned byte = pair_nibbles(nib);
      if (ok && !(threadIdx.x & 1)) mask[i >> 1] = (unsigned char)byte;
    }
  }
}

static int bn_apply_resup_impl(const float* x, const float* res_half, const float* mean, const float* invstd,
                               const float* gamma, const float* beta, float slope, float* y, unsigned char* mask,
                               int B, int C, int H, int W, hipStream_t stream, int segB = 0) {
  if (!x || !res_half || !mean || !invstd || !gamma || !beta || !y) return SIVAE_ERR_NULL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 3)) return SIVAE_ERR_SHAPE;
  if (segB < 0 || (segB > 0 && B % segB != 0)) return SIVAE_ERR_SHAPE;
  const size_t numel = (size_t)B * C * H * W;
  int nb = cdiv((long long)(numel >> 2), 256 * 4);
  if (nb > 8192) nb = 8192;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(bn_apply_resup_kernel, dim3(nb), dim3(256), 0, stream, x, res_half, mean, invstd, gamma, beta,
                     slope, y, mask, C, H, W, numel, segB);
  return sivae_launch_status();
}

extern "C" int sivae_bn_apply_act_resup(const float* x, const float* res_half, const float* mean, const float* invstd,
                                        const float* gamma, const float* beta, float slope, float* y, int B, int C,
                                        int H, int W, hipStream_t stream) {
  return bn_apply_resup_impl(x, res_half, mean, invstd, gamma, beta, slope, y, nullptr, B, C, H, W, stream);
}

// Same op, also writing AvgPool2d(2) of the result (the pool that follows every encoder block and the stem,
// :92,:98): one pass produces `y` (kept for backward) and the pooled tensor the next layer reads.
// One thread = 2 rows x 4 columns.  H even, W % 4 == 0.
template <bool HAS_RES>
__global__ void __launch_bounds__(256) bn_apply_pool_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float slope,
                                                            float* __restrict__ y, float* __restrict__ yp,
                                                            unsigned char* __restrict__ mask, int C, int H, int W,
                                                            size_t n_quads, int segB) {
  const size_t stride = (size_t)gridDim.x * 256;
  const int W4 = W >> 2, H2 = H >> 1;
  for (size_t base = (size_t)blockIdx.x * 256; base < n_quads; base += stride) {
    const size_t i = base + threadIdx.x;
    const bool ok = i < n_quads;
    unsigned nib0 = 0, nib1 = 0;
    size_t m0 = 0, m1 = 0;
    if (ok) {
      const int w4 = (int)(i % W4);
      size_t t = i / W4;
      const int h2 = (int)(t % H2);
      t /= H2;  // b*C + c
      const int c = (int)(t % C);
      const int sc = segB > 0 ? c + (int)(((unsigned)t / (unsigned)C) / (unsigned)segB) * C : c;
      const float m = mean[sc], g = invstd[sc] * gamma[c], bt = beta[c];
      const size_t o0 = (t * H + 2 * h2) * (size_t)W + 4 * w4, o1 = o0 + W;
      float4 a = *reinterpret_cast<const float4*>(x + o0), b = *reinterpret_cast<const float4*>(x + o1);
      a.x = (a.x - m) * g + bt; a.y = (a.y - m) * g + bt; a.z = (a.z - m) * g + bt; a.w = (a.w - m) * g + bt;
      b.x = (b.x - m) * g + bt; b.y = (b.y - m) * g + bt; b.z = (b.z - m) * g + bt; b.w = (b.w - m) * g + bt;
      if (HAS_RES) {
        const float4 ra = *reinterpret_cast<const float4*>(res + o0), rb = *reinterpret_cast<const float4*>(res + o1);
        a.x += ra.x; a.y += ra.y; a.z += ra.z; a.w += ra.w;
        b.x += rb.x; b.y += rb.y; b.z += rb.z; b.w += rb.w;
      }
      nib0 = sign_nibble(a);
      nib1 = sign_nibble(b);
      m0 = o0 >> 3;  // (W % 8 == 0 with a mask: the lane pair (w4, w4 ^ 1) shares one byte per row)
      m1 = o1 >> 3;
      a.x = lrelu(a.x, slope); a.y = lrelu(a.y, slope); a.z = lrelu(a.z, slope); a.w = lrelu(a.w, slope);
      b.x = lrelu(b.x, slope); b.y = lrelu(b.y, slope); b.z = lrelu(b.z, slope); b.w = lrelu(b.w, slope);
      if (y != nullptr) {
        *reinterpret_cast<float4*>(y + o0) = a;
        *reinterpret_cast<float4*>(y + o1) = b;
      }
      // same summation order as avgpool2_fwd_kernel: ((row0.l + row0.r) + (row1.l + row1.r)) * 0.25
      *reinterpret_cast<float2*>(yp + (t * H2 + h2) * (size_t)(W >> 1) + 2 * w4) =
          make_float2(((a.x + a.y) + (b.x + b.y)) * 0.25f, ((a.z + a.w) + (b.z + b.w)) * 0.25f);
    }
    if (mask != nullptr) {
      const unsigned byte0 = pair_nibbles(nib0), byte1 = pair_nibbles(nib1);
      if (ok && !(threadIdx.x & 1)) {
        mask[m0] = (unsigned char)byte0;
        mask[m1] = (unsigned char)byte1;
      }
    }
  }
}

static int bn_apply_pool_impl(const float* x, const float* res, const float* mean, const float* invstd,
                              const float* gamma, const float* beta, float slope, float* y, float* y_pooled,
                              unsigned char* mask, int B, int C, int H, int W, hipStream_t stream, int segB = 0) {
  if (!x || !mean || !invstd || !gamma || !beta || !y_pooled) return SIVAE_ERR_NULL;  // y may be NULL: pooled only
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 3)) return SIVAE_ERR_SHAPE;
  if (segB < 0 || (segB > 0 && B % segB != 0)) return SIVAE_ERR_SHAPE;
  if (mask && (W & 7)) return SIVAE_ERR_SHAPE;
  const size_t n_quads = (size_t)B * C * (H >> 1) * (W >> 2);
  int nb = cdiv((long long)n_quads, 256 * 2);
  if (nb > 8192) nb = 8192;
  if (nb < 1) nb = 1;
  if (res)
    hipLaunchKernelGGL(bn_apply_pool_kernel<true>, dim3(nb), dim3(256), 0, stream, x, res, mean, invstd, gamma, beta,
                       slope, y, y_pooled, mask, C, H, W, n_quads, segB);
  else
    hipLaunchKernelGGL(bn_apply_pool_kernel<false>, dim3(nb), dim3(256), 0, stream, x, res, mean, invstd, gamma, beta,
                       slope, y, y_pooled, mask, C, H, W, n_quads, segB);
  return sivae_launch_status();
}

extern "C" int sivae_bn_apply_act_pool(const float* x, const float* res, const float* mean, const float* invstd,
                                       const float* gamma, const float* beta, float slope, float* y, float* y_pooled,
                                       int B, int C, int H, int W, hipStream_t stream) {
  return bn_apply_pool_impl(x, res, mean, invstd, gamma, beta, slope, y, y_pooled, nullptr, B, C, H, W, stream);
}

// ---- the three apply flavours with the LeakyReLU sign mask as an extra output (see sign_nibble above):
//   y_pooled != NULL : BatchNorm + residual + LeakyReLU + AvgPool2d(2); y (full resolution) may be NULL
//   res_up != 0      : residual at half resolution, read through Upsample(2,'nearest') addressing
//   otherwise        : plain apply.           mask: sivae_bn_signmask_bytes(B, C, H*W) bytes.  H even, W % 8 == 0.
extern "C" size_t sivae_bn_signmask_bytes(int B, int C, int HW) {
  if (B <= 0 || C <= 0 || HW <= 0) return 0;
  return ((size_t)B * C * HW + 7) / 8;
}

static int bn_apply_signmask_impl(const float* x, const float* res, int res_up, const float* mean,
                                  const float* invstd, const float* gamma, const float* beta, float slope, float* y,
                                  float* y_pooled, unsigned char* mask, int B, int C, int H, int W, int segB,
                                  hipStream_t stream) {
  if (!mask) return SIVAE_ERR_NULL;
  if (H <= 0 || W <= 0 || (H & 1) || (W & 7)) return SIVAE_ERR_SHAPE;
  if (y_pooled) {
    if (res_up) return SIVAE_ERR_MODE;
    return bn_apply_pool_impl(x, res, mean, invstd, gamma, beta, slope, y, y_pooled, mask, B, C, H, W, stream, segB);
  }
  if (res_up) return bn_apply_resup_impl(x, res, mean, invstd, gamma, beta, slope, y, mask, B, C, H, W, stream, segB);
  return bn_apply_impl(x, res, mean, invstd, gamma, beta, slope, y, mask, B, C, H * W, stream, segB);
}

extern "C" int sivae_bn_apply_act_signmask(const float* x, const float* res, int res_up, const float* mean,
                                           const float* invstd, const float* gamma, const float* beta, float slope,
                                           float* y, float* y_pooled, unsigned char* mask, int B, int C, int H, int W,
                                           hipStream_t stream) {
  return bn_apply_signmask_impl(x, res, res_up, mean, invstd, gamma, beta, slope, y, y_pooled, mask, B, C, H, W, 0,
                                stream);
}

// ---- segmented forms (B = nseg * seg_images images; mean / invstd are [nseg][C], gamma / beta [C]): several passes of a
// network through the same weights run as ONE batch with per-pass BatchNorm statistics (functional.py "segments")
extern "C" int sivae_bn_apply_act_signmask_seg(const float* x, const float* res, int res_up, const float* mean,
                                               const float* invstd, const float* gamma, const float* beta, float slope,
                                               float* y, float* y_pooled, unsigned char* mask, int B, int C, int H,
                                               int W, int seg_images, hipStream_t stream) {
  if (seg_images <= 0) return SIVAE_ERR_SHAPE;
  return bn_apply_signmask_impl(x, res, res_up, mean, invstd, gamma, beta, slope, y, y_pooled, mask, B, C, H, W,
                                seg_images, stream);
}

// plain apply (+ residual; res_up: residual at half resolution; y_pooled != NULL: also AvgPool2d(2), y may be NULL)
extern "C" int sivae_bn_apply_act_seg(const float* x, const float* res, int res_up, const float* mean,
                                      const float* invstd, const float* gamma, const float* beta, float slope,
                                      float* y, float* y_pooled, int B, int C, int H, int W, int seg_images,
                                      hipStream_t stream) {
  if (seg_images <= 0) return SIVAE_ERR_SHAPE;
  if (y_pooled) {
    if (res_up) return SIVAE_ERR_MODE;
    return bn_apply_pool_impl(x, res, mean, invstd, gamma, beta, slope, y, y_pooled, nullptr, B, C, H, W, stream,
                              seg_images);
  }
  if (res_up)
    return bn_apply_resup_impl(x, res, mean, invstd, gamma, beta, slope, y, nullptr, B, C, H, W, stream, seg_images);
  return bn_apply_impl(x, res, mean, invstd, gamma, beta, slope, y, nullptr, B, C, H * W, stream, seg_images);
}

// ------------------------------------------------------------------------------------------------
// backward of  y = LeakyReLU(BN(x) [+ res])
//   dz = dy * (y > 0 ? 1 : slope)        (y is the saved OUTPUT; valid because slope > 0)
//   dbeta = sum dz ; dgamma = sum dz * xhat
//   dx = gamma*invstd * (dz - mean(dz) - xhat * mean(dz*xhat)) ;  dres = dz
// pass 1: per-channel partial sums (fp64), pass 2: coefficients, pass 3: dx / dz.
// If y == nullptr no activation mask is applied.
// ------------------------------------------------------------------------------------------------
// dy of 4 consecutive pixels of plane `pl` starting at hw (multiple of 4).  pool_w != 0: dy is the gradient of
// AvgPool2d(2)(y) stored at half resolution [.][H/2][W/2] and is read through the pool's adjoint
// (0.25 * dy_half[h>>1][w>>1]) — the full-resolution gradient of nn.AvgPool2d (:92,:98) is never written.
__device__ __forceinline__ float4 bn_load_dy4(const float* __restrict__ dy, size_t pl, int hw, int HW, int pool_w) {
  if (pool_w == 0) return *reinterpret_cast<const float4*>(dy + pl * HW + hw);
  const int h = hw / pool_w, w = hw - h * pool_w;
  const float2 d = *reinterpret_cast<const float2*>(dy + pl * (size_t)(HW >> 2) + (size_t)(h >> 1) * (pool_w >> 1) + (w >> 1));
  return make_float4(0.25f * d.x, 0.25f * d.x, 0.25f * d.y, 0.25f * d.y);
}

template <int ACT>
__global__ void __launch_bounds__(256) bn_bwd_partial_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                             const float* __restrict__ x,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ invstd,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float slope,
                                                             double* __restrict__ part, int C, int HW,
                                                             long long n_per_ch, long long slice_len, int S,
                                                             int pool_w, const unsigned char* __restrict__ mask,
                                                             int segB, unsigned* __restrict__ counters, int nseg,
                                                             double count, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, float* __restrict__ coef) {
  // blockIdx.x = g*C + c: pass (segment) g of segB images with its own statistics; n_per_ch counts ONE segment
  __shared__ double red[4];
  __shared__ int is_last;
  const int vc = blockIdx.x, s = blockIdx.y;
  const int c = vc % C;
  const long long b0 = (long long)(vc / C) * segB;
  const long long n0 = (long long)s * slice_len;
  long long n1 = n0 + slice_len;
  if (n1 > n_per_ch) n1 = n_per_ch;
  const float m = mean[vc], is = invstd[vc];
  const float gsc = ACT == 2 ? is * gamma[c] : 0.f, bt = ACT == 2 ? beta[c] : 0.f;
  double s1 = 0.0, s2 = 0.0;
  if ((HW & 3) == 0) {
    for (long long n = n0 + (long long)threadIdx.x * 4; n < n1; n += 1024) {
      const long long bl = n / HW, b = bl + b0;
      const size_t o = ((size_t)b * C + c) * HW + (size_t)(n - bl * HW);
      const float4 g = bn_load_dy4(dy, (size_t)b * C + c, (int)(n - bl * HW), HW, pool_w);
      const float4 xv = *reinterpret_cast<const float4*>(x + o);
      float gz[4] = {g.x, g.y, g.z, g.w};
      if (ACT == 1) {
        const float4 yv = *reinterpret_cast<const float4*>(y + o);
        gz[0] = yv.x > 0.f ? gz[0] : gz[0] * slope;
        gz[1] = yv.y > 0.f ? gz[1] : gz[1] * slope;
        gz[2] = yv.z > 0.f ? gz[2] : gz[2] * slope;
        gz[3] = yv.w > 0.f ? gz[3] : gz[3] * slope;
      } else if (ACT == 2) {
        gz[0] = ((xv.x - m) * gsc + bt) > 0.f ? gz[0] : gz[0] * slope;
        gz[1] = ((xv.y - m) * gsc + bt) > 0.f ? gz[1] : gz[1] * slope;
        gz[2] = ((xv.z - m) * gsc + bt) > 0.f ? gz[2] : gz[2] * slope;
        gz[3] = ((xv.w - m) * gsc + bt) > 0.f ? gz[3] : gz[3] * slope;
      } else if (ACT == 3) {
        apply_nibble(gz, load_nibble(mask, o >> 2), slope);
      }
      const float xh[4] = {(xv.x - m) * is, (xv.y - m) * is, (xv.z - m) * is, (xv.w - m) * is};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        s1 += (double)gz[k];
        s2 += (double)gz[k] * (double)xh[k];
      }
    }
  } else {
    for (long long n = n0 + threadIdx.x; n < n1; n += 256) {
      const long long bl = n / HW, b = bl + b0;
      const size_t o = ((size_t)b * C + c) * HW + (size_t)(n - bl * HW);
      float g = dy[o];
      if (ACT == 1) g = y[o] > 0.f ? g : g * slope;
      if (ACT == 2) g = ((x[o] - m) * gsc + bt) > 0.f ? g : g * slope;
      s1 += (double)g;
      s2 += (double)g * (double)((x[o] - m) * is);
    }
  }
  s1 = block_sum<256>(s1, red);
  s2 = block_sum<256>(s2, red);
  if (threadIdx.x == 0) {
    part[((size_t)vc * S + s) * 2 + 0] = s1;
    part[((size_t)vc * S + s) * 2 + 1] = s2;
  }
  if (counters == nullptr) return;
  // Fused finalize (saves one ~5 us launch per BatchNorm backward — ~110 per iteration, 1.3 % of an 8-image-shard
  // iteration): the block that completes channel c (all segments, all slices) folds the partials — in the SAME fixed
  // order as bn_bwd_finalize_kernel, so the result does not depend on which block it is.  counters[c] is zero on entry
  // and is left zero (self-resetting; one stream at a time per counter buffer).
  if (threadIdx.x == 0) {
    __threadfence();  // publish this block's partial (agent scope: the other blocks may sit on another XCD's L2)
    const unsigned done = atomicAdd(&counters[c], 1u);
    is_last = done == (unsigned)(nseg * S) - 1u;
  }
  __syncthreads();
  if (is_last && threadIdx.x == 0) {
    __threadfence();
    double t1 = 0.0, t2 = 0.0;
    for (int g = 0; g < nseg; ++g) {
      const int v = g * C + c;
      double a1 = 0.0, a2 = 0.0;
      for (int k = 0; k < S; ++k) {
        a1 += __hip_atomic_load(&part[((size_t)v * S + k) * 2 + 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a2 += __hip_atomic_load(&part[((size_t)v * S + k) * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      coef[v * 2 + 0] = (float)(a1 / count);
      coef[v * 2 + 1] = (float)(a2 / count);
      t1 += a1;
      t2 += a2;
    }
    if (dbeta) dbeta[c] = (float)t1;
    if (dgamma) dgamma[c] = (float)t2;
    counters[c] = 0u;
  }
}

// coefficients per (segment, channel); dgamma / dbeta are summed over the segments (gamma / beta are shared)
__global__ void __launch_bounds__(64) bn_bwd_finalize_kernel(const double* __restrict__ part, int S, int C,
                                                             double count, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, float* __restrict__ coef,
                                                             int nseg) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  double t1 = 0.0, t2 = 0.0;
  for (int g = 0; g < nseg; ++g) {
    const int vc = g * C + c;
    double s1 = 0.0, s2 = 0.0;
    for (int s = 0; s < S; ++s) {
      s1 += part[((size_t)vc * S + s) * 2 + 0];
      s2 += part[((size_t)vc * S + s) * 2 + 1];
    }
    coef[vc * 2 + 0] = (float)(s1 / count);
    coef[vc * 2 + 1] = (float)(s2 / count);
    t1 += s1;
    t2 += s2;
  }
  if (dbeta) dbeta[c] = (float)t1;
  if (dgamma) dgamma[c] = (float)t2;
}

template <int ACT, bool HAS_DZ, bool VEC>
__global__ void __launch_bounds__(256) bn_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                        const float* __restrict__ x, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        const float* __restrict__ coef, float slope,
                                                        float* __restrict__ dx, float* __restrict__ dz_out, int C,
                                                        int HW, size_t numel, int pool_w,
                                                        const unsigned char* __restrict__ mask, int segB) {
  const size_t stride = (size_t)gridDim.x * 256;
  if (VEC) {
    const size_t n4 = numel >> 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
      const size_t e = i << 2;
      const unsigned pl = (unsigned)(e / HW);
      const int c = (int)(pl % (unsigned)C);
      const int sc = segB > 0 ? c + (int)((pl / (unsigned)C) / (unsigned)segB) * C : c;
      const float m = mean[sc], is = invstd[sc], gs = gamma[c] * is, c1 = coef[sc * 2], c2 = coef[sc * 2 + 1];
      const float4 g = bn_load_dy4(dy, e / HW, (int)(e % HW), HW, pool_w);
      const float4 xv = reinterpret_cast<const float4*>(x)[i];
      float gz[4] = {g.x, g.y, g.z, g.w};
      if (ACT == 1) {
        const float4 yv = reinterpret_cast<const float4*>(y)[i];
        gz[0] = yv.x > 0.f ? gz[0] : gz[0] * slope;
        gz[1] = yv.y > 0.f ? gz[1] : gz[1] * slope;
        gz[2] = yv.z > 0.f ? gz[2] : gz[2] * slope;
        gz[3] = yv.w > 0.f ? gz[3] : gz[3] * slope;
      } else if (ACT == 2) {
        const float bt = beta[c];
        gz[0] = ((xv.x - m) * gs + bt) > 0.f ? gz[0] : gz[0] * slope;
        gz[1] = ((xv.y - m) * gs + bt) > 0.f ? gz[1] : gz[1] * slope;
        gz[2] = ((xv.z - m) * gs + bt) > 0.f ? gz[2] : gz[2] * slope;
        gz[3] = ((xv.w - m) * gs + bt) > 0.f ? gz[3] : gz[3] * slope;
      } else if (ACT == 3) {
        apply_nibble(gz, load_nibble(mask, i), slope);
      }
      float4 o;
      o.x = gs * (gz[0] - c1 - (xv.x - m) * is * c2);
      o.y = gs * (gz[1] - c1 - (xv.y - m) * is * c2);
      o.z = gs * (gz[2] - c1 - (xv.z - m) * is * c2);
      o.w = gs * (gz[3] - c1 - (xv.w - m) * is * c2);
      reinterpret_cast<float4*>(dx)[i] = o;
      if (HAS_DZ) reinterpret_cast<float4*>(dz_out)[i] = make_float4(gz[0], gz[1], gz[2], gz[3]);
    }
  } else {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < numel; e += stride) {
      const unsigned pl = (unsigned)(e / HW);
      const int c = (int)(pl % (unsigned)C);
      const int sc = segB > 0 ? c + (int)((pl / (unsigned)C) / (unsigned)segB) * C : c;
      const float m = mean[sc], is = invstd[sc];
      float g = dy[e];
      if (ACT == 1) g = y[e] > 0.f ? g : g * slope;
      if (ACT == 2) g = ((x[e] - m) * (is * gamma[c]) + beta[c]) > 0.f ? g : g * slope;
      dx[e] = gamma[c] * is * (g - coef[sc * 2] - (x[e] - m) * is * coef[sc * 2 + 1]);
      if (HAS_DZ) dz_out[e] = g;
    }
  }
}

// dx pass of the ACT == 1 backward (sign from the saved output y) that leaves the residual-branch gradient dz as its
// 2x2 BLOCK SUM at half resolution — the adjoint of the nn.Upsample (:155) the block's input went through — instead of
// the full-resolution tensor: in the decoder dz is only ever consumed through that sum (identity skip, or the 1x1
// expand conv that runs at half resolution).  One thread = 2 rows x 4 columns; H even, W % 4 == 0.
template <bool MASK>
__global__ void __launch_bounds__(256) bn_bwd_dx_dzsum_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                              const float* __restrict__ x,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ invstd,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ coef, float slope,
                                                              float* __restrict__ dx, float* __restrict__ dz_half, int C,
                                                              int H, int W, size_t n_quads,
                                                              const unsigned char* __restrict__ mask, int segB) {
  const size_t stride = (size_t)gridDim.x * 256;
  const int W4 = W >> 2, H2 = H >> 1;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_quads; i += stride) {
    const int w4 = (int)(i % W4);
    size_t t = i / W4;
    const int h2 = (int)(t % H2);
    t /= H2;  // b*C + c
    const int c = (int)(t % C);
    const int sc = segB > 0 ? c + (int)(((unsigned)t / (unsigned)C) / (unsigned)segB) * C : c;
    const float m = mean[sc], is = invstd[sc], gs = gamma[c] * is, c1 = coef[sc * 2], c2 = coef[sc * 2 + 1];
    const size_t o0 = (t * H + 2 * h2) * (size_t)W + 4 * w4;
    float gsum[2] = {0.f, 0.f};
    float row0[2] = {0.f, 0.f};
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const size_t o = o0 + (size_t)rr * W;
      const float4 g = *reinterpret_cast<const float4*>(dy + o);
      const float4 xv = *reinterpret_cast<const float4*>(x + o);
      float gzv[4] = {g.x, g.y, g.z, g.w};
      if (MASK) {
        apply_nibble(gzv, load_nibble(mask, o >> 2), slope);
      } else {
        const float4 yv = *reinterpret_cast<const float4*>(y + o);
        gzv[0] = yv.x > 0.f ? g.x : g.x * slope; gzv[1] = yv.y > 0.f ? g.y : g.y * slope;
        gzv[2] = yv.z > 0.f ? g.z : g.z * slope; gzv[3] = yv.w > 0.f ? g.w : g.w * slope;
      }
      const float gz0 = gzv[0], gz1 = gzv[1], gz2 = gzv[2], gz3 = gzv[3];
      float4 d;
      d.x = gs * (gz0 - c1 - (xv.x - m) * is * c2);
      d.y = gs * (gz1 - c1 - (xv.y - m) * is * c2);
      d.z = gs * (gz2 - c1 - (xv.z - m) * is * c2);
      d.w = gs * (gz3 - c1 - (xv.w - m) * is * c2);
      *reinterpret_cast<float4*>(dx + o) = d;
      // same order as upsample2_bwd_kernel: (row0.l + row0.r) + (row1.l + row1.r)
      if (rr == 0) {
        row0[0] = gz0 + gz1;
        row0[1] = gz2 + gz3;
      } else {
        gsum[0] = row0[0] + (gz0 + gz1);
        gsum[1] = row0[1] + (gz2 + gz3);
      }
    }
    *reinterpret_cast<float2*>(dz_half + (t * H2 + h2) * (size_t)(W >> 1) + 2 * w4) = make_float2(gsum[0], gsum[1]);
  }
}

static int bn_bwd_impl(const float* dy, const float* y, const float* x, const float* mean, const float* invstd,
                       const float* gamma, const float* beta, int act_mode, float slope, float* dx, float* dz_out,
                       float* dgamma, float* dbeta, int B, int C, int HW, int pool_w, void* workspace,
                       size_t workspace_bytes, hipStream_t stream, int dzsum_w = 0,
                       const unsigned char* mask = nullptr, int segB = 0, unsigned* counters = nullptr) {
  // segB > 0: B = nseg * segB images, statistics / coefficients per (segment, channel); the workspace layout is the
  // unsegmented one with nseg*C virtual channels (it is sized for (B, C, HW), which covers (segB, nseg*C, HW))
  if (!dy || !x || !mean || !invstd || !gamma || !dx) return SIVAE_ERR_NULL;
  if (act_mode < 0 || act_mode > 3) return SIVAE_ERR_MODE;
  if (act_mode == 1 && !y) return SIVAE_ERR_NULL;
  if (act_mode == 3 && (!mask || (HW & 3))) return SIVAE_ERR_NULL;  // sign from the 1-bit mask (float4 path only)
  if (act_mode == 2 && !beta) return SIVAE_ERR_NULL;
  if (B <= 0 || C <= 0 || HW <= 0) return SIVAE_ERR_SHAPE;
  if (segB < 0 || (segB > 0 && B % segB != 0)) return SIVAE_ERR_SHAPE;
  const int nseg = segB > 0 ? B / segB : 1, Bs = segB > 0 ? segB : B, VC = nseg * C;
  const long long n = (long long)Bs * HW;
  SlicePlan p = plan_slices(n, VC);
  if (!workspace || workspace_bytes < ((size_t)VC * p.S * 2 + (size_t)VC * 2) * sizeof(double)) return SIVAE_ERR_WORKSPACE;
  double* part = (double*)workspace;
  float* coef = (float*)(part + (size_t)VC * p.S * 2);
#define LAUNCHP(A) \
  hipLaunchKernelGGL((bn_bwd_partial_kernel<A>), dim3(VC, p.S), dim3(256), 0, stream, dy, y, x, mean, invstd, gamma, \
                     beta, slope, part, C, HW, n, p.len, p.S, pool_w, mask, Bs, counters, nseg, (double)n, dgamma, \
                     dbeta, coef)
  if (act_mode == 0) LAUNCHP(0); else if (act_mode == 1) LAUNCHP(1); else if (act_mode == 2) LAUNCHP(2); else LAUNCHP(3);
#undef LAUNCHP
  if (counters == nullptr)
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 64)), dim3(64), 0, stream, (const double*)part, p.S, C,
                       (double)n, dgamma, dbeta, coef, nseg);
  const size_t numel = (size_t)B * C * HW;
  if (dzsum_w > 0) {  // dz_out is [B][C][H/2][W/2] block sums (act_mode 1 only, checked by the caller)
    const int W_ = dzsum_w, H_ = HW / dzsum_w;
    const size_t n_quads = (size_t)B * C * (H_ >> 1) * (W_ >> 2);
    int nq = cdiv((long long)n_quads, 256 * 2);
    if (nq > 8192) nq = 8192;
    if (nq < 1) nq = 1;
    if (act_mode == 3)
      hipLaunchKernelGGL(bn_bwd_dx_dzsum_kernel<true>, dim3(nq), dim3(256), 0, stream, dy, y, x, mean, invstd, gamma,
                         (const float*)coef, slope, dx, dz_out, C, H_, W_, n_quads, mask, segB);
    else
      hipLaunchKernelGGL(bn_bwd_dx_dzsum_kernel<false>, dim3(nq), dim3(256), 0, stream, dy, y, x, mean, invstd, gamma,
                         (const float*)coef, slope, dx, dz_out, C, H_, W_, n_quads, mask, segB);
    return sivae_launch_status();
  }
  const bool vec = (HW & 3) == 0;
  long long work = vec ? (long long)(numel >> 2) : (long long)numel;
  int nb = cdiv(work, 256);
  if (nb > 8192) nb = 8192;
  if (nb < 1) nb = 1;
#define LAUNCH(A, Z, V) \
  hipLaunchKernelGGL((bn_bwd_dx_kernel<A, Z, V>), dim3(nb), dim3(256), 0, stream, dy, y, x, mean, invstd, gamma, \
                     beta, (const float*)coef, slope, dx, dz_out, C, HW, numel, pool_w, mask, segB)
#define LAUNCH_A(A) \
  { if (hz) { if (vec) LAUNCH(A, true, true); else LAUNCH(A, true, false); } \
    else { if (vec) LAUNCH(A, false, true); else LAUNCH(A, false, false); } }
  const bool hz = dz_out != nullptr;
  if (act_mode == 0) LAUNCH_A(0) else if (act_mode == 1) LAUNCH_A(1) else if (act_mode == 2) LAUNCH_A(2)
  else { if (hz) LAUNCH(3, true, true); else LAUNCH(3, false, true); }
#undef LAUNCH_A
#undef LAUNCH
  return sivae_launch_status();
}

extern "C" int sivae_bn_bwd(const float* dy, const float* y, const float* x, const float* mean,
                            const float* invstd, const float* gamma, const float* beta, int act_mode,
                            float slope, float* dx, float* dz_out, float* dgamma, float* dbeta, int B, int C,
                            int HW, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  return bn_bwd_impl(dy, y, x, mean, invstd, gamma, beta, act_mode, slope, dx, dz_out, dgamma, dbeta, B, C, HW, 0,
                     workspace, workspace_bytes, stream);
}

// act_mode-1 backward whose residual-branch gradient comes out as its 2x2 block sum dz_half [B][C][H/2][W/2]
// (the adjoint of the nn.Upsample in front of the block) instead of the full-resolution dz; H even, W % 4 == 0.
extern "C" int sivae_bn_bwd_dzsum(const float* dy, const float* y, const float* x, const float* mean,
                                  const float* invstd, const float* gamma, float slope, float* dx, float* dz_half,
                                  float* dgamma, float* dbeta, int B, int C, int H, int W, void* workspace,
                                  size_t workspace_bytes, hipStream_t stream) {
  if (!y || !dz_half) return SIVAE_ERR_NULL;
  if (H <= 0 || W <= 0 || (H & 1) || (W & 3)) return SIVAE_ERR_SHAPE;
  return bn_bwd_impl(dy, y, x, mean, invstd, gamma, nullptr, 1, slope, dx, dz_half, dgamma, dbeta, B, C, H * W, 0,
                     workspace, workspace_bytes, stream, W);
}

// Same with dy = the gradient of AvgPool2d(2)(y) at half resolution [B][C][H/2][W/2] (H even, W % 4 == 0): the
// pool's adjoint is applied while reading, so the block backward of "ResidualBlock -> AvgPool2d" (:95-99) and of the
// stem (:88-93) never materialises the full-resolution gradient.
extern "C" int sivae_bn_bwd_pooled_dy(const float* dy_half, const float* y, const float* x, const float* mean,
                                      const float* invstd, const float* gamma, const float* beta, int act_mode,
                                      float slope, float* dx, float* dz_out, float* dgamma, float* dbeta, int B, int C,
                                      int H, int W, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (H <= 0 || W <= 0 || (H & 1) || (W & 3)) return SIVAE_ERR_SHAPE;
  return bn_bwd_impl(dy_half, y, x, mean, invstd, gamma, beta, act_mode, slope, dx, dz_out, dgamma, dbeta, B, C, H * W,
                     W, workspace, workspace_bytes, stream);
}

// Backward of "LeakyReLU(BN(x) + res)" with the LeakyReLU sign taken from the 1-bit mask the apply pass wrote
// (sivae_bn_apply_act_signmask) — 1/32 of a tensor per pass instead of the saved output.  dy_pooled != 0: dy is
// the gradient of AvgPool2d(2)(output) at half resolution; dz_sum != 0: dz_out is the 2x2 block sum of the
// residual-branch gradient ([B][C][H/2][W/2]); not both.  H even, W % 8 == 0.
extern "C" int sivae_bn_bwd_signmask(const float* dy, const unsigned char* mask, const float* x, const float* mean,
                                     const float* invstd, const float* gamma, float slope, float* dx, float* dz_out,
                                     float* dgamma, float* dbeta, int B, int C, int H, int W, int dy_pooled, int dz_sum,
                                     void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!mask) return SIVAE_ERR_NULL;
  if (H <= 0 || W <= 0 || (H & 1) || (W & 7)) return SIVAE_ERR_SHAPE;
  if (dy_pooled && dz_sum) return SIVAE_ERR_MODE;
  if (dz_sum && !dz_out) return SIVAE_ERR_NULL;
  return bn_bwd_impl(dy, nullptr, x, mean, invstd, gamma, nullptr, 3, slope, dx, dz_out, dgamma, dbeta, B, C, H * W,
                     dy_pooled ? W : 0, workspace, workspace_bytes, stream, dz_sum ? W : 0, mask);
}

// General segmented backward: every variant above with B = nseg * seg_images images and per-(segment, channel)
// statistics (mean / invstd [nseg][C]); dgamma / dbeta [C] are summed over the segments.
//   act_mode 0 none, 1 sign from the saved output y, 2 sign recomputed from x (needs beta), 3 sign from `mask`
//   dy_pooled: dy is the gradient of AvgPool2d(2)(output) ([B][C][H/2][W/2]); dz_sum: dz_out receives the 2x2 block
//   sums of the residual-branch gradient (act_mode 1 or 3)
//   counters: NULL, or >= C zero-initialised unsigned ints that this call leaves zero: the per-channel finalize then runs
//   inside the reduction kernel (last block of a channel) instead of as a launch of its own; a counter buffer must not
//   be shared by calls running concurrently on different streams
extern "C" int sivae_bn_bwd_seg(const float* dy, const float* y, const unsigned char* mask, const float* x,
                                const float* mean, const float* invstd, const float* gamma, const float* beta,
                                int act_mode, float slope, float* dx, float* dz_out, float* dgamma, float* dbeta, int B,
                                int C, int H, int W, int dy_pooled, int dz_sum, int seg_images, unsigned int* counters,
                                void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (seg_images <= 0 || H <= 0 || W <= 0) return SIVAE_ERR_SHAPE;
  if (dy_pooled && dz_sum) return SIVAE_ERR_MODE;
  if ((dy_pooled || dz_sum) && ((H & 1) || (W & 3))) return SIVAE_ERR_SHAPE;
  if (act_mode == 3 && ((H & 1) || (W & 7))) return SIVAE_ERR_SHAPE;
  if (dz_sum && (!dz_out || (act_mode != 1 && act_mode != 3))) return SIVAE_ERR_MODE;
  return bn_bwd_impl(dy, y, x, mean, invstd, gamma, beta, act_mode, slope, dx, dz_out, dgamma, dbeta, B, C, H * W,
                     dy_pooled ? W : 0, workspace, workspace_bytes, stream, dz_sum ? W : 0, mask, seg_images, counters);
}

// ---- backward whose first reduction was done by the producer of dy (sivae_conv2d_wino_dgrad_bnbwd): per-tile
// {sum g, sum g*xhat} float partials [n_tiles][C][2] -> coefficients (+ dgamma, dbeta), then the dx pass.
__global__ void __launch_bounds__(256) bn_bwd_coef_partials_kernel(const float* __restrict__ part, int S, int C,
                                                                   double count, float* __restrict__ dgamma,
                                                                   float* __restrict__ dbeta, float* __restrict__ coef) {
  __shared__ double red[4];
  const int c = blockIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int s = threadIdx.x; s < S; s += 256) {
    const float2 v = *reinterpret_cast<const float2*>(part + ((size_t)s * C + c) * 2);
    s1 += (double)v.x;
    s2 += (double)v.y;
  }
  s1 = block_sum<256>(s1, red);
  s2 = block_sum<256>(s2, red);
  if (threadIdx.x != 0) return;
  if (dbeta) dbeta[c] = (float)s1;
  if (dgamma) dgamma[c] = (float)s2;
  coef[c * 2 + 0] = (float)(s1 / count);
  coef[c * 2 + 1] = (float)(s2 / count);
}

extern "C" int sivae_bn_bwd_from_partials(const float* dy, const float* x, const float* mean, const float* invstd,
                                          const float* gamma, const float* beta, float slope, const float* partials,
                                          int n_tiles, float* dx, float* dgamma, float* dbeta, int B, int C, int HW,
                                          void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!dy || !x || !mean || !invstd || !gamma || !beta || !partials || !dx) return SIVAE_ERR_NULL;
  if (B <= 0 || C <= 0 || HW <= 0 || n_tiles <= 0) return SIVAE_ERR_SHAPE;
  if (!workspace || workspace_bytes < sivae_bn_workspace_bytes(B, C, HW)) return SIVAE_ERR_WORKSPACE;
  SlicePlan p = plan_slices((long long)B * HW, C);
  float* coef = (float*)((double*)workspace + (size_t)C * p.S * 2);
  hipLaunchKernelGGL(bn_bwd_coef_partials_kernel, dim3(C), dim3(256), 0, stream, partials, n_tiles, C,
                     (double)B * HW, dgamma, dbeta, coef);
  const size_t numel = (size_t)B * C * HW;
  const bool vec = (HW & 3) == 0;
  long long work = vec ? (long long)(numel >> 2) : (long long)numel;
  int nb = cdiv(work, 256);
  if (nb > 8192) nb = 8192;
  if (nb < 1) nb = 1;
  const float* y = nullptr;
  float* dz_out = nullptr;
  if (vec)
    hipLaunchKernelGGL((bn_bwd_dx_kernel<2, false, true>), dim3(nb), dim3(256), 0, stream, dy, y, x, mean, invstd, gamma,
                       beta, (const float*)coef, slope, dx, dz_out, C, HW, numel, 0, (const unsigned char*)nullptr, 0);
  else
    hipLaunchKernelGGL((bn_bwd_dx_kernel<2, false, false>), dim3(nb), dim3(256), 0, stream, dy, y, x, mean, invstd,
                       gamma, beta, (const float*)coef, slope, dx, dz_out, C, HW, numel, 0, (const unsigned char*)nullptr, 0);
  return sivae_launch_status();
}

// ---- the same backward in two calls for synchronised BatchNorm: `reduce` leaves the local per-channel
// {sum dz, sum dz*xhat} in fp64 (the caller all-reduces a copy), `apply` takes the local sums (-> dgamma, dbeta, which
// the data-parallel gradient all-reduce sums across ranks anyway) and the global sums + global count (-> dx).
__global__ void __launch_bounds__(64) bn_bwd_sums_kernel(const double* __restrict__ part, int S, int C,
                                                         double* __restrict__ sums) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int s = 0; s < S; ++s) {
    s1 += part[((size_t)c * S + s) * 2 + 0];
    s2 += part[((size_t)c * S + s) * 2 + 1];
  }
  sums[c * 2 + 0] = s1;
  sums[c * 2 + 1] = s2;
}

__global__ void __launch_bounds__(64) bn_bwd_coef_kernel(const double* __restrict__ sums_local,
                                                         const double* __restrict__ sums_global, int C, double count,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                         float* __restrict__ coef) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  if (dbeta) dbeta[c] = (float)sums_local[c * 2 + 0];
  if (dgamma) dgamma[c] = (float)sums_local[c * 2 + 1];
  coef[c * 2 + 0] = (float)(sums_global[c * 2 + 0] / count);
  coef[c * 2 + 1] = (float)(sums_global[c * 2 + 1] / count);
}

extern "C" int sivae_bn_bwd_reduce(const float* dy, const float* y, const float* x, const float* mean,
                                   const float* invstd, const float* gamma, const float* beta, int act_mode,
                                   float slope, double* sums, int B, int C, int HW, void* workspace,
                                   size_t workspace_bytes, hipStream_t stream) {
  if (!dy || !x || !mean || !invstd || !gamma || !sums) return SIVAE_ERR_NULL;
  if (act_mode < 0 || act_mode > 2) return SIVAE_ERR_MODE;
  if (act_mode == 1 && !y) return SIVAE_ERR_NULL;
  if (act_mode == 2 && !beta) return SIVAE_ERR_NULL;
  if (B <= 0 || C <= 0 || HW <= 0) return SIVAE_ERR_SHAPE;
  if (!workspace || workspace_bytes < sivae_bn_workspace_bytes(B, C, HW)) return SIVAE_ERR_WORKSPACE;
  const long long n = (long long)B * HW;
  SlicePlan p = plan_slices(n, C);
  double* part = (double*)workspace;
#define LAUNCHP(A) \
  hipLaunchKernelGGL((bn_bwd_partial_kernel<A>), dim3(C, p.S), dim3(256), 0, stream, dy, y, x, mean, invstd, gamma, \
                     beta, slope, part, C, HW, n, p.len, p.S, 0, (const unsigned char*)nullptr, B, (unsigned*)nullptr, 1, \
                     0.0, (float*)nullptr, (float*)nullptr, (float*)nullptr)
  if (act_mode == 0) LAUNCHP(0); else if (act_mode == 1) LAUNCHP(1); else LAUNCHP(2);
#undef LAUNCHP
  hipLaunchKernelGGL(bn_bwd_sums_kernel, dim3(cdiv(C, 64)), dim3(64), 0, stream, (const double*)part, p.S, C, sums);
  return sivae_launch_status();
}

extern "C" int sivae_bn_bwd_apply(const float* dy, const float* y, const float* x, const float* mean,
                                  const float* invstd, const float* gamma, const float* beta, int act_mode,
                                  float slope, const double* sums_local, const double* sums_global,
                                  double count_global, float* dx, float* dz_out, float* dgamma, float* dbeta, int B,
                                  int C, int HW, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!dy || !x || !mean || !invstd || !gamma || !dx || !sums_local || !sums_global) return SIVAE_ERR_NULL;
  if (act_mode < 0 || act_mode > 2) return SIVAE_ERR_MODE;
  if (act_mode == 1 && !y) return SIVAE_ERR_NULL;
  if (act_mode == 2 && !beta) return SIVAE_ERR_NULL;
  if (B <= 0 || C <= 0 || HW <= 0 || !(count_global > 0.0)) return SIVAE_ERR_SHAPE;
  if (!workspace || workspace_bytes < sivae_bn_workspace_bytes(B, C, HW)) return SIVAE_ERR_WORKSPACE;
  SlicePlan p = plan_slices((long long)B * HW, C);
  float* coef = (float*)((double*)workspace + (size_t)C * p.S * 2);
  hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3(cdiv(C, 64)), dim3(64), 0, stream, sums_local, sums_global, C,
                     count_global, dgamma, dbeta, coef);
  const size_t numel = (size_t)B * C * HW;
  const bool vec = (HW & 3) == 0;
  long long work = vec ? (long long)(numel >> 2) : (long long)numel;
  int nb = cdiv(work, 256);
  if (nb > 8192) nb = 8192;
  if (nb < 1) nb = 1;
#define LAUNCH(A, Z, V) \
  hipLaunchKernelGGL((bn_bwd_dx_kernel<A, Z, V>), dim3(nb), dim3(256), 0, stream, dy, y, x, mean, invstd, gamma, \
                     beta, (const float*)coef, slope, dx, dz_out, C, HW, numel, 0, (const unsigned char*)nullptr, 0)
#define LAUNCH_A(A) \
  { if (hz) { if (vec) LAUNCH(A, true, true); else LAUNCH(A, true, false); } \
    else { if (vec) LAUNCH(A, false, true); else LAUNCH(A, false, false); } }
  const bool hz = dz_out != nullptr;
  if (act_mode == 0) LAUNCH_A(0) else if (act_mode == 1) LAUNCH_A(1) else LAUNCH_A(2)
#undef LAUNCH_A
#undef LAUNCH
  return sivae_launch_status();
}

// per-channel sum over (B, HW) — bias gradient of the `predict` conv (train_soft_intro_vae.py:159)
__global__ void __launch_bounds__(64) channel_sum_finalize_kernel(const double* __restrict__ part, int S, int C,
                                                                  float* __restrict__ out) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0;
  for (int s = 0; s < S; ++s) s1 += part[((size_t)c * S + s) * 2];
  out[c] = (float)s1;
}

extern "C" int sivae_channel_sum(const float* x, float* out, int B, int C, int HW, void* workspace,
                                 size_t workspace_bytes, hipStream_t stream) {
  if (!x || !out) return SIVAE_ERR_NULL;
  if (B <= 0 || C <= 0 || HW <= 0) return SIVAE_ERR_SHAPE;
  if (!workspace || workspace_bytes < sivae_bn_workspace_bytes(B, C, HW)) return SIVAE_ERR_WORKSPACE;
  const long long n = (long long)B * HW;
  SlicePlan p = plan_slices(n, C);
  double* part = (double*)workspace;
  hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(C, p.S), dim3(256), 0, stream, x, part, C, HW, n, p.len, p.S);
  hipLaunchKernelGGL(channel_sum_finalize_kernel, dim3(cdiv(C, 64)), dim3(64), 0, stream, (const double*)part, p.S,
                     C, out);
  return sivae_launch_status();
}
