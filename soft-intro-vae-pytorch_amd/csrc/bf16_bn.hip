// BatchNorm / elementwise / layout kernels of the bf16 mode (blocked bf16 activations, see bf16_common.h).
// All of them are HBM-bound streaming passes with 16-byte accesses per lane; statistics and parameter gradients are
// fp32 per thread, fp64 across blocks.  Reference ops: nn.BatchNorm2d (training mode) + nn.LeakyReLU(0.2) + the residual
// add of ResidualBlock.forward (soft_intro_vae/train_soft_intro_vae.py:65-75), nn.AvgPool2d(2) / nn.Upsample(2) that
// follow the blocks (:93,:98,:155) and their autograd adjoints.
#include "bf16_common.h"

namespace {

// per-channel-block parameter table in LDS: tab[cb][NP][8]
template <int NP>
__device__ __forceinline__ void load_tab(const float* tab, int cb, float (*out)[8]) {
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int e = 0; e < 8; ++e) out[p][e] = tab[(cb * NP + p) * 8 + e];
}

__device__ __forceinline__ u32x4_t ldv(const void* base, size_t vec) {
  return reinterpret_cast<const u32x4_t*>(base)[vec];
}
__device__ __forceinline__ void stv(void* base, size_t vec, u32x4_t v) { reinterpret_cast<u32x4_t*>(base)[vec] = v; }

// LeakyReLU sign of a vector's 8 (rounded) outputs as one byte: bit e set <=> output e > 0.  The backward reads this
// byte (1/16 of the tensor) instead of the saved output; "> 0" on the rounded value is exactly what the output-based
// path tests, so both give the same gradients bit for bit.
__device__ __forceinline__ unsigned char sign_byte(const u32x4_t o) {
  unsigned m = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    m |= (bf16_lo(o[i]) > 0.f ? 1u : 0u) << (2 * i);
    m |= (bf16_hi(o[i]) > 0.f ? 1u : 0u) << (2 * i + 1);
  }
  return (unsigned char)m;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// layout conversion
// ---------------------------------------------------------------------------------------------------------------
__global__ void bf16_from_f32_nchw_kernel(const float* __restrict__ src, void* __restrict__ dst, int B, int C, int Cb,
                                          int HW, float scale) {
  const size_t n = (size_t)B * Cb * HW;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (size_t)gridDim.x * blockDim.x) {
    const int p = (int)(v % HW);
    const size_t t = v / HW;
    const int cb = (int)(t % Cb);
    const int b = (int)(t / Cb);
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = cb * 8 + e;
      f[e] = c < C ? src[((size_t)b * C + c) * HW + p] * scale : 0.f;
    }
    stv(dst, v, pack8(f));
  }
}

__global__ void bf16_to_f32_nchw_kernel(const void* __restrict__ src, float* __restrict__ dst, int B, int C, int Cb,
                                        int HW) {
  const size_t n = (size_t)B * Cb * HW;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (size_t)gridDim.x * blockDim.x) {
    const int p = (int)(v % HW);
    const size_t t = v / HW;
    const int cb = (int)(t % Cb);
    const int b = (int)(t / Cb);
    float f[8];
    unpack8(ldv(src, v), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = cb * 8 + e;
      if (c < C) dst[((size_t)b * C + c) * HW + p] = f[e];
    }
  }
}

// kw-packed form of the RGB-side 5x5 layers (the stem conv reads 3 image channels, the decoder's predict conv writes 3:
// padded to a 16-channel k-step / a 32-channel MFMA tile they waste 13/16 of the matrix work over 25 taps).  The 5
// kernel COLUMNS are moved into the channel dimension instead: with X'[kw*C + c][h][w] = x[c][h][w + sgn*(kw-2)] a 5x5
// conv over C <= 3 channels is a 5x1 conv over 5C <= 15 channels (ks code 51 of bf16_conv.hip / bf16_wgrad.hip: 5 taps
// instead of 25), and a 5x5 conv INTO C channels is a 5x1 conv into 5C channels followed by the column fold below.
//   sgn = +1: the stem's image (forward + weight gradient);  sgn = -1: the predict conv's output gradient
// (a thread owns one pixel: 5C loads that neighbouring lanes share through the cache, two 16-byte stores; C is a
// template parameter and the batch index a grid dimension, which leaves one integer division per pixel)
template <int C>
__global__ void __launch_bounds__(256) bf16_im2col_kw5_kernel(const float* __restrict__ src, void* __restrict__ dst,
                                                              int H, int W, int sgn) {
  const int HW = H * W, b = blockIdx.y;
  const float* sb = src + (size_t)b * C * HW;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
    const int w = p % W;
    float f[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) f[j] = 0.f;
#pragma unroll
    for (int kw = 0; kw < 5; ++kw) {
      const int ws = w + sgn * (kw - 2);
      const bool ok = ws >= 0 && ws < W;
#pragma unroll
      for (int c = 0; c < C; ++c) f[kw * C + c] = ok ? sb[(size_t)c * HW + (p - w + ws)] : 0.f;
    }
    stv(dst, ((size_t)b * 2 + 0) * HW + p, pack8(f));
    stv(dst, ((size_t)b * 2 + 1) * HW + p, pack8(f + 8));
  }
}

// dst[b][c][h][w] = bias[c] + sum_kw g[b][kw*C + c][h][w + sgn*(kw-2)]   (g: fp32 NCHW with 5C channels)
//   sgn = +1: the predict conv's output;  sgn = -1: the stem's input gradient
template <int C>
__global__ void __launch_bounds__(256) bf16_fold_kw5_kernel(const float* __restrict__ g, const float* __restrict__ bias,
                                                            float* __restrict__ dst, int H, int W, int sgn) {
  const int HW = H * W, b = blockIdx.y;
  const float* gb = g + (size_t)b * 5 * C * HW;
  float* db = dst + (size_t)b * C * HW;
  float bv[C];
#pragma unroll
  for (int c = 0; c < C; ++c) bv[c] = bias ? bias[c] : 0.f;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
    const int w = p % W;
    float acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = bv[c];
#pragma unroll
    for (int kw = 0; kw < 5; ++kw) {
      const int ws = w + sgn * (kw - 2);
      if (ws >= 0 && ws < W) {
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] += gb[(size_t)(kw * C + c) * HW + (p - w + ws)];
      }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) db[(size_t)c * HW + p] = acc[c];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// y = LeakyReLU(x * scale[c] + shift[c] + res)        (BatchNorm apply + residual + activation)
// QUAD: a thread owns a 2x2 pixel quad (needed for the fused AvgPool2d output and for a half-resolution residual
// that is added through nearest-upsample addressing).
// ---------------------------------------------------------------------------------------------------------------
template <bool QUAD>
__global__ void __launch_bounds__(256) bf16_bn_apply_kernel(const void* __restrict__ x, const void* __restrict__ res,
                                                            int res_up, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float slope,
                                                            void* __restrict__ y, void* __restrict__ yp,
                                                            unsigned char* __restrict__ mask, int B, int C, int Cb,
                                                            int H, int W, int seg_images) {
  // SEGMENTED batch (seg_images < B: B / seg_images passes laid end to end, mean / invstd [nseg][C], shared gamma / beta):
  // one table row per (segment, channel block); image b reads row (b / seg_images) * Cb + cb
  extern __shared__ __attribute__((aligned(16))) float tab[];  // [nseg][Cb][2][8]
  const int nseg = B / seg_images;
  for (int i = threadIdx.x; i < nseg * Cb * 8; i += blockDim.x) {
    const int sg = i / (Cb * 8), c = i - sg * (Cb * 8);
    float sc = 0.f, sh = 0.f;
    if (c < C) {
      sc = invstd[sg * C + c] * gamma[c];
      sh = beta[c] - mean[sg * C + c] * sc;
    }
    tab[((sg * Cb + (c >> 3)) * 2 + 0) * 8 + (c & 7)] = sc;
    tab[((sg * Cb + (c >> 3)) * 2 + 1) * 8 + (c & 7)] = sh;
  }
  __syncthreads();
  const int HW = H * W;
  if (!QUAD) {
    const size_t n = (size_t)B * Cb * HW;
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (size_t)gridDim.x * blockDim.x) {
      const unsigned bc = (unsigned)(v / HW);  // b * Cb + cb
      const unsigned b = bc / (unsigned)Cb;
      const int cb = (int)(bc - b * (unsigned)Cb + (b / (unsigned)seg_images) * (unsigned)Cb);  // (table row)
      float p[2][8], f[8];
      load_tab<2>(tab, cb, p);
      unpack8(ldv(x, v), f);
      if (res != nullptr) {
        float r[8];
        unpack8(ldv(res, v), r);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = lrelu01(f[e] * p[0][e] + p[1][e] + r[e], slope);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = lrelu01(f[e] * p[0][e] + p[1][e], slope);
      }
      const u32x4_t o = pack8(f);
      if (y != nullptr) stv(y, v, o);
      if (mask != nullptr) mask[v] = sign_byte(o);
    }
  } else {
    const int Hh = H >> 1, Wh = W >> 1, HWh = Hh * Wh;
    const size_t n = (size_t)B * Cb * HWh;
    for (size_t qd = (size_t)blockIdx.x * blockDim.x + threadIdx.x; qd < n; qd += (size_t)gridDim.x * blockDim.x) {
      const int wq = (int)(qd % Wh);
      const size_t t = qd / Wh;
      const int hq = (int)(t % Hh);
      const size_t bc = t / Hh;  // b * Cb + cb
      const unsigned bi = (unsigned)bc / (unsigned)Cb;
      const int cb = (int)((unsigned)bc - bi * (unsigned)Cb + (bi / (unsigned)seg_images) * (unsigned)Cb);  // (table row)
      float p[2][8];
      load_tab<2>(tab, cb, p);
      float rh[8];
      if (res != nullptr && res_up) unpack8(ldv(res, bc * HWh + (size_t)hq * Wh + wq), rh);
      float pool[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int dw = 0; dw < 2; ++dw) {
          const size_t v = bc * HW + (size_t)(2 * hq + dh) * W + 2 * wq + dw;
          float f[8];
          unpack8(ldv(x, v), f);
          if (res != nullptr && !res_up) {
            float r[8];
            unpack8(ldv(res, v), r);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = f[e] * p[0][e] + p[1][e] + r[e];
          } else if (res != nullptr) {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = f[e] * p[0][e] + p[1][e] + rh[e];
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = f[e] * p[0][e] + p[1][e];
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = lrelu01(f[e], slope);
          const u32x4_t o = pack8(f);
          if (y != nullptr) stv(y, v, o);
          if (mask != nullptr) mask[v] = sign_byte(o);
          // the pooled value averages the ROUNDED outputs (what a separate AvgPool2d pass over y would read)
          float fr[8];
          unpack8(o, fr);
#pragma unroll
          for (int e = 0; e < 8; ++e) pool[e] += fr[e];
        }
      if (yp != nullptr) {
#pragma unroll
        for (int e = 0; e < 8; ++e) pool[e] *= 0.25f;
        stv(yp, qd, pack8(pool));
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// BatchNorm(+residual)+LeakyReLU backward.
//   g   = dy' * LeakyReLU'(out)        dy' = dy, or dy_pooled[h>>1][w>>1] / 4 (adjoint of the fused AvgPool2d)
//   sign of out: from the saved output y (ACT 1) or recomputed from x-hat*gamma + beta (ACT 2: no residual)
//   pass 1: per channel  sg = sum g,  sgx = sum g * xhat          (-> dbeta, dgamma)
//   pass 2: dx = gamma*invstd * (g - sg/N - xhat * sgx/N);  dz = g  (gradient of the residual branch), optionally as
//           its 2x2 block sums (all a block behind an nn.Upsample needs of it)
// ---------------------------------------------------------------------------------------------------------------
// ACT 1: sign from the saved output yv; ACT 2: recomputed from xhat * gamma + beta; ACT 3: from the sign byte `m`
template <int ACT>
__device__ __forceinline__ void bwd_g(const float* dy, const float* yv, const float* xhat, const float* gam,
                                      const float* bet, float slope, float* g, unsigned m = 0) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    bool pos;
    if (ACT == 1)
      pos = yv[e] > 0.f;
    else if (ACT == 2)
      pos = xhat[e] * gam[e] + bet[e] > 0.f;
    else
      pos = (m >> e) & 1u;
    g[e] = dy[e] * (pos ? 1.f : slope);
  }
}

template <int ACT>
__global__ void __launch_bounds__(256) bf16_bn_bwd_partial_kernel(const void* __restrict__ dy, int dy_pooled,
                                                                  const void* __restrict__ y,
                                                                  const unsigned char* __restrict__ mask,
                                                                  const void* __restrict__ x,
                                                                  const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, float slope,
                                                                  float* __restrict__ part, int B, int C, int Cb, int H,
                                                                  int W) {
  __shared__ float red[4][16];
  const int cb = blockIdx.y;
  const int HW = H * W, Wh = W >> 1, HWh = (H >> 1) * Wh;
  float mu[8], is[8], gm[8], bt[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = cb * 8 + e;
    const bool ok = c < C;
    mu[e] = ok ? mean[c] : 0.f;
    is[e] = ok ? invstd[c] : 0.f;
    gm[e] = ok ? gamma[c] : 0.f;
    bt[e] = (ok && beta != nullptr) ? beta[c] : 0.f;
  }
  float sg[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, sgx[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const size_t n = (size_t)B * HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / HW);
    const int p = (int)(i % HW);
    const size_t v = ((size_t)b * Cb + cb) * HW + p;
    float d[8], xv[8], yv[8], xh[8], g[8];
    if (dy_pooled) {
      const int h = p / W, w = p % W;
      unpack8(ldv(dy, ((size_t)b * Cb + cb) * HWh + (size_t)(h >> 1) * Wh + (w >> 1)), d);
#pragma unroll
      for (int e = 0; e < 8; ++e) d[e] *= 0.25f;
    } else {
      unpack8(ldv(dy, v), d);
    }
    unpack8(ldv(x, v), xv);
    if (ACT == 1) unpack8(ldv(y, v), yv);
    const unsigned mb = ACT == 3 ? mask[v] : 0u;
#pragma unroll
    for (int e = 0; e < 8; ++e) xh[e] = (xv[e] - mu[e]) * is[e];
    bwd_g<ACT>(d, yv, xh, gm, bt, slope, g, mb);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sg[e] += g[e];
      sgx[e] += g[e] * xh[e];
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float a = wave_sum(sg[e]), bsum = wave_sum(sgx[e]);
    if (lane == 0) {
      red[wave][e] = a;
      red[wave][8 + e] = bsum;
    }
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    const float s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    const int e = threadIdx.x & 7, which = threadIdx.x >> 3;
    const int c = cb * 8 + e;
    if (c < C) part[((size_t)blockIdx.x * C + c) * 2 + which] = s;
  }
}

// sums[c] = {sg, sgx} (fp64 over the slices, one wave per channel, fixed order); dgamma = sgx, dbeta = sg
__global__ void __launch_bounds__(64) bf16_bn_bwd_finalize_kernel(const float* __restrict__ part, int nslices, int C,
                                                                  float* __restrict__ sums, float* __restrict__ dgamma,
                                                                  float* __restrict__ dbeta) {
  const int c = blockIdx.x;
  double a = 0.0, b = 0.0;
  for (int s = threadIdx.x; s < nslices; s += 64) {
    const float2 v = *reinterpret_cast<const float2*>(part + ((size_t)s * C + c) * 2);
    a += (double)v.x;
    b += (double)v.y;
  }
  a = wave_sum(a);
  b = wave_sum(b);
  if (threadIdx.x == 0) {
    sums[2 * c + 0] = (float)a;
    sums[2 * c + 1] = (float)b;
    if (dgamma) dgamma[c] = (float)b;
    if (dbeta) dbeta[c] = (float)a;
  }
}

template <int ACT, bool QUAD>
__global__ void __launch_bounds__(256) bf16_bn_bwd_apply_kernel(const void* __restrict__ dy, int dy_pooled,
                                                                const void* __restrict__ y,
                                                                const unsigned char* __restrict__ mask,
                                                                const void* __restrict__ x,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ invstd,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float slope,
                                                                const float* __restrict__ sums, float inv_n,
                                                                void* __restrict__ dx, void* __restrict__ dz,
                                                                int dz_sum, int B, int C, int Cb, int H, int W) {
  extern __shared__ __attribute__((aligned(16))) float tab[];  // [Cb][6][8]: mean, invstd, gamma, beta, c1, c2
  for (int c = threadIdx.x; c < Cb * 8; c += blockDim.x) {
    const bool ok = c < C;
    float* t = tab + (size_t)(c >> 3) * 48 + (c & 7);
    t[0] = ok ? mean[c] : 0.f;
    t[8] = ok ? invstd[c] : 0.f;
    t[16] = ok ? gamma[c] : 0.f;
    t[24] = (ok && beta != nullptr) ? beta[c] : 0.f;
    t[32] = ok ? sums[2 * c] * inv_n : 0.f;
    t[40] = ok ? sums[2 * c + 1] * inv_n : 0.f;
  }
  __syncthreads();
  const int HW = H * W;
  if (!QUAD) {
    const size_t n = (size_t)B * Cb * HW;
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (size_t)gridDim.x * blockDim.x) {
      const int cb = (int)((v / HW) % Cb);
      float p[6][8], d[8], xv[8], yv[8], xh[8], g[8];
      load_tab<6>(tab, cb, p);
      unpack8(ldv(dy, v), d);
      unpack8(ldv(x, v), xv);
      if (ACT == 1) unpack8(ldv(y, v), yv);
      const unsigned mb = ACT == 3 ? mask[v] : 0u;
#pragma unroll
      for (int e = 0; e < 8; ++e) xh[e] = (xv[e] - p[0][e]) * p[1][e];
      bwd_g<ACT>(d, yv, xh, p[2], p[3], slope, g, mb);
      if (dz != nullptr) stv(dz, v, pack8(g));
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] = p[2][e] * p[1][e] * (g[e] - p[4][e] - xh[e] * p[5][e]);
      stv(dx, v, pack8(g));
    }
  } else {
    const int Hh = H >> 1, Wh = W >> 1, HWh = Hh * Wh;
    const size_t n = (size_t)B * Cb * HWh;
    for (size_t qd = (size_t)blockIdx.x * blockDim.x + threadIdx.x; qd < n; qd += (size_t)gridDim.x * blockDim.x) {
      const int wq = (int)(qd % Wh);
      const size_t t = qd / Wh;
      const int hq = (int)(t % Hh);
      const size_t bc = t / Hh;
      const int cb = (int)(bc % Cb);
      float p[6][8], dp[8];
      load_tab<6>(tab, cb, p);
      if (dy_pooled) {
        unpack8(ldv(dy, qd), dp);
#pragma unroll
        for (int e = 0; e < 8; ++e) dp[e] *= 0.25f;
      }
      float zs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int dw = 0; dw < 2; ++dw) {
          const size_t v = bc * HW + (size_t)(2 * hq + dh) * W + 2 * wq + dw;
          float d[8], xv[8], yv[8], xh[8], g[8];
          if (dy_pooled) {
#pragma unroll
            for (int e = 0; e < 8; ++e) d[e] = dp[e];
          } else {
            unpack8(ldv(dy, v), d);
          }
          unpack8(ldv(x, v), xv);
          if (ACT == 1) unpack8(ldv(y, v), yv);
          const unsigned mb = ACT == 3 ? mask[v] : 0u;
#pragma unroll
          for (int e = 0; e < 8; ++e) xh[e] = (xv[e] - p[0][e]) * p[1][e];
          bwd_g<ACT>(d, yv, xh, p[2], p[3], slope, g, mb);
          if (dz != nullptr && !dz_sum) stv(dz, v, pack8(g));
#pragma unroll
          for (int e = 0; e < 8; ++e) zs[e] += g[e];
#pragma unroll
          for (int e = 0; e < 8; ++e) g[e] = p[2][e] * p[1][e] * (g[e] - p[4][e] - xh[e] * p[5][e]);
          stv(dx, v, pack8(g));
        }
      if (dz != nullptr && dz_sum) stv(dz, qd, pack8(zs));
    }
  }
}

// dx[h][w] = sum of the 2x2 block of dy (adjoint of nn.Upsample(2,'nearest'))
__global__ void bf16_upsample2_bwd_kernel(const void* __restrict__ dy, void* __restrict__ dx, size_t nplanes, int Hs,
                                          int Ws) {
  const size_t n = nplanes * Hs * Ws;
  const int W = 2 * Ws;
  for (size_t qd = (size_t)blockIdx.x * blockDim.x + threadIdx.x; qd < n; qd += (size_t)gridDim.x * blockDim.x) {
    const int wq = (int)(qd % Ws);
    const size_t t = qd / Ws;
    const int hq = (int)(t % Hs);
    const size_t pl = t / Hs;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
      for (int dw = 0; dw < 2; ++dw) {
        float f[8];
        unpack8(ldv(dy, pl * 4 * Hs * Ws + (size_t)(2 * hq + dh) * W + 2 * wq + dw), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += f[e];
      }
    stv(dx, qd, pack8(s));
  }
}

__global__ void bf16_upsample2_fwd_kernel(const void* __restrict__ x, void* __restrict__ y, size_t nplanes, int Hs,
                                          int Ws) {
  const size_t n = nplanes * Hs * Ws;
  const int W = 2 * Ws;
  for (size_t qd = (size_t)blockIdx.x * blockDim.x + threadIdx.x; qd < n; qd += (size_t)gridDim.x * blockDim.x) {
    const int wq = (int)(qd % Ws);
    const size_t t = qd / Ws;
    const int hq = (int)(t % Hs);
    const size_t pl = t / Hs;
    const u32x4_t v = ldv(x, qd);
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
      for (int dw = 0; dw < 2; ++dw) stv(y, pl * 4 * Hs * Ws + (size_t)(2 * hq + dh) * W + 2 * wq + dw, v);
  }
}

__global__ void bf16_add_kernel(void* __restrict__ y, const void* __restrict__ x, size_t nvec) {
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
    float a[8], b[8];
    unpack8(ldv(y, v), a);
    unpack8(ldv(x, v), b);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] += b[e];
    stv(y, v, pack8(a));
  }
}

namespace {
unsigned grid_for(size_t n, int per_block = 256, unsigned cap = 256 * 16) {
  size_t g = (n + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (unsigned)g;
}
}  // namespace

extern "C" int sivae_bf16_from_f32_nchw(const float* src, void* dst, int B, int C, int H, int W, float scale,
                                        hipStream_t stream) {
  if (!src || !dst) return SIVAE_ERR_NULL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return SIVAE_ERR_SHAPE;
  const int Cb = bf16_cblocks(C);
  const size_t n = (size_t)B * Cb * H * W;
  hipLaunchKernelGGL(bf16_from_f32_nchw_kernel, dim3(grid_for(n)), dim3(256), 0, stream, src, dst, B, C, Cb, H * W,
                     scale);
  return sivae_launch_status();
}

extern "C" int sivae_bf16_to_f32_nchw(const void* src, float* dst, int B, int C, int H, int W, hipStream_t stream) {
  if (!src || !dst) return SIVAE_ERR_NULL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return SIVAE_ERR_SHAPE;
  const int Cb = bf16_cblocks(C);
  const size_t n = (size_t)B * Cb * H * W;
  hipLaunchKernelGGL(bf16_to_f32_nchw_kernel, dim3(grid_for(n)), dim3(256), 0, stream, src, dst, B, C, Cb, H * W);
  return sivae_launch_status();
}

extern "C" int sivae_bf16_im2col_kw5(const float* src, void* dst, int B, int C, int H, int W, int sgn,
                                     hipStream_t stream) {
  if (!src || !dst) return SIVAE_ERR_NULL;
  if (B <= 0 || C <= 0 || 5 * C > 16 || H <= 0 || W <= 0 || B > 65535) return SIVAE_ERR_SHAPE;
  if (sgn != 1 && sgn != -1) return SIVAE_ERR_MODE;
  const dim3 grid((unsigned)((H * W + 255) / 256 < 64 ? (H * W + 255) / 256 : 64), (unsigned)B);
  if (C == 1)
    hipLaunchKernelGGL(bf16_im2col_kw5_kernel<1>, grid, dim3(256), 0, stream, src, dst, H, W, sgn);
  else if (C == 2)
    hipLaunchKernelGGL(bf16_im2col_kw5_kernel<2>, grid, dim3(256), 0, stream, src, dst, H, W, sgn);
  else
    hipLaunchKernelGGL(bf16_im2col_kw5_kernel<3>, grid, dim3(256), 0, stream, src, dst, H, W, sgn);
  return sivae_launch_status();
}

extern "C" int sivae_bf16_fold_kw5(const float* g, const float* bias, float* dst, int B, int C, int H, int W, int sgn,
                                   hipStream_t stream) {
  if (!g || !dst) return SIVAE_ERR_NULL;
  if (B <= 0 || C <= 0 || 5 * C > 16 || H <= 0 || W <= 0 || B > 65535) return SIVAE_ERR_SHAPE;
  if (sgn != 1 && sgn != -1) return SIVAE_ERR_MODE;
  const dim3 grid((unsigned)((H * W + 255) / 256 < 64 ? (H * W + 255) / 256 : 64), (unsigned)B);
  if (C == 1)
    hipLaunchKernelGGL(bf16_fold_kw5_kernel<1>, grid, dim3(256), 0, stream, g, bias, dst, H, W, sgn);
  else if (C == 2)
    hipLaunchKernelGGL(bf16_fold_kw5_kernel<2>, grid, dim3(256), 0, stream, g, bias, dst, H, W, sgn);
  else
    hipLaunchKernelGGL(bf16_fold_kw5_kernel<3>, grid, dim3(256), 0, stream, g, bias, dst, H, W, sgn);
  return sivae_launch_status();
}

extern "C" size_t sivae_bf16_bn_signmask_bytes(int B, int C, int H, int W) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
  return (size_t)B * bf16_cblocks(C) * H * W;  // one byte per 8-channel pixel vector
}

// SEGMENTED batch: B = nseg * seg_images images, mean / invstd [nseg][C] (one set of batch statistics per pass of
// seg_images images), gamma / beta [C] — the bf16 twin of sivae_bn_apply_act_seg (bn.hip); seg_images == B is the plain form
extern "C" int sivae_bf16_bn_apply_act_seg(const void* x, const void* res, int res_up, const float* mean,
                                           const float* invstd, const float* gamma, const float* beta, float slope,
                                           void* y, void* y_pool, unsigned char* sign_mask, int B, int C, int H, int W,
                                           int seg_images, hipStream_t stream) {
  if (!x || !mean || !invstd || !gamma || !beta || (!y && !y_pool)) return SIVAE_ERR_NULL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return SIVAE_ERR_SHAPE;
  if (seg_images <= 0 || B % seg_images != 0) return SIVAE_ERR_SHAPE;
  if ((size_t)(B / seg_images) * bf16_cblocks(C) * 16 * sizeof(float) > 48 * 1024) return SIVAE_ERR_SHAPE;
  if (res_up && !res) return SIVAE_ERR_NULL;
  const bool quad = y_pool != nullptr || res_up;
  if (quad && ((H & 1) || (W & 1))) return SIVAE_ERR_SHAPE;
  const int Cb = bf16_cblocks(C);
  if ((unsigned long long)B * Cb >= 0xffffffffull) return SIVAE_ERR_RANGE;
  const size_t lds = (size_t)(B / seg_images) * Cb * 16 * sizeof(float);
  if (quad) {
    const size_t n = (size_t)B * Cb * (H / 2) * (W / 2);
    hipLaunchKernelGGL(bf16_bn_apply_kernel<true>, dim3(grid_for(n)), dim3(256), lds, stream, x, res, res_up, mean,
                       invstd, gamma, beta, slope, y, y_pool, sign_mask, B, C, Cb, H, W, seg_images);
  } else {
    if (!y) return SIVAE_ERR_NULL;
    const size_t n = (size_t)B * Cb * H * W;
    hipLaunchKernelGGL(bf16_bn_apply_kernel<false>, dim3(grid_for(n)), dim3(256), lds, stream, x, res, res_up, mean,
                       invstd, gamma, beta, slope, y, y_pool, sign_mask, B, C, Cb, H, W, seg_images);
  }
  return sivae_launch_status();
}

extern "C" int sivae_bf16_bn_apply_act(const void* x, const void* res, int res_up, const float* mean,
                                       const float* invstd, const float* gamma, const float* beta, float slope,
                                       void* y, void* y_pool, unsigned char* sign_mask, int B, int C, int H, int W,
                                       hipStream_t stream) {
  return sivae_bf16_bn_apply_act_seg(x, res, res_up, mean, invstd, gamma, beta, slope, y, y_pool, sign_mask, B, C, H, W,
                                     B, stream);
}

static int bn_bwd_slices(int B, int C, int H, int W) {
  const size_t per_cb = (size_t)B * H * W;
  const int Cb = bf16_cblocks(C);
  size_t s = (per_cb + 256 * 8 - 1) / (256 * 8);  // >= 8 vectors per thread
  const size_t want = (2048 + Cb - 1) / Cb;       // ~2048 blocks over the chip
  if (s > want) s = want;
  if (s < 1) s = 1;
  return (int)s;
}

extern "C" size_t sivae_bf16_bn_bwd_workspace_bytes(int B, int C, int H, int W) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
  return ((size_t)bn_bwd_slices(B, C, H, W) * C * 2 + (size_t)C * 2) * sizeof(float);
}

extern "C" int sivae_bf16_bn_bwd(const void* dy, int dy_pooled, const void* y, const unsigned char* sign_mask,
                                 const void* x, const float* mean, const float* invstd, const float* gamma,
                                 const float* beta, float slope, void* dx, void* dz, int dz_sum, float* dgamma,
                                 float* dbeta, int B, int C, int H, int W, void* workspace, size_t workspace_bytes,
                                 hipStream_t stream) {
  if (!dy || !x || !mean || !invstd || !gamma || !dx || !workspace) return SIVAE_ERR_NULL;
  // the activation sign comes from the sign mask, from the saved output y, or is recomputed (needs beta)
  if (!y && !sign_mask && !beta) return SIVAE_ERR_NULL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return SIVAE_ERR_SHAPE;
  if (dz_sum && !dz) return SIVAE_ERR_NULL;
  const bool quad = dy_pooled || dz_sum;
  if (quad && ((H & 1) || (W & 1))) return SIVAE_ERR_SHAPE;
  if (workspace_bytes < sivae_bf16_bn_bwd_workspace_bytes(B, C, H, W)) return SIVAE_ERR_WORKSPACE;
  const int Cb = bf16_cblocks(C);
  const int ns = bn_bwd_slices(B, C, H, W);
  float* part = reinterpret_cast<float*>(workspace);
  float* sums = part + (size_t)ns * C * 2;
  const int act = sign_mask ? 3 : (y ? 1 : 2);
  if (act == 3)
    hipLaunchKernelGGL(bf16_bn_bwd_partial_kernel<3>, dim3(ns, Cb), dim3(256), 0, stream, dy, dy_pooled, y, sign_mask, x,
                       mean, invstd, gamma, beta, slope, part, B, C, Cb, H, W);
  else if (act == 1)
    hipLaunchKernelGGL(bf16_bn_bwd_partial_kernel<1>, dim3(ns, Cb), dim3(256), 0, stream, dy, dy_pooled, y, sign_mask, x,
                       mean, invstd, gamma, beta, slope, part, B, C, Cb, H, W);
  else
    hipLaunchKernelGGL(bf16_bn_bwd_partial_kernel<2>, dim3(ns, Cb), dim3(256), 0, stream, dy, dy_pooled, y, sign_mask, x,
                       mean, invstd, gamma, beta, slope, part, B, C, Cb, H, W);
  hipLaunchKernelGGL(bf16_bn_bwd_finalize_kernel, dim3(C), dim3(64), 0, stream, part, ns, C, sums, dgamma, dbeta);
  const float inv_n = 1.0f / ((float)B * H * W);
  const size_t lds = (size_t)Cb * 48 * sizeof(float);
  const size_t n = quad ? (size_t)B * Cb * (H / 2) * (W / 2) : (size_t)B * Cb * H * W;
#define SIVAE_BWD_APPLY(ACT, Q)                                                                                      \
  hipLaunchKernelGGL((bf16_bn_bwd_apply_kernel<ACT, Q>), dim3(grid_for(n)), dim3(256), lds, stream, dy, dy_pooled, y, \
                     sign_mask, x, mean, invstd, gamma, beta, slope, sums, inv_n, dx, dz, dz_sum, B, C, Cb, H, W)
  if (act == 3) {
    if (quad)
      SIVAE_BWD_APPLY(3, true);
    else
      SIVAE_BWD_APPLY(3, false);
  } else if (act == 1) {
    if (quad)
      SIVAE_BWD_APPLY(1, true);
    else
      SIVAE_BWD_APPLY(1, false);
  } else {
    if (quad)
      SIVAE_BWD_APPLY(2, true);
    else
      SIVAE_BWD_APPLY(2, false);
  }
#undef SIVAE_BWD_APPLY
  return sivae_launch_status();
}

extern "C" int sivae_bf16_upsample2_bwd(const void* dy, void* dx, int B, int C, int Hs, int Ws, hipStream_t stream) {
  if (!dy || !dx) return SIVAE_ERR_NULL;
  if (B <= 0 || C <= 0 || Hs <= 0 || Ws <= 0) return SIVAE_ERR_SHAPE;
  const size_t planes = (size_t)B * bf16_cblocks(C);
  hipLaunchKernelGGL(bf16_upsample2_bwd_kernel, dim3(grid_for(planes * Hs * Ws)), dim3(256), 0, stream, dy, dx, planes,
                     Hs, Ws);
  return sivae_launch_status();
}

extern "C" int sivae_bf16_upsample2_fwd(const void* x, void* y, int B, int C, int Hs, int Ws, hipStream_t stream) {
  if (!x || !y) return SIVAE_ERR_NULL;
  if (B <= 0 || C <= 0 || Hs <= 0 || Ws <= 0) return SIVAE_ERR_SHAPE;
  const size_t planes = (size_t)B * bf16_cblocks(C);
  hipLaunchKernelGGL(bf16_upsample2_fwd_kernel, dim3(grid_for(planes * Hs * Ws)), dim3(256), 0, stream, x, y, planes,
                     Hs, Ws);
  return sivae_launch_status();
}

extern "C" int sivae_bf16_add_inplace(void* y, const void* x, size_t nvec, hipStream_t stream) {
  if (!y || !x) return SIVAE_ERR_NULL;
  if (nvec == 0) return SIVAE_OK;
  hipLaunchKernelGGL(bf16_add_kernel, dim3(grid_for(nvec)), dim3(256), 0, stream, y, x, nvec);
  return sivae_launch_status();
}
