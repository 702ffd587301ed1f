// HBM-bound elementwise kernels: 2x2 average pool, nearest 2x upsample (and their adjoints), ReLU,
// axpy-style accumulate.  NCHW fp32, one thread per OUTPUT element group, float2/float4 accesses
// along W so both sides of each kernel are coalesced.
//
// Reference ops being replaced: nn.AvgPool2d(2) (train_soft_intro_vae.py:92,98),
// nn.Upsample(scale_factor=2, mode='nearest') (:155), nn.ReLU(True) (:147).
#include "common.h"

// y[b,c,h,w] = 0.25 * sum_{i,j<2} x[b,c,2h+i,2w+j]      (H, W = OUTPUT size)
__global__ void __launch_bounds__(256) avgpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int H,
                                                           int W, size_t n_out) {
  const size_t stride = (size_t)gridDim.x * 256;
  const int W2 = 2 * W;
  for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < n_out; o += stride) {
    const int w = (int)(o % W);
    const size_t t = o / W;  // (b*C + c)*H + h
    const float* p = x + (t * 2) * W2 + 2 * w;
    const float2 a = *reinterpret_cast<const float2*>(p);
    const float2 b = *reinterpret_cast<const float2*>(p + W2);
    y[o] = ((a.x + a.y) + (b.x + b.y)) * 0.25f;
  }
}

// dx[b,c,2h+i,2w+j] = 0.25 * dy[b,c,h,w]
__global__ void __launch_bounds__(256) avgpool2_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                           int H, int W, size_t n_out) {
  const size_t stride = (size_t)gridDim.x * 256;
  const int W2 = 2 * W;
  for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < n_out; o += stride) {
    const int w = (int)(o % W);
    const size_t t = o / W;
    const float g = dy[o] * 0.25f;
    float* p = dx + (t * 2) * W2 + 2 * w;
    *reinterpret_cast<float2*>(p) = make_float2(g, g);
    *reinterpret_cast<float2*>(p + W2) = make_float2(g, g);
  }
}

// y[b,c,2h+i,2w+j] = x[b,c,h,w]       (H, W = INPUT size)
__global__ void __launch_bounds__(256) upsample2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int H,
                                                            int W, size_t n_in) {
  const size_t stride = (size_t)gridDim.x * 256;
  const int W2 = 2 * W;
  for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < n_in; o += stride) {
    const int w = (int)(o % W);
    const size_t t = o / W;
    const float v = x[o];
    float* p = y + (t * 2) * W2 + 2 * w;
    *reinterpret_cast<float2*>(p) = make_float2(v, v);
    *reinterpret_cast<float2*>(p + W2) = make_float2(v, v);
  }
}

// dx[b,c,h,w] = sum_{i,j<2} dy[b,c,2h+i,2w+j]
__global__ void __launch_bounds__(256) upsample2_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                            int H, int W, size_t n_in) {
  const size_t stride = (size_t)gridDim.x * 256;
  const int W2 = 2 * W;
  for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < n_in; o += stride) {
    const int w = (int)(o % W);
    const size_t t = o / W;
    const float* p = dy + (t * 2) * W2 + 2 * w;
    const float2 a = *reinterpret_cast<const float2*>(p);
    const float2 b = *reinterpret_cast<const float2*>(p + W2);
    dx[o] = (a.x + a.y) + (b.x + b.y);
  }
}

// general AvgPool2d(2) for odd input sizes (floor semantics, e.g. 7x7 -> 3x3): scalar accesses
__global__ void __launch_bounds__(256) avgpool2_fwd_odd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                               int Hin, int Win, size_t n_out) {
  const size_t stride = (size_t)gridDim.x * 256;
  const int H = Hin >> 1, W = Win >> 1;
  for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < n_out; o += stride) {
    const int w = (int)(o % W);
    const size_t t = o / W;
    const int h = (int)(t % H);
    const size_t row = t / H;
    const float* p = x + (row * Hin + 2 * h) * Win + 2 * w;
    y[o] = ((p[0] + p[1]) + (p[Win] + p[Win + 1])) * 0.25f;
  }
}
__global__ void __launch_bounds__(256) avgpool2_bwd_odd_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                               int Hin, int Win, size_t n_in) {
  const size_t stride = (size_t)gridDim.x * 256;
  const int H = Hin >> 1, W = Win >> 1;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_in; i += stride) {
    const int wi = (int)(i % Win);
    const size_t t = i / Win;
    const int hi = (int)(t % Hin);
    const size_t row = t / Hin;
    const int h = hi >> 1, w = wi >> 1;
    dx[i] = (h < H && w < W) ? 0.25f * dy[(row * H + h) * W + w] : 0.f;
  }
}

__global__ void __launch_bounds__(256) relu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) y[i] = fmaxf(x[i], 0.f);
}
// dx = dy * (y > 0)
__global__ void __launch_bounds__(256) relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                       float* __restrict__ dx, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}
// y += x
__global__ void __launch_bounds__(256) add_inplace_kernel(float* __restrict__ y, const float* __restrict__ x, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  const size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 a = reinterpret_cast<float4*>(y)[i];
    const float4 b = reinterpret_cast<const float4*>(x)[i];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    reinterpret_cast<float4*>(y)[i] = a;
  }
  for (size_t i = (n4 << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) y[i] += x[i];
}

static inline int grid_for(size_t n) {
  long long nb = (long long)((n + 255) / 256);
  if (nb > 8192) nb = 8192;
  if (nb < 1) nb = 1;
  return (int)nb;
}

// Hin, Win = INPUT (large side) size of the pool; output is (Hin/2, Win/2), floor semantics; rows = B*C.
extern "C" int sivae_avgpool2_fwd(const float* x, float* y, int rows, int Hin, int Win, hipStream_t stream) {
  if (!x || !y) return SIVAE_ERR_NULL;
  if (rows <= 0 || Hin < 2 || Win < 2) return SIVAE_ERR_SHAPE;
  const int H = Hin >> 1, W = Win >> 1;
  const size_t n = (size_t)rows * H * W;
  if ((Hin | Win) & 1)
    hipLaunchKernelGGL(avgpool2_fwd_odd_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, y, Hin, Win, n);
  else
    hipLaunchKernelGGL(avgpool2_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, y, H, W, n);
  return sivae_launch_status();
}
extern "C" int sivae_avgpool2_bwd(const float* dy, float* dx, int rows, int Hin, int Win, hipStream_t stream) {
  if (!dy || !dx) return SIVAE_ERR_NULL;
  if (rows <= 0 || Hin < 2 || Win < 2) return SIVAE_ERR_SHAPE;
  const int H = Hin >> 1, W = Win >> 1;
  if ((Hin | Win) & 1) {
    const size_t n = (size_t)rows * Hin * Win;
    hipLaunchKernelGGL(avgpool2_bwd_odd_kernel, dim3(grid_for(n)), dim3(256), 0, stream, dy, dx, Hin, Win, n);
  } else {
    const size_t n = (size_t)rows * H * W;
    hipLaunchKernelGGL(avgpool2_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, stream, dy, dx, H, W, n);
  }
  return sivae_launch_status();
}
// H, W = INPUT (small side) size of the upsample; output is (2H, 2W); rows = B*C.
extern "C" int sivae_upsample2_fwd(const float* x, float* y, int rows, int H, int W, hipStream_t stream) {
  if (!x || !y) return SIVAE_ERR_NULL;
  if (rows <= 0 || H <= 0 || W <= 0) return SIVAE_ERR_SHAPE;
  const size_t n = (size_t)rows * H * W;
  hipLaunchKernelGGL(upsample2_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, y, H, W, n);
  return sivae_launch_status();
}
extern "C" int sivae_upsample2_bwd(const float* dy, float* dx, int rows, int H, int W, hipStream_t stream) {
  if (!dy || !dx) return SIVAE_ERR_NULL;
  if (rows <= 0 || H <= 0 || W <= 0) return SIVAE_ERR_SHAPE;
  const size_t n = (size_t)rows * H * W;
  hipLaunchKernelGGL(upsample2_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, stream, dy, dx, H, W, n);
  return sivae_launch_status();
}
extern "C" int sivae_relu_fwd(const float* x, float* y, size_t n, hipStream_t stream) {
  if (!x || !y) return SIVAE_ERR_NULL;
  if (n == 0) return SIVAE_ERR_SHAPE;
  hipLaunchKernelGGL(relu_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, y, n);
  return sivae_launch_status();
}
extern "C" int sivae_relu_bwd(const float* dy, const float* y, float* dx, size_t n, hipStream_t stream) {
  if (!dy || !y || !dx) return SIVAE_ERR_NULL;
  if (n == 0) return SIVAE_ERR_SHAPE;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, stream, dy, y, dx, n);
  return sivae_launch_status();
}
extern "C" int sivae_add_inplace(float* y, const float* x, size_t n, hipStream_t stream) {
  if (!x || !y) return SIVAE_ERR_NULL;
  if (n == 0) return SIVAE_ERR_SHAPE;
  hipLaunchKernelGGL(add_inplace_kernel, dim3(grid_for(n >> 2 ? n >> 2 : n)), dim3(256), 0, stream, y, x, n);
  return sivae_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Input side of the path: uint8 image batch (as decoded: NCHW or NHWC) -> fp32 NCHW in [0, 1], with the
// per-sample horizontal mirror of the reference's dataset (dataset.py:27-28,46,68-70: random mirror, then
// transforms.ToTensor() = /255, train_soft_intro_vae.py:379).  Runs on the prefetch stream, so the step
// itself never sees a host tensor.  flip[b] != 0 mirrors sample b; flip may be null.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) u8_to_f32_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst,
                                                        const int* __restrict__ flip, int C, int H, int W, int nhwc,
                                                        float scale, size_t numel) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < numel; o += stride) {
    const int w = (int)(o % W);
    size_t t = o / W;
    const int h = (int)(t % H);
    t /= H;
    const int c = (int)(t % C);
    const size_t b = t / C;
    const int ws = (flip != nullptr && flip[b] != 0) ? (W - 1 - w) : w;
    const size_t si = nhwc ? (((b * H + h) * W + ws) * C + c) : (((b * C + c) * H + h) * W + ws);
    dst[o] = (float)src[si] * scale;
  }
}

extern "C" int sivae_u8_to_f32(const unsigned char* src, float* dst, const int* flip, int B, int C, int H, int W,
                               int nhwc, float scale, hipStream_t stream) {
  if (!src || !dst) return SIVAE_ERR_NULL;
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return SIVAE_ERR_SHAPE;
  const size_t numel = (size_t)B * C * H * W;
  int nb = cdiv((long long)numel, 256 * 4);
  if (nb > 8192) nb = 8192;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(u8_to_f32_kernel, dim3(nb), dim3(256), 0, stream, src, dst, flip, C, H, W, nhwc, scale, numel);
  return sivae_launch_status();
}

// ---- output side: generated images fp32 NCHW -> uint8, the quantisation the reference applies to every
// sample batch before the FID network sees it (metrics/fid_score.py:247-249: np.clip(images * 255, 0, 255)
// .astype(np.uint8) — clip, then truncate toward zero; NaN -> 0 here).  16 elements per thread, 16-byte stores.
__global__ void __launch_bounds__(256) f32_to_u8_kernel(const float* __restrict__ src, unsigned char* __restrict__ dst,
                                                        float scale, size_t numel) {
  const size_t stride = (size_t)gridDim.x * 256;
  const size_t n16 = numel >> 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
    unsigned w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = reinterpret_cast<const float4*>(src)[i * 4 + q];
      const float f[4] = {v.x, v.y, v.z, v.w};
      unsigned word = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float t = f[k] * scale;
        t = t > 0.f ? t : 0.f;  // (also NaN -> 0)
        t = t < 255.f ? t : 255.f;
        word |= (unsigned)(int)t << (8 * k);
      }
      w[q] = word;
    }
    reinterpret_cast<uint4*>(dst)[i] = make_uint4(w[0], w[1], w[2], w[3]);
  }
  // ragged tail
  for (size_t e = (n16 << 4) + (size_t)blockIdx.x * 256 + threadIdx.x; e < numel; e += stride) {
    float t = src[e] * scale;
    t = t > 0.f ? t : 0.f;
    t = t < 255.f ? t : 255.f;
    dst[e] = (unsigned char)(int)t;
  }
}

extern "C" int sivae_f32_to_u8(const float* src, unsigned char* dst, size_t numel, float scale, hipStream_t stream) {
  if (!src || !dst) return SIVAE_ERR_NULL;
  if (numel == 0) return SIVAE_ERR_SHAPE;
  if ((((uintptr_t)src) & 15u) || (((uintptr_t)dst) & 15u)) return SIVAE_ERR_SHAPE;  // 16-byte vector accesses
  int nb = cdiv((long long)((numel >> 4) + 1), 256);
  if (nb > 8192) nb = 8192;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(f32_to_u8_kernel, dim3(nb), dim3(256), 0, stream, src, dst, scale, numel);
  return sivae_launch_status();
}
