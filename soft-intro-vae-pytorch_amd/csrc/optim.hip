// Fused Adam step over a flat fp32 parameter buffer (one launch per network instead of ~6 tiny
// elementwise launches per tensor).  Arithmetic follows torch.optim.Adam's single-tensor path:
//   exp_avg.lerp_(g, 1-b1); exp_avg_sq = b2*exp_avg_sq + (1-b2)*g*g;
//   denom = sqrt(exp_avg_sq)/sqrt(1-b2^t) + eps; p -= (lr/(1-b1^t)) * exp_avg/denom
// Reference: optim.Adam(model.encoder.parameters(), lr=lr_e) soft_intro_vae/train_soft_intro_vae.py:450-451,589,624.
#include "common.h"

__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, size_t n,
                                                   float step_size, float beta1, float beta2, float eps,
                                                   float bc2_sqrt, float grad_scale) {
  const size_t stride = (size_t)gridDim.x * 256;
  const float w1 = 1.f - beta1, w2 = 1.f - beta2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float gi = g[i] * grad_scale;
    float mi = m[i], vi = v[i];
    mi = mi + w1 * (gi - mi);
    vi = vi * beta2 + w2 * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
  }
}

extern "C" int sivae_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                               float step_size, float beta1, float beta2, float eps, float bias_correction2_sqrt,
                               float grad_scale, hipStream_t stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq) return SIVAE_ERR_NULL;
  if (n == 0) return SIVAE_ERR_SHAPE;
  long long nb = (long long)((n + 255) / 256);
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(adam_kernel, dim3((int)nb), dim3(256), 0, stream, param, grad, exp_avg, exp_avg_sq, n, step_size,
                     beta1, beta2, eps, bias_correction2_sqrt, grad_scale);
  return sivae_launch_status();
}

// ---- the same step with its state on the device, for whole-iteration HIP graphs: the step count lives in
// state[0] (as a double), the learning rate in state[1]; a one-thread tick advances the count and derives the two
// bias-correction factors in double precision exactly as the host path does (python floats), so eager and
// graph-replayed runs agree.   state = double[4]: {t, lr, lr/(1-b1^t), sqrt(1-b2^t)}
__global__ void adam_tick_kernel(double* state, double beta1, double beta2) {
  const double t = state[0] + 1.0;
  state[0] = t;
  state[2] = state[1] / (1.0 - pow(beta1, t));
  state[3] = sqrt(1.0 - pow(beta2, t));
}

__global__ void __launch_bounds__(256) adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v, size_t n,
                                                       const double* __restrict__ state, float beta1, float beta2,
                                                       float eps, float grad_scale) {
  const float step_size = (float)state[2], bc2_sqrt = (float)state[3];
  const size_t stride = (size_t)gridDim.x * 256;
  const float w1 = 1.f - beta1, w2 = 1.f - beta2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float gi = g[i] * grad_scale;
    float mi = m[i], vi = v[i];
    mi = mi + w1 * (gi - mi);
    vi = vi * beta2 + w2 * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
  }
}

extern "C" int sivae_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                                   double* state, double beta1, double beta2, float eps, float grad_scale,
                                   hipStream_t stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !state) return SIVAE_ERR_NULL;
  if (n == 0) return SIVAE_ERR_SHAPE;
  long long nb = (long long)((n + 255) / 256);
  if (nb > 8192) nb = 8192;
  // (betas arrive as doubles: the bias corrections must come from 0.999, not from 0.999f widened)
  hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, stream, state, beta1, beta2);
  hipLaunchKernelGGL(adam_dev_kernel, dim3((int)nb), dim3(256), 0, stream, param, grad, exp_avg, exp_avg_sq, n,
                     (const double*)state, (float)beta1, (float)beta2, eps, grad_scale);
  return sivae_launch_status();
}

// flat_grad[i] += s0[i] + s1[i] + s2[i] + s3[i]  (s1 .. s3 may be NULL).  A parameter used by several passes of one backward gets
// one gradient per use; each use writes its own slab (no read-modify-write, no per-tensor add launches) and this single
// launch folds the slabs into the buffer the all-reduce and the fused Adam read (sivae_hip/optim.py).
__global__ void __launch_bounds__(256) sum_slabs_kernel(float* __restrict__ g, const float* __restrict__ s0,
                                                        const float* __restrict__ s1, const float* __restrict__ s2,
                                                        const float* __restrict__ s3, size_t n4, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 v = reinterpret_cast<float4*>(g)[i];
    const float4 a = reinterpret_cast<const float4*>(s0)[i];
    v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    if (s1) {
      const float4 b = reinterpret_cast<const float4*>(s1)[i];
      v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    if (s2) {
      const float4 c = reinterpret_cast<const float4*>(s2)[i];
      v.x += c.x; v.y += c.y; v.z += c.z; v.w += c.w;
    }
    if (s3) {
      const float4 d = reinterpret_cast<const float4*>(s3)[i];
      v.x += d.x; v.y += d.y; v.z += d.z; v.w += d.w;
    }
    reinterpret_cast<float4*>(g)[i] = v;
  }
  // ragged tail (n not a multiple of 4)
  for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    g[i] = (((g[i] + s0[i]) + (s1 ? s1[i] : 0.f)) + (s2 ? s2[i] : 0.f)) + (s3 ? s3[i] : 0.f);
}

extern "C" int sivae_sum_slabs(float* grad, const float* s0, const float* s1, const float* s2, const float* s3,
                               size_t n, hipStream_t stream) {
  if (!grad || !s0) return SIVAE_ERR_NULL;
  if (n == 0) return SIVAE_OK;
  const bool al = (((uintptr_t)grad | (uintptr_t)s0 | (uintptr_t)s1 | (uintptr_t)s2 | (uintptr_t)s3) & 15u) == 0;
  const size_t n4 = al ? n / 4 : 0;
  size_t nb = ((al ? n4 : n) + 255) / 256;
  if (nb > 4096) nb = 4096;
  if (nb == 0) nb = 1;
  hipLaunchKernelGGL(sum_slabs_kernel, dim3((unsigned)nb), dim3(256), 0, stream, grad, s0, s1, s2, s3, n4, n);
  return sivae_launch_status();
}
