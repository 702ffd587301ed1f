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
