// Sampler + loss kernels of the Soft-IntroVAE objective (fp32 in / fp64 reductions), gfx950.
//   reparameterize          z = mu + eps * exp(0.5 logvar)            train_soft_intro_vae.py:254-265
//   calc_kl                 per-sample KL to N(mu_o, exp(logvar_o))   :231-251
//   calc_reconstruction_loss  per-sample sum of mse / l1 / bce        :268-294
//   expELBO                 mean_i exp(-2 s (b_rec L_i + b_neg KL_i)) :580-581
// Row reductions are one wave64 per latent row (KL) or a 2-stage block reduction (image rows), all
// combined in a fixed order -> reproducible.  mu / logvar are addressed with a leading dimension so
// the two halves of the encoder fc output are consumed in place (no chunk copies).
#include "common.h"

// ---------------------------------------------------------------- reparameterize
__global__ void __launch_bounds__(256) reparam_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ lv,
                                                          int ld, const float* __restrict__ eps,
                                                          float* __restrict__ z, int B, int Z) {
  const int n = B * Z;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int b = i / Z, j = i - b * Z;
    const float m = mu[(size_t)b * ld + j], l = lv[(size_t)b * ld + j];
    z[i] = m + eps[i] * expf(0.5f * l);
  }
}
// dmu = dz ; dlogvar = dz * eps * 0.5 * exp(0.5 logvar)   (written with leading dimension ldg)
__global__ void __launch_bounds__(256) reparam_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ lv,
                                                          int ld, const float* __restrict__ eps,
                                                          float* __restrict__ dmu, float* __restrict__ dlv, int ldg,
                                                          int B, int Z) {
  const int n = B * Z;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int b = i / Z, j = i - b * Z;
    const float g = dz[i];
    dmu[(size_t)b * ldg + j] = g;
    dlv[(size_t)b * ldg + j] = g * eps[i] * 0.5f * expf(0.5f * lv[(size_t)b * ld + j]);
  }
}

// ---------------------------------------------------------------- KL
// kl_b = -0.5 * sum_j (1 + lv - lvo - exp(lv)/exp(lvo) - (mu-muo)^2/exp(lvo))
__global__ void __launch_bounds__(64) kl_fwd_kernel(const float* __restrict__ lv, const float* __restrict__ mu, int ld,
                                                    float mu_o, float lv_o, float* __restrict__ out, int Z) {
  const int b = blockIdx.x;
  const float elvo = expf(lv_o);
  double acc = 0.0;
  for (int j = threadIdx.x; j < Z; j += 64) {
    const float l = lv[(size_t)b * ld + j], m = mu[(size_t)b * ld + j];
    const float d = m - mu_o;
    const float t = 1.f + l - lv_o - expf(l) / elvo - d * d / elvo;
    acc += (double)t;
  }
  acc = wave_sum(acc);
  if (threadIdx.x == 0) out[b] = (float)(-0.5 * acc);
}
// g is per-sample upstream grad [B] (g_stride 1) or a broadcast scalar (g_stride 0), times g_scale
__global__ void __launch_bounds__(256) kl_bwd_kernel(const float* __restrict__ g, int g_stride, float g_scale,
                                                     const float* __restrict__ lv, const float* __restrict__ mu,
                                                     int ld, float mu_o, float lv_o, float* __restrict__ dlv,
                                                     float* __restrict__ dmu, int ldg, int B, int Z) {
  const int n = B * Z;
  const float elvo = expf(lv_o);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int b = i / Z, j = i - b * Z;
    const float gg = g[(size_t)b * g_stride] * g_scale;
    const float l = lv[(size_t)b * ld + j], m = mu[(size_t)b * ld + j];
    dlv[(size_t)b * ldg + j] = gg * (-0.5f) * (1.f - expf(l) / elvo);
    dmu[(size_t)b * ldg + j] = gg * (m - mu_o) / elvo;
  }
}

// calc_kl with TENSOR priors (train_soft_intro_vae.py:231-251 accepts tensors for mu_o / logvar_o): the prior parameters
// are read on the device through broadcast strides (row stride, column stride; 0 = broadcast), so a 0-d, [Z], [1, Z],
// [B, 1] or [B, Z] prior needs no host round trip.
__global__ void __launch_bounds__(64) kl_fwd_t_kernel(const float* __restrict__ lv, const float* __restrict__ mu, int ld,
                                                      const float* __restrict__ mo, int mo_rs, int mo_cs,
                                                      const float* __restrict__ lo, int lo_rs, int lo_cs,
                                                      float* __restrict__ out, int Z) {
  const int b = blockIdx.x;
  double acc = 0.0;
  for (int j = threadIdx.x; j < Z; j += 64) {
    const float l = lv[(size_t)b * ld + j], m = mu[(size_t)b * ld + j];
    const float mu_o = mo[(size_t)b * mo_rs + (size_t)j * mo_cs], lv_o = lo[(size_t)b * lo_rs + (size_t)j * lo_cs];
    const float elvo = expf(lv_o);
    const float d = m - mu_o;
    const float t = 1.f + l - lv_o - expf(l) / elvo - d * d / elvo;
    acc += (double)t;
  }
  acc = wave_sum(acc);
  if (threadIdx.x == 0) out[b] = (float)(-0.5 * acc);
}
__global__ void __launch_bounds__(256) kl_bwd_t_kernel(const float* __restrict__ g, int g_stride, float g_scale,
                                                       const float* __restrict__ lv, const float* __restrict__ mu,
                                                       int ld, const float* __restrict__ mo, int mo_rs, int mo_cs,
                                                       const float* __restrict__ lo, int lo_rs, int lo_cs,
                                                       float* __restrict__ dlv, float* __restrict__ dmu, int ldg, int B,
                                                       int Z) {
  const int n = B * Z;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int b = i / Z, j = i - b * Z;
    const float gg = g[(size_t)b * g_stride] * g_scale;
    const float l = lv[(size_t)b * ld + j], m = mu[(size_t)b * ld + j];
    const float mu_o = mo[(size_t)b * mo_rs + (size_t)j * mo_cs];
    const float elvo = expf(lo[(size_t)b * lo_rs + (size_t)j * lo_cs]);
    dlv[(size_t)b * ldg + j] = gg * (-0.5f) * (1.f - expf(l) / elvo);
    dmu[(size_t)b * ldg + j] = gg * (m - mu_o) / elvo;
  }
}

// ---------------------------------------------------------------- reconstruction losses
// TYPE 0 = mse (r-x)^2, 1 = l1 |r-x|, 2 = bce -(x log r + (1-x) log(1-r)) with logs clamped at -100
template <int TYPE>
__device__ __forceinline__ float recon_term(float x, float r) {
  if (TYPE == 0) {
    const float d = r - x;
    return d * d;
  } else if (TYPE == 1) {
    return fabsf(r - x);
  } else {
    const float lr = fmaxf(logf(r), -100.f), l1r = fmaxf(logf(1.f - r), -100.f);
    return -(x * lr + (1.f - x) * l1r);
  }
}
template <int TYPE>
__device__ __forceinline__ float recon_drecon(float x, float r) {
  if (TYPE == 0) return 2.f * (r - x);
  if (TYPE == 1) return (r > x) ? 1.f : ((r < x) ? -1.f : 0.f);
  return (r - x) / fmaxf((1.f - r) * r, 1e-12f);
}
template <int TYPE>
__device__ __forceinline__ float recon_dtarget(float x, float r) {
  if (TYPE == 0) return -2.f * (r - x);
  if (TYPE == 1) return (r > x) ? -1.f : ((r < x) ? 1.f : 0.f);
  return fmaxf(logf(1.f - r), -100.f) - fmaxf(logf(r), -100.f);
}

// partial[b][s] = sum over slice s of row b
template <int TYPE>
__global__ void __launch_bounds__(256) recon_rowsum_partial_kernel(const float* __restrict__ x,
                                                                   const float* __restrict__ r,
                                                                   double* __restrict__ part, int D, int S,
                                                                   int slice_len) {
  __shared__ double red[4];
  const int s = blockIdx.x, b = blockIdx.y;
  const int i0 = s * slice_len;
  int i1 = i0 + slice_len;
  if (i1 > D) i1 = D;
  const float* xr = x + (size_t)b * D;
  const float* rr = r + (size_t)b * D;
  double acc = 0.0;
  if ((D & 3) == 0) {
    for (int i = i0 + threadIdx.x * 4; i < i1; i += 1024) {
      const float4 a = *reinterpret_cast<const float4*>(xr + i);
      const float4 c = *reinterpret_cast<const float4*>(rr + i);
      acc += (double)recon_term<TYPE>(a.x, c.x) + (double)recon_term<TYPE>(a.y, c.y) +
             (double)recon_term<TYPE>(a.z, c.z) + (double)recon_term<TYPE>(a.w, c.w);
    }
  } else {
    for (int i = i0 + threadIdx.x; i < i1; i += 256) acc += (double)recon_term<TYPE>(xr[i], rr[i]);
  }
  acc = block_sum<256>(acc, red);
  if (threadIdx.x == 0) part[(size_t)b * S + s] = acc;
}
__global__ void __launch_bounds__(64) rowsum_finalize_kernel(const double* __restrict__ part, int S, int B,
                                                             float* __restrict__ out) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  double a = 0.0;
  for (int s = 0; s < S; ++s) a += part[(size_t)b * S + s];
  out[b] = (float)a;
}
// d_recon = g * dl/dr ; d_x = g * dl/dx.   g: per-row [B] (mode 0), scalar (mode 1), per-element (mode 2)
template <int TYPE>
__global__ void __launch_bounds__(256) recon_bwd_kernel(const float* __restrict__ x, const float* __restrict__ r,
                                                        const float* __restrict__ g, int g_mode, float g_scale,
                                                        float* __restrict__ d_r, float* __restrict__ d_x, int D,
                                                        size_t numel) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < numel; i += stride) {
    const float gg = (g_mode == 0 ? g[i / D] : (g_mode == 1 ? g[0] : g[i])) * g_scale;
    const float xv = x[i], rv = r[i];
    if (d_r) d_r[i] = gg * recon_drecon<TYPE>(xv, rv);
    if (d_x) d_x[i] = gg * recon_dtarget<TYPE>(xv, rv);
  }
}
template <int TYPE>
__global__ void __launch_bounds__(256) recon_elem_fwd_kernel(const float* __restrict__ x, const float* __restrict__ r,
                                                             float* __restrict__ out, size_t numel) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < numel; i += stride) out[i] = recon_term<TYPE>(x[i], r[i]);
}

// ---------------------------------------------------------------- batch-vector reductions
// out[0] = scale * sum_b v[b]   (one block, fixed order)
__global__ void __launch_bounds__(256) vec_sum_kernel(const float* __restrict__ v, int n, float scale,
                                                      float* __restrict__ out) {
  __shared__ double red[4];
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) a += (double)v[i];
  a = block_sum<256>(a, red);
  if (threadIdx.x == 0) out[0] = (float)(a * (double)scale);
}
// e[b] = exp(-2 s (b_rec L[b] + b_neg KL[b])) ; out[0] = mean_b e[b]
__global__ void __launch_bounds__(256) expelbo_fwd_kernel(const float* __restrict__ L, const float* __restrict__ KL,
                                                          float scale, float beta_rec, float beta_neg, int B,
                                                          float* __restrict__ e, float* __restrict__ out) {
  __shared__ double red[4];
  double a = 0.0;
  for (int i = threadIdx.x; i < B; i += 256) {
    const float v = expf(-2.f * scale * (beta_rec * L[i] + beta_neg * KL[i]));
    e[i] = v;
    a += (double)v;
  }
  a = block_sum<256>(a, red);
  if (threadIdx.x == 0) out[0] = (float)(a / (double)B);
}
__global__ void __launch_bounds__(256) expelbo_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ e,
                                                          float scale, float beta_rec, float beta_neg, int B,
                                                          float* __restrict__ dL, float* __restrict__ dKL) {
  const float g = gout[0] / (float)B;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < B; i += gridDim.x * 256) {
    const float t = g * e[i] * (-2.f * scale);
    dL[i] = t * beta_rec;
    dKL[i] = t * beta_neg;
  }
}

// ---------------------------------------------------------------- Philox4x32-10 standard normals
// counter = (offset + i/4), key = seed ; Box-Muller on the four 32-bit outputs -> 4 normals.
__device__ __forceinline__ void philox_round(unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3, unsigned k0,
                                             unsigned k1) {
  const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
  const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
  const unsigned n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
  c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}
__global__ void __launch_bounds__(256) randn_kernel(float* __restrict__ out, size_t n, unsigned long long seed,
                                                    unsigned long long offset,
                                                    const unsigned long long* __restrict__ offset_dev) {
  if (offset_dev != nullptr) offset += *offset_dev;  // stream position kept on the device (HIP-graph replay)
  const size_t n4 = (n + 3) >> 2;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const unsigned long long ctr = offset + i;
    unsigned c0 = (unsigned)ctr, c1 = (unsigned)(ctr >> 32), c2 = 0x5EED5EEDu, c3 = 0u;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int rd = 0; rd < 10; ++rd) {
      philox_round(c0, c1, c2, c3, k0, k1);
      k0 += 0x9E3779B9u;
      k1 += 0xBB67AE85u;
    }
    const float u0 = ((float)c0 + 0.5f) * 2.3283064365386963e-10f;  // (0,1)
    const float u1 = ((float)c1 + 0.5f) * 2.3283064365386963e-10f;
    const float u2 = ((float)c2 + 0.5f) * 2.3283064365386963e-10f;
    const float u3 = ((float)c3 + 0.5f) * 2.3283064365386963e-10f;
    const float r0 = sqrtf(-2.f * logf(u0)), r1 = sqrtf(-2.f * logf(u2));
    float s0, co0, s1, co1;
    sincosf(6.283185307179586f * u1, &s0, &co0);
    sincosf(6.283185307179586f * u3, &s1, &co1);
    const float v[4] = {r0 * co0, r0 * s0, r1 * co1, r1 * s1};
    const size_t base = i << 2;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (base + k < n) out[base + k] = v[k];
  }
}

// ================================================================= C ABI
static inline int g1d(size_t n) {
  long long nb = (long long)((n + 255) / 256);
  if (nb > 4096) nb = 4096;
  if (nb < 1) nb = 1;
  return (int)nb;
}

extern "C" int sivae_reparam_fwd(const float* mu, const float* logvar, int ld, const float* eps, float* z, int B,
                                 int Z, hipStream_t stream) {
  if (!mu || !logvar || !eps || !z) return SIVAE_ERR_NULL;
  if (B <= 0 || Z <= 0 || ld < Z) return SIVAE_ERR_SHAPE;
  hipLaunchKernelGGL(reparam_fwd_kernel, dim3(g1d((size_t)B * Z)), dim3(256), 0, stream, mu, logvar, ld, eps, z, B, Z);
  return sivae_launch_status();
}
extern "C" int sivae_reparam_bwd(const float* dz, const float* logvar, int ld, const float* eps, float* dmu,
                                 float* dlogvar, int ldg, int B, int Z, hipStream_t stream) {
  if (!dz || !logvar || !eps || !dmu || !dlogvar) return SIVAE_ERR_NULL;
  if (B <= 0 || Z <= 0 || ld < Z || ldg < Z) return SIVAE_ERR_SHAPE;
  hipLaunchKernelGGL(reparam_bwd_kernel, dim3(g1d((size_t)B * Z)), dim3(256), 0, stream, dz, logvar, ld, eps, dmu,
                     dlogvar, ldg, B, Z);
  return sivae_launch_status();
}
extern "C" int sivae_kl_fwd(const float* logvar, const float* mu, int ld, float mu_o, float logvar_o, float* out,
                            int B, int Z, hipStream_t stream) {
  if (!logvar || !mu || !out) return SIVAE_ERR_NULL;
  if (B <= 0 || Z <= 0 || ld < Z) return SIVAE_ERR_SHAPE;
  hipLaunchKernelGGL(kl_fwd_kernel, dim3(B), dim3(64), 0, stream, logvar, mu, ld, mu_o, logvar_o, out, Z);
  return sivae_launch_status();
}
// g_per_sample != 0: g is [B]; else g is a scalar broadcast to every sample. Effective grad = g * g_scale.
extern "C" int sivae_kl_bwd(const float* g, int g_per_sample, float g_scale, const float* logvar, const float* mu,
                            int ld, float mu_o, float logvar_o, float* dlogvar, float* dmu, int ldg, int B, int Z,
                            hipStream_t stream) {
  if (!g || !logvar || !mu || !dlogvar || !dmu) return SIVAE_ERR_NULL;
  if (B <= 0 || Z <= 0 || ld < Z || ldg < Z) return SIVAE_ERR_SHAPE;
  hipLaunchKernelGGL(kl_bwd_kernel, dim3(g1d((size_t)B * Z)), dim3(256), 0, stream, g, g_per_sample ? 1 : 0, g_scale,
                     logvar, mu, ld, mu_o, logvar_o, dlogvar, dmu, ldg, B, Z);
  return sivae_launch_status();
}

extern "C" int sivae_kl_fwd_t(const float* logvar, const float* mu, int ld, const float* mu_o, int mu_o_rs, int mu_o_cs,
                              const float* logvar_o, int lv_o_rs, int lv_o_cs, float* out, int B, int Z,
                              hipStream_t stream) {
  if (!logvar || !mu || !out || !mu_o || !logvar_o) return SIVAE_ERR_NULL;
  if (B <= 0 || Z <= 0 || ld < Z) return SIVAE_ERR_SHAPE;
  if (mu_o_rs < 0 || mu_o_cs < 0 || lv_o_rs < 0 || lv_o_cs < 0) return SIVAE_ERR_MODE;
  hipLaunchKernelGGL(kl_fwd_t_kernel, dim3(B), dim3(64), 0, stream, logvar, mu, ld, mu_o, mu_o_rs, mu_o_cs, logvar_o,
                     lv_o_rs, lv_o_cs, out, Z);
  return sivae_launch_status();
}
extern "C" int sivae_kl_bwd_t(const float* g, int g_per_sample, float g_scale, const float* logvar, const float* mu,
                              int ld, const float* mu_o, int mu_o_rs, int mu_o_cs, const float* logvar_o, int lv_o_rs,
                              int lv_o_cs, float* dlogvar, float* dmu, int ldg, int B, int Z, hipStream_t stream) {
  if (!g || !logvar || !mu || !dlogvar || !dmu || !mu_o || !logvar_o) return SIVAE_ERR_NULL;
  if (B <= 0 || Z <= 0 || ld < Z || ldg < Z) return SIVAE_ERR_SHAPE;
  if (mu_o_rs < 0 || mu_o_cs < 0 || lv_o_rs < 0 || lv_o_cs < 0) return SIVAE_ERR_MODE;
  hipLaunchKernelGGL(kl_bwd_t_kernel, dim3(g1d((size_t)B * Z)), dim3(256), 0, stream, g, g_per_sample ? 1 : 0, g_scale,
                     logvar, mu, ld, mu_o, mu_o_rs, mu_o_cs, logvar_o, lv_o_rs, lv_o_cs, dlogvar, dmu, ldg, B, Z);
  return sivae_launch_status();
}

static inline void recon_plan(int D, int* S, int* len) {
  int s = (D + 16383) / 16384;
  if (s < 1) s = 1;
  int l = (D + s - 1) / s;
  l = (l + 3) & ~3;
  *S = (D + l - 1) / l;
  *len = l;
}
extern "C" size_t sivae_recon_workspace_bytes(int B, int D) {
  if (B <= 0 || D <= 0) return 0;
  int S, len;
  recon_plan(D, &S, &len);
  return (size_t)B * S * sizeof(double);
}
// per-sample sums out[b] = sum_i term(x[b,i], recon[b,i]);  loss_type 0 mse, 1 l1, 2 bce
extern "C" int sivae_recon_rowsum_fwd(const float* x, const float* recon, int loss_type, float* out, int B, int D,
                                      void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!x || !recon || !out) return SIVAE_ERR_NULL;
  if (B <= 0 || D <= 0) return SIVAE_ERR_SHAPE;
  if (loss_type < 0 || loss_type > 2) return SIVAE_ERR_MODE;
  if (!workspace || workspace_bytes < sivae_recon_workspace_bytes(B, D)) return SIVAE_ERR_WORKSPACE;
  if (B > 65535) return SIVAE_ERR_RANGE;
  int S, len;
  recon_plan(D, &S, &len);
  double* part = (double*)workspace;
#define LAUNCH(T) \
  hipLaunchKernelGGL((recon_rowsum_partial_kernel<T>), dim3(S, B), dim3(256), 0, stream, x, recon, part, D, S, len)
  if (loss_type == 0) LAUNCH(0); else if (loss_type == 1) LAUNCH(1); else LAUNCH(2);
#undef LAUNCH
  hipLaunchKernelGGL(rowsum_finalize_kernel, dim3(cdiv(B, 64)), dim3(64), 0, stream, (const double*)part, S, B, out);
  return sivae_launch_status();
}
// g_mode 0: g[B] per-sample, 1: scalar g[0], 2: per-element g[B*D].  d_recon / d_x may be null.
extern "C" int sivae_recon_bwd(const float* x, const float* recon, int loss_type, const float* g, int g_mode,
                               float g_scale, float* d_recon, float* d_x, int B, int D, hipStream_t stream) {
  if (!x || !recon || !g) return SIVAE_ERR_NULL;
  if (!d_recon && !d_x) return SIVAE_ERR_NULL;
  if (B <= 0 || D <= 0) return SIVAE_ERR_SHAPE;
  if (loss_type < 0 || loss_type > 2 || g_mode < 0 || g_mode > 2) return SIVAE_ERR_MODE;
  const size_t n = (size_t)B * D;
  long long nb = (long long)((n + 255) / 256);
  if (nb > 16384) nb = 16384;
#define LAUNCH(T) \
  hipLaunchKernelGGL((recon_bwd_kernel<T>), dim3((int)nb), dim3(256), 0, stream, x, recon, g, g_mode, g_scale, \
                     d_recon, d_x, D, n)
  if (loss_type == 0) LAUNCH(0); else if (loss_type == 1) LAUNCH(1); else LAUNCH(2);
#undef LAUNCH
  return sivae_launch_status();
}
extern "C" int sivae_recon_elem_fwd(const float* x, const float* recon, int loss_type, float* out, size_t numel,
                                    hipStream_t stream) {
  if (!x || !recon || !out) return SIVAE_ERR_NULL;
  if (numel == 0) return SIVAE_ERR_SHAPE;
  if (loss_type < 0 || loss_type > 2) return SIVAE_ERR_MODE;
  long long nb = (long long)((numel + 255) / 256);
  if (nb > 16384) nb = 16384;
#define LAUNCH(T) \
  hipLaunchKernelGGL((recon_elem_fwd_kernel<T>), dim3((int)nb), dim3(256), 0, stream, x, recon, out, numel)
  if (loss_type == 0) LAUNCH(0); else if (loss_type == 1) LAUNCH(1); else LAUNCH(2);
#undef LAUNCH
  return sivae_launch_status();
}
extern "C" int sivae_vec_sum(const float* v, int n, float scale, float* out, hipStream_t stream) {
  if (!v || !out) return SIVAE_ERR_NULL;
  if (n <= 0) return SIVAE_ERR_SHAPE;
  hipLaunchKernelGGL(vec_sum_kernel, dim3(1), dim3(256), 0, stream, v, n, scale, out);
  return sivae_launch_status();
}
extern "C" int sivae_expelbo_fwd(const float* L, const float* KL, float scale, float beta_rec, float beta_neg, int B,
                                 float* e, float* out, hipStream_t stream) {
  if (!L || !KL || !e || !out) return SIVAE_ERR_NULL;
  if (B <= 0) return SIVAE_ERR_SHAPE;
  hipLaunchKernelGGL(expelbo_fwd_kernel, dim3(1), dim3(256), 0, stream, L, KL, scale, beta_rec, beta_neg, B, e, out);
  return sivae_launch_status();
}
extern "C" int sivae_expelbo_bwd(const float* gout, const float* e, float scale, float beta_rec, float beta_neg,
                                 int B, float* dL, float* dKL, hipStream_t stream) {
  if (!gout || !e || !dL || !dKL) return SIVAE_ERR_NULL;
  if (B <= 0) return SIVAE_ERR_SHAPE;
  hipLaunchKernelGGL(expelbo_bwd_kernel, dim3(g1d(B)), dim3(256), 0, stream, gout, e, scale, beta_rec, beta_neg, B, dL,
                     dKL);
  return sivae_launch_status();
}
// ---- loss assembly: lossE = scale * (beta_rec * loss_rec + beta_kl * kl_real) + 0.25 * (expelbo_rec + expelbo_fake)
// (train_soft_intro_vae.py:583-586) and lossD (:618-620) are weighted sums of up to six device scalars.  As torch arithmetic
// on 0-dim tensors each is 7-9 launches forward and as many backward (25 of an iteration's launches); here one each.
struct LinCombArgs {
  const float* p[6];
  float w[6];
  int n;
};
__global__ void lincomb_kernel(LinCombArgs a, float* __restrict__ out) {
  if (threadIdx.x != 0) return;
  float s = 0.f;
  for (int i = 0; i < a.n; ++i) s += a.w[i] * a.p[i][0];
  out[0] = s;
}
__global__ void lincomb_bwd_kernel(const float* __restrict__ g, LinCombArgs a, float* __restrict__ out) {
  if ((int)threadIdx.x < a.n) out[threadIdx.x] = g[0] * a.w[threadIdx.x];
}
// out[0] = sum_i w[i] * p[i][0], i < n <= 6 (terms added in index order); unused pointers may be NULL
extern "C" int sivae_lincomb(const float* p0, const float* p1, const float* p2, const float* p3, const float* p4,
                             const float* p5, float w0, float w1, float w2, float w3, float w4, float w5, int n,
                             float* out, hipStream_t stream) {
  if (!out) return SIVAE_ERR_NULL;
  if (n <= 0 || n > 6) return SIVAE_ERR_SHAPE;
  LinCombArgs a = {{p0, p1, p2, p3, p4, p5}, {w0, w1, w2, w3, w4, w5}, n};
  for (int i = 0; i < n; ++i)
    if (!a.p[i]) return SIVAE_ERR_NULL;
  hipLaunchKernelGGL(lincomb_kernel, dim3(1), dim3(64), 0, stream, a, out);
  return sivae_launch_status();
}
// its gradient: out[i] = g[0] * w[i], i < n
extern "C" int sivae_lincomb_bwd(const float* g, float w0, float w1, float w2, float w3, float w4, float w5, int n,
                                 float* out, hipStream_t stream) {
  if (!g || !out) return SIVAE_ERR_NULL;
  if (n <= 0 || n > 6) return SIVAE_ERR_SHAPE;
  LinCombArgs a = {{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, {w0, w1, w2, w3, w4, w5}, n};
  hipLaunchKernelGGL(lincomb_bwd_kernel, dim3(1), dim3(64), 0, stream, g, a, out);
  return sivae_launch_status();
}
extern "C" int sivae_randn(float* out, size_t n, unsigned long long seed, unsigned long long offset,
                           hipStream_t stream) {
  if (!out) return SIVAE_ERR_NULL;
  if (n == 0) return SIVAE_ERR_SHAPE;
  hipLaunchKernelGGL(randn_kernel, dim3(g1d((n + 3) >> 2)), dim3(256), 0, stream, out, n, seed, offset,
                     (const unsigned long long*)nullptr);
  return sivae_launch_status();
}

// Same stream with its position in DEVICE memory: draws from counter *offset_dev, then advances it by the number
// of Philox counters used — nothing a captured HIP graph bakes in changes between replays.
__global__ void counter_add_kernel(unsigned long long* c, unsigned long long inc) { *c += inc; }

extern "C" int sivae_randn_dev(float* out, size_t n, unsigned long long seed, unsigned long long* offset_dev,
                               hipStream_t stream) {
  if (!out || !offset_dev) return SIVAE_ERR_NULL;
  if (n == 0) return SIVAE_ERR_SHAPE;
  hipLaunchKernelGGL(randn_kernel, dim3(g1d((n + 3) >> 2)), dim3(256), 0, stream, out, n, seed, 0ull,
                     (const unsigned long long*)offset_dev);
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, stream, offset_dev, (unsigned long long)((n + 3) >> 2));
  return sivae_launch_status();
}
