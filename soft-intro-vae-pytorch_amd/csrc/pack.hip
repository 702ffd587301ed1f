// Weight re-layout for the implicit-GEMM kernels.
//   mode 0 (forward):  wp[tap][ci][co]  = w[co][ci][tap]                 dims [T][ci_pad(Ci)][co_pad(Co)]
//   mode 1 (dgrad):    wp[tap][co][ci]  = w[co][ci][T-1-tap]  (180° flip) dims [T][ci_pad(Co)][co_pad(Ci)]
// Padding entries are zero, so the GEMM kernels need no bounds checks on the weight operand.
// Runs once per optimizer step per layer (weights are reused by 5-8 passes per iteration).
#include "common.h"

extern "C" int sivae_conv_ci_pad(int ks, int ci);
extern "C" int sivae_conv_co_pad(int co);

__global__ void __launch_bounds__(256) pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wp,
                                                          int Co, int Ci, int taps, int mode, int kdim,
                                                          int kpad, int npad, size_t total) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i < total; i += stride) {
    const int n = (int)(i % npad);
    const size_t t = i / npad;
    const int k = (int)(t % kpad);
    const int tap = (int)(t / kpad);
    float v = 0.f;
    if (mode == 0) {
      // k = ci, n = co
      if (k < Ci && n < Co) v = w[((size_t)n * Ci + k) * taps + tap];
    } else {
      // k = co, n = ci, flipped tap
      if (k < Co && n < Ci) v = w[((size_t)k * Ci + n) * taps + (taps - 1 - tap)];
    }
    (void)kdim;
    wp[i] = v;
  }
}

extern "C" size_t sivae_pack_conv_weight_bytes(int Co, int Ci, int ks, int mode) {
  if (ks != 1 && ks != 3 && ks != 5) return 0;
  const int kdim = mode == 0 ? Ci : Co, ndim = mode == 0 ? Co : Ci;
  return (size_t)ks * ks * sivae_conv_ci_pad(ks, kdim) * sivae_conv_co_pad(ndim) * sizeof(float);
}

extern "C" int sivae_pack_conv_weight(const float* w, float* wp, int Co, int Ci, int ks, int mode,
                                      hipStream_t stream) {
  if (!w || !wp) return SIVAE_ERR_NULL;
  if (Co <= 0 || Ci <= 0) return SIVAE_ERR_SHAPE;
  if (ks != 1 && ks != 3 && ks != 5) return SIVAE_ERR_KSIZE;
  if (mode != 0 && mode != 1) return SIVAE_ERR_MODE;
  const int kdim = mode == 0 ? Ci : Co, ndim = mode == 0 ? Co : Ci;
  const int kpad = sivae_conv_ci_pad(ks, kdim), npad = sivae_conv_co_pad(ndim);
  const size_t total = (size_t)ks * ks * kpad * npad;
  int nb = cdiv((long long)total, 256);
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(nb), dim3(256), 0, stream, w, wp, Co, Ci, ks * ks, mode, kdim,
                     kpad, npad, total);
  return sivae_launch_status();
}
