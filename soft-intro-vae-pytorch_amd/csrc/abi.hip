// Version / architecture probes of libsivae_hip.
#include "common.h"
extern "C" int sivae_abi_version() { return SIVAE_ABI_VERSION; }
extern "C" const char* sivae_arch() { return "gfx950"; }
// number of visible HIP devices (0 on a box without a GPU); never throws
extern "C" int sivae_device_count() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// ---- test support (tests/kernel_checks.py::check_bn_fused_squatter): a kernel that does nothing but HOLD resources —
// `blocks` workgroups of `threads` threads with `lds_bytes` of dynamic LDS each stay resident for `ticks` periods of the
// 100 MHz wall clock (or until *stop becomes non-zero, when stop is given), the footprint of a collective's kernel on a
// side stream.  The persistent BatchNorm backward must make progress (more slowly) while such a kernel holds part of
// the chip, and must produce the bits it produces alone.
__global__ void sivae_squatter_kernel(long long ticks, const int* stop) {
  extern __shared__ int squat_lds[];
  if (threadIdx.x == 0) squat_lds[0] = 1;
  const long long t0 = (long long)wall_clock64();
  while ((long long)wall_clock64() - t0 < ticks) {
    if (stop != nullptr && __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
    __builtin_amdgcn_s_sleep(32);
  }
}
extern "C" int sivae_debug_squatter(int blocks, int threads, int lds_bytes, long long ticks, const int* stop,
                                    hipStream_t stream) {
  if (blocks <= 0 || threads <= 0 || threads > 1024 || lds_bytes < 4 || lds_bytes > 65536 || ticks <= 0)
    return SIVAE_ERR_SHAPE;
  hipLaunchKernelGGL(sivae_squatter_kernel, dim3((unsigned)blocks), dim3((unsigned)threads), (size_t)lds_bytes, stream,
                     ticks, stop);
  return sivae_launch_status();
}
