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
// side stream.  fat != 0: every wave also holds ~200 VGPRs, so that no wave of a 256-register kernel fits next to it —
// the persistent BatchNorm backward is then NOT fully resident until the squatter leaves: it must wait at its grid
// barrier (not trap, not give up within the spin limit) and finish with the bits it produces alone.
template <bool FAT>
__global__ void sivae_squatter_kernel(long long ticks, const int* stop) {
  extern __shared__ int squat_lds[];
  if (threadIdx.x == 0) squat_lds[0] = 1;
  if (FAT) asm volatile("v_mov_b32 v200, 0" ::: "v200");
  const long long t0 = (long long)wall_clock64();
  while ((long long)wall_clock64() - t0 < ticks) {
    if (stop != nullptr && __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
    __builtin_amdgcn_s_sleep(32);
  }
  if (FAT) asm volatile("v_mov_b32 v200, 0" ::: "v200");
}
extern "C" int sivae_debug_squatter(int blocks, int threads, int lds_bytes, int fat, long long ticks, const int* stop,
                                    hipStream_t stream) {
  if (blocks <= 0 || threads <= 0 || threads > 1024 || lds_bytes < 4 || lds_bytes > 65536 || ticks <= 0)
    return SIVAE_ERR_SHAPE;
  if (fat)
    hipLaunchKernelGGL(sivae_squatter_kernel<true>, dim3((unsigned)blocks), dim3((unsigned)threads), (size_t)lds_bytes,
                       stream, ticks, stop);
  else
    hipLaunchKernelGGL(sivae_squatter_kernel<false>, dim3((unsigned)blocks), dim3((unsigned)threads), (size_t)lds_bytes,
                       stream, ticks, stop);
  return sivae_launch_status();
}
