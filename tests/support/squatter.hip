// Test support, NOT part of the product ABI (libsivae_hip.so exports nothing of this): a kernel that does nothing but
// HOLD resources — `blocks` workgroups of `threads` threads with `lds_bytes` of dynamic LDS each stay resident for
// `ticks` periods of the 100 MHz wall clock (or until *stop becomes non-zero, when stop is given), the footprint of a
// collective's kernel on a side stream.  fat != 0: every wave also holds ~200 VGPRs, so that no wave of a 256-register
// kernel fits next to it — the persistent BatchNorm backward is then NOT fully resident until the squatter leaves: it must
// wait at its grid barrier (not trap, not give up within the spin limit) and finish with the bits it produces alone.
// Built by tests/support/build.sh into tests/support/libsivae_testsupport.so (tests/kernel_checks.py::check_bn_fused_squatter).
#include <hip/hip_runtime.h>

template <bool FAT>
__global__ void squatter_kernel(long long ticks, const int* stop) {
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

// 0 = launched; -2 = argument out of range; > 0 = hipError_t of the launch
extern "C" int testsupport_squatter(int blocks, int threads, int lds_bytes, int fat, long long ticks, const int* stop,
                                    hipStream_t stream) {
  if (blocks <= 0 || threads <= 0 || threads > 1024 || lds_bytes < 4 || lds_bytes > 65536 || ticks <= 0) return -2;
  if (fat)
    hipLaunchKernelGGL(squatter_kernel<true>, dim3((unsigned)blocks), dim3((unsigned)threads), (size_t)lds_bytes, stream,
                       ticks, stop);
  else
    hipLaunchKernelGGL(squatter_kernel<false>, dim3((unsigned)blocks), dim3((unsigned)threads), (size_t)lds_bytes, stream,
                       ticks, stop);
  return (int)hipGetLastError();
}
