// Pieces shared by the one-pass BatchNorm-backward kernels (bn_fused.hip: fp32 NCHW; bf16_bn_fused.hip: blocked bf16):
// the barrier / counter state layout, the fence-free XCD-hierarchical grid barrier, and the group plan.
#pragma once
#include "common.h"

namespace {

constexpr int BF_BAR_UINTS = 1024;   // barrier state per half-grid: 8 arrival counters, 1 top counter, 8 generation flags,
                                     // one 128-byte line each (17 x 32 uints used)
constexpr unsigned BF_OOB = 0xFFFFFF00u;  // byte offset no window reaches (windows are < 0xfffffe00 bytes)
constexpr int BF_CH_COUNTERS = 8192;  // per-channel segment-arrival counters behind the two barrier areas


__device__ __forceinline__ unsigned bf_load_u32(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double bf_load_f64(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// grid barrier of one half-grid, called by thread 0 of every block.  `bar`: arrival counters [xcd] at bar + 32*xcd, the
// top counter at bar + 256, generation flags at bar + 32*(9 + xcd).  Counters are reset by the last arriver (nobody
// arrives again before the generation flips), the generation only ever advances: the state needs no re-initialisation
// between launches.
// No fences: everything that crosses blocks (the partial sums, the counters, the flags) is written with agent-scope
// (write-through, `sc1`) stores / atomics and read with agent-scope loads, ordered by explicit vmcnt(0) waits.  A release
// fence here would write back the whole L2 of the XCD — the dx stores of the previous group, megabytes — once per block
// and barrier (first form of this kernel: 46 us per group instead of the ~13 us its bytes need).
__device__ __forceinline__ void bf_grid_barrier(unsigned* bar, int xcd, int nx, unsigned bpx, unsigned target) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this block's partial sums (sc1 stores) have reached memory
  unsigned* cnt = bar + xcd * 32;
  const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (old == bpx - 1u) {
    __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned* top = bar + 8 * 32;
    const unsigned o2 = __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (o2 == (unsigned)nx - 1u) {
      __hip_atomic_store(top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      for (int i = 0; i < nx; ++i) __hip_atomic_store(bar + (9 + i) * 32, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  const unsigned* gen = bar + (9 + xcd) * 32;
  unsigned spins = 0;
  while (bf_load_u32(gen) != target) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > (1u << 27)) __builtin_trap();  // (a block of this grid is not resident: fail loudly instead of hanging)
  }
}


}  // namespace
