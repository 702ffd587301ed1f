// Pieces shared by the one-pass BatchNorm-backward kernels (bn_fused.hip: fp32 NCHW; bf16_bn_fused.hip: blocked bf16):
// the barrier / counter state layout, the fence-free XCD-hierarchical grid barrier, and the group plan.
#pragma once
#include "common.h"
#include <stdlib.h>

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

// Word of the state that carries the "grid barrier timed out" flag (its own 128-byte line in the unused tail of the
// first barrier area).  Non-zero: some launch gave up waiting for a block that never arrived (the grid was not fully
// resident); every block that sees it leaves the kernel, the results of that launch are garbage and the barrier counters
// are inconsistent -> the host must zero the whole state before the next launch (sivae_hip.ops.bn_fused_check raises).
constexpr int BF_POISON_WORD = BF_BAR_UINTS - 32;
constexpr unsigned BF_SPIN_LIMIT_DEFAULT = 1u << 24;  // polls of ~1 us: tens of seconds, then the launch is abandoned

// grid barrier of one half-grid, called by thread 0 of every block.  `bar`: arrival counters [xcd] at bar + 32*xcd, the
// top counter at bar + 256, generation flags at bar + 32*(9 + xcd).  Every generation flag only ever advances BY ONE per
// barrier and a block compares its own XCD's flag with the value it read at kernel start plus the barriers it has passed —
// the flags need not agree with each other, so launches that use different numbers of XCD groups (nx) can share the state,
// and the flags need no re-initialisation between launches.
// The arrival counters are zero at the start of a launch and run ON through its barriers (round 6): the last arriver of
// barrier n (0-based, the same n in every block of the half-grid) is the one that finds (n + 1) * bpx - 1 — no reset store
// and no wait for it between the two levels (the round-5 form reset both counters inside every barrier: two more memory
// round trips of ~1.5 us on the critical path of each group).  bf_grid_reset puts them back to zero once, behind the
// launch's last barrier; a launch abandoned by the spin limit leaves them inconsistent, as before (the host zeroes the state).
// No fences: everything that crosses blocks (the partial sums, the counters, the flags) is written with agent-scope
// (write-through, `sc1`) stores / atomics and read with agent-scope loads, ordered by explicit vmcnt(0) waits.  A release
// fence here would write back the whole L2 of the XCD — the dx stores of the previous group, megabytes — once per block
// and barrier (first form of this kernel: 46 us per group instead of the ~13 us its bytes need).
// Returns false when the wait was abandoned (spin limit reached here or in another block: the poison word is set) — the
// caller's block must leave the kernel; nothing traps and nothing hangs.
// The two halves separately (a caller may issue independent memory traffic between its arrival and its wait — the
// persistent BatchNorm backward requests the next group's x there, bn_fused.hip):
__device__ __forceinline__ void bf_grid_arrive(unsigned* bar, int xcd, int nx, unsigned bpx, unsigned n) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this block's partial sums (sc1 stores) have reached memory
  unsigned* cnt = bar + xcd * 32;
  const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (old == (n + 1u) * bpx - 1u) {
    unsigned* top = bar + 8 * 32;
    const unsigned o2 = __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (o2 == (n + 1u) * (unsigned)nx - 1u) {
      for (int i = 0; i < nx; ++i)
        __hip_atomic_fetch_add(bar + (9 + i) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
// Behind the LAST barrier of a launch (every block of the half-grid has arrived at it: nobody touches the counters again),
// by one thread of the half-grid: the counters go back to zero for the next launch (ordered by the kernel boundary).
__device__ __forceinline__ void bf_grid_reset(unsigned* bar, int nx) {
  for (int i = 0; i < nx; ++i) __hip_atomic_store(bar + i * 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(bar + 8 * 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool bf_grid_wait(const unsigned* bar, unsigned* poison, int xcd, unsigned target,
                                             unsigned spin_limit) {
  const unsigned* gen = bar + (9 + xcd) * 32;
  unsigned spins = 0;
  while (bf_load_u32(gen) != target) {
    __builtin_amdgcn_s_sleep(1);
    ++spins;
    if ((spins & 255u) == 0u && bf_load_u32(poison) != 0u) return false;
    if (spins > spin_limit) {  // a block of this grid is not resident (or never will be): flag it and give up
      __hip_atomic_store(poison, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
  }
  return true;
}
__device__ __forceinline__ bool bf_grid_barrier(unsigned* bar, unsigned* poison, int xcd, int nx, unsigned bpx,
                                                unsigned target, unsigned spin_limit, unsigned n) {
  bf_grid_arrive(bar, xcd, nx, bpx, n);
  return bf_grid_wait(bar, poison, xcd, target, spin_limit);
}

// may a persistent (grid-barrier) launch assume that all CUs sivae_num_cus() reports are available to it?  Not under a CU
// mask (HSA_CU_MASK / ROC_GLOBAL_CU_MASK: the device still reports every CU); SIVAE_BN_FUSED_PERSISTENT=0 forces the
// answer (the callers then keep the three-launch form for plane sets that need the barrier).
static inline bool bf_persistent_allowed() {
  static int ok = -1;
  if (ok < 0) {
    const char* f = getenv("SIVAE_BN_FUSED_PERSISTENT");
    const char* m1 = getenv("HSA_CU_MASK");
    const char* m2 = getenv("ROC_GLOBAL_CU_MASK");
    if (f && f[0] == '0') ok = 0;
    else if ((m1 && m1[0]) || (m2 && m2[0])) ok = 0;
    else ok = 1;
  }
  return ok == 1;
}
// (read per launch, not cached: tests/kernel_checks.py::check_bn_fused_timeout shortens it for one call)
static inline unsigned bf_spin_limit() {
  const char* e = getenv("SIVAE_BN_FUSED_SPIN_LIMIT");
  long long lim = e ? atoll(e) : (long long)BF_SPIN_LIMIT_DEFAULT;
  if (lim < 16) lim = 16;
  if (lim > 0x7fffffffLL) lim = 0x7fffffffLL;
  return (unsigned)lim;
}


}  // namespace
