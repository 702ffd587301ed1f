// BatchNorm2d backward (+ LeakyReLU sign, + residual-branch gradient) as ONE persistent launch that touches every tensor
// exactly once: read dy and x, write dx (round 4).
//
// The two-kernel form in bn.hip (bn_bwd_partial_kernel -> bn_bwd_finalize_kernel -> bn_bwd_dx_kernel) streams dy and x
// TWICE — once for the per-channel sums {sum dz, sum dz*xhat}, once more for dx — because the sums cover the whole batch:
// 5 tensor passes at the copy roofline, 68 ms of the 415 ms headline iteration and 39 % of its HBM bytes.  Here the chip's
// register files hold the data between the two phases instead: 2 blocks per CU x 256 threads x up to 192 VGPRs of
// payload = 100 MB on 256 CUs.  The (segment, channel) planes are walked in GROUPS that fit that capacity; per group
//   phase 1   every block loads its slab of dy / x (and the sign source) into registers, turns dy into dz in place,
//             reduces {sum dz, sum dz*xhat} in fp64 and publishes one partial per slab;
//   barrier   grid-wide (XCD-hierarchical arrival counters + generation flags, sense-free: the state is left consistent);
//   phase 2   every block folds the partials of ITS channel in a fixed order (all blocks of a channel compute the same
//             coefficients — deterministic), forms dx from the registers and stores it (+ dz, or its 2x2 block sums).
// The grid is two independent half-grids (one block per CU each) that walk alternate groups with their own barrier state:
// while one half waits at its barrier the other half's loads / stores keep the HBM pipe busy.
// Round 5: between its arrival at a group's barrier and the wait, a block requests its slab of the NEXT group's x into LDS
// (LDS-direct loads, up to 6 of the 8 quads per thread = 48 KB per block): the memory pipe is otherwise idle from the
// moment the last block's loads have landed until the partial sums are folded (~1/3 of a group's 30 us on the 256x256
// layers, whose plane sets fill the whole grid so that the half-grid overlap above does not apply).  -4 % per call on
// those layers, -8..-11 % on the others, +1.4 % on the headline iteration (SIVAE_BN_FUSED_PREFETCH=0 switches it off).
// The request buffer is sized by what the chip's co-tenants allow, see PFQ below.
// dgamma / dbeta sum over the segments of a channel: the last (segment, channel) leader to finish adds the segments'
// sums in segment order (per-channel arrival counter), so the result does not depend on which one is last.
//
// Reference op: the backward of nn.BatchNorm2d + nn.LeakyReLU(0.2) (+ torch.add) in ResidualBlock / the encoder stem,
// soft_intro_vae/train_soft_intro_vae.py:57-63,71-74,90-91.
#include "bn_fused_common.h"
#include <stdlib.h>

namespace {

struct BnFusedArgs {
  const float* dy;
  const float* y;
  const float* x;
  const unsigned char* mask;
  const float* mean;
  const float* invstd;
  const float* gamma;
  const float* beta;
  float* dx;
  float* dz;
  float* dgamma;
  float* dbeta;
  double* part;   // [VC][spc][2]
  double* sums;   // [VC][2]
  unsigned* bar;  // 2 * BF_BAR_UINTS + BF_CH_COUNTERS uints, zero-initialised once by the caller
  double count;
  float slope;
  int C, H, W, Bs, nseg;
  int l2_qpp, l2_qw;  // log2(quads per plane), log2(quad columns per row); a quad = 2 rows x 4 columns
  int spc, cpg, ngroups, nx;
  int nsub;    // 2: two independent half-grids walking alternate groups; 1: one grid (plane sets too big for a half)
  int local;   // 1: every (segment, channel) plane set fits ONE block (spc == 1): grid = VC ordinary blocks, no barrier
  int dzmode;  // 0: none, 1: dz at full resolution, 2: 2x2 block sums [.][H/2][W/2]
  int pf;      // 1: the NEXT group's x is requested into LDS (LDS-direct loads) between the barrier's arrival and its wait
  unsigned spin_limit;  // polls of the barrier wait before the launch is abandoned (poison word set)
#ifdef BF_TIMING
  unsigned long long* ts;  // [block][group][8] s_memrealtime stamps of thread 0 (tools/bn_fused_timing.py; not a product build)
#endif
};
#ifdef BF_TIMING
#define BF_STAMP(K) \
  if (t == 0 && a.ts) a.ts[((size_t)blockIdx.x * a.ngroups + grp) * 8 + (K)] = __builtin_amdgcn_s_memrealtime();
#else
#define BF_STAMP(K)
#endif

__device__ __forceinline__ void bf_sign_nibble(float4& g, unsigned nib, float slope) {
  g.x = (nib & 1u) ? g.x : g.x * slope;
  g.y = (nib & 2u) ? g.y : g.y * slope;
  g.z = (nib & 4u) ? g.z : g.z * slope;
  g.w = (nib & 8u) ? g.w : g.w * slope;
}
__device__ __forceinline__ void bf_sign_val(float4& g, const float4 v, float slope) {
  g.x = v.x > 0.f ? g.x : g.x * slope;
  g.y = v.y > 0.f ? g.y : g.y * slope;
  g.z = v.z > 0.f ? g.z : g.z * slope;
  g.w = v.w > 0.f ? g.w : g.w * slope;
}
__device__ __forceinline__ void bf_store4(__amdgpu_buffer_rsrc_t r, const float4 v, unsigned voff, unsigned soff) {
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  f32x4 f;
  f[0] = v.x;
  f[1] = v.y;
  f[2] = v.z;
  f[3] = v.w;
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f), r, (int)voff, (int)soff, 0);
}
// Store-data lifetime (round 5; replaces round 4's `s_nop` pad).  Observed on gfx950 with the memory pipe saturated: a
// 16-byte buffer store whose data registers were rewritten by VALU instructions a few issue slots later stored the NEW
// values in the last quad of each 16-lane row.  Instead of padding with idle cycles, the data of every 16-byte store of
// this kernel lives in registers that NOTHING rewrites before the next group's loads are issued: dx is formed in place
// in the xhat registers of the payload, dz is stored from the payload itself, and BF_KEEP pins those values (an empty
// asm that "reads" them) at the point up to which their registers must not be recycled.  The next writers of those
// registers are VMEM loads, which the memory pipe executes in order behind the stores; the first VALU write follows an
// s_waitcnt vmcnt that — vmcnt being one in-order counter for loads and stores — covers the stores as well.
#define BF_KEEP(V) asm volatile("" ::"v"((V).x), "v"((V).y), "v"((V).z), "v"((V).w));

// ACT: 0 none, 1 sign from the saved output y, 2 sign recomputed from x (gamma, beta), 3 sign from the 1-bit mask.
// POOL: dy is the gradient of AvgPool2d(2)(output) at half resolution (read through the pool's adjoint).
// NQ: quads per thread and group (16 payload VGPRs each).
//
// Addressing: a block works on ONE (segment, channel) per group; its tensors are reached through buffer descriptors
// based at that channel's plane of the segment's first image, so a quad is a 32-bit byte offset (one VGPR; the second
// row of the quad is the same offset with the row pitch in the scalar offset) and a quad past the end of the plane set
// carries an out-of-range offset: its loads return 0 (dz = 0: no contribution to the sums) and its stores are skipped.
template <int ACT, bool POOL, int NQ>
__global__ void __launch_bounds__(256, 2) bn_bwd_fused_kernel(BnFusedArgs a) {
  __shared__ double red[8];
  // x of the NEXT group, requested while this block waits at the grid barrier (a.pf): [PFQ][2 rows][256 threads] float4 —
  // a wave's 64 lanes are 1 KB contiguous, the layout an LDS-direct load writes (M0 base + lane * 16)
  extern __shared__ __attribute__((aligned(16))) float4 bf_pfx[];
  const int t = threadIdx.x;
  const int wave64 = __builtin_amdgcn_readfirstlane(t >> 6) * 64;
  const bool pf = NQ <= 8 && a.pf != 0;  // (10 quads per thread: no code for the request buffer)
  // quads per thread that go through the request buffer: at most 6 = 48 KB per block.  Two blocks then fit a CU beside
  // 64 KB of somebody else's LDS, and no single co-tenant allocation can fragment the CU's 160 KB such that the second
  // block never fits (that needs a block > 160 / 3 KB): with all 8 quads (64 KB) a 32-KB co-tenant that left did exactly
  // that — the grid never became resident (tests/kernel_checks.py::check_bn_fused_squatter)
  constexpr int PFQ = NQ < 6 ? NQ : 6;
  const bool local = a.local != 0;
  const int nb_sub = local ? (int)gridDim.x : (int)gridDim.x / a.nsub;
  const int sub = (!local && (int)blockIdx.x >= nb_sub) ? 1 : 0;
  const int bid = (int)blockIdx.x - sub * nb_sub;
  unsigned* bar = a.bar + sub * BF_BAR_UINTS;
  unsigned* chcnt = a.bar + 2 * BF_BAR_UINTS;
  const int xcd = bid % a.nx;
  const unsigned bpx = (unsigned)(nb_sub / a.nx);
  __shared__ int bar_failed;
  unsigned target = 0, nbar = 0;  // (thread 0) generation to wait for; barriers of this launch passed so far
  if (t == 0 && !local) target = bf_load_u32(bar + (9 + xcd) * 32);
  const int C = a.C, W = a.W, HW = a.H * a.W;
  const int VC = a.nseg * C;
  const int nq = a.Bs << a.l2_qpp;
  const unsigned qpp_m = (1u << a.l2_qpp) - 1u, qw_m = (1u << a.l2_qw) - 1u;
  const int ci = local ? 0 : bid / a.spc, slab = local ? 0 : bid - ci * a.spc;
  const float slope = a.slope;
  const unsigned img_pitch = (unsigned)C * (unsigned)HW * 4u;  // bytes between two images of one channel
  const unsigned row_b = (unsigned)W * 4u;
  const unsigned long long win = ((unsigned long long)(a.Bs - 1) * C + 1ull) * HW * 4ull;  // bytes of a plane set's window

  // (local form: one pass of the loop body, vc = the block index)
  float4 g0[NQ], g1[NQ], x0[NQ], x1[NQ];
#pragma unroll
  for (int j = 0; j < NQ; ++j) g0[j] = g1[j] = x0[j] = x1[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  const unsigned qbase = (unsigned)(slab * (256 * NQ) + t);
  // byte offset of row 0 of quad q inside the window (or out of range)
  auto quad_off = [&](unsigned q) -> unsigned {
    const unsigned b = q >> a.l2_qpp, r = q & qpp_m, h2 = r >> a.l2_qw, w4 = r & qw_m;
    const unsigned off = b * img_pitch + (2u * h2 * (unsigned)W + 4u * w4) * 4u;
    return q < (unsigned)nq ? off : BF_OOB;
  };
  // the same quad in a half-resolution tensor [.][H/2][W/2] (pooled dy, dz block sums): 2 floats
  auto half_off = [&](unsigned q) -> unsigned {
    const unsigned b = q >> a.l2_qpp, r = q & qpp_m, h2 = r >> a.l2_qw, w4 = r & qw_m;
    const unsigned off = b * (img_pitch >> 2) + (h2 * (unsigned)(W >> 1) + 2u * w4) * 4u;
    return q < (unsigned)nq ? off : BF_OOB;
  };
  // x of plane set `vcn` (this block's slab of it) -> LDS, 2 * PFQ LDS-direct loads of 16 bytes per lane.  Issued for group
  // g + 1 between the arrival at group g's barrier and the wait: the memory pipe, otherwise idle until the last block has
  // arrived and the partial sums are folded, delivers a quarter of the next group's bytes meanwhile (round 5).
  auto request_x = [&](int vcn) {
    const int segn = vcn / C, cn = vcn - segn * C;
    const __amdgpu_buffer_rsrc_t rxn = make_rsrc(a.x + ((size_t)segn * a.Bs * C + cn) * (size_t)HW, win);
#pragma unroll
    for (int j = 0; j < PFQ; ++j) {
      const unsigned vo = quad_off(qbase + j * 256);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rxn, (float __attribute__((address_space(3)))*)(bf_pfx + (2 * j) * 256 + wave64),
                                               16, (int)vo, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rxn, (float __attribute__((address_space(3)))*)(bf_pfx + (2 * j + 1) * 256 + wave64),
                                               16, (int)vo, (int)row_b, 0, 0);
    }
  };
  if (pf && !local && ci < a.cpg && sub * a.cpg + ci < VC && sub < a.ngroups) request_x(sub * a.cpg + ci);
  for (int grp = sub; grp < (local ? 1 : a.ngroups); grp += a.nsub) {
    const int vc = local ? bid : grp * a.cpg + ci;
    const bool active = local ? true : (ci < a.cpg && vc < VC);
    double t1 = 0.0, t2 = 0.0;
    float m = 0.f, is = 0.f, gs = 0.f;
    int c = 0;
    size_t base = 0;  // element index of the plane (segment's first image, channel c)
    BF_STAMP(0)
    if (active) {
      const int seg = vc / C;
      c = vc - seg * C;
      m = a.mean[vc];
      is = a.invstd[vc];
      gs = a.gamma[c] * is;
      const float bt = ACT == 2 ? a.beta[c] : 0.f;
      base = ((size_t)seg * a.Bs * C + c) * (size_t)HW;
      const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x + base, win);
      const __amdgpu_buffer_rsrc_t rdy = POOL ? make_rsrc(a.dy + (base >> 2), win >> 2) : make_rsrc(a.dy + base, win);
      unsigned nib[ACT == 3 ? NQ : 1];
      // ---- phase 1: everything this block owns of the plane set goes into registers
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        const unsigned vo = quad_off(qbase + j * 256);
        // (the previous group's store data — dx in x0 / x1, dz in g0 / g1 — stays pinned up to here: see BF_KEEP)
        BF_KEEP(x0[j]) BF_KEEP(x1[j]) BF_KEEP(g0[j]) BF_KEEP(g1[j])
        if (!pf || j >= PFQ) {
          x0[j] = buf_load_f32x4(rx, vo, 0);
          x1[j] = buf_load_f32x4(rx, vo, row_b);
        }
        if (POOL) {
          const float2 d = buf_load_f32x2(rdy, half_off(qbase + j * 256), 0);
          g0[j] = make_float4(0.25f * d.x, 0.25f * d.x, 0.25f * d.y, 0.25f * d.y);
          g1[j] = g0[j];
        } else {
          g0[j] = buf_load_f32x4(rdy, vo, 0);
          g1[j] = buf_load_f32x4(rdy, vo, row_b);
        }
        if (ACT == 3) {
          // element e -> bit (e & 7) of byte e >> 3: row 0 of the quad is byte vo >> 5, nibble (vo >> 4) & 1; row 1 is
          // W / 8 bytes further with the same nibble parity (W % 8 == 0)
          const __amdgpu_buffer_rsrc_t rm = make_rsrc(a.mask + (base >> 3), win >> 5);
          const unsigned mo = vo == BF_OOB ? BF_OOB : (vo >> 5);
          const unsigned b0_ = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rm, (int)mo, 0, 0);
          const unsigned b1_ = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rm, (int)mo, (int)(W >> 3), 0);
          nib[j] = b0_ | (b1_ << 8);
        }
        if (ACT == 1) {
          const __amdgpu_buffer_rsrc_t ry = make_rsrc(a.y + base, win);
          bf_sign_val(g0[j], buf_load_f32x4(ry, vo, 0), slope);
          bf_sign_val(g1[j], buf_load_f32x4(ry, vo, row_b), slope);
        }
      }
      if (pf) {
        // x was requested a barrier ago (behind it in the in-order memory pipe: the previous group's stores and the loads
        // above, which are needed now anyway)
#pragma unroll
        for (int j = 0; j < PFQ; ++j) {
          x0[j] = bf_pfx[(2 * j) * 256 + t];
          x1[j] = bf_pfx[(2 * j + 1) * 256 + t];
        }
      }
      double s1 = 0.0, s2 = 0.0;
#ifdef BF_TIMING
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (stamp 1: every raw vector has landed)
#endif
      BF_STAMP(1)
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        if (ACT == 3) {
          const unsigned sh = (quad_off(qbase + j * 256) >> 4) & 1u ? 4u : 0u;
          bf_sign_nibble(g0[j], (nib[j] >> sh) & 0xfu, slope);
          bf_sign_nibble(g1[j], (nib[j] >> (8u + sh)) & 0xfu, slope);
        }
        if (ACT == 2) {
          float4 v = make_float4((x0[j].x - m) * gs + bt, (x0[j].y - m) * gs + bt, (x0[j].z - m) * gs + bt,
                                 (x0[j].w - m) * gs + bt);
          bf_sign_val(g0[j], v, slope);
          v = make_float4((x1[j].x - m) * gs + bt, (x1[j].y - m) * gs + bt, (x1[j].z - m) * gs + bt,
                          (x1[j].w - m) * gs + bt);
          bf_sign_val(g1[j], v, slope);
        }
        // x -> xhat in place (phase 2 needs only xhat); out-of-range quads have dz = 0 and contribute nothing
        x0[j] = make_float4((x0[j].x - m) * is, (x0[j].y - m) * is, (x0[j].z - m) * is, (x0[j].w - m) * is);
        x1[j] = make_float4((x1[j].x - m) * is, (x1[j].y - m) * is, (x1[j].z - m) * is, (x1[j].w - m) * is);
        // (fp32 within the quad, fp64 across quads: 8 products of one thread, then everything else in double)
        const float q1 = ((g0[j].x + g0[j].y) + (g0[j].z + g0[j].w)) + ((g1[j].x + g1[j].y) + (g1[j].z + g1[j].w));
        const float q2 = ((g0[j].x * x0[j].x + g0[j].y * x0[j].y) + (g0[j].z * x0[j].z + g0[j].w * x0[j].w)) +
                         ((g1[j].x * x1[j].x + g1[j].y * x1[j].y) + (g1[j].z * x1[j].z + g1[j].w * x1[j].w));
        s1 += (double)q1;
        s2 += (double)q2;
      }
      block_sum2<256>(s1, s2, red);
      if (local) {
        t1 = s1;
        t2 = s2;
      } else if (t == 0) {
        __hip_atomic_store(a.part + ((size_t)vc * a.spc + slab) * 2 + 0, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.part + ((size_t)vc * a.spc + slab) * 2 + 1, s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    BF_STAMP(2)
    if (!local) {
      // ---- barrier: every slab of every channel of this group is published
      if (t == 0) {
        ++target;
        bf_grid_arrive(bar, xcd, a.nx, bpx, nbar++);
      }
      BF_STAMP(3)
      if (pf) {
        const int vcn = (grp + a.nsub) * a.cpg + ci;
        if (grp + a.nsub < a.ngroups && ci < a.cpg && vcn < VC) request_x(vcn);
      }
      if (t == 0) bar_failed = bf_grid_wait(bar, a.bar + BF_POISON_WORD, xcd, target, a.spin_limit) ? 0 : 1;
      BF_STAMP(4)
      __syncthreads();
      if (bar_failed) return;  // abandoned launch (poison word set): no trap, no hang; the host raises
      if (!active) continue;
      // ---- phase 2: coefficients of this channel (fixed order: thread-strided slabs, then the block tree — the same
      // in every block of the channel), then dx straight from the registers
      // (a thread's slabs — two at most 512 — are requested together and added in slab order; one reduction for both sums)
      for (int s = t; s < a.spc; s += 512) {
        const int sb = s + 256 < a.spc ? s + 256 : s;
        const double p1 = bf_load_f64(a.part + ((size_t)vc * a.spc + s) * 2 + 0);
        const double p2 = bf_load_f64(a.part + ((size_t)vc * a.spc + s) * 2 + 1);
        const double q1 = bf_load_f64(a.part + ((size_t)vc * a.spc + sb) * 2 + 0);
        const double q2 = bf_load_f64(a.part + ((size_t)vc * a.spc + sb) * 2 + 1);
        t1 += p1;
        t2 += p2;
        if (s + 256 < a.spc) {
          t1 += q1;
          t2 += q2;
        }
      }
      block_sum2<256>(t1, t2, red);
    }
    const float c1 = (float)(t1 / a.count), c2 = (float)(t2 / a.count);
    BF_STAMP(5)
    if (slab == 0 && t == 0 && (a.dgamma != nullptr || a.dbeta != nullptr)) {
      if (a.nseg == 1) {
        if (a.dbeta) a.dbeta[c] = (float)t1;
        if (a.dgamma) a.dgamma[c] = (float)t2;
      } else {
        // publish this segment's sums; the last segment leader of the channel adds them in segment order
        __hip_atomic_store(a.sums + (size_t)vc * 2 + 0, t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.sums + (size_t)vc * 2 + 1, t2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned done = __hip_atomic_fetch_add(chcnt + c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == (unsigned)a.nseg - 1u) {
          double u1 = 0.0, u2 = 0.0;
          for (int g = 0; g < a.nseg; ++g) {
            u1 += bf_load_f64(a.sums + (size_t)(g * C + c) * 2 + 0);
            u2 += bf_load_f64(a.sums + (size_t)(g * C + c) * 2 + 1);
          }
          if (a.dbeta) a.dbeta[c] = (float)u1;
          if (a.dgamma) a.dgamma[c] = (float)u2;
          __hip_atomic_store(chcnt + c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    const __amdgpu_buffer_rsrc_t rdx = make_rsrc(a.dx + base, win);
    const __amdgpu_buffer_rsrc_t rdz = a.dzmode == 2 ? make_rsrc(a.dz + (base >> 2), win >> 2)
                                                     : make_rsrc(a.dzmode == 1 ? a.dz + base : a.dx + base, win);
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const unsigned vo = quad_off(qbase + j * 256);
      // (measured on gfx950: a 16-byte buffer store at voffset 0xFFFFFFFF is NOT dropped as a whole — dwords 1..3 wrap
      // into the window and zeros landed inside dz; so the stores of quads past the end are skipped by the exec mask and
      // the out-of-range marker keeps all four dwords of a vector beyond any window)
      // dx in place of xhat (dead after this): the registers a store reads are not rewritten before the next group
      x0[j].x = gs * (g0[j].x - c1 - x0[j].x * c2);
      x0[j].y = gs * (g0[j].y - c1 - x0[j].y * c2);
      x0[j].z = gs * (g0[j].z - c1 - x0[j].z * c2);
      x0[j].w = gs * (g0[j].w - c1 - x0[j].w * c2);
      x1[j].x = gs * (g1[j].x - c1 - x1[j].x * c2);
      x1[j].y = gs * (g1[j].y - c1 - x1[j].y * c2);
      x1[j].z = gs * (g1[j].z - c1 - x1[j].z * c2);
      x1[j].w = gs * (g1[j].w - c1 - x1[j].w * c2);
      if (vo == BF_OOB) continue;
      bf_store4(rdx, x0[j], vo, 0);
      bf_store4(rdx, x1[j], vo, row_b);
      if (a.dzmode == 1) {
        bf_store4(rdz, g0[j], vo, 0);
        bf_store4(rdz, g1[j], vo, row_b);
      } else if (a.dzmode == 2) {
        // same order as upsample2_bwd_kernel / bn_bwd_dx_dzsum_kernel: (row0.l + row0.r) + (row1.l + row1.r)
        const float sa = (g0[j].x + g0[j].y) + (g1[j].x + g1[j].y);
        const float sb = (g0[j].z + g0[j].w) + (g1[j].z + g1[j].w);
        buf_store_f32x2(rdz, sa, sb, half_off(qbase + j * 256), 0);
      }
    }
    BF_STAMP(6)
#ifdef BF_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BF_STAMP(7)
#endif
  }
  // the arrival counters of this half-grid back to zero (bn_fused_common.h: they run on through a launch's barriers)
  if (t == 0 && bid == 0 && nbar != 0u) bf_grid_reset(bar, a.nx);
  // the last group's stores: their data registers stay untouched until the stores have completed
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int j = 0; j < NQ; ++j) { BF_KEEP(x0[j]) BF_KEEP(x1[j]) BF_KEEP(g0[j]) BF_KEEP(g1[j]) }
}

struct BfPlan {
  int nq_per_thread;  // NQ
  int spc, cpg, ngroups, nb_sub, nsub;
  int local;  // plane sets of one block each: an ordinary launch of VC blocks
};

// do two blocks of the heaviest instantiation fit one CU on this device?  (They do by construction —
// __launch_bounds__(256, 2), a few hundred bytes of LDS —; the query guards against a runtime / driver that disagrees.
// Without a device — planning queries on a build box — the answer is yes.)
static bool bf_occupancy_ok() {
  static int ok = -1;
  if (ok < 0) {
    int ndev = 0, n = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
      (void)hipGetLastError();
      ok = 1;
    } else {
      // (8 quads per thread with the 48-KB request buffer is the LDS-heaviest launch, 10 quads the register-heaviest)
      hipError_t e = hipSuccess;
      int n8 = 0;
      if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n8, bn_bwd_fused_kernel<3, false, 8>, 256, 48 * 1024);
      if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, bn_bwd_fused_kernel<3, false, 10>, 256, 0);
      if (e == hipSuccess && n8 < n) n = n8;
      ok = (e != hipSuccess || n >= 2) ? 1 : 0;
      if (e != hipSuccess) (void)hipGetLastError();
    }
  }
  return ok == 1;
}

// the NQ (quads per thread) of {10, 8, 4} with the smallest modelled time: groups per (half-)grid x (fixed barrier /
// latency cost + streaming time of a group).  A channel plane set must fit one group (spc <= blocks of the (half-)grid).
static bool bf_plan(int Bs, int VC, int HW, int max_nq, BfPlan* out) {
  const long long nq = (long long)Bs * HW / 8;
  out->local = 0;
  // small plane sets (the 4x4 / 8x8 / 16x16 maps, small shards): one block holds a whole (segment, channel) — no
  // cross-block dependency, so no persistent grid and no barrier: VC ordinary blocks
  for (int NQ : {4, 8, 10}) {
    if (NQ > max_nq || nq > 256LL * NQ) continue;
    out->nq_per_thread = NQ;
    out->spc = 1;
    out->cpg = 1;
    out->ngroups = VC;
    out->nb_sub = VC;
    out->nsub = 1;
    out->local = 1;
    return true;
  }
  // the persistent form needs its whole grid (2 blocks per CU) resident: not under a CU mask, not when the runtime says
  // fewer than two blocks of the heaviest variant fit a CU -> the callers keep the three-launch form
  if (!bf_persistent_allowed() || !bf_occupancy_ok()) return false;
  // two half-grids (one block per CU each) when a plane set fits a half; otherwise one grid of two blocks per CU
  for (int nsub = 2; nsub >= 1; --nsub) {
    const int nb_sub = sivae_num_cus() * (nsub == 2 ? 1 : 2);
    double best = 0.0;
    bool found = false;
    static int force_nq = -1;  // SIVAE_BN_FUSED_NQ=<10|8|4>: pin the quads per thread of the persistent form (A/B)
    if (force_nq < 0) {
      const char* e = getenv("SIVAE_BN_FUSED_NQ");
      force_nq = e ? atoi(e) : 0;
    }
    for (int NQ : {10, 8, 4}) {
      if (NQ > max_nq) continue;
      if (force_nq > 0 && NQ != force_nq && NQ <= max_nq && force_nq <= max_nq) continue;
      const long long slabq = 256LL * NQ;
      const long long spc = (nq + slabq - 1) / slabq;
      if (spc > nb_sub) continue;
      const int cpg = (int)(nb_sub / spc);
      const int ngroups = (VC + cpg - 1) / cpg;
      const double cost = (double)((ngroups + nsub - 1) / nsub) * (8.0 + 2.5 * NQ);
      if (!found || cost < best) {
        best = cost;
        found = true;
        out->nq_per_thread = NQ;
        out->spc = (int)spc;
        out->cpg = cpg;
        out->ngroups = ngroups;
        out->nb_sub = nb_sub;
        out->nsub = nsub;
      }
    }
    if (found) return true;
  }
  return false;
}

static inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

#ifdef BF_TIMING
static unsigned long long* bf_timing_buffer = nullptr;
#endif

}  // namespace

#ifdef BF_TIMING
// timing build only (tools/bn_fused_timing.py)
extern "C" void sivae_debug_bf_timing(unsigned long long* buf) { bf_timing_buffer = buf; }
extern "C" int sivae_debug_bf_plan(int B, int C, int H, int W, int seg_images, int max_nq, int* out6) {
  BfPlan p;
  if (!bf_plan(seg_images, (B / seg_images) * C, H * W, max_nq, &p)) return -1;
  out6[0] = p.nq_per_thread; out6[1] = p.spc; out6[2] = p.cpg; out6[3] = p.ngroups; out6[4] = p.nb_sub; out6[5] = p.nsub;
  return 0;
}
#endif

// shapes the one-pass backward takes: power-of-two maps from 4x4 up (H even, W % 4 == 0 by construction), per-(segment,
// channel) plane sets that fit one group of a half-grid, at most 8192 channels
// (plans with the register budget of the tightest variant — act_mode 1, 8 quads per thread —, so that a shape this
// query accepts is accepted by every variant of the launch)
extern "C" int sivae_bn_bwd_fused_supported(int B, int C, int H, int W, int seg_images) {
  if (B <= 0 || C <= 0 || C > BF_CH_COUNTERS || !is_pow2(H) || !is_pow2(W) || H < 2 || W < 4) return 0;
  if (seg_images <= 0 || B % seg_images != 0) return 0;
  if ((long long)B * C * H * W >= 0xffffffffLL) return 0;
  if ((long long)seg_images * C * H * W * 4 >= 0xfffffe00LL) return 0;  // (a plane set's window: 32-bit byte offsets)
  BfPlan p;
  return bf_plan(seg_images, (B / seg_images) * C, H * W, 8, &p) ? 1 : 0;
}

// index (in unsigned ints) of the state word a timed-out grid barrier sets: the host checks it where it reads results
// back anyway (sivae_hip.ops.bn_fused_check) and must zero the whole state before the next launch when it is non-zero
extern "C" int sivae_bn_bwd_fused_poison_word() { return BF_POISON_WORD; }

// partial sums [VC][spc][2] + per-(segment, channel) sums [VC][2], doubles
extern "C" size_t sivae_bn_bwd_fused_workspace_bytes(int B, int C, int H, int W, int seg_images) {
  if (!sivae_bn_bwd_fused_supported(B, C, H, W, seg_images)) return 0;
  const int VC = (B / seg_images) * C;
  // (the launch picks its quads-per-thread from the variant's register budget: cover every choice)
  BfPlan p;
  size_t worst = 0;
  for (int mq : {10, 8, 4})
    if (bf_plan(seg_images, VC, H * W, mq, &p) && (size_t)p.spc > worst) worst = (size_t)p.spc;
  return ((size_t)VC * worst * 2 + (size_t)VC * 2) * sizeof(double);
}

// size (in unsigned ints) of the barrier / counter state sivae_bn_bwd_fused needs: zero-initialised ONCE by the caller,
// left consistent by every call; one buffer per stream (two calls sharing a buffer must not run concurrently)
extern "C" int sivae_bn_bwd_fused_state_uints() { return 2 * BF_BAR_UINTS + BF_CH_COUNTERS; }

// Same contract as sivae_bn_bwd_seg (bn.hip): every backward variant, B = nseg * seg_images images with per-(segment,
// channel) statistics (mean / invstd [nseg][C]); dgamma / dbeta [C] summed over the segments.
//   act_mode 0 none, 1 sign from the saved output y, 2 sign recomputed from x (needs beta), 3 sign from `mask`
//   dy_pooled: dy is the gradient of AvgPool2d(2)(output) ([B][C][H/2][W/2]); dz_sum: dz_out receives the 2x2 block sums
// The launch needs its whole grid (2 blocks per CU) resident: do not run it concurrently with another persistent kernel
// on the same device.
extern "C" int sivae_bn_bwd_fused(const float* dy, const float* y, const unsigned char* mask, const float* x,
                                  const float* mean, const float* invstd, const float* gamma, const float* beta,
                                  int act_mode, float slope, float* dx, float* dz_out, float* dgamma, float* dbeta, int B,
                                  int C, int H, int W, int dy_pooled, int dz_sum, int seg_images, unsigned int* state,
                                  void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!dy || !x || !mean || !invstd || !gamma || !dx || !state) return SIVAE_ERR_NULL;
  if (act_mode < 0 || act_mode > 3) return SIVAE_ERR_MODE;
  if (act_mode == 1 && !y) return SIVAE_ERR_NULL;
  if (act_mode == 2 && !beta) return SIVAE_ERR_NULL;
  if (act_mode == 3 && !mask) return SIVAE_ERR_NULL;
  if (act_mode == 3 && (W & 7)) return SIVAE_ERR_SHAPE;  // (a row of the 1-bit mask is whole bytes)
  if (dy_pooled && dz_sum) return SIVAE_ERR_MODE;
  if (dz_sum && !dz_out) return SIVAE_ERR_NULL;
  if (!sivae_bn_bwd_fused_supported(B, C, H, W, seg_images)) return SIVAE_ERR_SHAPE;
  if (((uintptr_t)dy & 15u) || ((uintptr_t)x & 15u) || ((uintptr_t)dx & 15u) || (dz_out && ((uintptr_t)dz_out & 15u)) ||
      (y && ((uintptr_t)y & 15u)))
    return SIVAE_ERR_SHAPE;
  const int nseg = B / seg_images, VC = nseg * C;
  BfPlan p;
  // (the sign-from-y variant carries two more transient vectors per quad)
  if (!bf_plan(seg_images, VC, H * W, act_mode == 1 ? 8 : 10, &p)) return SIVAE_ERR_SHAPE;
  const size_t need = ((size_t)VC * p.spc * 2 + (size_t)VC * 2) * sizeof(double);
  if (!workspace || workspace_bytes < need) return SIVAE_ERR_WORKSPACE;
  BnFusedArgs a;
  a.dy = dy;
  a.y = y;
  a.x = x;
  a.mask = mask;
  a.mean = mean;
  a.invstd = invstd;
  a.gamma = gamma;
  a.beta = beta;
  a.dx = dx;
  a.dz = dz_out;
  a.dgamma = dgamma;
  a.dbeta = dbeta;
  a.part = (double*)workspace;
  a.sums = a.part + (size_t)VC * p.spc * 2;
  a.bar = state;
  a.count = (double)seg_images * H * W;
  a.slope = slope;
  a.C = C;
  a.H = H;
  a.W = W;
  a.Bs = seg_images;
  a.nseg = nseg;
  a.l2_qpp = ilog2_exact(H * W / 8);
  a.l2_qw = ilog2_exact(W / 4);
  a.spc = p.spc;
  a.cpg = p.cpg;
  a.ngroups = p.ngroups;
  a.nx = (!p.local && p.nb_sub % 8 == 0) ? 8 : 1;
  a.nsub = p.nsub;
  a.local = p.local;
  a.dzmode = !dz_out ? 0 : (dz_sum ? 2 : 1);
  a.spin_limit = bf_spin_limit();
#ifdef BF_TIMING
  a.ts = bf_timing_buffer;
#endif
  // the next group's x requested into LDS across the barrier (up to 6 quads per thread = 48 KB per block): persistent
  // forms with up to 8 quads per thread and more than one group per (half-)grid
  static int pf_on = -1;
  if (pf_on < 0) {
    const char* e = getenv("SIVAE_BN_FUSED_PREFETCH");
    pf_on = (e && e[0] == '0') ? 0 : 1;
  }
  a.pf = (pf_on && !p.local && p.nq_per_thread <= 8 && p.ngroups > p.nsub) ? 1 : 0;
  const size_t lds = a.pf ? (size_t)(p.nq_per_thread < 6 ? p.nq_per_thread : 6) * 2 * 256 * sizeof(float4) : 0;
  const dim3 grid((unsigned)(p.local ? VC : p.nsub * p.nb_sub)), block(256);
#define BF_LAUNCH(A, P, Q) hipLaunchKernelGGL((bn_bwd_fused_kernel<A, P, Q>), grid, block, lds, stream, a);  // (lds <= 48 KB)
#define BF_NQ(A, P)                                 \
  {                                                 \
    if (p.nq_per_thread == 10) BF_LAUNCH(A, P, 10) \
    else if (p.nq_per_thread == 8) BF_LAUNCH(A, P, 8) \
    else BF_LAUNCH(A, P, 4)                         \
  }
#define BF_ACT(A)              \
  {                            \
    if (dy_pooled) BF_NQ(A, true) else BF_NQ(A, false) \
  }
  if (act_mode == 0) BF_ACT(0) else if (act_mode == 1) BF_ACT(1) else if (act_mode == 2) BF_ACT(2) else BF_ACT(3)
#undef BF_ACT
#undef BF_NQ
#undef BF_LAUNCH
  return sivae_launch_status();
}
