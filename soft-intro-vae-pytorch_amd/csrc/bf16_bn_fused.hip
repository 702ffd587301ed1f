// BatchNorm backward of the bf16 mode (blocked bf16 activations x[b][c/8][h][w][c%8], bf16_common.h) as ONE launch that
// reads dy and x once and writes dx once — the bf16 counterpart of bn_fused.hip (same structure: a persistent grid of two
// independent half-grids, plane sets walked in groups whose raw vectors stay in registers across a fence-free grid
// barrier; or, for plane sets that fit one block, an ordinary barrier-free launch).  The three-launch form of
// bf16_bn.hip (bf16_bn_bwd_partial_kernel -> finalize -> bf16_bn_bwd_apply_kernel) streams dy and x twice: 21 % of a
// 128x128 bf16 iteration.
//
// Unit of work: an 8-channel block `cb` (one 16-byte vector per pixel), a thread's payload = NU 2x2-pixel quads of raw
// dy and x vectors (32 registers per quad) + their 4 sign bytes; the per-channel sums are fp32 per thread (8 channels x
// {sum g, sum g*xhat}), fp32 across the block, fp64 across slabs (fixed order).  Phase 2 recomputes g and xhat from the
// raw vectors with the arithmetic of bf16_bn_bwd_apply_kernel.
//
// Reference op: backward of nn.BatchNorm2d + nn.LeakyReLU(0.2) (+ torch.add, nn.AvgPool2d(2) behind / nn.Upsample(2) in
// front of the block), soft_intro_vae/train_soft_intro_vae.py:57-63,71-74,90-93,98,155 — in config 3's bf16 storage.
#include "bf16_common.h"
#include "bn_fused_common.h"

namespace {

struct Bf16BnFusedArgs {
  const void* dy;
  const void* y;
  const void* x;
  const unsigned char* mask;
  const float* mean;
  const float* invstd;
  const float* gamma;
  const float* beta;
  void* dx;
  void* dz;
  float* dgamma;
  float* dbeta;
  double* part;  // [VCb][spc][16]
  double* sums;  // [VCb][16]: a (segment, channel block)'s 16 sums, published for the dgamma / dbeta fold (nseg > 1)
  unsigned* bar;
  float inv_n;
  float slope;
  int C, Cb, H, W, B;  // B: images of ONE segment (a plane set = one channel block of one segment's images)
  int nseg, VCb;       // segments of the batch; VCb = nseg * Cb "virtual" channel blocks, segment-major
  int l2_qpp, l2_qw;  // log2(quads per plane), log2(quads per row); a quad = 2 x 2 pixels
  int spc, cpg, ngroups, nx, nsub, local;
  int dzmode;  // 0 none, 1 full resolution, 2 2x2 block sums
  int pf;      // 1: the NEXT group's x is requested into LDS between the barrier's arrival and its wait (as bn_fused.hip)
  unsigned spin_limit;  // polls of the barrier wait before the launch is abandoned (bn_fused_common.h)
#ifdef B16_TIMING
  unsigned long long* ts;  // [block][group][8] s_memrealtime stamps of thread 0 (tools/b16_bn_timing.py; not a product build)
#endif
};
#ifdef B16_TIMING
#define B16_STAMP(K) \
  if (t == 0 && a.ts) a.ts[((size_t)blockIdx.x * a.ngroups + grp) * 8 + (K)] = __builtin_amdgcn_s_memrealtime();
#else
#define B16_STAMP(K)
#endif

// 16-byte buffer stores and the lifetime of their data registers.  Measured on gfx950 / ROCm 7.2 (this kernel, memory pipe
// saturated, round 4): when the VALU instructions that follow a buffer_store_dwordx4 rewrite its data registers within a
// few issue slots, the store can pick up the NEW values in lanes 12-15 of every 16-lane row (the last quad of each row is
// read last) — the two wait states hipcc inserts for the store-data hazard are not enough under back-pressure.  Round 4
// padded every store with idle cycles; since round 5 the data of the stores lives in registers that nothing rewrites
// before the next group's loads (packed dx replaces the raw x vector it was computed from, packed dz the raw dy vector;
// B16_KEEP pins them up to the loads that overwrite them — VMEM executes in order — and the kernel's end waits vmcnt(0)):
// see bn_fused.hip::BF_KEEP.  The one variant without a dead payload register for dz (pooled dy + full-resolution dz)
// keeps the padded store.
__device__ __forceinline__ void bf_store_u32x4(__amdgpu_buffer_rsrc_t r, u32x4_t v, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ void bf_store_u32x4_padded(__amdgpu_buffer_rsrc_t r, u32x4_t v, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)voff, (int)soff, 0);
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
#define B16_KEEP(V) asm volatile("" ::"v"((V)[0]), "v"((V)[1]), "v"((V)[2]), "v"((V)[3]));

// ACT: 1 sign from the saved output y, 2 recomputed from xhat * gamma + beta, 3 from the sign bytes
template <int ACT, bool POOL, int NU>
__global__ void __launch_bounds__(256, 2) bf16_bn_bwd_fused_kernel(Bf16BnFusedArgs a) {
  __shared__ float red[4][16];
  __shared__ double red2[32][16];
  __shared__ float coef[16];  // c1[8] = sg / N, c2[8] = sgx / N of this block's channel block
  // LeakyReLU-derivative factors of four sign bits at a time: entry n = {bit0 ? 1 : slope, ..., bit3 ? 1 : slope}.  One
  // ds_read_b128 per nibble (16 entries x 16 bytes = one bank row: distinct entries never conflict, equal ones broadcast)
  // instead of and + compare + select per element — the sign handling was 37 % of phase 2's vector instructions
  __shared__ float4 sel_lut[16];
  static_assert(sizeof(red) + sizeof(red2) + sizeof(coef) + sizeof(sel_lut) + 16 + 48 * 1024 <= 160 * 1024 / 3,
                "static LDS + the 48-KB request buffer must stay under a third of the CU's LDS (co-residency, see b16_occupancy)");
  // x of the NEXT group for up to 3 of the 4 units per thread (48 KB per block — the bound of bn_fused.hip's request
  // buffer, for the same reasons): [unit][vector][256 threads] 16-byte vectors, a wave's 64 lanes 1 KB contiguous
  extern __shared__ __attribute__((aligned(16))) u32x4_t b16_pfx[];
  constexpr int PFU = NU < 3 ? NU : 3;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wave64 = __builtin_amdgcn_readfirstlane(t >> 6) * 64;
  const bool pf = NU == 4 && a.pf != 0;
  const bool local = a.local != 0;
  const int nb_sub = local ? (int)gridDim.x : (int)gridDim.x / a.nsub;
  const int sub = (!local && (int)blockIdx.x >= nb_sub) ? 1 : 0;
  const int bid = (int)blockIdx.x - sub * nb_sub;
  unsigned* bar = a.bar + sub * BF_BAR_UINTS;
  unsigned* chcnt = a.bar + 2 * BF_BAR_UINTS;  // per-channel-block segment-arrival counters (bn_fused.hip's, zero at rest)
  const int xcd = bid % a.nx;
  const unsigned bpx = (unsigned)(nb_sub / a.nx);
  __shared__ int bar_failed;
  unsigned target = 0, nbar = 0;  // (thread 0) generation to wait for; barriers of this launch passed so far
  if (t == 0 && !local) target = bf_load_u32(bar + (9 + xcd) * 32);
  if (t < 16) sel_lut[t] = float4{(t & 1) ? 1.f : a.slope, (t & 2) ? 1.f : a.slope, (t & 4) ? 1.f : a.slope, (t & 8) ? 1.f : a.slope};
  __syncthreads();
  const int W = a.W, HW = a.H * a.W, Cb = a.Cb;
  const int nq = a.B << a.l2_qpp;
  const unsigned qpp_m = (1u << a.l2_qpp) - 1u, qw_m = (1u << a.l2_qw) - 1u;
  const int ci = local ? 0 : bid / a.spc, slab = local ? 0 : bid - ci * a.spc;
  const float slope = a.slope;
  const unsigned img_pitch = (unsigned)Cb * (unsigned)HW * 16u;  // bytes between two images of one channel block
  const unsigned row_b = (unsigned)W * 16u;
  const unsigned long long win = ((unsigned long long)(a.B - 1) * Cb + 1ull) * HW * 16ull;

  u32x4_t dq[NU][POOL ? 1 : 4], xq[NU][4];
#pragma unroll
  for (int j = 0; j < NU; ++j)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      xq[j][v] = u32x4_t{0u, 0u, 0u, 0u};
      if (!POOL || v == 0) dq[j][POOL ? 0 : v] = u32x4_t{0u, 0u, 0u, 0u};
    }
  const unsigned qbase = (unsigned)(slab * (256 * NU) + t);
  auto quad_off = [&](unsigned q) -> unsigned {  // byte offset of the quad's first vector (row 0) inside the window
    const unsigned b = q >> a.l2_qpp, r = q & qpp_m, h2 = r >> a.l2_qw, w2 = r & qw_m;
    return q < (unsigned)nq ? b * img_pitch + (2u * h2 * (unsigned)W + 2u * w2) * 16u : BF_OOB;
  };
  auto half_off = [&](unsigned q) -> unsigned {  // the quad's vector in a half-resolution tensor [.][H/2][W/2][8]
    const unsigned b = q >> a.l2_qpp, r = q & qpp_m;
    return q < (unsigned)nq ? b * (img_pitch >> 2) + r * 16u : BF_OOB;
  };
  auto request_x = [&](int vcbn) {  // x of virtual channel block vcbn (this block's slab) -> LDS, 4 * PFU LDS-direct loads
    const int segn = vcbn / Cb, cbn = vcbn - segn * Cb;
    const __amdgpu_buffer_rsrc_t rxn =
        make_rsrc(reinterpret_cast<const char*>(a.x) + ((size_t)segn * a.B * Cb + cbn) * HW * 16, win);
#pragma unroll
    for (int j = 0; j < PFU; ++j) {
      const unsigned vo = quad_off(qbase + j * 256);
      const unsigned vo1 = vo == BF_OOB ? BF_OOB : vo + 16u;
#pragma unroll
      for (int v = 0; v < 4; ++v)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rxn, (float __attribute__((address_space(3)))*)(b16_pfx + (4 * j + v) * 256 + wave64),
                                                 16, (int)((v & 1) ? vo1 : vo), (int)((v >> 1) ? row_b : 0u), 0, 0);
    }
  };
  if (pf && !local && ci < a.cpg && sub * a.cpg + ci < a.VCb && sub < a.ngroups) request_x(sub * a.cpg + ci);
  for (int grp = sub; grp < (local ? 1 : a.ngroups); grp += a.nsub) {
    const int vcb = local ? bid : grp * a.cpg + ci;  // (segment, channel block), segment-major
    const bool active = local ? true : (ci < a.cpg && vcb < a.VCb);
    const int seg = vcb / Cb, cb = vcb - seg * Cb;
    unsigned sg_bits[NU];  // sign bytes of the quad's 4 vectors (row 0: bytes 0, 1; row 1: bytes 2, 3)
    float mu[8], is[8];
    double tsum = 0.0;  // (threads t < 256: value t & 15 of the channel block's 16 sums)
    B16_STAMP(0)
    if (active) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = cb * 8 + e;
        const bool ok = c < a.C;
        mu[e] = ok ? a.mean[seg * a.C + c] : 0.f;
        is[e] = ok ? a.invstd[seg * a.C + c] : 0.f;
      }
      const size_t base = ((size_t)seg * a.B * Cb + cb) * HW;  // vector index of the plane in the segment's first image
      const __amdgpu_buffer_rsrc_t rx = make_rsrc(reinterpret_cast<const char*>(a.x) + base * 16, win);
      const __amdgpu_buffer_rsrc_t rdy =
          POOL ? make_rsrc(reinterpret_cast<const char*>(a.dy) + (base >> 2) * 16, win >> 2)
               : make_rsrc(reinterpret_cast<const char*>(a.dy) + base * 16, win);
      // ---- phase 1: raw vectors into registers
#pragma unroll
      for (int j = 0; j < NU; ++j) {
        const unsigned vo = quad_off(qbase + j * 256);
        // (the previous group's store data stays pinned in these registers up to here)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          B16_KEEP(xq[j][v])
          if (!POOL || v == 0) B16_KEEP(dq[j][POOL ? 0 : v])
        }
        if (!pf || j >= PFU) {
          xq[j][0] = buf_load_u32x4(rx, vo, 0);
          xq[j][1] = buf_load_u32x4(rx, vo == BF_OOB ? BF_OOB : vo + 16u, 0);
          xq[j][2] = buf_load_u32x4(rx, vo, row_b);
          xq[j][3] = buf_load_u32x4(rx, vo == BF_OOB ? BF_OOB : vo + 16u, row_b);
        }
        if (POOL) {
          dq[j][0] = buf_load_u32x4(rdy, half_off(qbase + j * 256), 0);
        } else {
          dq[j][0] = buf_load_u32x4(rdy, vo, 0);
          dq[j][1] = buf_load_u32x4(rdy, vo == BF_OOB ? BF_OOB : vo + 16u, 0);
          dq[j][2] = buf_load_u32x4(rdy, vo, row_b);
          dq[j][3] = buf_load_u32x4(rdy, vo == BF_OOB ? BF_OOB : vo + 16u, row_b);
        }
        if (ACT == 3) {
          // one sign byte per vector: bytes vo/16, vo/16 + 1 (row 0) and + W (row 1): two 16-bit loads
          const __amdgpu_buffer_rsrc_t rm = make_rsrc(a.mask + base, win >> 4);
          const unsigned mo = vo == BF_OOB ? BF_OOB : (vo >> 4);
          const unsigned m0 = (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rm, (int)mo, 0, 0);
          const unsigned m1 = (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rm, (int)mo, (int)W, 0);
          sg_bits[j] = m0 | (m1 << 16);
        } else if (ACT == 1) {
          const __amdgpu_buffer_rsrc_t ry = make_rsrc(reinterpret_cast<const char*>(a.y) + base * 16, win);
          unsigned bits = 0;
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const u32x4_t yv = buf_load_u32x4(ry, vo == BF_OOB ? BF_OOB : vo + (v & 1) * 16u, (v >> 1) ? row_b : 0u);
            unsigned m = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              m |= (bf16_lo(yv[i]) > 0.f ? 1u : 0u) << (2 * i);
              m |= (bf16_hi(yv[i]) > 0.f ? 1u : 0u) << (2 * i + 1);
            }
            bits |= m << (8 * v);
          }
          sg_bits[j] = bits;
        } else {
          sg_bits[j] = 0;
        }
      }
      if (pf) {
        // (requested a barrier ago: older in the in-order memory pipe than everything issued above)
#pragma unroll
        for (int j = 0; j < PFU; ++j)
#pragma unroll
          for (int v = 0; v < 4; ++v) xq[j][v] = b16_pfx[(4 * j + v) * 256 + t];
      }
      float gm[8], bt[8];
      if (ACT == 2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int c = cb * 8 + e;
          gm[e] = c < a.C ? a.gamma[c] : 0.f;
          bt[e] = c < a.C ? a.beta[c] : 0.f;
        }
      }
      float sg[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, sgx[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#ifdef B16_TIMING
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (stamp 1: every raw vector has landed)
#endif
      B16_STAMP(1)
#pragma unroll
      for (int j = 0; j < NU; ++j) {
        float dp[8];
        if (POOL) {
          unpack8(dq[j][0], dp);
#pragma unroll
          for (int e = 0; e < 8; ++e) dp[e] *= 0.25f;
        }
        unsigned bits = sg_bits[j];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          float d[8], xv[8];
          if (POOL) {
#pragma unroll
            for (int e = 0; e < 8; ++e) d[e] = dp[e];
          } else {
            unpack8(dq[j][v], d);
          }
          unpack8(xq[j][v], xv);
          unsigned m = 0;
          float sl[8];
          if (ACT != 2) {
            const float4 s0 = sel_lut[(bits >> (8 * v)) & 15u], s1 = sel_lut[(bits >> (8 * v + 4)) & 15u];
            sl[0] = s0.x; sl[1] = s0.y; sl[2] = s0.z; sl[3] = s0.w;
            sl[4] = s1.x; sl[5] = s1.y; sl[6] = s1.z; sl[7] = s1.w;
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float xh = (xv[e] - mu[e]) * is[e];
            float g;
            if (ACT == 2) {
              const bool pos = xh * gm[e] + bt[e] > 0.f;
              m |= (pos ? 1u : 0u) << e;
              g = d[e] * (pos ? 1.f : slope);
            } else {
              g = d[e] * sl[e];
            }
            sg[e] += g;
            sgx[e] += g * xh;
          }
          if (ACT == 2) bits |= m << (8 * v);
        }
        if (ACT == 2) sg_bits[j] = bits;  // (phase 2 reads the bits whatever their source)
      }
      // block fold of the 16 sums: wave butterflies, then the four waves through LDS
      if constexpr (ACT == 2 && NU == 4) {
        // (this instantiation has no register to spare: its 32 per-channel scalars already spill into VGPR lanes, and the
        // selects of the transposing form pushed 180 payload registers into scratch)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float s1 = wave_sum(sg[e]), s2 = wave_sum(sgx[e]);
          if (lane == 0) {
            red[wave][e] = s1;
            red[wave][8 + e] = s2;
          }
        }
      } else {
        // two transposing reductions of 8 values (common.h): lane l ends with the wave's total of value l & 7 — 20 VALU
        // + 6 cross-lane operations instead of 96 ds_bpermute butterflies on the path to the grid barrier
        wave_transpose_sum8(sg, lane);
        wave_transpose_sum8(sgx, lane);
        if (lane < 8) {
          red[wave][lane] = sg[0];
          red[wave][8 + lane] = sgx[0];
        }
      }
      __syncthreads();
      if (t < 16) {
        const float s = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
        if (local)
          tsum = (double)s;
        else
          __hip_atomic_store(a.part + ((size_t)vcb * a.spc + slab) * 16 + t, (double)s, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    B16_STAMP(2)
    if (!local) {
      if (t == 0) {
        ++target;
        bf_grid_arrive(bar, xcd, a.nx, bpx, nbar++);
      }
      B16_STAMP(3)
      if (pf) {
        const int cbn = (grp + a.nsub) * a.cpg + ci;
        if (grp + a.nsub < a.ngroups && ci < a.cpg && cbn < a.VCb) request_x(cbn);
      }
      if (t == 0) bar_failed = bf_grid_wait(bar, a.bar + BF_POISON_WORD, xcd, target, a.spin_limit) ? 0 : 1;
      B16_STAMP(4)
      __syncthreads();
      if (bar_failed) return;  // abandoned launch (poison word set; the host raises)
      if (!active) continue;
      // the channel block's 16 sums over its slabs: thread (value v = t & 15, row r = t >> 4) strides over the slabs,
      // rows folded in order through LDS — the same in every block of the channel block
      // (16-byte agent-scope loads: thread = (value pair p = t & 7, row r = t >> 3), eight in flight per thread, added in
      // slab order — 512 slabs are two memory round trips; with 8-byte loads, 16 rows and one dependent load per step this
      // fold was 32 round trips = 8.8 of a 37-us group on the 128 x 128 layers, tools/b16_bn_timing.py)
      const int pr = t & 7, r = t >> 3;
      double acc0 = 0.0, acc1 = 0.0;
      const __amdgpu_buffer_rsrc_t rpart = make_rsrc(a.part + (size_t)vcb * a.spc * 16, (unsigned long long)a.spc * 128ull);
      for (int s0 = r; s0 < a.spc; s0 += 32 * 8) {
        u32x4_t tmp[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int sk = s0 + 32 * k;
          // (cache policy 16 = sc1: agent scope, like bf_load_f64; past the end: out of the window, zeros, not added)
          tmp[k] = __builtin_amdgcn_raw_buffer_load_b128(rpart, sk < a.spc ? (sk * 128 + pr * 16) : (int)BF_OOB, 0, 16);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (s0 + 32 * k < a.spc) {
            acc0 += __builtin_bit_cast(double, ((unsigned long long)tmp[k][1] << 32) | tmp[k][0]);
            acc1 += __builtin_bit_cast(double, ((unsigned long long)tmp[k][3] << 32) | tmp[k][2]);
          }
      }
      red2[r][2 * pr] = acc0;
      red2[r][2 * pr + 1] = acc1;
      __syncthreads();
      if (t < 16) {
        double u = 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) u += red2[k][t];
        tsum = u;
      }
    }
    if (t < 16) {
      coef[t] = (float)tsum * a.inv_n;
      if (slab == 0 && (a.dbeta != nullptr || a.dgamma != nullptr)) {
        const int c = cb * 8 + (t & 7);
        if (a.nseg == 1) {
          if (c < a.C) {
            if (t < 8 && a.dbeta) a.dbeta[c] = (float)tsum;
            if (t >= 8 && a.dgamma) a.dgamma[c] = (float)tsum;
          }
        } else {
          // dgamma / dbeta sum over the segments of a channel: publish this segment's 16 sums; the LAST segment leader
          // of the channel block to arrive adds them in segment order (bn_fused.hip's protocol: the result does not
          // depend on which one is last) and puts the counter back to zero
          __hip_atomic_store(a.sums + (size_t)vcb * 16 + t, tsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          unsigned done = 0;
          if (t == 0) done = __hip_atomic_fetch_add(chcnt + cb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          done = (unsigned)__builtin_amdgcn_readfirstlane((int)done);
          if (done == (unsigned)a.nseg - 1u) {
            double u = 0.0;
            for (int g = 0; g < a.nseg; ++g) u += bf_load_f64(a.sums + ((size_t)g * Cb + cb) * 16 + t);
            if (c < a.C) {
              if (t < 8 && a.dbeta) a.dbeta[c] = (float)u;
              if (t >= 8 && a.dgamma) a.dgamma[c] = (float)u;
            }
            if (t == 0) __hip_atomic_store(chcnt + cb, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
    }
    __syncthreads();
    B16_STAMP(5)
    // ---- phase 2
    // dx = gi (g - c1 - xhat c2) with xhat = (x - mu) is, as two fused multiply-adds per element on the raw x:
    // dx = g gi + (x bc + cc), bc = -gi c2 is, cc = gi (c2 is mu - c1)  (24 per-channel scalars instead of 40)
    float gi[8], bc[8], cc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = cb * 8 + e;
      gi[e] = (c < a.C ? a.gamma[c] : 0.f) * is[e];
      const float c2i = coef[8 + e] * is[e];
      bc[e] = -gi[e] * c2i;
      cc[e] = gi[e] * (c2i * mu[e] - coef[e]);
    }
    const size_t base = ((size_t)seg * a.B * Cb + cb) * HW;
    const __amdgpu_buffer_rsrc_t rdx = make_rsrc(reinterpret_cast<char*>(a.dx) + base * 16, win);
    const __amdgpu_buffer_rsrc_t rdz =
        a.dzmode == 2 ? make_rsrc(reinterpret_cast<char*>(a.dz) + (base >> 2) * 16, win >> 2)
                      : make_rsrc(reinterpret_cast<char*>(a.dzmode == 1 ? a.dz : a.dx) + base * 16, win);
#pragma unroll
    for (int j = 0; j < NU; ++j) {
      const unsigned vo = quad_off(qbase + j * 256);
      if (vo == BF_OOB) continue;
      float dp[8];
      if (POOL) {
        unpack8(dq[j][0], dp);
#pragma unroll
        for (int e = 0; e < 8; ++e) dp[e] *= 0.25f;
      }
      float zs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const unsigned bits = sg_bits[j];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        float d[8], xv[8], g[8];
        if (POOL) {
#pragma unroll
          for (int e = 0; e < 8; ++e) d[e] = dp[e];
        } else {
          unpack8(dq[j][v], d);
        }
        unpack8(xq[j][v], xv);
        const float4 s0 = sel_lut[(bits >> (8 * v)) & 15u], s1 = sel_lut[(bits >> (8 * v + 4)) & 15u];
        const float sl[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          g[e] = d[e] * sl[e];
          zs[e] += g[e];
          xv[e] = fmaf(g[e], gi[e], fmaf(xv[e], bc[e], cc[e]));
        }
        const unsigned vv = vo + (v & 1) * 16u, so = (v >> 1) ? row_b : 0u;
        // store data in place of the raw vectors it was computed from (dead from here on)
        xq[j][v] = pack8(xv);
        if (a.dzmode == 1) {
          if (POOL) {
            bf_store_u32x4_padded(rdz, pack8(g), vv, so);
          } else {
            dq[j][v] = pack8(g);
            bf_store_u32x4(rdz, dq[j][v], vv, so);
          }
        }
        bf_store_u32x4(rdx, xq[j][v], vv, so);
      }
      if (a.dzmode == 2) {  // (never with pooled dy: the raw dy vectors are dead here)
        dq[j][0] = pack8(zs);
        bf_store_u32x4(rdz, dq[j][0], half_off(qbase + j * 256), 0);
      }
    }
    B16_STAMP(6)
    __syncthreads();  // (coef / red are reused by the next group)
    B16_STAMP(7)
  }
  // the arrival counters of this half-grid back to zero (bn_fused_common.h: they run on through a launch's barriers)
  if (t == 0 && bid == 0 && nbar != 0u) bf_grid_reset(bar, a.nx);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int j = 0; j < NU; ++j)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      B16_KEEP(xq[j][v])
      if (!POOL || v == 0) B16_KEEP(dq[j][POOL ? 0 : v])
    }
}

struct B16Plan {
  int nu, spc, cpg, ngroups, nb_sub, nsub, local;
};

// how many blocks of the heaviest instantiation (4 units per thread) does the runtime place on one CU, with and without
// the 48-KB request buffer?  (bn_fused.hip's bf_occupancy_ok for this kernel.  Two by construction —
// __launch_bounds__(256, 2), 4.6 KB of static LDS + 48 KB —; the query guards against a runtime that disagrees: fewer
// than two without the buffer -> no persistent form; fewer than two with it -> the buffer is switched off.  Margin next
// to a co-tenant: two blocks hold 2 x 52.6 KB, i.e. they fit beside up to 54 KB of somebody else's LDS, and no block is
// larger than 160 / 3 KB (the kernel asserts it: 0.7 KB to spare since the fold's 32-row table and the sign-factor table
// of round 6), so a co-tenant that leaves cannot fragment the CU against the second block.)
static int b16_occupancy(bool with_buffer) {
  static int occ[2] = {-1, -1};
  int& o = occ[with_buffer ? 1 : 0];
  if (o < 0) {
    int ndev = 0, n = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
      (void)hipGetLastError();
      o = 2;  // (planning queries on a build box)
    } else if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, bf16_bn_bwd_fused_kernel<3, false, 4>, 256,
                                                            with_buffer ? 48 * 1024 : 0) != hipSuccess) {
      (void)hipGetLastError();
      o = 2;
    } else {
      o = n;
    }
  }
  return o;
}

static bool b16_plan(int B, int Cb, int HW, B16Plan* out) {
  const long long nq = (long long)B * HW / 4;
  for (int NU : {1, 2, 4}) {
    if (nq > 256LL * NU) continue;
    *out = B16Plan{NU, 1, 1, Cb, Cb, 1, 1};
    return true;
  }
  if (!bf_persistent_allowed() || b16_occupancy(false) < 2) return false;  // (CU mask / switched off: three launches)
  for (int nsub = 2; nsub >= 1; --nsub) {
    const int nb_sub = sivae_num_cus() * (nsub == 2 ? 1 : 2);
    const int NU = 4;
    const long long slabq = 256LL * NU;
    const long long spc = (nq + slabq - 1) / slabq;
    if (spc > nb_sub) continue;
    const int cpg = (int)(nb_sub / spc);
    *out = B16Plan{NU, (int)spc, cpg, (Cb + cpg - 1) / cpg, nb_sub, nsub, 0};
    return true;
  }
  return false;
}

static inline bool b16_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

#ifdef B16_TIMING
static unsigned long long* b16_timing_buffer = nullptr;
#endif

}  // namespace

#ifdef B16_TIMING
// timing build only (tools/b16_bn_timing.py): device buffer of blocks * groups * 8 stamps, or null; returns the launch plan
extern "C" void sivae_debug_b16_timing(unsigned long long* buf) { b16_timing_buffer = buf; }
extern "C" int sivae_debug_b16_plan(int B, int C, int H, int W, int seg_images, int* out6) {
  B16Plan p;
  if (!b16_plan(seg_images, (B / seg_images) * bf16_cblocks(C), H * W, &p)) return -1;
  out6[0] = p.nu; out6[1] = p.spc; out6[2] = p.cpg; out6[3] = p.ngroups; out6[4] = p.nb_sub; out6[5] = p.nsub;
  return 0;
}
#endif

// power-of-two maps from 2x2 up whose (segment, channel block) plane sets fit one group of the grid; B = nseg * seg_images
extern "C" int sivae_bf16_bn_bwd_fused_seg_supported(int B, int C, int H, int W, int seg_images) {
  if (B <= 0 || C <= 0 || !b16_pow2(H) || !b16_pow2(W) || H < 2 || W < 2) return 0;
  if (seg_images <= 0 || B % seg_images != 0) return 0;
  const int Cb = bf16_cblocks(C);
  if (Cb > BF_CH_COUNTERS) return 0;
  if ((long long)seg_images * Cb * H * W * 16 >= 0xfffffe00LL) return 0;  // (a plane set's window: 32-bit byte offsets)
  B16Plan p;
  return b16_plan(seg_images, (B / seg_images) * Cb, H * W, &p) ? 1 : 0;
}

extern "C" int sivae_bf16_bn_bwd_fused_supported(int B, int C, int H, int W) {
  return sivae_bf16_bn_bwd_fused_seg_supported(B, C, H, W, B);
}

// partial sums [VCb][spc][16] + per-(segment, channel block) sums [VCb][16], doubles
extern "C" size_t sivae_bf16_bn_bwd_fused_seg_workspace_bytes(int B, int C, int H, int W, int seg_images) {
  if (!sivae_bf16_bn_bwd_fused_seg_supported(B, C, H, W, seg_images)) return 0;
  const int VCb = (B / seg_images) * bf16_cblocks(C);
  B16Plan p;
  b16_plan(seg_images, VCb, H * W, &p);
  return ((size_t)VCb * p.spc * 16 + (size_t)VCb * 16) * sizeof(double) + 16;
}

extern "C" size_t sivae_bf16_bn_bwd_fused_workspace_bytes(int B, int C, int H, int W) {
  return sivae_bf16_bn_bwd_fused_seg_workspace_bytes(B, C, H, W, B);
}

// sivae_bf16_bn_bwd (bf16_bn.hip) as one launch; same arguments + `state` (the zero-initialised-once barrier state of
// sivae_bn_bwd_fused: sivae_bn_bwd_fused_state_uints() unsigned ints, one buffer per stream).  The persistent form needs
// its whole grid (2 blocks per CU) resident.
//
// SEGMENTED batch (sivae_bf16_bn_bwd_fused_seg): B = nseg * seg_images images, mean / invstd [nseg][C] (per-pass batch
// statistics), dgamma / dbeta [C] summed over the segments in segment order — the bf16 twin of sivae_bn_bwd_fused (bn_fused.hip).
extern "C" int sivae_bf16_bn_bwd_fused_seg(const void* dy, int dy_pooled, const void* y, const unsigned char* sign_mask,
                                           const void* x, const float* mean, const float* invstd, const float* gamma,
                                           const float* beta, float slope, void* dx, void* dz, int dz_sum, float* dgamma,
                                           float* dbeta, int B, int C, int H, int W, int seg_images, unsigned int* state,
                                           void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!dy || !x || !mean || !invstd || !gamma || !dx || !workspace || !state) return SIVAE_ERR_NULL;
  if (!y && !sign_mask && !beta) return SIVAE_ERR_NULL;
  if (dz_sum && !dz) return SIVAE_ERR_NULL;
  if (dy_pooled && dz_sum) return SIVAE_ERR_MODE;
  if (!sivae_bf16_bn_bwd_fused_seg_supported(B, C, H, W, seg_images)) return SIVAE_ERR_SHAPE;
  if (workspace_bytes < sivae_bf16_bn_bwd_fused_seg_workspace_bytes(B, C, H, W, seg_images)) return SIVAE_ERR_WORKSPACE;
  const int Cb = bf16_cblocks(C);
  const int nseg = B / seg_images, VCb = nseg * Cb;
  B16Plan p;
  b16_plan(seg_images, VCb, H * W, &p);
  Bf16BnFusedArgs a;
  a.dy = dy;
  a.y = y;
  a.x = x;
  a.mask = sign_mask;
  a.mean = mean;
  a.invstd = invstd;
  a.gamma = gamma;
  a.beta = beta;
  a.dx = dx;
  a.dz = dz;
  a.dgamma = dgamma;
  a.dbeta = dbeta;
  a.part = (double*)workspace;
  a.sums = a.part + (size_t)VCb * p.spc * 16;
  a.bar = state;
  a.inv_n = 1.0f / ((float)seg_images * H * W);
  a.slope = slope;
  a.C = C;
  a.Cb = Cb;
  a.H = H;
  a.W = W;
  a.B = seg_images;
  a.nseg = nseg;
  a.VCb = VCb;
  a.l2_qpp = ilog2_exact(H * W / 4);
  a.l2_qw = ilog2_exact(W / 2);
  a.spc = p.spc;
  a.cpg = p.cpg;
  a.ngroups = p.ngroups;
  a.nx = (!p.local && p.nb_sub % 8 == 0) ? 8 : 1;
  a.nsub = p.nsub;
  a.local = p.local;
  a.dzmode = !dz ? 0 : (dz_sum ? 2 : 1);
  a.spin_limit = bf_spin_limit();
#ifdef B16_TIMING
  a.ts = b16_timing_buffer;
#endif
  static int pf_on = -1;
  if (pf_on < 0) {
    const char* e = getenv("SIVAE_BN_FUSED_PREFETCH");
    pf_on = (e && e[0] == '0') ? 0 : 1;
  }
  a.pf = (pf_on && !p.local && p.nu == 4 && p.ngroups > p.nsub && b16_occupancy(true) >= 2) ? 1 : 0;
  const size_t lds = a.pf ? (size_t)3 * 4 * 256 * 16 : 0;  // 48 KB
  const int act = sign_mask ? 3 : (y ? 1 : 2);
  const dim3 grid((unsigned)(p.local ? VCb : p.nsub * p.nb_sub)), block(256);
#define B16_LAUNCH(A, P, N) hipLaunchKernelGGL((bf16_bn_bwd_fused_kernel<A, P, N>), grid, block, lds, stream, a)
#define B16_NU(A, P)                       \
  {                                        \
    if (p.nu == 4) B16_LAUNCH(A, P, 4);    \
    else if (p.nu == 2) B16_LAUNCH(A, P, 2); \
    else B16_LAUNCH(A, P, 1);              \
  }
#define B16_ACT(A)                               \
  {                                              \
    if (dy_pooled) B16_NU(A, true) else B16_NU(A, false) \
  }
  if (act == 3) B16_ACT(3) else if (act == 1) B16_ACT(1) else B16_ACT(2)
#undef B16_ACT
#undef B16_NU
#undef B16_LAUNCH
  return sivae_launch_status();
}

extern "C" int sivae_bf16_bn_bwd_fused(const void* dy, int dy_pooled, const void* y, const unsigned char* sign_mask,
                                       const void* x, const float* mean, const float* invstd, const float* gamma,
                                       const float* beta, float slope, void* dx, void* dz, int dz_sum, float* dgamma,
                                       float* dbeta, int B, int C, int H, int W, unsigned int* state, void* workspace,
                                       size_t workspace_bytes, hipStream_t stream) {
  return sivae_bf16_bn_bwd_fused_seg(dy, dy_pooled, y, sign_mask, x, mean, invstd, gamma, beta, slope, dx, dz, dz_sum,
                                     dgamma, dbeta, B, C, H, W, B, state, workspace, workspace_bytes, stream);
}
