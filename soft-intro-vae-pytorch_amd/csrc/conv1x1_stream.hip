// 1x1 convolution (ResidualBlock.conv_expand, soft_intro_vae/train_soft_intro_vae.py:50-54, forward and data gradient)
// as a streaming kernel on v_mfma_f32_32x32x2_f32:   y[b][co][p] = sum_ci W[co][ci] x[b][ci][p],  p = pixel index of a plane.
//
// A 1x1 conv has no halo, so nothing about x needs LDS: a lane loads FOUR consecutive pixels of one input channel with
// one 16-byte buffer load and the four components serve four MFMA column tiles — MFMA column n of tile e stands for
// pixel 4n + e (any pixel <-> column map is legal as long as the epilogue knows it).  A wave therefore covers 128
// consecutive pixels x 64 output channels (2 row tiles x 4 column tiles = 128 accumulators), and the accumulators of
// one output channel at columns n of the four tiles are four CONSECUTIVE pixels: the epilogue is 32 sixteen-byte stores
// per lane, no transpose.  The weights of the block's 64-channel group (the direct pack [ci][co_pad], 256 B per input
// channel) sit in LDS for the whole life of the persistent block; per k-step (2 input channels) a wave issues 1
// sixteen-byte x load (ring of two groups of 8 k-steps), 2 ds_read_b32 and 8 MFMAs.
// The direct kernel (conv_fwd.hip) pushes x through LDS in 16-channel chunks with two barriers each and stores dwords.
#include "common.h"

struct Conv1sArgs {
  const float* x;   // [B][Ci][HW]
  const float* wp;  // direct pack [Ci_pad][Co_pad] (sivae_pack_conv_weight, ks = 1)
  float* y;         // [B][Co][HW]
  int B, Ci, Co, HW, Co_pad;
  int n_px_tiles;   // (32 * ET)-pixel tiles per image
  int n_co_groups;  // (32 * MT)-channel output groups
  int n_pt_items;   // B * n_px_tiles / 4 rounded up: a block takes 4 pixel tiles (one per wave) at a time
  int accumulate;
};

// k-steps per load group (a ring of two groups).  (Measured, round 5: 16 for the 32-channel tile is 3-6 % slower than 8.)
#define C1S_G 8
#define C1S_PIN(V) asm volatile("" ::"v"(V));  // (SIVAE_PIN4 of common.h for a 2- or 4-float vector)

// MT row tiles (32 output channels each) x ET column tiles (a lane loads ET consecutive pixels) per wave: <2, 4> is the
// streaming form described above; <1, 4> halves the wave's tile for launches that would otherwise leave CUs without a
// block, and for more than 256 input channels (its weight slab is 128 B per input channel).  These layers are bound by
// the matrix pipe, not by bytes (120 TF/s at batch 128): at the 8-image shard 256 -> 512 @ 32x32 takes 27.8 us with the
// 32-channel tile against 42.7 us.  A 32 x 64-pixel tile (ET = 2, 8-byte loads) was built and measured too: one LDS
// read per two MFMAs leaves the wave latency-bound (55 us for the same layer) — not instantiated.
template <int MT, int ET>
__global__ void __launch_bounds__(256, 2) conv1x1_stream_kernel(Conv1sArgs a) {
  constexpr int CG = 32 * MT;   // output channels of a block
  constexpr int PXT = 32 * ET;  // pixels of a wave's tile
  extern __shared__ __attribute__((aligned(16))) float wsm[];  // [Ci][CG]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, kk = lane >> 5;
  const int HW = a.HW, Ci = a.Ci;
  // The co groups of one pixel range read the same x: keep them on ONE XCD so that the second group finds x in that L2
  // (blocks go round-robin to the 8 XCDs by linear index: consecutive blockIdx.x are on DIFFERENT XCDs — round 4: the
  // sibling groups each fetched x from HBM, 1.85 GB per launch against ~1.4 of tensors).  lb enumerates the blocks of an
  // XCD consecutively when the grid is a multiple of 8 * groups.
  const int nblk = (int)gridDim.x;
  const bool remap = a.n_co_groups > 1 && (nblk % (8 * a.n_co_groups)) == 0;
  const int lb = remap ? ((int)blockIdx.x & 7) * (nblk >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  const int cog = lb % a.n_co_groups;
  const int co0 = cog * CG;

  // (co_pad is a multiple of 128: in range; padded channels are 0; 16-byte loads: the host checks the operand's alignment)
  for (int i = tid; i < Ci * (CG / 4); i += 256) {
    const int k = i / (CG / 4), c4 = i % (CG / 4);
    reinterpret_cast<float4*>(wsm)[i] = *reinterpret_cast<const float4*>(a.wp + (size_t)k * a.Co_pad + co0 + 4 * c4);
  }
  __syncthreads();

  const int nks = Ci >> 1;  // k-steps (Ci is even)
  const unsigned xlane = (unsigned)(kk * HW + ET * l31) * 4u;  // channel kk, pixels ET*l31 .. +ET-1 of the tile
  const unsigned xkstep = (unsigned)HW * 8u;                   // two channels
  const float* wl = wsm + kk * CG + l31;

  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
  typedef float xvec_t __attribute__((ext_vector_type(ET)));
  for (int it = lb / a.n_co_groups; it < a.n_pt_items; it += nblk / a.n_co_groups) {
    const int t = it * 4 + wave;  // this wave's pixel tile
    const int b = t / a.n_px_tiles, pt = t - b * a.n_px_tiles;
    const bool live = b < a.B;
    const int p0 = pt * PXT;
    // pixels past the end of the plane are masked through the buffer range (HW % 4 == 0: whole vectors in or out)
    const __amdgpu_buffer_rsrc_t xrs =
        make_rsrc(a.x + (size_t)(live ? b : 0) * Ci * HW, live ? (unsigned long long)Ci * HW * 4ull : 0ull);
    const unsigned pin = (p0 + ET * l31 < HW) ? (unsigned)p0 * 4u + xlane : SIVAE_OOB16;  // (8- / 16-byte loads)

    f32x16 acc[MT][ET];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int e = 0; e < ET; ++e)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][e][r] = 0.f;

    // (hipcc / ROCm 7.2: bit-cast the WHOLE loaded vector — a per-element cast of the u32x4 result narrows the load to one
    // dword and replicates it: common.h, buf_load_f32x4)
    xvec_t xa[C1S_G], xb[C1S_G];
#define C1S_LOAD(BUF, S0)                                                                          \
  _Pragma("unroll") for (int g = 0; g < C1S_G; ++g) {                                              \
    const int s_ = (S0) + g;                                                                       \
    if constexpr (ET == 4)                                                                         \
      BUF[g] = __builtin_bit_cast(xvec_t, __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)(s_ < nks ? pin : SIVAE_OOB16), (int)((unsigned)s_ * xkstep), 0)); \
    else                                                                                           \
      BUF[g] = __builtin_bit_cast(xvec_t, __builtin_amdgcn_raw_buffer_load_b64(xrs, (int)(s_ < nks ? pin : SIVAE_OOB16), (int)((unsigned)s_ * xkstep), 0)); \
  }
#define C1S_MMA(BUF, S0)                                                                           \
  _Pragma("unroll") for (int g = 0; g < C1S_G; ++g) {                                              \
    const int s_ = (S0) + g;                                                                       \
    if (s_ < nks) {                                                                                \
      float am[MT];                                                                                \
      _Pragma("unroll") for (int m = 0; m < MT; ++m) am[m] = wl[s_ * 2 * CG + 32 * m];             \
      _Pragma("unroll") for (int e = 0; e < ET; ++e) {                                             \
        const float bv = BUF[g][e];                                                                \
        _Pragma("unroll") for (int m = 0; m < MT; ++m)                                             \
          acc[m][e] = __builtin_amdgcn_mfma_f32_32x32x2f32(am[m], bv, acc[m][e], 0, 0, 0);         \
      }                                                                                            \
    }                                                                                              \
  }
    C1S_LOAD(xa, 0)
    for (int s0 = 0; s0 < nks; s0 += 2 * C1S_G) {
      C1S_LOAD(xb, s0 + C1S_G)
      C1S_MMA(xa, s0)
      C1S_LOAD(xa, s0 + 2 * C1S_G)
      C1S_MMA(xb, s0 + C1S_G)
    }
#undef C1S_LOAD
#undef C1S_MMA

    // ---- epilogue: acc[m][e][r] = channel co0 + m*32 + (r&3) + 8*(r>>2) + 4*kk of pixel p0 + ET*l31 + e
    const __amdgpu_buffer_rsrc_t yrs =
        make_rsrc(a.y + (size_t)(live ? b : 0) * a.Co * HW, live ? (unsigned long long)a.Co * HW * 4ull : 0ull);
    const unsigned pout = (p0 + ET * l31 < HW) ? (unsigned)(p0 + ET * l31) * 4u : SIVAE_OOB;
    xvec_t v_prev = 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
        // (16-byte accesses: SIVAE_OOB16 — at 0xFFFFFFFF the upper three dwords of a store would wrap into the window)
        const unsigned off = (co < a.Co && pout != SIVAE_OOB) ? pout + (unsigned)co * (unsigned)HW * 4u : SIVAE_OOB16;
        xvec_t v;
#pragma unroll
        for (int e = 0; e < ET; ++e) v[e] = acc[m][e][r];
        if constexpr (ET == 4) {
          if (a.accumulate) {
            const u32x4_t o = __builtin_amdgcn_raw_buffer_load_b128(yrs, (int)off, 0, 0);
            v += __builtin_bit_cast(xvec_t, o);
          }
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), yrs, (int)off, 0, 0);
        } else {
          if (a.accumulate) {
            const u32x2_t o = __builtin_amdgcn_raw_buffer_load_b64(yrs, (int)off, 0, 0);
            v += __builtin_bit_cast(xvec_t, o);
          }
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, v), yrs, (int)off, 0, 0);
        }
        C1S_PIN(v_prev)  // (store-data lifetime: common.h)
        v_prev = v;
      }
    C1S_PIN(v_prev)
  }
}

// shapes the streaming kernel takes: whole 16-byte pixel vectors, an even number of input channels whose weight tile
// (256 B per input channel for 64 output channels, 128 B for 32) leaves room for two blocks per CU: up to 256 input
// channels with the 64-channel tile, up to 512 with the 32-channel one
extern "C" int sivae_conv1x1_stream_supported(int B, int Ci, int Co, int HW) {
  if (B <= 0 || Ci <= 0 || Co <= 0 || HW <= 0) return 0;
  if ((HW & 3) || (Ci & 1) || Ci > 512) return 0;
  if ((long long)Ci * HW * 4 >= 0x7fffffffLL || (long long)Co * HW * 4 >= 0x7fffffffLL) return 0;
  return 1;
}

extern "C" int sivae_conv1x1_stream(const float* x, const float* wp, float* y, int B, int Ci, int Co, int HW,
                                    int accumulate, hipStream_t stream) {
  if (!x || !wp || !y) return SIVAE_ERR_NULL;
  if (!sivae_conv1x1_stream_supported(B, Ci, Co, HW)) return SIVAE_ERR_SHAPE;
  if (((uintptr_t)wp & 15u) != 0) return SIVAE_ERR_SHAPE;  // (the weight slab is staged with 16-byte loads)
  Conv1sArgs a;
  a.x = x;
  a.wp = wp;
  a.y = y;
  a.B = B;
  a.Ci = Ci;
  a.Co = Co;
  a.HW = HW;
  a.Co_pad = ((Co + 127) / 128) * 128;
  a.accumulate = accumulate;
  const int cus = sivae_num_cus();
  // the wave tile: 64 channels x 128 pixels when that leaves every CU two blocks, otherwise 32 x 128
  int MT = 2;
  const int ET = 4;
  auto items_of = [&](int mt, int et) { return (((long long)B * cdiv(HW, 32 * et) + 3) / 4) * cdiv(Co, 32 * mt); };
  static int small = -1;
  if (small < 0) {
    const char* e = getenv("SIVAE_CONV1X1_SMALL_TILES");
    small = (e && e[0] == '0') ? 0 : 1;
  }
  if (Ci > 256 || (small && items_of(2, 4) < 2LL * cus)) MT = 1;
  a.n_px_tiles = cdiv(HW, 32 * ET);
  a.n_co_groups = cdiv(Co, 32 * MT);
  const long long tiles = (long long)B * a.n_px_tiles;
  a.n_pt_items = (int)((tiles + 3) / 4);
  // grid: a multiple of the co-group count (block -> co group = blockIdx % groups), about two blocks per CU
  long long per_group = (2LL * cus) / a.n_co_groups;
  if (per_group < 1) per_group = 1;
  if (per_group > a.n_pt_items) per_group = a.n_pt_items;
  const long long grid = per_group * a.n_co_groups;
  const size_t lds = (size_t)Ci * 32 * MT * sizeof(float);
  void (*kern)(Conv1sArgs) = MT == 2 ? conv1x1_stream_kernel<2, 4> : conv1x1_stream_kernel<1, 4>;
  static size_t lds_hwm[2] = {0, 0};
  const int rc_lds = sivae_ensure_lds(reinterpret_cast<const void*>(kern), lds, &lds_hwm[MT == 2 ? 0 : 1]);
  if (rc_lds != SIVAE_OK) return rc_lds;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, stream, a);
  return sivae_launch_status();
}
