// Data gradient of "3x3 conv of a nearest-2x-upsampled input" with respect to the LOW-resolution input
// (adjoint of conv_wino_up.hip; replaces F(2x2,3x3) data gradient at full resolution + the 2x2 block sum of
// nn.Upsample's backward).  With the phase form  y_pq[i][j] = sum_ab g_pq[a][b] x[i-1+p+a][j-1+q+b]:
//
//     dx[u][v] = sum_{p,q} sum_{a',b'} gf_pq[a'][b'] * dy_pq[u - p + a'][v - q + b'],   gf_pq[a'][b'] = g_pq[1-a'][1-b'],
//
// i.e. ONE 2x2 convolution over 4*C input planes — the four parity planes dy_pq[i][j] = dy[2i+p][2j+q] of every
// dy channel, read in place with stride-2 addressing (the parity goes into the scalar offset of the buffer loads) —
// where plane (p, q) is read with a shift of (1-p, 1-q).  It runs as Winograd F(2x2,2x2) with the phases folded into the K dimension: 9 multiplies per
// (2x2 low-res outputs, channel pair, phase) = 36 per 4x4 block of dy pixels instead of 64.
//
// Work split: all 9 frequencies of an output in one wave (144 accumulator registers, output transform in
// registers, contiguous float2 stores); a block is 4 waves = 4 groups of 32 output channels x 32 tiles sharing the
// LDS halo of the current 16-plane chunk.  Chunks never straddle phases (K index = phase * Cpad + c), so the patch
// shift is uniform per chunk.  K loop, staging and persistent work items as in conv_wino.hip.
#include "common.h"
#include "pack_batch.h"
#include <stdlib.h>

struct WinoUpDgArgs {
  const float* dyp;  // dy [B][C][2Hs][2Ws], read as its four parity planes dy[2i+p][2j+q]
  const float* ud;   // packed [4*Cpad][Npad][12]
  float* dx;         // [B][N][Hs][Ws]
  int B, C, N, Hs, Ws;
  int Cpad, Npad;
  int nbh, nbw;
  int n_n_tiles;
  int n_items;
  int accumulate;
  // split-K (small shards: 512 -> 512 @16x16 with 16 images is 128 work items each walking 4 x 512 planes): item =
  // (slice, base item); slice s accumulates the chunk sequence [s * q_per_split, ...) and writes its partial dx to
  // dx + s * dx_split_stride (a workspace); wud_splitk_reduce_kernel sums the slices in a fixed order
  int n_items_base;
  int q_per_split;
  long long dx_split_stride;
  int xcd_group;
};

#define WUD_CK 16
#define WUD_TN 128

// KS2 = false: 4 waves = 4 groups of 32 output channels (128 per block), each wave runs all 8 k-steps of a chunk.
// KS2 = true (N <= 64): 2 groups of 32 output channels x 2 K-halves — waves 2,3 take k-steps 4..7 of every chunk —
// so that all four waves work when the layer has only 64 input channels; the two partial outputs (the output
// transform is linear, so already transformed: 64 floats per lane) are summed through LDS at the end.
template <int TTH_L2, int TTW_L2, bool KS2>
__global__ void __launch_bounds__(256, 2) conv_wino_up_dgrad_kernel(WinoUpDgArgs a) {
  constexpr int TTH = 1 << TTH_L2, TTW = 1 << TTW_L2;
  static_assert(TTH * TTW == 32, "a block is 32 tiles");
  constexpr int PXH = 2 * TTH, PXW = 2 * TTW;
  constexpr int LH = PXH + 2, LWU = PXW + 2;
  constexpr int PH = TTW + TTW / 4, RS = 2 * PH, PLANE = LH * RS;
  constexpr int NPOS = LH * LWU;
  static_assert(NPOS <= 256, "one halo position per thread");
  constexpr int CK = WUD_CK;
  constexpr int XBUF = CK * PLANE;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;  // [2][CK][PLANE]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int Hs = a.Hs, Ws = a.Ws, HWs = Hs * Ws;
  const int nch = a.Cpad / CK;  // chunks per phase
  const int nchunks = 4 * nch;
  constexpr int KPC = KS2 ? CK / 4 : CK / 2;  // k-steps per chunk and wave
  const int nksteps = nchunks * KPC;
  const int ng = KS2 ? (wave & 1) : wave;     // output-channel group of this wave
  const int kh = KS2 ? (wave >> 1) : 0;       // K half
  const int koff = kh * (CK / 2);             // first channel of this wave inside a chunk

  const int n_items = a.n_items;
  // XCD-aware start items (as conv_wino.hip): the blocks of XCD x walk the contiguous items x * grid/8 + slot (+ k * grid),
  // so vertically / horizontally adjacent tile blocks — which share halo rows of dy — meet in one L2
  int item = blockIdx.x;
  if (a.xcd_group) item = ((int)blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3);
  int b, r0, c0, n0;
  int q0, q1, sl;  // chunk-sequence range and output slice of the current item
  __amdgpu_buffer_rsrc_t xrsrc;
  const __amdgpu_buffer_rsrc_t ursrc = make_rsrc(a.ud, 4ull * a.Cpad * a.Npad * 48ull);
  unsigned xo, ua_base;
  static_assert(2 * NPOS >= 256, "duplicate-owner mapping");
  const int teff = tid < NPOS ? tid : tid - NPOS;  // (threads beyond the halo duplicate the first slots: no predication)
  const int xrr = teff / LWU, xcc = teff % LWU;
  const int xl = xrr * RS + (xcc & 1) * PH + (xcc >> 1);
#define WUD_SETUP(ITEM)                                                  \
  {                                                                      \
    sl = (ITEM) / a.n_items_base;                                        \
    const int it_ = (ITEM)-sl * a.n_items_base;                          \
    q0 = sl * a.q_per_split;                                             \
    q1 = q0 + a.q_per_split < nchunks ? q0 + a.q_per_split : nchunks;    \
    const int n_tile = it_ % a.n_n_tiles;                                \
    const int pt = it_ / a.n_n_tiles;                                    \
    const int tbx = pt % a.nbw;                                          \
    const int t2 = pt / a.nbw;                                           \
    const int tby = t2 % a.nbh;                                          \
    b = t2 / a.nbh;                                                      \
    r0 = tby * PXH;                                                      \
    c0 = tbx * PXW;                                                      \
    n0 = n_tile * (KS2 ? WUD_TN / 2 : WUD_TN);                           \
    xrsrc = make_rsrc(a.dyp + (size_t)b * 4 * a.C * HWs, 4ull * a.C * HWs * 4ull); \
    const int r = r0 + xrr - 1, c = c0 + xcc - 1;                        \
    xo = SIVAE_OOB;                                                      \
    /* parity plane (p, q), position (r, c) = dy[2r + p][2c + q]: the (p, q) part goes into the scalar offset */ \
    if (r >= 0 && r < Hs && c >= 0 && c < Ws) xo = (unsigned)(2 * r * (2 * Ws) + 2 * c) * 4u; \
    ua_base = (unsigned)(n0 + ng * 32) * 48u + (unsigned)koff * ua_step; \
  }

  const unsigned va0 = (unsigned)(hh * a.Npad + l31) * 48u;
  const unsigned ua_step = (unsigned)a.Npad * 48u;  // bytes per K index

  const int tx = l31 & (TTW - 1), ty = l31 >> TTW_L2;
  const int bb0 = (hh + koff) * PLANE + 2 * ty * RS + tx;
  int bc0, bc1, bc2;  // per-chunk patch bases (phase shift folded in)

  f32x16 acc[9];
  float xr[CK];
  float4 AR[4][3];

  // Chunk sequence q -> (phase = q & 3, first channel = (q >> 2) * CK): the four parity planes of the SAME 16 channels
  // are consecutive chunks, so the 128-byte lines of a dy row — shared by its two column parities (q = 0, 1) — are
  // fetched once and hit in L2 the second time (phase-major order re-fetched them a whole channel sweep later:
  // 12.5 GB of fabric traffic per launch on 64 -> 64 @256x256 against 2.7 GB of compulsory bytes, at 5.7 TB/s).
  // The packed filter's K index stays phase * Cpad + channel.
#define WUD_LOAD_X(Q)                                                    \
  {                                                                      \
    const int ph_ = (Q)&3, cc0_ = ((Q) >> 2) * CK;                           \
    _Pragma("unroll") for (int ck = 0; ck < CK; ++ck) {                  \
      const int c = cc0_ + ck;                                           \
      const int cclamp = c < a.C ? c : a.C - 1;                          \
      xr[ck] = buf_load_f32(xrsrc, xo, (unsigned)cclamp * (unsigned)HWs * 16u + (unsigned)((ph_ >> 1) * 2 * Ws + (ph_ & 1)) * 4u); \
    }                                                                    \
  }
  // this wave's k-step KS (counted over its own k-steps) -> K index (KS / KPC) * CK + koff + 2 * (KS % KPC)
#define WUD_LOAD_A(KS_ABS, SLOT)                                         \
  {                                                                      \
    const int q_ = (KS_ABS) / KPC; /* chunk sequence index -> K index (phase * Cpad + channel) */ \
    const unsigned so = ua_base + (unsigned)((q_ & 3) * a.Cpad + (q_ >> 2) * CK + 2 * ((KS_ABS) % KPC)) * ua_step; \
    AR[SLOT][0] = buf_load_f32x4(ursrc, va0, so);                        \
    AR[SLOT][1] = buf_load_f32x4(ursrc, va0 + 16u, so);                  \
    AR[SLOT][2] = buf_load_f32x4(ursrc, va0 + 32u, so);                  \
  }
#define WUD_STORE_X(Q, BUF)                                              \
  {                                                                      \
    const int ph_ = (Q)&3, cc0_ = ((Q) >> 2) * CK;                           \
    _Pragma("unroll") for (int ck = 0; ck < CK; ++ck) {                  \
      const float v = (cc0_ + ck) < a.C ? xr[ck] : 0.f;                  \
      xs[(BUF)*XBUF + ck * PLANE + xl] = v;                              \
    }                                                                    \
  }
  // plane (p, q) is read at rows (2*ty + (1-p) + r), columns (2*tx + (1-q) + c)
#define WUD_BASES(Q)                                                     \
  {                                                                      \
    const int ph_ = (Q)&3;                                               \
    const int ro = 1 - (ph_ >> 1), co = 1 - (ph_ & 1);                   \
    const int base = bb0 + ro * RS;                                      \
    bc0 = base + ((co + 0) & 1) * PH + ((co + 0) >> 1);                  \
    bc1 = base + ((co + 1) & 1) * PH + ((co + 1) >> 1);                  \
    bc2 = base + ((co + 2) & 1) * PH + ((co + 2) >> 1);                  \
  }
#define WUD_READ(BUF, KK, D)                                             \
  {                                                                      \
    const float* pb_ = xs + (BUF)*XBUF + 2 * (KK)*PLANE;                 \
    _Pragma("unroll") for (int r = 0; r < 3; ++r) {                      \
      D[r][0] = pb_[r * RS + bc0];                                       \
      D[r][1] = pb_[r * RS + bc1];                                       \
      D[r][2] = pb_[r * RS + bc2];                                       \
    }                                                                    \
  }
#define WUD_STEP(SLOT, D)                                                \
  {                                                                      \
    float t[3][3];                                                       \
    _Pragma("unroll") for (int r = 0; r < 3; ++r) {                      \
      t[r][0] = D[r][0] - D[r][1];                                       \
      t[r][1] = D[r][1];                                                 \
      t[r][2] = D[r][1] - D[r][2];                                       \
    }                                                                    \
    _Pragma("unroll") for (int j = 0; j < 3; ++j) {                      \
      const float v0 = t[0][j] - t[1][j], v1 = t[1][j], v2 = t[1][j] - t[2][j]; \
      const float u0 = (j == 0) ? AR[SLOT][0].x : ((j == 1) ? AR[SLOT][0].y : AR[SLOT][0].z); \
      const float u1 = (j == 0) ? AR[SLOT][0].w : ((j == 1) ? AR[SLOT][1].x : AR[SLOT][1].y); \
      const float u2 = (j == 0) ? AR[SLOT][1].z : ((j == 1) ? AR[SLOT][1].w : AR[SLOT][2].x); \
      acc[0 * 3 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(u0, v0, acc[0 * 3 + j], 0, 0, 0); \
      acc[1 * 3 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(u1, v1, acc[1 * 3 + j], 0, 0, 0); \
      acc[2 * 3 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(u2, v2, acc[2 * 3 + j], 0, 0, 0); \
    }                                                                    \
  }
#define WUD_KSTEP(Q, BUF, KK, D, DN)                                     \
  {                                                                      \
    if ((KK) + 1 < KPC) WUD_READ(BUF, (KK) + 1, DN)                      \
    __builtin_amdgcn_sched_barrier(0);                                   \
    WUD_STEP((KK)&3, D)                                                  \
    __builtin_amdgcn_sched_barrier(0);                                   \
    if ((Q) * KPC + (KK) + 4 < q1 * KPC) WUD_LOAD_A((Q) * KPC + (KK) + 4, (KK)&3) \
  }
#define WUD_MMA(Q, BUF, NEXT)                                            \
  {                                                                      \
    float d0[3][3], d1[3][3];                                            \
    if (NEXT) WUD_LOAD_X((Q) + 1)                                        \
    WUD_BASES(Q)                                                         \
    WUD_READ(BUF, 0, d0)                                                 \
    WUD_KSTEP(Q, BUF, 0, d0, d1)                                         \
    WUD_KSTEP(Q, BUF, 1, d1, d0)                                         \
    WUD_KSTEP(Q, BUF, 2, d0, d1)                                         \
    if (!KS2) {                                                          \
      WUD_KSTEP(Q, BUF, 3, d1, d0)                                       \
      WUD_KSTEP(Q, BUF, 4, d0, d1)                                       \
      WUD_KSTEP(Q, BUF, 5, d1, d0)                                       \
      WUD_KSTEP(Q, BUF, 6, d0, d1)                                       \
    }                                                                    \
    if (NEXT) WUD_STORE_X((Q) + 1, (BUF) ^ 1)                            \
    __builtin_amdgcn_sched_barrier(0);                                   \
    if (KS2) WUD_KSTEP(Q, BUF, 3, d1, d0) else WUD_KSTEP(Q, BUF, 7, d1, d0) \
    __syncthreads();                                                     \
  }
#define WUD_PREFETCH(ITEM)                                               \
  {                                                                      \
    WUD_SETUP(ITEM)                                                      \
    WUD_LOAD_X(q0)                                                       \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) WUD_LOAD_A(q0 * KPC + kk, kk) \
  }

  WUD_PREFETCH(item)
  for (;;) {
#pragma unroll
    for (int f = 0; f < 9; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
    WUD_STORE_X(q0, 0)
    __syncthreads();
    int q = q0;
    for (; q + 1 < q1; q += 2) {
      WUD_MMA(q, 0, true)
      const bool more = q + 2 < q1;
      WUD_MMA(q + 1, 1, more)
    }
    if (q < q1) WUD_MMA(q, 0, false)

    __builtin_amdgcn_s_setprio(1);  // serial tail at raised priority (see conv_wino.hip)
    const int e_b = b, e_r0 = r0, e_c0 = c0, e_n0 = n0, e_sl = sl;
    const int next = item + (int)gridDim.x;
    const bool has_next = next < n_items;
    const bool early = has_next && !a.accumulate;  // (accumulate loads dx: keep the prefetch behind those loads)
    if (early) WUD_PREFETCH(next)
    {
      const __amdgpu_buffer_rsrc_t yrsrc =
          make_rsrc(a.dx + (size_t)e_sl * a.dx_split_stride + (size_t)e_b * a.N * HWs, (unsigned long long)a.N * HWs * 4ull);
      const int li = e_r0 + 2 * ty, lj = e_c0 + 2 * tx;
      const bool okc = lj < Ws;  // Ws even: lj + 1 < Ws too
      const unsigned base = (unsigned)(li * Ws + lj) * 4u;
      float* red = smem;  // KS2: [2 groups][64 values][64 lanes] partial outputs of the upper K half (32 KB)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float s00 = acc[0][r] + acc[3][r], s01 = acc[1][r] + acc[4][r], s02 = acc[2][r] + acc[5][r];
        const float s10 = acc[3][r] - acc[6][r], s11 = acc[4][r] - acc[7][r], s12 = acc[5][r] - acc[8][r];
        // (reuse the first four accumulators of row r as the transformed outputs)
        acc[0][r] = s00 + s01;
        acc[1][r] = s01 - s02;
        acc[2][r] = s10 + s11;
        acc[3][r] = s11 - s12;
      }
      if (KS2) {
        // the K loop ended on a barrier: the halo buffers are free
        if (kh == 1) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int k = 0; k < 4; ++k) red[((ng * 64) + r * 4 + k) * 64 + lane] = acc[k][r];
        }
        __syncthreads();
        if (kh == 0) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k][r] += red[((ng * 64) + r * 4 + k) * 64 + lane];
        }
      }
      if (!KS2 || kh == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int chn = e_n0 + ng * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          float y00 = acc[0][r], y01 = acc[1][r], y10 = acc[2][r], y11 = acc[3][r];
          const bool cok = chn < a.N && okc;
          const unsigned cb = base + (unsigned)chn * (unsigned)HWs * 4u;
          const unsigned o0 = (cok && li < Hs) ? cb : SIVAE_OOB;
          const unsigned o1 = (cok && li + 1 < Hs) ? cb + (unsigned)Ws * 4u : SIVAE_OOB;
          if (a.accumulate) {
            const float2 p0 = buf_load_f32x2(yrsrc, o0, 0u), p1 = buf_load_f32x2(yrsrc, o1, 0u);
            y00 += p0.x;
            y01 += p0.y;
            y10 += p1.x;
            y11 += p1.y;
          }
          buf_store_f32x2(yrsrc, y00, y01, o0, 0u);
          buf_store_f32x2(yrsrc, y10, y11, o1, 0u);
        }
      }
      if (KS2) __syncthreads();  // the reduction area aliases the halo buffers of the next item
    }
    if (has_next && !early) WUD_PREFETCH(next)
    __builtin_amdgcn_s_setprio(0);
    if (!has_next) break;
    item = next;
  }
#undef WUD_SETUP
#undef WUD_LOAD_X
#undef WUD_LOAD_A
#undef WUD_STORE_X
#undef WUD_BASES
#undef WUD_READ
#undef WUD_STEP
#undef WUD_KSTEP
#undef WUD_MMA
#undef WUD_PREFETCH
}

// ---- dy [B][C][2Hs][2Ws] -> parity planes dyp [B][4][C][Hs][Ws], dyp[b][2p+q][c][i][j] = dy[b][c][2i+p][2j+q]
__global__ void __launch_bounds__(256) space_to_depth2_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                              int C, int Hs, int Ws, size_t n_pairs) {
  // one thread per (b, c, i, j2): 2 rows x 4 columns in, 4 planes x 2 columns out (Ws even)
  const size_t stride = (size_t)gridDim.x * 256;
  const int W2 = Ws >> 1, W = 2 * Ws;
  for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < n_pairs; o += stride) {
    const int j2 = (int)(o % W2);
    size_t t = o / W2;
    const int i = (int)(t % Hs);
    t /= Hs;  // b*C + c
    const int c = (int)(t % C);
    const size_t bb = t / C;
    const float* src = in + ((t * 2 * Hs + 2 * i) * (size_t)W) + 4 * j2;
    const float4 r0 = *reinterpret_cast<const float4*>(src);
    const float4 r1 = *reinterpret_cast<const float4*>(src + W);
    const size_t plane = (size_t)Hs * Ws;
    float* dst = out + ((bb * 4 * C + c) * Hs + i) * (size_t)Ws + 2 * j2;
    *reinterpret_cast<float2*>(dst) = make_float2(r0.x, r0.z);                              // p=0, q=0
    *reinterpret_cast<float2*>(dst + (size_t)C * plane) = make_float2(r0.y, r0.w);          // p=0, q=1
    *reinterpret_cast<float2*>(dst + (size_t)2 * C * plane) = make_float2(r1.x, r1.z);      // p=1, q=0
    *reinterpret_cast<float2*>(dst + (size_t)3 * C * plane) = make_float2(r1.y, r1.w);      // p=1, q=1
  }
}

extern "C" int sivae_space_to_depth2(const float* in, float* out, int B, int C, int Hs, int Ws, hipStream_t stream) {
  if (!in || !out) return SIVAE_ERR_NULL;
  if (B <= 0 || C <= 0 || Hs <= 0 || Ws <= 0 || (Ws & 1)) return SIVAE_ERR_SHAPE;
  const size_t n_pairs = (size_t)B * C * Hs * (Ws >> 1);
  int nb = cdiv((long long)n_pairs, 256 * 2);
  if (nb > 8192) nb = 8192;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(space_to_depth2_kernel, dim3(nb), dim3(256), 0, stream, in, out, C, Hs, Ws, n_pairs);
  return sivae_launch_status();
}

// ---- filter transform for the data gradient: K index = phase * Cpad + k (k = w's OUTPUT channel = dy channel),
// N index = w's input channel; U = G gf G^T with gf = 180-degree flip of the phase filter g_pq
__device__ __forceinline__ void pack_wino_up_dgrad_body(const float* __restrict__ w, float* __restrict__ ud,
                                                                 int Co, int Ci, int cpad, int npad, size_t idx0_, const size_t stride_) {
  const size_t total = (size_t)cpad * npad;
  for (size_t idx = idx0_; idx < total; idx += stride_) {
    const int n = (int)(idx % npad), k = (int)(idx / npad);
    float g[3][3];
    const bool ok = k < Co && n < Ci;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) g[r][c] = ok ? w[((size_t)k * Ci + n) * 9 + r * 3 + c] : 0.f;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float rw[2][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          rw[0][c] = p == 0 ? g[0][c] : g[0][c] + g[1][c];
          rw[1][c] = p == 0 ? g[1][c] + g[2][c] : g[2][c];
        }
        float gp[2][2];
#pragma unroll
        for (int a_ = 0; a_ < 2; ++a_) {
          gp[a_][0] = q == 0 ? rw[a_][0] : rw[a_][0] + rw[a_][1];
          gp[a_][1] = q == 0 ? rw[a_][1] + rw[a_][2] : rw[a_][2];
        }
        float gf[2][2] = {{gp[1][1], gp[1][0]}, {gp[0][1], gp[0][0]}};
        float gr[3][2];
#pragma unroll
        for (int b_ = 0; b_ < 2; ++b_) {
          gr[0][b_] = gf[0][b_];
          gr[1][b_] = gf[0][b_] + gf[1][b_];
          gr[2][b_] = gf[1][b_];
        }
        float u[9];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          u[i * 3 + 0] = gr[i][0];
          u[i * 3 + 1] = gr[i][0] + gr[i][1];
          u[i * 3 + 2] = gr[i][1];
        }
        float4* dst = reinterpret_cast<float4*>(ud + (((size_t)(p * 2 + q) * cpad + k) * npad + n) * 12);
        dst[0] = make_float4(u[0], u[1], u[2], u[3]);
        dst[1] = make_float4(u[4], u[5], u[6], u[7]);
        dst[2] = make_float4(u[8], 0.f, 0.f, 0.f);
      }
  }
}

__global__ void __launch_bounds__(256) pack_wino_up_dgrad_kernel(const float* __restrict__ w, float* __restrict__ ud,
                                                                 int Co, int Ci, int cpad, int npad) {
  pack_wino_up_dgrad_body(w, ud, Co, Ci, cpad, npad, (size_t)blockIdx.x * 256 + threadIdx.x, (size_t)gridDim.x * 256);
}

__global__ void __launch_bounds__(256) pack_wino_up_dgrad_batch_kernel(const SivaePackJob* __restrict__ jobs,
                                                                        const unsigned short* __restrict__ block_job) {
  const SivaePackJob j = jobs[block_job[blockIdx.x]];
  pack_wino_up_dgrad_body(j.w, j.dst, j.Co, j.Ci, j.kpad, j.npad, (size_t)(blockIdx.x - j.blk0) * 256 + threadIdx.x, (size_t)j.nblk * 256);
}


static inline int wud_cpad(int c) { return ((c + WUD_CK - 1) / WUD_CK) * WUD_CK; }
static inline int wud_npad(int n) { return ((n + WUD_TN - 1) / WUD_TN) * WUD_TN; }

extern "C" size_t sivae_pack_wino_up_dgrad_weight_bytes(int Co, int Ci) {
  if (Co <= 0 || Ci <= 0) return 0;
  return (size_t)4 * wud_cpad(Co) * wud_npad(Ci) * 12 * sizeof(float);
}

extern "C" int sivae_pack_wino_up_dgrad_weight(const float* w, float* ud, int Co, int Ci, hipStream_t stream) {
  if (!w || !ud) return SIVAE_ERR_NULL;
  if (Co <= 0 || Ci <= 0) return SIVAE_ERR_SHAPE;
  const int cpad = wud_cpad(Co), npad = wud_npad(Ci);
  int nb = cdiv((long long)cpad * npad, 256);
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(pack_wino_up_dgrad_kernel, dim3(nb), dim3(256), 0, stream, w, ud, Co, Ci, cpad, npad);
  return sivae_launch_status();
}

// Hs, Ws = LOW-resolution (output) size
extern "C" int sivae_conv2d_wino_up_dgrad_supported(int Hs, int Ws) {
  return (Hs >= 8 && Ws >= 16 && !(Ws & 1)) ? 1 : 0;
}

static int wud_grid_blocks() {
  static int g = 0;
  if (g == 0) {
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
        prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
    g = 2 * cus;
  }
  return g;
}

template <int TTH_L2, int TTW_L2, bool KS2>
static int wud_launch(WinoUpDgArgs& a, hipStream_t stream) {
  constexpr int PXH = 2 << TTH_L2, PXW = 2 << TTW_L2;
  constexpr int PH = (1 << TTW_L2) + (1 << TTW_L2) / 4, PLANE = (PXH + 2) * 2 * PH;
  a.nbh = cdiv(a.Hs, PXH);
  a.nbw = cdiv(a.Ws, PXW);
  a.n_n_tiles = cdiv(a.N, KS2 ? WUD_TN / 2 : WUD_TN);
  const long long nblk = (long long)a.B * a.nbh * a.nbw * a.n_n_tiles;
  if (nblk > 0x7fffffffLL) return SIVAE_ERR_RANGE;
  size_t lds = (size_t)2 * WUD_CK * PLANE * sizeof(float);
  if (KS2 && lds < 2 * 64 * 64 * sizeof(float)) lds = 2 * 64 * 64 * sizeof(float);
  a.n_items_base = (int)nblk;
  const int nchunks = 4 * (a.Cpad / WUD_CK);
  if (a.q_per_split <= 0) a.q_per_split = nchunks;
  const long long nitems = nblk * cdiv(nchunks, a.q_per_split);
  if (nitems > 0x7fffffffLL) return SIVAE_ERR_RANGE;
  a.n_items = (int)nitems;
  const int grid = nitems < wud_grid_blocks() ? (int)nitems : wud_grid_blocks();
  a.xcd_group = (sivae_xcd_remap() && !(grid & 7) && nitems > grid) ? 1 : 0;
  hipLaunchKernelGGL((conv_wino_up_dgrad_kernel<TTH_L2, TTW_L2, KS2>), dim3((unsigned)grid), dim3(256), lds, stream,
                     a);
  return sivae_launch_status();
}

static int wud_run(const float* dy, const float* ud, float* dx, int B, int C, int N, int Hs, int Ws, int accumulate,
                   hipStream_t stream, int q_per_split, long long dx_split_stride) {
  const float* dyp = dy;  // [B][C][2Hs][2Ws]
  if (!dyp || !ud || !dx) return SIVAE_ERR_NULL;
  if (B <= 0 || C <= 0 || N <= 0 || Hs <= 0 || Ws <= 0) return SIVAE_ERR_SHAPE;
  if (!sivae_conv2d_wino_up_dgrad_supported(Hs, Ws)) return SIVAE_ERR_SHAPE;
  if (((uintptr_t)dx & 7u) != 0) return SIVAE_ERR_SHAPE;
  const long long hw = (long long)Hs * Ws;
  if (4ll * C * hw * 4 >= 0x7fffffffLL || (long long)N * hw * 4 >= 0x7fffffffLL) return SIVAE_ERR_RANGE;
  WinoUpDgArgs a;
  a.dyp = dyp;
  a.ud = ud;
  a.dx = dx;
  a.B = B;
  a.C = C;
  a.N = N;
  a.Hs = Hs;
  a.Ws = Ws;
  a.Cpad = wud_cpad(C);
  a.Npad = wud_npad(N);
  a.accumulate = accumulate;
  if (4ull * a.Cpad * a.Npad * 48ull >= 0xffffffffull) return SIVAE_ERR_RANGE;
  a.q_per_split = q_per_split;
  a.dx_split_stride = dx_split_stride;
  if (N <= 64) return (Ws >= 32) ? wud_launch<1, 4, true>(a, stream) : wud_launch<2, 3, true>(a, stream);
  return (Ws >= 32) ? wud_launch<1, 4, false>(a, stream) : wud_launch<2, 3, false>(a, stream);
}

extern "C" int sivae_conv2d_wino_up_dgrad(const float* dy, const float* ud, float* dx, int B, int C, int N, int Hs,
                                          int Ws, int accumulate, hipStream_t stream) {
  return wud_run(dy, ud, dx, B, C, N, Hs, Ws, accumulate, stream, 0, 0);
}

// ---- split-K form (small shards) -----------------------------------------------------------------------------------
// dx[i] (+)= sum_s part[s][i], slices in a fixed order (deterministic)
__global__ void __launch_bounds__(256) wud_splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ dx,
                                                                int S, size_t n, size_t slice_stride, int accumulate) {
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * 1024) {
    float4 v = accumulate ? *reinterpret_cast<const float4*>(dx + i) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < S; ++k) {
      const float4 p = *reinterpret_cast<const float4*>(part + (size_t)k * slice_stride + i);
      v.x += p.x;
      v.y += p.y;
      v.z += p.z;
      v.w += p.w;
    }
    *reinterpret_cast<float4*>(dx + i) = v;
  }
}

// number of K slices sivae_conv2d_wino_up_dgrad_splitk_run will use (1: it is the plain kernel)
extern "C" int sivae_conv2d_wino_up_dgrad_splitk(int B, int C, int N, int Hs, int Ws) {
  if (B <= 0 || C <= 0 || N <= 0 || !sivae_conv2d_wino_up_dgrad_supported(Hs, Ws)) return SIVAE_ERR_SHAPE;
  static int enabled = -1;
  if (enabled < 0) {
    const char* e = getenv("SIVAE_WUD_SPLITK");
    enabled = (e && e[0] == '0') ? 0 : 1;
  }
  if (!enabled || N <= 64 || ((Hs * Ws) & 3)) return 1;
  const int pxh = Ws >= 32 ? 4 : 8, pxw = Ws >= 32 ? 32 : 16;
  const long long items = (long long)B * cdiv(Hs, pxh) * cdiv(Ws, pxw) * cdiv(N, WUD_TN);
  const int nch = wud_cpad(C) / WUD_CK;  // 16-channel chunks per phase
  const int cus = wud_grid_blocks() / 2;
  if (items > cus || nch < 8) return 1;
  int S = (int)(2 * cus / items);
  if (S > nch / 4) S = nch / 4;  // >= 4 channel chunks (x 4 phases) per slice
  if (S > 8) S = 8;
  return S < 2 ? 1 : S;
}

extern "C" size_t sivae_conv2d_wino_up_dgrad_splitk_workspace_bytes(int B, int C, int N, int Hs, int Ws) {
  const int S = sivae_conv2d_wino_up_dgrad_splitk(B, C, N, Hs, Ws);
  if (S <= 1) return 0;
  return (size_t)S * B * N * Hs * Ws * sizeof(float);
}

extern "C" int sivae_conv2d_wino_up_dgrad_splitk_run(const float* dy, const float* ud, float* dx, int B, int C, int N,
                                                     int Hs, int Ws, int accumulate, void* workspace,
                                                     size_t workspace_bytes, hipStream_t stream) {
  const int S = sivae_conv2d_wino_up_dgrad_splitk(B, C, N, Hs, Ws);
  if (S < 0) return S;
  if (S <= 1 || ((uintptr_t)dx & 15u) != 0)  // (the reduce kernel moves 16-byte vectors)
    return wud_run(dy, ud, dx, B, C, N, Hs, Ws, accumulate, stream, 0, 0);
  if (!workspace) return SIVAE_ERR_NULL;
  if (((uintptr_t)workspace & 15u) != 0) return SIVAE_ERR_SHAPE;
  if (workspace_bytes < sivae_conv2d_wino_up_dgrad_splitk_workspace_bytes(B, C, N, Hs, Ws)) return SIVAE_ERR_WORKSPACE;
  const int nch = wud_cpad(C) / WUD_CK;
  const int qps = 4 * cdiv(nch, S);  // whole channel chunks: the four parity planes of a chunk stay in one slice
  const size_t n = (size_t)B * N * Hs * Ws;
  float* part = reinterpret_cast<float*>(workspace);
  const int rc = wud_run(dy, ud, part, B, C, N, Hs, Ws, 0, stream, qps, (long long)n);
  if (rc != SIVAE_OK) return rc;
  const int nsl = cdiv(4 * nch, qps);
  size_t nb = (n / 4 + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(wud_splitk_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, stream, part, dx, nsl, n, n, accumulate);
  return sivae_launch_status();
}

// ---- batched packing (pack_batch.h)
int sivae_packjob_wino_up_dgrad(SivaePackJob* j, int Co, int Ci) {
  j->kdim = Co;
  j->ndim = Ci;
  j->kpad = wud_cpad(Co);
  j->npad = wud_npad(Ci);
  j->total = (unsigned long long)j->kpad * j->npad;
  return SIVAE_OK;
}
void sivae_packbatch_wino_up_dgrad(const SivaePackJob* jobs, const unsigned short* block_job, int nblocks, hipStream_t s) {
  hipLaunchKernelGGL(pack_wino_up_dgrad_batch_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, jobs, block_job);
}
