// 5x5 convolutions with a tiny channel count on one side (the RGB edges of the networks):
//   Decoder.predict            64 -> 3   (soft_intro_vae/train_soft_intro_vae.py:159)   forward + weight grad
//   Encoder stem  main.0        3 -> 64  (:89)                                          data grad + weight grad
// A plain implicit GEMM wastes 10x on them (3 channels padded to a 32-wide MFMA tile), and they are
// ~2.3 % of the iteration's FLOPs but were ~12 % of its time.  Here the small channel count is MERGED with
// the kernel column kw into one 16-wide MFMA dimension (3 x 5 = 15 of 16 rows used, 94 % efficiency) on
// v_mfma_f32_16x16x4_f32:
//
//   forward / dgrad (Co <= 3):   P[(co,kw)][q] = sum_{ci,kh} W[co][ci][kh][kw] * X[ci][r+kh-2][q]
//                                Y[co][r][c]   = sum_kw P[(co,kw)][c+kw-2]          (5-term shift-add via LDS)
//   wgrad, small Co:             dW[(co,kw)][ci] (per kh) = sum_{r,c'} dY[co][r][c'-kw+2] * X[ci][r+kh-2][c']
//   wgrad, small Ci:             dW[co][(ci,kw)] (per kh) = sum_{r,c}  dY[co][r][c]       * X[ci][r+kh-2][c+kw-2]
//
// All three stage zero-padded tiles through LDS with raw buffer loads (out-of-image lanes read 0) and keep
// the MFMA operands as conflict-free ds_read_b32.  These layers are close to the per-CU load bandwidth
// (each staged input element feeds only 75 MACs), so the aim is ~50 % MFMA utilisation, not 80 %.
#include "common.h"

namespace {

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------
// pack:  wq[(ci*5 + kh)*16 + (co*5 + kw)]  (rows = K index (ci,kh), 16 columns = (co,kw), zero padded)
//   mode 0 (forward):  wq = w[co][ci][kh][kw]            w is [Co][Ci][5][5], Co <= 3, K = Ci*5
//   mode 1 (dgrad):    wq = w[c][o][4-kh][4-kw]          w is [Cw][Cs][5][5] with Cs <= 3 (the stem's weight):
//                                                        "ci" = c in [0,Cw), "co" = o in [0,Cs)
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) pack5_smallco_kernel(const float* __restrict__ w, float* __restrict__ wq,
                                                            int n_small, int n_big, int mode) {
  const int total = n_big * 5 * 16;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int col = i & 15, row = i >> 4;
    const int cb = row / 5, kh = row - cb * 5;
    const int cs = col / 5, kw = col - cs * 5;
    float v = 0.f;
    if (cs < n_small) {
      if (mode == 0) v = w[((cs * n_big + cb) * 5 + kh) * 5 + kw];
      else v = w[((cb * n_small + cs) * 5 + (4 - kh)) * 5 + (4 - kw)];
    }
    wq[i] = v;
  }
}

extern "C" size_t sivae_pack_conv5_smallco_bytes(int n_small, int n_big) {
  if (n_small <= 0 || n_small > 3 || n_big <= 0) return 0;
  return (size_t)(((n_big + 7) / 8) * 8) * 5 * 16 * sizeof(float);
}

extern "C" int sivae_pack_conv5_smallco(const float* w, float* wq, int n_small, int n_big, int mode,
                                        hipStream_t stream) {
  if (!w || !wq) return SIVAE_ERR_NULL;
  if (n_small <= 0 || n_small > 3 || n_big <= 0) return SIVAE_ERR_SHAPE;
  if (mode != 0 && mode != 1) return SIVAE_ERR_MODE;
  hipError_t e = hipMemsetAsync(wq, 0, sivae_pack_conv5_smallco_bytes(n_small, n_big), stream);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(pack5_smallco_kernel, dim3(cdiv(n_big * 80, 256)), dim3(256), 0, stream, w, wq, n_small, n_big,
                     mode);
  return sivae_launch_status();
}

// ------------------------------------------------------------------------------------------------------------
// forward with Co <= 3:  block = 4 waves = 4 output rows x SW output columns of one image.
// ------------------------------------------------------------------------------------------------------------
struct Conv5FwdArgs {
  const float* x;
  const float* wq;  // packed, [Ci_pad*5][16]
  float* y;
  const float* bias;
  int B, Ci, Co, H, W;
  int xcd_remap;  // 1: blocks of one XCD take a contiguous item range (row groups sharing halo rows meet in one L2)
  int sw;      // output columns per block (<= 128)
  int ntile;   // 16-wide position tiles per wave (odd, <= NWT): 16*ntile >= sw + 4
  int nseg;    // column segments per row
  int nrow4;   // row groups of 4
};

// VEC (round 6; W % 4 == 0 and a 16-byte aligned x): the input tile is staged in 16-byte groups — the LDS column origin is
// c0 - 4 instead of c0 - 2, so that every group is aligned in memory and lies entirely inside or outside the image;
// 8 channels x 8 rows x 4 ntile groups = 256 ntile groups per chunk = ntile dwordx4 loads and ntile ds_write_b128 per thread
// instead of 40 dword loads / stores (the MFMA operand reads shift by two columns).
// EXACT (round 6): a.ntile == NWT at compile time.  With a run-time tile count every MFMA of the K loop and every staging
// load / store sat behind its own scalar branch (227 branches in the loop; an s_waitcnt vmcnt(0) in front of every LDS
// store) — the common widths (ntile = 3: 32 columns, 5: 64, 9: 128 and more) get branch-free instantiations.
template <int NWT, bool VEC, bool EXACT>
__global__ void __launch_bounds__(256, 2) conv5_smallco_fwd_kernel(Conv5FwdArgs a) {
  const int ntile = EXACT ? NWT : a.ntile;
  constexpr int CK = 8, ROWS = 8, NT = 256;
  constexpr int MAXPOS = VEC ? 1 : (ROWS * 16 * NWT + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, g = lane >> 4;
  const int LWX = 16 * ntile;       // odd multiple of 16 -> the two 16-lane groups of a half-wave hit disjoint banks
  const int XPL = ROWS * LWX + 16;    // channel plane stride, == 16 (mod 32)
  float* ws = smem;                   // [CK*5][16]
  float* xs = smem + CK * 5 * 16;     // [CK][XPL]

  // Consecutive blockIdx go round-robin to the 8 XCDs, and vertically adjacent row groups share 4 of their 8 input rows:
  // with the remap XCD x takes the contiguous items [x * grid/8, ...), so the two readers of a halo row are co-resident
  // and the second one hits in L2 (FETCH_SIZE 5.4 GB per launch on 64 -> 3 @256x256 against 2.25 GB of tensors before)
  int bid = blockIdx.x;
  if (a.xcd_remap) bid = ((int)blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3);
  const int seg = bid % a.nseg; bid /= a.nseg;
  const int rg = bid % a.nrow4;
  const int b = bid / a.nrow4;
  const int r0 = rg * 4, c0 = seg * a.sw;
  const int H = a.H, W = a.W, HW = H * W;

  const __amdgpu_buffer_rsrc_t xrs = make_rsrc(a.x + (size_t)b * a.Ci * HW, (unsigned long long)a.Ci * HW * 4ull);
  const int plane = ROWS * LWX;
  unsigned xo[MAXPOS];
  // VEC: group p of this thread = (channel gck, row, 16-byte group of the row); byte offset inside the image (+ the channel
  // part) or OOB, and its LDS float index
  unsigned gvo[VEC ? NWT : 1];
  int glds[VEC ? NWT : 1], gck[VEC ? NWT : 1];
  if (VEC) {
    const int gpr = 4 * ntile, gpc = ROWS * gpr;  // groups per row / per channel plane
#pragma unroll
    for (int p = 0; p < NWT; ++p) {
      const int gi = tid + p * NT;
      const int ck = gi / gpc, rem = gi - ck * gpc;
      const int rr = rem / gpr, q = rem - rr * gpr;
      const int r = r0 + rr - 2, c = c0 - 4 + 4 * q;
      const bool ok = p < ntile && r >= 0 && r < H && c >= 0 && c < W;  // (W % 4 == 0: a group never straddles the edge)
      gvo[p] = ok ? (unsigned)(ck * HW + r * W + c) * 4u : SIVAE_OOB16;
      glds[p] = ck * XPL + rr * LWX + 4 * q;
      gck[p] = ck;
    }
  } else {
#pragma unroll
    for (int p = 0; p < MAXPOS; ++p) {
      const int pos = tid + p * NT;
      unsigned off = SIVAE_OOB;
      if (pos < plane) {
        const int rr = pos / LWX, q = pos - rr * LWX;
        const int r = r0 + rr - 2, c = c0 + q - 2;
        if (r >= 0 && r < H && c >= 0 && c < W) off = (unsigned)(r * W + c) * 4u;
      }
      xo[p] = off;
    }
  }
  // per-lane K offsets: k = 4s + g -> (ci_local, kh)   (VEC: the LDS columns start at c0 - 4, two further left)
  int koff[10];
#pragma unroll
  for (int s = 0; s < 10; ++s) {
    const int k = 4 * s + g;
    const int cl = k / 5, kh = k - cl * 5;
    koff[s] = cl * XPL + (wave + kh) * LWX + li + (VEC ? 2 : 0);
  }
  f32x4 acc[NWT];
#pragma unroll
  for (int t = 0; t < NWT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  float xr[MAXPOS][VEC ? 1 : CK];
  float4 xg[VEC ? NWT : 1];
  float wr[3];  // CK*5*16 = 640 floats / 256 threads
  const int nchunks = (a.Ci + CK - 1) / CK;

#define C5_LOAD(CH)                                                                                   \
  {                                                                                                   \
    _Pragma("unroll") for (int q = 0; q < 3; ++q) {                                                   \
      const int e = tid + q * NT;                                                                     \
      wr[q] = e < CK * 5 * 16 ? a.wq[(size_t)(CH) * (CK * 5 * 16) + e] : 0.f;                         \
    }                                                                                                 \
    if (VEC) {                                                                                        \
      _Pragma("unroll") for (int p = 0; p < NWT; ++p)                                                 \
        if (p < ntile) /* (channels beyond Ci read as zeros: out-of-range offset) */                      \
          xg[p] = buf_load_f32x4(xrs, (CH) * CK + gck[p] < a.Ci ? gvo[p] : SIVAE_OOB16,               \
                                 (unsigned)((CH) * CK) * (unsigned)HW * 4u);                          \
    } else {                                                                                          \
      _Pragma("unroll") for (int ck = 0; ck < CK; ++ck) {                                             \
        const int ci = (CH) * CK + ck;                                                                \
        const int cic = ci < a.Ci ? ci : a.Ci - 1;                                                    \
        _Pragma("unroll") for (int p = 0; p < MAXPOS; ++p)                                            \
          xr[p][ck] = buf_load_f32(xrs, xo[p], (unsigned)cic * (unsigned)HW * 4u);                    \
      }                                                                                               \
    }                                                                                                 \
  }

  C5_LOAD(0)
  for (int ch = 0; ch < nchunks; ++ch) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int e = tid + q * NT;
      if (e < CK * 5 * 16) ws[e] = wr[q];
    }
    if (VEC) {
#pragma unroll
      for (int p = 0; p < NWT; ++p)
        if (p < ntile) *reinterpret_cast<float4*>(xs + glds[p]) = xg[p];
    } else {
#pragma unroll
      for (int ck = 0; ck < CK; ++ck) {
        const bool ok = ch * CK + ck < a.Ci;
#pragma unroll
        for (int p = 0; p < MAXPOS; ++p) {
          const int pos = tid + p * NT;
          if (pos < plane) xs[ck * XPL + pos] = ok ? xr[p][ck] : 0.f;
        }
      }
    }
    __syncthreads();
    if (ch + 1 < nchunks) C5_LOAD(ch + 1)
#pragma unroll
    for (int s = 0; s < 10; ++s) {
      const float aw = ws[(4 * s + g) * 16 + li];
#pragma unroll
      for (int t = 0; t < NWT; ++t) {
        if (t < ntile) acc[t] = mfma16(aw, xs[koff[s] + 16 * t], acc[t]);
      }
    }
    __syncthreads();
  }
#undef C5_LOAD

  // ---- epilogue: P -> LDS, 5-term shift-add, bias, coalesced store
  float* ds = xs + wave * 16 * LWX;  // [16][LWX] per wave (fits: 4*16*LWX <= CK*XPL)
#pragma unroll
  for (int t = 0; t < NWT; ++t) {
    if (t < ntile) {
#pragma unroll
      for (int r = 0; r < 4; ++r) ds[(4 * g + r) * LWX + 16 * t + li] = acc[t][r];
    }
  }
  __syncthreads();
  const int row = r0 + wave;
  if (row < H) {
    const int n_out = a.Co * a.sw;
    for (int idx = lane; idx < n_out; idx += 64) {
      const int co = idx / a.sw, c = idx - co * a.sw;
      if (c0 + c < W) {
        float v = a.bias ? a.bias[co] : 0.f;
#pragma unroll
        for (int kw = 0; kw < 5; ++kw) v += ds[(co * 5 + kw) * LWX + c + kw];
        a.y[((size_t)b * a.Co + co) * HW + (size_t)row * W + c0 + c] = v;
      }
    }
  }
}

extern "C" int sivae_conv5_smallco_fwd(const float* x, const float* wq, float* y, const float* bias, int B, int Ci,
                                       int Co, int H, int W, hipStream_t stream) {
  if (!x || !wq || !y) return SIVAE_ERR_NULL;
  if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return SIVAE_ERR_SHAPE;
  if (Co > 3) return SIVAE_ERR_SHAPE;
  if ((long long)Ci * H * W * 4 >= 0x7fffffffLL) return SIVAE_ERR_RANGE;
  Conv5FwdArgs a;
  a.x = x; a.wq = wq; a.y = y; a.bias = bias;
  a.B = B; a.Ci = Ci; a.Co = Co; a.H = H; a.W = W;
  a.sw = W < 128 ? W : 128;
  int nt = (a.sw + 4 + 15) / 16;
  if ((nt & 1) == 0) ++nt;
  a.ntile = nt;  // <= 9
  a.nseg = cdiv(W, a.sw);
  a.nrow4 = cdiv(H, 4);
  const int LWX = 16 * nt, XPL = 8 * LWX + 16;
  const size_t lds = (size_t)(8 * 5 * 16 + 8 * XPL) * sizeof(float);
  const long long nblk = (long long)B * a.nrow4 * a.nseg;
  if (nblk > 0x7fffffffLL) return SIVAE_ERR_RANGE;
  a.xcd_remap = (sivae_xcd_remap() && (nblk & 7) == 0) ? 1 : 0;
  // 16-byte staging where every group is aligned and never straddles the image edge (SIVAE_CONV5_VEC=0: the dword form)
  static int vec_on = -1;
  if (vec_on < 0) {
    const char* e = getenv("SIVAE_CONV5_VEC");
    vec_on = (e && e[0] == '0') ? 0 : 1;
  }
  const bool vec = vec_on && (W & 3) == 0 && (((uintptr_t)x) & 15u) == 0 && a.sw + 6 <= 16 * nt;  // (columns c0 - 4 .. c0 + sw + 1)
#define C5_GO(N, V, E) hipLaunchKernelGGL((conv5_smallco_fwd_kernel<N, V, E>), dim3((unsigned)nblk), dim3(256), lds, stream, a)
  if (vec && nt == 9) C5_GO(9, true, true);
  else if (vec && nt == 5) C5_GO(5, true, true);
  else if (vec && nt == 3) C5_GO(3, true, true);
  else if (vec) C5_GO(9, true, false);
  else if (nt == 9) C5_GO(9, false, true);
  else C5_GO(9, false, false);
#undef C5_GO
  return sivae_launch_status();
}

// ------------------------------------------------------------------------------------------------------------
// weight gradients.  Pixel tile = 4 rows x 32 columns of one image; block = 4 waves, each owning 16 channels of
// the LARGE side; per kh one 16x16 accumulator tile (5 tiles = 20 registers per wave).  No register prefetch
// (a tile's large-side slab is 64 floats per thread): two or more blocks per CU overlap staging with MFMA.
// SMALL_CO = true : dY has <= 3 channels (predict), X has Ci channels   -> D_kh[(co,kw)][ci]
// SMALL_CO = false: X  has <= 3 channels (stem),    dY has Co channels  -> D_kh[co][(ci,kw)]
// Partials go to part[slice][Co][Ci][5][5]; slices are added in a fixed order by the reduce kernel below.
// ------------------------------------------------------------------------------------------------------------
struct Conv5WgradArgs {
  const float* x;
  const float* dy;
  float* part;
  int B, Ci, Co, H, W;
  int nrow4, ncol32, n_tiles, tiles_per_slice;
};

template <bool SMALL_CO>
__global__ void __launch_bounds__(256, 2) conv5_edge_wgrad_kernel(Conv5WgradArgs a) {
  constexpr int ROWS_BIG = SMALL_CO ? 8 : 4;          // staged rows of the large-side tensor
  constexpr int BLD = ROWS_BIG * 32 + 1;              // odd channel stride of the large-side tile
  constexpr int SROWS = SMALL_CO ? 4 : 8;             // staged rows of the small-side tensor
  constexpr int SLD = 40, SPL = SROWS * SLD;          // small-side tile: [4 channels (last = zeros)][SROWS][40]
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* big = smem;                 // [64][BLD]
  float* sml = smem + 64 * BLD;      // [4][SROWS][40]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, g = lane >> 4;
  const int H = a.H, W = a.W, HW = H * W;
  const int Cbig = SMALL_CO ? a.Ci : a.Co, Csml = SMALL_CO ? a.Co : a.Ci;
  const float* pbig = SMALL_CO ? a.x : a.dy;
  const float* psml = SMALL_CO ? a.dy : a.x;
  const int cb0 = blockIdx.x * 64;
  const int slice = blockIdx.y;
  const int tile_begin = slice * a.tiles_per_slice;
  int tile_end = tile_begin + a.tiles_per_slice;
  if (tile_end > a.n_tiles) tile_end = a.n_tiles;

  f32x4 acc[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};

  // small-side operand address of this lane: row i (SMALL_CO) or column j (else) = (channel, kw)
  const int sc = li / 5, skw = li - sc * 5;  // li == 15 -> channel 3 = the all-zero plane
  const int big_row = (wave * 16 + li) * BLD;

  // ---- per-tile staging, split into "request into registers" and "write to LDS" so that the NEXT tile's global loads are
  // in flight during the current tile's MFMAs (round 6: the loop used to stage and compute strictly in turn, and two blocks
  // per CU — 66 KB of LDS each, no room for a third — covered only half of the load latency: 0.44-0.48 matrix-pipe busy).
  // A tile that continues the row ring (SMALL_CO, the row group below the previous one) brings 4 new rows = 32 floats per
  // thread for the large side and <= 3 for the small one; a tile that restarts the ring (first of a slice / of an image
  // column) needs 64 and is staged unpipelined as before.
  constexpr int PF = 32;  // large-side floats per thread of a pipelined tile
  constexpr int SML_PT = (4 * SPL + 255) / 256;
  float pv[PF], psv[SML_PT];
  auto tile_coords = [&](int tile, int& b, int& rg, int& r0, int& c0) {
    // row group fastest: consecutive tiles of a block are vertically adjacent, so the 4 halo rows of the large-side
    // slab that they share (SMALL_CO: 8 staged rows per 4 output rows) stay in the LDS row ring
    int t = tile;
    rg = t % a.nrow4; t /= a.nrow4;
    const int cseg = t % a.ncol32;
    b = t / a.ncol32;
    r0 = rg * 4;
    c0 = cseg * 32;
  };
  // small-side tile [4][SROWS][40]: cols c0-2 .. c0+37 (only 36 used), channel >= Csml -> 0
  auto load_sml = [&](int b, int r0, int c0) {
    const __amdgpu_buffer_rsrc_t srs = make_rsrc(psml + (size_t)b * Csml * HW, (unsigned long long)Csml * HW * 4ull);
#pragma unroll
    for (int q = 0; q < SML_PT; ++q) {
      const int e = tid + q * 256;
      const int ch = e / SPL, rem = e - ch * SPL;
      const int rr = rem / SLD, cc = rem - rr * SLD;
      const int r = r0 + rr - (SMALL_CO ? 0 : 2), c = c0 + cc - 2;
      const bool ok = e < 4 * SPL && ch < Csml && r >= 0 && r < H && c >= 0 && c < W;
      psv[q] = buf_load_f32(srs, ok ? (unsigned)(r * W + c) * 4u : SIVAE_OOB, (unsigned)(ch < Csml ? ch : 0) * (unsigned)HW * 4u);
    }
  };
  auto store_sml = [&]() {
#pragma unroll
    for (int q = 0; q < SML_PT; ++q) {
      const int e = tid + q * 256;
      if (e < 4 * SPL) sml[e] = psv[q];
    }
  };
  // large side of a PIPELINED tile: SMALL_CO — the 4 new ring rows (r0 + 2 .. r0 + 5) x 32 columns x 64 channels, 128
  // positions x 2 channel groups of 32; else — the 4 dY rows, the same shape
  auto load_big_pf = [&](int b, int r0, int c0) {
    const __amdgpu_buffer_rsrc_t brs = make_rsrc(pbig + (size_t)b * Cbig * HW, (unsigned long long)Cbig * HW * 4ull);
    const int pos = tid & 127, cgrp = tid >> 7;
    const int rr = (SMALL_CO ? 4 : 0) + (pos >> 5), cc = pos & 31;
    const int r = r0 + rr - (SMALL_CO ? 2 : 0), c = c0 + cc;
    const unsigned off = (r >= 0 && r < H && c < W) ? (unsigned)(r * W + c) * 4u : SIVAE_OOB;
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      int ch = cb0 + cgrp * PF + q;
      ch = ch < Cbig ? ch : Cbig - 1;
      pv[q] = buf_load_f32(brs, off, (unsigned)ch * (unsigned)HW * 4u);
    }
  };
  auto store_big_pf = [&](int r0) {
    const int pos = tid & 127, cgrp = tid >> 7;
    const int rr = (SMALL_CO ? 4 : 0) + (pos >> 5), cc = pos & 31;
    const int slot = SMALL_CO ? ((r0 + rr) & 7) : rr;
#pragma unroll
    for (int q = 0; q < PF; ++q) big[(cgrp * PF + q) * BLD + slot * 32 + cc] = pv[q];
  };
  // large side of a tile that RESTARTS the ring (SMALL_CO only): all 8 rows, staged in place
  auto stage_big_full = [&](int b, int r0, int c0) {
    const __amdgpu_buffer_rsrc_t brs = make_rsrc(pbig + (size_t)b * Cbig * HW, (unsigned long long)Cbig * HW * 4ull);
    const int rr = tid >> 5, cc = tid & 31;
    const int r = r0 - 2 + rr, c = c0 + cc;
    const int slot = (r0 + rr) & 7;
    const unsigned off = (r >= 0 && r < H && c < W) ? (unsigned)(r * W + c) * 4u : SIVAE_OOB;
    for (int q0 = 0; q0 < 64; q0 += 16) {
      float v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        int ch = cb0 + q0 + q;
        ch = ch < Cbig ? ch : Cbig - 1;
        v[q] = buf_load_f32(brs, off, (unsigned)ch * (unsigned)HW * 4u);
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) big[(q0 + q) * BLD + slot * 32 + cc] = v[q];
    }
  };
  // does `tile` continue the ring of the tile before it (same block, the row group below)?
  auto continues = [&](int tile) { return !SMALL_CO || (tile > tile_begin && tile % a.nrow4 != 0); };

  if (tile_begin < tile_end) {  // first tile of the slice: staged in place
    int b, rg, r0, c0;
    tile_coords(tile_begin, b, rg, r0, c0);
    if (SMALL_CO) {
      stage_big_full(b, r0, c0);
    } else {
      load_big_pf(b, r0, c0);
      store_big_pf(r0);
    }
    load_sml(b, r0, c0);
    store_sml();
  }
  __syncthreads();
  for (int tile = tile_begin; tile < tile_end; ++tile) {
    int b, rg, r0, c0;
    tile_coords(tile, b, rg, r0, c0);
    const bool has_next = tile + 1 < tile_end;
    const bool pipe = has_next && continues(tile + 1);
    int nb = b, nrg = rg, nr0 = r0, nc0 = c0;
    if (has_next) tile_coords(tile + 1, nb, nrg, nr0, nc0);
    if (pipe) {  // the next tile's operands: requested now, consumed behind the MFMAs
      load_big_pf(nb, nr0, nc0);
      load_sml(nb, nr0, nc0);
    }
#pragma unroll 1
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int cq = 4 * s + g;  // column inside the 32-wide tile handled by this lane's k index
        if (SMALL_CO) {
          // A[(co,kw)][c'] = dY[co][r][c'-kw+2]  (small tile col = c' - c0 - kw + 4) ; B[c'][ci] = X[ci][r+kh-2][c']
          const float av = sml[sc * SPL + r * SLD + cq - skw + 4];
#pragma unroll
          for (int kh = 0; kh < 5; ++kh) acc[kh] = mfma16(av, big[big_row + ((r0 + r + kh) & 7) * 32 + cq], acc[kh]);  // (row ring)
        } else {
          // A[co][c] = dY[co][r][c] ; B[c][(ci,kw)] = X[ci][r+kh-2][c+kw-2]  (small tile col = c - c0 + kw)
          const float av = big[big_row + r * 32 + cq];
#pragma unroll
          for (int kh = 0; kh < 5; ++kh) acc[kh] = mfma16(av, sml[sc * SPL + (r + kh) * SLD + cq + skw], acc[kh]);
        }
      }
    }
    __syncthreads();  // every wave is done with this tile's LDS operands
    if (has_next) {
      if (pipe) {
        store_big_pf(nr0);
        store_sml();
      } else {  // ring restart (SMALL_CO): stage the next tile in place
        stage_big_full(nb, nr0, nc0);
        load_sml(nb, nr0, nc0);
        store_sml();
      }
      __syncthreads();
    }
  }
  // ---- partial store: D_kh rows = 4g + reg, cols = li
  float* outp = a.part + (size_t)slice * a.Co * a.Ci * 25;
#pragma unroll
  for (int kh = 0; kh < 5; ++kh) {
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int row = 4 * g + rg, col = li;
      int co, ci, kw;
      if (SMALL_CO) { co = row / 5; kw = row - co * 5; ci = cb0 + wave * 16 + col; }
      else { co = cb0 + wave * 16 + row; ci = col / 5; kw = col - ci * 5; }
      if (co < a.Co && ci < a.Ci) outp[(((size_t)co * a.Ci + ci) * 5 + kh) * 5 + kw] = acc[kh][rg];
    }
  }
}

namespace {
struct EdgePlan { int nrow4, ncol32, n_tiles, tps, n_slices, nblkx; };
static EdgePlan edge_plan(int B, int Cbig, int H, int W) {
  EdgePlan p;
  p.nrow4 = cdiv(H, 4);
  p.ncol32 = cdiv(W, 32);
  p.n_tiles = B * p.nrow4 * p.ncol32;
  p.nblkx = cdiv(Cbig, 64);
  int want = cdiv(2048, p.nblkx);
  int tps = cdiv(p.n_tiles, want);
  if (tps < 8) tps = 8;
  if (tps > p.n_tiles) tps = p.n_tiles;
  p.tps = tps;
  p.n_slices = cdiv(p.n_tiles, tps);
  return p;
}
}  // namespace

extern "C" size_t sivae_conv5_edge_wgrad_workspace_bytes(int B, int Ci, int Co, int H, int W) {
  if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return 0;
  if (Ci > 3 && Co > 3) return 0;
  EdgePlan p = edge_plan(B, Co <= 3 ? Ci : Co, H, W);
  return (size_t)p.n_slices * Co * Ci * 25 * sizeof(float);
}

// dw[Co][Ci][5][5] for a 5x5 conv where min(Ci, Co) <= 3.
extern "C" int sivae_conv5_edge_wgrad(const float* x, const float* dy, float* dw, int B, int Ci, int Co, int H, int W,
                                      void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!x || !dy || !dw) return SIVAE_ERR_NULL;
  if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return SIVAE_ERR_SHAPE;
  if (Ci > 3 && Co > 3) return SIVAE_ERR_SHAPE;
  if ((long long)Ci * H * W * 4 >= 0x7fffffffLL || (long long)Co * H * W * 4 >= 0x7fffffffLL) return SIVAE_ERR_RANGE;
  const bool small_co = Co <= 3;
  EdgePlan p = edge_plan(B, small_co ? Ci : Co, H, W);
  const size_t need = (size_t)p.n_slices * Co * Ci * 25 * sizeof(float);
  if (!workspace || workspace_bytes < need) return SIVAE_ERR_WORKSPACE;
  Conv5WgradArgs a;
  a.x = x; a.dy = dy; a.part = (float*)workspace;
  a.B = B; a.Ci = Ci; a.Co = Co; a.H = H; a.W = W;
  a.nrow4 = p.nrow4; a.ncol32 = p.ncol32; a.n_tiles = p.n_tiles; a.tiles_per_slice = p.tps;
  dim3 grid(p.nblkx, p.n_slices);
  if (small_co) {
    const size_t lds = (size_t)(64 * (8 * 32 + 1) + 4 * 4 * 40) * sizeof(float);
    auto kern = conv5_edge_wgrad_kernel<true>;
    static size_t lds_hwm = 0;
    const int rc_lds = sivae_ensure_lds(reinterpret_cast<const void*>(kern), lds, &lds_hwm);
    if (rc_lds != SIVAE_OK) return rc_lds;
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, a);
  } else {
    const size_t lds = (size_t)(64 * (4 * 32 + 1) + 4 * 8 * 40) * sizeof(float);
    hipLaunchKernelGGL((conv5_edge_wgrad_kernel<false>), grid, dim3(256), lds, stream, a);
  }
  const int numel = Co * Ci * 25;
  sivae_launch_slice_reduce((const float*)workspace, dw, p.n_slices, (size_t)numel, stream);
  return sivae_launch_status();
}
