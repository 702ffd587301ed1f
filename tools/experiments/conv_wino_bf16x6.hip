// EXPERIMENT (not part of libsivae_hip.so; see NOTES_NEXT_ROUND.md): Winograd F(2x2,3x3) forward with the fp32 products
// built from bf16 pieces on v_mfma_f32_32x32x16_bf16 — every fp32 operand value a = a0 + a1 + a2 (three bf16 pieces,
// exact), the six piece products with p + q <= 2 accumulated in fp32: fp32-level accuracy (profiles/r1_probe_mfma_bf16.txt,
// profiles/r1_wino_numerics.txt) at 6 x 32 = 192 matrix-pipe cycles per 32x32 x 16-channel block instead of 8 x 64 = 512.
//
// Same block structure as conv_wino.hip (4 waves = the 4 frequency columns, 64 output channels x 32 tiles, persistent
// items, 16-channel chunks, double-buffered raw halo in LDS, one barrier per chunk); what changes is the K loop:
//   * one MFMA consumes a whole 16-channel chunk: lane (tile l31, half hh) supplies channels 2e + hh, e = 0..7, so it
//     transforms 8 channels (64 ds_read_b32 + 64 VALU), then per frequency splits its 8 values into three packed pieces;
//   * U is packed as bf16 pieces [j][chunk][i][piece][hh][co][8]: one 16-byte load per (frequency, co-subtile, piece),
//     ring of two frequencies (48 VGPRs) refilled two frequency-steps (768 MFMA cycles) ahead.
// Plain forward only (no prologue / statistics / accumulate): this file exists to settle registers, layout and speed.
//
// build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 -o tools/_build/wino_bf16x6 <this file> && tools/_build/wino_bf16x6
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define OOB 0xFFFFFFFFu
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned long long bytes) {
  const unsigned n = bytes > 0xFFFFFFFEull ? 0xFFFFFFFEu : (unsigned)bytes;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)n, 0x00020000);
}
__device__ __forceinline__ float buf_load_f32(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ u32x4 buf_load_b128(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ void buf_store_f32x2(__amdgpu_buffer_rsrc_t r, float a, float b, unsigned voff) {
  f32x2 v;
  v[0] = a;
  v[1] = b;
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, (int)voff, 0, 0);
}
static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

struct B6Args {
  const float* x;
  const void* up;  // bf16 pieces [4 j][nch][4 i][3 p][2 hh][Co_pad][8]
  float* y;
  int B, Ci, Co, H, W;
  int Ci_pad, Co_pad;
  int nbh, nbw, n_co_tiles, n_items;
};

#define B6_CK 16
#ifndef B6_OCC
#define B6_OCC 1  // waves per SIMD: 2 spills heavily (359 VGPRs) with this K loop, 1 = 512 registers, one block per CU
#endif
#ifdef B6_FENCE  // pin each frequency's MFMAs before the next frequency's split (first version); default: let the
#define B6_FREQ_FENCE __builtin_amdgcn_sched_barrier(0);  // compiler interleave the split of frequency i+1 with them
#else
#define B6_FREQ_FENCE
#endif
#ifndef B6_WM
#define B6_WM 2  // 32-channel output subtiles per wave
#endif
#define B6_EX_FLOATS (2 * 4 * B6_WM * 16 * 64)

template <int TTH_L2, int TTW_L2>
__global__ void __launch_bounds__(256, B6_OCC) conv_wino_b6_kernel(B6Args a) {
  constexpr int TTH = 1 << TTH_L2, TTW = 1 << TTW_L2;
  static_assert(TTH * TTW == 32, "one image per 32-tile block in this experiment");
  constexpr int WM = B6_WM, NW = 4;
  constexpr int PXH = 2 * TTH, PXW = 2 * TTW;
  constexpr int LH = PXH + 2, LWU = PXW + 2;
#ifdef B6_LDSDIRECT
  constexpr int PH = TTW + TTW / 4, RS = 2 * PH, PLANE = 256;  // plane slot == staging thread (LDS-direct loads)
  static_assert(LH * RS <= 256, "halo plane fits one slot per thread");
#else
  constexpr int PH = TTW + TTW / 4, RS = 2 * PH, PLANE = LH * RS;
#endif
  constexpr int NPOS = LH * LWU;
  static_assert(NPOS <= 256, "one halo position per thread");
  constexpr int CK = B6_CK, XBUF = CK * PLANE;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int wj = wave;
  const int H = a.H, W = a.W, HW = H * W;
  const int nch = a.Ci_pad / CK;

  int item = blockIdx.x;
  int pt, b, r0, c0, co0;
  __amdgpu_buffer_rsrc_t xrsrc;
  const __amdgpu_buffer_rsrc_t ursrc = make_rsrc(a.up, 96ull * a.Ci_pad * a.Co_pad);
  unsigned xo, ua_base;
#ifdef B6_LDSDIRECT
  // thread t stages plane slot t = xrr*RS + parity*PH + col/2 (slots outside the halo load nothing: OOB -> 0)
  const int xrr = tid / RS, xcc = 2 * ((tid % RS) % PH) + (tid % RS) / PH;
  const bool xslot = xrr < LH && xcc < LWU;
#else
  const int teff = tid % NPOS;
  const int xrr = teff / LWU, xcc = teff % LWU;
  const int xl = xrr * RS + (xcc & 1) * PH + (xcc >> 1);
  const bool xslot = true;
#endif
#define B6_SETUP(ITEM)                                                   \
  {                                                                      \
    const int co_tile = (ITEM) % a.n_co_tiles;                           \
    pt = (ITEM) / a.n_co_tiles;                                          \
    const int tbx = pt % a.nbw;                                          \
    const int t2 = pt / a.nbw;                                           \
    const int tby = t2 % a.nbh;                                          \
    b = t2 / a.nbh;                                                      \
    r0 = tby * PXH;                                                      \
    c0 = tbx * PXW;                                                      \
    co0 = co_tile * 32 * WM;                                                 \
    xrsrc = make_rsrc(a.x + (size_t)b * a.Ci * HW, (unsigned long long)a.Ci * HW * 4ull); \
    const int r = r0 + xrr - 1, c = c0 + xcc - 1;                        \
    xo = OOB;                                                            \
    if (xslot && r >= 0 && r < H && c >= 0 && c < W) xo = (unsigned)(r * W + c) * 4u; \
    ua_base = (unsigned)(wj * nch * 24) * (unsigned)a.Co_pad * 16u + (unsigned)co0 * 16u; \
  }

  // A operand: lane (co = l31 of subtile m, half hh): 16 bytes = channels 2e + hh of the chunk
  const unsigned va0 = (unsigned)(hh * a.Co_pad + l31) * 16u;
  const unsigned ua_step = (unsigned)a.Co_pad * 32u;  // bytes per (chunk, i, piece) record = 2 hh x Co_pad x 16

  // B operand: raw columns (ca, cb) and sign of frequency column j
  const int tx = l31 & (TTW - 1), ty = l31 >> TTW_L2;
  const int ca = (wj == 0) ? 0 : ((wj == 2) ? 2 : 1);
  const int cb = (wj == 0) ? 2 : ((wj == 1) ? 2 : ((wj == 2) ? 1 : 3));
  const float sgn = (wj == 1) ? 1.f : -1.f;
  const int bb = hh * PLANE + 2 * ty * RS + tx;
  const int base_a = bb + (ca & 1) * PH + (ca >> 1);
  const int base_b = bb + (cb & 1) * PH + (cb >> 1);

  f32x16 acc[4][WM];
#ifndef B6_LDSDIRECT
  float xr0[CK], xr1[CK];  // halo staging registers of chunk c: set c & 1, requested TWO chunks ahead
#else
  const int xr0 = 0, xr1 = 0;
  (void)xr0;
  (void)xr1;
#endif
#ifndef B6_RING
#define B6_RING 2
#endif
  u32x4 AR[B6_RING][WM][3];  // ring over frequency steps f = chunk * 4 + i: slot f % B6_RING
  const int nfsteps = nch * 4;

#define B6_LOAD_X(CH, XR)                                                \
  {                                                                      \
    _Pragma("unroll") for (int ck = 0; ck < CK; ++ck) {                  \
      const int ci = (CH)*CK + ck;                                       \
      const int cic = ci < a.Ci ? ci : a.Ci - 1;                         \
      XR[ck] = buf_load_f32(xrsrc, xo, (unsigned)cic * (unsigned)HW * 4u); \
    }                                                                    \
  }
#define B6_LOAD_A(F, SLOT)                                               \
  {                                                                      \
    _Pragma("unroll") for (int p = 0; p < 3; ++p) {                      \
      const unsigned so = ua_base + (unsigned)((F)*3 + p) * ua_step;     \
      _Pragma("unroll") for (int m = 0; m < WM; ++m) AR[SLOT][m][p] = buf_load_b128(ursrc, va0 + m * 512u, so); \
    }                                                                    \
  }
#define B6_STORE_X(CH, BUF, XR)                                          \
  {                                                                      \
    _Pragma("unroll") for (int ck = 0; ck < CK; ++ck) {                  \
      const int ci = (CH)*CK + ck;                                       \
      xs[(BUF)*XBUF + ck * PLANE + xl] = ci < a.Ci ? XR[ck] : 0.f;       \
    }                                                                    \
  }
  // frequency step I of chunk CH: split the lane's 8 channel values of frequency I into three packed bf16 pieces and
  // run the 12 MFMAs (2 co-subtiles x 6 piece pairs, smallest products first); then refill the U ring slot
#define B6_FREQ(CH, I)                                                   \
  {                                                                      \
    bf16x8 bp[3];                                                        \
    _Pragma("unroll") for (int e = 0; e < 8; ++e) {                      \
      const float val = v[e][I];                                         \
      const __bf16 p0 = (__bf16)val;                                     \
      const float r1 = val - (float)p0;                                  \
      const __bf16 p1 = (__bf16)r1;                                      \
      const __bf16 p2 = (__bf16)(r1 - (float)p1);                        \
      bp[0][e] = p0;                                                     \
      bp[1][e] = p1;                                                     \
      bp[2][e] = p2;                                                     \
    }                                                                    \
    constexpr int slot = (I) % B6_RING;                                  \
    _Pragma("unroll") for (int m = 0; m < WM; ++m) {                     \
      const bf16x8 a0 = __builtin_bit_cast(bf16x8, AR[slot][m][0]);      \
      const bf16x8 a1 = __builtin_bit_cast(bf16x8, AR[slot][m][1]);      \
      const bf16x8 a2 = __builtin_bit_cast(bf16x8, AR[slot][m][2]);      \
      acc[I][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bp[2], acc[I][m], 0, 0, 0); \
      acc[I][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bp[1], acc[I][m], 0, 0, 0); \
      acc[I][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, bp[0], acc[I][m], 0, 0, 0); \
      acc[I][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bp[1], acc[I][m], 0, 0, 0); \
      acc[I][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bp[0], acc[I][m], 0, 0, 0); \
      acc[I][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bp[0], acc[I][m], 0, 0, 0); \
    }                                                                    \
    B6_FREQ_FENCE                                                        \
    { /* unconditional refill (clamped past the end): a conditional one makes every wait assume NO younger loads */ \
      const int fnext = (CH)*4 + (I) + B6_RING;                          \
      B6_LOAD_A(fnext < nfsteps ? fnext : nfsteps - 1, slot)             \
    }                                                                    \
  }
#ifdef B6_LDSDIRECT
#define B6_LOAD_LDS(CH, BUF)                                             \
  {                                                                      \
    _Pragma("unroll") for (int ck = 0; ck < CK; ++ck)                    \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(                          \
          xrsrc, (float __attribute__((address_space(3)))*)(xs + (BUF)*XBUF + ck * PLANE + wave * 64), 4, xo, \
          (unsigned)((CH)*CK + ck) * (unsigned)HW * 4u, 0, 0);           \
  }
#define B6_TOP(CH, BUF, NEXT, XCUR) if (NEXT) B6_LOAD_LDS((CH) + 1, (BUF) ^ 1)
#define B6_BOTTOM(CH, BUF, NEXT, XNXT) __builtin_amdgcn_s_waitcnt(0x0F70);
#else
#define B6_TOP(CH, BUF, NEXT, XCUR) if ((CH) + 2 < nch) B6_LOAD_X((CH) + 2, XCUR)
#define B6_BOTTOM(CH, BUF, NEXT, XNXT) if (NEXT) B6_STORE_X((CH) + 1, (BUF) ^ 1, XNXT)
#endif
#define B6_MMA(CH, BUF, NEXT, XCUR, XNXT)                                \
  {                                                                      \
    B6_TOP(CH, BUF, NEXT, XCUR)                                          \
    float v[8][4];                                                       \
    _Pragma("unroll") for (int e = 0; e < 8; ++e) {                      \
      const float* pa = xs + (BUF)*XBUF + 2 * e * PLANE + base_a;        \
      const float* pb_ = xs + (BUF)*XBUF + 2 * e * PLANE + base_b;       \
      float t[4];                                                        \
      _Pragma("unroll") for (int r = 0; r < 4; ++r) t[r] = pa[r * RS] + sgn * pb_[r * RS]; \
      v[e][0] = t[0] - t[2];                                             \
      v[e][1] = t[1] + t[2];                                             \
      v[e][2] = t[2] - t[1];                                             \
      v[e][3] = t[1] - t[3];                                             \
    }                                                                    \
    B6_FREQ(CH, 0)                                                       \
    B6_FREQ(CH, 1)                                                       \
    B6_FREQ(CH, 2)                                                       \
    B6_BOTTOM(CH, BUF, NEXT, XNXT)                                       \
    __builtin_amdgcn_sched_barrier(0);                                   \
    B6_FREQ(CH, 3)                                                       \
    __syncthreads();                                                     \
  }

  B6_SETUP(item)
#ifdef B6_LDSDIRECT
  B6_LOAD_LDS(0, 0)
#else
  B6_LOAD_X(0, xr0)
  if (nch > 1) B6_LOAD_X(1, xr1)
#endif
  B6_LOAD_A(0, 0)
  B6_LOAD_A(1, 1)
  if (B6_RING == 4) {
    B6_LOAD_A(2, 2 % B6_RING)
    B6_LOAD_A(3, 3 % B6_RING)
  }
  for (;;) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][m][r] = 0.f;
#ifdef B6_LDSDIRECT
    __builtin_amdgcn_s_waitcnt(0x0F70);
#else
    B6_STORE_X(0, 0, xr0)
#endif
    __syncthreads();
    int ch = 0;
    for (; ch + 1 < nch; ch += 2) {
      B6_MMA(ch, 0, true, xr0, xr1)
      const bool more = ch + 2 < nch;
      B6_MMA(ch + 1, 1, more, xr1, xr0)
    }
    if (ch < nch) B6_MMA(ch, 0, false, xr0, xr1)

    const int e_b = b, e_r0 = r0, e_c0 = c0, e_co0 = co0;
    const int next = item + (int)gridDim.x;
    const bool has_next = next < a.n_items;
    if (has_next) {
      B6_SETUP(next)
#ifndef B6_LDSDIRECT
      B6_LOAD_X(0, xr0)
      if (nch > 1) B6_LOAD_X(1, xr1)
#endif
      B6_LOAD_A(0, 0)
      B6_LOAD_A(1, 1)
      if (B6_RING == 4) {
        B6_LOAD_A(2, 2 % B6_RING)
        B6_LOAD_A(3, 3 % B6_RING)
      }
    }
    // ---- output transform (as conv_wino.hip): rows in registers, columns across the four waves through LDS
    {
      float* ex = smem;  // [2 ar][4 j][2 cg][16 r][64 lanes]
      constexpr int PPW = 16 * WM / NW;
      const __amdgpu_buffer_rsrc_t yrsrc =
          make_rsrc(a.y + (size_t)e_b * a.Co * HW, (unsigned long long)a.Co * HW * 4ull);
      const int row_base = e_r0 + 2 * ty, col = e_c0 + 2 * tx;
#pragma unroll
      for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          ex[((wj * WM + m) * 16 + r) * 64 + lane] = acc[0][m][r] + acc[1][m][r] + acc[2][m][r];
          ex[(((4 + wj) * WM + m) * 16 + r) * 64 + lane] = acc[1][m][r] - acc[2][m][r] - acc[3][m][r];
        }
      __syncthreads();
#pragma unroll
      for (int rr = 0; rr < PPW; ++rr) {
        const int p = wave * PPW + rr;
        const int chn = e_co0 + (p >> 4) * 32 + (p & 3) + 8 * ((p & 15) >> 2) + 4 * hh;
        const unsigned yo = (chn < a.Co && col < W) ? (unsigned)((chn * H + row_base) * W + col) * 4u : OOB;
#pragma unroll
        for (int ar = 0; ar < 2; ++ar) {
          float e[4];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) e[jj] = ex[(((ar * 4 + jj) * WM + (p >> 4)) * 16 + (p & 15)) * 64 + lane];
          const bool ok = yo != OOB && row_base + ar < H;
          buf_store_f32x2(yrsrc, e[0] + e[1] + e[2], e[1] - e[2] - e[3], ok ? yo + (unsigned)(ar * W) * 4u : OOB);
        }
      }
      __syncthreads();
    }
    if (!has_next) break;
#ifdef B6_LDSDIRECT
    B6_LOAD_LDS(0, 0)
#endif
    item = next;
  }
}

// U = G g G^T split into bf16 pieces, packed [j][chunk][i][piece][hh][Co_pad][8]: element e of half hh = channel 2e + hh
__global__ void __launch_bounds__(256) pack_wino_b6_kernel(const float* __restrict__ w, unsigned short* __restrict__ up,
                                                           int Co, int Ci, int kpad, int npad) {
  const size_t total = (size_t)kpad * npad;
  const int nch = kpad / 16;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int n = (int)(idx % npad), k = (int)(idx / npad);
    float g[3][3];
    const bool ok = k < Ci && n < Co;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) g[r][c] = ok ? w[((size_t)n * Ci + k) * 9 + r * 3 + c] : 0.f;
    float gg[4][3];
    for (int c = 0; c < 3; ++c) {
      gg[0][c] = g[0][c];
      gg[1][c] = 0.5f * (g[0][c] + g[1][c] + g[2][c]);
      gg[2][c] = 0.5f * (g[0][c] - g[1][c] + g[2][c]);
      gg[3][c] = g[2][c];
    }
    float u[4][4];
    for (int i = 0; i < 4; ++i) {
      u[i][0] = gg[i][0];
      u[i][1] = 0.5f * (gg[i][0] + gg[i][1] + gg[i][2]);
      u[i][2] = 0.5f * (gg[i][0] - gg[i][1] + gg[i][2]);
      u[i][3] = gg[i][2];
    }
    const int chunk = k >> 4, kk = k & 15, hh = kk & 1, e = kk >> 1;
    for (int j = 0; j < 4; ++j)
      for (int i = 0; i < 4; ++i) {
        float val = u[i][j];
        for (int p = 0; p < 3; ++p) {
          const __bf16 h = (__bf16)val;
          val -= (float)h;
          const size_t rec = ((((size_t)j * nch + chunk) * 4 + i) * 3 + p) * 2 + hh;
          up[(rec * npad + n) * 8 + e] = __builtin_bit_cast(unsigned short, h);
        }
      }
  }
}

#define CK_(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

static int run(int B, int Ci, int Co, int H, int W, bool check, int reps) {
  const int kpad = (Ci + 15) / 16 * 16, npad = (Co + 63) / 64 * 64;
  std::vector<float> x((size_t)B * Ci * H * W), w((size_t)Co * Ci * 9), y((size_t)B * Co * H * W);
  srand(7);
  for (auto& v : x) v = (float)(rand() % 20001) / 10000.f - 1.f;
  for (auto& v : w) v = ((float)(rand() % 20001) / 10000.f - 1.f) / sqrtf((float)Ci * 9.f);
  float *dx, *dw, *dy;
  unsigned short* dup;
  CK_(hipMalloc(&dx, x.size() * 4)); CK_(hipMalloc(&dw, w.size() * 4)); CK_(hipMalloc(&dy, y.size() * 4));
  CK_(hipMalloc(&dup, (size_t)96 * kpad * npad));
  CK_(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
  CK_(hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice));
  CK_(hipMemset(dy, 0xff, y.size() * 4));
  hipLaunchKernelGGL(pack_wino_b6_kernel, dim3(cdiv((long long)kpad * npad, 256)), dim3(256), 0, 0, dw, dup, Co, Ci, kpad, npad);
  B6Args a;
  a.x = dx; a.up = dup; a.y = dy;
  a.B = B; a.Ci = Ci; a.Co = Co; a.H = H; a.W = W; a.Ci_pad = kpad; a.Co_pad = npad;
  a.nbh = cdiv(H, 4); a.nbw = cdiv(W, 32); a.n_co_tiles = cdiv(Co, 32 * B6_WM);
  a.n_items = B * a.nbh * a.nbw * a.n_co_tiles;
  const size_t lds_x = (size_t)2 * 16 * 256 * 4;  // two halo buffers (TTH = 2, TTW = 16)
  const size_t lds = (size_t)B6_EX_FLOATS * 4 > lds_x ? (size_t)B6_EX_FLOATS * 4 : lds_x;
  auto kern = conv_wino_b6_kernel<1, 4>;
  CK_(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int grid = a.n_items < 256 * B6_OCC ? a.n_items : 256 * B6_OCC;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, a);
  CK_(hipDeviceSynchronize());
  if (check) {
    CK_(hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost));
    double err = 0, ymax = 0;
    for (int b = 0; b < B; ++b)
      for (int co = 0; co < Co; ++co)
        for (int h = 0; h < H; ++h)
          for (int ww = 0; ww < W; ++ww) {
            double s = 0;
            for (int ci = 0; ci < Ci; ++ci)
              for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) {
                  const int hy = h + r - 1, wx = ww + c - 1;
                  if (hy < 0 || hy >= H || wx < 0 || wx >= W) continue;
                  s += (double)x[((size_t)(b * Ci + ci) * H + hy) * W + wx] * w[((size_t)co * Ci + ci) * 9 + r * 3 + c];
                }
            const double got = y[((size_t)(b * Co + co) * H + h) * W + ww];
            err = fmax(err, fabs(got - s));
            ymax = fmax(ymax, fabs(s));
          }
    printf("B=%d %d->%d @%dx%d: max-norm relative error vs fp64 direct conv = %.3e\n", B, Ci, Co, H, W, err / ymax);
  }
  if (reps > 0) {
    hipEvent_t e0, e1;
    CK_(hipEventCreate(&e0)); CK_(hipEventCreate(&e1));
    CK_(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, a);
    CK_(hipEventRecord(e1, 0));
    CK_(hipEventSynchronize(e1));
    float ms = 0;
    CK_(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double fl = 2.0 * B * H * W * (double)Ci * Co * 9;
    printf("B=%d %d->%d @%dx%d: %.3f ms  %.1f TF/s algorithmic (conv_wino.hip: see tools/bench_conv.py)\n", B, Ci, Co, H, W, ms,
           fl / ms / 1e9);
  }
  (void)hipFree(dx); (void)hipFree(dw); (void)hipFree(dy); (void)hipFree(dup);
  return 0;
}

int main(int argc, char** argv) {
  if (run(2, 48, 72, 8, 32, true, 0)) return 1;   // ragged output channels, three chunks
  if (run(1, 64, 64, 12, 64, true, 0)) return 1;
  if (argc > 1) return 0;
  if (run(128, 256, 256, 64, 64, false, 5)) return 1;
  if (run(128, 512, 512, 32, 32, false, 5)) return 1;
  if (run(128, 128, 128, 128, 128, false, 5)) return 1;
  if (run(128, 64, 64, 256, 256, false, 3)) return 1;
  return 0;
}
