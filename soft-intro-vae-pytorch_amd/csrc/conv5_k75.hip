// 5x5 stride-1 "same" convolution FROM <= 3 channels INTO <= 64 (the encoder stem, soft_intro_vae/train_soft_intro_vae.py:88,
// and the data gradient of Decoder.predict, :159) on v_mfma_f32_32x32x2_f32 with the whole contraction merged:
//
//     y[co][px] = sum_k W[co][k] * X[k][px],     k = (ci, kh, kw) = ci*25 + kh*5 + kw,   K = 75 (76 with one zero column)
//
// The direct kernel of conv_fwd.hip pads 3 input channels to 4 per tap (25 taps x 2 MFMAs = 50 per 32x32 outputs); here
// it is 38, the weights (76 x 64 floats) live in REGISTERS for the life of a persistent block (A operand: lane = output
// channel, k = 2*step + lane/32), and the B operand is one ds_read_b32 per MFMA pair straight from the zero-padded halo
// tile with an immediate offset: the two half-waves of a lane group need the patch elements k and k+1, whose distance
// is +1 column, +1 row - 4 columns or +1 plane - 4 rows - 4 columns depending only on the (compile-time) step, so three
// per-lane address registers (one per distance) cover every step.
//
// Block = 4 waves = one 8 x 32 pixel tile of one image; wave w owns rows 2w, 2w+1 (two 32-pixel MFMA column tiles) x 64
// output channels (two row tiles): 64 accumulator registers.  Persistent blocks (two per CU) walk the tiles; the next
// tile's halo is prefetched into registers during the MFMA phase.  Epilogue: bias, fp32 NCHW stores, per-tile
// {sum, sumsq} partials for the BatchNorm that follows the stem.
#include "common.h"

struct Conv5K75Args {
  const float* x;   // [B][Cs][H][W], Cs <= 3
  const float* wq;  // packed [76][64]: wq[k][co], zero padded
  float* y;         // [B][Co][H][W], Co <= 64
  const float* bias;
  float* stats;     // [n_items][Co][2] or null
  int B, Cs, Co, H, W;
  int nrow8, ncol32, n_items;
};

namespace {
constexpr int K75_TH = 8, K75_TW = 32, K75_LH = 12, K75_LW = 36, K75_PL = K75_LH * K75_LW;  // 432 floats per plane
constexpr int K75_NX = 3 * K75_PL;                                                            // staged floats
constexpr int K75_BUF = 4 * K75_PL;  // + one all-zero plane behind them (the padded column k = 75 reads into it)
constexpr int K75_NQ = (K75_NX + 255) / 256;  // staged elements per thread (6)
constexpr int K75_TS = 36;                    // row stride (floats) of a wave's output-transpose tile: 16-byte aligned rows
}  // namespace

__global__ void __launch_bounds__(256, 2) conv5_k75_kernel(Conv5K75Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;                   // [2][K75_BUF]
  float* red = smem + 2 * K75_BUF;    // [4 waves][64][2]
  float* trs = red + 4 * 64 * 2;      // [4 waves][32][K75_TS] output-transpose tiles
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, kk = lane >> 5;
  const int H = a.H, W = a.W, HW = H * W;

  // ---- weights: A[m][s] = W[m*32 + l31][2s + kk]
  float A[2][38];
#pragma unroll
  for (int s = 0; s < 38; ++s)
#pragma unroll
    for (int m = 0; m < 2; ++m) A[m][s] = a.wq[(2 * s + kk) * 64 + m * 32 + l31];

  // ---- the zero plane of both buffers
  for (int i = tid; i < K75_PL; i += 256) {
    xs[3 * K75_PL + i] = 0.f;
    xs[K75_BUF + 3 * K75_PL + i] = 0.f;
  }

  // ---- staging map (tile-invariant): element tid + q*256 of [3][LH][LW] -> coordinates and offset relative to the
  // tile origin ((ci*H + r)*W + c is linear in (r0, c0))
  int xrel[K75_NQ], xcrd[K75_NQ];
#pragma unroll
  for (int q = 0; q < K75_NQ; ++q) {
    const int pos = tid + q * 256;
    xcrd[q] = -1;
    xrel[q] = 0;
    if (pos < K75_NX) {
      const int ci = pos / K75_PL, rem = pos - ci * K75_PL;
      const int rr = rem / K75_LW, cc = rem - rr * K75_LW;
      if (ci < a.Cs) {
        xcrd[q] = rr | (cc << 8);
        xrel[q] = ((ci * H + rr - 2) * W + cc - 2) * 4;
      }
    }
  }

  // ---- B operand addresses (bytes) of the wave's two pixel rows; the upper half-wave is shifted by the distance
  // between patch elements k and k+1: A = next column, B = next row, C = next plane
  unsigned adrA[2], adrB[2], adrC[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int base = ((2 * wave + t) * K75_LW + l31) * 4;
    adrA[t] = (unsigned)(base + kk * 4);
    adrB[t] = (unsigned)(base + kk * (K75_LW - 4) * 4);
    adrC[t] = (unsigned)(base + kk * (K75_PL - 4 * K75_LW - 4) * 4);
  }

  float xr[K75_NQ];
  int b, r0, c0;
  __amdgpu_buffer_rsrc_t xrs;
#define K75_SETUP_LOAD(ITEM)                                                                                   \
  {                                                                                                            \
    int t_ = (ITEM);                                                                                           \
    const int cs_ = t_ % a.ncol32;                                                                             \
    t_ /= a.ncol32;                                                                                            \
    const int rg_ = t_ % a.nrow8;                                                                              \
    b = t_ / a.nrow8;                                                                                          \
    r0 = rg_ * K75_TH;                                                                                         \
    c0 = cs_ * K75_TW;                                                                                         \
    xrs = make_rsrc(a.x + (size_t)b * a.Cs * HW, (unsigned long long)a.Cs * HW * 4ull);                        \
    const int org_ = (r0 * W + c0) * 4;                                                                        \
    _Pragma("unroll") for (int q = 0; q < K75_NQ; ++q) {                                                       \
      const int crd = xcrd[q];                                                                                 \
      const int r = r0 + (crd & 255) - 2, c = c0 + (crd >> 8) - 2;                                             \
      const bool ok = crd >= 0 && r >= 0 && r < H && c >= 0 && c < W;                                          \
      xr[q] = buf_load_f32(xrs, ok ? (unsigned)(org_ + xrel[q]) : SIVAE_OOB, 0u);                              \
    }                                                                                                          \
  }

  int item = blockIdx.x;
  if (item < a.n_items) K75_SETUP_LOAD(item)
  int buf = 0;
  while (item < a.n_items) {
    float* xb = xs + buf * K75_BUF;
#pragma unroll
    for (int q = 0; q < K75_NQ; ++q) {
      const int pos = tid + q * 256;
      if (pos < K75_NX) xb[pos] = xr[q];  // (buffer loads outside the image / channel range returned 0)
    }
    const int e_b = b, e_r0 = r0, e_c0 = c0, e_item = item;
    __syncthreads();
    const int next = item + (int)gridDim.x;
    if (next < a.n_items) K75_SETUP_LOAD(next)

    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;
    const unsigned char* xbb = reinterpret_cast<const unsigned char*>(xb);
#pragma unroll
    for (int s = 0; s < 38; ++s) {
      const int k0 = 2 * s;
      const int ci0 = k0 / 25, kh0 = (k0 % 25) / 5, kw0 = k0 % 5;
      const int off = ((ci0 * K75_LH + kh0) * K75_LW + kw0) * 4;
      float bv[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const unsigned ad = kw0 < 4 ? adrA[t] : (kh0 < 4 ? adrB[t] : adrC[t]);
        bv[t] = *reinterpret_cast<const float*>(xbb + ad + off);
      }
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[m][s], bv[t], acc[m][t], 0, 0, 0);
    }

    // ---- epilogue: acc[m][t][r] = output channel m*32 + (r&3) + 8*(r>>2) + 4*kk of pixel (row 2*wave + t, column l31)
    const __amdgpu_buffer_rsrc_t yrs = make_rsrc(a.y + (size_t)e_b * a.Co * HW, (unsigned long long)a.Co * HW * 4ull);
    const int col = e_c0 + l31;
    bool okp[2];
    unsigned yo[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = e_r0 + 2 * wave + t;
      okp[t] = row < H && col < W;
      yo[t] = okp[t] ? (unsigned)(row * W + col) * 4u : SIVAE_OOB;
    }
    const bool want_stats = a.stats != nullptr;
    const bool vec = (W & 3) == 0;  // (block-uniform) rows are 16-byte aligned: 16-byte stores through an LDS transpose
    float* tw = trs + wave * (32 * K75_TS);  // this wave's [32 channels][32 pixels] transpose tile
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      float bias[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
        bias[r] = (a.bias != nullptr && co < a.Co) ? a.bias[co] : 0.f;
      }
      if (want_stats) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
          float s = 0.f, q = 0.f;
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const float vm = okp[t] ? acc[m][t][r] + bias[r] : 0.f;
            s += vm;
            q += vm * vm;
          }
          s = half_wave_sum_hi(s);
          q = half_wave_sum_hi(q);
          if (l31 == 31) {
            red[(wave * 64 + co) * 2 + 0] = s;
            red[(wave * 64 + co) * 2 + 1] = q;
          }
        }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (vec) {
          // The accumulator layout gives a lane ONE pixel of 16 channels: 64 dword stores per lane and tile, which is
          // store-issue bound (as long as the tile's MFMA phase).  Through the wave's LDS tile a lane gets 4 consecutive
          // pixels of one channel instead: 16 sixteen-byte stores.
#pragma unroll
          for (int r = 0; r < 16; ++r)
            tw[((r & 3) + 8 * (r >> 2) + 4 * kk) * K75_TS + l31] = acc[m][t][r] + bias[r];
          const int row = e_r0 + 2 * wave + t;
          typedef float f32x4p_t __attribute__((ext_vector_type(4)));
          f32x4p_t fv_prev = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int col_l = (lane >> 3) + 8 * i, c4 = (lane & 7) * 4;
            const float4 v = *reinterpret_cast<const float4*>(tw + col_l * K75_TS + c4);
            const int co = m * 32 + col_l;
            const bool ok = co < a.Co && row < H && e_c0 + c4 < W;
            const unsigned off = ok ? (unsigned)((co * H + row) * W + e_c0 + c4) * 4u : SIVAE_OOB16;  // (16-byte store)
            typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
            typedef float f32x4_t __attribute__((ext_vector_type(4)));
            f32x4_t fv = {v.x, v.y, v.z, v.w};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, fv), yrs, (int)off, 0, 0);
            SIVAE_PIN4(fv_prev)  // (store-data lifetime: common.h)
            fv_prev = fv;
          }
          SIVAE_PIN4(fv_prev)
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
            const bool cok = co < a.Co;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[m][t][r] + bias[r]), yrs,
                                                  (int)((cok && okp[t]) ? yo[t] + (unsigned)co * (unsigned)HW * 4u : SIVAE_OOB),
                                                  0, 0);
          }
        }
      }
    }
    if (want_stats) {
      __syncthreads();
      if (tid < 64 && tid < a.Co) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          s += red[(w * 64 + tid) * 2 + 0];
          q += red[(w * 64 + tid) * 2 + 1];
        }
        float* dst = a.stats + ((size_t)e_item * a.Co + tid) * 2;
        dst[0] = s;
        dst[1] = q;
      }
      // (the next tile's LDS traffic is its halo store into the OTHER buffer and a barrier: red is safe until then)
    }
    buf ^= 1;
    item = next;
  }
#undef K75_SETUP_LOAD
}

// ---- weight pack: wq[k][o], k = c*25 + kh*5 + kw (c = the narrow side's channel), o < 64 (zero padded), 76 rows
//   mode 0 (forward of a Cs -> Cb conv):        w [Cb][Cs][5][5]:  wq[k][o] = w[o][c][kh][kw]
//   mode 1 (data gradient of a Cb -> Cs conv):  w [Cs][Cb][5][5]:  wq[k][o] = w[c][o][4-kh][4-kw]
__global__ void __launch_bounds__(256) pack5_k75_kernel(const float* __restrict__ w, float* __restrict__ wq, int Cs,
                                                        int Cb, int mode) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 76 * 64) return;
  const int k = i / 64, o = i % 64;
  float v = 0.f;
  if (k < 75 && o < Cb) {
    const int c = k / 25, kh = (k % 25) / 5, kw = k % 5;
    if (c < Cs)
      v = mode == 0 ? w[(((size_t)o * Cs + c) * 5 + kh) * 5 + kw] : w[(((size_t)c * Cb + o) * 5 + (4 - kh)) * 5 + (4 - kw)];
  }
  wq[i] = v;
}

extern "C" size_t sivae_pack_conv5_k75_bytes(void) { return (size_t)76 * 64 * sizeof(float); }

extern "C" int sivae_pack_conv5_k75(const float* w, float* wq, int n_small, int n_big, int mode, hipStream_t stream) {
  if (!w || !wq) return SIVAE_ERR_NULL;
  if (n_small <= 0 || n_small > 3 || n_big <= 0 || n_big > 64) return SIVAE_ERR_SHAPE;
  if (mode != 0 && mode != 1) return SIVAE_ERR_MODE;
  hipLaunchKernelGGL(pack5_k75_kernel, dim3(cdiv(76 * 64, 256)), dim3(256), 0, stream, w, wq, n_small, n_big, mode);
  return sivae_launch_status();
}

extern "C" int sivae_conv5_k75_supported(int Ci, int Co) { return (Ci >= 1 && Ci <= 3 && Co >= 1 && Co <= 64) ? 1 : 0; }

extern "C" int sivae_conv5_k75_num_px_tiles(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return SIVAE_ERR_SHAPE;
  const long long n = (long long)B * cdiv(H, K75_TH) * cdiv(W, K75_TW);
  return n > 0x7fffffffLL ? SIVAE_ERR_RANGE : (int)n;
}

extern "C" int sivae_conv5_k75_fwd(const float* x, const float* wq, float* y, const float* bias, float* stats_partial,
                                   int B, int Ci, int Co, int H, int W, hipStream_t stream) {
  if (!x || !wq || !y) return SIVAE_ERR_NULL;
  if (B <= 0 || H <= 0 || W <= 0) return SIVAE_ERR_SHAPE;
  if (!sivae_conv5_k75_supported(Ci, Co)) return SIVAE_ERR_SHAPE;
  if ((long long)Co * H * W * 4 >= 0xfffffff0LL || (long long)Ci * H * W * 4 >= 0x7fffffffLL) return SIVAE_ERR_RANGE;
  Conv5K75Args a;
  a.x = x;
  a.wq = wq;
  a.y = y;
  a.bias = bias;
  a.stats = stats_partial;
  a.B = B;
  a.Cs = Ci;
  a.Co = Co;
  a.H = H;
  a.W = W;
  a.nrow8 = cdiv(H, K75_TH);
  a.ncol32 = cdiv(W, K75_TW);
  const int n = sivae_conv5_k75_num_px_tiles(B, H, W);
  if (n < 0) return n;
  a.n_items = n;
  int cus = 256;
  {
    static int cached = 0;
    if (cached == 0) {
      int dev = 0;
      hipDeviceProp_t prop;
      cached = 256;
      if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        cached = prop.multiProcessorCount;
    }
    cus = cached;
  }
  const int grid = n < 2 * cus ? n : 2 * cus;
  const size_t lds = (size_t)(2 * K75_BUF + 4 * 64 * 2 + 4 * 32 * K75_TS) * sizeof(float);
  hipLaunchKernelGGL(conv5_k75_kernel, dim3((unsigned)grid), dim3(256), lds, stream, a);
  return sivae_launch_status();
}
