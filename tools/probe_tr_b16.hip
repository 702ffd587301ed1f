// Hardware check of the ds_read_b64_tr_b16 addressing the bf16 weight-gradient kernel relies on (bf16_wgrad.hip):
// LDS image [position][32 channels] bf16 (64 bytes per position); lane (i16 = lane&15, grp = (lane>>4)&1, hh = lane>>5)
// supplies the address of position 8*hh + (i16>>2), channels 16*grp + 4*(i16&3) .. +3, and must receive the FOUR
// positions 8*hh + 0..3 of channel 16*grp + i16.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[64 * 32];
  for (int i = threadIdx.x; i < 64 * 32; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int lane = threadIdx.x, hh = lane >> 5, grp = (lane >> 4) & 1, i16 = lane & 15;
  const int pos = 8 * hh + (i16 >> 2);
  const char* base = (const char*)lds;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (s16x4 __attribute__((address_space(3)))*)(base + pos * 64 + (16 * grp + 4 * (i16 & 3)) * 2));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}
int main() {
  short* d;
  hipMalloc(&d, 256 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short h[256];
  hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane) {
    const int hh = lane >> 5, grp = (lane >> 4) & 1, i16 = lane & 15;
    for (int e = 0; e < 4; ++e) {
      const int want = (8 * hh + e) * 32 + 16 * grp + i16;
      if (h[lane * 4 + e] != want) ++bad;
    }
  }
  printf("tr_b16 probe: %d mismatches of 256\n", bad);
  for (int lane = 0; lane < 64; lane += 1)
    if (lane < 20 || (lane & 15) == 0) printf("lane %2d: %4d %4d %4d %4d\n", lane, h[lane * 4], h[lane * 4 + 1], h[lane * 4 + 2], h[lane * 4 + 3]);
  return bad != 0;
}
