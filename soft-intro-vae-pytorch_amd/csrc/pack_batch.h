// Batched weight packing (round 4): after an optimizer step every cached operand form of every weight of a network is
// rebuilt — direct pack, Winograd F(2x2,3x3) / F(4x4,3x3) transforms (forward + data-gradient modes), the two
// upsample-phase transforms — ~35 launches of 5-11 us per network that a 16-image shard pays in full.  A JOB TABLE in
// device memory (built once: the buffers are reused from step to step) lets ONE launch per operand form rebuild all
// weights of that form: block -> job through a uint16 map, the job's blocks walk its elements with the job's own stride.
#pragma once
#include "common.h"

struct SivaePackJob {
  const float* w;  // the weight [Co][Ci][k][k]
  float* dst;      // the packed operand
  int Co, Ci, mode, taps;
  int kdim, ndim, kpad, npad;
  unsigned blk0, nblk;  // this job's blocks are [blk0, blk0 + nblk) of the batch launch
  unsigned long long total;
  int aux;  // form-specific (bf16 operand slabs: number of input-channel chunks)
};

#define SIVAE_PACK_DIRECT 0
#define SIVAE_PACK_WINO 1
#define SIVAE_PACK_WINO4 2
#define SIVAE_PACK_WINO_UP 3
#define SIVAE_PACK_WINO_UP_DGRAD 4
#define SIVAE_PACK_WINO4_B6 5
#define SIVAE_PACK_BF16 6  // bf16 MFMA-operand slabs of the bf16 mode (bf16_conv.hip; dst is bf16, ks may be the code 51)
#define SIVAE_PACK_NTYPES 7

static inline unsigned sivae_pack_job_blocks(unsigned long long total) {
  unsigned long long nb = (total + 511) / 512;  // ~2 elements (weight pairs: 9 loads, 16-48 stores each) per thread
  if (nb < 1) nb = 1;
  if (nb > 1024) nb = 1024;
  return (unsigned)nb;
}

// per operand form, next to the kernels: fill the shape fields of a job (returns SIVAE_OK or an error code) / launch
int sivae_packjob_direct(SivaePackJob* j, int Co, int Ci, int ks, int mode);
int sivae_packjob_wino(SivaePackJob* j, int Co, int Ci, int mode);
int sivae_packjob_wino4(SivaePackJob* j, int Co, int Ci, int mode);
int sivae_packjob_wino4_b6(SivaePackJob* j, int Co, int Ci, int mode);
int sivae_packjob_wino_up(SivaePackJob* j, int Co, int Ci);
int sivae_packjob_wino_up_dgrad(SivaePackJob* j, int Co, int Ci);
int sivae_packjob_bf16(SivaePackJob* j, int Co, int Ci, int ks, int mode);
void sivae_packbatch_direct(const SivaePackJob* jobs, const unsigned short* block_job, int nblocks, hipStream_t s);
void sivae_packbatch_wino(const SivaePackJob* jobs, const unsigned short* block_job, int nblocks, hipStream_t s);
void sivae_packbatch_wino4(const SivaePackJob* jobs, const unsigned short* block_job, int nblocks, hipStream_t s);
void sivae_packbatch_wino4_b6(const SivaePackJob* jobs, const unsigned short* block_job, int nblocks, hipStream_t s);
void sivae_packbatch_wino_up(const SivaePackJob* jobs, const unsigned short* block_job, int nblocks, hipStream_t s);
void sivae_packbatch_wino_up_dgrad(const SivaePackJob* jobs, const unsigned short* block_job, int nblocks, hipStream_t s);
void sivae_packbatch_bf16(const SivaePackJob* jobs, const unsigned short* block_job, int nblocks, hipStream_t s);
