// Device helpers shared by the two bf16 encoder GEMM kernels of the hot path (encoder_gemm_fast.hip: 32x32x16 MFMAs,
// encoder_gemm_fast16.hip: 16x16x32 MFMAs): LDS-DMA from inline asm, accumulator reads, tile numbering.
#pragma once
#include "gemm_fast.h"

namespace clt {
namespace {

__device__ uint4 g_zero16[1];  // 16 zero bytes: source for absent bias chunks

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int ROWB = 64;                    // bytes of K per row per k-step (32 bf16)
constexpr int OP_BYTES = GEMM_TM * ROWB;    // 16 KiB per operand per stage
constexpr int NSTAGE = 4;
constexpr int AUX_OFF = 2 * OP_BYTES;     // per stage and wave 768 B: [0, 256) bias of 64 latents, [256, 768) tau of 128 tokens
constexpr int AUX_WAVE = 768;
constexpr int STAGE_BYTES = 2 * OP_BYTES + 4 * AUX_WAVE;
constexpr int NTHREADS = 256;
constexpr int NDMA = 9;                     // LDS-DMAs per wave per stage: 4 W pieces, 4 x pieces, 1 bias/tau piece
constexpr int QLANE = 10;                   // filter mode: 8-byte hit-queue entries per LANE and tile (5 KiB per wave)
constexpr int NSTORE = 64;                  // store instructions per wave in the epilogue of an interior tile

#define CLT_FENCE() __builtin_amdgcn_sched_barrier(0)

// LDS[m0 + 16 * lane] = *(sbase + voff): SGPR base, 32-bit lane offset.  Inline asm so that hipcc neither
// counts nor drains it (see encoder_gemm.hip).
__device__ __forceinline__ void dma_s(const char* sbase, uint32_t voff, uint32_t lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}
__device__ __forceinline__ void dma_v(const char* src, uint32_t lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(src), "s"(lds_dst)
      : "memory");
}

__device__ __forceinline__ f32x16 mfma(const uint4& a, const uint4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0,
                                                 0);
}

// Accumulator element -> VGPR at the point of use.  Volatile asm, so hipcc cannot hoist the 256 reads of an
// epilogue to its top (which needs 256 more VGPRs than there are and spills the loop-carried fragments).
__device__ __forceinline__ float acc_read(float a) {
  float r;
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r) : "a"(a));
  return r;
}

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Latent tiles can be a strided subset of W's 256-row tiles, so that one kernel serves the three launches of
// the fused encode+top-k (gemm_fast.h): virtual tile v -> actual tile.
__host__ __device__ __forceinline__ int lt_actual(int v, int mode, int S) {
  if (mode == LT_SAMPLE) return v * S;                 // every S-th tile
  if (mode == LT_REST) return v + v / (S - 1) + 1;     // all the others
  return v;
}

// The 16x16x32 kernels (encoder_gemm_fast16.hip, encoder_gemm_w8.hip) address W in GROUPS of 16 latent rows -- one
// group is one LDS-DMA piece and one MFMA row tile -- and a latent tile is 16 consecutive VIRTUAL groups:
//   LT_ALL      virtual group v = group v of W;
//   LT_GSAMPLE  the sample of the fused encode + top-k: every S-th group of W, v -> v S + S - 1;
//   LT_GREST    all the others, v -> v + v / (S - 1).
// Sampling ROW GROUPS inside every 256-row tile (not every S-th tile, as the 32x32x16 kernel does) keeps the sample
// representative when the latents of a tile fire together -- trained dictionaries are correlated like that.
__host__ __device__ __forceinline__ int grp_actual(int v, int mode, int S) {
  if (mode == LT_GSAMPLE) return v * S + S - 1;
  if (mode == LT_GREST) return v + v / (S - 1);
  return v;
}

}  // namespace
}  // namespace clt
