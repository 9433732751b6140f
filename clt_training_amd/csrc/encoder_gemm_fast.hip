// Encoder GEMM of the hot path, bf16 operands: pre[N, M] = relu(x[N, d] @ W[M, d]^T + b) on gfx950.
//
// Same product orientation as the generic kernel (encoder_gemm.hip): D[latent][token], W is the MFMA A operand,
// x the B operand, so a lane owns ONE token and runs of 4 consecutive latents.
//
// Structure (persistent, one 256-thread workgroup per CU, ONE wave per SIMD with the whole 512-register file):
//   tile 256 latents x 256 tokens; 4 waves as 2 (latent) x 2 (token), each wave 128 x 128 = 4 x 4 MFMA tiles
//   (v_mfma_f32_32x32x16_bf16), 256 accumulator registers;  per 32-wide k-step a wave issues 32 MFMAs,
//   16 ds_read_b128 and 9 LDS-DMAs (vs 16 / 12 / 5 for the 8-wave layout: half the LDS reads per flop);
//   staging by global_load_lds_dwordx4 (SGPR tile base + 32-bit lane offset: no address VALU in the loop) into a
//   4-deep ring of lane-linear LDS images, bank swizzle chunk ^= (row >> 2) & 3 on the source address and on
//   the ds_read address;
//   the bias is the MFMA C operand of a tile's first k-step (staged through LDS with the operands), so the
//   accumulators are never cleared and the epilogue has no bias add;
//   the instruction stream of a k-step is laid out by hand (sched_barrier fences): MFMAs back to back with
//   one ds_read or one LDS-DMA in the shadow of each; fragments of kk=1 are read during the MFMAs of kk=0, the
//   fragments of the NEXT stage's kk=0 during kk=1, so no MFMA waits on LDS latency across a stage boundary;
//   stage s+1 is certified (counted s_waitcnt vmcnt + s_barrier) early in stage s, one stage before its first
//   read; the same barrier frees the slot of stage s-1 for the LDS-DMAs of stage s+3.
// MFMA-bound: 2 * N * M * d flops per launch.
#include "gemm_fast.h"

namespace clt {
namespace {

__device__ uint4 g_zero16[1];  // 16 zero bytes: source for absent bias chunks

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int ROWB = 64;                    // bytes of K per row per k-step (32 bf16)
constexpr int OP_BYTES = GEMM_TM * ROWB;    // 16 KiB per operand per stage
constexpr int NSTAGE = 4;
constexpr int BIAS_OFF = 2 * OP_BYTES;
constexpr int STAGE_BYTES = 2 * OP_BYTES + 1024;
constexpr int NTHREADS = 256;
constexpr int NDMA = 9;                     // LDS-DMAs per wave per stage: 4 W pieces, 4 x pieces, 1 bias piece
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

template <typename OT>
__global__ __launch_bounds__(NTHREADS, 1) void gemm_fast_kernel(const char* __restrict__ xb, const char* __restrict__ Wb,
                                                                 const char* __restrict__ bias, int bias_esize,
                                                                 OT* __restrict__ pre, int64_t N, int64_t M, int kbytes,
                                                                 int nlt, int ntt, int dbg) {
  __shared__ __attribute__((aligned(1024))) char lds[NSTAGE * STAGE_BYTES];
  const int lane = threadIdx.x & 63;
  const int wave = uni(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int r32 = lane & 31, h = lane >> 5;

  // ---- this workgroup's tile ids: XCD group xg = blockIdx % 8 owns a contiguous id range and its workgroups
  //      walk it with stride nbx.  All of it is wave-uniform 32-bit arithmetic (host checks the ranges).
  const int ntiles = nlt * ntt;
  const int xg = blockIdx.x & 7, bi = blockIdx.x >> 3;
  const int nbx = ((int)gridDim.x - xg + 7) >> 3;
  const int qt = ntiles >> 3, rr = ntiles & 7;
  const int xstart = xg < rr ? xg * (qt + 1) : rr * (qt + 1) + (xg - rr) * qt;
  const int xcount = qt + (xg < rr ? 1 : 0);
  const int my_tiles = bi < xcount ? uni((xcount - bi + nbx - 1) / nbx) : 0;
  if (my_tiles == 0) return;

  const int nsteps = kbytes / ROWB;
  const int total = my_tiles * nsteps;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)lds;
  const char* zero = reinterpret_cast<const char*>(g_zero16);

  auto coords = [&](int id, int& lt, int& tt) {  // see tile_of (gemm_fast.h)
    const int full = nlt * GEMM_PANEL;
    const int p = id / full, rem = id - p * full;
    const int left = ntt - p * GEMM_PANEL;
    const int pt = left < GEMM_PANEL ? left : GEMM_PANEL;
    lt = uni(rem / pt);
    tt = uni(p * GEMM_PANEL + rem % pt);
  };

  // ---- loader state: runs three stages ahead of the multiplier, possibly in the next tile.  Past the last
  //      stage it keeps re-fetching the last tile into ring slots nobody reads again, so that every k-step issues
  //      the same NDMA operations and the counted waits need no special cases.
  int ld_t = 0, ld_ks = 0, ld_slot = 0;
  const char* ldW = Wb;
  const char* ldX = xb;
  const char* ldB = zero;
  uint32_t voffW[4], voffX[4];
  const int nl = 64 * bias_esize / 16;  // lanes of the bias piece (this wave's 64 latents)
  auto loader_tile = [&](int lt, int tt) {
    const int64_t lat0 = (int64_t)lt * GEMM_TM, tok0 = (int64_t)tt * GEMM_TN;
    ldW = Wb + lat0 * kbytes;
    ldX = xb + tok0 * kbytes;
    // rows past the end re-read the last row (their products are never stored)
    const int maxW = (int)(M - 1 - lat0 < GEMM_TM ? M - 1 - lat0 : GEMM_TM), maxX = (int)(N - 1 - tok0 < GEMM_TN ? N - 1 - tok0 : GEMM_TN);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = (wave + 4 * q) * 16 + (lane >> 2);
      const int c = (lane & 3) ^ ((r >> 2) & 3);
      voffW[q] = (uint32_t)((r < maxW ? r : maxW) * kbytes + c * 16);
      voffX[q] = (uint32_t)((r < maxX ? r : maxX) * kbytes + c * 16);
    }
    const int epc = 16 / bias_esize;
    const int64_t lat = lat0 + 64 * wave + lane * epc;
    ldB = (bias && lat + epc <= M) ? bias + lat * bias_esize : zero;
  };
  auto dma = [&](int q, uint32_t stage_lds, int kb0) {
    if (q < 4) {
      dma_s(ldW + kb0, voffW[q], stage_lds + (wave + 4 * q) * 1024);
    } else if (q < 8) {
      dma_s(ldX + kb0, voffX[q - 4], stage_lds + OP_BYTES + (wave + 4 * (q - 4)) * 1024);
    } else {
      if (lane < nl) dma_v(ldB, stage_lds + BIAS_OFF + wave * 64 * bias_esize);
    }
  };
  auto loader_advance = [&]() {
    ld_slot = (ld_slot + 1) & 3;
    if (++ld_ks == nsteps) {
      ld_ks = 0;
      if (++ld_t < my_tiles) {
        int lt, tt;
        coords(xstart + bi + ld_t * nbx, lt, tt);
        loader_tile(lt, tt);
      }
    }
  };

  // ---- fragment addresses: row * 64 + ((2 kk + h) ^ swizzle) * 16; the swizzle (row >> 2) & 3 is a lane constant
  const int swz = (r32 >> 2) & 3;
  const uint32_t fa = (uint32_t)((wm * 128 + r32) * ROWB);
  const uint32_t fb = (uint32_t)(OP_BYTES + (wn * 128 + r32) * ROWB);
  const uint32_t pk[2] = {(uint32_t)(((0 + h) ^ swz) * 16), (uint32_t)(((2 + h) ^ swz) * 16)};
  // piece p of a (stage, kk) fragment set, in the order the MFMAs need them: A0 B0 B1 B2 B3 A1 A2 A3
  auto read_piece = [&](const char* st, int kk, int p, uint4 (&A)[4], uint4 (&B)[4]) {
    if (p == 0) A[0] = *reinterpret_cast<const uint4*>(st + fa + pk[kk]);
    else if (p <= 4) B[p - 1] = *reinterpret_cast<const uint4*>(st + fb + (p - 1) * 32 * ROWB + pk[kk]);
    else A[p - 4] = *reinterpret_cast<const uint4*>(st + fa + (p - 4) * 32 * ROWB + pk[kk]);
  };

  f32x16 acc[4][4];
  uint4 A0[4], B0[4], A1[4], B1[4];

  // The bias enters as three extra K columns: A = (b_hi, b_mid, b_lo, 0...) of this lane's latent, B = (1, 1, 1,
  // 0...): bf16 x 1 products are exact and the three terms carry all 24 mantissa bits of an f32 bias, so a tile's
  // accumulators start at the bias without register moves (C and D of an MFMA share a register class) and the
  // epilogue has no bias add.  st: a landed stage of the tile (every stage carries the tile's bias).
  auto init_acc = [&](const char* st) {
    const uint4 ones = h == 0 ? make_uint4(0x3f803f80u, 0x00003f80u, 0u, 0u) : make_uint4(0u, 0u, 0u, 0u);
    uint4 Ab[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int off = wm * 128 + i * 32 + r32;
      float b;
      if (bias_esize == 4) b = *reinterpret_cast<const float*>(st + BIAS_OFF + off * 4);
      else b = bf16_bits_to_f32(*reinterpret_cast<const uint16_t*>(st + BIAS_OFF + off * 2));
      const uint32_t u0 = __float_as_uint(b) & 0xffff0000u;  // truncation splits: every remainder is exact
      const float r1 = b - __uint_as_float(u0);
      const uint32_t u1 = __float_as_uint(r1) & 0xffff0000u;
      const float r2 = r1 - __uint_as_float(u1);
      const uint32_t u2 = __float_as_uint(r2) & 0xffff0000u;
      Ab[i] = h == 0 ? make_uint4((u0 >> 16) | u1, u2 >> 16, 0u, 0u) : make_uint4(0u, 0u, 0u, 0u);
    }
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint4 oj = ones;
      asm volatile("" : "+v"(oj.x));  // 16 independent MFMAs, not 4 plus 192 accumulator moves
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = mfma(Ab[i], oj, zero16);
    }
  };

  // ---- prologue: three stages in flight, all landed before the first MFMA (once per launch)
  int cur_lt, cur_tt;
  coords(xstart + bi, cur_lt, cur_tt);
  loader_tile(cur_lt, cur_tt);
#pragma unroll 1
  for (int p = 0; p < 3; ++p) {
    const uint32_t st = lds0 + (uint32_t)ld_slot * STAGE_BYTES;
#pragma unroll
    for (int q = 0; q < NDMA; ++q) dma(q, st, ld_ks * ROWB);
    loader_advance();
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
  for (int p = 0; p < 8; ++p) read_piece(lds, 0, p, A0, B0);
  init_acc(lds);

  int ks = 0, t_idx = 0;
  int stores_behind = 0;  // k-steps (2, 1) for which the NSTORE stores of the last epilogue are still among the
                          // youngest operations of the vmcnt queue
#pragma unroll 1
  for (int s = 0; s < total; ++s) {
    const char* cb = lds + (s & 3) * STAGE_BYTES;
    const char* nb = lds + ((s + 1) & 3) * STAGE_BYTES;
    const uint32_t ld_stage = lds0 + (uint32_t)ld_slot * STAGE_BYTES;
    const int kb0 = ld_ks * ROWB;
    // Stage s+1 has landed once only the operations issued after its LDS-DMAs are outstanding: the NDMA DMAs of
    // stage s+2, plus the epilogue's NSTORE stores while they are younger than the DMAs of stage s+1.
    // vmcnt(n) is safe for any n <= that count (the counter saturates at 63).
    const bool many = stores_behind > 0;
    const bool last_k = ks + 1 == nsteps;

    {
      CLT_FENCE();
#pragma unroll
      for (int t = 0; t < 32; ++t) {
        const int kk = t >> 4, i = (t >> 2) & 3, j = t & 3;
        if (kk == 0) acc[i][j] = mfma(A0[i], B0[j], acc[i][j]);
        else acc[i][j] = mfma(A1[i], B1[j], acc[i][j]);
        CLT_FENCE();
        if (t == 1) {
          if (many) asm volatile("s_waitcnt vmcnt(63)\n\ts_barrier" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(9)\n\ts_barrier" ::: "memory");
        }
        if (t < 8) read_piece(cb, 1, t, A1, B1);
        if (t >= 16 && t < 24) read_piece(nb, 0, t - 16, A0, B0);
        if (t % 3 == 2 && t <= 26) dma((t - 2) / 3, ld_stage, kb0);
        CLT_FENCE();
      }
    }
    loader_advance();
    if (stores_behind > 0) --stores_behind;

    // ---- epilogue at the end of a tile: ReLU, one rounding, 4 consecutive latents per store
    if (last_k) {
      const int64_t lat0 = (int64_t)cur_lt * GEMM_TM, tok0 = (int64_t)cur_tt * GEMM_TN;
      const bool interior = tok0 + GEMM_TN <= N && lat0 + GEMM_TM <= M && (M & 3) == 0;
      if (!(dbg & 1)) {
        // One code path for interior and edge tiles (two paths make hipcc hoist all 256 accumulator reads above
        // the branch, which spills).  M is a multiple of 8, so a run of 4 latents starting at a multiple of 4 is
        // inside or outside as a whole and the test does not depend on the lane half.
        const int lat_left = (int)(M - lat0 - wm * 128 < 128 ? M - lat0 - wm * 128 : 128);  // wave-uniform
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int64_t tok = tok0 + wn * 128 + j * 32 + r32;
          const bool tok_ok = tok < N;
          OT* prow = pre + tok * M + lat0 + wm * 128 + 4 * h;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              if (tok_ok && i * 32 + 8 * g < lat_left) {
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = fmaxf(acc_read(acc[i][j][4 * g + q]), 0.f);
                if constexpr (sizeof(OT) == 2) {
                  uint2 u;
                  u.x = pack_bf16x2(v[0], v[1]);
                  u.y = pack_bf16x2(v[2], v[3]);
                  *reinterpret_cast<uint2*>(prow + i * 32 + 8 * g) = u;
                } else {
                  *reinterpret_cast<float4*>(prow + i * 32 + 8 * g) = make_float4(v[0], v[1], v[2], v[3]);
                }
              }
            }
            CLT_FENCE();
          }
        }
        if (interior) stores_behind = 2;
      }
      ks = 0;
      if (++t_idx < my_tiles) {
        coords(xstart + bi + t_idx * nbx, cur_lt, cur_tt);
        init_acc(nb);  // stage s+1, certified during this k-step, is the next tile's first
      }
    } else {
      ++ks;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the loader's trailing fetches
}

}  // namespace

bool gemm_fast_supported(int in_dtype, int64_t N, int64_t M, int d) {
  const int64_t kbytes = (int64_t)d * 2;
  const int64_t nlt = (M + GEMM_TM - 1) / GEMM_TM, ntt = (N + GEMM_TN - 1) / GEMM_TN;
  return in_dtype == CLT_BF16 && N > 0 && kbytes % ROWB == 0 && kbytes >= ROWB && (M & 7) == 0 &&
         kbytes * GEMM_TM < (1ll << 31) && nlt * ntt * (kbytes / ROWB) < (1ll << 30);
}

int launch_gemm_fast(const void* x, const void* W, const void* bias, int bias_dtype, void* pre, int out_dtype,
                     int64_t N, int64_t M, int d, int grid, int dbg, hipStream_t st) {
  const int64_t nlt = (M + GEMM_TM - 1) / GEMM_TM, ntt = (N + GEMM_TN - 1) / GEMM_TN;
  const int esize = bias_dtype == CLT_BF16 ? 2 : 4;
  if (out_dtype == CLT_BF16)
    hipLaunchKernelGGL((gemm_fast_kernel<bf16_t>), dim3(grid), dim3(NTHREADS), 0, st, (const char*)x, (const char*)W,
                       (const char*)bias, esize, (bf16_t*)pre, N, M, d * 2, (int)nlt, (int)ntt, dbg);
  else
    hipLaunchKernelGGL((gemm_fast_kernel<float>), dim3(grid), dim3(NTHREADS), 0, st, (const char*)x, (const char*)W,
                       (const char*)bias, esize, (float*)pre, N, M, d * 2, (int)nlt, (int)ntt, dbg);
  CLT_CHECK_LAUNCH("clt_encoder_gemm_relu");
  return CLT_OK;
}

}  // namespace clt
