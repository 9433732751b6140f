// Filter pass of the fused encode + top-k (gemm_fast.h) on TWO waves per SIMD: the encoder GEMM
// pre[N, M] = relu(x[N, d] @ W[M, d]^T + b) in bf16 on v_mfma_f32_16x16x32_bf16 whose epilogue keeps only the
// accumulators that reach their token's threshold.
//
// Same workgroup tile, LDS image, LDS-DMA ring and product orientation (D[latent][token]) as the store-mode kernel of
// encoder_gemm_fast16.hip, but the 256 x 256 tile is split over EIGHT waves (512 threads, 2 latent halves x 4 token
// quarters, 128 latents x 64 tokens per wave) instead of four:
//   * a wave holds 128 accumulators instead of 256, and at 256 registers per lane hipcc keeps them in ARCHITECTURAL
//     VGPRs (no "a" operand anywhere in this file): the threshold scan reads them in place -- no v_accvgpr_read, a
//     quarter of the old scan's instructions;
//   * what one wave per SIMD could not hide under its own MFMAs -- the issue cost of the LDS-DMAs (25-40 cycles
//     each), the scan, the waits -- now runs beside the partner wave's MFMAs (MI355X_MICROARCH.md, 'Two waves per
//     SIMD'); per wave and k-step: 32 MFMAs, 12 fragment reads (8 of W, 4 of x), 4 LDS-DMAs;
//   * the price is 12 instead of 8 ds_read_b128 per 32 MFMAs: 96 KiB of fragment reads per CU and k-step, ~40 % of
//     what the LDS array delivers at the ideal MFMA rate (256 B/clk for ds_read_b128).
// The accumulation order per output element is the four-wave kernel's (bias MFMA, then the k-steps in order), so the
// accumulators -- and with them the fused path's results -- are bit-identical.
//
// What leaves the kernel is a CANDIDATE STREAM, not per-(token, group) lists: a hit (4 consecutive latents of one token
// with at least one value at the threshold) becomes one 16-byte entry {2 x bf16, 2 x bf16, token, first latent}, appended
// by the lanes that hit (rank = mbcnt of the hit mask under EXEC = hit mask, one ds_write_b128) to a wave-private LDS
// queue of QCAP8 entries; a queue holding 64 or more entries at the end of a latent row is written out as dense 1 KiB
// rows to the wave's own region of the global stream (region = 8 * workgroup + wave, FastArgs.stream_rcap entries), the
// final count to FastArgs.stream_counts.  No atomics, no per-token list that could overflow: a region that runs full or
// a queue that cannot take a row's worst case (64 more entries) flags its token tile for the fallback (fb_flags), which
// no shape of BASELINE.json and no dictionary of tools/bench_fused.py reaches.  fused_bucket_kernel
// (encoder_gemm_fast.hip) turns the stream into per-token lists for the merge.
// MFMA-bound: 2 * N * M * d flops per launch.
#include <math.h>

#include <utility>

#include "gemm_fast_dev.h"

namespace clt {
namespace {

#ifndef CLT_ABL8  // compile-time ablations for tools/ (results are garbage): 4 = no LDS-DMA in the loop, 8 = no fragment
#define CLT_ABL8 0  // reads, 16 = no certification wait, 32 = no barrier, 64 = waves 4-7 at s_setprio 1
#endif

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#ifndef CLT_PANEL8
#define CLT_PANEL8 GEMM_PANEL
#endif
constexpr int PANEL8 = CLT_PANEL8;  // token tiles per L2 panel (a power of two)
constexpr int PANEL8_LOG = PANEL8 == 16 ? 4 : PANEL8 == 8 ? 3 : PANEL8 == 4 ? 2 : PANEL8 == 2 ? 1 : -1;
static_assert(PANEL8_LOG > 0 && GEMM_PANEL % PANEL8 == 0 || PANEL8 == 16, "panel");
constexpr int NT8 = 512;     // threads: 8 waves, wave = (wm = wave >> 2: latent half, wn = wave & 3: token quarter)
constexpr int PF8 = 4;       // stages the loader runs ahead of the multiplier (= ring slots: see encoder_gemm_fast16.hip)
constexpr int STAGE8 = 2 * OP_BYTES;
constexpr int AUX8_OFF = NSTAGE * STAGE8;
constexpr int AUX8_WAVE = 256;  // a wave's aux piece: [0, 128) bias of latents 32 w .. + 31, [128, 256) tau of tokens 32 w .. + 31
constexpr int QUEUE8_OFF = AUX8_OFF + 2 * 8 * AUX8_WAVE;
constexpr int QCAP8 = 224;   // 16-byte entries in a wave's candidate queue (3.5 KiB per wave)
constexpr int LDS8_BYTES = QUEUE8_OFF + 8 * QCAP8 * 16;  // 160 KiB: the ring, the aux pieces of two tiles, the queues
constexpr int NDMA8 = 4;     // data LDS-DMAs per wave and stage: 2 W pieces, 2 x pieces
static_assert(LDS8_BYTES <= 160 * 1024, "LDS");

__device__ __forceinline__ f32x4 mfma16(const u32x4& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0,
                                                 0);
}

template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// Instruction slots of a k-step (slot t = after MFMA t): the certification wait + barrier after MFMA 1, the twelve
// fragment reads of the next step after MFMAs 2, 4, .. 24, the four LDS-DMAs after MFMAs 5, 11, 17, 23 (an operand's m0
// write goes with its first piece).
__host__ __device__ constexpr int dma8_at(int t) { return (t >= 5 && (t - 5) % 6 == 0 && (t - 5) / 6 < 4) ? (t - 5) / 6 : -1; }
__host__ __device__ constexpr int read8_at(int t) { return (t >= 2 && t % 2 == 0 && (t - 2) / 2 < 12) ? (t - 2) / 2 : -1; }

template <int OFF>
__device__ __forceinline__ u32x4 lds_read128(uint32_t addr) {
  u32x4 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
  return r;
}
__device__ __forceinline__ void set_m0(uint32_t v) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" : : "s"(v) : "memory", "m0");
}
// LDS[m0 + OFF + 16 * lane] = *(sbase + voff + OFF)
template <int OFF>
__device__ __forceinline__ void dma_m0(const char* sbase, uint32_t voff) {
  asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" : : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
// the aux piece: the bias chunks (lanes of mask_b, from base_b + voff) and the tau chunks (lanes of mask_t, from
// base_t + voff); lane l lands at lds_dst + 16 l either way (every lane of the wave is active here: uniform control flow)
__device__ __forceinline__ void dma_aux(const char* base_b, const char* base_t, uint32_t voff, uint32_t lds_dst,
                                        unsigned long long mask_b, unsigned long long mask_t) {
  asm volatile(
      "s_mov_b32 m0, %3\n\t"
      "s_mov_b64 exec, %4\n\t"
      "global_load_lds_dwordx4 %0, %1\n\t"
      "s_mov_b64 exec, %5\n\t"
      "global_load_lds_dwordx4 %0, %2\n\t"
      "s_mov_b64 exec, -1"
      :
      : "v"(voff), "s"(base_b), "s"(base_t), "s"(lds_dst), "s"(mask_b), "s"(mask_t)
      : "memory", "m0");
}
// All LDS reads of this wave have returned; every fragment passes through the statement, so no MFMA that consumes one
// can be moved above the wait.
__device__ __forceinline__ void frag_wait(u32x4 (&A)[8], u32x4 (&B)[4]) {
  u32x4 a0 = A[0], a1 = A[1], a2 = A[2], a3 = A[3], a4 = A[4], a5 = A[5], a6 = A[6], a7 = A[7];
  u32x4 b0 = B[0], b1 = B[1], b2 = B[2], b3 = B[3];
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  asm volatile("" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
  A[0] = a0, A[1] = a1, A[2] = a2, A[3] = a3, A[4] = a4, A[5] = a5, A[6] = a6, A[7] = a7;
  B[0] = b0, B[1] = b1, B[2] = b2, B[3] = b3;
}

__global__ __launch_bounds__(NT8) void gemm_w8_filter_kernel(const FastArgs p) {
  __shared__ __attribute__((aligned(1024))) char lds[LDS8_BYTES];
  const int ntt = p.ntt;
  const char* __restrict__ xb = p.x;
  const char* __restrict__ Wb = p.W;
  const char* __restrict__ bias = p.bias;
  const int bias_esize = p.bias_esize, kbytes = p.kbytes, nlt = p.nlt;
#ifdef CLT_TUNING  // timing ablations (tools/): 2 = no scan, 8 = no list stores
  const int dbg = p.dbg;
#else
  constexpr int dbg = 0;
#endif
  const int64_t N = p.N;
  const int ngroups = p.ngroups;  // valid virtual groups of 16 latent rows (host: launch_gemm_fast)
  const int lane = threadIdx.x & 63;
  const int wave = uni(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int c16 = lane & 15, hq = lane >> 4;

  // ---- this workgroup's tile ids (as encoder_gemm_fast.hip): XCD group blockIdx % 8 owns a contiguous id range
  const int ntiles = nlt * ntt;
  const int xg = blockIdx.x & 7, bi = blockIdx.x >> 3;
  const int nbx = ((int)gridDim.x - xg + 7) >> 3;
  const int qt = ntiles >> 3, rr = ntiles & 7;
  const int xstart = xg < rr ? xg * (qt + 1) : rr * (qt + 1) + (xg - rr) * qt;
  const int xcount = qt + (xg < rr ? 1 : 0);
  const int my_tiles = bi < xcount ? uni((xcount - bi + nbx - 1) / nbx) : 0;
  if (my_tiles == 0) {  // (cannot happen with the host's grid: at most one block per tile)
    if (lane == 0) p.stream_counts[(int)blockIdx.x * 8 + wave] = 0;
    return;
  }

  const int nsteps = kbytes / ROWB;  // even and >= PF8 (host: gemm_fast16_ok)
  const int total = my_tiles * nsteps;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)lds;
  const char* zero = reinterpret_cast<const char*>(g_zero16);

  // Tile id -> (virtual latent tile, token tile), carried incrementally (encoder_gemm_fast16.hip).
  struct TilePos {
    int pp, rem, pt, full;
  };
  auto pos_panel = [&](TilePos& c) {
    const int left = ntt - c.pp * PANEL8;
    c.pt = left < PANEL8 ? left : PANEL8;
    c.full = nlt * c.pt;
  };
  auto pos_init = [&](TilePos& c, int id) {
    const int fullp = nlt * PANEL8;
    c.pp = uni(id / fullp);
    c.rem = uni(id - c.pp * fullp);
    pos_panel(c);
  };
  auto pos_next = [&](TilePos& c) {
    c.rem += nbx;
    while (c.rem >= c.full) {
      c.rem -= c.full;
      ++c.pp;
      pos_panel(c);
    }
  };
  auto pos_get = [&](const TilePos& c, int& lt, int& tt) {
    if (c.pt == PANEL8) {
      lt = c.rem >> PANEL8_LOG;
      tt = c.pp * PANEL8 + (c.rem & (PANEL8 - 1));
    } else {
      lt = uni(c.rem / c.pt);
      tt = uni(c.pp * PANEL8 + c.rem % c.pt);
    }
  };
  TilePos ld_pos, cur_pos;
  pos_init(ld_pos, xstart + bi);
  cur_pos = ld_pos;

  // ---- loader state: runs PF8 stages ahead of the multiplier, possibly in the next tile.  Past the last stage it keeps
  //      re-fetching the last tile into ring slots nobody reads again, so that every k-step issues the same NDMA8
  //      operations and the counted waits need no special cases.
  int ld_t = 0, ld_ks = 0, ld_slot = 0;
  const char* ldW = Wb;
  const char* ldX = xb;
  const char* ldAb = zero;  // aux piece: the tile's bias table (or 16 zero bytes) ...
  uint32_t ldA = 0;         // ... and the byte offset of this lane's bias chunk in it (lanes < nl <= 8) or of its tau chunk in
  uint32_t voffW[2], voffX[2];  // p.tau (lanes 8 .. 15)
  const int nl = 32 * bias_esize / 16;  // lanes of the bias piece (this wave's 32 latents)
  const unsigned long long aux_mask_b = __ballot(lane < nl), aux_mask_t = __ballot(lane >= 8 && lane < 16);
  const int ntt_layer = p.ntt_layer;
  const int ldx = ntt_layer ? (int)p.ldx_bytes : kbytes;  // bytes between consecutive token rows of x
  const char* __restrict__ xp = p.xp;                      // packed copy of x (fused path), or null
  const int64_t xstep = xp ? p.xp_kstep_bytes : ROWB;      // bytes from one k-step of x to the next
  constexpr int DMA_BIAS = 1024;  // piece q is loaded with immediate q * 1024: lane offsets carry + 1024 - q * 1024 >= 0
  auto loader_tile = [&](int lt, int tt) {
    const int64_t tok0 = (int64_t)tt * GEMM_TN;
    const char* Wt = Wb;
    const char* bt = bias;
    if (ntt_layer) {  // stacked layers: this token tile's layer picks the table, the bias and the x block
      const int layer = uni(tt / ntt_layer);
      Wt = p.Wl[layer];
      bt = p.biasl[layer];
      ldX = xb + (int64_t)layer * p.x_layer_bytes + (int64_t)(tt - layer * ntt_layer) * GEMM_TN * ldx - DMA_BIAS;
    } else {
      ldX = xb + tok0 * kbytes - DMA_BIAS;
    }
    if (xp) ldX = xp + tok0 * ROWB - DMA_BIAS;  // packed copy [k-step][token][64 B]: a piece is 1 KiB of consecutive bytes
    // W rows come in groups of 16 (grp_actual, gemm_fast_dev.h): piece q of this wave is virtual group 16 lt + 2 wave + q;
    // groups past the end re-read the last one (never scanned).  Offsets are relative to the tile's first group.
    const int vg_last = ngroups - 1;
    const int g_first = grp_actual(lt * 16 < vg_last ? lt * 16 : vg_last, p.lt_mode, p.lt_S);
    ldW = Wt + (int64_t)g_first * 16 * kbytes - DMA_BIAS;
    const int maxX = (int)(N - 1 - tok0 < GEMM_TN ? N - 1 - tok0 : GEMM_TN);  // (stacked layers: N_l % 256 == 0)
    int g_act[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int r = (wave * 2 + q) * 16 + (lane >> 2);  // wave w stages rows 32 w .. 32 w + 31 of either operand
      const int c = (lane & 3) ^ ((0 - (r >> 2)) & 3);  // the bank swizzle of encoder_gemm_fast16.hip
      const int vg = lt * 16 + wave * 2 + q;
      g_act[q] = uni(grp_actual(vg < vg_last ? vg : vg_last, p.lt_mode, p.lt_S));
      voffW[q] = (uint32_t)(((g_act[q] - g_first) * 16 + (lane >> 2)) * kbytes + c * 16 + DMA_BIAS - q * 1024);
      voffX[q] = xp ? (uint32_t)(r * ROWB + c * 16 + DMA_BIAS - q * 1024)  // (padding rows exist in the packed copy)
                    : (uint32_t)((r < maxX ? r : maxX) * ldx + c * 16 + DMA_BIAS - q * 1024);
    }
    // bias chunk of lane l < nl: latents 32 wave + l epc .. of the tile = group 2 wave + (l epc >> 4), offset (l epc) & 15
    // (32-bit lane arithmetic: a 64-bit form is hoisted out of the tile loop by hipcc and spilled, and its reload waits
    // for vmcnt(0) -- every LDS-DMA in flight -- once per tile)
    const int epc = 16 / bias_esize;
    const int lb = lane * epc;
    ldAb = bt ? bt : zero;
    ldA = bt ? (uint32_t)((lb < 16 ? g_act[0] : g_act[1]) * 16 + (lb & 15)) * (uint32_t)bias_esize : 0u;
    if (lane >= 8) ldA = (uint32_t)((tok0 + wave * 32 + (lane - 8) * 4) * 4);  // tau is padded to a multiple of 256 tokens
  };
  // LDS-DMA operation u of a stage: 0 = m0 for W, 1..2 W pieces, 3 = m0 for x, 4..5 x pieces, 6 aux piece (tile parity
  // ld_t & 1: the aux pieces of the running tile stay put while the loader is in the next one)
  auto dma = [&](auto uc, uint32_t stage_lds, int kb0) {
    constexpr int u = decltype(uc)::value;
    if constexpr (u == 0) set_m0(stage_lds + wave * 2048);
    else if constexpr (u <= 2) dma_m0<(u - 1) * 1024>(ldW + kb0, voffW[u - 1]);
    else if constexpr (u == 3) set_m0(stage_lds + OP_BYTES + wave * 2048);
    else if constexpr (u <= 5) dma_m0<(u - 4) * 1024>(ldX + (int64_t)(kb0 / ROWB) * xstep, voffX[u - 4]);
    else if (kb0 == 0)  // a tile's first stage (wave-uniform): the tile's aux piece; lane l lands at + 16 l
      dma_aux(ldAb, reinterpret_cast<const char*>(p.tau), ldA, lds0 + AUX8_OFF + ((ld_t & 1) * 8 + wave) * AUX8_WAVE,
              aux_mask_b, aux_mask_t);
  };
  auto loader_advance = [&]() {
    ld_slot = (ld_slot + 1) & 3;
    if (++ld_ks == nsteps) {
      ld_ks = 0;
      if (++ld_t < my_tiles) {
        int lt, tt;
        pos_next(ld_pos);
        pos_get(ld_pos, lt, tt);
        loader_tile(lt, tt);
      }
    }
  };

  // ---- fragment addresses (encoder_gemm_fast16.hip): row * 64 + (chunk ^ swizzle) * 16, chunk = hq
  const int swz = (0 - (c16 >> 2)) & 3;
  const uint32_t fa = (uint32_t)((wm * 128 + c16) * ROWB + ((hq ^ swz) * 16));
  const uint32_t fb = (uint32_t)(OP_BYTES + (wn * 64 + c16) * ROWB + ((hq ^ swz) * 16));
  // piece pc of a stage's fragment set, in the order the MFMAs need them: A0 B0 .. B3 A1 .. A7
  auto read_piece = [&](auto pcc, uint32_t sa, uint32_t sb, u32x4 (&A)[8], u32x4 (&B)[4]) {
    constexpr int pc = decltype(pcc)::value;
    if constexpr (pc == 0) A[0] = lds_read128<0>(sa);
    else if constexpr (pc <= 4) B[pc - 1] = lds_read128<(pc - 1) * 16 * ROWB>(sb);
    else A[pc - 4] = lds_read128<(pc - 4) * 16 * ROWB>(sa);
  };

  f32x4 acc[8][4];
  u32x4 FA0[8], FB0[4], FA1[8], FB1[4];
  // ---- candidate queue (LDS, wave-private) and this wave's region of the candidate stream (global)
  const uint32_t qaddr = lds0 + QUEUE8_OFF + wave * (QCAP8 * 16);
  int qn = 0;  // entries in the queue (wave-uniform)
  const int region = (int)blockIdx.x * 8 + wave;
  const int rcap = p.stream_rcap;  // entries per region (a multiple of 64)
  u32x4* const rbase = reinterpret_cast<u32x4*>(p.stream) + (int64_t)region * rcap;
  int rn = 0;  // entries of this wave already in its region (wave-uniform); runs on past rcap: the bucket kernel reports it
  auto flush_queue = [&]() {
    if (dbg & 4) {  // (timing ablation: the queue is dropped, and so is its count -- the stream stays consistent)
      qn = 0;
      return;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's queue writes have landed (one wave's LDS operations complete in order)
    // (the lane number afresh: at 256 registers hipcc otherwise keeps 16 * lane alive across the tile loop in scratch)
    const int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    for (int e0 = 0; e0 < qn; e0 += 64) {
      const int e = e0 + ln;
      if (e < qn && rn + e < rcap) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(lds + QUEUE8_OFF + wave * (QCAP8 * 16) + e * 16);
        rbase[rn + e] = v;  // (write-through and non-temporal forms were measured: no difference)
      }
    }
    rn += qn;
    qn = 0;
  };

  // The bias enters as three extra K columns (encoder_gemm_fast.hip): A = (b_hi, b_mid, b_lo, 0...) of this lane's
  // latent, B = (1, 1, 1, 0...), both in the lanes of K chunk 0.
  auto init_acc = [&](int par) {
    const char* st = lds + AUX8_OFF + par * 8 * AUX8_WAVE;
    const u32x4 zero_f = {0u, 0u, 0u, 0u}, ones_f = {0x3f803f80u, 0x00003f80u, 0u, 0u};
    const u32x4 ones = hq == 0 ? ones_f : zero_f;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int off = wm * 128 + i * 16 + c16;  // latent in the tile: aux piece of wave off / 32
      const char* bp = st + (off >> 5) * AUX8_WAVE + (off & 31) * bias_esize;
      float b;
      if (bias_esize == 4) b = *reinterpret_cast<const float*>(bp);
      else b = bf16_bits_to_f32(*reinterpret_cast<const uint16_t*>(bp));
      const uint32_t u0 = __float_as_uint(b) & 0xffff0000u;  // truncation splits: every remainder is exact
      const float r1 = b - __uint_as_float(u0);
      const uint32_t u1 = __float_as_uint(r1) & 0xffff0000u;
      const float r2 = r1 - __uint_as_float(u1);
      const uint32_t u2 = __float_as_uint(r2) & 0xffff0000u;
      const u32x4 bias_f = {(u0 >> 16) | u1, u2 >> 16, 0u, 0u};
      const u32x4 Ab = hq == 0 ? bias_f : zero_f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        u32x4 oj = ones;
        asm volatile("" : "+v"(oj));  // independent MFMAs, not one plus accumulator moves
        acc[i][j] = mfma16(Ab, oj, zero4);
      }
    }
  };

  // ---- prologue: PF8 stages in flight, all landed before the first MFMA (once per launch)
  int cur_lt, cur_tt;
  pos_get(cur_pos, cur_lt, cur_tt);
  loader_tile(cur_lt, cur_tt);
#pragma unroll 1
  for (int pp = 0; pp < PF8; ++pp) {
    const uint32_t st = lds0 + (uint32_t)ld_slot * STAGE8;
    static_for<7>([&](auto uc) { dma(uc, st, ld_ks * ROWB); });
    loader_advance();
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  static_for<12>([&](auto pcc) { read_piece(pcc, lds0 + fa, lds0 + fb, FA0, FB0); });
  init_acc(0);

  // One k-step: 32 MFMAs out of (Ac, Bc); in their shadows the certification of stage s + 1, the NDMA8 LDS-DMAs of
  // stage s + PF8 (into the slot of stage s, whose fragments are in registers) and the 12 fragment reads of stage s + 1
  // into (An, Bn).  Stage s + 1 has landed once only the operations issued after its LDS-DMAs are outstanding: the
  // 2 NDMA8 data DMAs of stages s + 2 and s + 3 (an aux piece or the list stores of an epilogue among the older
  // operations only make the wait stricter).  The barrier also tells every wave that all eight have finished reading
  // stage s (frag_wait: this wave's reads, issued during the step before, have returned).
  auto kstep = [&](u32x4 (&Ac)[8], u32x4 (&Bc)[4], u32x4 (&An)[8], u32x4 (&Bn)[4], uint32_t nb, uint32_t ld_stage,
                   int kb0) {
    frag_wait(Ac, Bc);
    const uint32_t sa = nb + fa, sb = nb + fb;
    CLT_FENCE();
    static_for<32>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      constexpr int i = t >> 2, j = t & 3;
      acc[i][j] = mfma16(Ac[i], Bc[j], acc[i][j]);
      CLT_FENCE();
      if constexpr (t == 1) {
        if constexpr ((CLT_ABL8 & 48) == 48) asm volatile("" ::: "memory");
        else if constexpr (CLT_ABL8 & 16) asm volatile("s_barrier" ::: "memory");
        else if constexpr (CLT_ABL8 & 32) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
      }
      constexpr int q = dma8_at(t), pc = read8_at(t);
      if constexpr (q >= 0 && !(CLT_ABL8 & 4)) {  // (the m0 write of an operand goes with its first piece)
        if constexpr (q == 0) dma(std::integral_constant<int, 0>{}, ld_stage, kb0);
        if constexpr (q == 2) dma(std::integral_constant<int, 3>{}, ld_stage, kb0);
        dma(std::integral_constant<int, (q < 2 ? q + 1 : q + 2)>{}, ld_stage, kb0);
        if constexpr (q == 3) dma(std::integral_constant<int, 6>{}, ld_stage, kb0);
      }
      if constexpr (pc >= 0 && !(CLT_ABL8 & 8)) read_piece(std::integral_constant<int, pc>{}, sa, sb, An, Bn);
      CLT_FENCE();
    });
  };
  static_assert(2 * NDMA8 == 8, "the counted wait of kstep");

  if constexpr (CLT_ABL8 & 64) {
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
  }
  int ks = 0, t_idx = 0;
#pragma unroll 1
  for (int s = 0; s < total; s += 2) {
    const bool last_k = ks + 2 == nsteps;
    // ---- even step of the pair (never a tile's last: nsteps is even)
    kstep(FA0, FB0, FA1, FB1, lds0 + (uint32_t)((s + 1) & 3) * STAGE8, lds0 + (uint32_t)ld_slot * STAGE8, ld_ks * ROWB);
    loader_advance();
    // ---- odd step
    kstep(FA1, FB1, FA0, FB0, lds0 + (uint32_t)((s + 2) & 3) * STAGE8, lds0 + (uint32_t)ld_slot * STAGE8, ld_ks * ROWB);
    loader_advance();

    if (last_k) {
      const int64_t tok0 = (int64_t)cur_tt * GEMM_TN;
      // a 16-latent MFMA row tile is one group: inside or outside the valid (virtual) groups as a whole -- wave-uniform tests
      const int vg0 = cur_lt * 16 + wm * 8;  // virtual group of row tile i = 0
      const int g_left = ngroups - vg0;
      const int lat_left = g_left < 8 ? (g_left > 0 ? g_left * 16 : 0) : 128;
      // actual groups of the eight row tiles: v -> v + v / (S - 1) (LT_GREST) from one division, then by counting
      int g_row[8];
      {
        const bool rest = p.lt_mode == LT_GREST;
        const int sm1 = rest ? p.lt_S - 1 : 0x7fffffff;
        int qd = rest ? uni(vg0 / sm1) : 0, rem = rest ? vg0 - qd * sm1 : 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          g_row[i] = vg0 + i + qd;
          if (++rem == sm1) rem = 0, ++qd;
        }
      }
      // ---- filter epilogue.  A register group (4 consecutive latents of one token) that holds an accumulator at or
      //      above its token's threshold becomes ONE 16-byte entry {4 bf16 values, token, first latent}; the merge side
      //      tests each value itself.  Scan: per group an integer max-of-4 read straight from the accumulator VGPRs + ONE
      //      compare; the four token groups of a latent row are compared together and their masks pinned in SGPR pairs
      //      before the four wave-uniform branches.  A group with a hit converts its values (two v_cvt_pk) and the lanes
      //      with a hit append their entries, COMPACTED (position = wave cursor + rank of the lane among the hits), to the
      //      wave's queue in LDS: one ds_write_b128 under EXEC = the compare mask.  The queue leaves for this wave's own
      //      region of the candidate stream in global memory as rows of 64 entries (1 KiB per store instruction) at the
      //      end of a tile once it holds a full row, and at the end of the kernel -- a scattered store per hit group
      //      (8 bytes into a per-token list) was measured at ~35 cycles of the CU's store path each, 90 us per call.
      float tl[4];
      const char* aux = lds + AUX8_OFF + (t_idx & 1) * 8 * AUX8_WAVE;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int tt = wn * 64 + j * 16 + c16;
        const float tj = *reinterpret_cast<const float*>(aux + (tt >> 5) * AUX8_WAVE + 128 + (tt & 31) * 4);
        tl[j] = tok0 + tt < N ? tj : __builtin_inff();
      }
      uint32_t tokv = (uint32_t)(tok0 + wn * 64 + c16);      // token of this lane's group j = 0 (group j: + 16 j)
      uint32_t latv = (uint32_t)(hq * 4);  // first latent of this lane's register group inside its 16-latent row tile
      asm volatile("" : "+v"(tokv), "+v"(latv));  // (opaque: hipcc otherwise hoists every tokv + 16 j / latv + ... and spills)
      bool q_over = false;  // wave-uniform
      if (!(dbg & 2))
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (i * 16 < lat_left) {
          unsigned long long hm[4];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            // thresholds are positive, so a >= t  <=>  bits(a) >= bits(t) as SIGNED integers
            const int mx = max(max(__float_as_int(acc[i][jj][0]), __float_as_int(acc[i][jj][1])),
                               max(__float_as_int(acc[i][jj][2]), __float_as_int(acc[i][jj][3])));
            hm[jj] = __ballot(mx >= __float_as_int(tl[jj]));
          }
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) asm volatile("" : "+s"(hm[jj]));  // all four masks exist before the first branch
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            if (hm[jj] != 0ull && !(dbg & 8)) {
              if (qn > QCAP8 - 64) {  // (no room for another group: this token tile goes through the exact fallback)
                q_over = true;
                continue;
              }
              const bf16x2_t p01 = {(__bf16)acc[i][jj][0], (__bf16)acc[i][jj][1]},
                             p23 = {(__bf16)acc[i][jj][2], (__bf16)acc[i][jj][3]};
              const u32x4 ent = {__builtin_bit_cast(uint32_t, p01), __builtin_bit_cast(uint32_t, p23), tokv + 16u * jj,
                                 latv + 16u * (uint32_t)g_row[i]};
              const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(hm[jj] >> 32),
                                                              __builtin_amdgcn_mbcnt_lo((uint32_t)hm[jj], 0u));
              const uint32_t qa = qaddr + (((uint32_t)qn + rank) << 4);
              asm volatile(
                  "s_mov_b64 exec, %0\n\t"
                  "ds_write_b128 %1, %2\n\t"
                  "s_mov_b64 exec, -1"
                  :
                  : "s"(hm[jj]), "v"(qa), "v"(ent)
                  : "memory");
              qn += __builtin_popcountll(hm[jj]);
            }
          }
          CLT_FENCE();
        }
        // The queue is written out whenever it holds a full row of 64 entries (dense 1 KiB stores) at the end of a latent
        // row: a latent row (4 groups x 64 lanes) can then always add 97 entries -- 50 times its mean at the BASELINE
        // shapes, 8 times that of a tp8 shard (where ~95 entries per tile and wave overflowed a queue that was only
        // emptied at tile ends: 26 of 256 token tiles fell back at N = 65536, M = 12288).  (Cadences of 1, 4 and 8
        // tiles, all waves together, were measured at C2: no difference.)
        if (qn >= 64) flush_queue();
      }
      if (q_over && __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u)
        atomicOr(p.fb_flags + 1 + cur_tt, 1);
      ks = 0;
      if (++t_idx < my_tiles) {
        pos_next(cur_pos);
        pos_get(cur_pos, cur_lt, cur_tt);
        init_acc(t_idx & 1);  // its aux pieces came with stage s + 2, certified during the odd step
      }
    } else {
      ks += 2;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the loader's trailing fetches
  flush_queue();
  if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u) p.stream_counts[region] = rn;
}

}  // namespace

// Filter mode only; same shapes as the 16x16x32 four-wave kernel (gemm_fast16_ok).
int launch_gemm_w8_filter(const FastArgs& a, int grid, hipStream_t st) {
  hipLaunchKernelGGL(gemm_w8_filter_kernel, dim3(grid), dim3(NT8), 0, st, a);
  CLT_CHECK_LAUNCH("clt_encode_topk_fwd");
  return CLT_OK;
}

}  // namespace clt
