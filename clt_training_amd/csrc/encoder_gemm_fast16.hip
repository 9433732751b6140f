// Encoder GEMM of the hot path, bf16 operands, on v_mfma_f32_16x16x32_bf16: pre[N, M] = relu(x[N, d] @ W[M, d]^T + b).
//
// Same tiling, staging and product orientation as encoder_gemm_fast.hip (persistent, one wave per SIMD, tile 256 latents
// x 256 tokens, 128 x 128 per wave, D[latent][token], LDS-DMA ring with the bank swizzle), but the wave tile is 8 x 8
// MFMA tiles of 16 x 16 with the whole 32-wide k-step in ONE fragment: under a bf16 MFMA load on random data this part
// holds a higher clock with this shape (MI355X_MICROARCH.md, DVFS item 7; tools/mfma_ab.hip: 1.98 against 1.73 PFLOP/s
// as bare loops).  What changes with the shape:
//   * a ds_read_b128 brings a 16-row fragment for all of K = 32 (lane l: row l & 15, 16-byte chunk l >> 4), so a k-step
//     is 16 reads + 64 MFMAs and its fragments are whole-step double-buffered: step s multiplies out of F[s & 1] while
//     the 16 reads of step s + 1 fill F[~s & 1].  The k loop is unrolled by two (K a multiple of 64 elements; other
//     shapes stay on the 32x32x16 kernel);
//   * stage s is in registers before step s starts, so its ring slot is free one step earlier: the loader runs FOUR
//     stages ahead (three in flight while one is certified) in the same four slots;
//   * a lane owns one token per token tile (8 tokens) and 4 consecutive latents per latent tile: a candidate list is
//     private to (token, latent tile, latent half wm, lane quarter) = 32 latents (8 x 4), 8 lists per latent tile;
//   * store mode writes 16 rows x 32 contiguous bytes per instruction (32 x 16 before).
// MFMA-bound: 2 * N * M * d flops per launch.
#include <math.h>

#include <utility>

#include "gemm_fast_dev.h"

namespace clt {
namespace {

#ifndef CLT_ABL16  // compile-time ablations for tools/ (results are garbage): 4 = no LDS-DMA in the loop, 8 = no fragment
#define CLT_ABL16 0  // reads, 16 = no certification wait, 32 = no barrier
#endif

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;  // a fragment: native vector, so that asm can tie it

constexpr int PF = 4;       // stages the loader runs ahead of the multiplier
// This kernel's LDS: a ring of NSTAGE stages of the two operands (32 KiB each), then the aux pieces (bias of 64 latents
// + tau of 128 tokens per wave, 768 B) of the current and the next tile -- loaded ONCE per tile with its first stage,
// not with every stage (an LDS-DMA costs ~40 cycles of a wave's issue here: tools note in DESIGN.md 4.1) -- then the
// filter mode's hit queue.
constexpr int STAGE16 = 2 * OP_BYTES;
constexpr int AUX16_OFF = NSTAGE * STAGE16;
constexpr int QUEUE16_OFF = AUX16_OFF + 2 * 4 * AUX_WAVE;
constexpr int STG_WAVE = 16 * 256;  // store mode, bf16: a wave's staging area for one 16-token x 128-latent block
constexpr int NDMA16 = 8;    // data LDS-DMAs per wave and stage (a tile's first stage has the aux piece on top)
static_assert(QUEUE16_OFF + 4 * STG_WAVE <= 160 * 1024, "LDS");
static_assert(2 * NDMA16 == 16 && 2 * NDMA16 + NSTORE >= 63 && NSTORE == 64, "the counted waits of kstep assume these");

__device__ __forceinline__ f32x4 mfma16(const u32x4& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0,
                                                 0);
}

// Compile-time loop: the slot number of a k-step's instruction stream must be a constant expression, because the
// fragment reads and the LDS-DMAs carry their displacement as an immediate.
template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// One wave per SIMD issues ONE instruction per ~4 cycles, and a 16x16x32 MFMA holds the vector issue for 8 of its 16
// cycles (MI355X_MICROARCH.md, cycle constants): a gap between two MFMAs hides about two other instructions, not the
// six or seven of the 32x32x16 kernel's LDS-DMA sequence (m0 save / set / nop / load / restore + an address add).  So:
//   * fragment reads carry their displacement as an immediate (two base adds per k-step instead of one per read);
//   * a wave's four pieces of an operand are ADJACENT in the LDS image (rows 64 w .. 64 w + 63), m0 is written once per
//     operand and k-step, and the piece displacement q * 1024 is the load's immediate -- which moves the LDS address and
//     the global address alike (measured: tools note in DESIGN.md 4.1), so the lane offsets are pre-compensated;
//   * m0 is declared clobbered instead of saved and restored; the aux piece masks EXEC by hand instead of branching.
// Fragment read from inline asm: as plain loads hipcc sank all 16 reads of a k-step below its last MFMA (nothing pins a
// load between sched_barriers), which exposed the LDS latency once per step.  Volatile asm keeps its place; the wait
// for the results is ours (frag_wait).
// Instruction slots of a k-step (slot t = after MFMA t): one table for all four waves (w = 0): the q-th DMA of the step
// after MFMA 2 + 7 q, the p-th fragment read after MFMA 3 + 3 p (or the next slot when that one holds a DMA).  A
// per-wave skew of the DMA slots (w > 0 below: each wave its own copy of the k-step behind a switch on the wave number)
// was built to keep the four waves from handing their LDS-DMAs to the CU's address path in the same cycles; hipcc
// spilled 744 registers over the four copies and the skew was dropped -- the kernel instantiates w = 0
// only.
__host__ __device__ constexpr int dma_q_at(int w, int t) {  // the real DMA (0..3 W pieces, 4..7 x pieces, 8 aux) or -1
  const int sh = w == 0 ? 0 : w == 1 ? 2 : w == 2 ? 4 : 5;
  const int r = t - 2 - sh;
  return (r >= 0 && r % 7 == 0 && r / 7 < 9) ? r / 7 : -1;
}
__host__ __device__ constexpr int read_at(int w, int t) {  // the fragment read (0..15) or -1: slot 3 + 3 p, or the next
  for (int pc = 0; pc < 16; ++pc) {                        // one when that is a DMA slot of this wave
    int c = 3 + 3 * pc;
    if (dma_q_at(w, c) >= 0) ++c;
    if (c == t) return pc;
  }
  return -1;
}
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
// the aux piece: lanes of `mask` only (every lane of the wave is active here: uniform control flow)
__device__ __forceinline__ void dma_aux(const char* src, uint32_t lds_dst, unsigned long long mask) {
  asm volatile(
      "s_mov_b32 m0, %1\n\t"
      "s_mov_b64 exec, %2\n\t"
      "global_load_lds_dwordx4 %0, off\n\t"
      "s_mov_b64 exec, -1"
      :
      : "v"(src), "s"(lds_dst), "s"(mask)
      : "memory", "m0");
}
// All LDS reads of this wave have returned; every fragment passes through the statement, so no MFMA that consumes one
// can be moved above the wait.  (LDS operations return in order and the counter counts hipcc's own reads too, so the
// asm reads only ever make hipcc's counted waits more conservative.)
__device__ __forceinline__ void frag_wait(u32x4 (&A)[8], u32x4 (&B)[8]) {
  u32x4 a0 = A[0], a1 = A[1], a2 = A[2], a3 = A[3], a4 = A[4], a5 = A[5], a6 = A[6], a7 = A[7];
  u32x4 b0 = B[0], b1 = B[1], b2 = B[2], b3 = B[3], b4 = B[4], b5 = B[5], b6 = B[6], b7 = B[7];
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  asm volatile("" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7));
  A[0] = a0, A[1] = a1, A[2] = a2, A[3] = a3, A[4] = a4, A[5] = a5, A[6] = a6, A[7] = a7;
  B[0] = b0, B[1] = b1, B[2] = b2, B[3] = b3, B[4] = b4, B[5] = b5, B[6] = b6, B[7] = b7;
}

// Store mode: relu(pre) as OT = bf16 or f32 (pass A of the fused encode + top-k, its fallback, clt_encoder_gemm_relu).
// The filter pass of the fused path runs on eight waves: encoder_gemm_w8.hip.
template <typename OT>
__global__ __launch_bounds__(NTHREADS, 1) void gemm_fast16_kernel(const FastArgs p) {
  __shared__ __attribute__((aligned(1024))) char lds[QUEUE16_OFF + (sizeof(OT) == 2 ? 4 * STG_WAVE : 0)];
  int ntt = p.ntt;
  if (p.gate) {  // fallback launch: runs only if the fused path raised a flag
    const int flagged = *p.gate;
    if (flagged == 0) return;
    if (p.tt_map) ntt = flagged;  // only the flagged token tiles: virtual token tile i is tile tt_map[i]
  }
  const char* __restrict__ xb = p.x;
  const char* __restrict__ Wb = p.W;
  const char* __restrict__ bias = p.bias;
  const int bias_esize = p.bias_esize, kbytes = p.kbytes, nlt = p.nlt;
#ifdef CLT_TUNING  // timing ablations (tools/): 1 = no stores
  const int dbg = p.dbg;
#else
  constexpr int dbg = 0;
#endif
  const int* __restrict__ tt_map = p.tt_map;
  const int64_t N = p.N;
  const int ngroups = p.ngroups;  // valid virtual groups of 16 latent rows (host: launch_gemm_fast)
  const int lane = threadIdx.x & 63;
  const int wave = uni(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int c16 = lane & 15, hq = lane >> 4;

  // ---- this workgroup's tile ids (as encoder_gemm_fast.hip): XCD group blockIdx % 8 owns a contiguous id range
  const int ntiles = nlt * ntt;
  const int xg = blockIdx.x & 7, bi = blockIdx.x >> 3;
  const int nbx = ((int)gridDim.x - xg + 7) >> 3;
  const int qt = ntiles >> 3, rr = ntiles & 7;
  const int xstart = xg < rr ? xg * (qt + 1) : rr * (qt + 1) + (xg - rr) * qt;
  const int xcount = qt + (xg < rr ? 1 : 0);
  const int my_tiles = bi < xcount ? uni((xcount - bi + nbx - 1) / nbx) : 0;
  if (my_tiles == 0) return;

  const int nsteps = kbytes / ROWB;  // even and >= PF (host: gemm_fast16_ok)
  const int total = my_tiles * nsteps;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)lds;
  const char* zero = reinterpret_cast<const char*>(g_zero16);

  // Tile id -> (virtual latent tile, token tile), see tile_of (gemm_fast.h).  A workgroup's ids advance by nbx, so the
  // position (panel pp, offset rem inside the panel) is carried along instead of divided out per tile: the two integer
  // divisions of the straightforward form are ~150 scalar instructions between two k-steps, twice per tile (loader and
  // multiplier), with the MFMA pipe idle.  Panels are GEMM_PANEL = 8 token tiles wide except the last one.
  struct TilePos {
    int pp, rem, pt, full;
  };
  auto pos_panel = [&](TilePos& c) {
    const int left = ntt - c.pp * GEMM_PANEL;
    c.pt = left < GEMM_PANEL ? left : GEMM_PANEL;
    c.full = nlt * c.pt;
  };
  auto pos_init = [&](TilePos& c, int id) {
    const int fullp = nlt * GEMM_PANEL;
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
    if (c.pt == GEMM_PANEL) {
      lt = c.rem >> 3;
      tt = c.pp * GEMM_PANEL + (c.rem & 7);
    } else {
      lt = uni(c.rem / c.pt);
      tt = uni(c.pp * GEMM_PANEL + c.rem % c.pt);
    }
    if (tt_map) tt = uni(tt_map[tt]);
  };
  static_assert(GEMM_PANEL == 8, "pos_get shifts by 3");
  TilePos ld_pos, cur_pos;
  pos_init(ld_pos, xstart + bi);
  cur_pos = ld_pos;

  // ---- loader state: runs PF stages ahead of the multiplier, possibly in the next tile.  Past the last stage it keeps
  //      re-fetching the last tile into ring slots nobody reads again, so that every k-step issues the same NDMA
  //      operations and the counted waits need no special cases.
  int ld_t = 0, ld_ks = 0, ld_slot = 0;
  const char* ldW = Wb;
  const char* ldX = xb;
  const char* ldA = zero;  // per-lane source of the aux piece: bias chunk (lanes < nl <= 16) or tau chunk (lanes 16 .. 47)
  uint32_t voffW[4], voffX[4];
  const int nl = 64 * bias_esize / 16;  // lanes of the bias piece (this wave's 64 latents)
  const unsigned long long aux_mask = __ballot(lane < nl);
  const int ntt_layer = p.ntt_layer;
  const int ldx = ntt_layer ? (int)p.ldx_bytes : kbytes;  // bytes between consecutive token rows of x
  const char* __restrict__ xp = p.xp;                      // packed copy of x (fused path), or null
  const int64_t xstep = xp ? p.xp_kstep_bytes : ROWB;      // bytes from one k-step of x to the next
  constexpr int DMA_BIAS = 3 * 1024;  // piece q is loaded with immediate q * 1024: lane offsets carry + 3072 - q * 1024 >= 0
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
    // W rows come in groups of 16 (grp_actual): piece q of this wave is virtual group 16 lt + 4 wave + q; groups past the
    // end re-read the last one (their products are never stored).  Offsets are relative to the tile's first group.
    const int vg_last = ngroups - 1;
    const int g_first = grp_actual(lt * 16 < vg_last ? lt * 16 : vg_last, p.lt_mode, p.lt_S);
    ldW = Wt + (int64_t)g_first * 16 * kbytes - DMA_BIAS;
    const int maxX = (int)(N - 1 - tok0 < GEMM_TN ? N - 1 - tok0 : GEMM_TN);  // (stacked layers: N_l % 256 == 0)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = (wave * 4 + q) * 16 + (lane >> 2);  // wave w stages rows 64 w .. 64 w + 63 of either operand
      const int c = (lane & 3) ^ ((0 - (r >> 2)) & 3);  // this kernel's bank swizzle (see the fragment addresses)
      const int vg = lt * 16 + wave * 4 + q;
      const int g_rel = uni(grp_actual(vg < vg_last ? vg : vg_last, p.lt_mode, p.lt_S) - g_first);
      voffW[q] = (uint32_t)((g_rel * 16 + (lane >> 2)) * kbytes + c * 16 + DMA_BIAS - q * 1024);
      voffX[q] = xp ? (uint32_t)(r * ROWB + c * 16 + DMA_BIAS - q * 1024)  // (padding rows exist in the packed copy)
                    : (uint32_t)((r < maxX ? r : maxX) * ldx + c * 16 + DMA_BIAS - q * 1024);
    }
    // bias chunk of lane l < nl: latents 64 wave + l epc .. of the tile, i.e. group 4 wave + (l epc >> 4), offset (l epc) & 15
    const int epc = 16 / bias_esize;
    const int vgb = lt * 16 + wave * 4 + ((lane * epc) >> 4);
    const int64_t lat = (int64_t)grp_actual(vgb < vg_last ? vgb : vg_last, p.lt_mode, p.lt_S) * 16 + ((lane * epc) & 15);
    ldA = (bt && lane < nl) ? bt + lat * bias_esize : zero;
  };
  // LDS-DMA operation u of a stage: 0 = m0 for W, 1..4 W pieces, 5 = m0 for x, 6..9 x pieces, 10 aux piece (tile parity
  // ld_t & 1: the aux pieces of the running tile stay put while the loader is in the next one)
  auto dma = [&](auto uc, uint32_t stage_lds, int kb0) {
    constexpr int u = decltype(uc)::value;
    if constexpr (u == 0) set_m0(stage_lds + wave * 4096);
    else if constexpr (u <= 4) dma_m0<(u - 1) * 1024>(ldW + kb0, voffW[u - 1]);
    else if constexpr (u == 5) set_m0(stage_lds + OP_BYTES + wave * 4096);
    else if constexpr (u <= 9) dma_m0<(u - 6) * 1024>(ldX + (int64_t)(kb0 / ROWB) * xstep, voffX[u - 6]);
    else if (kb0 == 0)  // a tile's first stage (wave-uniform): the tile's aux piece; lane l lands at + 16 l
      dma_aux(ldA, lds0 + AUX16_OFF + ((ld_t & 1) * 4 + wave) * AUX_WAVE, aux_mask);
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

  // ---- fragment addresses: row * 64 + (chunk ^ swizzle) * 16, chunk = hq; the swizzle is a function of (row >> 2) & 3
  //      and so a lane constant, because every fragment starts at a multiple of 16 rows.  ds_read_b128 serves four
  //      groups of 16 lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32: MI355X_MICROARCH.md, LDS), each
  //      of which must touch 16 different 16-byte slots of the 256-byte bank row: slot = (row & 3) * 4 + position.  With
  //      lane = (row l & 15, chunk l >> 4) a group holds rows g = 0 and g = 3 of one chunk and rows g = 1, 2 of the next
  //      (g = row >> 2), so the positions f(0), f(3), 1 ^ f(1), 1 ^ f(2) must differ: f(g) = -g & 3 (the 32x32x16
  //      kernel's f(g) = g is 2-way here: measured 4 % slower than that kernel instead of faster).
  const int swz = (0 - (c16 >> 2)) & 3;
  const uint32_t fa = (uint32_t)((wm * 128 + c16) * ROWB + ((hq ^ swz) * 16));
  const uint32_t fb = (uint32_t)(OP_BYTES + (wn * 128 + c16) * ROWB + ((hq ^ swz) * 16));
  // piece pc of a stage's fragment set, in the order the MFMAs need them: A0 B0 .. B7 A1 .. A7
  // (sa, sb: LDS byte address of the stage + fa / + fb)
  auto read_piece = [&](auto pcc, uint32_t sa, uint32_t sb, u32x4 (&A)[8], u32x4 (&B)[8]) {
    constexpr int pc = decltype(pcc)::value;
    if constexpr (pc == 0) A[0] = lds_read128<0>(sa);
    else if constexpr (pc <= 8) B[pc - 1] = lds_read128<(pc - 1) * 16 * ROWB>(sb);
    else A[pc - 8] = lds_read128<(pc - 8) * 16 * ROWB>(sa);
  };

  f32x4 acc[8][8];
  u32x4 FA0[8], FB0[8], FA1[8], FB1[8];

  // The bias enters as three extra K columns (encoder_gemm_fast.hip): A = (b_hi, b_mid, b_lo, 0...) of this lane's
  // latent, B = (1, 1, 1, 0...), both in the lanes of K chunk 0.
  auto init_acc = [&](int par) {
    const char* st = lds + AUX16_OFF + par * 4 * AUX_WAVE;
    const u32x4 zero_f = {0u, 0u, 0u, 0u}, ones_f = {0x3f803f80u, 0x00003f80u, 0u, 0u};
    const u32x4 ones = hq == 0 ? ones_f : zero_f;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int off = wm * 128 + i * 16 + c16;  // latent in the tile: aux piece of wave off / 64
      const char* bp = st + (off >> 6) * AUX_WAVE + (off & 63) * bias_esize;
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
      for (int j = 0; j < 8; ++j) {
        u32x4 oj = ones;
        asm volatile("" : "+v"(oj));  // independent MFMAs, not one plus accumulator moves
        acc[i][j] = mfma16(Ab, oj, zero4);
      }
    }
  };

  // ---- prologue: PF stages in flight, all landed before the first MFMA (once per launch)
  int cur_lt, cur_tt;
  pos_get(cur_pos, cur_lt, cur_tt);
  loader_tile(cur_lt, cur_tt);
#pragma unroll 1
  for (int pp = 0; pp < PF; ++pp) {
    const uint32_t st = lds0 + (uint32_t)ld_slot * STAGE16;
    static_for<11>([&](auto uc) { dma(uc, st, ld_ks * ROWB); });
    loader_advance();
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  static_for<16>([&](auto pcc) { read_piece(pcc, lds0 + fa, lds0 + fb, FA0, FB0); });
  init_acc(0);

  // One k-step: 64 MFMAs out of (Ac, Bc); in their shadows the certification of stage s + 1, the NDMA LDS-DMAs of
  // stage s + PF (into the slot of stage s, whose fragments are in registers) and the 16 fragment reads of stage s + 1
  // into (An, Bn), one LDS operation per 2-3 MFMAs.  Stage s + 1 has landed once only the operations issued after its
  // LDS-DMAs are outstanding: the 2 NDMA16 data DMAs of stages s + 2 and s + 3 (an aux piece among them only makes the
  // wait stricter), plus -- for the three k-steps after a store epilogue -- the epilogue's stores, which are younger
  // than the DMAs of stage s + 1 (the counter saturates at 63).
  // The barrier also tells every wave that all four have finished reading stage s (frag_wait: this wave's reads, issued
  // during the step before, have returned).
  auto kstep = [&](auto wc, u32x4 (&Ac)[8], u32x4 (&Bc)[8], u32x4 (&An)[8], u32x4 (&Bn)[8], uint32_t nb,
                   uint32_t ld_stage, int kb0, bool after_stores) {
    constexpr int W = decltype(wc)::value;
    frag_wait(Ac, Bc);
    const uint32_t sa = nb + fa, sb = nb + fb;
    CLT_FENCE();
    static_for<64>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      constexpr int i = t >> 3, j = t & 7;
      acc[i][j] = mfma16(Ac[i], Bc[j], acc[i][j]);
      CLT_FENCE();
      if constexpr (t == 1) {
#if defined(CLT_ABL16) && (CLT_ABL16 & 16)  // (ablation: no certification wait)
        asm volatile("s_barrier" ::: "memory");
#elif defined(CLT_ABL16) && (CLT_ABL16 & 32)  // (ablation: no barrier)
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
#else
        if (after_stores) {  // 16 + the stores of an interior tile: 64 (f32: saturates at 63) or 32 (bf16, staged)
          if constexpr (sizeof(OT) == 2) asm volatile("s_waitcnt vmcnt(48)\n\ts_barrier" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(63)\n\ts_barrier" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
        }
#endif
      }
      constexpr int q = dma_q_at(W, t), pc = read_at(W, t);
      if constexpr (q >= 0 && !(CLT_ABL16 & 4)) {  // (the m0 write of an operand goes with its first piece)
        if constexpr (q == 0) dma(std::integral_constant<int, 0>{}, ld_stage, kb0);
        if constexpr (q == 4) dma(std::integral_constant<int, 5>{}, ld_stage, kb0);
        dma(std::integral_constant<int, (q < 4 ? q + 1 : q < 8 ? q + 2 : 10)>{}, ld_stage, kb0);
      }
      if constexpr (pc >= 0 && !(CLT_ABL16 & 8)) read_piece(std::integral_constant<int, pc>{}, sa, sb, An, Bn);
      CLT_FENCE();
    });
  };

  int ks = 0, t_idx = 0;
  int stores_behind = 0;
#pragma unroll 1
  for (int s = 0; s < total; s += 2) {
    const bool last_k = ks + 2 == nsteps;
    auto step_pair = [&](auto wc) {
      // ---- even step of the pair (never a tile's last: nsteps is even)
      kstep(wc, FA0, FB0, FA1, FB1, lds0 + (uint32_t)((s + 1) & 3) * STAGE16,
            lds0 + (uint32_t)ld_slot * STAGE16, ld_ks * ROWB, stores_behind > 0);
      loader_advance();
      if (stores_behind > 0) --stores_behind;
      // ---- odd step
      kstep(wc, FA1, FB1, FA0, FB0, lds0 + (uint32_t)((s + 2) & 3) * STAGE16,
            lds0 + (uint32_t)ld_slot * STAGE16, ld_ks * ROWB, stores_behind > 0);
      loader_advance();
      if (stores_behind > 0) --stores_behind;
    };
    step_pair(std::integral_constant<int, 0>{});

    if (last_k) {
      const int64_t tok0 = (int64_t)cur_tt * GEMM_TN;
      // a 16-latent MFMA tile is one group: inside or outside the valid (virtual) groups as a whole -- wave-uniform tests
      const int g_left = ngroups - (cur_lt * 16 + wm * 8);
      const int lat_left = g_left < 8 ? (g_left > 0 ? g_left * 16 : 0) : 128;
      {
        // ---- store epilogue: ReLU, one rounding.  bf16: each 16-token x 128-latent block goes through a wave-private
        //      staging area in LDS (row = token, 256 B; 16-byte chunk c of row r at position c ^ r: the 8-byte writes of
        //      a lane = (token, 4 latents) spread over all banks and the 16-byte reads of a lane = (token, 8 latents) are
        //      conflict-free) and leaves as 16-byte stores, 4 tokens x 256 contiguous bytes per instruction: 32 store
        //      instructions of 8 full lines per tile and wave instead of 64 of 16 x 32 bytes (a store is priced like an
        //      LDS-DMA: by the lines it touches).  f32: 4 consecutive latents per 16-byte store, as before.  One code path
        //      for interior and edge tiles (two paths make hipcc hoist all 256 accumulator reads above the branch: spills).
        if (!(dbg & 1)) {
          OT* pre = reinterpret_cast<OT*>(p.pre);
          const bool interior = tok0 + GEMM_TN <= N && lat_left == 128;
          if constexpr (sizeof(OT) == 2) {
            char* stg = lds + QUEUE16_OFF + wave * STG_WAVE;
            const int rq = lane >> 4, cc = lane & 15;  // read side: row 4 r4 + rq of the block, 16-byte chunk cc
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#pragma unroll
              for (int i = 0; i < 8; ++i) {
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = fmaxf(acc_read(acc[i][j][q]), 0.f);
                uint2 u;
                u.x = pack_bf16x2(v[0], v[1]);
                u.y = pack_bf16x2(v[2], v[3]);
                const int ch = 2 * i + (hq >> 1);  // 16-byte chunk of latents i * 16 + hq * 4 .. + 3 in the token's row
                *reinterpret_cast<uint2*>(stg + c16 * 256 + ((ch ^ c16) & 15) * 16 + (hq & 1) * 8) = u;
              }
              CLT_FENCE();
#pragma unroll
              for (int r4 = 0; r4 < 4; ++r4) {
                const int row = r4 * 4 + rq;
                const uint4 w = *reinterpret_cast<const uint4*>(stg + row * 256 + ((cc ^ row) & 15) * 16);
                const int64_t tok = tok0 + wn * 128 + j * 16 + row;
                if (tok < N && cc * 8 < lat_left)
                  *reinterpret_cast<uint4*>(pre + tok * p.ldp + (int64_t)cur_lt * GEMM_TM + wm * 128 + cc * 8) = w;
              }
              CLT_FENCE();
            }
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const int64_t tok = tok0 + wn * 128 + j * 16 + c16;
              const bool tok_ok = tok < N;
              OT* prow = pre + tok * p.ldp + (int64_t)cur_lt * GEMM_TM + wm * 128 + 4 * hq;
#pragma unroll
              for (int i = 0; i < 8; ++i) {
                if (tok_ok && i * 16 < lat_left) {
                  float v[4];
#pragma unroll
                  for (int q = 0; q < 4; ++q) v[q] = fmaxf(acc_read(acc[i][j][q]), 0.f);
                  *reinterpret_cast<float4*>(prow + i * 16) = make_float4(v[0], v[1], v[2], v[3]);
                }
              }
              CLT_FENCE();
            }
          }
          if (interior) stores_behind = 3;
        }
      }
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
}

// x[N, d] (row stride ldx bytes; stacked layers: FastArgs) -> xp[k-step][token][64 B], tokens padded to whole 256-token
// tiles with copies of the last row.  A 16-row LDS-DMA piece of x is then 1 KiB of consecutive bytes instead of 16
// segments of 64 B in 16 rows: the wave that issues it is held ~24 instead of ~42 cycles (tools/dma_cost.hip).
__global__ void pack_x_kernel(const char* __restrict__ x, int64_t ldx, int64_t x_layer_bytes, int64_t N_l, int64_t N,
                              int64_t Npad, int nsteps, uint4* __restrict__ xp, int* __restrict__ zero_ints, int n_zero,
                              int* __restrict__ zero2, int n_zero2) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g < n_zero) zero_ints[g] = 0;  // the call's fallback flags and counters (launch_fused_encode_topk)
  if (g < n_zero2) zero2[g] = 0;     // and the per-token candidate counts
  if (g >= Npad * nsteps * 4) return;
  const int c = (int)(g & 3);
  const int64_t t = (g >> 2) % Npad, ks = (g >> 2) / Npad;
  const int64_t tt = t < N ? t : N - 1;
  const int64_t layer = N_l ? tt / N_l : 0, local = N_l ? tt - layer * N_l : tt;
  xp[g] = *reinterpret_cast<const uint4*>(x + layer * x_layer_bytes + local * ldx + ks * ROWB + c * 16);
}

}  // namespace

void launch_pack_x(const FastArgs& a, void* xp, int* zero_ints, int n_zero, int* zero2, int n_zero2, hipStream_t st) {
  const int64_t Npad = (int64_t)a.ntt * GEMM_TN;
  const int nsteps = a.kbytes / ROWB;
  const int64_t n = Npad * nsteps * 4;
  hipLaunchKernelGGL(pack_x_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a.x,
                     a.ntt_layer ? a.ldx_bytes : (int64_t)a.kbytes, a.x_layer_bytes,
                     a.ntt_layer ? (int64_t)a.ntt_layer * GEMM_TN : (int64_t)0, a.N, Npad, nsteps, (uint4*)xp, zero_ints,
                     n_zero, zero2, n_zero2);
}

// K a multiple of 64 elements (the k loop is unrolled by two) and at least PF k-steps per tile (the loader, PF stages
// ahead, is then never more than one tile ahead, which is what the two aux areas allow); M a multiple of 16.
bool gemm_fast16_ok(int64_t M, int kbytes) { return kbytes % (2 * ROWB) == 0 && kbytes >= PF * ROWB && (M & 15) == 0; }

int launch_gemm_fast16(const FastArgs& a, int out_dtype, int grid, hipStream_t st) {
  if (out_dtype == CLT_BF16)
    hipLaunchKernelGGL((gemm_fast16_kernel<bf16_t>), dim3(grid), dim3(NTHREADS), 0, st, a);
  else
    hipLaunchKernelGGL((gemm_fast16_kernel<float>), dim3(grid), dim3(NTHREADS), 0, st, a);
  CLT_CHECK_LAUNCH("clt_encoder_gemm_relu");
  return CLT_OK;
}

}  // namespace clt
