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
#include <math.h>

#include "gemm_fast_dev.h"
#include "topk_select.h"

namespace clt {
namespace {


// MODE 0: store relu(pre) (OT = bf16 or f32).  MODE 1: per-token threshold filter, no pre matrix (see FastArgs).
template <typename OT, int MODE>
__global__ __launch_bounds__(NTHREADS, 1) void gemm_fast_kernel(const FastArgs p) {
  __shared__ __attribute__((aligned(1024))) char lds[NSTAGE * STAGE_BYTES + (MODE == 1 ? 4 * QLANE * 64 * 8 : 0)];
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
#ifdef CLT_TUNING  // timing ablations (tools/): 1 = no stores, 2 = no filter scan, 4 = no LDS-DMA, 8 = no MFMAs
  const int dbg = p.dbg;
#else
  constexpr int dbg = 0;
#endif
  const int* __restrict__ tt_map = p.tt_map;
  const int64_t N = p.N, M = p.M;
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

  auto coords = [&](int id, int& lt, int& tt) {  // see tile_of (gemm_fast.h); lt is the VIRTUAL latent tile
    const int full = nlt * GEMM_PANEL;
    const int pp = id / full, rem = id - pp * full;
    const int left = ntt - pp * GEMM_PANEL;
    const int pt = left < GEMM_PANEL ? left : GEMM_PANEL;
    lt = uni(rem / pt);
    tt = uni(pp * GEMM_PANEL + rem % pt);
    if (tt_map) tt = uni(tt_map[tt]);
  };

  // ---- loader state: runs three stages ahead of the multiplier, possibly in the next tile.  Past the last
  //      stage it keeps re-fetching the last tile into ring slots nobody reads again, so that every k-step issues
  //      the same NDMA operations and the counted waits need no special cases.
  int ld_t = 0, ld_ks = 0, ld_slot = 0;
  const char* ldW = Wb;
  const char* ldX = xb;
  const char* ldA = zero;  // per-lane source of the aux piece: bias chunk (lanes < nl <= 16) or tau chunk (lanes 16 .. 47)
  uint32_t voffW[4], voffX[4];
  const int nl = 64 * bias_esize / 16;  // lanes of the bias piece (this wave's 64 latents)
  const bool aux_lane = lane < nl || (MODE == 1 && lane >= 16 && lane < 48);
  const int ntt_layer = p.ntt_layer;
  const int ldx = ntt_layer ? (int)p.ldx_bytes : kbytes;  // bytes between consecutive token rows of x
  auto loader_tile = [&](int lt, int tt) {
    const int64_t lat0 = (int64_t)lt_actual(lt, p.lt_mode, p.lt_S) * GEMM_TM, tok0 = (int64_t)tt * GEMM_TN;
    const char* Wt = Wb;
    const char* bt = bias;
    if (ntt_layer) {  // stacked layers: this token tile's layer picks the table, the bias and the x block
      const int layer = uni(tt / ntt_layer);
      Wt = p.Wl[layer];
      bt = p.biasl[layer];
      ldX = xb + (int64_t)layer * p.x_layer_bytes + (int64_t)(tt - layer * ntt_layer) * GEMM_TN * ldx;
    } else {
      ldX = xb + tok0 * kbytes;
    }
    ldW = Wt + lat0 * kbytes;
    // rows past the end re-read the last row (their products are never stored)
    const int maxW = (int)(M - 1 - lat0 < GEMM_TM ? M - 1 - lat0 : GEMM_TM);
    const int maxX = (int)(N - 1 - tok0 < GEMM_TN ? N - 1 - tok0 : GEMM_TN);  // (stacked layers: N_l % 256 == 0)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = (wave + 4 * q) * 16 + (lane >> 2);
      const int c = (lane & 3) ^ ((r >> 2) & 3);
      voffW[q] = (uint32_t)((r < maxW ? r : maxW) * kbytes + c * 16);
      voffX[q] = (uint32_t)((r < maxX ? r : maxX) * ldx + c * 16);
    }
    const int epc = 16 / bias_esize;
    const int64_t lat = lat0 + 64 * wave + lane * epc;
    ldA = (bt && lat + epc <= M) ? bt + lat * bias_esize : zero;
    if (MODE == 1 && lane >= 16 && lane < 48)  // tau is padded to a multiple of 256 tokens
      ldA = reinterpret_cast<const char*>(p.tau + tok0 + wn * 128 + (lane - 16) * 4);
  };
  auto dma = [&](int q, uint32_t stage_lds, int kb0) {
    if (q < 4) {
      dma_s(ldW + kb0, voffW[q], stage_lds + (wave + 4 * q) * 1024);
    } else if (q < 8) {
      dma_s(ldX + kb0, voffX[q - 4], stage_lds + OP_BYTES + (wave + 4 * (q - 4)) * 1024);
    } else {
      if (aux_lane) dma_v(ldA, stage_lds + AUX_OFF + wave * AUX_WAVE);  // lane-linear: lane l lands at + 16 l
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
  // piece pc of a (stage, kk) fragment set, in the order the MFMAs need them: A0 B0 B1 B2 B3 A1 A2 A3
  auto read_piece = [&](const char* st, int kk, int pc, uint4 (&A)[4], uint4 (&B)[4]) {
    if (pc == 0) A[0] = *reinterpret_cast<const uint4*>(st + fa + pk[kk]);
    else if (pc <= 4) B[pc - 1] = *reinterpret_cast<const uint4*>(st + fb + (pc - 1) * 32 * ROWB + pk[kk]);
    else A[pc - 4] = *reinterpret_cast<const uint4*>(st + fa + (pc - 4) * 32 * ROWB + pk[kk]);
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
      const int off = wm * 128 + i * 32 + r32;  // latent in the tile: aux piece of wave off / 64
      const char* bp = st + AUX_OFF + (off >> 6) * AUX_WAVE + (off & 63) * bias_esize;
      float b;
      if (bias_esize == 4) b = *reinterpret_cast<const float*>(bp);
      else b = bf16_bits_to_f32(*reinterpret_cast<const uint16_t*>(bp));
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
  for (int pp = 0; pp < 3; ++pp) {
    const uint32_t st = lds0 + (uint32_t)ld_slot * STAGE_BYTES;
#pragma unroll
    for (int q = 0; q < NDMA; ++q) dma(q, st, ld_ks * ROWB);
    loader_advance();
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
  for (int pc = 0; pc < 8; ++pc) read_piece(lds, 0, pc, A0, B0);
  init_acc(lds);

  int ks = 0, t_idx = 0;
  // Stage s+1 has landed once only the operations issued after its LDS-DMAs are outstanding: the NDMA DMAs of
  // stage s+2, plus -- for the two k-steps after an epilogue -- the epilogue's stores, which are younger than the
  // DMAs of stage s+1.  vmcnt(n) is safe for any n <= that count (the counter saturates at 63): wait_allow.
  int stores_behind = 0, wait_allow = NDMA;
#pragma unroll 1
  for (int s = 0; s < total; ++s) {
    const char* cb = lds + (s & 3) * STAGE_BYTES;
    const char* nb = lds + ((s + 1) & 3) * STAGE_BYTES;
    const uint32_t ld_stage = lds0 + (uint32_t)ld_slot * STAGE_BYTES;
    const int kb0 = ld_ks * ROWB;
    const int allow = stores_behind > 0 ? wait_allow : NDMA;
    const bool last_k = ks + 1 == nsteps;

    CLT_FENCE();
#pragma unroll
    for (int t = 0; t < 32; ++t) {
      const int kk = t >> 4, i = (t >> 2) & 3, j = t & 3;
      if (kk == 0) acc[i][j] = mfma(A0[i], B0[j], acc[i][j]);
      else acc[i][j] = mfma(A1[i], B1[j], acc[i][j]);
      CLT_FENCE();
      if (t == 1) {
        if (allow >= 63) asm volatile("s_waitcnt vmcnt(63)\n\ts_barrier" ::: "memory");
        else if (allow >= 45) asm volatile("s_waitcnt vmcnt(45)\n\ts_barrier" ::: "memory");
        else if (allow >= 27) asm volatile("s_waitcnt vmcnt(27)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(9)\n\ts_barrier" ::: "memory");
      }
#ifdef CLT_TUNING  // ablations (results are garbage): 8 = no fragment reads, 4 = no LDS-DMA after the prologue
      if (!(dbg & 8)) {
        if (t < 8) read_piece(cb, 1, t, A1, B1);
        if (t >= 16 && t < 24) read_piece(nb, 0, t - 16, A0, B0);
      }
      if (!(dbg & 4) && t % 3 == 2 && t <= 26) dma((t - 2) / 3, ld_stage, kb0);
#else
      if (t < 8) read_piece(cb, 1, t, A1, B1);
      if (t >= 16 && t < 24) read_piece(nb, 0, t - 16, A0, B0);
      if (t % 3 == 2 && t <= 26) dma((t - 2) / 3, ld_stage, kb0);
#endif
      CLT_FENCE();
    }
    loader_advance();
    if (stores_behind > 0) --stores_behind;

    if (last_k) {
      const int lt_act = lt_actual(cur_lt, p.lt_mode, p.lt_S);
      const int64_t lat0 = (int64_t)lt_act * GEMM_TM, tok0 = (int64_t)cur_tt * GEMM_TN;
      // M is a multiple of 8, so a run of 4 latents starting at a multiple of 4 is inside or outside M as a whole
      // and the test does not depend on the lane half.
      const int lat_left = (int)(M - lat0 - wm * 128 < 128 ? M - lat0 - wm * 128 : 128);  // wave-uniform
      if constexpr (MODE == 0) {
        // ---- store epilogue: ReLU, one rounding, 4 consecutive latents per store.  One code path for interior
        //      and edge tiles (two paths make hipcc hoist all 256 accumulator reads above the branch: spills).
        if (!(dbg & 1)) {
          OT* pre = reinterpret_cast<OT*>(p.pre);
          const bool interior = tok0 + GEMM_TN <= N && lat_left == 128;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int64_t tok = tok0 + wn * 128 + j * 32 + r32;
            const bool tok_ok = tok < N;
            OT* prow = pre + tok * p.ldp + (int64_t)cur_lt * GEMM_TM + wm * 128 + 4 * h;
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
          if (interior) stores_behind = 2, wait_allow = NDMA + NSTORE;
        }
      } else {
        // ---- filter epilogue: an accumulator that reaches its token's threshold (the lower edge of the bf16
        //      rounding interval of the m-th best of the sample tiles, from pass A) is appended, as
        //      (bf16 bits << 16 | position in the list's 64 latents), to the private list of this (token, latent
        //      tile, latent half, lane half); the list's length goes to counts.  No atomics, no pre matrix.
        //      Scan: per group of 4 accumulators of one token, 4 reads + max-of-4 + ONE compare + one wave-uniform
        //      branch (9 instructions; ~63 % of the groups have no hit in any lane).  A group with a hit runs a short
        //      INLINE straight-line body (10 instructions, no EXEC change, no branch per element): EVERY lane writes
        //      the group's four values, rounded to bf16, as ONE 8-byte entry at its lane-private LDS queue cursor; only
        //      lanes with a hit advance the cursor and set the group's bit in a per-lane mask (a miss is overwritten by
        //      the next write).  The e-th queue entry of a lane belongs to the e-th set bit of its mask, so the entry
        //      needs no code.  A rolled loop after each QUARTER of the tile turns queue entries into list entries,
        //      testing each of the four values against the threshold's bf16 value (bf16(a) >= v_m is the fp32 test
        //      a >= tau up to the rounding interval's lower edge, which only ever ADDED candidates).
        //      (History: 64 unrolled out-of-line handlers, ~100 KB of code and an instruction-cache miss per hit group;
        //      then per-element 4-byte entries with a code, ~33 instructions per hit group.)
        const int list = (cur_lt * 2 + wm) * 2 + h;
        float tl[4];
        int t16[4];  // (threshold's bf16 value << 16), at least the smallest positive bf16: zeros carry nothing
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int64_t tok = tok0 + wn * 128 + j * 32 + r32;
          const float t = *reinterpret_cast<const float*>(cb + AUX_OFF + wave * AUX_WAVE + 256 + (j * 32 + r32) * 4);
          tl[j] = tok < N ? t : __builtin_inff();
          const uint32_t vm = (__float_as_uint(tl[j]) + 0x8000u) >> 16;
          t16[j] = (int)((vm < 1u ? 1u : vm) << 16);
        }
        // first entry of the (token j = 0, list) list, in entries; token j adds j * lstride (host checks 32 bits)
        const uint32_t lbase0 = (uint32_t)(((tok0 + wn * 128 + r32) * p.nL + list) * FUSED_SLOTS);
        const uint32_t lstride = 32u * (uint32_t)p.nL * FUSED_SLOTS;
        uint2* q = reinterpret_cast<uint2*>(lds + NSTAGE * STAGE_BYTES) + wave * (QLANE * 64);
        // byte cursor into this lane's queue: entry e of lane l sits at q[e * 64 + l]; a full queue keeps overwriting
        // its last slot while the cursor runs on and reports the overflow
        char* const qb = reinterpret_cast<char*>(q) + lane * 8;
        constexpr uint32_t QLAST = (QLANE - 1) * 512;
        uint32_t cntp = 0;  // the four list lengths of this lane, one byte each
        bool q_over = false;
        uint32_t qoff = 0;           // bytes: 512 per entry
        uint32_t gm[2] = {0u, 0u};   // bit (i * 4 + g) * 4 + j: this lane had a hit in that group
        const bool qflush = p.qflush != 0;  // wave-uniform (host: fused_plan)
        if (!(dbg & 2))
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          // Scan a quarter of the tile (32 latents x 128 tokens per wave) into the queue; the queue is turned into list
          // entries after the last quarter or -- when the host expects more than the queue's QLANE - 1 = 9 hit groups per
          // lane and tile to be likely (latent shards: ~2.3 per tile against ~0.5 unsharded) -- after every quarter.
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            if (i * 32 + 8 * g < lat_left) {
              // the four token groups of this (i, g) together: 16 reads, 8 max, 4 compares into SGPR pairs, THEN the four
              // wave-uniform branches -- one VALU -> SALU dependency stall per four groups instead of one per group
              float a[4][4];
              unsigned long long hm[4];
#pragma unroll
              for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) a[j][qq] = acc_read(acc[i][j][4 * g + qq]);
              }
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                // thresholds are positive, so a >= t  <=>  bits(a) >= bits(t) as SIGNED integers (a negative float is a
                // negative integer); integer max needs no NaN canonicalisation: v_max3_i32 + v_max_i32 + one compare
                const int mx = max(max(__float_as_int(a[j][0]), __float_as_int(a[j][1])),
                                   max(__float_as_int(a[j][2]), __float_as_int(a[j][3])));
                hm[j] = __ballot(mx >= __float_as_int(tl[j]));
              }
#pragma unroll
              for (int j = 0; j < 4; ++j) asm volatile("" : "+s"(hm[j]));  // all four masks exist before the first branch
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                if (hm[j] != 0ull) {
                  uint32_t flag;  // 1 in the lanes with a hit (the mask is already an SGPR pair: one v_cndmask)
                  asm("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(flag) : "s"(hm[j]));
                  const bf16x2_t p01 = {(__bf16)a[j][0], (__bf16)a[j][1]}, p23 = {(__bf16)a[j][2], (__bf16)a[j][3]};  // 2 x v_cvt_pk
                  *reinterpret_cast<uint2*>(qb + (qoff < QLAST ? qoff : QLAST)) =
                      make_uint2(__builtin_bit_cast(uint32_t, p01), __builtin_bit_cast(uint32_t, p23));
                  qoff += flag << 9;                                              // v_lshl_add_u32
                  gm[((i * 4 + g) * 4 + j) >> 5] |= flag << (((i * 4 + g) * 4 + j) & 31);  // v_lshl_or_b32
                }
              }
            }
            CLT_FENCE();
          }
          if (!qflush && i < 3) continue;
          // ---- queue -> lists (rolled)
          int nq = (int)(qoff >> 9);
          q_over = q_over || nq > QLANE - 1;  // (slot QLANE - 1 is where writes go once the queue is full: not an entry)
          nq = nq > QLANE - 1 ? QLANE - 1 : nq;
#pragma unroll 1
          for (int e = 0; __ballot(e < nq) != 0ull; ++e) {
            if (e < nq) {
              const uint2 ent = q[e * 64 + lane];
              const int gi = gm[0] != 0u ? __builtin_ctz(gm[0]) : 32 + __builtin_ctz(gm[1]);  // the e-th set bit
              if (gm[0] != 0u) gm[0] &= gm[0] - 1u;
              else gm[1] &= gm[1] - 1u;
              const uint32_t j = (uint32_t)gi & 3u, sh = j * 8u, pos0 = ((uint32_t)gi >> 2) * 4u;
              const int thr = j == 0 ? t16[0] : j == 1 ? t16[1] : j == 2 ? t16[2] : t16[3];
              const uint32_t lb = lbase0 + j * lstride;
#pragma unroll
              for (int qq = 0; qq < 4; ++qq) {
                const uint32_t w = qq < 2 ? ent.x : ent.y;
                const uint32_t vb = (qq & 1) ? (w & 0xffff0000u) : (w << 16);  // bf16 bits << 16
                if ((int)vb >= thr) {
                  const uint32_t c = (cntp >> sh) & 255u;
                  if (c < (uint32_t)FUSED_SLOTS && !(dbg & 1)) p.lists[lb + c] = vb | (pos0 + (uint32_t)qq);
                  cntp += 1u << sh;
                }
              }
            }
          }
          qoff = 0;
          gm[0] = gm[1] = 0u;
          CLT_FENCE();
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int64_t tok = tok0 + wn * 128 + j * 32 + r32;
          const uint32_t c = (cntp >> (8 * j)) & 255u;  // > FUSED_SLOTS: the merge kernel reports the overflow
          if (tok < N) p.counts[tok * p.nL + list] = (uint8_t)(q_over ? 255u : c);
        }
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

// tau[tok] = the smallest f32 whose bf16 rounding can reach the m-th best sample value of the token (half a bf16
// ulp below it: the rounding interval's lower edge; including the edge itself only adds candidates), or the
// smallest positive f32 when the sample holds fewer than m positive values.  Padding tokens get +inf.
__global__ void tau_kernel(const uint16_t* __restrict__ svals, int k, int m, int64_t N, int64_t Npad,
                           float* __restrict__ tau) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= Npad) return;
  float v = __builtin_inff();
  if (t < N) {
    const uint32_t bits = (uint32_t)svals[t * k + m - 1] << 16;
    v = bits == 0u ? __uint_as_float(1u) : __uint_as_float(bits - 0x8000u);
  }
  tau[t] = v;
}

// Row top-k of the SAMPLE matrix (pass A) + the token's threshold in one launch, for rows that fit the registers of a
// wave (bf16, k <= 32, at most 512 NCH <= 6144 columns): the results are clt_topk_rows' bit for bit (same keys, same
// WaveSelect), what differs is how few elements reach the selection.  The generic kernel streams a row chunk by chunk and
// learns its threshold on the way (~k ln(L / k) survivors, five or six queue merges per row, each chunk's load waited
// for in turn); here every load of the row is in flight at once, a packed integer max per lane (post-ReLU bf16: the
// signed 16-bit order is the value order, negative patterns stay below every candidate) gives 128 sub-maxima -- 128
// DISTINCT elements of the row -- whose k-th largest (a 16-step bit search over two ballots) is a lower bound of the
// row's k-th largest value, and only the elements at or above it (~40 at L = 6144, k = 32) are offered.  One merge
// per row.  HBM-bound: L * 2 bytes per row, read once (the sample matrix is still in the Infinity Cache).
// tau[t] = lower edge of the bf16 rounding interval of the sample's m-th value (tau_kernel's rule), +inf for the padding
// rows of the last token tile.
typedef short i16x2 __attribute__((ext_vector_type(2)));
template <int NCH>
__global__ __launch_bounds__(256) void sample_topk_kernel(const uint16_t* __restrict__ pre, int64_t ld, int64_t N,
                                                          int64_t Npad, int L, int k, int m,
                                                          uint16_t* __restrict__ svals, int64_t* __restrict__ sidx,
                                                          float* __restrict__ tau) {
  __shared__ u64 lds_q[4][192];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // (uniform: the token's scalars are scalar loads)
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;
  if (row >= Npad) return;
  if (row >= N) {
    if (lane == 0) tau[row] = __builtin_inff();
    return;
  }
  const uint16_t* rp = pre + row * ld;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  u32x4 c[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int e0 = (j * 64 + lane) * 8;  // (L is a multiple of 16: a chunk is inside the row or outside)
    c[j] = e0 < L ? *reinterpret_cast<const u32x4*>(rp + e0) : zero4;
  }
  i16x2 mx = {0, 0};
#pragma unroll
  for (int j = 0; j < NCH; ++j)
#pragma unroll
    for (int w = 0; w < 4; ++w) mx = __builtin_elementwise_max(mx, __builtin_bit_cast(i16x2, c[j][w]));
  const int lo = mx.x, hi = mx.y;  // >= 0
  const int kk = k < 128 ? k : 128;
  int T0 = 0;  // largest t with at least kk sub-maxima >= t (0: fewer than kk positive ones, every positive value is offered)
#pragma unroll
  for (int b = 14; b >= 0; --b) {
    const int t = T0 | (1 << b);
    const int cnt = __popcll(__ballot(lo >= t)) + __popcll(__ballot(hi >= t));
    if (cnt >= kk) T0 = t;
  }
  const int T1 = T0 > 1 ? T0 : 1;  // candidates are positive
  WaveSelect<1, 4, true> ws;
  ws.init(lds_q[wave], lane, k);
  ws.set_bits(48, 32 - __clz(L > 1 ? L - 1 : 1));
  // survivors -> the wave's LDS queue, densely (straight-line code: the row stays in registers); a row with more than
  // the queue takes (heavy ties at the bound) is redone below the generic way
  int cnt = 0;
  bool over = false, sorted = false;
  const i16x2 T1pk = {(short)T1, (short)T1};
#pragma unroll
  for (int j = 0; j < NCH; ++j)
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const uint32_t word = c[j][w];
      const i16x2 pr = __builtin_bit_cast(i16x2, word);
      // either half at the bound?  One packed saturating subtract: the sign bits of (half - T1) are clear where it is
      if (__ballot((~__builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(pr, T1pk)) & 0x80008000u) != 0u) == 0ull) continue;
      const bool c0 = pr.x >= T1, c1 = pr.y >= T1;
      const u64 b0 = __ballot(c0), b1 = __ballot(c1);
      if (!over) {  // (wave-uniform)
        if (cnt + __popcll(b0) + __popcll(b1) > 192) {
          over = true;
        } else {
          const unsigned col = (unsigned)((j * 64 + lane) * 8 + w * 2);
          const int p0 = cnt + __builtin_amdgcn_mbcnt_hi((unsigned)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b0, 0));
          if (c0) ws.q[p0] = ((u64)(word << 16) << 32) | (u64)(~col);
          cnt += __popcll(b0);
          const int p1 = cnt + __builtin_amdgcn_mbcnt_hi((unsigned)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b1, 0));
          if (c1) ws.q[p1] = ((u64)(word & 0xffff0000u) << 32) | (u64)(~(col + 1u));
          cnt += __popcll(b1);
        }
      }
    }
  if (!over && cnt <= 64) {  // the usual case: one key per lane, one 64-key bitonic sort, done
    wave_lds_fence();
    u64 k1[1] = {lane < cnt ? ws.q[lane] : 0ull};
    wave_lds_fence();
    bitonic_sort_desc<1>(k1, lane);
    ws.key[0] = k1[0];
    sorted = true;
  } else if (!over) {
    ws.cnt = cnt;
    ws.flush();
  } else {
    for (int e0 = lane * 8; e0 - lane * 8 < L; e0 += 512) {  // (wave-uniform trip count)
      const u32x4 ch = e0 < L ? *reinterpret_cast<const u32x4*>(rp + e0) : zero4;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t word = ch[i >> 1];
        const float v = __uint_as_float((i & 1) ? (word & 0xffff0000u) : (word << 16));
        ws.offer(v >= ws.thr_val && v > 0.f, v, (unsigned)(e0 + i));
      }
    }
    if (ws.cnt > 0) ws.flush();
  }
  if (!sorted) ws.finalize();
  write_out<1, 4>(ws, svals, CLT_BF16, row, sidx, k);
  const uint32_t bits = (uint32_t)(readlane_u64(ws.key[0], m - 1) >> 32);  // f32 pattern of the m-th value, 0 if there is none
  if (lane == 0) tau[row] = bits == 0u ? __uint_as_float(1u) : __uint_as_float(bits - 0x8000u);
}

// Final selection of the fused path: one wave per token merges the sample's top-k (pass A) with the token's
// candidate lists (pass B) under the same 64-bit key order as clt_topk_rows.  The threshold was the sample's m-th
// value, m <= k: every element of the row that reaches it was collected, so the merged top-k is the row's top-k
// if and only if its k-th value reaches the threshold too.  If it does not (or a list overflowed), *flag is raised
// and the caller's gated fallback launches redo the token batch the plain way.
template <int S, int R>
__global__ __launch_bounds__(256) void fused_select_kernel(const uint16_t* __restrict__ svals,
                                                           const int64_t* __restrict__ sidx,
                                                           const uint8_t* __restrict__ counts,
                                                           const uint32_t* __restrict__ lists, int nL, int lt_S, int64_t N,
                                                           int64_t M, int k, int m, bf16_t* __restrict__ out_vals,
                                                           int64_t* __restrict__ out_idx, int* __restrict__ flag) {
  constexpr int C = (R - S) * 64;
  __shared__ u64 lds_q[4][C];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // (uniform: the token's scalars are scalar loads)
  const int64_t tok = (int64_t)blockIdx.x * 4 + wave;
  if (tok >= N) return;
  WaveSelect<S, R, (S <= 2)> ws;
  ws.init(lds_q[wave], lane, k);
  ws.set_bits(48, 32 - __clz((int)(M > 1 ? M - 1 : 1)));  // bf16 values, latent indices < M
  // the sample's best k: column c of the sample matrix is latent (c / 256) * S * 256 + c % 256
  for (int e0 = 0; e0 < k; e0 += 64) {
    const int e = e0 + lane;
    float v = -1.f;
    unsigned col = 0;
    if (e < k) {
      v = bf16_bits_to_f32(svals[tok * k + e]);
      const int64_t c = sidx[tok * k + e];
      col = (unsigned)((c >> 8) * lt_S * 256 + (c & 255));
    }
    ws.offer(v > 0.f, v, col);
  }
  if (ws.cnt > 0) ws.flush();  // the threshold is now the sample's k-th value
  bool over = false;
  const uint8_t* crow = counts + tok * nL;
  const uint32_t* lrow = lists + tok * (int64_t)nL * FUSED_SLOTS;
  {
  for (int l0 = 0; l0 < nL; l0 += 64) {
    const int l = l0 + lane;
    int c = l < nL ? (int)crow[l] : 0;
    if (c > FUSED_SLOTS) over = true, c = FUSED_SLOTS;
    // list l = ((virtual tile * 2 + wm) * 2 + h); position pos = i * 16 + g * 4 + q
    const int vt = l >> 2, wm = (l >> 1) & 1, hh = l & 1;
    const unsigned lat_base = (unsigned)lt_actual(vt, LT_REST, lt_S) * 256u + wm * 128 + 4 * hh;
    for (int e = 0; __ballot(e < c) != 0ull; ++e) {  // while some lane's list still has an entry
      float v = -1.f;
      unsigned col = 0;
      if (e < c) {
        const uint32_t ent = lrow[(int64_t)l * FUSED_SLOTS + e];
        v = __uint_as_float(ent & 0xffff0000u);
        const unsigned pos = ent & 63u;
        col = lat_base + (pos >> 4) * 32 + ((pos >> 2) & 3) * 8 + (pos & 3);
      }
      ws.offer(v >= ws.thr_val && v > 0.f, v, col);
    }
  }
  }
  ws.flush();
  const float thr_m = bf16_bits_to_f32(svals[tok * k + m - 1]);
  if (__ballot(over) != 0ull || ws.thr_val < thr_m) {
    if (lane == 0) atomicOr(flag + 1 + tok / GEMM_TN, 1);  // this token's 256-token tile goes through the fallback
  }
  ws.finalize();
  write_out<S, R>(ws, out_vals, CLT_BF16, tok, out_idx, k);
}

// ---- candidate stream of the eight-wave filter kernel (encoder_gemm_w8.hip) -> per-token candidate lists.
// One block per region (= one wave of the filter launch): every entry {4 bf16 values, token, first latent} goes to a
// free slot of its token's list (the order of a list does not matter to the selection).  A region holds the entries of
// the 64 tokens of one tile quarter, for a few token tiles, and the 16 regions that share those tokens are processed at
// the same time: one global atomic per entry made every token a chain of ~170 same-address atomics (49 us for 1.4 M
// entries).  So a block first counts its entries per token in LDS (a small open-addressing table keyed by token),
// reserves each token's range with ONE global atomic, and then places the entries.
// A region that overflowed has dropped entries of unknown tokens: every token tile is flagged for the exact fallback.
constexpr int BUCKET_SLOTS = 1024;  // table slots per block (a power of two); more distinct tokens than fit go the slow way
__device__ __forceinline__ int bucket_slot(uint32_t* keys, uint32_t tok, bool insert) {
  uint32_t h = (tok * 2654435761u) >> 22;
  for (int probe = 0; probe < BUCKET_SLOTS; ++probe) {
    const uint32_t cur = insert ? atomicCAS(keys + h, 0xffffffffu, tok) : keys[h];
    if (cur == tok || (insert && cur == 0xffffffffu)) return (int)h;
    if (!insert && cur == 0xffffffffu) return -1;
    h = (h + 1) & (BUCKET_SLOTS - 1);
  }
  return -1;  // table full
}
__global__ __launch_bounds__(256) void fused_bucket_kernel(const u32x4* __restrict__ stream,
                                                           const int* __restrict__ scounts, int rcap, int tcap,
                                                           int* __restrict__ tcnt, u32x4* __restrict__ tlists,
                                                           int* __restrict__ flags, int ntt, int64_t N) {
  __shared__ uint32_t keys[BUCKET_SLOTS];
  __shared__ int lcnt[BUCKET_SLOTS], lbase[BUCKET_SLOTS];
  const int region = blockIdx.x;
  int n = scounts[region];
  if (n > rcap) {
    for (int t = threadIdx.x; t < ntt; t += blockDim.x) flags[1 + t] = 1;
    n = rcap;
  }
  for (int h = threadIdx.x; h < BUCKET_SLOTS; h += blockDim.x) keys[h] = 0xffffffffu, lcnt[h] = 0;
  __syncthreads();
  const u32x4* src = stream + (int64_t)region * rcap;
  // (a token >= N is never written by the filter kernel: a bound on what this kernel can touch all the same)
  for (int e = threadIdx.x; e < n; e += blockDim.x) {
    const uint32_t tok = src[e].z;
    if ((int64_t)tok >= N) continue;
    const int h = bucket_slot(keys, tok, true);
    if (h >= 0) atomicAdd(lcnt + h, 1);
  }
  __syncthreads();
  for (int h = threadIdx.x; h < BUCKET_SLOTS; h += blockDim.x) {
    const int c = lcnt[h];
    if (c > 0) lbase[h] = atomicAdd(tcnt + keys[h], c);
    lcnt[h] = 0;  // (now the cursor inside the reserved range)
  }
  __syncthreads();
  for (int e = threadIdx.x; e < n; e += blockDim.x) {
    const u32x4 v = src[e];
    const uint32_t tok = v.z;
    if ((int64_t)tok >= N) continue;
    const int h = bucket_slot(keys, tok, false);
    const int pos = h >= 0 ? lbase[h] + atomicAdd(lcnt + h, 1) : atomicAdd(tcnt + tok, 1);  // (table full: one by one)
    if (pos < tcap) tlists[(int64_t)tok * tcap + pos] = v;
  }
}

// Final selection of the fused path on the per-token lists: one wave per token merges the sample's top-k (pass A) with
// the token's candidates under the same 64-bit key order as clt_topk_rows, and verifies as fused_select_kernel does.
template <int S, int R>
__global__ __launch_bounds__(256) void fused_select_stream_kernel(const uint16_t* __restrict__ svals,
                                                                  const int64_t* __restrict__ sidx,
                                                                  const int* __restrict__ tcnt,
                                                                  const u32x4* __restrict__ tlists, int tcap, int lt_S,
                                                                  int64_t N, int64_t M, int k, int m,
                                                                  bf16_t* __restrict__ out_vals,
                                                                  int64_t* __restrict__ out_idx, int* __restrict__ flag) {
  constexpr int C = (R - S) * 64;
  __shared__ u64 lds_q[4][C];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // (uniform: the token's scalars are scalar loads)
  const int64_t tok = (int64_t)blockIdx.x * 4 + wave;
  if (tok >= N) return;
  WaveSelect<S, R, (S <= 2)> ws;
  ws.init(lds_q[wave], lane, k);
  ws.set_bits(48, 32 - __clz((int)(M > 1 ? M - 1 : 1)));  // bf16 values, latent indices < M
  int n = tcnt[tok];
  const bool over = n > tcap;  // (wave-uniform)
  n = over ? tcap : n;
  const u32x4* lrow = tlists + tok * (int64_t)tcap;
  const u32x4 none = {0u, 0u, 0u, 0u};
  u32x4 ent = lane < n ? lrow[lane] : none;  // (in flight while the sample's entries are offered)
  // The selection starts AT the token's threshold (the sample's m-th value): the result stands only if its k-th value
  // reaches the threshold (the check below), and then nothing under the threshold is part of it -- not the sample's
  // entries m + 1 .. k either.  Until round 3 all k sample entries were offered and merged first (one queue merge of
  // ~480 instructions per token) to raise the selection's threshold to the sample's k-th value, which is lower.
  const float thr_m = bf16_bits_to_f32(svals[tok * k + m - 1]);
  ws.thr_val = thr_m;
  ws.thr_key = thr_m > 0.f ? (((u64)__float_as_uint(thr_m)) << 32) - 1ull : 0ull;  // keys of value >= thr_m pass
  // the sample's best k: column c of the sample matrix is latent c % 16 of group (c / 16) S + S - 1 (grp_actual)
  for (int e0 = 0; e0 < k; e0 += 64) {
    const int e = e0 + lane;
    float v = -1.f;
    unsigned col = 0;
    if (e < k) {
      v = bf16_bits_to_f32(svals[tok * k + e]);
      const int64_t c = sidx[tok * k + e];
      col = (unsigned)grp_actual((int)(c >> 4), LT_GSAMPLE, lt_S) * 16u + (unsigned)(c & 15);
    }
    ws.offer(v > 0.f && v >= thr_m, v, col);
  }
  for (int e0 = 0; e0 < n; e0 += 64) {
    const u32x4 cur = ent;
    const bool have = e0 + lane < n;
    if (e0 + 64 < n) ent = e0 + 64 + lane < n ? lrow[e0 + 64 + lane] : none;
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const uint32_t w = qq < 2 ? cur.x : cur.y;
      const float v = __uint_as_float((qq & 1) ? (w & 0xffff0000u) : (w << 16));
      ws.offer(have && v >= ws.thr_val && v > 0.f, v, cur.w + (unsigned)qq);
    }
  }
  ws.flush();
  if (over || ws.thr_val < thr_m) {
    if (lane == 0) atomicOr(flag + 1 + tok / GEMM_TN, 1);  // this token's 256-token tile goes through the fallback
  }
  ws.finalize();
  write_out<S, R>(ws, out_vals, CLT_BF16, tok, out_idx, k);
}

// flags[1 .. ntt] -> flags[0] = number of flagged token tiles, flags[1 + ntt ..] = their ids in ascending order.
// One wave: ballot + prefix count per 64 tiles (order preserving).
__global__ void fused_compact_kernel(int* __restrict__ flags, int ntt) {
  const int lane = threadIdx.x;
  int base = 0;
  for (int t0 = 0; t0 < ntt; t0 += 64) {
    const int t = t0 + lane;
    const bool f = t < ntt && flags[1 + t] != 0;
    const unsigned long long b = __ballot(f);
    if (f) flags[1 + ntt + base + __builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0))] = t;
    base += __popcll(b);
  }
  if (lane == 0) flags[0] = base;
}

}  // namespace

bool gemm_fast_supported(int in_dtype, int64_t N, int64_t M, int d) {
  const int64_t kbytes = (int64_t)d * 2;
  const int64_t nlt = (M + GEMM_TM - 1) / GEMM_TM, ntt = (N + GEMM_TN - 1) / GEMM_TN;
  return in_dtype == CLT_BF16 && N > 0 && kbytes % ROWB == 0 && kbytes >= ROWB && (M & 7) == 0 &&
         kbytes * GEMM_TM < (1ll << 31) && nlt * ntt * (kbytes / ROWB) < (1ll << 30);
}

static int grid_for(int64_t ntiles, int cus) { return (int)(ntiles < cus ? ntiles : cus); }

int launch_gemm_fast(FastArgs a, int out_dtype, bool filter, int cus, hipStream_t st) {
  const int grid = grid_for((int64_t)a.nlt * a.ntt, cus);
  if (grid <= 0) return CLT_OK;
  if (gemm_fast16_ok(a.M, a.kbytes) && !(a.dbg & 64)) {  // (-DCLT_TUNING builds: CLT_GEMM_DEBUG 64 = the 32x32x16 kernels)
    if (a.ngroups <= 0) a.ngroups = (int)(a.M / 16);  // every group of W (M % 16 == 0 here)
    if (filter) return launch_gemm_w8_filter(a, grid, st);  // eight waves, candidate stream (encoder_gemm_w8.hip)
    return launch_gemm_fast16(a, out_dtype, grid, st);
  }
  if (filter)
    hipLaunchKernelGGL((gemm_fast_kernel<bf16_t, 1>), dim3(grid), dim3(NTHREADS), 0, st, a);
  else if (out_dtype == CLT_BF16)
    hipLaunchKernelGGL((gemm_fast_kernel<bf16_t, 0>), dim3(grid), dim3(NTHREADS), 0, st, a);
  else
    hipLaunchKernelGGL((gemm_fast_kernel<float, 0>), dim3(grid), dim3(NTHREADS), 0, st, a);
  CLT_CHECK_LAUNCH("clt_encoder_gemm_relu");
  return CLT_OK;
}

// ---- fused encode + top-k ------------------------------------------------------------------------------------
FusedPlan fused_plan(int in_dtype, int out_dtype, int64_t N, int64_t M, int d, int k, int cus) {
  FusedPlan pl = {};
  if (!gemm_fast_supported(in_dtype, N, M, d) || out_dtype != CLT_BF16 || k < 1 || k > 256) return pl;
  // Sample stride: pass A (store mode, ~1.4x the cost per tile of the filter mode, plus a row top-k over M / S columns)
  // wants S large; the candidates (~m * S per token: filter epilogue, bucket and merge kernels) want it small.  S up to
  // M / (40 k), capped at 16 (swept again on the eight-wave kernel: 12 / 16 / 20 / 24 / 32, 16 stays).  Latent shards
  // (M / tp rows) still take the fused path: M = 12288, k = 32 -> S = 9; M = 16384, k = 64 -> S = 6.
  int64_t S = M / (tuning_env("CLT_FUSED_SDIV", 40) * (int64_t)k);
  const int64_t s_cap = tuning_env("CLT_FUSED_SCAP", 16);  // -DCLT_TUNING builds only
  if (S > s_cap) S = s_cap;
  if (S < 2) return pl;
  const int64_t nlt = (M + GEMM_TM - 1) / GEMM_TM, ntt = (N + GEMM_TN - 1) / GEMM_TN;
  pl.l16 = gemm_fast16_ok(M, d * 2) && !(tuning_env("CLT_GEMM_DEBUG", 0) & 64);
  int64_t nltA, nltB;
  if (pl.l16) {
    // 16x16x32 kernels: the sample is every S-th GROUP of 16 latent rows (grp_actual, gemm_fast_dev.h), spread over all
    // tiles; sample tiles and filter tiles are runs of 16 virtual groups
    const int64_t G = M / 16;
    pl.ngA = (int)(G / S);
    pl.ngB = (int)(G - G / S);
    nltA = (pl.ngA + 15) / 16;
    nltB = (pl.ngB + 15) / 16;
    pl.Ls = (int64_t)pl.ngA * 16;
    // group offsets inside a tile are 32-bit byte offsets from the tile's first group: at most 17 S groups apart
    if ((17 * S * 16 + 16) * (int64_t)d * 2 >= (1ll << 31)) return pl;
  } else {
    nltA = (nlt + S - 1) / S, nltB = nlt - nltA;
    // valid columns of the sample matrix: only its last tile can be W's (partial) last tile
    const int64_t lastA = (nltA - 1) * S;
    const int64_t last_valid = (M - lastA * GEMM_TM) < GEMM_TM ? (M - lastA * GEMM_TM) : GEMM_TM;
    pl.Ls = (nltA - 1) * GEMM_TM + last_valid;
  }
  if (nltA < 1 || nltB < 1) return pl;
  if (pl.Ls < k) return pl;
  pl.S = (int)S;
  // Threshold rank m <= k: the sample's m-th value admits about m * S candidates per token instead of k * S.  The
  // row then holds fewer than k elements at or above it only if fewer than k - m of them fall outside the sample,
  // which for exchangeable latents is P(Poisson((k - m) / (S - 1)) >= m) (the sample's m-th value sits where the
  // row's density is m per sample size): take the smallest m that keeps this under 1e-6 per token.  A failing token
  // costs one 256-token tile of plain GEMM + top-k (the per-tile fallback: ~60 us at the C2 shape, and only for that
  // tile), so one failure per ~10 layer calls of 8192 tokens is far cheaper than the extra candidates of a safer
  // rank (m = 11 instead of 16 at C2: 176 candidates per token instead of 256).  Results are exact either way.
  pl.m = k;
  for (int m = 1; m <= k; ++m) {
    const double lam = (double)(k - m) / (double)(S - 1);
    double tail = 0.0;
    if (lam > 0.0) {
      // P(X >= m) <= pmf(m) * 1 / (1 - lam / (m + 1)) for lam < m + 1
      if (lam >= m + 1) continue;
      const double logp = -lam + m * log(lam) - lgamma((double)m + 1.0);
      tail = exp(logp) / (1.0 - lam / (m + 1.0));
    }
    if (tail < 1e-6) {
      pl.m = m;
      break;
    }
  }
  // (32x32x16 filter kernel only) hit groups per lane and tile: 4 lists x 64 latents x (m S / M) candidates per latent.
  // Its queue holds 9 per lane between flushes; P(Poisson(0.6) >= 10) ~ 1e-9 per lane and tile
  pl.qflush = 256.0 * pl.m * (double)S / (double)M > 0.6 ? 1 : 0;
  pl.nltA = (int)nltA;
  pl.nltB = (int)nltB;
  pl.ntt = (int)ntt;
  pl.nL = (int)nltB * 4;  // (32x32x16 kernels: lists of 64 latents)
  // the 32x32x16 filter kernel addresses list entries with 32 bits
  if (!pl.l16 && (int64_t)ntt * GEMM_TN * pl.nL * FUSED_SLOTS >= (1ll << 32)) return pl;
  pl.lds = nltA * GEMM_TM;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off = align_up(off + bytes, 256); return o; };
  pl.off_pre_s = take((size_t)N * pl.lds * 2);
  pl.off_svals = take((size_t)N * k * 2);
  pl.off_sidx = take((size_t)N * k * 8);
  pl.off_tau = take((size_t)ntt * GEMM_TN * 4);
  if (pl.l16) {
    // Eight-wave filter kernel: 16-byte candidate entries in one stream region per wave, then per-token lists.  About
    // m * S entries per token are expected whatever the data (that is what the sampled threshold fixes).  An overflowing
    // list costs its token tile's fallback, an overflowing region every tile's.
    const int64_t tilesB = nltB * ntt;
    pl.nregions = (int)(tilesB < cus ? tilesB : cus) * 8;
    const double per_tok = (double)pl.m * (double)S;
    // (a region sums thousands of (token, tile) pairs: 1.5 x its expected share + 256 is far out; a token's own count
    // follows the sampled threshold's rank-m statistic -- Gamma(m) -- and gets 3 x)
    pl.rcap = (int)align_up((size_t)(1.5 * per_tok * (double)N / pl.nregions) + 256, 64);
    pl.tcap = (int)align_up((size_t)(3.0 * per_tok) + 64, 64);
    pl.off_xp = take((size_t)ntt * GEMM_TN * d * 2);  // packed x (16x16x32 kernels)
    pl.off_stream = take((size_t)pl.nregions * pl.rcap * 16);
    pl.off_scounts = take((size_t)pl.nregions * 4);
    pl.off_tcnt = take((size_t)N * 4);
    pl.off_tlists = take((size_t)N * pl.tcap * 16);
  } else {
    pl.off_counts = take((size_t)N * pl.nL);
    pl.off_lists = take((size_t)N * pl.nL * FUSED_SLOTS * 4);
  }
  pl.bytes = off;
  pl.ok = true;
  return pl;
}

int launch_fused_encode_topk(const FusedPlan& pl, const void* x, const void* W, const void* bias, int bias_dtype,
                             int64_t N, int64_t M, int d, int k, void* out_vals, int64_t* out_idx, char* ws, int* flag,
                             int cus, int dbg, hipStream_t st, const FusedLayers* layers) {
  FastArgs a = {};
  if (layers) {  // N = n_layers * N_l stacked tokens; every launch below sees one problem of N tokens
    a.ntt_layer = (int)(N / layers->n_layers / GEMM_TN);
    a.ldx_bytes = layers->ldx_bytes;
    a.x_layer_bytes = layers->x_layer_bytes;
    for (int l = 0; l < layers->n_layers; ++l) {
      a.Wl[l] = (const char*)layers->W[l];
      a.biasl[l] = layers->bias ? (const char*)layers->bias[l] : nullptr;
    }
  }
  a.x = (const char*)x;
  a.W = (const char*)W;
  a.bias = (const char*)bias;
  a.bias_esize = bias_dtype == CLT_BF16 ? 2 : 4;
  a.N = N;
  a.M = M;
  a.kbytes = d * 2;
  a.ntt = pl.ntt;
  a.lt_S = pl.S;
  a.dbg = dbg;
  // The flag area (gemm_fast.h) starts at zero; with the 16x16x32 kernel the x-packing launch clears it (one launch
  // fewer: a memset costs ~5 us on a stream that has nothing else to run).  (Letting the merge kernel's last wave
  // compact the flagged tile ids was tried too: a fence + one counter for 8192 waves took the kernel from 50 to 357 us.)
  if (!pl.l16 && hipMemsetAsync(flag, 0, (size_t)(1 + pl.ntt) * sizeof(int), st) != hipSuccess) {
    set_error("clt_encode_topk_fwd: hipMemsetAsync failed");
    return CLT_ERR_LAUNCH;
  }
  if (pl.l16) {  // pass A and pass B read x from a k-step-major copy (the rare fallback reads x in place)
    launch_pack_x(a, ws + pl.off_xp, flag, 1 + pl.ntt, (int*)(ws + pl.off_tcnt), (int)N, st);
    a.xp = ws + pl.off_xp;
    a.xp_kstep_bytes = (int64_t)pl.ntt * GEMM_TN * ROWB;
  }
  // pass A: every S-th latent tile, stored, and its row top-k -> the token thresholds
  FastArgs pa = a;
  pa.nlt = pl.nltA;
  pa.lt_mode = pl.l16 ? LT_GSAMPLE : LT_SAMPLE;
  pa.ngroups = pl.ngA;
  pa.pre = ws + pl.off_pre_s;
  pa.ldp = pl.lds;
  int rc = launch_gemm_fast(pa, CLT_BF16, false, cus, st);
  if (rc != CLT_OK) return rc;
  uint16_t* svals = (uint16_t*)(ws + pl.off_svals);
  int64_t* sidx = (int64_t*)(ws + pl.off_sidx);
  float* tau = (float*)(ws + pl.off_tau);
  const int64_t Npad = (int64_t)pl.ntt * GEMM_TN;
  // register-resident rows: top-k and threshold in one launch.  Measured against the generic kernel + tau_kernel (same
  // box, whole call): N = 65536, M = 12288, k = 32 (tp8 shard) 1.408 against 1.461 ms; C2 equal (1.065 / 1.063); k = 64
  // at M = 131072 (~90 survivors: two queue merges, 78 registers) 3.263 against 3.222 ms -- so k <= 32 only.
  if (pl.l16 && k <= 32 && pl.Ls <= 6144 && !(dbg & 128)) {
    const dim3 sg((unsigned)(Npad / 4));
    const uint16_t* pre_s = (const uint16_t*)(ws + pl.off_pre_s);
    const int nch = (int)((pl.Ls + 511) / 512);
#define CLT_SAMPLE_TOPK(NCH)                                                                                      \
  hipLaunchKernelGGL((sample_topk_kernel<NCH>), sg, dim3(256), 0, st, pre_s, pl.lds, N, Npad, (int)pl.Ls, k, pl.m, svals, \
                     sidx, tau)
    if (nch <= 4)
      CLT_SAMPLE_TOPK(4);
    else if (nch <= 8)
      CLT_SAMPLE_TOPK(8);
    else
      CLT_SAMPLE_TOPK(12);
#undef CLT_SAMPLE_TOPK
  } else {
    rc = topk_rows_gated(ws + pl.off_pre_s, CLT_BF16, pl.lds, nullptr, nullptr, N, pl.Ls, k, svals, CLT_BF16, sidx,
                         nullptr, 0, st);
    if (rc != CLT_OK) return rc;
    hipLaunchKernelGGL(tau_kernel, dim3((unsigned)((Npad + 255) / 256)), dim3(256), 0, st, svals, k, pl.m, N, Npad, tau);
  }
  // pass B: all other tiles, filtered against the thresholds
  FastArgs pb = a;
  pb.nlt = pl.nltB;
  pb.lt_mode = pl.l16 ? LT_GREST : LT_REST;
  pb.ngroups = pl.ngB;
  pb.tau = tau;
  const dim3 grid((unsigned)((N + 3) / 4));
  if (pl.l16) {
    pb.stream = ws + pl.off_stream;
    pb.stream_counts = (int*)(ws + pl.off_scounts);
    pb.stream_rcap = pl.rcap;
    pb.fb_flags = flag;
    rc = launch_gemm_fast(pb, CLT_BF16, true, cus, st);  // (its grid: the pl.nregions / 8 of fused_plan)
    if (rc != CLT_OK) return rc;
    int* tcnt = (int*)(ws + pl.off_tcnt);
    const u32x4* tlists = (const u32x4*)(ws + pl.off_tlists);
    hipLaunchKernelGGL(fused_bucket_kernel, dim3(pl.nregions), dim3(256), 0, st, (const u32x4*)pb.stream,
                       pb.stream_counts, pl.rcap, pl.tcap, tcnt, (u32x4*)(ws + pl.off_tlists), flag, pl.ntt, N);
    if (k <= 64)
      hipLaunchKernelGGL((fused_select_stream_kernel<1, 4>), grid, dim3(256), 0, st, svals, sidx, tcnt, tlists, pl.tcap,
                         pl.S, N, M, k, pl.m, (bf16_t*)out_vals, out_idx, flag);
    else if (k <= 128)
      hipLaunchKernelGGL((fused_select_stream_kernel<2, 8>), grid, dim3(256), 0, st, svals, sidx, tcnt, tlists, pl.tcap,
                         pl.S, N, M, k, pl.m, (bf16_t*)out_vals, out_idx, flag);
    else
      hipLaunchKernelGGL((fused_select_stream_kernel<4, 8>), grid, dim3(256), 0, st, svals, sidx, tcnt, tlists, pl.tcap,
                         pl.S, N, M, k, pl.m, (bf16_t*)out_vals, out_idx, flag);
  } else {
    pb.lists = (uint32_t*)(ws + pl.off_lists);
    pb.counts = (uint8_t*)(ws + pl.off_counts);
    pb.nL = pl.nL;
    pb.qflush = pl.qflush;
    rc = launch_gemm_fast(pb, CLT_BF16, true, cus, st);
    if (rc != CLT_OK) return rc;
    if (k <= 64)
      hipLaunchKernelGGL((fused_select_kernel<1, 4>), grid, dim3(256), 0, st, svals, sidx, pb.counts, pb.lists, pl.nL,
                         pl.S, N, M, k, pl.m, (bf16_t*)out_vals, out_idx, flag);
    else if (k <= 128)
      hipLaunchKernelGGL((fused_select_kernel<2, 8>), grid, dim3(256), 0, st, svals, sidx, pb.counts, pb.lists, pl.nL,
                         pl.S, N, M, k, pl.m, (bf16_t*)out_vals, out_idx, flag);
    else
      hipLaunchKernelGGL((fused_select_kernel<4, 8>), grid, dim3(256), 0, st, svals, sidx, pb.counts, pb.lists, pl.nL,
                         pl.S, N, M, k, pl.m, (bf16_t*)out_vals, out_idx, flag);
  }
  CLT_CHECK_LAUNCH("clt_encode_topk_fwd");
  // gated fallback (a list overflowed / a token failed verification): the flagged 256-token tiles the plain way, into
  // the same outputs.  The pre rows of those tiles land on the front of the workspace, over the lists (consumed by now).
  hipLaunchKernelGGL(fused_compact_kernel, dim3(1), dim3(64), 0, st, flag, pl.ntt);
  FastArgs pf = a;
  pf.xp = nullptr;  // (its pre rows may land on the packed copy)
  pf.nlt = (int)((M + GEMM_TM - 1) / GEMM_TM);
  pf.lt_mode = LT_ALL;
  pf.ngroups = 0;
  pf.pre = ws;
  pf.ldp = M;
  pf.gate = flag;
  pf.tt_map = flag + 1 + pl.ntt;
  rc = launch_gemm_fast(pf, CLT_BF16, false, cus, st);
  if (rc != CLT_OK) return rc;
  return topk_rows_gated(ws, CLT_BF16, M, nullptr, nullptr, N, M, k, out_vals, CLT_BF16, out_idx, flag, GEMM_TN, st);
}

}  // namespace clt
