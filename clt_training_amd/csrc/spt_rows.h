// Row pass over prebuilt segments, shared by segments.hip (clt_spt_dense, clt_spt_dense_adam) and decode_rows.hip
// (clt_decode_bwd_rows*): one wave owns RW consecutive rows of an [M, d] table.
#pragma once
#include <math.h>
#include <stdlib.h>

#include "common.h"

namespace clt {

// Token of entry e = n * k + j.  k is a power of two in every BASELINE config: the shift replaces the ~25
// instructions of an emulated 32-bit division (the row passes are issue-bound).  Wave-uniform branch.
__device__ __forceinline__ int entry_token(int e, int k) {
  return (k & (k - 1)) == 0 ? (e >> (31 - __clz(k))) : e / k;
}


template <typename OT, int E>
__device__ __forceinline__ void store_row_chunk(OT* p, const float (&a)[E]) {
  if constexpr (sizeof(OT) == 2 && E == 8) {
    nt_store8(p, a);  // the gradient is read back once, by the optimizer, after every other kernel of the step
  } else if constexpr (sizeof(OT) == 4 && E == 4) {
    Chunk<float>::store(p, a);
  } else if constexpr (sizeof(OT) == 4 && E == 8) {
    reinterpret_cast<float4*>(p)[0] = make_float4(a[0], a[1], a[2], a[3]);
    reinterpret_cast<float4*>(p)[1] = make_float4(a[4], a[5], a[6], a[7]);
  } else {
    uint2 u2;
    u2.x = pack_bf16x2(a[0], a[1]);
    u2.y = pack_bf16x2(a[2], a[3]);
    *reinterpret_cast<uint2*>(p) = u2;
  }
}

// One wave owns RW consecutive output rows.  Their segment bounds (one load) and all their entries (one
// 64-lane load per 64 entries: entry id, weight, source row, owning output row) are fetched up front, so the
// dependent chain offsets -> perm -> dense row is paid once per RW rows instead of once per row; the gathers of
// consecutive output rows then issue back to back (most rows hold 0-8 entries).  Rows are finished in order:
// when the owning row of the next entry changes, the accumulator is stored (zeros for empty rows) and cleared.
// RW = 8 when there are rows to spare (M / 8 waves still fill the chip); fewer rows per wave when the table is
// short and its rows long (a latent shard of a large token batch: M / tp rows, N * k / M * tp entries each).
// The summation order within a row does not depend on RW: results are bitwise the same.
static int pick_rw(int64_t M) { return M >= 8 * 8192 ? 8 : M >= 4 * 8192 ? 4 : M >= 2 * 8192 ? 2 : 1; }

// Optimizer state of the rows for the fused variant (clt_spt_dense_adam): the finished gradient row never goes
// to memory; it is rounded to the parameter dtype (as the .grad tensor would be) and consumed on the spot by
// the same update arithmetic as adam.hip.
struct AdamRows {
  void* m;
  void* v;
  float step_size, b1, b2, inv_sqrt_bc2, eps;
  int project;  // SAE with a unit-norm decoder: remove the gradient's component along the row first
                // (remove_gradient_parallel_to_decoder_directions, sparse_coder.py:696-711)
};

// g - (g . w) w for one row chunk, with the row's dot product already reduced; one rounding to the gradient dtype
// (the reference updates W_dec.grad in place).  Shared by the fused row passes and clt_rows_project_out.
template <bool BF16>
__device__ __forceinline__ float project_elem(float g, float w, float dot) {
  const float r = fmaf(-dot, w, g);
  return BF16 ? bf16_bits_to_f32(f32_to_bf16_bits(r)) : r;
}

// The same host arithmetic as clt_adam_step (adam.hip): bias corrections in double, one rounding each.
static inline AdamRows make_adam_rows(void* m, void* v, float lr, float b1, float b2, float eps, int step,
                                      int project = 0) {
  const double bc1 = 1.0 - pow((double)b1, (double)step);
  const double bc2 = 1.0 - pow((double)b2, (double)step);
  AdamRows ad;
  ad.m = m;
  ad.v = v;
  ad.step_size = lr * 1.0f * (float)(1.0 / bc1);
  ad.b1 = b1;
  ad.b2 = b2;
  ad.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  ad.eps = eps;
  ad.project = project;
  return ad;
}

// Decoder backward in the same row pass (clt_decode_bwd_rows*): the table row IS the decoder row W[r,:], the
// gathered dense rows are the loss gradient G[n,:], so the gradient wrt the bag weight of entry e = (n, j),
//   ga = G[n,:] . W[r,:]                                             (xformers.py:167-171),
// costs one dot product against registers that are already loaded -- no second gather of W rows per token.  The
// backward of the post-encoder affine (sparse_coder.py:136-140) and its per-latent sums follow on the spot.
struct BwdActs {
  const void* v;            // [N*k] top-k values (v_dtype): the (v > 0) mask and d post_enc_scale
  int v_dtype;
  const float* post_enc;    // [M] or null
  const float* post_scale;  // [M] or null
  float* grad_acts;         // [N*k] fp32, accumulated into (the detached leaf's .grad, sparse_coder.py:72-76)
  float* g_post;            // [M] or null:  sum_e u * (v > 0)
  float* g_scale;           // [M] or null:  sum_e ga * (v + post_enc * (v > 0))
  const void* W;            // [M, d] table rows when they are not the Adam parameter (plain gradient output)
};

template <int E>
__device__ __forceinline__ void store_elems_f32(float* p, const float (&a)[E]) {
#pragma unroll
  for (int q = 0; q < E; q += 4) *reinterpret_cast<float4*>(p + q) = make_float4(a[q], a[q + 1], a[q + 2], a[q + 3]);
}
template <int E>
__device__ __forceinline__ void load_elems_f32(const float* p, float (&a)[E]) {
#pragma unroll
  for (int q = 0; q < E; q += 4) {
    const float4 f = *reinterpret_cast<const float4*>(p + q);
    a[q] = f.x; a[q + 1] = f.y; a[q + 2] = f.z; a[q + 3] = f.w;
  }
}

template <typename PT, int E>
__device__ __forceinline__ void load_row_chunk(const PT* p, float (&a)[E]) {
  if constexpr (sizeof(PT) == 2 && E == 8) {
    nt_load8(p, a);
  } else if constexpr (sizeof(PT) == 4) {
#pragma unroll
    for (int q = 0; q < E; q += 4) {
      const float4 f = *reinterpret_cast<const float4*>(p + q);
      a[q] = f.x; a[q + 1] = f.y; a[q + 2] = f.z; a[q + 3] = f.w;
    }
  } else {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    a[0] = bf16_lo(u.x); a[1] = bf16_hi(u.x); a[2] = bf16_lo(u.y); a[3] = bf16_hi(u.y);
  }
}

template <typename DT, typename OT, int CH, int U, bool ADAM, int RW, bool BWD = false>
__global__ __launch_bounds__(256) void spt_dense_kernel(const int32_t* __restrict__ perm,
                                                        const int32_t* __restrict__ offsets, int64_t M,
                                                        const float* __restrict__ vals,
                                                        const void* __restrict__ gate, int gate_dtype,
                                                        const DT* __restrict__ dense, int d, int k,
                                                        OT* __restrict__ out, float* __restrict__ rowsum,
                                                        const AdamRows ad, const BwdActs bw = BwdActs()) {
  constexpr int E = Chunk<DT>::E;
  const int lane = threadIdx.x & 63;
  const int64_t r0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RW;
  if (r0 >= M) return;
  const int nrows = (int)min((int64_t)RW, M - r0);
  const int nchunks = d / E;
  const int my_off = lane <= nrows ? offsets[r0 + lane] : 0;  // lanes 0..nrows hold the RW+1 bounds
  const int beg = readlane_i(my_off, 0), end = readlane_i(my_off, nrows);
  // The row bounds as wave-uniform values, read HERE, in uniform control flow.  A v_readlane inside the divergent
  // `if (pos < end)` below may name a lane that is inactive there, and under register pressure hipcc spills and
  // restores my_off around such a region for the active lanes only: the inactive lanes then hold garbage and rows
  // with few entries landed on neighbouring rows (d = 4096: the fused variants need > 256 VGPRs).
  int bnd[RW];
#pragma unroll
  for (int q = 0; q < RW; ++q) bnd[q] = q < nrows ? readlane_i(my_off, q) : 0x7fffffff;

  float acc[CH][E];
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int e = 0; e < E; ++e) acc[c][e] = 0.f;
  float rs = 0.f;   // wave-uniform row sum
  int cur = 0;      // row (relative to r0) being accumulated
  // BWD: per-row post-encoder parameters (lane q holds row q's), the running per-row sums, and -- when the row is
  // not prefetched as optimizer state -- the table row itself, loaded at the row's first entry
  float pe_l = 0.f, ps_l = 1.f, pe_cur = 0.f, ps_cur = 1.f, gp_acc = 0.f, gs_acc = 0.f;
  float wrow[BWD ? CH : 1][E];
  int wrow_of = -1;
  if constexpr (BWD) {
    if (lane < nrows) {
      if (bw.post_enc) pe_l = bw.post_enc[r0 + lane];
      if (bw.post_scale) ps_l = bw.post_scale[r0 + lane];
    }
    pe_cur = readlane_f(pe_l, 0);
    ps_cur = readlane_f(ps_l, 0);
  }

  // Fused optimizer variant: the parameter / exp_avg / exp_avg_sq chunks of the row being accumulated are fetched
  // when the row STARTS (16 bytes per lane and chunk each), so they are in flight behind the row's gathers and
  // the next row's fetch is issued before the current row's update arithmetic.
  constexpr bool PREFETCH = ADAM && sizeof(OT) * E == 16;
  u32x4 pq[PREFETCH ? CH : 1], mq[PREFETCH ? CH : 1], vq[PREFETCH ? CH : 1];
  auto prefetch_chunk = [&](int row, int c) {
    if constexpr (PREFETCH) {
      const int ch = lane + 64 * c;
      if (ch < nchunks) {
        const int64_t off = (r0 + row) * (int64_t)d + ch * E;
        pq[c] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(out + off));
        mq[c] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(reinterpret_cast<const OT*>(ad.m) + off));
        vq[c] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(reinterpret_cast<const OT*>(ad.v) + off));
      }
    }
  };
  auto unpack = [&](const u32x4& u, float (&f)[E]) {
    if constexpr (sizeof(OT) == 2) {
      f[0] = bf16_lo(u.x); f[1] = bf16_hi(u.x); f[2] = bf16_lo(u.y); f[3] = bf16_hi(u.y);
      if constexpr (E == 8) {
        f[4] = bf16_lo(u.z); f[5] = bf16_hi(u.z); f[6] = bf16_lo(u.w); f[7] = bf16_hi(u.w);
      }
    } else {
      f[0] = __uint_as_float(u.x); f[1] = __uint_as_float(u.y); f[2] = __uint_as_float(u.z); f[3] = __uint_as_float(u.w);
    }
  };
#pragma unroll
  for (int c = 0; c < CH; ++c) prefetch_chunk(0, c);

  auto flush_until = [&](int upto) {  // store rows cur .. upto-1 (wave-uniform loop)
    while (cur < upto) {
      OT* orow = out + (r0 + cur) * (int64_t)d;
      if constexpr (ADAM) {
        const bool more = cur + 1 < nrows;  // wave-uniform
        float dot = 0.f;
        if (ad.project) {  // wave-uniform
#pragma unroll
          for (int c = 0; c < CH; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nchunks) {
              float wv[E];
              if constexpr (PREFETCH) unpack(pq[c], wv);
              else load_row_chunk<OT, E>(orow + ch * E, wv);
#pragma unroll
              for (int e = 0; e < E; ++e) {
                const float g = sizeof(OT) == 2 ? bf16_bits_to_f32(f32_to_bf16_bits(acc[c][e])) : acc[c][e];
                dot = fmaf(g, wv[e], dot);
              }
            }
          }
          dot = wave_sum_dpp(dot);
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int ch = lane + 64 * c;
          if (ch < nchunks) {
            const int64_t off = (r0 + cur) * (int64_t)d + ch * E;
            OT* mrow = reinterpret_cast<OT*>(ad.m) + off;
            OT* vrow = reinterpret_cast<OT*>(ad.v) + off;
            float pf[E], mf[E], vf[E];
            if constexpr (PREFETCH) {
              unpack(pq[c], pf);
              unpack(mq[c], mf);
              unpack(vq[c], vf);
              if (more) prefetch_chunk(cur + 1, c);  // the next row's chunk, behind this row's arithmetic
            } else {
              load_row_chunk<OT, E>(orow + ch * E, pf);
              load_row_chunk<OT, E>(mrow, mf);
              load_row_chunk<OT, E>(vrow, vf);
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
              float g = sizeof(OT) == 2 ? bf16_bits_to_f32(f32_to_bf16_bits(acc[c][e])) : acc[c][e];
              if (ad.project) g = project_elem<sizeof(OT) == 2>(g, pf[e], dot);
              adam_update(pf[e], mf[e], vf[e], g, ad.step_size, ad.b1, ad.b2, ad.inv_sqrt_bc2, ad.eps);
              acc[c][e] = 0.f;
            }
            store_row_chunk<OT, E>(orow + ch * E, pf);
            store_row_chunk<OT, E>(mrow, mf);
            store_row_chunk<OT, E>(vrow, vf);
          }
        }
      } else {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int ch = lane + 64 * c;
          if (ch < nchunks) {
            store_row_chunk<OT, E>(orow + ch * E, acc[c]);
#pragma unroll
            for (int e = 0; e < E; ++e) acc[c][e] = 0.f;
          }
        }
      }
      if (rowsum && lane == 0) rowsum[r0 + cur] = rs;
      rs = 0.f;
      if constexpr (BWD) {
        if (lane == 0) {
          if (bw.g_post) bw.g_post[r0 + cur] = gp_acc;
          if (bw.g_scale) bw.g_scale[r0 + cur] = gs_acc;
        }
        gp_acc = gs_acc = 0.f;
      }
      ++cur;
      if constexpr (BWD) {
        if (cur < nrows) {
          pe_cur = readlane_f(pe_l, cur);
          ps_cur = readlane_f(ps_l, cur);
        }
      }
    }
  };

  for (int s0 = beg; s0 < end; s0 += 64) {
    int my_n = 0, my_row = 0, my_e = 0;
    float my_w = 0.f, my_v = 0.f, my_u = 0.f;
    const int pos = s0 + lane;
    if (pos < end) {
      const int e = perm[pos];
      my_e = e;
      uint32_t v_raw = 0, g_raw = 0;  // (every load issued before any is converted: rows_lds_row)
      if constexpr (BWD) {
        v_raw = load_scalar_word(bw.v, bw.v_dtype, e);
        my_u = bw.grad_acts[e];  // fetched with the entry: only a store is left at the end of the chunk
      }
      float w = vals[e];
      if (gate) g_raw = load_scalar_word(gate, gate_dtype, e);
      if constexpr (BWD) my_v = scalar_from_word(v_raw, bw.v, bw.v_dtype, e);
      if (gate) w = scalar_from_word(g_raw, gate, gate_dtype, e) > 0.f ? w : 0.f;
      my_w = w;
      my_n = entry_token(e, k);
#pragma unroll
      for (int q = 1; q < RW; ++q) my_row += pos >= bnd[q] ? 1 : 0;
    }
    const int cnt = min(64, end - s0);
    for (int jj = 0; jj < cnt; jj += U) {
      uint4 raw[U][CH];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int jl = min(jj + u, cnt - 1);
        const int n = readlane_i(my_n, jl);
        const char* row = reinterpret_cast<const char*>(dense) + (size_t)n * (size_t)d * sizeof(DT);
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int ch = lane + 64 * c;
          raw[u][c] = *reinterpret_cast<const uint4*>(row + (size_t)(ch < nchunks ? ch : nchunks - 1) * 16);  // (unconditional: rows_lds_row, spt_rows.h)
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (jj + u < cnt) {  // wave-uniform
          const int row_e = readlane_i(my_row, jj + u);
          const float w = readlane_f(my_w, jj + u);
          if (row_e != cur) flush_until(row_e);
          rs += w;
          if constexpr (BWD && !PREFETCH) {
            if (wrow_of != row_e) {  // wave-uniform: the row's first entry
              const OT* wsrc = (ADAM ? out : reinterpret_cast<const OT*>(bw.W)) + (r0 + row_e) * (int64_t)d;
#pragma unroll
              for (int c = 0; c < CH; ++c) {
                const int ch = lane + 64 * c;
                if (ch < nchunks) load_row_chunk<OT, E>(wsrc + ch * E, wrow[c]);
              }
              wrow_of = row_e;
            }
          }
          float part = 0.f;
#pragma unroll
          for (int c = 0; c < CH; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nchunks) {
              float x[E];
              Chunk<DT>::expand(raw[u][c], x);
#pragma unroll
              for (int e = 0; e < E; ++e) acc[c][e] = fmaf(w, x[e], acc[c][e]);
              if constexpr (BWD) {
                if constexpr (PREFETCH) {
                  float wv[E];
                  unpack(pq[c], wv);  // the parameter row before this step's update
#pragma unroll
                  for (int e = 0; e < E; ++e) part = fmaf(x[e], wv[e], part);
                } else {
#pragma unroll
                  for (int e = 0; e < E; ++e) part = fmaf(x[e], wrow[c][e], part);
                }
              }
            }
          }
          if constexpr (BWD) {
            part = wave_sum_dpp(part);  // ga of this entry, in every lane
            const float v_e = readlane_f(my_v, jj + u);
            const float posf = v_e > 0.f ? 1.f : 0.f;
            const float u_ = bw.post_scale ? part * ps_cur : part;
            // products rounded before they are added (no fma contraction): the sums are bitwise those of
            // clt_decode_bwd_acts -> clt_segment_sum
            gs_acc += mul_rounded(part, v_e + pe_cur * posf);
            gp_acc += mul_rounded(u_, posf);
            if (lane == jj + u) my_u += u_;
          }
        }
      }
    }
    if constexpr (BWD) {
      if (pos < end) bw.grad_acts[my_e] = my_u;  // every entry belongs to exactly one row: no conflicts
    }
  }
  flush_until(nrows);
}

// ---------------------------------------------------------------------------------------------------------------
// LDS-staged variant of the fused-optimizer row pass (same arithmetic, same order, bitwise the same results).
// spt_dense_kernel keeps the optimizer state of one row ahead in registers; with the decoder backward fused it needs
// 180 VGPRs (two waves per SIMD) and the parameter row at the row's FIRST entry.  Here one wave owns ONE row: it asks
// for the row's parameter, exp_avg and exp_avg_sq chunks with LDS-DMA (global_load_lds_dwordx4, non-temporal: no
// registers held while in flight), issues the segment loads and the gathers behind them and finds the row in LDS
// afterwards.  ~100 VGPRs, four waves per SIMD; measured 5.5 TB/s without and 4.5 TB/s with the backward (the
// register version: 5.1 / 3.7).  Workgroup = 2 waves, dynamic LDS = 2 * 3 * row bytes.
typedef __attribute__((address_space(3))) void* lds_void_ptr;

__device__ __forceinline__ void rows_dma16(const void* src, uint32_t lds_dst) {
  // LDS[lds_dst + 16 * lane] = *(16 bytes at src), lanes masked by EXEC do nothing.  Inline asm: hipcc does not count
  // it (its own vmcnt waits only become more conservative); our reads of the staged rows wait for vmcnt(0).
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off nt\n\t"  // streamed once: keep the gathered dense rows in L2 / Infinity Cache
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(src), "s"(lds_dst)
      : "memory");
}

// ---- long rows.  A latent that fires on most tokens owns a segment of thousands of entries; one wave walking it
// alone would outlast the rest of the launch (the reference balances such rows by processing embeddings in
// count-descending order, xformers.py:195,214-230, and sums with atomics).  Here the segment array is cut into blocks
// of ROWS_LT positions; a helper launch lets one wave per block pre-reduce the part of every LONG row (more than
// ROWS_LT entries) that falls into its block -- at most two such rows touch a block: one that began before it
// (slot 0) and one that begins inside it (slot 1) -- into an fp32 partial row plus three partial scalars, and (BWD)
// finishes the per-entry work of those entries.  The row pass then adds a long row's partials in block order instead
// of gathering: deterministic, atomic-free, and no wave handles more than ROWS_LT entries.
constexpr int ROWS_LT = 512;

struct LongRows {
  float* partial;  // [nblocks][2][d + 4] fp32 (row, then rs, gp, gs, pad); null = feature off
  int stride;      // d + 4
};

static inline size_t rows_long_workspace_bytes(int64_t n_entries, int d) {
  const int64_t nb = (n_entries + ROWS_LT - 1) / ROWS_LT;
  return (size_t)nb * 2 * (size_t)(d + 4) * sizeof(float) + 256;
}

// first row r with offsets[r + 1] > pos (the row that holds segment position pos), rows are [0, M)
__device__ __forceinline__ int row_of_position(const int32_t* __restrict__ offsets, int64_t M, int pos) {
  int lo = 0, hi = (int)M - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (offsets[mid + 1] > pos) hi = mid;
    else lo = mid + 1;
  }
  return lo;
}

template <typename T, int CH, int U, bool BWD>
__device__ __forceinline__ void rows_long_block(const int32_t* __restrict__ perm,
                                                const int32_t* __restrict__ offsets, int64_t M,
                                                const float* __restrict__ vals, const void* __restrict__ gate,
                                                int gate_dtype, const T* __restrict__ dense, int d, int k,
                                                const T* __restrict__ W, const BwdActs& bw, const LongRows& lr) {
  constexpr int E = Chunk<T>::E;
  const int lane = threadIdx.x;
  if (offsets[M + 1] == 0) return;  // clt_segments_build found no long segment: nothing to pre-reduce
  const int n_owned = offsets[M];
  const int p0 = (int)blockIdx.x * ROWS_LT;
  if (p0 >= n_owned) return;
  const int p1 = min(p0 + ROWS_LT, n_owned);
  const int nchunks = d / E;
  const int row_a = row_of_position(offsets, M, p0), row_b = row_of_position(offsets, M, p1 - 1);
  for (int which = 0; which < 2; ++which) {
    const int row = which == 0 ? row_a : row_b;
    if (which == 1 && row_b == row_a) break;
    const int rbeg = offsets[row], rend = offsets[row + 1];
    if (rend - rbeg <= ROWS_LT) continue;  // an ordinary row: the row pass gathers it
    const int slot = rbeg < p0 ? 0 : 1;
    const int beg = max(p0, rbeg), end = min(p1, rend);
    float acc[CH][E];
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int e = 0; e < E; ++e) acc[c][e] = 0.f;
    float rs = 0.f, gp_acc = 0.f, gs_acc = 0.f, pe_r = 0.f, ps_r = 1.f;
    float wrow[BWD ? CH : 1][E];
    if constexpr (BWD) {
      if (bw.post_enc) pe_r = bw.post_enc[row];
      if (bw.post_scale) ps_r = bw.post_scale[row];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nchunks) load_row_chunk<T, E>(W + (int64_t)row * d + ch * E, wrow[c]);
      }
    }
    for (int s0 = beg; s0 < end; s0 += 64) {
      int my_n = 0, my_e = 0;
      float my_w = 0.f, my_v = 0.f, my_u = 0.f;
      const int pos = s0 + lane;
      if (pos < end) {
        const int e = perm[pos];
        my_e = e;
        uint32_t v_raw = 0, g_raw = 0;  // (every load issued before any is converted: rows_lds_row)
        if constexpr (BWD) {
          v_raw = load_scalar_word(bw.v, bw.v_dtype, e);
          my_u = bw.grad_acts[e];
        }
        float w = vals[e];
        if (gate) g_raw = load_scalar_word(gate, gate_dtype, e);
        if constexpr (BWD) my_v = scalar_from_word(v_raw, bw.v, bw.v_dtype, e);
        if (gate) w = scalar_from_word(g_raw, gate, gate_dtype, e) > 0.f ? w : 0.f;
        my_w = w;
        my_n = entry_token(e, k);
      }
      const int cnt = min(64, end - s0);
      for (int jj = 0; jj < cnt; jj += U) {
        uint4 raw[U][CH];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int jl = min(jj + u, cnt - 1);
          const int n = readlane_i(my_n, jl);
          const char* grow = reinterpret_cast<const char*>(dense) + (size_t)n * (size_t)d * sizeof(T);
#pragma unroll
          for (int c = 0; c < CH; ++c) {
            const int ch = lane + 64 * c;
            raw[u][c] = *reinterpret_cast<const uint4*>(grow + (size_t)(ch < nchunks ? ch : nchunks - 1) * 16);  // (unconditional: rows_lds_row, spt_rows.h)
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (jj + u < cnt) {  // wave-uniform
            const float w = readlane_f(my_w, jj + u);
            rs += w;
            float part = 0.f;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
              const int ch = lane + 64 * c;
              if (ch < nchunks) {
                float x[E];
                Chunk<T>::expand(raw[u][c], x);
#pragma unroll
                for (int e = 0; e < E; ++e) acc[c][e] = fmaf(w, x[e], acc[c][e]);
                if constexpr (BWD) {
#pragma unroll
                  for (int e = 0; e < E; ++e) part = fmaf(x[e], wrow[c][e], part);
                }
              }
            }
            if constexpr (BWD) {
              part = wave_sum_dpp(part);
              const float v_e = readlane_f(my_v, jj + u);
              const float posf = v_e > 0.f ? 1.f : 0.f;
              const float u_ = bw.post_scale ? part * ps_r : part;
              gs_acc += mul_rounded(part, v_e + pe_r * posf);
              gp_acc += mul_rounded(u_, posf);
              if (lane == jj + u) my_u += u_;
            }
          }
        }
      }
      if constexpr (BWD) {
        if (pos < end) bw.grad_acts[my_e] = my_u;
      }
    }
    float* prow = lr.partial + ((size_t)blockIdx.x * 2 + slot) * lr.stride;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nchunks) store_elems_f32<E>(prow + ch * E, acc[c]);
    }
    if (lane == 0) {
      prow[d] = rs;
      prow[d + 1] = gp_acc;
      prow[d + 2] = gs_acc;
    }
  }
}

template <typename T, int CH, int U, bool BWD>
__global__ __launch_bounds__(64) void rows_long_kernel(const int32_t* __restrict__ perm,
                                                       const int32_t* __restrict__ offsets, int64_t M,
                                                       const float* __restrict__ vals, const void* __restrict__ gate,
                                                       int gate_dtype, const T* __restrict__ dense, int d, int k,
                                                       const T* __restrict__ W, const BwdActs bw, const LongRows lr) {
  rows_long_block<T, CH, U, BWD>(perm, offsets, M, vals, gate, gate_dtype, dense, d, k, W, bw, lr);
}

// Several row passes that share M, d, k and the dense rows (the decoder backward of every live source of ONE target:
// the same loss gradient G, one table per source) as ONE launch: blockIdx.y names the job.  One ramp-up and one tail
// instead of one per (source, target) pair, and one helper launch instead of one per pair.
constexpr int ROWS_MAX_JOBS = 16;
struct RowsJob {
  const int32_t* perm;
  const int32_t* offsets;
  const float* vals;
  void* out;  // ADAM: the parameter table
  const void* dense;  // this job's dense rows, or null: the launch's common ones (the loss gradient of a target)
  const void* gate;   // [n_entries] in the launch's gate dtype or null (encoder backward: the top-k values)
  float* rowsum;      // [M] or null (encoder backward: grad b_enc)
  AdamRows ad;
  BwdActs bw;
  LongRows lr;
};
struct RowsJobs {
  RowsJob j[ROWS_MAX_JOBS];
};

template <typename T, int CH, int U, bool BWD>
__global__ __launch_bounds__(64) void rows_long_multi_kernel(const RowsJobs jobs, int64_t M, const T* __restrict__ dense,
                                                             int gate_dtype, int d, int k) {
  const RowsJob& jb = jobs.j[blockIdx.y];
  rows_long_block<T, CH, U, BWD>(jb.perm, jb.offsets, M, jb.vals, jb.gate, gate_dtype,
                                 jb.dense ? (const T*)jb.dense : dense, d, k, (const T*)jb.out, jb.bw, jb.lr);
}

// ADAM = false: the plain-gradient form (clt_spt_dense, clt_decode_bwd_rows): `out` receives the gradient row, only
// the table row bw.W is staged (BWD) or nothing at all, and the long-segment pre-reduction is shared.
template <typename T, int CH, int U, bool BWD, bool ADAM = true>
__device__ __forceinline__ void rows_lds_row(
    const int32_t* __restrict__ perm, const int32_t* __restrict__ offsets, int64_t M, const float* __restrict__ vals,
    const void* __restrict__ gate, int gate_dtype, const T* __restrict__ dense, int d, int k, T* __restrict__ out,
    float* __restrict__ rowsum, const AdamRows& ad, const BwdActs& bw, const LongRows& lr) {
  constexpr int E = Chunk<T>::E;  // 16-byte chunks of the state rows and of the dense rows alike
  extern __shared__ __attribute__((aligned(16))) char rows_lds[];
  // (the wave number through readfirstlane: everything indexed by the row -- its segment bounds, its post-encoder
  // scalars -- is then a scalar load, ~a quarter of a vector load's latency at the head of the wave's dependent chain
  // bounds -> perm -> entry scalars -> gathered rows)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t row = (int64_t)blockIdx.x * 2 + wave;
  if (row >= M) return;
  const int nchunks = d / E;
  const int rowbytes = d * (int)sizeof(T);
  constexpr int NARR = ADAM ? 3 : (BWD ? 1 : 0);  // staged arrays per row
  char* my_lds = rows_lds + (size_t)wave * NARR * rowbytes;
  // (wave-uniform, but hipcc cannot know: readfirstlane puts it in an SGPR for the m0 moves)
  const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_void_ptr)my_lds);
  const int beg = offsets[row], end = offsets[row + 1];
  const int64_t roff = row * (int64_t)d;
#pragma unroll
  for (int c = 0; c < CH; ++c) {  // [p, m, v][chunk], chunk-linear
    const int ch = lane + 64 * c;
    if (ch < nchunks) {
      const uint32_t dst = lds0 + (uint32_t)(c * 1024);
      if constexpr (ADAM) {
        rows_dma16(out + roff + ch * E, dst);
        rows_dma16(reinterpret_cast<const T*>(ad.m) + roff + ch * E, dst + rowbytes);
        rows_dma16(reinterpret_cast<const T*>(ad.v) + roff + ch * E, dst + 2 * rowbytes);
      } else if constexpr (BWD) {
        rows_dma16(reinterpret_cast<const T*>(bw.W) + roff + ch * E, dst);
      }
    }
  }

  float acc[CH][E];
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int e = 0; e < E; ++e) acc[c][e] = 0.f;
  float rs = 0.f, pe_r = 0.f, ps_r = 1.f, gp_acc = 0.f, gs_acc = 0.f;
  if constexpr (BWD) {
    if (bw.post_enc) pe_r = bw.post_enc[row];
    if (bw.post_scale) ps_r = bw.post_scale[row];
  }
  auto lds_chunk = [&](int arr, int c, float (&f)[E]) {
    const uint4 u = *reinterpret_cast<const uint4*>(my_lds + arr * rowbytes + (lane + 64 * c) * 16);
    Chunk<T>::expand(u, f);
  };
  float wrow[BWD ? CH : 1][E];
  bool staged = false;  // wave-uniform
  const bool is_long = lr.partial != nullptr && end - beg > ROWS_LT;  // pre-reduced by rows_long_kernel
  if (is_long) {
    for (int b = beg / ROWS_LT; b <= (end - 1) / ROWS_LT; ++b) {  // block order: fixed summation order
      const float* prow = lr.partial + ((size_t)b * 2 + (beg < b * ROWS_LT ? 0 : 1)) * lr.stride;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int ch = lane + 64 * c;
        if (ch < nchunks) {
          float x[E];
          load_elems_f32<E>(prow + ch * E, x);
#pragma unroll
          for (int e = 0; e < E; ++e) acc[c][e] += x[e];
        }
      }
      rs += prow[d];
      gp_acc += prow[d + 1];
      gs_acc += prow[d + 2];
    }
  }

  for (int s0 = beg; s0 < (is_long ? beg : end); s0 += 64) {
    int my_n = 0, my_e = 0;
    float my_w = 0.f, my_v = 0.f, my_u = 0.f;
    const int pos = s0 + lane;
    if (pos < end) {
      const int e = perm[pos];
      my_e = e;
      // the entry's scalars: every load is issued before any is converted (one memory round trip, not one per array)
      uint32_t v_raw = 0, g_raw = 0;
      if constexpr (BWD) {
        v_raw = load_scalar_word(bw.v, bw.v_dtype, e);
        my_u = bw.grad_acts[e];  // fetched with the entry: only a store is left at the end of the chunk
      }
      float w = vals[e];
      if (gate) g_raw = load_scalar_word(gate, gate_dtype, e);
      if constexpr (BWD) my_v = scalar_from_word(v_raw, bw.v, bw.v_dtype, e);
      if (gate) w = scalar_from_word(g_raw, gate, gate_dtype, e) > 0.f ? w : 0.f;
      my_w = w;
      my_n = entry_token(e, k);
    }
    const int cnt = min(64, end - s0);
    for (int jj = 0; jj < cnt; jj += U) {
      uint4 raw[U][CH];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int jl = min(jj + u, cnt - 1);
        const int n = readlane_i(my_n, jl);
        const char* grow = reinterpret_cast<const char*>(dense) + (size_t)n * (size_t)d * sizeof(T);
        // Unconditional loads (lanes past the row's last chunk re-read it: same cache line, no traffic).  With the load
        // under `if (ch < nchunks)` every load sat in an EXEC-masked block of its own, and from the second batch of a
        // segment on hipcc put `s_waitcnt vmcnt(0)` in front of each (the destination registers of the previous
        // batch "may" still be pending on the path that skipped their use): eight serialised round trips per batch.
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int ch = lane + 64 * c;
          raw[u][c] = *reinterpret_cast<const uint4*>(grow + (size_t)(ch < nchunks ? ch : nchunks - 1) * 16);
        }
      }
      if constexpr (BWD) {
        // the row's first entries: the parameter row (before its update) out of LDS.  (Taking it BEFORE the gather loads
        // are requested -- the wait is free there -- was measured in round 3: 197 -> 206 us per pair; the LDS read then
        // sits on the path to the gathers.)
        if (!staged) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          staged = true;
#pragma unroll
          for (int c = 0; c < CH; ++c)
            if (lane + 64 * c < nchunks) lds_chunk(0, c, wrow[c]);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (jj + u < cnt) {  // wave-uniform
          const float w = readlane_f(my_w, jj + u);
          rs += w;
          float part = 0.f;
#pragma unroll
          for (int c = 0; c < CH; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nchunks) {
              float x[E];
              Chunk<T>::expand(raw[u][c], x);
#pragma unroll
              for (int e = 0; e < E; ++e) acc[c][e] = fmaf(w, x[e], acc[c][e]);
              if constexpr (BWD) {
#pragma unroll
                for (int e = 0; e < E; ++e) part = fmaf(x[e], wrow[c][e], part);
              }
            }
          }
          if constexpr (BWD) {
            part = wave_sum_dpp(part);
            const float v_e = readlane_f(my_v, jj + u);
            const float posf = v_e > 0.f ? 1.f : 0.f;
            const float u_ = bw.post_scale ? part * ps_r : part;
            gs_acc += mul_rounded(part, v_e + pe_r * posf);
            gp_acc += mul_rounded(u_, posf);
            if (lane == jj + u) my_u += u_;
          }
        }
      }
    }
    if constexpr (BWD) {
      if (pos < end) bw.grad_acts[my_e] = my_u;  // every entry belongs to exactly one row: no conflicts
    }
  }

  // ---- the row is complete: optimizer update out of LDS, or the gradient row itself
  if (NARR > 0 && !staged) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (also: no DMA may outlive the wave)
  if constexpr (ADAM) {
    float dot = 0.f;
    if (ad.project) {  // wave-uniform
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (lane + 64 * c < nchunks) {
          float wv[E];
          lds_chunk(0, c, wv);
#pragma unroll
          for (int e = 0; e < E; ++e) {
            const float g = sizeof(T) == 2 ? bf16_bits_to_f32(f32_to_bf16_bits(acc[c][e])) : acc[c][e];
            dot = fmaf(g, wv[e], dot);
          }
        }
      }
      dot = wave_sum_dpp(dot);
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nchunks) {
        float pf[E], mf[E], vf[E];
        lds_chunk(0, c, pf);
        lds_chunk(1, c, mf);
        lds_chunk(2, c, vf);
#pragma unroll
        for (int e = 0; e < E; ++e) {
          float g = sizeof(T) == 2 ? bf16_bits_to_f32(f32_to_bf16_bits(acc[c][e])) : acc[c][e];
          if (ad.project) g = project_elem<sizeof(T) == 2>(g, pf[e], dot);
          adam_update(pf[e], mf[e], vf[e], g, ad.step_size, ad.b1, ad.b2, ad.inv_sqrt_bc2, ad.eps);
        }
        store_row_chunk<T, E>(out + roff + ch * E, pf);
        store_row_chunk<T, E>(reinterpret_cast<T*>(ad.m) + roff + ch * E, mf);
        store_row_chunk<T, E>(reinterpret_cast<T*>(ad.v) + roff + ch * E, vf);
      }
    }
  } else {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nchunks) store_row_chunk<T, E>(out + roff + ch * E, acc[c]);  // zeros for an empty row
    }
  }
  if (lane == 0) {
    if (rowsum) rowsum[row] = rs;
    if constexpr (BWD) {
      if (bw.g_post) bw.g_post[row] = gp_acc;
      if (bw.g_scale) bw.g_scale[row] = gs_acc;
    }
  }
}

template <typename T, int CH, int U, bool BWD, bool ADAM = true>
__global__ __launch_bounds__(128, (CH <= 2 ? 4 : CH == 4 ? 3 : 1)) void rows_lds_kernel(
    const int32_t* __restrict__ perm, const int32_t* __restrict__ offsets, int64_t M, const float* __restrict__ vals,
    const void* __restrict__ gate, int gate_dtype, const T* __restrict__ dense, int d, int k, T* __restrict__ out,
    float* __restrict__ rowsum, const AdamRows ad, const BwdActs bw, const LongRows lr) {
  rows_lds_row<T, CH, U, BWD, ADAM>(perm, offsets, M, vals, gate, gate_dtype, dense, d, k, out, rowsum, ad, bw, lr);
}

template <typename T, int CH, int U, bool BWD>
__global__ __launch_bounds__(128, (CH <= 2 ? 4 : CH == 4 ? 3 : 1)) void rows_lds_multi_kernel(
    const RowsJobs jobs, int64_t M, const T* __restrict__ dense, int gate_dtype, int d, int k) {
  const RowsJob& jb = jobs.j[blockIdx.y];
  rows_lds_row<T, CH, U, BWD, true>(jb.perm, jb.offsets, M, jb.vals, jb.gate, gate_dtype,
                                    jb.dense ? (const T*)jb.dense : dense, d, k, (T*)jb.out, jb.rowsum, jb.ad, jb.bw, jb.lr);
}

// The LDS-staged kernel serves rows of up to 10 KiB (2 waves * 3 arrays * row <= 64 KiB of LDS) and 8 chunks per lane.
static inline bool rows_lds_ok(int d, size_t elem, int ch, int narr = 3) {
  static const bool off = tuning_env("CLT_ROWS_LDS", 1) == 0;  // -DCLT_TUNING builds only: A/B against the register kernel
  return !off && ch <= 8 && 2 * (size_t)narr * (size_t)d * elem <= 65536;
}

template <typename T, bool BWD, bool ADAM = true>
static int launch_rows_lds(int ch, hipStream_t st, const int32_t* perm, const int32_t* offsets, int64_t M,
                           const float* vals, const void* gate, int gate_dtype, const void* dense, int d, int k, void* out,
                           float* rowsum, AdamRows ad, BwdActs bw, int64_t n_entries, void* workspace,
                           size_t workspace_bytes) {
  const dim3 grid((unsigned)((M + 1) / 2));
  const size_t lds = 2 * (size_t)(ADAM ? 3 : (BWD ? 1 : 0)) * (size_t)d * sizeof(T);
  const void* table = ADAM ? out : bw.W;  // the rows the decoder backward dots against (the long-row helper reads them)
  LongRows lr = {nullptr, d + 4};
  const bool longs = workspace != nullptr && n_entries > ROWS_LT && workspace_bytes >= rows_long_workspace_bytes(n_entries, d);
  const dim3 lgrid((unsigned)((n_entries + ROWS_LT - 1) / ROWS_LT));
  if (longs) lr.partial = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
#define LAUNCH(CH, U)                                                                                                 \
  do {                                                                                                                \
    if (longs)                                                                                                        \
      hipLaunchKernelGGL((rows_long_kernel<T, CH, U, BWD>), lgrid, dim3(64), 0, st, perm, offsets, M, vals, gate,     \
                         gate_dtype, (const T*)dense, d, k, (const T*)table, bw, lr);                                 \
    hipLaunchKernelGGL((rows_lds_kernel<T, CH, U, BWD, ADAM>), grid, dim3(128), lds, st, perm, offsets, M, vals,      \
                       gate, gate_dtype, (const T*)dense, d, k, (T*)out, rowsum, ad, bw, lr);                         \
  } while (0)
  switch (ch) {
    case 1: LAUNCH(1, 8); break;
    case 2: LAUNCH(2, 4); break;
    case 4: LAUNCH(4, 2); break;  // 3 waves per SIMD (<= 168 VGPRs) beat 4 gathers in flight at d = 2048
    case 8: LAUNCH(8, 2); break;
    default: return CLT_ERR_UNSUPPORTED;
  }
#undef LAUNCH
  return CLT_OK;
}

// n_jobs decoder-backward + Adam row passes in one launch (plus one helper launch for their long segments); job i's
// long-row partials take the i-th slice of the workspace
template <typename T, bool BWD = true>
static int launch_rows_lds_multi(int ch, hipStream_t st, RowsJobs& jobs, int n_jobs, int64_t M, const void* dense, int d,
                                 int k, int64_t n_entries, void* workspace, size_t workspace_bytes, int gate_dtype = 0) {
  const dim3 grid((unsigned)((M + 1) / 2), (unsigned)n_jobs);
  const size_t lds = 2 * (size_t)3 * (size_t)d * sizeof(T);
  const size_t per = (rows_long_workspace_bytes(n_entries, d) + 255) & ~(size_t)255;
  const bool longs = workspace != nullptr && n_entries > ROWS_LT && workspace_bytes >= per * (size_t)n_jobs + 256;
  const dim3 lgrid((unsigned)((n_entries + ROWS_LT - 1) / ROWS_LT), (unsigned)n_jobs);
  char* base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
  for (int i = 0; i < n_jobs; ++i) {
    jobs.j[i].lr.partial = longs ? reinterpret_cast<float*>(base + per * (size_t)i) : nullptr;
    jobs.j[i].lr.stride = d + 4;
  }
#define LAUNCH(CH, U)                                                                                              \
  do {                                                                                                             \
    if (longs)                                                                                                     \
      hipLaunchKernelGGL((rows_long_multi_kernel<T, CH, U, BWD>), lgrid, dim3(64), 0, st, jobs, M, (const T*)dense,   \
                         gate_dtype, d, k);                                                                        \
    hipLaunchKernelGGL((rows_lds_multi_kernel<T, CH, U, BWD>), grid, dim3(128), lds, st, jobs, M, (const T*)dense,   \
                       gate_dtype, d, k);                                                                          \
  } while (0)
  switch (ch) {
    case 1: LAUNCH(1, 8); break;
    case 2: LAUNCH(2, 4); break;
    case 4: LAUNCH(4, 2); break;
    case 8: LAUNCH(8, 2); break;
    default: return CLT_ERR_UNSUPPORTED;
  }
#undef LAUNCH
  return CLT_OK;
}

template <typename DT, typename OT, bool ADAM = false, bool BWD = false>
static int launch_spt(int ch, hipStream_t st, const int32_t* perm, const int32_t* offsets, int64_t M,
                      const float* vals, const void* gate, int gate_dtype, const void* dense, int d, int k, void* out,
                      float* rowsum, AdamRows ad = AdamRows(), BwdActs bw = BwdActs()) {
  const int rw = pick_rw(M);
  const dim3 grid((unsigned)((M + 4 * rw - 1) / (4 * rw)));
#define LAUNCH_RW(CH, U, RW_)                                                                                         \
  hipLaunchKernelGGL((spt_dense_kernel<DT, OT, CH, U, ADAM, RW_, BWD>), grid, dim3(256), 0, st, perm, offsets, M,     \
                     vals, gate, gate_dtype, (const DT*)dense, d, k, (OT*)out, rowsum, ad, bw)
#define LAUNCH(CH, U)                    \
  do {                                   \
    if (rw == 8) LAUNCH_RW(CH, U, 8);    \
    else if (rw == 4) LAUNCH_RW(CH, U, 4); \
    else if (rw == 2) LAUNCH_RW(CH, U, 2); \
    else LAUNCH_RW(CH, U, 1);            \
  } while (0)
  switch (ch) {
    case 1: LAUNCH(1, 8); break;
    case 2:
      if (ADAM) { LAUNCH(2, 4); }  // fewer gathers in flight, more waves: the state streams need the occupancy
      else { LAUNCH(2, 8); }
      break;
    case 4: LAUNCH(4, 4); break;
    case 8: LAUNCH(8, 2); break;
    case 16: LAUNCH(16, 1); break;
    default: return CLT_ERR_UNSUPPORTED;
  }
#undef LAUNCH
#undef LAUNCH_RW
  return CLT_OK;
}


}  // namespace clt
