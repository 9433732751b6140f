// Segments (entries grouped by latent), sparse^T @ dense and per-latent sums for gfx950.
//
// clt_segments_build: stable LSD radix sort of the N*k int32 latent keys with the
// entry id as payload (rocPRIM device radix sort: plumbing, bit range limited to
// ceil(log2 M)), then a boundary-scan kernel that writes offsets[M+1].
// clt_spt_dense: one wave per output row; gathers the rows of `dense` named by the
// segment (they are re-read from L2 / Infinity Cache: dense is only N*d) and writes
// each output row exactly once -- no atomics, fixed summation order.
// HBM-bound: algorithmic bytes = M*d*sizeof(out) written + N*k*d*sizeof(dense) gathered.
#include <math.h>
#include <string.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"

namespace clt {

// key = latent - lat_begin, or the sentinel M for entries of other shards (sorted behind every real segment)
__global__ void seg_prepare_kernel(const int64_t* __restrict__ idx, int32_t* __restrict__ keys,
                                   int32_t* __restrict__ ids, int64_t n, int64_t M, int64_t lat_begin) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const int64_t r = idx[i] - lat_begin;
    keys[i] = (r >= 0 && r < M) ? (int32_t)r : (int32_t)M;
    ids[i] = (int32_t)i;
  }
}

// offsets[r] = first position p with sorted_keys[p] >= r;  offsets[M] = n.
__global__ void seg_offsets_kernel(const int32_t* __restrict__ sorted_keys, int32_t* __restrict__ offsets,
                                   int64_t n, int64_t M) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  int64_t lo = (i == 0) ? 0 : (int64_t)sorted_keys[i - 1] + 1;
  int64_t hi = (i == n) ? M : (int64_t)sorted_keys[i];
  if (hi > M) hi = M;  // defensive: out-of-range keys never index offsets
  for (int64_t r = lo; r <= hi; ++r) offsets[r] = (int32_t)i;
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
};

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

template <typename DT, typename OT, int CH, int U, bool ADAM, int RW>
__global__ __launch_bounds__(256) void spt_dense_kernel(const int32_t* __restrict__ perm,
                                                        const int32_t* __restrict__ offsets, int64_t M,
                                                        const float* __restrict__ vals,
                                                        const void* __restrict__ gate, int gate_dtype,
                                                        const DT* __restrict__ dense, int d, int k,
                                                        OT* __restrict__ out, float* __restrict__ rowsum,
                                                        const AdamRows ad) {
  constexpr int E = Chunk<DT>::E;
  const int lane = threadIdx.x & 63;
  const int64_t r0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RW;
  if (r0 >= M) return;
  const int nrows = (int)min((int64_t)RW, M - r0);
  const int nchunks = d / E;
  const int my_off = lane <= nrows ? offsets[r0 + lane] : 0;  // lanes 0..nrows hold the RW+1 bounds
  const int beg = readlane_i(my_off, 0), end = readlane_i(my_off, nrows);

  float acc[CH][E];
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int e = 0; e < E; ++e) acc[c][e] = 0.f;
  float rs = 0.f;   // wave-uniform row sum
  int cur = 0;      // row (relative to r0) being accumulated

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
              const float g = sizeof(OT) == 2 ? bf16_bits_to_f32(f32_to_bf16_bits(acc[c][e])) : acc[c][e];
              mf[e] = mf[e] + (g - mf[e]) * (1.f - ad.b1);
              vf[e] = ad.b2 * vf[e] + (1.f - ad.b2) * g * g;
              const float denom = sqrtf(vf[e]) * ad.inv_sqrt_bc2 + ad.eps;
              pf[e] = pf[e] - ad.step_size * (mf[e] / denom);
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
      ++cur;
    }
  };

  for (int s0 = beg; s0 < end; s0 += 64) {
    int my_n = 0, my_row = 0;
    float my_w = 0.f;
    const int pos = s0 + lane;
    if (pos < end) {
      const int e = perm[pos];
      float w = vals[e];
      if (gate) w = load_scalar(gate, gate_dtype, e) > 0.f ? w : 0.f;
      my_w = w;
      my_n = e / k;
#pragma unroll
      for (int q = 1; q < RW; ++q) my_row += (q < nrows && pos >= readlane_i(my_off, q)) ? 1 : 0;
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
          if (ch < nchunks) raw[u][c] = *reinterpret_cast<const uint4*>(row + (size_t)ch * 16);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (jj + u < cnt) {  // wave-uniform
          const int row_e = readlane_i(my_row, jj + u);
          const float w = readlane_f(my_w, jj + u);
          if (row_e != cur) flush_until(row_e);
          rs += w;
#pragma unroll
          for (int c = 0; c < CH; ++c) {
            const int ch = lane + 64 * c;
            if (ch < nchunks) {
              float x[E];
              Chunk<DT>::expand(raw[u][c], x);
#pragma unroll
              for (int e = 0; e < E; ++e) acc[c][e] = fmaf(w, x[e], acc[c][e]);
            }
          }
        }
      }
    }
  }
  flush_until(nrows);
}

__global__ void segment_sum_kernel(const int32_t* __restrict__ perm, const int32_t* __restrict__ offsets, int64_t M,
                                   const float* __restrict__ a, float* __restrict__ out_a,
                                   const float* __restrict__ b, float* __restrict__ out_b) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= M) return;
  const int beg = offsets[r], end = offsets[r + 1];
  float sa = 0.f, sb = 0.f;
  for (int s = beg; s < end; ++s) {
    const int e = perm[s];
    sa += a[e];
    if (b) sb += b[e];
  }
  out_a[r] = sa;
  if (b) out_b[r] = sb;
}

static int key_bits(int64_t M) {
  int bits = 1;
  while ((1ll << bits) < M + 1) ++bits;  // + the sentinel key M
  return bits;
}

static size_t sort_temp_bytes(int64_t n, int64_t M) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs<rocprim::default_config, const int32_t*, int32_t*, const int32_t*, int32_t*>(
      nullptr, bytes, nullptr, nullptr, nullptr, nullptr, (size_t)n, 0, key_bits(M), 0, false);
  return bytes;
}

}  // namespace clt

using namespace clt;

extern "C" size_t clt_segments_workspace_bytes(int64_t n_entries, int64_t M) {
  if (n_entries <= 0 || M <= 0) return 256;
  size_t a = align_up((size_t)n_entries * 4, 256);
  return 3 * a + align_up(sort_temp_bytes(n_entries, M), 256) + 256;
}

extern "C" int clt_segments_build(const int64_t* idx, int64_t n, int64_t M, int64_t lat_begin, int32_t* perm,
                                  int32_t* offsets,
                                  void* workspace, size_t workspace_bytes, clt_stream_t stream) {
  CLT_CHECK_ARG(idx && perm && offsets && workspace, "clt_segments_build: null pointer");
  CLT_CHECK_ARG(n > 0 && n < (1ll << 31) && M > 0 && M < (1ll << 31), "clt_segments_build: bad sizes n=%lld M=%lld",
                (long long)n, (long long)M);
  if (workspace_bytes < clt_segments_workspace_bytes(n, M)) {
    set_error("clt_segments_build: workspace too small (%zu < %zu)", workspace_bytes,
              clt_segments_workspace_bytes(n, M));
    return CLT_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const size_t a = align_up((size_t)n * 4, 256);
  char* ws = reinterpret_cast<char*>(workspace);
  int32_t* keys_in = reinterpret_cast<int32_t*>(ws);
  int32_t* ids_in = reinterpret_cast<int32_t*>(ws + a);
  int32_t* keys_out = reinterpret_cast<int32_t*>(ws + 2 * a);
  void* temp = ws + 3 * a;
  size_t temp_bytes = sort_temp_bytes(n, M);
  const unsigned blocks = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(seg_prepare_kernel, dim3(blocks), dim3(256), 0, st, idx, keys_in, ids_in, n, M, lat_begin);
  hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, (const int32_t*)keys_in, keys_out,
                                           (const int32_t*)ids_in, perm, (size_t)n, 0, key_bits(M), st, false);
  if (e != hipSuccess) {
    set_error("clt_segments_build: radix sort failed: %s", hipGetErrorString(e));
    return CLT_ERR_LAUNCH;
  }
  const unsigned oblocks = (unsigned)((n + 1 + 255) / 256);
  hipLaunchKernelGGL(seg_offsets_kernel, dim3(oblocks), dim3(256), 0, st, (const int32_t*)keys_out, offsets, n, M);
  CLT_CHECK_LAUNCH("clt_segments_build");
  return CLT_OK;
}

template <typename DT, typename OT, bool ADAM = false>
static int launch_spt(int ch, hipStream_t st, const int32_t* perm, const int32_t* offsets, int64_t M,
                      const float* vals, const void* gate, int gate_dtype, const void* dense, int d, int k, void* out,
                      float* rowsum, AdamRows ad = AdamRows()) {
  const int rw = pick_rw(M);
  const dim3 grid((unsigned)((M + 4 * rw - 1) / (4 * rw)));
#define LAUNCH_RW(CH, U, RW_)                                                                                         \
  hipLaunchKernelGGL((spt_dense_kernel<DT, OT, CH, U, ADAM, RW_>), grid, dim3(256), 0, st, perm, offsets, M, vals,    \
                     gate, gate_dtype, (const DT*)dense, d, k, (OT*)out, rowsum, ad)
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

extern "C" int clt_spt_dense(const int32_t* perm, const int32_t* offsets, int64_t M, const float* vals,
                             const void* vals_gate, int gate_dtype, const void* dense, int dense_dtype, int d, int k,
                             void* out, int out_dtype, float* rowsum, clt_stream_t stream) {
  CLT_CHECK_ARG(perm && offsets && vals && dense && out, "clt_spt_dense: null pointer");
  CLT_CHECK_ARG(M > 0 && M < (1ll << 31) && d > 0 && k > 0, "clt_spt_dense: bad sizes");
  CLT_CHECK_ARG(valid_dtype(dense_dtype) && valid_dtype(out_dtype) && (!vals_gate || valid_dtype(gate_dtype)),
                "clt_spt_dense: bad dtype");
  CLT_CHECK_ARG(aligned16(dense) && aligned16(out), "clt_spt_dense: pointers must be 16-byte aligned");
  CLT_UNSUPPORTED(d % 8 == 0 || (d % 4 == 0 && dense_dtype == CLT_F32 && out_dtype == CLT_F32),
                  "clt_spt_dense: d=%d must be a multiple of 8 (or 4 for all-fp32)", d);
  const int E = dense_dtype == CLT_BF16 ? 8 : 4;
  int chunks = d / E, ch = (chunks + 63) / 64;
  ch = ch <= 1 ? 1 : ch <= 2 ? 2 : ch <= 4 ? 4 : ch <= 8 ? 8 : ch <= 16 ? 16 : -1;
  CLT_UNSUPPORTED(ch > 0, "clt_spt_dense: d=%d too wide", d);
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (dense_dtype == CLT_BF16 && out_dtype == CLT_BF16)
    rc = launch_spt<bf16_t, bf16_t>(ch, st, perm, offsets, M, vals, vals_gate, gate_dtype, dense, d, k, out, rowsum);
  else if (dense_dtype == CLT_BF16 && out_dtype == CLT_F32)
    rc = launch_spt<bf16_t, float>(ch, st, perm, offsets, M, vals, vals_gate, gate_dtype, dense, d, k, out, rowsum);
  else if (dense_dtype == CLT_F32 && out_dtype == CLT_F32)
    rc = launch_spt<float, float>(ch, st, perm, offsets, M, vals, vals_gate, gate_dtype, dense, d, k, out, rowsum);
  else
    rc = launch_spt<float, bf16_t>(ch, st, perm, offsets, M, vals, vals_gate, gate_dtype, dense, d, k, out, rowsum);
  if (rc != CLT_OK) {
    set_error("clt_spt_dense: unsupported width");
    return rc;
  }
  CLT_CHECK_LAUNCH("clt_spt_dense");
  return CLT_OK;
}

extern "C" int clt_spt_dense_adam(const int32_t* perm, const int32_t* offsets, int64_t M, const float* vals,
                                  const void* vals_gate, int gate_dtype, const void* dense, int dense_dtype, int d, int k,
                                  void* p, void* m, void* v, int p_dtype, float* rowsum, float lr, float b1, float b2,
                                  float eps, int step, clt_stream_t stream) {
  CLT_CHECK_ARG(perm && offsets && vals && dense && p && m && v, "clt_spt_dense_adam: null pointer");
  CLT_CHECK_ARG(M > 0 && M < (1ll << 31) && d > 0 && k > 0 && step >= 1, "clt_spt_dense_adam: bad sizes");
  CLT_CHECK_ARG(valid_dtype(dense_dtype) && valid_dtype(p_dtype) && (!vals_gate || valid_dtype(gate_dtype)),
                "clt_spt_dense_adam: bad dtype");
  CLT_CHECK_ARG(aligned16(dense) && aligned16(p) && aligned16(m) && aligned16(v),
                "clt_spt_dense_adam: pointers must be 16-byte aligned");
  CLT_UNSUPPORTED(d % 8 == 0 || (d % 4 == 0 && dense_dtype == CLT_F32 && p_dtype == CLT_F32),
                  "clt_spt_dense_adam: d=%d must be a multiple of 8 (or 4 for all-fp32)", d);
  const int E = dense_dtype == CLT_BF16 ? 8 : 4;
  int chunks = d / E, ch = (chunks + 63) / 64;
  ch = ch <= 1 ? 1 : ch <= 2 ? 2 : ch <= 4 ? 4 : ch <= 8 ? 8 : ch <= 16 ? 16 : -1;
  CLT_UNSUPPORTED(ch > 0, "clt_spt_dense_adam: d=%d too wide", d);
  const double bc1 = 1.0 - pow((double)b1, (double)step);  // the same host arithmetic as clt_adam_step
  const double bc2 = 1.0 - pow((double)b2, (double)step);
  AdamRows ad;
  ad.m = m;
  ad.v = v;
  ad.step_size = lr * 1.0f * (float)(1.0 / bc1);
  ad.b1 = b1;
  ad.b2 = b2;
  ad.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  ad.eps = eps;
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (dense_dtype == CLT_BF16 && p_dtype == CLT_BF16)
    rc = launch_spt<bf16_t, bf16_t, true>(ch, st, perm, offsets, M, vals, vals_gate, gate_dtype, dense, d, k, p, rowsum, ad);
  else if (dense_dtype == CLT_BF16 && p_dtype == CLT_F32)
    rc = launch_spt<bf16_t, float, true>(ch, st, perm, offsets, M, vals, vals_gate, gate_dtype, dense, d, k, p, rowsum, ad);
  else if (dense_dtype == CLT_F32 && p_dtype == CLT_F32)
    rc = launch_spt<float, float, true>(ch, st, perm, offsets, M, vals, vals_gate, gate_dtype, dense, d, k, p, rowsum, ad);
  else
    rc = launch_spt<float, bf16_t, true>(ch, st, perm, offsets, M, vals, vals_gate, gate_dtype, dense, d, k, p, rowsum, ad);
  if (rc != CLT_OK) {
    set_error("clt_spt_dense_adam: unsupported width");
    return rc;
  }
  CLT_CHECK_LAUNCH("clt_spt_dense_adam");
  return CLT_OK;
}

extern "C" int clt_segment_sum(const int32_t* perm, const int32_t* offsets, int64_t M, const float* a, float* out_a,
                               const float* b, float* out_b, clt_stream_t stream) {
  CLT_CHECK_ARG(perm && offsets && a && out_a && (!b || out_b), "clt_segment_sum: null pointer");
  CLT_CHECK_ARG(M > 0 && M < (1ll << 31), "clt_segment_sum: bad M");
  hipLaunchKernelGGL(segment_sum_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, perm,
                     offsets, M, a, out_a, b, out_b);
  CLT_CHECK_LAUNCH("clt_segment_sum");
  return CLT_OK;
}
