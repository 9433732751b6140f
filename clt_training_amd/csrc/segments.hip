// Segments (entries grouped by latent), sparse^T @ dense and per-latent sums for gfx950.
//
// clt_segments_build: the N*k (token, slot) entries grouped by latent, ascending entry id inside a latent -- a stable
// LSD radix sort of the keys (latent - lat_begin) with the entry id as payload, written for this job:
//   * digits of at most 9 bits, ceil(log2 M) bits in all (17 bits at M = 98304: two passes);
//   * one WAVE per block of 512 entries; per pass three launches: per-block digit histograms (LDS atomics), one wave
//     per digit row for the exclusive scan over blocks, and the scatter, whose stable rank inside a 64-entry round is a
//     ballot match over the digit bits (no sorting network, no cross-wave traffic);
//   * entries that belong to OTHER latent shards are dropped by the first pass (they have no segment), so under tp the
//     later passes and the offsets scan see 1/tp of the entries;
//   * the first pass reads the int64 indices directly (no key / id preparation pass).
// 7 launches for two passes; the generic library sort this replaces took 17 launches and about twice the time.
// Then a boundary-scan kernel writes offsets[M + 1] and the long-segment flag.  perm[offsets[M] ...] is unspecified.
// clt_spt_dense: one wave per output row; gathers the rows of `dense` named by the
// segment (they are re-read from L2 / Infinity Cache: dense is only N*d) and writes
// each output row exactly once -- no atomics, fixed summation order.
// HBM-bound: algorithmic bytes = M*d*sizeof(out) written + N*k*d*sizeof(dense) gathered.
#include <math.h>
#include <string.h>

#include <cstring>

#include "spt_rows.h"

namespace clt {

constexpr int RS_T = 512;      // entries per block (one wave: 8 rounds of 64)
constexpr int RS_ROUNDS = RS_T / 64;
constexpr int RS_MAXBINS = 512;

// key of entry i in the first pass: latent - lat_begin, or -1 for an entry of another shard
__device__ __forceinline__ int32_t rs_first_key(const int64_t* __restrict__ idx, int64_t i, int64_t M, int64_t lat_begin) {
  const int64_t r = idx[i] - lat_begin;
  return (r >= 0 && r < M) ? (int32_t)r : -1;
}

// hist[digit * nblocks + block] = entries of the block whose key has that digit
template <bool FIRST>
__global__ __launch_bounds__(64) void rs_hist_kernel(const int64_t* __restrict__ idx, const int32_t* __restrict__ keys,
                                                     const int32_t* __restrict__ n_dev, int64_t n, int64_t M,
                                                     int64_t lat_begin, int shift, int nbins, int nblocks,
                                                     int32_t* __restrict__ hist, int32_t* __restrict__ offsets) {
  __shared__ int h[RS_MAXBINS];
  const int lane = threadIdx.x;
  if (FIRST && blockIdx.x == 0 && lane == 0) offsets[M + 1] = 0;  // "some segment is long" flag (seg_offsets_kernel)
  for (int b = lane; b < nbins; b += 64) h[b] = 0;
  __syncthreads();
  const int64_t cnt = FIRST ? n : (int64_t)*n_dev;
  const int64_t base = (int64_t)blockIdx.x * RS_T;
  if (base < cnt) {
#pragma unroll 4
    for (int r = 0; r < RS_ROUNDS; ++r) {
      const int64_t i = base + r * 64 + lane;
      if (i < cnt) {
        const int32_t key = FIRST ? rs_first_key(idx, i, M, lat_begin) : keys[i];
        if (key >= 0) atomicAdd(&h[(key >> shift) & (nbins - 1)], 1);
      }
    }
  }
  __syncthreads();
  for (int b = lane; b < nbins; b += 64) hist[(int64_t)b * nblocks + blockIdx.x] = h[b];
}

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

// one wave per digit: exclusive scan of its row over the blocks (in place) and the row's total
__global__ __launch_bounds__(64) void rs_scan_kernel(int32_t* __restrict__ hist, int nblocks, int32_t* __restrict__ totals) {
  const int lane = threadIdx.x;
  int32_t* row = hist + (int64_t)blockIdx.x * nblocks;  // (nblocks is a multiple of 4: rows stay 16-byte aligned)
  int carry = 0;
  for (int j0 = 0; j0 < nblocks; j0 += 256) {  // four consecutive blocks per lane and iteration
    const int j = j0 + 4 * lane;
    int4 v = make_int4(0, 0, 0, 0);
    if (j < nblocks) v = *reinterpret_cast<const int4*>(row + j);
    const int s4 = v.x + v.y + v.z + v.w;
    const int inc = wave_incl_scan(s4, lane);
    const int e0 = carry + inc - s4;
    if (j < nblocks) *reinterpret_cast<int4*>(row + j) = make_int4(e0, e0 + v.x, e0 + v.x + v.y, e0 + v.x + v.y + v.z);
    carry += __shfl(inc, 63, 64);
  }
  if (lane == 0) totals[blockIdx.x] = carry;
}

// stable scatter: position = (entries with a smaller digit) + (entries with this digit in earlier blocks) + (earlier
// entries with this digit in this block)
template <bool FIRST>
__global__ __launch_bounds__(64) void rs_scatter_kernel(const int64_t* __restrict__ idx, const int32_t* __restrict__ keys,
                                                        const int32_t* __restrict__ ids, int32_t* __restrict__ n_dev,
                                                        int64_t n, int64_t M, int64_t lat_begin, int shift, int nbits,
                                                        int nblocks, const int32_t* __restrict__ hist,
                                                        const int32_t* __restrict__ totals, int32_t* __restrict__ keys_out,
                                                        int32_t* __restrict__ ids_out) {
  __shared__ int base[RS_MAXBINS];
  const int lane = threadIdx.x;
  const int nbins = 1 << nbits;
  // exclusive scan of the digit totals: lane l owns bins [l * per, l * per + per)
  const int per = (nbins + 63) / 64;
  int mine = 0;
  for (int q = 0; q < per; ++q) {
    const int b = lane * per + q;
    if (b < nbins) mine += totals[b];
  }
  const int inc = wave_incl_scan(mine, lane);
  int run = inc - mine;
  for (int q = 0; q < per; ++q) {
    const int b = lane * per + q;
    if (b < nbins) {
      base[b] = run + hist[(int64_t)b * nblocks + blockIdx.x];
      run += totals[b];
    }
  }
  if (FIRST && blockIdx.x == 0 && lane == 63) *n_dev = inc;  // number of owned entries: the later passes' length
  const int64_t cnt = FIRST ? n : (int64_t)__shfl(inc, 63, 64);
  __syncthreads();
  const int64_t b0 = (int64_t)blockIdx.x * RS_T;
  if (b0 >= cnt) return;
  int32_t key[RS_ROUNDS], id[RS_ROUNDS];
#pragma unroll
  for (int r = 0; r < RS_ROUNDS; ++r) {
    const int64_t i = b0 + r * 64 + lane;
    key[r] = -1;
    id[r] = 0;
    if (i < cnt) {
      key[r] = FIRST ? rs_first_key(idx, i, M, lat_begin) : keys[i];
      id[r] = FIRST ? (int32_t)i : ids[i];
    }
  }
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll  // (key[] / id[] live in registers: the round index must be a constant)
  for (int r = 0; r < RS_ROUNDS; ++r) {
    const bool valid = key[r] >= 0;
    const int digit = valid ? (key[r] >> shift) & (nbins - 1) : 0;
    unsigned long long peers = __ballot(valid);
    for (int b = 0; b < nbits; ++b) {
      const bool bit = (digit >> b) & 1;
      const unsigned long long bb = __ballot(valid && bit);
      peers &= bit ? bb : ~bb;
    }
    const int rank = __popcll(peers & lt);
    int pos = 0;
    if (valid) pos = base[digit] + rank;
    __syncthreads();  // every lane has read its base before a leader moves it
    if (valid && rank == 0) base[digit] += __popcll(peers);
    __syncthreads();
    if (valid) {
      keys_out[pos] = key[r];
      ids_out[pos] = id[r];
    }
  }
}

// offsets[r] = first position p with sorted_keys[p] >= r;  offsets[M] = number of owned entries (*n_dev);
// offsets[M + 1] = 1 if some segment holds more than ROWS_LT entries (the same key ROWS_LT positions apart).
__global__ void seg_offsets_kernel(const int32_t* __restrict__ sorted_keys, int32_t* __restrict__ offsets,
                                   const int32_t* __restrict__ n_dev, int64_t M) {
  const int64_t n = *n_dev;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  if (i + ROWS_LT < n && sorted_keys[i] == sorted_keys[i + ROWS_LT]) offsets[M + 1] = 1;
  int64_t lo = (i == 0) ? 0 : (int64_t)sorted_keys[i - 1] + 1;
  int64_t hi = (i == n) ? M : (int64_t)sorted_keys[i];
  if (hi > M) hi = M;  // defensive: out-of-range keys never index offsets
  for (int64_t r = lo; r <= hi; ++r) offsets[r] = (int32_t)i;
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

struct RsPlan {
  int passes, nbits[4], shift[4], nblocks;
  size_t a, hist_bytes, bytes;  // a: bytes of one int32 array of n entries
};

static RsPlan rs_plan(int64_t n, int64_t M) {
  RsPlan p = {};
  int bits = 1;
  while ((1ll << bits) < M) ++bits;
  p.passes = (bits + 8) / 9;
  int done = 0;
  for (int i = 0; i < p.passes; ++i) {
    const int left = bits - done, w = (left + (p.passes - i) - 1) / (p.passes - i);
    p.nbits[i] = w;
    p.shift[i] = done;
    done += w;
  }
  p.nblocks = (int)align_up((size_t)((n + RS_T - 1) / RS_T), 4);  // (rs_scan_kernel reads rows in 16-byte pieces)
  p.a = align_up((size_t)n * 4, 256);
  p.hist_bytes = align_up((size_t)RS_MAXBINS * p.nblocks * 4, 256);
  p.bytes = 4 * p.a + p.hist_bytes + align_up(RS_MAXBINS * 4, 256) + 256;
  return p;
}

}  // namespace clt

using namespace clt;

extern "C" size_t clt_segments_workspace_bytes(int64_t n_entries, int64_t M) {
  if (n_entries <= 0 || M <= 0) return 256;
  return rs_plan(n_entries, M).bytes;
}

extern "C" int clt_segments_build(const int64_t* idx, int64_t n, int64_t M, int64_t lat_begin, int32_t* perm,
                                  int32_t* offsets,
                                  void* workspace, size_t workspace_bytes, clt_stream_t stream) {
  CLT_CHECK_ARG(idx && perm && offsets && workspace, "clt_segments_build: null pointer");
  CLT_CHECK_ARG(n > 0 && n < (1ll << 31) && M > 0 && M < (1ll << 31), "clt_segments_build: bad sizes n=%lld M=%lld",
                (long long)n, (long long)M);
  const RsPlan pl = rs_plan(n, M);
  if (workspace_bytes < pl.bytes) {
    set_error("clt_segments_build: workspace too small (%zu < %zu)", workspace_bytes, pl.bytes);
    return CLT_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  char* ws = reinterpret_cast<char*>(workspace);
  int32_t* kbuf[2] = {reinterpret_cast<int32_t*>(ws), reinterpret_cast<int32_t*>(ws + pl.a)};
  int32_t* ibuf[2] = {reinterpret_cast<int32_t*>(ws + 2 * pl.a), reinterpret_cast<int32_t*>(ws + 3 * pl.a)};
  int32_t* hist = reinterpret_cast<int32_t*>(ws + 4 * pl.a);
  int32_t* totals = reinterpret_cast<int32_t*>(ws + 4 * pl.a + pl.hist_bytes);
  int32_t* n_dev = totals + RS_MAXBINS;  // (inside the trailing 256 bytes)
  const int32_t* kin = nullptr;
  const int32_t* iin = nullptr;
  for (int p = 0; p < pl.passes; ++p) {
    const int nbins = 1 << pl.nbits[p];
    int32_t* kout = kbuf[p & 1];
    int32_t* iout = p + 1 == pl.passes ? perm : ibuf[p & 1];
    if (p == 0) {
      hipLaunchKernelGGL((rs_hist_kernel<true>), dim3(pl.nblocks), dim3(64), 0, st, idx, kin, n_dev, n, M, lat_begin,
                         pl.shift[p], nbins, pl.nblocks, hist, offsets);
      hipLaunchKernelGGL(rs_scan_kernel, dim3(nbins), dim3(64), 0, st, hist, pl.nblocks, totals);
      hipLaunchKernelGGL((rs_scatter_kernel<true>), dim3(pl.nblocks), dim3(64), 0, st, idx, kin, iin, n_dev, n, M,
                         lat_begin, pl.shift[p], pl.nbits[p], pl.nblocks, hist, totals, kout, iout);
    } else {
      hipLaunchKernelGGL((rs_hist_kernel<false>), dim3(pl.nblocks), dim3(64), 0, st, idx, kin, n_dev, n, M, lat_begin,
                         pl.shift[p], nbins, pl.nblocks, hist, offsets);
      hipLaunchKernelGGL(rs_scan_kernel, dim3(nbins), dim3(64), 0, st, hist, pl.nblocks, totals);
      hipLaunchKernelGGL((rs_scatter_kernel<false>), dim3(pl.nblocks), dim3(64), 0, st, idx, kin, iin, n_dev, n, M,
                         lat_begin, pl.shift[p], pl.nbits[p], pl.nblocks, hist, totals, kout, iout);
    }
    kin = kout;
    iin = iout;
  }
  const unsigned oblocks = (unsigned)((n + 1 + 255) / 256);
  hipLaunchKernelGGL(seg_offsets_kernel, dim3(oblocks), dim3(256), 0, st, kin, offsets, (const int32_t*)n_dev, M);
  CLT_CHECK_LAUNCH("clt_segments_build");
  return CLT_OK;
}

extern "C" int clt_spt_dense(const int32_t* perm, const int32_t* offsets, int64_t M, const float* vals,
                             const void* vals_gate, int gate_dtype, const void* dense, int dense_dtype, int d, int k,
                             void* out, int out_dtype, float* rowsum, int64_t n_entries, void* workspace,
                             size_t workspace_bytes, clt_stream_t stream) {
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
  const bool lds_ok = dense_dtype == out_dtype && rows_lds_ok(d, dtype_size(out_dtype), ch, 0);
  if (lds_ok && out_dtype == CLT_BF16)
    rc = launch_rows_lds<bf16_t, false, false>(ch, st, perm, offsets, M, vals, vals_gate, gate_dtype, dense, d, k, out, rowsum,
                                               AdamRows(), BwdActs(), n_entries, workspace, workspace_bytes);
  else if (lds_ok)
    rc = launch_rows_lds<float, false, false>(ch, st, perm, offsets, M, vals, vals_gate, gate_dtype, dense, d, k, out, rowsum,
                                              AdamRows(), BwdActs(), n_entries, workspace, workspace_bytes);
  else if (dense_dtype == CLT_BF16 && out_dtype == CLT_BF16)
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

extern "C" size_t clt_rows_workspace_bytes(int64_t n_entries, int d) {
  return n_entries > 0 && d > 0 ? rows_long_workspace_bytes(n_entries, d) : 256;
}

extern "C" int clt_spt_dense_adam(const int32_t* perm, const int32_t* offsets, int64_t M, const float* vals,
                                  const void* vals_gate, int gate_dtype, const void* dense, int dense_dtype, int d, int k,
                                  void* p, void* m, void* v, int p_dtype, float* rowsum, float lr, float b1, float b2,
                                  float eps, int step, int64_t n_entries, void* workspace, size_t workspace_bytes,
                                  clt_stream_t stream) {
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
  const AdamRows ad = make_adam_rows(m, v, lr, b1, b2, eps, step);
  hipStream_t st = (hipStream_t)stream;
  int rc;
  const bool lds_ok = dense_dtype == p_dtype && rows_lds_ok(d, dtype_size(p_dtype), ch);
  if (lds_ok && p_dtype == CLT_BF16)
    rc = launch_rows_lds<bf16_t, false>(ch, st, perm, offsets, M, vals, vals_gate, gate_dtype, dense, d, k, p, rowsum, ad, BwdActs(), n_entries, workspace, workspace_bytes);
  else if (lds_ok)
    rc = launch_rows_lds<float, false>(ch, st, perm, offsets, M, vals, vals_gate, gate_dtype, dense, d, k, p, rowsum, ad, BwdActs(), n_entries, workspace, workspace_bytes);
  else if (dense_dtype == CLT_BF16 && p_dtype == CLT_BF16)
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

extern "C" int clt_spt_dense_adam_multi(const clt_spt_job* h_jobs, int n_jobs, int64_t M, int gate_dtype,
                                        int dense_dtype, int d, int k, int p_dtype, float lr, float b1, float b2,
                                        float eps, int step, int64_t n_entries, void* workspace,
                                        size_t workspace_bytes, clt_stream_t stream) {
  CLT_CHECK_ARG(h_jobs && n_jobs >= 1 && n_jobs <= CLT_ROWS_MAX_JOBS, "clt_spt_dense_adam_multi: 1..%d jobs",
                CLT_ROWS_MAX_JOBS);
  CLT_CHECK_ARG(M > 0 && M < (1ll << 31) && d > 0 && k > 0 && step >= 1 && valid_dtype(dense_dtype) &&
                    valid_dtype(p_dtype), "clt_spt_dense_adam_multi: bad sizes / dtypes");
  const int E = dense_dtype == CLT_BF16 ? 8 : 4;
  int chunks = d / E, ch = (chunks + 63) / 64;
  ch = ch <= 1 ? 1 : ch <= 2 ? 2 : ch <= 4 ? 4 : ch <= 8 ? 8 : -1;
  const bool lds_ok = d % E == 0 && ch > 0 && dense_dtype == p_dtype && rows_lds_ok(d, dtype_size(p_dtype), ch);
  RowsJobs jobs = {};
  for (int i = 0; i < n_jobs; ++i) {
    const clt_spt_job& h = h_jobs[i];
    CLT_CHECK_ARG(h.perm && h.offsets && h.vals && h.dense && h.p && h.m && h.v && aligned16(h.dense) && aligned16(h.p) &&
                      aligned16(h.m) && aligned16(h.v) && (!h.vals_gate || valid_dtype(gate_dtype)),
                  "clt_spt_dense_adam_multi: job %d: null or misaligned pointer", i);
    if (!lds_ok) {  // shapes the LDS-staged kernel does not take: one call per job
      const int rc = clt_spt_dense_adam(h.perm, h.offsets, M, h.vals, h.vals_gate, gate_dtype, h.dense, dense_dtype, d, k,
                                        h.p, h.m, h.v, p_dtype, h.rowsum, lr, b1, b2, eps, step, n_entries, workspace,
                                        workspace_bytes, stream);
      if (rc != CLT_OK) return rc;
      continue;
    }
    RowsJob& jb = jobs.j[i];
    jb.perm = h.perm;
    jb.offsets = h.offsets;
    jb.vals = h.vals;
    jb.out = h.p;
    jb.dense = h.dense;
    jb.gate = h.vals_gate;
    jb.rowsum = h.rowsum;
    jb.ad = make_adam_rows(h.m, h.v, lr, b1, b2, eps, step);
  }
  if (!lds_ok) return CLT_OK;
  hipStream_t st = (hipStream_t)stream;
  const int rc = p_dtype == CLT_BF16
                     ? launch_rows_lds_multi<bf16_t, false>(ch, st, jobs, n_jobs, M, nullptr, d, k, n_entries, workspace,
                                                            workspace_bytes, gate_dtype)
                     : launch_rows_lds_multi<float, false>(ch, st, jobs, n_jobs, M, nullptr, d, k, n_entries, workspace,
                                                           workspace_bytes, gate_dtype);
  if (rc != CLT_OK) {
    set_error("clt_spt_dense_adam_multi: unsupported width");
    return rc;
  }
  CLT_CHECK_LAUNCH("clt_spt_dense_adam_multi");
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
