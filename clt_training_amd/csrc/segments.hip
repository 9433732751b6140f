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

#include "spt_rows.h"

namespace clt {

// key = latent - lat_begin, or the sentinel M for entries of other shards (sorted behind every real segment)
__global__ void seg_prepare_kernel(const int64_t* __restrict__ idx, int32_t* __restrict__ keys,
                                   int32_t* __restrict__ ids, int64_t n, int64_t M, int64_t lat_begin,
                                   int32_t* __restrict__ offsets) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) offsets[M + 1] = 0;  // "some segment is long" flag, raised by seg_offsets_kernel
  if (i < n) {
    const int64_t r = idx[i] - lat_begin;
    keys[i] = (r >= 0 && r < M) ? (int32_t)r : (int32_t)M;
    ids[i] = (int32_t)i;
  }
}

// offsets[r] = first position p with sorted_keys[p] >= r;  offsets[M] = number of owned entries;
// offsets[M + 1] = 1 if some segment holds more than ROWS_LT entries (the same key ROWS_LT positions apart).
__global__ void seg_offsets_kernel(const int32_t* __restrict__ sorted_keys, int32_t* __restrict__ offsets,
                                   int64_t n, int64_t M) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  if (i + ROWS_LT < n && sorted_keys[i] < M && sorted_keys[i] == sorted_keys[i + ROWS_LT]) offsets[M + 1] = 1;
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
  hipLaunchKernelGGL(seg_prepare_kernel, dim3(blocks), dim3(256), 0, st, idx, keys_in, ids_in, n, M, lat_begin, offsets);
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

extern "C" int clt_segment_sum(const int32_t* perm, const int32_t* offsets, int64_t M, const float* a, float* out_a,
                               const float* b, float* out_b, clt_stream_t stream) {
  CLT_CHECK_ARG(perm && offsets && a && out_a && (!b || out_b), "clt_segment_sum: null pointer");
  CLT_CHECK_ARG(M > 0 && M < (1ll << 31), "clt_segment_sum: bad M");
  hipLaunchKernelGGL(segment_sum_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, perm,
                     offsets, M, a, out_a, b, out_b);
  CLT_CHECK_LAUNCH("clt_segment_sum");
  return CLT_OK;
}
