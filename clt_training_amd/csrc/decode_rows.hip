// Decoder backward of one (source, target) pair as ONE row pass over the prebuilt segments (gfx950).
//
// For every decoder row r (read once: it is the Adam parameter row, or the row of W next to a plain gradient
// output) and every entry e = (n, j) of its segment, the gathered loss-gradient row G[n,:] serves twice:
//   grad W[r,:] += a'[e] * G[n,:]      (xformers.py:125-166: the dense weight gradient, written / consumed once)
//   ga[e]        = G[n,:] . W[r,:]     (xformers.py:167-171: gradient wrt the bag weight)
// followed by the backward of the post-encoder affine (sparse_coder.py:136-140) with its per-latent sums.  This
// replaces clt_decode_bwd_acts (a second gather of k decoder rows per token: N*k*d*sizeof(W) bytes per pair) and
// clt_segment_sum; G is only N*d and its rows are re-read from L2 / Infinity Cache.
// HBM-bound: algorithmic bytes = the table rows (Adam: 6*M*d*sizeof(W); plain: 2*M*d) + N*k*d*sizeof(G) gathered.
#include "spt_rows.h"

using namespace clt;

static int check_rows_args(const char* name, const int32_t* perm, const int32_t* offsets, int64_t M, const float* acts,
                           const void* v, int v_dtype, const float* post_scale, const void* G, int g_dtype, int d, int k,
                           const void* W, int w_dtype, float* grad_acts, float* g_scale, int* ch_out) {
  CLT_CHECK_ARG(perm && offsets && acts && v && G && W && grad_acts, "%s: null pointer", name);
  CLT_CHECK_ARG(M > 0 && M < (1ll << 31) && d > 0 && k > 0, "%s: bad sizes", name);
  CLT_CHECK_ARG(valid_dtype(v_dtype) && valid_dtype(g_dtype) && valid_dtype(w_dtype), "%s: bad dtype", name);
  CLT_CHECK_ARG(aligned16(G) && aligned16(W), "%s: pointers must be 16-byte aligned", name);
  CLT_CHECK_ARG(!g_scale || post_scale, "%s: g_scale requires post_scale", name);
  CLT_UNSUPPORTED(g_dtype == w_dtype, "%s: G and W must have the same dtype", name);
  const int E = g_dtype == CLT_BF16 ? 8 : 4;
  CLT_UNSUPPORTED(d % E == 0, "%s: d=%d must be a multiple of %d", name, d, E);
  int chunks = d / E, ch = (chunks + 63) / 64;
  ch = ch <= 1 ? 1 : ch <= 2 ? 2 : ch <= 4 ? 4 : ch <= 8 ? 8 : ch <= 16 ? 16 : -1;
  CLT_UNSUPPORTED(ch > 0, "%s: d=%d too wide", name, d);
  *ch_out = ch;
  return CLT_OK;
}

extern "C" int clt_decode_bwd_rows(const int32_t* perm, const int32_t* offsets, int64_t M, const float* acts,
                                   const void* v, int v_dtype, const float* post_enc, const float* post_scale,
                                   const void* G, int g_dtype, int d, int k, const void* W, int w_dtype, void* grad_W,
                                   float* grad_acts, float* g_post, float* g_scale, int64_t n_entries, void* workspace,
                                   size_t workspace_bytes, clt_stream_t stream) {
  int ch;
  int rc = check_rows_args("clt_decode_bwd_rows", perm, offsets, M, acts, v, v_dtype, post_scale, G, g_dtype, d, k, W,
                           w_dtype, grad_acts, g_scale, &ch);
  if (rc != CLT_OK) return rc;
  CLT_CHECK_ARG(grad_W && aligned16(grad_W), "clt_decode_bwd_rows: grad_W is null or misaligned");
  BwdActs bw = {v, v_dtype, post_enc, post_scale, grad_acts, g_post, g_scale, W};
  hipStream_t st = (hipStream_t)stream;
  const bool lds_ok = rows_lds_ok(d, dtype_size(w_dtype), ch, 1);
  if (lds_ok && g_dtype == CLT_BF16)
    rc = launch_rows_lds<bf16_t, true, false>(ch, st, perm, offsets, M, acts, nullptr, 0, G, d, k, grad_W, nullptr, AdamRows(),
                                              bw, n_entries, workspace, workspace_bytes);
  else if (lds_ok)
    rc = launch_rows_lds<float, true, false>(ch, st, perm, offsets, M, acts, nullptr, 0, G, d, k, grad_W, nullptr, AdamRows(),
                                             bw, n_entries, workspace, workspace_bytes);
  else if (g_dtype == CLT_BF16)
    rc = launch_spt<bf16_t, bf16_t, false, true>(ch, st, perm, offsets, M, acts, nullptr, 0, G, d, k, grad_W, nullptr, AdamRows(), bw);
  else
    rc = launch_spt<float, float, false, true>(ch, st, perm, offsets, M, acts, nullptr, 0, G, d, k, grad_W, nullptr, AdamRows(), bw);
  if (rc != CLT_OK) {
    set_error("clt_decode_bwd_rows: unsupported width");
    return rc;
  }
  CLT_CHECK_LAUNCH("clt_decode_bwd_rows");
  return CLT_OK;
}

extern "C" int clt_decode_bwd_rows_adam(const int32_t* perm, const int32_t* offsets, int64_t M, const float* acts,
                                        const void* v, int v_dtype, const float* post_enc, const float* post_scale,
                                        const void* G, int g_dtype, int d, int k, void* W, void* exp_avg,
                                        void* exp_avg_sq, int w_dtype, float* grad_acts, float* g_post, float* g_scale,
                                        float lr, float b1, float b2, float eps, int step, int project_out,
                                        int64_t n_entries, void* workspace, size_t workspace_bytes,
                                        clt_stream_t stream) {
  int ch;
  int rc = check_rows_args("clt_decode_bwd_rows_adam", perm, offsets, M, acts, v, v_dtype, post_scale, G, g_dtype, d, k,
                           W, w_dtype, grad_acts, g_scale, &ch);
  if (rc != CLT_OK) return rc;
  CLT_CHECK_ARG(exp_avg && exp_avg_sq && aligned16(exp_avg) && aligned16(exp_avg_sq) && step >= 1,
                "clt_decode_bwd_rows_adam: bad optimizer state");
  BwdActs bw = {v, v_dtype, post_enc, post_scale, grad_acts, g_post, g_scale, nullptr};
  const AdamRows ad = make_adam_rows(exp_avg, exp_avg_sq, lr, b1, b2, eps, step, project_out != 0);
  hipStream_t st = (hipStream_t)stream;
  const bool lds_ok = rows_lds_ok(d, dtype_size(w_dtype), ch);
  if (lds_ok && g_dtype == CLT_BF16)
    rc = launch_rows_lds<bf16_t, true>(ch, st, perm, offsets, M, acts, nullptr, 0, G, d, k, W, nullptr, ad, bw, n_entries, workspace, workspace_bytes);
  else if (lds_ok)
    rc = launch_rows_lds<float, true>(ch, st, perm, offsets, M, acts, nullptr, 0, G, d, k, W, nullptr, ad, bw, n_entries, workspace, workspace_bytes);
  else if (g_dtype == CLT_BF16)
    rc = launch_spt<bf16_t, bf16_t, true, true>(ch, st, perm, offsets, M, acts, nullptr, 0, G, d, k, W, nullptr, ad, bw);
  else
    rc = launch_spt<float, float, true, true>(ch, st, perm, offsets, M, acts, nullptr, 0, G, d, k, W, nullptr, ad, bw);
  if (rc != CLT_OK) {
    set_error("clt_decode_bwd_rows_adam: unsupported width");
    return rc;
  }
  CLT_CHECK_LAUNCH("clt_decode_bwd_rows_adam");
  return CLT_OK;
}

extern "C" size_t clt_rows_multi_workspace_bytes(int64_t n_entries, int d, int n_jobs) {
  if (n_entries <= 0 || d <= 0 || n_jobs <= 0) return 256;
  return (((rows_long_workspace_bytes(n_entries, d) + 255) & ~(size_t)255)) * (size_t)n_jobs + 512;
}

extern "C" int clt_decode_bwd_rows_adam_multi(const clt_rows_job* h_jobs, int n_jobs, int64_t M, int v_dtype,
                                              const void* G, int g_dtype, int d, int k, int w_dtype, float lr, float b1,
                                              float b2, float eps, int step, int project_out, int64_t n_entries,
                                              void* workspace, size_t workspace_bytes, clt_stream_t stream) {
  CLT_CHECK_ARG(h_jobs && n_jobs >= 1 && n_jobs <= CLT_ROWS_MAX_JOBS, "clt_decode_bwd_rows_adam_multi: 1..%d jobs",
                CLT_ROWS_MAX_JOBS);
  static_assert(CLT_ROWS_MAX_JOBS == ROWS_MAX_JOBS, "header and kernel disagree");
  int ch = 0;
  RowsJobs jobs = {};
  for (int i = 0; i < n_jobs; ++i) {
    const clt_rows_job& h = h_jobs[i];
    const int rc = check_rows_args("clt_decode_bwd_rows_adam_multi", h.perm, h.offsets, M, h.acts, h.v, v_dtype,
                                   h.post_scale, G, g_dtype, d, k, h.W, w_dtype, h.grad_acts, h.g_scale, &ch);
    if (rc != CLT_OK) return rc;
    CLT_CHECK_ARG(h.exp_avg && h.exp_avg_sq && aligned16(h.exp_avg) && aligned16(h.exp_avg_sq) && step >= 1,
                  "clt_decode_bwd_rows_adam_multi: bad optimizer state");
    RowsJob& jb = jobs.j[i];
    jb.perm = h.perm;
    jb.offsets = h.offsets;
    jb.vals = h.acts;
    jb.out = h.W;
    jb.ad = make_adam_rows(h.exp_avg, h.exp_avg_sq, lr, b1, b2, eps, step, project_out != 0);
    jb.bw = BwdActs{h.v, v_dtype, h.post_enc, h.post_scale, h.grad_acts, h.g_post, h.g_scale, nullptr};
  }
  hipStream_t st = (hipStream_t)stream;
  if (!rows_lds_ok(d, dtype_size(w_dtype), ch)) {  // rows wider than the LDS-staged kernel takes: pair by pair
    for (int i = 0; i < n_jobs; ++i) {
      const clt_rows_job& h = h_jobs[i];
      const int rc = clt_decode_bwd_rows_adam(h.perm, h.offsets, M, h.acts, h.v, v_dtype, h.post_enc, h.post_scale, G,
                                              g_dtype, d, k, h.W, h.exp_avg, h.exp_avg_sq, w_dtype, h.grad_acts,
                                              h.g_post, h.g_scale, lr, b1, b2, eps, step, project_out, n_entries,
                                              workspace, workspace_bytes, stream);
      if (rc != CLT_OK) return rc;
    }
    return CLT_OK;
  }
  const int rc = g_dtype == CLT_BF16
                     ? launch_rows_lds_multi<bf16_t>(ch, st, jobs, n_jobs, M, G, d, k, n_entries, workspace, workspace_bytes)
                     : launch_rows_lds_multi<float>(ch, st, jobs, n_jobs, M, G, d, k, n_entries, workspace, workspace_bytes);
  if (rc != CLT_OK) {
    set_error("clt_decode_bwd_rows_adam_multi: unsupported width");
    return rc;
  }
  CLT_CHECK_LAUNCH("clt_decode_bwd_rows_adam_multi");
  return CLT_OK;
}

// ---- SAE extras (sparse_coder.py:687-711; trainer.py:595-596, 793-795): row-wise passes over the decoder table, one wave
//      per row, 16-byte chunks, fp32 arithmetic.  HBM-bound: one read + one write of the table (the reference's torch
//      expressions make 3-6 passes).
namespace clt {

template <typename T, int CH>
__global__ __launch_bounds__(256) void rows_unit_norm_kernel(T* __restrict__ W, int64_t M, int d, float eps) {
  constexpr int E = Chunk<T>::E;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nchunks = d / E;
  T* wr = W + row * (int64_t)d;
  float w[CH][E];
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nchunks) {
      load_row_chunk<T, E>(wr + ch * E, w[c]);
#pragma unroll
      for (int e = 0; e < E; ++e) ss = fmaf(w[c][e], w[c][e], ss);
    }
  }
  ss = wave_sum_dpp(ss);
  float nrm = sqrtf(ss);
  if (sizeof(T) == 2) nrm = bf16_bits_to_f32(f32_to_bf16_bits(nrm));  // torch.norm returns the table's dtype
  const float den = nrm + eps;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nchunks) {
#pragma unroll
      for (int e = 0; e < E; ++e) w[c][e] = w[c][e] / den;
      store_row_chunk<T, E>(wr + ch * E, w[c]);
    }
  }
}

template <typename T, int CH>
__global__ __launch_bounds__(256) void rows_project_out_kernel(T* __restrict__ G, const T* __restrict__ W, int64_t M,
                                                               int d) {
  constexpr int E = Chunk<T>::E;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nchunks = d / E;
  T* gr = G + row * (int64_t)d;
  const T* wr = W + row * (int64_t)d;
  float g[CH][E], w[CH][E];
  float dot = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nchunks) {
      load_row_chunk<T, E>(gr + ch * E, g[c]);
      load_row_chunk<T, E>(wr + ch * E, w[c]);
#pragma unroll
      for (int e = 0; e < E; ++e) dot = fmaf(g[c][e], w[c][e], dot);
    }
  }
  dot = wave_sum_dpp(dot);
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nchunks) {
#pragma unroll
      for (int e = 0; e < E; ++e) g[c][e] = project_elem<sizeof(T) == 2>(g[c][e], w[c][e], dot);
      store_row_chunk<T, E>(gr + ch * E, g[c]);
    }
  }
}

template <typename T>
static int launch_rows_simple(int which, void* A, const void* B, int64_t M, int d, float eps, hipStream_t st) {
  const int E = Chunk<T>::E;
  int chunks = d / E, ch = (chunks + 63) / 64;
  ch = ch <= 1 ? 1 : ch <= 2 ? 2 : ch <= 4 ? 4 : ch <= 8 ? 8 : ch <= 16 ? 16 : -1;
  if (ch < 0 || d % E != 0) return CLT_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)((M + 3) / 4));
#define LAUNCH(CH)                                                                                               \
  do {                                                                                                           \
    if (which == 0) hipLaunchKernelGGL((rows_unit_norm_kernel<T, CH>), grid, dim3(256), 0, st, (T*)A, M, d, eps); \
    else hipLaunchKernelGGL((rows_project_out_kernel<T, CH>), grid, dim3(256), 0, st, (T*)A, (const T*)B, M, d); \
  } while (0)
  switch (ch) {
    case 1: LAUNCH(1); break;
    case 2: LAUNCH(2); break;
    case 4: LAUNCH(4); break;
    case 8: LAUNCH(8); break;
    default: LAUNCH(16); break;
  }
#undef LAUNCH
  return CLT_OK;
}

}  // namespace clt

extern "C" int clt_rows_unit_norm(void* W, int w_dtype, int64_t M, int d, float eps, clt_stream_t stream) {
  CLT_CHECK_ARG(W && aligned16(W) && M > 0 && d > 0 && valid_dtype(w_dtype), "clt_rows_unit_norm: bad argument");
  const int rc = w_dtype == CLT_BF16 ? launch_rows_simple<bf16_t>(0, W, nullptr, M, d, eps, (hipStream_t)stream)
                                     : launch_rows_simple<float>(0, W, nullptr, M, d, eps, (hipStream_t)stream);
  if (rc != CLT_OK) {
    set_error("clt_rows_unit_norm: unsupported width d=%d", d);
    return rc;
  }
  CLT_CHECK_LAUNCH("clt_rows_unit_norm");
  return CLT_OK;
}

extern "C" int clt_rows_project_out(void* G, const void* W, int dtype, int64_t M, int d, clt_stream_t stream) {
  CLT_CHECK_ARG(G && W && aligned16(G) && aligned16(W) && M > 0 && d > 0 && valid_dtype(dtype),
                "clt_rows_project_out: bad argument");
  const int rc = dtype == CLT_BF16 ? launch_rows_simple<bf16_t>(1, G, W, M, d, 0.f, (hipStream_t)stream)
                                   : launch_rows_simple<float>(1, G, W, M, d, 0.f, (hipStream_t)stream);
  if (rc != CLT_OK) {
    set_error("clt_rows_project_out: unsupported width d=%d", d);
    return rc;
  }
  CLT_CHECK_LAUNCH("clt_rows_project_out");
  return CLT_OK;
}
