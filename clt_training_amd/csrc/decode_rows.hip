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
                                   float* grad_acts, float* g_post, float* g_scale, clt_stream_t stream) {
  int ch;
  int rc = check_rows_args("clt_decode_bwd_rows", perm, offsets, M, acts, v, v_dtype, post_scale, G, g_dtype, d, k, W,
                           w_dtype, grad_acts, g_scale, &ch);
  if (rc != CLT_OK) return rc;
  CLT_CHECK_ARG(grad_W && aligned16(grad_W), "clt_decode_bwd_rows: grad_W is null or misaligned");
  BwdActs bw = {v, v_dtype, post_enc, post_scale, grad_acts, g_post, g_scale, W};
  hipStream_t st = (hipStream_t)stream;
  if (g_dtype == CLT_BF16)
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
                                        float lr, float b1, float b2, float eps, int step, int64_t n_entries,
                                        void* workspace, size_t workspace_bytes, clt_stream_t stream) {
  int ch;
  int rc = check_rows_args("clt_decode_bwd_rows_adam", perm, offsets, M, acts, v, v_dtype, post_scale, G, g_dtype, d, k,
                           W, w_dtype, grad_acts, g_scale, &ch);
  if (rc != CLT_OK) return rc;
  CLT_CHECK_ARG(exp_avg && exp_avg_sq && aligned16(exp_avg) && aligned16(exp_avg_sq) && step >= 1,
                "clt_decode_bwd_rows_adam: bad optimizer state");
  BwdActs bw = {v, v_dtype, post_enc, post_scale, grad_acts, g_post, g_scale, nullptr};
  const AdamRows ad = make_adam_rows(exp_avg, exp_avg_sq, lr, b1, b2, eps, step);
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
