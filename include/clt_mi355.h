/*
 * clt_mi355.h -- C ABI of libcltmi355.so: the MI355X (gfx950) implementation of the
 * SAE / cross-layer-transcoder train-step hot path of EleutherAI/clt-training.
 *
 * Conventions:
 *   - every pointer is a DEVICE pointer unless its name starts with h_;
 *   - every entry point returns 0 on success and a negative clt_status on
 *     failure, never throws, never synchronises and never allocates: outputs
 *     and workspaces are passed in by the caller (the reference's ops allocate
 *     with torch.empty/zeros: kernels.py:81,244; xformers.py:56-58,208-211);
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued on it;
 *   - dtype codes: CLT_F32 = 0, CLT_BF16 = 1;
 *   - indices at the seam are int64 exactly as torch.topk returns them
 *     (fused_encoder.py:30, xformers.py:59); latent counts must be < 2^31;
 *   - row-major, contiguous tensors; d (feature width) must be a multiple of
 *     8 (bf16) or 4 (fp32) so that every row is a whole number of 16-byte chunks.
 *
 * Citations are file:line under /root/reference/.
 */
#ifndef CLT_MI355_H
#define CLT_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* clt_stream_t;

enum clt_dtype { CLT_F32 = 0, CLT_BF16 = 1 };

enum clt_status {
  CLT_OK = 0,
  CLT_ERR_BAD_ARG = -1,      /* null pointer, non-positive size, misaligned pointer   */
  CLT_ERR_UNSUPPORTED = -2,  /* shape / dtype outside what the kernels are built for  */
  CLT_ERR_LAUNCH = -3,       /* hipGetLastError() after a launch was not hipSuccess   */
  CLT_ERR_WORKSPACE = -4     /* workspace smaller than the *_workspace_bytes() answer */
};

/* Library version (major*10000 + minor*100 + patch) and last error text (thread local). */
int clt_version(void);
const char* clt_last_error(void);

/* ------------------------------------------------------------------ encoder
 * Replaces the no_grad body of sparsify.fused_encoder.fused_encoder
 * (fused_encoder.py:310-312, nanogpt.py:321):  pre = relu(x @ W^T + b).
 *   x [N,d], W [M,d] in `in_dtype` (bf16 -> bf16 MFMA, f32 -> exact f32 MFMA);
 *   bias [M] in `bias_dtype` or NULL;  pre [N,M] in `out_dtype`, ld = M.
 * fp32 accumulation, one rounding to out_dtype after the bias add.  */
int clt_encoder_gemm_relu(const void* x, const void* W, int in_dtype, const void* bias, int bias_dtype,
                          void* pre, int out_dtype, int64_t N, int64_t M, int d, clt_stream_t stream);

/* Row top-k: replaces torch.topk(pre, k, dim=1, sorted=False) (fused_encoder.py:30,372)
 * and, with cand_idx != NULL, the TP merge top-k + gather (fused_encoder.py:359-360).
 *   vals [N,L] (leading dimension ld, `val_dtype`), values must be >= 0 (post-ReLU);
 *   cand_idx [N,L] int64 or NULL (NULL: the column number is the index);
 *   counts  [N] int32 or NULL: number of valid leading entries of each row (NULL: L);
 *   out_vals [N,k] in `out_dtype`, out_idx [N,k] int64.
 * Tie rule: larger value first, then LOWEST index.  Output order: value
 * descending, index ascending.  With cand_idx == NULL zero entries are never
 * candidates; a row with fewer than k positive entries is completed with
 * (value 0, the lowest column numbers not yet selected) -- exactly what the
 * tie rule gives for a post-ReLU row.  With cand_idx != NULL zeros are ordinary
 * candidates and every row must hold at least k of them.   k <= 256.   */
int clt_topk_rows(const void* vals, int val_dtype, int64_t ld, const int64_t* cand_idx, const int32_t* counts,
                  int64_t N, int64_t L, int k, void* out_vals, int out_dtype, int64_t* out_idx,
                  clt_stream_t stream);

/* clt_encode_topk_fwd for n_layers layers of the same shape in ONE set of launches (all layer inputs of a step are known
 * at its start, and no encoder reads what a decoder writes within the step: trainer.py:492-545 encodes them one hook at
 * a time only because the hooks fire that way).  Layer l reads x + l * x_layer_stride + t * x_row_stride (elements; a
 * block of layer inputs side by side has x_row_stride = n_layers * d, x_layer_stride = d), multiplies with h_W[l]
 * (HOST array of device pointers) and adds h_bias[l]; outputs are stacked: out_vals / out_idx [n_layers * N_l, k].
 * Results are those of n_layers clt_encode_topk_fwd calls bit for bit.  Takes bf16 operands on the fused path with
 * N_l a multiple of 2048 and at most 16 layers (CLT_ERR_UNSUPPORTED otherwise: call layer by layer).  workspace:
 * clt_encode_topk_workspace_bytes(n_layers * N_l, M, out_dtype).                                                    */
int clt_encode_topk_fwd_layers(const void* x, int64_t x_row_stride, int64_t x_layer_stride, const void* const* h_W,
                               int in_dtype, const void* const* h_bias, int bias_dtype, int n_layers, int64_t N_l,
                               int64_t M, int d, int k, void* out_vals, int out_dtype, int64_t* out_idx,
                               void* workspace, size_t workspace_bytes, clt_stream_t stream);

/* The exchange format of the TP top-k merge (fused_encoder.py:347-360 all-gathers values and int64 indices as two
 * tensors; here one int64 per candidate carries both through ONE collective):
 *   packed = (bits of float32(value) << 32) | (idx + idx_offset),  values >= 0, idx + idx_offset < 2^32.
 * clt_candidates_pack:   vals [n] in `val_dtype`, idx [n] int64 -> packed [n] int64.
 * clt_candidates_unpack: packed as the all-to-all delivers it, [groups][rows][k] (group r = what rank r sent), ->
 *   vals / idx [rows][groups * k]: row t holds the candidates of token t from every rank, rank by rank (groups = 1:
 *   a plain unpack).  vals in `val_dtype` (the value was widened exactly, so the narrowing is exact too).           */
int clt_candidates_pack(const void* vals, int val_dtype, const int64_t* idx, int64_t idx_offset, int64_t n,
                        int64_t* packed, clt_stream_t stream);
int clt_candidates_unpack(const int64_t* packed, int64_t groups, int64_t rows, int64_t k, void* vals, int val_dtype,
                          int64_t* idx, clt_stream_t stream);

/* Encoder GEMM + bias + ReLU + row top-k in one call (whole of fused_encoder.py:310-322,372 for
 * activation="topk"); results are those of clt_encoder_gemm_relu -> clt_topk_rows bit for bit.
 * `workspace`: clt_encode_topk_workspace_bytes(N, M, out_dtype) bytes = room for the pre-activation matrix
 * [N,M] in out_dtype, plus max(1/8 of that, 8 KiB per token), plus a tail that holds the device-side fallback
 * flags (one per 256-token tile).  bf16 operands with M >= 80 k never form that matrix: the same bytes hold, laid
 * over its front, the sample matrix [N, M/S], the sample's top-k (values, int64 indices), the per-token thresholds
 * tau[N] and the candidates -- where the 16x16x32 MFMA kernels run (d % 64 == 0, d >= 128, M % 16 == 0: every
 * BASELINE width) a k-step-major copy of x, the candidate stream of the filter pass (16-byte entries, one region per
 * wave of the launch) and the per-token candidate lists built from it; on other shapes 8 x u32 list slots per token
 * and 64 latents with u8 lengths.  The full matrix is written only by the device-gated fallback launches (a token
 * failed verification or a candidate buffer ran full: only the 256-token tiles concerned are redone).
 * Other dtypes / shapes: GEMM into the workspace, then clt_topk_rows over it.  Contents are scratch:
 * nothing in the workspace needs to survive the call, and calls sharing a workspace must be stream-ordered. */
size_t clt_encode_topk_workspace_bytes(int64_t N, int64_t M, int out_dtype);
int clt_encode_topk_fwd(const void* x, const void* W, int in_dtype, const void* bias, int bias_dtype,
                        int64_t N, int64_t M, int d, int k, void* out_vals, int out_dtype, int64_t* out_idx,
                        void* workspace, size_t workspace_bytes, clt_stream_t stream);

/* groupmax activation: replaces `preacts.unflatten(-1, (k, -1)).max(dim=-1)` + the group offsets
 * (fused_encoder.py:408-413).  vals [N,M] (leading dimension ld), k must divide M; out_vals [N,k] = the largest
 * entry of each run of M/k consecutive columns, out_idx [N,k] int64 = its column (ties: the lowest column, torch's
 * CPU rule; an all-zero post-ReLU group reports its first column with value 0).  Output order: group order.
 * Under latent sharding each rank calls it on its [N, M/tp] block with k/tp groups and adds its column offset
 * (fused_encoder.py:374-389).  */
int clt_groupmax_rows(const void* vals, int val_dtype, int64_t ld, int64_t N, int64_t M, int k, void* out_vals,
                      int out_dtype, int64_t* out_idx, clt_stream_t stream);

/* Fused convenience for activation="groupmax" (fused_encoder.py:310-312, 373-413): encoder GEMM + bias + ReLU into
 * the workspace (clt_encode_topk_workspace_bytes), then clt_groupmax_rows.  */
int clt_encode_groupmax_fwd(const void* x, const void* W, int in_dtype, const void* bias, int bias_dtype,
                            int64_t N, int64_t M, int d, int k, void* out_vals, int out_dtype, int64_t* out_idx,
                            void* workspace, size_t workspace_bytes, clt_stream_t stream);

/* ------------------------------------------------------------------ decoder
 * Replaces decoder_impl / xformers_embedding_bag forward (utils.py:442-457,
 * xformers.py:15-90) with the post-encoder affine (sparse_coder.py:136-140) and the
 * decoder bias add (sparse_coder.py:661) optionally fused:
 *   a'[n,j] = (v[n,j] + post_enc[idx]*(v>0)) * post_scale[idx]      (either may be NULL)
 *   out[n,:] = (flags & CLT_DECODE_ACCUMULATE ? out[n,:] : 0) + bias + sum_j a'[n,j] * W[idx[n,j],:]
 * v [N,k] in `v_dtype`; W [M,d] in `w_dtype`; post_enc/post_scale/bias fp32;
 * out [N,d] in `out_dtype`; acts_out [N,k] fp32 or NULL receives a'.  fp32 accumulation.
 * Latent sharding (replaces parallelize_decoder's mask-and-relocate, utils.py:520-530): W holds the M rows
 * [lat_begin, lat_begin + M) of the full table; entries whose index falls outside are SKIPPED (their a' is
 * reported as 0), so each shard's output is a partial sum to be combined by a reduce-scatter. lat_begin = 0
 * and M = full table size is the unsharded case. post_enc / post_scale are indexed by the LOCAL row. */
#define CLT_DECODE_ACCUMULATE 1 /* out += ... instead of out = ...                                   */
#define CLT_DECODE_ROUND_ACTS 2 /* round a' to bf16 first: top_acts.to(self.dtype), sparse_coder.py:660 */
int clt_decode_fwd(const int64_t* idx, const void* v, int v_dtype, const float* post_enc, const float* post_scale,
                   const void* W, int w_dtype, int64_t M, int64_t lat_begin, int d, const float* bias, void* out,
                   int out_dtype, int flags, float* acts_out, int64_t N, int k, clt_stream_t stream);

/* Gradient w.r.t. the bag weights (xformers.py:167-171: g_acts[n,j] = G[n,:].W[idx[n,j],:])
 * with the backward of the post-encoder affine fused (autograd of sparse_coder.py:136-140):
 *   ga = G[n,:] . W[idx,:];  u = ga * post_scale[idx]
 *   grad_acts[n,j]  (+)= u                       (accumulate_grad_acts: CLT leaf grad, sparse_coder.py:72-76)
 *   e_post[n,j]  = u * (v>0)                     (NULL if post_enc grads are not needed)
 *   e_scale[n,j] = ga * (v + post_enc[idx]*(v>0))  (NULL unless post_scale is given)
 * G [N,d] in `g_dtype`; grad_acts/e_post/e_scale fp32 [N,k].  Entries outside the shard
 * [lat_begin, lat_begin + M) get 0 (and leave grad_acts untouched when accumulating).      */
int clt_decode_bwd_acts(const int64_t* idx, const void* v, int v_dtype, const float* post_enc,
                        const float* post_scale, const void* W, int w_dtype, int64_t M, int64_t lat_begin, int d,
                        const void* G, int g_dtype, float* grad_acts, int accumulate_grad_acts, float* e_post,
                        float* e_scale, int64_t N, int k, clt_stream_t stream);

/* CLT fan-out in one launch (runner.py:43-71: the target layer's prediction is the sum of the decodes of EVERY
 * live source layer; sparse_coder.py:171-176 picks each source's slot).  All sources share N, k, M, d and the
 * dtypes; `h_sources` is a HOST array of n_sources <= CLT_DECODE_MAX_SOURCES descriptors (copied into the kernel
 * arguments: nothing is retained).  One wave per token walks the entries of all sources as one list, so a latent
 * shard that owns only k/tp entries per source still keeps 8 row gathers in flight, and the output row is
 * written (or the gradient row read) once per target instead of once per (source, target) pair.
 *   fwd:  out[n,:] = (ACCUMULATE ? out[n,:] : 0) + bias + sum_s weight_s * sum_j a'_s[n,j] * W_s[idx_s[n,j],:]
 *         (fp32 accumulation over all sources, ONE rounding to out_dtype; acts_out_s receives a'_s)
 *   bwd:  as clt_decode_bwd_acts per source with G replaced by weight_s * G (weight: 1, or 1/divide_by of
 *         runner.py:46-49 for the other sources when divide_cross_layer is set).                            */
#define CLT_DECODE_MAX_SOURCES 16
typedef struct clt_decode_source {
  const int64_t* idx;      /* [N,k] */
  const void* v;           /* [N,k] v_dtype */
  const float* post_enc;   /* [M] or NULL */
  const float* post_scale; /* [M] or NULL */
  const void* W;           /* [M,d] w_dtype: rows [lat_begin, lat_begin + M) of the source's table for this target */
  float* acts_out;         /* fwd: [N,k] or NULL */
  float* grad_acts;        /* bwd: [N,k] */
  float* e_post;           /* bwd: [N,k] or NULL */
  float* e_scale;          /* bwd: [N,k] or NULL (requires post_scale) */
  int64_t lat_begin;
  float weight;
  int32_t accumulate_grad_acts;
} clt_decode_source;
int clt_decode_fwd_multi(const clt_decode_source* h_sources, int n_sources, int v_dtype, int w_dtype, int64_t M, int d,
                         const float* bias, void* out, int out_dtype, int flags, int64_t N, int k,
                         clt_stream_t stream);
int clt_decode_bwd_acts_multi(const clt_decode_source* h_sources, int n_sources, int v_dtype, int w_dtype, int64_t M,
                              int d, const void* G, int g_dtype, int64_t N, int k, clt_stream_t stream);

/* ------------------------------------------------------------------ segments
 * Stable sort of the N*k (token, slot) entries by latent: replaces the COO sort of
 * kernels.py:42-51 and the count/cumsum/reverse-map kernels of xformers.py:93-122,
 * deterministically (entries of one latent stay in ascending entry id).
 *   perm [N*k] int32: entry ids (n*k + j) grouped by latent; offsets [M+2] int32: offsets[r] .. offsets[r+1] is the
 *   segment of local row r, offsets[M] = number of entries this shard owns, offsets[M+1] = 1 if some segment
 *   holds more than 512 entries (read by the long-row helper of clt_spt_dense_adam / clt_decode_bwd_rows_adam).
 * Sharded: only entries with idx in [lat_begin, lat_begin + M) form segments (local row = idx - lat_begin);
 * the others are dropped by the first sorting pass: perm[offsets[M] ...] is unspecified.                 */
size_t clt_segments_workspace_bytes(int64_t n_entries, int64_t M);
int clt_segments_build(const int64_t* idx, int64_t n_entries, int64_t M, int64_t lat_begin, int32_t* perm,
                       int32_t* offsets, void* workspace, size_t workspace_bytes, clt_stream_t stream);

/* sparse^T @ dense over prebuilt segments: replaces triton_sparse_transpose_dense_matmul
 * (kernels.py:14-54,113-216: encoder grad-W, fused_encoder.py:150-155) and
 * aggregate_gradient_for_embedding_k (xformers.py:125-176: decoder grad-W):
 *   out[r,:] = sum_{e in segment r} vals[e] * dense[e / k, :]     (every row written, zeros if empty)
 *   rowsum[r] = sum_{e in segment r} vals[e] * (mask ? mask_vals[e] > 0 : 1)   (NULL to skip; this is
 *               grad_b of fused_encoder.py:140-143)
 * vals [N*k] fp32; vals_gate [N*k] (`gate_dtype`) or NULL: when given, vals[e] is multiplied by
 * (vals_gate[e] > 0) -- the (values > 0) mask of fused_encoder.py:97; dense [N,d] `dense_dtype`;
 * out [M,d] `out_dtype`.  Atomic-free and bitwise reproducible.  (bf16 per-entry arrays -- vals_gate here, v of the
 * decoder row passes -- are read as the aligned 32-bit word around the element: their natural 2-byte alignment is
 * enough, and up to two bytes before the first / after the last element are touched, inside any allocation granule.)
 * n_entries / workspace: the optional long-segment
 * pre-reduction described at clt_spt_dense_adam (workspace NULL, or dense_dtype != out_dtype: every row is walked
 * by one wave).                                                                                  */
int clt_spt_dense(const int32_t* perm, const int32_t* offsets, int64_t M, const float* vals, const void* vals_gate,
                  int gate_dtype, const void* dense, int dense_dtype, int d, int k, void* out, int out_dtype,
                  float* rowsum, int64_t n_entries, void* workspace, size_t workspace_bytes, clt_stream_t stream);

/* clt_spt_dense with the optimizer step applied to each finished row, for steps in which the gradient has no
 * other consumer (grad_acc_steps == 1, no clipping): the fusion of the dense grad-W write of xformers.py:179-231 /
 * fused_encoder.py:150-155 with the torch.optim.Adam update of trainer.py:178-186,797-802 for that tensor.
 *   g = round_to(p_dtype, sum_{e in segment r} vals[e] * dense[e / k, :])   (what .grad would have held)
 *   m, v, p updated exactly as clt_adam_step does (lr is the group's current lr; step is 1-based).
 * p, m, v [M,d] `p_dtype`, updated in place; the gradient row never reaches memory (4 of the 16 bytes per
 * parameter that grad write + Adam move).  rowsum as in clt_spt_dense.  Bitwise equal to clt_spt_dense followed
 * by clt_adam_step.
 * Long rows: a latent active on most tokens owns thousands of entries (the reference balances such embeddings by a
 * count-descending schedule and atomics, xformers.py:195,214-230).  With a `workspace` of
 * clt_rows_workspace_bytes(n_entries, d) bytes (n_entries = length of perm = N*k) every segment of more than 512
 * entries is pre-reduced in blocks of 512 by a helper launch and its partial rows are added in block order:
 * deterministic, atomic-free; only the fp32 summation order of THOSE rows differs from clt_spt_dense.  workspace
 * NULL: every row is walked by one wave.                                                                        */
size_t clt_rows_workspace_bytes(int64_t n_entries, int d);
int clt_spt_dense_adam(const int32_t* perm, const int32_t* offsets, int64_t M, const float* vals, const void* vals_gate,
                       int gate_dtype, const void* dense, int dense_dtype, int d, int k, void* p, void* m, void* v,
                       int p_dtype, float* rowsum, float lr, float b1, float b2, float eps, int step,
                       int64_t n_entries, void* workspace, size_t workspace_bytes, clt_stream_t stream);

/* n_jobs clt_spt_dense_adam passes that share M, d, k, the dtypes and the Adam hyper-parameters as one launch (the
 * encoder backward of every layer that finishes in the same hook call -- all of them at the last layer when
 * cross_layer covers the model, runner.py:56-59, 193-197); bitwise the results of n_jobs separate calls.  h_jobs is a
 * HOST array; workspace: clt_rows_multi_workspace_bytes(n_entries, d, n_jobs) or NULL.                       */
typedef struct clt_spt_job {
  const int32_t* perm;
  const int32_t* offsets;
  const float* vals;      /* [N*k] fp32 */
  const void* vals_gate;  /* [N*k] gate_dtype or NULL */
  const void* dense;      /* [N,d] dense_dtype: this layer's encoder input */
  void* p;                /* [M,d] p_dtype, updated in place */
  void* m;
  void* v;
  float* rowsum;          /* [M] or NULL */
} clt_spt_job;
int clt_spt_dense_adam_multi(const clt_spt_job* h_jobs, int n_jobs, int64_t M, int gate_dtype, int dense_dtype, int d,
                             int k, int p_dtype, float lr, float b1, float b2, float eps, int step, int64_t n_entries,
                             void* workspace, size_t workspace_bytes, clt_stream_t stream);

/* Decoder backward of one (source, target) pair as ONE row pass: the weight gradient of clt_spt_dense (xformers.py:
 * 125-166), the gradient wrt the bag weights of clt_decode_bwd_acts (xformers.py:167-171) and the per-latent sums
 * of clt_segment_sum together.  For every decoder row r and every entry e = (n, j) of its segment:
 *   grad_W[r,:] += acts[e] * G[n,:];   ga = G[n,:] . W[r,:];   u = ga * post_scale[r]
 *   grad_acts[e] += u;   g_post[r] += u * (v[e] > 0);   g_scale[r] += ga * (v[e] + post_enc[r] * (v[e] > 0))
 * so the k decoder rows per token are NOT gathered a second time: the row of W is in registers when G[n,:] arrives.
 * acts [N*k] fp32 = a' (what clt_decode_fwd reported); v [N*k] `v_dtype` = the top-k values; G [N,d] and W [M,d]
 * share `g_dtype` == `w_dtype`; grad_acts [N*k] fp32 is ACCUMULATED into (entries of other shards are not
 * touched: zero it before the source's first target); g_post / g_scale [M] fp32 or NULL are overwritten.
 * clt_decode_bwd_rows writes the gradient to grad_W [M,d] (`w_dtype`); clt_decode_bwd_rows_adam consumes it in the
 * Adam update of (W, exp_avg, exp_avg_sq) exactly as clt_spt_dense_adam does, with ga taken against the row
 * BEFORE the update.  Results are bitwise those of the three separate entry points (long rows, when a workspace
 * is given to the _adam form: see clt_spt_dense_adam).                                                     */
int clt_decode_bwd_rows(const int32_t* perm, const int32_t* offsets, int64_t M, const float* acts, const void* v,
                        int v_dtype, const float* post_enc, const float* post_scale, const void* G, int g_dtype, int d,
                        int k, const void* W, int w_dtype, void* grad_W, float* grad_acts, float* g_post,
                        float* g_scale, int64_t n_entries, void* workspace, size_t workspace_bytes,
                        clt_stream_t stream);
int clt_decode_bwd_rows_adam(const int32_t* perm, const int32_t* offsets, int64_t M, const float* acts, const void* v,
                             int v_dtype, const float* post_enc, const float* post_scale, const void* G, int g_dtype,
                             int d, int k, void* W, void* exp_avg, void* exp_avg_sq, int w_dtype, float* grad_acts,
                             float* g_post, float* g_scale, float lr, float b1, float b2, float eps, int step,
                             int project_out, int64_t n_entries, void* workspace, size_t workspace_bytes,
                             clt_stream_t stream);

/* The decoder backward of EVERY live source of one target (runner.py:43-71 in reverse: the same loss gradient G
 * for all of them, one table W_decs[j - i] per source) as one launch: n_jobs clt_decode_bwd_rows_adam passes that
 * share M, d, k, G, the dtypes and the Adam hyper-parameters; results are bitwise those of n_jobs separate calls.
 * One ramp-up / tail and one long-segment helper launch per target instead of one per (source, target) pair.
 * h_jobs is a HOST array (copied into the launch); workspace: clt_rows_multi_workspace_bytes(n_entries, d, n_jobs)
 * or NULL (long segments then are walked by one wave each).                                                  */
#define CLT_ROWS_MAX_JOBS 16
typedef struct clt_rows_job {
  const int32_t* perm;     /* the source's segments (clt_segments_build) */
  const int32_t* offsets;
  const float* acts;       /* [N*k] a' of this (source, target) pair */
  const void* v;           /* [N*k] v_dtype */
  const float* post_enc;   /* [M] or NULL */
  const float* post_scale; /* [M] or NULL */
  void* W;                 /* [M,d] w_dtype, updated in place */
  void* exp_avg;
  void* exp_avg_sq;
  float* grad_acts;        /* [N*k], accumulated into */
  float* g_post;           /* [M] or NULL */
  float* g_scale;          /* [M] or NULL (requires post_scale) */
} clt_rows_job;
size_t clt_rows_multi_workspace_bytes(int64_t n_entries, int d, int n_jobs);
int clt_decode_bwd_rows_adam_multi(const clt_rows_job* h_jobs, int n_jobs, int64_t M, int v_dtype, const void* G,
                                   int g_dtype, int d, int k, int w_dtype, float lr, float b1, float b2, float eps,
                                   int step, int project_out, int64_t n_entries, void* workspace,
                                   size_t workspace_bytes, clt_stream_t stream);

/* ------------------------------------------------------------------ SAE extras
 * Plain SAEs keep unit-norm decoder rows (sparse_coder.py:687-711; trainer.py:595-596, 793-795):
 *   clt_rows_unit_norm:   W[r,:] /= (||W[r,:]|| + eps)            (set_decoder_norm_to_unit_norm; eps = finfo(dtype).eps)
 *   clt_rows_project_out: G[r,:] -= (G[r,:] . W[r,:]) W[r,:]      (remove_gradient_parallel_to_decoder_directions)
 * one read + one write of the [M,d] table each (the torch expressions make 3-6 passes), in place, fp32 arithmetic,
 * one rounding per stored value.  `project_out` != 0 in clt_decode_bwd_rows_adam applies the projection to the
 * finished gradient row inside the row pass (bitwise equal to clt_decode_bwd_rows -> clt_rows_project_out ->
 * clt_adam_step), so an SAE's decoder backward, projection and Adam update are ONE pass as well.              */
int clt_rows_unit_norm(void* W, int w_dtype, int64_t M, int d, float eps, clt_stream_t stream);
int clt_rows_project_out(void* G, const void* W, int dtype, int64_t M, int d, clt_stream_t stream);

/* Per-latent sums of up to two per-entry arrays (post_enc / post_enc_scale grads: the
 * scatter-adds autograd generates for sparse_coder.py:136-140):
 *   out_a[r] = sum_{e in segment r} a[e];  out_b likewise (b/out_b may be NULL).  fp32.  */
int clt_segment_sum(const int32_t* perm, const int32_t* offsets, int64_t M, const float* a, float* out_a,
                    const float* b, float* out_b, clt_stream_t stream);

/* ------------------------------------------------------------------ FVU
 * Replaces sparse_coder.py:207-221.
 *   stats(y, mask): colsum[d] = sum_n m*y, count = sum m, then tv = sum m (y - colsum/count)^2.
 *   loss(y, yhat, mask): l2 = sum (m (y - yhat))^2, fvu = l2 / tv, and
 *   G[n,:] = gscale * (-2 / tv) * m (y - yhat)  in `g_dtype`  (d fvu / d yhat times gscale).
 * mask [N] uint8 (torch.bool) or NULL.  scalars (device, fp32): out[0] = fvu, out[1] = l2, out[2] = tv.
 * workspace: clt_fvu_workspace_bytes(N, d).                                                   */
size_t clt_fvu_workspace_bytes(int64_t N, int d);
int clt_fvu_fwd_bwd(const void* y, int y_dtype, const void* yhat, int yhat_dtype, const uint8_t* mask, int64_t N,
                    int d, float gscale, float* out_scalars, void* G, int g_dtype, void* workspace,
                    size_t workspace_bytes, clt_stream_t stream);

/* clt_fvu_fwd_bwd WITHOUT a caller workspace and without any allocation.  No counterpart in the reference: it is for
 * callers that cannot allocate -- code under HIP graph capture, tight serving / evaluation loops, harnesses that
 * forbid hipMalloc.  Same arguments, kernels and results (bit for bit) as clt_fvu_fwd_bwd; the reduction partials live
 * in a device array inside the library (about 16 MiB on every GPU the library runs on, held until the process ends).
 * Limit: d <= 8192 (any N); beyond it the call fails with CLT_ERR_UNSUPPORTED and clt_last_error() says so.
 * Calls no hipMalloc / hipFree / allocating library routine.  The first call on each device looks the array's
 * address up (hipGetSymbolAddress): make that call outside graph capture.  Every call on a GPU shares that GPU's
 * array: calls that overlap on one GPU (two streams, two threads) race -- serialise them on one stream.  Nothing in
 * the array has to start at zero (every partial is written before it is read within one call).               */
int clt_fvu_fwd_bwd_static(const void* y, int y_dtype, const void* yhat, int yhat_dtype, const uint8_t* mask, int64_t N,
                           int d, float gscale, float* out_scalars, void* G, int g_dtype, clt_stream_t stream);

/* The three stages of the FVU as separate entry points, for TOKEN-SHARDED targets (tensor parallel: each GPU
 * holds its own tokens of y / yhat after the reduce-scatter).  The caller all-reduces (SUM) colsum_count[d+1]
 * after stage 1 and l2tv[2] after stage 2 -- the scalar / [d] all-reduces DTensor generates for
 * sparse_coder.py:214-221.  colsum_count = (sum_n m*y per column, then sum_n m); l2tv = (l2, tv).   */
int clt_fvu_colsum(const void* y, int y_dtype, const uint8_t* mask, int64_t N, int d, float* colsum_count,
                   void* workspace, size_t workspace_bytes, clt_stream_t stream);
int clt_fvu_partial(const void* y, int y_dtype, const void* yhat, int yhat_dtype, const uint8_t* mask, int64_t N,
                    int d, const float* colsum_count, float* l2tv, void* workspace, size_t workspace_bytes,
                    clt_stream_t stream);
int clt_fvu_grad(const void* y, int y_dtype, const void* yhat, int yhat_dtype, const uint8_t* mask, int64_t N, int d,
                 const float* l2tv, float gscale, float* out_scalars, void* G, int g_dtype, clt_stream_t stream);

/* ------------------------------------------------------------------ Adam
 * torch.optim.Adam defaults (trainer.py:178-186: betas=(b1,b2), eps 1e-8, no weight decay,
 * no amsgrad) over a table of tensors in ONE launch.
 * Each clt_adam_tensor describes one parameter; `tensors` is a DEVICE array of n_tensors of
 * them; `chunk_tensor`/`chunk_start` are DEVICE arrays of n_chunks int32/int64 mapping each
 * workgroup to (tensor id, first element).  lr may differ per tensor (trainer.py:166-173); the
 * effective rate is tensor.lr * lr_factor (the LambdaLR factor of trainer.py:45-58).
 * Math in fp32, one rounding per stored value.  step is 1-based.                            */
typedef struct clt_adam_tensor {
  void* p;
  const void* g;
  void* m;
  void* v;
  int64_t numel;
  float lr;
  int32_t p_dtype; /* dtype of p, m and v */
  int32_t g_dtype;
  int32_t pad_;
} clt_adam_tensor;
int clt_adam_step(const clt_adam_tensor* tensors, int n_tensors, const int32_t* chunk_tensor,
                  const int64_t* chunk_start, int64_t n_chunks, int chunk_elems, float lr_factor, float b1,
                  float b2, float eps, int step, clt_stream_t stream);

/* SignSGD (sparsify/sign_sgd.py:19-23, the update under optimizer="signum"): p -= lr * lr_factor * sign(g) for
 * every tensor of the table (m and v of the records are ignored and may be NULL); sign(0) = 0.  */
int clt_signsgd_step(const clt_adam_tensor* tensors, int n_tensors, const int32_t* chunk_tensor,
                     const int64_t* chunk_start, int64_t n_chunks, int chunk_elems, float lr_factor,
                     clt_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CLT_MI355_H */
