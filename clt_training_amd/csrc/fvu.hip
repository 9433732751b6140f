// FVU loss forward + gradient for gfx950 (sparse_coder.py:207-221).
//   fvu = sum (m (y - yhat))^2 / sum m (y - ybar_m)^2,   G = gscale * (-2 / tv) * m * (y - yhat)
// Four small launches, all deterministic two-stage reductions (no atomics):
//   1 column sums of m*y per row block   2 mean[d]   3 per-block (l2, tv)   4 finalize + write G.
// HBM-bound, algorithmic bytes ~ N*d*(3*sizeof(y) + 2*sizeof(yhat) + sizeof(G)).
#include "common.h"

namespace clt {

struct FvuWs {
  float* pcol;   // [NB][d]
  float* pcnt;   // [NB]
  float* mean;   // [d] then count at mean[d]
  float* psc;    // [NB][2]
  int NB;
  int64_t rpb;
};

static int fvu_blocks(int64_t N) {
  int64_t nb = (N + 63) / 64;
  if (nb > 128) nb = 128;
  if (nb < 1) nb = 1;
  return (int)nb;
}

__device__ __forceinline__ float ld(const void* p, int dt, int64_t i) { return load_scalar(p, dt, i); }

__global__ __launch_bounds__(256) void fvu_colsum_kernel(const void* __restrict__ y, int y_dtype,
                                                         const uint8_t* __restrict__ mask, int64_t N, int d,
                                                         int64_t rpb, float* __restrict__ pcol,
                                                         float* __restrict__ pcnt) {
  // grid = (row blocks, column blocks): one column per thread, rows of a block read coalesced across threads
  const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = min(N, r0 + rpb);
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c < d) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int64_t r = r0;
    for (; r + 3 < r1; r += 4) {
      const float m0 = mask ? (mask[r] ? 1.f : 0.f) : 1.f, m1 = mask ? (mask[r + 1] ? 1.f : 0.f) : 1.f;
      const float m2 = mask ? (mask[r + 2] ? 1.f : 0.f) : 1.f, m3 = mask ? (mask[r + 3] ? 1.f : 0.f) : 1.f;
      s0 += m0 * ld(y, y_dtype, r * d + c);
      s1 += m1 * ld(y, y_dtype, (r + 1) * d + c);
      s2 += m2 * ld(y, y_dtype, (r + 2) * d + c);
      s3 += m3 * ld(y, y_dtype, (r + 3) * d + c);
    }
    for (; r < r1; ++r) s0 += (mask ? (mask[r] ? 1.f : 0.f) : 1.f) * ld(y, y_dtype, r * d + c);
    pcol[(int64_t)blockIdx.x * d + c] = (s0 + s1) + (s2 + s3);
  }
  if (blockIdx.y == 0 && threadIdx.x == 0) {
    float cnt = 0.f;
    for (int64_t r = r0; r < r1; ++r) cnt += mask ? (mask[r] ? 1.f : 0.f) : 1.f;
    pcnt[blockIdx.x] = cnt;
  }
}

__global__ void fvu_total_kernel(const float* __restrict__ psc, int NB, float* __restrict__ out2) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float l2 = 0.f, tv = 0.f;
    for (int b = 0; b < NB; ++b) {
      l2 += psc[2 * b];
      tv += psc[2 * b + 1];
    }
    out2[0] = l2;
    out2[1] = tv;
  }
}

__global__ void fvu_mean_kernel(const float* __restrict__ pcol, const float* __restrict__ pcnt, int NB, int d,
                                float* __restrict__ mean) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  float cnt = 0.f;
  for (int b = 0; b < NB; ++b) cnt += pcnt[b];
  if (c < d) {
    float s = 0.f;
    for (int b = 0; b < NB; ++b) s += pcol[(int64_t)b * d + c];
    mean[c] = s;  // column SUM; the consumer divides by the (possibly all-reduced) count in mean[d]
  }
  if (c == 0) mean[d] = cnt;
}

__global__ __launch_bounds__(256) void fvu_partial_kernel(const void* __restrict__ y, int y_dtype,
                                                          const void* __restrict__ yhat, int yh_dtype,
                                                          const uint8_t* __restrict__ mask, int64_t N, int d,
                                                          int64_t rpb, const float* __restrict__ mean,
                                                          float* __restrict__ psc) {
  __shared__ float red[2][4];
  const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = min(N, r0 + rpb);
  const float inv_cnt = 1.f / mean[d];
  float l2 = 0.f, tv = 0.f;
  for (int64_t r = r0; r < r1; ++r) {
    const float m = mask ? (mask[r] ? 1.f : 0.f) : 1.f;
    if (m == 0.f) continue;
    for (int c = threadIdx.x; c < d; c += 256) {
      const float yy = ld(y, y_dtype, r * d + c);
      const float e = yy - ld(yhat, yh_dtype, r * d + c);
      const float dv = yy - mean[c] * inv_cnt;
      l2 = fmaf(e, e, l2);
      tv = fmaf(dv, dv, tv);
    }
  }
  l2 = wave_sum(l2);
  tv = wave_sum(tv);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[0][w] = l2;
    red[1][w] = tv;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    psc[2 * blockIdx.x + 0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    psc[2 * blockIdx.x + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

template <typename GT>
__global__ __launch_bounds__(256) void fvu_grad_kernel(const void* __restrict__ y, int y_dtype,
                                                       const void* __restrict__ yhat, int yh_dtype,
                                                       const uint8_t* __restrict__ mask, int64_t N, int d,
                                                       const float* __restrict__ psc, int NB, float gscale,
                                                       float* __restrict__ out_scalars, GT* __restrict__ G) {
  // every block re-derives the totals in the same fixed order (NB == 0: psc already holds the totals)
  float l2 = 0.f, tv = 0.f;
  if (NB == 0) {
    l2 = psc[0];
    tv = psc[1];
  }
  for (int b = 0; b < NB; ++b) {
    l2 += psc[2 * b];
    tv += psc[2 * b + 1];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    out_scalars[0] = l2 / tv;
    out_scalars[1] = l2;
    out_scalars[2] = tv;
  }
  if (!G) return;
  const float coef = -2.f * gscale / tv;
  const int64_t total = N * (int64_t)d;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < total; i += (int64_t)gridDim.x * 1024) {
    const int64_t r = i / d;  // d % 4 == 0: the 4 elements share a row
    const float m = mask ? (mask[r] ? 1.f : 0.f) : 1.f;
    float g[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) g[q] = coef * m * (ld(y, y_dtype, i + q) - ld(yhat, yh_dtype, i + q));
    if constexpr (sizeof(GT) == 2) {
      uint2 u;
      u.x = pack_bf16x2(g[0], g[1]);
      u.y = pack_bf16x2(g[2], g[3]);
      *reinterpret_cast<uint2*>(G + i) = u;
    } else {
      *reinterpret_cast<float4*>(G + i) = make_float4(g[0], g[1], g[2], g[3]);
    }
  }
}

}  // namespace clt

using namespace clt;

extern "C" size_t clt_fvu_workspace_bytes(int64_t N, int d) {
  if (N <= 0 || d <= 0) return 256;
  const size_t nb = (size_t)fvu_blocks(N);
  return (nb * (size_t)d + nb + (size_t)d + 1 + 2 * nb + 16) * sizeof(float);
}

extern "C" int clt_fvu_fwd_bwd(const void* y, int y_dtype, const void* yhat, int yhat_dtype, const uint8_t* mask,
                               int64_t N, int d, float gscale, float* out_scalars, void* G, int g_dtype,
                               void* workspace, size_t workspace_bytes, clt_stream_t stream) {
  CLT_CHECK_ARG(y && yhat && out_scalars && workspace, "clt_fvu_fwd_bwd: null pointer");
  CLT_CHECK_ARG(N > 0 && d > 0, "clt_fvu_fwd_bwd: bad sizes");
  CLT_CHECK_ARG(valid_dtype(y_dtype) && valid_dtype(yhat_dtype) && (!G || valid_dtype(g_dtype)), "clt_fvu_fwd_bwd: bad dtype");
  CLT_UNSUPPORTED(d % 4 == 0, "clt_fvu_fwd_bwd: d=%d must be a multiple of 4", d);
  CLT_CHECK_ARG(!G || aligned16(G), "clt_fvu_fwd_bwd: G must be 16-byte aligned");
  if (workspace_bytes < clt_fvu_workspace_bytes(N, d)) {
    set_error("clt_fvu_fwd_bwd: workspace too small");
    return CLT_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int NB = fvu_blocks(N);
  const int64_t rpb = (N + NB - 1) / NB;
  float* ws = reinterpret_cast<float*>(workspace);
  float* pcol = ws;
  float* pcnt = pcol + (size_t)NB * d;
  float* mean = pcnt + NB;
  float* psc = mean + d + 1;
  hipLaunchKernelGGL(fvu_colsum_kernel, dim3(NB, (d + 255) / 256), dim3(256), 0, st, y, y_dtype, mask, N, d, rpb, pcol, pcnt);
  hipLaunchKernelGGL(fvu_mean_kernel, dim3((d + 255) / 256), dim3(256), 0, st, (const float*)pcol, (const float*)pcnt, NB,
                     d, mean);
  hipLaunchKernelGGL(fvu_partial_kernel, dim3(NB), dim3(256), 0, st, y, y_dtype, yhat, yhat_dtype, mask, N, d, rpb,
                     (const float*)mean, psc);
  const int64_t total = N * (int64_t)d;
  int64_t gb64 = (total + 1023) / 1024;
  int gb = gb64 > 2048 ? 2048 : (int)gb64;
  if (g_dtype == CLT_BF16)
    hipLaunchKernelGGL((fvu_grad_kernel<bf16_t>), dim3(gb), dim3(256), 0, st, y, y_dtype, yhat, yhat_dtype, mask, N, d,
                       (const float*)psc, NB, gscale, out_scalars, (bf16_t*)G);
  else
    hipLaunchKernelGGL((fvu_grad_kernel<float>), dim3(gb), dim3(256), 0, st, y, y_dtype, yhat, yhat_dtype, mask, N, d,
                       (const float*)psc, NB, gscale, out_scalars, (float*)G);
  CLT_CHECK_LAUNCH("clt_fvu_fwd_bwd");
  return CLT_OK;
}

// ---- the same three stages as separate entry points, for token-sharded (multi-GPU) FVU: the caller all-reduces
//      colsum_count[d+1] after stage 1 and l2tv[2] after stage 2 (sparse_coder.py:214-221 under DTensor).
extern "C" int clt_fvu_colsum(const void* y, int y_dtype, const uint8_t* mask, int64_t N, int d, float* colsum_count,
                              void* workspace, size_t workspace_bytes, clt_stream_t stream) {
  CLT_CHECK_ARG(y && colsum_count && workspace && N > 0 && d > 0 && valid_dtype(y_dtype), "clt_fvu_colsum: bad argument");
  if (workspace_bytes < clt_fvu_workspace_bytes(N, d)) {
    set_error("clt_fvu_colsum: workspace too small");
    return CLT_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int NB = fvu_blocks(N);
  const int64_t rpb = (N + NB - 1) / NB;
  float* pcol = reinterpret_cast<float*>(workspace);
  float* pcnt = pcol + (size_t)NB * d;
  hipLaunchKernelGGL(fvu_colsum_kernel, dim3(NB, (d + 255) / 256), dim3(256), 0, st, y, y_dtype, mask, N, d, rpb, pcol, pcnt);
  hipLaunchKernelGGL(fvu_mean_kernel, dim3((d + 255) / 256), dim3(256), 0, st, (const float*)pcol, (const float*)pcnt, NB,
                     d, colsum_count);
  CLT_CHECK_LAUNCH("clt_fvu_colsum");
  return CLT_OK;
}

extern "C" int clt_fvu_partial(const void* y, int y_dtype, const void* yhat, int yhat_dtype, const uint8_t* mask,
                               int64_t N, int d, const float* colsum_count, float* l2tv, void* workspace,
                               size_t workspace_bytes, clt_stream_t stream) {
  CLT_CHECK_ARG(y && yhat && colsum_count && l2tv && workspace && N > 0 && d > 0, "clt_fvu_partial: bad argument");
  CLT_CHECK_ARG(valid_dtype(y_dtype) && valid_dtype(yhat_dtype), "clt_fvu_partial: bad dtype");
  if (workspace_bytes < clt_fvu_workspace_bytes(N, d)) {
    set_error("clt_fvu_partial: workspace too small");
    return CLT_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int NB = fvu_blocks(N);
  const int64_t rpb = (N + NB - 1) / NB;
  float* psc = reinterpret_cast<float*>(workspace);
  hipLaunchKernelGGL(fvu_partial_kernel, dim3(NB), dim3(256), 0, st, y, y_dtype, yhat, yhat_dtype, mask, N, d, rpb,
                     colsum_count, psc);
  hipLaunchKernelGGL(fvu_total_kernel, dim3(1), dim3(64), 0, st, (const float*)psc, NB, l2tv);
  CLT_CHECK_LAUNCH("clt_fvu_partial");
  return CLT_OK;
}

extern "C" int clt_fvu_grad(const void* y, int y_dtype, const void* yhat, int yhat_dtype, const uint8_t* mask, int64_t N,
                            int d, const float* l2tv, float gscale, float* out_scalars, void* G, int g_dtype,
                            clt_stream_t stream) {
  CLT_CHECK_ARG(y && yhat && l2tv && out_scalars && G && N > 0 && d > 0, "clt_fvu_grad: bad argument");
  CLT_CHECK_ARG(valid_dtype(y_dtype) && valid_dtype(yhat_dtype) && valid_dtype(g_dtype), "clt_fvu_grad: bad dtype");
  CLT_UNSUPPORTED(d % 4 == 0, "clt_fvu_grad: d=%d must be a multiple of 4", d);
  CLT_CHECK_ARG(aligned16(G), "clt_fvu_grad: G must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = N * (int64_t)d;
  int64_t gb64 = (total + 1023) / 1024;
  int gb = gb64 > 2048 ? 2048 : (int)gb64;
  if (g_dtype == CLT_BF16)
    hipLaunchKernelGGL((fvu_grad_kernel<bf16_t>), dim3(gb), dim3(256), 0, st, y, y_dtype, yhat, yhat_dtype, mask, N, d,
                       l2tv, 0, gscale, out_scalars, (bf16_t*)G);
  else
    hipLaunchKernelGGL((fvu_grad_kernel<float>), dim3(gb), dim3(256), 0, st, y, y_dtype, yhat, yhat_dtype, mask, N, d,
                       l2tv, 0, gscale, out_scalars, (float*)G);
  CLT_CHECK_LAUNCH("clt_fvu_grad");
  return CLT_OK;
}
