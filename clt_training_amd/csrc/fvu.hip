// FVU loss forward + gradient for gfx950 (sparse_coder.py:207-221).
//   fvu = sum (m (y - yhat))^2 / sum m (y - ybar_m)^2,   G = gscale * (-2 / tv) * m * (y - yhat)
// Four small launches, all deterministic two-stage reductions (no atomics):
//   1 column sums of m*y per row block   2 mean[d]   3 per-block (l2, tv)   4 finalize + write G.
// HBM-bound, algorithmic bytes ~ N*d*(3*sizeof(y) + 2*sizeof(yhat) + sizeof(G)).
#include <mutex>

#include <type_traits>

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
  int64_t nb = (N + 15) / 16;
  if (nb > 512) nb = 512;
  if (nb < 1) nb = 1;
  return (int)nb;
}

__device__ __forceinline__ float ld(const void* p, int dt, int64_t i) { return load_scalar(p, dt, i); }

// 4 consecutive elements starting at element i (i % 4 == 0, 16-byte aligned base), dtype known at compile time.
// (A version with the dtype as a runtime argument was called from inside the loops below until round 3: with the runtime
// branch of ld4 inside an unrolled loop every load sat in a conditional block of its own and hipcc put `s_waitcnt vmcnt(0)`
// in front of each -- eight serialised memory round trips per unrolled iteration instead of eight loads in flight.)
template <bool BF>
__device__ __forceinline__ void ld4t(const void* p, int64_t i, float (&f)[4]) {
  if constexpr (BF) {
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(p) + i);
    f[0] = bf16_lo(u.x); f[1] = bf16_hi(u.x); f[2] = bf16_lo(u.y); f[3] = bf16_hi(u.y);
  } else {
    const float4 u = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + i);
    f[0] = u.x; f[1] = u.y; f[2] = u.z; f[3] = u.w;
  }
}
template <typename F>
__device__ __forceinline__ void by_dtype(int dt, F&& f) {
  if (dt == CLT_BF16) f(std::true_type{});
  else f(std::false_type{});
}

// grid = (row blocks, column blocks of 1024): four columns per thread, a row block's rows in order
__global__ __launch_bounds__(256) void fvu_colsum_kernel(const void* __restrict__ y, int y_dtype,
                                                         const uint8_t* __restrict__ mask, int64_t N, int d,
                                                         int64_t rpb, float* __restrict__ pcol,
                                                         float* __restrict__ pcnt) {
  const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = min(N, r0 + rpb);
  const int c = (blockIdx.y * 256 + threadIdx.x) * 4;
  if (c < d) {
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    by_dtype(y_dtype, [&](auto yb) {
#pragma unroll 8
      for (int64_t r = r0; r < r1; ++r) {
        const bool m = mask ? mask[r] != 0 : true;
        float v[4];
        ld4t<decltype(yb)::value>(y, r * d + c, v);
#pragma unroll
        for (int q = 0; q < 4; ++q) s[q] += m ? v[q] : 0.f;
      }
    });
    *reinterpret_cast<float4*>(pcol + (int64_t)blockIdx.x * d + c) = make_float4(s[0], s[1], s[2], s[3]);
  }
  if (blockIdx.y == 0 && threadIdx.x == 0) {
    float cnt = 0.f;
    for (int64_t r = r0; r < r1; ++r) cnt += mask ? (mask[r] ? 1.f : 0.f) : 1.f;
    pcnt[blockIdx.x] = cnt;
  }
}

__global__ void fvu_total_kernel(const float* __restrict__ psc, int NB, float* __restrict__ out2) {
  // one wave: lane-strided partial sums in a fixed order, then a fixed butterfly
  float l2 = 0.f, tv = 0.f;
  for (int b = threadIdx.x; b < NB; b += 64) {
    l2 += psc[2 * b];
    tv += psc[2 * b + 1];
  }
  l2 = wave_sum(l2);
  tv = wave_sum(tv);
  if (threadIdx.x == 0) {
    out2[0] = l2;
    out2[1] = tv;
  }
}

// grid = column blocks of 64: the 16 waves of a block take sixteenths of the row blocks (loads issued 8 at a time),
// then combine through LDS in a fixed order
__global__ __launch_bounds__(1024) void fvu_mean_kernel(const float* __restrict__ pcol, const float* __restrict__ pcnt,
                                                        int NB, int d, float* __restrict__ mean) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int per = (NB + 15) / 16;
  const int b0 = w * per, b1 = min(NB, b0 + per);
  float s = 0.f;
  if (c < d) {
#pragma unroll 8
    for (int b = b0; b < b1; ++b) s += pcol[(int64_t)b * d + c];
  }
  red[w][lane] = s;
  __syncthreads();
  if (w == 0 && c < d) {  // column SUM
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][lane];
    mean[c] = t;
  }
  if (blockIdx.x == 0 && w == 1) {  // the consumer divides by the (possibly all-reduced) count in mean[d]
    float cnt = 0.f;
    for (int b = lane; b < NB; b += 64) cnt += pcnt[b];
    cnt = wave_sum(cnt);
    if (lane == 0) mean[d] = cnt;
  }
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
  for (int c = threadIdx.x * 4; c < d; c += 1024) {
    const float4 mu4 = *reinterpret_cast<const float4*>(mean + c);
    const float mu[4] = {mu4.x * inv_cnt, mu4.y * inv_cnt, mu4.z * inv_cnt, mu4.w * inv_cnt};
    by_dtype(y_dtype, [&](auto yb) {
      by_dtype(yh_dtype, [&](auto hb) {
#pragma unroll 8
        for (int64_t r = r0; r < r1; ++r) {
          const bool m = mask ? mask[r] != 0 : true;
          float yy[4], yh[4];
          ld4t<decltype(yb)::value>(y, r * d + c, yy);
          ld4t<decltype(hb)::value>(yhat, r * d + c, yh);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float e = m ? yy[q] - yh[q] : 0.f;
            const float dv = m ? yy[q] - mu[q] : 0.f;
            l2 = fmaf(e, e, l2);
            tv = fmaf(dv, dv, tv);
          }
        }
      });
    });
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
  if (NB > 0) {  // the same lane-strided order as fvu_total_kernel, in every wave
    for (int b = threadIdx.x & 63; b < NB; b += 64) {
      l2 += psc[2 * b];
      tv += psc[2 * b + 1];
    }
    l2 = wave_sum(l2);
    tv = wave_sum(tv);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    out_scalars[0] = l2 / tv;
    out_scalars[1] = l2;
    out_scalars[2] = tv;
  }
  if (!G) return;
  const float coef = -2.f * gscale / tv;
  const int64_t total = N * (int64_t)d;
  by_dtype(y_dtype, [&](auto yb) {
    by_dtype(yh_dtype, [&](auto hb) {
      for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < total; i += (int64_t)gridDim.x * 1024) {
        const int64_t r = i / d;  // d % 4 == 0: the 4 elements share a row
        const float m = mask ? (mask[r] ? 1.f : 0.f) : 1.f;
        float g[4], yy[4], yh[4];
        ld4t<decltype(yb)::value>(y, i, yy);
        ld4t<decltype(hb)::value>(yhat, i, yh);
#pragma unroll
        for (int q = 0; q < 4; ++q) g[q] = coef * m * (yy[q] - yh[q]);
        if constexpr (sizeof(GT) == 2) {
          uint2 u;
          u.x = pack_bf16x2(g[0], g[1]);
          u.y = pack_bf16x2(g[2], g[3]);
          *reinterpret_cast<uint2*>(G + i) = u;
        } else {
          *reinterpret_cast<float4*>(G + i) = make_float4(g[0], g[1], g[2], g[3]);
        }
      }
    });
  });
}

}  // namespace clt

using namespace clt;

extern "C" size_t clt_fvu_workspace_bytes(int64_t N, int d) {
  if (N <= 0 || d <= 0) return 256;
  const size_t nb = (size_t)fvu_blocks(N);
  return (nb * (size_t)d + nb + (size_t)d + 1 + 2 * nb + 32) * sizeof(float);
}

// The launches of clt_fvu_fwd_bwd over a float workspace of clt_fvu_workspace_bytes(N, d) bytes (arguments checked).
static int fvu_fwd_bwd_launch(const char* who, const void* y, int y_dtype, const void* yhat, int yhat_dtype,
                              const uint8_t* mask, int64_t N, int d, float gscale, float* out_scalars, void* G,
                              int g_dtype, float* ws, hipStream_t st) {
  const int NB = fvu_blocks(N);
  const int64_t rpb = (N + NB - 1) / NB;
  float* pcol = ws;
  float* pcnt = pcol + (size_t)NB * d;
  float* mean = pcnt + (NB + 3) / 4 * 4;  // float4 loads of the column sums: keep 16-byte alignment
  float* psc = mean + d + 4;
  hipLaunchKernelGGL(fvu_colsum_kernel, dim3(NB, (d + 1023) / 1024), dim3(256), 0, st, y, y_dtype, mask, N, d, rpb, pcol, pcnt);
  hipLaunchKernelGGL(fvu_mean_kernel, dim3((d + 63) / 64), dim3(1024), 0, st, (const float*)pcol, (const float*)pcnt, NB,
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
  CLT_CHECK_LAUNCH(who);
  return CLT_OK;
}

static int fvu_check_args(const char* who, const void* y, int y_dtype, const void* yhat, int yhat_dtype, int64_t N, int d,
                          const float* out_scalars, const void* G, int g_dtype) {
  if (!(y && yhat && out_scalars)) { set_error("%s: null pointer", who); return CLT_ERR_BAD_ARG; }
  if (!(N > 0 && d > 0)) { set_error("%s: bad sizes", who); return CLT_ERR_BAD_ARG; }
  if (!(valid_dtype(y_dtype) && valid_dtype(yhat_dtype) && (!G || valid_dtype(g_dtype)))) {
    set_error("%s: bad dtype", who);
    return CLT_ERR_BAD_ARG;
  }
  if (d % 4 != 0) { set_error("%s: d=%d must be a multiple of 4", who, d); return CLT_ERR_UNSUPPORTED; }
  if (G && !aligned16(G)) { set_error("%s: G must be 16-byte aligned", who); return CLT_ERR_BAD_ARG; }
  return CLT_OK;
}

extern "C" int clt_fvu_fwd_bwd(const void* y, int y_dtype, const void* yhat, int yhat_dtype, const uint8_t* mask,
                               int64_t N, int d, float gscale, float* out_scalars, void* G, int g_dtype,
                               void* workspace, size_t workspace_bytes, clt_stream_t stream) {
  const int rc = fvu_check_args("clt_fvu_fwd_bwd", y, y_dtype, yhat, yhat_dtype, N, d, out_scalars, G, g_dtype);
  if (rc != CLT_OK) return rc;
  CLT_CHECK_ARG(workspace, "clt_fvu_fwd_bwd: null pointer");
  if (workspace_bytes < clt_fvu_workspace_bytes(N, d)) {
    set_error("clt_fvu_fwd_bwd: workspace too small");
    return CLT_ERR_WORKSPACE;
  }
  return fvu_fwd_bwd_launch("clt_fvu_fwd_bwd", y, y_dtype, yhat, yhat_dtype, mask, N, d, gscale, out_scalars, G, g_dtype,
                            reinterpret_cast<float*>(workspace), (hipStream_t)stream);
}

// ---- clt_fvu_fwd_bwd for callers that cannot allocate (HIP graph capture, serving loops, harnesses that forbid
//      hipMalloc): the reduction partials live in a file-scope device array instead of a caller workspace.  Each GPU
//      the library runs on holds its own copy of the array (at its own address) until the process ends.  In host
//      code the array's NAME is only a stand-in, not a device address: the per-device address comes from
//      hipGetSymbolAddress, once per device, and only that pointer is handed to the kernels.
constexpr int FVU_STATIC_MAX_D = 8192;
constexpr size_t FVU_SCRATCH_BYTES = 16 * 1024 * 1024 + 64 * 1024;  // >= clt_fvu_workspace_bytes(any N, 8192)
static __device__ __align__(256) unsigned char scratch[FVU_SCRATCH_BYTES];

constexpr int FVU_MAX_DEVICES = 64;
static void* g_scratch_ptr[FVU_MAX_DEVICES];  // per device; written once under the mutex, read-only afterwards
static std::mutex g_scratch_mutex;

static int fvu_scratch_for_current_device(float** out) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= FVU_MAX_DEVICES) {
    set_error("clt_fvu_fwd_bwd_static: no current HIP device (or device index >= %d)", FVU_MAX_DEVICES);
    return CLT_ERR_LAUNCH;
  }
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[dev]) {
    void* ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));  // the address on THIS device
    if (e != hipSuccess || !ptr) {
      set_error("clt_fvu_fwd_bwd_static: hipGetSymbolAddress failed: %s", hipGetErrorString(e));
      return CLT_ERR_LAUNCH;
    }
    g_scratch_ptr[dev] = ptr;
  }
  *out = reinterpret_cast<float*>(g_scratch_ptr[dev]);
  return CLT_OK;
}

extern "C" int clt_fvu_fwd_bwd_static(const void* y, int y_dtype, const void* yhat, int yhat_dtype, const uint8_t* mask,
                                      int64_t N, int d, float gscale, float* out_scalars, void* G, int g_dtype,
                                      clt_stream_t stream) {
  const char* who = "clt_fvu_fwd_bwd_static";
  const int rc = fvu_check_args(who, y, y_dtype, yhat, yhat_dtype, N, d, out_scalars, G, g_dtype);
  if (rc != CLT_OK) return rc;
  if (d > FVU_STATIC_MAX_D || clt_fvu_workspace_bytes(N, d) > FVU_SCRATCH_BYTES) {
    set_error("%s: d=%d exceeds the static scratch (d <= %d); use clt_fvu_fwd_bwd with a caller workspace", who, d,
              FVU_STATIC_MAX_D);
    return CLT_ERR_UNSUPPORTED;
  }
  float* ws = nullptr;
  const int rs = fvu_scratch_for_current_device(&ws);
  if (rs != CLT_OK) return rs;
  // every partial the kernels read is written earlier in the same call: nothing in the scratch has to start at zero
  return fvu_fwd_bwd_launch(who, y, y_dtype, yhat, yhat_dtype, mask, N, d, gscale, out_scalars, G, g_dtype, ws,
                            (hipStream_t)stream);
}

// ---- the same three stages as separate entry points, for token-sharded (multi-GPU) FVU: the caller all-reduces
//      colsum_count[d+1] after stage 1 and l2tv[2] after stage 2 (sparse_coder.py:214-221 under DTensor).
extern "C" int clt_fvu_colsum(const void* y, int y_dtype, const uint8_t* mask, int64_t N, int d, float* colsum_count,
                              void* workspace, size_t workspace_bytes, clt_stream_t stream) {
  CLT_CHECK_ARG(y && colsum_count && workspace && N > 0 && d > 0 && valid_dtype(y_dtype), "clt_fvu_colsum: bad argument");
  CLT_UNSUPPORTED(d % 4 == 0, "clt_fvu_colsum: d=%d must be a multiple of 4", d);
  if (workspace_bytes < clt_fvu_workspace_bytes(N, d)) {
    set_error("clt_fvu_colsum: workspace too small");
    return CLT_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int NB = fvu_blocks(N);
  const int64_t rpb = (N + NB - 1) / NB;
  float* pcol = reinterpret_cast<float*>(workspace);
  float* pcnt = pcol + (size_t)NB * d;
  hipLaunchKernelGGL(fvu_colsum_kernel, dim3(NB, (d + 1023) / 1024), dim3(256), 0, st, y, y_dtype, mask, N, d, rpb, pcol, pcnt);
  hipLaunchKernelGGL(fvu_mean_kernel, dim3((d + 63) / 64), dim3(1024), 0, st, (const float*)pcol, (const float*)pcnt, NB,
                     d, colsum_count);
  CLT_CHECK_LAUNCH("clt_fvu_colsum");
  return CLT_OK;
}

extern "C" int clt_fvu_partial(const void* y, int y_dtype, const void* yhat, int yhat_dtype, const uint8_t* mask,
                               int64_t N, int d, const float* colsum_count, float* l2tv, void* workspace,
                               size_t workspace_bytes, clt_stream_t stream) {
  CLT_CHECK_ARG(y && yhat && colsum_count && l2tv && workspace && N > 0 && d > 0, "clt_fvu_partial: bad argument");
  CLT_CHECK_ARG(valid_dtype(y_dtype) && valid_dtype(yhat_dtype), "clt_fvu_partial: bad dtype");
  CLT_UNSUPPORTED(d % 4 == 0, "clt_fvu_partial: d=%d must be a multiple of 4", d);
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
