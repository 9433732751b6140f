// Multi-tensor Adam for gfx950: torch.optim.Adam defaults (trainer.py:178-186) over a table of
// tensors in ONE launch.  One workgroup per chunk of `chunk_elems` elements; 16 B per lane.
// HBM-bound: algorithmic bytes per parameter = 3 reads (p, m, v) + 1 grad read + 3 writes.
#include <stdlib.h>

#include "common.h"

namespace clt {

template <typename PT, typename GT, int VAR>
__device__ __forceinline__ void adam_chunk(PT* __restrict__ p, const GT* __restrict__ g, PT* __restrict__ m,
                                           PT* __restrict__ v, int64_t n, float step_size, float b1, float b2,
                                           float inv_sqrt_bc2, float eps) {
  constexpr int V = 8;  // elements per thread per pass
  const int64_t nvec = n / V;
#pragma unroll(VAR == 2 ? 1 : 4)
  for (int64_t i = threadIdx.x; i < nvec; i += 256) {
    float pf[V], gf[V], mf[V], vf[V];
    if constexpr (sizeof(PT) == 2 && VAR >= 1) {
      nt_load8(p + i * V, pf);
      nt_load8(m + i * V, mf);
      nt_load8(v + i * V, vf);
    } else if constexpr (sizeof(PT) == 2) {
      Chunk<bf16_t>::load(p + i * V, pf);
      Chunk<bf16_t>::load(m + i * V, mf);
      Chunk<bf16_t>::load(v + i * V, vf);
    } else {
      float a[4], b[4];
      Chunk<float>::load(p + i * V, a); Chunk<float>::load(p + i * V + 4, b);
#pragma unroll
      for (int q = 0; q < 4; ++q) pf[q] = a[q], pf[q + 4] = b[q];
      Chunk<float>::load(m + i * V, a); Chunk<float>::load(m + i * V + 4, b);
#pragma unroll
      for (int q = 0; q < 4; ++q) mf[q] = a[q], mf[q + 4] = b[q];
      Chunk<float>::load(v + i * V, a); Chunk<float>::load(v + i * V + 4, b);
#pragma unroll
      for (int q = 0; q < 4; ++q) vf[q] = a[q], vf[q + 4] = b[q];
    }
    if constexpr (sizeof(GT) == 2 && VAR >= 1) {
      nt_load8(g + i * V, gf);
    } else if constexpr (sizeof(GT) == 2) {
      Chunk<bf16_t>::load(g + i * V, gf);
    } else {
      float a[4], b[4];
      Chunk<float>::load(g + i * V, a); Chunk<float>::load(g + i * V + 4, b);
#pragma unroll
      for (int q = 0; q < 4; ++q) gf[q] = a[q], gf[q + 4] = b[q];
    }
#pragma unroll
    for (int q = 0; q < V; ++q) {
      adam_update(pf[q], mf[q], vf[q], gf[q], step_size, b1, b2, inv_sqrt_bc2, eps);
    }
    if constexpr (sizeof(PT) == 2 && VAR >= 1) {
      nt_store8(p + i * V, pf);
      nt_store8(m + i * V, mf);
      nt_store8(v + i * V, vf);
    } else if constexpr (sizeof(PT) == 2) {
      Chunk<bf16_t>::store(p + i * V, pf);
      Chunk<bf16_t>::store(m + i * V, mf);
      Chunk<bf16_t>::store(v + i * V, vf);
    } else {
      *reinterpret_cast<float4*>(p + i * V) = make_float4(pf[0], pf[1], pf[2], pf[3]);
      *reinterpret_cast<float4*>(p + i * V + 4) = make_float4(pf[4], pf[5], pf[6], pf[7]);
      *reinterpret_cast<float4*>(m + i * V) = make_float4(mf[0], mf[1], mf[2], mf[3]);
      *reinterpret_cast<float4*>(m + i * V + 4) = make_float4(mf[4], mf[5], mf[6], mf[7]);
      *reinterpret_cast<float4*>(v + i * V) = make_float4(vf[0], vf[1], vf[2], vf[3]);
      *reinterpret_cast<float4*>(v + i * V + 4) = make_float4(vf[4], vf[5], vf[6], vf[7]);
    }
  }
  // scalar tail
  for (int64_t i = nvec * V + threadIdx.x; i < n; i += 256) {
    float pf = (float)p[i], gf = (float)g[i], mf = (float)m[i], vf = (float)v[i];
    adam_update(pf, mf, vf, gf, step_size, b1, b2, inv_sqrt_bc2, eps);
    p[i] = (PT)pf;
    m[i] = (PT)mf;
    v[i] = (PT)vf;
  }
}

template <int VAR>
__global__ __launch_bounds__(256) void adam_kernel(const clt_adam_tensor* __restrict__ tensors,
                                                   const int32_t* __restrict__ chunk_tensor,
                                                   const int64_t* __restrict__ chunk_start, int chunk_elems,
                                                   float lr_factor, float b1, float b2, float eps, float inv_bc1,
                                                   float inv_sqrt_bc2) {
  const clt_adam_tensor t = tensors[chunk_tensor[blockIdx.x]];
  const int64_t start = chunk_start[blockIdx.x];
  const int64_t n = min((int64_t)chunk_elems, t.numel - start);
  if (n <= 0) return;
  const float step_size = t.lr * lr_factor * inv_bc1;
  if (t.p_dtype == CLT_BF16) {
    bf16_t* p = (bf16_t*)t.p + start;
    bf16_t* m = (bf16_t*)t.m + start;
    bf16_t* v = (bf16_t*)t.v + start;
    if (t.g_dtype == CLT_BF16)
      adam_chunk<bf16_t, bf16_t, VAR>(p, (const bf16_t*)t.g + start, m, v, n, step_size, b1, b2, inv_sqrt_bc2, eps);
    else
      adam_chunk<bf16_t, float, VAR>(p, (const float*)t.g + start, m, v, n, step_size, b1, b2, inv_sqrt_bc2, eps);
  } else {
    float* p = (float*)t.p + start;
    float* m = (float*)t.m + start;
    float* v = (float*)t.v + start;
    if (t.g_dtype == CLT_BF16)
      adam_chunk<float, bf16_t, VAR>(p, (const bf16_t*)t.g + start, m, v, n, step_size, b1, b2, inv_sqrt_bc2, eps);
    else
      adam_chunk<float, float, VAR>(p, (const float*)t.g + start, m, v, n, step_size, b1, b2, inv_sqrt_bc2, eps);
  }
}

// SignSGD (sparsify/sign_sgd.py:19-23): p -= lr * sign(g); sign(0) = 0, so a parameter whose gradient is zero keeps its
// bits.  HBM-bound: one read of p and g, one write of p.
template <typename PT, typename GT>
__device__ __forceinline__ void signsgd_chunk(PT* __restrict__ p, const GT* __restrict__ g, int64_t n, float lr) {
  constexpr int V = 8;
  const int64_t nvec = n / V;
#pragma unroll 2
  for (int64_t i = threadIdx.x; i < nvec; i += 256) {
    float pf[V], gf[V];
    if constexpr (sizeof(PT) == 2) {
      nt_load8(p + i * V, pf);
    } else {
      float a[4], b[4];
      Chunk<float>::load(p + i * V, a); Chunk<float>::load(p + i * V + 4, b);
#pragma unroll
      for (int q = 0; q < 4; ++q) pf[q] = a[q], pf[q + 4] = b[q];
    }
    if constexpr (sizeof(GT) == 2) {
      nt_load8(g + i * V, gf);
    } else {
      float a[4], b[4];
      Chunk<float>::load(g + i * V, a); Chunk<float>::load(g + i * V + 4, b);
#pragma unroll
      for (int q = 0; q < 4; ++q) gf[q] = a[q], gf[q + 4] = b[q];
    }
#pragma unroll
    for (int q = 0; q < V; ++q) {
      const float sgn = gf[q] > 0.f ? 1.f : (gf[q] < 0.f ? -1.f : gf[q]);  // +-0 -> 0 (and a NaN gradient stays NaN)
      pf[q] = fmaf(-lr, sgn, pf[q]);                                       // Tensor.add_(sign, alpha=-lr)
    }
    if constexpr (sizeof(PT) == 2) {
      nt_store8(p + i * V, pf);
    } else {
      *reinterpret_cast<float4*>(p + i * V) = make_float4(pf[0], pf[1], pf[2], pf[3]);
      *reinterpret_cast<float4*>(p + i * V + 4) = make_float4(pf[4], pf[5], pf[6], pf[7]);
    }
  }
  for (int64_t i = nvec * V + threadIdx.x; i < n; i += 256) {
    const float gf = (float)g[i];
    const float sgn = gf > 0.f ? 1.f : (gf < 0.f ? -1.f : gf);
    p[i] = (PT)fmaf(-lr, sgn, (float)p[i]);
  }
}

__global__ __launch_bounds__(256) void signsgd_kernel(const clt_adam_tensor* __restrict__ tensors,
                                                      const int32_t* __restrict__ chunk_tensor,
                                                      const int64_t* __restrict__ chunk_start, int chunk_elems,
                                                      float lr_factor) {
  const clt_adam_tensor t = tensors[chunk_tensor[blockIdx.x]];
  const int64_t start = chunk_start[blockIdx.x];
  const int64_t n = min((int64_t)chunk_elems, t.numel - start);
  if (n <= 0) return;
  const float lr = t.lr * lr_factor;
  if (t.p_dtype == CLT_BF16) {
    if (t.g_dtype == CLT_BF16) signsgd_chunk((bf16_t*)t.p + start, (const bf16_t*)t.g + start, n, lr);
    else signsgd_chunk((bf16_t*)t.p + start, (const float*)t.g + start, n, lr);
  } else {
    if (t.g_dtype == CLT_BF16) signsgd_chunk((float*)t.p + start, (const bf16_t*)t.g + start, n, lr);
    else signsgd_chunk((float*)t.p + start, (const float*)t.g + start, n, lr);
  }
}

}  // namespace clt

using namespace clt;

extern "C" int clt_signsgd_step(const clt_adam_tensor* tensors, int n_tensors, const int32_t* chunk_tensor,
                                const int64_t* chunk_start, int64_t n_chunks, int chunk_elems, float lr_factor,
                                clt_stream_t stream) {
  CLT_CHECK_ARG(tensors && chunk_tensor && chunk_start, "clt_signsgd_step: null pointer");
  CLT_CHECK_ARG(n_tensors > 0 && n_chunks > 0 && n_chunks < (1ll << 31), "clt_signsgd_step: bad sizes");
  CLT_CHECK_ARG(chunk_elems > 0 && chunk_elems % 8 == 0, "clt_signsgd_step: chunk_elems must be a positive multiple of 8");
  hipLaunchKernelGGL(signsgd_kernel, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)stream, tensors, chunk_tensor,
                     chunk_start, chunk_elems, lr_factor);
  CLT_CHECK_LAUNCH("clt_signsgd_step");
  return CLT_OK;
}

extern "C" int clt_adam_step(const clt_adam_tensor* tensors, int n_tensors, const int32_t* chunk_tensor,
                             const int64_t* chunk_start, int64_t n_chunks, int chunk_elems, float lr_factor, float b1,
                             float b2, float eps, int step, clt_stream_t stream) {
  CLT_CHECK_ARG(tensors && chunk_tensor && chunk_start, "clt_adam_step: null pointer");
  CLT_CHECK_ARG(n_tensors > 0 && n_chunks > 0 && n_chunks < (1ll << 31) && step >= 1, "clt_adam_step: bad sizes");
  CLT_CHECK_ARG(chunk_elems > 0 && chunk_elems % 8 == 0, "clt_adam_step: chunk_elems must be a positive multiple of 8");
  const double bc1 = 1.0 - pow((double)b1, (double)step);
  const double bc2 = 1.0 - pow((double)b2, (double)step);
  const int var = tuning_env("CLT_ADAM_VARIANT", 1);  // -DCLT_TUNING builds only: 0 = cached loads/stores, 2 = no unrolling
#define CLT_ADAM_LAUNCH(VARIANT)                                                                                    \
  hipLaunchKernelGGL(adam_kernel<VARIANT>, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)stream, tensors,    \
                     chunk_tensor, chunk_start, chunk_elems, lr_factor, b1, b2, eps, (float)(1.0 / bc1),            \
                     (float)(1.0 / sqrt(bc2)))
  if (var == 0) CLT_ADAM_LAUNCH(0);
  else if (var == 2) CLT_ADAM_LAUNCH(2);
  else CLT_ADAM_LAUNCH(1);
  CLT_CHECK_LAUNCH("clt_adam_step");
  return CLT_OK;
}
