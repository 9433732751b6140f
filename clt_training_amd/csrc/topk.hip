// Row top-k for gfx950: wave64 streaming select with a register-resident bitonic sorter.
//
// Each wave streams its share of a row (16-byte loads, 8 bf16 / 4 f32 per lane) and
// compares every value against the current k-th best.  Survivors (rare once the
// threshold has warmed up: ~k*ln(L/k) per row) are appended to a small per-wave LDS
// queue; when the queue fills it is merged into the wave's sorted top list by one
// bitonic sort over 64*R 64-bit keys held R per lane (cross-lane steps by ds_bpermute,
// steps >= 64 apart are register-to-register).  Keys are (value_bits << 32) | ~index,
// so the u64 order IS the tie rule: larger value first, then lowest index.  Four waves
// share a long row and merge through LDS; short rows get one wave each.
// The kernel reads the row exactly once: HBM-bound, algorithmic bytes = L*sizeof(val) per row.
#include "topk_select.h"

namespace clt {

// WPR waves cooperate on one row (WPR = 4: one row per 256-thread block; WPR = 1: four rows per block).
template <typename VT, int S, int R, int WPR, bool VEC>
__global__ __launch_bounds__(256) void topk_rows_kernel(const VT* __restrict__ vals, int64_t ld,
                                                        const int64_t* __restrict__ cand_idx,
                                                        const int32_t* __restrict__ counts, int64_t N, int64_t L,
                                                        int k, void* __restrict__ out_vals, int out_dtype,
                                                        int64_t* __restrict__ out_idx, const int* __restrict__ gate,
                                                        int gate_rows) {
  // fallback launch of the fused encoder path: runs only if a flag was raised -- gate[0] = number of flagged token
  // tiles; with gate_rows > 0 a row runs only if ITS tile of gate_rows rows is flagged (gate[1 + row / gate_rows])
  if (gate && *gate == 0) return;
  constexpr int C = (R - S) * 64;
  constexpr int EPL = VEC ? (16 / (int)sizeof(VT)) : 1;
  __shared__ u64 lds_q[4][C];
  __shared__ u64 lds_m[4][S * 64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t row = WPR == 4 ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * 4 + wave;
  const int part = WPR == 4 ? wave : 0;
  if (WPR == 1 && row >= N) return;
  if (gate && gate_rows > 0 && gate[1 + row / gate_rows] == 0) return;  // wave-uniform (block-uniform for WPR == 4)

  constexpr bool RX = WPR == 1 && S <= 2;  // radix flush for one wave per row; the 4-wave LDS merge keeps sorted lists
  WaveSelect<S, R, RX> ws;
  ws.init(lds_q[wave], lane, k);
  ws.set_bits(sizeof(VT) == 2 ? 48 : 32, cand_idx ? 32 : 32 - __clz((int)(L > 1 ? L - 1 : 1)));

  int64_t len = counts ? (int64_t)counts[row] : L;
  if (len > L) len = L;
  const VT* rp = vals + row * ld;
  const int64_t* ip = cand_idx ? cand_idx + row * ld : nullptr;
  const bool accept_zero = cand_idx != nullptr;

  // this wave's span, in units of 64*EPL elements
  const int64_t step = 64 * EPL;
  const int64_t nsteps = (len + step - 1) / step;
  const int64_t per = (nsteps + WPR - 1) / WPR;
  const int64_t s_beg = per * part, s_end = min(nsteps, per * (part + 1));

  for (int64_t s = s_beg; s < s_end; ++s) {
    const int64_t e0 = s * step + (int64_t)lane * EPL;
    if constexpr (VEC) {
      float f[EPL];
      if (e0 + EPL <= len) {
        Chunk<VT>::load(rp + e0, f);
      } else {
#pragma unroll
        for (int i = 0; i < EPL; ++i) f[i] = (e0 + i < len) ? (float)rp[e0 + i] : -1.f;
      }
#pragma unroll
      for (int i = 0; i < EPL; ++i) {
        const float v = f[i];
        const bool cand = v >= ws.thr_val && v > 0.f;
        ws.offer(cand, v, (unsigned)(e0 + i));
      }
    } else {
      float v = -1.f;
      unsigned col = 0;
      if (e0 < len) {
        v = (float)rp[e0];
        col = ip ? (unsigned)ip[e0] : (unsigned)e0;
      }
      const bool cand = v >= ws.thr_val && (accept_zero ? v >= 0.f : v > 0.f);
      ws.offer(cand, v, col);
    }
  }
  if (ws.cnt > 0) ws.flush();

  if constexpr (WPR == 4) {
    // tree merge through LDS: (0,1) and (2,3), then (0,2)
    lds_u64* mine = (lds_u64*)lds_m[wave];
#pragma unroll
    for (int r = 0; r < S; ++r) mine[r * 64 + lane] = ws.key[r];
    __syncthreads();
    if ((wave & 1) == 0) {
      merge_from<S, R>(ws, (const lds_u64*)lds_m[wave + 1]);
    }
    __syncthreads();
    if (wave == 2) {
#pragma unroll
      for (int r = 0; r < S; ++r) mine[r * 64 + lane] = ws.key[r];
    }
    __syncthreads();
    if (wave == 0) {
      merge_from<S, R>(ws, (const lds_u64*)lds_m[2]);
      write_out<S, R>(ws, out_vals, out_dtype, row, out_idx, k);
    }
  } else {
    ws.finalize();
    write_out<S, R>(ws, out_vals, out_dtype, row, out_idx, k);
  }
}

}  // namespace clt

using namespace clt;

template <typename VT, int S, int R>
static void launch_topk(bool long_rows, bool vec, hipStream_t st, const void* vals, int64_t ld, const int64_t* cand_idx,
                        const int32_t* counts, int64_t N, int64_t L, int k, void* out_vals, int out_dtype, int64_t* out_idx,
                        const int* gate, int gate_rows) {
  if (long_rows) {
    dim3 grid((unsigned)N);
    if (vec)
      hipLaunchKernelGGL((topk_rows_kernel<VT, S, R, 4, true>), grid, dim3(256), 0, st, (const VT*)vals, ld, cand_idx,
                         counts, N, L, k, out_vals, out_dtype, out_idx, gate, gate_rows);
    else
      hipLaunchKernelGGL((topk_rows_kernel<VT, S, R, 4, false>), grid, dim3(256), 0, st, (const VT*)vals, ld,
                         cand_idx, counts, N, L, k, out_vals, out_dtype, out_idx, gate, gate_rows);
  } else {
    dim3 grid((unsigned)((N + 3) / 4));
    if (vec)
      hipLaunchKernelGGL((topk_rows_kernel<VT, S, R, 1, true>), grid, dim3(256), 0, st, (const VT*)vals, ld, cand_idx,
                         counts, N, L, k, out_vals, out_dtype, out_idx, gate, gate_rows);
    else
      hipLaunchKernelGGL((topk_rows_kernel<VT, S, R, 1, false>), grid, dim3(256), 0, st, (const VT*)vals, ld,
                         cand_idx, counts, N, L, k, out_vals, out_dtype, out_idx, gate, gate_rows);
  }
}

template <typename VT>
static void launch_topk_s(int S, bool long_rows, bool vec, hipStream_t st, const void* vals, int64_t ld,
                          const int64_t* cand_idx, const int32_t* counts, int64_t N, int64_t L, int k, void* out_vals,
                          int out_dtype, int64_t* out_idx, const int* gate, int gate_rows) {
  if (S == 1)
    launch_topk<VT, 1, 4>(long_rows, vec, st, vals, ld, cand_idx, counts, N, L, k, out_vals, out_dtype, out_idx, gate, gate_rows);
  else if (S == 2)
    launch_topk<VT, 2, 8>(long_rows, vec, st, vals, ld, cand_idx, counts, N, L, k, out_vals, out_dtype, out_idx, gate, gate_rows);
  else
    launch_topk<VT, 4, 8>(long_rows, vec, st, vals, ld, cand_idx, counts, N, L, k, out_vals, out_dtype, out_idx, gate, gate_rows);
}

int clt::topk_rows_gated(const void* vals, int val_dtype, int64_t ld, const int64_t* cand_idx, const int32_t* counts,
                         int64_t N, int64_t L, int k, void* out_vals, int out_dtype, int64_t* out_idx, const int* gate,
                         int gate_rows, hipStream_t stream) {
  CLT_CHECK_ARG(vals && out_vals && out_idx, "clt_topk_rows: null pointer");
  CLT_CHECK_ARG(N >= 0 && L > 0 && ld >= L && L < (1ll << 31), "clt_topk_rows: bad sizes N=%lld L=%lld ld=%lld",
                (long long)N, (long long)L, (long long)ld);
  CLT_CHECK_ARG(valid_dtype(val_dtype) && valid_dtype(out_dtype), "clt_topk_rows: bad dtype");
  CLT_UNSUPPORTED(k >= 1 && k <= 256, "clt_topk_rows: k=%d outside [1, 256]", k);
  CLT_CHECK_ARG(k <= L, "clt_topk_rows: k=%d > L=%lld", k, (long long)L);
  if (N == 0) return CLT_OK;
  const int S = k <= 64 ? 1 : k <= 128 ? 2 : 4;
  const size_t es = dtype_size(val_dtype);
  const bool vec = cand_idx == nullptr && aligned16(vals) && ((size_t)ld * es) % 16 == 0;
  // four waves per row pay a fixed LDS merge per row: worth it for very long rows, or when there are too few rows
  // to fill the chip with one wave each
  const bool long_rows = L >= 32768 || (L >= 4096 && N < 4096);
  hipStream_t st = stream;
  if (val_dtype == CLT_BF16)
    launch_topk_s<bf16_t>(S, long_rows, vec, st, vals, ld, cand_idx, counts, N, L, k, out_vals, out_dtype, out_idx, gate, gate_rows);
  else
    launch_topk_s<float>(S, long_rows, vec, st, vals, ld, cand_idx, counts, N, L, k, out_vals, out_dtype, out_idx, gate, gate_rows);
  CLT_CHECK_LAUNCH("clt_topk_rows");
  return CLT_OK;
}

extern "C" int clt_topk_rows(const void* vals, int val_dtype, int64_t ld, const int64_t* cand_idx,
                             const int32_t* counts, int64_t N, int64_t L, int k, void* out_vals, int out_dtype,
                             int64_t* out_idx, clt_stream_t stream) {
  return topk_rows_gated(vals, val_dtype, ld, cand_idx, counts, N, L, k, out_vals, out_dtype, out_idx, nullptr, 0,
                         (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------------
// groupmax (fused_encoder.py:373-413): the M latents of a row are k consecutive groups of M / k; each group keeps its
// largest pre-activation (ties: lowest index, torch's CPU rule) -- out[n, g] = max, idx[n, g] = g * (M / k) + argmax.
// One wave per (row, group): the group is a contiguous run of the row, read once in 16-byte chunks (HBM-bound:
// N * M * sizeof(T) bytes per call).  Within a lane the indices ascend, so a strict '>' keeps the lowest; across lanes
// the reduction compares (value, then index).
namespace clt {

template <typename VT, typename OT, bool VEC>
__global__ __launch_bounds__(256) void groupmax_rows_kernel(const VT* __restrict__ vals, int64_t ld, int64_t N, int k,
                                                            int gs, OT* __restrict__ out_vals,
                                                            int64_t* __restrict__ out_idx) {
  const int lane = threadIdx.x & 63;
  const int64_t unit = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // (row, group), group fastest
  if (unit >= N * k) return;  // wave-uniform
  const int64_t row = unit / k;
  const int g = (int)(unit - row * k);
  const VT* p = vals + row * ld + (int64_t)g * gs;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  if (VEC) {
    constexpr int E = Chunk<VT>::E;
    for (int c = lane; c * E < gs; c += 64) {
      float f[E];
      Chunk<VT>::load(p + c * E, f);
#pragma unroll
      for (int e = 0; e < E; ++e)
        if (f[e] > best) {
          best = f[e];
          bi = c * E + e;
        }
    }
  } else {
    for (int i = lane; i < gs; i += 64) {
      const float f = (float)p[i];
      if (f > best) {
        best = f;
        bi = i;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) {
      best = ov;
      bi = oi;
    }
  }
  if (lane == 0) {
    out_vals[unit] = (OT)best;
    out_idx[unit] = (int64_t)g * gs + bi;
  }
}

}  // namespace clt

namespace clt {
__global__ void candidates_pack_kernel(const void* __restrict__ vals, int val_dtype, const int64_t* __restrict__ idx,
                                       int64_t idx_offset, int64_t n, int64_t* __restrict__ packed) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t vb = __float_as_uint(load_scalar(vals, val_dtype, i));
  packed[i] = (int64_t)((vb << 32) | (uint64_t)(uint32_t)(idx[i] + idx_offset));
}

template <typename VT>
__global__ void candidates_unpack_kernel(const int64_t* __restrict__ packed, int64_t groups, int64_t rows, int64_t k,
                                         VT* __restrict__ vals, int64_t* __restrict__ idx) {
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // output position: (t, r, j)
  const int64_t gk = groups * k;
  if (o >= rows * gk) return;
  const int64_t t = o / gk, rj = o - t * gk, r = rj / k, j = rj - r * k;
  const uint64_t p = (uint64_t)packed[(r * rows + t) * k + j];
  const float v = __uint_as_float((uint32_t)(p >> 32));
  if constexpr (sizeof(VT) == 2) vals[o] = (VT)v;
  else vals[o] = v;
  idx[o] = (int64_t)(p & 0xffffffffull);
}
}  // namespace clt

extern "C" int clt_candidates_pack(const void* vals, int val_dtype, const int64_t* idx, int64_t idx_offset, int64_t n,
                                   int64_t* packed, clt_stream_t stream) {
  CLT_CHECK_ARG(vals && idx && packed, "clt_candidates_pack: null pointer");
  CLT_CHECK_ARG(valid_dtype(val_dtype) && n > 0 && n < (1ll << 40) && idx_offset >= 0 && idx_offset < (1ll << 32),
                "clt_candidates_pack: bad arguments");
  hipLaunchKernelGGL(candidates_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, vals,
                     val_dtype, idx, idx_offset, n, packed);
  CLT_CHECK_LAUNCH("clt_candidates_pack");
  return CLT_OK;
}

extern "C" int clt_candidates_unpack(const int64_t* packed, int64_t groups, int64_t rows, int64_t k, void* vals,
                                     int val_dtype, int64_t* idx, clt_stream_t stream) {
  CLT_CHECK_ARG(packed && vals && idx, "clt_candidates_unpack: null pointer");
  CLT_CHECK_ARG(valid_dtype(val_dtype) && groups > 0 && rows > 0 && k > 0 && groups * rows * k < (1ll << 40),
                "clt_candidates_unpack: bad arguments");
  const int64_t n = groups * rows * k;
  const dim3 grid((unsigned)((n + 255) / 256));
  if (val_dtype == CLT_BF16)
    hipLaunchKernelGGL((candidates_unpack_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, packed, groups, rows, k,
                       (bf16_t*)vals, idx);
  else
    hipLaunchKernelGGL((candidates_unpack_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, packed, groups, rows, k,
                       (float*)vals, idx);
  CLT_CHECK_LAUNCH("clt_candidates_unpack");
  return CLT_OK;
}

extern "C" int clt_groupmax_rows(const void* vals, int val_dtype, int64_t ld, int64_t N, int64_t M, int k,
                                 void* out_vals, int out_dtype, int64_t* out_idx, clt_stream_t stream) {
  using namespace clt;
  CLT_CHECK_ARG(vals && out_vals && out_idx, "clt_groupmax_rows: null pointer");
  CLT_CHECK_ARG(valid_dtype(val_dtype) && valid_dtype(out_dtype), "clt_groupmax_rows: bad dtype");
  CLT_CHECK_ARG(N >= 0 && M > 0 && M < (1ll << 31) && k >= 1 && k <= M && ld >= M, "clt_groupmax_rows: bad sizes");
  CLT_CHECK_ARG(M % k == 0, "clt_groupmax_rows: k=%d must divide the %lld latents (fused_encoder.py:409)", k, (long long)M);
  if (N == 0) return CLT_OK;
  const int gs = (int)(M / k);
  const size_t es = dtype_size(val_dtype);
  const bool vec = aligned16(vals) && (ld * es) % 16 == 0 && (gs * es) % 16 == 0;
  const int64_t units = N * k;
  CLT_CHECK_ARG((units + 3) / 4 < (1ll << 31), "clt_groupmax_rows: too many (row, group) pairs");
  dim3 grid((unsigned)((units + 3) / 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
#define CLT_GM(VT, OT)                                                                                              \
  do {                                                                                                              \
    if (vec)                                                                                                        \
      hipLaunchKernelGGL((groupmax_rows_kernel<VT, OT, true>), grid, block, 0, st, (const VT*)vals, ld, N, k, gs,   \
                         (OT*)out_vals, out_idx);                                                                   \
    else                                                                                                            \
      hipLaunchKernelGGL((groupmax_rows_kernel<VT, OT, false>), grid, block, 0, st, (const VT*)vals, ld, N, k, gs,  \
                         (OT*)out_vals, out_idx);                                                                   \
  } while (0)
  if (val_dtype == CLT_BF16 && out_dtype == CLT_BF16) CLT_GM(bf16_t, bf16_t);
  else if (val_dtype == CLT_BF16) CLT_GM(bf16_t, float);
  else if (out_dtype == CLT_BF16) CLT_GM(float, bf16_t);
  else CLT_GM(float, float);
#undef CLT_GM
  CLT_CHECK_LAUNCH("clt_groupmax_rows");
  return CLT_OK;
}

// relu(x W^T + b) -> groupmax: the GEMM writes the pre-activation matrix into the workspace (same size rule as
// clt_encode_topk_workspace_bytes), the row pass above reads it once.
extern "C" int clt_encode_groupmax_fwd(const void* x, const void* W, int in_dtype, const void* bias, int bias_dtype,
                                       int64_t N, int64_t M, int d, int k, void* out_vals, int out_dtype,
                                       int64_t* out_idx, void* workspace, size_t workspace_bytes, clt_stream_t stream) {
  using namespace clt;
  CLT_CHECK_ARG(workspace && aligned16(workspace), "clt_encode_groupmax_fwd: workspace must be a 16-byte aligned pointer");
  CLT_CHECK_ARG(M > 0 && k >= 1 && M % k == 0, "clt_encode_groupmax_fwd: k must divide the latent count");
  const size_t need = clt_encode_topk_workspace_bytes(N, M, out_dtype);
  if (workspace_bytes < need) {
    set_error("clt_encode_groupmax_fwd: workspace too small (%zu < %zu)", workspace_bytes, need);
    return CLT_ERR_WORKSPACE;
  }
  int rc = clt_encoder_gemm_relu(x, W, in_dtype, bias, bias_dtype, workspace, out_dtype, N, M, d, stream);
  if (rc != CLT_OK) return rc;
  return clt_groupmax_rows(workspace, out_dtype, M, N, M, k, out_vals, out_dtype, out_idx, stream);
}
