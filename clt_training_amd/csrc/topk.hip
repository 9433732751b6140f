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
                                                        int64_t* __restrict__ out_idx, const int* __restrict__ gate) {
  if (gate && *gate == 0) return;  // fallback launch of the fused encoder path: runs only if the flag was raised
  constexpr int C = (R - S) * 64;
  constexpr int EPL = VEC ? (16 / (int)sizeof(VT)) : 1;
  __shared__ u64 lds_q[4][C];
  __shared__ u64 lds_m[4][S * 64];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t row = WPR == 4 ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * 4 + wave;
  const int part = WPR == 4 ? wave : 0;
  if (WPR == 1 && row >= N) return;

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
    volatile u64* mine = lds_m[wave];
#pragma unroll
    for (int r = 0; r < S; ++r) mine[r * 64 + lane] = ws.key[r];
    __syncthreads();
    if ((wave & 1) == 0) {
      merge_from<S, R>(ws, lds_m[wave + 1]);
    }
    __syncthreads();
    if (wave == 2) {
#pragma unroll
      for (int r = 0; r < S; ++r) mine[r * 64 + lane] = ws.key[r];
    }
    __syncthreads();
    if (wave == 0) {
      merge_from<S, R>(ws, lds_m[2]);
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
                        const int* gate) {
  if (long_rows) {
    dim3 grid((unsigned)N);
    if (vec)
      hipLaunchKernelGGL((topk_rows_kernel<VT, S, R, 4, true>), grid, dim3(256), 0, st, (const VT*)vals, ld, cand_idx,
                         counts, N, L, k, out_vals, out_dtype, out_idx, gate);
    else
      hipLaunchKernelGGL((topk_rows_kernel<VT, S, R, 4, false>), grid, dim3(256), 0, st, (const VT*)vals, ld,
                         cand_idx, counts, N, L, k, out_vals, out_dtype, out_idx, gate);
  } else {
    dim3 grid((unsigned)((N + 3) / 4));
    if (vec)
      hipLaunchKernelGGL((topk_rows_kernel<VT, S, R, 1, true>), grid, dim3(256), 0, st, (const VT*)vals, ld, cand_idx,
                         counts, N, L, k, out_vals, out_dtype, out_idx, gate);
    else
      hipLaunchKernelGGL((topk_rows_kernel<VT, S, R, 1, false>), grid, dim3(256), 0, st, (const VT*)vals, ld,
                         cand_idx, counts, N, L, k, out_vals, out_dtype, out_idx, gate);
  }
}

template <typename VT>
static void launch_topk_s(int S, bool long_rows, bool vec, hipStream_t st, const void* vals, int64_t ld,
                          const int64_t* cand_idx, const int32_t* counts, int64_t N, int64_t L, int k, void* out_vals,
                          int out_dtype, int64_t* out_idx, const int* gate) {
  if (S == 1)
    launch_topk<VT, 1, 4>(long_rows, vec, st, vals, ld, cand_idx, counts, N, L, k, out_vals, out_dtype, out_idx, gate);
  else if (S == 2)
    launch_topk<VT, 2, 8>(long_rows, vec, st, vals, ld, cand_idx, counts, N, L, k, out_vals, out_dtype, out_idx, gate);
  else
    launch_topk<VT, 4, 8>(long_rows, vec, st, vals, ld, cand_idx, counts, N, L, k, out_vals, out_dtype, out_idx, gate);
}

int clt::topk_rows_gated(const void* vals, int val_dtype, int64_t ld, const int64_t* cand_idx, const int32_t* counts,
                         int64_t N, int64_t L, int k, void* out_vals, int out_dtype, int64_t* out_idx, const int* gate,
                         hipStream_t stream) {
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
    launch_topk_s<bf16_t>(S, long_rows, vec, st, vals, ld, cand_idx, counts, N, L, k, out_vals, out_dtype, out_idx, gate);
  else
    launch_topk_s<float>(S, long_rows, vec, st, vals, ld, cand_idx, counts, N, L, k, out_vals, out_dtype, out_idx, gate);
  CLT_CHECK_LAUNCH("clt_topk_rows");
  return CLT_OK;
}

extern "C" int clt_topk_rows(const void* vals, int val_dtype, int64_t ld, const int64_t* cand_idx,
                             const int32_t* counts, int64_t N, int64_t L, int k, void* out_vals, int out_dtype,
                             int64_t* out_idx, clt_stream_t stream) {
  return topk_rows_gated(vals, val_dtype, ld, cand_idx, counts, N, L, k, out_vals, out_dtype, out_idx, nullptr,
                         (hipStream_t)stream);
}
