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
#include "common.h"

namespace clt {

typedef unsigned long long u64;

__device__ __forceinline__ u64 shfl_xor_u64(u64 x, int mask) {
  unsigned lo = (unsigned)x, hi = (unsigned)(x >> 32);
  lo = __shfl_xor((int)lo, mask, 64);
  hi = __shfl_xor((int)hi, mask, 64);
  return ((u64)hi << 32) | lo;
}

__device__ __forceinline__ u64 readlane_u64(u64 x, int lane) {
  unsigned lo = (unsigned)readlane_i((int)(unsigned)x, lane);
  unsigned hi = (unsigned)readlane_i((int)(unsigned)(x >> 32), lane);
  return ((u64)hi << 32) | lo;
}

// Descending bitonic sort of 64*R keys; element e lives in key[e / 64] of lane e % 64.
template <int R>
__device__ __forceinline__ void bitonic_sort_desc(u64 (&key)[R], int lane) {
  constexpr int n = 64 * R;
#pragma unroll
  for (int size = 2; size <= n; size <<= 1) {
#pragma unroll
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      if (stride >= 64) {
        const int rs = stride >> 6;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if ((r & rs) == 0) {
            const bool desc = ((r * 64) & size) == 0;
            const u64 a = key[r], b = key[r | rs];
            const bool sw = desc ? (a < b) : (a > b);
            key[r] = sw ? b : a;
            key[r | rs] = sw ? a : b;
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const u64 a = key[r];
          const u64 p = shfl_xor_u64(a, stride);
          const bool lower = (lane & stride) == 0;
          const bool desc = size >= 64 ? (((r * 64) & size) == 0) : ((lane & size) == 0);
          const bool keep_max = (lower == desc);
          const u64 mx = a > p ? a : p, mn = a > p ? p : a;
          key[r] = keep_max ? mx : mn;
        }
      }
    }
  }
}

template <int R>
__device__ __forceinline__ u64 element_at(const u64 (&key)[R], int e) {
  u64 out = 0;
#pragma unroll
  for (int r = 0; r < R; ++r)
    if ((e >> 6) == r) out = readlane_u64(key[r], e & 63);
  return out;
}

__device__ __forceinline__ u64 make_key(float v, unsigned col) {
  unsigned bits = v > 0.f ? __float_as_uint(v) : 0u;
  return ((u64)bits << 32) | (u64)(~col);
}

// Per-wave selection state.
template <int S, int R>
struct WaveSelect {
  static constexpr int C = (R - S) * 64;  // queue capacity
  u64 key[R];
  u64 thr_key;
  float thr_val;
  int cnt;
  volatile u64* q;  // this wave's LDS queue [C]
  int lane, k;

  __device__ __forceinline__ void init(volatile u64* queue, int lane_, int k_) {
#pragma unroll
    for (int r = 0; r < R; ++r) key[r] = 0;
    thr_key = 0;
    thr_val = 0.f;
    cnt = 0;
    q = queue;
    lane = lane_;
    k = k_;
  }

  __device__ __forceinline__ void flush() {
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's queue writes have landed
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = S; r < R; ++r) {
      const int e = (r - S) * 64 + lane;
      key[r] = e < cnt ? q[e] : 0ull;
    }
    __builtin_amdgcn_wave_barrier();
    bitonic_sort_desc<R>(key, lane);
    thr_key = element_at<R>(key, k - 1);
    thr_val = __uint_as_float((unsigned)(thr_key >> 32));
    cnt = 0;
  }

  // Offer one candidate per lane.  `cand` = this lane holds a possible candidate.
  __device__ __forceinline__ void offer(bool cand, float v, unsigned col) {
    if (__ballot(cand) == 0ull) return;
    const u64 kk = make_key(v, col);
    cand = cand && (kk > thr_key);
    const u64 b = __ballot(cand);
    if (b == 0ull) return;
    if (cnt + 64 > C) flush(), cand = cand && (kk > thr_key);
    const u64 b2 = __ballot(cand);
    if (b2 == 0ull) return;
    const int pos = cnt + __builtin_amdgcn_mbcnt_hi((unsigned)(b2 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b2, 0));
    if (cand) q[pos] = kk;
    cnt += __popcll(b2);
  }
};

// Load the top list of another wave (written as [S*64] keys) into regs [S, 2S) and re-sort.
template <int S, int R>
__device__ __forceinline__ void merge_from(WaveSelect<S, R>& ws, const volatile u64* other) {
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (r >= S && r < 2 * S) ws.key[r] = other[(r - S) * 64 + ws.lane];
    else if (r >= 2 * S) ws.key[r] = 0ull;
  }
  bitonic_sort_desc<R>(ws.key, ws.lane);
}

__device__ __forceinline__ void store_val(void* base, int dtype, int64_t i, float v) {
  if (dtype == CLT_BF16) reinterpret_cast<bf16_t*>(base)[i] = (bf16_t)v;
  else reinterpret_cast<float*>(base)[i] = v;
}

// Write the k best of a finished selection; complete short rows (fewer than k
// candidates) with (0, lowest unselected column numbers).
template <int S, int R>
__device__ __forceinline__ void write_out(WaveSelect<S, R>& ws, void* out_base, int out_dtype, int64_t row,
                                          int64_t* out_idx, int k) {
  const int64_t ob = row * k;
  out_idx += ob;
  const int lane = ws.lane;
  int P = 0;  // number of real candidates among the first k elements
#pragma unroll
  for (int r = 0; r < S; ++r) {
    const int e = r * 64 + lane;
    const bool real = e < k && ws.key[r] != 0ull;
    P += __popcll(__ballot(real));
    if (real) {
      store_val(out_base, out_dtype, ob + e, __uint_as_float((unsigned)(ws.key[r] >> 32)));
      out_idx[e] = (int64_t)(unsigned)(~(unsigned)ws.key[r]);
    }
  }
  if (P >= k) return;  // wave-uniform
  // rare path: lowest column numbers c in [0, 64*S) that are not selected
  bool taken[S];
#pragma unroll
  for (int q = 0; q < S; ++q) taken[q] = false;
  for (int e2 = 0; e2 < P; ++e2) {
    const unsigned sidx = ~(unsigned)element_at<R>(ws.key, e2);
#pragma unroll
    for (int q = 0; q < S; ++q) taken[q] = taken[q] || (sidx == (unsigned)(q * 64 + lane));
  }
  int base = P;
#pragma unroll
  for (int q = 0; q < S; ++q) {
    const bool fr = !taken[q];
    const u64 b = __ballot(fr);
    const int slot = base + __builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0));
    if (fr && slot < k) {
      store_val(out_base, out_dtype, ob + slot, 0.f);
      out_idx[slot] = (int64_t)(q * 64 + lane);
    }
    base += __popcll(b);
  }
}

// WPR waves cooperate on one row (WPR = 4: one row per 256-thread block; WPR = 1: four rows per block).
template <typename VT, int S, int R, int WPR, bool VEC>
__global__ __launch_bounds__(256) void topk_rows_kernel(const VT* __restrict__ vals, int64_t ld,
                                                        const int64_t* __restrict__ cand_idx,
                                                        const int32_t* __restrict__ counts, int64_t N, int64_t L,
                                                        int k, void* __restrict__ out_vals, int out_dtype,
                                                        int64_t* __restrict__ out_idx) {
  constexpr int C = (R - S) * 64;
  constexpr int EPL = VEC ? (16 / (int)sizeof(VT)) : 1;
  __shared__ u64 lds_q[4][C];
  __shared__ u64 lds_m[4][S * 64];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t row = WPR == 4 ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * 4 + wave;
  const int part = WPR == 4 ? wave : 0;
  if (WPR == 1 && row >= N) return;

  WaveSelect<S, R> ws;
  ws.init(lds_q[wave], lane, k);

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
    write_out<S, R>(ws, out_vals, out_dtype, row, out_idx, k);
  }
}

}  // namespace clt

using namespace clt;

template <typename VT, int S, int R>
static void launch_topk(bool long_rows, bool vec, hipStream_t st, const void* vals, int64_t ld, const int64_t* cand_idx,
                        const int32_t* counts, int64_t N, int64_t L, int k, void* out_vals, int out_dtype, int64_t* out_idx) {
  if (long_rows) {
    dim3 grid((unsigned)N);
    if (vec)
      hipLaunchKernelGGL((topk_rows_kernel<VT, S, R, 4, true>), grid, dim3(256), 0, st, (const VT*)vals, ld, cand_idx,
                         counts, N, L, k, out_vals, out_dtype, out_idx);
    else
      hipLaunchKernelGGL((topk_rows_kernel<VT, S, R, 4, false>), grid, dim3(256), 0, st, (const VT*)vals, ld,
                         cand_idx, counts, N, L, k, out_vals, out_dtype, out_idx);
  } else {
    dim3 grid((unsigned)((N + 3) / 4));
    if (vec)
      hipLaunchKernelGGL((topk_rows_kernel<VT, S, R, 1, true>), grid, dim3(256), 0, st, (const VT*)vals, ld, cand_idx,
                         counts, N, L, k, out_vals, out_dtype, out_idx);
    else
      hipLaunchKernelGGL((topk_rows_kernel<VT, S, R, 1, false>), grid, dim3(256), 0, st, (const VT*)vals, ld,
                         cand_idx, counts, N, L, k, out_vals, out_dtype, out_idx);
  }
}

template <typename VT>
static void launch_topk_s(int S, bool long_rows, bool vec, hipStream_t st, const void* vals, int64_t ld,
                          const int64_t* cand_idx, const int32_t* counts, int64_t N, int64_t L, int k, void* out_vals,
                          int out_dtype, int64_t* out_idx) {
  if (S == 1)
    launch_topk<VT, 1, 4>(long_rows, vec, st, vals, ld, cand_idx, counts, N, L, k, out_vals, out_dtype, out_idx);
  else if (S == 2)
    launch_topk<VT, 2, 8>(long_rows, vec, st, vals, ld, cand_idx, counts, N, L, k, out_vals, out_dtype, out_idx);
  else
    launch_topk<VT, 4, 8>(long_rows, vec, st, vals, ld, cand_idx, counts, N, L, k, out_vals, out_dtype, out_idx);
}

extern "C" int clt_topk_rows(const void* vals, int val_dtype, int64_t ld, const int64_t* cand_idx,
                             const int32_t* counts, int64_t N, int64_t L, int k, void* out_vals, int out_dtype,
                             int64_t* out_idx, clt_stream_t stream) {
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
  const bool long_rows = L >= 4096;
  hipStream_t st = (hipStream_t)stream;
  if (val_dtype == CLT_BF16)
    launch_topk_s<bf16_t>(S, long_rows, vec, st, vals, ld, cand_idx, counts, N, L, k, out_vals, out_dtype, out_idx);
  else
    launch_topk_s<float>(S, long_rows, vec, st, vals, ld, cand_idx, counts, N, L, k, out_vals, out_dtype, out_idx);
  CLT_CHECK_LAUNCH("clt_topk_rows");
  return CLT_OK;
}
