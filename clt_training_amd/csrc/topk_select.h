// Wave-level streaming selection shared by the row top-k kernels (topk.hip) and the fused encoder path
// (encoder_gemm_fast.hip): 64-bit keys (value_bits << 32) | ~index, register-resident bitonic sorter.
#pragma once
#include "common.h"

namespace clt {

typedef unsigned long long u64;
// The per-wave queue lives in LDS and is addressed as LDS: through a generic `volatile u64*` every append compiled to
// flat_store_dwordx2 + s_waitcnt vmcnt(0) (LLVM keeps volatile accesses flat and completes each one), a full memory
// round trip per offered element in every selection kernel.
typedef __attribute__((address_space(3))) u64 lds_u64;
// Orders this wave's LDS accesses for the compiler and the hardware (LDS executes a wave's operations in order; the wait
// covers the returning reads).
__device__ __forceinline__ void wave_lds_fence() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

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

// Per-wave selection state.  key[0, S) hold the best S*64 keys seen so far, key[S, R) receive the queue on a flush.
// RX = false: a flush is one bitonic sort of all 64*R keys (the list stays sorted).
// RX = true : a flush finds the k-th largest key by a most-significant-bit-first search over the bits that can
//             differ (value bits [vlo, 62] and index bits [0, lob): one v_cmp_ge_u64 + s_bcnt1 per register and
//             bit, ~5x fewer instructions than the sort), keeps the k survivors UNSORTED, and finalize() orders
//             them once at the end by counting ranks through LDS.  Needs 2*S*64 <= C (S <= 2).
template <int S, int R, bool RX = false>
struct WaveSelect {
  static constexpr int C = (R - S) * 64;  // queue capacity
  u64 key[R];
  u64 thr_key;
  float thr_val;
  int cnt;
  lds_u64* q;  // this wave's LDS queue [C]
  int lane, k;
  int vlo, lob, nsel;

  __device__ __forceinline__ void init(volatile u64* queue, int lane_, int k_) {
#pragma unroll
    for (int r = 0; r < R; ++r) key[r] = 0;
    thr_key = 0;
    thr_val = 0.f;
    cnt = 0;
    q = (lds_u64*)(u64*)queue;
    lane = lane_;
    k = k_;
    vlo = 32;
    lob = 32;
    nsel = 0;
  }
  // value_low_bit: 48 when every value is a bf16 (low 16 bits of the f32 pattern are zero), else 32;
  // index_bits: every column index is < 2^index_bits
  __device__ __forceinline__ void set_bits(int value_low_bit, int index_bits) {
    vlo = value_low_bit;
    lob = index_bits;
  }

  __device__ __forceinline__ void flush() {
    wave_lds_fence();  // this wave's queue writes are ordered before the reads
#pragma unroll
    for (int r = S; r < R; ++r) {
      const int e = (r - S) * 64 + lane;
      key[r] = e < cnt ? q[e] : 0ull;
    }
    wave_lds_fence();
    if constexpr (!RX) {
      bitonic_sort_desc<R>(key, lane);
      thr_key = element_at<R>(key, k - 1);
    } else {
      int nz = 0;
#pragma unroll
      for (int r = 0; r < R; ++r) nz += __popcll(__ballot(key[r] != 0ull));
      u64 T = 1ull;  // fewer than k real keys: keep them all
      if (nz >= k) {
        // invariant: at least k keys are >= prefix.  Index bits [lob, 32) are ones in every real key.
        u64 prefix = lob < 32 ? (1ull << 32) - (1ull << lob) : 0ull;
        int at_prefix = nz;  // number of keys >= prefix
        auto step = [&](int b) {
          const u64 t = prefix | (1ull << b);
          int c = 0;
#pragma unroll
          for (int r = 0; r < R; ++r) c += __popcll(__ballot(key[r] >= t));
          if (c >= k) prefix = t, at_prefix = c;
        };
        for (int b = 62; b >= vlo; --b) step(b);
        // exactly k keys reach the k-th VALUE: no tie to break at the boundary, every one of them is selected and the
        // index bits need no search (half of the steps; the usual case)
        if (at_prefix != k)
          for (int b = lob - 1; b >= 0; --b) step(b);
        T = prefix;  // the k-th largest key (keys are distinct), or a bound that exactly k keys reach
      }
      int base = 0;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const bool sel = key[r] >= T;
        const u64 m = __ballot(sel);
        const int pos = base + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
        if (sel) q[pos] = key[r];
        base += __popcll(m);
      }
      wave_lds_fence();
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int e = r * 64 + lane;
        key[r] = (r < S && e < base) ? q[e] : 0ull;
      }
      wave_lds_fence();
      nsel = base;
      thr_key = nz >= k ? T : 0ull;
    }
    thr_val = __uint_as_float((unsigned)(thr_key >> 32));
    cnt = 0;
  }

  // Order the survivors (largest first).  Call after the last flush, before write_out.
  __device__ __forceinline__ void finalize() {
    if constexpr (RX) {
      static_assert(2 * S * 64 <= C, "rank buffer does not fit the queue");
      // q[0, nsel) still holds the survivors (written by the last flush): rank = number of larger keys
      int rank[S];
#pragma unroll
      for (int r = 0; r < S; ++r) rank[r] = 0;
      for (int j = 0; j < nsel; ++j) {
        const u64 o = q[j];  // same address in every lane: LDS broadcast
#pragma unroll
        for (int r = 0; r < S; ++r) rank[r] += o > key[r] ? 1 : 0;
      }
#pragma unroll
      for (int r = 0; r < S; ++r)
        if (r * 64 + lane < nsel) q[S * 64 + rank[r]] = key[r];
      wave_lds_fence();
#pragma unroll
      for (int r = 0; r < S; ++r) key[r] = r * 64 + lane < nsel ? q[S * 64 + r * 64 + lane] : 0ull;
      wave_lds_fence();
    }
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
__device__ __forceinline__ void merge_from(WaveSelect<S, R>& ws, const lds_u64* other) {
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
template <int S, int R, bool RX>
__device__ __forceinline__ void write_out(WaveSelect<S, R, RX>& ws, void* out_base, int out_dtype, int64_t row,
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

// clt_topk_rows with a device-side gate: when gate != nullptr the kernel returns at once unless gate[0] != 0, and with
// gate_rows > 0 a row is processed only if gate[1 + row / gate_rows] != 0 (the fused encoder's per-token-tile fallback).
int topk_rows_gated(const void* vals, int val_dtype, int64_t ld, const int64_t* cand_idx, const int32_t* counts,
                    int64_t N, int64_t L, int k, void* out_vals, int out_dtype, int64_t* out_idx, const int* gate,
                    int gate_rows, hipStream_t stream);

}  // namespace clt
