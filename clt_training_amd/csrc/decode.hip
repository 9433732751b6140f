// Sparse decoder (embedding bag) forward and grad-wrt-activations for gfx950.
//
// Layout: one 64-lane wave per token.  A weight row is d*sizeof(WT) bytes = a
// whole number of 16-byte chunks; lane l owns chunks l, l+64, l+128, ... (CH of
// them), so every wave-load is a contiguous, fully coalesced run of up to 1 KiB
// of one row.  U rows are kept in flight per wave to cover HBM latency
// (MI355X_MICROARCH.md, "Indexed rows": gathered whole rows into registers reach
// 5.5-5.8 TB/s with one wave per destination and >= 4 rows in flight).
// HBM-bound: algorithmic bytes per token = (k + 1) * d * sizeof(WT).
#include "common.h"

namespace clt {

template <typename GT, int E>
__device__ __forceinline__ void load_elems(const GT* p, float (&f)[E]);

template <>
__device__ __forceinline__ void load_elems<bf16_t, 8>(const bf16_t* p, float (&f)[8]) {
  Chunk<bf16_t>::load(p, f);
}
template <>
__device__ __forceinline__ void load_elems<float, 4>(const float* p, float (&f)[4]) {
  Chunk<float>::load(p, f);
}
template <>
__device__ __forceinline__ void load_elems<float, 8>(const float* p, float (&f)[8]) {
  float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
template <>
__device__ __forceinline__ void load_elems<bf16_t, 4>(const bf16_t* p, float (&f)[4]) {
  uint2 u = *reinterpret_cast<const uint2*>(p);
  f[0] = bf16_lo(u.x); f[1] = bf16_hi(u.x); f[2] = bf16_lo(u.y); f[3] = bf16_hi(u.y);
}

template <typename OT, int E>
__device__ __forceinline__ void store_elems(OT* p, const float (&f)[E]);
template <>
__device__ __forceinline__ void store_elems<bf16_t, 8>(bf16_t* p, const float (&f)[8]) {
  Chunk<bf16_t>::store(p, f);
}
template <>
__device__ __forceinline__ void store_elems<float, 4>(float* p, const float (&f)[4]) {
  Chunk<float>::store(p, f);
}
template <>
__device__ __forceinline__ void store_elems<float, 8>(float* p, const float (&f)[8]) {
  reinterpret_cast<float4*>(p)[0] = make_float4(f[0], f[1], f[2], f[3]);
  reinterpret_cast<float4*>(p)[1] = make_float4(f[4], f[5], f[6], f[7]);
}
template <>
__device__ __forceinline__ void store_elems<bf16_t, 4>(bf16_t* p, const float (&f)[4]) {
  uint2 u;
  u.x = pack_bf16x2(f[0], f[1]);
  u.y = pack_bf16x2(f[2], f[3]);
  *reinterpret_cast<uint2*>(p) = u;
}

// a' = (v + post_enc[idx] * (v > 0)) * post_scale[idx]   (sparse_coder.py:136-140)
__device__ __forceinline__ float post_affine(float v, int r, const float* post_enc, const float* post_scale) {
  float a = v;
  if (post_enc) a += (v > 0.f) ? post_enc[r] : 0.f;
  if (post_scale) a *= post_scale[r];
  return a;
}

// Latent sharding (tensor parallel): this GPU owns table rows [lat_begin, lat_begin + M).  Entries that
// name another shard are skipped -- not masked: the valid entries of a 64-entry chunk are compacted to the
// front of the wave (ds_permute push with a bijective destination map), so a shard pays k/tp row gathers
// per token, not k (the reference masks and still gathers all k: utils.py:520-530).
__device__ __forceinline__ int compact_dest(bool valid, bool active, int& n_valid) {
  const unsigned long long mv = __ballot(valid), mi = __ballot(active && !valid);
  n_valid = __popcll(mv);
  const int pv = __builtin_amdgcn_mbcnt_hi((unsigned)(mv >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mv, 0));
  const int pi = __builtin_amdgcn_mbcnt_hi((unsigned)(mi >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mi, 0));
  const int n_act = __popcll(mv | mi);
  // valid -> [0, n_valid), invalid-but-active -> [n_valid, n_act), inactive lanes keep the tail in lane order
  const unsigned long long mo = ~(mv | mi);
  const int po = __builtin_amdgcn_mbcnt_hi((unsigned)(mo >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mo, 0));
  return valid ? pv : (active ? n_valid + pi : n_act + po);
}
__device__ __forceinline__ int permute_push_i(int dest_lane, int v) {
  return __builtin_amdgcn_ds_permute(dest_lane << 2, v);
}
__device__ __forceinline__ float permute_push_f(int dest_lane, float v) {
  return __int_as_float(__builtin_amdgcn_ds_permute(dest_lane << 2, __float_as_int(v)));
}

template <typename WT, typename OT, int CH, int U>
__global__ __launch_bounds__(256) void decode_fwd_kernel(const int64_t* __restrict__ idx, const void* __restrict__ v,
                                                         int v_dtype, const float* __restrict__ post_enc,
                                                         const float* __restrict__ post_scale,
                                                         const WT* __restrict__ W, int64_t M, int64_t lat_begin,
                                                         int d, const float* __restrict__ bias, OT* out,
                                                         int flags, float* __restrict__ acts_out, int64_t N,
                                                         int k) {
  constexpr int E = Chunk<WT>::E;
  const int lane = threadIdx.x & 63;
  const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;  // wave-uniform
  const int nchunks = d / E;
  float acc[CH][E];
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int e = 0; e < E; ++e) acc[c][e] = 0.f;

  for (int j0 = 0; j0 < k; j0 += 64) {
    const int j = j0 + lane;
    int my_r = 0;
    float my_a = 0.f;
    bool valid = false;
    if (j < k) {
      const int64_t rr = idx[n * k + j] - lat_begin;
      valid = rr >= 0 && rr < M;
      my_r = valid ? (int)rr : 0;
      float a = 0.f;
      if (valid) {
        const float vv = load_scalar(v, v_dtype, n * k + j);
        a = post_affine(vv, my_r, post_enc, post_scale);
        if (flags & CLT_DECODE_ROUND_ACTS) a = bf16_bits_to_f32(f32_to_bf16_bits(a));
      }
      my_a = a;
      if (acts_out) acts_out[n * k + j] = a;
    }
    int cnt = min(64, k - j0);
    if (__ballot(valid) != __ballot(j < k)) {  // some entries belong to other shards: compact
      int nv;
      const int dest = compact_dest(valid, j < k, nv);
      my_r = permute_push_i(dest, my_r);
      my_a = permute_push_f(dest, my_a);
      cnt = nv;
    }
    for (int jj = 0; jj < cnt; jj += U) {
      uint4 raw[U][CH];
      float a_u[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int jl = min(jj + u, cnt - 1);
        const int r = readlane_i(my_r, jl);
        a_u[u] = (jj + u < cnt) ? readlane_f(my_a, jl) : 0.f;
        const char* row = reinterpret_cast<const char*>(W) + (size_t)r * (size_t)d * sizeof(WT);
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int ch = lane + 64 * c;
          raw[u][c] = *reinterpret_cast<const uint4*>(row + (size_t)(ch < nchunks ? ch : nchunks - 1) * 16);  // (unconditional: rows_lds_row, spt_rows.h)
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int ch = lane + 64 * c;
          if (ch < nchunks) {
            float w[E];
            Chunk<WT>::expand(raw[u][c], w);
#pragma unroll
            for (int e = 0; e < E; ++e) acc[c][e] = fmaf(a_u[u], w[e], acc[c][e]);
          }
        }
      }
    }
  }

  OT* orow = out + n * (int64_t)d;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nchunks) {
      if (bias) {
        float b[E];
        load_elems<float, E>(bias + ch * E, b);
#pragma unroll
        for (int e = 0; e < E; ++e) acc[c][e] += b[e];
      }
      if (flags & CLT_DECODE_ACCUMULATE) {
        float o[E];
        load_elems<OT, E>(orow + ch * E, o);
#pragma unroll
        for (int e = 0; e < E; ++e) acc[c][e] += o[e];
      }
      store_elems<OT, E>(orow + ch * E, acc[c]);
    }
  }
}

template <typename WT, typename GT, int CH, int U>
__global__ __launch_bounds__(256) void decode_bwd_acts_kernel(
    const int64_t* __restrict__ idx, const void* __restrict__ v, int v_dtype, const float* __restrict__ post_enc,
    const float* __restrict__ post_scale, const WT* __restrict__ W, int64_t M, int64_t lat_begin, int d,
    const GT* __restrict__ G, float* grad_acts, int accumulate, float* __restrict__ e_post,
    float* __restrict__ e_scale, int64_t N, int k) {
  constexpr int E = Chunk<WT>::E;
  const int lane = threadIdx.x & 63;
  const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const int nchunks = d / E;
  float g[CH][E];
  const GT* grow = G + n * (int64_t)d;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nchunks) {
      load_elems<GT, E>(grow + ch * E, g[c]);
    } else {
#pragma unroll
      for (int e = 0; e < E; ++e) g[c][e] = 0.f;
    }
  }

  for (int j0 = 0; j0 < k; j0 += 64) {
    const int j = j0 + lane;
    int my_r = 0;
    bool valid = false;
    if (j < k) {
      const int64_t rr = idx[n * k + j] - lat_begin;
      valid = rr >= 0 && rr < M;
      my_r = valid ? (int)rr : 0;
    }
    int my_j = j;  // the entry this lane reports on (changes under compaction)
    int cnt = min(64, k - j0);
    if (__ballot(valid) != __ballot(j < k)) {
      int nv;
      const int dest = compact_dest(valid, j < k, nv);
      my_r = permute_push_i(dest, my_r);
      my_j = permute_push_i(dest, j);
      valid = lane < nv;
      cnt = nv;
    }
    float my_ga = 0.f;
    for (int jj = 0; jj < cnt; jj += U) {
      uint4 raw[U][CH];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int jl = min(jj + u, cnt - 1);
        const int r = readlane_i(my_r, jl);
        const char* row = reinterpret_cast<const char*>(W) + (size_t)r * (size_t)d * sizeof(WT);
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int ch = lane + 64 * c;
          raw[u][c] = *reinterpret_cast<const uint4*>(row + (size_t)(ch < nchunks ? ch : nchunks - 1) * 16);  // (unconditional: rows_lds_row, spt_rows.h)
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float part = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int ch = lane + 64 * c;
          if (ch < nchunks) {
            float w[E];
            Chunk<WT>::expand(raw[u][c], w);
#pragma unroll
            for (int e = 0; e < E; ++e) part = fmaf(g[c][e], w[e], part);
          }
        }
        part = wave_sum_dpp(part);
        if (jj + u < cnt && lane == jj + u) my_ga = part;
      }
    }
    if (my_j < k) {  // every entry of the chunk is reported by exactly one lane
      const int64_t o = n * k + my_j;
      float u_ = 0.f, es = 0.f, ep = 0.f;
      if (valid) {
        const float vv = load_scalar(v, v_dtype, o);
        const float pos = vv > 0.f ? 1.f : 0.f;
        u_ = post_scale ? my_ga * post_scale[my_r] : my_ga;
        es = my_ga * (vv + (post_enc ? post_enc[my_r] * pos : 0.f));
        ep = u_ * pos;
      }
      if (e_scale) e_scale[o] = es;
      if (e_post) e_post[o] = ep;
      if (accumulate) {
        if (valid) grad_acts[o] += u_;
      } else {
        grad_acts[o] = u_;
      }
    }
  }
}

static int pick_ch(int d, int E) {
  int chunks = d / E;
  int ch = (chunks + 63) / 64;
  if (ch <= 1) return 1;
  if (ch <= 2) return 2;
  if (ch <= 4) return 4;
  if (ch <= 8) return 8;
  if (ch <= 16) return 16;
  return -1;
}

}  // namespace clt

using namespace clt;

namespace clt {
// the LDS-staged fan-out kernel (below) when the problem fits it: bf16 table, rows of at most 8 KiB, bf16 / fp32 output.
// Returns false without launching otherwise.
static bool decode_fwd_lds_route(const clt_decode_source* hs, int ns, int v_dtype, int w_dtype, int64_t M, int d,
                                 const float* bias, void* out, int out_dtype, int flags, int64_t N, int k, hipStream_t st);
}  // namespace clt

template <typename WT, typename OT>
static int launch_decode_fwd(int ch, dim3 grid, hipStream_t st, const int64_t* idx, const void* v, int v_dtype,
                             const float* pe, const float* ps, const void* W, int64_t M, int64_t lat_begin, int d,
                             const float* bias, void* out, int flags, float* acts_out, int64_t N, int k) {
#define LAUNCH(CH, U)                                                                                            \
  hipLaunchKernelGGL((decode_fwd_kernel<WT, OT, CH, U>), grid, dim3(256), 0, st, idx, v, v_dtype, pe, ps,        \
                     (const WT*)W, M, lat_begin, d, bias, (OT*)out, flags, acts_out, N, k)
  switch (ch) {
    case 1: LAUNCH(1, 8); break;
    case 2: LAUNCH(2, 8); break;
    case 4: LAUNCH(4, 4); break;
    case 8: LAUNCH(8, 2); break;
    case 16: LAUNCH(16, 1); break;
    default: return CLT_ERR_UNSUPPORTED;
  }
#undef LAUNCH
  return CLT_OK;
}

extern "C" int clt_decode_fwd(const int64_t* idx, const void* v, int v_dtype, const float* post_enc,
                              const float* post_scale, const void* W, int w_dtype, int64_t M, int64_t lat_begin, int d,
                              const float* bias, void* out, int out_dtype, int flags, float* acts_out, int64_t N,
                              int k, clt_stream_t stream) {
  CLT_CHECK_ARG(idx && v && W && out, "clt_decode_fwd: null pointer");
  CLT_CHECK_ARG(N >= 0 && k > 0 && d > 0 && M > 0 && M < (1ll << 31), "clt_decode_fwd: bad sizes N=%lld k=%d d=%d M=%lld",
                (long long)N, k, d, (long long)M);
  CLT_CHECK_ARG(valid_dtype(v_dtype) && valid_dtype(w_dtype) && valid_dtype(out_dtype), "clt_decode_fwd: bad dtype");
  CLT_CHECK_ARG(aligned16(W) && aligned16(out) && (!bias || aligned16(bias)), "clt_decode_fwd: pointers must be 16-byte aligned");
  const int E = w_dtype == CLT_BF16 ? 8 : 4;
  CLT_UNSUPPORTED(d % 8 == 0 || (d % 4 == 0 && w_dtype == CLT_F32 && out_dtype == CLT_F32),
                  "clt_decode_fwd: d=%d must be a multiple of 8 (or 4 for all-fp32)", d);
  const int ch = pick_ch(d, E);
  CLT_UNSUPPORTED(ch > 0, "clt_decode_fwd: d=%d too wide", d);
  if (N == 0) return CLT_OK;
  dim3 grid((unsigned)((N + 3) / 4));
  hipStream_t st = (hipStream_t)stream;
  {  // one source of the fan-out kernel (weight 1: the same products and the same summation order)
    clt_decode_source one = {};
    one.idx = idx, one.v = v, one.post_enc = post_enc, one.post_scale = post_scale, one.W = W, one.acts_out = acts_out;
    one.lat_begin = lat_begin, one.weight = 1.f;
    if (decode_fwd_lds_route(&one, 1, v_dtype, w_dtype, M, d, bias, out, out_dtype, flags, N, k, st)) {
      CLT_CHECK_LAUNCH("clt_decode_fwd");
      return CLT_OK;
    }
  }
  int rc;
  if (w_dtype == CLT_BF16 && out_dtype == CLT_BF16)
    rc = launch_decode_fwd<bf16_t, bf16_t>(ch, grid, st, idx, v, v_dtype, post_enc, post_scale, W, M, lat_begin, d, bias, out, flags, acts_out, N, k);
  else if (w_dtype == CLT_BF16 && out_dtype == CLT_F32)
    rc = launch_decode_fwd<bf16_t, float>(ch, grid, st, idx, v, v_dtype, post_enc, post_scale, W, M, lat_begin, d, bias, out, flags, acts_out, N, k);
  else if (w_dtype == CLT_F32 && out_dtype == CLT_F32)
    rc = launch_decode_fwd<float, float>(ch, grid, st, idx, v, v_dtype, post_enc, post_scale, W, M, lat_begin, d, bias, out, flags, acts_out, N, k);
  else
    rc = launch_decode_fwd<float, bf16_t>(ch, grid, st, idx, v, v_dtype, post_enc, post_scale, W, M, lat_begin, d, bias, out, flags, acts_out, N, k);
  if (rc != CLT_OK) {
    set_error("clt_decode_fwd: unsupported width d=%d", d);
    return rc;
  }
  CLT_CHECK_LAUNCH("clt_decode_fwd");
  return CLT_OK;
}

template <typename WT, typename GT>
static int launch_decode_bwd(int ch, dim3 grid, hipStream_t st, const int64_t* idx, const void* v, int v_dtype,
                             const float* pe, const float* ps, const void* W, int64_t M, int64_t lat_begin, int d,
                             const void* G, float* ga, int accumulate, float* e_post, float* e_scale, int64_t N,
                             int k) {
#define LAUNCH(CH, U)                                                                                            \
  hipLaunchKernelGGL((decode_bwd_acts_kernel<WT, GT, CH, U>), grid, dim3(256), 0, st, idx, v, v_dtype, pe, ps,   \
                     (const WT*)W, M, lat_begin, d, (const GT*)G, ga, accumulate, e_post, e_scale, N, k)
  switch (ch) {
    case 1: LAUNCH(1, 8); break;
    case 2: LAUNCH(2, 8); break;
    case 4: LAUNCH(4, 4); break;
    case 8: LAUNCH(8, 2); break;
    case 16: LAUNCH(16, 1); break;
    default: return CLT_ERR_UNSUPPORTED;
  }
#undef LAUNCH
  return CLT_OK;
}

extern "C" int clt_decode_bwd_acts(const int64_t* idx, const void* v, int v_dtype, const float* post_enc,
                                   const float* post_scale, const void* W, int w_dtype, int64_t M, int64_t lat_begin,
                                   int d, const void* G, int g_dtype, float* grad_acts, int accumulate_grad_acts,
                                   float* e_post, float* e_scale, int64_t N, int k, clt_stream_t stream) {
  CLT_CHECK_ARG(idx && v && W && G && grad_acts, "clt_decode_bwd_acts: null pointer");
  CLT_CHECK_ARG(N >= 0 && k > 0 && d > 0 && M > 0 && M < (1ll << 31), "clt_decode_bwd_acts: bad sizes");
  CLT_CHECK_ARG(valid_dtype(v_dtype) && valid_dtype(w_dtype) && valid_dtype(g_dtype), "clt_decode_bwd_acts: bad dtype");
  CLT_CHECK_ARG(aligned16(W) && aligned16(G), "clt_decode_bwd_acts: pointers must be 16-byte aligned");
  CLT_CHECK_ARG(!e_scale || post_scale, "clt_decode_bwd_acts: e_scale requires post_scale");
  const int E = w_dtype == CLT_BF16 ? 8 : 4;
  CLT_UNSUPPORTED(d % 8 == 0 || (d % 4 == 0 && w_dtype == CLT_F32 && g_dtype == CLT_F32),
                  "clt_decode_bwd_acts: d=%d must be a multiple of 8 (or 4 for all-fp32)", d);
  const int ch = pick_ch(d, E);
  CLT_UNSUPPORTED(ch > 0, "clt_decode_bwd_acts: d=%d too wide", d);
  if (N == 0) return CLT_OK;
  dim3 grid((unsigned)((N + 3) / 4));
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (w_dtype == CLT_BF16 && g_dtype == CLT_BF16)
    rc = launch_decode_bwd<bf16_t, bf16_t>(ch, grid, st, idx, v, v_dtype, post_enc, post_scale, W, M, lat_begin, d, G, grad_acts, accumulate_grad_acts, e_post, e_scale, N, k);
  else if (w_dtype == CLT_BF16 && g_dtype == CLT_F32)
    rc = launch_decode_bwd<bf16_t, float>(ch, grid, st, idx, v, v_dtype, post_enc, post_scale, W, M, lat_begin, d, G, grad_acts, accumulate_grad_acts, e_post, e_scale, N, k);
  else if (w_dtype == CLT_F32 && g_dtype == CLT_F32)
    rc = launch_decode_bwd<float, float>(ch, grid, st, idx, v, v_dtype, post_enc, post_scale, W, M, lat_begin, d, G, grad_acts, accumulate_grad_acts, e_post, e_scale, N, k);
  else
    rc = launch_decode_bwd<float, bf16_t>(ch, grid, st, idx, v, v_dtype, post_enc, post_scale, W, M, lat_begin, d, G, grad_acts, accumulate_grad_acts, e_post, e_scale, N, k);
  if (rc != CLT_OK) {
    set_error("clt_decode_bwd_acts: unsupported width d=%d", d);
    return rc;
  }
  CLT_CHECK_LAUNCH("clt_decode_bwd_acts");
  return CLT_OK;
}

// ----------------------------------------------------------------------------------------------------------------
// Multi-source variants (CLT fan-out, runner.py:43-71): the entries of all live sources of one target form ONE list
// per token: entry e = s * k + j.  Each lane resolves its entry's source through a copy of the descriptor table in
// LDS, turns it into (row address, weight), entries of other shards are compacted away, and the gather loop no
// longer knows about sources: U rows in flight whatever k / tp is.
namespace clt {

struct SrcTable {
  clt_decode_source s[CLT_DECODE_MAX_SOURCES];
};

__device__ __forceinline__ void stage_table(const SrcTable& tab, int ns, clt_decode_source* lt) {
  // static indices only: the descriptors are read straight from the kernel-argument segment (scalar loads)
#pragma unroll
  for (int i = 0; i < CLT_DECODE_MAX_SOURCES; ++i)
    if (i < ns && threadIdx.x == i) lt[i] = tab.s[i];
  __syncthreads();
}

template <typename WT, typename OT, int CH, int U>
__global__ __launch_bounds__(256) void decode_fwd_multi_kernel(const SrcTable tab, int ns, int v_dtype, int64_t M, int d,
                                                               const float* __restrict__ bias, OT* out, int flags,
                                                               int64_t N, int k) {
  constexpr int E = Chunk<WT>::E;
  __shared__ clt_decode_source lt[CLT_DECODE_MAX_SOURCES];
  stage_table(tab, ns, lt);
  const int lane = threadIdx.x & 63;
  const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;  // wave-uniform, after the only barrier
  const int nchunks = d / E;
  const size_t row_bytes = (size_t)d * sizeof(WT);
  float acc[CH][E];
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int e = 0; e < E; ++e) acc[c][e] = 0.f;

  const int total = ns * k;
  for (int e0 = 0; e0 < total; e0 += 64) {
    const int e = e0 + lane;
    const bool active = e < total;
    bool valid = false;
    float my_a = 0.f;
    unsigned my_lo = 0, my_hi = 0;
    if (active) {
      const int s = e / k, j = e - s * k;
      const clt_decode_source& S = lt[s];
      const int64_t o = n * k + j;
      const int64_t rr = S.idx[o] - S.lat_begin;
      valid = rr >= 0 && rr < M;
      float a = 0.f;
      if (valid) {
        const float vv = load_scalar(S.v, v_dtype, o);
        a = post_affine(vv, (int)rr, S.post_enc, S.post_scale);
        if (flags & CLT_DECODE_ROUND_ACTS) a = bf16_bits_to_f32(f32_to_bf16_bits(a));
        const uint64_t addr = reinterpret_cast<uint64_t>(S.W) + (uint64_t)rr * row_bytes;
        my_lo = (unsigned)addr;
        my_hi = (unsigned)(addr >> 32);
      }
      if (S.acts_out) S.acts_out[o] = a;
      my_a = a * S.weight;
    }
    int cnt = min(64, total - e0);
    if (__ballot(valid) != __ballot(active)) {
      int nv;
      const int dest = compact_dest(valid, active, nv);
      my_lo = (unsigned)permute_push_i(dest, (int)my_lo);
      my_hi = (unsigned)permute_push_i(dest, (int)my_hi);
      my_a = permute_push_f(dest, my_a);
      cnt = nv;
    }
    for (int jj = 0; jj < cnt; jj += U) {
      uint4 raw[U][CH];
      float a_u[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int jl = min(jj + u, cnt - 1);
        const uint64_t addr = ((uint64_t)(unsigned)readlane_i((int)my_hi, jl) << 32) | (unsigned)readlane_i((int)my_lo, jl);
        a_u[u] = (jj + u < cnt) ? readlane_f(my_a, jl) : 0.f;
        const char* row = reinterpret_cast<const char*>(addr);
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int ch = lane + 64 * c;
          raw[u][c] = *reinterpret_cast<const uint4*>(row + (size_t)(ch < nchunks ? ch : nchunks - 1) * 16);  // (unconditional: rows_lds_row, spt_rows.h)
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int ch = lane + 64 * c;
          if (ch < nchunks) {
            float w[E];
            Chunk<WT>::expand(raw[u][c], w);
#pragma unroll
            for (int q = 0; q < E; ++q) acc[c][q] = fmaf(a_u[u], w[q], acc[c][q]);
          }
        }
      }
    }
  }

  OT* orow = out + n * (int64_t)d;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nchunks) {
      if (bias) {
        float b[E];
        load_elems<float, E>(bias + ch * E, b);
#pragma unroll
        for (int q = 0; q < E; ++q) acc[c][q] += b[q];
      }
      if (flags & CLT_DECODE_ACCUMULATE) {
        float o[E];
        load_elems<OT, E>(orow + ch * E, o);
#pragma unroll
        for (int q = 0; q < E; ++q) acc[c][q] += o[q];
      }
      store_elems<OT, E>(orow + ch * E, acc[c]);
    }
  }
}

template <typename WT, typename GT, int CH, int U>
__global__ __launch_bounds__(256) void decode_bwd_acts_multi_kernel(const SrcTable tab, int ns, int v_dtype, int64_t M,
                                                                    int d, const GT* __restrict__ G, int64_t N, int k) {
  constexpr int E = Chunk<WT>::E;
  __shared__ clt_decode_source lt[CLT_DECODE_MAX_SOURCES];
  stage_table(tab, ns, lt);
  const int lane = threadIdx.x & 63;
  const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const int nchunks = d / E;
  const size_t row_bytes = (size_t)d * sizeof(WT);
  float g[CH][E];
  const GT* grow = G + n * (int64_t)d;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nchunks) {
      load_elems<GT, E>(grow + ch * E, g[c]);
    } else {
#pragma unroll
      for (int q = 0; q < E; ++q) g[c][q] = 0.f;
    }
  }

  const int total = ns * k;
  for (int e0 = 0; e0 < total; e0 += 64) {
    const int e = e0 + lane;
    const bool active = e < total;
    bool valid = false;
    int my_r = 0;
    unsigned my_lo = 0, my_hi = 0;
    if (active) {
      const int s = e / k, j = e - s * k;
      const clt_decode_source& S = lt[s];
      const int64_t rr = S.idx[n * k + j] - S.lat_begin;
      valid = rr >= 0 && rr < M;
      if (valid) {
        my_r = (int)rr;
        const uint64_t addr = reinterpret_cast<uint64_t>(S.W) + (uint64_t)rr * row_bytes;
        my_lo = (unsigned)addr;
        my_hi = (unsigned)(addr >> 32);
      }
    }
    int my_e = e;  // the entry this lane reports on (changes under compaction)
    int cnt = min(64, total - e0);
    if (__ballot(valid) != __ballot(active)) {
      int nv;
      const int dest = compact_dest(valid, active, nv);
      my_lo = (unsigned)permute_push_i(dest, (int)my_lo);
      my_hi = (unsigned)permute_push_i(dest, (int)my_hi);
      my_r = permute_push_i(dest, my_r);
      my_e = permute_push_i(dest, e);
      valid = lane < nv;
      cnt = nv;
    }
    float my_ga = 0.f;
    for (int jj = 0; jj < cnt; jj += U) {
      uint4 raw[U][CH];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int jl = min(jj + u, cnt - 1);
        const uint64_t addr = ((uint64_t)(unsigned)readlane_i((int)my_hi, jl) << 32) | (unsigned)readlane_i((int)my_lo, jl);
        const char* row = reinterpret_cast<const char*>(addr);
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int ch = lane + 64 * c;
          raw[u][c] = *reinterpret_cast<const uint4*>(row + (size_t)(ch < nchunks ? ch : nchunks - 1) * 16);  // (unconditional: rows_lds_row, spt_rows.h)
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float part = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int ch = lane + 64 * c;
          if (ch < nchunks) {
            float w[E];
            Chunk<WT>::expand(raw[u][c], w);
#pragma unroll
            for (int q = 0; q < E; ++q) part = fmaf(g[c][q], w[q], part);
          }
        }
        part = wave_sum_dpp(part);
        if (jj + u < cnt && lane == jj + u) my_ga = part;
      }
    }
    if (my_e < total) {  // every entry of the chunk is reported by exactly one lane
      const int s = my_e / k, j = my_e - s * k;
      const clt_decode_source& S = lt[s];
      const int64_t o = n * k + j;
      float u_ = 0.f, es = 0.f, ep = 0.f;
      if (valid) {
        const float ga = my_ga * S.weight;
        const float vv = load_scalar(S.v, v_dtype, o);
        const float pos = vv > 0.f ? 1.f : 0.f;
        u_ = S.post_scale ? ga * S.post_scale[my_r] : ga;
        es = ga * (vv + (S.post_enc ? S.post_enc[my_r] * pos : 0.f));
        ep = u_ * pos;
      }
      if (S.e_scale) S.e_scale[o] = es;
      if (S.e_post) S.e_post[o] = ep;
      if (S.accumulate_grad_acts) {
        if (valid) S.grad_acts[o] += u_;
      } else {
        S.grad_acts[o] = u_;
      }
    }
  }
}

// clt_decode_fwd_multi with the gathered rows staged in LDS by LDS-DMA instead of landing in registers (bf16 tables, rows
// of at most CH KiB).  MI355X_MICROARCH.md ('Indexed rows: gather into LDS') reads random rows faster through
// global_load_lds_dwordx4 than through global_load_dwordx4 into registers; here every lane reads back exactly the bytes
// it asked for (lane-linear landing: lane l's chunk of piece c sits at slot + c * 1024 + 16 l), so LDS is only a place
// for loads in flight -- RS rows of CH KiB per wave, in two halves: the rows of batch b + 1 are requested before the
// rows of batch b are consumed (counted vmcnt wait).  Same arithmetic, in the same order, as decode_fwd_multi_kernel.
// rows in flight per wave: 4 waves * RS rows * (at most CH KiB) stays at or under 64 KiB a block, two blocks a CU (at
// C2's 1536-byte rows 4 to 10 rows measured the same, 12 and 16 -- one block a CU -- 25 % slower)
__host__ __device__ constexpr int lds_rows_per_wave(int pieces) { return pieces <= 2 ? 8 : pieces <= 4 ? 4 : 2; }

__device__ __forceinline__ void lds_dma_row_piece(const char* src, uint32_t lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(src), "s"(lds_dst)
      : "memory");
}

template <typename OT, int CH>
__global__ __launch_bounds__(256) void decode_fwd_multi_lds_kernel(const SrcTable tab, int ns, int v_dtype, int64_t M, int d,
                                                                   const float* __restrict__ bias, OT* out, int flags,
                                                                   int64_t N, int k) {
  typedef bf16_t WT;
  constexpr int E = Chunk<WT>::E;
  constexpr int RS = lds_rows_per_wave(CH);
  constexpr int H = RS / 2;  // rows per batch
  __shared__ clt_decode_source lt[CLT_DECODE_MAX_SOURCES];
  extern __shared__ __attribute__((aligned(1024))) char stage[];  // [4 waves][RS rows][SLOT = row bytes]
  stage_table(tab, ns, lt);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t n = (int64_t)blockIdx.x * 4 + wave;
  if (n >= N) return;  // wave-uniform, after the only barrier
  const int nchunks = d / E;
  const size_t row_bytes = (size_t)d * sizeof(WT);
  const int SLOT = (int)row_bytes;  // piece c of a row lands at slot + c * 1024: the last piece is the short one
  char* my = stage + (size_t)wave * RS * SLOT;
  const uint32_t my_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)my;
  float acc[CH][E];
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int e = 0; e < E; ++e) acc[c][e] = 0.f;

  const int total = ns * k;
  for (int e0 = 0; e0 < total; e0 += 64) {
    const int e = e0 + lane;
    const bool active = e < total;
    bool valid = false;
    float my_a = 0.f;
    unsigned my_lo = 0, my_hi = 0;
    if (active) {
      const int s = e / k, j = e - s * k;
      const clt_decode_source& S = lt[s];
      const int64_t o = n * k + j;
      const int64_t rr = S.idx[o] - S.lat_begin;
      valid = rr >= 0 && rr < M;
      float a = 0.f;
      if (valid) {
        const float vv = load_scalar(S.v, v_dtype, o);
        a = post_affine(vv, (int)rr, S.post_enc, S.post_scale);
        if (flags & CLT_DECODE_ROUND_ACTS) a = bf16_bits_to_f32(f32_to_bf16_bits(a));
        const uint64_t addr = reinterpret_cast<uint64_t>(S.W) + (uint64_t)rr * row_bytes;
        my_lo = (unsigned)addr;
        my_hi = (unsigned)(addr >> 32);
      }
      if (S.acts_out) S.acts_out[o] = a;
      my_a = a * S.weight;
    }
    int cnt = min(64, total - e0);
    if (__ballot(valid) != __ballot(active)) {
      int nv;
      const int dest = compact_dest(valid, active, nv);
      my_lo = (unsigned)permute_push_i(dest, (int)my_lo);
      my_hi = (unsigned)permute_push_i(dest, (int)my_hi);
      my_a = permute_push_f(dest, my_a);
      cnt = nv;
    }
    // (the scalar loads above are done: their results fed the address computation; the counted waits below see DMAs only.
    // Requesting the NEXT group's index and activation here, to have them in flight under this group's rows, was measured
    // in round 3: 413 -> 416 us per launch at C2, no gain -- twelve waves per CU hide the header already.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    auto issue = [&](int jj0, int half) {
#pragma unroll
      for (int u = 0; u < H; ++u) {
        const int jl = min(jj0 + u, cnt - 1);  // (rows past the end repeat the last one: every batch issues H * CH loads)
        const uint64_t addr = ((uint64_t)(unsigned)readlane_i((int)my_hi, jl) << 32) | (unsigned)readlane_i((int)my_lo, jl);
        const char* row = reinterpret_cast<const char*>(addr);
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int ch = lane + 64 * c;
          // (lanes past the row's last chunk sit out; the wave still issues the instruction -- CH is the exact piece
          // count, so every piece has at least one lane -- and a batch is H * CH loads for the counted wait)
          if (ch < nchunks) lds_dma_row_piece(row + (size_t)ch * 16, my_lds + (uint32_t)((half * H + u) * SLOT + c * 1024));
        }
      }
    };
    if (cnt > 0) issue(0, 0);
    int half = 0;
    for (int jj = 0; jj < cnt; jj += H, half ^= 1) {
      const bool more = jj + H < cnt;
      if (more) {
        issue(jj + H, half ^ 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(H * CH) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      // every LDS read of the batch first (lanes past the row's last chunk re-read the slot's first bytes), then the FMAs:
      // under `if (ch < nchunks)` each read sat in a block of its own, in ONE register quad, with its wait right behind it
      uint4 raw[H][CH];
#pragma unroll
      for (int u = 0; u < H; ++u)
#pragma unroll
        for (int c = 0; c < CH; ++c)
          raw[u][c] = *reinterpret_cast<const uint4*>(my + (half * H + u) * SLOT + (lane + 64 * c < nchunks ? c * 1024 + lane * 16 : 0));
#pragma unroll
      for (int u = 0; u < H; ++u) {
        const float a_u = (jj + u < cnt) ? readlane_f(my_a, min(jj + u, cnt - 1)) : 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int ch = lane + 64 * c;
          if (ch < nchunks) {
            float w[E];
            Chunk<WT>::expand(raw[u][c], w);
#pragma unroll
            for (int q = 0; q < E; ++q) acc[c][q] = fmaf(a_u, w[q], acc[c][q]);
          }
        }
      }
      // (the next batch's DMAs into this half are issued after these reads have returned: their data fed the FMAs)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }

  OT* orow = out + n * (int64_t)d;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nchunks) {
      if (bias) {
        float b[E];
        load_elems<float, E>(bias + ch * E, b);
#pragma unroll
        for (int q = 0; q < E; ++q) acc[c][q] += b[q];
      }
      if (flags & CLT_DECODE_ACCUMULATE) {
        float o[E];
        load_elems<OT, E>(orow + ch * E, o);
#pragma unroll
        for (int q = 0; q < E; ++q) acc[c][q] += o[q];
      }
      store_elems<OT, E>(orow + ch * E, acc[c]);
    }
  }
}

template <typename OT>
static void launch_decode_fwd_multi_lds(int pieces, dim3 grid, hipStream_t st, const SrcTable& tab, int ns, int v_dtype,
                                        int64_t M, int d, const float* bias, void* out, int flags, int64_t N, int k) {
#define LAUNCH(CH)                                                                                                 \
  hipLaunchKernelGGL((decode_fwd_multi_lds_kernel<OT, CH>), grid, dim3(256), (size_t)4 * lds_rows_per_wave(CH) * d * 2, \
                     st, tab, ns, v_dtype, M, d, bias, (OT*)out, flags, N, k)
  switch (pieces) {  // CH = the 1 KiB pieces of a row, exactly: every batch is H * CH loads for the counted wait
    case 1: LAUNCH(1); break;
    case 2: LAUNCH(2); break;
    case 3: LAUNCH(3); break;
    case 4: LAUNCH(4); break;
    case 5: LAUNCH(5); break;
    case 6: LAUNCH(6); break;
    case 7: LAUNCH(7); break;
    default: LAUNCH(8); break;
  }
#undef LAUNCH
}

static bool decode_fwd_lds_route(const clt_decode_source* hs, int ns, int v_dtype, int w_dtype, int64_t M, int d,
                                 const float* bias, void* out, int out_dtype, int flags, int64_t N, int k, hipStream_t st) {
  const int pieces = (d / 8 + 63) / 64;
  // (CLT_DECODE_LDS=0 in a -DCLT_TUNING build keeps the register gathers for A/B runs)
  if (w_dtype != CLT_BF16 || d % 8 != 0 || pieces > 8 || (out_dtype != CLT_BF16 && out_dtype != CLT_F32) ||
      !tuning_env("CLT_DECODE_LDS", 1))
    return false;
  SrcTable tab = {};
  for (int i = 0; i < ns; ++i) tab.s[i] = hs[i];
  const dim3 grid((unsigned)((N + 3) / 4));
  if (out_dtype == CLT_BF16)
    launch_decode_fwd_multi_lds<bf16_t>(pieces, grid, st, tab, ns, v_dtype, M, d, bias, out, flags, N, k);
  else
    launch_decode_fwd_multi_lds<float>(pieces, grid, st, tab, ns, v_dtype, M, d, bias, out, flags, N, k);
  return true;
}

static int check_sources(const char* name, const clt_decode_source* hs, int ns, bool bwd) {
  CLT_CHECK_ARG(hs && ns >= 1, "%s: no sources", name);
  CLT_UNSUPPORTED(ns <= CLT_DECODE_MAX_SOURCES, "%s: %d sources > CLT_DECODE_MAX_SOURCES", name, ns);
  for (int i = 0; i < ns; ++i) {
    CLT_CHECK_ARG(hs[i].idx && hs[i].v && hs[i].W && aligned16(hs[i].W), "%s: source %d: null or misaligned pointer", name, i);
    if (bwd) {
      CLT_CHECK_ARG(hs[i].grad_acts, "%s: source %d: grad_acts is null", name, i);
      CLT_CHECK_ARG(!hs[i].e_scale || hs[i].post_scale, "%s: source %d: e_scale requires post_scale", name, i);
    }
  }
  return CLT_OK;
}

}  // namespace clt

template <typename WT, typename OT>
static int launch_decode_fwd_multi(int ch, dim3 grid, hipStream_t st, const SrcTable& tab, int ns, int v_dtype, int64_t M,
                                   int d, const float* bias, void* out, int flags, int64_t N, int k) {
#define LAUNCH(CH, U)                                                                                                \
  hipLaunchKernelGGL((decode_fwd_multi_kernel<WT, OT, CH, U>), grid, dim3(256), 0, st, tab, ns, v_dtype, M, d, bias, \
                     (OT*)out, flags, N, k)
  switch (ch) {
    case 1: LAUNCH(1, 8); break;
    case 2: LAUNCH(2, 8); break;
    case 4: LAUNCH(4, 4); break;
    case 8: LAUNCH(8, 2); break;
    case 16: LAUNCH(16, 1); break;
    default: return CLT_ERR_UNSUPPORTED;
  }
#undef LAUNCH
  return CLT_OK;
}

extern "C" int clt_decode_fwd_multi(const clt_decode_source* h_sources, int n_sources, int v_dtype, int w_dtype,
                                    int64_t M, int d, const float* bias, void* out, int out_dtype, int flags, int64_t N,
                                    int k, clt_stream_t stream) {
  int rc = check_sources("clt_decode_fwd_multi", h_sources, n_sources, false);
  if (rc != CLT_OK) return rc;
  CLT_CHECK_ARG(out, "clt_decode_fwd_multi: null pointer");
  CLT_CHECK_ARG(N >= 0 && k > 0 && d > 0 && M > 0 && M < (1ll << 31) && (int64_t)n_sources * k < (1ll << 30),
                "clt_decode_fwd_multi: bad sizes N=%lld k=%d d=%d M=%lld", (long long)N, k, d, (long long)M);
  CLT_CHECK_ARG(valid_dtype(v_dtype) && valid_dtype(w_dtype) && valid_dtype(out_dtype), "clt_decode_fwd_multi: bad dtype");
  CLT_CHECK_ARG(aligned16(out) && (!bias || aligned16(bias)), "clt_decode_fwd_multi: pointers must be 16-byte aligned");
  const int E = w_dtype == CLT_BF16 ? 8 : 4;
  CLT_UNSUPPORTED(d % 8 == 0 || (d % 4 == 0 && w_dtype == CLT_F32 && out_dtype == CLT_F32),
                  "clt_decode_fwd_multi: d=%d must be a multiple of 8 (or 4 for all-fp32)", d);
  const int ch = pick_ch(d, E);
  CLT_UNSUPPORTED(ch > 0, "clt_decode_fwd_multi: d=%d too wide", d);
  if (N == 0) return CLT_OK;
  SrcTable tab = {};
  for (int i = 0; i < n_sources; ++i) tab.s[i] = h_sources[i];
  dim3 grid((unsigned)((N + 3) / 4));
  hipStream_t st = (hipStream_t)stream;
  if (decode_fwd_lds_route(h_sources, n_sources, v_dtype, w_dtype, M, d, bias, out, out_dtype, flags, N, k, st)) {
    CLT_CHECK_LAUNCH("clt_decode_fwd_multi");
    return CLT_OK;
  }
  if (w_dtype == CLT_BF16 && out_dtype == CLT_BF16)
    rc = launch_decode_fwd_multi<bf16_t, bf16_t>(ch, grid, st, tab, n_sources, v_dtype, M, d, bias, out, flags, N, k);
  else if (w_dtype == CLT_BF16 && out_dtype == CLT_F32)
    rc = launch_decode_fwd_multi<bf16_t, float>(ch, grid, st, tab, n_sources, v_dtype, M, d, bias, out, flags, N, k);
  else if (w_dtype == CLT_F32 && out_dtype == CLT_F32)
    rc = launch_decode_fwd_multi<float, float>(ch, grid, st, tab, n_sources, v_dtype, M, d, bias, out, flags, N, k);
  else
    rc = launch_decode_fwd_multi<float, bf16_t>(ch, grid, st, tab, n_sources, v_dtype, M, d, bias, out, flags, N, k);
  if (rc != CLT_OK) {
    set_error("clt_decode_fwd_multi: unsupported width d=%d", d);
    return rc;
  }
  CLT_CHECK_LAUNCH("clt_decode_fwd_multi");
  return CLT_OK;
}

template <typename WT, typename GT>
static int launch_decode_bwd_multi(int ch, dim3 grid, hipStream_t st, const SrcTable& tab, int ns, int v_dtype, int64_t M,
                                   int d, const void* G, int64_t N, int k) {
#define LAUNCH(CH, U)                                                                                               \
  hipLaunchKernelGGL((decode_bwd_acts_multi_kernel<WT, GT, CH, U>), grid, dim3(256), 0, st, tab, ns, v_dtype, M, d, \
                     (const GT*)G, N, k)
  switch (ch) {
    case 1: LAUNCH(1, 8); break;
    case 2: LAUNCH(2, 8); break;
    case 4: LAUNCH(4, 4); break;
    case 8: LAUNCH(8, 2); break;
    case 16: LAUNCH(16, 1); break;
    default: return CLT_ERR_UNSUPPORTED;
  }
#undef LAUNCH
  return CLT_OK;
}

extern "C" int clt_decode_bwd_acts_multi(const clt_decode_source* h_sources, int n_sources, int v_dtype, int w_dtype,
                                         int64_t M, int d, const void* G, int g_dtype, int64_t N, int k,
                                         clt_stream_t stream) {
  int rc = check_sources("clt_decode_bwd_acts_multi", h_sources, n_sources, true);
  if (rc != CLT_OK) return rc;
  CLT_CHECK_ARG(G && aligned16(G), "clt_decode_bwd_acts_multi: G is null or misaligned");
  CLT_CHECK_ARG(N >= 0 && k > 0 && d > 0 && M > 0 && M < (1ll << 31) && (int64_t)n_sources * k < (1ll << 30),
                "clt_decode_bwd_acts_multi: bad sizes");
  CLT_CHECK_ARG(valid_dtype(v_dtype) && valid_dtype(w_dtype) && valid_dtype(g_dtype), "clt_decode_bwd_acts_multi: bad dtype");
  const int E = w_dtype == CLT_BF16 ? 8 : 4;
  CLT_UNSUPPORTED(d % 8 == 0 || (d % 4 == 0 && w_dtype == CLT_F32 && g_dtype == CLT_F32),
                  "clt_decode_bwd_acts_multi: d=%d must be a multiple of 8 (or 4 for all-fp32)", d);
  const int ch = pick_ch(d, E);
  CLT_UNSUPPORTED(ch > 0, "clt_decode_bwd_acts_multi: d=%d too wide", d);
  if (N == 0) return CLT_OK;
  SrcTable tab = {};
  for (int i = 0; i < n_sources; ++i) tab.s[i] = h_sources[i];
  dim3 grid((unsigned)((N + 3) / 4));
  hipStream_t st = (hipStream_t)stream;
  if (w_dtype == CLT_BF16 && g_dtype == CLT_BF16)
    rc = launch_decode_bwd_multi<bf16_t, bf16_t>(ch, grid, st, tab, n_sources, v_dtype, M, d, G, N, k);
  else if (w_dtype == CLT_BF16 && g_dtype == CLT_F32)
    rc = launch_decode_bwd_multi<bf16_t, float>(ch, grid, st, tab, n_sources, v_dtype, M, d, G, N, k);
  else if (w_dtype == CLT_F32 && g_dtype == CLT_F32)
    rc = launch_decode_bwd_multi<float, float>(ch, grid, st, tab, n_sources, v_dtype, M, d, G, N, k);
  else
    rc = launch_decode_bwd_multi<float, bf16_t>(ch, grid, st, tab, n_sources, v_dtype, M, d, G, N, k);
  if (rc != CLT_OK) {
    set_error("clt_decode_bwd_acts_multi: unsupported width d=%d", d);
    return rc;
  }
  CLT_CHECK_LAUNCH("clt_decode_bwd_acts_multi");
  return CLT_OK;
}
