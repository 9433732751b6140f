// Encoder GEMM for gfx950: pre[N, M] = relu(x[N, d] @ W[M, d]^T + b), bf16 or exact-f32 MFMA.
//
// Both operands are K-contiguous ("NT"), so an MFMA fragment is one 16-byte run of a row
// for A and for B alike.  The product is computed TRANSPOSED, D[latent][token] with W as
// the A operand and x as the B operand: in the 32x32 accumulator layout
// (col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)) a lane then owns
// ONE token and runs of 4 consecutive latents, i.e. 8/16 contiguous bytes of a pre row,
// and a per-token threshold test for a fused top-k filter is a per-lane scalar compare.
//
// Tile: 128 latents x 128 tokens x (128 bytes of K) per step, 4 waves as 2 x 2, each wave
// 64 x 64 = 2 x 2 MFMA tiles (v_mfma_f32_32x32x16_bf16 / v_mfma_f32_32x32x2_f32).
// Staging: global_load_lds 16 B/lane straight into a double-buffered LDS image
// (2 x 2 x 16 KiB); the image is lane-linear, the bank swizzle
// (chunk ^= (row >> 1) & 7) is applied to the per-lane SOURCE address and to the
// ds_read_b128 address (cdna_hip_programming.md rule 21).  Out-of-range rows / K tails
// read a 16-byte zero chunk, so every shape with d % 8 == 0 (bf16) or d % 4 == 0 (f32)
// runs on the same path.  Workgroups are renumbered so that each XCD walks a contiguous
// range of (token panel, latent tile) pairs: a panel of 16 token tiles of x stays in the
// XCD's 4 MiB L2 while W tiles stream through.
// MFMA-bound: 2 * N * M * d flops.
#include "common.h"

namespace clt {

__device__ uint4 g_zero_chunk[1];  // 16 zero bytes, source for padding lanes

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

constexpr int TM = 128;   // latents per tile
constexpr int TN = 128;   // tokens per tile
constexpr int ROWB = 128; // bytes of K per row per step
constexpr int TILE_BYTES = TM * ROWB;  // 16 KiB per operand per buffer
constexpr int PANEL = 16;              // token tiles per L2 panel

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// Stage one 128-row x 128-byte operand tile. 16 pieces of 1 KiB; wave w issues pieces w, w+4, w+8, w+12.
// Piece p covers LDS rows 8p .. 8p+7; lane l writes row 8p + l/8, chunk position l%8 and must
// therefore fetch source chunk (l%8) ^ ((row >> 1) & 7) of that row.
__device__ __forceinline__ void stage_tile(const char* __restrict__ base, int64_t row0, int64_t nrows, int64_t ld_bytes,
                                           int kbyte0, int kbytes, char* lds_tile, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = wave + 4 * i;
    const int r = piece * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    const int64_t grow = row0 + r;
    const int kb = kbyte0 + c * 16;
    const char* src = (grow < nrows && kb < kbytes) ? base + grow * ld_bytes + kb
                                                     : reinterpret_cast<const char*>(g_zero_chunk);
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(lds_tile + piece * 1024), 16, 0, 0);
  }
}

__device__ __forceinline__ uint4 lds_frag(const char* lds_tile, int row, int chunk) {
  const int pos = chunk ^ ((row >> 1) & 7);
  return *reinterpret_cast<const uint4*>(lds_tile + row * ROWB + pos * 16);
}

template <typename T>
__device__ __forceinline__ void mma_step(const uint4& a, const uint4& b, f32x16& acc);

template <>
__device__ __forceinline__ void mma_step<bf16_t>(const uint4& a, const uint4& b, f32x16& acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_step<float>(const uint4& a, const uint4& b, f32x16& acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}

// Bijective XCD-aware renumbering (cdna_hip_programming.md, T1): blocks b and b+8 share an XCD;
// give each XCD group a contiguous range of logical ids.
__device__ __forceinline__ int64_t xcd_remap(int64_t orig, int64_t nwg) {
  const int64_t q = nwg / 8, r = nwg % 8, x = orig % 8;
  const int64_t start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return start + orig / 8;
}

template <typename T, typename OT>
__global__ __launch_bounds__(256, 2) void encoder_gemm_kernel(const T* __restrict__ x, const T* __restrict__ W,
                                                               const void* __restrict__ bias, int bias_dtype,
                                                               OT* __restrict__ pre, int64_t N, int64_t M, int d,
                                                               int64_t nlt, int64_t ntt) {
  __shared__ __attribute__((aligned(16))) char lds[2 * 2 * TILE_BYTES];  // [buf][A|B][16 KiB]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // ---- logical tile id -> (token panel, latent tile, token tile)
  const int64_t id = xcd_remap(blockIdx.x, nlt * ntt);
  const int64_t full = nlt * PANEL;
  int64_t p = id / full;
  int64_t rem = id - p * full;
  const int64_t pt = min((int64_t)PANEL, ntt - p * PANEL);
  const int64_t lt = rem / pt, tt = p * PANEL + rem % pt;
  const int64_t lat0 = lt * TM, tok0 = tt * TN;

  const int kbytes = d * (int)sizeof(T);
  const int nsteps = (kbytes + ROWB - 1) / ROWB;
  const int64_t ld_bytes = (int64_t)kbytes;
  const char* Wb = reinterpret_cast<const char*>(W);
  const char* xb = reinterpret_cast<const char*>(x);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  stage_tile(Wb, lat0, M, ld_bytes, 0, kbytes, lds, wave, lane);
  stage_tile(xb, tok0, N, ld_bytes, 0, kbytes, lds + TILE_BYTES, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int r32 = lane & 31, h = lane >> 5;
  for (int t = 0; t < nsteps; ++t) {
    char* cur = lds + (t & 1) * 2 * TILE_BYTES;
    char* nxt = lds + ((t + 1) & 1) * 2 * TILE_BYTES;
    if (t + 1 < nsteps) {
      stage_tile(Wb, lat0, M, ld_bytes, (t + 1) * ROWB, kbytes, nxt, wave, lane);
      stage_tile(xb, tok0, N, ld_bytes, (t + 1) * ROWB, kbytes, nxt + TILE_BYTES, wave, lane);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      uint4 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = lds_frag(cur, wm * 64 + i * 32 + r32, 2 * kk + h);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = lds_frag(cur + TILE_BYTES, wn * 64 + j * 32 + r32, 2 * kk + h);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) mma_step<T>(a[i], b[j], acc[i][j]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue: + bias, ReLU, one rounding, 4 consecutive latents per store
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int64_t tok = tok0 + wn * 64 + j * 32 + r32;
    if (tok >= N) continue;
    OT* prow = pre + tok * M;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int64_t lat = lat0 + wm * 64 + i * 32 + 8 * g + 4 * h;
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float bq = 0.f;
          if (bias && lat + q < M) bq = load_scalar(bias, bias_dtype, lat + q);
          const float s = acc[i][j][4 * g + q] + bq;
          v[q] = s > 0.f ? s : 0.f;
        }
        if (lat + 3 < M && (M & 3) == 0) {
          if constexpr (sizeof(OT) == 2) {
            uint2 u;
            u.x = pack_bf16x2(v[0], v[1]);
            u.y = pack_bf16x2(v[2], v[3]);
            *reinterpret_cast<uint2*>(prow + lat) = u;
          } else {
            *reinterpret_cast<float4*>(prow + lat) = make_float4(v[0], v[1], v[2], v[3]);
          }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (lat + q < M) prow[lat + q] = (OT)v[q];
        }
      }
    }
  }
}

}  // namespace clt

using namespace clt;

extern "C" int clt_encoder_gemm_relu(const void* x, const void* W, int in_dtype, const void* bias, int bias_dtype,
                                     void* pre, int out_dtype, int64_t N, int64_t M, int d, clt_stream_t stream) {
  CLT_CHECK_ARG(x && W && pre, "clt_encoder_gemm_relu: null pointer");
  CLT_CHECK_ARG(N >= 0 && M > 0 && d > 0 && M < (1ll << 31), "clt_encoder_gemm_relu: bad sizes");
  CLT_CHECK_ARG(valid_dtype(in_dtype) && valid_dtype(out_dtype) && (!bias || valid_dtype(bias_dtype)),
                "clt_encoder_gemm_relu: bad dtype");
  CLT_CHECK_ARG(aligned16(x) && aligned16(W) && aligned16(pre), "clt_encoder_gemm_relu: pointers must be 16-byte aligned");
  CLT_UNSUPPORTED((d * (int)dtype_size(in_dtype)) % 16 == 0,
                  "clt_encoder_gemm_relu: d=%d must give rows that are a multiple of 16 bytes", d);
  if (N == 0) return CLT_OK;
  const int64_t nlt = (M + TM - 1) / TM, ntt = (N + TN - 1) / TN;
  CLT_UNSUPPORTED(nlt * ntt < (1ll << 31), "clt_encoder_gemm_relu: too many tiles");
  dim3 grid((unsigned)(nlt * ntt));
  hipStream_t st = (hipStream_t)stream;
  if (in_dtype == CLT_BF16 && out_dtype == CLT_BF16)
    hipLaunchKernelGGL((encoder_gemm_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)W,
                       bias, bias_dtype, (bf16_t*)pre, N, M, d, nlt, ntt);
  else if (in_dtype == CLT_BF16 && out_dtype == CLT_F32)
    hipLaunchKernelGGL((encoder_gemm_kernel<bf16_t, float>), grid, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)W,
                       bias, bias_dtype, (float*)pre, N, M, d, nlt, ntt);
  else if (in_dtype == CLT_F32 && out_dtype == CLT_F32)
    hipLaunchKernelGGL((encoder_gemm_kernel<float, float>), grid, dim3(256), 0, st, (const float*)x, (const float*)W,
                       bias, bias_dtype, (float*)pre, N, M, d, nlt, ntt);
  else
    hipLaunchKernelGGL((encoder_gemm_kernel<float, bf16_t>), grid, dim3(256), 0, st, (const float*)x, (const float*)W,
                       bias, bias_dtype, (bf16_t*)pre, N, M, d, nlt, ntt);
  CLT_CHECK_LAUNCH("clt_encoder_gemm_relu");
  return CLT_OK;
}

extern "C" size_t clt_encode_topk_workspace_bytes(int64_t N, int64_t M, int out_dtype) {
  if (N <= 0 || M <= 0) return 256;
  return align_up((size_t)N * (size_t)M * dtype_size(out_dtype), 256);
}

extern "C" int clt_encode_topk_fwd(const void* x, const void* W, int in_dtype, const void* bias, int bias_dtype,
                                   int64_t N, int64_t M, int d, int k, void* out_vals, int out_dtype,
                                   int64_t* out_idx, void* workspace, size_t workspace_bytes, clt_stream_t stream) {
  CLT_CHECK_ARG(workspace, "clt_encode_topk_fwd: null workspace");
  if (workspace_bytes < clt_encode_topk_workspace_bytes(N, M, out_dtype)) {
    set_error("clt_encode_topk_fwd: workspace too small (%zu < %zu)", workspace_bytes,
              clt_encode_topk_workspace_bytes(N, M, out_dtype));
    return CLT_ERR_WORKSPACE;
  }
  int rc = clt_encoder_gemm_relu(x, W, in_dtype, bias, bias_dtype, workspace, out_dtype, N, M, d, stream);
  if (rc != CLT_OK) return rc;
  return clt_topk_rows(workspace, out_dtype, M, nullptr, nullptr, N, M, k, out_vals, out_dtype, out_idx, stream);
}
