// Encoder GEMM for gfx950: pre[N, M] = relu(x[N, d] @ W[M, d]^T + b), bf16 or exact-f32 MFMA.
//
// Both operands are K-contiguous ("NT"), so an MFMA fragment is one 16-byte run of a row
// for A and for B alike.  The product is computed TRANSPOSED, D[latent][token] with W as
// the A operand and x as the B operand: in the 32x32 accumulator layout
// (col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)) a lane then owns
// ONE token and runs of 4 consecutive latents, i.e. 8/16 contiguous bytes of a pre row,
// and a per-token threshold test for a fused top-k filter is a per-lane scalar compare.
//
// Structure (persistent, one 512-thread workgroup per CU):
//   tile 256 latents x 256 tokens x (64 bytes of K per step); 8 waves as 2 (latent) x 4
//   (token), each wave 128 x 64 = 4 x 2 MFMA tiles (v_mfma_f32_32x32x16_bf16 /
//   v_mfma_f32_32x32x2_f32), 128 accumulator VGPRs;
//   staging by global_load_lds, 16 B per lane, into a 4-deep ring of lane-linear LDS images
//   (4 stages x (2 operands x 16 KiB + 1 KiB of bias) = 132 KiB, 64 bytes of K per stage, counted
//   s_waitcnt vmcnt(10): three k-steps of LDS-DMA stay in flight); the bank swizzle chunk ^= (row >> 2) & 3
//   is applied to the per-lane SOURCE address and to the ds_read_b128 address
//   (cdna_hip_programming.md rule 21); rows beyond M / N and K tails read a 16-byte zero
//   chunk, so every shape with 16-byte rows runs on the same path;
//   the (tile, k-step) sequence of a workgroup is flattened: the loads of the NEXT tile's
//   first k-step are already in flight while the current tile's epilogue runs;
//   workgroups are numbered so that the 32 workgroups of an XCD work on neighbouring
//   (token panel, latent tile, token tile) ids: an 8-tile token panel of x (3 MiB) stays in the
//   XCD's 4 MiB L2 while W tiles stream through once per panel.
// MFMA-bound: 2 * N * M * d flops per launch.
#include <stdlib.h>

#include "gemm_fast.h"

namespace clt {

__device__ uint4 g_zero_chunk[1];  // 16 zero bytes, source for padding lanes

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

constexpr int TM = GEMM_TM;  // latents per tile
constexpr int TN = GEMM_TN;  // tokens per tile
constexpr int ROWB = 64;   // bytes of K per row per step
constexpr int OP_BYTES = TM * ROWB;  // 16 KiB per operand per stage
constexpr int NSTAGE = 4;            // LDS ring: 3 k-steps of LDS-DMA in flight behind the one being multiplied
constexpr int BIAS_OFF = 2 * OP_BYTES;      // each stage also carries the tile's 256 bias values (<= 1 KiB)
constexpr int STAGE_BYTES = 2 * OP_BYTES + 1024;
constexpr int NTHREADS = 512;

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// One LDS-DMA of 16 B per lane: LDS[dst + 16 * lane] = *src.  Issued from inline asm so that hipcc
// neither counts it nor drains it: with the builtin, hipcc puts s_waitcnt vmcnt(0) in front of the
// next ds_read (it cannot prove that the DMA target and the tile being read are different buffers),
// which serialises every prefetch (cdna_hip_programming.md, "Pipelining across barriers", 5.7 item 1).
// Completion is waited for by the explicit s_waitcnt vmcnt(0) in front of the k-step barrier.
__device__ __forceinline__ void glds16(const char* src, char* lds_dst) {
  const unsigned dst = (unsigned)(uintptr_t)(lds_ptr_t)lds_dst;
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(src), "s"(dst)
      : "memory");
}

// Stage one k-step of both operand tiles.  An operand stage is 16 pieces of 1 KiB (16 rows x 64 B);
// wave w issues pieces w and w+8 of each operand (4 LDS-DMAs per wave per k-step).  Lane l of a piece
// writes LDS row 16*piece + l/4, chunk position l%4, and therefore fetches source chunk
// (l%4) ^ ((row >> 2) & 3).
constexpr int GLDS_PER_STEP = 5;  // 2 W pieces + 2 x pieces + this wave's 32 bias values
__device__ __forceinline__ void stage_step(const char* __restrict__ Wb, const char* __restrict__ xb,
                                           const char* __restrict__ bias, int bias_esize, TileCoord tc,
                                           int64_t M, int64_t N, int64_t ld_bytes, int kbyte0, int kbytes,
                                           char* buf, int wave, int lane) {
  const char* zero = reinterpret_cast<const char*>(g_zero_chunk);
#pragma unroll
  for (int op = 0; op < 2; ++op) {
    const char* base = op == 0 ? Wb : xb;
    const int64_t row0 = op == 0 ? tc.lat0 : tc.tok0;
    const int64_t nrows = op == 0 ? M : N;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int piece = wave + 8 * i;
      const int r = piece * 16 + (lane >> 2);
      const int c = (lane & 3) ^ ((r >> 2) & 3);
      const int64_t grow = row0 + r;
      const int kb = kbyte0 + c * 16;
      const char* src = (grow < nrows && kb < kbytes) ? base + grow * ld_bytes + kb : zero;
      glds16(src, buf + op * OP_BYTES + piece * 1024);
    }
  }
  // bias of latents [lat0 + 32*wave, +32): 8 (f32) or 4 (bf16) lanes of one exec-masked LDS-DMA.  Every wave
  // issues it on every k-step (even without a bias), so that the counted vmcnt waits are the same everywhere.
  const int nl = 32 * bias_esize / 16;
  if (lane < nl) {
    const int epc = 16 / bias_esize;  // elements per 16-byte chunk
    const int64_t lat = tc.lat0 + 32 * wave + lane * epc;
    const char* src = (bias && lat + epc <= M) ? bias + lat * bias_esize : zero;
    glds16(src, buf + BIAS_OFF + wave * nl * 16);
  }
}

__device__ __forceinline__ uint4 lds_frag(const char* lds_tile, int row, int chunk) {
  const int pos = chunk ^ ((row >> 2) & 3);
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

// Bias of 8 consecutive latents starting at the WAVE-UNIFORM index lat8.  The address is uniform and the
// buffer is const __restrict__, so hipcc emits scalar loads (s_load_dwordx4/x8, lgkmcnt): they neither
// occupy vector registers nor share the vmcnt queue with the LDS-DMA prefetch and the epilogue stores.
__device__ __forceinline__ void load_bias8(const void* __restrict__ bias, int bias_dtype, int64_t lat8, int64_t M,
                                           float (&b)[8]) {
  if (!bias) {
#pragma unroll
    for (int q = 0; q < 8; ++q) b[q] = 0.f;
  } else if (lat8 + 7 < M) {
    if (bias_dtype == CLT_BF16) {
      const uint16_t* p = reinterpret_cast<const uint16_t*>(bias) + lat8;
#pragma unroll
      for (int q = 0; q < 8; ++q) b[q] = bf16_bits_to_f32(p[q]);
    } else {
      const float* p = reinterpret_cast<const float*>(bias) + lat8;
#pragma unroll
      for (int q = 0; q < 8; ++q) b[q] = p[q];
    }
  } else {
#pragma unroll
    for (int q = 0; q < 8; ++q) b[q] = lat8 + q < M ? load_scalar(bias, bias_dtype, lat8 + q) : 0.f;
  }
}

template <typename T, typename OT>
__global__ __launch_bounds__(NTHREADS, 2) void encoder_gemm_kernel(const T* __restrict__ x, const T* __restrict__ W,
                                                                    const void* __restrict__ bias, int bias_dtype,
                                                                    OT* __restrict__ pre, int64_t N, int64_t M, int d,
                                                                    int64_t nlt, int64_t ntt, int dbg) {
  __shared__ __attribute__((aligned(16))) char lds[NSTAGE * STAGE_BYTES];  // ring of [W | x] k-steps
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int r32 = lane & 31, h = lane >> 5;

  // ---- this workgroup's tile ids: XCD group x = blockIdx % 8 owns a contiguous id range;
  //      its workgroups walk it with stride (#workgroups in the group)
  const int64_t ntiles = nlt * ntt;
  const int64_t xg = blockIdx.x & 7, bi = blockIdx.x >> 3;
  const int64_t nbx = (gridDim.x - xg + 7) >> 3;
  const int64_t q = ntiles >> 3, rr = ntiles & 7;
  const int64_t xstart = xg < rr ? xg * (q + 1) : rr * (q + 1) + (xg - rr) * q;
  const int64_t xcount = q + (xg < rr ? 1 : 0);
  const int64_t my_tiles = bi < xcount ? (xcount - bi + nbx - 1) / nbx : 0;
  if (my_tiles == 0) return;

  const int kbytes = d * (int)sizeof(T);
  const int nsteps = (kbytes + ROWB - 1) / ROWB;
  const int64_t ld_bytes = (int64_t)kbytes;
  const char* Wb = reinterpret_cast<const char*>(W);
  const char* xb = reinterpret_cast<const char*>(x);
  const char* bias_b = reinterpret_cast<const char*>(bias);
  const int bias_esize = bias_dtype == CLT_BF16 ? 2 : 4;
  const bool bias_lds = (M & 7) == 0;  // otherwise a 16-byte bias chunk could straddle the end of the array

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // ---- flattened (tile, k-step) sequence with a 4-deep LDS ring
  const int64_t total = my_tiles * nsteps;
  int64_t t_idx = 0;          // tile being multiplied (index among my tiles)
  TileCoord cur = tile_of(xstart + bi, nlt, ntt);
  int ks = 0;                 // its k-step
  int64_t ld_s = 0;           // next (tile, k-step) to stage
  int64_t ld_t = 0;
  int ld_ks = 0;
  TileCoord ld_tc = cur;
  auto issue_next = [&]() {
    if (ld_s < total && !(dbg & 4)) {
      stage_step(Wb, xb, bias_b, bias_esize, ld_tc, M, N, ld_bytes, ld_ks * ROWB, kbytes,
                 lds + (ld_s % NSTAGE) * STAGE_BYTES, wave, lane);
    }
    ++ld_s;
    if (++ld_ks == nsteps) {
      ld_ks = 0;
      ++ld_t;
      if (ld_t < my_tiles) ld_tc = tile_of(xstart + bi + ld_t * nbx, nlt, ntt);
    }
  };
  issue_next();
  issue_next();
  issue_next();

  for (int64_t s = 0; s < total; ++s) {
    // stage s has landed once at most the LDS-DMAs of the (up to two) younger stages are outstanding
    const int64_t younger = min((int64_t)2, total - 1 - s);
    if (younger == 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // every wave's share of stage s is in LDS; every wave is done reading stage s-1
    const bool last_k = ks + 1 == nsteps;
    if (!last_k) issue_next();  // into the slot of stage s-1
    char* cb = lds + (s % NSTAGE) * STAGE_BYTES;
    // ---- MFMA over this k-step
    if (!(dbg & 2))
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint4 a[4], b[2];
      if (dbg & 8) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = make_uint4(lane + i, lane, s, kk);
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = make_uint4(lane, lane + j, s, kk);
      } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = lds_frag(cb, wm * 128 + i * 32 + r32, 2 * kk + h);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = lds_frag(cb + OP_BYTES, wn * 64 + j * 32 + r32, 2 * kk + h);
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) mma_step<T>(a[i], b[j], acc[i][j]);
      __builtin_amdgcn_s_setprio(0);
    }
    // ---- epilogue at the end of a tile: + bias, ReLU, one rounding, 4 consecutive latents per store
    if (last_k) {
      // token-major store order: the 16 stores of one pre row (512 contiguous bytes per wave) go out back to
      // back, so L2 assembles whole lines (interleaving two rows measured 35 % slower for the whole kernel)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int64_t tok = cur.tok0 + wn * 64 + j * 32 + r32;
        OT* prow = pre + tok * M;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int off = wm * 128 + i * 32 + 8 * g;  // wave-uniform latent offset in the tile
            const int64_t lat = cur.lat0 + off + 4 * h;
            float bq[4], v[4];
            if (bias_lds) {
              if (bias_esize == 4) {
                const float4 f = *reinterpret_cast<const float4*>(cb + BIAS_OFF + (off + 4 * h) * 4);
                bq[0] = f.x; bq[1] = f.y; bq[2] = f.z; bq[3] = f.w;
              } else {
                const uint2 u = *reinterpret_cast<const uint2*>(cb + BIAS_OFF + (off + 4 * h) * 2);
                bq[0] = bf16_lo(u.x); bq[1] = bf16_hi(u.x); bq[2] = bf16_lo(u.y); bq[3] = bf16_hi(u.y);
              }
            } else {
              float b8[8];
              load_bias8(bias, bias_dtype, cur.lat0 + off, M, b8);
#pragma unroll
              for (int qq = 0; qq < 4; ++qq) bq[qq] = h ? b8[4 + qq] : b8[qq];
            }
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
              const float sum = acc[i][j][4 * g + qq] + bq[qq];
              v[qq] = sum > 0.f ? sum : 0.f;
              acc[i][j][4 * g + qq] = 0.f;
            }
            if (tok < N && !(dbg & 1)) {
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
                for (int qq = 0; qq < 4; ++qq)
                  if (lat + qq < M) prow[lat + qq] = (OT)v[qq];
              }
            }
          }
        }
      }
      ks = 0;
      ++t_idx;
      if (t_idx < my_tiles) cur = tile_of(xstart + bi + t_idx * nbx, nlt, ntt);
      // the epilogue's stores are older than the LDS-DMAs issued now: the counted waits stay valid
      issue_next();
    } else {
      ++ks;
    }
  }
}

}  // namespace clt

using namespace clt;

static int num_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

extern "C" int clt_encoder_gemm_relu(const void* x, const void* W, int in_dtype, const void* bias, int bias_dtype,
                                     void* pre, int out_dtype, int64_t N, int64_t M, int d, clt_stream_t stream) {
  CLT_CHECK_ARG(x && W && pre, "clt_encoder_gemm_relu: null pointer");
  CLT_CHECK_ARG(N >= 0 && M > 0 && d > 0 && M < (1ll << 31), "clt_encoder_gemm_relu: bad sizes");
  CLT_CHECK_ARG(valid_dtype(in_dtype) && valid_dtype(out_dtype) && (!bias || valid_dtype(bias_dtype)),
                "clt_encoder_gemm_relu: bad dtype");
  CLT_CHECK_ARG(aligned16(x) && aligned16(W) && aligned16(pre) && (!bias || aligned16(bias)),
                "clt_encoder_gemm_relu: pointers must be 16-byte aligned");
  CLT_UNSUPPORTED((d * (int)dtype_size(in_dtype)) % 16 == 0,
                  "clt_encoder_gemm_relu: d=%d must give rows that are a multiple of 16 bytes", d);
  if (N == 0) return CLT_OK;
  const int64_t nlt = (M + TM - 1) / TM, ntt = (N + TN - 1) / TN;
  const int64_t ntiles = nlt * ntt;
  const int64_t grid_n = ntiles < num_cus() ? ntiles : num_cus();
  dim3 grid((unsigned)grid_n);
  hipStream_t st = (hipStream_t)stream;
  const int dbg = tuning_env("CLT_GEMM_DEBUG", 0);  // -DCLT_TUNING builds only: skip stores / MFMA / loads
  if (gemm_fast_supported(in_dtype, N, M, d) && !(dbg & 16)) {  // 16: force the generic kernel
    FastArgs a = {};
    a.x = (const char*)x;
    a.W = (const char*)W;
    a.bias = (const char*)bias;
    a.bias_esize = bias_dtype == CLT_BF16 ? 2 : 4;
    a.pre = pre;
    a.ldp = M;
    a.N = N;
    a.M = M;
    a.kbytes = d * 2;
    a.nlt = (int)nlt;
    a.ntt = (int)ntt;
    a.lt_mode = LT_ALL;
    a.dbg = dbg;
    return launch_gemm_fast(a, out_dtype, false, num_cus(), st);
  }
  if (in_dtype == CLT_BF16 && out_dtype == CLT_BF16)
    hipLaunchKernelGGL((encoder_gemm_kernel<bf16_t, bf16_t>), grid, dim3(NTHREADS), 0, st, (const bf16_t*)x,
                       (const bf16_t*)W, bias, bias_dtype, (bf16_t*)pre, N, M, d, nlt, ntt, dbg);
  else if (in_dtype == CLT_BF16 && out_dtype == CLT_F32)
    hipLaunchKernelGGL((encoder_gemm_kernel<bf16_t, float>), grid, dim3(NTHREADS), 0, st, (const bf16_t*)x,
                       (const bf16_t*)W, bias, bias_dtype, (float*)pre, N, M, d, nlt, ntt, dbg);
  else if (in_dtype == CLT_F32 && out_dtype == CLT_F32)
    hipLaunchKernelGGL((encoder_gemm_kernel<float, float>), grid, dim3(NTHREADS), 0, st, (const float*)x,
                       (const float*)W, bias, bias_dtype, (float*)pre, N, M, d, nlt, ntt, dbg);
  else
    hipLaunchKernelGGL((encoder_gemm_kernel<float, bf16_t>), grid, dim3(NTHREADS), 0, st, (const float*)x,
                       (const float*)W, bias, bias_dtype, (bf16_t*)pre, N, M, d, nlt, ntt, dbg);
  CLT_CHECK_LAUNCH("clt_encoder_gemm_relu");
  return CLT_OK;
}

// Workspace: the pre-activation matrix of the plain path (which is also the fused path's fallback), with the
// fused path's buffers laid over it (sample matrix, packed input, candidate stream and per-token candidate lists: a
// few hundred MB at the BASELINE shapes, but comparable to the matrix itself for a narrow dictionary -- hence the
// margin; a shape whose buffers still do not fit takes the plain path), and a tail that holds the fallback flags
// (gemm_fast.h).
static size_t encode_topk_body_bytes(int64_t N, int64_t M, int out_dtype) {
  const size_t pre = (size_t)N * (size_t)M * dtype_size(out_dtype);
  const size_t extra = pre / 8 > (size_t)N * 8192 ? pre / 8 : (size_t)N * 8192;  // (narrow dictionaries: 8 KiB per token)
  return align_up(pre + extra, 256);
}
extern "C" size_t clt_encode_topk_workspace_bytes(int64_t N, int64_t M, int out_dtype) {
  if (N <= 0 || M <= 0) return 4096 + 256;
  return encode_topk_body_bytes(N, M, out_dtype) + fused_flag_bytes(N);
}

extern "C" int clt_encode_topk_fwd(const void* x, const void* W, int in_dtype, const void* bias, int bias_dtype,
                                   int64_t N, int64_t M, int d, int k, void* out_vals, int out_dtype,
                                   int64_t* out_idx, void* workspace, size_t workspace_bytes, clt_stream_t stream) {
  CLT_CHECK_ARG(workspace, "clt_encode_topk_fwd: null workspace");
  CLT_CHECK_ARG(x && W && out_vals && out_idx, "clt_encode_topk_fwd: null pointer");
  CLT_CHECK_ARG(aligned16(workspace), "clt_encode_topk_fwd: workspace must be 16-byte aligned");
  const size_t need = clt_encode_topk_workspace_bytes(N, M, out_dtype);
  if (workspace_bytes < need) {
    set_error("clt_encode_topk_fwd: workspace too small (%zu < %zu)", workspace_bytes, need);
    return CLT_ERR_WORKSPACE;
  }
  if (N == 0) return CLT_OK;
  const int dbg = tuning_env("CLT_GEMM_DEBUG", 0);  // -DCLT_TUNING builds only; 32: never take the fused path
  if (!(dbg & (16 | 32)) && valid_dtype(in_dtype) && valid_dtype(out_dtype) && aligned16(x) && aligned16(W) &&
      (!bias || (valid_dtype(bias_dtype) && aligned16(bias))) && k >= 1 && k <= M) {
    const FusedPlan pl = fused_plan(in_dtype, out_dtype, N, M, d, k, num_cus());
    const size_t tail = fused_flag_bytes(N);
    if (pl.ok && pl.bytes <= need - tail) {
      char* ws = (char*)workspace;
      return launch_fused_encode_topk(pl, x, W, bias, bias_dtype, N, M, d, k, out_vals, out_idx, ws,
                                      (int*)(ws + need - tail), num_cus(), dbg, (hipStream_t)stream);
    }
  }
  int rc = clt_encoder_gemm_relu(x, W, in_dtype, bias, bias_dtype, workspace, out_dtype, N, M, d, stream);
  if (rc != CLT_OK) return rc;
  return clt_topk_rows(workspace, out_dtype, M, nullptr, nullptr, N, M, k, out_vals, out_dtype, out_idx, stream);
}

extern "C" int clt_encode_topk_fwd_layers(const void* x, int64_t x_row_stride, int64_t x_layer_stride,
                                          const void* const* h_W, int in_dtype, const void* const* h_bias,
                                          int bias_dtype, int n_layers, int64_t N_l, int64_t M, int d, int k,
                                          void* out_vals, int out_dtype, int64_t* out_idx, void* workspace,
                                          size_t workspace_bytes, clt_stream_t stream) {
  CLT_CHECK_ARG(x && h_W && out_vals && out_idx && workspace && aligned16(workspace) && aligned16(x),
                "clt_encode_topk_fwd_layers: null or misaligned pointer");
  CLT_CHECK_ARG(n_layers >= 1 && N_l > 0 && M > 0 && d > 0 && k >= 1 && k <= M && valid_dtype(in_dtype) &&
                    valid_dtype(out_dtype) && (!h_bias || valid_dtype(bias_dtype)),
                "clt_encode_topk_fwd_layers: bad arguments");
  for (int l = 0; l < n_layers; ++l)
    CLT_CHECK_ARG(h_W[l] && aligned16(h_W[l]) && (!h_bias || (h_bias[l] && aligned16(h_bias[l]))),
                  "clt_encode_topk_fwd_layers: layer %d: null or misaligned table / bias", l);
  const int64_t N = N_l * n_layers;
  // what the stacked launch needs: the bf16 fused path, whole 256-token tiles per layer, panels that do not straddle
  // layers, 16-byte aligned x rows; anything else is for the caller to run layer by layer (clt_encode_topk_fwd)
  const int64_t es = dtype_size(in_dtype);
  const FusedPlan pl = fused_plan(in_dtype, out_dtype, N, M, d, k, num_cus());
  const size_t need = clt_encode_topk_workspace_bytes(N, M, out_dtype);
  const size_t tail = fused_flag_bytes(N);
  CLT_UNSUPPORTED(pl.ok && n_layers <= GEMM_MAX_LAYERS && N_l % (GEMM_TN * GEMM_PANEL) == 0 &&
                      (x_row_stride * es) % 16 == 0 && (x_layer_stride * es) % 16 == 0 && x_row_stride >= d &&
                      x_row_stride * es * GEMM_TN < (1ll << 31) && pl.bytes <= need - tail,
                  "clt_encode_topk_fwd_layers: shape not taken by the stacked fused path (bf16, N_l %% %d == 0, <= %d layers)",
                  GEMM_TN * GEMM_PANEL, GEMM_MAX_LAYERS);
  if (workspace_bytes < need) {
    set_error("clt_encode_topk_fwd_layers: workspace too small (%zu < %zu)", workspace_bytes, need);
    return CLT_ERR_WORKSPACE;
  }
  const FusedLayers fl = {n_layers, x_row_stride * es, x_layer_stride * es, h_W, h_bias};
  char* ws = (char*)workspace;
  return launch_fused_encode_topk(pl, x, h_W[0], h_bias ? h_bias[0] : nullptr, bias_dtype, N, M, d, k, out_vals, out_idx,
                                  ws, (int*)(ws + need - tail), num_cus(), 0, (hipStream_t)stream, &fl);
}

