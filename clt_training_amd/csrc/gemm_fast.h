// Internal interface between encoder_gemm.hip (ABI entry points, generic kernel) and encoder_gemm_fast.hip
// (the bf16 kernel of the hot path).
#pragma once
#include "common.h"

namespace clt {

constexpr int GEMM_TM = 256;   // latents per tile
constexpr int GEMM_TN = 256;   // tokens per tile
constexpr int GEMM_PANEL = 8;  // token tiles per L2 panel

struct TileCoord {
  int64_t lat0, tok0;
};

// Tile id -> coordinates.  Ids run panel by panel (GEMM_PANEL token tiles wide); inside a panel the token tile
// is the fast index, so the workgroups of an XCD that work on neighbouring ids share the panel's x rows in L2
// and stream each W tile once per panel.
__host__ __device__ __forceinline__ TileCoord tile_of(int64_t id, int64_t nlt, int64_t ntt) {
  const int64_t full = nlt * GEMM_PANEL;
  const int64_t p = id / full;
  const int64_t rem = id - p * full;
  const int64_t left = ntt - p * GEMM_PANEL;
  const int64_t pt = left < GEMM_PANEL ? left : GEMM_PANEL;
  TileCoord c;
  c.lat0 = (rem / pt) * GEMM_TM;
  c.tok0 = (p * GEMM_PANEL + rem % pt) * GEMM_TN;
  return c;
}

// true when the fast kernel takes this problem (bf16 operands, K rows a multiple of 64 bytes and at least 4
// k-steps, bias chunks that cannot straddle the end of the array)
bool gemm_fast_supported(int in_dtype, int64_t N, int64_t M, int d);

// pre[N, M] = relu(x @ W^T + b); same contract as clt_encoder_gemm_relu.  dbg: timing experiments only.
int launch_gemm_fast(const void* x, const void* W, const void* bias, int bias_dtype, void* pre, int out_dtype,
                     int64_t N, int64_t M, int d, int grid, int dbg, hipStream_t st);

}  // namespace clt
