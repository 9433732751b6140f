// Internal interface between encoder_gemm.hip (ABI entry points, generic kernel) and encoder_gemm_fast.hip
// (the bf16 kernels of the hot path).
#pragma once
#include "common.h"

namespace clt {

constexpr int GEMM_TM = 256;   // latents per tile
constexpr int GEMM_TN = 256;   // tokens per tile
constexpr int GEMM_PANEL = 8;  // token tiles per L2 panel
constexpr int FUSED_SLOTS = 8; // entries per candidate list (64 latents of one token; 32 with the 16x16x32 kernel)
constexpr int GEMM_MAX_LAYERS = 16;

enum { LT_ALL = 0, LT_SAMPLE = 1, LT_REST = 2, LT_GSAMPLE = 3, LT_GREST = 4 };  // (see grp_actual, gemm_fast_dev.h)

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

// Arguments of the fast kernel.  nlt counts VIRTUAL latent tiles: with lt_mode LT_SAMPLE they are every lt_S-th
// 256-row tile of W, with LT_REST all the others.  Store mode writes virtual tile v to columns [256 v, 256 v + 256)
// of pre (leading dimension ldp).  Filter mode reads tau and writes lists / counts (nL = 4 * nlt lists per token).
struct FastArgs {
  const char* x;
  const char* W;
  const char* bias;
  int bias_esize;
  void* pre;
  int64_t ldp;
  int64_t N, M;
  int kbytes;
  int nlt, ntt;
  int lt_mode, lt_S;
  int ngroups;        // 16x16x32 kernels: valid virtual groups of 16 latent rows (0: M / 16, every group of W)
  const float* tau;
  uint32_t* lists;
  uint8_t* counts;
  int nL;
  int qflush;         // filter mode: turn the hit queue into list entries after every quarter tile, not once per tile
  // Eight-wave filter kernel (encoder_gemm_w8.hip): candidates leave as 16-byte entries {4 bf16 values of 4 consecutive
  // latents, token, first latent} in a stream with one region of stream_rcap entries per wave (region = 8 * block +
  // wave); stream_counts[region] = entries the wave produced (more than stream_rcap: the excess was dropped).
  void* stream;
  int* stream_counts;
  int stream_rcap;
  int* fb_flags;      // the fused path's flag area (below): a wave whose queue could not take a group flags its token tile
  // Several layers in one launch (same N_l, M, d; each layer its own W and bias): the layers' tokens are stacked, token
  // tile tt belongs to layer tt / ntt_layer (ntt_layer = N_l / 256; 0 = one layer: W / bias above).  x row t of layer l
  // starts at x + l * x_layer_bytes + t * ldx_bytes (a [N_l, L * d] block of layer inputs side by side: ldx = L * d * 2,
  // x_layer_bytes = d * 2).  Everything else -- tau, lists, counts, pre, outputs -- is indexed by the stacked token id.
  int ntt_layer;
  int64_t ldx_bytes, x_layer_bytes;
  const char* Wl[GEMM_MAX_LAYERS];
  const char* biasl[GEMM_MAX_LAYERS];
  // 16x16x32 kernel only: a packed copy of x, [k-step][token (padded to whole tiles)][64 B], from launch_pack_x; k-step s
  // starts at xp + s * xp_kstep_bytes.  Null: x is read in place.
  const char* xp;
  int64_t xp_kstep_bytes;
  const int* gate;    // if set, the kernel returns at once unless *gate != 0 ...
  const int* tt_map;  // ... and, if tt_map is set too, *gate is the number of token tiles to do and tt_map[i] their ids
  int dbg;
};

// true when the fast kernel takes this problem (bf16 operands, K rows a multiple of 64 bytes, M a multiple of 8)
bool gemm_fast_supported(int in_dtype, int64_t N, int64_t M, int d);
int launch_gemm_fast(FastArgs a, int out_dtype, bool filter, int cus, hipStream_t st);
// The 16x16x32 build of the same kernel (encoder_gemm_fast16.hip) takes the shapes with K a multiple of 64 elements, at
// least 128, and M a multiple of 16; launch_gemm_fast picks it by this predicate, so the launches of one call agree.
bool gemm_fast16_ok(int64_t M, int kbytes);
int launch_gemm_fast16(const FastArgs& a, int out_dtype, int grid, hipStream_t st);
// The filter pass on eight waves, two per SIMD (encoder_gemm_w8.hip): same shapes, same lists, bit-identical accumulators.
int launch_gemm_w8_filter(const FastArgs& a, int grid, hipStream_t st);
// xp[a.kbytes / 64][a.ntt * 256][64 B] <- a.x (a.N, a.kbytes, a.ntt and the stacked-layer fields must be set)
// and zero_ints[0, n_zero) <- 0, zero2[0, n_zero2) <- 0 (both counts <= the launch's thread count)
void launch_pack_x(const FastArgs& a, void* xp, int* zero_ints, int n_zero, int* zero2, int n_zero2, hipStream_t st);

// Fused encode + top-k without the pre-activation matrix:
//   pass A  GEMM over every S-th latent tile, stored [N, lds], and clt_topk_rows over it: per token the best k of the
//           sample and, from its k-th value, a threshold no larger than the k-th value of the whole row;
//   pass B  GEMM over the other tiles whose epilogue keeps only accumulators that reach the token's threshold;
//   merge   per token, sample top-k + candidates -> top-k (same key order as clt_topk_rows);
//   gated fallback: a token whose candidate list overflowed or whose merged top-k failed verification raises the flag of
//           its 256-token tile (device memory); a one-wave launch compacts the flagged tile ids, and the plain GEMM +
//           top-k launches that follow redo ONLY those token tiles -- or return at once when there is none.  No host
//           synchronisation anywhere.  Flag area (ints): [0] number of flagged tiles, [1 .. ntt] per-tile flags,
//           [1 + ntt .. 1 + 2 ntt) the compacted tile ids.
struct FusedPlan {
  bool ok;
  bool l16;         // the 16x16x32 kernels: packed x, eight-wave filter pass with a candidate stream; else the 32x32x16
                    // kernel with candidate lists of 64 latents, 4 per latent tile
  int S, nltA, nltB, ntt, nL;
  int m;            // rank of the sample value that serves as threshold (<= k)
  int qflush;       // see FastArgs
  int64_t lds, Ls;  // sample matrix: leading dimension and valid columns
  // eight-wave filter kernel: stream regions (one per wave of its launch), entries per region / per token list
  int nregions, rcap, tcap;
  int ngA, ngB;     // 16x16x32 kernels: groups of 16 latent rows in the sample / in the filter pass
  size_t off_pre_s, off_svals, off_sidx, off_tau, off_counts, off_lists, off_xp, off_stream, off_scounts, off_tcnt,
      off_tlists, bytes;
};
// cus: the device's compute units (the filter launch's grid, and with it the number of stream regions, follows from it)
FusedPlan fused_plan(int in_dtype, int out_dtype, int64_t N, int64_t M, int d, int k, int cus);
// bytes of the flag area at the end of the clt_encode_topk_fwd workspace
inline size_t fused_flag_bytes(int64_t N) {
  const int64_t ntt = (N + GEMM_TN - 1) / GEMM_TN;
  return align_up((size_t)(1 + 2 * (ntt > 0 ? ntt : 0)) * sizeof(int), 4096);
}
// layers: null for one layer; else n_layers tables / biases and the x strides (FastArgs), N = n_layers * N_l stacked tokens
struct FusedLayers {
  int n_layers;
  int64_t ldx_bytes, x_layer_bytes;
  const void* const* W;
  const void* const* bias;
};
int launch_fused_encode_topk(const FusedPlan& pl, const void* x, const void* W, const void* bias, int bias_dtype,
                             int64_t N, int64_t M, int d, int k, void* out_vals, int64_t* out_idx, char* ws, int* flag,
                             int cus, int dbg, hipStream_t st, const FusedLayers* layers = nullptr);

}  // namespace clt
