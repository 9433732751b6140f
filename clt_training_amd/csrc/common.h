// Shared helpers for the gfx950 kernels of libcltmi355.so.
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/clt_mi355.h"

#define CLT_WAVE 64

typedef __bf16 bf16_t;

namespace clt {

// Tuning switches (timing ablations, A/B kernels) are read from the environment ONLY in builds made with -DCLT_TUNING
// (python -m clt_training_amd.build --tuning -> libcltmi355_tuning.so); the shipped library never reads the environment.
static inline int tuning_env(const char* name, int dflt) {
#ifdef CLT_TUNING
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
#else
  (void)name;
  return dflt;
#endif
}


void set_error(const char* fmt, ...);

#define CLT_CHECK_ARG(cond, ...)         \
  do {                                   \
    if (!(cond)) {                       \
      clt::set_error(__VA_ARGS__);       \
      return CLT_ERR_BAD_ARG;            \
    }                                    \
  } while (0)

#define CLT_UNSUPPORTED(cond, ...)       \
  do {                                   \
    if (!(cond)) {                       \
      clt::set_error(__VA_ARGS__);       \
      return CLT_ERR_UNSUPPORTED;        \
    }                                    \
  } while (0)

#define CLT_CHECK_LAUNCH(name)                                                    \
  do {                                                                            \
    hipError_t e__ = hipGetLastError();                                           \
    if (e__ != hipSuccess) {                                                      \
      clt::set_error("%s: launch failed: %s", name, hipGetErrorString(e__));      \
      return CLT_ERR_LAUNCH;                                                      \
    }                                                                             \
  } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline size_t dtype_size(int dt) { return dt == CLT_BF16 ? 2 : 4; }
static inline bool valid_dtype(int dt) { return dt == CLT_F32 || dt == CLT_BF16; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- device helpers -------------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ float bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
  bf16_t b = (bf16_t)f;  // v_cvt_pk_bf16_f32: round to nearest even, NaN preserving
  return (uint32_t)__builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  return f32_to_bf16_bits(lo) | (f32_to_bf16_bits(hi) << 16);
}

// 16-byte chunk of a row, as 8 bf16 or 4 f32, expanded to floats.
template <typename T>
struct Chunk;
template <>
struct Chunk<bf16_t> {
  static constexpr int E = 8;
  __device__ static __forceinline__ void load(const void* p, float (&f)[8]) {
    uint4 u = *reinterpret_cast<const uint4*>(p);
    f[0] = bf16_lo(u.x); f[1] = bf16_hi(u.x); f[2] = bf16_lo(u.y); f[3] = bf16_hi(u.y);
    f[4] = bf16_lo(u.z); f[5] = bf16_hi(u.z); f[6] = bf16_lo(u.w); f[7] = bf16_hi(u.w);
  }
  __device__ static __forceinline__ void expand(const uint4& u, float (&f)[8]) {
    f[0] = bf16_lo(u.x); f[1] = bf16_hi(u.x); f[2] = bf16_lo(u.y); f[3] = bf16_hi(u.y);
    f[4] = bf16_lo(u.z); f[5] = bf16_hi(u.z); f[6] = bf16_lo(u.w); f[7] = bf16_hi(u.w);
  }
  __device__ static __forceinline__ void store(void* p, const float (&f)[8]) {
    uint4 u;
    u.x = pack_bf16x2(f[0], f[1]); u.y = pack_bf16x2(f[2], f[3]);
    u.z = pack_bf16x2(f[4], f[5]); u.w = pack_bf16x2(f[6], f[7]);
    *reinterpret_cast<uint4*>(p) = u;
  }
};
template <>
struct Chunk<float> {
  static constexpr int E = 4;
  __device__ static __forceinline__ void load(const void* p, float (&f)[4]) {
    float4 u = *reinterpret_cast<const float4*>(p);
    f[0] = u.x; f[1] = u.y; f[2] = u.z; f[3] = u.w;
  }
  __device__ static __forceinline__ void expand(const uint4& u, float (&f)[4]) {
    f[0] = __uint_as_float(u.x); f[1] = __uint_as_float(u.y);
    f[2] = __uint_as_float(u.z); f[3] = __uint_as_float(u.w);
  }
  __device__ static __forceinline__ void store(void* p, const float (&f)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
  }
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Non-temporal 16-byte accesses for streams that are touched once per launch (optimizer state, gradient
// outputs): they do not displace the lines that ARE reused (gathered rows) from L2 / Infinity Cache.
__device__ __forceinline__ void nt_load8(const bf16_t* p, float (&f)[8]) {
  const u32x4 u = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  f[0] = bf16_lo(u.x); f[1] = bf16_hi(u.x); f[2] = bf16_lo(u.y); f[3] = bf16_hi(u.y);
  f[4] = bf16_lo(u.z); f[5] = bf16_hi(u.z); f[6] = bf16_lo(u.w); f[7] = bf16_hi(u.w);
}
__device__ __forceinline__ void nt_store8(bf16_t* p, const float (&f)[8]) {
  u32x4 u;
  u.x = pack_bf16x2(f[0], f[1]); u.y = pack_bf16x2(f[2], f[3]);
  u.z = pack_bf16x2(f[4], f[5]); u.w = pack_bf16x2(f[6], f[7]);
  __builtin_nontemporal_store(u, reinterpret_cast<u32x4*>(p));
}

__device__ __forceinline__ float load_scalar(const void* p, int dtype, int64_t i) {
  if (dtype == CLT_BF16) return bf16_bits_to_f32(reinterpret_cast<const uint16_t*>(p)[i]);
  return reinterpret_cast<const float*>(p)[i];
}

// load_scalar in two halves, for code that wants several such loads IN FLIGHT together: the aligned 32-bit word that holds
// element i (one load instruction whatever the dtype), and its conversion.  load_scalar's own dtype branch puts each load
// in a block of its own with the conversion right behind it, so hipcc waits for every load before it issues the next.
// (A bf16 array needs no more than its natural 2-byte alignment: the word is taken at the element's address rounded
// down, so at most two bytes before the first or after the last element are touched -- inside any allocation granule.)
__device__ __forceinline__ uint32_t load_scalar_word(const void* p, int dtype, int64_t i) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(p) + (uintptr_t)(dtype == CLT_BF16 ? i * 2 : i * 4);
  return *reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3);
}
__device__ __forceinline__ float scalar_from_word(uint32_t w, const void* p, int dtype, int64_t i) {
  const bool hi = ((reinterpret_cast<uintptr_t>(p) >> 1) + (uintptr_t)i) & 1;
  return __uint_as_float(dtype == CLT_BF16 ? (hi ? (w & 0xffff0000u) : (w << 16)) : w);
}

__device__ __forceinline__ int readlane_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float readlane_f(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// a * b rounded to fp32 on its own: never contracted into an fma with a following add (hipcc contracts by default;
// __fmul_rn is a plain product in this toolchain).  For sums that must match a product stored by another kernel.
__device__ __forceinline__ float mul_rounded(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// One element of torch.optim.Adam's update (trainer.py:178-186; exp_avg.lerp_, exp_avg_sq.mul_().addcmul_(),
// param.addcdiv_()), written with explicit roundings: no contraction is left to the compiler, so every kernel that
// applies it (clt_adam_step, clt_spt_dense_adam, clt_decode_bwd_rows_adam) produces the same bits.
// The square root and the quotient are the hardware's 1-ulp v_sqrt_f32 / v_rcp_f32: the correctly rounded forms hipcc
// emits for sqrtf and '/' are ~10 instructions each, and the row passes are issue-bound (SQ counters).
// The update term they feed is ~1e-3 of the parameter, so the parameter moves by < 1e-10 relative.
__device__ __forceinline__ void adam_update(float& p, float& m, float& v, float g, float step_size, float b1, float b2,
                                            float inv_sqrt_bc2, float eps) {
#pragma clang fp contract(off)
  m = fmaf(g - m, 1.f - b1, m);
  v = fmaf((1.f - b2) * g, g, b2 * v);
  const float denom = fmaf(__builtin_amdgcn_sqrtf(v), inv_sqrt_bc2, eps);
  p = fmaf(-step_size, m * __builtin_amdgcn_rcpf(denom), p);
}

// Wave sum without the LDS crossbar: four DPP row shifts leave each row-of-16 total in its lane 15, the four row
// totals are then added in a fixed order.  ~6x shorter dependency chain than six ds_bpermute steps.  Needs all 64
// lanes active (call it in wave-uniform control flow).  The summation ORDER differs from wave_sum: kernels whose
// results are compared bit for bit must use the same one.
template <int CTRL>
__device__ __forceinline__ float dpp_add_(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v = dpp_add_<0x111>(v);  // row_shr:1 (lanes shifted in from outside the row read 0)
  v = dpp_add_<0x112>(v);  // row_shr:2
  v = dpp_add_<0x114>(v);  // row_shr:4
  v = dpp_add_<0x118>(v);  // row_shr:8
  const float a = readlane_f(v, 15), b = readlane_f(v, 31), c = readlane_f(v, 47), d = readlane_f(v, 63);
  return ((a + b) + c) + d;
}

}  // namespace clt
