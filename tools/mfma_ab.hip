// A/B of the two dense bf16 MFMA shapes of gfx950 as BARE loops (no loads in the loop, one wave per SIMD, 256
// accumulator registers each): v_mfma_f32_32x32x16_bf16 (16 independent accumulators of 16 registers) against
// v_mfma_f32_16x16x32_bf16 (64 independent accumulators of 4 registers), random operands, a few hundred ms of
// sustained load so that the clock settles.  Same flop count per launch.  Prints wall time and PFLOP/s per shape;
// run it under `rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES` for the cycle counts (clock = cycles / time).
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_ab tools/mfma_ab.hip && tools/mfma_ab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define CHECK(x)                                                                     \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                        \
      exit(1);                                                                       \
    }                                                                                \
  } while (0)

template <int KIND>
__global__ __launch_bounds__(256, 1) void mfma_loop(const uint4* __restrict__ in, float* __restrict__ out, int iters) {
  const int t = threadIdx.x;
  if constexpr (KIND == 0) {
    bf16x8 A[4], B[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      A[i] = __builtin_bit_cast(bf16x8, in[t + 256 * i]);
      B[i] = __builtin_bit_cast(bf16x8, in[t + 256 * (4 + i)]);
    }
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 2; ++rep)  // 2 x 16 MFMAs of 32 x 32 x 16 = one 32-wide k-step of a 128 x 128 wave tile
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i], B[j], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[blockIdx.x * 256 + t] = s;
  } else {
    bf16x8 A[8], B[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      A[i] = __builtin_bit_cast(bf16x8, in[t + 256 * i]);
      B[i] = __builtin_bit_cast(bf16x8, in[t + 256 * (8 + i)]);
    }
    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i)  // 64 MFMAs of 16 x 16 x 32 = the same 32-wide k-step of a 128 x 128 wave tile
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[i], B[j], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) s += acc[i][j][e];
    out[blockIdx.x * 256 + t] = s;
  }
}

// Two waves per SIMD, 128 accumulator registers each (the wave tile of gemm_w8_filter_kernel: 128 x 64 = 8 x 4 MFMA
// tiles of 16 x 16 x 32); BAR = 1 adds one s_barrier per 32 MFMAs (per k-step), BAR = 2 a barrier and 45 "tiles" of
// iters / 45 k-steps with the accumulators re-initialised by an MFMA in between.
template <int BAR>
__global__ __launch_bounds__(512, 1) void mfma_loop8(const uint4* __restrict__ in, float* __restrict__ out, int iters) {
  const int t = threadIdx.x & 255;
  bf16x8 A[8], B[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) A[i] = __builtin_bit_cast(bf16x8, in[t + 256 * i]);
#pragma unroll
  for (int i = 0; i < 4; ++i) B[i] = __builtin_bit_cast(bf16x8, in[t + 256 * (8 + i)]);
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[i], B[j], acc[i][j], 0, 0, 0);
    if (BAR) __builtin_amdgcn_s_barrier();
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) s += acc[i][j][e];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

// The tile structure of the filter pass as a bare loop: 24 k-steps of 32 MFMAs + barrier, then an "epilogue" of EPI
// dependent-free VALU instructions per accumulator (a stand-in for the threshold scan) during which a wave issues no MFMA.
// NT = 512: eight waves in lockstep (both waves of a SIMD reach the epilogue together: the MFMA pipe idles);
// NT = 256 with two workgroups per CU: the two waves of a SIMD belong to different workgroups and drift apart.
template <int NT, int EPI>
__global__ __launch_bounds__(NT, 512 / NT) void mfma_tiles(const uint4* __restrict__ in, float* __restrict__ out, int tiles) {
  const int t = threadIdx.x & 255;
  bf16x8 A[8], B[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) A[i] = __builtin_bit_cast(bf16x8, in[t + 256 * i]);
#pragma unroll
  for (int i = 0; i < 4; ++i) B[i] = __builtin_bit_cast(bf16x8, in[t + 256 * (8 + i)]);
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
  float keep = 0.f;
#pragma unroll 1
  for (int tile = 0; tile < tiles; ++tile) {
#pragma unroll 1
    for (int it = 0; it < 24; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[i], B[j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_barrier();
    }
    // epilogue: EPI VALU instructions per accumulator register quad, then the accumulators restart
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float m = fmaxf(fmaxf(acc[i][j][0], acc[i][j][1]), fmaxf(acc[i][j][2], acc[i][j][3]));
#pragma unroll
        for (int e = 0; e < EPI; ++e) m = __builtin_fmaf(m, 1.0001f, keep);
        keep = m * 1e-30f;
        acc[i][j] = f32x4{keep, keep, keep, keep};
      }
  }
  out[blockIdx.x * NT + threadIdx.x] = keep;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;  // k-steps per wave and launch
  const int reps = argc > 2 ? atoi(argv[2]) : 12;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const size_t n_in = 256 * 16;
  uint4* h = (uint4*)malloc(n_in * sizeof(uint4));
  srand(1);
  for (size_t i = 0; i < n_in * 8; ++i) {  // bf16 values in [-1, 1): sign, exponent 118..126, random mantissa
    const uint16_t sign = (rand() & 1) << 15, ex = (uint16_t)(118 + rand() % 9) << 7, man = rand() & 127;
    ((uint16_t*)h)[i] = sign | ex | man;
  }
  uint4* din;
  float* dout;
  CHECK(hipMalloc(&din, n_in * sizeof(uint4)));
  CHECK(hipMalloc(&dout, (size_t)cus * 1024 * sizeof(float)));
  CHECK(hipMemcpy(din, h, n_in * sizeof(uint4), hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  // flops per launch: CUs x 4 waves x iters x (128 x 128 x 32 x 2)
  const double flop = (double)cus * 4 * iters * (128.0 * 128 * 32 * 2);
  const char* names[4] = {"32x32x16", "16x16x32", "16x16x32 two waves per SIMD", "16x16x32 two waves per SIMD + barrier per k-step"};
  for (int kind = 0; kind < 4; ++kind) {
    for (int r = 0; r < reps; ++r) {
      CHECK(hipEventRecord(e0, 0));
      if (kind == 0) hipLaunchKernelGGL((mfma_loop<0>), dim3(cus), dim3(256), 0, 0, din, dout, iters);
      else if (kind == 1) hipLaunchKernelGGL((mfma_loop<1>), dim3(cus), dim3(256), 0, 0, din, dout, iters);
      else if (kind == 2) hipLaunchKernelGGL((mfma_loop8<0>), dim3(cus), dim3(512), 0, 0, din, dout, iters);
      else hipLaunchKernelGGL((mfma_loop8<1>), dim3(cus), dim3(512), 0, 0, din, dout, iters);
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= reps - 3)
        printf("%s  launch %2d  %.3f ms  %.3f PFLOP/s\n", names[kind], r, ms, flop / ms / 1e12);
    }
  }
  // tile structure: 8 waves in lockstep against 2 x 4 waves per CU, same work per wave (iters / 24 tiles)
  const int tiles = iters / 24;
  const double flop_t = (double)cus * 8 * tiles * 24 * (128.0 * 64 * 32 * 2);
  for (int kind = 0; kind < 4; ++kind) {
    for (int r = 0; r < reps; ++r) {
      CHECK(hipEventRecord(e0, 0));
      if (kind == 0) hipLaunchKernelGGL((mfma_tiles<512, 20>), dim3(cus), dim3(512), 0, 0, din, dout, tiles);
      else if (kind == 1) hipLaunchKernelGGL((mfma_tiles<256, 20>), dim3(2 * cus), dim3(256), 0, 0, din, dout, tiles);
      else if (kind == 2) hipLaunchKernelGGL((mfma_tiles<512, 10>), dim3(cus), dim3(512), 0, 0, din, dout, tiles);
      else hipLaunchKernelGGL((mfma_tiles<256, 10>), dim3(2 * cus), dim3(256), 0, 0, din, dout, tiles);
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= reps - 2)
        printf("tiles, %s, epilogue %2d VALU per accumulator quad  launch %2d  %.3f ms  %.3f PFLOP/s\n",
               (kind & 1) ? "2 x 4 waves per CU" : "8 waves in lockstep", kind < 2 ? 20 : 10, r, ms, flop_t / ms / 1e12);
    }
  }
  return 0;
}
