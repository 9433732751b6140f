// Issue cost of global_load_lds_dwordx4 among 16x16x32 MFMAs, by address pattern.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int PAT, int NDMA>
__global__ __launch_bounds__(256, 1) void k(const char* src, const uint4* in, float* out, long long* cyc, int iters, int rowstride) {
  __shared__ __attribute__((aligned(1024))) char lds[131072];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  bf16x8 A[8], B[8];
  for (int i = 0; i < 8; ++i) { A[i] = __builtin_bit_cast(bf16x8, in[t + 256 * i]); B[i] = __builtin_bit_cast(bf16x8, in[t + 256 * (8 + i)]); }
  f32x4 acc[8][8];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)lds;
  uint32_t voff;
  if (PAT == 0) voff = (lane >> 2) * rowstride + (lane & 3) * 16;        // 16 rows x 64 B
  else if (PAT == 1) voff = (lane >> 3) * rowstride + (lane & 7) * 16;   // 8 rows x 128 B
  else if (PAT == 2) voff = (lane >> 4) * rowstride + (lane & 15) * 16;  // 4 rows x 256 B
  else voff = lane * 16;                                                 // 1 KiB contiguous
  const char* base = src + (size_t)blockIdx.x * 8192 + wave * 65536;
  long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    const char* sb = base + (it & 15) * 64;
    const uint32_t dst = lds0 + (uint32_t)((it & 3) * 32768 + wave * 8192);
#pragma unroll
    for (int m = 0; m < 64; ++m) {
      acc[m >> 3][m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[m >> 3], B[m & 7], acc[m >> 3][m & 7], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (NDMA > 0 && m % 7 == 2 && m / 7 < NDMA) {
        if (m == 2) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(dst) : "memory");
        asm volatile("global_load_lds_dwordx4 %0, %1 offset:0" ::"v"(voff + (uint32_t)(m / 7) * 1024u * (PAT == 3 ? 1 : 0) + (uint32_t)(m / 7) * 16u * rowstride * (PAT == 3 ? 0 : 1)), "s"(sb) : "memory");
      }
      if (m == 1) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) for (int e = 0; e < 4; ++e) s += acc[i][j][e];
  out[blockIdx.x * 256 + t] = s + lds[t];
  if (t == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int PAT, int NDMA>
void run(const char* name, const char* src, const uint4* in, float* out, long long* cyc, int iters, int cus) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int r = 0; r < 4; ++r) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<PAT, NDMA>), dim3(cus), dim3(256), 0, 0, src, in, out, cyc, iters, 1536);
    hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  }
  long long h[1024]; hipMemcpy(h, cyc, cus * sizeof(long long), hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < cus; ++i) avg += h[i]; avg /= cus;
  printf("%-28s %8.3f ms  %9.1f memtime ticks per step  (%.1f ns per step)\n", name, ms, avg / iters, ms * 1e6 / iters);
}
int main() {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount, iters = 20000;
  char* src; uint4* in; float* out; long long* cyc;
  hipMalloc(&src, 64 << 20); hipMemset(src, 1, 64 << 20);
  uint4* h = (uint4*)malloc(4096 * 16); srand(1);
  for (int i = 0; i < 4096 * 8; ++i) ((uint16_t*)h)[i] = (uint16_t)(((rand() & 1) << 15) | ((118 + rand() % 9) << 7) | (rand() & 127));
  hipMalloc(&in, 4096 * 16); hipMemcpy(in, h, 4096 * 16, hipMemcpyHostToDevice);
  hipMalloc(&out, cus * 256 * 4); hipMalloc(&cyc, 1024 * 8);
  run<0, 0>("no DMA", src, in, out, cyc, iters, cus);
  run<0, 9>("9 DMA 16 rows x 64 B", src, in, out, cyc, iters, cus);
  run<1, 9>("9 DMA 8 rows x 128 B", src, in, out, cyc, iters, cus);
  run<2, 9>("9 DMA 4 rows x 256 B", src, in, out, cyc, iters, cus);
  run<3, 9>("9 DMA 1 KiB contiguous", src, in, out, cyc, iters, cus);
  run<0, 5>("5 DMA 16 rows x 64 B", src, in, out, cyc, iters, cus);
  run<0, 0>("no DMA", src, in, out, cyc, iters, cus);
  return 0;
}
