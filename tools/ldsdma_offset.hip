#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__global__ void k(const uint32_t* src, uint32_t* out) {
  __shared__ __attribute__((aligned(1024))) uint32_t lds[2048];  // 8 KiB
  for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  const uint32_t base = (uint32_t)(uintptr_t)(lds_ptr_t)lds;
  uint32_t voff = threadIdx.x * 16;
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %1 offset:1024\n\t"
      "s_waitcnt vmcnt(0)"
      :
      : "v"(voff), "s"(src), "s"(base + 2048)
      : "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += 64) out[i] = lds[i];
}
int main() {
  uint32_t h[4096], *d, *o, r[2048];
  for (int i = 0; i < 4096; ++i) h[i] = i;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
  hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  int first = -1, n = 0;
  for (int i = 0; i < 2048; ++i) if (r[i] != 0xdeadbeefu) { if (first < 0) first = i; ++n; }
  printf("first written dword %d (byte %d), count %d, value there %u (= source dword; source byte %u)\n", first, first * 4, n, r[first], r[first] * 4);
  return 0;
}
