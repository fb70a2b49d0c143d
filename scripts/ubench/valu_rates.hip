// Micro-benchmark: issue rate of the VALU ops the XNOR conv is made of (v_xor_b32, v_bcnt_u32_b32).
//   hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned* out, unsigned seed, int iters) {
  unsigned a[8], acc[8];
  for (int i = 0; i < 8; ++i) { a[i] = seed * (threadIdx.x + 1 + i); acc[i] = i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(acc[i]) : "v"(a[i]));
        if (MODE == 1) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[i]) : "v"(a[i]));
        if (MODE == 2) asm volatile("v_add_u32 %0, %1, %0" : "+v"(acc[i]) : "v"(a[i]));
        if (MODE == 3) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(acc[i]) : "v"(a[i]));
      }
    }
  }
  unsigned s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name) {
  unsigned* out;
  const int blocks = 256 * 8, iters = 4096;
  hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t s, e;
  hipEventCreate(&s); hipEventCreate(&e);
  k<MODE><<<blocks, 256>>>(out, 12345u, 16);
  hipEventRecord(s);
  k<MODE><<<blocks, 256>>>(out, 12345u, iters);
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  const double ops = (double)blocks * 256 * iters * 64;     // lane-ops
  printf("%-16s %8.3f ms  %7.2f T lane-ops/s\n", name, ms, ops / ms / 1e9);
  hipFree(out);
}

int main() {
  run<0>("v_xor_b32");
  run<1>("v_bcnt_u32_b32");
  run<2>("v_add_u32");
  run<3>("v_fma_f32");
  return 0;
}
