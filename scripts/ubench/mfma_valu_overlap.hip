// Micro-benchmark: does VALU work issued between integer MFMAs hide behind them on gfx950?
// Each wave runs a chain of v_mfma_i32_32x32x32_i8 on two accumulators with N independent v_and_b32 after every
// MFMA; 256 CUs x (1 or 2) waves per SIMD.  Overlap: time per MFMA = max(32, 4 N) cycles; no overlap: 32 + 4 N.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_valu_overlap.hip -o /tmp/mfma_valu_overlap && /tmp/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int N, bool DEP>
__global__ void k(int* out, int iters) {
  v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, 7, (int)blockIdx.x};
  v16i c0 = {}, c1 = {};
  unsigned t[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) t[i] = threadIdx.x * 7 + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (DEP) {   // the MFMA's B operand is written by the VALU instructions right before it (as in the conv kernel)
#pragma unroll
        for (int i = 0; i < 4 && i < N; ++i) asm volatile("v_and_b32 %0, %1, %2" : "=v"(b[i]) : "v"(t[i]), "v"(0x01010101u << (i + u)));
#pragma unroll
        for (int i = 4; i < N; ++i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(t[i]) : "v"(0xFFFFFF0Fu));
      } else {
#pragma unroll
        for (int i = 0; i < N; ++i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(t[i]) : "v"(0xFFFFFF0Fu));
      }
      if (u & 1) c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
      else c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
    }
  }
  int s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
#pragma unroll
  for (int i = 0; i < 12; ++i) s += t[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int N, bool DEP>
void run(int* d, int threads) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<N, DEP>), dim3(256), dim3(threads), 0, 0, d, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<N, DEP>), dim3(256), dim3(threads), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = 4.0 * iters * (threads / 256);
  printf("N=%2d %s waves/SIMD=%d: %8.3f ms  %6.1f ns per MFMA per SIMD\n", N, DEP ? "dep  " : "indep", threads / 256, ms,
         ms * 1e6 / mfma_per_simd);
}

int main() {
  int* d;
  hipMalloc(&d, 256 * 512 * 4);
  for (int threads : {256, 512}) {
    run<0, false>(d, threads);
    run<2, false>(d, threads);
    run<4, false>(d, threads);
    run<8, false>(d, threads);
    run<12, false>(d, threads);
    run<4, true>(d, threads);
    run<8, true>(d, threads);
  }
  return 0;
}
