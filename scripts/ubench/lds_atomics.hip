// Micro-benchmark (round 5): throughput of no-return LDS atomic adds on gfx950, 512-thread workgroups, one per CU -- what paces
// pass 1 of the single-launch quantizer (one 64-bit histogram atomic per sub-sampled key).
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/lds_atomics.hip -o /tmp/lds_atomics && /tmp/lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>   // 0: u64 random bins; 1: u32 random bins; 2: two u32; 3: u64, all lanes of a wave in 8 bins; 4: u32 same; 5: u64 8192 bins sequential per lane (conflict-free)
__global__ __launch_bounds__(512) void k(unsigned* out, int iters) {
  __shared__ unsigned long long h[8192];
  unsigned* h32 = reinterpret_cast<unsigned*>(h);
  for (int i = threadIdx.x; i < 8192; i += 512) h[i] = 0;
  __syncthreads();
  unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int it = 0; it < iters; ++it) {
    x = x * 1664525u + 1013904223u;
    unsigned b = (x >> 12) & 8191u;
    if (MODE == 3 || MODE == 4) b = 4000u + ((x >> 12) & 7u);
    if (MODE == 5) b = (threadIdx.x + it * 512) & 8191u;
    if (MODE == 0 || MODE == 3 || MODE == 5) atomicAdd(&h[b], (1ull << 42) | (x & 8191u));
    if (MODE == 1 || MODE == 4) atomicAdd(&h32[b], x & 8191u);
    if (MODE == 2) { atomicAdd(&h32[b], 1u); atomicAdd(&h32[8192 + b], x & 8191u); }
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = (unsigned)h[blockIdx.x & 8191];
}

template <int MODE>
void run(const char* what) {
  unsigned* out;
  hipMalloc(&out, 4096);
  const int iters = 4096;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k<MODE><<<256, 512>>>(out, iters);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<MODE><<<256, 512>>>(out, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  // per CU: 512 lanes x iters keys
  printf("%-60s %8.1f us  -> %.2f keys per ns per CU (%.1f cycles per wave-instruction at 2.1 GHz, 8 waves)\n", what, ms * 1e3,
         512.0 * iters / (ms * 1e6), ms * 1e-3 * 2.1e9 / (iters * 8.0));
  hipFree(out);
}

int main() {
  run<0>("u64 add, random bins of 8192");
  run<1>("u32 add, random bins of 8192");
  run<2>("two u32 adds (count, low bits), random bins");
  run<3>("u64 add, 8 bins per wave (same-address conflicts)");
  run<4>("u32 add, 8 bins per wave");
  run<5>("u64 add, conflict-free (lane-linear addresses)");
  return 0;
}
