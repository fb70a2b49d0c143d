// Micro-benchmark: issue rate of v_mfma_f32_32x32x16_bf16 from ONE wave per SIMD in the register pattern of the
// sign-weight convolution (4 accumulators, hi then lo products, 8 MFMAs per tap), in s_memtime ticks and wall time.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(const float* in, float* out, unsigned long long* ticks, int iters) {
  bf16x8 a[2], b[4];
  for (int i = 0; i < 2; ++i)
    for (int q = 0; q < 8; ++q) a[i][q] = (__bf16)in[(threadIdx.x + i * 64 + q) & 1023];
  for (int i = 0; i < 4; ++i)
    for (int q = 0; q < 8; ++q) b[i][q] = (__bf16)in[(threadIdx.x * 3 + i * 64 + q) & 1023];
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
      for (int m = 0; m < 8; ++m)
        acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(m >> 1) & 1], b[(m & 1) + 2 * (m >> 2)], acc[m % NACC], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < NACC; ++i)
    for (int q = 0; q < 16; ++q) s += acc[i][q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int NACC, int WAVES>
void run(const char* name, int blocks) {
  float *in, *out;
  unsigned long long* ticks;
  hipMalloc(&in, 4096);
  std::vector<float> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xFFFF) / 65536.f - 0.5f;
  hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
  hipMalloc(&out, blocks * 64 * WAVES * 4);
  hipMalloc(&ticks, blocks * 8);
  const int iters = 200;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, in, out, ticks, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, in, out, ticks, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> t(blocks);
  hipMemcpy(t.data(), ticks, blocks * 8, hipMemcpyDeviceToHost);
  const double n_mfma = (double)iters * 72;
  const double per_simd = n_mfma * WAVES / 4.0;      // MFMAs each SIMD issues (waves are spread over the 4 SIMDs)
  printf("%-28s blocks %4d: %7.1f ticks / MFMA / wave, wall %8.1f us -> %6.1f ns per MFMA per SIMD, %7.1f TFLOP/s\n", name, blocks,
         (double)t[0] / n_mfma, ms * 1e3, ms * 1e6 / per_simd, blocks * WAVES * n_mfma * 32768.0 / (ms * 1e-3) / 1e12);
}

int main() {
  run<4, 4>("4 acc, 1 wave/SIMD", 256);
  run<4, 8>("4 acc, 2 waves/SIMD", 256);
  run<8, 4>("8 acc, 1 wave/SIMD", 256);
  run<2, 4>("2 acc, 1 wave/SIMD", 256);
  run<4, 4>("4 acc, 1 wave/SIMD, 1 block", 1);
  return 0;
}
