// Developer micro-benchmark: GPU-side cost of a kernel launch as a function of workgroup size and LDS
// allocation (empty kernels; hipEvent time per launch over a back-to-back train, and the shader-clock
// lifetime of a workgroup).   hipcc --offload-arch=gfx950 -O3 launch_overhead.hip -o launch_overhead
#include <hip/hip_runtime.h>
#include <cstdio>

template <int LDS_BYTES>
__global__ void empty_kernel(int* out) {
  __shared__ unsigned char smem[LDS_BYTES > 0 ? LDS_BYTES : 4];
  if (out && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) out[0] = smem[0];
}

template <int LDS_BYTES>
float time_it(int blocks, int threads, int iters) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(empty_kernel<LDS_BYTES>, dim3(blocks), dim3(threads), 0, 0, nullptr);
  hipEventRecord(a, 0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(empty_kernel<LDS_BYTES>, dim3(blocks), dim3(threads), 0, 0, nullptr);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / iters;
}

int main() {
  const int it = 2000;
  printf("us per launch, back-to-back train of %d empty kernels\n", it);
  printf("256 x 1024 thr, LDS 156 KB : %6.2f\n", time_it<156 * 1024>(256, 1024, it));
  printf("256 x 1024 thr, LDS  92 KB : %6.2f\n", time_it<92 * 1024>(256, 1024, it));
  printf("256 x 1024 thr, LDS   0    : %6.2f\n", time_it<0>(256, 1024, it));
  printf("1024 x 256 thr, LDS   0    : %6.2f\n", time_it<0>(1024, 256, it));
  printf("256 x  256 thr, LDS   0    : %6.2f\n", time_it<0>(256, 256, it));
  printf("12544 x 256 thr, LDS 17 KB : %6.2f\n", time_it<17 * 1024>(12544, 256, it));
  printf("1 x 64 thr                 : %6.2f\n", time_it<0>(1, 64, it));
  return 0;
}
