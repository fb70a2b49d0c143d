// Micro-benchmark: the tap loop of the sign-weight convolution kernels in isolation -- per tap 8 x
// v_mfma_f32_32x32x16_bf16 (four accumulators, two passes) fed by 6 x ds_read_b128 issued one tap ahead.
// Variants: fragment reads on/off, conflict-free or patch-like addresses, 1 or 2 waves per SIMD, 1 or 2 workgroups per CU.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/tap_loop.hip -o /tmp/tap_loop && /tmp/tap_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { if ((x) != hipSuccess) { printf("%s failed\n", #x); return; } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
union Frag { bf16x8 v; uint4 q; };

template <int READS, int LDSKB>
__global__ __launch_bounds__(256, 2) void taps(float* out, unsigned long long* cyc, int iters, int stride) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDSKB * 1024];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < LDSKB * 64; i += blockDim.x) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u);
  __syncthreads();
  const int col = lane & 31, kh8 = lane >> 5;
  int xa[9][2], wa[2];
  for (int t = 0; t < 9; ++t)
    for (int j = 0; j < 2; ++j) {
      const int row = ((tid >> 6) * 64 + j * 32 + col) * stride + (t / 3) * 30 + t % 3;      // patch-like: consecutive entries
      const int p = (row & ~15) | ((row & 3) << 2) | ((row >> 2) & 3);
      xa[t][j] = (p * 32 + ((kh8 ^ ((p >> 3) & 1)) << 4)) & (16 * 1024 - 16);
    }
  for (int i = 0; i < 2; ++i) { const int r = i * 32 + col; wa[i] = 40960 + r * 32 + ((kh8 ^ ((r >> 3) & 1)) << 4); }
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  Frag wf[2][2], xh[2][2], xl[2][2];
  for (int s = 0; s < 2; ++s) for (int i = 0; i < 2; ++i) { wf[s][i].q = make_uint4(0x3F803F80u, 0, 0, 0); xh[s][i].q = wf[s][i].q; xl[s][i].q = wf[s][i].q; }
  auto load = [&](int t, int s) {
    if (READS) {
#pragma unroll
      for (int i = 0; i < 2; ++i) wf[s][i].q = *reinterpret_cast<const uint4*>(smem + t * 2048 + wa[i]);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        xh[s][j].q = *reinterpret_cast<const uint4*>(smem + xa[t][j]);
        xl[s][j].q = *reinterpret_cast<const uint4*>(smem + 16896 + xa[t][j]);
      }
    }
  };
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    load(0, 0);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if (t + 1 < 9) load(t + 1, (t + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      const int s = t & 1;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[s][j].v, wf[s][i].v, acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl[s][j].v, wf[s][i].v, acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float sum = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 16; ++q) sum += acc[i][j][q];
  out[blockIdx.x * blockDim.x + tid] = sum;
  if (lane == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (tid >> 6)] = t1 - t0;
}

template <int READS, int LDSKB>
void run(const char* name, int threads, int grid, int stride) {
  float* out; unsigned long long* cyc;
  CK(hipMalloc(&out, sizeof(float) * grid * threads));
  CK(hipMalloc(&cyc, 8 * grid * 8));
  const int iters = 200;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  taps<READS, LDSKB><<<grid, threads>>>(out, cyc, iters, stride);
  CK(hipEventRecord(e0));
  taps<READS, LDSKB><<<grid, threads>>>(out, cyc, iters, stride);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(grid * (threads / 64));
  CK(hipMemcpy(h.data(), cyc, 8 * h.size(), hipMemcpyDeviceToHost));
  double avg = 0; for (auto v : h) avg += v; avg /= h.size();
  const double mf = 72.0 * iters;
  const double waves_per_simd = (double)grid * (threads / 64) / (256 * 4);
  printf("%-58s %7.1f cyc/MFMA per wave  -> %5.1f cyc/MFMA per SIMD   %.3f ms  %.0f TF\n", name, avg / mf, avg / mf / (waves_per_simd < 1 ? 1 : waves_per_simd),
         ms, 2.0 * 32 * 32 * 16 * mf * grid * (threads / 64) / ms * 1e-9);
  CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
  run<0, 60>("no reads, 1 wave/SIMD (256 WG x 256 thr)", 256, 256, 1);
  run<0, 60>("no reads, 2 waves/SIMD (512 WG x 256 thr)", 256, 512, 1);
  run<1, 60>("6 reads/tap, 1 wave/SIMD", 256, 256, 1);
  run<1, 60>("6 reads/tap, 2 waves/SIMD (2 WG/CU)", 256, 512, 1);
  run<1, 60>("6 reads/tap, stride-2 entries, 2 waves/SIMD", 256, 512, 2);
  return 0;
}
