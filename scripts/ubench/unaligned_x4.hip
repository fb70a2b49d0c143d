// Micro-test: do dword-aligned (not 16-byte-aligned) global_load_dwordx4 / buffer_load_dwordx4 return the right data on gfx950,
// and what does a buffer load return past num_records?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* x, float* y, int n_floats) {
  const int off = threadIdx.x * 7 + 1;                      // 4-byte aligned only
  const f4 v = *reinterpret_cast<const f4*>(x + off);       // global_load_dwordx4
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, n_floats * 4, 0x00020000);
  const u4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, off * 4, 0, 0);
  for (int i = 0; i < 4; ++i) {
    y[threadIdx.x * 8 + i] = v[i];
    y[threadIdx.x * 8 + 4 + i] = __builtin_bit_cast(float, b[i]);
  }
}
int main() {
  const int n = 64 * 7 + 8;
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = (float)i;
  float *x, *y;
  (void)hipMalloc(&x, n * 4);
  (void)hipMalloc(&y, 64 * 8 * 4);
  (void)hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
  const int limit = 64 * 7 - 2;                             // the last lanes' buffer loads run past num_records
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, x, y, limit);
  std::vector<float> r(64 * 8);
  (void)hipMemcpy(r.data(), y, 64 * 8 * 4, hipMemcpyDeviceToHost);
  int bad_g = 0, bad_b = 0;
  for (int t = 0; t < 64; ++t)
    for (int i = 0; i < 4; ++i) {
      const int idx = t * 7 + 1 + i;
      if (r[t * 8 + i] != (float)idx) ++bad_g;
      const float want = idx < limit ? (float)idx : 0.f;
      if (r[t * 8 + 4 + i] != want) { ++bad_b; printf("lane %d elem %d idx %d: buffer load gave %g, want %g\n", t, i, idx, r[t * 8 + 4 + i], want); }
    }
  printf("unaligned global_load_dwordx4: %d wrong; buffer_load_dwordx4 (with out-of-range tail): %d wrong\n", bad_g, bad_b);
  return 0;
}
