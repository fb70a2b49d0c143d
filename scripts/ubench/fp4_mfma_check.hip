#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

// A[m][k] (32 x 64), B[n][k] (32 x 64) as sign / bit tensors; D[m][n] = sum_k a[m][k] * b[n][k]
// lane (r = lane & 31, h = lane >> 5) holds k-slots h*32 + (4 j + q) for register q, nibble j  -- OUR convention, both operands
__global__ void k(const unsigned* abits /*[32][2] dwords of sign bits*/, const unsigned* bbits /*[32][2] dwords of 0/1*/, float* d, int mode) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const unsigned aw = abits[r * 2 + h], bw = bbits[r * 2 + h];
  v8i a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
  // activation: bit -> nibble code
  if (mode == 0) {        // codes 0x1 (0.5), 0x2 (1), 0x4 (2), 0x1 (0.5)
    b[0] = (int)(bw & 0x11111111u);
    b[1] = (int)(bw & 0x22222222u);
    b[2] = (int)(bw & 0x44444444u);
    b[3] = (int)((bw >> 3) & 0x11111111u);
  } else {                // no subnormal: 1, 1, 2, 2
    b[0] = (int)((bw << 1) & 0x22222222u);
    b[1] = (int)(bw & 0x22222222u);
    b[2] = (int)(bw & 0x44444444u);
    b[3] = (int)((bw >> 1) & 0x44444444u);
  }
  // weights: sign bit s (1 = +1): nibble = (s ? 0 : 8) | mag code
  const unsigned mag0[4] = {0x66666666u, 0x44444444u, 0x22222222u, 0x66666666u};   // 4, 2, 1, 4
  const unsigned mag1[4] = {0x44444444u, 0x44444444u, 0x22222222u, 0x22222222u};   // 2, 2, 1, 1
  for (int q = 0; q < 4; ++q) {
    const unsigned ones = (aw >> q) & 0x11111111u;       // sign bit of channel q + 4 j in nibble j
    const unsigned neg = (ones ^ 0x11111111u) << 3;      // 0x8 where the weight is -1
    a[q] = (int)((mode == 0 ? mag0[q] : mag1[q]) | neg);
  }
  v16f acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  // scale exponents: E8M0 127 = 1.0 in byte 0
  const int sc = 0x7F7F7F7F;
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 4, 4, 0, sc, 0, sc);
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * h;     // m
    d[row * 32 + r] = acc[i];
  }
}
int main() {
  std::vector<unsigned> A(64), B(64);
  srand(1);
  for (auto& x : A) x = (unsigned)rand() * 2654435761u ^ (unsigned)rand();
  for (auto& x : B) x = (unsigned)rand() * 40503u ^ ((unsigned)rand() << 16);
  unsigned *da, *db; float* dd;
  hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 4096);
  hipMemcpy(da, A.data(), 256, hipMemcpyHostToDevice); hipMemcpy(db, B.data(), 256, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd, mode);
    std::vector<float> D(1024);
    hipMemcpy(D.data(), dd, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) {
      int s = 0;
      for (int hh = 0; hh < 2; ++hh) for (int c = 0; c < 32; ++c) {
        const int sa = ((A[m * 2 + hh] >> c) & 1) ? 1 : -1, bb = (B[n * 2 + hh] >> c) & 1;
        s += 2 * sa * bb;
      }
      if (D[m * 32 + n] != (float)s) { if (bad < 5) printf("mode %d m %d n %d got %g want %d\n", mode, m, n, D[m * 32 + n], s); ++bad; }
    }
    printf("mode %d: %d mismatches\n", mode, bad);
  }
  return 0;
}
