// Micro-benchmark (round 5): what of a kernel's OUTPUT does the next kernel of the same stream find in the L2 of the XCD that
// wrote it?  256 workgroups (workgroup b runs on XCD b % 8), each writes (or reads) its own chunk; the consumer's workgroup b then
// reads chunk (b + shift) % 256: shift 0 = the chunk its own CU produced, 8 = another CU of the same XCD, 1 = another XCD.
// Decides whether aligning a producer's tiles with the consumer's rows per XCD could keep small activation tensors out of the
// Infinity Cache path.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/l2_retention.hip -o /tmp/l2_retention && /tmp/l2_retention
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(512) void produce(float4* buf, size_t chunk4, float v) {
  float4* p = buf + (size_t)blockIdx.x * chunk4;
  for (size_t i = threadIdx.x; i < chunk4; i += 512) p[i] = make_float4(v, v + 1.f, v + 2.f, (float)i);
}
__global__ __launch_bounds__(512) void touch(const float4* buf, size_t chunk4, float* out) {     // a producer that only READS (clean lines)
  const float4* p = buf + (size_t)blockIdx.x * chunk4;
  float s = 0.f;
  for (size_t i = threadIdx.x; i < chunk4; i += 512) { const float4 v = p[i]; s += v.x + v.w; }
  if (s == 12345.678f) out[blockIdx.x] = s;
}
__global__ __launch_bounds__(512) void consume(const float4* buf, size_t chunk4, int shift, float* out) {
  const float4* p = buf + (size_t)((blockIdx.x + shift) % gridDim.x) * chunk4;
  float s = 0.f;
  for (size_t i = threadIdx.x; i < chunk4; i += 512) { const float4 v = p[i]; s += v.x + v.w; }
  if (s == 12345.678f) out[blockIdx.x] = s;
}

int main() {
  float* out;
  hipMalloc(&out, 4096);
  float4 *buf, *junk;
  const size_t maxb = 256ull << 20;
  hipMalloc(&buf, maxb);
  hipMalloc(&junk, 1ull << 30);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const size_t totals[] = {4ull << 20, 16ull << 20, 25ull << 20, 51ull << 20, 103ull << 20};
  printf("consumer time in us after a WRITING / a READING producer of the same stream (median of 9); rows: bytes of the tensor\n");
  printf("%10s | %28s | %28s | %s\n", "tensor", "written: shift 0 / 8 / 1", "read: shift 0 / 8 / 1", "after 1 GiB of other traffic: shift 0");
  for (size_t total : totals) {
    const size_t chunk4 = total / 256 / 16;
    float res[7];
    int col = 0;
    for (int prod = 0; prod < 3; ++prod)
      for (int shift : {0, 8, 1}) {
        if (prod == 2 && shift != 0) continue;
        std::vector<float> ts;
        for (int rep = 0; rep < 9; ++rep) {
          if (prod == 0) produce<<<256, 512>>>(buf, chunk4, (float)rep);
          else if (prod == 1) touch<<<256, 512>>>(buf, chunk4, out);
          else { produce<<<256, 512>>>(buf, chunk4, (float)rep); produce<<<256, 512>>>(junk, (1ull << 30) / 256 / 16, 1.f); }
          hipEventRecord(a);
          consume<<<256, 512>>>(buf, chunk4, shift, out);
          hipEventRecord(b);
          hipEventSynchronize(b);
          float ms;
          hipEventElapsedTime(&ms, a, b);
          ts.push_back(ms * 1e3f);
        }
        std::sort(ts.begin(), ts.end());
        res[col++] = ts[4];
      }
    printf("%7zu MB | %8.1f %8.1f %8.1f   | %8.1f %8.1f %8.1f   | %8.1f\n", total >> 20, res[0], res[1], res[2], res[3], res[4], res[5], res[6]);
  }
  return 0;
}
