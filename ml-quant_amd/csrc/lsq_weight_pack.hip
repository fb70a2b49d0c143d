// Weight sign packing (eval mode, cached scales) for gfx950.
//
// Replaces the eval branch of WeightQuantizer{LS1,LS2,LST,GF}.forward
// (quant/binary/weight_quantization.py:32-34, :57-58, :80-81, :106-108): plane q of output
// channel o is sign(w - sum_{r<q} u_r[o] * plane_r) with sign(+-0) = +1 (ste.py:16-18).
// Weights are tiny next to activations (1.37 MB packed for ResNet-18) and are packed once
// per eval() session, so this kernel is written for clarity: one thread per (o, tap).
//
// Layout written here and read by the conv kernels:
//   wbits[q][tap][j][o']  uint64, o' = grp * og_pad + (o - grp * og), og_pad = ceil16(O/groups)
//   wsum [q][o][tap]      int32 = sum_c sign(w[o][c][tap])   (border correction of the XNOR conv)

#include "lsq_common.h"

namespace lsq {
namespace {

struct PackArgs {
  const float* w;
  const float* scales;  // [k][O]
  unsigned long long* wbits;
  int* wsum;
  long long plane_words;
  int O, og, og_pad, cg, taps, Gg, k;
};

__global__ __launch_bounds__(256) void pack_weight_kernel(PackArgs a) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= a.O * a.taps) return;
  const int o = idx / a.taps, tap = idx - o * a.taps;
  const int grp = o / a.og;
  const int o_pad = grp * a.og_pad + (o - grp * a.og);
  const int opad_total = (a.O / a.og) * a.og_pad;
  float u[LSQ_MAX_PLANES];
  int sums[LSQ_MAX_PLANES];
#pragma unroll
  for (int q = 0; q < LSQ_MAX_PLANES; ++q) {
    u[q] = q < a.k ? a.scales[(long long)q * a.O + o] : 0.f;
    sums[q] = 0;
  }
  for (int j = 0; j < a.Gg; ++j) {
    unsigned long long words[LSQ_MAX_PLANES];
#pragma unroll
    for (int q = 0; q < LSQ_MAX_PLANES; ++q) words[q] = 0ull;
    const int nch = min(64, a.cg - 64 * j);
    for (int b = 0; b < nch; ++b) {
      const float x = a.w[((long long)o * a.cg + 64 * j + b) * a.taps + tap];
      float result = 0.f;
#pragma unroll
      for (int q = 0; q < LSQ_MAX_PLANES; ++q) {
        if (q < a.k) {
          const bool bit = (x - result) >= 0.f;
          words[q] |= (unsigned long long)bit << b;
          sums[q] += bit ? 1 : -1;
          result = result + (bit ? u[q] : -u[q]);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < LSQ_MAX_PLANES; ++q)
      if (q < a.k) a.wbits[(long long)q * a.plane_words + ((long long)tap * a.Gg + j) * opad_total + o_pad] = words[q];
  }
#pragma unroll
  for (int q = 0; q < LSQ_MAX_PLANES; ++q)
    if (q < a.k) a.wsum[((long long)q * a.O + o) * a.taps + tap] = sums[q];
}

}  // namespace
}  // namespace lsq

using namespace lsq;

extern "C" int64_t lsq_weight_plane_words(const lsq_conv_geom* g) {
  if (check_geom(g)) return -1;
  const int64_t cg = g->C / g->groups, Gg = (cg + 63) / 64;
  const int64_t og = g->O / g->groups, og_pad = (og + 15) / 16 * 16;
  return (int64_t)g->KH * g->KW * Gg * g->groups * og_pad;
}

extern "C" int lsq_pack_weight(const float* w, const lsq_conv_geom* g, int k, const float* scales,
                               uint64_t* wbits, int32_t* wsum, void* stream) {
  if (!w || !scales || !wbits || !wsum) return LSQ_E_NULL;
  if (int e = check_geom(g)) return e;
  if (k < 1 || k > LSQ_MAX_PLANES) return LSQ_E_SCHEME;
  PackArgs a;
  a.w = w;
  a.scales = scales;
  a.wbits = (unsigned long long*)wbits;
  a.wsum = wsum;
  a.plane_words = lsq_weight_plane_words(g);
  a.O = g->O;
  a.og = g->O / g->groups;
  a.og_pad = (a.og + 15) / 16 * 16;
  a.cg = g->C / g->groups;
  a.taps = g->KH * g->KW;
  a.Gg = (a.cg + 63) / 64;
  a.k = k;
  // padded output-channel slots must read as zero words
  hipError_t e = hipMemsetAsync(wbits, 0, (size_t)a.plane_words * k * sizeof(uint64_t), (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  const int total = a.O * a.taps;
  hipLaunchKernelGGL(pack_weight_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
