// Shared device helpers for the gfx950 least-squares quantization kernels.
// Wavefront = 64 lanes everywhere (CDNA4); no portability layer.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lsq_hip.h"

namespace lsq {

constexpr int kWave = 64;

__device__ __forceinline__ float clamp_sym(float x, float alpha) {
  // torch.clamp(x, -alpha, alpha); a negative alpha encodes clamp_identity
  return alpha >= 0.f ? fminf(fmaxf(x, -alpha), alpha) : x;
}

__device__ __forceinline__ unsigned abs_key(float x) {
  // order-preserving integer key of |x| (IEEE-754 bits without the sign)
  return __float_as_uint(x) & 0x7FFFFFFFu;
}

__device__ __forceinline__ float key_value(unsigned key) { return __uint_as_float(key); }

// ---- wave-level scans / reductions on 64 lanes (DPP/ds_bpermute via __shfl) ----------------
__device__ __forceinline__ unsigned wave_incl_scan(unsigned v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    unsigned o = __shfl_up(v, d);
    if (lane >= d) v += o;
  }
  return v;
}

__device__ __forceinline__ double wave_incl_scan(double v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    double o = __shfl_up(v, d);
    if (lane >= d) v += o;
  }
  return v;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

__device__ __forceinline__ unsigned wave_min(unsigned v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v = min(v, (unsigned)__shfl_xor(v, d));
  return v;
}

__device__ __forceinline__ unsigned wave_sum(unsigned v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

inline int check_geom(const lsq_conv_geom* g) {
  if (!g) return LSQ_E_NULL;
  if (g->N <= 0 || g->C <= 0 || g->H <= 0 || g->W <= 0 || g->O <= 0 || g->KH <= 0 || g->KW <= 0 ||
      g->stride_h <= 0 || g->stride_w <= 0 || g->pad_h < 0 || g->pad_w < 0 || g->dil_h <= 0 ||
      g->dil_w <= 0 || g->groups <= 0)
    return LSQ_E_SHAPE;
  if (g->C % g->groups || g->O % g->groups) return LSQ_E_SHAPE;
  return LSQ_OK;
}

inline int out_h(const lsq_conv_geom* g) {
  return (g->H + 2 * g->pad_h - g->dil_h * (g->KH - 1) - 1) / g->stride_h + 1;
}
inline int out_w(const lsq_conv_geom* g) {
  return (g->W + 2 * g->pad_w - g->dil_w * (g->KW - 1) - 1) / g->stride_w + 1;
}

}  // namespace lsq
