// Shared device helpers for the gfx950 least-squares quantization kernels.
// Wavefront = 64 lanes everywhere (CDNA4); no portability layer.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lsq_hip.h"

namespace lsq {

constexpr int kWave = 64;

// Workgroup barrier for phases whose waves talk to each other through LDS only.  __syncthreads() is a release fence +
// s_barrier + acquire fence, and on gfx950 the release fence drains the vector-memory counter too: every global load
// in flight is waited for at every barrier.  A kernel that requests data early and keeps it in flight across a chain
// of LDS-only phases needs the barrier without that drain: LDS operations of a CU complete in order, so
// lgkmcnt(0) + s_barrier is all the ordering those phases need.  (The asm is a compiler barrier for memory operations;
// registers with loads pending are still tracked by the compiler, which waits for them at their first use.)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float clamp_sym(float x, float alpha) {
  // torch.clamp(x, -alpha, alpha); a negative alpha encodes clamp_identity
  return alpha >= 0.f ? fminf(fmaxf(x, -alpha), alpha) : x;
}

__device__ __forceinline__ unsigned abs_key(float x) {
  // order-preserving integer key of |x| (IEEE-754 bits without the sign)
  return __float_as_uint(x) & 0x7FFFFFFFu;
}

__device__ __forceinline__ float key_value(unsigned key) { return __uint_as_float(key); }

// ---- wave-level scans / reductions on 64 lanes -------------------------------------------------
// Inclusive scans on DPP moves (row shifts inside the rows of 16 lanes, then the two row broadcasts): a dozen VALU
// instructions per scan where the ds_bpermute form of __shfl_up pays six dependent LDS round trips.  A lane without
// a source keeps `old` = 0, so the adds need no lane tests.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_or_zero(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}

__device__ __forceinline__ unsigned wave_incl_scan(unsigned v) {
  v += dpp_or_zero<0x111, 0xF>(v);     // row_shr:1
  v += dpp_or_zero<0x112, 0xF>(v);     // row_shr:2
  v += dpp_or_zero<0x114, 0xF>(v);     // row_shr:4
  v += dpp_or_zero<0x118, 0xF>(v);     // row_shr:8
  v += dpp_or_zero<0x142, 0xA>(v);     // row_bcast:15 into rows 1 and 3
  v += dpp_or_zero<0x143, 0xC>(v);     // row_bcast:31 into rows 2 and 3
  return v;
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_or_zero(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = dpp_or_zero<CTRL, ROW_MASK>((unsigned)u), hi = dpp_or_zero<CTRL, ROW_MASK>((unsigned)(u >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

__device__ __forceinline__ double wave_incl_scan(double v) {
  v += dpp_or_zero<0x111, 0xF>(v);
  v += dpp_or_zero<0x112, 0xF>(v);
  v += dpp_or_zero<0x114, 0xF>(v);
  v += dpp_or_zero<0x118, 0xF>(v);
  v += dpp_or_zero<0x142, 0xA>(v);
  v += dpp_or_zero<0x143, 0xC>(v);
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
