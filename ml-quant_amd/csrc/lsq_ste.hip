// Training-side elementwise pieces of the quantizers for gfx950:
//
//   lsq_quant_values   x_q = sum_i v_i b_i, the value every quantizer of quant/binary/quantization.py returns
//                      (:56 ls-1, :89-92 ls-2, :112-115 ls-T, :137-146 gf-k), from the input and given per-row scales --
//                      the fp32 operand of the weight-gradient convolution;
//   lsq_ste_backward   the gradient of that value with respect to the input through the straight-through estimator of
//                      quant/binary/ste.py:51-66 (a sign passes the gradient where its argument lies in [-1, 1]) and the
//                      symmetric clamp in front of it (quantization.py:22-24: the gradient passes inside [-alpha, alpha]).
//
// Every scheme is the same chain: r_0 = 0, d_i = clamp(x) - r_i, b_i = sign(d_i), r_{i+1} = r_i + v_i b_i (ls-2 = two
// steps with its solved scales, ls-T = two steps with v_2 = v_1, ls-1 = one step, gf-k = k steps); the scales are computed
// from detached data in the reference, so no gradient flows through them.  Backwards, with G_k = g:
//   t_i = G_{i+1} v_i [|d_i| <= 1],   G_i = G_{i+1} - t_i,   dL/dx = [|x| <= alpha] sum_i t_i.
// Rows are samples (activations, M = C H W) or output channels (weights, M = C KH KW, no clamp).  HBM-bound: x and g read
// once, the gradient written once, 16-byte accesses when the row length allows.

#include "lsq_common.h"

namespace lsq {
namespace {

constexpr int kSteThreads = 256;
constexpr int kMaxK = LSQ_MAX_PLANES;

struct SteArgs {
  const float* x;
  const float* g;       // null: forward values only
  float* out;
  const float* scales;  // [k][rows]
  long long rows, M;
  int k;
  float alpha;          // < 0: no clamp
};

__device__ __forceinline__ float ste_one(float xv, float gv, const float (&v)[kMaxK], int k, float alpha, bool backward) {
  const bool inside = alpha < 0.f || (xv >= -alpha && xv <= alpha);        // clamp backward: min <= x <= max
  const float xc = clamp_sym(xv, alpha);
  float r = 0.f;
  float d[kMaxK];
#pragma unroll
  for (int i = 0; i < kMaxK; ++i) {
    if (i < k) {
      d[i] = xc - r;
      r = r + v[i] * (d[i] >= 0.f ? 1.f : -1.f);       // sign(+-0) = +1 (ste.py:16-18)
    }
  }
  if (!backward) return k == 0 ? xc : r;
  float G = gv, acc = 0.f;
#pragma unroll
  for (int i = kMaxK - 1; i >= 0; --i) {
    if (i < k) {
      const float t = (fabsf(d[i]) <= 1.f) ? G * v[i] : 0.f;
      acc += t;
      G -= t;
    }
  }
  if (k == 0) acc = gv;
  return inside ? acc : 0.f;
}

template <int VEC, bool BACKWARD>
__global__ __launch_bounds__(kSteThreads) void ste_kernel(SteArgs a) {
  const long long row = blockIdx.y;
  float v[kMaxK];
#pragma unroll
  for (int i = 0; i < kMaxK; ++i) v[i] = i < a.k ? a.scales[(long long)i * a.rows + row] : 0.f;
  const float* __restrict__ xr = a.x + row * a.M;
  const float* __restrict__ gr = BACKWARD ? a.g + row * a.M : nullptr;
  float* __restrict__ outr = a.out + row * a.M;
  const long long nvec = a.M / VEC;
  const long long step = (long long)gridDim.x * kSteThreads;
  for (long long i = (long long)blockIdx.x * kSteThreads + threadIdx.x; i < nvec; i += step) {
    if constexpr (VEC == 4) {
      const float4 xv = reinterpret_cast<const float4*>(xr)[i];
      float4 gv = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (BACKWARD) gv = reinterpret_cast<const float4*>(gr)[i];
      float4 o;
      o.x = ste_one(xv.x, gv.x, v, a.k, a.alpha, BACKWARD);
      o.y = ste_one(xv.y, gv.y, v, a.k, a.alpha, BACKWARD);
      o.z = ste_one(xv.z, gv.z, v, a.k, a.alpha, BACKWARD);
      o.w = ste_one(xv.w, gv.w, v, a.k, a.alpha, BACKWARD);
      reinterpret_cast<float4*>(outr)[i] = o;
    } else {
      outr[i] = ste_one(xr[i], BACKWARD ? gr[i] : 0.f, v, a.k, a.alpha, BACKWARD);
    }
  }
}

int launch_ste(const SteArgs& a, hipStream_t st) {
  const bool vec4 = a.M % 4 == 0 && ((uintptr_t)a.x % 16) == 0 && ((uintptr_t)a.out % 16) == 0 && (!a.g || ((uintptr_t)a.g % 16) == 0);
  const long long nvec = vec4 ? a.M / 4 : a.M;
  // about four workgroups per CU over all rows, at least one per row
  long long per_row = (nvec + kSteThreads - 1) / kSteThreads;
  const long long want = (1024 + a.rows - 1) / a.rows;
  if (per_row > want) per_row = want;
  if (per_row < 1) per_row = 1;
  const dim3 grid((unsigned)per_row, (unsigned)a.rows);
  if (a.g) {
    if (vec4) hipLaunchKernelGGL((ste_kernel<4, true>), grid, dim3(kSteThreads), 0, st, a);
    else hipLaunchKernelGGL((ste_kernel<1, true>), grid, dim3(kSteThreads), 0, st, a);
  } else {
    if (vec4) hipLaunchKernelGGL((ste_kernel<4, false>), grid, dim3(kSteThreads), 0, st, a);
    else hipLaunchKernelGGL((ste_kernel<1, false>), grid, dim3(kSteThreads), 0, st, a);
  }
  return (int)hipGetLastError();
}

int check(const float* x, int64_t rows, int64_t M, int k, const float* scales, float* out) {
  if (!x || !out || (k > 0 && !scales)) return LSQ_E_NULL;
  if (rows <= 0 || M <= 0 || rows > 65535) return LSQ_E_SHAPE;
  if (k < 0 || k > kMaxK) return LSQ_E_SCHEME;
  return LSQ_OK;
}

}  // namespace
}  // namespace lsq

extern "C" int lsq_quant_values(const float* x, int64_t rows, int64_t M, int k, const float* scales, float clamp_alpha,
                                float* x_q, void* stream) {
  if (const int e = lsq::check(x, rows, M, k, scales, x_q)) return e;
  lsq::SteArgs a = {x, nullptr, x_q, scales, rows, M, k, clamp_alpha};
  return lsq::launch_ste(a, (hipStream_t)stream);
}

extern "C" int lsq_ste_backward(const float* x, const float* grad_q, int64_t rows, int64_t M, int k, const float* scales,
                                float clamp_alpha, float* grad_x, void* stream) {
  if (!grad_q) return LSQ_E_NULL;
  if (const int e = lsq::check(x, rows, M, k, scales, grad_x)) return e;
  lsq::SteArgs a = {x, grad_q, grad_x, scales, rows, M, k, clamp_alpha};
  return lsq::launch_ste(a, (hipStream_t)stream);
}
