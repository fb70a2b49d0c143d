// Full-precision activation x sign-weight convolution on the gfx950 matrix cores.
//
// Replaces F.conv2d(x, w_q, ...) of quant/binary/binary_conv.py:165-173 when x_quant == 'fp' and the
// weights are sums of scaled sign planes:  y[n,o] = bias[o] + sum_q u_q[o] * conv(clamp(x), s_q)[n,o].
//
// Implicit GEMM, transposed so that stores are coalesced:  D[o][pixel] = sum_k S[o][k] * X[k][pixel],
// k = (tap, channel).  v_mfma_f32_32x32x16_bf16: A = 32 out-channels x 16 channels of +-1 (exact in
// bf16), B = 16 channels x 32 pixels of activations.  fp32 activations are split x = hi + lo with
// hi = bf16 truncation of x and lo = bf16(x - hi): two MFMAs per tile, relative error <= 2^-16 per
// product, fp32 accumulation -- inside the 1e-4 bound where single bf16 (2^-8) is not.
//
// One wave owns 32 consecutive output pixels x (32*OTW) output channels: each activation fragment is
// loaded once (8 floats per lane, lanes = consecutive pixels => coalesced 128 B segments per channel)
// and reused for OTW weight tiles and both halves of the split.  A weight fragment is ONE byte of the
// packed sign plane per lane (8 consecutive channels of one out-channel), expanded to 8 bf16 with
// three VALU ops per pair.  Zero padding and channel padding are exact (x = 0 contributes 0).

#include "lsq_common.h"

namespace lsq {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct SwArgs {
  const float* x;                      // [N][C][H][W]
  const unsigned long long* wbits;     // [taps][Gg][Opad]  (one weight plane)
  const float* wscale;                 // [O]
  const float* bias;                   // [O] or null
  const float* pre_scale;              // [C] or null (folded eval batch norm)
  const float* pre_shift;
  float* y;                            // [N][O][Ho][Wo]
  float alpha;
  int N, C, H, W, O, KH, KW, sh, sw, ph, pw, dh, dw;
  int Gg, Ho, Wo, cg, og, og_pad, opad_total, tiles_per_group;
  int accumulate;
  int final_pass;
  int relu;                            // epilogue: y = relu(conv + bias + res_pre) + res_post
  const float* res_pre;                // [N][O][Ho][Wo] or null
  const float* res_post;
};

union Frag {
  unsigned u[4];
  bf16x8 v;
};

// 8 sign bits (bit j = channel j) -> 8 bf16 values +1 / -1
__device__ __forceinline__ bf16x8 expand_signs(unsigned byte) {
  Frag f;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const unsigned b0 = (byte >> (2 * p)) & 1u, b1 = (byte >> (2 * p + 1)) & 1u;
    f.u[p] = 0x3F803F80u | ((b0 ^ 1u) << 15) | ((b1 ^ 1u) << 31);
  }
  return f.v;
}

template <int OTW>
__global__ __launch_bounds__(256) void signw_conv_kernel(SwArgs a) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int col = lane & 31, kh8 = lane >> 5;          // pixel / out-channel within the tile, k half
  const long long total = (long long)a.N * a.Ho * a.Wo;
  const long long pix = ((long long)blockIdx.x * 4 + wid) * 32 + col;
  const bool pvalid = pix < total;
  const int tile = blockIdx.y;
  const int grp = tile / a.tiles_per_group;
  const int t = tile - grp * a.tiles_per_group;
  const int o_pad0 = grp * a.og_pad + t * (32 * OTW);
  const int o0 = grp * a.og + t * (32 * OTW);
  const int HoWo = a.Ho * a.Wo, HW = a.H * a.W;
  const long long pc = pvalid ? pix : 0;
  const int n = (int)(pc / HoWo);
  const int r = (int)(pc - (long long)n * HoWo);
  const int ho = r / a.Wo, wo = r - ho * a.Wo;
  const float* xg = a.x + ((long long)n * a.C + (long long)grp * a.cg) * HW;

  f32x16 acc[OTW];
#pragma unroll
  for (int i = 0; i < OTW; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;

  for (int kh = 0; kh < a.KH; ++kh) {
    for (int kw = 0; kw < a.KW; ++kw) {
      const int hi = ho * a.sh - a.ph + kh * a.dh, wi = wo * a.sw - a.pw + kw * a.dw;
      const bool inb = pvalid && hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
      const float* xp = xg + (long long)hi * a.W + wi;
      const int tap = kh * a.KW + kw;
      for (int c0 = 0; c0 < a.cg; c0 += 16) {
        // B fragment: 8 channels of this lane's pixel
        float xv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int c = c0 + 8 * kh8 + j;
          float t = 0.f;
          if (inb && c < a.cg) {
            t = xp[(long long)c * HW];
            if (a.pre_scale) t = fmaf(t, a.pre_scale[grp * a.cg + c], a.pre_shift[grp * a.cg + c]);
            t = clamp_sym(t, a.alpha);
          }
          xv[j] = t;
        }
        Frag bhi, blo;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const unsigned x0 = __float_as_uint(xv[2 * p]), x1 = __float_as_uint(xv[2 * p + 1]);
          const unsigned h0 = x0 & 0xFFFF0000u, h1 = x1 & 0xFFFF0000u;              // truncation to bf16
          const unsigned l0 = __float_as_uint(xv[2 * p] - __uint_as_float(h0));     // exact in fp32
          const unsigned l1 = __float_as_uint(xv[2 * p + 1] - __uint_as_float(h1));
          const unsigned r0 = (l0 + 0x7FFFu + ((l0 >> 16) & 1u)) >> 16;             // round to nearest even
          const unsigned r1 = (l1 + 0x7FFFu + ((l1 >> 16) & 1u)) & 0xFFFF0000u;
          bhi.u[p] = (h0 >> 16) | h1;
          blo.u[p] = r0 | r1;
        }
        // A fragments: one byte of the packed plane per lane and tile
        const int g = c0 >> 6, byte_sel = ((c0 & 63) >> 3) + kh8;
        const unsigned long long* wp = a.wbits + ((long long)tap * a.Gg + g) * a.opad_total + o_pad0 + col;
#pragma unroll
        for (int i = 0; i < OTW; ++i) {
          // out-channel slots beyond the padded group width do not exist in the plane
          const unsigned long long wv = (t * (32 * OTW) + 32 * i + col) < a.og_pad ? wp[32 * i] : 0ull;
          const bf16x8 af = expand_signs((unsigned)(wv >> (8 * byte_sel)) & 0xFFu);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bhi.v, acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, blo.v, acc[i], 0, 0, 0);
        }
      }
    }
  }

  if (!pvalid) return;
  float* yp = a.y + ((long long)n * a.O + o0) * HoWo + r;
#pragma unroll
  for (int i = 0; i < OTW; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int ol = 32 * i + (j & 3) + 8 * (j >> 2) + 4 * kh8;     // C/D layout of the 32x32 MFMA
      if (t * (32 * OTW) + ol < a.og) {
        const int o = o0 + ol;
        const float v = acc[i][j] * a.wscale[o];
        const float base = a.accumulate ? yp[(long long)ol * HoWo] : (a.bias ? a.bias[o] : 0.f);
        float out = base + v;
        if (a.final_pass) {
          const long long yi = ((long long)n * a.O + o) * HoWo + r;
          if (a.res_pre) out += a.res_pre[yi];
          if (a.relu) out = fmaxf(out, 0.f);
          if (a.res_post) out += a.res_post[yi];
        }
        yp[(long long)ol * HoWo] = out;
      }
    }
  }
}

}  // namespace
}  // namespace lsq

using namespace lsq;

extern "C" int lsq_signw_conv2d(const float* x, float clamp_alpha, const float* pre_scale, const float* pre_shift,
                                const uint64_t* wbits, int kw_planes, const float* wscales, const float* bias,
                                const lsq_conv_geom* g, int relu, const float* res_pre, const float* res_post,
                                float* y, void* stream) {
  if (!x || !wbits || !wscales || !y) return LSQ_E_NULL;
  if (int e = check_geom(g)) return e;
  if (kw_planes < 1 || kw_planes > LSQ_MAX_PLANES) return LSQ_E_SCHEME;
  const int Ho = out_h(g), Wo = out_w(g);
  if (Ho <= 0 || Wo <= 0) return LSQ_E_SHAPE;
  SwArgs a = {};
  if ((pre_scale == nullptr) != (pre_shift == nullptr)) return LSQ_E_NULL;
  a.x = x; a.bias = bias; a.y = y; a.alpha = clamp_alpha;
  a.pre_scale = pre_scale; a.pre_shift = pre_shift;
  a.relu = relu; a.res_pre = res_pre; a.res_post = res_post;
  a.N = g->N; a.C = g->C; a.H = g->H; a.W = g->W; a.O = g->O; a.KH = g->KH; a.KW = g->KW;
  a.sh = g->stride_h; a.sw = g->stride_w; a.ph = g->pad_h; a.pw = g->pad_w; a.dh = g->dil_h; a.dw = g->dil_w;
  a.cg = g->C / g->groups;
  a.Gg = (a.cg + 63) / 64;
  a.Ho = Ho; a.Wo = Wo;
  a.og = g->O / g->groups;
  a.og_pad = (a.og + 15) / 16 * 16;
  a.opad_total = g->groups * a.og_pad;
  const long long wplane_words = lsq_weight_plane_words(g);
  const long long total = (long long)g->N * Ho * Wo;
  hipStream_t st = (hipStream_t)stream;
  // tiles of 128 out-channels when the group is wide enough, else 64 (reads of padded slots beyond
  // og_pad are avoided by the per-lane guard below: tiles never start beyond og)
  const int otw = a.og > 64 ? 4 : 2;
  a.tiles_per_group = (a.og + 32 * otw - 1) / (32 * otw);
  for (int q = 0; q < kw_planes; ++q) {
    a.wbits = (const unsigned long long*)wbits + (long long)q * wplane_words;
    a.wscale = wscales + (long long)q * g->O;
    a.accumulate = q ? 1 : 0;
    a.final_pass = q == kw_planes - 1 ? 1 : 0;
    dim3 grid((unsigned)((total + 127) / 128), (unsigned)(g->groups * a.tiles_per_group));
    if (otw == 4) hipLaunchKernelGGL((signw_conv_kernel<4>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((signw_conv_kernel<2>), grid, dim3(256), 0, st, a);
    if (int e = (int)hipGetLastError()) return e;
  }
  return LSQ_OK;
}
