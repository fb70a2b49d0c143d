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


// ---------------------------------------------------------------------------------------------
// Tiled version: a 256-thread workgroup computes BM out-channels x 128 pixels.  Per K-chunk (one tap,
// 32 channels) the block stages into LDS, ONCE for its four waves, (a) the +-1 weight fragment expanded
// from 4 bytes of the packed plane per out-channel and (b) the activation chunk already clamped /
// batch-norm-folded and split into bf16 hi and lo, both as [row][32 k] with a 16-byte row pad so that
// the 16-byte MFMA fragment reads of a 32-lane group hit distinct banks.  Global loads of chunk i+1
// are issued before the MFMAs of chunk i and written to the other LDS buffer after them (one barrier
// per chunk).  Each wave owns TM x TN 32x32 tiles: 2*TM*TN*2 MFMAs per chunk against
// (TM + 2*TN)*2 fragment reads.
constexpr int kKC = 32;                         // channels per K-chunk
constexpr int kRowB = kKC * 2 + 16;             // LDS row pitch in bytes (64 data + 16 pad)
constexpr int kBN = 128;                        // pixels per block
#ifndef LSQ_SIGNW_LDS_BUFS
#define LSQ_SIGNW_LDS_BUFS 1
#endif
constexpr int kLdsBufs = LSQ_SIGNW_LDS_BUFS;    // 2: one barrier per chunk; 1: two barriers, half the LDS, more blocks per CU

template <int BM, int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256) void signw_conv_tiled(SwArgs a) {
  static_assert(WM * WN == 4 && WM * TM * 32 == BM && WN * TN * 32 == kBN, "tile shape");
  __shared__ __attribute__((aligned(16))) unsigned char smem[kLdsBufs][(BM + 2 * kBN) * kRowB];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid - wm * WN;
  const int col = lane & 31, kh8 = lane >> 5;
  const int tile = blockIdx.y;
  const int grp = tile / a.tiles_per_group;
  const int t = tile - grp * a.tiles_per_group;
  const int o_pad0 = grp * a.og_pad + t * BM;
  const int o0 = grp * a.og + t * BM;
  const int HoWo = a.Ho * a.Wo, HW = a.H * a.W;
  const long long total = (long long)a.N * HoWo;
  const int taps = a.KH * a.KW;
  const int cchunks = (a.cg + kKC - 1) / kKC;
  const int nchunks = taps * cchunks;

  // ---- staging roles
  // activations: thread -> pixel (tid & 127), k half (tid >> 7) of 16 channels
  const int sp = tid & (kBN - 1), skh = tid >> 7;
  const long long spix = (long long)blockIdx.x * kBN + sp;
  const bool sp_valid = spix < total;
  const long long spc = sp_valid ? spix : 0;
  const int sn = (int)(spc / HoWo);
  const int sr = (int)(spc - (long long)sn * HoWo);
  const int sho = sr / a.Wo, swo = sr - sho * a.Wo;
  const float* xg = a.x + ((long long)sn * a.C + (long long)grp * a.cg) * HW;
  // weights: thread -> out-channel (tid % BM), k part (tid / BM) of KC / (256 / BM) channels
  constexpr int WPARTS = 256 / BM;              // 2 (BM = 128) or 4 (BM = 64)
  constexpr int WK = kKC / WPARTS;              // 16 or 8 channels per thread
  const int so = tid % BM, swp = tid / BM;
  const bool so_valid = t * BM + so < a.og_pad;

  float xr[16];
  unsigned wr = 0;
  auto load_chunk = [&](int ch) {
    const int tap = ch / cchunks, cc = ch - tap * cchunks;
    const int kh = tap / a.KW, kw = tap - kh * a.KW;
    const int hi = sho * a.sh - a.ph + kh * a.dh, wi = swo * a.sw - a.pw + kw * a.dw;
    const bool inb = sp_valid && hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
    const int cbase = cc * kKC + skh * 16;
    // all 16 loads are issued unconditionally (out-of-image / padded channels read a valid dummy
    // address and are zeroed afterwards): a load inside a divergent branch is waited for on the spot,
    // which serialises 16 memory latencies per chunk
    const float* xp = inb ? xg + (long long)hi * a.W + wi : xg;
    float raw[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) raw[j] = xp[(long long)min(cbase + j, a.cg - 1) * HW];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int c = cbase + j;
      float v = raw[j];
      if (a.pre_scale) {
        // the channel is the same for all lanes of a wave (k half = tid >> 7): scalar loads
        const int cs = __builtin_amdgcn_readfirstlane(grp * a.cg + min(c, a.cg - 1));
        v = fmaf(v, a.pre_scale[cs], a.pre_shift[cs]);
      }
      v = clamp_sym(v, a.alpha);
      xr[j] = (inb && c < a.cg) ? v : 0.f;
    }
    const int c0 = cc * kKC;
    const int g = c0 >> 6;
    const int bit0 = (c0 & 63) + swp * WK;
    const unsigned long long wv =
        so_valid ? a.wbits[((long long)tap * a.Gg + g) * a.opad_total + o_pad0 + so] : 0ull;
    wr = (unsigned)(wv >> bit0) & ((1u << WK) - 1u);
  };
  auto store_chunk = [&](int buf) {
    unsigned char* sA = smem[buf];
    unsigned char* sBh = sA + BM * kRowB;
    unsigned char* sBl = sBh + kBN * kRowB;
    // activations -> bf16 hi (truncation) and lo (rounded remainder), 16 values = 32 bytes each
    unsigned hi[8], lo[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const unsigned x0 = __float_as_uint(xr[2 * p]), x1 = __float_as_uint(xr[2 * p + 1]);
      const unsigned h0 = x0 & 0xFFFF0000u, h1 = x1 & 0xFFFF0000u;
      const unsigned l0 = __float_as_uint(xr[2 * p] - __uint_as_float(h0));
      const unsigned l1 = __float_as_uint(xr[2 * p + 1] - __uint_as_float(h1));
      hi[p] = (h0 >> 16) | h1;
      lo[p] = ((l0 + 0x7FFFu + ((l0 >> 16) & 1u)) >> 16) | ((l1 + 0x7FFFu + ((l1 >> 16) & 1u)) & 0xFFFF0000u);
    }
    uint4* dh = reinterpret_cast<uint4*>(sBh + sp * kRowB + skh * 32);
    uint4* dl = reinterpret_cast<uint4*>(sBl + sp * kRowB + skh * 32);
    dh[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    dh[1] = make_uint4(hi[4], hi[5], hi[6], hi[7]);
    dl[0] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    dl[1] = make_uint4(lo[4], lo[5], lo[6], lo[7]);
    // weights: WK sign bits -> WK bf16 +-1
    unsigned* dw = reinterpret_cast<unsigned*>(sA + so * kRowB + swp * WK * 2);
#pragma unroll
    for (int p = 0; p < WK / 2; ++p) {
      const unsigned b0 = (wr >> (2 * p)) & 1u, b1 = (wr >> (2 * p + 1)) & 1u;
      dw[p] = 0x3F803F80u | ((b0 ^ 1u) << 15) | ((b1 ^ 1u) << 31);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  load_chunk(0);
  store_chunk(0);
  __syncthreads();
  for (int ch = 0; ch < nchunks; ++ch) {
    const int buf = kLdsBufs == 2 ? (ch & 1) : 0;
    if (ch + 1 < nchunks) load_chunk(ch + 1);          // in flight during the MFMAs below
    const unsigned char* sA = smem[buf];
    const unsigned char* sBh = sA + BM * kRowB;
    const unsigned char* sBl = sBh + kBN * kRowB;
#pragma unroll
    for (int ks = 0; ks < kKC / 16; ++ks) {
      Frag af[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const uint4 v = *reinterpret_cast<const uint4*>(sA + ((wm * TM + i) * 32 + col) * kRowB + ks * 32 + kh8 * 16);
        af[i].u[0] = v.x; af[i].u[1] = v.y; af[i].u[2] = v.z; af[i].u[3] = v.w;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = ((wn * TN + j) * 32 + col) * kRowB + ks * 32 + kh8 * 16;
        const uint4 vh = *reinterpret_cast<const uint4*>(sBh + row);
        const uint4 vl = *reinterpret_cast<const uint4*>(sBl + row);
        bh[j].u[0] = vh.x; bh[j].u[1] = vh.y; bh[j].u[2] = vh.z; bh[j].u[3] = vh.w;
        bl[j].u[0] = vl.x; bl[j].u[1] = vl.y; bl[j].u[2] = vl.z; bl[j].u[3] = vl.w;
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i].v, bh[j].v, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i].v, bl[j].v, acc[i][j], 0, 0, 0);
        }
    }
    if (kLdsBufs == 1) __syncthreads();          // everyone is done reading before the single buffer is rewritten
    if (ch + 1 < nchunks) store_chunk(kLdsBufs == 2 ? (buf ^ 1) : 0);
    __syncthreads();
  }

  // ---- epilogue: lane = pixel column, registers = out-channel rows (coalesced 128-byte stores)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const long long pix = (long long)blockIdx.x * kBN + (wn * TN + j) * 32 + col;
    if (pix >= total) continue;
    const int n = (int)(pix / HoWo);
    const int r = (int)(pix - (long long)n * HoWo);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int ol = (wm * TM + i) * 32 + (q & 3) + 8 * (q >> 2) + 4 * kh8;
        if (t * BM + ol < a.og) {
          const int o = o0 + ol;
          const long long yi = ((long long)n * a.O + o) * HoWo + r;
          const float v = acc[i][j][q] * a.wscale[o];
          float out = (a.accumulate ? a.y[yi] : (a.bias ? a.bias[o] : 0.f)) + v;
          if (a.final_pass) {
            if (a.res_pre) out += a.res_pre[yi];
            if (a.relu) out = fmaxf(out, 0.f);
            if (a.res_post) out += a.res_post[yi];
          }
          a.y[yi] = out;
        }
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
  // 128 out-channels x 128 pixels per workgroup when the group is wide enough, else 64 x 128
  const bool wide = a.og > 64;
  const int bm = wide ? 128 : 64;
  a.tiles_per_group = (a.og + bm - 1) / bm;
  for (int q = 0; q < kw_planes; ++q) {
    a.wbits = (const unsigned long long*)wbits + (long long)q * wplane_words;
    a.wscale = wscales + (long long)q * g->O;
    a.accumulate = q ? 1 : 0;
    a.final_pass = q == kw_planes - 1 ? 1 : 0;
    dim3 grid((unsigned)((total + kBN - 1) / kBN), (unsigned)(g->groups * a.tiles_per_group));
    if (wide) hipLaunchKernelGGL((signw_conv_tiled<128, 2, 2, 2, 2>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((signw_conv_tiled<64, 1, 4, 2, 1>), grid, dim3(256), 0, st, a);
    if (int e = (int)hipGetLastError()) return e;
  }
  return LSQ_OK;
}
