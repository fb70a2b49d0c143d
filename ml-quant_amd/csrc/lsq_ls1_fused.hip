// 1-bit activations on small images: quantizer AND convolution in one launch (gfx950).
//
// QuantConv2d.forward with x_quant = 'ls-1' (quant/binary/binary_conv.py:161-173; quantizer_ls_1, quantization.py:35-56)
// is y[n,o] = u_o v1_n (b (*) s_o) + bias_o with b = sign(clamp(x)), v1_n = mean |clamp(x_n)|: the activation scale only
// enters the epilogue.  On CIFAR-sized images (cifar100_ls1_kd.yaml: 32^2 ... 4^2 pixels, batch 100) the two-kernel path
// -- lsq_act_quant writes the bit plane and the scale, lsq_xnor_conv2d reads them back -- is launch- and latency-bound:
// 32 launches of 8-17 us for a few hundred KB each, one workgroup per sample on 100 of 256 CUs.  Here ONE workgroup owns
// a sample (or an out-channel slice of it, so that a batch of 100 fills the chip):
//   phase 1  read the sample's input once (folded batch norm, clamp): sign bits -> a bit plane with a zero halo in LDS
//            (byte stores: the 8 channels of an octet are one byte of the word), sum |x| in the exact arithmetic of the
//            plain sweeps -- the fp32 sum of every octet of channels rounded to a multiple of 2^e, added in fp64, so the
//            scale is the same NUMBER as lsq_act_quant's (bitwise; lsq_act_quant.hip, EXACT);
//   phase 2  XNOR + popcount convolution from the LDS plane, OT out-channels per lane in registers, weight words from the
//            packed planes of lsq_pack_weight (global memory, L1-resident); border taps corrected with the tap sums;
//   epilogue the popcount kernel's arithmetic in the same order (lsq_xnor_conv.hip): bias, weight scale, residual adds,
//            ReLU / PReLU -- the output equals the two-kernel path's bit for bit (tests/test_gpu_round4.py).
// Covered: kernels up to 3 x 3, groups 1, dilation 1, C a multiple of 64 up to 512, images up to 32 x 32 + halo,
// a symmetric clamp (the exact row sum needs the bound).  Anything else: LSQ_E_UNSUPPORTED, the caller takes the two kernels.

#include "lsq_common.h"

namespace lsq {
namespace {

constexpr int kT = 1024;                // threads per workgroup: sixteen waves hide the global-memory round trips of the
                                        // epilogue (residuals, stores) that four waves exposed -- 120 us per 32 x 32 layer
constexpr int kOT = 8;                  // out-channels per lane and pass
constexpr int kMaxWords = 1280;         // plane words in LDS: C / 64 x (H + 2 ph) x (W + 2 pw)  (64 x 34 x 34 = 1156)
constexpr int kMaxK = 3;                // kernel rows / columns
constexpr int kMaxTaps = kMaxK * kMaxK;
constexpr int kMaxSlice = 256;          // out-channels per workgroup
constexpr int kMaxWWords = 5120;        // weight words of a slice staged in LDS: taps x C / 64 x out-channels (40 KB)

struct L1Args {
  const float* x;
  const float* pre_scale;
  const float* pre_shift;
  const unsigned long long* wbits;   // [q][taps][Gg][Opad]
  const int* wsum;                   // [q][O][taps]
  const float* wscales;              // [q][O]
  const float* bias;
  const float* slope;
  const float* res_pre;
  const float* res_post;
  float* y;
  float* scales;                     // [1][N] out
  long long wplane_words;
  double qmagic;
  float alpha;
  int N, C, H, W, O, KH, KW, sh, sw, ph, pw, Hp, Wp, Ho, Wo, Gg, opad, kw_planes, act, oslice, nslices, ngroups;
};

struct L1Lds {
  unsigned long long plane[kMaxWords];
  int ws[kMaxTaps][kMaxSlice];       // tap sums of the slice's out-channels (one weight plane at a time)
  int rs[kMaxK][kMaxSlice], cs[kMaxK][kMaxSlice];
  __attribute__((aligned(16))) unsigned long long wts[kMaxWWords];   // [tap][word][out-channel of the slice, padded to kOT]
  float wsc[kMaxSlice], bia[kMaxSlice];   // weight scales (one plane at a time) and bias of the slice
  double red[kT / 64];
  float v1;
};

template <bool K3>
__global__ __launch_bounds__(kT) void ls1_conv_kernel(L1Args a) {
  __shared__ L1Lds lds;
  const int tid = threadIdx.x;
  // a workgroup owns one out-channel slice and walks over the samples group, group + ngroups, ...: the slice's weight
  // words, tap sums, scales and bias are staged ONCE (deep layers: 36 KB of weights against 32 KB of input per sample)
  const int slice = blockIdx.x % a.nslices, group = blockIdx.x / a.nslices;
  const int o_lo = slice * a.oslice, o_n = min(a.oslice, a.O - o_lo);
  const int HW = a.H * a.W, HpWp = a.Hp * a.Wp, taps = a.KH * a.KW;
  const int HoWo = a.Ho * a.Wo;
  const int otiles = (o_n + kOT - 1) / kOT;
  const int wslice = otiles * kOT;                          // out-channels of the slice, padded to whole tiles
  const int cunits = otiles * HoWo;
  const int full = a.C * taps;
  const int iters = (cunits + kT - 1) / kT;

  auto stage = [&](int q) {                                  // weight plane q of the slice -> LDS (all lanes call)
    for (int i = tid; i < taps * o_n; i += kT) {
      const int tp = i / o_n, o = i - tp * o_n;
      lds.ws[tp][o] = a.wsum[((long long)q * a.O + o_lo + o) * taps + tp];
    }
    for (int o = tid; o < o_n; o += kT) {
      lds.wsc[o] = a.wscales[(long long)q * a.O + o_lo + o];
      lds.bia[o] = a.bias ? a.bias[o_lo + o] : 0.f;
    }
    // (the packed planes are padded to 16 out-channels: a slice's tail tile reads defined words)
    const unsigned long long* __restrict__ wsrc = a.wbits + (long long)q * a.wplane_words + o_lo;
    const int rows = taps * a.Gg;
    for (int i = tid; i < rows * wslice; i += kT) {
      const int rw = i / wslice, o = i - rw * wslice;
      lds.wts[i] = (o_lo + o < a.opad) ? wsrc[rw * a.opad + o] : 0ull;
    }
    __syncthreads();
    for (int i = tid; i < (a.KH + a.KW) * o_n; i += kT) {
      const int k = i / o_n, o = i - k * o_n;
      int s = 0;
      if (k < a.KH) {
        for (int kw = 0; kw < a.KW; ++kw) s += lds.ws[k * a.KW + kw][o];
        lds.rs[k][o] = s;
      } else {
        for (int kh = 0; kh < a.KH; ++kh) s += lds.ws[kh * a.KW + (k - a.KH)][o];
        lds.cs[k - a.KH][o] = s;
      }
    }
    __syncthreads();
  };
  for (int i = tid; i < a.Gg * HpWp; i += kT) lds.plane[i] = 0ull;     // (the halo stays zero for every sample)
  if (a.kw_planes == 1) stage(0);
  else __syncthreads();

  // what a lane's units look like does not depend on the sample: (out-channel tile, output pixel) of iteration `it`
  for (int n = group; n < a.N; n += a.ngroups) {
    const float* __restrict__ xn = a.x + (long long)n * a.C * HW;
    // ---- phase 1: units of 8 channels x 4 consecutive pixels (HW % 4 == 0: a unit never leaves its image plane)
    double acc = 0.0;
    {
      const int quads = HW >> 2;
      const int units = (a.C >> 3) * quads;
      const bool affine = a.pre_scale != nullptr;
      // two units (16 float4 loads) in flight per lane: the phase is latency-bound on small images
      for (int u0 = tid; u0 < units; u0 += 2 * kT) {
        float4 v[2][8];
        int octs[2], p0s[2];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const int u = min(u0 + h2 * kT, units - 1);
          const int oct = u / quads, qd = u - oct * quads;   // octet of channels, quad of pixels
          octs[h2] = oct;
          p0s[h2] = qd << 2;
          const float* __restrict__ src = xn + (oct << 3) * HW + p0s[h2];
#pragma unroll
          for (int k = 0; k < 8; ++k) v[h2][k] = *reinterpret_cast<const float4*>(src + k * HW);
        }
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          if (u0 + h2 * kT < units) {
            const int oct = octs[h2], c0 = oct << 3, p0 = p0s[h2];
            unsigned bits[4] = {0u, 0u, 0u, 0u};
            float facc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              float e[4] = {v[h2][k].x, v[h2][k].y, v[h2][k].z, v[h2][k].w};
              const float sc = affine ? a.pre_scale[c0 + k] : 1.f, sh = affine ? a.pre_shift[c0 + k] : 0.f;
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                const float xv = clamp_sym(affine ? fmaf(e[t], sc, sh) : e[t], a.alpha);
                bits[t] |= (xv >= 0.f ? 1u : 0u) << k;       // sign(+-0) = +1 (ste.py:16-18)
                facc[t] += fabsf(xv);                        // channel order inside the octet: the sweeps' order
              }
            }
            const int j = oct >> 3, byte = oct & 7;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              acc += ((double)facc[t] + a.qmagic) - a.qmagic;   // rounded to a multiple of 2^e: fp64 adds those exactly
              const int pix = p0 + t, h = pix / a.W, w = pix - h * a.W;
              reinterpret_cast<unsigned char*>(&lds.plane[(j * a.Hp + h + a.ph) * a.Wp + w + a.pw])[byte] = (unsigned char)bits[t];
            }
          }
        }
      }
    }
    acc = wave_sum(acc);
    if ((tid & 63) == 0) lds.red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
      for (int w = 0; w < kT / 64; ++w) t += lds.red[w];
      const float v1 = (float)(t / (double)((long long)a.C * HW));
      lds.v1 = v1;
      if (slice == 0) a.scales[n] = v1;
    }
    __syncthreads();                                         // plane, v1 (and the staged tables) are visible
    const float xs = lds.v1;

    // ---- phase 2: units of (out-channel tile, output pixel); outputs are finished over the weight planes in order:
    // out = fma(v1 * (b * s_q), u_q, out), out starting at the bias -- the sequence of the two-kernel path's launches
    for (int it = 0; it < iters; ++it) {
      const int u = it * kT + tid;
      const bool live = u < cunits;
      const int uu = live ? u : 0;
      const int ot = uu / HoWo, r = uu - ot * HoWo;
      const int ho = r / a.Wo, wo = r - ho * a.Wo;
      const int ol = ot * kOT;                               // first out-channel of the tile inside the slice
      const int ov = min(kOT, o_n - ol);
      // the residual operands of the epilogue are requested now, a whole popcount loop ahead of their use
      const int ybase = (n * a.O + o_lo + ol) * HoWo + r;    // (outputs below 2^31 elements: the entry point checks)
      float rpre[kOT], rpost[kOT];
#pragma unroll
      for (int o = 0; o < kOT; ++o) {
        const int yi = ybase + (o < ov ? o : 0) * HoWo;
        rpre[o] = (live && a.res_pre) ? a.res_pre[yi] : 0.f;
        rpost[o] = (live && a.res_post) ? a.res_post[yi] : 0.f;
      }
      unsigned bad_h = 0, bad_w = 0;
#pragma unroll
      for (int k = 0; k < kMaxK; ++k) {
        const int hi = ho * a.sh - a.ph + k, wi = wo * a.sw - a.pw + k;
        bad_h |= (k < a.KH && (hi < 0 || hi >= a.H)) ? 1u << k : 0u;
        bad_w |= (k < a.KW && (wi < 0 || wi >= a.W)) ? 1u << k : 0u;
      }
      float out[kOT];
      for (int q = 0; q < a.kw_planes; ++q) {
        if (a.kw_planes > 1) {
          __syncthreads();                                   // the previous plane's tables are done with
          stage(q);
        }
        if (q == 0) {
#pragma unroll
          for (int o = 0; o < kOT; ++o) out[o] = lds.bia[min(ol + o, o_n - 1)];
        }
        if (live) {
          int pc[kOT];
#pragma unroll
          for (int o = 0; o < kOT; ++o) pc[o] = 0;
          // weights of the tile from the LDS copy: [tap][word][slice channel], 8 consecutive channels = four 16-byte reads
          const unsigned long long* __restrict__ wl = lds.wts + ol;
          auto tap_step = [&](const unsigned long long xa, const unsigned long long* __restrict__ wp) {
            unsigned long long wv[kOT];
#pragma unroll
            for (int o = 0; o < kOT; o += 2) {
              const ulonglong2 t2 = *reinterpret_cast<const ulonglong2*>(wp + o);
              wv[o] = t2.x;
              wv[o + 1] = t2.y;
            }
#pragma unroll
            for (int o = 0; o < kOT; ++o) {
              const unsigned lo = (unsigned)xa ^ (unsigned)wv[o];
              asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(pc[o]) : "v"(lo));
            }
#pragma unroll
            for (int o = 0; o < kOT; ++o) {
              const unsigned hi = (unsigned)(xa >> 32) ^ (unsigned)(wv[o] >> 32);
              asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(pc[o]) : "v"(hi));
            }
          };
          for (int j = 0; j < a.Gg; ++j) {
            const unsigned long long* __restrict__ xp = &lds.plane[(j * a.Hp + ho * a.sh) * a.Wp + wo * a.sw];
            if constexpr (K3) {
#pragma unroll
              for (int kh = 0; kh < 3; ++kh) {
                const unsigned long long x0 = xp[kh * a.Wp], x1 = xp[kh * a.Wp + 1], x2 = xp[kh * a.Wp + 2];
                tap_step(x0, wl + ((kh * 3 + 0) * a.Gg + j) * wslice);
                tap_step(x1, wl + ((kh * 3 + 1) * a.Gg + j) * wslice);
                tap_step(x2, wl + ((kh * 3 + 2) * a.Gg + j) * wslice);
              }
            } else {
              for (int kh = 0; kh < a.KH; ++kh)
                for (int kw = 0; kw < a.KW; ++kw) tap_step(xp[kh * a.Wp + kw], wl + ((kh * a.KW + kw) * a.Gg + j) * wslice);
            }
          }
          int corr[kOT];
#pragma unroll
          for (int o = 0; o < kOT; ++o) corr[o] = 0;
          if (bad_h | bad_w) {
            for (int kh = 0; kh < a.KH; ++kh) {
              if ((bad_h >> kh) & 1u) {
#pragma unroll
                for (int o = 0; o < kOT; ++o) corr[o] += lds.rs[kh][min(ol + o, o_n - 1)];
                for (int kw = 0; kw < a.KW; ++kw)
                  if ((bad_w >> kw) & 1u) {
#pragma unroll
                    for (int o = 0; o < kOT; ++o) corr[o] -= lds.ws[kh * a.KW + kw][min(ol + o, o_n - 1)];
                  }
              }
            }
            for (int kw = 0; kw < a.KW; ++kw)
              if ((bad_w >> kw) & 1u) {
#pragma unroll
                for (int o = 0; o < kOT; ++o) corr[o] += lds.cs[kw][min(ol + o, o_n - 1)];
              }
          }
#pragma unroll
          for (int o = 0; o < kOT; ++o) {
            if (o < ov) {
              const float v = xs * (float)(full + corr[o] - (pc[o] << 1));
              out[o] = fmaf(v, lds.wsc[ol + o], out[o]);
            }
          }
        }
      }
      if (live) {
#pragma unroll
        for (int o = 0; o < kOT; ++o) {
          if (o < ov) {
            float v = out[o];
            if (a.res_pre) v += rpre[o];
            if (a.act == LSQ_ACT_RELU) v = fmaxf(v, 0.f);
            else if (a.act >= LSQ_ACT_PRELU) v = v > 0.f ? v : a.slope[a.act == LSQ_ACT_PRELU ? 0 : o_lo + ol + o] * v;
            if (a.res_post) v += rpost[o];
            a.y[ybase + o * HoWo] = v;
          }
        }
      }
    }
    __syncthreads();                                         // the plane is rewritten by the next sample
  }
}

}  // namespace
}  // namespace lsq

using namespace lsq;

extern "C" int lsq_ls1_conv2d(const float* x, const lsq_conv_geom* g, float clamp_alpha, const float* pre_scale,
                              const float* pre_shift, const uint64_t* wbits, const int32_t* wsum, int kw_planes,
                              const float* wscales, const float* bias, int act, const float* act_slope,
                              const float* res_pre, const float* res_post, float* y, float* scales, void* stream) {
  if (!x || !wbits || !wsum || !wscales || !y || !scales) return LSQ_E_NULL;
  if (int e = check_geom(g)) return e;
  if ((pre_scale == nullptr) != (pre_shift == nullptr)) return LSQ_E_NULL;
  if (kw_planes < 1 || kw_planes > LSQ_MAX_PLANES) return LSQ_E_SCHEME;
  if (act < LSQ_ACT_NONE || act > LSQ_ACT_PRELU_CHANNEL || (act >= LSQ_ACT_PRELU && !act_slope)) return LSQ_E_SCHEME;
  const int Ho = out_h(g), Wo = out_w(g);
  if (Ho <= 0 || Wo <= 0) return LSQ_E_SHAPE;
  const int Hp = g->H + 2 * g->pad_h, Wp = g->W + 2 * g->pad_w, Gg = g->C / 64;
  const long long HW = (long long)g->H * g->W;
  if (g->groups != 1 || g->dil_h != 1 || g->dil_w != 1 || g->C % 64 || g->C > 512 || HW % 4 || HW > 1024 ||
      (long long)Gg * Hp * Wp > kMaxWords || g->KH > kMaxK || g->KW > kMaxK || !(clamp_alpha > 0.f) ||
      ((uintptr_t)x % 16) || (long long)g->N * g->O * Ho * Wo >= (1ll << 31) || (long long)g->N * g->C * HW >= (1ll << 31))
    return LSQ_E_UNSUPPORTED;
  L1Args a = {};
  a.x = x; a.pre_scale = pre_scale; a.pre_shift = pre_shift;
  a.wbits = (const unsigned long long*)wbits; a.wsum = wsum; a.wscales = wscales; a.bias = bias; a.slope = act_slope;
  a.res_pre = res_pre; a.res_post = res_post; a.y = y; a.scales = scales;
  a.wplane_words = lsq_weight_plane_words(g);
  int e2 = 0;
  (void)frexpf(clamp_alpha, &e2);                     // the plain sweeps' rounding unit (lsq_act_quant.hip: qmagic)
  a.qmagic = ldexp(1.5, 52 + e2 - 31);
  a.alpha = clamp_alpha;
  a.N = g->N; a.C = g->C; a.H = g->H; a.W = g->W; a.O = g->O; a.KH = g->KH; a.KW = g->KW;
  a.sh = g->stride_h; a.sw = g->stride_w; a.ph = g->pad_h; a.pw = g->pad_w;
  a.Hp = Hp; a.Wp = Wp; a.Ho = Ho; a.Wo = Wo; a.Gg = Gg;
  a.opad = (g->O + 15) / 16 * 16;
  a.kw_planes = kw_planes; a.act = act;
  // Out-channel slices per sample: every slice re-reads (and re-packs) the sample's input, so no more of them than
  // keep that within about 512 KB per sample; at most one 1024-thread workgroup per CU; whole tiles of kOT channels; and the
  // slice's weight words must fit their LDS stage.
  const int taps = g->KH * g->KW;
  const long long sample_bytes = 4ll * g->C * HW;
  int nslices = 1;
  while (nslices < 8 && (long long)g->N * nslices * 2 <= 256 && sample_bytes * nslices * 2 <= (512ll << 10) &&
         g->O % (nslices * 2 * kOT) == 0 && g->O / (nslices * 2) >= 16)
    nslices *= 2;
  while ((long long)taps * Gg * ((g->O + nslices - 1) / nslices + kOT - 1) > kMaxWWords || (g->O + nslices - 1) / nslices > kMaxSlice) {
    nslices *= 2;
    if (nslices > 64) return LSQ_E_UNSUPPORTED;
  }
  a.nslices = nslices;
  a.oslice = ((g->O + nslices - 1) / nslices + kOT - 1) / kOT * kOT;
  a.nslices = (g->O + a.oslice - 1) / a.oslice;
  // sample groups: about one workgroup per CU over slices x groups (a workgroup walks over its group's samples)
  int ngroups = 256 / a.nslices;
  if (ngroups < 1) ngroups = 1;
  if (ngroups > g->N) ngroups = g->N;
  a.ngroups = ngroups;
  const long long blocks = (long long)ngroups * a.nslices;
  if (blocks > 0x7FFFFFFF) return LSQ_E_SHAPE;
  if (g->KH == 3 && g->KW == 3) hipLaunchKernelGGL(ls1_conv_kernel<true>, dim3((unsigned)blocks), dim3(kT), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(ls1_conv_kernel<false>, dim3((unsigned)blocks), dim3(kT), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
