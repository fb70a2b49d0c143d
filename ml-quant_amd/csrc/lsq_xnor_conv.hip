// Binary x binary convolution for gfx950: XNOR + popcount over channel-packed 64-bit words.
//
// Replaces F.conv2d(x_q, w_q, ...) of quant/binary/binary_conv.py:165-173 when both operands
// are sums of scaled sign planes.  For planes b_p (activation) and s_q (weight)
//     (b_p (*) s_q)[n,o,y,x] = sum_{taps, c} b*s = cg*taps - 2*popcount(b XOR s)   (exact integer)
// so the whole fp32 convolution collapses to integer popcounts plus a tiny fp32 epilogue.
//
// Mapping: one lane = one output pixel, OT output channels per lane held in registers.  The
// lanes of a wave walk consecutive pixels, so activation words are coalesced 8-byte loads
// that hit L1/L2 (a whole layer's bit planes are a few MB); every lane of a wave needs the
// SAME weight words, so they are wave-uniform and come through the scalar cache into SGPRs:
// the inner loop is v_xor_b32 + v_bcnt_u32_b32 on a VGPR and an SGPR operand, two VALU ops
// per 32 binary MACs -- the VALU popcount roofline of SURVEY.md section 8(d).
//
// Zero padding: the bit planes carry a physical halo of zero words (= all -1).  A padded tap
// therefore contributes -sum_c s[o][c][tap] instead of 0; the epilogue adds wsum[o][tap] back
// for every out-of-image tap of a border pixel, which makes the result exact.

#include "lsq_common.h"
#include <atomic>

#include "lsq_xnor_conv.h"

#include <cstdlib>

namespace lsq {
namespace {

constexpr int kDefaultOT = 16;

template <int KX, int OT>
__global__ __launch_bounds__(256) void xnor_conv_kernel(ConvArgs a) {
  // tap sums of this block's OT out-channels for the halo correction: whole-tap, per kernel row and per
  // kernel column (inclusion-exclusion), so that a border pixel pays ~3 LDS reads per channel instead of
  // up to KH*KW global reads -- every wave contains border pixels once rows are narrower than 64
  __shared__ int s_ws[64][OT];
  __shared__ int s_rs[8][OT], s_cs[8][OT];
  const long long pix_raw = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)a.N * a.Ho * a.Wo;
  const bool pvalid = pix_raw < total;
  const long long pix = pvalid ? pix_raw : total - 1;
  const int tile = blockIdx.y;
  const int grp = tile / a.tiles_per_group;
  const int t = tile - grp * a.tiles_per_group;
  const int o_pad0 = grp * a.og_pad + t * OT;
  const int o0 = grp * a.og + t * OT;
  const int o_valid = min(OT, a.og - t * OT);
  const int HoWo = a.Ho * a.Wo;
  // (a 64-bit integer division is ~100 VALU instructions: take it only for tensors past 2^31 pixels)
  const int n = total <= 0x7FFFFFFFll ? (int)((unsigned)pix / (unsigned)HoWo) : (int)(pix / HoWo);
  const int r = (int)(pix - (long long)n * HoWo);
  const int ho = (int)((unsigned)r / (unsigned)a.Wo), wo = r - ho * a.Wo;
  const long long HpWp = (long long)a.Hp * a.Wp;

  int acc[KX][OT];
#pragma unroll
  for (int p = 0; p < KX; ++p)
#pragma unroll
    for (int o = 0; o < OT; ++o) acc[p][o] = 0;

  const unsigned long long* __restrict__ xp =
      a.xplanes + ((long long)n * a.Gt + (long long)grp * a.Gg) * HpWp + (long long)(ho * a.sh) * a.Wp + wo * a.sw;
  const unsigned long long* __restrict__ wbase = a.wbits + o_pad0;

  {
    const int ntap = a.KH * a.KW;
    for (int i = threadIdx.x; i < ntap * OT; i += 256) {
      const int tp = i / OT, o = i - tp * OT;
      s_ws[tp][o] = o < o_valid ? a.wsum[(long long)(o0 + o) * ntap + tp] : 0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (a.KH + a.KW) * OT; i += 256) {
      const int k = i / OT, o = i - k * OT;
      int acc_s = 0;
      if (k < a.KH) {
        for (int kw = 0; kw < a.KW; ++kw) acc_s += s_ws[k * a.KW + kw][o];
        s_rs[k][o] = acc_s;
      } else {
        for (int kh = 0; kh < a.KH; ++kh) acc_s += s_ws[kh * a.KW + (k - a.KH)][o];
        s_cs[k - a.KH][o] = acc_s;
      }
    }
    __syncthreads();
  }

  // Loop over (channel word j, tap).  Everything address-like is a 32-bit word offset that advances by
  // a loop-invariant stride (the first version spent as many SALU instructions on 64-bit index math and
  // integer divisions as VALU instructions on popcounts); tap offsets come precomputed in the kernel
  // arguments.  The activation words of the next step are requested before this step's popcounts.
  const int taps = a.KH * a.KW;
  const int plane_stride = (int)a.xplane_words;
  const int wstep = a.Gg * a.opad_total;            // weight words between consecutive taps
  int xbase = 0;                                     // j * Hp * Wp
  unsigned long long xn[KX];
#pragma unroll
  for (int p = 0; p < KX; ++p) xn[p] = xp[p * plane_stride + a.tap_xoff[0]];
  // OT = 8: weight words one step ahead as well (scalar loads return out of order, so the only usable wait is
  // "all of them"; issued a whole step early, that wait finds them there).  With 16 channels per lane two sets
  // of weight words do not fit the 102 SGPRs.
  constexpr bool kWeightsAhead = OT <= 8;
  unsigned long long wnext[OT];
#pragma unroll
  for (int o = 0; o < OT; ++o) wnext[o] = kWeightsAhead ? wbase[o] : 0ull;
  for (int j = 0; j < a.Gg; ++j) {
    const unsigned long long* __restrict__ wp = wbase + j * a.opad_total;
    for (int tp = 0; tp < taps; ++tp) {
      unsigned long long xa[KX];
#pragma unroll
      for (int p = 0; p < KX; ++p) xa[p] = xn[p];
      {
        const bool last_tap = tp + 1 == taps;
        const int nb = last_tap ? xbase + (int)HpWp : xbase;
        const int noff = nb + a.tap_xoff[last_tap ? 0 : tp + 1];
        if (!(last_tap && j + 1 == a.Gg)) {
#pragma unroll
          for (int p = 0; p < KX; ++p) xn[p] = xp[p * plane_stride + noff];
        }
      }
      // low halves of all OT x KX words first, then the high halves: the two popcounts that feed one
      // accumulator end up 2*OT*KX instructions apart instead of back to back (v_bcnt is a 2-pass op)
      unsigned long long wv[OT];
#pragma unroll
      for (int o = 0; o < OT; ++o) wv[o] = kWeightsAhead ? wnext[o] : wp[o];   // wave-uniform: scalar loads, SGPR operands
      if (kWeightsAhead) {
        const bool last_tap = tp + 1 == taps;
        const unsigned long long* __restrict__ wq = last_tap ? wbase + (j + 1 < a.Gg ? j + 1 : j) * a.opad_total : wp + wstep;
#pragma unroll
        for (int o = 0; o < OT; ++o) wnext[o] = wq[o];       // (the very last step re-reads a valid address)
      }
#pragma unroll
      for (int o = 0; o < OT; ++o) {
#pragma unroll
        for (int p = 0; p < KX; ++p) {
          const unsigned lo = (unsigned)xa[p] ^ (unsigned)wv[o];
          asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[p][o]) : "v"(lo));   // acc = popcount(lo) + acc
        }
      }
#pragma unroll
      for (int o = 0; o < OT; ++o) {
#pragma unroll
        for (int p = 0; p < KX; ++p) {
          const unsigned hi = (unsigned)(xa[p] >> 32) ^ (unsigned)(wv[o] >> 32);
          asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[p][o]) : "v"(hi));
        }
      }
      wp += wstep;
    }
    xbase += (int)HpWp;
  }

  // border correction: out-of-image taps were computed against zero words
  int corr[OT];
#pragma unroll
  for (int o = 0; o < OT; ++o) corr[o] = 0;
  const int hi0 = ho * a.sh - a.ph, wi0 = wo * a.sw - a.pw;
  unsigned bad_h = 0, bad_w = 0;                   // bit k set: kernel row / column k falls outside the image
  for (int kh = 0; kh < a.KH; ++kh) {
    const int hi = hi0 + kh * a.dh;
    bad_h |= (hi < 0 || hi >= a.H) ? 1u << kh : 0u;
  }
  for (int kw = 0; kw < a.KW; ++kw) {
    const int wi = wi0 + kw * a.dw;
    bad_w |= (wi < 0 || wi >= a.W) ? 1u << kw : 0u;
  }
  if ((bad_h | bad_w) && !a.dbg_no_corr) {
    for (int kh = 0; kh < a.KH; ++kh) {
      if ((bad_h >> kh) & 1u) {
#pragma unroll
        for (int o = 0; o < OT; ++o) corr[o] += s_rs[kh][o];
        for (int kw = 0; kw < a.KW; ++kw) {
          if ((bad_w >> kw) & 1u) {
#pragma unroll
            for (int o = 0; o < OT; ++o) corr[o] -= s_ws[kh * a.KW + kw][o];
          }
        }
      }
    }
    for (int kw = 0; kw < a.KW; ++kw) {
      if ((bad_w >> kw) & 1u) {
#pragma unroll
        for (int o = 0; o < OT; ++o) corr[o] += s_cs[kw][o];
      }
    }
  }
  if (!pvalid) return;

  float xs[KX];
#pragma unroll
  for (int p = 0; p < KX; ++p) xs[p] = a.xscales[(long long)p * a.N + n];
  const int full = a.cg * taps;
  const long long ybase = ((long long)n * a.O + o0) * HoWo + r;
  float* yp = a.y + ybase;
  // the residual loads are issued together, before the first store: a load consumed right after it is
  // issued costs one memory latency per out-channel.  One residual array in registers (the block adds its
  // shortcut either before or after the ReLU); with both present the second is read in place.
  const bool want_pre = a.final_pass && a.res_pre, want_post = a.final_pass && a.res_post;
  const float* rsrc = want_pre ? a.res_pre : a.res_post;
  float rv[OT];
#pragma unroll
  for (int o = 0; o < OT; ++o)
    rv[o] = (want_pre || want_post) ? rsrc[ybase + (o < o_valid ? (long long)o * HoWo : 0)] : 0.f;
#pragma unroll
  for (int o = 0; o < OT; ++o) {
    if (o < o_valid) {
      // (b * s) = full + corr - 2 * popcount, per plane; y = ws * sum_p xs_p * (b_p * s) + bias: one integer
      // op, one convert and one fma per plane, one fma for scale + bias
      const int fc = full + corr[o];
      float v = xs[0] * (float)(fc - (acc[0][o] << 1));
#pragma unroll
      for (int p = 1; p < KX; ++p) v = fmaf(xs[p], (float)(fc - (acc[p][o] << 1)), v);
      const float base = a.accumulate ? yp[(long long)o * HoWo] : (a.bias ? a.bias[o0 + o] : 0.f);
      float out = fmaf(v, a.wscale[o0 + o], base);
      if (a.final_pass) {        // fused block epilogue (non-linearity and shortcut adds of resnet.py:182-190)
        if (want_pre) out += rv[o];
        if (a.relu == LSQ_ACT_RELU) out = fmaxf(out, 0.f);
        else if (a.relu >= LSQ_ACT_PRELU) out = out > 0.f ? out : a.slope[a.relu == LSQ_ACT_PRELU ? 0 : o0 + o] * out;
        if (want_post) out += want_pre ? a.res_post[ybase + (long long)o * HoWo] : rv[o];
      }
      yp[(long long)o * HoWo] = out;
    }
  }
}

template <int KX>
int launch_kx(ConvArgs a, int groups, hipStream_t st) {
  const long long total = (long long)a.N * a.Ho * a.Wo;
  // out-channels per lane: the weight plane is padded to 16 per group; 32 needs og_pad % 32 == 0
  // measured on MI355X: 8 channels per lane (59 VGPRs, 8 waves/SIMD) wins once the channel loop is long
  // (C >= 256); 16 (78 VGPRs) wins for C = 64/128 where per-lane prologue/epilogue weigh more; 32 loses
  int ot = a.Gg >= 4 ? 8 : kDefaultOT;
#ifdef LSQ_TUNE
  if (getenv("LSQ_XNOR_NOCORR")) a.dbg_no_corr = 1;
  if (const char* e = getenv("LSQ_XNOR_OT")) ot = atoi(e);
  if (ot == 32 && a.og_pad % 32) ot = 16;
#endif
  a.tiles_per_group = a.og_pad / ot;
  dim3 grid((unsigned)((total + 255) / 256), (unsigned)(groups * a.tiles_per_group));
  if (ot == 32) hipLaunchKernelGGL((xnor_conv_kernel<KX, 32>), grid, dim3(256), 0, st, a);
  else if (ot == 8) hipLaunchKernelGGL((xnor_conv_kernel<KX, 8>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((xnor_conv_kernel<KX, 16>), grid, dim3(256), 0, st, a);
  return (int)hipGetLastError();
}

}  // namespace
}  // namespace lsq

using namespace lsq;

// test hook (include/lsq_hip_debug.h, not part of the product ABI): 1 = every geometry through the popcount kernel
static std::atomic<int> g_force_popcount{0};
extern "C" int lsq_debug_xnor_impl(int impl) { return g_force_popcount.exchange(impl, std::memory_order_relaxed); }

static int xnor_conv2d_impl(const uint64_t* xplanes, int kx, const float* xscales, const uint64_t* wbits,
                            const int32_t* wsum, int kw_planes, const float* wscales, const float* bias,
                            const lsq_conv_geom* g, int relu, const float* act_slope, const float* res_pre,
                            const float* res_post, float* y, void* stream,
                            const int64_t* x_units, float x_alpha, const lsq_next_ls1* next,
                            int y_layout = LSQ_LAYOUT_NCHW, int res_layout = LSQ_LAYOUT_NCHW) {
  if (!xplanes || (!xscales && !x_units) || !wbits || !wsum || !wscales || !y) return LSQ_E_NULL;
  if (int e = check_geom(g)) return e;
  if (kx < 1 || kx > LSQ_MAX_PLANES || kw_planes < 1 || kw_planes > LSQ_MAX_PLANES) return LSQ_E_SCHEME;
  const int Ho = out_h(g), Wo = out_w(g);
  if (Ho <= 0 || Wo <= 0) return LSQ_E_SHAPE;
  ConvArgs a = {};
  a.N = g->N; a.H = g->H; a.W = g->W; a.O = g->O; a.KH = g->KH; a.KW = g->KW;
  a.sh = g->stride_h; a.sw = g->stride_w; a.ph = g->pad_h; a.pw = g->pad_w; a.dh = g->dil_h; a.dw = g->dil_w;
  a.cg = g->C / g->groups;
  a.Gg = (a.cg + 63) / 64;
  a.Gt = g->groups * a.Gg;
  a.Hp = g->H + 2 * g->pad_h; a.Wp = g->W + 2 * g->pad_w;
  a.Ho = Ho; a.Wo = Wo;
  a.og = g->O / g->groups;
  a.og_pad = (a.og + 15) / 16 * 16;
  a.opad_total = g->groups * a.og_pad;
  a.tiles_per_group = a.og_pad / 16;
  a.xplane_words = lsq_act_plane_words(g);
  if (g->KH * g->KW > 64 || g->KH > 8 || g->KW > 8 || a.xplane_words * kx >= (1ll << 31)) return LSQ_E_UNSUPPORTED;
  for (int kh = 0; kh < g->KH; ++kh)
    for (int kw = 0; kw < g->KW; ++kw) a.tap_xoff[kh * g->KW + kw] = kh * g->dil_h * a.Wp + kw * g->dil_w;
  a.bias = bias;
  a.y = y;
  if (relu < LSQ_ACT_NONE || relu > LSQ_ACT_PRELU_CHANNEL || (relu >= LSQ_ACT_PRELU && !act_slope)) return LSQ_E_SCHEME;
  a.relu = relu;
  a.slope = act_slope;
  a.res_pre = res_pre;
  a.res_post = res_post;
  a.res_stream = 2ll * 4 * (long long)g->N * g->O * Ho * Wo > kInfinityCacheBytes;
  if ((y_layout != LSQ_LAYOUT_NCHW && y_layout != LSQ_LAYOUT_SPLIT3) || (res_layout != LSQ_LAYOUT_NCHW && res_layout != LSQ_LAYOUT_SPLIT3))
    return LSQ_E_SCHEME;
  if (!res_pre && !res_post) res_layout = LSQ_LAYOUT_NCHW;
  const bool layouts = y_layout != LSQ_LAYOUT_NCHW || res_layout != LSQ_LAYOUT_NCHW;
  if (layouts) {
    // three-stream rows (include/lsq_hip.h): the integer-MFMA kernel only
    const int64_t S = lsq_split3_stream_floats(g->O, Ho, Wo);
    if (S <= 0 || 3 * S * g->N >= (1ll << 31) || g_force_popcount.load(std::memory_order_relaxed) == 1 || x_units || next) return LSQ_E_UNSUPPORTED;
    a.s3_hp = (int)(S / g->O);
    a.y_s3 = y_layout == LSQ_LAYOUT_SPLIT3 ? (int)S : 0;
    a.res_s3 = res_layout == LSQ_LAYOUT_SPLIT3 ? (int)S : 0;
  }
  const bool chained = x_units != nullptr || next != nullptr;
  if (chained) {
    // chained 1-bit layers run on the integer-MFMA kernel only (3x3, C in {64, 128, 256, 512}); one activation plane
    if (kx != 1) return LSQ_E_SCHEME;
    if (x_units) {
      if (!(x_alpha > 0.f)) return LSQ_E_UNSUPPORTED;
      int e2 = 0;
      (void)frexpf(x_alpha, &e2);
      a.xunits = (const long long*)x_units;
      a.xunit = ldexp(1.0, e2 - 31);
      a.xM = (double)g->C * g->H * g->W;
    }
    if (next) {
      if (!next->planes || !next->sum_units) return LSQ_E_NULL;
      if ((next->pre_scale == nullptr) != (next->pre_shift == nullptr)) return LSQ_E_NULL;
      if (!(next->clamp_alpha > 0.f) || g->O % 64 || next->pad_h < 0 || next->pad_w < 0) return LSQ_E_UNSUPPORTED;
      int e2 = 0;
      (void)frexpf(next->clamp_alpha, &e2);
      a.nq_planes32 = (unsigned*)next->planes;
      a.nq_units = (unsigned long long*)next->sum_units;
      a.nq_scale = next->pre_scale;
      a.nq_shift = next->pre_shift;
      a.nq_alpha = next->clamp_alpha;
      a.nq_magic = ldexp(1.5, 52 + e2 - 31);
      a.nq_inv_unit = ldexp(1.0, 31 - e2);
      a.nq_ph = next->pad_h; a.nq_pw = next->pad_w;
      a.nq_Hp = Ho + 2 * next->pad_h; a.nq_Wp = Wo + 2 * next->pad_w;
      if ((long long)g->N * (g->O / 64) * a.nq_Hp * a.nq_Wp >= (1ll << 30)) return LSQ_E_UNSUPPORTED;
    }
  }
  const long long wplane_words = lsq_weight_plane_words(g);
  const int taps = g->KH * g->KW;
  hipStream_t st = (hipStream_t)stream;
  bool first = true;
  for (int q = 0; q < kw_planes; ++q) {
    for (int p0 = 0; p0 < kx; p0 += 2) {
      const int np = (kx - p0) >= 2 ? 2 : 1;
      a.xplanes = (const unsigned long long*)xplanes + (long long)p0 * a.xplane_words;
      a.xscales = xscales + (long long)p0 * g->N;
      a.wbits = (const unsigned long long*)wbits + (long long)q * wplane_words;
      a.wsum = wsum + (long long)q * g->O * taps;
      a.wscale = wscales + (long long)q * g->O;
      a.accumulate = first ? 0 : 1;
      a.final_pass = (q == kw_planes - 1 && p0 + np >= kx) ? 1 : 0;
      const int impl = g_force_popcount.load(std::memory_order_relaxed);      // 0: matrix cores (fp4), 1: popcount kernel, 2: matrix cores (int8)
      a.int8_mfma = impl == 2;
      int e = (impl == 1 && !chained) ? kXnorMfmaNotEligible : xnor_conv_mfma(a, np, g->groups, st);
      if (e == kXnorMfmaNoLayout || (e == kXnorMfmaNotEligible && (chained || layouts))) return LSQ_E_UNSUPPORTED;
      if (e == kXnorMfmaNotEligible) e = np == 2 ? launch_kx<2>(a, g->groups, st) : launch_kx<1>(a, g->groups, st);
      if (e) return e;
      first = false;
    }
  }
  return LSQ_OK;
}

extern "C" int lsq_xnor_conv2d(const uint64_t* xplanes, int kx, const float* xscales, const uint64_t* wbits,
                               const int32_t* wsum, int kw_planes, const float* wscales, const float* bias,
                               const lsq_conv_geom* g, int relu, const float* act_slope, const float* res_pre,
                               const float* res_post,
                               float* y, void* stream) {
  return xnor_conv2d_impl(xplanes, kx, xscales, wbits, wsum, kw_planes, wscales, bias, g, relu, act_slope, res_pre, res_post, y,
                          stream, nullptr, -1.f, nullptr);
}

extern "C" int lsq_xnor_conv2d_layout(const uint64_t* xplanes, int kx, const float* xscales, const uint64_t* wbits,
                                      const int32_t* wsum, int kw_planes, const float* wscales, const float* bias,
                                      const lsq_conv_geom* g, int relu, const float* act_slope, const float* res_pre,
                                      const float* res_post, int res_layout, float* y, int y_layout, void* stream) {
  return xnor_conv2d_impl(xplanes, kx, xscales, wbits, wsum, kw_planes, wscales, bias, g, relu, act_slope, res_pre, res_post, y,
                          stream, nullptr, -1.f, nullptr, y_layout, res_layout);
}

extern "C" int64_t lsq_split3_stream_floats(int64_t C, int64_t H, int64_t W) {
  if (C <= 0 || H <= 0 || W <= 0 || (H * W) % 3 != 1) return -1;
  // per channel and stream ceil(H W / 3) values (+ 3: the quantizer's last items read a float4 that starts at the last one),
  // rounded up to whole 128-byte lines
  return C * (((H * W + 2) / 3 + 3 + 31) / 32 * 32);
}

extern "C" int lsq_layout_support(const lsq_conv_geom* g, int scheme, int kx) {
  if (check_geom(g)) return 0;
  int mask = 0;
  const int cg = g->C / g->groups;
  // bit 0: the single-launch solving quantizer (lsq_act_fused.hip) on three-stream rows
  const long long M = (long long)g->C * g->H * g->W;
  if ((scheme == LSQ_SCHEME_LS2 || scheme == LSQ_SCHEME_LST) && lsq_split3_stream_floats(g->C, g->H, g->W) > 0 && cg % 64 == 0 &&
      g->C <= 1024 && M % 4 == 0 && (long long)g->H * g->W >= 4 && (lsq_split3_stream_floats(g->C, g->H, g->W) / 4 + 511) / 512 <= 33)
    mask |= 1;
  // bits 1, 2: the integer-MFMA convolution's three-stream kernels (lsq_xnor_mfma.hip)
  const int Ho = out_h(g), Wo = out_w(g);
  if (g->groups == 1 && g->KH == 3 && g->KW == 3 && g->dil_w == 1 && g->O % 32 == 0 && kx == 2 && (cg == 64 || cg == 128) &&
      Ho > 0 && Wo > 0 && lsq_split3_stream_floats(g->O, Ho, Wo) > 0 && 3 * lsq_split3_stream_floats(g->O, Ho, Wo) * g->N < (1ll << 30) &&
      g_force_popcount.load(std::memory_order_relaxed) != 1)
    mask |= 2 | 4;
  return mask;
}

extern "C" int lsq_xnor_conv2d_chain(const uint64_t* xplanes, const float* xscales, const int64_t* x_units, float x_alpha,
                                     const uint64_t* wbits, const int32_t* wsum, int kw_planes, const float* wscales,
                                     const float* bias, const lsq_conv_geom* g, int relu, const float* act_slope,
                                     const float* res_pre, const float* res_post, const lsq_next_ls1* next, float* y,
                                     void* stream) {
  if (x_units && xscales) return LSQ_E_SCHEME;               // one source for the activation scale
  return xnor_conv2d_impl(xplanes, 1, xscales, wbits, wsum, kw_planes, wscales, bias, g, relu, act_slope, res_pre, res_post, y,
                          stream, x_units, x_alpha, next);
}
