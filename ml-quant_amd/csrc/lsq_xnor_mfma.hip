// Binary x binary 3x3 convolution on the integer matrix cores of gfx950 (v_mfma_i32_32x32x32_i8), for the layers
// whose whole weight operand fits the register file of a wave: C = 64 (72 VGPRs of weights per 32 out-channels).
// Second implementation of F.conv2d(x_q, w_q, ...) of quant/binary/binary_conv.py:165-173 next to the popcount
// kernel (lsq_xnor_conv.hip); same operands, same epilogue arithmetic, bit-identical output.
//
// Why only these layers, and why it wins there: the popcount kernel spends two VALU instructions per 32 binary
// MACs and, with 64 channels, 38 % of its instructions outside the popcount core.  The matrix cores do 32768 MACs
// per instruction, but only if their operands arrive as bytes -- the earlier attempt (bits expanded through an LDS
// table into an im2col patch, both operands read from LDS) was fed at a third of the rate it needed.  Here
//   * the WEIGHTS are expanded once per workgroup and stay in registers for its whole life (weight-stationary:
//     every wave walks many pixel tiles with the same 9 x 2 operand fragments);
//   * the ACTIVATION operand is built from the packed word in registers with ONE v_and_b32 per four channels:
//     an MFMA sums over k in any order as long as both operands agree on it, so k-slot (register q, byte b) of a
//     lane is defined as channel q + 8 b of the lane's 32-bit half of the word, and `word & (0x01010101 << q)`
//     IS that register -- its bytes hold 0 or 2^q.  The weight byte of the same slot is +-(64 >> q), so every
//     product is +-64 or 0 and the accumulator is 64 * sum_c s_c [b_c = +1], an exact integer.  (q = 7 would be
//     the sign bit of an int8: that register is `(word >> 7) & 0x01010101` against weights of +-64.)
//   * no LDS and no barrier inside the tile loop; the words of the next tile are requested before this tile's
//     MFMAs (the popcount kernel's access pattern: lanes = consecutive pixels = consecutive words).
//
// (b * s) = 2 * sum_c s_c [b_c = +1] - sum_c s_c over the taps inside the image; halo words are zero bits and
// contribute nothing to the first term, the second comes from the same wsum tables as the popcount kernel.
//
// Tile = 32 consecutive output pixels (flat over n, ho, wo) x 32 out-channels per wave; D[o][pixel]: a lane holds
// ONE pixel (column lane & 31) and 16 out-channels (rows (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)).

#include "lsq_xnor_conv.h"

namespace lsq {
namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr unsigned kM0 = 0x01010101u;

template <int KX, int GG, int TAPS>
__global__ __launch_bounds__(256, 2) void xnor_mfma_kernel(ConvArgs a) {
  constexpr int NF = TAPS * GG * 2;              // 16-byte operand fragments per lane: (tap, word, half of the dword's bits)
  constexpr int NW = TAPS * GG * KX;             // activation dwords per lane and tile
  __shared__ v4i s_w[NF][64];
  __shared__ int s_ws[TAPS][32], s_rs[8][32], s_cs[8][32], s_tot[32];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int col = lane & 31, hh = lane >> 5;
  const int o0 = blockIdx.y * 32;

  // ---- once per workgroup: weight fragments and halo tables of its 32 out-channels -------------------------------
  for (int e = tid; e < NF * 256; e += 256) {
    const int r = e & 3, L = (e >> 2) & 63, f = e >> 8;
    const int q = 4 * (f & 1) + r;
    const unsigned long long w = a.wbits[(long long)(f >> 1) * a.opad_total + o0 + (L & 31)];   // [tap][word][O]
    const unsigned d = (L >> 5) ? (unsigned)(w >> 32) : (unsigned)w;
    const int mag = q < 7 ? (64 >> q) : 64;
    unsigned out = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) out |= (unsigned)(((d >> (q + 8 * b)) & 1u ? mag : -mag) & 0xFF) << (8 * b);
    reinterpret_cast<unsigned*>(&s_w[f][L])[r] = out;
  }
  for (int i = tid; i < TAPS * 32; i += 256) s_ws[i >> 5][i & 31] = a.wsum[(long long)(o0 + (i & 31)) * TAPS + (i >> 5)];
  __syncthreads();
  for (int i = tid; i < (a.KH + a.KW + 1) * 32; i += 256) {
    const int k = i >> 5, o = i & 31;
    int acc_s = 0;
    if (k < a.KH) {
      for (int kw = 0; kw < a.KW; ++kw) acc_s += s_ws[k * a.KW + kw][o];
      s_rs[k][o] = acc_s;
    } else if (k < a.KH + a.KW) {
      for (int kh = 0; kh < a.KH; ++kh) acc_s += s_ws[kh * a.KW + (k - a.KH)][o];
      s_cs[k - a.KH][o] = acc_s;
    } else {
      for (int tp = 0; tp < TAPS; ++tp) acc_s += s_ws[tp][o];
      s_tot[o] = acc_s;
    }
  }
  __syncthreads();
  v4i wreg[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) wreg[f] = s_w[f][lane];

  // ---- tiles ------------------------------------------------------------------------------------------------------
  const long long total = (long long)a.N * a.Ho * a.Wo;
  const int ntiles = (int)((total + 31) >> 5);
  const int HoWo = a.Ho * a.Wo;
  const int HpWp = a.Hp * a.Wp;
  const unsigned* __restrict__ xd = reinterpret_cast<const unsigned*>(a.xplanes);
  const unsigned plane_stride = 2u * (unsigned)a.xplane_words;
  const int tstride = gridDim.x * 4;

  struct Pix {
    int n, r, ho, wo;
  };
  auto decode = [&](int tile) {
    const unsigned p = (unsigned)min((long long)tile * 32 + col, total - 1);
    Pix px;
    px.n = (int)(p / (unsigned)HoWo);
    px.r = (int)(p - (unsigned)px.n * (unsigned)HoWo);
    px.ho = (int)((unsigned)px.r / (unsigned)a.Wo);
    px.wo = px.r - px.ho * a.Wo;
    return px;
  };
  auto request = [&](const Pix& px, unsigned (&x)[NW]) {
    // dword index of (word, half) = 2 * word + hh; words: [plane][n][GG][Hp][Wp]
    const unsigned base = 2u * (unsigned)((px.n * GG * a.Hp + px.ho * a.sh) * a.Wp + px.wo * a.sw) + (unsigned)hh;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int j = 0; j < GG; ++j)
#pragma unroll
        for (int p = 0; p < KX; ++p)
          x[(t * GG + j) * KX + p] = xd[base + 2u * (unsigned)(a.tap_xoff[t] + j * HpWp) + (unsigned)p * plane_stride];
  };

  int tile = blockIdx.x * 4 + wid;
  if (tile >= ntiles) return;
  Pix cur = decode(tile);
  unsigned xc[NW];
  request(cur, xc);
  for (;;) {
    const int nxt = tile + tstride;
    const bool more = nxt < ntiles;
    Pix nx = cur;
    unsigned xn[NW];
    if (more) {
      nx = decode(nxt);
      request(nx, xn);
    }

    v16i acc[KX];
#pragma unroll
    for (int p = 0; p < KX; ++p)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[p][i] = 0;
#pragma unroll
    for (int tj = 0; tj < TAPS * GG; ++tj) {
#pragma unroll
      for (int p = 0; p < KX; ++p) {
        const unsigned d = xc[tj * KX + p];
        v4i b0, b1;
        b0[0] = (int)(d & kM0);
        b0[1] = (int)(d & (kM0 << 1));
        b0[2] = (int)(d & (kM0 << 2));
        b0[3] = (int)(d & (kM0 << 3));
        b1[0] = (int)(d & (kM0 << 4));
        b1[1] = (int)(d & (kM0 << 5));
        b1[2] = (int)(d & (kM0 << 6));
        b1[3] = (int)((d >> 7) & kM0);
        acc[p] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wreg[2 * tj], b0, acc[p], 0, 0, 0);
        acc[p] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wreg[2 * tj + 1], b1, acc[p], 0, 0, 0);
      }
    }

    // ---- epilogue of this tile: the popcount kernel's arithmetic on the same integers -> the same floats --------
    const bool pvalid = (long long)tile * 32 + col < total;
    int corr[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) corr[i] = 0;
    const int hi0 = cur.ho * a.sh - a.ph, wi0 = cur.wo * a.sw - a.pw;
    unsigned bad_h = 0, bad_w = 0;
    for (int kh = 0; kh < a.KH; ++kh) {
      const int hi = hi0 + kh * a.dh;
      bad_h |= (hi < 0 || hi >= a.H) ? 1u << kh : 0u;
    }
    for (int kw = 0; kw < a.KW; ++kw) {
      const int wi = wi0 + kw * a.dw;
      bad_w |= (wi < 0 || wi >= a.W) ? 1u << kw : 0u;
    }
    const int ob = 4 * hh;                       // the lane's out-channel of register i: ob + (i & 3) + 8 (i >> 2)
    if (bad_h | bad_w) {
      for (int kh = 0; kh < a.KH; ++kh) {
        if ((bad_h >> kh) & 1u) {
#pragma unroll
          for (int i = 0; i < 16; ++i) corr[i] += s_rs[kh][ob + (i & 3) + 8 * (i >> 2)];
          for (int kw = 0; kw < a.KW; ++kw) {
            if ((bad_w >> kw) & 1u) {
#pragma unroll
              for (int i = 0; i < 16; ++i) corr[i] -= s_ws[kh * a.KW + kw][ob + (i & 3) + 8 * (i >> 2)];
            }
          }
        }
      }
      for (int kw = 0; kw < a.KW; ++kw) {
        if ((bad_w >> kw) & 1u) {
#pragma unroll
          for (int i = 0; i < 16; ++i) corr[i] += s_cs[kw][ob + (i & 3) + 8 * (i >> 2)];
        }
      }
    }
    if (pvalid) {
      float xs[KX];
#pragma unroll
      for (int p = 0; p < KX; ++p) xs[p] = a.xscales[(long long)p * a.N + cur.n];
      const long long ybase = ((long long)cur.n * a.O + o0 + ob) * HoWo + cur.r;
      float* __restrict__ yp = a.y + ybase;
      const bool want_pre = a.final_pass && a.res_pre, want_post = a.final_pass && a.res_post;
      const float* rsrc = (want_pre ? a.res_pre : a.res_post) + ybase;
      float rv[16], basev[16], wsv[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int oo = (i & 3) + 8 * (i >> 2);
        rv[i] = (want_pre || want_post) ? rsrc[(long long)oo * HoWo] : 0.f;
        basev[i] = a.accumulate ? yp[(long long)oo * HoWo] : (a.bias ? a.bias[o0 + ob + oo] : 0.f);
        wsv[i] = a.wscale[o0 + ob + oo];
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int oo = (i & 3) + 8 * (i >> 2);
        const int fc = corr[i] - s_tot[ob + oo];
        float v = xs[0] * (float)(fc + (acc[0][i] >> 5));
#pragma unroll
        for (int p = 1; p < KX; ++p) v = fmaf(xs[p], (float)(fc + (acc[p][i] >> 5)), v);
        float out = fmaf(v, wsv[i], basev[i]);
        if (a.final_pass) {
          if (want_pre) out += rv[i];
          if (a.relu) out = fmaxf(out, 0.f);
          if (want_post) out += want_pre ? a.res_post[ybase + (long long)oo * HoWo] : rv[i];
        }
        yp[(long long)oo * HoWo] = out;
      }
    }

    if (!more) break;
    tile = nxt;
    cur = nx;
#pragma unroll
    for (int i = 0; i < NW; ++i) xc[i] = xn[i];
  }
}

template <int KX>
int launch(const ConvArgs& a, hipStream_t st) {
  const long long total = (long long)a.N * a.Ho * a.Wo;
  const long long ntiles = (total + 31) >> 5;
  const int n_ot = a.O / 32;
  // two workgroups per CU, each wave strides over the tiles of its out-channel tile
  long long gx = (512 + n_ot - 1) / n_ot;
  gx = gx < 1 ? 1 : gx;
  if (gx * 4 > ntiles) gx = (ntiles + 3) / 4;
  hipLaunchKernelGGL((xnor_mfma_kernel<KX, 1, 9>), dim3((unsigned)gx, (unsigned)n_ot), dim3(256), 0, st, a);
  return (int)hipGetLastError();
}

}  // namespace

int xnor_conv_mfma(const ConvArgs& a, int kx, int groups, hipStream_t st) {
  const long long total = (long long)a.N * a.Ho * a.Wo;
  if (groups != 1 || a.cg != 64 || a.KH != 3 || a.KW != 3 || a.O % 32 || total >= (1ll << 31) - 64 ||
      a.xplane_words * kx >= (1ll << 30))
    return kXnorMfmaNotEligible;
  return kx == 2 ? launch<2>(a, st) : launch<1>(a, st);
}

}  // namespace lsq
