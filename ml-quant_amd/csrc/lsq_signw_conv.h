// Shared declarations of the full-precision-activation x sign-weight convolution kernels
// (lsq_signw_conv.hip: general kernels; lsq_signw_lean.hip: the 3x3 fast path with weights expanded once per eval session).
#pragma once

#include "lsq_common.h"

namespace lsq {
namespace signw {

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
  int relu;                            // epilogue: y = act(conv + bias + res_pre) + res_post; LSQ_ACT_*
  const float* slope;                  // PReLU slope(s): [1] or [O]
  const float* res_pre;                // [N][O][Ho][Wo] or null
  const float* res_post;
};

union Frag {
  unsigned u[4];
  bf16x8 v;
};

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) unsigned short u16x2;

// x = hi + lo in bf16: hi = bf16(x) (v_cvt_pk_bf16_f32, round to nearest even), lo = bf16(x - hi).
// x - hi is exact in fp32, so |x - hi - lo| <= 2^-18 |x|.
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& hi, unsigned& lo) {
  const f32x2 v = {x0, x1};
  const bf16x2 h = __builtin_convertvector(v, bf16x2);
  const f32x2 r = v - __builtin_convertvector(h, f32x2);
  const bf16x2 l = __builtin_convertvector(r, bf16x2);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}

constexpr int kPC = 16;                          // channels per chunk (one MFMA k-step)
constexpr int kPRow = 32;                        // bytes per LDS row (16 bf16)

// LDS rows of 32 bytes without padding: the two 16-byte halves of row r are swapped when bit 3 of r is set, which
// makes the ds_read_b128 lane groups (16 lanes on consecutive rows) hit 16 distinct 4-bank groups.
__device__ __forceinline__ int swz(int row, int half) { return row * kPRow + ((half ^ ((row >> 3) & 1)) << 4); }

// Epilogue shared by all the kernels: lane = pixel column, registers = out-channel rows (coalesced 128-byte
// stores); y = relu(u * acc + bias|y + res_pre) + res_post on the last weight plane.  Off = unsigned when
// the host has checked 4*N*O*Ho*Wo < 2^32 (one vector add per output address), else size_t.
template <typename Off, int BM, int BN, int TM, int TN>
__device__ __forceinline__ void store_tiles(const SwArgs& a, const f32x16 (&acc)[TM][TN], int ptile, int t, int o0,
                                              int wm, int wn, int col, int kh8) {
  const int HoWo = a.Ho * a.Wo;
  const long long total = (long long)a.N * HoWo;
  Off pbase[TN];                                       // byte offset of (n, out-channel o0 + 4*kh8, pixel)
  bool pok[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const long long pix = (long long)ptile * BN + (wn * TN + j) * 32 + col;
    pok[j] = pix < total;
    const int n = pok[j] ? (int)(pix / HoWo) : 0;
    const int r = pok[j] ? (int)(pix - (long long)n * HoWo) : 0;
    pbase[j] = (Off)4 * ((Off)(n * a.O + o0 + 4 * kh8) * (Off)HoWo + (Off)r);
  }
  char* yb = reinterpret_cast<char*>(a.y);
  const char* rpre = reinterpret_cast<const char*>(a.res_pre);
  const char* rpost = reinterpret_cast<const char*>(a.res_post);
  const bool want_pre = a.final_pass && a.res_pre, want_post = a.final_pass && a.res_post;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int qh = 0; qh < 2; ++qh) {
      // batch of 8 out-channel rows x TN pixel tiles: every load of the batch (previous partial sum,
      // residuals, scale, bias) is issued before the first use -- a load consumed right after it is issued
      // costs one memory latency per output
      float ws[8], bs[8], sl[8], prev[8][TN], r1[8][TN], r2[8][TN];
      Off yo[8][TN];
      bool ok[8][TN];
#pragma unroll
      for (int qq = 0; qq < 8; ++qq) {
        const int q = qh * 8 + qq;
        const int olu = (wm * TM + i) * 32 + (q & 3) + 8 * (q >> 2);      // wave-uniform part of the row
        const int ol = olu + 4 * kh8;                                     // C/D layout of the 32x32 MFMA
        const bool rok = t * BM + ol < a.og;
        const int o = o0 + (rok ? ol : 0);
        ws[qq] = a.wscale[o];
        bs[qq] = (!a.accumulate && a.bias) ? a.bias[o] : 0.f;
        sl[qq] = (a.final_pass && a.relu >= LSQ_ACT_PRELU) ? a.slope[a.relu == LSQ_ACT_PRELU ? 0 : o] : 0.f;
        const Off rowoff = (Off)4 * (Off)olu * (Off)HoWo;                 // scalar
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          ok[qq][j] = rok && pok[j];
          yo[qq][j] = ok[qq][j] ? pbase[j] + rowoff : (Off)0;
          prev[qq][j] = a.accumulate ? *reinterpret_cast<const float*>(yb + yo[qq][j]) : 0.f;
          r1[qq][j] = want_pre ? *reinterpret_cast<const float*>(rpre + yo[qq][j]) : 0.f;
          r2[qq][j] = want_post ? *reinterpret_cast<const float*>(rpost + yo[qq][j]) : 0.f;
        }
      }
#pragma unroll
      for (int qq = 0; qq < 8; ++qq) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          float out = (a.accumulate ? prev[qq][j] : bs[qq]) + acc[i][j][qh * 8 + qq] * ws[qq];
          if (a.final_pass) {
            out += r1[qq][j];
            if (a.relu == LSQ_ACT_RELU) out = fmaxf(out, 0.f);
            else if (a.relu >= LSQ_ACT_PRELU) out = out > 0.f ? out : sl[qq] * out;
            out += r2[qq][j];
          }
          if (ok[qq][j]) *reinterpret_cast<float*>(yb + yo[qq][j]) = out;
        }
      }
    }
  }
}


// lsq_signw_lean.hip
// bytes of the expanded weight operand of `planes` weight planes, 0 when the geometry is outside the fast path
long long lean_weight_bytes(const lsq_conv_geom* g, int planes);
int lean_prepare(const uint64_t* wbits, int planes, const lsq_conv_geom* g, void* wprep, hipStream_t st);
// LSQ_E_UNSUPPORTED when the call is outside the fast path (the caller then takes the general kernels)
int lean_launch(const SwArgs& a, const void* wprep, int plane, const lsq_conv_geom* g, hipStream_t st);

}  // namespace signw
}  // namespace lsq
