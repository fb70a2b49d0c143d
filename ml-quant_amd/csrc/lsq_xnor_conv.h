// Shared between the two implementations of lsq_xnor_conv2d: the popcount kernel (lsq_xnor_conv.hip, every
// geometry) and the integer-MFMA kernel (lsq_xnor_mfma.hip, 3x3 taps over 64 channels).
#ifndef LSQ_XNOR_CONV_H_
#define LSQ_XNOR_CONV_H_

#include "lsq_common.h"

namespace lsq {

struct ConvArgs {
  const unsigned long long* xplanes;   // [KX][N][Gt][Hp][Wp]
  const float* xscales;                // [KX][N]
  const unsigned long long* wbits;     // [taps][Gg][Opad]  (one weight plane)
  const int* wsum;                     // [O][taps]
  const float* wscale;                 // [O]
  const float* bias;                   // [O] or null
  float* y;                            // [N][O][Ho][Wo]
  long long xplane_words;
  int N, H, W, O, KH, KW, sh, sw, ph, pw, dh, dw;
  int Gg, Gt, Hp, Wp, Ho, Wo, cg, og, og_pad, opad_total, tiles_per_group;
  int accumulate;
  int final_pass;                      // last launch of a multi-plane sequence: apply the epilogue
  int relu;                            // epilogue: y = act(conv + bias + res_pre) + res_post; LSQ_ACT_*
  const float* slope;                  // PReLU slope(s): [1] (LSQ_ACT_PRELU) or [O] (LSQ_ACT_PRELU_CHANNEL)
  const float* res_pre;                // [N][O][Ho][Wo] or null
  const float* res_post;
  int res_stream;                      // the residual operand is read with the non-temporal hint: it and the output together exceed the
                                       // Infinity Cache and this is its last use (set by the entry point, kStreamBytes)
  int dbg_no_corr;                     // tuning builds only
  int int8_mfma;                       // test hook: the int8 matrix-core kernel instead of the fp4 one (two-plane, unchained launches)
  // ---- three-stream rows (LSQ_LAYOUT_SPLIT3, include/lsq_hip.h): 0 = NCHW, else S = floats per stream of the tensor's rows
  int y_s3;                            // layout of y (and of the partial sums read back with `accumulate`)
  int res_s3;                          // layout of res_pre / res_post (both)
  int s3_hp;                           // floats per channel of a stream (S = O * s3_hp)
  // ---- chained 1-bit layers (lsq_xnor_conv2d_chain): the NEXT layer's ls-1 quantizer in this layer's epilogue, and this
  // layer's activation scale from the exact row sum the PREVIOUS layer's epilogue left
  const long long* xunits;             // [N] or null: row sum of |clamp(x)| in units of 2^e; xscale = float(units * xunit / xM)
  double xunit, xM;
  unsigned* nq_planes32;               // or null: next layer's plane [N][O/64][nq_Hp][nq_Wp] as dwords
  unsigned long long* nq_units;        // [N]: += this launch's part of the next layer's row sums (zeroed by the caller)
  const float* nq_scale;               // [O] or null: folded batch norm in front of the next quantizer
  const float* nq_shift;
  float nq_alpha;                      // next layer's clamp (> 0)
  double nq_magic, nq_inv_unit;        // 1.5 * 2^(52 + e) and 2^-e
  int nq_Hp, nq_Wp, nq_ph, nq_pw;
  int tap_xoff[64];                    // (kh*dil_h)*Wp + kw*dil_w per tap
};

// A block's tensors go from kernel to kernel through the 256 MiB Infinity Cache.  Where a launch reads a tensor for the last time
// while it writes one of the same size and the two do not fit together (the 56 x 56 layers at batch 256: 205 MB each), the dead
// tensor's loads carry the non-temporal hint so that the new lines push out consumed ones instead of lines the launch (or the
// next quantizer) still needs: measured on the ResNet-18 step, xnor conv 88 -> 86 us and the quantizer BEHIND it 93 -> 89 us per
// 56 x 56 layer; on tensors that do fit the hint costs 2-5 us per launch (28 x 28: 61 -> 67 us), hence the threshold.
constexpr long long kInfinityCacheBytes = 256ll << 20;

constexpr int kXnorMfmaNotEligible = 1;
constexpr int kXnorMfmaNoLayout = 2;     // a three-stream operand on a geometry without such a kernel: LSQ_E_UNSUPPORTED
// One launch (one weight plane x kx <= 2 activation planes) on the matrix cores; kXnorMfmaNotEligible when the
// geometry is not covered (the caller then takes the popcount kernel), else hipGetLastError().
int xnor_conv_mfma(const ConvArgs& a, int kx, int groups, hipStream_t st);

}  // namespace lsq
#endif  // LSQ_XNOR_CONV_H_
