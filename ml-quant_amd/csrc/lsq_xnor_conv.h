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
  int dbg_no_corr;                     // tuning builds only
  int tap_xoff[64];                    // (kh*dil_h)*Wp + kw*dil_w per tap
};

constexpr int kXnorMfmaNotEligible = 1;
// One launch (one weight plane x kx <= 2 activation planes) on the matrix cores; kXnorMfmaNotEligible when the
// geometry is not covered (the caller then takes the popcount kernel), else hipGetLastError().
int xnor_conv_mfma(const ConvArgs& a, int kx, int groups, hipStream_t st);

}  // namespace lsq
#endif  // LSQ_XNOR_CONV_H_
