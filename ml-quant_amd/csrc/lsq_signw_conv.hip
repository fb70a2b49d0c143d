// fp-activation x sign-weight convolution (placeholder until the MFMA kernel lands).
#include "lsq_common.h"

extern "C" int lsq_signw_conv2d(const float* x, float clamp_alpha, const uint64_t* wbits, int kw_planes,
                                const float* wscales, const float* bias, const lsq_conv_geom* g, float* y,
                                void* stream) {
  (void)x; (void)clamp_alpha; (void)wbits; (void)kw_planes; (void)wscales; (void)bias; (void)g; (void)y; (void)stream;
  return LSQ_E_UNSUPPORTED;
}
