// Single-launch activation quantizer (lsq_act_fused.hip), called by lsq_act_quant for the LS-2 / LS-T
// solver schemes when the row's sub-sample fits the register file of one workgroup.
#pragma once

#include "lsq_common.h"

namespace lsq {

struct FusedArgs {
  const float* x;
  long long row_elems;      // M = C*H*W
  int C, H, W, cg, Gg, Gt, Hp, Wp, pad_h, pad_w;
  float alpha;
  const float* pre_scale;   // [C] or null (folded eval batch norm)
  const float* pre_shift;
  unsigned long long* planes;
  long long plane_words;    // words of one plane (all rows)
  long long row_words;      // words of one row of one plane
  float* scales;            // [2][N]
  int N;
  int ternary;
  int debug;                // developer / test switches: 1 every flagged bin through the block path, 2 a 2048-key list,
                            // 4 no windowed level-1 histogram (the round-2..4 solve only), 8 windowed histogram built, then dropped
  unsigned win_k0;          // windowed level-1 histogram (set by fused_act_quant): fine bin of key k >= win_k0 is
  int win_sh;               // (k - win_k0) >> win_sh, 8192 bins up to the top of the clamp value's binade; 0: off
  const float* forced;      // [2][N] scales given by the caller (moving-average inference): no solve, planes only
  int* trace;               // test hook: chosen sorted position per row (lsq_debug_solver_trace), or null
  int x_s3;                 // 0: rows in NCHW order; else S, the floats per stream of three-stream rows (LSQ_LAYOUT_SPLIT3):
                            // element (c, pixel) at ((c + pixel) % 3) * S + c * x_hp + pixel / 3, rows 3 S floats apart
                            // (fused_act_quant_s3 only)
  int x_hp;                 // ... floats per channel of a stream (x_s3 = C * x_hp; a multiple of 32)
  int greedy;               // gf-2 (quantization.py:118-148 with k = 2): v1 = mean |x| instead of the solve; planes and
                            // v2 = mean |x - v1 b1| are the 2-bit least-squares scheme's
};

constexpr int kFusedNotEligible = 1;   // the shape is left to the streaming three-kernel path

// LSQ_OK, an hipError_t, or kFusedNotEligible (nothing launched).  skip is the reference's 3.
int fused_act_quant(const FusedArgs& a, hipStream_t st);
// the same for rows in the three-stream layout (a.x_s3 != 0; LS-2 / LS-T solve under a symmetric clamp, skip 3)
int fused_act_quant_s3(const FusedArgs& a, hipStream_t st);

}  // namespace lsq
