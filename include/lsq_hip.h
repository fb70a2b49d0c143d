/*
 * lsq_hip.h -- C ABI of the MI355X (gfx950) least-squares binary quantized forward path.
 *
 * The reference (apple/ml-quant) is pure Python on stock torch ops and has no FFI of its
 * own; each entry point below replaces the Python functions cited next to it (paths are
 * relative to the reference root).  INTEGRATION.md shows the ctypes binding a maintainer
 * of the reference would add to `QuantConv2d.forward` (quant/binary/binary_conv.py:161-173).
 *
 * Conventions
 *  - All pointers are DEVICE pointers owned by the caller (torch allocator); the library
 *    allocates nothing and is re-entrant per stream.  The entry points below keep no state between calls; the only
 *    process-wide variables in the library are the three test hooks of lsq_hip_debug.h (default off), which select
 *    between implementations that return identical results.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).
 *  - Every function returns 0 on success, a negative LSQ_E_* code for an argument error,
 *    or a positive hipError_t if a launch failed.
 *  - Activations are NCHW fp32, one "row" = one sample (quantization.py:77); weights are
 *    [O][C/groups][KH][KW] fp32, one row = one output channel.
 *
 * Packed sign planes ("bit tensors")
 *  - Activation plane p of a batch: uint64 words [p][N][Gt][Hp][Wp] with
 *        cg = C / groups,  Gg = ceil(cg / 64),  Gt = groups * Gg,
 *        Hp = H + 2*pad_h, Wp = W + 2*pad_w.
 *    Word (n, grp*Gg + j, h + pad_h, w + pad_w) holds, in bit b, the sign of channel
 *    grp*cg + 64*j + b at pixel (h, w): 1 <=> value >= 0 (so sign(+-0) = +1,
 *    quant/binary/ste.py:16-18), 0 for channels beyond cg.  The halo words are never
 *    written: the caller zero-fills the buffer once (zero words = all -1, corrected
 *    analytically in the conv epilogue so padded taps contribute exactly 0).
 *  - Weight plane q: uint64 words [q][KH*KW][Gg][O]; bit b of word (tap, j, o) is the sign of
 *    w[o][64*j + b][tap]; int32 tap sums wsum[q][O][KH*KW] = sum_c sign(w[o][c][tap]).
 */
#ifndef LSQ_HIP_H_
#define LSQ_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSQ_ABI_VERSION 11

/* fused non-linearity of the convolution epilogues (quant/models/resnet.py non_linearity_map) */
#define LSQ_ACT_NONE 0
#define LSQ_ACT_RELU 1
#define LSQ_ACT_PRELU 2            /* x > 0 ? x : slope[0] * x */
#define LSQ_ACT_PRELU_CHANNEL 3    /* x > 0 ? x : slope[o] * x */

/* quantization schemes (quant/binary/binary_conv.py:99-101) */
enum {
  LSQ_SCHEME_LS1 = 1, /* quantizer_ls_1       quantization.py:35-56  */
  LSQ_SCHEME_LS2 = 2, /* quantizer_ls_2       quantization.py:59-92  */
  LSQ_SCHEME_LST = 3, /* quantizer_ls_ternary quantization.py:95-115 */
  LSQ_SCHEME_GF = 4   /* quantizer_gf (k planes) quantization.py:118-148 */
};

enum {
  LSQ_OK = 0,
  LSQ_E_NULL = -1,      /* required pointer is NULL */
  LSQ_E_SHAPE = -2,     /* non-positive / inconsistent dimension */
  LSQ_E_SCHEME = -3,    /* unknown scheme or plane count */
  LSQ_E_TOO_LONG = -4,  /* a sub-sampled row has >= 2^22 elements */
  LSQ_E_WORKSPACE = -5, /* workspace too small */
  LSQ_E_UNSUPPORTED = -6
};

#define LSQ_MAX_PLANES 8

int lsq_abi_version(void);
const char* lsq_error_string(int code);

/* Geometry of one QuantConv2d call (nn.Conv2d arguments, binary_conv.py:165-173). */
typedef struct lsq_conv_geom {
  int32_t N, C, H, W;      /* input  */
  int32_t O, KH, KW;       /* weight [O][C/groups][KH][KW] */
  int32_t stride_h, stride_w;
  int32_t pad_h, pad_w;
  int32_t dil_h, dil_w;
  int32_t groups;
} lsq_conv_geom;

/*
 * Layouts of fp32 activation tensors (ABI v11).
 *  LSQ_LAYOUT_NCHW    [N][C][H][W], what the reference's modules produce.
 *  LSQ_LAYOUT_SPLIT3  "three-stream rows": a sample's row of M = C*H*W values (flat NCHW index e = c*H*W + p, p = h*W + w) is
 *                     stored as three streams of S = lsq_split3_stream_floats(C, H, W) = C * hp floats: element (c, p) in
 *                     stream s = e % 3 = (c + p) % 3 at s * S + c * hp + p / 3, i.e. per stream and channel a block of hp
 *                     floats (ceil(H*W / 3) + 3 rounded up to whole 128-byte lines; the tail of a block is padding of
 *                     any content); rows are 3*S floats apart.  Why: the v1 search of quantizer_ls_2 /
 *                     quantizer_ls_ternary looks at every third element of the row (quant/binary/quantization.py:63,
 *                     skip = 3; optimal.py:121-155) -- in NCHW that is a third of every cache line, i.e. a full read of
 *                     the row; here it is stream 0, one contiguous third.  A tensor that goes from a convolution's
 *                     epilogue to the next layer's quantizer (and to a later epilogue as a residual operand) can stay in
 *                     this layout from its producer to its last consumer: lsq_xnor_conv2d_layout writes and reads it,
 *                     lsq_act_quant_layout and lsq_pointwise_conv_layout read it.  Needs H*W % 3 == 1 (every layer of the
 *                     reference's ResNets: 56^2, 28^2, 14^2, 7^2, 32^2 ... 4^2), so that e % 3 = (c + pixel) % 3.
 *                     Values are the NCHW tensor's, bit for bit; only their addresses differ.
 */
#define LSQ_LAYOUT_NCHW 0
#define LSQ_LAYOUT_SPLIT3 1
/* floats per stream of a SPLIT3 row (C blocks of hp floats, hp a multiple of 32); -1 when H*W % 3 != 1 */
int64_t lsq_split3_stream_floats(int64_t C, int64_t H, int64_t W);
/* which operands of a call with geometry g may be SPLIT3: bit 0 -- the input of lsq_act_quant_layout(scheme, skip 3, no
 * forced scales, clamp > 0); bit 1 -- y of lsq_xnor_conv2d_layout with kx activation planes; bit 2 -- its residual operands;
 * bit 3 -- x of lsq_pointwise_conv_layout for a 1x1 convolution of stride g->stride_h over g's input */
int lsq_layout_support(const lsq_conv_geom* g, int scheme, int kx);

/* number of uint64 words of ONE activation plane for `g` (N * Gt * Hp * Wp) */
int64_t lsq_act_plane_words(const lsq_conv_geom* g);
/* number of uint64 words of ONE weight plane (KH*KW * Gg * O) */
int64_t lsq_weight_plane_words(const lsq_conv_geom* g);

/*
 * Activation quantization: clamp -> per-sample scale solve -> packed sign planes.
 * Replaces ActivationQuantizer*._batch_quantization / _moving_average_quantization
 * (quant/binary/activation_quantization.py:68-102) together with clamp_symmetric
 * (quantization.py:22-24), opt_v1 / compute_mask / cost_function (optimal.py:16-155).
 *
 *   x            [N][C][H][W] fp32
 *   scheme,k     LS1: k=1; LS2, LST: k=2; GF: k = number of bits (1..LSQ_MAX_PLANES)
 *   skip         sub-sampling stride of the v1 search (quantization.py:63; 3 in the reference)
 *   clamp_alpha  symmetric clamp bound, or a negative value for clamp_identity
 *   pre_scale, pre_shift  NULL, or [C] per-channel affine applied before the clamp, x' = x*s[c] + t[c]:
 *                the eval-mode BatchNorm2d that feeds the layer (quant/models/resnet.py:182,185) folded
 *                into the read, s = gamma / sqrt(var + eps), t = beta - mean * s
 *   forced       NULL, or [k][N] scales to use instead of solving (eval with moving average,
 *                activation_quantization.py:90-98)
 *   planes       out, [k] activation planes laid out as described above (halo pre-zeroed)
 *   scales       out, [k][N] fp32: v1..vk per sample (LST: row 1 repeats v1)
 *   workspace    LS2 / LST without forced scales: lsq_solver_workspace_bytes(N) bytes, 8-byte aligned (required).
 *                LS1 / GF without forced scales: NULL, or lsq_sweep_workspace_bytes(N) bytes, 8-byte aligned, one
 *                buffer per stream, ANY content (the arrival slots are tagged with a per-launch epoch and released by
 *                the last arrival, so neither zero-filling, the leftovers of an aborted launch nor the replay of a
 *                captured launch matter): with it a row may be shared by several
 *                workgroups when the batch alone would leave CUs idle -- same planes, the scale is the same
 *                fixed-order sum either way.
 */
int lsq_act_quant(const float* x, const lsq_conv_geom* g, int scheme, int k, int skip,
                  float clamp_alpha, const float* pre_scale, const float* pre_shift,
                  const float* forced, uint64_t* planes, float* scales,
                  void* workspace, size_t workspace_bytes, void* stream);

/* lsq_act_quant for an input in `x_layout` (LSQ_LAYOUT_*); SPLIT3: LS2 / LST without forced scales, skip 3, clamp_alpha > 0,
 * C a multiple of 64 per group, LSQ_E_UNSUPPORTED otherwise.  Planes and scales are those of the NCHW call, bit for bit. */
int lsq_act_quant_layout(const float* x, int x_layout, const lsq_conv_geom* g, int scheme, int k, int skip,
                         float clamp_alpha, const float* pre_scale, const float* pre_shift,
                         const float* forced, uint64_t* planes, float* scales,
                         void* workspace, size_t workspace_bytes, void* stream);

/* Bytes of scratch the LS2 / LST scale solve needs for `rows` rows (slot records handed from the
 * histogram sweep to the solve kernel).  Not needed (may be NULL / 0) for LS1, GF or forced scales. */
int64_t lsq_solver_workspace_bytes(int64_t rows);

/* Bytes of the optional row workspace of the LS1 / GF sweeps (partial sums and an arrival counter per row). */
int64_t lsq_sweep_workspace_bytes(int64_t rows);

/*
 * Stand-alone optimal-v1 solve on a dense [R][M] fp32 matrix (rows need not be activations):
 * opt_v1(matrix, ternary, skip) of quant/binary/optimal.py:121-155 followed, for the
 * non-ternary case, by v2 = mean|x - v1 sign(x)| over the full row (quantization.py:84-85).
 *   v12 [2][R] out: row 0 = v1, row 1 = v2 (ternary: v2 = v1);  status [R] out (may be NULL):
 *   number of distinct candidate values found (0 => v1 = 0, the reference's zero padding wins).
 */
int lsq_solve_rows(const float* rows, int64_t R, int64_t M, int skip, int ternary,
                   float clamp_alpha, float* v12, int32_t* status, void* workspace,
                   size_t workspace_bytes, void* stream);

/*
 * Chains of 1-bit (LS1) layers (ABI v10): lsq_xnor_conv2d with the NEXT layer's quantizer in its epilogue and / or this
 * layer's activation scale taken from what the previous layer's epilogue left -- the lsq_act_quant launch between two
 * quantized convolutions and its read of the activation tensor disappear (QuantConv2d.forward, binary_conv.py:161-173,
 * for x_quant = 'ls-1': quantizer_ls_1, quantization.py:35-56, is a sign and a mean).
 *   next        NULL, or where the epilogue leaves the next layer's input: sign(clamp(y * pre_scale[o] + pre_shift[o]))
 *               as plane words [N][O/64][Ho + 2 pad_h][Wo + 2 pad_w] (halo pre-zeroed by the caller, interior fully
 *               written) and, ADDED to sum_units[N] (int64, zeroed by the caller before the launch), the row sums of
 *               |clamp(.)| in units of 2^e, e = ceil(log2 clamp_alpha) - 31 -- the exact arithmetic of lsq_act_quant's
 *               plain sweeps, so that float(sum_units * 2^e / (O Ho Wo)) IS the scale lsq_act_quant(LS1) would return
 *   x_units     NULL (then xscales [1][N] as in lsq_xnor_conv2d), or the sum_units a previous call left for THIS layer's
 *               input; x_alpha = this layer's clamp (> 0), which fixes the unit
 * One activation plane; 3x3 kernels over 64 / 128 / 256 / 512 channels (the integer-MFMA kernel), O a multiple of 64 when
 * next is given; LSQ_E_UNSUPPORTED otherwise (the caller takes lsq_act_quant + lsq_xnor_conv2d: same bits).
 */
typedef struct {
  uint64_t* planes;
  int64_t* sum_units;
  const float* pre_scale; /* [O] or NULL: eval-mode batch norm in front of the next quantizer, folded */
  const float* pre_shift;
  float clamp_alpha;      /* the next layer's symmetric clamp, > 0 */
  int pad_h, pad_w;       /* the next layer's padding (halo of its plane) */
} lsq_next_ls1;
int lsq_xnor_conv2d_chain(const uint64_t* xplanes, const float* xscales, const int64_t* x_units, float x_alpha,
                          const uint64_t* wbits, const int32_t* wsum, int kw_planes, const float* wscales,
                          const float* bias, const lsq_conv_geom* g, int act, const float* act_slope,
                          const float* res_pre, const float* res_post, const lsq_next_ls1* next, float* y, void* stream);

/*
 * Training-side pieces of the quantizers (ABI v8).  Rows are samples (activations, M = C*H*W)
 * or output channels (weights, M = C/groups*KH*KW; clamp_alpha < 0); scales [k][rows] are the sign planes' scales
 * (LS2: v1, v2;  LST: v1, v1;  LS1: v1;  GF: v1..vk;  k = 0: full precision, the clamp alone).
 *
 * lsq_quant_values: x_q = sum_i v_i b_i with b_i = sign(clamp(x) - sum_{r<i} v_r b_r) -- the tensor the
 *   reference's quantizers return (quantization.py:56, :89-92, :112-115, :137-146), the fp32 operand of the
 *   weight-gradient convolution in training.
 * lsq_ste_backward: grad_x = d<grad_q, x_q>/dx through the straight-through estimator of every sign
 *   (quant/binary/ste.py:51-66: the gradient passes where the sign's argument lies in [-1, 1]) and the symmetric
 *   clamp (quantization.py:22-24: passes inside [-alpha, alpha]); no gradient flows through the scales (the
 *   reference computes them from detached data, quantization.py:53, :77, :109, :133).
 * rows <= 65535.
 */
int lsq_quant_values(const float* x, int64_t rows, int64_t M, int k, const float* scales, float clamp_alpha,
                     float* x_q, void* stream);
int lsq_ste_backward(const float* x, const float* grad_q, int64_t rows, int64_t M, int k, const float* scales,
                     float clamp_alpha, float* grad_x, void* stream);

/*
 * Weight sign packing with cached per-output-channel scales (eval mode):
 * replaces WeightQuantizer*.forward in eval (quant/binary/weight_quantization.py:32-34,
 * :57-58, :80-81, :106-108) -- plane q = sign(w - sum_{r<q} u_r * plane_r).
 *   w       [O][C/groups][KH][KW] fp32
 *   scales  [k][O] fp32 (the module's v1..vk buffers)
 *   wbits   out, [k] weight planes;  wsum out, [k][O][KH*KW] int32
 */
int lsq_pack_weight(const float* w, const lsq_conv_geom* g, int k, const float* scales,
                    uint64_t* wbits, int32_t* wsum, void* stream);

/*
 * Binary x binary convolution -- exact integer inner products of sign planes: 3x3 kernels over 64 / 128 / 256 / 512 channels on the
 * matrix cores (sign bits as fp4 operands of v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales: every product +-2 or 0, the fp32
 * accumulator holds the integer), every other geometry by XNOR + popcount; both give the same bits:
 *   y[n][o] = bias[o] + sum_p sum_q xs[p][n] * ws[q][o] * (plane_p (*) wplane_q)[n][o]
 * which equals F.conv2d(x_q, w_q, bias, ...) of binary_conv.py:165-173 for
 * x_q = sum_p xs_p b_p, w_q = sum_q ws_q s_q (exact integer inner products).
 *   kx, kw_planes  number of activation / weight planes
 *   act, act_slope, res_pre, res_post   optional fused block epilogue, y = act(conv + bias + res_pre) + res_post
 *                  (the non-linearity and shortcut additions of quant/models/resnet.py:95-100, :182-190);
 *                  act: LSQ_ACT_NONE / LSQ_ACT_RELU / LSQ_ACT_PRELU (one slope, act_slope[0], nn.PReLU()) /
 *                  LSQ_ACT_PRELU_CHANNEL (act_slope[o]); act_slope is a device pointer, NULL unless a PReLU;
 *                  residuals are [N][O][Ho][Wo] fp32 or NULL
 *   y              out, [N][O][Ho][Wo] fp32
 */
int lsq_xnor_conv2d(const uint64_t* xplanes, int kx, const float* xscales,
                    const uint64_t* wbits, const int32_t* wsum, int kw_planes,
                    const float* wscales, const float* bias, const lsq_conv_geom* g,
                    int act, const float* act_slope, const float* res_pre, const float* res_post,
                    float* y, void* stream);

/* lsq_xnor_conv2d with y in `y_layout` and res_pre / res_post in `res_layout` (LSQ_LAYOUT_*).  SPLIT3 operands: the
 * integer-MFMA kernel's 3x3 geometries over 64 or 128 channels with two activation planes (lsq_layout_support), else
 * LSQ_E_UNSUPPORTED.  Same values as the NCHW call. */
int lsq_xnor_conv2d_layout(const uint64_t* xplanes, int kx, const float* xscales,
                           const uint64_t* wbits, const int32_t* wsum, int kw_planes,
                           const float* wscales, const float* bias, const lsq_conv_geom* g,
                           int act, const float* act_slope, const float* res_pre, const float* res_post, int res_layout,
                           float* y, int y_layout, void* stream);

/*
 * Full-precision activation x sign-weight convolution on bf16 MFMA (x split hi+lo):
 *   y[n][o] = bias[o] + sum_q ws[q][o] * conv(clamp(x), wplane_q)[n][o]
 * Replaces F.conv2d(x, w_q, ...) of binary_conv.py:165-173 when x_quant == 'fp'.
 *   wprep   NULL, or the buffer lsq_signw_prepare_weight filled for the same wbits / geometry: 9-tap kernels
 *           (3x3) then take the fast path, whose workgroups copy ready-made bf16 +-1 fragments instead of
 *           expanding the sign bits again in every workgroup.  Results are identical with and without it.
 */
int lsq_signw_conv2d(const float* x, float clamp_alpha, const float* pre_scale, const float* pre_shift,
                     const uint64_t* wbits, const void* wprep, int kw_planes,
                     const float* wscales, const float* bias, const lsq_conv_geom* g,
                     int act, const float* act_slope, const float* res_pre, const float* res_post,
                     float* y, void* stream);

/*
 * Once per eval session, next to lsq_pack_weight (the eval branch of WeightQuantizer*.forward,
 * quant/binary/weight_quantization.py:32-34): the sign planes expanded to the bf16 operand image of
 * lsq_signw_conv2d's fast path.  lsq_signw_weight_bytes = size of that buffer in bytes, 0 when the
 * geometry has no fast path (kernels other than 9 taps, groups, channels not a multiple of 16).
 */
int64_t lsq_signw_weight_bytes(const lsq_conv_geom* g, int kw_planes);
int lsq_signw_prepare_weight(const uint64_t* wbits, int kw_planes, const lsq_conv_geom* g, void* wprep, void* stream);

/*
 * Stem tail in front of the first quantized convolution:
 *   y[n][c][ho][wo] = act( max_{kh,kw} x[n][ho*stride-pad+kh][wo*stride-pad+kw][c] + bias[c] ),  act = ReLU or identity
 * x is channels-last (NHWC, the layout MIOpen's convolution is fastest in), y is NCHW (what lsq_act_quant
 * reads).  Replaces nn.ReLU + nn.MaxPool2d of the reference's first block in eval mode
 * (quant/models/resnet.py: Sequential(conv1, bn1, relu, maxpool), forward :393-397) once the batch norm is
 * folded into the convolution: bias and ReLU commute with the max.  Padding behaves like
 * nn.MaxPool2d (-inf), 2*pad <= kernel, floor mode, no dilation.  bias may be NULL.
 */
int lsq_pool_bias_relu_nhwc(const float* x_nhwc, int N, int C, int H, int W, int kernel, int stride, int pad,
                            const float* bias, int relu, float* y_nchw, void* stream);

/*
 * ResNet stem in one kernel: y = max_pool2d(relu(conv2d(x, w, stride 2, padding 3) + bias), 3, 2, 1) for a
 * 7x7 convolution from 3 to 64 channels whose eval-mode batch norm has been folded into w / bias by the caller.
 * Replaces Sequential(conv1, bn1, relu, maxpool) of the reference's QResNet (quant/models/resnet.py: __init__
 * layer0, forward :393-397) in front of the first QuantConv2d.  x [N,3,H,W] fp32 NCHW (W even, 8-byte aligned),
 * w [64,3,7,7], bias [64], y [N,64,Hp,Wp] with Hc = (H - 1) / 2 + 1, Hp = (Hc - 1) / 2 + 1 (same for W).
 * 16-bit MFMA on split fp32 operands, fp32 accumulation.  split = 3: three bf16 terms (six passes, dropped terms
 * <= 2^-24 relative per product: fp32 rounding level, any finite operand); 2: two bf16 terms (three passes, ~2^-17
 * per product); 22: fp16 leading term + fp16 remainder scaled by 2^11 (three passes, 2^-23 per product; operands
 * must be below 65504 in magnitude -- beyond that the leading term is inf and so is the output).
 *   overflow   NULL, or an int32 on the device the caller has zeroed: split 22 sets it to 1 when an element of x or
 *              w is at or beyond 65504 (or NaN), i.e. outside that split's domain -- the caller then knows the output
 *              is not to be used and can repeat the call with split 3.
 */
int lsq_stem_conv_pool(const float* x, int N, int H, int W, const float* w, const float* bias, int split,
                       float* y, int32_t* overflow, void* stream);

/*
 * Strided 1x1 convolution: y[n][o][ho][wo] = sum_c w[o][c] * x[n][c][ho*stride][wo*stride] + bias[o], the projection
 * shortcut Sequential(Conv2d(in, out, 1, stride), BatchNorm2d(out)) of the reference's residual blocks
 * (quant/models/resnet.py: XnorBasicBlock / RegularBasicBlock, forward :180-190) with the eval-mode batch norm
 * folded into w / bias by the caller.  x [N,C,H,W], w [O,C], y [N,O,Ho,Wo] fp32 NCHW, Ho = (H-1)/stride + 1;
 * C and O multiples of 64 (LSQ_E_UNSUPPORTED otherwise); bias may be NULL.  Exact fp32 (fp32 MFMA).
 */
int lsq_pointwise_conv(const float* x, int N, int C, int H, int W, const float* w, const float* bias, int O,
                       int stride, float* y, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LSQ_HIP_H_ */
