// Strided 1x1 convolution (the projection shortcut of a down-sampling residual block) for gfx950:
//     y[n, o, ho, wo] = sum_c w[o, c] * x[n, c, ho * s, wo * s] + b[o]            (batch norm folded into w, b)
// NCHW fp32 in and out, exact fp32 arithmetic on the matrix cores (v_mfma_f32_32x32x2_f32: a k-ordered fmaf
// chain, bit for bit).
//
// Replaces Sequential(Conv2d(in, out, 1, stride), BatchNorm2d(out)) built by the reference's blocks
// (quant/models/resnet.py: XnorBasicBlock / RegularBasicBlock shortcut, forward :180-190) in eval mode.  The
// previous path was a strided gather copy into a dense buffer plus a batched GEMM (three kernels per projection
// with the bias trick, 0.32 ms per forward for the three projections); here the gather happens while staging the
// GEMM operand, so x is read once, in place.
//
// GEMM view per workgroup: 64 out-channels x 64 pixels (flat index over n, ho, wo), K = C in chunks of 64 staged
// in LDS as fp32, both operands channel-minor: W chunk [64 o][64 c], X chunk [64 pixels][64 c].  Each of the four
// waves owns one 32 x 32 tile; an MFMA takes ONE fp32 per lane per operand (A[o = lane & 31][k = lane >> 5],
// B[k = lane >> 5][pixel = lane & 31]), fetched 32 at a time with ds_read_b128.  The global loads of chunk i + 1 are
// in flight during the MFMAs of chunk i.  The work is small (10 GFLOP per forward, 63 us at the fp32 MFMA peak).

#include "lsq_common.h"

namespace lsq {
namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

struct PwArgs {
  const float* x;      // [N][C][H][W]
  const float* w;      // [O][C]
  const float* bias;   // [O] or null
  float* y;            // [N][O][Ho][Wo]
  int N, C, H, W, O, Ho, Wo, s;
  long long P;         // N * Ho * Wo
};

constexpr int kPitch = 68;   // floats per LDS row: 16-byte aligned rows, ds_read_b128 of 16 consecutive rows conflict-free

__global__ __launch_bounds__(256) void pointwise_conv_kernel(PwArgs a) {
  __shared__ __attribute__((aligned(16))) float ws[64][kPitch];   // [out-channel][channel]
  __shared__ __attribute__((aligned(16))) float xs[64][kPitch];   // [pixel][channel]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // out-channel tile = fast index: the O / 64 workgroups that read the same pixels are dispatched together, so
  // the strided x tile comes from HBM once and from L2 afterwards
  const unsigned n_ot = (unsigned)a.O >> 6;
  const long long p0 = (long long)(blockIdx.x / n_ot) * 64;
  const int o0 = (int)(blockIdx.x % n_ot) * 64;
  const int HoWo = a.Ho * a.Wo;
  const long long HW = (long long)a.H * a.W;
  // staging role: pixel column `lane`, wave w brings channels 16 w .. 16 w + 15 of the chunk (pixels past the end
  // are clamped to the last one: loaded, multiplied, never stored -- no branch around any load)
  const long long sp = min(p0 + lane, a.P - 1);
  const int sn = (int)(sp / HoWo);
  const int sr = (int)(sp - (long long)sn * HoWo);
  const int sho = sr / a.Wo, swo = sr - sho * a.Wo;
  // (addresses = workgroup-uniform 64-bit base + one 32-bit offset per lane: the entry point admits tensors below
  // 2^30 elements; 64-bit per-lane addresses for 32 loads in flight would cost 64 VGPRs)
  const unsigned xoff = (unsigned)(((long long)sn * a.C + 16 * wid) * HW + (long long)sho * a.s * a.W + swo * a.s);
  const float* __restrict__ xsrc = a.x;
  // W chunk: consecutive lanes -> consecutive channels of one out-channel row; wave w brings rows 16 w .. 16 w + 15
  const unsigned woff = (unsigned)(16 * wid * a.C + lane);
  const float* __restrict__ wsrc = a.w + (long long)o0 * a.C;
  // compute role: out-channel tile mt, pixel tile nt
  const int mt = wid >> 1, nt = wid & 1;
  const int col = lane & 31, g = lane >> 5;
  f32x16 acc = {};
  float xv[16], wv[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) xv[i] = (xsrc + (long long)i * HW)[xoff];
#pragma unroll
  for (int i = 0; i < 16; ++i) wv[i] = (wsrc + (long long)i * a.C)[woff];
  for (int c0 = 0; c0 < a.C; c0 += 64) {
    __syncthreads();                       // the fragments of the previous chunk have been read
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<float4*>(&xs[lane][16 * wid + 4 * q]) = make_float4(xv[4 * q], xv[4 * q + 1], xv[4 * q + 2], xv[4 * q + 3]);
#pragma unroll
    for (int i = 0; i < 16; ++i) ws[16 * wid + i][lane] = wv[i];
    __syncthreads();
    if (c0 + 64 < a.C) {                   // the next chunk's loads fly during this chunk's MFMAs
#pragma unroll
      for (int i = 0; i < 16; ++i) xv[i] = (xsrc + (long long)(c0 + 64 + i) * HW)[xoff];
#pragma unroll
      for (int i = 0; i < 16; ++i) wv[i] = (wsrc + (long long)i * a.C + c0 + 64)[woff];
    }
    // MFMA step s multiplies channel 32 g + s of the chunk (any assignment of channels to steps is a valid GEMM as
    // long as both operands use the same one): a lane's 32 operands per matrix are 128 contiguous bytes
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float4 af[4], bf[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        af[q] = *reinterpret_cast<const float4*>(&ws[mt * 32 + col][32 * g + 16 * h + 4 * q]);
        bf[q] = *reinterpret_cast<const float4*>(&xs[nt * 32 + col][32 * g + 16 * h + 4 * q]);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].x, bf[q].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].y, bf[q].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].z, bf[q].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].w, bf[q].w, acc, 0, 0, 0);
      }
    }
  }
  // lane: pixel p0 + nt * 32 + col, out-channels o0 + mt * 32 + (reg & 3) + 8 (reg >> 2) + 4 g
  const long long p = p0 + nt * 32 + col;
  const int ob = o0 + mt * 32 + 4 * g;
  float bv[16];
  if (a.bias) {                            // (uniform) all bias loads before the first store
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) bv[reg] = a.bias[ob + (reg & 3) + 8 * (reg >> 2)];
  } else {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) bv[reg] = 0.f;
  }
  if (p >= a.P) return;
  const int n = (int)(p / HoWo);
  const int r = (int)(p - (long long)n * HoWo);
  const unsigned yoff = (unsigned)(((long long)n * a.O + ob) * HoWo + r);
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) (a.y + (long long)((reg & 3) + 8 * (reg >> 2)) * HoWo)[yoff] = acc[reg] + bv[reg];
}

}  // namespace
}  // namespace lsq

using namespace lsq;

extern "C" int lsq_pointwise_conv(const float* x, int N, int C, int H, int W, const float* w, const float* bias, int O,
                                  int stride, float* y, void* stream) {
  if (!x || !w || !y) return LSQ_E_NULL;
  if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || O <= 0 || stride <= 0) return LSQ_E_SHAPE;
  if (C % 64 || O % 64) return LSQ_E_UNSUPPORTED;
  if ((long long)N * C * H * W >= (1ll << 30) || (long long)N * O * ((H - 1) / stride + 1) * ((W - 1) / stride + 1) >= (1ll << 30))
    return LSQ_E_UNSUPPORTED;              // 32-bit lane offsets
  PwArgs a = {};
  a.x = x; a.w = w; a.bias = bias; a.y = y;
  a.N = N; a.C = C; a.H = H; a.W = W; a.O = O; a.s = stride;
  a.Ho = (H - 1) / stride + 1;
  a.Wo = (W - 1) / stride + 1;
  a.P = (long long)N * a.Ho * a.Wo;
  const long long blocks = (a.P + 63) / 64;
  if (blocks * (O / 64) > 0x7FFFFFFF) return LSQ_E_SHAPE;
  hipLaunchKernelGGL(pointwise_conv_kernel, dim3((unsigned)(blocks * (O / 64))), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
