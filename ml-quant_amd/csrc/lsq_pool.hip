// Stem tail for gfx950: max-pool -> (+ per-channel bias) -> ReLU, channels-last in, NCHW out.
//
// Replaces, in eval mode, nn.ReLU + nn.MaxPool2d at the end of the reference's first block
// (quant/models/resnet.py: QResNet.__init__ builds Sequential(conv1, bn1, relu, maxpool); forward :393-397)
// for the tensor that feeds the first quantized convolution.  With the batch norm folded into the
// convolution, relu(pool(conv + b)) = relu(pool(conv) + b): the bias and the ReLU commute with the max, so
// they run on the pooled (stride^2 times smaller) tensor.  The convolution itself stays on MIOpen, which
// is fastest in channels-last; this kernel reads that NHWC tensor once (HBM-bound) and writes the NCHW
// tensor the quantizer kernels read -- three PyTorch passes (pool, bias add with layout change, ReLU)
// collapse into one.
//
// One workgroup = one output row (n, ho) x 64 channels.  Loads: lane = channel, so the 64 lanes of a wave
// read 256 contiguous bytes of one input pixel.  The results go through an LDS tile [64 channels][64 wo]
// (row pitch 65 floats: conflict-free both ways) and leave with lane = wo: 256-byte stores per channel row.

#include "lsq_common.h"

namespace lsq {
namespace {

struct PoolArgs {
  const float* x;      // [N][H][W][C]
  const float* bias;   // [C] or null
  float* y;            // [N][C][Ho][Wo]
  int N, C, H, W, Ho, Wo, k, stride, pad, relu;
};

// K, S > 0: kernel size and stride known at compile time -- each wave takes 16 adjacent output columns,
// issues the (15*S + K) * K input loads of its lane's channel unconditionally (out-of-image taps read a
// clamped address and are replaced by -inf) and reduces columns first, then windows.  K = S = 0: any
// geometry, one output at a time.
template <int K, int S>
__global__ __launch_bounds__(256) void pool_bias_relu_kernel(PoolArgs a) {
  __shared__ float tile[64][65];
  const int lane = threadIdx.x & 63, grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // workgroups are dealt round-robin to the 8 XCDs (each with its own L2): give every XCD a contiguous
  // range of output rows so that the input rows shared by vertically adjacent outputs hit in its L2
  const int per_xcd = gridDim.x >> 3;
  const int row = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (row >= a.N * a.Ho) return;
  const int n = row / a.Ho, ho = row - n * a.Ho;
  const int c0 = blockIdx.y * 64;
  const int c = c0 + lane;
  const bool c_ok = c < a.C;
  const float b = (a.bias && c_ok) ? a.bias[c] : 0.f;
  const int hi0 = ho * a.stride - a.pad;
  const float* xn = a.x + (long long)n * a.H * a.W * a.C + (c_ok ? c : 0);
  const float ninf = -__builtin_inff();                  // max_pool2d pads with -inf
  for (int w0 = 0; w0 < a.Wo; w0 += 64) {
    if constexpr (K > 0) {
      constexpr int NIN = 15 * S + K;
      const int wbase = w0 + grp * 16;
      if (wbase < a.Wo) {
        const int wi_start = wbase * S - a.pad;
        float cm[NIN];
        constexpr int CH = 11;                           // columns per batch: CH * K loads in flight per lane
#pragma unroll
        for (int i0 = 0; i0 < NIN; i0 += CH) {
          float t[CH][K];
#pragma unroll
          for (int ii = 0; ii < CH; ++ii) {
            const int wic = min(max(wi_start + i0 + ii, 0), a.W - 1);
#pragma unroll
            for (int kh = 0; kh < K; ++kh)
              t[ii][kh] = xn[((long long)min(max(hi0 + kh, 0), a.H - 1) * a.W + wic) * a.C];
          }
#pragma unroll
          for (int ii = 0; ii < CH; ++ii) {
            if (i0 + ii < NIN) {
              const int wi = wi_start + i0 + ii;
              const bool w_ok = wi >= 0 && wi < a.W;
              float v = ninf;
#pragma unroll
              for (int kh = 0; kh < K; ++kh) {
                const int hi = hi0 + kh;
                v = fmaxf(v, (w_ok && hi >= 0 && hi < a.H) ? t[ii][kh] : ninf);
              }
              cm[i0 + ii] = v;
            }
          }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          float m = cm[j * S];
#pragma unroll
          for (int kw = 1; kw < K; ++kw) m = fmaxf(m, cm[j * S + kw]);
          m += b;
          tile[lane][grp * 16 + j] = a.relu ? fmaxf(m, 0.f) : m;
        }
      }
    } else {
      // lane = channel, the four waves take every fourth output column
      for (int wl = grp; wl < 64 && w0 + wl < a.Wo; wl += 4) {
        const int wi0 = (w0 + wl) * a.stride - a.pad;
        float m = ninf;
        for (int kh = 0; kh < a.k; ++kh) {
          const int hi = hi0 + kh;
          if (hi < 0 || hi >= a.H) continue;
          for (int kw = 0; kw < a.k; ++kw) {
            const int wi = wi0 + kw;
            if (wi < 0 || wi >= a.W) continue;
            m = fmaxf(m, xn[((long long)hi * a.W + wi) * a.C]);
          }
        }
        m += b;
        tile[lane][wl] = a.relu ? fmaxf(m, 0.f) : m;
      }
    }
    __syncthreads();
    // lane = output column, wave w stores channel rows w, w + 4, ...
    const int wo = w0 + lane;
    if (wo < a.Wo) {
      for (int r = grp; r < 64 && c0 + r < a.C; r += 4)
        a.y[(((long long)n * a.C + c0 + r) * a.Ho + ho) * a.Wo + wo] = tile[r][lane];
    }
    __syncthreads();
  }
}

}  // namespace
}  // namespace lsq

using namespace lsq;

extern "C" int lsq_pool_bias_relu_nhwc(const float* x_nhwc, int N, int C, int H, int W, int kernel, int stride,
                                       int pad, const float* bias, int relu, float* y_nchw, void* stream) {
  if (!x_nhwc || !y_nchw) return LSQ_E_NULL;
  if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || kernel <= 0 || stride <= 0 || pad < 0) return LSQ_E_SHAPE;
  if (2 * pad > kernel) return LSQ_E_SHAPE;              // (torch's own constraint: pad <= kernel / 2)
  const int Ho = (H + 2 * pad - kernel) / stride + 1, Wo = (W + 2 * pad - kernel) / stride + 1;
  if (H + 2 * pad < kernel || W + 2 * pad < kernel) return LSQ_E_SHAPE;
  if ((long long)N * Ho > 0x7FFFFFF0ll) return LSQ_E_TOO_LONG;
  PoolArgs a = {x_nhwc, bias, y_nchw, N, C, H, W, Ho, Wo, kernel, stride, pad, relu};
  const dim3 grid((unsigned)((N * Ho + 7) / 8 * 8), (unsigned)((C + 63) / 64));
  if (kernel == 3 && stride == 2) hipLaunchKernelGGL((pool_bias_relu_kernel<3, 2>), grid, dim3(256), 0, (hipStream_t)stream, a);
  else if (kernel == 2 && stride == 2) hipLaunchKernelGGL((pool_bias_relu_kernel<2, 2>), grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((pool_bias_relu_kernel<0, 0>), grid, dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
