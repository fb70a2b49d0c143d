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

// (the waves of a workgroup exchange the operand tiles through LDS only: __syncthreads() would also wait for the NEXT tile's
//  global loads, which are issued precisely so that they are in flight during the MFMAs)
#ifdef LSQ_PW_SYNCTHREADS
#define PW_BARRIER() __syncthreads()
#else
#define PW_BARRIER() lds_barrier()
#endif
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

// ---------------------------------------------------------------------------------------------------------------
// All out-channel tiles in one workgroup (round 3).  The kernel above gives every (64 pixels, 64 out-channels) tile
// its own workgroup, so the strided gather of x -- 4-byte loads, each a full 16-cycle address pass of the CU's
// vector-memory path -- is repeated O / 64 times (PMC: 4-8x the input bytes), and together with the 4-byte weight
// loads it takes as long as the MFMAs.  Here a workgroup owns 64 pixels and ALL out-channels (NOT tiles of 64, at most
// 8: 128 accumulator registers per lane): per 64-channel chunk the x tile is gathered ONCE -- as 16-byte loads that
// cover the pixel pair (wo, wo + 1) of a stride-2 row when Wo is even -- and the weight tiles stream through a
// double-buffered LDS tile as 16-byte loads.  Same k order as above: same fmaf chain, same bits.
typedef float f32x4 __attribute__((ext_vector_type(4)));

// VEC4: Ho * Wo is a multiple of 4 (an aligned group of four pixels never straddles two images): 16-byte stores.
template <int NOT, bool PAIRS, bool VEC4>
__global__ __launch_bounds__(256) void pointwise_all_kernel(PwArgs a) {
  __shared__ __attribute__((aligned(16))) float ws[2][64][kPitch];   // [buffer][out-channel][channel]
  __shared__ __attribute__((aligned(16))) float xs[64][kPitch];      // [pixel][channel]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const long long p0 = (long long)blockIdx.x * 64;
  const int ob = (int)blockIdx.y * (NOT * 64);           // (few pixel tiles: the out-channels are spread over blockIdx.y)
  const int HoWo = a.Ho * a.Wo;
  const long long HW = (long long)a.H * a.W;
  const float* __restrict__ xsrc = a.x;
  // ---- x staging roles
  //   PAIRS (stride 2, Wo even): thread = (pixel pair pp = tid & 31, channel octet tid >> 5): one 16-byte load per channel
  //   brings x[2 wo .. 2 wo + 3] of the row, i.e. pixels wo and wo + 1 (elements 0 and 2);
  //   else: thread = (pixel = lane, 16 channels 16 wid ..), 4-byte loads as in the kernel above.
  unsigned xoff;
  if constexpr (PAIRS) {
    const long long sp = min(p0 + 2 * (tid & 31), a.P - 2);          // (P is even here: Wo even)
    const int sn = (int)(sp / HoWo);
    const int sr = (int)(sp - (long long)sn * HoWo);
    const int sho = sr / a.Wo, swo = sr - sho * a.Wo;
    xoff = (unsigned)(((long long)sn * a.C + 8 * (tid >> 5)) * HW + (long long)sho * a.s * a.W + swo * a.s);
  } else {
    const long long sp = min(p0 + lane, a.P - 1);
    const int sn = (int)(sp / HoWo);
    const int sr = (int)(sp - (long long)sn * HoWo);
    const int sho = sr / a.Wo, swo = sr - sho * a.Wo;
    xoff = (unsigned)(((long long)sn * a.C + 16 * wid) * HW + (long long)sho * a.s * a.W + swo * a.s);
  }
  // ---- weight staging role: thread = (row tid >> 4 (+16 k), 4 channels 4 (tid & 15) ..): four 16-byte loads per tile
  const unsigned woff = (unsigned)((tid >> 4) * a.C + 4 * (tid & 15));
  const int mt = wid >> 1, nt = wid & 1;
  const int col = lane & 31, g = lane >> 5;
  f32x16 acc[NOT];
#pragma unroll
  for (int t = 0; t < NOT; ++t)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;

  f32x4 xq[8];
  float xv[16];
  auto load_x = [&](int c0) {
    if constexpr (PAIRS) {
#pragma unroll
      for (int i = 0; i < 8; ++i) xq[i] = *reinterpret_cast<const f32x4*>(xsrc + (long long)(c0 + i) * HW + xoff);   // 8-byte aligned at least
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) xv[i] = (xsrc + (long long)(c0 + i) * HW)[xoff];
    }
  };
  auto store_x = [&]() {
    if constexpr (PAIRS) {
      const int px = 2 * (tid & 31), cb = 8 * (tid >> 5);
      *reinterpret_cast<float4*>(&xs[px][cb]) = make_float4(xq[0][0], xq[1][0], xq[2][0], xq[3][0]);
      *reinterpret_cast<float4*>(&xs[px][cb + 4]) = make_float4(xq[4][0], xq[5][0], xq[6][0], xq[7][0]);
      *reinterpret_cast<float4*>(&xs[px + 1][cb]) = make_float4(xq[0][2], xq[1][2], xq[2][2], xq[3][2]);
      *reinterpret_cast<float4*>(&xs[px + 1][cb + 4]) = make_float4(xq[4][2], xq[5][2], xq[6][2], xq[7][2]);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(&xs[lane][16 * wid + 4 * q]) = make_float4(xv[4 * q], xv[4 * q + 1], xv[4 * q + 2], xv[4 * q + 3]);
    }
  };
  f32x4 wq[4];
  auto load_w = [&](int t, int c0) {
    const float* wsrc = a.w + (long long)(ob + t * 64) * a.C + c0;
#pragma unroll
    for (int k = 0; k < 4; ++k) wq[k] = *reinterpret_cast<const f32x4*>(wsrc + (long long)(16 * k) * a.C + woff);
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      *reinterpret_cast<float4*>(&ws[buf][(tid >> 4) + 16 * k][4 * (tid & 15)]) = make_float4(wq[k][0], wq[k][1], wq[k][2], wq[k][3]);
  };

  const int nchunks = a.C >> 6;
  load_x(0);
  load_w(0, 0);
  int buf = 0;
  for (int cc = 0; cc < nchunks; ++cc) {
    PW_BARRIER();                          // the x fragments of the previous chunk have been read
    store_x();
    if (cc + 1 < nchunks) load_x(64 * (cc + 1));
#pragma unroll
    for (int t = 0; t < NOT; ++t) {
      store_w(buf);                        // (the tile two steps back was read before the barrier of the step in between)
      PW_BARRIER();
      if (t + 1 < NOT) load_w(t + 1, 64 * cc);
      else if (cc + 1 < nchunks) load_w(0, 64 * (cc + 1));
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float4 af[4], bf[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          af[q] = *reinterpret_cast<const float4*>(&ws[buf][mt * 32 + col][32 * g + 16 * h + 4 * q]);
          bf[q] = *reinterpret_cast<const float4*>(&xs[nt * 32 + col][32 * g + 16 * h + 4 * q]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          // (A = pixels, B = weights: the accumulator is D[pixel][out-channel] -- a lane owns one out-channel and
          //  groups of four consecutive pixels, 16 contiguous bytes of the NCHW output)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[q].x, af[q].x, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[q].y, af[q].y, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[q].z, af[q].z, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[q].w, af[q].w, acc[t], 0, 0, 0);
        }
      }
      buf ^= 1;
    }
  }
  // lane: out-channel 64 t + mt * 32 + col, pixels p0 + nt * 32 + (reg & 3) + 8 (reg >> 2) + 4 g
#pragma unroll
  for (int t = 0; t < NOT; ++t) {
    const int o = ob + 64 * t + mt * 32 + col;
    const float b = a.bias ? a.bias[o] : 0.f;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const long long p = p0 + nt * 32 + 8 * rg + 4 * g;
      if constexpr (VEC4) {
        if (p < a.P) {                       // (P is a multiple of 4: the whole group is inside)
          const int n = (int)(p / HoWo);
          const int r = (int)(p - (long long)n * HoWo);
          const f32x4 v = {acc[t][4 * rg] + b, acc[t][4 * rg + 1] + b, acc[t][4 * rg + 2] + b, acc[t][4 * rg + 3] + b};
          *reinterpret_cast<f32x4*>(a.y + ((long long)n * a.O + o) * HoWo + r) = v;
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const long long pk = p + k;
          if (pk < a.P) {
            const int n = (int)(pk / HoWo);
            const int r = (int)(pk - (long long)n * HoWo);
            a.y[((long long)n * a.O + o) * HoWo + r] = acc[t][4 * rg + k] + b;
          }
        }
      }
    }
  }
}

template <int NOT>
void launch_all(const PwArgs& a, dim3 grid, hipStream_t st) {
  // pixel pairs: stride 2 with an even output width (a pair never straddles two rows), the second pixel inside the row
  const bool vec4 = ((a.Ho * a.Wo) & 3) == 0;
  if (a.s == 2 && (a.Wo & 1) == 0 && a.P >= 2) {
    if (vec4) hipLaunchKernelGGL((pointwise_all_kernel<NOT, true, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((pointwise_all_kernel<NOT, true, false>), grid, dim3(256), 0, st, a);
  } else {
    if (vec4) hipLaunchKernelGGL((pointwise_all_kernel<NOT, false, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((pointwise_all_kernel<NOT, false, false>), grid, dim3(256), 0, st, a);
  }
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
  // all out-channel tiles in one workgroup when they fit the accumulators (O <= 512) and the last load of a pixel pair
  // stays inside the tensor (W even: the pair's 16 bytes end at the row's end at the latest)
  const int n_ot = O / 64;
  const bool pair_ok = stride != 2 || (a.Wo & 1) || (W & 1) == 0;
  if (n_ot <= 8 && (n_ot & (n_ot - 1)) == 0 && pair_ok) {
    // out-channel tiles per workgroup: all of them (x gathered once) unless that leaves CUs without a workgroup
    int per_wg = n_ot;
    long long min_wgs = 256;
#ifdef LSQ_TUNE
    if (const char* e = getenv("LSQ_PW_MINWG")) min_wgs = atoll(e);
#endif
    while (per_wg > 1 && blocks * (n_ot / per_wg) < min_wgs) per_wg >>= 1;
    const dim3 grid((unsigned)blocks, (unsigned)(n_ot / per_wg));
    switch (per_wg) {
      case 1: launch_all<1>(a, grid, (hipStream_t)stream); break;
      case 2: launch_all<2>(a, grid, (hipStream_t)stream); break;
      case 4: launch_all<4>(a, grid, (hipStream_t)stream); break;
      default: launch_all<8>(a, grid, (hipStream_t)stream); break;
    }
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL(pointwise_conv_kernel, dim3((unsigned)(blocks * (O / 64))), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
