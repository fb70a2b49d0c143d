// ResNet stem for gfx950 in ONE kernel: 7x7 stride-2 convolution (3 -> 64 channels, batch norm folded into the
// weights) + bias + ReLU + 3x3 stride-2 max-pool, NCHW fp32 in, NCHW fp32 out.
//
// Replaces, in eval mode, Sequential(conv1, bn1, relu, maxpool) in front of the first quantized convolution
// (quant/models/resnet.py: QResNet.__init__ / forward :393-397).  The previous path was three kernels -- a layout
// change of the input to channels-last, MIOpen's fp32 implicit GEMM, and the pool / bias / ReLU tail -- with the
// 112x112x64 convolution output (822 MB at batch 256) written to HBM and read back in between.  Here the
// convolution output never leaves the chip.
//
// Arithmetic: 16-bit MFMA on fp32 operands split into 16-bit terms, fp32 accumulation.
//   fp16 terms (split code 22, the host default): x = h + l 2^-11 with h = fp16(x), l = fp16((x - h) 2^11): 22 bits of
//     x in two terms, three passes hh + (hl + lh) 2^-11, 2^-23 per product -- as accurate as three bf16 terms at half
//     the MFMA work (measured 2.0e-7 of max|y| against fp64, torch's own fp32 convolution 5.4e-7).  Operands must
//     be below 65504 in magnitude (normalised images, folded weights): beyond that h is inf and so is the output.
//   SPLIT = 3 (any finite operand): x = h + m + l (each the bf16 rounding of what is left; the remainders are exact in fp32),
//     six passes hh + (hm + mh) + (hl + lh + mm); the dropped terms are <= 2^-24 relative per product, i.e. fp32
//     rounding level -- a binarized network amplifies any perturbation in front of its first quantizer (one
//     flipped sign is worth 3 % of a block's output), so the stem keeps fp32-class accuracy.
//   SPLIT = 2: x = h + l, three passes hh + hl + lh, ~2^-17 relative per product, half the MFMA work.
// Exact fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at the VALU rate: 2.5x the cycles of the six bf16 passes.
//
// GEMM view: M = 64 out-channels (two 32-row tiles), N = output pixels (32-column tiles of one conv row),
// K = (kh, c, kw) with kw padded 7 -> 8 so that the 8 k-values of one MFMA operand register group are EIGHT
// CONSECUTIVE INPUT COLUMNS of one (kh, c) input row: K = 7 * 3 * 8 = 168 -> 11 steps of 16 (two input rows per
// step; the 22nd row has zero weights).  The pad slot sits in FRONT (k = 0 is column 2x - 4, a zero weight), which
// makes the operand start at an even 16-bit index: two ds_read2_b32 per operand, conflict-free across the
// 32 lanes (consecutive pixels = consecutive dwords).
//
// One 256-thread workgroup = one image x four pooled rows (nine conv rows: one halo row recomputed), walked in
// chunks of 32 conv columns.  Per chunk: stage the 23 x 72 input patch of the three channels as 16-bit term
// planes in LDS; wave w owns out-channel tile w / 2 and conv rows w % 2, w % 2 + 2, ...: 66 (33) MFMAs per
// row on two accumulators (leading products, cross terms) with the operands in the order (pixels, weights), i.e.
// D[pixel][out-channel]: a lane holds ONE out-channel and four groups of four consecutive columns, so the horizontal
// 3-max (stride 2) is lane-local except for the column in front of each group, which the other lane half holds (one
// v_permlane32_swap per group; the column left of the chunk comes from an LDS carry the same wave wrote for the previous
// chunk) -- round 5: a third of the VALU instructions of the lane-per-pixel layout of rounds 2-4, in a kernel that was
// VALU-bound (6.8 VALU instructions per MFMA): 0.40 -> 0.37 ms in the network.  The 16 x 9 x 64 partial maxima go to
// LDS; then all lanes take the vertical 3-max, add the bias, apply the ReLU (both commute with the max) and store
// 64-byte segments of the NCHW output.  Every barrier is LDS-only (lds_barrier): the output stores and the prefetched
// patch of the next chunk stay in flight across them.

#include <type_traits>

#include "lsq_common.h"

// The waves of a workgroup talk to each other through LDS only (patch, horizontal maxima, carry); __syncthreads() would also
// drain the vector-memory counter -- the output stores of the chunk before and the prefetched patch of the next one -- at
// every one of the three barriers per chunk.
#ifndef LSQ_STEM_SYNCTHREADS
#define STEM_BARRIER() lds_barrier()
#else
#define STEM_BARRIER() __syncthreads()
#endif
namespace lsq {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

union Frag {
  unsigned u[4];
  bf16x8 v;
  __attribute__((ext_vector_type(8))) _Float16 h;
};

// two fp32 values -> SPLIT packed 16-bit pairs t[0] (leading term) .. t[SPLIT - 1].
// bf16 terms (HALF = false): each the bf16 rounding of what is left; x - t[0] - ... is exact in fp32.
// fp16 terms (HALF = true, SPLIT = 2): h = fp16(x), l = fp16((x - h) * 2^11) -- the remainder is at most 2^-12 |x|, and
// scaled up it is an ordinary fp16 number with 11 significant bits instead of a subnormal; the cross-term accumulator
// is scaled back by 2^-11 (exact).  h + l 2^-11 carries 22 bits of x: products are good to 2^-23 with THREE MFMA
// passes where bf16 needs six.  Domain: |x| below the largest fp16 (65504).
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
constexpr float kLoScale = 2048.f;
constexpr float kHalfMax = 65504.f;        // largest fp16: operands at or beyond it (or NaN) are outside the fp16 split's domain

template <int SPLIT, bool HALF>
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned (&t)[SPLIT]) {
  f32x2 r = {x0, x1};
  if constexpr (HALF) {
    static_assert(SPLIT == 2, "fp16 terms: two");
    // operands outside the fp16 domain are REPORTED (the overflow flag: the caller switches to the bf16 split); they are
    // also saturated here, so that a batch that slips through before the report is read yields finite numbers, not inf / nan
    r[0] = __builtin_amdgcn_fmed3f(r[0], -kHalfMax, kHalfMax);
    r[1] = __builtin_amdgcn_fmed3f(r[1], -kHalfMax, kHalfMax);
    const f16x2 h = __builtin_convertvector(r, f16x2);
    t[0] = __builtin_bit_cast(unsigned, h);
    r = (r - __builtin_convertvector(h, f32x2)) * kLoScale;
    t[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
  } else {
#pragma unroll
    for (int i = 0; i < SPLIT; ++i) {
      const bf16x2 b = __builtin_convertvector(r, bf16x2);
      t[i] = __builtin_bit_cast(unsigned, b);
      r = r - __builtin_convertvector(b, f32x2);
    }
  }
}

constexpr int kPH = 4;                    // pooled rows per workgroup
constexpr int kCR = 2 * kPH + 1;          // conv rows per workgroup (one halo row)
constexpr int kIR = 2 * kCR + 5;          // input rows per workgroup (23)
constexpr int kIC = 72;                   // staged input columns per chunk: 2 * 32 + 5, padded to a multiple of 8
constexpr int kSteps = 11;                // K = 176 = 11 x 16
constexpr int kO = 64;
constexpr int kHS = 20;                   // floats per (row, channel) of the horizontal maxima: 16 + 4 (16-byte aligned rows, 2-way banks)

struct StemArgs {
  const float* x;      // [N][3][H][W]
  const float* w;      // [64][3][7][7]  (batch norm folded)
  const float* bias;   // [64]
  float* y;            // [N][64][Hp][Wp]
  int N, H, W, Hc, Wc, Hp, Wp;
  int* overflow;       // fp16 split only: set to 1 when an operand is outside the split's domain (may be null)
};

template <int SPLIT>
struct StemLds {
  unsigned xs[SPLIT][3 * kIR * kIC / 2];  // bf16 pairs of each split term, [c][row][col]
  float hbuf[kCR][kO][kHS];               // horizontal 3-max (stride 2) of every conv row of the chunk (rows padded: bank spread)
  float carry[kCR][2][32];                 // last conv column of the previous chunk: [row][out-channel tile][32 out-channels]
  float bias[kO];
};

template <int SPLIT, bool HALF>
__global__ __launch_bounds__(256, 2) void stem_conv_pool_kernel(StemArgs a) {   // two workgroups per CU: <= 256 registers (VGPR + AGPR)
  __shared__ StemLds<SPLIT> lds;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int strips = (a.Hp + kPH - 1) / kPH;
  const int n = blockIdx.x / strips;
  const int pr0 = (blockIdx.x - n * strips) * kPH;
  const int cr0 = 2 * pr0 - 1;            // first conv row of the strip (may be -1)
  const int ir0 = 2 * cr0 - 3;            // first input row
  const int mt = wid >> 1;                // out-channel tile
  const int q0 = wid & 1;                 // first conv row of this wave
  const int xl_ = lane & 31, g = lane >> 5;
  const float ninf = -__builtin_inff();
  bool bad = false;                       // fp16 split: an operand at or beyond 65504 (or NaN) was seen

  // ---- A fragments (weights) of this wave's 32 out-channels, all 11 k-steps, hi and lo
  Frag af[SPLIT][kSteps];
  {
    const int o = mt * 32 + xl_;
#pragma unroll
    for (int s = 0; s < kSteps; ++s) {
      const int rr2 = 2 * s + g;          // (kh, c) row of K: kh = rr2 / 3, c = rr2 % 3; row 21 is padding
      const int kh = rr2 / 3, c = rr2 - kh * 3;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (rr2 < 21 && j > 0) ? a.w[((o * 3 + c) * 7 + kh) * 7 + (j - 1)] : 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned t[SPLIT];
        if constexpr (HALF) bad = bad || !(fmaxf(fabsf(v[2 * j]), fabsf(v[2 * j + 1])) < kHalfMax);
        split_pair<SPLIT, HALF>(v[2 * j], v[2 * j + 1], t);
#pragma unroll
        for (int i = 0; i < SPLIT; ++i) af[i][s].u[j] = t[i];
      }
    }
  }
  for (int i = tid; i < kCR * kO; i += 256) (&lds.carry[0][0][0])[i] = ninf;
  if (tid < kO) lds.bias[tid] = a.bias[tid];

  const float* __restrict__ xn = a.x + (long long)n * 3 * a.H * a.W;
  const int chunks = (a.Wc + 31) / 32;
  // ---- staging of the input patch: pairs of columns, fp32 -> 16-bit terms.  Pair e = tid + 256 it of the [3][23][36]
  // patch: 256 = 7 * 36 + 4, so (c, row, pair) advance without divisions; W is even and the patch starts at an even
  // column, so a pair is inside the image or outside it as a whole.
  constexpr int kPairs = 3 * kIR * (kIC / 2), kIt = (kPairs + 255) / 256;
  // PREFETCH (the two-term variants have the registers): all ten loads of a lane's share of the NEXT chunk are issued
  // before this chunk's MFMAs and converted after them -- the patch's memory latency, paid three times per chunk
  // with four loads in flight, was 118 of the kernel's 729 us.
  constexpr bool kPrefetch = SPLIT == 2;
  constexpr int kHalf = kPrefetch ? kIt : 4;                                      // loads in flight per lane
  float2 pre[kPrefetch ? kIt : 1];
  auto request = [&](int ck, int h0, float2* t) {                                 // iterations h0 .. h0 + kHalf - 1
    const int ic0 = 64 * ck - 4;                                                  // input column of staged column 0
    int pp = tid % (kIC / 2), rr = tid / (kIC / 2), c = 0;
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
      if (it >= h0 && it < h0 + kHalf) {
        const int ir = ir0 + rr, ic = ic0 + 2 * pp;
        const bool in = tid + 256 * it < kPairs && ir >= 0 && ir < a.H && ic >= 0 && ic < a.W;
        t[it - h0] = make_float2(0.f, 0.f);
        if (in) t[it - h0] = *reinterpret_cast<const float2*>(xn + ((long long)c * a.H + ir) * a.W + ic);
      }
      pp += 4;
      rr += 7;
      if (pp >= kIC / 2) {
        pp -= kIC / 2;
        rr += 1;
      }
      if (rr >= kIR) {
        rr -= kIR;
        c += 1;
      }
    }
  };
  auto convert = [&](int h0, const float2* t) {
#pragma unroll
    for (int it = h0; it < h0 + kHalf; ++it) {
      if (it < kIt && tid + 256 * it < kPairs) {
        unsigned sp[SPLIT];
        if constexpr (HALF) bad = bad || !(fmaxf(fabsf(t[it - h0].x), fabsf(t[it - h0].y)) < kHalfMax);
        split_pair<SPLIT, HALF>(t[it - h0].x, t[it - h0].y, sp);
#pragma unroll
        for (int i = 0; i < SPLIT; ++i) lds.xs[i][tid + 256 * it] = sp[i];
      }
    }
  };
  if constexpr (kPrefetch) request(0, 0, pre);
#ifdef LSQ_STEM_CLOCKS
  long long clk[24];
  int nclk = 0;
#define SCLK() do { __builtin_amdgcn_sched_barrier(0); if (nclk < 24) clk[nclk++] = (long long)__builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SCLK() do {} while (0)
#endif
  SCLK();
  for (int ck = 0; ck < chunks; ++ck) {
    STEM_BARRIER();                       // the previous chunk's patch and hbuf are done with
    SCLK();
    if constexpr (kPrefetch) {
      convert(0, pre);
    } else {
#pragma unroll
      for (int h0 = 0; h0 < kIt; h0 += kHalf) {
        float2 t[kHalf];
        request(ck, h0, t);
        convert(h0, t);
      }
    }
    STEM_BARRIER();
    SCLK();
    if constexpr (kPrefetch) {
      if (ck + 1 < chunks) request(ck + 1, 0, pre);
    }

    // ---- this wave's conv rows of the chunk.  PAIR (the last chunk of a row when at most 16 of its 32 columns exist: 112 = 3.5
    // x 32): one tile holds TWO of the wave's rows x 16 columns instead of one row x 32 -- pixels 0..15 row qA, 16..31 row qB --
    // so the half-empty chunk costs half the MFMAs, operand reads and maxima (round 5: an eighth of the kernel's row work).
    auto rows = [&](int qA, int qB, auto pair_tag) {
      constexpr bool PAIR = decltype(pair_tag)::value;
      const bool okA = cr0 + qA >= 0 && cr0 + qA < a.Hc;
      const bool okB = PAIR && qB < kCR && cr0 + qB >= 0 && cr0 + qB < a.Hc;
      if (!okA && !okB) return;                                // (wave-uniform) padding rows of the pool
      f32x16 acc0 = {}, acc1 = {};        // leading products; all cross terms (<= 2^-8 of them: summed apart)
      // pixel fragments: the reads of step s + 1 are issued before the MFMAs of step s (two register sets); the
      // scheduling barriers keep it at two -- left alone, the scheduler hoists several steps' reads and spills
      constexpr int kDepth = 2;           // register sets of pixel fragments (three and four measured the same: round 5)
      Frag bf[kDepth][SPLIT];
      const int ql = PAIR ? (xl_ < 16 ? qA : min(qB, kCR - 1)) : qA;       // the lane's conv row and column inside the chunk
      const int xc = PAIR ? (xl_ & 15) : xl_;
      auto load_b = [&](int s, int which) {
        const int rr2 = min(2 * s + g, 20);
        const int kh = rr2 / 3, c = rr2 - kh * 3;
        const int off = ((c * kIR + 2 * ql + kh) * kIC) / 2 + xc;   // dword index: 2 * x bf16 = x dwords
#pragma unroll
        for (int i = 0; i < SPLIT; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) bf[which][i].u[j] = lds.xs[i][off + j];
      };
#pragma unroll
      for (int s = 0; s < kDepth - 1; ++s) load_b(s, s % kDepth);
#pragma unroll
      for (int s = 0; s < kSteps; ++s) {
        __builtin_amdgcn_sched_barrier(0);
        if (s + kDepth - 1 < kSteps) load_b(s + kDepth - 1, (s + kDepth - 1) % kDepth);
        __builtin_amdgcn_sched_barrier(0);
        const Frag* b = bf[s % kDepth];
        // (the leading product BETWEEN the two cross terms: two MFMAs on the same accumulator back to back wait for each other)
        if constexpr (HALF) {
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[0].h, af[1][s].h, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[0].h, af[0][s].h, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[1].h, af[0][s].h, acc1, 0, 0, 0);
        } else {
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[0].v, af[1][s].v, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[0].v, af[0][s].v, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[1].v, af[0][s].v, acc1, 0, 0, 0);
          if constexpr (SPLIT == 3) {
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[2].v, af[0][s].v, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[0].v, af[2][s].v, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[1].v, af[1][s].v, acc1, 0, 0, 0);
          }
        }
      }
      // The MFMA operands are (pixels, weights): D[pixel][out-channel], so a lane holds ONE out-channel (mt * 32 + lane % 32)
      // and sixteen pixels x = (reg & 3) + 8 (reg >> 2) + 4 g of the tile -- groups of four consecutive columns.  The
      // horizontal 3-max with stride 2 is then mostly lane-local: of a group [4k, 4k + 3] the centre 4k + 2 needs the
      // group's own three last columns, the centre 4k the column in front of the group, which the OTHER lane half holds
      // (groups alternate between the halves): one v_permlane32_swap per group.  44 VALU instructions per row and tile
      // where the lane-per-pixel layout (two one-lane shifts and two maxima per accumulator register, half of the lanes
      // idle at the store) took 130 -- the stem was VALU-bound (6.8 VALU instructions per MFMA, round 4's counters).
      const bool full = 32 * ck + 32 <= a.Wc;                  // (uniform) no column of the chunk is past the row's end
      float v[16];
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) v[reg] = HALF ? fmaf(acc1[reg], 1.f / kLoScale, acc0[reg]) : acc1[reg] + acc0[reg];
      if (!full) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int x = (reg & 3) + 8 * (reg >> 2) + 4 * g;    // pixel of the tile; PAIR: column x & 15 of row qA (x < 16) / qB
          v[reg] = 32 * ck + (PAIR ? (x & 15) : x) < a.Wc ? v[reg] : ninf;
        }
      }
      float prevA = 0.f, prevB = 0.f;                          // column 31 of the previous chunk (32 floats per row and tile)
      if (g == 0) {
        prevA = lds.carry[qA][mt][xl_];
        if (PAIR) prevB = lds.carry[min(qB, kCR - 1)][mt][xl_];
      }
      float lo_from_hi[4], hi_from_lo[4];
#pragma unroll
      for (int G = 0; G < 4; ++G) {
        const unsigned t = __float_as_uint(v[4 * G + 3]);
        const auto sw = __builtin_amdgcn_permlane32_swap(t, t, false, false);
        hi_from_lo[G] = __uint_as_float(sw[0]);                // upper lanes: the lower half's column 8 G + 3
        lo_from_hi[G] = __uint_as_float(sw[1]);                // lower lanes: the upper half's column 8 G + 7
      }
      const int ch = mt * 32 + xl_;
#pragma unroll
      for (int G = 0; G < 4; ++G) {
        const bool second = PAIR && G >= 2;                    // (compile time) the group belongs to row qB
        const bool first_of_row = G == 0 || (PAIR && G == 2);
        const float before = g ? hi_from_lo[G] : (first_of_row ? (second ? prevB : prevA) : lo_from_hi[G ? G - 1 : 0]);
        const float ca = fmaxf(fmaxf(before, v[4 * G]), v[4 * G + 1]);                           // pooled column 4 G' + 2 g
        const float cb = fmaxf(fmaxf(v[4 * G + 1], v[4 * G + 2]), v[4 * G + 3]);                 // pooled column 4 G' + 2 g + 1
        const int Gc = PAIR ? (G & 1) : G;
        if (second ? okB : okA)
          *reinterpret_cast<float2*>(&lds.hbuf[second ? min(qB, kCR - 1) : qA][ch][4 * Gc + 2 * g]) = make_float2(ca, cb);
      }
      if (!PAIR && g == 1) lds.carry[qA][mt][xl_] = v[15];  // column 31, for the next chunk (this wave's own: no barrier)
    };
    if (a.Wc - 32 * ck <= 16) {                                // (uniform)
      for (int q = q0; q < kCR; q += 4) rows(q, q + 2, std::true_type{});
    } else {
      for (int q = q0; q < kCR; q += 2) rows(q, q, std::false_type{});
    }
    SCLK();
    STEM_BARRIER();
    SCLK();
    // ---- vertical 3-max (stride 2), bias, ReLU, store: 64 channels x 4 pooled rows x 16 pooled columns
    if ((a.Wp & 3) == 0) {
      // four columns per lane: 16-byte LDS reads and global stores (a group of four is inside the row or outside it).  A
      // thread keeps its pooled row p and its column group j4 over the four passes and moves 16 channels on: every LDS
      // read of the four passes (three rows of maxima and the bias, staged in LDS) goes out before the first is used --
      // written as a loop with the bias read from memory inside, each pass paid a round trip of its own (5 - 7 k of the
      // 22 k cycles of a chunk, round 5's stamps).
      const int j4 = tid & 3, p = (tid >> 2) & (kPH - 1), ch0 = tid >> 4;
      const int pr = pr0 + p, pc = 16 * ck + 4 * j4;
      if (pr < a.Hp && pc < a.Wp) {
        float4 h[4][3];
        float b[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
#pragma unroll
          for (int d = 0; d < 3; ++d) {
            const int q = 2 * p + d, cr = cr0 + q;
            h[it][d] = make_float4(ninf, ninf, ninf, ninf);
            if (cr >= 0 && cr < a.Hc) h[it][d] = *reinterpret_cast<const float4*>(&lds.hbuf[q][ch0 + 16 * it][4 * j4]);
          }
          b[it] = lds.bias[ch0 + 16 * it];
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const float4 m = make_float4(fmaxf(fmaxf(h[it][0].x, h[it][1].x), h[it][2].x), fmaxf(fmaxf(h[it][0].y, h[it][1].y), h[it][2].y),
                                       fmaxf(fmaxf(h[it][0].z, h[it][1].z), h[it][2].z), fmaxf(fmaxf(h[it][0].w, h[it][1].w), h[it][2].w));
          *reinterpret_cast<float4*>(a.y + (((long long)n * kO + ch0 + 16 * it) * a.Hp + pr) * a.Wp + pc) =
              make_float4(fmaxf(m.x + b[it], 0.f), fmaxf(m.y + b[it], 0.f), fmaxf(m.z + b[it], 0.f), fmaxf(m.w + b[it], 0.f));
        }
      }
    } else {
      for (int e = tid; e < kO * kPH * 16; e += 256) {
        const int j = e & 15, p = (e >> 4) & (kPH - 1), ch = e >> 6;
        const int pr = pr0 + p, pc = 16 * ck + j;
        if (pr < a.Hp && pc < a.Wp) {
          float m = ninf;
#pragma unroll
          for (int d = 0; d < 3; ++d) {
            const int q = 2 * p + d, cr = cr0 + q;
            if (cr >= 0 && cr < a.Hc) m = fmaxf(m, lds.hbuf[q][ch][j]);
          }
          a.y[(((long long)n * kO + ch) * a.Hp + pr) * a.Wp + pc] = fmaxf(m + lds.bias[ch], 0.f);
        }
      }
    }
  }
  SCLK();
#ifdef LSQ_STEM_CLOCKS
  // dev build: wave 0 of the workgroups 1024..1027 (a middle round) overwrites the start of y with its stamps: per chunk
  // (barrier, convert + barrier, rows: MFMAs + horizontal max, barrier, vertical max + stores)
  if (tid == 0 && blockIdx.x >= 1024 && blockIdx.x < 1028)
    for (int i = 0; i < 24; ++i) reinterpret_cast<long long*>(a.y)[(blockIdx.x - 1024) * 24 + i] = i < nclk ? clk[i] : 0;
#endif
  if constexpr (HALF) {
    if (a.overflow && __builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) atomicOr(a.overflow, 1);
  }
}

}  // namespace
}  // namespace lsq

using namespace lsq;

extern "C" int lsq_stem_conv_pool(const float* x, int N, int H, int W, const float* w, const float* bias, int split,
                                  float* y, int32_t* overflow, void* stream) {
  if (!x || !w || !bias || !y) return LSQ_E_NULL;
  if (N <= 0 || H < 8 || W < 8 || (W & 1) || ((uintptr_t)x % 8)) return LSQ_E_SHAPE;
  if (split != 2 && split != 3 && split != 22) return LSQ_E_SCHEME;
  StemArgs a = {};
  a.x = x; a.w = w; a.bias = bias; a.y = y; a.overflow = overflow;
  a.N = N; a.H = H; a.W = W;
  a.Hc = (H + 6 - 7) / 2 + 1;
  a.Wc = (W + 6 - 7) / 2 + 1;
  a.Hp = (a.Hc + 2 - 3) / 2 + 1;
  a.Wp = (a.Wc + 2 - 3) / 2 + 1;
  const long long blocks = (long long)N * ((a.Hp + kPH - 1) / kPH);
  if (blocks > 0x7FFFFFFF) return LSQ_E_SHAPE;
  if (split == 3) hipLaunchKernelGGL((stem_conv_pool_kernel<3, false>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  else if (split == 2) hipLaunchKernelGGL((stem_conv_pool_kernel<2, false>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((stem_conv_pool_kernel<2, true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
