// Binary x binary 3x3 convolution on the integer matrix cores of gfx950 (v_mfma_i32_32x32x32_i8) for 64, 128, 256
// and 512 input channels -- all sixteen quantized layers of ResNet-18.  Second implementation of
// F.conv2d(x_q, w_q, ...) of quant/binary/binary_conv.py:165-173 next to the popcount kernel (lsq_xnor_conv.hip, every
// other geometry): same operands, same epilogue arithmetic, bit-identical output (tests/test_gpu_parity.py).
//
// The popcount kernel spends two VALU instructions per 32 binary MACs (v_bcnt at half rate: 688 T MAC/s at best) and,
// with 64 channels, 38 % of its instructions outside the popcount core.  The matrix cores do 32768 MACs per
// instruction, but want bytes; a first attempt (bits expanded through an LDS table into an im2col patch, both
// operands read from LDS) was fed at a third of the rate it needed.  What makes this one work:
//   * the ACTIVATION operand is built from the packed word in registers with ONE v_and_b32 per four channels:
//     an MFMA sums over k in any order as long as both operands agree on it, so k-slot (register q, byte b) of a
//     lane is defined as channel q + 8 b of the lane's 32-bit half of the word, and `word & (0x01010101 << q)`
//     IS that register -- its bytes hold 0 or 2^q.  The weight byte of the same slot is +-(64 >> q), so every
//     product is +-64 or 0 and the accumulator is 64 * sum_c s_c [b_c = +1], an exact integer.  (q = 7 would be
//     the sign bit of an int8: that register is `(word >> 7) & 0x01010101` against weights of +-64.)
//   * the WEIGHTS are expanded once per workgroup (six VALU instructions per register) into LDS, 1 KB per (word, tap,
//     half) fragment, and streamed from there through a ring of three register fragments, two taps ahead; all waves
//     of a workgroup work on the same 32 out-channels and stride over its pixel tiles;
//   * no LDS patch and no barrier inside the tile loop: a lane loads the dword of ITS pixel and ITS half of the word
//     straight from the bit planes (the popcount kernel's coalesced access pattern), one channel word ahead of the
//     MFMAs that consume it; scheduling barriers pin the order of issue;
//   * the epilogue has no integer work: (b * s) = 2 * sum_c s_c [b_c = +1] - sum over the taps inside the image of
//     sum_c s_c; the second term depends only on the pixel's border pattern (a 4 KB table per workgroup) and is
//     loaded into the accumulators, times 32, before the first MFMA.  Residuals and scales are requested a tile ahead.
//
// Tile = 32 consecutive output pixels (flat over n, ho, wo) x 32 out-channels per wave; D[o][pixel]: a lane holds
// ONE pixel (column lane & 31) and 16 out-channels (rows (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)).

#include <type_traits>

#include "lsq_xnor_conv.h"
#ifndef LSQ_Y3_WAVES
#define LSQ_Y3_WAVES 12
#endif

namespace lsq {
namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr unsigned kM0 = 0x01010101u;

// The weight fragments are re-read from LDS for every tile (conflict-free ds_read_b128 at 256 B/clk: a quarter of the
// LDS bandwidth at full MFMA rate); keeping the 72 VGPRs of the 64-channel case in registers instead measured slower
// once the epilogue inputs were requested early (fewer waves per SIMD).
// NWAVES: waves per workgroup, all on the same 32 out-channels: 8 -- one workgroup per CU, two waves per SIMD -- for 256 and
// 512 channels (the fragments of 512 channels fill the LDS of a CU; at 256 channels one workgroup of 8 waves expands
// the weights and builds the tables ONCE where two workgroups of 4 did it twice: 59-60 -> 55-56 us per 14 x 14 layer, round 4).
//
// S3: kernels that read / write fp32 tensors in the THREE-STREAM ROW layout (include/lsq_hip.h, LSQ_LAYOUT_SPLIT3): element
// (c, pixel) of a sample's row lives in stream s = (c + pixel) % 3 at s * S + c * hp + pixel / 3 (hp floats per channel and
// stream, a multiple of 32; S = C * hp), so that the sub-sample e % 3 == 0 of the flat NCHW index e = c * HoWo + pixel --
// what the next layer's scale solve reads (quantization.py:63, skip = 3) -- is ONE contiguous third of the row instead of
// every third float of all of it: HoWo % 3 == 1 (every ResNet shape), hence e % 3 = (c + pixel) % 3.
// When the OUTPUT has that layout a tile is 32 pixels at stride 3 of ONE image -- pixel 96 group + j + 3 lane, j = 0, 1, 2,
// ceil(HoWo / 96) groups per image --, so the 32 lanes of a channel's store are 32 consecutive floats of one stream that
// START ON A 128-BYTE LINE (the first version -- e at (e % 3) S + e / 3, groups running across images -- had them start
// anywhere: every line was written by two tiles on two CUs, and the launch took 110 us against 75).  The bit-plane reads
// of such a tile are the 32 lanes' 24-byte windows back to back: 768 bytes without overlap where 32 neighbouring pixels share
// all but 272 -- so the three tiles of a group, which read the same lines shifted by one word, go to three waves of ONE
// workgroup at the same time (wave = (slot, j), NWAVES = 6: two groups per workgroup and round) and meet in the CU's L1; on
// three different CUs (the first version) the planes went through the L2s 2.8 times and the 56 x 56 layers took 141 us
// against 92.  Residual operands may have either layout.
//
// FP4 (round 6): the same kernel on v_mfma_scale_f32_32x32x64_f8f6f4 with both operands in fp4 (E2M1) and unit block scales --
// the matrix cores' fastest format, twice the int8 rate (MI355X_MICROARCH.md: 9.1 PF measured against 4.4 POP/s), and
// still EXACT here: a sign bit of the packed word becomes an fp4 code with ONE v_and_b32 per EIGHT channels -- nibble j of
// register q of a lane is channel q + 4 j of its dword, `d & (0x11111111 << q)` holds codes 1 / 2 / 4 = 0.5 / 1 / 2 (q = 3 would
// be the sign bit: that register is `(d >> 3) & 0x11111111`, 0.5 again) --, the weight nibble of the same slot is +-(4, 2, 1,
// 4), so every product is +-2 or 0 and the fp32 accumulator, started at the border term fc, IS (b * s): integers below 2^13,
// exact in any order.  Half the operand-build instructions, half the MFMAs and half the LDS bytes per binary MAC;
// scripts/ubench/fp4_mfma_check.hip is the known-answer test of the instruction (subnormal code 1 = 0.5 included).
template <int KX, int GG, int TAPS, int NWAVES, int WPC, bool CHAIN, bool YS3, bool RS3, bool FP4>
__global__ __launch_bounds__(64 * NWAVES, WPC) void xnor_mfma_kernel(ConvArgs a) {
  constexpr bool S3 = YS3 || RS3;                // (which operands have the three-stream layout is fixed per instantiation:
                                                 //  runtime selects between two sets of sixteen channel offsets cost scalar registers)
  constexpr int NT = 64 * NWAVES;
  constexpr int FPS = FP4 ? 1 : 2;               // 16-byte weight fragments per (word, tap) step: int8 K = 32 twice, fp4 K = 64 once
  constexpr int NF = TAPS * GG * FPS;            // 16-byte operand fragments per lane
  constexpr int NW = TAPS * KX;                  // activation dwords per lane and group (= one channel word of a tile)
  __shared__ v4i s_w[NF][64];
  __shared__ int s_ws[TAPS][32];
  // fc[bad rows][bad columns][o] = sum of wsum over the taps OUTSIDE the image - sum over all taps: what a pixel
  // with that border pattern adds to 2 * (matrix-core sum).  3 x 3 taps: 8 x 8 patterns, one table look-up per
  // output instead of loops over kernel rows and columns.
  __shared__ __attribute__((aligned(16))) short s_fc[8][8][32];    // |.| <= taps * channels = 4608
  __shared__ __attribute__((aligned(16))) float s_scale[32], s_bias[32], s_slope[32];
  __shared__ float s_nqs[CHAIN ? 32 : 1], s_nqt[CHAIN ? 32 : 1];      // chained layers: the next quantizer's folded batch norm
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int col = lane & 31, hh = lane >> 5;
  const int o0 = blockIdx.y * 32;

  // ---- tile bookkeeping (before the tables: the first loads go out early) --------------------------------------
  const unsigned total = (unsigned)(a.N * a.Ho * a.Wo);
  const int HoWo = a.Ho * a.Wo;
  constexpr bool map3 = YS3;                     // tiles of 32 pixels at stride 3: the output is a three-stream tensor
  // (stride-3 tiles: `tile` counts GROUPS of 96 pixels, the wave's j picks its third)
  const int GPI = (HoWo + 95) / 96;              // stride-3 tiles: groups per image
  const int ntiles = map3 ? a.N * GPI : (int)((total + 31u) >> 5);
  constexpr int kSlots = map3 ? NWAVES / 3 : NWAVES;       // tiles (groups) a workgroup works on at a time
  static_assert(!map3 || NWAVES % 3 == 0, "stride-3 tiles: three waves per group");
  const int wslot = map3 ? wid / 3 : wid, j3 = map3 ? wid - 3 * (wid / 3) : 0;
  const unsigned s3_S = (unsigned)(a.y_s3 ? a.y_s3 : a.res_s3), s3_h = (unsigned)a.s3_hp;      // floats per stream / per channel of a stream
  const unsigned* __restrict__ xd = reinterpret_cast<const unsigned*>(a.xplanes);
  const unsigned plane_stride = 2u * (unsigned)a.xplane_words;
  const int tstride = gridDim.x * kSlots;
  // a wave's pixel index advances by the same amount from tile to tile: (n, ho, wo) follow with adds and carries
  const unsigned dstep = (map3 ? 96u : 32u) * (unsigned)tstride;
  const int d_n = (int)(dstep / (unsigned)HoWo);
  const int d_r = (int)(dstep - (unsigned)d_n * (unsigned)HoWo);
  const int d_ho = d_r / a.Wo, d_wo = d_r - d_ho * a.Wo;
  const bool acc_in = a.accumulate != 0, fin = a.final_pass != 0;
  const bool want_pre = fin && a.res_pre, want_post = fin && a.res_post;
  const bool relu = fin && a.relu == LSQ_ACT_RELU, prelu = fin && a.relu >= LSQ_ACT_PRELU;
  const int ob = 4 * hh;                         // the lane's out-channel of register i: ob + (i & 3) + 8 (i >> 2)

  struct Pix {
    int n, ho, wo;
  };
  auto advance = [&](Pix& px) {
    px.wo += d_wo;
    px.ho += d_ho;
    px.n += d_n;
    if (px.wo >= a.Wo) {
      px.wo -= a.Wo;
      px.ho += 1;
    }
    if (px.ho >= a.Ho) {
      px.ho -= a.Ho;
      px.n += 1;
    }
  };
  // The three words a pixel's kernel row touches (kw = 0, 1, 2) are 24 contiguous bytes of the plane; the lane of
  // half hh = 0 takes the first 12 (word 0 low, word 0 high, word 1 low), the lane of half 1 the last 12 (word 1 high,
  // word 2 low, word 2 high): ONE 12-byte load per (kernel row, plane) instead of three 4-byte ones -- a vector-memory
  // instruction costs a 16-cycle address pass whatever its width, and with dword loads that pass, not the matrix
  // core, paced the layers (round 3: 1.20 -> 1.06 ms per forward).  The MFMA sums over k in any order, so dword d of
  // a lane simply IS k-step (row, d); the weight fragments are laid out to match (see the expansion below).
  typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
  auto request = [&](const Pix& px, int j, unsigned (&x)[NW]) {
    // dword index of (word, half) = 2 * word + half; words: [plane][n][GG][Hp][Wp]; lanes past the last pixel (last tile
    // only) read the words of image 0 and store nothing
    const int n = px.n < a.N ? px.n : 0;
    const unsigned base = 2u * (unsigned)(((n * GG + j) * a.Hp + px.ho * a.sh) * a.Wp + px.wo * a.sw) + 3u * (unsigned)hh;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int p = 0; p < KX; ++p) {
        const u32x3 v = *reinterpret_cast<const u32x3*>(xd + base + 2u * (unsigned)a.tap_xoff[3 * kh] + (unsigned)p * plane_stride);
        x[(3 * kh + 0) * KX + p] = v.x;
        x[(3 * kh + 1) * KX + p] = v.y;
        x[(3 * kh + 2) * KX + p] = v.z;
      }
  };

  // Element offset of (sample, out-channel o0 + ob + k, the lane's pixel) = v[k % 3] + koff(k), koff uniform: NCHW
  // v[.] = ((n O + o0 + ob) HoWo + pixel), koff = k HoWo; three streams: with u = o0 + ob + pixel, channel k sits in stream
  // (u + k) % 3 at (o0 + ob + k) hp + pixel / 3 -- so v[m] = n * 3 S + ((u + m) % 3) S + (o0 + ob) hp + pixel / 3, koff = k hp.
  struct Lay {
    unsigned v[S3 ? 3 : 1];
  };
  auto lay_of = [&](unsigned pix, int ln, bool s3, Lay& L) {
    if constexpr (S3) {
      if (s3) {
        const unsigned u0 = (unsigned)(o0 + ob) + pix;
        const unsigned r0 = u0 - 3u * (__umulhi(u0, 0xAAAAAAABu) >> 1);
        const unsigned base = (unsigned)ln * 3u * s3_S + (unsigned)(o0 + ob) * s3_h + (__umulhi(pix, 0xAAAAAAABu) >> 1);
        L.v[0] = base + r0 * s3_S;
        L.v[1] = base + (r0 == 2u ? 0u : r0 + 1u) * s3_S;
        L.v[2] = base + (r0 == 0u ? 2u : r0 - 1u) * s3_S;
        return;
      }
    }
    const unsigned yoff = (unsigned)((ln * a.O + o0 + ob) * HoWo) + pix;
#pragma unroll
    for (int m = 0; m < (S3 ? 3 : 1); ++m) L.v[m] = yoff;
  };
  constexpr bool y_s3 = YS3, r_s3 = RS3;
  // (uniform) element offset of register i's channel k = (i & 3) + 8 (i >> 2) on top of Lay::v[k % 3]
  auto koff = [&](int i, bool s3) -> long long {
    const int k = (i & 3) + 8 * (i >> 2);
    return (S3 && s3) ? (long long)k * s3_h : (long long)k * HoWo;
  };
  auto kv = [&](const Lay& L, int i) -> unsigned { return L.v[S3 ? ((i & 3) + 8 * (i >> 2)) % 3 : 0]; };

  // wave-major numbering: when the tiles do not divide evenly, the waves with one tile more sit in different
  // workgroups (on different SIMDs) instead of filling one
  int tile = wslot * gridDim.x + blockIdx.x;
  const bool have_tile = tile < ntiles;
  // stride-3 tiles: group `t` = (image t / GPI, group t % GPI of that image); lanes past the image's last pixel (its last
  // group: 3136 = 32 * 96 + 64) are marked like the lanes past the last pixel of all, n = N
  auto pix3 = [&](int t) {
    const int gn = t / GPI, gi = t - gn * GPI;
    const int p = 96 * gi + j3 + 3 * col;
    Pix px;
    const bool in = p < HoWo;
    px.n = in ? gn : a.N;
    px.ho = in ? (int)((unsigned)p / (unsigned)a.Wo) : 0;
    px.wo = in ? p - px.ho * a.Wo : 0;
    return px;
  };
  Pix cur;
  if constexpr (map3) {
    cur = pix3(tile);
  } else {
    const unsigned p = (unsigned)tile * 32u + (unsigned)col;
    cur.n = (int)(p / (unsigned)HoWo);
    const int r = (int)(p - (unsigned)cur.n * (unsigned)HoWo);
    cur.ho = (int)((unsigned)r / (unsigned)a.Wo);
    cur.wo = r - cur.ho * a.Wo;
  }
  // the first tile's words are requested before the workgroup builds its tables: their latency hides behind it
  unsigned xc[NW];
  if (have_tile) request(cur, 0, xc);

  // ---- once per workgroup: weight fragments and epilogue tables of its 32 out-channels ----------------------------
  // one thread = one (fragment pair, lane) entry: the weight word's dword of the lane's half, eight registers of four
  // bytes +-(64 >> q): bytes (d >> q) & 0x01010101 -> 0x00 / 0xFF masks -> select between the two byte patterns
  // EVERY global load of the set-up goes out first -- the weight words of all of the thread's entries, its tap sums, the
  // channel's scale / bias / slope -- and is waited for once: written as "load, expand, store" per entry the loop paid one
  // L2 round trip per entry, up to nine in a row (round 4: the set-up was a tenth of the 256-channel launches).
  constexpr int kWEntries = TAPS * GG * 64, kWIter = (kWEntries + NT - 1) / NT;
  constexpr int kSIter = (TAPS * 32 + NT - 1) / NT;
  auto entry = [&](int e, int& L, int& tj, int& pos) {
    L = e & 63;
    tj = e >> 6;
    const int fj = tj / TAPS, ft = tj - fj * TAPS;                                               // fragment order: word-major
    // step ft = (kernel row, dword dd of the lane's 12 bytes): half 0 holds (kw 0 low, kw 0 high, kw 1 low), half 1
    // (kw 1 high, kw 2 low, kw 2 high)
    const int fkh = ft / 3, dd = ft - 3 * fkh, lh = L >> 5;
    pos = 3 * lh + dd;                                                                            // 0..5: dword of the 24 bytes
    const int tap = 3 * fkh + (pos >> 1);
    return (long long)(tap * GG + fj) * a.opad_total + o0 + (L & 31);                             // [tap][word][O]
  };
  unsigned long long wword[kWIter];
  int wsv[kSIter];
#pragma unroll
  for (int k = 0; k < kWIter; ++k) {
    int L, tj, pos;
    const int e = tid + k * NT;
    const long long at = entry(e < kWEntries ? e : 0, L, tj, pos);
    wword[k] = a.wbits[at];
  }
#pragma unroll
  for (int k = 0; k < kSIter; ++k) {
    const int i = min(tid + k * NT, TAPS * 32 - 1);
    wsv[k] = a.wsum[(long long)(o0 + (i & 31)) * TAPS + (i >> 5)];
  }
  float c_sc = 0.f, c_bi = 0.f, c_sl = 0.f, c_ns = 1.f, c_nt = 0.f;
  if (tid < 32) {
    c_sc = a.wscale[o0 + tid];
    c_bi = a.bias ? a.bias[o0 + tid] : 0.f;
    c_sl = a.relu >= LSQ_ACT_PRELU ? a.slope[a.relu == LSQ_ACT_PRELU ? 0 : o0 + tid] : 0.f;     // (0: ReLU)
    if constexpr (CHAIN) {
      c_ns = (a.nq_planes32 && a.nq_scale) ? a.nq_scale[o0 + tid] : 1.f;
      c_nt = (a.nq_planes32 && a.nq_scale) ? a.nq_shift[o0 + tid] : 0.f;
    }
  }
  // one thread = one (fragment pair, lane) entry: the weight word's dword of the lane's half, eight registers of four
  // bytes +-(64 >> q): bytes (d >> q) & 0x01010101 -> 0x00 / 0xFF masks -> select between the two byte patterns
#pragma unroll
  for (int k = 0; k < kWIter; ++k) {
    const int e = tid + k * NT;
    if (e < kWEntries) {
      int L, tj, pos;
      entry(e, L, tj, pos);
      const unsigned long long w = wword[k];
      const unsigned d = (pos & 1) ? (unsigned)(w >> 32) : (unsigned)w;
      if constexpr (FP4) {
        // nibble j of register q = channel q + 4 j: +-(4, 2, 1, 4) as E2M1 codes 6 / 4 / 2 / 6, sign bit 8 where the weight is -1
        v4i out;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const unsigned mag = (q == 1 ? 0x4u : (q == 2 ? 0x2u : 0x6u)) * 0x11111111u;
          const unsigned ones = (d >> q) & 0x11111111u;
          out[q] = (int)(mag | ((ones ^ 0x11111111u) << 3));
        }
        s_w[tj][L] = out;
      } else {
      v4i out[2];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const unsigned mag = q < 7 ? (64u >> q) : 64u;
        const unsigned pos = mag * kM0, neg = ((256u - mag) & 0xFFu) * kM0;
        const unsigned ones = (d >> q) & kM0;
        const unsigned mask = (ones << 8) - ones;                                                // 0xFF where the bit is set
        out[q >> 2][q & 3] = (int)(neg ^ ((pos ^ neg) & mask));
      }
      s_w[2 * tj][L] = out[0];
      s_w[2 * tj + 1][L] = out[1];
      }
    }
  }
#pragma unroll
  for (int k = 0; k < kSIter; ++k) {
    const int i = tid + k * NT;
    if (i < TAPS * 32) s_ws[i >> 5][i & 31] = wsv[k];
  }
  if (tid < 32) {
    s_scale[tid] = c_sc;
    s_bias[tid] = c_bi;
    s_slope[tid] = c_sl;
    if constexpr (CHAIN) {
      s_nqs[tid] = c_ns;
      s_nqt[tid] = c_nt;
    }
  }
  __syncthreads();
  for (int i = tid; i < 8 * 8 * 32; i += NT) {
    const int o = i & 31, bw = (i >> 5) & 7, bh = i >> 8;
    int v = 0;
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp) {
      const bool outside = ((bh >> (tp / 3)) & 1) | ((bw >> (tp % 3)) & 1);
      v -= outside ? 0 : s_ws[tp][o];
    }
    s_fc[bh][bw][o] = (short)v;
  }
  __syncthreads();

  if (!have_tile) return;

  // ---- tiles ------------------------------------------------------------------------------------------------------
  constexpr int kRing = 3;                       // weight fragments of kRing consecutive (word, tap) steps; 9 GG % kRing == 0
  v4i wring[kRing][FPS];
#pragma unroll
  for (int i = 0; i < kRing - 1; ++i) {
#pragma unroll
    for (int f = 0; f < FPS; ++f) wring[i][f] = s_w[FPS * i + f][lane];
  }
  const int unit_scale = 0x7F7F7F7F;             // E8M0 127 = 2^0 in every byte (fp4: the block scales of both operands)
#ifdef LSQ_XNOR_CLOCKS
  long long clk[15];
  int nclk = 0;
#define XCLK() do { __builtin_amdgcn_sched_barrier(0); if (nclk < 15) clk[nclk++] = (long long)__builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define XCLK() do {} while (0)
#endif
  for (;;) {
    XCLK();
    const int nxt = tile + tstride;
    const bool more = nxt < ntiles;
    Pix nx = cur;
    if (more) {
      if constexpr (map3) nx = pix3(nxt);
      else advance(nx);
    }

    // What the epilogue reads from memory is requested NOW, a whole tile of MFMAs ahead: the two waves of a SIMD run
    // in lockstep (same start, same tile length), so a load waited for in the epilogue stalls the matrix core for
    // its full latency -- that, not the MFMA rate, set the pace of the first version.
    const int ln = cur.n < a.N ? cur.n : 0;
    // 32-bit element offsets from uniform bases (the entry point admits outputs below 2^30 elements)
    // (three-stream kernels: the offsets are worked out where they are used -- here for the loads, again in front of the
    //  stores from an opaque copy of the pixel -- instead of living through the MFMA loop: the 128-channel kernel sits at
    //  its 168-register budget)
    Lay ly, lr;
    unsigned pixv = (unsigned)(cur.ho * a.Wo + cur.wo);
    lay_of(pixv, ln, y_s3, ly);
    lay_of(pixv, ln, r_s3, lr);
    float xs[KX], rv[16], basev[16];
    // (int8: the accumulators hold 32 (b * s), hence xs / 32 -- exact; fp4: they hold (b * s) itself)
    constexpr float kAccUnit = FP4 ? 1.0f : 0.03125f;
#pragma unroll
    for (int p = 0; p < KX; ++p) xs[p] = (!CHAIN || a.xscales) ? a.xscales[p * a.N + ln] * kAccUnit : 0.f;
    if (CHAIN && a.xunits)          // chained ls-1 layer: the scale is the exact row sum the producer's epilogue left, / M as the sweeps do
      xs[0] = (float)(((double)a.xunits[ln] * a.xunit) / a.xM) * kAccUnit;
    if (want_pre || want_post) {
      const float* __restrict__ rsrc = want_pre ? a.res_pre : a.res_post;
      if (a.res_stream) {           // (uniform) last use of a tensor that does not fit the Infinity Cache next to the output: lsq_xnor_conv.h
#pragma unroll
        for (int i = 0; i < 16; ++i) rv[i] = __builtin_nontemporal_load(rsrc + koff(i, r_s3) + kv(lr, i));
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) rv[i] = (rsrc + koff(i, r_s3))[kv(lr, i)];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) rv[i] = 0.f;
    }
    if (acc_in) {
#pragma unroll
      for (int i = 0; i < 16; ++i) basev[i] = (a.y + koff(i, y_s3))[kv(ly, i)];
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) basev[i] = s_bias[ob + (i & 3) + 8 * (i >> 2)];
    }
    __builtin_amdgcn_sched_barrier(0);

    // The accumulators start at 32 * fc (fc = what the pixel's border pattern adds to 2 * the matrix-core sum, from
    // the table): the matrix core does the epilogue's integer additions, and 64 * S + 32 * fc = 32 * (b * s).
    typedef typename std::conditional<FP4, v16f, v16i>::type acc_t;
    acc_t acc[KX];
    {
      const int hi0 = cur.ho * a.sh - a.ph, wi0 = cur.wo * a.sw - a.pw;
      unsigned bad_h = 0, bad_w = 0;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int hi = hi0 + k * a.dh, wi = wi0 + k * a.dw;
        bad_h |= (hi < 0 || hi >= a.H) ? 1u << k : 0u;
        bad_w |= (wi < 0 || wi >= a.W) ? 1u << k : 0u;
      }
      const short* __restrict__ fcp = &s_fc[bad_h][bad_w][ob];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const short4 f4 = *reinterpret_cast<const short4*>(fcp + 8 * g);
#pragma unroll
        for (int p = 0; p < KX; ++p) {
          if constexpr (FP4) {
            acc[p][4 * g + 0] = (float)f4.x;
            acc[p][4 * g + 1] = (float)f4.y;
            acc[p][4 * g + 2] = (float)f4.z;
            acc[p][4 * g + 3] = (float)f4.w;
          } else {
            acc[p][4 * g + 0] = (int)f4.x << 5;
            acc[p][4 * g + 1] = (int)f4.y << 5;
            acc[p][4 * g + 2] = (int)f4.z << 5;
            acc[p][4 * g + 3] = (int)f4.w << 5;
          }
        }
      }
    }
    // Order of issue, enforced with scheduling barriers (left alone, the compiler sinks the prefetches to a few
    // instructions before their use): the words of the NEXT group first -- the next channel word of this tile, or
    // the first word of the wave's next tile: a whole group of 36 MFMAs ahead --, then per tap the weight fragments
    // of the tap after the next (a ring of three, wrapping around into the next tile: the fragments are the same
    // for every tile), then this tap's 18 v_and and 4 MFMAs.
    int wl = lane;                               // opaque per tile: the fragment reads must not be hoisted out of
    asm volatile("" : "+v"(wl));                 // the tile loop (36 and more fragments)
#pragma unroll
    for (int j = 0; j < GG; ++j) {
      unsigned xn[NW];
      if (j + 1 < GG) request(cur, j + 1, xn);
      else if (more) request(nx, 0, xn);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        constexpr int kAll = GG * TAPS;
        const int idx = j * TAPS + t;
        const int pre = (idx + kRing - 1) % kAll;
#pragma unroll
        for (int f = 0; f < FPS; ++f) wring[(idx + kRing - 1) % kRing][f] = s_w[FPS * pre + f][wl];
        __builtin_amdgcn_sched_barrier(0);         // reads first: they must not sink below this step's MFMAs
        if constexpr (FP4) {
          const v4i w4 = wring[idx % kRing][0];
          const v8i w8 = {w4[0], w4[1], w4[2], w4[3], 0, 0, 0, 0};      // (fp4 operands: the low four registers are read)
          v8i b8[KX];
#pragma unroll
          for (int p = 0; p < KX; ++p) {
            const unsigned d = xc[t * KX + p];
            b8[p] = v8i{(int)(d & 0x11111111u), (int)(d & 0x22222222u), (int)(d & 0x44444444u), (int)((d >> 3) & 0x11111111u), 0, 0, 0, 0};
          }
#pragma unroll
          for (int p = 0; p < KX; ++p)
            acc[p] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w8, b8[p], acc[p], 4, 4, 0, unit_scale, 0, unit_scale);
          // (the scaled MFMA is a pure value to the instruction selector, which -- unlike with the int8 builtin -- gathered all of a
          //  tile's fragment reads in front of the first one and spilled them; an empty asm on the accumulators ties each step
          //  to its place between the scheduling barriers)
#pragma unroll
          for (int p = 0; p < KX; ++p) asm volatile("" : "+v"(acc[p]));
        } else {
          const v4i w0 = wring[idx % kRing][0], w1 = wring[idx % kRing][FPS - 1];
          v4i b0[KX], b1[KX];
#pragma unroll
          for (int p = 0; p < KX; ++p) {
            const unsigned d = xc[t * KX + p];
            b0[p][0] = (int)(d & kM0);
            b0[p][1] = (int)(d & (kM0 << 1));
            b0[p][2] = (int)(d & (kM0 << 2));
            b0[p][3] = (int)(d & (kM0 << 3));
            b1[p][0] = (int)(d & (kM0 << 4));
            b1[p][1] = (int)(d & (kM0 << 5));
            b1[p][2] = (int)(d & (kM0 << 6));
            b1[p][3] = (int)((d >> 7) & kM0);
          }
#pragma unroll
          for (int p = 0; p < KX; ++p) acc[p] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w0, b0[p], acc[p], 0, 0, 0);
#pragma unroll
          for (int p = 0; p < KX; ++p) acc[p] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w1, b1[p], acc[p], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (j + 1 < GG || more) {
#pragma unroll
        for (int i = 0; i < NW; ++i) xc[i] = xn[i];
      }
    }

    XCLK();
    // ---- epilogue of this tile: the popcount kernel's arithmetic on the same integers -> the same floats --------
    float nqv[CHAIN ? 16 : 1];                   // (chained layers: the tile's outputs, kept for the next layer's quantizer)
    if constexpr (CHAIN) {
#pragma unroll
      for (int i = 0; i < 16; ++i) nqv[i] = 0.f;
    }
    if constexpr (S3) {
      asm volatile("" : "+v"(pixv));
      lay_of(pixv, ln, y_s3, ly);
      if (want_post && want_pre) lay_of(pixv, ln, r_s3, lr);
    }
    if (cur.n < a.N) {                           // (false only for the lanes past the last pixel)
      // float(acc) = 32 * (b * s) exactly (|.| < 2^18) and xs / 32 is exact, so (xs / 32) * float(acc) is the very
      // product xs * float(b * s) of the popcount kernel
      float4 scv[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) scv[g] = *reinterpret_cast<const float4*>(&s_scale[ob + 8 * g]);
      float outv[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float v = xs[0] * (float)acc[0][i];
#pragma unroll
        for (int p = 1; p < KX; ++p) v = fmaf(xs[p], (float)acc[p][i], v);
        outv[i] = fmaf(v, reinterpret_cast<const float*>(&scv[i >> 2])[i & 3], basev[i]);
      }
#ifdef LSQ_XNOR_CLOCKS
      asm volatile("" :: "v"(outv[0]), "v"(outv[15]));
      XCLK();
#endif
      if (want_pre) {
#pragma unroll
        for (int i = 0; i < 16; ++i) outv[i] += rv[i];
      }
      if (relu) {
#pragma unroll
        for (int i = 0; i < 16; ++i) outv[i] = fmaxf(outv[i], 0.f);
      }
      if (prelu) {
        // (slopes as four 16-byte reads and the product taken unconditionally: written as `v > 0 ? v : slope * v` with the
        //  slope read inside, the compiler emitted sixteen branches with an LDS round trip each)
        float4 slv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) slv[g] = *reinterpret_cast<const float4*>(&s_slope[ob + 8 * g]);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float neg = reinterpret_cast<const float*>(&slv[i >> 2])[i & 3] * outv[i];
          outv[i] = outv[i] > 0.f ? outv[i] : neg;
        }
      }
      if (want_post && !want_pre) {
#pragma unroll
        for (int i = 0; i < 16; ++i) outv[i] += rv[i];
      }
      if (want_post && want_pre) {
#pragma unroll
        for (int i = 0; i < 16; ++i) outv[i] += (a.res_post + koff(i, r_s3))[kv(lr, i)];
      }
#ifdef LSQ_XNOR_CLOCKS
      asm volatile("" :: "v"(outv[0]), "v"(outv[15]));
      XCLK();
#endif
#pragma unroll
      for (int i = 0; i < 16; ++i) (a.y + koff(i, y_s3))[kv(ly, i)] = outv[i];
      if constexpr (CHAIN) {
#pragma unroll
        for (int i = 0; i < 16; ++i) nqv[i] = outv[i];
      }
    }
    if (CHAIN && fin && a.nq_planes32) {         // (uniform; compiled into the one-plane kernels only)
      // The NEXT layer's 1-bit quantizer (quantizer_ls_1 behind the block's batch norm and clamp, quantization.py:35-56)
      // on the values this tile just produced: sign bits straight into the next plane -- this wave's 32 out-channels are
      // one dword of the next layer's channel word --, and sum |x| in the exact arithmetic of the plain sweeps
      // (lsq_act_quant.hip, EXACT): the fp32 sum of every octet of channels in channel order, rounded to a multiple of
      // 2^e; such multiples add exactly, here as 64-bit integers in units of 2^e (atomics in any order give the same
      // total).  A lane of half hh holds channels 4 hh + (i & 3) + 8 (i >> 2): octet g = registers 4 g .. 4 g + 3 of
      // both halves; half 0 sums its four, half 1 continues the chain with its own.
      const bool valid = cur.n < a.N;
      unsigned mask = 0u;
      double units = 0.0;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        float av[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = 4 * g4 + k;
          const int cl = ob + k + 8 * g4;                                    // channel inside the 32-channel tile
          float t = valid ? nqv[CHAIN ? i : 0] : 0.f;
          if (a.nq_scale) t = fmaf(t, s_nqs[CHAIN ? cl : 0], s_nqt[CHAIN ? cl : 0]);
          t = fminf(fmaxf(t, -a.nq_alpha), a.nq_alpha);
          mask |= (t >= 0.f ? 1u : 0u) << (ob + k + 8 * g4);
          av[k] = fabsf(t);
        }
        const float lo4 = ((av[0] + av[1]) + av[2]) + av[3];                  // channels 8 g .. 8 g + 3 (half 0)
        // (v_permlane32_swap: the upper 32 lanes receive the lower 32 lanes' value -- one VALU instruction where a
        //  shuffle is an LDS round trip)
        const float from0 = __uint_as_float(__builtin_amdgcn_permlane32_swap(__float_as_uint(lo4), __float_as_uint(lo4), false, false)[0]);
        const float oct = (((from0 + av[0]) + av[1]) + av[2]) + av[3];        // half 1: ... + channels 8 g + 4 .. 8 g + 7
        const double r = ((double)oct + a.nq_magic) - a.nq_magic;
        units += (hh == 1 && valid) ? r * a.nq_inv_unit : 0.0;                // an integer below 2^53
      }
      mask |= __builtin_amdgcn_permlane32_swap(mask, mask, false, false)[1];   // lower lanes: | the upper lanes' bits
      if (valid && hh == 0) {
        const int j = o0 >> 6, half = (o0 >> 5) & 1;
        a.nq_planes32[2u * (unsigned)(((cur.n * (a.O >> 6) + j) * a.nq_Hp + cur.ho + a.nq_ph) * a.nq_Wp + cur.wo + a.nq_pw) + (unsigned)half] = mask;
      }
      // one atomic per wave and sample of the tile (small images: a tile of 32 pixels spans two samples): the lanes of
      // the first remaining sample are summed with DPP shifts (fp64 holds these integers exactly), then the next sample
      bool todo = valid && hh == 1;
      while (__builtin_amdgcn_ballot_w64(todo) != 0ull) {
        const int lead = __builtin_ctzll(__builtin_amdgcn_ballot_w64(todo));
        const int n_cur = __builtin_amdgcn_readlane(cur.n, lead);
        const bool mine = todo && cur.n == n_cur;
        const double tot = wave_incl_scan(mine ? units : 0.0);
        const unsigned long long total = (unsigned long long)__longlong_as_double(
            ((long long)__builtin_amdgcn_readlane((int)(__double_as_longlong(tot) >> 32), 63) << 32) |
            (unsigned)__builtin_amdgcn_readlane((int)__double_as_longlong(tot), 63));
        if (lane == 0) atomicAdd(&a.nq_units[n_cur], total);
        todo = todo && !mine;
      }
    }

    XCLK();
    if (!more) break;
    tile = nxt;
    cur = nx;
  }
#ifdef LSQ_XNOR_CLOCKS
  // dev build: the time stamps of (tile start, main loop end, epilogue end) x 4 tiles of wave 0 of workgroups 0..3 overwrite
  // the start of y (the caller reads them as int64)
  if (lane == 0 && wid == 0 && blockIdx.y == 0 && blockIdx.x < 4)
    for (int i = 0; i < 15; ++i) reinterpret_cast<long long*>(a.y)[blockIdx.x * 15 + i] = i < nclk ? clk[i] : 0;
#endif
}

// three-stream OUTPUT: workgroups of six waves (two groups of 96 pixels at a time), two per CU -- the same three waves per
// SIMD and 168 registers as the NCHW kernels
template <int KX, int GG>
int launch_y3(const ConvArgs& a, hipStream_t st) {
  constexpr int NWAVES = LSQ_Y3_WAVES, WPC = 3, kSlots = NWAVES / 3;   // (WPC: waves per SIMD the registers must allow)
  const long long total = (long long)a.N * a.Ho * a.Wo;
  const long long ngroups = (long long)a.N * (((long long)a.Ho * a.Wo + 95) / 96);
  const int n_ot = a.O / 32;
  long long gx = (256 * (12 / NWAVES) + n_ot - 1) / n_ot;
  if (gx * kSlots > ngroups) gx = (ngroups + kSlots - 1) / kSlots;
  gx = gx < 1 ? 1 : gx;
  const dim3 grid((unsigned)gx, (unsigned)n_ot), block(64 * NWAVES);
  if (a.res_s3) hipLaunchKernelGGL((xnor_mfma_kernel<KX, GG, 9, NWAVES, WPC, false, true, true, true>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((xnor_mfma_kernel<KX, GG, 9, NWAVES, WPC, false, true, false, true>), grid, block, 0, st, a);
  return (int)hipGetLastError();
}

template <int KX, int GG, bool FP4 = true>
int launch(const ConvArgs& a, hipStream_t st) {
  // launch shape per channel count: waves per workgroup, workgroups per CU, and WPC = the waves per SIMD the register
  // allocation must allow (__launch_bounds__'s second argument).  The fp4 kernel runs as ONE workgroup of TWELVE waves per CU
  // for every channel count: the set-up -- weight expansion, tables -- is paid once per CU, and 145 / 165 / 167 / 159 registers
  // allow three waves per SIMD (the int8 kernel's 181 allowed two at 256 / 512 channels).  Measured (scripts/xnor_shapes.py,
  // profiles/r06_xnor_ablation.txt): 64 / 128 channels against rounds 2-5's three workgroups of 4 waves 91.3 -> 89.9 / 55.1 ->
  // 51.1 / 60.8 -> 57.2 / 43.5 -> 39.7 us per layer (two workgroups of 6: 119 us -- the second one does not always fit its
  // waves onto the SIMDs the first left free); 256 / 512 channels against one workgroup of 8: 49.0 -> 47.5 / 29.8 -> 28.0 /
  // 40.9 -> 40.1.
#ifndef LSQ_G4_SHAPE
#define LSQ_G4_SHAPE 12, 1, 3
#endif
#ifndef LSQ_G8_SHAPE
#define LSQ_G8_SHAPE 12, 1, 3
#endif
#ifndef LSQ_G1_SHAPE
#define LSQ_G1_SHAPE 12, 1, 3
#endif
  constexpr int kShape[5][3] = {{LSQ_G1_SHAPE}, {LSQ_G4_SHAPE}, {LSQ_G8_SHAPE}, {8, 1, 1}, {4, 3, 3}};
  constexpr int kS = !FP4 ? (GG < 4 ? 4 : 3) : (GG >= 8 ? 2 : (GG >= 4 ? 1 : 0));   // (the int8 comparator: its shapes of rounds 4-5)
  constexpr int NWAVES = kShape[kS][0], kWgsPerCu = kShape[kS][1], WPC = kShape[kS][2];
  const long long total = (long long)a.N * a.Ho * a.Wo;
  const long long ntiles = (total + 31) >> 5;
  const int n_ot = a.O / 32;
  // every resident wave strides over the pixel tiles of its workgroup's out-channel tile
  const int wgs = 256 * kWgsPerCu;
  long long gx = (wgs + n_ot - 1) / n_ot;
  gx = gx < 1 ? 1 : gx;
  if (gx * NWAVES > ntiles) gx = (ntiles + NWAVES - 1) / NWAVES;
  if constexpr (FP4) if (a.y_s3 || a.res_s3) {
    // three-stream tensors: the two-plane kernels of 64 / 128 channels (the 56 x 56 and 28 x 28 layers of the network, whose
    // rows are the long ones); a three-stream OUTPUT means tiles at pixel stride 3, whose bookkeeping wants the tile stride
    // of a wave to be a multiple of 3
    if constexpr (KX == 2 && GG <= 2) {
      if (a.y_s3) return launch_y3<KX, GG>(a, st);
      hipLaunchKernelGGL((xnor_mfma_kernel<KX, GG, 9, NWAVES, WPC, false, false, true, true>), dim3((unsigned)gx, (unsigned)n_ot), dim3(64 * NWAVES), 0, st, a);
      return (int)hipGetLastError();
    } else {
      return kXnorMfmaNoLayout;
    }
  }
  if constexpr (!FP4) {
    hipLaunchKernelGGL((xnor_mfma_kernel<KX, GG, 9, NWAVES, WPC, false, false, false, false>), dim3((unsigned)gx, (unsigned)n_ot), dim3(64 * NWAVES), 0, st, a);
  } else {
    if (KX == 1 && (a.xunits || a.nq_planes32))
      hipLaunchKernelGGL((xnor_mfma_kernel<KX, GG, 9, NWAVES, WPC, KX == 1, false, false, true>), dim3((unsigned)gx, (unsigned)n_ot), dim3(64 * NWAVES), 0, st, a);
    else
      hipLaunchKernelGGL((xnor_mfma_kernel<KX, GG, 9, NWAVES, WPC, false, false, false, true>), dim3((unsigned)gx, (unsigned)n_ot), dim3(64 * NWAVES), 0, st, a);
  }
  return (int)hipGetLastError();
}

template <int KX>
int launch_gg(const ConvArgs& a, hipStream_t st) {
  // (test hook lsq_debug_xnor_impl(2): rounds 2-5's int8 kernel, the fp4 kernel's comparator -- same bits; plain calls only)
  if (a.int8_mfma && !a.y_s3 && !a.res_s3 && !a.xunits && !a.nq_planes32) {
    switch (a.cg) {
      case 64: return launch<KX, 1, false>(a, st);
      case 128: return launch<KX, 2, false>(a, st);
      case 256: return launch<KX, 4, false>(a, st);
      case 512: return launch<KX, 8, false>(a, st);
    }
  }
  switch (a.cg) {
    case 64: return launch<KX, 1>(a, st);
    case 128: return launch<KX, 2>(a, st);
    case 256: return launch<KX, 4>(a, st);
    case 512: return launch<KX, 8>(a, st);
  }
  return kXnorMfmaNotEligible;
}

}  // namespace

int xnor_conv_mfma(const ConvArgs& a, int kx, int groups, hipStream_t st) {
  const long long total = (long long)a.N * a.Ho * a.Wo;
  if (groups != 1 || a.KH != 3 || a.KW != 3 || a.dw != 1 || a.O % 32 || total * a.O >= (1ll << 30) || a.xplane_words * kx >= (1ll << 30))
    return kXnorMfmaNotEligible;
  return kx == 2 ? launch_gg<2>(a, st) : launch_gg<1>(a, st);
}

}  // namespace lsq
