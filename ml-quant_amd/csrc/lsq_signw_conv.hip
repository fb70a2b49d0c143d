// Full-precision activation x sign-weight convolution on the gfx950 matrix cores.
//
// Replaces F.conv2d(x, w_q, ...) of quant/binary/binary_conv.py:165-173 when x_quant == 'fp' and the
// weights are sums of scaled sign planes:  y[n,o] = bias[o] + sum_q u_q[o] * conv(clamp(x), s_q)[n,o].
//
// Implicit GEMM, transposed so that stores are coalesced:  D[o][pixel] = sum_k S[o][k] * X[k][pixel],
// k = (tap, channel).  v_mfma_f32_32x32x16_bf16: A = 32 out-channels x 16 channels of +-1 (exact in
// bf16), B = 16 channels x 32 pixels of activations.  fp32 activations are split x = hi + lo with
// hi = bf16(x) and lo = bf16(x - hi): two MFMAs per tile, relative error <= 2^-18 per product, fp32
// accumulation -- inside the 1e-4 bound where single bf16 (2^-9) is not.  Zero padding and channel
// padding are exact (x = 0 contributes 0).
//
// Two kernels.  signw_conv_patch (any stride whose input patch fits LDS -- every ResNet layer): per 16-channel
// chunk the workgroup keeps the whole input patch its output pixels touch in LDS (already batch-norm-folded,
// clamped and split) together with the expanded +-1 weights of all taps, and the taps read their fragments
// at precomputed addresses -- each input element is loaded and converted once per workgroup instead of
// once per tap.  signw_conv_tiled (fall-back): im2col staging per (tap, 32-channel chunk).

#include "lsq_signw_conv.h"

namespace lsq {
namespace signw {
namespace {

// ---------------------------------------------------------------------------------------------
// Tiled (im2col) version: a 256-thread workgroup computes BM out-channels x 128 pixels.  Per K-chunk (one tap,
// 32 channels) the block stages into LDS, ONCE for its four waves, (a) the +-1 weight fragment expanded
// from 4 bytes of the packed plane per out-channel and (b) the activation chunk already clamped /
// batch-norm-folded and split into bf16 hi and lo, both as [row][32 k] with a 16-byte row pad so that
// the 16-byte MFMA fragment reads of a 32-lane group hit distinct banks.  Global loads of chunk i+1
// are issued before the MFMAs of chunk i and written to LDS after them (one buffer, two barriers per
// chunk: half the LDS, three workgroups per CU, measured faster than two buffers and one barrier).  Each wave owns TM x TN 32x32 tiles: 2*TM*TN*2 MFMAs per chunk against
// (TM + 2*TN)*2 fragment reads.
constexpr int kKC = 32;                         // channels per K-chunk
constexpr int kRowB = kKC * 2 + 16;             // LDS row pitch in bytes (64 data + 16 pad)
constexpr int kBN = 128;                        // pixels per block
#ifndef LSQ_SIGNW_LDS_BUFS
#define LSQ_SIGNW_LDS_BUFS 1
#endif
constexpr int kLdsBufs = LSQ_SIGNW_LDS_BUFS;    // 2: one barrier per chunk; 1: two barriers, half the LDS, more blocks per CU

template <int BM, int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256) void signw_conv_tiled(SwArgs a) {
  static_assert(WM * WN == 4 && WM * TM * 32 == BM && WN * TN * 32 == kBN, "tile shape");
  __shared__ __attribute__((aligned(16))) unsigned char smem[kLdsBufs][(BM + 2 * kBN) * kRowB];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid - wm * WN;
  const int col = lane & 31, kh8 = lane >> 5;
  const int tile = blockIdx.y;
  const int grp = tile / a.tiles_per_group;
  const int t = tile - grp * a.tiles_per_group;
  const int o_pad0 = grp * a.og_pad + t * BM;
  const int o0 = grp * a.og + t * BM;
  const int HoWo = a.Ho * a.Wo, HW = a.H * a.W;
  const long long total = (long long)a.N * HoWo;
  const int taps = a.KH * a.KW;
  const int cchunks = (a.cg + kKC - 1) / kKC;
  const int nchunks = taps * cchunks;

  // ---- staging roles
  // activations: thread -> pixel (tid & 127), k half (tid >> 7) of 16 channels
  const int sp = tid & (kBN - 1), skh = tid >> 7;
  const long long spix = (long long)blockIdx.x * kBN + sp;
  const bool sp_valid = spix < total;
  const long long spc = sp_valid ? spix : 0;
  const int sn = (int)(spc / HoWo);
  const int sr = (int)(spc - (long long)sn * HoWo);
  const int sho = sr / a.Wo, swo = sr - sho * a.Wo;
  const float* xg = a.x + ((long long)sn * a.C + (long long)grp * a.cg) * HW;
  // weights: thread -> out-channel (tid % BM), k part (tid / BM) of KC / (256 / BM) channels
  constexpr int WPARTS = 256 / BM;              // 2 (BM = 128) or 4 (BM = 64)
  constexpr int WK = kKC / WPARTS;              // 16 or 8 channels per thread
  const int so = tid % BM, swp = tid / BM;
  const bool so_valid = t * BM + so < a.og_pad;

  float xr[16];
  unsigned wr = 0;
  auto load_chunk = [&](int ch) {
    const int tap = ch / cchunks, cc = ch - tap * cchunks;
    const int kh = tap / a.KW, kw = tap - kh * a.KW;
    const int hi = sho * a.sh - a.ph + kh * a.dh, wi = swo * a.sw - a.pw + kw * a.dw;
    const bool inb = sp_valid && hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
    const int cbase = cc * kKC + skh * 16;
    // all 16 loads are issued unconditionally (out-of-image / padded channels read a valid dummy
    // address and are zeroed afterwards): a load inside a divergent branch is waited for on the spot,
    // which serialises 16 memory latencies per chunk
    const float* xp = inb ? xg + (long long)hi * a.W + wi : xg;
    float raw[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) raw[j] = xp[(long long)min(cbase + j, a.cg - 1) * HW];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int c = cbase + j;
      float v = raw[j];
      if (a.pre_scale) {
        // the channel is the same for all lanes of a wave (k half = tid >> 7): scalar loads
        const int cs = __builtin_amdgcn_readfirstlane(grp * a.cg + min(c, a.cg - 1));
        v = fmaf(v, a.pre_scale[cs], a.pre_shift[cs]);
      }
      v = clamp_sym(v, a.alpha);
      xr[j] = (inb && c < a.cg) ? v : 0.f;
    }
    const int c0 = cc * kKC;
    const int g = c0 >> 6;
    const int bit0 = (c0 & 63) + swp * WK;
    const unsigned long long wv =
        so_valid ? a.wbits[((long long)tap * a.Gg + g) * a.opad_total + o_pad0 + so] : 0ull;
    wr = (unsigned)(wv >> bit0) & ((1u << WK) - 1u);
  };
  auto store_chunk = [&](int buf) {
    unsigned char* sA = smem[buf];
    unsigned char* sBh = sA + BM * kRowB;
    unsigned char* sBl = sBh + kBN * kRowB;
    // activations -> bf16 hi and lo = bf16(x - hi) (x - hi is exact in fp32; |x - hi - lo| <= 2^-18 |x|),
    // 16 values = 32 bytes each
    unsigned hi[8], lo[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) split_pair(xr[2 * p], xr[2 * p + 1], hi[p], lo[p]);
    uint4* dh = reinterpret_cast<uint4*>(sBh + sp * kRowB + skh * 32);
    uint4* dl = reinterpret_cast<uint4*>(sBl + sp * kRowB + skh * 32);
    dh[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    dh[1] = make_uint4(hi[4], hi[5], hi[6], hi[7]);
    dl[0] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    dl[1] = make_uint4(lo[4], lo[5], lo[6], lo[7]);
    // weights: WK sign bits -> WK bf16 +-1
    unsigned* dw = reinterpret_cast<unsigned*>(sA + so * kRowB + swp * WK * 2);
#pragma unroll
    for (int p = 0; p < WK / 2; ++p) {
      const unsigned b0 = (wr >> (2 * p)) & 1u, b1 = (wr >> (2 * p + 1)) & 1u;
      dw[p] = 0x3F803F80u | ((b0 ^ 1u) << 15) | ((b1 ^ 1u) << 31);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  load_chunk(0);
  store_chunk(0);
  __syncthreads();
  for (int ch = 0; ch < nchunks; ++ch) {
    const int buf = kLdsBufs == 2 ? (ch & 1) : 0;
    if (ch + 1 < nchunks) load_chunk(ch + 1);          // in flight during the MFMAs below
    const unsigned char* sA = smem[buf];
    const unsigned char* sBh = sA + BM * kRowB;
    const unsigned char* sBl = sBh + kBN * kRowB;
#pragma unroll
    for (int ks = 0; ks < kKC / 16; ++ks) {
      Frag af[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const uint4 v = *reinterpret_cast<const uint4*>(sA + ((wm * TM + i) * 32 + col) * kRowB + ks * 32 + kh8 * 16);
        af[i].u[0] = v.x; af[i].u[1] = v.y; af[i].u[2] = v.z; af[i].u[3] = v.w;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = ((wn * TN + j) * 32 + col) * kRowB + ks * 32 + kh8 * 16;
        const uint4 vh = *reinterpret_cast<const uint4*>(sBh + row);
        const uint4 vl = *reinterpret_cast<const uint4*>(sBl + row);
        bh[j].u[0] = vh.x; bh[j].u[1] = vh.y; bh[j].u[2] = vh.z; bh[j].u[3] = vh.w;
        bl[j].u[0] = vl.x; bl[j].u[1] = vl.y; bl[j].u[2] = vl.z; bl[j].u[3] = vl.w;
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i].v, bh[j].v, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i].v, bl[j].v, acc[i][j], 0, 0, 0);
        }
    }
    if (kLdsBufs == 1) __syncthreads();          // everyone is done reading before the single buffer is rewritten
    if (ch + 1 < nchunks) store_chunk(kLdsBufs == 2 ? (buf ^ 1) : 0);
    __syncthreads();
  }

  store_tiles<size_t, BM, kBN, TM, TN>(a, acc, blockIdx.x, t, o0, wm, wn, col, kh8);
}

// ---------------------------------------------------------------------------------------------
// Patch version (any stride / padding / dilation).  Padded coordinates hp = hi + pad_h, wp = wi + pad_w
// and the linear index L = (n*Hp + hp)*Wp + wp: output pixel (n, ho, wo) and tap (kh, kw) read
// L = B + T with B = (n*Hp + ho*stride_h)*Wp + wo*stride_w and T = kh*dil_h*Wp + kw*dil_w.  The BN consecutive output
// pixels of a workgroup therefore touch the contiguous range [B_first, B_last + T_max] of L -- the patch,
// at most PLr <= 512 entries (host-side bound, multiple of 128).  Per 16-channel chunk (one MFMA k-step)
// the workgroup
//   1. converts the patch once: lanes = consecutive entries = consecutive addresses of one channel
//      (coalesced), folded batch norm + clamp + bf16 hi / lo split, LDS rows [entry][16 channels] of 32
//      bytes in a hi and a lo plane;
//   2. expands the +-1 weights of up to 9 taps x BM out-channels x 16 channels from the packed plane
//      into LDS rows of 32 bytes, cooperatively (each 16-bit piece once per workgroup);
//   3. runs the taps back to back: A fragments at compile-time LDS offsets, B fragments at per-lane
//      addresses (entry of the lane's pixel + T) computed once per workgroup -- no address arithmetic,
//      no barrier and no global access inside the tap loop.
// The global loads of chunk i+1 (activations and weight words) are issued right before the tap loop of
// chunk i and consumed after it.  Rows have no padding: the two 16-byte halves of row r are swapped when
// bit 3 of r is set, which makes any 16 consecutive rows hit 16 distinct 4-bank groups.
// Each wave owns a 64 x 64 tile (2 x 2 MFMA tiles): 16 MFMAs per tap against 6 fragment reads.
constexpr int kTapGroup = 9;                     // taps whose weights are resident at a time


// PMAX: patch entries the LDS planes hold; PREMAX: channels per group the folded-batch-norm table holds.
// Stride 1: 128 x 128 (or 64 x 256) tiles, 64 x 64 per wave, PMAX 512.  Stride 2: only every second entry of a
// row feeds a given tap, so the same 128 pixels would need a patch twice as long: 128 x 64 tiles (64 x 32 per
// wave), PMAX 640.
template <int BM, int BN, int WM, int WN, int TN, int PMAX, int PREMAX, bool MANY_TAPS>
__global__ __launch_bounds__(256) void signw_conv_patch(SwArgs a, int Hp, int Wp, int PLr) {
  constexpr int TM = 2;
  constexpr int kPMaxEntries = PMAX, kPlane = PMAX * kPRow, kMaxPre = PREMAX;
  static_assert(WM * WN == 4 && WM * TM * 32 == BM && WN * TN * 32 == BN, "tile shape");
  constexpr int kItems = (kPMaxEntries * 2 + 255) / 256; // (entry, octet) items per thread, at most
  constexpr int WPARTS = 256 / BM;                       // threads per weight row
  constexpr int WTAPS = (kTapGroup + WPARTS - 1) / WPARTS;   // taps per thread and group
  __shared__ __attribute__((aligned(16))) unsigned char sP[2 * kPlane];
  __shared__ __attribute__((aligned(16))) unsigned char sW[kTapGroup * BM * kPRow];
  __shared__ __attribute__((aligned(16))) float sPre[2 * kMaxPre];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid - wm * WN;
  const int col = lane & 31, kh8 = lane >> 5;
  const int tile = blockIdx.y;
  const int grp = tile / a.tiles_per_group;
  const int t = tile - grp * a.tiles_per_group;
  const int o_pad0 = grp * a.og_pad + t * BM;
  const int o0 = grp * a.og + t * BM;
  const int HoWo = a.Ho * a.Wo, HW = a.H * a.W, HpWp = Hp * Wp;
  const int total = a.N * HoWo;
  const int taps = a.KH * a.KW;
  const int cchunks = (a.cg + kPC - 1) / kPC;
  const bool ragged = (a.cg % kPC) != 0;
  const float lim = a.alpha >= 0.f ? a.alpha : __builtin_inff();   // clamp_identity: no-op bounds

  // 32-bit index math throughout: the host takes this kernel only when N*Hp*Wp and N*C*H*W fit in 31 bits
  auto base_of = [&](int p) {
    const int n = p / HoWo;
    const int r = p - n * HoWo;
    const int ho = r / a.Wo;
    return (n * Hp + ho * a.sh) * Wp + (r - ho * a.Wo) * a.sw;
  };
  const int p0 = blockIdx.x * BN;
  const int bmin = base_of(p0);
  const float* xg = a.x + (long long)grp * a.cg * HW;

  if (a.pre_scale) {
    for (int c = tid; c < cchunks * kPC; c += 256) {
      const int cs = grp * a.cg + min(c, a.cg - 1);
      sPre[c] = a.pre_scale[cs];
      sPre[kMaxPre + c] = a.pre_shift[cs];
    }
  }

  // ---- per-lane fragment addresses (fixed for the whole kernel)
  int e_pix[TN];                                         // patch entry of this lane's pixel, per column tile
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int pix = p0 + (wn * TN + j) * 32 + col;
    e_pix[j] = pix < total ? base_of(pix) - bmin : 0;
  }
  int b_addr[kTapGroup][TN];                             // hi-plane byte address of the B fragment, first tap group
  {
    int kh = 0, kw = 0;
#pragma unroll
    for (int tt = 0; tt < kTapGroup; ++tt) {
      const int toff = kh * a.dh * Wp + kw * a.dw;
#pragma unroll
      for (int j = 0; j < TN; ++j) b_addr[tt][j] = swz(e_pix[j] + (tt < taps ? toff : 0), kh8);
      if (++kw == a.KW) { kw = 0; ++kh; }
    }
  }
  int a_addr[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) a_addr[i] = swz((wm * TM + i) * 32 + col, kh8);

  // ---- patch staging roles: item = (entry, channel octet); the octet is uniform per wave (PLr % 128 == 0)
  const int n_items = PLr >> 7;
  int it_off[kItems], it_dst[kItems];
#pragma unroll
  for (int u = 0; u < kItems; ++u) {
    const int i = tid + 256 * u;
    const int oct = i >= PLr;
    const int e = i - oct * PLr;
    const int L = bmin + e;
    const int n = L / HpWp;
    const int rem = L - n * HpWp;
    const int hp = rem / Wp;
    const int hi = hp - a.ph, wi = rem - hp * Wp - a.pw;
    const bool inside = u < n_items && n < a.N && hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
    it_off[u] = inside ? n * a.C * HW + hi * a.W + wi : -1;
    it_dst[u] = swz(e, oct) | (oct << 30);               // bit 30 carries the octet
  }
  float raw[kItems][8];
  auto issue_xloads = [&](int cc) {
    // address = wave-uniform channel base (scalar registers) + the lane's 32-bit byte offset, which never
    // changes: no vector address arithmetic.  Halo / padded channels read a valid dummy address and are
    // zeroed at conversion.
#pragma unroll
    for (int u = 0; u < kItems; ++u) {
      if (u < n_items) {
        const int c0 = __builtin_amdgcn_readfirstlane(cc * kPC + (it_dst[u] >> 30) * 8);
        const unsigned boff = it_off[u] < 0 ? 0u : (unsigned)it_off[u] * 4u;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const char* cbase = reinterpret_cast<const char*>(xg + (long long)min(c0 + j, a.cg - 1) * HW);
          raw[u][j] = *reinterpret_cast<const float*>(cbase + boff);
        }
      }
    }
  };
  auto convert_store = [&](int cc) {
#pragma unroll
    for (int u = 0; u < kItems; ++u) {
      if (u < n_items) {
        const int c0 = cc * kPC + (it_dst[u] >> 30) * 8;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = raw[u][j];
        if (a.pre_scale) {                               // wave-uniform addresses: broadcast LDS reads
          const float4* ps = reinterpret_cast<const float4*>(sPre + c0);
          const float4* pb = reinterpret_cast<const float4*>(sPre + kMaxPre + c0);
          const float4 s0 = ps[0], s1 = ps[1], b0 = pb[0], b1 = pb[1];
          v[0] = fmaf(v[0], s0.x, b0.x); v[1] = fmaf(v[1], s0.y, b0.y);
          v[2] = fmaf(v[2], s0.z, b0.z); v[3] = fmaf(v[3], s0.w, b0.w);
          v[4] = fmaf(v[4], s1.x, b1.x); v[5] = fmaf(v[5], s1.y, b1.y);
          v[6] = fmaf(v[6], s1.z, b1.z); v[7] = fmaf(v[7], s1.w, b1.w);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
          v[j] = (it_off[u] >= 0 && (!ragged || c0 + j < a.cg)) ? __builtin_amdgcn_fmed3f(v[j], -lim, lim) : 0.f;
        unsigned hi[4], lo[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) split_pair(v[2 * q], v[2 * q + 1], hi[q], lo[q]);
        const int dst = it_dst[u] & 0xFFFFF;
        *reinterpret_cast<uint4*>(sP + dst) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        *reinterpret_cast<uint4*>(sP + kPlane + dst) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
      }
    }
  };

  // ---- weight staging roles: thread -> out-channel row (tid % BM), taps part + WPARTS * k of the group
  const int so = tid % BM, part = tid / BM;
  const bool so_valid = t * BM + so < a.og_pad;
  const unsigned* wcol = reinterpret_cast<const unsigned*>(a.wbits + o_pad0 + (so_valid ? so : 0));
  unsigned one_bf16x2;                                   // 0x3F803F80 held in a vector register (one scalar operand per VALU op)
  asm volatile("v_mov_b32 %0, 0x3f803f80" : "=v"(one_bf16x2));
  unsigned wraw[WTAPS];
  auto issue_wloads = [&](unsigned (&dst)[WTAPS], int tg, int cc) {
    const int c0 = cc * kPC;
#pragma unroll
    for (int k = 0; k < WTAPS; ++k) {
      const int tap = min(tg + part + WPARTS * k, taps - 1);
      dst[k] = wcol[2 * ((tap * a.Gg + (c0 >> 6)) * a.opad_total) + ((c0 >> 5) & 1)];
    }
  };
  auto expand_store = [&](const unsigned (&src)[WTAPS], int tg, int cc) {
    const int sh = (cc * kPC) & 16;
#pragma unroll
    for (int k = 0; k < WTAPS; ++k) {
      const int ts = part + WPARTS * k;
      if (ts < kTapGroup && tg + ts < taps) {
        // 16 sign bits (set = +1) -> 16 bf16.  The inverted bits replicated into both 16-bit halves, one
        // packed 16-bit shift moves bit 2q / 2q+1 to the sign position of the low / high half, one
        // and-or merges it into 0x3F80 (= 1.0): two VALU ops per pair of weights.
        const unsigned m = so_valid ? ~(src[k] >> sh) & 0xFFFFu : 0xFFFFu;     // rows past the group: -1 * 0 anyway
        const u16x2 rep = __builtin_bit_cast(u16x2, m | (m << 16));
        unsigned d[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const u16x2 shq = {(unsigned short)(15 - 2 * q), (unsigned short)(14 - 2 * q)};
          const unsigned xq = __builtin_bit_cast(unsigned, rep << shq);
          asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(d[q]) : "v"(xq), "s"(0x80008000u), "v"(one_bf16x2));
        }
        unsigned char* row = sW + ts * (BM * kPRow);
        *reinterpret_cast<uint4*>(row + swz(so, 0)) = make_uint4(d[0], d[1], d[2], d[3]);
        *reinterpret_cast<uint4*>(row + swz(so, 1)) = make_uint4(d[4], d[5], d[6], d[7]);
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  auto mfma_tap = [&](int tt, const int (&baddr)[TN]) {
    Frag af[TM], bh[TN], bl[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const uint4 v = *reinterpret_cast<const uint4*>(sW + tt * (BM * kPRow) + a_addr[i]);
      af[i].u[0] = v.x; af[i].u[1] = v.y; af[i].u[2] = v.z; af[i].u[3] = v.w;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const uint4 vh = *reinterpret_cast<const uint4*>(sP + baddr[j]);
      const uint4 vl = *reinterpret_cast<const uint4*>(sP + kPlane + baddr[j]);
      bh[j].u[0] = vh.x; bh[j].u[1] = vh.y; bh[j].u[2] = vh.z; bh[j].u[3] = vh.w;
      bl[j].u[0] = vl.x; bl[j].u[1] = vl.y; bl[j].u[2] = vl.z; bl[j].u[3] = vl.w;
    }
    // hi products of all four tiles first, then lo: dependent MFMAs on one accumulator are 4 apart
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i].v, bh[j].v, acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i].v, bl[j].v, acc[i][j], 0, 0, 0);
  };

  issue_xloads(0);
  issue_wloads(wraw, 0, 0);
  __syncthreads();                                       // sPre visible
  for (int cc = 0; cc < cchunks; ++cc) {
    convert_store(cc);
    expand_store(wraw, 0, cc);
    __syncthreads();                                     // patch and first tap group of this chunk visible
    if constexpr (!MANY_TAPS) {
      if (cc + 1 < cchunks) {                            // next chunk's loads fly during the MFMAs below
        issue_xloads(cc + 1);
        issue_wloads(wraw, 0, cc + 1);
      }
#pragma unroll
      for (int tt = 0; tt < kTapGroup; ++tt)
        if (tt < taps) mfma_tap(tt, b_addr[tt]);
    } else {
      // more than 9 taps (5x5 ...): later groups are staged in place, B addresses recomputed per tap
      int kh = 0, kw = 0;
      for (int tg = 0; tg < taps; tg += kTapGroup) {
        if (tg) {
          unsigned wtmp[WTAPS];
          issue_wloads(wtmp, tg, cc);
          __syncthreads();                               // previous group's fragment reads finished
          expand_store(wtmp, tg, cc);
          __syncthreads();
        }
        for (int tt = 0; tt < kTapGroup && tg + tt < taps; ++tt) {
          const int toff = kh * a.dh * Wp + kw * a.dw;
          int baddr[TN];
#pragma unroll
          for (int j = 0; j < TN; ++j) baddr[j] = swz(e_pix[j] + toff, kh8);
          // (tt is not a compile-time constant here: the A offset becomes an address add)
          Frag af[TM], bh[TN], bl[TN];
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const uint4 v = *reinterpret_cast<const uint4*>(sW + tt * (BM * kPRow) + a_addr[i]);
            af[i].u[0] = v.x; af[i].u[1] = v.y; af[i].u[2] = v.z; af[i].u[3] = v.w;
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const uint4 vh = *reinterpret_cast<const uint4*>(sP + baddr[j]);
            const uint4 vl = *reinterpret_cast<const uint4*>(sP + kPlane + baddr[j]);
            bh[j].u[0] = vh.x; bh[j].u[1] = vh.y; bh[j].u[2] = vh.z; bh[j].u[3] = vh.w;
            bl[j].u[0] = vl.x; bl[j].u[1] = vl.y; bl[j].u[2] = vl.z; bl[j].u[3] = vl.w;
          }
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i].v, bh[j].v, acc[i][j], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i].v, bl[j].v, acc[i][j], 0, 0, 0);
          if (++kw == a.KW) { kw = 0; ++kh; }
        }
      }
      if (cc + 1 < cchunks) {
        issue_xloads(cc + 1);
        issue_wloads(wraw, 0, cc + 1);
      }
    }
    __syncthreads();                                     // every wave is done reading before LDS is rewritten
  }
  store_tiles<unsigned, BM, BN, TM, TN>(a, acc, blockIdx.x, t, o0, wm, wn, col, kh8);
}

}  // namespace
}  // namespace signw
}  // namespace lsq

using namespace lsq;
using namespace lsq::signw;

extern "C" int64_t lsq_signw_weight_bytes(const lsq_conv_geom* g, int kw_planes) {
  if (check_geom(g) || kw_planes < 1 || kw_planes > LSQ_MAX_PLANES) return 0;
  return lean_weight_bytes(g, kw_planes);
}

extern "C" int lsq_signw_prepare_weight(const uint64_t* wbits, int kw_planes, const lsq_conv_geom* g, void* wprep, void* stream) {
  if (!wbits || !wprep) return LSQ_E_NULL;
  if (int e = check_geom(g)) return e;
  if (kw_planes < 1 || kw_planes > LSQ_MAX_PLANES) return LSQ_E_SCHEME;
  return lean_prepare(wbits, kw_planes, g, wprep, (hipStream_t)stream);
}

extern "C" int lsq_signw_conv2d(const float* x, float clamp_alpha, const float* pre_scale, const float* pre_shift,
                                const uint64_t* wbits, const void* wprep, int kw_planes, const float* wscales, const float* bias,
                                const lsq_conv_geom* g, int relu, const float* act_slope, const float* res_pre,
                                const float* res_post,
                                float* y, void* stream) {
  if (!x || !wbits || !wscales || !y) return LSQ_E_NULL;
  if (int e = check_geom(g)) return e;
  if (kw_planes < 1 || kw_planes > LSQ_MAX_PLANES) return LSQ_E_SCHEME;
  const int Ho = out_h(g), Wo = out_w(g);
  if (Ho <= 0 || Wo <= 0) return LSQ_E_SHAPE;
  SwArgs a = {};
  if ((pre_scale == nullptr) != (pre_shift == nullptr)) return LSQ_E_NULL;
  a.x = x; a.bias = bias; a.y = y; a.alpha = clamp_alpha;
  a.pre_scale = pre_scale; a.pre_shift = pre_shift;
  if (relu < LSQ_ACT_NONE || relu > LSQ_ACT_PRELU_CHANNEL || (relu >= LSQ_ACT_PRELU && !act_slope)) return LSQ_E_SCHEME;
  a.relu = relu; a.slope = act_slope; a.res_pre = res_pre; a.res_post = res_post;
  a.N = g->N; a.C = g->C; a.H = g->H; a.W = g->W; a.O = g->O; a.KH = g->KH; a.KW = g->KW;
  a.sh = g->stride_h; a.sw = g->stride_w; a.ph = g->pad_h; a.pw = g->pad_w; a.dh = g->dil_h; a.dw = g->dil_w;
  a.cg = g->C / g->groups;
  a.Gg = (a.cg + 63) / 64;
  a.Ho = Ho; a.Wo = Wo;
  a.og = g->O / g->groups;
  a.og_pad = (a.og + 15) / 16 * 16;
  a.opad_total = g->groups * a.og_pad;
  const long long wplane_words = lsq_weight_plane_words(g);
  const long long total = (long long)g->N * Ho * Wo;
  hipStream_t st = (hipStream_t)stream;
  // 128 out-channels x 128 pixels per workgroup when the group is wide enough, else 64 x 128
  const bool wide = a.og > 64;
  const int bm = wide ? 128 : 64;
  a.tiles_per_group = (a.og + bm - 1) / bm;
  // Layers whose input patch fits the LDS planes take the patch kernel.  Stride 1: 128 out-channels x 128
  // pixels per workgroup, or 64 x 256 for narrow layers (the patch conversion is shared by more pixels);
  // other strides: 64 pixels per workgroup (the patch of the same pixels is stride^2 times sparser).
  const int Hp = g->H + 2 * g->pad_h, Wp = g->W + 2 * g->pad_w;
  const bool unit = g->stride_h == 1 && g->stride_w == 1;
  const int pbn = unit ? (wide ? 128 : 256) : (wide ? 64 : 128);
  const int pmax = unit ? 512 : 640, premax = unit ? 512 : 256;
  const long long row_gap = (long long)g->stride_h * Wp - (long long)Wo * g->stride_w;
  const long long img_gap = ((long long)Hp - (long long)Ho * g->stride_h) * Wp;
  const long long patch = (long long)(pbn - 1) * g->stride_w + ((pbn - 1) / Wo + 1) * (row_gap > 0 ? row_gap : 0) +
                          ((pbn - 1) / (Ho * Wo) + 1) * (img_gap > 0 ? img_gap : 0) +
                          (long long)(g->KH - 1) * g->dil_h * Wp + (long long)(g->KW - 1) * g->dil_w + 1;
  const int PLr = (int)((patch + 127) / 128 * 128);
  bool use_patch = PLr <= pmax && (!pre_scale || (a.cg + kPC - 1) / kPC * kPC <= premax) &&
                   (long long)g->N * g->C * g->H * g->W < (1ll << 30) && (long long)g->N * Hp * Wp + PLr < (1ll << 31) &&
                   (long long)g->N * g->O * Ho * Wo < (1ll << 30);
#ifdef LSQ_TUNE
  if (getenv("LSQ_SIGNW_NOPATCH")) use_patch = false;
#endif
  for (int q = 0; q < kw_planes; ++q) {
    a.wbits = (const unsigned long long*)wbits + (long long)q * wplane_words;
    a.wscale = wscales + (long long)q * g->O;
    a.accumulate = q ? 1 : 0;
    a.final_pass = q == kw_planes - 1 ? 1 : 0;
    const unsigned otiles = (unsigned)(g->groups * a.tiles_per_group);
    // 3x3 fast path on the weights expanded by lsq_signw_prepare_weight (lsq_signw_lean.hip)
    const int fast = lean_launch(a, wprep, q, g, st);
    if (fast != LSQ_E_UNSUPPORTED) {
      if (fast) return fast;
      continue;
    }
    if (use_patch) {
      dim3 grid((unsigned)((total + pbn - 1) / pbn), otiles);
      const bool many = g->KH * g->KW > kTapGroup;
#define LSQ_PATCH(BM_, BN_, WM_, WN_, TN_, PM_, PRE_)                                                              \
  do {                                                                                                             \
    if (many) hipLaunchKernelGGL((signw_conv_patch<BM_, BN_, WM_, WN_, TN_, PM_, PRE_, true>), grid, dim3(256), 0, st, a, Hp, Wp, PLr); \
    else hipLaunchKernelGGL((signw_conv_patch<BM_, BN_, WM_, WN_, TN_, PM_, PRE_, false>), grid, dim3(256), 0, st, a, Hp, Wp, PLr);     \
  } while (0)
      if (unit && wide) LSQ_PATCH(128, 128, 2, 2, 2, 512, 512);
      else if (unit) LSQ_PATCH(64, 256, 1, 4, 2, 512, 512);
      else if (wide) LSQ_PATCH(128, 64, 2, 2, 1, 640, 256);
      else LSQ_PATCH(64, 128, 1, 4, 1, 640, 256);
#undef LSQ_PATCH
    } else {
      dim3 grid((unsigned)((total + kBN - 1) / kBN), otiles);
      if (wide) hipLaunchKernelGGL((signw_conv_tiled<128, 2, 2, 2, 2>), grid, dim3(256), 0, st, a);
      else hipLaunchKernelGGL((signw_conv_tiled<64, 1, 4, 2, 1>), grid, dim3(256), 0, st, a);
    }
    if (int e = (int)hipGetLastError()) return e;
  }
  return LSQ_OK;
}
