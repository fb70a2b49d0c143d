// Activation quantization for gfx950: clamp -> per-row optimal-scale solve -> packed sign planes.  This file holds
// the entry point lsq_act_quant, the single-sweep schemes (ls-1, gf-k, forced scales) and the STREAMING solver;
// the ls-2 / ls-T solve of the reference's skip = 3 sub-sample -- the hot configuration -- runs as one launch of
// aq_fused_kernel (lsq_act_fused.hip) and comes here only when that kernel does not take the shape (row length not
// a multiple of 4, fewer than 4 pixels, other skips, more than 2^22 keys) or under lsq_debug_force_streaming.
//
// Streaming path: one 1024-thread workgroup owns one row (one sample, quantization.py:77) in every kernel of the
// sequence (histogram sweep -> solve -> one sweep per further plane); rows are independent, so a batch of N rows is
// a grid of N workgroups.  The kernels are separate so that each gets its own register allocation.
//
// The optimal-v1 solve (quant/binary/optimal.py:41-155) never sorts.  It is a radix select over the IEEE
// bit pattern of |x|: a 13-bit level-1 histogram over the whole sub-sample, then, for the few bins that can
// hold a candidate, an 8-bit level per wave (block-level fall-back: 10 + 8 bits).  Each histogram bin
// accumulates, with ONE 64-bit LDS atomic per element, its count and the exact integer sum of the low key
// bits; within a bin the exponent is fixed, so value = 2^e * mantissa is linear in those bits and the
// bin's sum, hence rank and prefix sum at every bin boundary, is exact.  m1(i), m2(i) of optimal.py:66-74
// are monotone in the sorted position i, so only bins whose value range can intersect them are refined
// (typically 10-25 of 8192); their keys are gathered once into LDS.  Candidates are tested exactly (fp64)
// and their least-squares cost is evaluated in closed form from the prefix sums, replacing the reference's
// [N,K,M] broadcast (optimal.py:31-38).  The arithmetic shared with the fused kernel is in lsq_solver_math.h.
//
// Memory traffic per row of M floats (streaming path): the histogram sweep reads M (plane 0 + level-1 histogram),
// the solve's gather reads M again (the sub-sample touches every line at skip = 3), the plane-1 sweep reads M
// (plane 1 + v2).  Writes are M/8 bytes per plane.  The fused kernel reads M twice.

#include <type_traits>

#include <atomic>

#include "lsq_common.h"
#include "lsq_solver_math.h"
#include "lsq_act_fused.h"

namespace lsq {
#ifdef LSQ_PHASE_CLOCKS
__device__ long long g_phase_clocks[32];
__device__ long long g_block_times[1024][2];    // solve kernel: constant-rate clock at entry / exit of each workgroup
__device__ long long g_sweep_times[2][1024][2]; // the same for the histogram sweep [0] and the last plane sweep [1]
__device__ long long g_wave_stats[16][4];       // block 0: cycles in the wave path, slots, flagged sub-bins, ranked keys
#define LSQ_MARK(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_phase_clocks[i] = (long long)clock64(); } while (0)
#define LSQ_NOTE(i, v) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_phase_clocks[i] = (long long)(v); } while (0)
#define LSQ_WSTAT(k, v) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) g_wave_stats[threadIdx.x >> 6][k] += (long long)(v); } while (0)
#define LSQ_WSTAT0() do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) for (int k_ = 0; k_ < 4; ++k_) g_wave_stats[threadIdx.x >> 6][k_] = 0; } while (0)
#else
#define LSQ_MARK(i) do {} while (0)
#define LSQ_NOTE(i, v) do {} while (0)
#define LSQ_WSTAT(k, v) do {} while (0)
#define LSQ_WSTAT0() do {} while (0)
#endif
namespace {

constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / kWave;
constexpr int kBinsPerThread = L1_BINS / kThreads;
constexpr int L2_SHIFT = 8, L2_BINS = 1024;    // block path: key bits [17:8]
constexpr int L3_BINS = 256;                   // block path: key bits [7:0]
constexpr int kKeysPerLane = L3_BINS / kWave;
constexpr int kSlotCap = 256;                  // flagged level-1 bins held as slot records
constexpr int kSubSlots = 126;                 // slots gathered per sub-round (role table bytes)
constexpr int kSeg3 = 8;                       // level-2 bins refined per level-3 pass
constexpr int kListExt = 22528;                // gathered keys held in LDS (88 KB)
constexpr int kWaveSub = 256;                  // wave-level refinement: at most 256 sub-bins (key bits [17:10])
constexpr int kSmallSeg = 1024;                // segments up to this many keys use 64 sub-bins (<= 64 keys each, typically 16)

struct Args {
  const float* x;
  long long row_elems;      // M = C*H*W (or M of the flat matrix)
  int C, H, W, cg, Gg, Gt, Hp, Wp, pad_h, pad_w;
  int scheme, k, skip;
  float alpha;
  const float* forced;      // [k][N] or null
  const float* pre_scale;   // [C] or null: x' = x * pre_scale[c] + pre_shift[c] before the clamp (folded eval BN)
  const float* pre_shift;
  unsigned long long* planes;
  long long plane_words;    // words of one plane (all rows)
  long long row_words;      // words of one row of one plane
  float* scales;            // [k][N]
  int* status;              // flat mode: number of candidates per row
  int* trace;               // test hook (lsq_debug_solver_trace): chosen sorted position per row, or null
  int N;
  int flat;                 // 1: dense [R][M] rows, no planes
  int ternary;
  int write_scale;          // sweep kernels: store mean|residual| as scale q
  int csplit_log2;          // sweep kernels: 2^csplit_log2 lanes share one (64 channels x VEC pixels) item
  int parts;                // sweep kernels without histogram: workgroups that share one row (grid = parts x N)
  double qmagic;            // != 0: exact row sums -- every 8-channel group sum is rounded to a multiple of 2^e through
                            // (g + qmagic) - qmagic, qmagic = 1.5 * 2^(52 + e), and fp64 adds such multiples exactly
  unsigned char* ws;        // workspace (kWsRow bytes per row)
  unsigned char* rws;       // row-split sweeps: kSplitRow bytes per row (partial sums + (epoch, arrivals) slot); any content
  unsigned epoch;           // row-split sweeps: tag of this sweep launch (never 0), see the arrival slot
};
constexpr int kMaxParts = 8;
constexpr int kSplitRow = 8 * (kMaxParts + 1);           // bytes: kMaxParts fp64 partial sums + a 64-bit slot for the counter

// one flagged sub-bin of a slot, queued by the wave that scanned the slot and resolved by whichever wave is
// free: the 16 waves then share the expensive part instead of one wave walking all sub-bins of its slot
struct SubTask {
  unsigned short slot, bits;     // slot index; width of the sub-bin index (6 or 8, +6 per refinement, <= 18)
  unsigned sub;                  // sub-bin index inside the level-1 bin (`bits` wide)
  unsigned cc, rs, ns;           // keys in the sub-bin, sorted position in front of it, next non-empty sub-bin at this
                                 // level (kNoKey: none -- the successor key is succ_key, or the slot's when that is kNoKey too)
  unsigned succ_key;
  double ps;                     // prefix sum in front of it
};
constexpr int kTaskCap = (L1_BINS * 2) / (int)sizeof(SubTask);   // what the role table's bytes hold (682)

struct Seg3 {
  unsigned pref, next_pref, cnt, r0;
  double p0;
};

struct BlockHists {
  unsigned long long hist2[L2_BINS];
  unsigned hist3[kSeg3][L3_BINS];
};

struct SolverLds {
  // gathered keys of the flagged bins; rows with more than kSlotCap flagged bins never use the list and
  // re-scan the level-1 histogram instead, which then lives in the same bytes
  union {
    unsigned list[kListExt];
    struct {
      unsigned long long hist1[L1_BINS];
      unsigned short nzlist1[L1_BINS];
    } big;
  } k;
  union {
    BlockHists blk;                                    // block-level (slow) refinement
    unsigned long long whist[kWaves][kWaveSub];        // wave-level (fast) refinement
  } u;
  unsigned wkeys[kWaves][kWave];
  unsigned short nzlist[L2_BINS];                      // block path, level 2
  union {
    unsigned short role[L1_BINS];                      // gather: low byte 1 + gather slot, high byte 1 + successor slot
    SubTask task[kTaskCap];                            // wave path, after the gather
  };
  Slot1 slot[kSlotCap];
  unsigned fill[kSubSlots];
  unsigned short sub_begin[kSlotCap + 2];
  unsigned short slow[kSubSlots];
  Seg3 seg[kSeg3];
  unsigned succ3[kSeg3];
  // block-scan scratch
  unsigned wa[kWaves], wb[kWaves], wc[kWaves];
  double ws[kWaves];
  Best wbest[kWaves];
  // scalars
  unsigned n_sub, n_cand, n_slow, blk_succ, best_order;
  unsigned task_cnt[3];                           // tasks queued in the current / next / after-next round
  unsigned dbg_slow, dbg_gathered, dbg_rowpass;   // diagnostics written back to the row header
  unsigned rg_lo[4], rg_len[4], rg_first[4], rg_succbin[4], rg_last[4], n_rg;   // runs of consecutive flagged bins
  unsigned minkey;
  double total;
  float sv[LSQ_MAX_PLANES];
  Args args;                  // one copy per workgroup: non-inlined phases read it from LDS
};

struct SmallLds {
  double ws[kWaves];
  float sv[LSQ_MAX_PLANES];
  Args args;
};

// first sweep of a solver scheme: level-1 histogram, its scan, and the slot records handed to the
// solve kernel through the workspace
constexpr int kNzCap = 2048;                   // non-empty level-1 bins the balanced scan handles (2 per thread)
struct SweepLds {
  unsigned long long hist1[L1_BINS];
  unsigned short nzlist[L1_BINS];
  unsigned nz_r0[kNzCap];                       // sorted position / prefix sum in front of the i-th non-empty bin
  double nz_p0[kNzCap];
  Slot1 slot[kSlotCap];
  unsigned wa[kWaves], wb[kWaves], wc[kWaves];
  double ws[kWaves];
  double total;
  float sv[LSQ_MAX_PLANES];
  Args args;
};

// per-row record in the caller-provided workspace (sweep kernel -> solve kernel)
struct RowHeader {
  unsigned tflag, minkey, n, pad;
  double total, pad2;
};
constexpr long long kWsSlots = sizeof(RowHeader);
constexpr long long kWsHist = kWsSlots + (long long)kSlotCap * (long long)sizeof(Slot1);
constexpr long long kWsRow = (kWsHist + (long long)L1_BINS * 8 + 63) / 64 * 64;


// exclusive block scan of (a, b, s); returns block totals. All 1024 threads must call.
template <class L>
__device__ void block_excl_scan(unsigned& a, unsigned& b, double& s, unsigned& ta, unsigned& tb, double& ts,
                                L* lds) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const unsigned ia = wave_incl_scan(a), ib = wave_incl_scan(b);
  const double is = wave_incl_scan(s);
  __syncthreads();
  if (lane == 63) {
    lds->wa[wid] = ia;
    lds->wb[wid] = ib;
    lds->ws[wid] = is;
  }
  __syncthreads();
  unsigned oa = 0, ob = 0;
  double os = 0.0;
  ta = tb = 0;
  ts = 0.0;
  for (int w = 0; w < kWaves; ++w) {
    if (w == wid) {
      oa = ta;
      ob = tb;
      os = ts;
    }
    ta += lds->wa[w];
    tb += lds->wb[w];
    ts += lds->ws[w];
  }
  a = oa + ia - a;
  b = ob + ib - b;
  s = os + (is - s);
}

template <class L>
__device__ double block_sum(double v, L* lds) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds->ws[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < kWaves; ++w) t += lds->ws[w];
  return t;
}

__device__ __forceinline__ unsigned mod_small(unsigned r, unsigned m) {
  // r < m + 4
  if (r >= m) r -= m;
  if (r >= m) r -= m;
  if (r >= m) r -= m;
  if (r >= m) r -= m;
  return r;
}

// ---------------------------------------------------------------------------------------------
// One streaming pass over the row: plane q bits, sum |residual_q|, optional level-1 histogram.
// result chain / residual chain exactly as quantization.py:89-92, :112-115, :137-146.
struct Chain {
  int q;
  float v[LSQ_MAX_PLANES];   // wave-uniform (SGPRs)
};

// QM = 0: plane 0 (no scale yet); QM = 1: plane 1 (one scale); QM = 2: any depth
template <int QM>
__device__ __forceinline__ void chain_eval(const Chain& ch, float xv, bool& bit, float& absres) {
  if constexpr (QM == 0) {
    bit = xv >= 0.f;
    absres = fabsf(xv);
  } else if constexpr (QM == 1) {
    const float v0 = ch.v[0];
    const float r = xv - ((xv >= 0.f) ? v0 : -v0);      // x - v1*b1: one rounding, as the reference
    bit = r >= 0.f;
    absres = fabsf(r);
  } else {
    float result = 0.f, res = xv;
#pragma unroll
    for (int i = 0; i < LSQ_MAX_PLANES - 1; ++i) {
      if (i < ch.q) {
        const float vi = ch.v[i];
        result = result + ((xv - result >= 0.f) ? vi : -vi);
        res = res - ((res >= 0.f) ? vi : -vi);
      }
    }
    bit = (xv - result) >= 0.f;
    absres = fabsf(res);
  }
}

struct PassOut {
  double sum;
  unsigned minkey;
};

template <class L>
__device__ __forceinline__ Chain load_chain(const L* lds, int q) {
  Chain ch;
  ch.q = q;
#pragma unroll
  for (int i = 0; i < LSQ_MAX_PLANES; ++i)
    ch.v[i] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(lds->sv[i])));
  return ch;
}

// Which of VEC consecutive elements (flat index = base, base+1, ...) belong to the sub-sample
// flat % skip == 0 (optimal.py:134)?  `rem` = base % skip.  For the reference's skip = 3 the answer is
// a select, not a per-element modulo + branch: exactly one of the first three elements is a hit.
template <int VEC, class F>
__device__ __forceinline__ void emit_subsample(const float (&x)[VEC], unsigned rem, unsigned skip, F emit) {
  if (skip == 3u) {
    if constexpr (VEC == 4) {
      emit(rem == 0u ? x[0] : (rem == 1u ? x[2 % VEC] : x[1 % VEC]));
      if (rem == 0u) emit(x[3 % VEC]);
    } else if constexpr (VEC == 2) {
      if (rem != 1u) emit(rem == 0u ? x[0] : x[1 % VEC]);
    } else {
      if (rem == 0u) emit(x[0]);
    }
  } else {
#pragma unroll
    for (int v = 0; v < VEC; ++v)
      if (mod_small(rem + v, skip) == 0u) emit(x[v]);
  }
}

// The lane = pixel(s), loop over 64 channels sweep shared by the pack passes and the gather pass.
// body(cc, x[VEC], rem) is called for every channel of the item with the clamped values.
// SPLIT (round 3): a row with fewer items than threads (small images: 32 x 32 and below) used to leave most of the
// workgroup idle while every busy lane walked its 64 channels in a chain of dependent load batches -- 20 us per
// launch on CIFAR-sized rows, latency, not bandwidth.  With SPLIT 2^csplit_log2 consecutive lanes share an item:
// each walks `cper` = 64 >> csplit_log2 channels from its own base pointer (uniform loop bounds and bit positions,
// exactly the code of the unsplit case), and `end` shifts the lane's partial word to its place and ORs the lanes'
// words together (they sit in one wave and leave the loop together).
template <int VEC, bool NEED_REM, bool SPLIT, class Begin, class Body, class End>
__device__ __forceinline__ void sweep_row(const Args& a, const float* __restrict__ xrow, Begin begin, Body body, End end) {
  const int HW = a.H * a.W;
  const int PV = (HW + VEC - 1) / VEC;
  const int items = a.Gt * PV;
  const unsigned skip = (unsigned)a.skip;
  const unsigned step = (unsigned)(HW % a.skip);
  const int csl = SPLIT ? a.csplit_log2 : 0, cs = 1 << csl;
  const int part = SPLIT ? (int)(threadIdx.x & (cs - 1)) : 0;
  const int cper = 64 >> csl;
  // (row-split sweeps: the row's a.parts workgroups form one pool of parts * kThreads lanes; the lanes that share an item
  //  are consecutive lanes of one wave either way)
  const int wparts = a.parts > 1 ? a.parts : 1, wpart = a.parts > 1 ? (int)blockIdx.x : 0;
  for (int item = (wpart * kThreads + (int)threadIdx.x) >> csl; item < items; item += (wparts * kThreads) >> csl) {
    const int j = item / PV;
    const int p = (item - j * PV) * VEC;
    const int grp = j / a.Gg;
    const int jj = j - grp * a.Gg;
    const int cfirst = part * cper;                         // the lane's first channel inside the 64-channel word
    const int c0 = grp * a.cg + jj * 64 + cfirst;
    const int nch = max(0, min(cper, a.cg - jj * 64 - cfirst));
    const float* src = xrow + (long long)c0 * HW + p;
    unsigned rem = NEED_REM ? (unsigned)(((long long)c0 * HW + p) % (long long)skip) : 0u;
    begin();
    // U independent loads are issued before any of them is consumed: LDS atomics in the body
    // would otherwise pin every load to its use and leave one request in flight per lane.
    constexpr int U = 32 / VEC;
    const bool affine = a.pre_scale != nullptr;
    for (int cb = 0; cb < nch; cb += U) {
      float vals[U][VEC];
      float scs[U], shs[U];          // folded batch norm of the U channels, requested with the data
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int cc = min(cb + u, nch - 1);
        if (VEC == 4) {
          const float4 t = *reinterpret_cast<const float4*>(src + (long long)cc * HW);
          vals[u][0] = t.x; vals[u][1 % VEC] = t.y; vals[u][2 % VEC] = t.z; vals[u][3 % VEC] = t.w;
        } else if (VEC == 2) {
          const float2 t = *reinterpret_cast<const float2*>(src + (long long)cc * HW);
          vals[u][0] = t.x; vals[u][1 % VEC] = t.y;
        } else {
          vals[u][0] = src[(long long)cc * HW];
        }
        scs[u] = affine ? a.pre_scale[c0 + cc] : 1.f;
        shs[u] = affine ? a.pre_shift[c0 + cc] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int cc = cb + u;
        if (cc < nch) {
          float x[VEC];
          if (affine) {                 // one fma per element
#pragma unroll
            for (int v = 0; v < VEC; ++v) x[v] = clamp_sym(fmaf(vals[u][v], scs[u], shs[u]), a.alpha);
          } else {
#pragma unroll
            for (int v = 0; v < VEC; ++v) x[v] = clamp_sym(vals[u][v], a.alpha);
          }
          body(cc, x, rem);             // (cc: bit position relative to the lane's first channel)
          if constexpr (NEED_REM) {
            rem += step;
            if (rem >= skip) rem -= skip;
          }
        }
      }
    }
    end(j, p, cfirst, cs);
  }
}

// EXACT (plain sweeps under a clamp): the row sum is the same NUMBER however the row's elements are dealt to lanes and
// workgroups -- whatever the batch size, lane sharing or row split.  The fp32 sum of every group of 8 channels (fixed
// membership and order) is rounded to a multiple of 2^e, e = ceil(log2 alpha) - 31 (|x| <= alpha: a relative step of
// 2^-32 of a typical group sum), and multiples of 2^e below 2^(53 + e) -- 4 million elements at the clamp -- add without
// rounding in fp64.
template <int VEC, bool HIST, int QM, bool SPLIT, bool EXACT, class L>
__device__ __forceinline__ PassOut pack_pass(const Args& a, L* lds, const float* __restrict__ xrow, unsigned long long* __restrict__ prow, int q) {
  // (a: the kernel's own argument block -- scalar registers; a copy staged through LDS cost a microsecond per launch)
  const Chain ch = load_chain(lds, q);
  const unsigned skip = (unsigned)a.skip;
  double acc = 0.0;
  unsigned mk = kNoKey;
  unsigned long long word[VEC];
  float facc[VEC];            // fp32 over one 64-channel column, folded into fp64 per item
  sweep_row<VEC, HIST, SPLIT>(
      a, xrow,
      [&]() {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          word[v] = 0ull;
          facc[v] = 0.f;
        }
      },
      [&](int cc, const float (&x)[VEC], unsigned rem) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          bool bit;
          float ar;
          chain_eval<QM>(ch, x[v], bit, ar);
          word[v] |= (unsigned long long)bit << cc;
          facc[v] += ar;
          if constexpr (EXACT) {
            if ((cc & 7) == 7) {                       // (cc counts from the lane's first channel, a multiple of 8)
              acc += ((double)facc[v] + a.qmagic) - a.qmagic;
              facc[v] = 0.f;
            }
          }
        }
        if constexpr (HIST) {
          emit_subsample<VEC>(x, rem, skip, [&](float xs) {
            const unsigned key = abs_key(xs);
            atomicAdd(&lds->hist1[key >> L1_SHIFT], kOne | (unsigned long long)(key & ((1u << L1_SHIFT) - 1u)));
            mk = min(mk, key);
          });
        }
      },
      [&](int j, int p, int cfirst, int cs) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          if constexpr (EXACT) acc += ((double)facc[v] + a.qmagic) - a.qmagic;    // (a last group of fewer than 8 channels; else 0)
          else acc += (double)facc[v];
          unsigned long long wd = word[v];
          if constexpr (SPLIT) {
            wd <<= cfirst;                             // the lane's channels start at bit cfirst of the word
            for (int d = 1; d < cs; d <<= 1)          // the lanes that share the item each hold some of its channels' bits
              wd |= ((unsigned long long)(unsigned)__shfl_xor((int)(wd >> 32), d) << 32) | (unsigned)__shfl_xor((int)(unsigned)wd, d);
          }
          const int pix = p + v;
          const int h = pix / a.W;
          const int w = pix - h * a.W;
          if (!SPLIT || cfirst == 0) prow[((long long)j * a.Hp + h + a.pad_h) * a.Wp + w + a.pad_w] = wd;
        }
      });
  PassOut out;
  out.sum = acc;
  out.minkey = mk;
  return out;
}

// flat rows (no planes): sum |residual| and optional histogram, coalesced
template <bool HIST, int QM, class L>
__device__ __forceinline__ PassOut flat_pass(const Args& a, L* lds, const float* __restrict__ xrow, int q) {
  const Chain ch = load_chain(lds, q);
  const long long M = a.row_elems;
  const unsigned skip = (unsigned)a.skip;
  double acc = 0.0;
  unsigned mk = kNoKey;
  for (long long i = threadIdx.x; i < M; i += kThreads) {
    const float xv = clamp_sym(xrow[i], a.alpha);
    bool bit;
    float ar;
    chain_eval<QM>(ch, xv, bit, ar);
    acc += (double)ar;
    if constexpr (HIST) {
      if ((i % skip) == 0) {
        const unsigned key = abs_key(xv);
        atomicAdd(&lds->hist1[key >> L1_SHIFT], kOne | (unsigned long long)(key & ((1u << L1_SHIFT) - 1u)));
        mk = min(mk, key);
      }
    }
  }
  PassOut out;
  out.sum = acc;
  out.minkey = mk;
  return out;
}

// sub-sampled keys of the row (optimal.py:134), straight from memory (L2 / Infinity Cache)
template <class F>
__device__ __forceinline__ void for_each_row_key(const Args& a, const float* __restrict__ xrow, unsigned n, F f) {
  constexpr int U = 8;
  const long long skip = a.skip;
  for (unsigned j0 = threadIdx.x; j0 < n; j0 += kThreads * U) {
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned j = min(j0 + (unsigned)u * kThreads, n - 1u);
      v[u] = xrow[(long long)j * skip];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (j0 + (unsigned)u * kThreads < n) {
        float xv = v[u];
        if (a.pre_scale) {
          const long long flat = (long long)(j0 + (unsigned)u * kThreads) * skip;
          const int c = (int)(flat / ((long long)a.H * a.W));
          xv = fmaf(xv, a.pre_scale[c], a.pre_shift[c]);
        }
        f(abs_key(clamp_sym(xv, a.alpha)));
      }
  }
}

// ---------------------------------------------------------------------------------------------
// The solve.  On entry hist1 holds the level-1 histogram of the n sub-sampled keys.  The phases are
// separate non-inlined functions on purpose: each gets its own register allocation, so the rarely
// used block-level path cannot push the streaming sweeps of the kernel into scratch.
// level 1: scan the 8192-bin histogram (8 bins per thread), flag bins that may hold a candidate, and
// write a slot record for the flagged bins with ordinal in [round0, round0 + kSlotCap)
template <class L>
__device__ __forceinline__ unsigned l1_scan(L* lds, const unsigned long long* hist1, unsigned short* nzl, unsigned n,
                                        unsigned round0) {
  const float* xrow = nullptr;
  const Args& a = lds->args;
  const unsigned* list = nullptr;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const bool ternary = a.ternary != 0;
  double total = 0.0;
  (void)a; (void)xrow; (void)n; (void)list; (void)tid; (void)lane; (void)wid; (void)ternary;
  unsigned cnt[kBinsPerThread];
  double sum[kBinsPerThread];
  unsigned my_nz = 0, my_cnt = 0;
  double my_sum = 0.0;
#pragma unroll
  for (int u = 0; u < kBinsPerThread; ++u) {
    const unsigned b = (unsigned)kBinsPerThread * tid + u;
    const unsigned long long h = hist1[b];
    cnt[u] = (unsigned)(h >> 42);
    sum[u] = cnt[u] ? bin_sum_exact(b << L1_SHIFT, cnt[u], h & kLowMask) : 0.0;
    my_nz += cnt[u] ? 1u : 0u;
    my_cnt += cnt[u];
    my_sum += sum[u];
  }
  unsigned enz = my_nz, ecnt = my_cnt, tnz, tcnt;
  double esum = my_sum;
  block_excl_scan(enz, ecnt, esum, tnz, tcnt, total, lds);
  {
    unsigned z = enz;
#pragma unroll
    for (int u = 0; u < kBinsPerThread; ++u)
      if (cnt[u]) nzl[z++] = (unsigned short)((unsigned)kBinsPerThread * tid + u);
  }
  if constexpr (std::is_same<L, SweepLds>::value) {
    // The non-empty bins sit in a few binades, i.e. in the 8-bin ranges of a few dozen threads, and the
    // candidate test is ~100 fp64 instructions: run it over the COMPACT list instead, two non-empty bins
    // per thread, with the prefixes handed over through LDS.
    if (tnz <= (unsigned)kNzCap && round0 == 0u) {
      {
        unsigned z = enz, c = ecnt;
        double sacc = esum;
#pragma unroll
        for (int u = 0; u < kBinsPerThread; ++u) {
          if (cnt[u]) {
            lds->nz_r0[z] = c;
            lds->nz_p0[z] = sacc;
            ++z;
          }
          c += cnt[u];
          sacc += sum[u];
        }
      }
      __syncthreads();
      unsigned bz[2], cz[2], rz[2];
      unsigned short nz2[2];
      double sz[2], pz[2];
      bool fz[2];
      unsigned nflag = 0;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const unsigned z = 2u * (unsigned)tid + (unsigned)e;
        fz[e] = false;
        bz[e] = cz[e] = rz[e] = 0;
        nz2[e] = 0xFFFFu;
        sz[e] = pz[e] = 0.0;
        if (z < tnz) {
          bz[e] = nzl[z];
          const unsigned long long h = hist1[bz[e]];
          cz[e] = (unsigned)(h >> 42);
          sz[e] = bin_sum_exact(bz[e] << L1_SHIFT, cz[e], h & kLowMask);
          rz[e] = lds->nz_r0[z];
          pz[e] = lds->nz_p0[z];
          const double vlo = (double)key_value(bz[e] << L1_SHIFT);
          const double vhi = (double)key_value((bz[e] << L1_SHIFT) | ((1u << L1_SHIFT) - 1u));
          double next_hi = vhi;
          if (z + 1 < tnz) {
            nz2[e] = nzl[z + 1];
            next_hi = (double)key_value(((unsigned)nz2[e] << L1_SHIFT) | ((1u << L1_SHIFT) - 1u));
          }
          fz[e] = n >= 3u && may_hold_candidate(rz[e], cz[e], pz[e], sz[e], vlo, vhi, next_hi, n, total, ternary);
          nflag += fz[e] ? 1u : 0u;
        }
      }
      unsigned eflag = nflag, dummy = 0, tflag, tdummy;
      double dzero = 0.0, tdz;
      block_excl_scan(eflag, dummy, dzero, tflag, tdummy, tdz, lds);
      unsigned ord = eflag;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if (fz[e]) {
          if (ord < (unsigned)kSlotCap) {
            Slot1 sl;
            sl.bin = (unsigned short)bz[e];
            sl.next_bin = nz2[e];
            sl.cnt = cz[e];
            sl.r0 = rz[e];
            sl.succ = kNoKey;
            sl.base = 0;
            sl.pad = 0;
            sl.p0 = pz[e];
            sl.sum = sz[e];
            lds->slot[ord] = sl;
          }
          ++ord;
        }
      }
      if (tid == 0) lds->total = total;
      __syncthreads();
      return tflag;
    }
  }
  __syncthreads();
  unsigned my_flags = 0;
  bool flag[kBinsPerThread];
  unsigned short nxt[kBinsPerThread];
  {
    unsigned z = enz, c = ecnt;
    double s = esum;
#pragma unroll
    for (int u = 0; u < kBinsPerThread; ++u) {
      flag[u] = false;
      nxt[u] = 0xFFFFu;
      if (cnt[u]) {
        const unsigned b = (unsigned)kBinsPerThread * tid + u;
        const double vlo = (double)key_value(b << L1_SHIFT);
        const double vhi = (double)key_value((b << L1_SHIFT) | ((1u << L1_SHIFT) - 1u));
        double next_hi = vhi;
        if (z + 1 < tnz) {
          nxt[u] = nzl[z + 1];
          next_hi = (double)key_value(((unsigned)nxt[u] << L1_SHIFT) | ((1u << L1_SHIFT) - 1u));
        }
        flag[u] = n >= 3u && may_hold_candidate(c, cnt[u], s, sum[u], vlo, vhi, next_hi, n, total, ternary);
        my_flags += flag[u] ? 1u : 0u;
        ++z;
      }
      c += cnt[u];
      s += sum[u];
    }
  }
  unsigned eflag = my_flags, dummy = 0, tflag, tdummy;
  double dzero = 0.0, tdz;
  block_excl_scan(eflag, dummy, dzero, tflag, tdummy, tdz, lds);
  {
    unsigned ord = eflag, c = ecnt;
    double s = esum;
#pragma unroll
    for (int u = 0; u < kBinsPerThread; ++u) {
      if (flag[u]) {
        if (ord >= round0 && ord < round0 + (unsigned)kSlotCap) {
          Slot1 sl;
          sl.bin = (unsigned short)((unsigned)kBinsPerThread * tid + u);
          sl.next_bin = nxt[u];
          sl.cnt = cnt[u];
          sl.r0 = c;
          sl.succ = kNoKey;
          sl.base = 0;
          sl.pad = 0;
          sl.p0 = s;
          sl.sum = sum[u];
          lds->slot[ord - round0] = sl;
        }
        ++ord;
      }
      c += cnt[u];
      s += sum[u];
    }
  }
  if (tid == 0) lds->total = total;
  __syncthreads();
  return tflag;
}

// block-level refinement of one slot: levels 2 (10 bits) and 3 (9 bits) as histograms.  Fully general
// (any count, any ties) but costs ~15 workgroup barriers, so it only serves what the wave-level path
// gives up on.  from_row: histogram straight from the row (bins too large for the LDS list);
// otherwise from the slot's segment of the list.
__device__ __noinline__ Best resolve_slot_block(SolverLds* lds, const float* __restrict__ xrow, unsigned n, unsigned si,
                                               bool from_row, Best best) {
  const Args& a = lds->args;
  unsigned* const list = lds->k.list;          // kListExt keys (runs on into hist1)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const bool ternary = a.ternary != 0;
  const double total = lds->total;
  (void)a; (void)xrow; (void)n; (void)list; (void)tid; (void)lane; (void)wid; (void)ternary;
  const Slot1 s1 = lds->slot[si];
  const unsigned s1_bin = s1.bin;
  const unsigned s1_next = s1.next_bin == 0xFFFFu ? kNoKey : (unsigned)s1.next_bin;
  __syncthreads();
  for (int i = tid; i < L2_BINS; i += kThreads) lds->u.blk.hist2[i] = 0ull;
  if (tid == 0) lds->blk_succ = s1.succ;
  __syncthreads();
  auto l2 = [&](unsigned key) {
    const unsigned b = key >> L1_SHIFT;
    if (b == s1_bin)
      atomicAdd(&lds->u.blk.hist2[(key >> L2_SHIFT) & (L2_BINS - 1)],
                kOne | (unsigned long long)(key & ((1u << L2_SHIFT) - 1u)));
    else if (from_row && b == s1_next && key < lds->blk_succ)
      atomicMin(&lds->blk_succ, key);
  };
  if (from_row) {
    for_each_row_key(a, xrow, n, l2);
  } else {
    for (unsigned i = tid; i < s1.cnt; i += kThreads) l2(list[s1.base + i]);
  }
  __syncthreads();
  const unsigned succ_b = lds->blk_succ;
  // thread t owns sub-bin t
  const unsigned long long h2 = lds->u.blk.hist2[tid];
  const unsigned c2 = (unsigned)(h2 >> 42);
  const unsigned hi_key2 = (s1_bin << L1_SHIFT) | ((unsigned)tid << L2_SHIFT);
  const double s2 = c2 ? bin_sum_exact(hi_key2, c2, h2 & kLowMask) : 0.0;
  unsigned enz2 = c2 ? 1u : 0u, ec2 = c2, tnz2, tc2;
  double es2 = s2, ts2;
  block_excl_scan(enz2, ec2, es2, tnz2, tc2, ts2, lds);
  if (c2) lds->nzlist[enz2] = (unsigned short)tid;   // level-1 use of nzlist is over
  __syncthreads();
  const unsigned r02 = s1.r0 + ec2;
  const double p02 = s1.p0 + es2;
  unsigned next_sub = kNoKey;
  bool f2 = false;
  if (c2) {
    const double vlo = (double)key_value(hi_key2);
    const double vhi = (double)key_value(hi_key2 | ((1u << L2_SHIFT) - 1u));
    double next_hi = vhi;
    if (enz2 + 1 < tnz2) {
      next_sub = lds->nzlist[enz2 + 1];
      next_hi = (double)key_value((s1_bin << L1_SHIFT) | (next_sub << L2_SHIFT) | ((1u << L2_SHIFT) - 1u));
    } else if (succ_b != kNoKey) {
      next_hi = (double)key_value(succ_b);
    }
    f2 = may_hold_candidate(r02, c2, p02, s2, vlo, vhi, next_hi, n, total, ternary);
  }
  unsigned ef2 = f2 ? 1u : 0u, d2 = 0, tf2, td2;
  double dz2 = 0.0, tdz2;
  block_excl_scan(ef2, d2, dz2, tf2, td2, tdz2, lds);

  // ---- level 3 in batches of kSeg3 flagged sub-bins
  for (unsigned b3 = 0; b3 < tf2; b3 += kSeg3) {
    const unsigned nseg = min((unsigned)kSeg3, tf2 - b3);
    __syncthreads();
    if (f2 && ef2 >= b3 && ef2 < b3 + nseg) {
      Seg3 g;
      g.pref = (s1_bin << 10) | (unsigned)tid;
      g.next_pref = next_sub != kNoKey ? ((s1_bin << 10) | next_sub) : kNoKey;
      g.cnt = c2;
      g.r0 = r02;
      g.p0 = p02;
      lds->seg[ef2 - b3] = g;
      lds->succ3[ef2 - b3] = next_sub != kNoKey ? kNoKey : succ_b;
    }
    for (int i = tid; i < kSeg3 * L3_BINS; i += kThreads) (&lds->u.blk.hist3[0][0])[i] = 0u;
    __syncthreads();
    auto l3 = [&](unsigned key) {
      const unsigned p = key >> L2_SHIFT;
      for (unsigned j = 0; j < nseg; ++j) {
        if (p == lds->seg[j].pref)
          atomicAdd(&lds->u.blk.hist3[j][key & (L3_BINS - 1)], 1u);
        else if (p == lds->seg[j].next_pref && key < lds->succ3[j])
          atomicMin(&lds->succ3[j], key);
      }
    };
    if (from_row) {
      for_each_row_key(a, xrow, n, l3);
    } else {
      for (unsigned i = tid; i < s1.cnt; i += kThreads) l3(list[s1.base + i]);
    }
    __syncthreads();
    // wave j resolves segment j: lane owns 8 consecutive keys
    if ((unsigned)wid < nseg) {
      const Seg3 g = lds->seg[wid];
      const unsigned succ_s = lds->succ3[wid];
      unsigned kc[kKeysPerLane];
      unsigned lane_cnt = 0;
      double lane_sum = 0.0;
      unsigned first_key = kNoKey;
#pragma unroll
      for (int u = kKeysPerLane - 1; u >= 0; --u) {
        kc[u] = lds->u.blk.hist3[wid][lane * kKeysPerLane + u];
        if (kc[u]) first_key = (g.pref << L2_SHIFT) | (unsigned)(lane * kKeysPerLane + u);
      }
#pragma unroll
      for (int u = 0; u < kKeysPerLane; ++u) {
        lane_cnt += kc[u];
        lane_sum += (double)kc[u] * (double)key_value((g.pref << L2_SHIFT) | (unsigned)(lane * kKeysPerLane + u));
      }
      const unsigned ic = wave_incl_scan(lane_cnt);
      const double is = wave_incl_scan(lane_sum);
      unsigned after = kNoKey;          // next non-empty key in a higher lane
      {
        unsigned sfx = first_key;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const unsigned o = __shfl_down(sfx, d);
          if (lane + d < 64) sfx = min(sfx, o);
        }
        const unsigned up1 = __shfl_down(sfx, 1);
        after = lane < 63 ? up1 : kNoKey;
      }
      unsigned run_r0 = g.r0 + (ic - lane_cnt);
      double run_p0 = g.p0 + (is - lane_sum);
      unsigned nextk[kKeysPerLane];
      unsigned cur = after != kNoKey ? after : succ_s;
#pragma unroll
      for (int u = kKeysPerLane - 1; u >= 0; --u) {
        nextk[u] = cur;
        if (kc[u]) cur = (g.pref << L2_SHIFT) | (unsigned)(lane * kKeysPerLane + u);
      }
#pragma unroll
      for (int u = 0; u < kKeysPerLane; ++u) {
        if (kc[u]) {
          const unsigned key = (g.pref << L2_SHIFT) | (unsigned)(lane * kKeysPerLane + u);
          const double v = (double)key_value(key);
          const double succ_v = nextk[u] != kNoKey ? (double)key_value(nextk[u]) : INFINITY;
          if (run_has_candidate(v, kc[u], run_r0, run_p0, succ_v, n, total, ternary)) {
            Best c;
            c.cost = cost_of(v, run_r0, run_p0, kc[u], n, total, ternary);
            c.order = run_r0;
            c.value = key_value(key);
            if (better(c, best)) best = c;
            atomicAdd(&lds->n_cand, 1u);
          }
          run_r0 += kc[u];
          run_p0 += (double)kc[u] * v;
        }
      }
    }
  }
  return best;
}

// wave-level refinement of one slot whose keys sit in list[base, base+cnt): 256-way histogram of key
// bits [18:11] private to the wave, then every flagged sub-bin is either a run of equal keys (resolved
// analytically) or has <= 64 keys (ranked by brute force with shuffles).  No workgroup barrier: the 16
// waves resolve 16 slots concurrently.  Returns false to hand the slot to resolve_slot_block.
struct WaveOut {
  Best best;
  int ok;
};

// PER sub-bins per lane: 4 (256 sub-bins of key bits [17:10]) or, for segments of at most kSmallSeg keys, 1
// (64 sub-bins of bits [17:12]: a quarter of the candidate tests and scan work per slot)
template <int PER>
__device__ __forceinline__ WaveOut resolve_slot_wave(SolverLds* lds, unsigned n, unsigned si, Best best) {
  constexpr int kSub = 64 * PER;                           // sub-bins
  constexpr int kSubBits = PER == 4 ? 8 : 6;
  constexpr int kShift = L1_SHIFT - kSubBits;              // low key bits below the sub-bin index
  const float* xrow = nullptr;
  const Args& a = lds->args;
  unsigned* const list = lds->k.list;          // kListExt keys (runs on into hist1)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const bool ternary = a.ternary != 0;
  const double total = lds->total;
  (void)a; (void)xrow; (void)n; (void)list; (void)tid; (void)lane; (void)wid; (void)ternary;
  const Slot1 s1 = lds->slot[si];
  const unsigned s1_bin = s1.bin;
  const unsigned succ_b = s1.succ;
  const unsigned* const seg = list + s1.base;
  const unsigned seg_n = s1.cnt;
  unsigned long long* const h = lds->u.whist[wid];
  LSQ_MARK(20);
  LSQ_NOTE(25, seg_n);
#pragma unroll
  for (int u = 0; u < PER; ++u) h[lane * PER + u] = 0ull;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  for (unsigned i = lane; i < seg_n; i += kWave) {
    const unsigned key = seg[i];
    atomicAdd(&h[(key >> kShift) & (kSub - 1)], kOne | (unsigned long long)(key & ((1u << kShift) - 1u)));
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  LSQ_MARK(21);
  unsigned c[PER], lane_cnt = 0, first_sub = kNoKey;
  double sm[PER], lane_sum = 0.0;
#pragma unroll
  for (int u = PER - 1; u >= 0; --u) {
    const unsigned long long hv = h[lane * PER + u];
    c[u] = (unsigned)(hv >> 42);
    const unsigned hi_key = (s1_bin << L1_SHIFT) | ((unsigned)(lane * PER + u) << kShift);
    sm[u] = c[u] ? bin_sum_exact(hi_key, c[u], hv & kLowMask) : 0.0;
    if (c[u]) first_sub = (unsigned)(lane * PER + u);
  }
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    lane_cnt += c[u];
    lane_sum += sm[u];
  }
  const unsigned ic = wave_incl_scan(lane_cnt);
  const double is = wave_incl_scan(lane_sum);
  unsigned after = kNoKey;                      // first non-empty sub-bin in a higher lane
  {
    unsigned sfx = first_sub;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned o = __shfl_down(sfx, d);
      if (lane + d < 64) sfx = min(sfx, o);
    }
    const unsigned up1 = __shfl_down(sfx, 1);
    after = lane < 63 ? up1 : kNoKey;
  }
  unsigned nsub[PER], r0s[PER];
  double p0s[PER];
  bool fl[PER];
  {
    unsigned cur = after;
#pragma unroll
    for (int u = PER - 1; u >= 0; --u) {
      nsub[u] = cur;
      if (c[u]) cur = (unsigned)(lane * PER + u);
    }
    unsigned rr = s1.r0 + (ic - lane_cnt);
    double pp = s1.p0 + (is - lane_sum);
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      r0s[u] = rr;
      p0s[u] = pp;
      rr += c[u];
      pp += sm[u];
      fl[u] = false;
      if (c[u]) {
        const unsigned hi_key = (s1_bin << L1_SHIFT) | ((unsigned)(lane * PER + u) << kShift);
        const double vlo = (double)key_value(hi_key);
        const double vhi = (double)key_value(hi_key | ((1u << kShift) - 1u));
        double next_hi = vhi;
        if (nsub[u] != kNoKey)
          next_hi = (double)key_value((s1_bin << L1_SHIFT) | (nsub[u] << kShift) | ((1u << kShift) - 1u));
        else if (succ_b != kNoKey)
          next_hi = (double)key_value(succ_b);
        fl[u] = may_hold_candidate(r0s[u], c[u], p0s[u], sm[u], vlo, vhi, next_hi, n, total, ternary);
      }
    }
  }
  // queue the flagged sub-bins
  bool ok = true;
  LSQ_MARK(22);
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    if (fl[u]) {
      const unsigned at = atomicAdd(&lds->task_cnt[0], 1u);
      if (at < (unsigned)kTaskCap) {
        SubTask tk;
        tk.slot = (unsigned short)si;
        tk.sub = (unsigned)(lane * PER + u);
        tk.bits = (unsigned short)kSubBits;
        tk.succ_key = kNoKey;
        tk.cc = c[u];
        tk.rs = r0s[u];
        tk.ns = nsub[u];
        tk.ps = p0s[u];
        lds->task[at] = tk;
      } else {
        ok = false;                              // (never seen: > 682 flagged sub-bins) block path for the slot
      }
    }
  }
  LSQ_MARK(23);
  WaveOut wo;
  wo.best = best;
  wo.ok = __ballot(!ok) == 0ull ? 1 : 0;
  return wo;
}

// A flagged sub-bin with more than 64 keys that are not all equal (a value of high multiplicity -- the image of
// the zeros of a ReLU under the next layer's batch norm -- next to ordinary keys): histogram its keys over
// the next (up to) 6 key bits, scan, and queue the children that may hold a candidate.  Same arithmetic as
// the first level: exact integer sums per child, conservative candidate test.  Returns false only when
// the queue is full.
__device__ __forceinline__ bool refine_task_wave(SolverLds* lds, unsigned n, const SubTask tk, unsigned succ_k,
                                                 unsigned child_base, unsigned* child_cnt) {
  const Args& a = lds->args;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const bool ternary = a.ternary != 0;
  const double total = lds->total;
  const Slot1 s1 = lds->slot[tk.slot];
  const unsigned* const seg = lds->k.list + s1.base;
  const unsigned seg_n = s1.cnt;
  unsigned long long* const h = lds->u.whist[wid];
  const unsigned pbits = tk.bits, pshift = (unsigned)L1_SHIFT - pbits;        // parent: low bits below its index
  const unsigned nb = min(6u, pshift);                                          // child index bits (pshift > 0 here)
  const unsigned cshift = pshift - nb;                                          // child: low bits below its index
  const unsigned pref = ((unsigned)s1.bin << pbits) | tk.sub;
  h[lane] = 0ull;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  for (unsigned i = lane; i < seg_n; i += kWave) {
    const unsigned key = seg[i];
    if ((key >> pshift) == pref)
      atomicAdd(&h[(key >> cshift) & ((1u << nb) - 1u)], kOne | (unsigned long long)(key & ((1u << cshift) - 1u)));
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  const unsigned long long hv = h[lane];
  const unsigned c = (unsigned)lane < (1u << nb) ? (unsigned)(hv >> 42) : 0u;
  const unsigned child = (tk.sub << nb) | (unsigned)lane;                       // index at the new level
  const unsigned hi_key = ((unsigned)s1.bin << L1_SHIFT) | (child << cshift);
  const double sm = c ? bin_sum_exact(hi_key, c, hv & kLowMask) : 0.0;
  const unsigned ic = wave_incl_scan(c);
  const double is = wave_incl_scan(sm);
  unsigned nxt = c ? (unsigned)lane : kNoKey;                                   // first non-empty child in a higher lane
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned o = __shfl_down(nxt, d);
    if (lane + d < 64) nxt = min(nxt, o);
  }
  const unsigned up1 = __shfl_down(nxt, 1);
  const unsigned after = lane < 63 ? up1 : kNoKey;
  bool flag = false;
  const unsigned r0 = tk.rs + (ic - c);
  const double p0 = tk.ps + (is - sm);
  if (c) {
    const double vlo = (double)key_value(hi_key);
    const double vhi = (double)key_value(hi_key | ((1u << cshift) - 1u));
    double next_hi = vhi;
    if (after != kNoKey)
      next_hi = (double)key_value(((unsigned)s1.bin << L1_SHIFT) | ((((tk.sub << nb) | after)) << cshift) | ((1u << cshift) - 1u));
    else if (succ_k != kNoKey)
      next_hi = (double)key_value(succ_k);
    flag = may_hold_candidate(r0, c, p0, sm, vlo, vhi, next_hi, n, total, ternary);
  }
  bool ok = true;
  if (flag) {
    const unsigned at = child_base + atomicAdd(child_cnt, 1u);
    if (at < (unsigned)kTaskCap) {
      SubTask ch;
      ch.slot = tk.slot;
      ch.bits = (unsigned short)(pbits + nb);
      ch.sub = child;
      ch.cc = c;
      ch.rs = r0;
      ch.ns = after != kNoKey ? ((tk.sub << nb) | after) : kNoKey;
      ch.succ_key = succ_k;
      ch.ps = p0;
      lds->task[at] = ch;
    } else {
      ok = false;
    }
  }
  return __ballot(!ok) == 0ull;
}

// phase 2 of the wave path: one flagged sub-bin.  Returns false when the sub-bin holds more than 64
// distinct keys (the slot then goes to the block path).
__device__ __forceinline__ bool resolve_task_wave(SolverLds* lds, unsigned n, const SubTask tk, Best& best,
                                                  unsigned child_base, unsigned* child_cnt) {
  const Args& a = lds->args;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const bool ternary = a.ternary != 0;
  const double total = lds->total;
  const Slot1 s1 = lds->slot[tk.slot];
  const unsigned s1_bin = s1.bin, succ_b = s1.succ;
  const unsigned* const seg = lds->k.list + s1.base;
  const unsigned seg_n = s1.cnt;
  unsigned* const wk = lds->wkeys[wid];
  const unsigned kSubBits = tk.bits, kShift = (unsigned)L1_SHIFT - kSubBits;
  const unsigned sub = tk.sub, cc = tk.cc, rs = tk.rs, ns = tk.ns;
  const double ps = tk.ps;
  LSQ_WSTAT(2, 1);
  LSQ_WSTAT(3, cc);
  const unsigned pref = (s1_bin << kSubBits) | sub;
  const unsigned npref = ns != kNoKey ? ((s1_bin << kSubBits) | ns) : kNoKey;
  // one sweep of the segment: this sub-bin's keys (first 64) + its min/max + successor key
  unsigned pos = 0, succ_l = kNoKey, kmin = kNoKey, kmax = 0u;
  for (unsigned i0 = 0; i0 < seg_n; i0 += kWave) {
    const unsigned i = i0 + lane;
    const unsigned key = i < seg_n ? seg[i] : kNoKey;
    const unsigned pk = key >> kShift;
    const bool mine = i < seg_n && pk == pref;
    const unsigned long long mm = __ballot(mine);
    if (mine) {
      const unsigned at = pos + (unsigned)__popcll(mm & ((1ull << lane) - 1ull));
      if (at < (unsigned)kWave) wk[at] = key;
      kmin = min(kmin, key);
      kmax = max(kmax, key);
    }
    pos += (unsigned)__popcll(mm);
    if (i < seg_n && pk == npref) succ_l = min(succ_l, key);
  }
  const unsigned succ_k = ns != kNoKey ? wave_min(succ_l) : (tk.succ_key != kNoKey ? tk.succ_key : succ_b);
  const double succ_v = succ_k != kNoKey ? (double)key_value(succ_k) : INFINITY;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (cc > (unsigned)kWave) {
    kmin = wave_min(kmin);
    kmax = ~wave_min(~kmax);
    if (kmin != kmax) return refine_task_wave(lds, n, tk, succ_k, child_base, child_cnt);   // dense and mixed: 6 bits deeper
    // a run of cc equal keys (saturated clamp value, exact zeros, constant rows)
    const double v = (double)key_value(kmin);
    if (run_has_candidate(v, cc, rs, ps, succ_v, n, total, ternary)) {
      Best cb;
      cb.cost = cost_of(v, rs, ps, cc, n, total, ternary);
      cb.order = rs;
      cb.value = key_value(kmin);
      if (better(cb, best)) best = cb;
      if (lane == 0) atomicAdd(&lds->n_cand, 1u);
    }
    return true;
  }
  const bool act = (unsigned)lane < cc;
  const unsigned key = act ? wk[lane] : kNoKey;
  unsigned rank = 0, below = 0, eq = 0;
  double bsum = 0.0, psum = 0.0;
  // four shuffles in flight per step (a shuffle is an LDS round trip; one per iteration made the loop a
  // latency chain of ~240 cycles per key).  Lanes >= cc hold kNoKey, which no live key is below or equal
  // to, so the padded iterations change nothing.
  for (unsigned j0 = 0; j0 < cc; j0 += 4u) {
    unsigned kj[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) kj[q] = (unsigned)__shfl((int)key, (int)((j0 + (unsigned)q) & 63u));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned j = j0 + (unsigned)q;
      const double vj = (double)key_value(kj[q]);
      const bool lt = kj[q] < key, e = kj[q] == key;
      below += lt ? 1u : 0u;
      eq += e ? 1u : 0u;
      if (lt) bsum += vj;
      if (lt || (e && j <= (unsigned)lane)) psum += vj;
      if (lt || (e && j < (unsigned)lane)) ++rank;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if (act) wk[rank] = key;                       // sorted order
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  bool cand = false;
  if (act) {
    const unsigned nk = rank + 1u < cc ? wk[rank + 1u] : succ_k;
    const double v = (double)key_value(key);
    const double nv = nk != kNoKey ? (double)key_value(nk) : INFINITY;
    const long long i = (long long)rs + rank;
    cand = i >= 1 && i <= (long long)n - 2 &&
           position_is_candidate(v, nv, (double)(i + 1), ps + psum, (double)n, total, ternary);
    if (cand) {
      Best cb;
      cb.cost = cost_of(v, rs + below, ps + bsum, eq, n, total, ternary);
      cb.order = rs + below;
      cb.value = key_value(key);
      if (better(cb, best)) best = cb;
    }
  }
  const unsigned long long firsts = __ballot(cand && below == rank);
  if (lane == 0 && firsts) atomicAdd(&lds->n_cand, (unsigned)__popcll(firsts));
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  return true;
}

// gather sweep: every sub-sampled key of a flagged bin goes to its slot's segment of the LDS list; keys
// of the bin right above a flagged bin update that slot's successor key.  Activations are re-read with
// the coalesced full-row sweep (a strided 4-byte gather of the sub-sample alone was slower); dense
// flat rows use the strided walk.
template <int VEC>
__device__ __forceinline__ void gather_keys(SolverLds* lds, const float* __restrict__ xrow, unsigned n, unsigned sb) {
  const Args a = lds->args;
  unsigned* const list = lds->k.list;
  // The flagged bins of a sub-round usually form <= 4 runs of consecutive bins: membership is then a
  // few VALU compares against SGPR constants instead of an LDS table lookup per key (the gather is
  // bound by LDS instruction issue, not by the loads), and fill[] starts at the segment base so the
  // returning atomic yields the absolute list position.
  const unsigned n_rg = (unsigned)__builtin_amdgcn_readfirstlane((int)lds->n_rg);
  unsigned lo[4], len[4], first[4], succbin[4], last[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    lo[r] = (unsigned)__builtin_amdgcn_readfirstlane((int)lds->rg_lo[r]);
    len[r] = (unsigned)__builtin_amdgcn_readfirstlane((int)lds->rg_len[r]);
    first[r] = (unsigned)__builtin_amdgcn_readfirstlane((int)lds->rg_first[r]);
    succbin[r] = (unsigned)__builtin_amdgcn_readfirstlane((int)lds->rg_succbin[r]);
    last[r] = (unsigned)__builtin_amdgcn_readfirstlane((int)lds->rg_last[r]);
  }
  auto take = [&](unsigned key) {
    const unsigned bin = key >> L1_SHIFT;
    if (n_rg) {
      unsigned gs = kNoKey;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned d = bin - lo[r];
        if (d < len[r]) gs = first[r] + d;
      }
      if (gs != kNoKey) list[atomicAdd(&lds->fill[gs], 1u)] = key;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (bin == succbin[r]) {
          unsigned* sp = &lds->slot[sb + last[r]].succ;
          if (key < *sp) atomicMin(sp, key);
        }
      }
      return;
    }
    const unsigned r = lds->role[bin];
    const unsigned gs = r & 0xFFu, ss = r >> 8;
    if (gs) list[atomicAdd(&lds->fill[gs - 1u], 1u)] = key;
    if (ss) {
      unsigned* sp = &lds->slot[sb + ss - 1u].succ;
      if (key < *sp) atomicMin(sp, key);
    }
  };
  // The gather needs no bit packing, so it does not use the lane = pixel sweep: a flat walk keeps all 1024
  // lanes busy whatever the layer's H*W is.  For skip = 3 a lane takes THREE consecutive float4s (12
  // elements starting at a multiple of 12): the sub-sampled elements are then always x of the first, w of
  // the first, z of the second and y of the third -- four keys per three loads, no per-lane phase
  // bookkeeping, every loaded line fully consumed by the three loads of the wave.  Two phases per batch
  // keep the returning LDS atomics in flight together.
  const long long M = a.row_elems;
  const bool vec_ok = a.skip == 3 && (M % 4) == 0 && (((uintptr_t)xrow) % 16) == 0 && n_rg != 0u &&
                      (a.pre_scale == nullptr || ((a.H * a.W) % 4) == 0);
  if (!vec_ok) {
    for_each_row_key(a, xrow, n, take);
    return;
  }
  const float4* __restrict__ row4 = reinterpret_cast<const float4*>(xrow);
  const unsigned nvec = (unsigned)(M / 4);
  const unsigned ntrip = (nvec + 2u) / 3u;
  const bool affine = a.pre_scale != nullptr;
  const unsigned qv = (unsigned)(a.H * a.W) / 4u;      // float4s per channel
  const float qinv = 1.0f / (float)(qv ? qv : 1u);
  constexpr int U = 3;
  for (unsigned j0 = threadIdx.x; j0 < ntrip; j0 += kThreads * U) {
    float4 v[U][3];
    float scs[U][3], shs[U][3];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned j = j0 + (unsigned)u * kThreads;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const unsigned i = min(3u * j + (unsigned)t, nvec - 1u);
        v[u][t] = row4[i];
        scs[u][t] = 1.f;
        shs[u][t] = 0.f;
        if (affine) {                 // channel of this float4: i / qv via a reciprocal, corrected
          unsigned c = (unsigned)((float)i * qinv);
          if ((c + 1u) * qv <= i) ++c;
          if (c * qv > i) --c;
          scs[u][t] = a.pre_scale[c];
          shs[u][t] = a.pre_shift[c];
        }
      }
    }
    unsigned keys[U][4], pos[U][4];
    bool tk[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned j = j0 + (unsigned)u * kThreads;
      const unsigned e0 = 12u * j;                     // flat index of the triple's first element
      float xs[4];
      xs[0] = fmaf(v[u][0].x, scs[u][0], shs[u][0]);   // (scale 1, shift 0 without a folded batch norm)
      xs[1] = fmaf(v[u][0].w, scs[u][0], shs[u][0]);
      xs[2] = fmaf(v[u][1].z, scs[u][1], shs[u][1]);
      xs[3] = fmaf(v[u][2].y, scs[u][2], shs[u][2]);
      if (!affine) {
        xs[0] = v[u][0].x; xs[1] = v[u][0].w; xs[2] = v[u][1].z; xs[3] = v[u][2].y;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool has = j < ntrip && (long long)e0 + 3 * e < M;
        const unsigned key = abs_key(clamp_sym(xs[e], a.alpha));
        const unsigned bin = key >> L1_SHIFT;
        unsigned gs = kNoKey;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const unsigned d = bin - lo[r];
          if (d < len[r]) gs = first[r] + d;
        }
        keys[u][e] = key;
        tk[u][e] = has && gs != kNoKey;
        pos[u][e] = 0u;
        if (tk[u][e]) pos[u][e] = atomicAdd(&lds->fill[gs], 1u);
        if (has) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (bin == succbin[r]) {
              unsigned* sp = &lds->slot[sb + last[r]].succ;
              if (key < *sp) atomicMin(sp, key);
            }
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (tk[u][e]) list[pos[u][e]] = keys[u][e];
  }
}

// The solve proper.  On entry the slot records of the flagged level-1 bins are in lds->slot (or, for
// tflag > kSlotCap, the level-1 histogram is back in lds->hist1) and lds->total is set.
__device__ __noinline__ unsigned l1_rescan(SolverLds* lds, unsigned n, unsigned round0) {
  return l1_scan(lds, lds->k.big.hist1, lds->k.big.nzlist1, n, round0);
}

template <int VEC>
__device__ __forceinline__ float solve_v1(SolverLds* lds, const float* __restrict__ xrow, unsigned n, unsigned minkey,
                                          unsigned tflag) {
  const Args& a = lds->args;
  const bool ternary = a.ternary != 0;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  Best best;
  best.cost = INFINITY;
  best.order = kNoKey;
  best.value = 0.f;
  unsigned* const list = lds->k.list;          // kListExt keys (runs on into hist1)
  (void)list;
  LSQ_MARK(2);
  LSQ_MARK(3);
  if (tflag > (unsigned)kSlotCap) {
    // pathological rows (hundreds of crossing bins): no LDS list, so hist1 stays intact and the
    // scan can be repeated for each group of kSlotCap flagged bins
    for (unsigned round0 = 0; round0 < tflag; round0 += kSlotCap) {
      l1_rescan(lds, n, round0);
      const unsigned nslot = min((unsigned)kSlotCap, tflag - round0);
      for (unsigned si = 0; si < nslot; ++si) best = resolve_slot_block(lds, xrow, n, si, true, best);
    }
  } else if (tflag) {
    // wave 0 prepares a sub-round: runs of consecutive flagged bins for the gather's membership test and
    // the list offsets of the slots (fill[] starts at the segment base so that the gather's returning
    // atomic yields the absolute list position)
    auto prepare_round = [&](unsigned sb, unsigned se) {
        // runs of consecutive flagged bins (slots are in ascending bin order; at most 126 per sub-round):
        // lane l looks at slots l and l + 64, a slot starts a run when its bin is not its predecessor's + 1
        const unsigned ns = se - sb;
        const unsigned b0 = (unsigned)lane < ns ? lds->slot[sb + lane].bin : kNoKey;
        const unsigned b1 = (unsigned)lane + 64u < ns ? lds->slot[sb + 64u + lane].bin : kNoKey;
        const unsigned p0 = (unsigned)__shfl_up((int)b0, 1);
        const unsigned p1x = (unsigned)__shfl_up((int)b1, 1);
        const unsigned p1 = lane == 0 ? (unsigned)__shfl((int)b0, 63) : p1x;
        const unsigned long long m0 = __ballot((unsigned)lane < ns && (lane == 0 || b0 != p0 + 1u));
        const unsigned long long m1 = __ballot((unsigned)lane + 64u < ns && b1 != p1 + 1u);
        // lane r < 4 fills in run r (its LDS accesses proceed in parallel with the other runs')
        const unsigned nr = (unsigned)(__popcll(m0) + __popcll(m1));
        auto nth_start = [&](unsigned r) {                 // slot index of the r-th run start, ns past the last
          unsigned long long w0 = m0, w1 = m1;
          unsigned st = ns;
          for (unsigned k = 0; k <= r; ++k) {
            if (w0) {
              st = (unsigned)__ffsll((long long)w0) - 1u;
              w0 &= w0 - 1ull;
            } else if (w1) {
              st = 64u + (unsigned)__ffsll((long long)w1) - 1u;
              w1 &= w1 - 1ull;
            } else {
              st = ns;
            }
          }
          return st;
        };
        if (lane < 4) {
          const unsigned r = (unsigned)lane;
          if (r >= nr || nr > 4u) {
            lds->rg_lo[r] = 0; lds->rg_len[r] = 0; lds->rg_first[r] = 0; lds->rg_succbin[r] = kNoKey; lds->rg_last[r] = 0;
          } else {
            const unsigned st = nth_start(r), en = nth_start(r + 1u);
            const unsigned lastq = en - 1u;
            lds->rg_lo[r] = lds->slot[sb + st].bin;
            lds->rg_len[r] = en - st;
            lds->rg_first[r] = st;
            lds->rg_last[r] = lastq;
            const unsigned nb = lds->slot[sb + lastq].next_bin;
            lds->rg_succbin[r] = nb == 0xFFFFu ? kNoKey : nb;
          }
        }
        if (lane == 0) {
          lds->n_slow = 0;
          lds->n_rg = nr <= 4u ? nr : 0u;
        }
        for (unsigned q = (unsigned)lane; q < se - sb; q += kWave) lds->fill[q] = lds->slot[sb + q].base;
    };
    // split the flagged bins into sub-rounds by gathered-key capacity.  Usual case: everything fits one
    // round -- wave 0 turns the counts into list offsets with a scan; otherwise one lane splits greedily.
    if (wid == 0) {
      unsigned cq[4], lane_tot = 0;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned q = (unsigned)lane * 4u + (unsigned)u;
        cq[u] = q < tflag ? lds->slot[q].cnt : 0u;
        lane_tot += cq[u];
      }
      const unsigned incl = wave_incl_scan(lane_tot);
      const unsigned all = (unsigned)__shfl((int)incl, 63);
      if (lane == 0) lds->dbg_gathered = all;          // diagnostics: keys of all flagged bins
      if (all <= (unsigned)kListExt && tflag <= (unsigned)kSubSlots) {
        unsigned acc = incl - lane_tot;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const unsigned q = (unsigned)lane * 4u + (unsigned)u;
          if (q < tflag) lds->slot[q].base = acc;
          acc += cq[u];
        }
        if (lane == 0) {
          lds->sub_begin[0] = 0;
          lds->sub_begin[1] = (unsigned short)tflag;
          lds->n_sub = 1;
        }
      } else if (lane == 0) {
        unsigned ns = 0, begin = 0, acc = 0;
        for (unsigned q = 0; q < tflag; ++q) {
          const unsigned c1 = lds->slot[q].cnt;
          if (q > begin && (acc + c1 > (unsigned)kListExt || q - begin >= (unsigned)kSubSlots)) {
            lds->sub_begin[ns++] = (unsigned short)begin;
            begin = q;
            acc = 0;
          }
          lds->slot[q].base = acc;
          acc += c1;
        }
        lds->sub_begin[ns++] = (unsigned short)begin;
        lds->sub_begin[ns] = (unsigned short)tflag;
        lds->n_sub = ns;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // this wave reads what its lanes just wrote
      prepare_round(lds->sub_begin[0], lds->sub_begin[1]);         // first (usually only) sub-round: no extra barrier
    }
    __syncthreads();
    const unsigned n_sub = lds->n_sub;
    for (unsigned sr = 0; sr < n_sub; ++sr) {
      const unsigned sb = lds->sub_begin[sr], se = lds->sub_begin[sr + 1];
      if (sr) {
        __syncthreads();                               // the previous sub-round's list and slots are done with
        if (wid == 0) prepare_round(sb, se);
        __syncthreads();
      }
      if (se - sb == 1u && lds->slot[sb].cnt > (unsigned)kListExt) {
        if (tid == 0) lds->dbg_rowpass += 1;
        best = resolve_slot_block(lds, xrow, n, sb, true, best);              // one huge bin: histogram straight from the row
        continue;
      }
      if (lds->n_rg == 0u) {            // > 4 runs: membership through the role table instead
        for (int i = tid; i < L1_BINS / 2; i += kThreads) reinterpret_cast<unsigned*>(lds->role)[i] = 0u;
        __syncthreads();
        if ((unsigned)tid < se - sb) {
          const Slot1 sl = lds->slot[sb + tid];
          atomicOr((unsigned*)&lds->role[sl.bin & ~1u], (unsigned)(tid + 1) << (16 * (sl.bin & 1u)));
          if (sl.next_bin != 0xFFFFu)
            atomicOr((unsigned*)&lds->role[sl.next_bin & ~1u], (unsigned)(tid + 1) << (8 + 16 * (sl.next_bin & 1u)));
        }
        __syncthreads();
      }
      // gather pass: every key of a flagged bin goes to its slot's segment of the LDS list; keys of
      // the bin right above a flagged bin update that slot's successor key
      LSQ_MARK(13);
      gather_keys<VEC>(lds, xrow, n, sb);
      __syncthreads();
      if (lds->n_rg) {
        // inside a run the successor of a bin's largest key is the smallest key of the next slot's segment
        for (unsigned si = sb + (unsigned)wid; si + 1 < se; si += kWaves) {
          const Slot1 nx = lds->slot[si + 1];
          if (lds->slot[si].next_bin == nx.bin) {
            unsigned mn = kNoKey;
            for (unsigned i = lane; i < nx.cnt; i += kWave) mn = min(mn, list[nx.base + i]);
            mn = wave_min(mn);
            if (lane == 0) lds->slot[si].succ = mn;
          }
        }
        __syncthreads();
      }
      LSQ_MARK(4);
      LSQ_WSTAT0();
      const long long wp_t0 = (long long)clock64();
      (void)wp_t0;
      // phase 1: a wave per slot -- sub-bin histogram and scan; flagged sub-bins are queued (the role table's
      // bytes hold the queue: the gather is done with it)
      if (tid < 3) lds->task_cnt[tid] = 0u;
      __syncthreads();
      for (unsigned si = sb + (unsigned)wid; si < se; si += kWaves) {
        LSQ_WSTAT(1, 1);
        const WaveOut wo = lds->slot[si].cnt <= (unsigned)kSmallSeg ? resolve_slot_wave<1>(lds, n, si, best)
                                                                    : resolve_slot_wave<4>(lds, n, si, best);
        best = wo.best;
        if (!wo.ok) {
          if (lane == 0 && atomicOr(&lds->slot[si].pad, 1u) == 0u) lds->slow[atomicAdd(&lds->n_slow, 1u)] = (unsigned short)si;
        }
      }
      __syncthreads();
      // phase 2: the queued sub-bins, round-robin over all 16 waves.  A dense mixed sub-bin queues children (6
      // key bits deeper) behind the current round; they form the next round.  Three rotating counters make one
      // barrier per round enough: round r appends through counter (r+1)%3, which everyone reads after the
      // barrier, while thread 0 clears counter (r+2)%3 (last read a full round ago, next used a full round on).
      {
        unsigned t_begin = 0, t_end = min(lds->task_cnt[0], (unsigned)kTaskCap);
        if (t_end == 0u) __syncthreads();            // (no round, no barrier: the slow list must still be visible)
        for (unsigned r = 0; t_begin < t_end; ++r) {
          unsigned* const child_cnt = &lds->task_cnt[(r + 1u) % 3u];
          if (tid == 0) lds->task_cnt[(r + 2u) % 3u] = 0u;
          for (unsigned t = t_begin + (unsigned)wid; t < t_end; t += kWaves) {
            const SubTask tk = lds->task[t];
            if (!resolve_task_wave(lds, n, tk, best, t_end, child_cnt)) {
              if (lane == 0 && atomicOr(&lds->slot[tk.slot].pad, 1u) == 0u)
                lds->slow[atomicAdd(&lds->n_slow, 1u)] = (unsigned short)tk.slot;
            }
          }
          __syncthreads();
          t_begin = t_end;
          t_end = min(t_end + *child_cnt, (unsigned)kTaskCap);
        }
      }
      LSQ_WSTAT(0, (long long)clock64() - wp_t0);
      LSQ_MARK(5);
      const unsigned n_slow = lds->n_slow;
      if (tid == 0) lds->dbg_slow += n_slow;
      for (unsigned q = 0; q < n_slow; ++q) best = resolve_slot_block(lds, xrow, n, lds->slow[q], false, best);
    }
  }
  LSQ_MARK(6);

  // ---- ternary: min > mean/2 adds mean/2 (optimal.py:86-118)
  if (ternary && n > 0u && tid == 0) {
    const double mean = lds->total / (double)n;
    if ((double)key_value(minkey) > 0.5 * mean) {
      const float half = (float)((double)((float)mean) / 2.0);
      Best c;
      c.cost = cost_of((double)half, 0u, 0.0, 0u, n, lds->total, true);
      c.order = n + 1u;
      c.value = half;
      if (better(c, best)) best = c;
      atomicAdd(&lds->n_cand, 1u);
    }
  }

  // ---- block argmin (first minimum in sorted order, optimal.py:151)
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    Best o;
    o.cost = __shfl_xor(best.cost, d);
    o.order = __shfl_xor(best.order, d);
    o.value = __shfl_xor(best.value, d);
    if (better(o, best)) best = o;
  }
  if (lane == 0) lds->wbest[wid] = best;
  __syncthreads();
  // (only thread 0 uses the result: wave 0 reduces the 16 wave minima with four more shuffles)
  Best r = lds->wbest[lane & (kWaves - 1)];
#pragma unroll
  for (int d = kWaves / 2; d > 0; d >>= 1) {
    Best o;
    o.cost = __shfl_xor(r.cost, d);
    o.order = __shfl_xor(r.order, d);
    o.value = __shfl_xor(r.value, d);
    if (better(o, r)) r = o;
  }
  if (tid == 0) lds->best_order = r.order;   // (read by thread 0 only, after its own store)
  return r.value;   // 0.0 when no candidate exists (zero padding wins, optimal.py:148-153)
}

// ---------------------------------------------------------------------------------------------
// Kernel A: one streaming sweep of every row: plane q, mean |residual_q| -> scale q, and for the first
// sweep of a solver scheme the level-1 histogram, its scan and the slot records (-> workspace).
template <int VEC, bool HIST, int QM>
__global__ __launch_bounds__(kThreads) void aq_sweep_kernel(Args a, int q) {
  using L = typename std::conditional<HIST, SweepLds, SmallLds>::type;
  __shared__ __attribute__((aligned(16))) unsigned char smem[sizeof(L)];
  L* lds = reinterpret_cast<L*>(smem);
  const int row = a.parts > 1 ? (int)blockIdx.y : (int)blockIdx.x;
  const int tid = threadIdx.x;
#ifdef LSQ_PHASE_CLOCKS
  if (tid == 0 && row < 1024) g_sweep_times[HIST ? 0 : 1][row][0] = (long long)wall_clock64();
#endif
  const float* xrow = a.x + (long long)row * a.row_elems;
  if constexpr (HIST) {
    for (int i = tid; i < L1_BINS; i += kThreads) lds->hist1[i] = 0ull;
  }
  if constexpr (HIST) {
    if (tid == 0) lds->args = a;         // (the scan / slot routines below read it from here)
  }
  if (tid < LSQ_MAX_PLANES) lds->sv[tid] = tid < q ? a.scales[(long long)tid * a.N + row] : 0.f;
  __syncthreads();
  if constexpr (HIST) LSQ_MARK(15);
  LSQ_MARK(0);
  PassOut po;
  if (a.flat) {
    po = flat_pass<HIST, QM>(a, lds, xrow, q);
  } else {
    unsigned long long* prow = a.planes + (long long)q * a.plane_words + (long long)row * a.row_words;
    // (the split variant only where it is used: 8- and 16-byte loads; one pixel per lane has nothing to split)
    if (!HIST && a.qmagic != 0.0) {
      if (VEC > 1 && a.csplit_log2 > 0) po = pack_pass<VEC, HIST, QM, (VEC > 1), !HIST>(a, lds, xrow, prow, q);
      else po = pack_pass<VEC, HIST, QM, false, !HIST>(a, lds, xrow, prow, q);
    } else {
      if (VEC > 1 && a.csplit_log2 > 0) po = pack_pass<VEC, HIST, QM, (VEC > 1), false>(a, lds, xrow, prow, q);
      else po = pack_pass<VEC, HIST, QM, false, false>(a, lds, xrow, prow, q);
    }
  }
  LSQ_MARK(1);
  // row sum (and, for the histogram sweep, the smallest key) in one LDS exchange
  double tot = 0.0;
  unsigned minkey = kNoKey;
  {
    const double wsum = wave_sum(po.sum);
    const unsigned wmin = HIST ? wave_min(po.minkey) : kNoKey;
    __syncthreads();
    if ((tid & 63) == 0) {
      lds->ws[tid >> 6] = wsum;
      if constexpr (HIST) lds->wa[tid >> 6] = wmin;
    }
    __syncthreads();
    for (int w = 0; w < kWaves; ++w) {
      tot += lds->ws[w];
      if constexpr (HIST) minkey = min(minkey, lds->wa[w]);
    }
  }
  LSQ_MARK(7);
  if (a.write_scale && tid == 0) {
    if (!HIST && a.parts > 1) {
      // the row's workgroups leave their partial sums and count themselves in; the last one adds the partials up in
      // part order (the same sum whichever workgroup comes last).  The 64-bit slot is (launch epoch << 32 | arrivals):
      // the first arrival of a launch finds another epoch -- whatever an earlier call, an aborted launch or an
      // uninitialised buffer left there -- and restarts the count, so the workspace needs no zero-filling and a stale
      // counter can never keep a scale from being written.
      double* partial = reinterpret_cast<double*>(a.rws + (long long)row * kSplitRow);
      unsigned long long* slot = reinterpret_cast<unsigned long long*>(partial + kMaxParts);
      __hip_atomic_store(&partial[blockIdx.x], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __atomic_thread_fence(__ATOMIC_RELEASE);
      unsigned long long seen = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), want;
      do {
        want = (unsigned)(seen >> 32) == a.epoch ? seen + 1ull : (((unsigned long long)a.epoch << 32) | 1ull);
      } while (!__hip_atomic_compare_exchange_strong(slot, &seen, want, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      if ((unsigned)want == (unsigned)a.parts) {
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        double t = 0.0;
        for (int p = 0; p < a.parts; ++p) t += __hip_atomic_load(&partial[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.scales[(long long)q * a.N + row] = (float)(t / (double)a.row_elems);
        // the launch is done with this row: tag 0 (no launch carries it), so that a REPLAY of this very launch -- a HIP
        // graph re-issues it with the same epoch argument -- starts its count afresh like any other launch
        __hip_atomic_store(slot, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      a.scales[(long long)q * a.N + row] = (float)(tot / (double)a.row_elems);
    }
  }
  if constexpr (HIST) {
    const unsigned n_sub = (unsigned)((a.row_elems + a.skip - 1) / a.skip);
    LSQ_MARK(8);
    const unsigned tflag = l1_scan(lds, lds->hist1, lds->nzlist, n_sub, 0);
    LSQ_MARK(14);
    LSQ_MARK(14);
    unsigned char* wrow = a.ws + (long long)row * kWsRow;
    if (tid == 0) {
      RowHeader h;
      h.tflag = tflag;
      h.minkey = minkey;
      h.n = n_sub;
      h.pad = 0;
      h.total = lds->total;
      h.pad2 = 0.0;
      *reinterpret_cast<RowHeader*>(wrow) = h;
    }
    const unsigned nslot = min(tflag, (unsigned)kSlotCap);
    const unsigned words = nslot * (unsigned)(sizeof(Slot1) / 4);
    const unsigned* src = reinterpret_cast<const unsigned*>(lds->slot);
    unsigned* dst = reinterpret_cast<unsigned*>(wrow + kWsSlots);
    for (unsigned i = tid; i < words; i += kThreads) dst[i] = src[i];
    if (tflag > (unsigned)kSlotCap) {           // rare: the solve kernel re-scans the histogram in groups
      unsigned long long* hd = reinterpret_cast<unsigned long long*>(wrow + kWsHist);
      for (int i = tid; i < L1_BINS; i += kThreads) hd[i] = lds->hist1[i];
    }
  }
  if constexpr (HIST) LSQ_MARK(16);
  LSQ_MARK(9);
#ifdef LSQ_PHASE_CLOCKS
  __syncthreads();
  if (tid == 0 && row < 1024) g_sweep_times[HIST ? 0 : 1][row][1] = (long long)wall_clock64();
#endif
}

// Kernel B: the solve.  Reads the row's slot records, gathers the flagged bins' keys with a second
// sweep, refines them and writes v1 (ternary: v1 also as the second scale).
template <int VEC>
__global__ __launch_bounds__(kThreads) void aq_solve_kernel(Args a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[sizeof(SolverLds)];
  SolverLds* lds = reinterpret_cast<SolverLds*>(smem);
  const int row = blockIdx.x;
  const int tid = threadIdx.x;
  const float* xrow = a.x + (long long)row * a.row_elems;
  const unsigned char* wrow = a.ws + (long long)row * kWsRow;
  LSQ_MARK(10);
#ifdef LSQ_PHASE_CLOCKS
  if (tid == 0 && row < 1024) g_block_times[row][0] = (long long)wall_clock64();
#endif
  const RowHeader h = *reinterpret_cast<const RowHeader*>(wrow);
  if (tid == 0) {
    lds->args = a;
    lds->total = h.total;
    lds->n_cand = 0;
    lds->dbg_slow = lds->dbg_gathered = lds->dbg_rowpass = 0;
  }
  if (h.tflag > (unsigned)kSlotCap) {
    const unsigned long long* hs = reinterpret_cast<const unsigned long long*>(wrow + kWsHist);
    for (int i = tid; i < L1_BINS; i += kThreads) lds->k.big.hist1[i] = hs[i];
  } else {
    const unsigned words = h.tflag * (unsigned)(sizeof(Slot1) / 4);
    const unsigned* src = reinterpret_cast<const unsigned*>(wrow + kWsSlots);
    unsigned* dst = reinterpret_cast<unsigned*>(lds->slot);
    for (unsigned i = tid; i < words; i += kThreads) dst[i] = src[i];
  }
  __syncthreads();
  LSQ_MARK(11);
  const float v1 = solve_v1<VEC>(lds, xrow, h.n, h.minkey, h.tflag);
  LSQ_MARK(12);
  if (tid == 0) {
    a.scales[row] = v1;
    if (a.ternary) a.scales[(long long)a.N + row] = v1;
    if (a.status) a.status[row] = (int)lds->n_cand;
    if (a.trace) a.trace[row] = (int)lds->best_order;
    // diagnostics for tooling (scripts/solve_stats.py): slow slots, gathered keys, row-pass slots
    RowHeader* hw = reinterpret_cast<RowHeader*>(a.ws + (long long)row * kWsRow);
    hw->pad = lds->dbg_slow | (lds->dbg_rowpass << 16);
    hw->pad2 = (double)lds->dbg_gathered;
#ifdef LSQ_PHASE_CLOCKS
    if (row < 1024) g_block_times[row][1] = (long long)wall_clock64();
#endif
  }
}

// tag of a row-split sweep launch (the arrival slot's epoch): unique per launch of this process, never 0
static std::atomic<unsigned> g_sweep_epoch{0x5eed0001u};

template <int VEC>
int launch_sweep(const Args& a_in, int q, bool hist, hipStream_t st) {
  Args a = a_in;
  do {
    a.epoch = g_sweep_epoch.fetch_add(1u, std::memory_order_relaxed);
  } while (a.epoch == 0u);
  const dim3 grid = (!hist && a.parts > 1) ? dim3((unsigned)a.parts, (unsigned)a.N) : dim3((unsigned)a.N);
  const dim3 block(kThreads);
  if (hist) hipLaunchKernelGGL((aq_sweep_kernel<VEC, true, 0>), grid, block, 0, st, a, q);
  else if (q == 0) hipLaunchKernelGGL((aq_sweep_kernel<VEC, false, 0>), grid, block, 0, st, a, q);
  else if (q == 1) hipLaunchKernelGGL((aq_sweep_kernel<VEC, false, 1>), grid, block, 0, st, a, q);
  else hipLaunchKernelGGL((aq_sweep_kernel<VEC, false, 2>), grid, block, 0, st, a, q);
  return (int)hipGetLastError();
}

// the whole sequence for one batch of rows
template <int VEC>
int run(Args a, hipStream_t st) {
  // lanes per item: as many as leave no thread idle, as long as a lane's share of the 64 channels is still a full
  // batch of loads (32 / VEC channels)
  if (!a.flat) {
    const long long items = (long long)a.Gt * ((a.H * a.W + VEC - 1) / VEC);
    // Workgroups per row (row-split sweeps): when the batch alone leaves CUs idle (small batches), up to kMaxParts
    // workgroups share a row, at most one workgroup per CU over the batch.  Needs the caller's row
    // workspace and a scale to reduce (the plain sweeps of ls-1 / gf-k; not the histogram sweep, not forced scales).
    // (measured, scripts/sweep_split.py: up to 32 rows every shape gains -- 56 x 56 x 64: 30 -> 11 us --; from 33 to 128
    //  rows only rows of half a megabyte and more do, two workgroups each; beyond, one workgroup per row fills the chip)
    // exact row sums under a clamp (the plain sweeps: not the solver's histogram sweep, not given scales)
    const bool solver_scheme = a.scheme == LSQ_SCHEME_LS2 || a.scheme == LSQ_SCHEME_LST;
    a.qmagic = 0.0;
    if (!solver_scheme && !a.forced && a.alpha > 0.f && a.row_elems <= (1ll << 22)) {
      int e2 = 0;
      (void)frexpf(a.alpha, &e2);                          // alpha = m * 2^e2, 0.5 <= m < 1: 2^e2 >= alpha
      a.qmagic = ldexp(1.5, 52 + e2 - 31);
    }
    long long parts = 1;
    if (a.rws && !a.forced && a.qmagic != 0.0) {           // (shared rows change who adds what: only with exact sums)
      if (a.N <= 32) parts = 256 / (a.N > 0 ? a.N : 1);
      else if (a.N <= 128 && a.row_elems * 4 >= (512ll << 10)) parts = 2;
      if (parts > kMaxParts) parts = kMaxParts;
      if (parts < 1) parts = 1;
    }
#ifdef LSQ_TUNE
    if (const char* e = getenv("LSQ_SWEEP_PARTS")) { if (a.rws && !a.forced) parts = atoi(e) < 1 ? 1 : (atoi(e) > kMaxParts ? kMaxParts : atoi(e)); }
#endif
    // lanes per item: as many as leave no thread of the row's workgroups idle, as long as a lane's share of the 64
    // channels is still a full batch of loads (32 / VEC channels)
    int csl = 0;
    if (VEC > 1)                                           // (one pixel per lane: 32 channels per batch, nothing to split)
      while ((items << (csl + 1)) <= parts * kThreads && (64 >> (csl + 1)) >= 32 / VEC && (64 >> (csl + 1)) >= 8) ++csl;
    a.csplit_log2 = csl;
#ifdef LSQ_TUNE
    if (const char* e = getenv("LSQ_CSPLIT")) a.csplit_log2 = atoi(e) < csl ? atoi(e) : csl;
#endif
    // (no more workgroups than the row's lanes fill)
    const long long fill = ((items << a.csplit_log2) + kThreads - 1) / kThreads;
    if (parts > fill) parts = fill;
    a.parts = (int)(parts < 1 ? 1 : parts);
  }
  const bool solver = (a.scheme == LSQ_SCHEME_LS2 || a.scheme == LSQ_SCHEME_LST) && !a.forced;
  if (a.forced) {
    hipError_t e = hipMemcpyAsync(a.scales, a.forced, sizeof(float) * (size_t)a.k * a.N, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return (int)e;
  }
  for (int q = 0; q < a.k; ++q) {
    const bool hist = solver && q == 0;
    a.write_scale = (!a.forced && !(solver && q == 0) && !(a.scheme == LSQ_SCHEME_LST && q == 1)) ? 1 : 0;
    if (a.flat && q > 0 && !a.write_scale) break;          // dense rows: no planes, nothing left to do
    if (int e = launch_sweep<VEC>(a, q, hist, st)) return e;
    if (hist) {
      hipLaunchKernelGGL((aq_solve_kernel<VEC>), dim3(a.N), dim3(kThreads), 0, st, a);
      if (int e = (int)hipGetLastError()) return e;
    }
  }
  return LSQ_OK;
}

}  // namespace
}  // namespace lsq

using namespace lsq;

// test hooks (include/lsq_hip_debug.h, not part of the product ABI): relaxed process-wide atomics, default 0
static std::atomic<int> g_force_streaming{0}, g_fused_debug{0};
static std::atomic<int*> g_solver_trace{nullptr};
// test hook: a device buffer of one int32 per row that the LS-2 / LS-T solvers (both paths) fill with the SORTED POSITION
// of the candidate they chose (first position of its run of equal keys; n + 1 = the ternary extra candidate of
// optimal.py:86-118; -1 = no candidate).  The caller sizes it for the largest batch it runs; null switches it off.
extern "C" int lsq_debug_solver_trace(int32_t* device_rows) {
  g_solver_trace.store(device_rows, std::memory_order_relaxed);
  return 0;
}
extern "C" int lsq_debug_force_streaming(int on) { return g_force_streaming.exchange(on, std::memory_order_relaxed); }
extern "C" int lsq_debug_fused_mode(int mode) { return g_fused_debug.exchange(mode, std::memory_order_relaxed); }

extern "C" int64_t lsq_act_plane_words(const lsq_conv_geom* g) {
  if (check_geom(g)) return -1;
  const int64_t cg = g->C / g->groups, Gg = (cg + 63) / 64;
  return (int64_t)g->N * g->groups * Gg * (g->H + 2 * g->pad_h) * (g->W + 2 * g->pad_w);
}

extern "C" int64_t lsq_solver_workspace_bytes(int64_t rows) { return rows > 0 ? rows * kWsRow : -1; }
extern "C" int64_t lsq_sweep_workspace_bytes(int64_t rows) { return rows > 0 ? rows * kSplitRow : -1; }

static int act_quant_impl(const float* x, int x_layout, const lsq_conv_geom* g, int scheme, int k, int skip,
                          float clamp_alpha, const float* pre_scale, const float* pre_shift,
                          const float* forced, uint64_t* planes, float* scales,
                          void* workspace, size_t workspace_bytes, void* stream) {
  if (!x || !planes || !scales) return LSQ_E_NULL;
  if (x_layout != LSQ_LAYOUT_NCHW && x_layout != LSQ_LAYOUT_SPLIT3) return LSQ_E_SCHEME;
  if (int e = check_geom(g)) return e;
  if (skip < 1) return LSQ_E_SHAPE;
  if (k < 1 || k > LSQ_MAX_PLANES) return LSQ_E_SCHEME;
  if ((scheme == LSQ_SCHEME_LS1 && k != 1) || ((scheme == LSQ_SCHEME_LS2 || scheme == LSQ_SCHEME_LST) && k != 2) ||
      scheme < LSQ_SCHEME_LS1 || scheme > LSQ_SCHEME_GF)
    return LSQ_E_SCHEME;
  Args a = {};
  a.x = x;
  a.C = g->C; a.H = g->H; a.W = g->W;
  a.row_elems = (long long)g->C * g->H * g->W;
  a.cg = g->C / g->groups;
  a.Gg = (a.cg + 63) / 64;
  a.Gt = g->groups * a.Gg;
  a.pad_h = g->pad_h; a.pad_w = g->pad_w;
  a.Hp = g->H + 2 * g->pad_h; a.Wp = g->W + 2 * g->pad_w;
  a.scheme = scheme; a.k = k; a.skip = skip; a.alpha = clamp_alpha;
  a.forced = forced;
  if ((pre_scale == nullptr) != (pre_shift == nullptr)) return LSQ_E_NULL;
  a.pre_scale = pre_scale;
  a.pre_shift = pre_shift;
  a.planes = (unsigned long long*)planes;
  a.row_words = (long long)a.Gt * a.Hp * a.Wp;
  a.plane_words = a.row_words * g->N;
  a.scales = scales;
  a.N = g->N;
  a.ternary = scheme == LSQ_SCHEME_LST;
  a.ws = (unsigned char*)workspace;
  a.trace = g_solver_trace.load(std::memory_order_relaxed);
  const bool solver = (scheme == LSQ_SCHEME_LS2 || scheme == LSQ_SCHEME_LST) && !forced;
  // the plain sweeps (ls-1, gf-k) take an OPTIONAL workspace of lsq_sweep_workspace_bytes(N) (any content): with it a row
  // may be shared by several workgroups (small batches)
  a.rws = (!solver && !forced && workspace && (long long)workspace_bytes >= (long long)g->N * kSplitRow && ((uintptr_t)workspace % 8) == 0)
              ? (unsigned char*)workspace : nullptr;
  if (solver) {
    if ((a.row_elems + skip - 1) / skip >= (1ll << 22)) return LSQ_E_TOO_LONG;
    if (!workspace) return LSQ_E_NULL;
    if ((long long)workspace_bytes < (long long)g->N * kWsRow || ((uintptr_t)workspace % 8)) return LSQ_E_WORKSPACE;
  }
  const int HW = g->H * g->W;
  const bool al16 = ((uintptr_t)x % 16) == 0, al8 = ((uintptr_t)x % 8) == 0;
  hipStream_t st = (hipStream_t)stream;
  // (gf-2 has the planes of ls-2: with given scales the same kernel serves it; without, v1 = mean |x| replaces the solve)
  const bool gf2 = scheme == LSQ_SCHEME_GF && k == 2;
  const bool forced2 = forced && (scheme == LSQ_SCHEME_LS2 || scheme == LSQ_SCHEME_LST || gf2);
  if (((solver && skip == 3) || forced2 || gf2) && !g_force_streaming.load(std::memory_order_relaxed)) {
    // single launch with the sub-sample resident on chip (lsq_act_fused.hip) when the row fits; with the caller's
    // scales (moving-average inference) both planes in one read of the input
    FusedArgs f = {};
    f.x = x; f.row_elems = a.row_elems;
    f.C = a.C; f.H = a.H; f.W = a.W; f.cg = a.cg; f.Gg = a.Gg; f.Gt = a.Gt; f.Hp = a.Hp; f.Wp = a.Wp;
    f.pad_h = a.pad_h; f.pad_w = a.pad_w;
    f.alpha = clamp_alpha; f.pre_scale = pre_scale; f.pre_shift = pre_shift;
    f.planes = a.planes; f.plane_words = a.plane_words; f.row_words = a.row_words;
    f.scales = scales; f.N = a.N; f.ternary = a.ternary; f.debug = g_fused_debug.load(std::memory_order_relaxed);
    f.forced = forced2 ? forced : nullptr;
    f.trace = g_solver_trace.load(std::memory_order_relaxed);
    f.greedy = (gf2 && !forced) ? 1 : 0;
    if (x_layout == LSQ_LAYOUT_SPLIT3) {
      // three-stream rows (include/lsq_hip.h): the single-launch solving kernels read them, nothing else does
      const int64_t S = lsq_split3_stream_floats(g->C, g->H, g->W);
      if (S <= 0 || !solver || ((uintptr_t)x % 16) != 0) return LSQ_E_UNSUPPORTED;
      f.x_s3 = (int)S;
      f.x_hp = (int)(S / g->C);
      const int e = fused_act_quant_s3(f, st);
      return e == kFusedNotEligible ? LSQ_E_UNSUPPORTED : e;
    }
    const int e = fused_act_quant(f, st);
    if (e != kFusedNotEligible) return e;
  }
  if (x_layout != LSQ_LAYOUT_NCHW) return LSQ_E_UNSUPPORTED;
  // widest loads the image allows; rows with fewer items than threads split the 64 channels of an item over lanes
  // (run<VEC>), so small images take 16-byte loads too
  if (HW % 4 == 0 && al16 && (long long)a.Gt * (HW / 4) >= 16) return run<4>(a, st);
  if (HW % 2 == 0 && al8 && (long long)a.Gt * (HW / 2) >= 16) return run<2>(a, st);
  return run<1>(a, st);
}

extern "C" int lsq_act_quant(const float* x, const lsq_conv_geom* g, int scheme, int k, int skip,
                             float clamp_alpha, const float* pre_scale, const float* pre_shift,
                             const float* forced, uint64_t* planes, float* scales,
                             void* workspace, size_t workspace_bytes, void* stream) {
  return act_quant_impl(x, LSQ_LAYOUT_NCHW, g, scheme, k, skip, clamp_alpha, pre_scale, pre_shift, forced, planes, scales, workspace,
                        workspace_bytes, stream);
}

extern "C" int lsq_act_quant_layout(const float* x, int x_layout, const lsq_conv_geom* g, int scheme, int k, int skip,
                                    float clamp_alpha, const float* pre_scale, const float* pre_shift,
                                    const float* forced, uint64_t* planes, float* scales,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  return act_quant_impl(x, x_layout, g, scheme, k, skip, clamp_alpha, pre_scale, pre_shift, forced, planes, scales, workspace,
                        workspace_bytes, stream);
}

extern "C" int lsq_solve_rows(const float* rows, int64_t R, int64_t M, int skip, int ternary,
                              float clamp_alpha, float* v12, int32_t* status, void* workspace,
                              size_t workspace_bytes, void* stream) {
  if (!rows || !v12 || !workspace) return LSQ_E_NULL;
  if (R <= 0 || M <= 0 || skip < 1 || R > 0x7FFFFFFF) return LSQ_E_SHAPE;
  if ((M + skip - 1) / skip >= (1ll << 22)) return LSQ_E_TOO_LONG;
  if ((long long)workspace_bytes < R * kWsRow || ((uintptr_t)workspace % 8)) return LSQ_E_WORKSPACE;
  Args a = {};
  a.x = rows;
  a.row_elems = M;
  a.C = 1; a.H = 1; a.W = 1; a.cg = 1; a.Gg = 1; a.Gt = 1; a.Hp = 1; a.Wp = 1;
  a.scheme = ternary ? LSQ_SCHEME_LST : LSQ_SCHEME_LS2;
  a.k = 2;
  a.skip = skip;
  a.alpha = clamp_alpha;
  a.N = (int)R;
  a.flat = 1;
  a.ternary = ternary ? 1 : 0;
  a.status = status;
  a.trace = g_solver_trace.load(std::memory_order_relaxed);
  a.scales = v12;
  a.ws = (unsigned char*)workspace;
  return run<1>(a, (hipStream_t)stream);
}

#ifdef LSQ_PHASE_CLOCKS
extern "C" int lsq_debug_read_clocks(long long* host32) {
  return (int)hipMemcpyFromSymbol(host32, HIP_SYMBOL(lsq::g_phase_clocks), 32 * sizeof(long long));
}
extern "C" int lsq_debug_read_sweep_times(long long* host4096) {
  return (int)hipMemcpyFromSymbol(host4096, HIP_SYMBOL(lsq::g_sweep_times), 4096 * sizeof(long long));
}
extern "C" int lsq_debug_read_block_times(long long* host2048) {
  return (int)hipMemcpyFromSymbol(host2048, HIP_SYMBOL(lsq::g_block_times), 2048 * sizeof(long long));
}
extern "C" int lsq_debug_read_wave_stats(long long* host64) {
  return (int)hipMemcpyFromSymbol(host64, HIP_SYMBOL(lsq::g_wave_stats), 64 * sizeof(long long));
}
#endif
