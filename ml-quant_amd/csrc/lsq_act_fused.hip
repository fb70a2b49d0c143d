// Single-launch LS-2 / LS-T activation quantizer for gfx950 (quant/binary/quantization.py:59-115 +
// quant/binary/optimal.py:41-155 behind QuantConv2d.forward, binary_conv.py:161-164).
//
// One 512-thread workgroup owns one row (one sample) from the first load to the last plane word; the streaming
// path (lsq_act_quant.hip) needs three kernels and three reads of the row for the same work.
//
//   pass 1   flat coalesced walk over the row (three consecutive float4 per lane and step: the sub-sampled elements
//            flat % 3 == 0 are .x and .w of the first, .z of the second, .y of the third): folded batch norm (from
//            an LDS copy of scale / shift), clamp, one 64-bit LDS atomic per key into the 13-bit level-1 histogram
//            (count | exact sum of the low key bits), and the KEYS STAY IN REGISTERS (132 per lane at 56x56).
//   solve    level-1 scan over the occupied bin range -> the ~10 bins that can hold a candidate (optimal.py:78-80);
//            their keys (15-20 % of the row's) are copied from the registers to an LDS list (ballot + mbcnt, one
//            list allocation per wave and 16 keys); the refinement then works on the list: per round every node
//            (a key prefix of 13 / 19 / 25 bits) is histogrammed over its next 6 bits, one wave scans a node's
//            64 children, flagged children become an analytic run (all keys equal), a brute-force task (<= 64
//            keys, captured into an LDS arena and ranked from LDS broadcast reads) or a node of the next round.
//            Same exact arithmetic as the streaming path: integer bin sums, fp64 prefix sums in fixed order,
//            conservative bin tests, exact candidate tests, closed-form cost, (cost, sorted position) argmin.
//   pass 2   lane = VEC pixels x 64 channels: both planes and sum |x - v1 b1|; compile-time channel indices,
//            v_cmp + v_addc_co_u32 bit packing (8 VALU instructions per element), double-buffered loads.
//
// HBM traffic: two reads of the row (the second one of the short rows is served by L2 / Infinity Cache) instead
// of three, one launch instead of three, no workspace.  Why 512 lanes: the per-lane working set of the solver is
// the same whatever the number of lanes, so fewer lanes with 256 VGPRs each leave room for the resident keys
// where 1024 lanes with 128 VGPRs spill; every kernel here has private_segment_fixed_size 0.
// Anything the fixed-size tables cannot take (more than 40 flagged level-1 bins, table or list overflow) goes to a
// block-level path that histograms straight from the row in memory: slow, general, exact.
//
// Every barrier of this file is lds_barrier() (lsq_common.h): the waves of a workgroup exchange data through LDS only --
// each workgroup reads its own row and writes its own plane words and scales --, and the loads pass 2 requests early
// must stay in flight across the refinement's barriers (a __syncthreads() would wait for them every time).
//
// Results are bit-identical to the streaming path and to oracle/lsq_exact.py (same candidate set, same
// closed-form cost, same argmin); scripts/fused_vs_streaming.py and tests/test_gpu_parity.py check it, the rare
// paths included (lsq_debug_fused_mode).

#include <cstring>
#include <type_traits>

#include "lsq_act_fused.h"
#include "lsq_solver_math.h"

#ifndef LSQ_WIN_HIST
#define LSQ_WIN_HIST 0
#endif
// This file is compiled TWICE (Makefile): as it is -- fused_act_quant, rows in NCHW order -- and with -DLSQ_FUSED_S3=1 --
// fused_act_quant_s3, the same kernels for rows in the THREE-STREAM layout (LSQ_LAYOUT_SPLIT3, include/lsq_hip.h: element
// (c, pixel) in stream s = (c + pixel) % 3 at s * S + c * hp + pixel / 3, S = C * hp).  There the sub-sample e % 3 == 0 of
// the v1 search (quantization.py:63, skip = 3; e = c * H W + pixel, H W % 3 == 1) is stream 0, the first contiguous third of
// the row: pass 1 reads a third of the bytes (one float4 per four keys instead of three; the keys of a channel's block, the
// pad at its end masked), pass 2 reads all three streams -- 4/3 reads of the row instead of 2.  The solve works on exact
// integer sums and ranks keys by value, so WHICH lane holds a key does not matter; pass 2 sums every pixel's |r| in the
// same channel order: planes and scales are the NCHW kernels' bit for bit.
#ifndef LSQ_FUSED_S3
#define LSQ_FUSED_S3 0
#endif
#if LSQ_FUSED_S3                                   // (developer builds: this translation unit's own phase-clock tables)
#define g_fused_times g_fused_times_s3
#define g_win_stats g_win_stats_s3
#define lsq_debug_read_fused_times lsq_debug_read_fused_times_s3
#define lsq_debug_read_win_stats lsq_debug_read_win_stats_s3
#endif
namespace lsq {
#if defined(LSQ_PHASE_CLOCKS)
__device__ long long g_fused_times[1024][16];    // constant-rate clock (100 MHz) at the phase marks of each workgroup
#define FMARK(i) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_fused_times[blockIdx.x][i] = (long long)wall_clock64(); } while (0)
__device__ int g_win_stats[1024][4];             // windowed level 1: 1 solved / 2 fell back, flags, flagged groups, flagged fine bins
#define WSTAT(p, f, g, s) do { if (threadIdx.x == 0 && blockIdx.x < 1024) { g_win_stats[blockIdx.x][0] = (p); g_win_stats[blockIdx.x][1] = (int)(f); \
  g_win_stats[blockIdx.x][2] = (int)(g); g_win_stats[blockIdx.x][3] = (int)(s); } } while (0)
#else
#define WSTAT(p, f, g, s) do {} while (0)
#define FMARK(i) do {} while (0)
#endif
namespace {

constexpr bool kS3 = LSQ_FUSED_S3 != 0;        // rows in the three-stream layout (this translation unit's kernels)

constexpr int kNzCap = 2048;                   // non-empty level-1 bins tested per chunk of the level-1 scan
constexpr int kSlotCap = 64;                   // flagged level-1 bins per scan round (slot records)
constexpr int kFastSlots = 40;                 // flagged level-1 bins the on-chip refinement takes
constexpr int kNodeCap = 48;                   // nodes over all rounds (round 0 = the flagged level-1 bins)
constexpr int kTaskCap = 256;                  // brute-force tasks over all rounds
constexpr int kCellCap = kNodeCap + kTaskCap;  // successor cells
constexpr int kArena = 4096;                   // captured keys of the brute-force tasks
constexpr int kSeg3 = 8;
constexpr int kBlkSlots = 8;                   // flagged level-1 bins the block path histograms per read of the row
constexpr unsigned kTaskBit = 0x8000u;
constexpr int kBnCap = 1024;                   // channels whose folded batch norm is staged in LDS
constexpr int kWinSlots = 32;                  // windowed level 1: flagged fine bins (each one brute-force task)
constexpr int kWinTask = 128;                  // keys a flagged fine bin may hold (two per lane of the wave that ranks them)
constexpr int kWinGroups = 24;                 // windowed level 1: flagged groups of 16 fine bins
constexpr int kLowBins = 256;                  // windowed level 1: one bin per binade for the keys below the window

struct Node {
  unsigned prefix, cnt, r0, cell;   // key >> (31 - bits); keys; sorted position in front; successor cell
  double p0;                        // prefix sum in front
  unsigned root, pad;               // level-1 slot it descends from
  unsigned kmin, kmax;              // nodes below round 0: smallest / largest key (a run of equal keys ends there)
};
struct Task {
  double ps;
  unsigned cc, rs, base, cell;      // keys, sorted position in front, arena offset, successor cell
  unsigned short node, child;       // table entry to clear once resolved
  unsigned root;
};
struct Seg3 {
  unsigned pref, next_pref, cnt, r0;
  double p0;
};
struct WGroup {                     // a flagged group of 16 fine bins: the lane that owns it, rank and prefix sum in front
  unsigned lane, r0;
  double p0;
};
struct WSlot {                      // a flagged fine bin
  unsigned bin, cnt, r0, pad;
  double p0;
};

// Everything below is parameterised by the workgroup size.
template <int kThreads>
struct Impl {
static constexpr int kWaves = kThreads / kWave;
static constexpr int kBinsPerThread = L1_BINS / kThreads;
static constexpr int kEntries = kNzCap / kThreads;      // compact-list entries per lane and chunk
// block path: one lane per level-2 bin (key bits [17:L2_SHIFT]), level 3 = the remaining low bits
static constexpr int L2_BINS = kThreads;
static constexpr int L2_SHIFT = kThreads == 1024 ? 8 : (kThreads == 512 ? 9 : 10);
static constexpr int L3_BINS = 1 << L2_SHIFT;
static constexpr int kKeysPerLane = L3_BINS / kWave;
static_assert((L2_BINS << L2_SHIFT) == (1 << L1_SHIFT), "block path covers the 18 low key bits");
struct BlockHists {
  unsigned long long hist2[kBlkSlots][L2_BINS];
  unsigned hist3[kSeg3][L3_BINS];
};

struct FixedLds {
  Slot1 slot[kSlotCap];
  Node node[kNodeCap];
  Task task[kTaskCap];
  unsigned cell[kCellCap];
  unsigned task_fill[kTaskCap];
  unsigned short nzlist2[L2_BINS];
  unsigned short slow[kFastSlots];
  Seg3 seg[kSeg3];
  unsigned succ3[kSeg3];
  unsigned blk_bin[kBlkSlots], blk_next[kBlkSlots], blk_succs[kBlkSlots];   // block path: the slots of one read of the row
  unsigned wa[kWaves], wb[kWaves], wc[kWaves];
  double ws[kWaves];
  Best wbest[kWaves];
  unsigned n_nodes, n_tasks, n_cells, arena_fill, blk_succ, n_slow, n_list, maxkey, best_order;
  unsigned long long slow_mask;
  double total;
  float v1;
  FusedArgs args;
  float bn_s[kBnCap], bn_t[kBnCap];                // folded batch norm of the row's channels (C <= kBnCap)
};
static constexpr int kRefineFixed = 2 * L1_BINS + 8 + kNodeCap * 64 * 12 + kArena * 4;   // role, succ, nhist, centry, arena
static constexpr int kListCap = ((160 * 1024 - (int)sizeof(FixedLds) - kRefineFixed) / 4) & ~63;   // keys of the flagged bins

struct WinLds {                                     // windowed level 1 (solve_windowed): lives where the scan tables of the
  unsigned long long hist_low[kLowBins];           // round-2 solve live (that solve rebuilds everything it reads)
  unsigned short role[L1_BINS + 2];                // fine bin -> low byte: slot + 1; high byte: slot + 1 of the flagged bin whose
                                                   // successor bin this is; [L1_BINS] = 0 for keys outside the window
  unsigned arena[kWinSlots][kWinTask];             // keys of the flagged fine bins
  unsigned fill[kWinSlots], succ[kWinSlots];       // keys captured; smallest key above the bin
  WGroup group[kWinGroups];
  WSlot slot[kWinSlots];
  unsigned n_groups, n_slots, flags, pad0;
  WSlot run;                                       // the bin of the row's largest key when it is a run of that key (pad = 1)
  unsigned first_bin[kWaves];                      // per wave: its lowest non-empty fine bin
  double wl[kWaves];                               // inclusive low-region sums of the block scan
};
struct FusedLds : FixedLds {
  union {
    struct {                                       // pass 1 and the level-1 scan(s); blk: block path
      unsigned long long hist1[L1_BINS];
      union {
        struct {
          unsigned short nzlist[L1_BINS];
          union {
            struct {
              unsigned nz_r0[kNzCap];
              double nz_p0[kNzCap];
            };
            BlockHists blk;                        // (the level-1 scan's prefix tables are dead between scans)
          };
        };
        WinLds w;
      };
    } a;
    struct {                                       // on-chip refinement
      unsigned short role[L1_BINS + 2];            // level-1 bin -> low byte: node + 1; high byte: successor cell + 1 of
                                                   // the flagged bin below it; [L1_BINS] = 0 for padding keys
      unsigned long long nhist[kNodeCap][64];      // node histograms of the current round
      unsigned centry[kNodeCap][64];               // low half: child node + 1 or kTaskBit | task; high half: cell + 1
      unsigned arena[kArena];                      // captured keys of the brute-force tasks
      unsigned klist[kListCap];                    // every key of the flagged bins (compacted out of the registers)
    } b;
  };
};
static_assert(sizeof(FusedLds) <= 160 * 1024, "LDS budget");

// exclusive block scan of (a, b, s); returns block totals.  All 1024 threads must call.
static __device__ __forceinline__ void block_excl_scan(unsigned& a, unsigned& b, double& s, unsigned& ta, unsigned& tb,
                                                double& ts, FusedLds* lds) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const unsigned ia = wave_incl_scan(a), ib = wave_incl_scan(b);
  const double is = wave_incl_scan(s);
  lds_barrier();
  if (lane == 63) {
    lds->wa[wid] = ia;
    lds->wb[wid] = ib;
    lds->ws[wid] = is;
  }
  lds_barrier();
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

static __device__ __forceinline__ double block_sum(double v, FusedLds* lds) {
  v = wave_sum(v);
  lds_barrier();
  if ((threadIdx.x & 63) == 0) lds->ws[threadIdx.x >> 6] = v;
  lds_barrier();
  double t = 0.0;
  for (int w = 0; w < kWaves; ++w) t += lds->ws[w];
  return t;
}

static __device__ __forceinline__ void store_lo16(unsigned* w, unsigned v) { reinterpret_cast<unsigned short*>(w)[0] = (unsigned short)v; }
static __device__ __forceinline__ void store_hi16(unsigned* w, unsigned v) { reinterpret_cast<unsigned short*>(w)[1] = (unsigned short)v; }

// ---------------------------------------------------------------------------------------------
// Level 1: scan the 8192-bin histogram, flag the bins that may hold a candidate (optimal.py:78-80) and write
// slot records for the flagged bins with ordinal in [round0, round0 + kSlotCap).  Same arithmetic as the
// streaming path's scan: exact integer bin sums, fixed-order fp64 prefix scan, conservative test.  The test
// runs over the COMPACT list of non-empty bins (they sit in a few binades, i.e. in the bin ranges of a few
// dozen lanes), kNzCap entries at a time; loops are rolled and per-entry state is re-read from LDS instead of
// being carried in registers: the resident keys of the caller stay live across this function.
static __device__ __forceinline__ unsigned l1_scan(FusedLds* lds, unsigned n, unsigned round0, bool ternary, unsigned bin_lo,
                                                   unsigned bin_hi) {
  const unsigned long long* const hist1 = lds->a.hist1;
  unsigned short* const nzl = lds->a.nzlist;
  const int tid = threadIdx.x;
  // The keys of a row sit in [bin_lo, bin_hi] (smallest / largest key of pass 1), typically a few hundred bins: the
  // lanes share THAT range, `per` consecutive bins each (1 or 2), instead of 8192 / kThreads bins of which all but
  // a few dozen lanes' are empty.
  const unsigned per = (bin_hi - bin_lo + (unsigned)kThreads) / (unsigned)kThreads;
  const unsigned b0 = bin_lo + per * (unsigned)tid;
  unsigned my_nz = 0, my_cnt = 0;
  double my_sum = 0.0;
#pragma unroll 2
  for (unsigned u = 0; u < per; ++u) {
    const unsigned b = b0 + u;
    if (b <= bin_hi) {
      const unsigned long long h = hist1[b];
      const unsigned c = (unsigned)(h >> 42);
      my_nz += c ? 1u : 0u;
      my_cnt += c;
      my_sum += c ? bin_sum_exact(b << L1_SHIFT, c, h & kLowMask) : 0.0;
    }
  }
  unsigned enz = my_nz, ecnt = my_cnt, tnz, tcnt;
  double esum = my_sum, total;
  block_excl_scan(enz, ecnt, esum, tnz, tcnt, total, lds);
  if (tid == 0) lds->total = total;
  FMARK(2);
  unsigned flag_base = 0;                       // flagged bins in front of the current chunk
  for (unsigned chunk = 0; chunk < tnz; chunk += kNzCap) {
    if (chunk) lds_barrier();                 // the previous chunk's prefixes are done with
    if (my_nz) {
      unsigned z = enz, c = ecnt;
      double sacc = esum;
#pragma unroll 2
      for (unsigned u = 0; u < per; ++u) {
        const unsigned b = b0 + u;
        if (b <= bin_hi) {
          const unsigned long long h = hist1[b];
          const unsigned cn = (unsigned)(h >> 42);
          if (cn) {
            if (chunk == 0u) nzl[z] = (unsigned short)b;
            if (z - chunk < (unsigned)kNzCap) {   // (unsigned wrap: entries in front of the chunk fail too)
              lds->a.nz_r0[z - chunk] = c;
              lds->a.nz_p0[z - chunk] = sacc;
            }
            ++z;
            c += cn;
            sacc += bin_sum_exact(b << L1_SHIFT, cn, h & kLowMask);
          }
        }
      }
    }
    lds_barrier();
    FMARK(3);
    // entry zl = e * kThreads + tid: with the usual few hundred non-empty bins every lane tests at most one
    for (int e = 0; e < kEntries && chunk + (unsigned)e * kThreads < tnz; ++e) {
      const unsigned zl = (unsigned)e * kThreads + (unsigned)tid, z = chunk + zl;
      bool fl = false;
      unsigned b = 0, cn = 0;
      double sm = 0.0;
      if (z < tnz) {
        b = nzl[z];
        const unsigned long long h = hist1[b];
        cn = (unsigned)(h >> 42);
        sm = bin_sum_exact(b << L1_SHIFT, cn, h & kLowMask);
        const double vlo = (double)key_value(b << L1_SHIFT);
        const double vhi = (double)key_value((b << L1_SHIFT) | ((1u << L1_SHIFT) - 1u));
        // next_hi bounds the smallest key above the bin: the upper edge of the next non-empty bin, and -- tighter, about
        // half a bin on dense data -- that bin's MEAN, which its exact count and sum give (the minimum of a set is at
        // most its mean; the factor covers the rounding of the quotient)
        double next_hi = vhi;
        if (z + 1 < tnz) {
          const unsigned b2 = nzl[z + 1];
          const unsigned long long h2 = hist1[b2];
          const unsigned c2 = (unsigned)(h2 >> 42);
          next_hi = (double)key_value((b2 << L1_SHIFT) | ((1u << L1_SHIFT) - 1u));
          const double mean2 = quick_div(bin_sum_exact(b2 << L1_SHIFT, c2, h2 & kLowMask), (double)c2) * (1.0 + 1e-12);
          next_hi = mean2 < next_hi ? mean2 : next_hi;
        }
        fl = n >= 3u && may_hold_candidate(lds->a.nz_r0[zl], cn, lds->a.nz_p0[zl], sm, vlo, vhi, next_hi, n, total, ternary);
      }
      unsigned eflag = fl ? 1u : 0u, dummy = 0, cflag, tdummy;
      double dzero = 0.0, tdz;
      block_excl_scan(eflag, dummy, dzero, cflag, tdummy, tdz, lds);
      const unsigned ord = flag_base + eflag;
      if (fl && ord >= round0 && ord < round0 + (unsigned)kSlotCap) {
        Slot1 sl;
        sl.bin = (unsigned short)b;
        sl.next_bin = z + 1 < tnz ? nzl[z + 1] : (unsigned short)0xFFFFu;
        sl.cnt = cn;
        sl.r0 = lds->a.nz_r0[zl];
        sl.succ = kNoKey;
        sl.base = z + 1 < tnz ? (unsigned)(hist1[nzl[z + 1]] >> 42) : 0u;   // keys in the next non-empty bin
        sl.pad = 0;
        sl.p0 = lds->a.nz_p0[zl];
        sl.sum = sm;
        lds->slot[ord - round0] = sl;
      }
      flag_base += cflag;
    }
  }
  lds_barrier();
  return flag_base;
}

// ---------------------------------------------------------------------------------------------
// Block path: the sub-sampled keys of the row straight from memory (L2 / Infinity Cache).
// three-stream rows: entries of channel c in stream 0 = pixels congruent to -c modulo 3 below H W
static __device__ __forceinline__ unsigned s3_count(unsigned c, unsigned HW) {
  const unsigned pmin = (3u - c % 3u) % 3u;
  return (HW - pmin + 2u) / 3u;
}
template <class F>
static __device__ __forceinline__ void for_each_row_key(const FusedArgs& a, const float* __restrict__ xrow, unsigned n, F f) {
  constexpr int U = 8;
  if constexpr (kS3) {
    // the sub-sample is stream 0: C blocks of x_hp floats, the first s3_count(c) of each are keys
    const unsigned hp = (unsigned)a.x_hp, tot = (unsigned)a.C * hp, HW = (unsigned)(a.H * a.W);
    for (unsigned j0 = threadIdx.x; j0 < tot; j0 += kThreads * U) {
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = xrow[min(j0 + (unsigned)u * kThreads, tot - 1u)];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const unsigned j = j0 + (unsigned)u * kThreads;
        const unsigned c = j / hp;
        if (j < tot && j - c * hp < s3_count(c, HW)) {
          float xv = v[u];
          if (a.pre_scale) xv = fmaf(xv, a.pre_scale[c], a.pre_shift[c]);
          f(abs_key(clamp_sym(xv, a.alpha)));
        }
      }
    }
    return;
  }
  for (unsigned j0 = threadIdx.x; j0 < n; j0 += kThreads * U) {
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned j = min(j0 + (unsigned)u * kThreads, n - 1u);
      v[u] = xrow[(long long)j * 3];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (j0 + (unsigned)u * kThreads < n) {
        float xv = v[u];
        if (a.pre_scale) {
          const long long flat = (long long)(j0 + (unsigned)u * kThreads) * 3;
          const int c = (int)(flat / ((long long)a.H * a.W));
          xv = fmaf(xv, a.pre_scale[c], a.pre_shift[c]);
        }
        f(abs_key(clamp_sym(xv, a.alpha)));
      }
  }
}

// Flagged level-1 bins resolved with block-wide histograms of key bits [17:L2_SHIFT] and the remaining low bits,
// read from the row in memory.  Fully general (any count, any ties).  Up to kBlkSlots bins share ONE read of the
// row for their level-2 histograms, and their flagged level-2 bins share the reads for the level-3 histograms,
// kSeg3 at a time -- rows with many oversized bins (the zeros of a ReLU spread over a few dozen per-channel constants
// by the next batch norm) took two reads of the row PER BIN before.
template <class SlotOf>
static __device__ __forceinline__ Best resolve_slots_block(FusedLds* lds, const float* __restrict__ xrow, unsigned n, unsigned nsl,
                                                    SlotOf slot_of, bool ternary, Best best) {
  const FusedArgs& a = lds->args;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const double total = lds->total;
  lds_barrier();
  for (unsigned i = tid; i < nsl * (unsigned)L2_BINS; i += kThreads) (&lds->a.blk.hist2[0][0])[i] = 0ull;
  if ((unsigned)tid < nsl) {
    const Slot1 sl = lds->slot[slot_of((unsigned)tid)];
    lds->blk_bin[tid] = sl.bin;
    lds->blk_next[tid] = sl.next_bin == 0xFFFFu ? kNoKey : (unsigned)sl.next_bin;
    lds->blk_succs[tid] = kNoKey;
  }
  lds_barrier();
  for_each_row_key(a, xrow, n, [&](unsigned key) {
    const unsigned b = key >> L1_SHIFT;
    for (unsigned j = 0; j < nsl; ++j) {
      if (b == lds->blk_bin[j])
        atomicAdd(&lds->a.blk.hist2[j][(key >> L2_SHIFT) & (L2_BINS - 1)], kOne | (unsigned long long)(key & ((1u << L2_SHIFT) - 1u)));
      if (b == lds->blk_next[j] && key < lds->blk_succs[j]) atomicMin(&lds->blk_succs[j], key);
    }
  });
  lds_barrier();

  // level 3 for the `pend` queued segments: one read of the row, one wave per segment
  auto flush = [&](unsigned pend) {
    lds_barrier();
    for (int i = tid; i < kSeg3 * L3_BINS; i += kThreads) (&lds->a.blk.hist3[0][0])[i] = 0u;
    lds_barrier();
    for_each_row_key(a, xrow, n, [&](unsigned key) {
      const unsigned p = key >> L2_SHIFT;
      for (unsigned j = 0; j < pend; ++j) {
        if (p == lds->seg[j].pref)
          atomicAdd(&lds->a.blk.hist3[j][key & (L3_BINS - 1)], 1u);
        else if (p == lds->seg[j].next_pref && key < lds->succ3[j])
          atomicMin(&lds->succ3[j], key);
      }
    });
    lds_barrier();
    if ((unsigned)wid < pend) {
      const Seg3 g = lds->seg[wid];
      const unsigned succ_s = lds->succ3[wid];
      unsigned kc[kKeysPerLane];
      unsigned lane_cnt = 0;
      double lane_sum = 0.0;
      unsigned first_key = kNoKey;
#pragma unroll
      for (int u = kKeysPerLane - 1; u >= 0; --u) {
        kc[u] = lds->a.blk.hist3[wid][lane * kKeysPerLane + u];
        if (kc[u]) first_key = (g.pref << L2_SHIFT) | (unsigned)(lane * kKeysPerLane + u);
      }
#pragma unroll
      for (int u = 0; u < kKeysPerLane; ++u) {
        lane_cnt += kc[u];
        lane_sum += (double)kc[u] * (double)key_value((g.pref << L2_SHIFT) | (unsigned)(lane * kKeysPerLane + u));
      }
      const unsigned ic = wave_incl_scan(lane_cnt);
      const double is = wave_incl_scan(lane_sum);
      unsigned after = kNoKey;
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
          }
          run_r0 += kc[u];
          run_p0 += (double)kc[u] * v;
        }
      }
    }
    lds_barrier();                               // the segment records may be overwritten
  };

  unsigned pend = 0;                               // segments queued for the next level-3 read (uniform)
  for (unsigned j = 0; j < nsl; ++j) {
    const Slot1 s1 = lds->slot[slot_of(j)];
    const unsigned s1_bin = s1.bin;
    const unsigned succ_b = lds->blk_succs[j];
    const unsigned long long h2 = lds->a.blk.hist2[j][tid];
    const unsigned c2 = (unsigned)(h2 >> 42);
    const unsigned hi_key2 = (s1_bin << L1_SHIFT) | ((unsigned)tid << L2_SHIFT);
    const double s2 = c2 ? bin_sum_exact(hi_key2, c2, h2 & kLowMask) : 0.0;
    unsigned enz2 = c2 ? 1u : 0u, ec2 = c2, tnz2, tc2;
    double es2 = s2, ts2;
    block_excl_scan(enz2, ec2, es2, tnz2, tc2, ts2, lds);
    if (c2) lds->nzlist2[enz2] = (unsigned short)tid;
    lds_barrier();
    const unsigned r02 = s1.r0 + ec2;
    const double p02 = s1.p0 + es2;
    unsigned next_sub = kNoKey;
    bool f2 = false;
    if (c2) {
      const double vlo = (double)key_value(hi_key2);
      const double vhi = (double)key_value(hi_key2 | ((1u << L2_SHIFT) - 1u));
      double next_hi = vhi;
      if (enz2 + 1 < tnz2) {
        next_sub = lds->nzlist2[enz2 + 1];
        next_hi = (double)key_value((s1_bin << L1_SHIFT) | (next_sub << L2_SHIFT) | ((1u << L2_SHIFT) - 1u));
      } else if (succ_b != kNoKey) {
        next_hi = (double)key_value(succ_b);
      }
      f2 = may_hold_candidate(r02, c2, p02, s2, vlo, vhi, next_hi, n, total, ternary);
    }
    unsigned ef2 = f2 ? 1u : 0u, d2 = 0, tf2, td2;
    double dz2 = 0.0, tdz2;
    block_excl_scan(ef2, d2, dz2, tf2, td2, tdz2, lds);
    unsigned taken = 0;                            // flagged level-2 bins of this slot already queued
    while (taken < tf2) {
      const unsigned take = min((unsigned)kSeg3 - pend, tf2 - taken);
      if (f2 && ef2 >= taken && ef2 < taken + take) {
        Seg3 g;
        g.pref = (s1_bin << (L1_SHIFT - L2_SHIFT)) | (unsigned)tid;
        g.next_pref = next_sub != kNoKey ? ((s1_bin << (L1_SHIFT - L2_SHIFT)) | next_sub) : kNoKey;
        g.cnt = c2;
        g.r0 = r02;
        g.p0 = p02;
        lds->seg[pend + ef2 - taken] = g;
        lds->succ3[pend + ef2 - taken] = next_sub != kNoKey ? kNoKey : succ_b;
      }
      pend += take;
      taken += take;
      if (pend == (unsigned)kSeg3) {
        flush(pend);
        pend = 0;
      }
    }
    lds_barrier();                               // nzlist2 and the scan scratch are reused by the next slot
  }
  if (pend) flush(pend);
  return best;
}

// ---------------------------------------------------------------------------------------------
// On-chip refinement, wave-level pieces.
static __device__ __forceinline__ void mark_slow(FusedLds* lds, unsigned root) {
  const unsigned long long bit = 1ull << root;
  if ((atomicOr(&lds->slow_mask, bit) & bit) == 0ull) lds->slow[atomicAdd(&lds->n_slow, 1u)] = (unsigned short)root;
}

// One wave scans the 64 children of node k (a key prefix of 13 + 6*depth bits): exact counts and sums, the
// conservative test, and for every child that passes it: analytic run / brute-force task / node of the next round.
static __device__ __forceinline__ void scan_node(FusedLds* lds, unsigned n, unsigned k, unsigned cur_lo, int depth, bool ternary,
                                          Best& best) {
  const int lane = threadIdx.x & 63;
  const double total = lds->total;
  const Node nd = lds->node[k];
  if (depth > 0 && nd.kmin == nd.kmax) {
    // more than 64 equal keys (the clamp value alpha, exact zeros, constants): one analytic run, no deeper rounds
    const double v = (double)key_value(nd.kmin);
    const unsigned sk = lds->cell[nd.cell];
    const double succ_v = sk != kNoKey ? (double)key_value(sk) : INFINITY;
    if (lane == 0 && run_has_candidate(v, nd.cnt, nd.r0, nd.p0, succ_v, n, total, ternary)) {
      Best cb;
      cb.cost = cost_of(v, nd.r0, nd.p0, nd.cnt, n, total, ternary);
      cb.order = nd.r0;
      cb.value = key_value(nd.kmin);
      if (better(cb, best)) best = cb;
    }
    return;
  }
  const int sh = 12 - 6 * depth;                       // key bits below the child index
  const unsigned long long hv = lds->b.nhist[k - cur_lo][lane];
  const unsigned c = (unsigned)(hv >> 42);
  const unsigned base_key = nd.prefix << (sh + 6);
  const unsigned hi_key = base_key | ((unsigned)lane << sh);
  const double sm = c ? bin_sum_exact(hi_key, c, hv & kLowMask) : 0.0;
  const unsigned ic = wave_incl_scan(c);
  const double is = wave_incl_scan(sm);
  unsigned after;                                      // first non-empty child above this one
  {
    unsigned sfx = c ? (unsigned)lane : kNoKey;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned o = __shfl_down(sfx, d);
      if (lane + d < 64) sfx = min(sfx, o);
    }
    const unsigned up1 = __shfl_down(sfx, 1);
    after = lane < 63 ? up1 : kNoKey;
  }
  const unsigned r0 = nd.r0 + (ic - c);
  const double p0 = nd.p0 + (is - sm);
  const unsigned nsucc = lds->cell[nd.cell];           // smallest key above the node (exact after this round's pass)
  const unsigned lowmask = (1u << sh) - 1u;
  bool flag = false;
  if (c) {
    const double vlo = (double)key_value(hi_key);
    const double vhi = (double)key_value(hi_key | lowmask);
    double next_hi = vhi;
    if (after != kNoKey)
      next_hi = (double)key_value(base_key | (after << sh) | lowmask);
    else if (nsucc != kNoKey)
      next_hi = (double)key_value(nsucc);
    flag = may_hold_candidate(r0, c, p0, sm, vlo, vhi, next_hi, n, total, ternary);
  }
  if (!flag) return;
  if (sh == 0) {
    // the child is one key value of multiplicity c
    const double v = (double)key_value(hi_key);
    const unsigned sk = after != kNoKey ? (base_key | after) : nsucc;
    const double succ_v = sk != kNoKey ? (double)key_value(sk) : INFINITY;
    if (run_has_candidate(v, c, r0, p0, succ_v, n, total, ternary)) {
      Best cb;
      cb.cost = cost_of(v, r0, p0, c, n, total, ternary);
      cb.order = r0;
      cb.value = key_value(hi_key);
      if (better(cb, best)) best = cb;
    }
    return;
  }
  {
    // The clamp value alpha is usually the row's largest key and comes in hundreds to thousands of copies.  A
    // child that holds the largest key holds nothing above it, so its low-bit sum equals c * low(maxkey) exactly
    // when all its keys ARE the largest key: one analytic run instead of two more rounds over those keys.
    const unsigned mx = lds->maxkey;
    if (c > (unsigned)kWave && (mx >> sh) == (hi_key >> sh) &&
        (hv & kLowMask) == (unsigned long long)c * (unsigned long long)(mx & lowmask)) {
      const double v = (double)key_value(mx);
      if (run_has_candidate(v, c, r0, p0, INFINITY, n, total, ternary)) {
        Best cb;
        cb.cost = cost_of(v, r0, p0, c, n, total, ternary);
        cb.order = r0;
        cb.value = key_value(mx);
        if (better(cb, best)) best = cb;
      }
      return;
    }
  }
  // (a record whose table slot exists but whose resources ran out is written dead -- no keys, no table entry --
  // because the round bookkeeping counts it; its level-1 bin goes to the block path)
  const unsigned cid = atomicAdd(&lds->n_cells, 1u);
  bool ok = cid < (unsigned)kCellCap;
  if (c <= (unsigned)kWave) {
    const unsigned t = atomicAdd(&lds->n_tasks, 1u);
    const unsigned ab = atomicAdd(&lds->arena_fill, c);
    ok = ok && t < (unsigned)kTaskCap && ab + c <= (unsigned)kArena;
    if (t < (unsigned)kTaskCap) {
      Task tk;
      tk.ps = p0;
      tk.cc = ok ? c : 0u;
      tk.rs = r0;
      tk.base = ok ? ab : 0u;
      tk.cell = ok ? cid : 0u;
      tk.node = (unsigned short)k;
      tk.child = (unsigned short)lane;
      tk.root = nd.root;
      lds->task[t] = tk;
      if (ok) store_lo16(&lds->b.centry[k][lane], kTaskBit | t);
    }
  } else {
    const unsigned q = atomicAdd(&lds->n_nodes, 1u);
    ok = ok && q < (unsigned)kNodeCap;
    if (q < (unsigned)kNodeCap) {
      Node ch;
      ch.prefix = (nd.prefix << 6) | (unsigned)lane;
      ch.cnt = c;
      ch.r0 = r0;
      ch.cell = ok ? cid : 0u;
      ch.p0 = p0;
      ch.root = nd.root;
      ch.pad = 0;
      ch.kmin = kNoKey;
      ch.kmax = 0u;
      lds->node[q] = ch;
      if (ok) store_lo16(&lds->b.centry[k][lane], q + 1u);
    }
  }
  if (ok) {
    lds->cell[cid] = after != kNoKey ? kNoKey : nsucc;
    if (after != kNoKey) store_hi16(&lds->b.centry[k][after], cid + 1u);
  } else {
    mark_slow(lds, nd.root);
  }
}

// One wave resolves a brute-force task: its <= 64 captured keys are ranked with shuffles, every position is
// tested exactly (optimal.py:78-80) and costed in closed form.
static __device__ __forceinline__ void resolve_task(FusedLds* lds, unsigned n, unsigned t, bool ternary, Best& best) {
  const int lane = threadIdx.x & 63;
  const double total = lds->total;
  const Task tk = lds->task[t];
  const unsigned cc = tk.cc, rs = tk.rs;
  const double ps = tk.ps;
  unsigned* const wk = lds->b.arena + tk.base;
  const unsigned succ_k = lds->cell[tk.cell];
  const bool act = (unsigned)lane < cc;
  const unsigned key = act ? wk[lane] : kNoKey;
  unsigned rank = 0, below = 0, eq = 0;
  double bsum = 0.0, psum = 0.0;
  // every lane reads the task's keys from the arena (uniform addresses: LDS broadcast, independent loads)
  // instead of passing them around with shuffles.  Slots >= cc of the padded last group compare as kNoKey.
  for (unsigned j0 = 0; j0 < cc; j0 += 4u) {
    unsigned kj[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) kj[q] = j0 + (unsigned)q < cc ? wk[j0 + (unsigned)q] : kNoKey;
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
  if (act) {
    const unsigned nk = rank + 1u < cc ? wk[rank + 1u] : succ_k;
    const double v = (double)key_value(key);
    const double nv = nk != kNoKey ? (double)key_value(nk) : INFINITY;
    const long long i = (long long)rs + rank;
    const bool cand = i >= 1 && i <= (long long)n - 2 &&
                      position_is_candidate(v, nv, (double)(i + 1), ps + psum, (double)n, total, ternary);
    if (cand) {
      Best cb;
      cb.cost = cost_of(v, rs + below, ps + bsum, eq, n, total, ternary);
      cb.order = rs + below;
      cb.value = key_value(key);
      if (better(cb, best)) best = cb;
    }
  }
  if (lane == 0) store_lo16(&lds->b.centry[tk.node][tk.child], 0u);   // captured once
}

// What a key of the list does in the pass of round `round` >= 1 once its depth-0 table entry `ce` (of node k0) is
// known to be non-zero: successor minimum, capture into a task, or further down to the histogram of a node.
static __device__ __forceinline__ void walk_key(FusedLds* lds, unsigned key, unsigned ce, int round, unsigned cur_lo) {
  for (int d = 0;; ++d) {
    if (d + 1 == round && (ce >> 16)) atomicMin(&lds->cell[(ce >> 16) - 1u], key);
    const unsigned a = ce & 0xFFFFu;
    if (a == 0u) return;
    if (a & kTaskBit) {
      const unsigned t = a & 0x7FFFu;
      const unsigned pos = atomicAdd(&lds->task_fill[t], 1u);
      if (pos < (unsigned)kWave) lds->b.arena[lds->task[t].base + pos] = key;
      return;
    }
    const unsigned k = a - 1u;                    // a node of depth d + 1
    const int sh = 6 - 6 * d;
    const unsigned c = (key >> sh) & 63u;
    if (d + 1 == round) {
      atomicAdd(&lds->b.nhist[k - cur_lo][c], kOne | (unsigned long long)(key & ((1u << sh) - 1u)));
      atomicMin(&lds->node[k].kmin, key);
      atomicMax(&lds->node[k].kmax, key);
      return;
    }
    ce = lds->b.centry[k][c];
  }
}

// The refinement of all flagged level-1 bins.  kreg = the sub-sampled keys the calling lane holds in registers
// (kNoKey = padding).  They are used ONCE: the keys of the flagged bins and of the bins right above them (15-20 %
// of the row's keys) are copied to an LDS list -- table look-ups in groups of 16 independent loads, one list
// allocation per wave -- and everything after that walks the list, so the key registers are dead before the
// histogram atomics and the wave-level routines run.
template <int NK, class Early>
static __device__ __forceinline__ Best refine_resident(FusedLds* lds, unsigned n, unsigned tflag, bool ternary, Best best,
                                                const unsigned (&kreg)[NK], Early keys_dead) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int i = tid; i < L1_BINS / 2 + 1; i += kThreads) reinterpret_cast<unsigned*>(lds->b.role)[i] = 0u;
  for (int i = tid; i < kNodeCap * 64; i += kThreads) (&lds->b.centry[0][0])[i] = 0u;
  for (unsigned i = tid; i < tflag * 64u; i += kThreads) (&lds->b.nhist[0][0])[i] = 0ull;
  if (tid == 0) {
    lds->n_nodes = tflag;
    lds->n_tasks = 0;
    lds->n_cells = tflag;
    lds->arena_fill = 0;
    lds->n_slow = 0;
    lds->n_list = 0;
    lds->slow_mask = 0ull;
  }
  lds_barrier();
  if (wid == 0) {
    // bins whose keys no longer fit the list go to the block path (ascending bin order; tflag <= kFastSlots <= 64).
    // A listed bin brings the bin right above it along (its smallest key is the bin's successor): 2 x cnt bounds it.
    const Slot1 sl = lds->slot[min((unsigned)lane, tflag - 1u)];
    const unsigned cn = (unsigned)lane < tflag ? sl.cnt : 0u;
    // Listed: the bin's keys and those of the next non-empty bin (the successor of the bin's largest key is the
    // smallest key up there).  Cumulative distinct keys: the next bin counts once when it is flagged itself.
    const unsigned above = (unsigned)__shfl_down((int)(unsigned)sl.bin, 1);        // the next flagged bin
    const bool next_flagged = (unsigned)lane + 1u < tflag && above == (unsigned)sl.next_bin;
    const unsigned nb = (unsigned)lane < tflag && !next_flagged ? sl.base : 0u;
    const unsigned incl = wave_incl_scan(cn + nb) + ((unsigned)lane < tflag && next_flagged ? sl.base : 0u);
    if ((unsigned)lane < tflag) {
      Node nd;
      nd.prefix = sl.bin;
      nd.cnt = sl.cnt;
      nd.r0 = sl.r0;
      nd.cell = (unsigned)lane;
      nd.p0 = sl.p0;
      nd.root = (unsigned)lane;
      nd.pad = 0;
      nd.kmin = kNoKey;
      nd.kmax = 0u;
      lds->node[lane] = nd;
      lds->cell[lane] = kNoKey;
      if (incl <= ((lds->args.debug & 2) ? 2048u : (unsigned)kListCap)) {
        reinterpret_cast<unsigned char*>(&lds->b.role[sl.bin])[0] = (unsigned char)(lane + 1);
        if (sl.next_bin != 0xFFFFu) reinterpret_cast<unsigned char*>(&lds->b.role[sl.next_bin])[1] = (unsigned char)(lane + 1);
      } else {
        mark_slow(lds, (unsigned)lane);
      }
    }
  }
  lds_barrier();
  // One sweep over the key registers, 16 at a time: table look-ups (independent loads), one ballot per key, ONE
  // list allocation per wave and group, then every kept key goes to base + (kept lanes below it).  Padding keys
  // (kNoKey) index the always-zero extra table entry.
  constexpr int G = 16;
#pragma unroll
  for (int g0 = 0; g0 < NK; g0 += G) {
    unsigned ent[G];
#pragma unroll
    for (int g = 0; g < G; ++g)
      if (g0 + g < NK) ent[g] = lds->b.role[min(kreg[g0 + g] >> L1_SHIFT, (unsigned)L1_BINS)];
    unsigned long long bm[G];
    unsigned tot = 0;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (g0 + g < NK) {
        bm[g] = __ballot(ent[g] != 0u);
        tot += (unsigned)__popcll(bm[g]);
      }
    }
    if (tot) {                                    // (wave-uniform)
      unsigned base = 0;
      if (lane == 0) base = atomicAdd(&lds->n_list, tot);
      base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (g0 + g < NK) {
          if (ent[g] != 0u) {
            const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(bm[g] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm[g], 0u));
            lds->b.klist[base + below] = kreg[g0 + g];
          }
          base += (unsigned)__popcll(bm[g]);
        }
      }
    }
    // This group's 16 key registers are dead: pass 2's early loads take their place, a few per group, so that the
    // request is spread over the whole copy phase (see pass2_request_part).
    {
      constexpr int NG = (NK + G - 1) / G;
      const int gi = g0 / G;
      unsigned parts = 0;
#pragma unroll
      for (int k = 0; k < kPfParts; ++k)
        if (k * NG / kPfParts == gi) parts |= 1u << k;
      if (parts) keys_dead(parts);
    }
  }
  lds_barrier();
  FMARK(5);
  const unsigned n_list = lds->n_list;
  unsigned cur_lo = 0, cur_n = tflag, tp_lo = 0, tp_n = 0;
  for (int round = 0; round < 4; ++round) {
    if (round) {
      for (unsigned i = tid; i < cur_n * 64u; i += kThreads) (&lds->b.nhist[0][0])[i] = 0ull;
      for (unsigned i = tid; i < tp_n; i += kThreads) lds->task_fill[tp_lo + i] = 0u;
      lds_barrier();
    }
    // list pass: the table look-ups of 8 keys in flight.  Round 0: every key of a flagged bin into its node's
    // histogram, keys of a successor bin into the cell minimum.  Later rounds: the (rare) non-zero depth-0 entries
    // are walked one by one.
    constexpr int LG = 8;
    const unsigned top_key = lds->maxkey;
    unsigned top_cnt = 0;
    for (unsigned i0 = tid; i0 < n_list; i0 += LG * kThreads) {
      unsigned kk[LG], ce[LG];
#pragma unroll
      for (int g = 0; g < LG; ++g) kk[g] = lds->b.klist[min(i0 + (unsigned)g * kThreads, n_list - 1u)];
#pragma unroll
      for (int g = 0; g < LG; ++g) ce[g] = lds->b.role[kk[g] >> L1_SHIFT];
      if (round == 0) {
#pragma unroll
        for (int g = 0; g < LG; ++g) {
          if (i0 + (unsigned)g * kThreads < n_list) {
            const unsigned nd = ce[g] & 0xFFu, sc = ce[g] >> 8;
            if (sc) atomicMin(&lds->cell[sc - 1u], kk[g]);
            if (kk[g] == top_key)
              ++top_cnt;                         // (thousands of copies of one key: counted here, added once)
            else if (nd)
              atomicAdd(&lds->b.nhist[nd - 1u][(kk[g] >> 12) & 63u], kOne | (unsigned long long)(kk[g] & 0xFFFu));
          }
        }
      } else {
#pragma unroll
        for (int g = 0; g < LG; ++g) {
          const unsigned nd = ce[g] & 0xFFu;
          ce[g] = nd ? lds->b.centry[nd - 1u][(kk[g] >> 12) & 63u] : 0u;
        }
        // (few keys have work left: each lane walks ITS hits one after the other -- as many steps as the busiest
        // lane has hits -- instead of the wave stepping through all LG key slots)
        unsigned nz = 0;
#pragma unroll
        for (int g = 0; g < LG; ++g) nz |= (ce[g] != 0u && i0 + (unsigned)g * kThreads < n_list ? 1u : 0u) << g;
        while (__ballot(nz != 0u)) {
          if (nz) {
            const int g = __ffs((int)nz) - 1;
            nz &= nz - 1u;
            unsigned key = kk[0], c = ce[0];
#pragma unroll
            for (int q = 1; q < LG; ++q) {
              key = g == q ? kk[q] : key;
              c = g == q ? ce[q] : c;
            }
            walk_key(lds, key, c, round, cur_lo);
          }
        }
      }
    }
    if (round == 0) {
      top_cnt = wave_sum(top_cnt);
      const unsigned nd = (unsigned)lds->b.role[min(top_key >> L1_SHIFT, (unsigned)L1_BINS)] & 0xFFu;
      if (lane == 0 && top_cnt && nd)
        atomicAdd(&lds->b.nhist[nd - 1u][(top_key >> 12) & 63u],
                  (unsigned long long)top_cnt * kOne + (unsigned long long)top_cnt * (unsigned long long)(top_key & 0xFFFu));
    }
    lds_barrier();
    if (round == 0) FMARK(6);
    if (round == 1) FMARK(11);
    const unsigned items = cur_n + tp_n;
    for (unsigned it = (unsigned)wid; it < items; it += kWaves) {
      if (it < cur_n)
        scan_node(lds, n, cur_lo + it, cur_lo, round, ternary, best);
      else
        resolve_task(lds, n, tp_lo + (it - cur_n), ternary, best);
    }
    lds_barrier();
    if (round == 0) FMARK(7);
    if (round == 1) FMARK(12);
    cur_lo += cur_n;
    tp_lo += tp_n;
    cur_n = min(lds->n_nodes, (unsigned)kNodeCap) - cur_lo;
    tp_n = min(lds->n_tasks, (unsigned)kTaskCap) - tp_lo;
    if (cur_n == 0u && tp_n == 0u) break;
  }
  FMARK(8);
#ifdef LSQ_PHASE_CLOCKS
  if (threadIdx.x == 0 && blockIdx.x < 1024) {
    g_fused_times[blockIdx.x][13] = lds->n_list;
    g_fused_times[blockIdx.x][14] = lds->n_tasks;
    g_fused_times[blockIdx.x][15] = lds->n_nodes * 1000 + tflag;
  }
#endif
  return best;
}

// ---------------------------------------------------------------------------------------------
// Windowed level 1 (round 5).  Under a symmetric clamp every key is at most key(alpha), and the rows of a network sit in the
// few binades below it: the 13 histogram bits are spent on a WINDOW of key space -- fine bin of key k >= k0 is (k - k0) >> sh,
// 8192 bins up to the top of alpha's binade (sh = 13: 1024 bins per binade over 8 binades; sh = 14: 512 over 16) -- instead
// of 32 bins per binade over all 254.  A bin still never straddles a binade (k0 is a multiple of 2^sh, 2^sh divides 2^23),
// so its sum is the same exact integer arithmetic (bin_sum_exact); the keys below the window are counted in one bin per
// binade (exact sums again).  With bins sixteen to thirty-two times finer the bins that can hold a candidate
// (optimal.py:78-80) hold a few dozen keys at most: each is ONE brute-force task, captured straight out of the key
// registers -- no key list, no node histograms, no refinement rounds: level-1 scan, group test, fine test, one sweep over
// the registers, one wave per flagged bin, argmin.  Anything the fixed tables cannot take -- a candidate possible below the
// window, a flagged fine bin with more than 64 keys that is not a run of the row's largest key, more than kWinGroups /
// kWinSlots flagged -- makes solve_windowed return false BEFORE it has touched the key registers, and the caller rebuilds
// the 32-bins-per-binade histogram from them and runs the round-2 solve (solve_from_hist): same candidates, same argmin.
static __device__ __forceinline__ void hist_add_win(FusedLds* lds, unsigned key, unsigned k0, int sh, unsigned low_base) {
#if LSQ_WIN_HIST == 1
  // one atomic per key, its address and operand selected (a branch per key costs more than the selects)
  const unsigned t = key - k0;
  const bool in = key >= k0;
  const unsigned idx = in ? min(t >> sh, (unsigned)L1_BINS - 1u) : low_base + (key >> 23);
  const unsigned low = in ? (t & ((1u << sh) - 1u)) : (key & 0x7FFFFFu);
  atomicAdd(&lds->a.hist1[idx], kOne | (unsigned long long)low);
#else
  if (key >= k0) {
    const unsigned t = key - k0;
    // (min: every key is at most key(alpha) -- the clamp maps a NaN to -alpha, fminf(fmaxf(NaN, -a), a) --, so the index is below
    //  L1_BINS by construction; the bound keeps a build with other NaN semantics from writing past the histogram)
    atomicAdd(&lds->a.hist1[min(t >> sh, (unsigned)L1_BINS - 1u)], kOne | (unsigned long long)(t & ((1u << sh) - 1u)));
  } else {                                       // below the window (rare): one bin per binade, the 23 mantissa bits summed
    atomicAdd(&lds->a.hist1[low_base + (key >> 23)], kOne | (unsigned long long)(key & 0x7FFFFFu));
  }
#endif
}

// One wave resolves a flagged fine bin: its <= 128 captured keys (two per lane beyond 64) are ranked from LDS broadcast reads,
// every position is tested exactly (optimal.py:78-80) and costed in closed form -- resolve_task's arithmetic on the windowed
// tables.
static __device__ __forceinline__ void resolve_wslot(FusedLds* lds, unsigned n, unsigned s, bool ternary, Best& best) {
  const int lane = threadIdx.x & 63;
  const double total = lds->total;
  const WSlot sl = lds->a.w.slot[s];
  const unsigned cc = sl.cnt, rs = sl.r0;
  const double ps = sl.p0;
  unsigned* const wk = lds->a.w.arena[s];
  const unsigned succ_k = lds->a.w.succ[s];
  // (the keys of a fine bin share one binade: value = 2^sc * mantissa, so the sums of the keys below a key are sums of 24-bit
  //  integers -- at most 128 of them: 32 bits -- converted and scaled once; the same numbers as fp64 sums of the values)
  const unsigned k0bin = lds->args.win_k0 + (sl.bin << lds->args.win_sh);
  const unsigned hidden = (k0bin >> 23) ? (1u << 23) : 0u;
  const double unit = __longlong_as_double((long long)(((k0bin >> 23) ? (int)(k0bin >> 23) - 150 : -149) + 1023) << 52);
  struct R {
    unsigned key, rank, below, eq;
    double bsum, psum;
  };
  auto rank_of = [&](unsigned idx) {             // key `idx` of the bin against all of them
    R r;
    r.key = idx < cc ? wk[idx] : kNoKey;
    r.rank = r.below = r.eq = 0u;
    unsigned bs = 0u, psm = 0u;
    for (unsigned j0 = 0; j0 < cc; j0 += 4u) {
      unsigned kj[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) kj[q] = j0 + (unsigned)q < cc ? wk[j0 + (unsigned)q] : kNoKey;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned j = j0 + (unsigned)q;
        const unsigned mj = (kj[q] & 0x7FFFFFu) | hidden;
        const bool lt = kj[q] < r.key, e = kj[q] == r.key;
        r.below += lt ? 1u : 0u;
        r.eq += e ? 1u : 0u;
        bs += lt ? mj : 0u;
        psm += (lt || (e && j <= idx)) ? mj : 0u;
        r.rank += (lt || (e && j < idx)) ? 1u : 0u;
      }
    }
    r.bsum = (double)bs * unit;
    r.psum = (double)psm * unit;
    return r;
  };
  auto test = [&](const R& r, unsigned idx) {    // (after the keys were rewritten in sorted order)
    if (idx >= cc) return;
    const unsigned nk = r.rank + 1u < cc ? wk[r.rank + 1u] : succ_k;
    const double v = (double)key_value(r.key);
    const double nv = nk != kNoKey ? (double)key_value(nk) : INFINITY;
    const long long i = (long long)rs + r.rank;
    const bool cand = i >= 1 && i <= (long long)n - 2 &&
                      position_is_candidate(v, nv, (double)(i + 1), ps + r.psum, (double)n, total, ternary);
    if (cand) {
      Best cb;
      cb.cost = cost_of(v, rs + r.below, ps + r.bsum, r.eq, n, total, ternary);
      cb.order = rs + r.below;
      cb.value = key_value(r.key);
      if (better(cb, best)) best = cb;
    }
  };
  const R r0 = rank_of((unsigned)lane);
  if (cc <= (unsigned)kWave) {                   // (uniform)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if ((unsigned)lane < cc) wk[r0.rank] = r0.key;            // sorted order
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    test(r0, (unsigned)lane);
  } else {
    const R r1 = rank_of((unsigned)lane + (unsigned)kWave);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    wk[r0.rank] = r0.key;
    if ((unsigned)lane + (unsigned)kWave < cc) wk[r1.rank] = r1.key;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    test(r0, (unsigned)lane);
    test(r1, (unsigned)lane + (unsigned)kWave);
  }
}

// Part 1 (S1-S3): from the windowed histogram to the slot records of the flagged fine bins; false when the row is the
// round-2 solve's.  The key registers are not touched (they stay live in the caller: loops rolled, state in LDS).
static __device__ __forceinline__ bool win_plan(FusedLds* lds, unsigned n, unsigned minkey, unsigned maxkey, bool ternary) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const unsigned k0 = lds->args.win_k0;
  const int sh = lds->args.win_sh;
  const unsigned lowmask = (1u << sh) - 1u;
  const unsigned long long* const hist = lds->a.hist1;
  WinLds& w = lds->a.w;
  constexpr unsigned BPL = L1_BINS / kThreads;                 // fine bins per lane = one group
  static_assert(BPL == 16, "a group is one DPP row of the fine test");
  // ---- S1: every lane sums ITS group of 16 consecutive fine bins (bin order), lanes 0..255 one low bin each; block scan
  // (a lane's 16 bins are 128 bytes apart from its neighbour's: read in the same order by every lane that is a 32-way bank
  //  conflict; lane L starts at bin (L >> 1) & 15 of its group and wraps around -- 32 lanes, 32 different bank pairs.  The
  //  order of the fp64 additions inside a group is then the lane's own, fixed, order: the terms are exact, the sum is
  //  the same number on every run)
  // (the 16 bins of a group share one binade -- groups are aligned and a binade is a multiple of 16 bins --, so the group's sum
  //  is ONE integer: mant0 * sum(c) + (sum(c * u) << sh) + sum(low bits), converted and scaled once: the same number as the
  //  sum of the sixteen exact bin sums, at a third of the instructions)
  unsigned gc = 0, first = kNoKey, cu = 0;
  unsigned long long lowsum = 0ull;
#pragma unroll 4
  for (unsigned uu = 0; uu < BPL; ++uu) {
    const unsigned u = (uu + ((unsigned)tid >> 1)) & (BPL - 1u);
    const unsigned b = (unsigned)tid * BPL + u;
    const unsigned long long h = hist[b];
    const unsigned c = (unsigned)(h >> 42);
    gc += c;
    cu += c * u;
    lowsum += h & kLowMask;
    first = c ? min(first, b) : first;
  }
  double gs = 0.0;
  if (gc) {
    const unsigned hk = k0 + (((unsigned)tid * BPL) << sh);   // lowest key of the group
    const int e = (int)(hk >> 23);
    long long mant = (long long)(hk & 0x7FFFFFu);
    int sc = -149;
    if (e > 0) {
      mant += 1ll << 23;
      sc = e - 150;
    }
    const long long integer = (long long)gc * mant + ((long long)cu << sh) + (long long)lowsum;     // < 2^47: exact in fp64
    gs = (double)integer * __longlong_as_double((long long)(sc + 1023) << 52);
  }
  unsigned lc = 0;
  double ls = 0.0;
  if (tid < kLowBins) {
    const unsigned long long h = w.hist_low[tid];
    lc = (unsigned)(h >> 42);
    if (lc) ls = bin_sum_exact((unsigned)tid << 23, lc, h & kLowMask);
  }
  const unsigned igc = wave_incl_scan(gc), ilc = wave_incl_scan(lc);
  const double igs = wave_incl_scan(gs), ils = wave_incl_scan(ls);
  const unsigned first_own = first;
  first = wave_min(first);
  if (lane == 63) {
    lds->wa[wid] = igc;
    lds->wb[wid] = ilc;
    lds->ws[wid] = igs;
    w.wl[wid] = ils;
  }
  if (lane == 0) {
    w.first_bin[wid] = first;
  }
  lds_barrier();
  unsigned ogc = 0, tgc = 0, tlc = 0, fbin = kNoKey;
  double ogs = 0.0, tgs = 0.0, tls = 0.0;
  for (int q = 0; q < kWaves; ++q) {
    if (q == wid) {
      ogc = tgc;
      ogs = tgs;
    }
    tgc += lds->wa[q];
    tlc += lds->wb[q];
    tgs += lds->ws[q];
    tls += w.wl[q];
    fbin = min(fbin, w.first_bin[q]);
  }
  const unsigned r0g = tlc + ogc + (igc - gc);                 // keys in front of this lane's group
  const double p0g = tls + (ogs + (igs - gs));                 // and their sum
  const double total = tls + tgs;
  if (tid == 0) lds->total = total;
  FMARK(2);
  // ---- S2: the conservative test on whole groups (and, by the last lane, on everything below the window)
  // the first non-empty fine bin ABOVE this lane's group bounds the first key above the group: the lowest non-empty bin of
  // the lanes above it in the wave (suffix minimum over the lanes), else of the waves above
  unsigned nfb;
  {
    unsigned sfx = gc ? first_own : kNoKey;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned o = (unsigned)__shfl_down((int)sfx, d);
      if (lane + d < 64) sfx = min(sfx, o);
    }
    const unsigned up1 = (unsigned)__shfl_down((int)sfx, 1);
    nfb = lane < 63 ? up1 : kNoKey;
    for (int q = wid + 1; q < kWaves; ++q) nfb = min(nfb, w.first_bin[q]);
  }
  if (gc && n >= 3u) {
    const unsigned b0 = (unsigned)tid * BPL;
    const double vlo = (double)key_value(k0 + (b0 << sh));
    const double vhi = (double)key_value(k0 + ((b0 + BPL) << sh) - 1u);
    // (no bin above: the group holds the row's largest key, and the test does not look at next_hi)
    const double next_hi = nfb != kNoKey ? (double)key_value(k0 + ((nfb + 1u) << sh) - 1u) : vhi;
    if (may_hold_candidate(r0g, gc, p0g, gs, vlo, vhi, next_hi, n, total, ternary)) {
      const unsigned gi = atomicAdd(&w.n_groups, 1u);
      if (gi < (unsigned)kWinGroups) {
        WGroup g;
        g.lane = (unsigned)tid;
        g.r0 = r0g;
        g.p0 = p0g;
        w.group[gi] = g;
      }
    }
  }
  if (tid == kThreads - 1 && tlc && n >= 3u) {
    // the keys below the window as ONE bin: [smallest key, largest key below k0]; the first key above them is the smallest
    // key of the lowest non-empty fine bin (at most that bin's upper edge and its mean)
    const double vlo = (double)key_value(minkey);
    const double vhi = (double)key_value(min(k0 - 1u, maxkey));
    double next_hi = vhi;
    if (fbin != kNoKey) {
      const unsigned long long h2 = hist[fbin];
      const unsigned c2 = (unsigned)(h2 >> 42);
      const double edge = (double)key_value(k0 + ((fbin + 1u) << sh) - 1u);
      const double mean = quick_div(bin_sum_exact(k0 + (fbin << sh), c2, h2 & kLowMask), (double)c2) * (1.0 + 1e-12);
      next_hi = mean < edge ? mean : edge;
    }
    if (may_hold_candidate(0u, tlc, 0.0, tls, vlo, vhi, next_hi, n, total, ternary)) atomicOr(&w.flags, 1u);
  }
  lds_barrier();
  FMARK(3);
  // (readfirstlane: values read from LDS are divergent as far as the compiler knows, and a divergent exit here is turned into
  //  predicated regions that keep pass 2's early request alive -- as zeros -- across the fall-back: scratch on both paths)
  const unsigned ngr = (unsigned)__builtin_amdgcn_readfirstlane((int)w.n_groups);
  const unsigned fl2 = (unsigned)__builtin_amdgcn_readfirstlane((int)(w.flags | (unsigned)(lds->args.debug & 8)));
  if (ngr > (unsigned)kWinGroups || (fl2 & 9u)) {
    WSTAT(2, w.flags, ngr, 0);
    return false;
  }
  // ---- S3: the fine bins of the flagged groups, one lane each (a group = one DPP row of 16 lanes)
  {
    const unsigned gi = (unsigned)tid >> 4, bi = (unsigned)tid & 15u;
    const bool act = gi < ngr;
    WGroup g;
    g.lane = 0u;
    g.r0 = 0u;
    g.p0 = 0.0;
    unsigned bin = 0u, c = 0u;
    unsigned long long h = 0ull;
    double sm = 0.0;
    if (act) {
      g = w.group[gi];
      bin = g.lane * BPL + bi;
      h = hist[bin];
      c = (unsigned)(h >> 42);
      if (c) sm = bin_sum_exact(k0 + (bin << sh), c, h & kLowMask);
    }
    unsigned ic = c;                                           // inclusive scan inside the row (all lanes execute the moves)
    ic += dpp_or_zero<0x111, 0xF>(ic);
    ic += dpp_or_zero<0x112, 0xF>(ic);
    ic += dpp_or_zero<0x114, 0xF>(ic);
    ic += dpp_or_zero<0x118, 0xF>(ic);
    double is = sm;
    is += dpp_or_zero<0x111, 0xF>(is);
    is += dpp_or_zero<0x112, 0xF>(is);
    is += dpp_or_zero<0x114, 0xF>(is);
    is += dpp_or_zero<0x118, 0xF>(is);
    if (act && c) {
      const unsigned r0 = g.r0 + (ic - c);
      const double p0 = g.p0 + (is - sm);
      unsigned nb = bin + 1u;                                  // the next non-empty fine bin (usually the very next)
      unsigned long long h2 = 0ull;
      while (nb < (unsigned)L1_BINS) {
        h2 = hist[nb];
        if (h2 >> 42) break;
        ++nb;
      }
      const double vlo = (double)key_value(k0 + (bin << sh));
      const double vhi = (double)key_value(k0 + ((bin + 1u) << sh) - 1u);
      double next_hi = vhi;
      if (nb < (unsigned)L1_BINS) {
        const unsigned c2 = (unsigned)(h2 >> 42);
        const double edge = (double)key_value(k0 + ((nb + 1u) << sh) - 1u);
        const double mean = quick_div(bin_sum_exact(k0 + (nb << sh), c2, h2 & kLowMask), (double)c2) * (1.0 + 1e-12);
        next_hi = mean < edge ? mean : edge;
      }
      if (may_hold_candidate(r0, c, p0, sm, vlo, vhi, next_hi, n, total, ternary)) {
        if (c <= (unsigned)kWinTask) {
          const unsigned sidx = atomicAdd(&w.n_slots, 1u);
          if (sidx < (unsigned)kWinSlots) {
            WSlot sl;
            sl.bin = bin;
            sl.cnt = c;
            sl.r0 = r0;
            sl.pad = 0u;
            sl.p0 = p0;
            w.slot[sidx] = sl;
            w.fill[sidx] = 0u;
            w.succ[sidx] = kNoKey;
            reinterpret_cast<unsigned char*>(&w.role[bin])[0] = (unsigned char)(sidx + 1u);
            if (nb < (unsigned)L1_BINS) reinterpret_cast<unsigned char*>(&w.role[nb])[1] = (unsigned char)(sidx + 1u);
          }
        } else if (maxkey >= k0 && bin == ((maxkey - k0) >> sh) &&
                   (h & kLowMask) == (unsigned long long)c * (unsigned long long)((maxkey - k0) & lowmask)) {
          // hundreds to thousands of copies of the clamp value: the bin of the row's largest key holds nothing above it, so
          // its low-bit sum equals c * low(maxkey) exactly when all its keys ARE that key: one analytic run, no keys needed.
          // Evaluated behind the key sweep (run_has_candidate is a CALL: with the keys live, half of them would be saved to
          // scratch around it).
          WSlot sl;
          sl.bin = bin;
          sl.cnt = c;
          sl.r0 = r0;
          sl.pad = 1u;
          sl.p0 = p0;
          w.run = sl;
        } else {
          atomicOr(&w.flags, 2u);                              // more keys than a task takes: the round-2 solve takes the row
        }
      }
    }
  }
  lds_barrier();
  FMARK(4);
  const unsigned ns = (unsigned)__builtin_amdgcn_readfirstlane((int)w.n_slots);
  const unsigned fl3 = (unsigned)__builtin_amdgcn_readfirstlane((int)w.flags);
  if (ns > (unsigned)kWinSlots || (fl3 & 2u)) {
    WSTAT(2, w.flags | 16u, ngr, ns);
    return false;
  }
  WSTAT(1, fl3, ngr, ns);
  return true;
}

// Part 2 (S4-S5): the keys of the flagged fine bins out of the registers, one wave per bin.  Only run when win_plan said
// yes: the early request of pass 2 (keys_dead) is DEFINED here and nowhere else -- a join of this path with the fall-back's
// (which still holds all the keys) would have both register sets live at once, and the allocator then homes the request in
// scratch and waits for every one of its loads.
template <int NK, class Early>
static __device__ __forceinline__ void win_sweep(FusedLds* lds, unsigned n, unsigned maxkey, bool ternary,
                                                 const unsigned (&kreg)[NK], Early keys_dead, Best& best) {
  const int tid = threadIdx.x, wid = tid >> 6;
  const unsigned k0 = lds->args.win_k0;
  const int sh = lds->args.win_sh;
  WinLds& w = lds->a.w;
  const unsigned ns = (unsigned)__builtin_amdgcn_readfirstlane((int)w.n_slots);
  // ---- S4: one sweep over the key registers, 16 at a time: table look-ups (independent loads), the rare hits captured into
  // their bin's arena / folded into the successor minimum.  Pass 2's early loads take the registers over group by group.
  constexpr int G = 16;
#pragma unroll
  for (int g0 = 0; g0 < NK; g0 += G) {
    unsigned ent[G];
#pragma unroll
    for (int g = 0; g < G; ++g)
      if (g0 + g < NK) ent[g] = w.role[min((kreg[g0 + g] - k0) >> sh, (unsigned)L1_BINS)];   // (below the window / padding: wraps past the table)
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (g0 + g < NK) {
        if (ent[g] != 0u) {
          const unsigned key = kreg[g0 + g];
          const unsigned own = ent[g] & 0xFFu, below = ent[g] >> 8;
          if (own) {
            const unsigned pos = atomicAdd(&w.fill[own - 1u], 1u);
            w.arena[own - 1u][pos & (unsigned)(kWinTask - 1)] = key;
          }
          if (below && key < w.succ[below - 1u]) atomicMin(&w.succ[below - 1u], key);   // (the plain read spares most of the atomics)
        }
      }
    }
    {
      constexpr int NG = (NK + G - 1) / G;
      const int gi = g0 / G;
      unsigned parts = 0;
#pragma unroll
      for (int k = 0; k < kPfParts; ++k)
        if (k * NG / kPfParts == gi) parts |= 1u << k;
      if (parts) keys_dead(parts);
    }
  }
  lds_barrier();
  FMARK(5);
  FMARK(6);
  FMARK(7);
  // ---- S5: one wave per flagged fine bin
  for (unsigned s = (unsigned)wid; s < ns; s += kWaves) resolve_wslot(lds, n, s, ternary, best);
  if (tid == kThreads - 1 && w.run.pad) {                      // (the last wave has the fewest slots)
    const WSlot sl = w.run;
    const double v = (double)key_value(maxkey);
    bool hit;
    [[clang::always_inline]] hit = run_has_candidate(v, sl.cnt, sl.r0, sl.p0, INFINITY, n, lds->total, ternary);   // (a call here
    // would save the early request's registers to scratch around it)
    if (hit) {
      Best cb;
      cb.cost = cost_of(v, sl.r0, sl.p0, sl.cnt, n, lds->total, ternary);
      cb.order = sl.r0;
      cb.value = key_value(maxkey);
      if (better(cb, best)) best = cb;
    }
  }
  FMARK(8);
#ifdef LSQ_PHASE_CLOCKS
  if (threadIdx.x == 0 && blockIdx.x < 1024) {
    g_fused_times[blockIdx.x][11] = g_fused_times[blockIdx.x][12] = g_fused_times[blockIdx.x][8];
    g_fused_times[blockIdx.x][13] = 0;
    g_fused_times[blockIdx.x][14] = ns;
    g_fused_times[blockIdx.x][15] = ns;
  }
#endif
}

// block argmin (first minimum in sorted order, optimal.py:151) -> lds->v1.  Inside a wave the minimum travels upwards on
// DPP moves (row shifts, then the two row broadcasts: lane 63 ends up with the wave's best; a lane without a source compares
// with itself) -- the butterfly of __shfl_xor is six times four LDS-crossbar round trips --, and every lane then takes the
// minimum over the eight waves' entries itself (broadcast reads), so no second barrier is needed for the result.
template <int CTRL, int ROW_MASK>
static __device__ __forceinline__ Best dpp_best(const Best& b) {
  Best o;
  const unsigned long long c = (unsigned long long)__double_as_longlong(b.cost);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)c, (int)(unsigned)c, CTRL, ROW_MASK, 0xF, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(c >> 32), (int)(unsigned)(c >> 32), CTRL, ROW_MASK, 0xF, false);
  o.cost = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
  o.order = (unsigned)__builtin_amdgcn_update_dpp((int)b.order, (int)b.order, CTRL, ROW_MASK, 0xF, false);
  o.value = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(b.value), __float_as_int(b.value), CTRL, ROW_MASK, 0xF, false));
  return o;
}
static __device__ __forceinline__ float block_argmin(FusedLds* lds, Best best) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  Best o;
  o = dpp_best<0x111, 0xF>(best); if (better(o, best)) best = o;     // row_shr:1
  o = dpp_best<0x112, 0xF>(best); if (better(o, best)) best = o;     // row_shr:2
  o = dpp_best<0x114, 0xF>(best); if (better(o, best)) best = o;     // row_shr:4
  o = dpp_best<0x118, 0xF>(best); if (better(o, best)) best = o;     // row_shr:8
  o = dpp_best<0x142, 0xA>(best); if (better(o, best)) best = o;     // row_bcast:15 into rows 1 and 3
  o = dpp_best<0x143, 0xC>(best); if (better(o, best)) best = o;     // row_bcast:31 into rows 2 and 3
  if (lane == 63) lds->wbest[wid] = best;
  lds_barrier();
  Best r = lds->wbest[0];
#pragma unroll
  for (int q = 1; q < kWaves; ++q) {
    const Best c = lds->wbest[q];
    if (better(c, r)) r = c;
  }
  if (tid == 0) {
    lds->v1 = r.value;     // 0.0 when no candidate exists (zero padding wins, optimal.py:148-153)
    lds->best_order = r.order;
  }
  return r.value;
}

// The ternary scheme's extra candidate (optimal.py:86-118: min > mean / 2 adds mean / 2), then the block argmin -> v1.
static __device__ __forceinline__ float finish_best(FusedLds* lds, unsigned n, unsigned minkey, bool ternary, Best best) {
  if (ternary && n > 0u && threadIdx.x == 0) {
    const double mean = lds->total / (double)n;
    if ((double)key_value(minkey) > 0.5 * mean) {
      const float half = (float)((double)((float)mean) / 2.0);
      Best c;
      c.cost = cost_of((double)half, 0u, 0.0, 0u, n, lds->total, true);
      c.order = n + 1u;
      c.value = half;
      if (better(c, best)) best = c;
    }
  }
  return block_argmin(lds, best);
}

// Everything between the level-1 histogram and v1.  `each_key` as in refine_resident.
template <int NK, class Early, class Drop>
static __device__ __forceinline__ float solve_from_hist(FusedLds* lds, const float* __restrict__ xrow, unsigned n, unsigned minkey,
                                                 unsigned maxkey, bool ternary, const unsigned (&kreg)[NK], Early keys_dead,
                                                 Drop early_drop, bool& early_ok) {
  const int tid = threadIdx.x;
  const unsigned bin_lo = min(minkey >> L1_SHIFT, (unsigned)L1_BINS - 1u), bin_hi = min(maxkey >> L1_SHIFT, (unsigned)L1_BINS - 1u);
  if (tid == 0) lds->maxkey = maxkey;
  Best best;
  best.cost = INFINITY;
  best.order = kNoKey;
  best.value = 0.f;
  // Usual case: <= kFastSlots flagged bins -> on-chip refinement, nothing left for the block path.  Many
  // crossing bins (hist1 stays intact): every flagged bin through the block path, kSlotCap at a time.  (The
  // block path sits behind the last use of the resident keys, so they are not live across it.)
  unsigned tflag = l1_scan(lds, n, 0u, ternary, bin_lo, bin_hi);
  FMARK(4);
  unsigned nslot, round0 = 0;
  bool listed = false;
  bool requested = false;                        // pass 2's early loads are out (and still wanted)
  if (tflag <= (unsigned)kFastSlots && !(lds->args.debug & 1)) {
    nslot = 0;
    if (tflag) {
      best = refine_resident<NK>(lds, n, tflag, ternary, best, kreg, keys_dead);
      requested = true;
      nslot = lds->n_slow;
      listed = true;
    }
    tflag = 0;
  } else {
    nslot = min((unsigned)kSlotCap, tflag);
  }
  if (nslot) {
    // (the block path needs the whole register file: what was requested early is dropped -- pass 2 loads it again --,
    // so that nothing is kept alive, or spilled, across this rare path)
    for (;;) {
      for (unsigned q = 0; q < nslot; q += kBlkSlots)
        best = resolve_slots_block(lds, xrow, n, min((unsigned)kBlkSlots, nslot - q),
                                   [&](unsigned j) { return listed ? (unsigned)lds->slow[q + j] : q + j; }, ternary, best);
      round0 += kSlotCap;
      if (round0 >= tflag) break;
      l1_scan(lds, n, round0, ternary, bin_lo, bin_hi);
      nslot = min((unsigned)kSlotCap, tflag - round0);
    }
    requested = false;
    early_drop();
  }
  early_ok = requested;
  return finish_best(lds, n, minkey, ternary, best);
}

static __device__ __forceinline__ void block_minmax(unsigned& mn, unsigned& mx, FusedLds* lds) {
  mn = wave_min(mn);
  mx = ~wave_min(~mx);
  lds_barrier();
  if ((threadIdx.x & 63) == 0) {
    lds->wa[threadIdx.x >> 6] = mn;
    lds->wb[threadIdx.x >> 6] = mx;
  }
  lds_barrier();
  mn = kNoKey;
  mx = 0u;
  for (int w = 0; w < kWaves; ++w) {
    mn = min(mn, lds->wa[w]);
    mx = max(mx, lds->wb[w]);
  }
}

static __device__ __forceinline__ void hist_add(FusedLds* lds, unsigned key) {
  atomicAdd(&lds->a.hist1[key >> L1_SHIFT], kOne | (unsigned long long)(key & ((1u << L1_SHIFT) - 1u)));
}

// ---------------------------------------------------------------------------------------------
// Pass 2.  Per element: bit0 = x >= 0 (sign(+-0) = +1, ste.py:16-18), r = xc - v1*b1 with xc the clamped value,
// bit1 = r >= 0, and |r| for the second scale (quantization.py:84-92).  With d = min(|x|, alpha) - v1:
// r = d for b1 = +1 and r = -d for b1 = -1 EXACTLY (|xc| - v1 and xc -+ v1 round the same way), so
// bit1 = ((b0 ? d : -d) >= 0) and |r| = |d|.  The two compares go straight from VCC into the plane words with
// v_addc_co_u32 (w = 2 w + carry-in): one VALU instruction per packed bit; channels ascending, so the words
// come out bit-reversed and one v_bfrev_b32 per 32 channels fixes that.
static __device__ __forceinline__ void push_bits(unsigned& w0, unsigned& w1, float& facc, float x, float d) {
  float t;
  // (the |d| accumulation sits in the same block: left to the scheduler, all 64 channels' adds sink to the end of
  // the item and their operands spill)
  asm volatile(
      "v_cmp_le_f32 vcc, 0, %[x]\n\t"
      "v_cndmask_b32_e64 %[t], -%[d], %[d], vcc\n\t"
      "v_addc_co_u32_e32 %[w0], vcc, %[w0], %[w0], vcc\n\t"
      "v_cmp_le_f32 vcc, 0, %[t]\n\t"
      "v_addc_co_u32_e32 %[w1], vcc, %[w1], %[w1], vcc\n\t"
      "v_add_f32_e64 %[f], %[f], |%[d]|"
      : [w0] "+v"(w0), [w1] "+v"(w1), [f] "+v"(facc), [t] "=&v"(t)
      : [x] "v"(x), [d] "v"(d)
      : "vcc");
}

// Pass-2 data requested EARLY (single-launch kernel): the first kPfFloats floats of the lane's first item, issued when
// the key registers die -- after the copy phase of the refinement -- so that they arrive during the refinement rounds
// and the argmin, a 10-15 us stretch in which the CU otherwise issues no load at all.
#ifndef LSQ_PB
#define LSQ_PB 4
#endif
template <int VEC>
struct Pf {
  static constexpr int UB = VEC == 1 ? 64 : 32 / VEC;     // channels per batch of loads (as in pass2_full)
  static constexpr int NB = 64 / UB;
  static constexpr int PB = VEC == 1 ? 1 : (NB < LSQ_PB ? NB : LSQ_PB);   // batches requested early: 128 floats (VEC = 1: the item's 64)
  float v[PB][UB][VEC];
};

// Loads [K, K + 1) / kPfParts of the early request.  The request is dealt out in kPfParts parts over the copy phase of
// the refinement (refine_resident), one or two behind every group of 16 key registers that phase has finished with: a
// CU keeps only some 50-60 KB of loads in flight, so a wave that asks for its 128 floats AT ONCE sits in the issue of
// those loads until most of the 256 KB of its workgroup have arrived -- measured: the phase behind the request grew by
// exactly what pass 2 saved.  An eighth at a time (32 KB per CU) fits, and the copy runs meanwhile.
static constexpr int kPfParts = 8;
// Three-stream rows: where a lane's item -- VEC pixels p0, p0 + 3, ... (one class mod 3: consecutive floats of a stream,
// starting at a multiple of VEC: aligned loads) x 64 channels -- finds channel c0 + cc: stream (c0 + cc + p0) % 3 at
// (c0 + cc) hp + p0 / 3, i.e. qb[cc % 3] + cc * hp from three per-lane pointers.
struct S3Item {
  const float* qb[3];
  int step;                                                          // hp floats
};
static __device__ __forceinline__ S3Item s3_item(const FusedArgs& a, const float* __restrict__ xrow, int c0, int p0) {
  const unsigned S = (unsigned)a.x_s3, hp = (unsigned)a.x_hp;
  const unsigned u0 = (unsigned)(c0 + p0);
  const unsigned r0 = u0 - 3u * (__umulhi(u0, 0xAAAAAAABu) >> 1);
  const float* base = xrow + ((long long)c0 * hp + (__umulhi((unsigned)p0, 0xAAAAAAABu) >> 1));
  S3Item it;
  it.qb[0] = base + (long long)r0 * S;
  it.qb[1] = base + (long long)(r0 == 2u ? 0u : r0 + 1u) * S;
  it.qb[2] = base + (long long)(r0 == 0u ? 2u : r0 - 1u) * S;
  it.step = (int)hp;
  return it;
}
// item index -> (channel word j, first pixel p0) of a three-stream row: per word 3 * PG items, PG = ceil(HW / (3 VEC))
// groups of 3 VEC consecutive pixels, item (g, s) = pixels 3 VEC g + s + 3 v
template <int VEC>
static __device__ __forceinline__ void s3_decode(int item, int PG, int& j, int& p0) {
  j = item / (3 * PG);
  const int idx = item - j * 3 * PG;
  const int g = idx / 3;
  p0 = 3 * VEC * g + (idx - 3 * g);
}

template <int VEC, int K>
static __device__ __forceinline__ void pass2_request_part(const float* __restrict__ q0, int HW, Pf<VEC>& pf, const S3Item* s3 = nullptr) {
  constexpr int L = Pf<VEC>::PB * Pf<VEC>::UB;                       // loads of the whole request
  constexpr int lo = K * L / kPfParts, hi = (K + 1) * L / kPfParts;
#pragma unroll
  for (int i = lo; i < hi; ++i) {
    constexpr int UB = Pf<VEC>::UB;
    const float* __restrict__ q = kS3 ? s3->qb[i % 3] + (long long)i * s3->step : q0 + (long long)i * HW;
    const int b = i / UB, u = i % UB;
    if constexpr (VEC == 4) {
      const float4 t = *reinterpret_cast<const float4*>(q);
      pf.v[b][u][0] = t.x; pf.v[b][u][1 % VEC] = t.y; pf.v[b][u][2 % VEC] = t.z; pf.v[b][u][3 % VEC] = t.w;
    } else if constexpr (VEC == 2) {
      const float2 t = *reinterpret_cast<const float2*>(q);
      pf.v[b][u][0] = t.x; pf.v[b][u][1 % VEC] = t.y;
    } else {
      pf.v[b][u][0] = *q;
    }
  }
}

// parts: bit mask of the parts to request now
template <int VEC>
static __device__ __forceinline__ void pass2_request(const FusedArgs& a, const float* __restrict__ xrow, int item0, Pf<VEC>& pf,
                                                     unsigned parts) {
  const int HW = a.H * a.W;
  const int PV = kS3 ? 3 * ((HW + 3 * VEC - 1) / (3 * VEC)) : (HW + VEC - 1) / VEC;
  const int items = a.Gt * PV;
  if (item0 >= items) return;
  int j, p;
  if constexpr (kS3) {
    s3_decode<VEC>(item0, PV / 3, j, p);
  } else {
    j = item0 / PV;
    p = (item0 - j * PV) * VEC;
  }
  const int grp = j / a.Gg;
  const int jj = j - grp * a.Gg;
  const int c0 = grp * a.cg + jj * 64;
  const float* __restrict__ q = xrow + (long long)c0 * HW + p;
  S3Item s3v;
  if constexpr (kS3) s3v = s3_item(a, xrow, c0, p);
  const S3Item* s3 = kS3 ? &s3v : nullptr;
  if (parts & 1u) pass2_request_part<VEC, 0>(q, HW, pf, s3);           // (compile-time masks at every call site)
  if (parts & 2u) pass2_request_part<VEC, 1>(q, HW, pf, s3);
  if (parts & 4u) pass2_request_part<VEC, 2>(q, HW, pf, s3);
  if (parts & 8u) pass2_request_part<VEC, 3>(q, HW, pf, s3);
  if (parts & 16u) pass2_request_part<VEC, 4>(q, HW, pf, s3);
  if (parts & 32u) pass2_request_part<VEC, 5>(q, HW, pf, s3);
  if (parts & 64u) pass2_request_part<VEC, 6>(q, HW, pf, s3);
  if (parts & 128u) pass2_request_part<VEC, 7>(q, HW, pf, s3);
}

// full 64-channel groups: every channel index is a compile-time constant.  PRE: the caller may hold the first batches
// of the lane's first item (pass2_request; `have_pre` is uniform over the workgroup).
template <int VEC, bool AFFINE, bool PRE = false>
static __device__ __forceinline__ double pass2_full(const FusedArgs& a, const float* bn_s, const float* bn_t,
                                                    const float* __restrict__ xrow, float v1,
                                                    unsigned long long* __restrict__ prow0, unsigned long long* __restrict__ prow1,
                                                    int item0, int item_step, const Pf<VEC>* pre = nullptr, bool have_pre = false) {
  const int HW = a.H * a.W;
  const int PV = kS3 ? 3 * ((HW + 3 * VEC - 1) / (3 * VEC)) : (HW + VEC - 1) / VEC;
  const int items = a.Gt * PV;
  constexpr int PSTEP = kS3 ? 3 : 1;                     // pixel stride of a lane's VEC values
  const float alpha = a.alpha >= 0.f ? a.alpha : INFINITY;
  double acc = 0.0;
  // loads per batch; two batches in flight.  One pixel per lane (the short rows): all 64 channels at once --
  // those launches are bound by memory round trips per lane, not by bytes
  constexpr int UB = Pf<VEC>::UB;
  constexpr int NB = Pf<VEC>::NB;
  constexpr int PB = Pf<VEC>::PB;
  auto one_item = [&](int item, auto first_tag) {
    constexpr bool FIRST = decltype(first_tag)::value;    // the batches b < PB are already in `pre`
    int j, p;
    if constexpr (kS3) {
      s3_decode<VEC>(item, PV / 3, j, p);
    } else {
      j = item / PV;
      p = (item - j * PV) * VEC;
    }
    const int grp = j / a.Gg;
    const int jj = j - grp * a.Gg;
    const int c0 = grp * a.cg + jj * 64;
    const float* __restrict__ src = xrow + (long long)c0 * HW + p;
    S3Item s3v;
    if constexpr (kS3) s3v = s3_item(a, xrow, c0, p);
    const float* __restrict__ bs = bn_s + c0;      // (LDS copies of the folded batch norm)
    const float* __restrict__ bt = bn_t + c0;
    unsigned w0[VEC][2], w1[VEC][2];
    float facc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      w0[v][0] = w0[v][1] = w1[v][0] = w1[v][1] = 0u;
      facc[v] = 0.f;
    }
    float buf[2][UB][VEC];
    const float* __restrict__ q = src + (FIRST ? (long long)PB * UB * HW : 0ll);   // running channel pointer (no table of 64 addresses)
    auto load = [&](int which, int bb) {
#pragma unroll
      for (int u = 0; u < UB; ++u, q += HW) {
        if constexpr (kS3) {
          const int cc = bb * UB + u;
          q = s3v.qb[cc % 3] + (long long)cc * s3v.step;
        }
        if constexpr (VEC == 4) {
          const float4 t = *reinterpret_cast<const float4*>(q);
          buf[which][u][0] = t.x; buf[which][u][1 % VEC] = t.y; buf[which][u][2 % VEC] = t.z; buf[which][u][3 % VEC] = t.w;
        } else if constexpr (VEC == 2) {
          const float2 t = *reinterpret_cast<const float2*>(q);
          buf[which][u][0] = t.x; buf[which][u][1 % VEC] = t.y;
        } else {
          buf[which][u][0] = *q;
        }
      }
    };
    if constexpr (!FIRST) load(0, 0);
    else if constexpr (PB < NB) load(PB & 1, PB);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      // (the scheduling barriers keep exactly two batches of loads in flight: without them the scheduler hoists all
      // 64 channels' loads to the top and spills)
      __builtin_amdgcn_sched_barrier(0);
      if (b + 1 < NB && (!FIRST || b + 1 > PB)) load((b + 1) & 1, b + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int cc = b * UB + u;                     // compile-time
        float sc = 1.f, sh = 0.f;
        if constexpr (AFFINE) {
          sc = bs[cc];
          sh = bt[cc];
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          float xv;
          if constexpr (FIRST) xv = b < PB ? pre->v[b < PB ? b : 0][u][v] : buf[b & 1][u][v];
          else xv = buf[b & 1][u][v];
          if constexpr (AFFINE) xv = fmaf(xv, sc, sh);
          const float d = fminf(fabsf(xv), alpha) - v1;
          push_bits(w0[v][cc >> 5], w1[v][cc >> 5], facc[v], xv, d);
        }
      }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const int pix = p + PSTEP * v;
      if (!kS3 || pix < HW) acc += (double)facc[v];      // (three-stream rows: a value past the image belongs to the next channel)
      if (pix < HW) {
        const int h = pix / a.W;
        const int w = pix - h * a.W;
        const long long widx = ((long long)j * a.Hp + h + a.pad_h) * a.Wp + w + a.pad_w;
        prow0[widx] = (unsigned long long)__brev(w0[v][0]) | ((unsigned long long)__brev(w0[v][1]) << 32);
        prow1[widx] = (unsigned long long)__brev(w1[v][0]) | ((unsigned long long)__brev(w1[v][1]) << 32);
      }
    }
  };
  int item = item0;
  if constexpr (PRE) {
    if (have_pre && item < items) {                    // (uniform condition; lanes without an item skip)
      one_item(item, std::true_type{});
      item += item_step;
    }
  }
  for (; item < items; item += item_step) one_item(item, std::false_type{});
  return acc;
}

// any channel count (groups that are not multiples of 64: LeNet's 20 channels, grouped convolutions)
template <int VEC>
static __device__ __forceinline__ double pass2_any(const FusedArgs& a, const float* __restrict__ xrow, float v1,
                                                   unsigned long long* __restrict__ prow0, unsigned long long* __restrict__ prow1,
                                                   int item0, int item_step) {
  const int HW = a.H * a.W;
  const int PV = (HW + VEC - 1) / VEC;
  const int items = a.Gt * PV;
  const bool affine = a.pre_scale != nullptr;
  double acc = 0.0;
  for (int item = item0; item < items; item += item_step) {
    const int j = item / PV;
    const int p = (item - j * PV) * VEC;
    const int grp = j / a.Gg;
    const int jj = j - grp * a.Gg;
    const int c0 = grp * a.cg + jj * 64;
    const int nch = min(64, a.cg - jj * 64);
    const float* __restrict__ src = xrow + (long long)c0 * HW + p;
    unsigned long long w0[VEC], w1[VEC];
    float facc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      w0[v] = w1[v] = 0ull;
      facc[v] = 0.f;
    }
    constexpr int UB = 8;
    for (int cb = 0; cb < nch; cb += UB) {
      float vals[UB][VEC], scs[UB], shs[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int cc = min(cb + u, nch - 1);
        if constexpr (VEC == 4) {
          const float4 t = *reinterpret_cast<const float4*>(src + (long long)cc * HW);
          vals[u][0] = t.x; vals[u][1 % VEC] = t.y; vals[u][2 % VEC] = t.z; vals[u][3 % VEC] = t.w;
        } else if constexpr (VEC == 2) {
          const float2 t = *reinterpret_cast<const float2*>(src + (long long)cc * HW);
          vals[u][0] = t.x; vals[u][1 % VEC] = t.y;
        } else {
          vals[u][0] = src[(long long)cc * HW];
        }
        scs[u] = affine ? a.pre_scale[c0 + cc] : 1.f;
        shs[u] = affine ? a.pre_shift[c0 + cc] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int cc = cb + u;
        if (cc < nch) {
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            const float xv = clamp_sym(affine ? fmaf(vals[u][v], scs[u], shs[u]) : vals[u][v], a.alpha);
            const bool b0 = xv >= 0.f;
            const float r = xv - (b0 ? v1 : -v1);      // x - v1*b1: one rounding, as the reference
            w0[v] |= (unsigned long long)b0 << cc;
            w1[v] |= (unsigned long long)(r >= 0.f) << cc;
            facc[v] += fabsf(r);
          }
        }
      }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      acc += (double)facc[v];
      const int pix = p + v;
      if (pix < HW) {
        const int h = pix / a.W;
        const int w = pix - h * a.W;
        const long long widx = ((long long)j * a.Hp + h + a.pad_h) * a.Wp + w + a.pad_w;
        prow0[widx] = w0[v];
        prow1[widx] = w1[v];
      }
    }
  }
  return acc;
}

// ---------------------------------------------------------------------------------------------
// The kernel body.  Pass 1 = flat walk over the row (three consecutive float4 per lane and step: the
// sub-sampled elements flat % 3 == 0 are always .x and .w of the first, .z of the second, .y of the third):
// level-1 histogram, the keys kept in registers.  Solve.  Pass 2 = lane = VEC pixels x 64 channels sweep that
// packs both planes and sums |x - v1 b1| (quantization.py:84-92, :112-115); for the short rows this second
// read is served by the L2 / Infinity Cache the first one filled.
template <int U, int VEC, bool WIN>
static __device__ __forceinline__ void run(const FusedArgs& a, FusedLds* lds) {
  const int row = blockIdx.x;
  const int tid = threadIdx.x;
  const float* __restrict__ xrow = a.x + (long long)row * (kS3 ? 3ll * a.x_s3 : a.row_elems);
  FMARK(0);
  for (int i = tid; i < L1_BINS; i += kThreads) lds->a.hist1[i] = 0ull;
  // windowed level-1 histogram (solve_windowed): fine bins over the binades below the clamp value, one bin per binade below them
  constexpr bool win = WIN;                      // (the host launches the WIN kernels exactly when it has set a window)
  const unsigned win_k0 = a.win_k0;
  const int win_sh = a.win_sh;
  const unsigned low_base = (unsigned)(lds->a.w.hist_low - lds->a.hist1);
  if (win) {
    for (int i = tid; i < kLowBins; i += kThreads) lds->a.w.hist_low[i] = 0ull;
    for (int i = tid; i < (L1_BINS + 2) / 2; i += kThreads) reinterpret_cast<unsigned*>(lds->a.w.role)[i] = 0u;
    if (tid == 0) lds->a.w.n_groups = lds->a.w.n_slots = lds->a.w.flags = lds->a.w.run.pad = 0u;
  }
  if (tid == 0) lds->args = a;
  if (a.pre_scale != nullptr && a.C <= kBnCap) {
    for (int i = tid; i < a.C; i += kThreads) {
      lds->bn_s[i] = a.pre_scale[i];
      lds->bn_t[i] = a.pre_shift[i];
    }
  }
  lds_barrier();

  const int HW = a.H * a.W;
  const long long M = a.row_elems;
  const bool affine = a.pre_scale != nullptr;
  const bool bn_lds = a.C <= kBnCap;
  unsigned kreg[4 * U];
  unsigned mk = kNoKey, xk = 0u;
  {
    const float4* __restrict__ row4 = reinterpret_cast<const float4*>(xrow);
    const unsigned nvec = (unsigned)(M / 4);
    const unsigned ntrip = (nvec + 2u) / 3u;
    const float hinv = 1.0f / (float)HW;
    // triples in flight per lane (more in flight measured slower: the histogram atomics of a batch overlap the
    // loads of the next one)
    constexpr int B = kS3 ? 9 : 3;                 // (three-stream rows: the same 144 bytes per lane in flight)
    // three-stream rows: a lane's step is ONE float4 of stream 0 -- four entries of one channel's block, keys up to the
    // channel's count (s3_count), a third of the bytes
    const unsigned nvec0 = kS3 ? (unsigned)a.x_s3 / 4u : 0u, hp4 = kS3 ? (unsigned)a.x_hp / 4u : 1u;
    const unsigned hp4_magic = (unsigned)((0x100000000ull + hp4 - 1u) / hp4);      // jt / hp4 = umulhi(jt, magic): exact while jt * hp4 < 2^32
#pragma unroll
    for (int u0 = 0; u0 < U; u0 += B) {
      float4 v[B][kS3 ? 1 : 3];
#pragma unroll
      for (int b = 0; b < B; ++b) {
        if (u0 + b < U) {
          const unsigned jt = (unsigned)tid + (unsigned)(u0 + b) * kThreads;
          if constexpr (kS3) {
            v[b][0] = row4[min(jt, nvec0 - 1u)];
          } else {
#pragma unroll
            for (int t = 0; t < 3; ++t) v[b][t] = row4[min(3u * jt + (unsigned)t, nvec - 1u)];
          }
        }
      }
#pragma unroll
      for (int b = 0; b < B; ++b) {
        if (u0 + b < U) {
          const unsigned jt = (unsigned)tid + (unsigned)(u0 + b) * kThreads;
          const unsigned e0 = 12u * jt;
          float xs[4] = {v[b][0].x, kS3 ? v[b][0].y : v[b][0].w, kS3 ? v[b][0].z : v[b][kS3 ? 0 : 1].z,
                         kS3 ? v[b][0].w : v[b][kS3 ? 0 : 2].y};
          if constexpr (kS3) {
            const unsigned ch = min(__umulhi(jt, hp4_magic), (unsigned)a.C - 1u);
            const unsigned t0 = 4u * (jt - ch * hp4), cnt = s3_count(ch, (unsigned)HW);
            float sc = 1.f, sh = 0.f;
            if (affine) {
              sc = bn_lds ? lds->bn_s[ch] : a.pre_scale[ch];
              sh = bn_lds ? lds->bn_t[ch] : a.pre_shift[ch];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const bool has = jt < nvec0 && t0 + (unsigned)e < cnt;
              const unsigned key = abs_key(clamp_sym(affine ? fmaf(xs[e], sc, sh) : xs[e], a.alpha));
              if (has) {
                if (win) hist_add_win(lds, key, win_k0, win_sh, low_base);
                else hist_add(lds, key);
                mk = min(mk, key);
                xk = max(xk, key);
              }
              kreg[4 * (u0 + b) + e] = has ? key : kNoKey;
            }
            continue;
          }
          if (affine) {
            // channel of element e0 (e0 / HW via a reciprocal, corrected); the other three sub-sampled elements
            // are at most one channel boundary further each (HW >= 4).  Scale / shift come from the LDS copy.
            const unsigned ec = min(e0, (unsigned)(M - 1));
            unsigned c = (unsigned)((float)ec * hinv);
            if ((c + 1u) * (unsigned)HW <= ec) ++c;
            if (c * (unsigned)HW > ec) --c;
            unsigned r = ec - c * (unsigned)HW;        // position inside the channel
            auto bn = [&](unsigned ch, float& sc, float& sh) {      // (uniform branch: LDS copy or global)
              if (bn_lds) {
                sc = lds->bn_s[ch];
                sh = lds->bn_t[ch];
              } else {
                sc = a.pre_scale[ch];
                sh = a.pre_shift[ch];
              }
            };
            if (r + 9u < (unsigned)HW) {               // the usual case: all four in one channel
              float sc, sh;
              bn(c, sc, sh);
#pragma unroll
              for (int e = 0; e < 4; ++e) xs[e] = fmaf(xs[e], sc, sh);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float sc, sh;
                bn(min(c, (unsigned)a.C - 1u), sc, sh);
                xs[e] = fmaf(xs[e], sc, sh);
                r += 3u;
                if (r >= (unsigned)HW) {
                  r -= (unsigned)HW;
                  ++c;
                }
              }
            }
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool has = jt < ntrip && (long long)e0 + 3 * e < M;
            const unsigned key = abs_key(clamp_sym(xs[e], a.alpha));
            if (has) {
              if (win) hist_add_win(lds, key, win_k0, win_sh, low_base);
              else hist_add(lds, key);
              mk = min(mk, key);
              xk = max(xk, key);
            }
            kreg[4 * (u0 + b) + e] = has ? key : kNoKey;
          }
        }
      }
    }
  }
  unsigned minkey = mk, maxkey = xk;
  block_minmax(minkey, maxkey, lds);               // (its barriers also close the histogram)
  FMARK(1);
  const unsigned n = (unsigned)((M + 2) / 3);
  const bool ternary = a.ternary != 0;
  // The first 128 floats of the lane's first pass-2 item are requested as soon as the refinement has copied the keys
  // it needs out of the registers (the usual path; rows that take the block path load them in pass 2 as before).
  const bool full = (a.cg & 63) == 0 && a.C <= kBnCap;
  unsigned long long* __restrict__ prow0 = a.planes + (long long)row * a.row_words;
  unsigned long long* __restrict__ prow1 = prow0 + a.plane_words;
  auto finish_row = [&](double acc, float v1) {
    const double tot = block_sum(acc, lds);
    FMARK(10);
    if (tid == 0) {
      a.scales[row] = v1;
      a.scales[(long long)a.N + row] = ternary ? v1 : (float)(tot / (double)M);
      if (a.trace) a.trace[row] = (int)lds->best_order;
    }
  };
  auto pass2_late = [&](float v1) {              // pass 2 without an early request
    if (full)
      return affine ? pass2_full<VEC, true, false>(a, lds->bn_s, lds->bn_t, xrow, v1, prow0, prow1, tid, kThreads)
                    : pass2_full<VEC, false, false>(a, lds->bn_s, lds->bn_t, xrow, v1, prow0, prow1, tid, kThreads);
    return pass2_any<VEC>(a, xrow, v1, prow0, prow1, tid, kThreads);
  };
  Pf<VEC> pf;
  auto keys_dead = [&](unsigned parts) {
#ifndef LSQ_NO_EARLY
    if (full) pass2_request<VEC>(a, xrow, tid, pf, parts);
#endif
  };
  auto pass2_early = [&](float v1, bool have_pf) {   // pass 2 whose first loads went out during the solve
#ifdef LSQ_NO_EARLY
    have_pf = false;
#endif
    if (full)
      return affine ? pass2_full<VEC, true, true>(a, lds->bn_s, lds->bn_t, xrow, v1, prow0, prow1, tid, kThreads, &pf, have_pf)
                    : pass2_full<VEC, false, true>(a, lds->bn_s, lds->bn_t, xrow, v1, prow0, prow1, tid, kThreads, &pf, have_pf);
    return pass2_any<VEC>(a, xrow, v1, prow0, prow1, tid, kThreads);
  };
  if constexpr (WIN) {
    Best best;
    best.cost = INFINITY;
    best.order = kNoKey;
    best.value = 0.f;
    if (tid == 0) lds->maxkey = maxkey;
    // (readfirstlane: the compiler must SEE that the whole workgroup takes the same side -- a divergent branch becomes two
    //  predicated regions with everything of both sides live across them)
    if (__builtin_amdgcn_readfirstlane((int)win_plan(lds, n, minkey, maxkey, ternary)) != 0) {
      win_sweep<4 * U>(lds, n, maxkey, ternary, kreg, keys_dead, best);
      const float v1 = finish_best(lds, n, minkey, ternary, best);
      FMARK(9);
      finish_row(pass2_early(v1, true), v1);
    } else {
      // The row is the round-2 solve's, from its first load on, as a CALL: in line, its code -- which holds the keys far
      // longer -- raises the register pressure of the whole function and the early request of the windowed path ends up
      // in scratch (every early load then waited for).  The row is read once more, from L2 / Infinity Cache; rows that end
      // here are rare (scripts/win_stats_net.py counts them).
      run_round2<U, VEC>(lds);
    }
  } else {
    bool have_pf = false;
    const float v1 = solve_from_hist<4 * U>(lds, xrow, n, minkey, maxkey, ternary, kreg, keys_dead, [&]() {
#pragma unroll
      for (int b = 0; b < Pf<VEC>::PB; ++b)
#pragma unroll
        for (int u = 0; u < Pf<VEC>::UB; ++u)
#pragma unroll
          for (int v = 0; v < VEC; ++v) pf.v[b][u][v] = 0.f;
    }, have_pf);
    FMARK(9);
    finish_row(pass2_early(v1, have_pf), v1);
  }
}
// the round-2 kernel body as a function of its own (the windowed kernels' fall-back, see run)
template <int U, int VEC>
static __device__ __attribute__((noinline)) void run_round2(FusedLds* lds) {
  const FusedArgs a = lds->args;                 // (the kernel's own argument block must not escape: it would live in scratch)
  lds_barrier();                                 // every wave has left the windowed tables (and read the arguments)
  run<U, VEC, false>(a, lds);
}
};  // struct Impl

// WIN: windowed level-1 histogram (solve_windowed) with the round-2 solve as the fall-back of single rows; !WIN: the
// round-2 solve alone (rows without a symmetric clamp, lsq_debug_fused_mode 4)
template <int T, int U, int VEC, bool WIN>
__global__ __launch_bounds__(T) void aq_fused_kernel(FusedArgs a) {
  using I = Impl<T>;
  __shared__ __attribute__((aligned(16))) unsigned char smem[sizeof(typename I::FusedLds)];
  I::template run<U, VEC, WIN>(a, reinterpret_cast<typename I::FusedLds*>(smem));
}

// Scales given by the caller (moving-average inference modes, activation_quantization.py:90-98: the deployment
// configuration -- no solve at inference): pass 2 alone, ONE read of the input for both planes (the streaming
// sweeps read it once per plane).  No row reduction is left, so the grid is decoupled from N: blockIdx.y = row,
// blockIdx.x = one of `parts` interleaved shares of the row's items.
template <int T, int VEC>
__global__ __launch_bounds__(T) void aq_forced_kernel(FusedArgs a) {
  using I = Impl<T>;
  __shared__ float bn_s[kBnCap], bn_t[kBnCap];
  const int tid = threadIdx.x, row = blockIdx.y;
  const bool stage = a.pre_scale != nullptr && a.C <= kBnCap;
  if (stage) {
    for (int i = tid; i < a.C; i += T) {
      bn_s[i] = a.pre_scale[i];
      bn_t[i] = a.pre_shift[i];
    }
    lds_barrier();
  }
  const float* __restrict__ xrow = a.x + (long long)row * a.row_elems;
  unsigned long long* __restrict__ prow0 = a.planes + (long long)row * a.row_words;
  unsigned long long* __restrict__ prow1 = prow0 + a.plane_words;
  const float v1 = a.forced[row];
  const int item0 = (int)blockIdx.x * T + tid, step = (int)gridDim.x * T;
  if ((a.cg & 63) == 0 && a.C <= kBnCap) {
    if (a.pre_scale != nullptr) I::template pass2_full<VEC, true>(a, bn_s, bn_t, xrow, v1, prow0, prow1, item0, step);
    else I::template pass2_full<VEC, false>(a, bn_s, bn_t, xrow, v1, prow0, prow1, item0, step);
  } else {
    I::template pass2_any<VEC>(a, xrow, v1, prow0, prow1, item0, step);
  }
  if (blockIdx.x == 0 && tid == 0) {
    a.scales[row] = v1;
    a.scales[(long long)a.N + row] = a.forced[(long long)a.N + row];
  }
}

// Greedy 2-bit scheme (gf-2, quantization.py:118-148): v1 = mean |x| of the row, then exactly the planes and the second
// scale of the least-squares 2-bit scheme.  One launch, one workgroup per row: a flat coalesced sweep sums |x| (fp64 per
// lane), pass 2 follows from the same workgroup -- its read is served by the L2 / Infinity Cache the first one filled.
// (The streaming path takes two launches of 1024-thread sweeps.)
template <int T, int VEC>
__global__ __launch_bounds__(T) void aq_greedy2_kernel(FusedArgs a) {
  using I = Impl<T>;
  __shared__ float bn_s[kBnCap], bn_t[kBnCap];
  __shared__ double red[T / 64];
  const int tid = threadIdx.x, row = blockIdx.x;
  const bool affine = a.pre_scale != nullptr;
  if (affine && a.C <= kBnCap) {
    for (int i = tid; i < a.C; i += T) {
      bn_s[i] = a.pre_scale[i];
      bn_t[i] = a.pre_shift[i];
    }
  }
  lds_barrier();
  const float* __restrict__ xrow = a.x + (long long)row * a.row_elems;
  const long long M = a.row_elems;
  const int HW = a.H * a.W;
  auto block_total = [&](double v) {
    v = wave_sum(v);
    lds_barrier();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    lds_barrier();
    double t = 0.0;
    for (int w = 0; w < T / 64; ++w) t += red[w];
    return t;
  };
  // sum |clamp(bn(x))| over the whole row, flat and coalesced
  double acc = 0.0;
  {
    const float4* __restrict__ row4 = reinterpret_cast<const float4*>(xrow);
    const unsigned nvec = (unsigned)(M / 4);
    const float hinv = 1.0f / (float)HW;
    constexpr int B = 8;                          // independent loads in flight per lane
    for (unsigned i0 = (unsigned)tid; i0 < nvec; i0 += B * T) {
      float4 v[B];
#pragma unroll
      for (int b = 0; b < B; ++b) v[b] = row4[min(i0 + (unsigned)b * T, nvec - 1u)];
#pragma unroll
      for (int b = 0; b < B; ++b) {
        const unsigned i = i0 + (unsigned)b * T;
        if (i < nvec) {
          float xs[4] = {v[b].x, v[b].y, v[b].z, v[b].w};
          if (affine) {
            const unsigned e = 4u * i;
            unsigned c = (unsigned)((float)e * hinv);
            if ((c + 1u) * (unsigned)HW <= e) ++c;
            if (c * (unsigned)HW > e) --c;
            const unsigned r = e - c * (unsigned)HW;            // position inside the channel; HW >= 4: at most one boundary
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const unsigned ck = min(c + (r + (unsigned)k >= (unsigned)HW ? 1u : 0u), (unsigned)a.C - 1u);
              const float sc = a.C <= kBnCap ? bn_s[ck] : a.pre_scale[ck];
              const float sh = a.C <= kBnCap ? bn_t[ck] : a.pre_shift[ck];
              xs[k] = fmaf(xs[k], sc, sh);
            }
          }
          const float x0 = clamp_sym(xs[0], a.alpha), x1 = clamp_sym(xs[1], a.alpha);
          const float x2 = clamp_sym(xs[2], a.alpha), x3 = clamp_sym(xs[3], a.alpha);
          acc += (double)((fabsf(x0) + fabsf(x1)) + (fabsf(x2) + fabsf(x3)));
        }
      }
    }
  }
  const float v1 = (float)(block_total(acc) / (double)M);
  unsigned long long* __restrict__ prow0 = a.planes + (long long)row * a.row_words;
  unsigned long long* __restrict__ prow1 = prow0 + a.plane_words;
  double acc2;
  if ((a.cg & 63) == 0 && a.C <= kBnCap) {
    acc2 = affine ? I::template pass2_full<VEC, true>(a, bn_s, bn_t, xrow, v1, prow0, prow1, tid, T)
                  : I::template pass2_full<VEC, false>(a, bn_s, bn_t, xrow, v1, prow0, prow1, tid, T);
  } else {
    acc2 = I::template pass2_any<VEC>(a, xrow, v1, prow0, prow1, tid, T);
  }
  const double tot2 = block_total(acc2);
  if (tid == 0) {
    a.scales[row] = v1;
    a.scales[(long long)a.N + row] = (float)(tot2 / (double)M);
  }
}

#if !LSQ_FUSED_S3
template <int T>
int launch_greedy(const FusedArgs& a, int vec, hipStream_t st) {
  if (vec == 4) hipLaunchKernelGGL((aq_greedy2_kernel<T, 4>), dim3(a.N), dim3(T), 0, st, a);
  else if (vec == 2) hipLaunchKernelGGL((aq_greedy2_kernel<T, 2>), dim3(a.N), dim3(T), 0, st, a);
  else hipLaunchKernelGGL((aq_greedy2_kernel<T, 1>), dim3(a.N), dim3(T), 0, st, a);
  return (int)hipGetLastError();
}

template <int T>
int launch_forced(const FusedArgs& a, int vec, hipStream_t st) {
  const long long HW = (long long)a.H * a.W;
  const long long items = (long long)a.Gt * ((HW + vec - 1) / vec);
  // shares per row: every lane an item if there are that many, about four workgroups per CU over the batch
  long long parts = (items + T - 1) / T;
  const long long want = (1024 + a.N - 1) / a.N;
  if (parts > want) parts = want;
  if (parts < 1) parts = 1;
  const dim3 grid((unsigned)parts, (unsigned)a.N);
  if (vec == 4) hipLaunchKernelGGL((aq_forced_kernel<T, 4>), grid, dim3(T), 0, st, a);
  else if (vec == 2) hipLaunchKernelGGL((aq_forced_kernel<T, 2>), grid, dim3(T), 0, st, a);
  else hipLaunchKernelGGL((aq_forced_kernel<T, 1>), grid, dim3(T), 0, st, a);
  return (int)hipGetLastError();
}
#endif

template <int T, int U>
int launch(const FusedArgs& a, int vec, hipStream_t st) {
#ifdef LSQ_DEV_VEC      // developer builds: one instantiation only (register-pressure experiments compile in seconds)
#ifdef LSQ_DEV_NOWIN
  hipLaunchKernelGGL((aq_fused_kernel<T, U, LSQ_DEV_VEC, false>), dim3(a.N), dim3(T), 0, st, a);
#else
  hipLaunchKernelGGL((aq_fused_kernel<T, U, LSQ_DEV_VEC, true>), dim3(a.N), dim3(T), 0, st, a);
#endif
#else
  if (a.win_sh) {
    if (vec == 4) hipLaunchKernelGGL((aq_fused_kernel<T, U, 4, true>), dim3(a.N), dim3(T), 0, st, a);
    else if (vec == 2) hipLaunchKernelGGL((aq_fused_kernel<T, U, 2, true>), dim3(a.N), dim3(T), 0, st, a);
    else hipLaunchKernelGGL((aq_fused_kernel<T, U, 1, true>), dim3(a.N), dim3(T), 0, st, a);
  } else {
#if !LSQ_FUSED_S3
    if (vec == 4) hipLaunchKernelGGL((aq_fused_kernel<T, U, 4, false>), dim3(a.N), dim3(T), 0, st, a);
    else if (vec == 2) hipLaunchKernelGGL((aq_fused_kernel<T, U, 2, false>), dim3(a.N), dim3(T), 0, st, a);
    else hipLaunchKernelGGL((aq_fused_kernel<T, U, 1, false>), dim3(a.N), dim3(T), 0, st, a);
#endif
  }
#endif
  return (int)hipGetLastError();
}

}  // namespace

#if LSQ_FUSED_S3
int fused_act_quant_s3(const FusedArgs& a_in, hipStream_t st) {
  FusedArgs a = a_in;
  // three-stream rows: the solving kernels only (given scales and gf-2 read the row once as it is); H W = 3 h + 1, streams
  // of S floats (a multiple of 4, at least ceil(M / 3) + 8: pass 2's last items read a few floats past a stream's end)
  if (a.forced || a.greedy || a.x_s3 <= 0 || a.x_hp <= 0 || a.x_hp % 32 || a.x_s3 != (long long)a.C * a.x_hp ||
      ((long long)a.H * a.W) % 3 != 1 || a.x_hp < ((long long)a.H * a.W + 2) / 3 + 3 || (a.cg & 63) != 0 || a.C > kBnCap)
    return kFusedNotEligible;
#else
int fused_act_quant(const FusedArgs& a_in, hipStream_t st) {
  FusedArgs a = a_in;
  if (a.x_s3) return kFusedNotEligible;
#endif
  const long long HW = (long long)a.H * a.W;
  const long long M = a.row_elems;
  // windowed level-1 histogram of the solve (solve_windowed): under a symmetric clamp every key is at most key(alpha), so the
  // 8192 bins cover the binades below the top of alpha's binade -- 1024 bins per binade over 8 binades for long rows (a
  // flagged bin must hold at most 64 keys), 512 over 16 for short ones; k0 = 0 when alpha's binade is that low already
  a.win_k0 = 0u;
  a.win_sh = 0;
  if (a.alpha > 0.f && a.alpha < INFINITY && !(a.debug & 4) && !a.forced && !a.greedy) {
    unsigned ka;
    memcpy(&ka, &a.alpha, 4);
    const int sh = (M + 2) / 3 >= 16384 ? 13 : 14;
    const unsigned long long ktop = (unsigned long long)((ka >> 23) + 1u) << 23, span = (unsigned long long)L1_BINS << sh;
    a.win_k0 = ktop > span ? (unsigned)(ktop - span) : 0u;
    a.win_sh = sh;
  }
  if (M + 2 >= (1ll << 31) || M % 4 != 0 || HW < 4 || ((uintptr_t)a.x % 16) != 0) return kFusedNotEligible;
  constexpr int T = 512;
  // pass 2, pixels per lane (measured on the four ResNet-18 shapes, scripts/kbench.py under LSQ_FUSED_VEC): four
  // when that is a single round of items (every load 16 bytes; idle lanes cost less than 4-byte loads, each a full
  // address pass of the load path) or many rounds, two for the 1..4 rounds in between (56 x 56 x 64: 784 four-pixel
  // items on 512 lanes are two rounds with the second half empty)
  int vec = 1;
  const long long items4 = kS3 ? (long long)a.Gt * 3 * ((HW + 11) / 12) : (long long)a.Gt * (HW / 4);
  if ((kS3 || HW % 4 == 0) && (items4 <= T || items4 >= 4 * T)) vec = 4;
  else if (kS3 || HW % 2 == 0) vec = 2;
  if (kS3 && (long long)a.Gt * 3 * ((HW + 2) / 3) <= T) vec = 1;      // (the 7 x 7 rows: one pixel per lane, as the NCHW kernels)
#ifdef LSQ_TUNE
  if (const char* e = getenv("LSQ_FUSED_VEC")) {
    const int v = atoi(e);
    if ((v == 4 || v == 2 || v == 1) && HW % v == 0) vec = v;
  }
#endif
#if !LSQ_FUSED_S3
  if (a.forced) {
    if (a.N > 65535) return kFusedNotEligible;       // (grid y)
    // no rounds to balance here: the widest loads the image allows
    return launch_forced<T>(a, HW % 4 == 0 ? 4 : (HW % 2 == 0 ? 2 : 1), st);
  }
  if (a.greedy) return launch_greedy<T>(a, vec, st);
#else
  if (!a.win_sh) return kFusedNotEligible;           // (rows under a symmetric clamp: the windowed kernels, with their fall-back)
#endif
  const long long ntrip = kS3 ? a.x_s3 / 4 : (M / 4 + 2) / 3;
  const long long need = (ntrip + T - 1) / T;        // triples (4 keys each) per lane
#ifdef LSQ_DEV_U
  return need <= LSQ_DEV_U ? launch<T, LSQ_DEV_U>(a, vec, st) : kFusedNotEligible;
#else
#if LSQ_FUSED_S3
  if (need <= 18) return launch<T, 18>(a, vec, st);     // (28 x 28 x 128: 128 blocks of 288 floats = 18 float4 per lane)
  if (need <= 33) return launch<T, 33>(a, vec, st);
  return kFusedNotEligible;
#else
  if (need <= 5) return launch<T, 5>(a, vec, st);
  if (need <= 9) return launch<T, 9>(a, vec, st);
  if (need <= 17) return launch<T, 17>(a, vec, st);
  if (need <= 33) return launch<T, 33>(a, vec, st);
  return kFusedNotEligible;
#endif
#endif
}

#if defined(LSQ_PHASE_CLOCKS)
extern "C" int lsq_debug_read_fused_times(long long* host16384) {
  return (int)hipMemcpyFromSymbol(host16384, HIP_SYMBOL(g_fused_times), 16384 * sizeof(long long));
}
extern "C" int lsq_debug_read_win_stats(int* host4096) {
  return (int)hipMemcpyFromSymbol(host4096, HIP_SYMBOL(g_win_stats), 4096 * sizeof(int));
}
#endif

}  // namespace lsq
