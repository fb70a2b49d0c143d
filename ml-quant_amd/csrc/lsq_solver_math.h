// Solver arithmetic shared by the activation-quantizer kernels (lsq_act_quant.hip: streaming three-kernel
// path; lsq_act_fused.hip: single-launch path with the sub-sample resident on chip).  Exact rank / prefix-sum
// bookkeeping over histogram bins of the IEEE bit pattern of |x|, the candidate tests of
// quant/binary/optimal.py:66-80 and the closed-form least-squares cost of optimal.py:31-38.
#pragma once

#include "lsq_common.h"

namespace lsq {

constexpr int L1_SHIFT = 18, L1_BINS = 8192;   // key bits [30:18]: 32 bins per binade
constexpr unsigned kNoKey = 0xFFFFFFFFu;
constexpr unsigned long long kOne = 1ull << 42;          // count field of a histogram word
constexpr unsigned long long kLowMask = kOne - 1;
constexpr double kSlack = 1e-9;

struct Best {
  double cost;
  unsigned order;
  float value;
};

__device__ __forceinline__ bool better(const Best& a, const Best& b) {
  return a.cost < b.cost || (a.cost == b.cost && a.order < b.order);
}

struct Slot1 {
  unsigned short bin, next_bin;   // next_bin = 0xFFFF: none
  unsigned cnt, r0, succ;         // succ: smallest key above the bin (kNoKey until known)
  unsigned base;                  // first key of the slot's segment in the LDS list
  unsigned pad;
  double p0, sum;
};

// exact sum of a histogram bin whose keys share `hi_key` above the low bits
__device__ __forceinline__ double bin_sum_exact(unsigned hi_key, unsigned cnt, unsigned long long lowsum) {
  const int e = (int)(hi_key >> 23);
  long long mant = (long long)(hi_key & 0x7FFFFFu);
  int sc = -149;
  if (e > 0) {
    mant += 1ll << 23;
    sc = e - 150;
  }
  const long long integer = (long long)cnt * mant + (long long)lowsum;
  // integer < 2^47 is exact in fp64; multiply by the exact power of two 2^sc (sc >= -149)
  return (double)integer * __longlong_as_double((long long)(sc + 1023) << 52);
}

struct MPair {
  double m2, m1;
};
__device__ __forceinline__ MPair m_pair(double lo_cnt, double lo_sum, double n, double total) {
  const double hi_mean = (total - lo_sum) / (n - lo_cnt);
  MPair r;
  r.m2 = 0.5 * hi_mean;                          // optimal.py:74
  r.m1 = 0.5 * (lo_sum / lo_cnt + hi_mean);      // optimal.py:73
  return r;
}

// a / b for the conservative bin tests only: hardware reciprocal + one Newton step (relative error ~1e-15,
// far inside kSlack) instead of the ~45-instruction IEEE division; b >= 1 here.
__device__ __forceinline__ double quick_div(double a, double b) {
  double r = __builtin_amdgcn_rcp(b);
  r = fma(fma(-b, r, 1.0), r, r);
  return a * r;
}
__device__ __forceinline__ MPair m_pair_quick(double lo_cnt, double lo_sum, double n, double total) {
  const double hi_mean = quick_div(total - lo_sum, n - lo_cnt);
  MPair r;
  r.m2 = 0.5 * hi_mean;
  r.m1 = 0.5 * (quick_div(lo_sum, lo_cnt) + hi_mean);
  return r;
}

// Can a position inside [r0, r0+cnt) be a candidate (optimal.py:78-80)?  Conservative, never misses one.
// The keys of the bin lie in [vlo, vhi]; the first key above it is at most next_hi.  m2(i) and m1(i) grow with i.
// A position i whose successor is in the same bin needs vlo <= m(i) <= vhi, and m over the bin's positions lies between
// m_lo (taken one position in front of the bin) and m_hi (at the bin's last position): m_hi >= vlo and m_lo <= vhi.
// The bin's LAST position has its successor above the bin: vlo <= m_hi <= next_hi.  (Until round 4 the test was
// m_hi >= vlo and m_lo <= next_hi -- it let every bin within one bin width below a crossing through, twelve flagged
// level-1 bins for the two that hold the candidates of a Gaussian row.)
__device__ inline bool may_hold_candidate(unsigned r0, unsigned cnt, double p0, double s, double vlo, double vhi,
                                   double next_hi, unsigned n, double total, bool ternary) {
  const long long r1 = (long long)r0 + cnt;
  const long long ilo = r0 > 1u ? (long long)r0 : 1ll;
  const long long ihi = (r1 - 1) < ((long long)n - 2) ? (r1 - 1) : ((long long)n - 2);
  if (ilo > ihi) return false;
  double m2_lo, m1_lo, m2_hi, m1_hi;
  if (r0 >= 1u) {
    const MPair m = m_pair_quick((double)r0, p0, (double)n, total);
    m2_lo = m.m2;
    m1_lo = m.m1;
  } else {
    m2_lo = m1_lo = 0.5 * quick_div(total - vhi, (double)n - 1.0);
  }
  const bool has_last = r1 <= (long long)n - 1;    // the bin's last position is an inner position of the row
  if (has_last) {
    const MPair m = m_pair_quick((double)r1, p0 + s, (double)n, total);
    m2_hi = m.m2;
    m1_hi = m.m1;
  } else {
    // the bin holds the row's largest key: the last inner position is n - 2, its upper tail is that one key (<= vhi)
    // and its lower head sums to total - (largest key) <= total - vlo
    m2_hi = 0.5 * vhi;
    m1_hi = 0.5 * (quick_div(total - vlo, (double)n - 1.0) + vhi);
  }
  const double up = 1.0 + kSlack, dn = 1.0 - kSlack;
  bool hit = (m2_hi * up >= vlo) && ((m2_lo * dn <= vhi) || (has_last && m2_hi * dn <= next_hi));
  if (!ternary) hit = hit || ((m1_hi * up >= vlo) && ((m1_lo * dn <= vhi) || (has_last && m1_hi * dn <= next_hi)));
  return hit;
}

__device__ __forceinline__ bool position_is_candidate(double v, double nxt, double lo_cnt, double lo_sum,
                                                      double n, double total, bool ternary) {
  const MPair m = m_pair(lo_cnt, lo_sum, n, total);
  bool hit = (v <= m.m2) && (m.m2 <= nxt);
  if (!ternary) hit = hit || ((v <= m.m1) && (m.m1 <= nxt));
  return hit;
}

// A run of `c` equal keys of value v at sorted positions [r0, r0+c): is any position a candidate?
__device__ inline bool run_has_candidate(double v, unsigned c, unsigned r0, double p0, double succ_v, unsigned n,
                                  double total, bool ternary) {
  const double dn = (double)n;
  // last element of the run: successor is the next distinct value
  {
    const long long i = (long long)r0 + c - 1;
    if (i >= 1 && i <= (long long)n - 2 &&
        position_is_candidate(v, succ_v, (double)(i + 1), p0 + (double)c * v, dn, total, ternary))
      return true;
  }
  if (c < 2u) return false;
  // interior positions: a[i] == a[i+1] == v, so m must equal v exactly
  long long tlo = 0, thi = (long long)c - 2;
  if ((long long)r0 + tlo < 1) tlo = 1 - (long long)r0;
  if ((long long)r0 + thi > (long long)n - 2) thi = (long long)n - 2 - (long long)r0;
  if (tlo > thi) return false;
  if (thi - tlo < 8) {
    for (long long t = tlo; t <= thi; ++t)
      if (position_is_candidate(v, v, (double)(r0 + t + 1), p0 + (double)(t + 1) * v, dn, total, ternary))
        return true;
    return false;
  }
  // Longer runs: m2(t), m1(t) are monotone in t, so a run holds a candidate exactly when the FIRST t with m >= v
  // has m == v.  With lo_cnt = k = r0 + t + 1 and lo_sum = A + k v (A = p0 - r0 v) both crossings are roots of
  // linear equations in k:  m2 = v  <=>  k = (2 v n + A - total) / v,   m1 = v  <=>  k = A n / (v n + 2 A - total).
  // The root brackets the first t to a window of four positions, which is then searched with the exact predicate
  // (the same expressions as m_pair); a window that does not bracket (degenerate slopes: v = 0, all-equal rows)
  // falls back to the bisection over the whole run.  Same answer as the bisection, a third of its divisions.
  const double A = p0 - (double)r0 * v;
  for (int which = 0; which < (ternary ? 1 : 2); ++which) {
    auto m_at = [&](long long t) {
      const double lo_cnt = (double)(r0 + t + 1), lo_sum = p0 + (double)(t + 1) * v;
      const double hi_mean = (total - lo_sum) / (dn - lo_cnt);
      return which ? 0.5 * (lo_sum / lo_cnt + hi_mean) : 0.5 * hi_mean;
    };
    long long lo = tlo, hi = thi;
    const double kstar = which ? (A * dn) / (v * dn + 2.0 * A - total) : (2.0 * v * dn + A - total) / v;
    if (kstar == kstar && kstar > -1e18 && kstar < 1e18) {
      const long long tc = (long long)floor(kstar) - (long long)r0 - 1;
      const long long wlo = tc - 1 < tlo ? tlo : (tc - 1 > thi ? thi : tc - 1);
      const long long whi = tc + 2 < tlo ? tlo : (tc + 2 > thi ? thi : tc + 2);
      if ((wlo == tlo || m_at(wlo - 1) < v) && (whi == thi || m_at(whi) >= v)) {
        lo = wlo;
        hi = whi;
      }
    }
    while (lo < hi) {
      const long long mid = (lo + hi) >> 1;
      if (m_at(mid) >= v) hi = mid; else lo = mid + 1;
    }
    if (m_at(lo) == v) return true;
  }
  return false;
}

// closed-form cost^2 (minus the constant sum a^2) of candidate v (optimal.py:31-38)
__device__ __forceinline__ double cost_of(double v, unsigned below_cnt, double below_sum, unsigned eq_cnt,
                                          unsigned n, double total, bool ternary) {
  const double dn = (double)n;
  const double above_cnt = dn - (double)below_cnt - (double)eq_cnt;
  const double above_sum = total - below_sum - (double)eq_cnt * v;
  const double dev = (v * (double)below_cnt - below_sum) + (above_sum - v * above_cnt);
  const double quad = -2.0 * v * total + dn * v * v;
  if (ternary) return quad - 2.0 * v * dev + dn * v * v;
  return quad - dev * dev / dn;
}

}  // namespace lsq
