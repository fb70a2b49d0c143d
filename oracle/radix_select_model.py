"""Host model (numpy) of the sort-free idea behind the GPU optimal-v1 solver -- TEST INFRASTRUCTURE ONLY.

The HIP solvers (ml-quant_amd/csrc/lsq_act_fused.hip, lsq_act_quant.hip, shared math in
lsq_solver_math.h) never sort a row.  They find the reference's candidate positions
(quant/binary/optimal.py:55-83) with a radix select over the IEEE-754 bit pattern of |x| in
which every histogram bin carries an *exact integer* sum of its elements, so rank and prefix sum
at any bin boundary are exact; only bins that can contain a crossing of the monotone functions
m1(i), m2(i) with the sorted sequence are refined.  This file states that idea step by step with a
fixed three-level split (12 + 10 + 9 key bits) so that its logic (conservative bin test, successor
handling, runs of equal keys, edge cases) can be checked on the CPU against ``oracle/lsq_exact.py``.

It is NOT a transcription of the shipped kernels: those use a 13-bit first level, 6-bit refinement
rounds over an LDS key list, brute-force ranking of small sub-bins and a closed-form test for
runs; their results are checked bit for bit against ``oracle/lsq_exact.py`` on the device
(tests/test_gpu_parity.py), not against this model.
"""

from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np

L1_SHIFT, L2_SHIFT = 19, 9          # 12 + 10 + 9 key bits (this model's split)
L2_BITS, L3_BITS = 10, 9
KEY_INF = 0xFFFFFFFF
SLACK = 1e-9


def keys_of(abs_vals: np.ndarray) -> np.ndarray:
    return np.asarray(abs_vals, dtype=np.float32).view(np.uint32).astype(np.uint64)


def key_value(key: int) -> float:
    return float(np.array([key & 0x7FFFFFFF], dtype=np.uint32).view(np.float32)[0])


def bin_sum(level_shift: int, prefix: int, cnt: np.ndarray, lowsum: np.ndarray, nbits: int) -> np.ndarray:
    """Exact fp64 sum of the elements of each bin of one histogram.

    All keys of a bin share the bits above ``level_shift``; a key is
    ((prefix << nbits | bin) << level_shift) | low.  For a fixed exponent field the value
    is linear in the mantissa, so sum = 2^(e-150) * (cnt * mant_hi + sum(low)).
    """
    bins = np.arange(cnt.shape[0], dtype=np.uint64)
    hi_key = ((np.uint64(prefix) << np.uint64(nbits)) | bins) << np.uint64(level_shift)
    expo = (hi_key >> np.uint64(23)).astype(np.int64)
    mant_hi = (hi_key & np.uint64(0x7FFFFF)).astype(np.int64)
    normal = expo > 0
    mant_hi = np.where(normal, mant_hi + (1 << 23), mant_hi)
    scale_e = np.where(normal, expo - 150, -149)
    integer = cnt.astype(np.int64) * mant_hi + lowsum.astype(np.int64)
    return np.ldexp(integer.astype(np.float64), scale_e.astype(np.int64))


def m_pair(lo_cnt: float, lo_sum: float, n: int, total: float) -> Tuple[float, float]:
    """(m2, m1) at a position whose inclusive prefix has lo_cnt elements summing to lo_sum."""
    hi_cnt = n - lo_cnt
    hi_mean = (total - lo_sum) / hi_cnt
    return 0.5 * hi_mean, 0.5 * (lo_sum / lo_cnt + hi_mean)


def may_hold_candidate(r0: int, cnt: int, p0: float, s: float, vlo: float, vhi: float,
                       next_hi: float, n: int, total: float, ternary: bool) -> bool:
    """Conservative test: can a position i in [r0, r0+cnt) be a candidate?

    m1 and m2 are non-decreasing in i.  The keys of the bin lie in [vlo, vhi]; ``next_hi`` bounds the first key above
    the bin.  A position whose successor sits in the same bin needs vlo <= m(i) <= vhi, and m over the bin's positions
    lies between its value one position in front of the bin (m_lo) and at the bin's last position (m_hi); the bin's
    last position has its successor above the bin and needs vlo <= m_hi <= next_hi.  (Same test as
    ml-quant_amd/csrc/lsq_solver_math.h since round 4.)
    """
    r1 = r0 + cnt
    if max(r0, 1) > min(r1 - 1, n - 2):
        return False
    if r0 >= 1:
        m2_lo, m1_lo = m_pair(r0, p0, n, total)
    else:
        m2_lo = m1_lo = 0.5 * (total - vhi) / (n - 1)
    has_last = r1 <= n - 1
    if has_last:
        m2_hi, m1_hi = m_pair(r1, p0 + s, n, total)
    else:
        m2_hi, m1_hi = 0.5 * vhi, 0.5 * ((total - vlo) / (n - 1) + vhi)
    up, dn = 1.0 + SLACK, 1.0 - SLACK
    hit = (m2_hi * up >= vlo) and ((m2_lo * dn <= vhi) or (has_last and m2_hi * dn <= next_hi))
    if not ternary:
        hit = hit or ((m1_hi * up >= vlo) and ((m1_lo * dn <= vhi) or (has_last and m1_hi * dn <= next_hi)))
    return hit


def _level_hist(keys: np.ndarray, shift: int, nbits: int, prefix: Optional[int]):
    """Histogram of keys whose bits above (shift+nbits) equal prefix: counts, low-bit sums, and
    the smallest key beyond the segment (the segment's successor)."""
    nb = 1 << nbits
    top = keys >> np.uint64(shift + nbits)
    if prefix is None:
        sel = np.ones(keys.shape, dtype=bool)
        beyond = np.zeros(keys.shape, dtype=bool)
    else:
        sel = top == np.uint64(prefix)
        beyond = top > np.uint64(prefix)
    k = keys[sel]
    bins = ((k >> np.uint64(shift)) & np.uint64(nb - 1)).astype(np.int64)
    low = (k & np.uint64((1 << shift) - 1)).astype(np.int64)
    cnt = np.bincount(bins, minlength=nb).astype(np.int64)
    lowsum = np.bincount(bins, weights=low.astype(np.float64), minlength=nb).astype(np.int64)
    succ = int(keys[beyond].min()) if beyond.any() else KEY_INF
    return cnt, lowsum, succ


def solve_row_model(row: np.ndarray, ternary: bool, skip: int = 1,
                    stats: Optional[Dict] = None) -> np.float32:
    a = np.abs(np.asarray(row, dtype=np.float32).reshape(-1)[::skip])
    keys = keys_of(a)
    n = int(keys.shape[0])
    passes = 0
    best = (np.inf, -1, np.float32(0.0))          # (cost, order, value)
    cands: List[Tuple[int, float]] = []

    cnt1, low1, _ = _level_hist(keys, L1_SHIFT, 12, None)
    sum1 = bin_sum(L1_SHIFT, 0, cnt1, low1, 12)
    total = float(np.sum(sum1))                    # (the kernel scans in bin order)
    sq = None

    def cost_of(v: float, below_cnt: int, below_sum: float, eq_cnt: int) -> float:
        above_cnt = n - below_cnt - eq_cnt
        above_sum = total - below_sum - eq_cnt * v
        dev = (v * below_cnt - below_sum) + (above_sum - v * above_cnt)
        quad = -2.0 * v * total + n * v * v        # + sum a^2, constant over candidates
        if ternary:
            return quad - 2.0 * v * dev + n * v * v
        return quad - dev * dev / n

    def consider(v: float, order: int, below_cnt: int, below_sum: float, eq_cnt: int):
        nonlocal best
        c = cost_of(v, below_cnt, below_sum, eq_cnt)
        cands.append((order, v))
        if (c, order) < (best[0], best[1]):
            best = (c, order, np.float32(v))

    if n >= 3:
        pre_c1 = np.concatenate([[0], np.cumsum(cnt1)[:-1]])
        pre_s1 = np.concatenate([[0.0], np.cumsum(sum1)[:-1]])
        nz1 = np.nonzero(cnt1)[0]
        for idx, b in enumerate(nz1):
            vlo = key_value(int(b) << L1_SHIFT)
            vhi = key_value((int(b) << L1_SHIFT) | ((1 << L1_SHIFT) - 1))
            nb = nz1[idx + 1] if idx + 1 < len(nz1) else None
            next_hi = key_value((int(nb) << L1_SHIFT) | ((1 << L1_SHIFT) - 1)) if nb is not None else vhi
            if not may_hold_candidate(int(pre_c1[b]), int(cnt1[b]), float(pre_s1[b]), float(sum1[b]),
                                      vlo, vhi, next_hi, n, total, ternary):
                continue
            # ---- level 2 over bin b
            passes += 1
            cnt2, low2, succ_b = _level_hist(keys, L2_SHIFT, L2_BITS, int(b))
            sum2 = bin_sum(L2_SHIFT, int(b), cnt2, low2, L2_BITS)
            pre_c2 = pre_c1[b] + np.concatenate([[0], np.cumsum(cnt2)[:-1]])
            pre_s2 = pre_s1[b] + np.concatenate([[0.0], np.cumsum(sum2)[:-1]])
            nz2 = np.nonzero(cnt2)[0]
            for j, s in enumerate(nz2):
                base = (int(b) << L2_BITS | int(s)) << L2_SHIFT
                vlo2 = key_value(base)
                vhi2 = key_value(base | ((1 << L2_SHIFT) - 1))
                if j + 1 < len(nz2):
                    nbase = (int(b) << L2_BITS | int(nz2[j + 1])) << L2_SHIFT
                    next_hi2 = key_value(nbase | ((1 << L2_SHIFT) - 1))
                else:
                    next_hi2 = key_value(succ_b) if succ_b != KEY_INF else vhi2
                if not may_hold_candidate(int(pre_c2[s]), int(cnt2[s]), float(pre_s2[s]), float(sum2[s]),
                                          vlo2, vhi2, next_hi2, n, total, ternary):
                    continue
                # ---- level 3: single keys of sub-bin (b, s)
                passes += 1
                pref3 = (int(b) << L2_BITS) | int(s)
                cnt3, _, succ_s = _level_hist(keys, 0, L3_BITS, pref3)
                run_r0 = int(pre_c2[s])
                run_p0 = float(pre_s2[s])
                nz3 = np.nonzero(cnt3)[0]
                for q, kk in enumerate(nz3):
                    key = (pref3 << L3_BITS) | int(kk)
                    v = key_value(key)
                    c = int(cnt3[kk])
                    if q + 1 < len(nz3):
                        succ_v = key_value((pref3 << L3_BITS) | int(nz3[q + 1]))
                    else:
                        succ_v = key_value(succ_s) if succ_s != KEY_INF else np.inf
                    hit = False
                    for t in range(c):              # (the kernel spreads t over lanes)
                        i = run_r0 + t
                        if i < 1 or i > n - 2:
                            continue
                        m2, m1 = m_pair(i + 1, run_p0 + (t + 1) * v, n, total)
                        nxt = v if t < c - 1 else succ_v
                        if (v <= m2 <= nxt) or ((not ternary) and (v <= m1 <= nxt)):
                            hit = True
                            break
                    if hit:
                        consider(v, run_r0, run_r0, run_p0, c)
                    run_r0 += c
                    run_p0 += c * v
    if ternary and n > 0:
        amin = key_value(int(keys.min()))
        mean = total / n
        if amin > 0.5 * mean:
            v = float(np.float32(np.float32(mean)) ) / 2
            v = float(np.float32(v))
            below = keys < keys_of(np.array([v], dtype=np.float32))[0]
            eq = keys == keys_of(np.array([v], dtype=np.float32))[0]
            consider(v, n + 1, int(below.sum()), float(a[below].astype(np.float64).sum()), int(eq.sum()))
    if stats is not None:
        stats.update(passes=passes, candidates=cands, n=n)
    return best[2]
