"""Exact-arithmetic (fp64) per-row restatement of the optimal-v1 solve -- TEST INFRASTRUCTURE ONLY.

The reference decides between near-tied candidates with whatever fp32 rounding its
``torch.norm``/``mean`` reductions produce (hard part 1).  This module
states the same algorithm (quant/binary/optimal.py:41-155) with every sum carried in
fp64 and the per-candidate cost evaluated in closed form from prefix sums, which is
what the HIP solver does on the GPU.  It is the *tight* checker for the HIP solver:
both must choose the same sorted element.  Its relation to the fp32 reference
(``oracle/ref_port.py``) is pinned in ``tests/test_oracle_golden.py``: same candidate
positions up to the reference's rounding clusters, chosen v1 within 1e-3 relative, and
true least-squares cost not worse than the reference's by more than 1e-5 relative.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import this.
"""

from __future__ import annotations

from typing import Dict, Optional

import numpy as np


def abs_subsample(row: np.ndarray, skip: int) -> np.ndarray:
    """|row[::skip]| as float32 (quant/binary/optimal.py:134)."""
    return np.abs(np.asarray(row, dtype=np.float32).reshape(-1)[::skip])


def solve_row(row: np.ndarray, ternary: bool, skip: int = 1,
              details: Optional[Dict] = None) -> np.float32:
    """Optimal v1 for one row, exact arithmetic.

    Candidates (optimal.py:55-83): inner sorted positions i in [1, L-2] with
    a_i <= m <= a_{i+1} for m = half the upper-tail mean, or (non-ternary) the average of
    lower-head mean and upper-tail mean.  Ternary rows whose minimum exceeds half their mean
    get mean/2 appended (optimal.py:86-118).  A row without any candidate yields 0.0 (the
    reference's zero padding wins by default, optimal.py:148-153).  Among candidates the one
    of least cost ||(a-v1) - v2 sign(a-v1)|| wins, first one on ties (optimal.py:151).
    """
    a32 = np.sort(abs_subsample(row, skip))
    a = a32.astype(np.float64)
    n = a.shape[0]
    prefix = np.cumsum(a)
    total = prefix[-1] if n else 0.0
    sq_total = float(np.dot(a, a))
    cand_val = []
    cand_pos = []
    if n >= 3:
        i = np.arange(1, n - 1)
        lo_cnt = (i + 1).astype(np.float64)
        hi_cnt = (n - 1 - i).astype(np.float64)
        hi_mean = (total - prefix[i]) / hi_cnt
        m2 = 0.5 * hi_mean
        hit = (a[i] <= m2) & (m2 <= a[i + 1])
        if not ternary:
            m1 = 0.5 * (prefix[i] / lo_cnt + hi_mean)
            hit |= (a[i] <= m1) & (m1 <= a[i + 1])
        cand_pos = i[hit].tolist()
        cand_val = a[i[hit]].tolist()
    extra = None
    if ternary and n > 0:
        mean = total / n
        if a[0] > 0.5 * mean:
            # the reference forms float32(mean)/2 (optimal.py:116)
            extra = float(np.float32(mean)) / 2
            cand_val.append(float(np.float32(extra)))
            cand_pos.append(-1)

    def scores(v: np.ndarray) -> np.ndarray:
        # closed form of cost^2 = sum (|a - v1| - v2)^2, vectorised over candidates
        k = np.searchsorted(a, v, side='left')                  # elements < v1
        below = np.where(k > 0, prefix[np.maximum(k, 1) - 1], 0.0)
        dev = (v * k - below) + ((total - below) - v * (n - k))     # sum |a - v1|
        quad = sq_total - 2.0 * v * total + n * v * v               # sum (a - v1)^2
        if ternary:
            return quad - 2.0 * v * dev + n * v * v
        return quad - dev * dev / n

    if not cand_val:
        best, costs = np.float32(0.0), []
    else:
        costs = scores(np.asarray(cand_val, dtype=np.float64))
        best = np.float32(cand_val[int(np.argmin(costs))])
        costs = costs.tolist()
    if details is not None:
        details.update(sorted=a32, positions=cand_pos, values=cand_val, cost_sq=costs, n=n)
    return best


def solve_rows(rows: np.ndarray, ternary: bool, skip: int = 1) -> np.ndarray:
    rows = np.asarray(rows, dtype=np.float32)
    rows = rows.reshape(rows.shape[0], -1)
    return np.array([solve_row(r, ternary, skip) for r in rows], dtype=np.float32)


def true_cost(row: np.ndarray, v1: float, ternary: bool, skip: int = 1) -> float:
    """sqrt of the least-squares objective the reference minimises (optimal.py:31-38), in fp64."""
    a = abs_subsample(row, skip).astype(np.float64)
    s = a - v1
    v2 = v1 if ternary else np.abs(s).mean()
    return float(np.sqrt(np.sum((np.abs(s) - v2) ** 2)))
