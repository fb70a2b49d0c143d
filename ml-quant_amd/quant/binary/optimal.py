"""Optimal first scale of the 2-bit / ternary least-squares quantizers (host implementation).

Same contract as the reference's ``quant/binary/optimal.py`` (``opt_v1`` :121-155,
``compute_mask`` :41-83, ``cost_function`` :16-38) but stated the way the gfx950 solver
computes it (``csrc/lsq_act_quant.hip``): prefix sums are carried in fp64 and a candidate's
cost is evaluated in closed form from them, O(M log M) per row with no [N, K, M] temporary.
CUDA tensors go to the HIP solver; this torch formulation serves CPU tensors and autograd.

Where this (and the HIP solver, which computes the same thing) intentionally differs from the reference:
  * exact cost ties -- integer-valued or heavily clamped rows, where several candidates have exactly the same
    true cost -- are broken by (fp64 closed-form cost, smallest sorted position); the reference takes the first
    argmin of fp32 costs over its zero-padded candidate list, so which of the tied candidates it returns depends
    on fp32 rounding of its own norm / mean reductions.  Either answer has the same least-squares cost.
  * the ternary extra candidate (half the row mean, optimal.py:147-152) is the fp64 mean rounded to fp32, up to one
    ulp from the reference's fp32 ``matrix.mean() / 2``;
  * a row with no candidate returns 0.  The reference raises only when NO row of the batch has one (``argmin`` over
    an empty dimension, optimal.py:147-151: an ``IndexError``; a row without candidates next to one with gets the zero
    of the padding) -- in practice rows of fewer than three elements.  ``STRICT_NO_CANDIDATE = True`` makes
    ``opt_v1`` raise the same ``IndexError`` in that case (on the GPU it reads the solver's per-row status back: one
    synchronisation per call, which is why it is not the default); the fused eval path of ``QuantConv2d`` never raises.
On continuous data the results match the reference's bit for bit in most rows and within 1e-3 relative in the
rest (near-tied candidates); tests/test_oracle_golden.py::test_exact_solver_vs_reference.
"""

from typing import Tuple

import torch

#: True: ``opt_v1`` raises ``IndexError`` like the reference when no row of the batch has a candidate (see above)
STRICT_NO_CANDIDATE = False


def _prefix(matrix: torch.Tensor):
    values, _ = torch.sort(matrix, dim=1)
    run = values.to(torch.float64).cumsum(dim=1)
    return values, run


def compute_mask(matrix: torch.Tensor, ternary: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """Candidate mask over inner sorted positions of a matrix of absolute values.

    Position i (1 <= i <= L-2) is kept when half the mean of the elements above it, or (2-bit
    case) the average of the means below-and-including / above it, lies in [a_i, a_{i+1}].
    Returns (mask [N, L-2], selected values as a flat vector).
    """
    values, run = _prefix(matrix)
    n = matrix.shape[1]
    idx = torch.arange(1, n - 1, device=matrix.device)
    lo_cnt = (idx + 1).to(torch.float64)
    hi_cnt = (n - 1 - idx).to(torch.float64)
    v64 = values.to(torch.float64)
    hi_mean = (run[:, -1:] - run[:, 1:-1]) / hi_cnt
    left, right = v64[:, 1:-1], v64[:, 2:]
    half = 0.5 * hi_mean
    mask = (left <= half) & (half <= right)
    if not ternary:
        mid = 0.5 * (run[:, 1:-1] / lo_cnt + hi_mean)
        mask = mask | ((left <= mid) & (mid <= right))
    return mask, torch.masked_select(values[:, 1:-1], mask)


def cost_function(matrix: torch.Tensor, v1s: torch.Tensor, ternary: bool = False) -> torch.Tensor:
    """||(a - v1) - v2 sign(a - v1)||_2 per (row, candidate), v2 = mean|a - v1| (or v1 if ternary).

    ``matrix`` [N, L] holds absolute values, ``v1s`` [N, K] candidates; closed form via prefix sums.
    """
    sq = (matrix.to(torch.float64) ** 2).sum(dim=1, keepdim=True)
    return (_cost_sq(matrix, v1s, ternary) + sq).clamp_min(0).sqrt().to(matrix.dtype)


def _opt_v1_torch(matrix_skipped: torch.Tensor, ternary: bool) -> torch.Tensor:
    a = matrix_skipped
    rows, n = a.shape
    out = a.new_zeros((rows, 1))
    if n < 3 and not ternary:
        return out
    mask, _ = compute_mask(a, ternary) if n >= 3 else (a.new_zeros((rows, 0), dtype=torch.bool), None)
    values, _ = torch.sort(a, dim=1)
    inner = values[:, 1:-1] if n >= 3 else values[:, :0]
    big = torch.finfo(torch.float64).max
    if inner.shape[1] > 0:
        costs = _cost_sq(a, inner, ternary)      # squared cost minus a per-row constant
        costs = torch.where(mask, costs, torch.full_like(costs, big))
        best_cost, best_idx = costs.min(dim=1, keepdim=True)
        found = mask.any(dim=1, keepdim=True)
        out = torch.where(found, torch.gather(inner, 1, best_idx), out)
    else:
        best_cost = a.new_full((rows, 1), big, dtype=torch.float64)
        found = torch.zeros((rows, 1), dtype=torch.bool, device=a.device)
    if ternary:
        mean = a.to(torch.float64).mean(dim=1, keepdim=True)
        low = a.min(dim=1, keepdim=True).values.to(torch.float64)
        extra = low > 0.5 * mean
        half = (mean.to(a.dtype).to(torch.float64) / 2).to(a.dtype)
        extra_cost = _cost_sq(a, half, True)
        take = extra & (~found | (extra_cost < best_cost))
        out = torch.where(take, half, out)
    return out


def _cost_sq(matrix: torch.Tensor, v1s: torch.Tensor, ternary: bool) -> torch.Tensor:
    values, run = _prefix(matrix)
    n = matrix.shape[1]
    v64 = values.to(torch.float64)
    cand = v1s.to(torch.float64)
    total = run[:, -1:]
    below = torch.searchsorted(v64.contiguous(), cand.contiguous(), right=False)
    below_sum = torch.gather(torch.cat([torch.zeros_like(total), run], dim=1), 1, below)
    dev = (cand * below - below_sum) + ((total - below_sum) - cand * (n - below))
    quad = -2.0 * cand * total + n * cand * cand           # minus the constant sum a^2
    return quad - 2.0 * cand * dev + n * cand * cand if ternary else quad - dev * dev / n


def opt_v1(matrix: torch.Tensor, ternary: bool, skip: int = 1) -> torch.Tensor:
    """Optimal v1 per row of a 2-D tensor, searched over every ``skip``-th element. Returns [N, 1]."""
    with torch.no_grad():
        if matrix.is_cuda and matrix.dtype == torch.float32:     # (other dtypes: the torch formulation below)
            from quant import _hip
            v12, status = _hip.solve_rows(matrix, skip, ternary)
            if STRICT_NO_CANDIDATE and int(status.sum()) == 0:
                raise IndexError('argmin(): Expected reduction dim 2 to have non-zero size.')     # (optimal.py:151 of the reference)
            return v12[0].view(-1, 1)
        a = matrix[..., ::skip].abs()
        if STRICT_NO_CANDIDATE and not _any_candidate(a, ternary):
            raise IndexError('argmin(): Expected reduction dim 2 to have non-zero size.')
        return _opt_v1_torch(a, ternary)


def _any_candidate(a: torch.Tensor, ternary: bool) -> bool:
    """Does any row have a candidate (a masked inner position, or the ternary scheme's extra one)?"""
    n = a.shape[1]
    if n >= 3 and bool(compute_mask(a, ternary)[0].any()):
        return True
    if ternary and n > 0:
        return bool((a.min(dim=1).values.to(torch.float64) > 0.5 * a.to(torch.float64).mean(dim=1)).any())
    return False
