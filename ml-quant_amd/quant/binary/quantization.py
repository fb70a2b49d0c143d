"""Functional quantizers (host side) with the reference's call conventions.

``quantizer_ls_1`` / ``quantizer_ls_2`` / ``quantizer_ls_ternary`` / ``quantizer_gf`` return the
same tuples as the reference's ``quant/binary/quantization.py`` (:35, :59, :95, :118); these
torch formulations serve CPU tensors and training (autograd through the STE).  Eval-mode
CUDA tensors never come here: ``QuantConv2d`` hands them to the gfx950 kernels.
"""

from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from quant.binary.optimal import opt_v1
from quant.binary.ste import binarize, binary_sign


def clamp_identity(x: torch.Tensor) -> torch.Tensor:
    """No clamping."""
    return x


def clamp_symmetric(x: torch.Tensor, alpha: float) -> torch.Tensor:
    """Clamp to [-alpha, alpha]."""
    return torch.clamp(x, min=-alpha, max=alpha)


class QuantizerFP(nn.Module):
    """Full precision: the identity."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return x


def _per_row(v: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    return v.reshape(x.shape[0], *([1] * (x.dim() - 1)))


def quantizer_ls_1(x: torch.Tensor, v1: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """1-bit least squares: v1 = mean |x| per row (dim 0), x_q = v1 * sign(x)."""
    if v1 is None:
        mag = x.detach().abs()
        while mag.dim() > 1:                      # nested means, innermost first
            mag = mag.mean(dim=-1)
        v1 = mag
    return v1, _per_row(v1, x) * binarize(x)


def quantizer_ls_2(x: torch.Tensor, v1: Optional[torch.Tensor] = None, v2: Optional[torch.Tensor] = None,
                   skip: int = 3) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """2-bit least squares: x_q = v1 b1 + v2 sign(x - v1 b1) with the optimal v1 and v2 = mean|x - v1 b1|."""
    flat = x.detach().reshape(x.shape[0], -1)
    v1 = opt_v1(flat, ternary=False, skip=skip).view(-1) if v1 is None else v1.reshape(-1)
    if v2 is None:
        v2 = (flat - v1.view(-1, 1) * binary_sign(flat)).abs().mean(dim=-1)
    else:
        v2 = v2.reshape(-1)
    first = _per_row(v1, x) * binarize(x)
    return v1, v2, first + _per_row(v2, x) * binarize(x - first)


def quantizer_ls_ternary(x: torch.Tensor, v1: Optional[torch.Tensor] = None,
                         skip: int = 3) -> Tuple[torch.Tensor, torch.Tensor]:
    """Ternary least squares: x_q = v1 (b1 + sign(x - v1 b1)) in {-2 v1, 0, 2 v1}."""
    if v1 is None:
        v1 = opt_v1(x.detach().reshape(x.shape[0], -1), ternary=True, skip=skip)
    v1 = v1.reshape(-1)
    s1 = _per_row(v1, x)
    b1 = binarize(x)
    return v1, s1 * (b1 + binarize(x - s1 * b1))


def quantizer_gf(x: torch.Tensor, k: int,
                 vs: Optional[List[torch.Tensor]] = None) -> Tuple[List[torch.Tensor], torch.Tensor]:
    """Greedy foldable k-bit quantization: v_i = mean |residual|, one sign plane per step."""
    if vs is not None and len(vs) != k:
        raise ValueError('If vs is passed in, all vs from v_1 to v_k must be passed in (could be None).')
    residual = x.detach().reshape(x.shape[0], -1).clone()
    scales: List[torch.Tensor] = []
    approx = 0
    for i in range(k):
        v = vs[i] if vs is not None else residual.abs().mean(dim=-1)
        scales.append(v)
        residual = residual - v.view(-1, 1) * binary_sign(residual)
        approx = approx + _per_row(v, x) * binarize(x - approx)
    return scales, approx
