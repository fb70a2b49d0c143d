"""Activation quantizer modules with optional moving-average scales.

Mirrors the interface of the reference's ``quant/binary/activation_quantization.py``:
``MovingAverageMode`` (:19-28), ``ActivationQuantizer`` (:31-114) and the LS1 / LS2 / LST / GF
subclasses (:117-239).  With mode ``'off'`` (every example yaml) scales are re-solved for
every batch, per sample, in training *and* eval -- which is what puts the scale solve on the
inference hot path.  Each subclass also exposes ``hip_scheme`` / ``n_planes`` so that
``QuantConv2d`` can drive the gfx950 kernels for eval-mode CUDA tensors.
"""

from abc import abstractmethod
from enum import Enum
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

import quant.binary.quantization as quantization
from quant.utils.moving_average import MovingAverage


class MovingAverageMode(Enum):
    """'off' | 'eval_only' (tracked in training, used in eval) | 'train_and_eval'."""

    off = 'off'
    eval_only = 'eval_only'
    train_and_eval = 'train_and_eval'


class ActivationQuantizer(nn.Module):
    """Base class: batch quantization + exponential moving average of the batch-mean scales."""

    hip_scheme = 0      # LSQ_SCHEME_* of include/lsq_hip.h
    scheme = ''

    def __init__(self, num_scaling_factors: int, moving_average_mode: str = 'off',
                 moving_average_momentum: float = 0.99) -> None:
        super().__init__()
        self.num_scaling_factors = num_scaling_factors
        self.moving_avg_module = MovingAverage(torch.tensor([moving_average_momentum] * num_scaling_factors))
        self.moving_average_mode = MovingAverageMode(moving_average_mode)
        # test hook: per-sample scales [num_scaling_factors, N] to use instead of solving
        self._forced_scales: Optional[torch.Tensor] = None

    @property
    def n_planes(self) -> int:
        """Sign planes produced per activation."""
        return self.num_scaling_factors

    def eval_scales(self, batch: int) -> Optional[torch.Tensor]:
        """[planes, batch] scales fixed ahead of the forward in eval mode, or None (solve per sample)."""
        if self._forced_scales is not None:
            return self.plane_scales(self._forced_scales)
        if self.moving_average_mode != MovingAverageMode.off:
            avg = self.moving_avg_module.moving_average
            return self.plane_scales(avg.view(-1, 1).expand(-1, batch))
        return None

    def plane_scales(self, scales: torch.Tensor) -> torch.Tensor:
        return scales.contiguous()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self._forced_scales is not None:
            return self._moving_average_quantization(x, list(self._forced_scales))
        use_average = self.moving_average_mode != MovingAverageMode.off
        if self.training:
            batch_vs, x_q = self._batch_quantization(x)
            if use_average:
                tracked = self.moving_avg_module(batch_vs.mean(1))
                if self.moving_average_mode == MovingAverageMode.train_and_eval:
                    x_q = self._moving_average_quantization(
                        x, [tracked[i].expand(x.shape[0]) for i in range(self.num_scaling_factors)])
            return x_q
        if use_average:
            avg = self.moving_avg_module.moving_average
            return self._moving_average_quantization(x, [avg[i].expand(x.shape[0]) for i in range(avg.size(0))])
        return self._batch_quantization(x)[1]

    @abstractmethod
    def _batch_quantization(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """(scales [num_scaling_factors, N], quantized x)."""

    @abstractmethod
    def _moving_average_quantization(self, x: torch.Tensor, vs: List[torch.Tensor]) -> torch.Tensor:
        """Quantized x for given scales."""


class ActivationQuantizerLS1(ActivationQuantizer):
    """Least squares, 1 bit."""

    hip_scheme = 1
    scheme = 'ls-1'

    def __init__(self, moving_average_mode: str = 'off', moving_average_momentum: float = 0.99) -> None:
        super().__init__(1, moving_average_mode, moving_average_momentum)

    def _batch_quantization(self, x):
        v1, x_q = quantization.quantizer_ls_1(x)
        return v1.view(1, -1), x_q

    def _moving_average_quantization(self, x, vs):
        return quantization.quantizer_ls_1(x, vs[0])[1]


class ActivationQuantizerLS2(ActivationQuantizer):
    """Least squares, 2 bits."""

    hip_scheme = 2
    scheme = 'ls-2'

    def __init__(self, moving_average_mode: str = 'off', moving_average_momentum: float = 0.99) -> None:
        super().__init__(2, moving_average_mode, moving_average_momentum)

    def _batch_quantization(self, x):
        v1, v2, x_q = quantization.quantizer_ls_2(x)
        return torch.stack([v1, v2]), x_q

    def _moving_average_quantization(self, x, vs):
        return quantization.quantizer_ls_2(x, vs[0], vs[1])[2]


class ActivationQuantizerLST(ActivationQuantizer):
    """Least squares, ternary (two sign planes sharing one scale)."""

    hip_scheme = 3
    scheme = 'ls-T'

    def __init__(self, moving_average_mode: str = 'off', moving_average_momentum: float = 0.99) -> None:
        super().__init__(1, moving_average_mode, moving_average_momentum)

    @property
    def n_planes(self) -> int:
        return 2

    def plane_scales(self, scales: torch.Tensor) -> torch.Tensor:
        return torch.cat([scales[:1], scales[:1]]).contiguous()

    def _batch_quantization(self, x):
        v1, x_q = quantization.quantizer_ls_ternary(x)
        return v1.view(1, -1), x_q

    def _moving_average_quantization(self, x, vs):
        return quantization.quantizer_ls_ternary(x, vs[0])[1]


class ActivationQuantizerGF(ActivationQuantizer):
    """Greedy foldable, k bits."""

    hip_scheme = 4

    def __init__(self, k: int, moving_average_mode: str = 'off', moving_average_momentum: float = 0.99) -> None:
        super().__init__(k, moving_average_mode, moving_average_momentum)
        self.k = k
        self.scheme = f'gf-{k}'

    def _batch_quantization(self, x):
        vs, x_q = quantization.quantizer_gf(x, self.k)
        return torch.stack(vs), x_q

    def _moving_average_quantization(self, x, vs):
        return quantization.quantizer_gf(x, self.k, vs)[1]
