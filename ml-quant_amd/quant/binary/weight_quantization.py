"""Weight quantizer modules: compute-and-cache scales in training, reuse them in eval.

Buffer names (``v1`` ... ``vk``, length ``out_channels``, initialised to 0) follow the
reference's ``quant/binary/weight_quantization.py`` (:25, :48-49, :73, :97-98) so checkpoints
are interchangeable; a module that was never trained therefore quantizes to all zeros in eval.
"""

from typing import List

import torch
import torch.nn as nn

import quant.binary.quantization as quantization


class _CachedScaleQuantizer(nn.Module):
    """Shared train/eval protocol; subclasses say how scales are computed and applied."""

    n_scales = 1
    scheme = ''

    def __init__(self, size: int) -> None:
        super().__init__()
        for i in range(1, self.n_scales + 1):
            self.register_buffer(f'v{i}', torch.zeros(size))

    def cached_scales(self) -> List[torch.Tensor]:
        return [getattr(self, f'v{i}') for i in range(1, self.n_scales + 1)]

    def plane_scales(self) -> torch.Tensor:
        """[planes, out_channels] scales of the sign planes the HIP kernels consume."""
        return torch.stack(self.cached_scales())

    def _store(self, scales) -> None:
        for buf, v in zip(self.cached_scales(), scales):
            buf.copy_(v)


class WeightQuantizerLS1(_CachedScaleQuantizer):
    """Least squares, 1 bit."""

    scheme = 'ls-1'

    def forward(self, w: torch.Tensor) -> torch.Tensor:
        if self.training:
            v1, w_q = quantization.quantizer_ls_1(w)
            self._store([v1])
        else:
            _, w_q = quantization.quantizer_ls_1(w, self.v1)
        return w_q


class WeightQuantizerLS2(_CachedScaleQuantizer):
    """Least squares, 2 bits."""

    n_scales = 2
    scheme = 'ls-2'

    def forward(self, w: torch.Tensor, skip: int = 3) -> torch.Tensor:
        if self.training:
            v1, v2, w_q = quantization.quantizer_ls_2(w, skip=skip)
            self._store([v1, v2])
        else:
            _, _, w_q = quantization.quantizer_ls_2(w, self.v1, self.v2, skip=skip)
        return w_q


class WeightQuantizerLST(_CachedScaleQuantizer):
    """Least squares, ternary."""

    scheme = 'ls-T'

    def forward(self, w: torch.Tensor, skip: int = 3) -> torch.Tensor:
        if self.training:
            v1, w_q = quantization.quantizer_ls_ternary(w, skip=skip)
            self._store([v1])
        else:
            _, w_q = quantization.quantizer_ls_ternary(w, self.v1, skip=skip)
        return w_q

    def plane_scales(self) -> torch.Tensor:
        return torch.stack([self.v1, self.v1])


class WeightQuantizerGF(_CachedScaleQuantizer):
    """Greedy foldable, k bits."""

    def __init__(self, size: int, k: int) -> None:
        self.n_scales = k
        self.k = k
        self.scheme = f'gf-{k}'
        super().__init__(size)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.training:
            vs, x_q = quantization.quantizer_gf(x, k=self.k)
            self._store(vs)
        else:
            _, x_q = quantization.quantizer_gf(x, k=self.k, vs=self.cached_scales())
        return x_q
