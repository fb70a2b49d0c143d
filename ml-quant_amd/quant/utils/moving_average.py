"""Exponential moving average of activation scales.

Buffer names (``num_batches_tracked``, ``momentum``, ``moving_average``) match the
reference's ``quant/utils/moving_average.py:23-25`` so its checkpoints load with strict keys.
"""

import torch
import torch.nn as nn


class MovingAverage(nn.Module):
    """``m * old + (1 - m) * new`` per entry; the first update copies the input."""

    def __init__(self, momentum: torch.Tensor) -> None:
        super().__init__()
        self.register_buffer('num_batches_tracked', torch.tensor(0))
        self.register_buffer('momentum', momentum)
        self.register_buffer('moving_average', torch.zeros(len(momentum)))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.training:
            with torch.no_grad():
                if int(self.num_batches_tracked) == 0:
                    self.moving_average.copy_(x)
                else:
                    keep = self.momentum * self.moving_average
                    self.moving_average.copy_(keep + (1.0 - self.momentum) * x)
                self.num_batches_tracked += 1
        return self.moving_average
