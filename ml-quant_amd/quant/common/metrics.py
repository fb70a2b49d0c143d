"""Evaluation metrics of the classification tasks.

Interface of the reference's ``quant/common/metrics.py``: ``Metric`` (:50-90), ``LossMetric`` (:93-141),
``Top1Accuracy`` (:144-177), ``TopKAccuracy`` (:180-218): ``update(output, target)`` per batch, ``compute()`` at
the end, ``reset()`` between epochs; ``accumulate=True`` sums over batches, ``False`` keeps the last batch.
The counts stay on the device until ``compute`` (one host sync per epoch instead of one per batch).
"""

from abc import ABC, abstractmethod
from typing import Any, Callable, Optional

import torch
from torch import Tensor


class Metric(ABC):
    DEFAULT_PRECISION = 4

    def __init__(self, accumulate: bool) -> None:
        self.accumulate = accumulate
        self.reset()

    def reset(self) -> None:
        self.n_examples = 0
        self.total = 0.0

    def _add(self, value, n: int) -> None:
        if self.accumulate:
            self.n_examples += n
            self.total = self.total + value
        else:
            self.n_examples = n
            self.total = value

    def _total(self) -> float:
        return float(self.total.item()) if isinstance(self.total, Tensor) else float(self.total)

    @abstractmethod
    def update(self, output: Tensor, target: Tensor, **kwargs: Any) -> None:
        raise NotImplementedError

    @abstractmethod
    def compute(self) -> float:
        raise NotImplementedError


class LossMetric(Metric):
    """Mean of a loss criterion (``criterion(output, target, reduction=...)``)."""

    def __init__(self, criterion: Callable[..., Tensor], accumulate: bool) -> None:
        super().__init__(accumulate)
        self.criterion = criterion

    def update(self, output: Tensor, target: Tensor, teacher_output: Optional[Tensor] = None, **kwargs: Any) -> None:
        n = output.shape[0]
        if teacher_output is not None:            # knowledge distillation criterion: batch mean
            kd = self.criterion(output, teacher_output, target).detach()
            self._add(kd * n if self.accumulate else kd, n)
        elif self.accumulate:
            self._add(self.criterion(output, target, reduction='sum').detach(), n)
        else:
            self._add(self.criterion(output, target, reduction='mean').detach(), n)

    def compute(self) -> float:
        return self._total() / self.n_examples if self.accumulate else self._total()

    def __str__(self) -> str:
        return '{0:.{1}f}'.format(self.compute(), 8)


class TopKAccuracy(Metric):
    """Fraction of samples whose target is among the ``k`` largest outputs."""

    def __init__(self, k: int, accumulate: bool) -> None:
        super().__init__(accumulate)
        self.k = k

    def update(self, output: Tensor, target: Tensor, **kwargs: Any) -> None:
        top = output.topk(self.k, dim=1).indices
        self._add(top.eq(target.view(-1, 1)).any(dim=1).sum().detach(), output.shape[0])

    def compute(self) -> float:
        return self._total() / self.n_examples

    def __str__(self) -> str:
        return '{0}/{1} ({2:.{3}f}%)'.format(int(self._total()), self.n_examples, 100 * self.compute(), self.DEFAULT_PRECISION)


class Top1Accuracy(TopKAccuracy):
    def __init__(self, accumulate: bool) -> None:
        super().__init__(1, accumulate)
