"""Linearly decaying learning rate, stepped once per batch (interface of the reference's
``quant/utils/linear_lr_scheduler.py:14-54``)."""

from typing import List

from torch.optim.lr_scheduler import _LRScheduler


class LinearLR(_LRScheduler):
    """lr(step) = max(lr_0 - step / ((total_epochs - 1) * steps_per_epoch) * (lr_0 + min_lr), min_lr).

    ``last_epoch`` counts BATCHES (the training loop steps the scheduler after every batch); -1 starts afresh."""

    def __init__(self, optimizer, min_lr: float, total_epochs: int, steps_per_epoch: int, last_epoch: int = -1) -> None:
        self.min_lr = min_lr
        self.total_epochs = total_epochs
        self.steps_per_epoch = steps_per_epoch
        super().__init__(optimizer, last_epoch)

    def get_lr(self) -> List[float]:
        span = (self.total_epochs - 1) * self.steps_per_epoch
        done = self.last_epoch
        return [max(g['initial_lr'] - done / span * (g['initial_lr'] + self.min_lr), self.min_lr)
                for g in self.optimizer.param_groups]
