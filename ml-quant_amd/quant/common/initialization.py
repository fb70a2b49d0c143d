"""Model and loss factories used to build a network from a yaml ``model`` section.

The factories of the reference's ``quant/common/initialization.py``: ``model_mapping`` (:21-24), ``get_loss_fn``
(:27-47), ``get_model`` (:97-131), ``get_optimizer`` (:134-157) and ``get_lr_scheduler`` (:160-216).  The reference
wraps multi-GPU models in ``nn.DataParallel``; here multi-GPU inference is one process per GPU
(``quant.common.sharded_eval``), so ``get_model`` always returns the bare module.
"""

import copy
from typing import Callable, Dict

import torch
import torch.nn as nn
import torch.nn.functional as F

from quant.models.lenet import QLeNet5
from quant.models.resnet import QResNet

model_mapping = {'lenet5': QLeNet5, 'resnet': QResNet}

_LOSSES: Dict[str, Callable[..., torch.Tensor]] = {
    'cross_entropy': F.cross_entropy, 'nll_loss': F.nll_loss, 'kl_div': F.kl_div}


def get_loss_fn(loss: str) -> Callable[..., torch.Tensor]:
    """'cross_entropy' | 'nll_loss' | 'kl_div' -> the functional loss."""
    if loss not in _LOSSES:
        raise ValueError(f'Loss function {loss} is not supported.')
    return _LOSSES[loss]


def get_model(architecture: str, loss_fn: Callable[..., torch.Tensor], arch_config: dict,
              device: torch.device, ngpus: int = 1) -> nn.Module:
    """Instantiate ``model_mapping[architecture](loss_fn=..., **arch_config)`` on ``device``."""
    if architecture not in model_mapping:
        raise ValueError(f'Model architecture {architecture} is not found.')
    if ngpus > max(torch.cuda.device_count(), 0) and ngpus > 0 and torch.device(device).type == 'cuda':
        raise ValueError(f'Device only has {torch.cuda.device_count()} GPUs, but {ngpus} are specified.')
    return model_mapping[architecture](loss_fn=loss_fn, **arch_config).to(device)


def get_optimizer(parameters, config: dict) -> torch.optim.Optimizer:
    """``config['algorithm']`` in 'adadelta' | 'adam' | 'sgd'; the remaining keys are the optimizer's arguments."""
    config = copy.deepcopy(config)
    algorithm = config.pop('algorithm')
    table = {'adadelta': torch.optim.Adadelta, 'adam': torch.optim.Adam, 'sgd': torch.optim.SGD}
    return table[algorithm](parameters, **config)


def get_lr_scheduler(optimizer, config: dict, epochs: int, steps_per_epoch: int):
    """``config['scheduler']`` in 'linear_lr' | 'lambda_lr' | 'step_lr' | 'multi_step_lr'.  The training loop steps the
    scheduler after every BATCH, so epoch-denominated arguments (``step_size``, ``milestones``) are scaled by
    ``steps_per_epoch`` and a ``lambda_lr`` function receives the global batch index."""
    from torch.optim import lr_scheduler
    from quant.utils.linear_lr_scheduler import LinearLR
    config = copy.deepcopy(config)
    kind = config.pop('scheduler')
    table = {'linear_lr': LinearLR, 'lambda_lr': lr_scheduler.LambdaLR, 'step_lr': lr_scheduler.StepLR,
             'multi_step_lr': lr_scheduler.MultiStepLR}
    if kind == 'linear_lr':
        config.update(steps_per_epoch=steps_per_epoch, total_epochs=epochs, min_lr=float(config['min_lr']))
    elif kind == 'lambda_lr':
        config['lr_lambda'] = eval(config['lr_lambda'])       # (the yaml holds the lambda as source text, as in the reference)
    elif kind == 'step_lr':
        config['step_size'] *= steps_per_epoch
    elif kind == 'multi_step_lr':
        config['milestones'] = [m * steps_per_epoch for m in config['milestones']]
    return table[kind](optimizer, **config)
