"""Model and loss factories used to build a network from a yaml ``model`` section.

Subset of the reference's ``quant/common/initialization.py`` that the inference path needs:
``model_mapping`` (:21-24), ``get_loss_fn`` (:27-47) and ``get_model`` (:97-131).  The reference
wraps multi-GPU models in ``nn.DataParallel``; here multi-GPU inference is one process per GPU
(``quant.common.sharded_eval``), so ``get_model`` always returns the bare module.
"""

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
