"""The evaluation loop (inference side of the reference's ``quant/common/training.py``).

``evaluate`` follows ``training.py:155-204``: ``model.eval()``, metrics reset, ``torch.no_grad()``, one forward
per batch of the test loader, metrics updated on the device, hooks called once at the end.  The reference runs
multi-GPU evaluation through ``nn.DataParallel``; here, when a process group is initialised (one process per
GPU), every rank takes its slice of each batch and the logits are all-gathered before the metrics see them
(``quant.common.sharded_eval``), so every rank reports the metrics of the whole test set.

Training (``train``, :66-152) is outside this build's scope (the path is the eval forward).
"""

import logging
from typing import Callable, Dict, Optional, Sequence

import torch
import torch.distributed as dist
import torch.nn as nn

from quant.common.metrics import Metric
from quant.common.sharded_eval import evaluate_sharded, local_slice

logger = logging.getLogger(__name__)
Hook = Callable[..., None]


def evaluate(model: nn.Module, test_loader, metrics: Dict[str, Metric], device: torch.device, epoch: int,
             hooks: Optional[Sequence[Hook]] = None) -> Dict[str, float]:
    """Evaluate ``model`` on ``test_loader``; returns {metric name: value}."""
    hooks = hooks or []
    model.eval()
    for metric in metrics.values():
        metric.reset()
    sharded = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    batch_idx = -1
    with torch.no_grad():
        for batch_idx, (data, target) in enumerate(test_loader):
            target = target.to(device)
            if sharded:
                # the shard is cut on the host: only this rank's samples cross PCIe
                part = local_slice(data.shape[0], dist.get_rank(), dist.get_world_size())
                output = evaluate_sharded(model, data[part].to(device), total=data.shape[0])
            else:
                output = model(data.to(device))
            for metric in metrics.values():
                metric.update(output, target)
    for hook in hooks:
        hook(epoch=epoch, global_step=1 + (epoch - 1) * len(test_loader.dataset) + batch_idx)
    computed = {name: metric.compute() for name, metric in metrics.items()}
    logger.info('Test set evaluation metrics:')
    for name, metric in metrics.items():
        logger.info(f'{name}: {metric}')
    return computed


def train(*args, **kwargs):
    raise NotImplementedError('training is outside the scope of this build (inference path only): run the drivers '
                              'with --skip-training')
