"""The training and evaluation loops of the reference's ``quant/common/training.py``.

``train`` follows ``training.py:66-152``: one optimizer step and one scheduler step per batch, optional knowledge
distillation from a teacher, metrics on the training outputs, hooks once per batch.  On a GPU every train-mode
``QuantConv2d`` runs its forward and backward through the gfx950 kernels (``quant.binary.hip_train``).

``evaluate`` follows ``training.py:155-204``: ``model.eval()``, metrics reset, ``torch.no_grad()``, one forward
per batch of the test loader, metrics updated on the device, hooks called once at the end.  The reference runs
multi-GPU evaluation through ``nn.DataParallel``; here, when a process group is initialised (one process per
GPU), every rank takes its slice of each batch and the logits are all-gathered before the metrics see them
(``quant.common.sharded_eval``), so every rank reports the metrics of the whole test set.

"""

import logging
from typing import Callable, Dict, Optional, Sequence

import torch
import torch.distributed as dist
import torch.nn as nn

from quant.common.metrics import Metric
from quant.common.sharded_eval import all_gather_logits, local_forward, local_slice
from quant.common.stream_pipeline import StreamPipeline, eval_streams

logger = logging.getLogger(__name__)
Hook = Callable[..., None]


def _stem_flag(device) -> bool:
    """Is THIS rank's device flag of the fused stem up (an operand outside the fp16 split's domain since the last reset)?"""
    if torch.device(device).type != 'cuda':
        return False
    from quant import _hip
    return bool(_hip.available() and _hip.stem_overflow_flag_raised(device))


def _stem_flag_reset(device) -> None:
    if torch.device(device).type == 'cuda':
        from quant import _hip
        _hip.stem_overflow_reset(device, keep_tripped=True)


def _any_rank(flag: bool, device, sharded: bool) -> bool:
    """``flag`` or-ed over the ranks of a sharded evaluation: the flag is per device, so only the rank whose shard held the
    offending sample sees it -- and a rank that repeated the pass alone would issue a second series of all-gathers that the
    others never join.  Every rank calls this at the same point; all get the same answer."""
    if not sharded:
        return flag
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(int(t.item()))


def evaluate(model: nn.Module, test_loader, metrics: Dict[str, Metric], device: torch.device, epoch: int,
             hooks: Optional[Sequence[Hook]] = None, _retry: bool = False) -> Dict[str, float]:
    """Evaluate ``model`` on ``test_loader``; returns {metric name: value}."""
    hooks = hooks or []
    model.eval()
    for metric in metrics.values():
        metric.reset()
    sharded = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if not _retry and _stem_flag(device):
        # the flag is sticky and shared by everything that runs on this device: one raised BEFORE this pass (an earlier
        # forward, another model) is noted and lowered, so that only overflows of THIS pass make it repeat
        _stem_flag_reset(device)
    batch_idx = -1
    # consecutive batches alternate between two HIP streams (stream_pipeline.py: the dispatch ramp and the last tiles of one
    # forward's kernels run under the other forward's kernels); the metrics of batch i - 1 -- and, when the batch is sharded
    # over ranks, the all-gather of its logits -- are taken on THIS stream while batch i runs.  Every rank submits and
    # consumes in the same order, so the collectives line up.  One stream on the CPU: the reference's order of operations.
    pipe = StreamPipeline(lambda shard: local_forward(model, shard) if sharded else model(shard), device,
                          eval_streams(device, sharded))
    window = []

    def consume():
        pending, target, total = window.pop(0)
        output = pending.result()
        if sharded:
            output = all_gather_logits(output, total=total)
        for metric in metrics.values():
            metric.update(output, target)

    with torch.no_grad():
        for batch_idx, (data, target) in enumerate(test_loader):
            target = target.to(device)
            if sharded:
                # the shard is cut on the host: only this rank's samples cross PCIe
                data = data[local_slice(data.shape[0], dist.get_rank(), dist.get_world_size())]
            window.append((pipe.submit(data.to(device)), target, target.shape[0]))
            if len(window) >= pipe.depth:
                consume()
        while window:
            consume()
    if torch.device(device).type == 'cuda' or sharded:
        # batches of this pass computed on saturated stem operands (inf / nan logits) on ANY rank: the stem has switched to
        # the bf16 split, which takes any finite input -- the whole pass is repeated on it, by every rank or by none
        if _any_rank(_stem_flag(device), device, sharded):
            if not _retry:
                logger.warning('lsq_stem_conv_pool saw operands at or beyond 65504 during this evaluation; repeating it with '
                               'the bf16 split of the stem')
                _stem_flag_reset(device)
                return evaluate(model, test_loader, metrics, device, epoch, hooks, _retry=True)
            logger.error('lsq_stem_conv_pool: the flag is still raised after the repeated evaluation (another model on this '
                         'device?); the metrics below may come from saturated logits')
    for hook in hooks:
        hook(epoch=epoch, global_step=1 + (epoch - 1) * len(test_loader.dataset) + batch_idx)
    computed = {name: metric.compute() for name, metric in metrics.items()}
    logger.info('Test set evaluation metrics:')
    for name, metric in metrics.items():
        logger.info(f'{name}: {metric}')
    return computed


def _get_lr(optimizer) -> float:
    for group in optimizer.param_groups:
        return group['lr']
    raise ValueError('Cannot get optimizer LR: optimizer does not have any parameter groups.')


def project(optimizer) -> None:
    """Placeholder kept for interface parity (training.py:55-63: projecting the latent weights to [-1, 1] made no
    difference in the reference's experiments, so it is a no-op there too)."""
    return None


def train(model: nn.Module, train_loader, metrics: Dict[str, Metric], optimizer, scheduler, device: torch.device,
          epoch: int, log_interval: int, hooks: Optional[Sequence[Hook]] = None,
          teacher: Optional[nn.Module] = None) -> Dict[str, float]:
    """One epoch over ``train_loader``; returns {metric name: value} of the training outputs.

    ``model.loss_fn(output, target)`` is the criterion, or ``model.loss_fn(output, teacher_output, target)`` when a
    ``teacher`` is given (knowledge distillation).  The scheduler steps once per batch, as in the reference."""
    hooks = hooks or []
    model.train()
    for metric in metrics.values():
        metric.reset()
    loss_fn = model.module.loss_fn if isinstance(model, nn.DataParallel) else model.loss_fn
    seen = 0
    for batch_idx, (data, target) in enumerate(train_loader):
        data, target = data.to(device), target.to(device)
        optimizer.zero_grad()
        output = model(data)
        if teacher is None:
            teacher_output = None
            loss = loss_fn(output, target)
        else:
            teacher_output = teacher(data)
            loss = loss_fn(output, teacher_output, target)
        loss.backward()
        optimizer.step()
        project(optimizer)
        scheduler.step()
        with torch.no_grad():
            for metric in metrics.values():
                metric.update(output, target, teacher_output=teacher_output)
        for hook in hooks:
            hook(epoch=epoch, global_step=1 + (epoch - 1) * len(train_loader.dataset) + batch_idx,
                 values_dict={'lr': _get_lr(optimizer)}, log_interval=log_interval)
        seen += len(data)
        if batch_idx % log_interval == 0:
            logger.info('Train Epoch: {} [{}/{} ({:.0f}%)]\tBatch Loss: {:.6f}'.format(
                epoch, seen, len(train_loader.dataset), 100 * batch_idx / len(train_loader), loss.item()))
    computed = {name: metric.compute() for name, metric in metrics.items()}
    logger.info('Training set evaluation metrics:')
    for name, metric in metrics.items():
        logger.info(f'{name}: {metric}')
    return computed
