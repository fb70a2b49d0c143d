"""Batch-sharded multi-GPU inference: one process per GPU, full replica per rank, logits all-gathered.

Replaces the reference's single-process ``nn.DataParallel`` wrap (``quant/common/initialization.py:
125-127``: per-forward parameter broadcast + gather to device 0).  In eval mode with
``moving_average_mode: 'off'`` every activation scale is per sample (quantization.py:55,77), batch
norm uses running statistics and weights are read-only, so samples are independent: rank r takes
``x[r*B/W:(r+1)*B/W]`` and the only exchange is ONE all-gather of the fp32 logits per step (RCCL over
xGMI on the GPU, gloo in the CPU tests).  Weights are loaded and bit-packed once per rank.
"""

from typing import Optional

import torch
import torch.distributed as dist


def local_slice(n: int, rank: int, world: int) -> slice:
    """Contiguous shard of ``n`` samples owned by ``rank`` (remainder spread over the first ranks)."""
    base, extra = divmod(n, world)
    start = rank * base + min(rank, extra)
    return slice(start, start + base + (1 if rank < extra else 0))


def all_gather_logits(local: torch.Tensor, out: Optional[torch.Tensor] = None, group=None) -> torch.Tensor:
    """Concatenate every rank's ``[B_local, classes]`` logits along dim 0 (equal B_local on all ranks)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    if out is None:
        out = local.new_empty((world * local.shape[0],) + tuple(local.shape[1:]))
    local = local.contiguous()
    if local.is_cuda:
        dist.all_gather_into_tensor(out, local, group=group)
    else:
        dist.all_gather(list(out.chunk(world, dim=0)), local, group=group)
    return out


@torch.no_grad()
def evaluate_sharded(model: torch.nn.Module, x_local: torch.Tensor, out: Optional[torch.Tensor] = None,
                     group=None) -> torch.Tensor:
    """One inference step: local forward on this rank's shard, then the all-gather of logits."""
    return all_gather_logits(model(x_local), out, group)
