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


def all_gather_logits(local: torch.Tensor, out: Optional[torch.Tensor] = None, group=None,
                      total: Optional[int] = None, always_collective: bool = False) -> torch.Tensor:
    """Concatenate every rank's ``[B_local, classes]`` logits along dim 0.

    ``total`` = the number of samples over all ranks when they were split with :func:`local_slice`; it may be
    omitted when every rank holds the same number.  Collectives need identical shapes on every rank, so uneven
    shards are padded to the largest one for the exchange and trimmed afterwards.  ``always_collective`` issues the
    collective with a single rank too (the one-GPU check that the RCCL path works end to end)."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    if dist.get_world_size(group) == 1 and not always_collective:
        return local
    world = dist.get_world_size(group)
    sizes = None
    if total is not None:
        sizes = [local_slice(total, r, world) for r in range(world)]
        sizes = [s.stop - s.start for s in sizes]
        if sizes[dist.get_rank(group)] != local.shape[0]:
            raise ValueError(f'rank holds {local.shape[0]} samples, local_slice({total}, ...) gives {sizes[dist.get_rank(group)]}')
        if len(set(sizes)) == 1:
            sizes = None
    if sizes is None:
        if out is None:
            out = local.new_empty((world * local.shape[0],) + tuple(local.shape[1:]))
        local = local.contiguous()
        if local.is_cuda:
            dist.all_gather_into_tensor(out, local, group=group)
        else:
            dist.all_gather(list(out.chunk(world, dim=0)), local, group=group)
        return out
    widest = max(sizes)
    padded = local.new_zeros((widest,) + tuple(local.shape[1:]))
    padded[:local.shape[0]] = local
    buf = local.new_empty((world * widest,) + tuple(local.shape[1:]))
    if local.is_cuda:
        dist.all_gather_into_tensor(buf, padded, group=group)
    else:
        dist.all_gather(list(buf.chunk(world, dim=0)), padded, group=group)
    parts = [buf[r * widest:r * widest + sizes[r]] for r in range(world)]
    if out is None:
        return torch.cat(parts, dim=0)
    torch.cat(parts, dim=0, out=out)
    return out


@torch.no_grad()
def local_forward(model: torch.nn.Module, x_local: torch.Tensor) -> torch.Tensor:
    """This rank's logits for its shard -- the half of :func:`evaluate_sharded` in front of the collective (the evaluation
    loop queues it on a side stream and all-gathers on its own).  An empty shard: see :func:`evaluate_sharded`."""
    if x_local.shape[0] == 0:
        probe = model(x_local.new_zeros((1,) + tuple(x_local.shape[1:])))
        return probe[:0]
    return model(x_local)


@torch.no_grad()
def evaluate_sharded(model: torch.nn.Module, x_local: torch.Tensor, out: Optional[torch.Tensor] = None,
                     group=None, total: Optional[int] = None, always_collective: bool = False) -> torch.Tensor:
    """One inference step: local forward on this rank's shard, then the all-gather of logits.

    A batch smaller than the number of ranks leaves some ranks without a sample (``local_slice`` gives them an empty
    shard): the kernels refuse empty batches, and a rank that raised would leave the others waiting in the
    collective -- such a rank runs the model on ONE sample borrowed from its own padding (zeros of the input's
    shape), to learn the width of the logits, and contributes no rows."""
    return all_gather_logits(local_forward(model, x_local), out, group, total, always_collective)
