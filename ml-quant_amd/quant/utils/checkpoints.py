"""Checkpoint files in the reference's on-disk layout.

A checkpoint is ``torch.save({'epoch', 'model_state_dict', 'optimizer_state_dict',
'scheduler_state_dict'}, <dir>/checkpoint_<epoch>.pt)`` (reference ``quant/utils/checkpoints.py``:
``log_checkpoints`` :17-51, ``restore_from_checkpoint`` :54-104, ``get_path_to_checkpoint`` :107-136).
Because every module of this package keeps the reference's parameter and buffer names (weight-scale
buffers ``v1..vk``, moving-average buffers, batch-norm statistics), a file written by the reference loads
here unchanged and vice versa; loading invalidates the packed-weight caches of every ``QuantConv2d``
(``_load_from_state_dict`` hook), so the next eval-mode forward re-packs from the restored weights.
"""

from pathlib import Path
from typing import Optional, Tuple

import torch
import torch.nn as nn


def _unwrap(model: nn.Module) -> nn.Module:
    """The replica inside a ``DataParallel`` / ``DistributedDataParallel`` wrapper, else the model."""
    return model.module if isinstance(model, (nn.DataParallel, nn.parallel.DistributedDataParallel)) else model


def log_checkpoints(checkpoint_dir: Path, model: nn.Module, optimizer, scheduler, epoch: int) -> None:
    """Write ``checkpoint_<epoch>.pt`` under ``checkpoint_dir`` (created if missing)."""
    checkpoint_dir = Path(checkpoint_dir)
    checkpoint_dir.mkdir(exist_ok=True, parents=True)
    torch.save({'epoch': epoch,
                'model_state_dict': _unwrap(model).state_dict(),
                'optimizer_state_dict': optimizer.state_dict(),
                'scheduler_state_dict': scheduler.state_dict()},
               checkpoint_dir / f'checkpoint_{epoch}.pt')


def restore_from_checkpoint(model: nn.Module, optimizer, scheduler, checkpoint_path: str,
                            device: torch.device, strict_keys: bool = True) -> Tuple[nn.Module, object, object, int]:
    """Load model (and, when given, optimizer / scheduler) state; tensors are mapped onto ``device``
    whatever device they were saved from.  Returns ``(model, optimizer, scheduler, epoch)``."""
    state = torch.load(checkpoint_path, map_location=device)
    _unwrap(model).load_state_dict(state['model_state_dict'], strict=strict_keys)
    if optimizer is not None:
        optimizer.load_state_dict(state['optimizer_state_dict'])
        for slots in optimizer.state.values():
            for key, value in slots.items():
                if isinstance(value, torch.Tensor):
                    slots[key] = value.to(device)
    if scheduler is not None:
        scheduler.load_state_dict(state['scheduler_state_dict'])
    return model, optimizer, scheduler, state['epoch']


def get_path_to_checkpoint(experiment_path: Path, epoch: Optional[int] = None) -> str:
    """Path of ``<experiment>/checkpoints/checkpoint_<epoch>.pt``; the latest epoch when ``epoch`` is None.
    ``ValueError`` when the directory holds no checkpoint or not the requested one."""
    found = {}
    for path in (Path(experiment_path) / 'checkpoints').iterdir():
        found[int(path.name.split('_')[1].split('.')[0])] = path
    if not found:
        raise ValueError(f'No checkpoint exists in the experiment directory: {experiment_path}')
    if epoch is None:
        epoch = max(found)
    elif epoch not in found:
        raise ValueError(f'Could not find checkpoint for epoch {epoch}.')
    return str(found[epoch])
