"""The classification task the driver scripts run.

``classification_task`` keeps the signature and the flow of the reference's ``quant/common/tasks.py:85-232``: device,
loaders, ``get_model`` from the yaml ``model`` section, optional teacher + distillation loss (``kd_config``, :33-82),
optimizer and per-batch learning-rate scheduler, optional checkpoint (``--restore-experiment`` /
``--init-from-checkpoint``), Loss / Top-1 / Top-5 metrics; then either one ``evaluate`` (``--skip-training``, :185-194)
or ``epochs`` rounds of ``train`` + ``evaluate`` with checkpoints every ``save_model_freq`` epochs (:195-228).  The test
loss is always the plain ``model.loss`` (:176-180).
"""

import logging
from pathlib import Path
from typing import Callable, Dict, List, Optional, Tuple, Type

import torch

from functools import partial

from quant.common.initialization import get_loss_fn, get_lr_scheduler, get_model, get_optimizer
from quant.common.metrics import LossMetric, Top1Accuracy, TopKAccuracy
from quant.common.training import evaluate, train
from quant.data.data_loaders import QuantDataLoader
from quant.utils.checkpoints import get_path_to_checkpoint, log_checkpoints, restore_from_checkpoint
from quant.utils.kd_criterion import kd_criterion


def get_device(ngpus: int, seed: Optional[int] = None, **cuda_flags) -> torch.device:
    """``cuda`` (the current device of this process) when ``ngpus`` > 0, else ``cpu``; seeds torch when asked.
    (The reference also sets cuDNN flags here, initialization.py:50-94; they have no counterpart on this path.)"""
    if seed is not None:
        torch.manual_seed(seed)
    if ngpus > 0:
        if not torch.cuda.is_available():
            raise ValueError('ngpus > 0 but no GPU is visible.')
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


def get_teacher_and_kd_loss(teacher_config_path: str, teacher_checkpoint_path: str, train_mode: bool, criterion_config: dict,
                            device: torch.device, ngpus: int, freeze_teacher: bool = True, strict_keys: bool = True):
    """(teacher, distillation loss) of a ``kd_config`` section (tasks.py:33-82): the teacher is built from ITS yaml's
    ``model`` section, restored from its checkpoint, frozen unless told otherwise and put in train or eval mode."""
    import yaml
    with open(teacher_config_path) as f:
        teacher_model = yaml.safe_load(f)['model']
    teacher = get_model(teacher_model['architecture'], get_loss_fn(teacher_model['loss']), teacher_model['arch_config'],
                        device, ngpus)
    restore_from_checkpoint(teacher, None, None, teacher_checkpoint_path, device, strict_keys)
    if freeze_teacher:
        for p in teacher.parameters():
            p.requires_grad_(False)
    teacher.train(train_mode)
    return teacher, partial(kd_criterion, freeze_teacher=freeze_teacher, **criterion_config)


def classification_task(config: dict, experiment_root_directory: Path, data_loader_cls: Type[QuantDataLoader],
                        get_hooks: Optional[Callable] = None, restore_experiment: Optional[Path] = None
                        ) -> Tuple[List[Dict[str, float]], List[Dict[str, float]]]:
    """(training metrics per epoch, test metrics per epoch)."""
    env, data_config, model_config = config['environment'], config['data'], config['model']
    log_config = config.get('log', {})
    logging.basicConfig(level=getattr(logging, str(log_config.get('level', 'INFO'))))
    skip = bool(config.get('skip_training'))
    device = get_device(env['ngpus'], config.get('seed'), **env.get('cuda', {}))
    data_loader = data_loader_cls(**data_config)
    train_loader = None if skip else data_loader.get_train_loader()
    test_loader = data_loader.get_test_loader()
    strict = model_config.get('strict_keys', True)
    teacher, loss_fn = None, get_loss_fn(model_config['loss'])
    if 'kd_config' in model_config and not skip:            # (a teacher only matters for training)
        teacher, loss_fn = get_teacher_and_kd_loss(device=device, ngpus=env['ngpus'], strict_keys=strict,
                                                   **model_config['kd_config'])
    model = get_model(model_config['architecture'], loss_fn, model_config['arch_config'], device, env['ngpus'])
    optimizer = scheduler = None
    epochs = 1
    if not skip:
        opt_config = config['optimization']
        epochs = opt_config['epochs']
        optimizer = get_optimizer(model.parameters(), opt_config['optimizer'])
        scheduler = get_lr_scheduler(optimizer, opt_config['lr_scheduler'], epochs, len(train_loader))
    start_epoch = 1
    if restore_experiment is not None:
        _, optimizer, scheduler, last = restore_from_checkpoint(model, optimizer, scheduler,
                                                                get_path_to_checkpoint(Path(restore_experiment)), device, strict)
        start_epoch = last + 1
    elif config.get('init_from_checkpoint'):
        restore_from_checkpoint(model, None, None, config['init_from_checkpoint'], device, strict)
    train_metrics = {'Loss': LossMetric(loss_fn, accumulate=True), 'Top-1 Accuracy': Top1Accuracy(accumulate=True),
                     'Top-5 Accuracy': TopKAccuracy(5, accumulate=True)}
    test_metrics = {'Loss': LossMetric(get_loss_fn(model_config['loss']), accumulate=True),
                    'Top-1 Accuracy': Top1Accuracy(accumulate=True), 'Top-5 Accuracy': TopKAccuracy(5, accumulate=True)}
    train_hooks, test_hooks = ([], [])
    if get_hooks is not None:
        train_hooks, test_hooks = get_hooks(config, experiment_root_directory, train_metrics, test_metrics)
    train_epochs: List[Dict[str, float]] = []
    test_epochs: List[Dict[str, float]] = []
    if skip:
        test_epochs.append(evaluate(model=model, test_loader=test_loader, metrics=test_metrics, device=device, epoch=1,
                                    hooks=test_hooks))
    else:
        for epoch in range(start_epoch, start_epoch + epochs):
            train_epochs.append(train(model=model, train_loader=train_loader, metrics=train_metrics, optimizer=optimizer,
                                      scheduler=scheduler, device=device, epoch=epoch,
                                      log_interval=log_config.get('interval', 10), hooks=train_hooks, teacher=teacher))
            test_epochs.append(evaluate(model=model, test_loader=test_loader, metrics=test_metrics, device=device,
                                        epoch=epoch, hooks=test_hooks))
            if epoch % log_config.get('save_model_freq', 1) == 0 or epoch == epochs:
                log_checkpoints(Path(experiment_root_directory) / config.get('experiment_name', 'experiment') / 'checkpoints',
                                model, optimizer, scheduler, epoch)
    data_loader.cleanup()
    return train_epochs, test_epochs
