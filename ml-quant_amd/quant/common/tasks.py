"""The classification task the driver scripts run (inference side).

``classification_task`` keeps the signature and the flow of the reference's ``quant/common/tasks.py:85-232`` for
``--skip-training`` (``:185-194``): device, test loader, ``get_model`` from the yaml ``model`` section, optional
checkpoint (``--restore-experiment`` / ``--init-from-checkpoint``), Loss / Top-1 / Top-5 metrics, ``evaluate``.
A knowledge-distillation config (``kd_config``) only matters for training: its student loss is replaced by the
plain ``model.loss`` for evaluation, as the reference's test metrics do (``:176-180``).
"""

import logging
from pathlib import Path
from typing import Callable, Dict, List, Optional, Tuple, Type

import torch

from quant.common.initialization import get_loss_fn, get_model
from quant.common.metrics import LossMetric, Top1Accuracy, TopKAccuracy
from quant.common.training import evaluate
from quant.data.data_loaders import QuantDataLoader
from quant.utils.checkpoints import get_path_to_checkpoint, restore_from_checkpoint


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


def classification_task(config: dict, experiment_root_directory: Path, data_loader_cls: Type[QuantDataLoader],
                        get_hooks: Optional[Callable] = None, restore_experiment: Optional[Path] = None
                        ) -> Tuple[List[Dict[str, float]], List[Dict[str, float]]]:
    """(training metrics per epoch, test metrics per epoch); only ``skip_training`` runs are supported."""
    if not config.get('skip_training'):
        raise NotImplementedError('training is outside the scope of this build: pass --skip-training')
    env, data_config, model_config = config['environment'], config['data'], config['model']
    logging.basicConfig(level=getattr(logging, str(config.get('log', {}).get('level', 'INFO'))))
    device = get_device(env['ngpus'], config.get('seed'), **env.get('cuda', {}))
    data_loader = data_loader_cls(**data_config)
    test_loader = data_loader.get_test_loader()
    loss_fn = get_loss_fn(model_config['loss'])
    model = get_model(model_config['architecture'], loss_fn, model_config['arch_config'], device, env['ngpus'])
    strict = model_config.get('strict_keys', True)
    if restore_experiment is not None:
        restore_from_checkpoint(model, None, None, get_path_to_checkpoint(Path(restore_experiment)), device, strict)
    elif config.get('init_from_checkpoint'):
        restore_from_checkpoint(model, None, None, config['init_from_checkpoint'], device, strict)
    test_metrics = {'Loss': LossMetric(loss_fn, accumulate=True), 'Top-1 Accuracy': Top1Accuracy(accumulate=True),
                    'Top-5 Accuracy': TopKAccuracy(5, accumulate=True)}
    hooks = get_hooks(config, experiment_root_directory, {}, test_metrics)[1] if get_hooks is not None else []
    computed = evaluate(model=model, test_loader=test_loader, metrics=test_metrics, device=device, epoch=1, hooks=hooks)
    data_loader.cleanup()
    return [], [computed]
