"""Experiment bookkeeping for the driver scripts: the resolved config is written to
``<root>/<experiment_name>/config.yaml`` and the task is run (reference ``quant/common/experiment.py:60-125`` and
``compute_platform.py:73-114`` without the TensorBoard subprocess)."""

from pathlib import Path
from typing import Callable, Optional

import yaml


class Experiment:
    def __init__(self, task: Callable, config: dict, data_loader_cls, get_hooks: Optional[Callable] = None) -> None:
        self.task, self.config, self.data_loader_cls, self.get_hooks = task, config, data_loader_cls, get_hooks
        self.name = config['experiment_name']

    def run(self, experiments_dir: Path):
        root = Path(experiments_dir)
        (root / self.name).mkdir(parents=True, exist_ok=True)
        with open(root / self.name / 'config.yaml', 'w') as f:
            yaml.safe_dump(self.config, f)
        restore = self.config.get('restore_experiment')
        return self.task(self.config, root, self.data_loader_cls, self.get_hooks, Path(restore) if restore else None)


class LocalComputePlatform:
    def __init__(self, root_experiments_dir: str = '.') -> None:
        self.root = Path(root_experiments_dir) / 'experiments'

    def run(self, experiment: Experiment):
        return experiment.run(self.root)
