"""Command line + yaml -> one resolved config dictionary, for the driver scripts under ``examples/``.

Same contract as the reference's ``quant/common/parser.py`` (flag names :227-261, precedence :160-224, the two
``ValueError`` conditions :149-157) so its yaml files and command lines work unchanged, built differently: the flags
are a table, and a single ``resolve`` applies the layers in order --

    restored experiment's config.yaml  <  --config file  <  command-line flags  <  defaults for what is still missing
"""

from argparse import ArgumentParser, Namespace
from datetime import datetime
from pathlib import Path
from typing import Any, Callable, Dict, Optional

import torch
import yaml

#: flag -> (argparse keyword arguments, help text)
FLAGS = {
    '--config': (dict(type=str), 'Path to a yaml config file.'),
    '--experiment-name': (dict(type=str, default=None), 'Name of the experiment.'),
    '--ngpus': (dict(type=int, default=None), 'Number of GPUs. Use 0 for CPU.'),
    '--skip-training': (dict(action='store_true', default=False),
                        'Skip training and only run evaluation. Checkpoint must be passed in as well.'),
    '--restore-experiment': (dict(type=str), 'Path to experiments directory to restore checkpoint from.'),
    '--init-from-checkpoint': (dict(type=str), 'Path to model file to initialize model parameters.'),
}


def get_base_argument_parser(description: str) -> ArgumentParser:
    parser = ArgumentParser(description)
    for flag, (kwargs, text) in FLAGS.items():
        parser.add_argument(flag, help=text, **kwargs)
    return parser


def _validate_args(args: Namespace) -> None:
    """What cannot be resolved: nothing to read the model from, or two sources of initial weights."""
    problems = {
        '--config must be specified if not restoring from experiment.': not (args.restore_experiment or args.config),
        'Only one of --restore-experiment / --init-from-checkpoint can be set.':
            bool(args.restore_experiment and args.init_from_checkpoint),
    }
    for message, hit in problems.items():
        if hit:
            raise ValueError(message)


def _load_yaml(path: Path) -> Dict[str, Any]:
    with open(path) as handle:
        return yaml.safe_load(handle)


def _default_experiment_name(config_path: str, now: Optional[datetime] = None) -> str:
    return f"{(now or datetime.now()).strftime('%b%d_%H-%M-%S')}_{Path(config_path).stem}"


def parse_common_fields(args: Namespace, config: dict) -> None:
    """Apply the command-line layer and the defaults to ``config`` in place (the reference's helper of the same name)."""
    overrides = {'experiment_name': args.experiment_name, 'init_from_checkpoint': args.init_from_checkpoint or None}
    config.update({key: value for key, value in overrides.items() if value is not None})
    config['skip_training'] = args.skip_training
    if 'experiment_name' not in config or args.experiment_name is None:
        config['experiment_name'] = args.experiment_name or _default_experiment_name(config['config'])
    environment = config.get('environment')
    if not isinstance(environment, dict) or 'platform' not in environment:
        environment = config['environment'] = {'platform': 'local'}
    if args.ngpus is not None:
        environment['ngpus'] = args.ngpus
    environment.setdefault('ngpus', 1 if torch.cuda.is_available() else 0)


def resolve(args: Namespace) -> dict:
    """File layers, then flags, then defaults."""
    layers = []
    if args.restore_experiment:
        layers.append(_load_yaml(Path(args.restore_experiment) / 'config.yaml'))
    if args.config:
        layers.append(dict(_load_yaml(Path(args.config)), config=args.config))
    config = dict(layers[-1]) if layers else {}              # (--config REPLACES a restored config, it does not merge)
    parse_common_fields(args, config)
    if args.restore_experiment:
        config['restore_experiment'] = args.restore_experiment
    return config


def parse_config(args: Namespace, validator: Callable[[Namespace], None] = _validate_args) -> dict:
    """The resolved config: file contents with the command-line arguments applied on top."""
    validator(args)
    return resolve(args)
