"""YAML config + command-line merge for the driver scripts.

Interface of the reference's ``quant/common/parser.py``: ``get_base_argument_parser`` (:227-261, the six flags
``--config --experiment-name --ngpus --skip-training --restore-experiment --init-from-checkpoint``),
``parse_config`` (:196-224) and ``parse_common_fields`` (:160-193) with the same precedence: a restored
experiment's ``config.yaml`` first, ``--config`` replaces it, command-line values override the file
(``--ngpus`` beats ``environment.ngpus``), and the same two ``ValueError`` s.
"""

from argparse import ArgumentParser, Namespace
from datetime import datetime
from pathlib import Path
from typing import Callable

import torch
import yaml


def _validate_args(args: Namespace) -> None:
    if not args.restore_experiment and not args.config:
        raise ValueError('--config must be specified if not restoring from experiment.')
    if args.restore_experiment and args.init_from_checkpoint:
        raise ValueError('Only one of --restore-experiment / --init-from-checkpoint can be set.')


def parse_common_fields(args: Namespace, config: dict) -> None:
    """Fill ``experiment_name``, ``environment``, ``skip_training`` and ``init_from_checkpoint`` from the flags."""
    if args.experiment_name is not None:
        config['experiment_name'] = args.experiment_name
    else:
        stamp = datetime.now().strftime('%b%d_%H-%M-%S')
        config['experiment_name'] = f"{stamp}_{Path(config['config']).stem}"
    if 'environment' not in config or 'platform' not in config['environment']:
        config['environment'] = {'platform': 'local'}
    if args.ngpus is not None:
        config['environment']['ngpus'] = args.ngpus
    if 'ngpus' not in config['environment']:
        config['environment']['ngpus'] = 1 if torch.cuda.is_available() else 0
    config['skip_training'] = args.skip_training
    if args.init_from_checkpoint:
        config['init_from_checkpoint'] = args.init_from_checkpoint


def parse_config(args: Namespace, validator: Callable[[Namespace], None] = _validate_args) -> dict:
    """The resolved config: file contents with the command-line arguments applied on top."""
    validator(args)
    config: dict = {}
    if args.restore_experiment:
        with open(Path(args.restore_experiment) / 'config.yaml') as f:
            config = yaml.safe_load(f)
    if args.config:
        with open(args.config) as f:
            config = yaml.safe_load(f)
        config['config'] = args.config
    parse_common_fields(args, config)
    if args.restore_experiment:
        config['restore_experiment'] = args.restore_experiment
    return config


def get_base_argument_parser(description: str) -> ArgumentParser:
    parser = ArgumentParser(description)
    parser.add_argument('--config', type=str, help='Path to a yaml config file.')
    parser.add_argument('--experiment-name', type=str, default=None, help='Name of the experiment.')
    parser.add_argument('--ngpus', type=int, default=None, help='Number of GPUs. Use 0 for CPU.')
    parser.add_argument('--skip-training', default=False, action='store_true',
                        help='Skip training and only run evaluation. Checkpoint must be passed in as well.')
    parser.add_argument('--restore-experiment', type=str, help='Path to experiments directory to restore checkpoint from.')
    parser.add_argument('--init-from-checkpoint', type=str, help='Path to model file to initialize model parameters.')
    return parser
