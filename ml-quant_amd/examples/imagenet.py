#!/usr/bin/env python3
"""Driver in the shape of the reference's examples/imagenet/imagenet.py: yaml config + flags -> evaluation.

    python ml-quant_amd/examples/imagenet.py --config <the reference's examples/imagenet/*.yaml> --skip-training \
        [--ngpus 1] [--init-from-checkpoint checkpoint_N.pt]

Multi-GPU: launch one process per GPU with torchrun; every rank evaluates its slice of each batch and the logits
are all-gathered (quant.common.training.evaluate).  Test data is synthetic (quant/data/data_loaders.py).
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from quant.common.experiment import Experiment, LocalComputePlatform  # noqa: E402
from quant.common.parser import get_base_argument_parser, parse_config  # noqa: E402
from quant.common.tasks import classification_task  # noqa: E402
from quant.data.data_loaders import ImageNetDataLoader  # noqa: E402

if __name__ == '__main__':
    args = get_base_argument_parser('Driver script for running imagenet.').parse_args()
    config = parse_config(args)
    if 'RANK' in os.environ and config['environment']['ngpus'] > 0:
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=torch.device('cuda', torch.cuda.current_device()))
    platform = LocalComputePlatform(config.get('log', {}).get('root_experiments_dir', '.'))
    _, test = platform.run(Experiment(classification_task, config, ImageNetDataLoader))
    if not dist.is_initialized() or dist.get_rank() == 0:
        print(json.dumps(test[-1]))
    if dist.is_initialized():
        dist.destroy_process_group()
