"""yaml / CLI merge, metrics, the --skip-training evaluate loop and the driver scripts
(reference quant/common/parser.py:196-261, metrics.py, training.py:155-204, tasks.py:85-232, examples/*/*.py)."""

import json
import os
import subprocess
import sys

import pytest
import torch
import yaml

import detgen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# same sections and keys as the reference's examples/mnist/mnist_ls1_weight_ls2_activation.yaml
CONFIG = {
    'seed': None,
    'environment': {'platform': 'local', 'ngpus': 1, 'cuda': {'cudnn_deterministic': True, 'cudnn_benchmark': False}},
    'data': {'dataset_path': 'data/mnist/', 'train_batch_size': 64, 'test_batch_size': 50, 'workers': 4},
    'model': {'architecture': 'lenet5', 'loss': 'nll_loss',
              'arch_config': {'moving_average_mode': 'off', 'moving_average_momentum': 0.99, 'x_quant': 'ls-2',
                              'w_quant': 'ls-1', 'clamp': {'kind': 'symmetric', 'alpha': 2}, 'conv1_filters': 20,
                              'conv2_filters': 50, 'output_classes': 10}},
    'optimization': {'epochs': 10, 'optimizer': {'algorithm': 'adadelta', 'lr': 1.0},
                     'lr_scheduler': {'scheduler': 'step_lr', 'step_size': 1, 'gamma': 0.7}},
    'log': {'level': 'INFO', 'interval': 10, 'tensorboard': True},
}


def _write(tmp_path, cfg=CONFIG, name='mnist_ls2.yaml'):
    path = tmp_path / name
    path.write_text(yaml.safe_dump(cfg))
    return str(path)


def test_parse_config_merges_yaml_and_flags(tmp_path):
    from quant.common.parser import get_base_argument_parser, parse_config
    parser = get_base_argument_parser('test')
    cfg = parse_config(parser.parse_args(['--config', _write(tmp_path), '--ngpus', '0', '--skip-training']))
    assert cfg['environment']['ngpus'] == 0 and cfg['skip_training'] is True            # the flag beats the file
    assert cfg['experiment_name'].endswith('_mnist_ls2') and cfg['config'].endswith('mnist_ls2.yaml')
    assert cfg['model']['arch_config']['x_quant'] == 'ls-2' and 'init_from_checkpoint' not in cfg
    cfg = parse_config(parser.parse_args(['--config', _write(tmp_path), '--experiment-name', 'exp7',
                                          '--init-from-checkpoint', 'ck.pt']))
    assert cfg['experiment_name'] == 'exp7' and cfg['environment']['ngpus'] == 1 and cfg['skip_training'] is False
    assert cfg['init_from_checkpoint'] == 'ck.pt'
    bare = {k: v for k, v in CONFIG.items() if k != 'environment'}                       # no environment section
    cfg = parse_config(parser.parse_args(['--config', _write(tmp_path, bare, 'bare.yaml')]))
    assert cfg['environment'] == {'platform': 'local', 'ngpus': 1 if torch.cuda.is_available() else 0}
    with pytest.raises(ValueError):
        parse_config(parser.parse_args([]))
    with pytest.raises(ValueError):
        parse_config(parser.parse_args(['--restore-experiment', 'x', '--init-from-checkpoint', 'y']))
    # a restored experiment supplies the config
    exp = tmp_path / 'old'
    exp.mkdir()
    (exp / 'config.yaml').write_text(yaml.safe_dump(dict(CONFIG, config='orig.yaml')))
    cfg = parse_config(parser.parse_args(['--restore-experiment', str(exp), '--skip-training']))
    assert cfg['restore_experiment'] == str(exp) and cfg['model']['architecture'] == 'lenet5'


def test_metrics_match_direct_computation():
    from quant.common.metrics import LossMetric, Top1Accuracy, TopKAccuracy
    out = torch.log_softmax(detgen.normal('plumb.out', (37, 10)), dim=1)
    tgt = torch.arange(37) % 10
    loss, top1, top5 = LossMetric(torch.nn.functional.nll_loss, True), Top1Accuracy(True), TopKAccuracy(5, True)
    for lo, hi in ((0, 16), (16, 32), (32, 37)):
        for m in (loss, top1, top5):
            m.update(out[lo:hi], tgt[lo:hi])
    assert loss.compute() == pytest.approx(float(torch.nn.functional.nll_loss(out, tgt)), rel=1e-6)
    assert top1.compute() == pytest.approx(float((out.argmax(1) == tgt).float().mean()))
    assert top5.compute() == pytest.approx(float((out.topk(5, 1).indices == tgt.view(-1, 1)).any(1).float().mean()))
    assert top1.n_examples == 37 and '/37 (' in str(top1)
    last = Top1Accuracy(False)
    last.update(out[:16], tgt[:16])
    last.update(out[32:], tgt[32:])
    assert last.n_examples == 5 and last.compute() == pytest.approx(float((out[32:].argmax(1) == tgt[32:]).float().mean()))
    top1.reset()
    assert top1.n_examples == 0


def test_skip_training_task_and_checkpoint_on_cpu(tmp_path):
    """tasks.py:185-194: model from the yaml model section, optional checkpoint, evaluate() over the test loader."""
    from quant.common.experiment import Experiment, LocalComputePlatform
    from quant.common.parser import get_base_argument_parser, parse_config
    from quant.common.tasks import classification_task
    from quant.data.data_loaders import MNISTDataLoader
    from quant.models.lenet import QLeNet5
    from quant.utils.checkpoints import log_checkpoints
    src = QLeNet5(loss_fn=torch.nn.functional.nll_loss, **CONFIG['model']['arch_config'])
    detgen.fill_module(src, seed=12)
    with torch.no_grad():
        src.conv2.w_approximate.v1.copy_(src.conv2.weight.abs().mean(dim=(1, 2, 3)))
    opt = torch.optim.SGD(src.parameters(), lr=0.1)
    log_checkpoints(tmp_path / 'ck', src, opt, torch.optim.lr_scheduler.StepLR(opt, 1), 4)
    parser = get_base_argument_parser('t')
    cfg = parse_config(parser.parse_args(['--config', _write(tmp_path), '--ngpus', '0', '--skip-training', '--experiment-name',
                                          'e1', '--init-from-checkpoint', str(tmp_path / 'ck' / 'checkpoint_4.pt')]))
    cfg['log']['root_experiments_dir'] = str(tmp_path)
    train, test = LocalComputePlatform(str(tmp_path)).run(Experiment(classification_task, cfg, MNISTDataLoader))
    assert train == [] and set(test[0]) == {'Loss', 'Top-1 Accuracy', 'Top-5 Accuracy'}
    assert (tmp_path / 'experiments' / 'e1' / 'config.yaml').exists()
    # the same numbers by hand: the restored weights on the loader's synthetic test set
    loader = MNISTDataLoader(**cfg['data']).get_test_loader()
    src.eval()
    with torch.no_grad():
        outs, tgts = zip(*[(src(d), t) for d, t in loader])
    out, tgt = torch.cat(outs), torch.cat(tgts)
    assert len(tgt) == 200
    assert test[0]['Top-1 Accuracy'] == pytest.approx(float((out.argmax(1) == tgt).float().mean()))
    assert test[0]['Loss'] == pytest.approx(float(torch.nn.functional.nll_loss(out, tgt)), rel=1e-5)
    # without --skip-training the task trains (tasks.py:195-228): two epochs, metrics per epoch, a checkpoint per epoch
    cfg2 = json.loads(json.dumps(dict(cfg, skip_training=False)))
    cfg2['optimization']['epochs'] = 2
    cfg2['log']['save_model_freq'] = 1
    cfg2['experiment_name'] = 'e2'
    tr, te = classification_task(cfg2, tmp_path, MNISTDataLoader)
    assert len(tr) == len(te) == 2 and set(tr[0]) == {'Loss', 'Top-1 Accuracy', 'Top-5 Accuracy'}
    assert tr[1]['Loss'] < tr[0]['Loss']                                    # it learns its 200 synthetic samples
    assert (tmp_path / 'e2' / 'checkpoints' / 'checkpoint_2.pt').exists()


def test_driver_script_runs_a_config(tmp_path):
    cfg = json.loads(json.dumps(CONFIG))
    cfg['data']['test_batch_size'] = 16
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'ml-quant_amd', 'examples', 'mnist.py'), '--config',
                        _write(tmp_path, cfg), '--skip-training', '--ngpus', '0', '--experiment-name', 'drv'],
                       cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    metrics = json.loads(r.stdout.strip().splitlines()[-1])
    assert set(metrics) == {'Loss', 'Top-1 Accuracy', 'Top-5 Accuracy'} and 0.0 <= metrics['Top-5 Accuracy'] <= 1.0


@pytest.mark.gpu
def test_skip_training_task_on_the_gpu(tmp_path):
    """`--skip-training --init-from-checkpoint ...` (tasks.py:185-194 of the reference) with ngpus = 1 against the same
    command with ngpus = 0: the evaluate loop drives the HIP path, and its metrics equal the CPU run's within what
    the free-running ls-2 solve can move the logits (measured here, bounded by tests/test_gpu_parity.FREE_LIMIT):
    |dLoss| <= 2 max|dlogit| (cross entropy is 2-Lipschitz in the sup norm), and a top-k count can only change for a
    sample whose k-th margin is below 2 max|dlogit|."""
    from quant.binary.binary_conv import QuantConv2d
    from quant.common.initialization import get_loss_fn, get_model
    from quant.common.tasks import classification_task
    from quant.data.data_loaders import CIFAR100DataLoader
    from quant.common.parser import get_base_argument_parser, parse_config
    from quant.utils.checkpoints import log_checkpoints
    cfg = json.loads(json.dumps(CONFIG))
    cfg['data'] = {'dataset_path': 'data/cifar100/', 'train_batch_size': 128, 'test_batch_size': 50, 'workers': 16}
    layer = {'x_quant': 'ls-2', 'w_quant': 'ls-1', 'clamp': {'kind': 'symmetric', 'alpha': 2}, 'double_shortcut': True}
    cfg['model'] = {'architecture': 'resnet', 'loss': 'cross_entropy', 'arch_config': {
        'moving_average_mode': 'off', 'moving_average_momentum': 0.99, 'block': 'xnor',
        'layer0': {'n_in_channels': 64, 'kernel_size': 3, 'stride': 1, 'padding': 1, 'bias': False, 'maxpool': {'type': 'identity'}},
        'layer1': layer, 'layer2': layer, 'layer3': layer, 'layer4': layer, 'nonlins': ['relu', 'relu'],
        'num_blocks': [2, 2, 2, 2], 'output_classes': 100}}
    # a checkpoint with cached weight scales (a never-trained module has all-zero scales: weight_quantization.py:25)
    torch.manual_seed(7)
    src = get_model('resnet', get_loss_fn('cross_entropy'), cfg['model']['arch_config'], torch.device('cpu'), 0)
    with torch.no_grad():
        for m in src.modules():
            if isinstance(m, QuantConv2d):
                m.w_approximate.v1.copy_(m.weight.abs().mean(dim=(1, 2, 3)))
    opt = torch.optim.SGD(src.parameters(), lr=0.1)
    log_checkpoints(tmp_path / 'ck', src, opt, torch.optim.lr_scheduler.StepLR(opt, 1), 1)
    results = {}
    for ngpus in (0, 1):
        args = get_base_argument_parser('t').parse_args(['--config', _write(tmp_path, cfg, 'cifar.yaml'), '--skip-training', '--ngpus',
                                                         str(ngpus), '--init-from-checkpoint', str(tmp_path / 'ck' / 'checkpoint_1.pt')])
        _, test = classification_task(parse_config(args), tmp_path, CIFAR100DataLoader)
        results[ngpus] = test[0]
        assert set(test[0]) == {'Loss', 'Top-1 Accuracy', 'Top-5 Accuracy'}
    assert 'liblsq_hip.so' in open('/proc/self/maps').read()
    # how far the two paths' logits are apart on this test set
    loader = CIFAR100DataLoader(**cfg['data']).get_test_loader()
    src.eval()
    dev = src.__class__(loss_fn=src.loss_fn, **cfg['model']['arch_config'])
    dev.load_state_dict(src.state_dict())
    dev = dev.eval().to('cuda:0')
    with torch.no_grad():
        lc = torch.cat([src(d) for d, _ in loader])
        lg = torch.cat([dev(d.to('cuda:0')).cpu() for d, _ in loader])
        tgt = torch.cat([t for _, t in loader])
    d = float((lc - lg).abs().max())
    assert d <= 0.1 * float(lc.abs().max()), d                     # the free-running bound of the whole network
    assert abs(results[1]['Loss'] - results[0]['Loss']) <= 2 * d + 1e-6
    for k, name in ((1, 'Top-1 Accuracy'), (5, 'Top-5 Accuracy')):
        top = lc.topk(k + 1, dim=1).values
        tval = lc.gather(1, tgt.view(-1, 1)).squeeze(1)
        kth, nxt = top[:, k - 1], top[:, k]
        fragile = ((tval - nxt).abs() < 2 * d) | ((tval - kth).abs() < 2 * d)     # samples whose membership in the top k can flip
        assert abs(results[1][name] - results[0][name]) <= float(fragile.float().mean()) + 1e-9, name
