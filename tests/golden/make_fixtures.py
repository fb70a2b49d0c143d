#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE itself.

Run in the build container only (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference python tests/golden/make_fixtures.py

Inputs and parameters come from ``detgen.py`` (frozen numpy RandomState streams), so the
``.npz`` files hold expected outputs (plus the few hand-made edge-case inputs).  Nothing
from the reference's source is stored: only tensors it computed.
"""

import json
import os
import sys

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get('LSQ_REFERENCE_ROOT', '/root/reference')
sys.path.insert(0, HERE)
if REF not in sys.path:
    sys.path.insert(0, REF)

import detgen  # noqa: E402
import quant.binary.optimal as r_opt  # noqa: E402
import quant.binary.quantization as r_q  # noqa: E402
from quant.binary.binary_conv import QuantConv2d as RefQuantConv2d  # noqa: E402
from quant.binary.ste import binary_sign as r_sign  # noqa: E402
from quant.models.lenet import QLeNet5 as RefLeNet  # noqa: E402
from quant.models.resnet import QResNet as RefResNet  # noqa: E402

assert os.path.realpath(r_q.__file__).startswith(os.path.realpath(REF)), r_q.__file__
torch.set_num_threads(8)


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print(f'{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB')


def packbits(t):
    """{-1,+1} tensor -> packed uint8 (1 = +1), C order."""
    return np.packbits((t.detach().numpy() > 0).astype(np.uint8).reshape(-1))


# ----------------------------------------------------------------------------- F1 sign table
def f1_sign():
    x = torch.tensor([42, -42, 42, 42, 0, -1, 1, -4.2, 4.2, 0.0, -0.0, 1e-45, -1e-45,
                      1e-38, -1e-38, 3.4e38, -3.4e38], dtype=torch.float32)
    save('f1_sign', x=x, sign=r_sign(x))


# ----------------------------------------------------------------------------- F2/F4 quantizers
def f24_quantizers():
    x = detgen.normal('f24.x', (4, 64, 14, 14), scale=1.3).clamp(-3, 3)
    out = {}
    v1, xq = r_q.quantizer_ls_1(x)
    out.update(ls1_v1=v1, ls1_xq=xq)
    v1, v2, xq = r_q.quantizer_ls_2(x)
    out.update(ls2_v1=v1, ls2_v2=v2, ls2_xq=xq)
    v1, v2, xq = r_q.quantizer_ls_2(x, skip=1)
    out.update(ls2s1_v1=v1, ls2s1_v2=v2)
    v1, xq = r_q.quantizer_ls_ternary(x)
    out.update(lst_v1=v1, lst_xq=xq)
    vs, xq = r_q.quantizer_gf(x, 2)
    out.update(gf2_v1=vs[0], gf2_v2=vs[1], gf2_xq=xq)
    vs, xq = r_q.quantizer_gf(x, 3)
    out.update(gf3_v1=vs[0], gf3_v2=vs[1], gf3_v3=vs[2], gf3_b=packbits(torch.sign(xq)))
    # injected scales (eval-mode / moving-average style calls)
    inj1 = detgen.uniform('f24.inj1', (4,), 0.6, 1.4)
    inj2 = detgen.uniform('f24.inj2', (4,), 0.2, 0.6)
    _, _, xq = r_q.quantizer_ls_2(x, inj1, inj2)
    out.update(inj1=inj1, inj2=inj2, ls2_inj_xq=xq)
    _, xq = r_q.quantizer_ls_ternary(x, inj1)
    out.update(lst_inj_xq=xq)
    _, xq = r_q.quantizer_ls_1(x, inj1)
    out.update(ls1_inj_xq=xq)
    # v2 only injected v1 (quantization.py:82-85)
    _, v2, _ = r_q.quantizer_ls_2(x, inj1)
    out.update(ls2_v2_from_inj1=v2)
    save('f24_quantizers', **out)


# ----------------------------------------------------------------------------- F3 solver internals
def solver_case(tag, rows, out):
    for ternary in (False, True):
        for skip in (1, 3):
            a = rows[..., ::skip].abs()
            mask, vals = r_opt.compute_mask(a, ternary)
            sizes = mask.sum(dim=1)
            if ternary:
                vals, sizes = r_opt._handle_ternary_min_gt_half_avg(a, vals, sizes.clone())
            key = f'{tag}_t{int(ternary)}_s{skip}'
            try:
                v1 = r_opt.opt_v1(rows, ternary, skip)
            except IndexError:          # no candidate in any row: argmin over an empty dim
                out[key + '_raises'] = np.array(1)
                continue
            lists = torch.split(vals, sizes.tolist())
            kmax = max(int(s) for s in sizes.tolist())
            padded = torch.zeros(rows.shape[0], kmax)
            for r, c in enumerate(lists):
                padded[r, :c.numel()] = c
            costs = r_opt.cost_function(a, padded, ternary)
            out[key + '_sizes'] = sizes
            out[key + '_cands'] = padded
            out[key + '_costs'] = costs
            out[key + '_v1'] = v1.view(-1)
            # positions of mask hits (inner index + 1 = sorted position)
            pos = torch.full((rows.shape[0], kmax), -1, dtype=torch.int64)
            for r in range(rows.shape[0]):
                idx = torch.nonzero(mask[r]).flatten() + 1
                pos[r, :idx.numel()] = idx
            out[key + '_pos'] = pos


def f3_solver():
    out = {}
    # long rows: the ResNet-18 layer-4 row length (512*7*7 = 25088 -> 8363 at skip 3)
    rows = detgen.normal('f3.long', (4, 25088), scale=1.0).clamp(-3, 3)
    solver_case('long', rows, out)
    # relu-like rows (half exact zeros) and saturated rows (clamp alpha=0.5)
    relu = detgen.normal('f3.relu', (4, 3000)).clamp(min=0)
    solver_case('relu', relu, out)
    sat = detgen.normal('f3.sat', (4, 3001)).clamp(-0.5, 0.5)
    solver_case('sat', sat, out)
    # short rows incl. lengths not divisible by 3
    for n in (3, 4, 5, 7, 10, 11, 64):
        solver_case(f'short{n}', detgen.normal(f'f3.short{n}', (6, n)), out)
    # all-equal and min>mean/2 rows mixed with ordinary rows (tests/binary/test_quantization.py:95-111)
    mix = detgen.uniform('f3.mix', (8, 768))
    mix[1] = 2.0
    mix[5] = -3.0
    mix[6] = detgen.uniform('f3.mix6', (768,), 1.0, 1.2)
    out['mix_rows'] = mix
    solver_case('mix', mix, out)
    save('f3_solver', **out)


# ----------------------------------------------------------------------------- F5 QuantConv2d
PAIRS = [('ls-1', 'ls-1'), ('ls-2', 'ls-1'), ('ls-T', 'ls-1'), ('gf-2', 'ls-1'),
         ('fp', 'ls-1'), ('fp', 'fp'), ('ls-2', 'ls-2'), ('ls-1', 'gf-2'), ('ls-T', 'ls-T')]


def make_ref_conv(xs, ws, cin, cout, k, clamp, **kw):
    conv = RefQuantConv2d(xs, ws, cin, cout, k, clamp, **kw)
    tag = f'f5.w.{cin}.{cout}.{k}'
    fan_in = int(np.prod(conv.weight.shape[1:]))
    with torch.no_grad():
        conv.weight.copy_(detgen.normal(tag, conv.weight.shape, scale=fan_in ** -0.5))
        if conv.bias is not None:
            conv.bias.copy_(detgen.normal(tag + '.b', conv.bias.shape, scale=0.1))
    conv.train()
    return conv


def f5_conv():
    out = {}
    x = detgen.normal('f5.x', (2, 64, 14, 14), scale=1.2)
    for xs, ws in PAIRS:
        for stride in (1, 2):
            for alpha in (2, 3):
                if alpha == 3 and xs not in ('ls-2', 'ls-1'):
                    continue
                clamp = {'kind': 'symmetric', 'alpha': alpha}
                conv = make_ref_conv(xs, ws, 64, 64, 3, clamp, stride=stride, padding=1, bias=True)
                with torch.no_grad():
                    conv(x)                      # train-mode forward populates weight scales
                    conv.eval()
                    y = conv(x)
                key = f'{xs}_{ws}_s{stride}_a{alpha}'
                out[key + '_y'] = y
                for name, buf in conv.w_approximate.named_buffers():
                    out[key + '_w_' + name] = buf
                # activation scales the reference used (recomputed from the same input)
                xc = conv.clamping_fn(x)
                if xs == 'ls-2':
                    v1, v2, _ = r_q.quantizer_ls_2(xc)
                    out[key + '_xv1'], out[key + '_xv2'] = v1, v2
                elif xs == 'ls-T':
                    out[key + '_xv1'] = r_q.quantizer_ls_ternary(xc)[0]
    # LeNet conv2 geometry: Cin=20 (not a multiple of 64), 5x5, no padding, identity clamp
    xl = detgen.normal('f5.xl', (2, 20, 12, 12))
    for xs, ws in [('ls-1', 'ls-1'), ('ls-2', 'ls-1'), ('fp', 'ls-1'), ('gf-2', 'ls-1'), ('ls-T', 'ls-1')]:
        conv = make_ref_conv(xs, ws, 20, 50, 5, None, stride=1)
        with torch.no_grad():
            conv(xl)
            conv.eval()
            out[f'lenet_{xs}_{ws}_y'] = conv(xl)
            out[f'lenet_{xs}_{ws}_w_v1'] = conv.w_approximate.v1
    # never-trained module in eval: all-zero weight scales (weight_quantization.py:25)
    conv = make_ref_conv('ls-1', 'ls-1', 64, 64, 3, None, padding=1)
    conv.eval()
    with torch.no_grad():
        out['untrained_y'] = conv(x)
    # dilation / groups / rectangular kernel / asymmetric stride+padding
    geo = dict(stride=(2, 1), padding=(2, 1), dilation=(2, 1), groups=2, bias=True)
    conv = make_ref_conv('ls-2', 'ls-1', 64, 64, (3, 2), {'kind': 'symmetric', 'alpha': 2}, **geo)
    with torch.no_grad():
        conv(x)
        conv.eval()
        out['geo_y'] = conv(x)
        out['geo_w_v1'] = conv.w_approximate.v1
    save('f5_conv', **out)


# ----------------------------------------------------------------------------- F6/F7 models
def set_weight_scales(model):
    """Populate weight-quantizer buffers exactly as a train-mode forward would
    (weight_quantization.py:29-31), without touching BatchNorm statistics."""
    for m in model.modules():
        if isinstance(m, RefQuantConv2d):
            wq = m.w_approximate
            if hasattr(wq, 'v1'):
                wq.train()
                with torch.no_grad():
                    wq(m.weight)
                wq.eval()


def arch_from_yaml(rel):
    with open(os.path.join(REF, 'examples', rel)) as f:
        cfg = yaml.safe_load(f)
    return cfg['model']['arch_config']


def capture_scales(model, x):
    rec = {}
    hooks = []
    names = {m: n for n, m in model.named_modules()}

    def pre(mod, args):
        xc = mod.clamping_fn(args[0])
        aq = type(mod.x_approximate).__name__
        if aq.endswith('LS2'):
            v1, v2, _ = r_q.quantizer_ls_2(xc)
            rec[names[mod]] = torch.stack([v1, v2])
        elif aq.endswith('LST'):
            rec[names[mod]] = r_q.quantizer_ls_ternary(xc)[0].view(1, -1)
        elif aq.endswith('LS1'):
            rec[names[mod]] = r_q.quantizer_ls_1(xc)[0].view(1, -1)
    for m in model.modules():
        if isinstance(m, RefQuantConv2d):
            hooks.append(m.register_forward_pre_hook(pre))
    with torch.no_grad():
        y = model(x)
    for h in hooks:
        h.remove()
    return y, rec


def f6_models():
    out = {}
    cases = [
        ('imagenet_ls1w_ls2a', 'imagenet/imagenet_ls1_weight_ls2_activation_kd.yaml', (2, 3, 64, 64)),
        ('imagenet_ls1w_fpa', 'imagenet/imagenet_ls1_weight_fp_activation_kd.yaml', (2, 3, 64, 64)),
        ('imagenet_ls1w_lsTa', 'imagenet/imagenet_ls1_weight_lsT_activation_kd.yaml', (2, 3, 64, 64)),
        ('imagenet_ls1w_gf2a', 'imagenet/imagenet_ls1_weight_gf2_activation_kd.yaml', (2, 3, 64, 64)),
        ('cifar100_ls1', 'cifar100/cifar100_ls1_kd.yaml', (4, 3, 32, 32)),
    ]
    for tag, rel, shape in cases:
        arch = arch_from_yaml(rel)
        model = RefResNet(loss_fn=torch.nn.functional.cross_entropy, **arch)
        detgen.fill_module(model, seed=1)
        set_weight_scales(model)
        model.eval()
        x = detgen.normal(tag + '.x', shape)
        y, rec = capture_scales(model, x)
        out[tag + '_arch'] = np.frombuffer(json.dumps(arch).encode(), dtype=np.uint8)
        out[tag + '_logits'] = y
        for name, sc in rec.items():
            out[f'{tag}_scales_{name}'] = sc
    # single XnorBasicBlock, double shortcut, stride 2
    from quant.models.resnet import XnorBasicBlock
    blk = XnorBasicBlock(64, 128, 'ls-2', 'ls-1', ['relu', 'relu'], stride=2, double_shortcut=True,
                         clamp={'kind': 'symmetric', 'alpha': 3})
    detgen.fill_module(blk, seed=2)
    set_weight_scales(blk)
    blk.eval()
    with torch.no_grad():
        out['block_y'] = blk(detgen.normal('block.x', (2, 64, 16, 16)))
    save('f6_models', **out)


def f7_lenet():
    out = {}
    for tag, rel in [('mnist_ls1w_fpa', 'mnist/mnist_ls1_weight_fp_activation.yaml'),
                     ('mnist_ls1', 'mnist/mnist_ls1.yaml'),
                     ('mnist_ls1w_ls2a', 'mnist/mnist_ls1_weight_ls2_activation.yaml')]:
        arch = arch_from_yaml(rel)
        model = RefLeNet(loss_fn=torch.nn.functional.nll_loss, **arch)
        detgen.fill_module(model, seed=3)
        set_weight_scales(model)
        model.eval()
        x = detgen.normal(tag + '.x', (64, 1, 28, 28))
        with torch.no_grad():
            out[tag + '_logp'] = model(x)
        out[tag + '_arch'] = np.frombuffer(json.dumps(arch).encode(), dtype=np.uint8)
    # config 0 as BASELINE.json words it: mnist_fp.yaml with w_quant overridden to ls-1
    arch = arch_from_yaml('mnist/mnist_fp.yaml')
    arch['w_quant'] = 'ls-1'
    assert arch == arch_from_yaml('mnist/mnist_ls1_weight_fp_activation.yaml')
    # state_dict key contract (checkpoint row)
    arch = arch_from_yaml('imagenet/imagenet_ls1_weight_ls2_activation_kd.yaml')
    keys = list(RefResNet(loss_fn=None, **arch).state_dict().keys())
    out['resnet_ls2_state_keys'] = np.frombuffer('\n'.join(keys).encode(), dtype=np.uint8)
    arch = arch_from_yaml('mnist/mnist_ls1_weight_gf2_activation.yaml')
    keys = list(RefLeNet(loss_fn=None, **arch).state_dict().keys())
    out['lenet_gf2_state_keys'] = np.frombuffer('\n'.join(keys).encode(), dtype=np.uint8)
    save('f7_lenet', **out)


# ----------------------------------------------------------------------------- F8 moving-average modes + checkpoints
def tiny_resnet_arch(mode, x_quant='ls-2'):
    """A QResNet small enough for a fixture with the real structure (xnor blocks, double shortcuts, two
    stride-2 stages with projection shortcuts; 16 / 32 / 64 channels, no fourth stage)."""
    def layer():
        return {'x_quant': x_quant, 'w_quant': 'ls-1', 'clamp': {'kind': 'symmetric', 'alpha': 3}, 'double_shortcut': True}
    return {'moving_average_mode': mode, 'moving_average_momentum': 0.9, 'block': 'xnor',
            'layer0': {'n_in_channels': 16, 'kernel_size': 3, 'stride': 1, 'padding': 1, 'bias': False,
                       'maxpool': {'type': 'identity'}},
            'layer1': layer(), 'layer2': layer(), 'layer3': layer(), 'layer4': None,
            'nonlins': ['relu', 'relu'], 'num_blocks': [1, 1, 1], 'output_classes': 10}


def f8_checkpoints():
    """Reference models in the moving-average inference modes (activation_quantization.py:68-102): two train-mode
    batches (weight scales cached, EMA of the activation scales and batch-norm statistics updated), the
    reference's own log_checkpoints writes checkpoint_2.pt, then an eval-mode forward gives the expected logits.
    The .pt files are what torch.save made of tensors and plain numbers (state dicts); no source."""
    from pathlib import Path
    from quant.utils.checkpoints import log_checkpoints
    out = {}
    ckdir = Path(HERE) / 'ref_checkpoints'
    cases = [('resnet_eval_only_ls2', 'resnet', 'eval_only', 'ls-2'), ('resnet_train_and_eval_lsT', 'resnet', 'train_and_eval', 'ls-T'),
             ('lenet_eval_only_ls2', 'lenet', 'eval_only', 'ls-2'), ('lenet_train_and_eval_gf2', 'lenet', 'train_and_eval', 'gf-2')]
    for tag, kind, mode, xq in cases:
        if kind == 'resnet':
            arch = tiny_resnet_arch(mode, xq)
            model = RefResNet(loss_fn=torch.nn.functional.cross_entropy, **arch)
            shape = (6, 3, 32, 32)
        else:
            arch = arch_from_yaml('mnist/mnist_ls1_weight_ls2_activation.yaml')
            arch.update(moving_average_mode=mode, moving_average_momentum=0.9, x_quant=xq, conv1_filters=8, conv2_filters=12)
            model = RefLeNet(loss_fn=torch.nn.functional.nll_loss, **arch)
            shape = (6, 1, 28, 28)
        detgen.fill_module(model, seed=5)
        model.train()
        with torch.no_grad():
            for b in range(2):
                model(detgen.normal(f'{tag}.train{b}', shape))
        opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9)
        sched = torch.optim.lr_scheduler.StepLR(opt, 1)
        log_checkpoints(ckdir / tag / 'checkpoints', model, opt, sched, 2)
        model.eval()
        with torch.no_grad():
            out[tag + '_logits'] = model(detgen.normal(f'{tag}.eval', shape))
        out[tag + '_arch'] = np.frombuffer(json.dumps(arch).encode(), dtype=np.uint8)
        sd = model.state_dict()
        ma = [k for k in sd if k.endswith('moving_avg_module.moving_average')]
        out[tag + '_ma_first'] = sd[ma[0]]
        print(tag, 'checkpoint', os.path.getsize(ckdir / tag / 'checkpoints' / 'checkpoint_2.pt') // 1024, 'KiB')
    save('f8_checkpoints', **out)


# ----------------------------------------------------------------------------- F9 train-mode step
TRAIN_PAIRS = [('ls-2', 'ls-1'), ('ls-1', 'ls-1'), ('gf-2', 'ls-1'), ('ls-T', 'ls-1'), ('fp', 'ls-1'), ('ls-1', 'gf-2'),
               ('ls-1', 'ls-2'), ('fp', 'fp')]


def f9_train():
    """One train-mode forward + backward of the reference's QuantConv2d (ste.py:51-66, weight_quantization.py:29-31,
    :97-105): output, gradients of input / weight / bias for a fixed upstream gradient, the weight-scale buffers the
    forward cached, and the eval-mode output that then reuses them."""
    out = {}
    x0 = detgen.normal('f9.x', (3, 32, 10, 10), scale=1.1)
    gy = detgen.normal('f9.gy', (3, 24, 10, 10))
    clamp = {'kind': 'symmetric', 'alpha': 2}
    for xs, ws in TRAIN_PAIRS:
        conv = RefQuantConv2d(xs, ws, 32, 24, 3, clamp, padding=1, bias=True)
        with torch.no_grad():
            conv.weight.copy_(detgen.normal('f9.w', conv.weight.shape, scale=0.3))
            conv.bias.copy_(detgen.normal('f9.b', conv.bias.shape, scale=0.1))
        conv.train()
        x = x0.clone().requires_grad_()
        y = conv(x)
        y.backward(gy)
        key = f'{xs}_{ws}'
        out[key + '_y'], out[key + '_gx'] = y, x.grad
        out[key + '_gw'], out[key + '_gb'] = conv.weight.grad, conv.bias.grad
        for name, buf in conv.w_approximate.named_buffers():
            out[key + '_w_' + name] = buf
        conv.eval()
        with torch.no_grad():
            out[key + '_y_eval'] = conv(x0)
    save('f9_train', **out)


if __name__ == '__main__':
    only = sys.argv[1:]
    if only:
        for name in only:
            globals()[name]()
        sys.exit(0)
    f8_checkpoints()
    f1_sign()
    f24_quantizers()
    f3_solver()
    f5_conv()
    f6_models()
    f7_lenet()
    f9_train()
