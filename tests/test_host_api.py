"""Host side of the drop-in boundary on CPU: the reference's module/quantizer API (SURVEY.md
section 8b), its known-answer and inequality tests (tests/binary/*, tests/models/test_resnet.py of the
reference, restated), state_dict key compatibility, and config 0 (LeNet, ls-1 weights) plumbing."""

import itertools

import numpy as np
import pytest
import torch
import torch.nn as nn

import detgen
import quant.binary.quantization as quantization
from oracle import lsq_exact as E
from oracle import ref_port as P
from quant.binary.activation_quantization import (ActivationQuantizerGF, ActivationQuantizerLS1,
                                                  ActivationQuantizerLS2, ActivationQuantizerLST)
from quant.binary.binary_conv import QuantConv2d
from quant.binary.optimal import compute_mask, cost_function, opt_v1
from quant.binary.ste import binarize, binary_sign
from quant.binary.weight_quantization import (WeightQuantizerGF, WeightQuantizerLS1, WeightQuantizerLS2,
                                              WeightQuantizerLST)
from quant.common.initialization import get_loss_fn, get_model, model_mapping
from quant.models.lenet import QLeNet5
from quant.models.resnet import QResNet, XnorBasicBlock
from quant.utils.moving_average import MovingAverage

torch.set_num_threads(8)


# ------------------------------------------------------------------ ste
def test_sign_known_answers(golden):
    x = torch.tensor([42, -42, 42, 42, 0, -1, 1, -4.2, 4.2])
    assert torch.equal(binarize(x), torch.tensor([1., -1, 1, 1, 1, -1, 1, -1, 1]))
    g = golden('f1_sign')
    assert torch.equal(binary_sign(g['x']), g['sign'])


def test_ste_gradient_mask():
    x = torch.tensor([42, -42, 0, -1, 1, -0.2, 0.2], requires_grad=True)
    binarize(x).sum().backward()
    assert torch.equal(x.grad, torch.tensor([0., 0, 1, 1, 1, 1, 1]))


# ------------------------------------------------------------------ functional quantizers
def test_clamps():
    x = torch.tensor([-1.0, 0.0, 1.0, 2.0])
    assert torch.equal(quantization.clamp_identity(x), x)
    assert torch.equal(quantization.clamp_symmetric(x, 1), torch.tensor([-1., 0, 1, 1]))
    assert torch.equal(quantization.clamp_symmetric(x, 0.5), torch.tensor([-0.5, 0, 0.5, 0.5]))
    assert torch.equal(quantization.clamp_symmetric(x, 3), x)
    assert torch.equal(quantization.QuantizerFP()(x), x)


def test_quantizers_match_reference_fixture(golden):
    """ls-1 / gf-k / injected-scale calls are bit-exact; ls-2 / ls-T use the exact solver, so v1 may
    sit one rounding cluster away from the fp32 reference (1e-3) and equals the exact oracle."""
    g = golden('f24_quantizers')
    x = detgen.normal('f24.x', (4, 64, 14, 14), scale=1.3).clamp(-3, 3)
    v1, xq = quantization.quantizer_ls_1(x)
    assert torch.equal(v1, g['ls1_v1']) and torch.equal(xq, g['ls1_xq'])
    vs, xq = quantization.quantizer_gf(x, 2)
    assert torch.equal(vs[0], g['gf2_v1']) and torch.equal(vs[1], g['gf2_v2']) and torch.equal(xq, g['gf2_xq'])
    assert torch.equal(quantization.quantizer_ls_2(x, g['inj1'], g['inj2'])[2], g['ls2_inj_xq'])
    assert torch.equal(quantization.quantizer_ls_ternary(x, g['inj1'])[1], g['lst_inj_xq'])
    assert torch.equal(quantization.quantizer_ls_1(x, g['inj1'])[1], g['ls1_inj_xq'])
    assert torch.equal(quantization.quantizer_ls_2(x, g['inj1'])[1], g['ls2_v2_from_inj1'])
    v1, v2, _ = quantization.quantizer_ls_2(x)
    assert np.array_equal(v1.numpy(), E.solve_rows(x.numpy(), False, 3))
    assert torch.allclose(v1, g['ls2_v1'], rtol=1e-3) and torch.allclose(v2, g['ls2_v2'], rtol=1e-3)
    v1, _ = quantization.quantizer_ls_ternary(x)
    assert np.array_equal(v1.numpy(), E.solve_rows(x.numpy(), True, 3))
    assert torch.allclose(v1, g['lst_v1'], rtol=1e-3)


@pytest.mark.parametrize('ternary', [False, True])
@pytest.mark.parametrize('skip', [1, 3])
def test_opt_v1_equals_exact_oracle(ternary, skip):
    rs = np.random.RandomState(11)
    rows = [rs.standard_normal((6, 2500)).astype(np.float32).clip(-3, 3),
            np.maximum(rs.standard_normal((4, 999)), 0).astype(np.float32),
            np.full((2, 300), -3.0, dtype=np.float32), rs.standard_normal((5, 7)).astype(np.float32),
            rs.standard_normal((3, 2)).astype(np.float32)]
    for r in rows:
        mine = opt_v1(torch.from_numpy(r), ternary, skip).view(-1).numpy()
        assert np.array_equal(mine, E.solve_rows(r, ternary, skip))


def test_mask_and_cost_helpers_agree_with_reference_port():
    a = detgen.normal('host.mask', (5, 400)).abs()
    for ternary in (False, True):
        mask, vals = compute_mask(a, ternary)
        table, srt = P.candidate_table(a, ternary)
        # fp64 prefix sums can only drop the fp32 reference's rounding neighbours, never move a crossing
        assert mask.shape == table.shape and bool((mask & ~table).sum() <= 2)
        cands = srt[:, 100:104].contiguous()
        assert torch.allclose(cost_function(a, cands, ternary), P.candidate_costs(a, cands, ternary), rtol=1e-5)


def test_optimality_inequalities():
    """tests/binary/test_quantization.py:36-165 of the reference, at a CPU-friendly size."""
    torch.manual_seed(1234)
    x = torch.randn(64, 3, 32, 32)
    flat = x.view(64, -1)

    def err(xq):
        return torch.norm((xq - x).view(64, -1), dim=1)
    e1 = err(quantization.quantizer_ls_1(x)[1])
    assert torch.all(e1 <= err(torch.randn(64, 1, 1, 1).abs() * binarize(x)))
    e2 = err(quantization.quantizer_ls_2(x, skip=1)[2])
    eT = err(quantization.quantizer_ls_ternary(x, skip=1)[1])
    idx = torch.randint(0, flat.shape[1], (64,))
    sub_v1 = flat[torch.arange(64), idx].abs()
    assert torch.all(e2 <= err(quantization.quantizer_ls_2(x, sub_v1)[2]))
    assert torch.all(eT <= err(quantization.quantizer_ls_ternary(x, sub_v1)[1]))
    g = [err(quantization.quantizer_gf(x, k)[1]) for k in (1, 2, 3, 4)]
    assert torch.all(g[3] <= g[2]) and torch.all(g[2] <= g[1]) and torch.all(g[1] <= g[0])
    assert torch.all(e2 <= eT * (1 + 1e-6)) and torch.all(eT <= e1 * (1 + 1e-6))
    assert torch.all(e2 <= g[1] * (1 + 1e-6)) and torch.all(g[1] <= e1 * (1 + 1e-6))


def test_ternary_all_equal_rows():
    x = torch.ones(32, 3, 16, 16) * 2
    assert torch.all(quantization.quantizer_ls_ternary(x)[1] == 2.0)
    x = torch.rand(32, 3, 16, 16)
    x[1] = 2.0
    x[9] = -3.0
    _, xq = quantization.quantizer_ls_ternary(x)
    assert torch.all(xq[1] == 2) and torch.all(xq[9] == -3)


# ------------------------------------------------------------------ quantizer modules
def test_weight_quantizers_cache_in_train_and_reuse_in_eval():
    for make, nbuf in ((lambda: WeightQuantizerLS1(8), 1), (lambda: WeightQuantizerLS2(8), 2),
                       (lambda: WeightQuantizerLST(8), 1), (lambda: WeightQuantizerGF(8, 3), 3)):
        q = make()
        assert all(float(b.abs().sum()) == 0 for b in q.cached_scales()) and len(q.cached_scales()) == nbuf
        w1, w2 = torch.randn(8, 4, 3, 3), torch.randn(8, 4, 3, 3) * 3
        q.eval()
        assert float(q(w1).abs().sum()) == 0          # never trained: all-zero scales
        q.train()
        q(w1)
        cached = [b.clone() for b in q.cached_scales()]
        assert all(float(b.abs().sum()) > 0 for b in cached)
        q.eval()
        q(w2)
        assert all(torch.equal(a, b) for a, b in zip(cached, q.cached_scales()))


@pytest.mark.parametrize('cls,nsc', [(ActivationQuantizerLS1, 1), (ActivationQuantizerLS2, 2),
                                     (ActivationQuantizerLST, 1), (lambda m, mo: ActivationQuantizerGF(2, m, mo), 2)])
def test_activation_quantizer_moving_average_modes(cls, nsc):
    """tests/binary/test_activation_quantization.py: constant inputs give exact scales and EMA values."""
    tern = cls is ActivationQuantizerLST
    x2, x4 = torch.ones(8, 3, 4, 4) * 2, torch.ones(8, 3, 4, 4) * 4
    q = cls('off', 0.9)
    assert torch.all(q(x2) == 2.0)
    q.eval()
    assert torch.all(q(x4) == 4.0)                    # eval with mode 'off' recomputes
    q = cls('eval_only', 0.9)
    q.train()
    assert torch.all(q(x2) == 2.0) and torch.all(q(x4) == 4.0)       # tracked, not applied in training
    first = 1.0 if tern else 2.0
    second = 2.0 if tern else 4.0
    assert torch.allclose(q.moving_avg_module.moving_average[0], torch.tensor(0.9 * first + 0.1 * second))
    q.eval()
    out = q(x4)
    assert out.shape == x4.shape and bool((out == out.flatten()[0]).all())
    if cls is ActivationQuantizerLS1:
        assert torch.allclose(out, torch.full_like(out, 2.2))          # 0.9 * 2 + 0.1 * 4
    q = cls('train_and_eval', 0.9)
    q.train()
    q(x2)
    out = q(x4)                                       # applied in training: EMA scale, not 4
    assert float(out.flatten()[0]) != 4.0 or nsc > 1


def test_moving_average_module():
    m = MovingAverage(torch.tensor([0.9, 0.5]))
    assert set(dict(m.named_buffers())) == {'num_batches_tracked', 'momentum', 'moving_average'}
    m.train()
    assert torch.equal(m(torch.tensor([2.0, 2.0])), torch.tensor([2.0, 2.0]))
    assert torch.allclose(m(torch.tensor([4.0, 4.0])), torch.tensor([2.2, 3.0]))
    m.eval()
    assert torch.allclose(m(torch.tensor([100.0, 100.0])), torch.tensor([2.2, 3.0]))
    assert int(m.num_batches_tracked) == 2


# ------------------------------------------------------------------ QuantConv2d
def test_fp_quant_conv2d_equals_nn_conv2d_forward_and_input_grad():
    torch.manual_seed(1234)
    x = torch.randn(4, 3, 40, 40, requires_grad=True)
    x2 = x.clone().detach().requires_grad_(True)
    ref = nn.Conv2d(3, 30, 5)
    mine = QuantConv2d('fp', 'fp', 3, 30, 5)
    mine.weight, mine.bias = nn.Parameter(ref.weight), nn.Parameter(ref.bias)
    y_ref, y = ref(x), mine(x2)
    y_ref.sum().backward()
    y.sum().backward()
    assert torch.equal(y_ref, y) and torch.equal(x.grad, x2.grad)


def test_ls1_magnitude_bound_and_fp_act_structure():
    torch.manual_seed(1234)
    conv = QuantConv2d('ls-1', 'ls-1', 3, 16, (2, 2))
    y = conv(torch.randn(4, 3, 8, 8))
    for j in range(16):
        assert torch.max(y[:, j].abs()) <= 2 * 2 * 3 + conv.bias[j]
    x = torch.zeros(1, 3, 8, 8)
    x[0, :, :4, 4:], x[0, :, 4:, :4], x[0, :, 4:, 4:] = -1, 2, -3
    y = QuantConv2d('fp', 'ls-1', 3, 1, (4, 4), stride=4, bias=False)(x).squeeze()
    assert y.shape == (2, 2) and y[0, 0] == 0
    assert torch.isclose(y[1, 0], -2 * y[0, 1]) and torch.isclose(y[1, 1], 3 * y[0, 1])


def test_parameter_groups_and_scheme_validation():
    clamp = {'alpha': 2, 'kind': 'symmetric'}
    conv = QuantConv2d('ls-2', 'ls-1', 3, 1, (4, 4), clamp=clamp, stride=4, bias=False)
    assert len(conv.quantized_parameters['fp']) == 0 and len(conv.quantized_parameters['ls-1']) == 1
    assert set(conv.quantized_parameters) - {'fp', 'ls-1'} == set() and len(list(conv.parameters())) == 1
    conv = QuantConv2d('ls-2', 'ls-2', 3, 1, (4, 4), clamp=clamp, stride=4)
    assert len(conv.quantized_parameters['fp']) == 1 and len(conv.quantized_parameters['ls-2']) == 1
    schemes = ['fp', 'ls-1', 'ls-2', 'ls-T', 'gf-2', 'gf-3']
    for xs, ws in itertools.product(schemes, schemes):
        QuantConv2d(xs, ws, 3, 1, (4, 4))
    for bad in (('ls', 'ls-1'), ('l2', 'ls-1'), ('ls-1', 'ls-3'), ('ls-1', 'l2'), ('gf-', 'fp')):
        with pytest.raises(ValueError):
            QuantConv2d(bad[0], bad[1], 3, 1, (4, 4))
    with pytest.raises(ValueError):
        QuantConv2d('ls-1', 'ls-2', 3, 1, (4, 4), clamp={'kind': 'sym'})


@pytest.mark.parametrize('xs,ws', [('ls-1', 'ls-1'), ('gf-2', 'ls-1'), ('fp', 'ls-1'), ('fp', 'fp'), ('ls-1', 'gf-2')])
def test_quant_conv2d_cpu_path_matches_reference_fixture(golden, xs, ws):
    g = golden('f5_conv')
    x = detgen.normal('f5.x', (2, 64, 14, 14), scale=1.2)
    for stride in (1, 2):
        key = f'{xs}_{ws}_s{stride}_a2'
        conv = QuantConv2d(xs, ws, 64, 64, 3, {'kind': 'symmetric', 'alpha': 2}, stride=stride, padding=1, bias=True)
        with torch.no_grad():
            conv.weight.copy_(detgen.normal('f5.w.64.64.3', (64, 64, 3, 3), scale=(64 * 9) ** -0.5))
            conv.bias.copy_(detgen.normal('f5.w.64.64.3.b', (64,), scale=0.1))
            conv.train()
            conv(x)                              # train-mode forward caches the weight scales
            conv.eval()
            assert torch.equal(conv(x), g[key + '_y']), key


# ------------------------------------------------------------------ models / factories
def test_state_dict_keys_match_the_reference(golden):
    g = golden('f7_lenet')
    arch = golden('f6_models').json('imagenet_ls1w_ls2a_arch')
    assert list(QResNet(loss_fn=None, **arch).state_dict().keys()) == g.text('resnet_ls2_state_keys').split('\n')
    arch = dict(golden('f7_lenet').json('mnist_ls1_arch'), x_quant='gf-2')
    assert list(QLeNet5(loss_fn=None, **arch).state_dict().keys()) == g.text('lenet_gf2_state_keys').split('\n')
    conv = QuantConv2d('ls-2', 'ls-1', 3, 4, 3)
    assert not any('hip' in k or 'cache' in k for k in conv.state_dict())     # derived state never persisted


def test_resnet_variants_forward_shapes(golden):
    arch = golden('f6_models').json('imagenet_ls1w_ls2a_arch')
    x = torch.randn(2, 3, 32, 32)
    for block, nl, ds in (('xnor', ['prelu', 'prelu'], True), ('xnor', ['relu', 'relu'], False),
                          ('regular', ['relu', 'relu'], None)):
        cfg = dict(arch, block=block, nonlins=nl)
        for k in ('layer1', 'layer2', 'layer3', 'layer4'):
            lay = dict(cfg[k], x_quant='ls-1', clamp={'kind': 'symmetric', 'alpha': 2})
            if ds is None:
                lay.pop('double_shortcut')
            else:
                lay['double_shortcut'] = ds
            cfg[k] = lay
        assert QResNet(loss_fn=None, **cfg)(x).shape == (2, 1000)
    with pytest.raises(ValueError):
        QResNet(loss_fn=None, **dict(arch, block='bottleneck'))
    with pytest.raises(ValueError):
        QResNet(loss_fn=None, **dict(arch, layer0=dict(arch['layer0'], maxpool={'type': 'avg'})))
    with pytest.raises(ValueError):
        XnorBasicBlock(4, 4, 'ls-1', 'ls-1', ['relu'])


def test_factories():
    assert set(model_mapping) == {'lenet5', 'resnet'}
    assert get_loss_fn('nll_loss') is torch.nn.functional.nll_loss
    with pytest.raises(ValueError):
        get_loss_fn('mse')
    with pytest.raises(ValueError):
        get_model('vgg', None, {}, torch.device('cpu'), 0)
    m = get_model('lenet5', get_loss_fn('nll_loss'), {'w_quant': 'ls-1'}, torch.device('cpu'), 0)
    assert isinstance(m, QLeNet5) and m.w_quant == 'ls-1'


@pytest.mark.parametrize('tag', ['mnist_ls1w_fpa', 'mnist_ls1'])
def test_config0_lenet_plumbing(golden, tag):
    """BASELINE config 0: LeNet from the mnist yaml with ls-1 weights, CPU, batch 64 -> [64, 10] log-probs
    equal to the reference's (no search involved for these schemes, so bit-exact)."""
    g = golden('f7_lenet')
    model = QLeNet5(loss_fn=torch.nn.functional.nll_loss, **g.json(tag + '_arch'))
    detgen.fill_module(model, seed=3)
    with torch.no_grad():
        model.conv2.w_approximate.v1.copy_(P.weight_scales(model.conv2.weight, 'ls-1')[0])
    model.eval()
    with torch.no_grad():
        y = model(detgen.normal(tag + '.x', (64, 1, 28, 28)))
    assert y.shape == (64, 10) and torch.equal(y, g[tag + '_logp'])


def test_checkpoint_files_in_the_reference_layout(golden, tmp_path):
    """A ``checkpoint_<epoch>.pt`` in the reference's layout (utils/checkpoints.py:40-51) restores into a
    fresh model, which then reproduces the reference's log-probs; latest / requested epoch lookup and the
    error cases follow ``get_path_to_checkpoint`` (:107-136)."""
    from quant.utils.checkpoints import get_path_to_checkpoint, log_checkpoints, restore_from_checkpoint
    g, tag = golden('f7_lenet'), 'mnist_ls1'
    src = QLeNet5(loss_fn=torch.nn.functional.nll_loss, **g.json(tag + '_arch'))
    detgen.fill_module(src, seed=3)
    with torch.no_grad():
        src.conv2.w_approximate.v1.copy_(P.weight_scales(src.conv2.weight, 'ls-1')[0])
    opt = torch.optim.SGD(src.parameters(), lr=0.1, momentum=0.9)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=2)
    with pytest.raises(ValueError):
        (tmp_path / 'exp' / 'checkpoints').mkdir(parents=True)
        get_path_to_checkpoint(tmp_path / 'exp')
    log_checkpoints(tmp_path / 'exp' / 'checkpoints', src, opt, sched, 3)
    log_checkpoints(tmp_path / 'exp' / 'checkpoints', nn.DataParallel(src), opt, sched, 11)
    assert get_path_to_checkpoint(tmp_path / 'exp').endswith('checkpoint_11.pt')
    assert get_path_to_checkpoint(tmp_path / 'exp', 3).endswith('checkpoint_3.pt')
    with pytest.raises(ValueError):
        get_path_to_checkpoint(tmp_path / 'exp', 4)
    raw = torch.load(get_path_to_checkpoint(tmp_path / 'exp'))
    assert set(raw) == {'epoch', 'model_state_dict', 'optimizer_state_dict', 'scheduler_state_dict'}
    assert list(raw['model_state_dict']) == list(src.state_dict())            # no 'module.' prefix
    dst = QLeNet5(loss_fn=torch.nn.functional.nll_loss, **g.json(tag + '_arch'))
    dst.eval()
    dst.conv2._hip_cache['stale'] = object()                                   # derived state must not survive
    _, o2, s2, epoch = restore_from_checkpoint(dst, torch.optim.SGD(dst.parameters(), lr=0.5),
                                               None, get_path_to_checkpoint(tmp_path / 'exp'),
                                               torch.device('cpu'))
    assert epoch == 11 and s2 is None and o2.param_groups[0]['lr'] == 0.1 and not dst.conv2._hip_cache
    with torch.no_grad():
        assert torch.equal(dst(detgen.normal(tag + '.x', (64, 1, 28, 28))), g[tag + '_logp'])
    with pytest.raises(RuntimeError):
        restore_from_checkpoint(QLeNet5(loss_fn=None, w_quant='ls-2'), None, None,
                                get_path_to_checkpoint(tmp_path / 'exp'), torch.device('cpu'))
    restore_from_checkpoint(QLeNet5(loss_fn=None, w_quant='ls-2'), None, None,
                            get_path_to_checkpoint(tmp_path / 'exp'), torch.device('cpu'), strict_keys=False)


def _restore_reference_checkpoint(golden, tag, device):
    """Our model of the fixture's arch, restored (strict keys) from the checkpoint file the REFERENCE's
    log_checkpoints wrote after two train-mode batches (tests/golden/make_fixtures.py f8_checkpoints)."""
    import os
    from quant.models.resnet import QResNet
    from quant.utils.checkpoints import get_path_to_checkpoint, restore_from_checkpoint
    g = golden('f8_checkpoints')
    arch = g.json(tag + '_arch')
    if tag.startswith('resnet'):
        model, shape = QResNet(loss_fn=torch.nn.functional.cross_entropy, **arch), (6, 3, 32, 32)
    else:
        model, shape = QLeNet5(loss_fn=torch.nn.functional.nll_loss, **arch), (6, 1, 28, 28)
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ref_checkpoints', tag)
    model, _, _, epoch = restore_from_checkpoint(model.to(device), None, None, get_path_to_checkpoint(root), device)
    assert epoch == 2
    return model.eval(), detgen.normal(f'{tag}.eval', shape), g


CHECKPOINT_TAGS = ['resnet_eval_only_ls2', 'resnet_train_and_eval_lsT', 'lenet_eval_only_ls2', 'lenet_train_and_eval_gf2']


@pytest.mark.parametrize('tag', CHECKPOINT_TAGS)
def test_reference_checkpoint_moving_average_inference(golden, tag):
    """a checkpoint written by the reference in eval_only / train_and_eval mode
    (activation_quantization.py:68-102, utils/checkpoints.py:17-51) loads with strict keys and the eval forward
    -- activation scales = the restored moving averages, weight scales = the restored v1 buffers -- gives the
    reference's logits."""
    model, x, g = _restore_reference_checkpoint(golden, tag, torch.device('cpu'))
    sd = model.state_dict()
    ma = [k for k in sd if k.endswith('moving_avg_module.moving_average')]
    assert torch.equal(sd[ma[0]], g[tag + '_ma_first'])
    with torch.no_grad():
        y = model(x)
    ref = g[tag + '_logits']
    assert y.shape == ref.shape and torch.allclose(y, ref, rtol=1e-5, atol=1e-5 * float(ref.abs().max()))


def test_fused_forward_cpu_composition_with_prelu_and_identity():
    """QuantConv2d.fused_forward off the GPU is the plain composition of the modules it stands for, for every
    non-linearity of non_linearity_map; blocks report which of them the fused path takes."""
    import quant.models.resnet as R
    from quant.binary.binary_conv import QuantConv2d
    clamp = {'kind': 'symmetric', 'alpha': 2}
    conv = QuantConv2d('ls-2', 'ls-1', 8, 12, 3, clamp, padding=1)
    detgen.fill_module(conv, seed=3)
    bn = torch.nn.BatchNorm2d(8).eval()
    conv.train()
    x = detgen.normal('host.fused.x', (2, 8, 6, 6))
    with torch.no_grad():
        conv(x)
    conv.eval()
    res = detgen.normal('host.fused.r', (2, 12, 6, 6))
    slope = torch.tensor([0.3])
    with torch.no_grad():
        base = conv(bn(x))
        assert torch.equal(conv.fused_forward(x, bn, res_pre=res, prelu=slope), torch.nn.functional.prelu(base + res, slope))
        assert torch.equal(conv.fused_forward(x, bn, relu=True, res_post=res), torch.relu(base) + res)
        assert torch.equal(conv.fused_forward(x, bn), base)
    assert R._act_args(torch.nn.ReLU()) == {'relu': True} and R._act_args(torch.nn.Identity()) == {}
    p = torch.nn.PReLU()
    assert R._act_args(p)['prelu'] is p.weight and R._act_args(torch.nn.Sigmoid()) is None


def test_opt_v1_strict_flag_on_the_host():
    """The reference's opt_v1 raises (argmin over an empty dimension, optimal.py:147-151) when NO row of the batch has a
    candidate; the product returns zeros by default and raises the same IndexError under STRICT_NO_CANDIDATE."""
    from quant.binary import optimal
    no_candidate = torch.tensor([[1.0, 2.0], [0.5, -0.25]])              # two elements per row: no inner position
    some = torch.tensor([[0.1, 0.2, 0.9, 1.0, 1.1], [1.0, 1.0, 1.0, 1.0, 1.0]])
    assert float(optimal.opt_v1(no_candidate, False).abs().sum()) == 0.0
    optimal.STRICT_NO_CANDIDATE = True
    try:
        with pytest.raises(IndexError):
            optimal.opt_v1(no_candidate, False)
        assert torch.equal(optimal.opt_v1(torch.ones(3, 7), False), torch.ones(3, 1))   # all-equal rows DO have candidates (m = v)
        assert optimal.opt_v1(some, False).shape == (2, 1)                   # one row with a candidate is enough
        assert optimal.opt_v1(torch.ones(3, 7), True).shape == (3, 1)        # ternary: the extra candidate mean / 2 exists
    finally:
        optimal.STRICT_NO_CANDIDATE = False
