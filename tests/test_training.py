"""the training loop, the optimizer / scheduler factories, the distillation criterion and the
straight-through arithmetic the kernels implement (reference quant/common/training.py:66-152, initialization.py:134-216,
utils/kd_criterion.py, utils/linear_lr_scheduler.py, binary/ste.py:51-66)."""

import math

import pytest
import torch

import detgen


def test_train_runs_the_reference_loop_on_cpu():
    """One epoch: optimizer and scheduler step once per BATCH, metrics see every training output, hooks once per batch."""
    from quant.common.initialization import get_lr_scheduler, get_optimizer
    from quant.common.metrics import LossMetric, Top1Accuracy
    from quant.common.training import evaluate, train
    from quant.data.data_loaders import MNISTDataLoader
    from quant.models.lenet import QLeNet5
    torch.manual_seed(3)
    model = QLeNet5(loss_fn=torch.nn.functional.nll_loss, x_quant='ls-2', w_quant='ls-1', clamp={'kind': 'symmetric', 'alpha': 2})
    loader = MNISTDataLoader(train_batch_size=32, test_batch_size=32, dataset_path='', workers=0, n_train=96).get_train_loader()
    opt = get_optimizer(model.parameters(), {'algorithm': 'sgd', 'lr': 0.05, 'momentum': 0.9})
    sched = get_lr_scheduler(opt, {'scheduler': 'step_lr', 'step_size': 1, 'gamma': 0.5}, epochs=2, steps_per_epoch=len(loader))
    seen = []
    metrics = {'Loss': LossMetric(model.loss_fn, accumulate=True), 'Top-1 Accuracy': Top1Accuracy(accumulate=True)}
    before = model.conv2.weight.detach().clone()
    out = [train(model, loader, metrics, opt, sched, torch.device('cpu'), epoch, 1, hooks=[lambda **kw: seen.append(kw)])
           for epoch in (1, 2)]
    assert len(seen) == 6 and seen[0]['global_step'] == 1 and seen[0]['values_dict']['lr'] == 0.05
    assert seen[3]['global_step'] == 1 + 96 and seen[3]['values_dict']['lr'] == pytest.approx(0.025)   # step_size scaled to one EPOCH
    assert out[1]['Loss'] < out[0]['Loss'] and not torch.equal(before, model.conv2.weight)
    assert model.training and float(model.conv2.w_approximate.v1.abs().sum()) > 0      # scales cached by the train-mode forwards
    ev = evaluate(model, loader, metrics, torch.device('cpu'), 1)
    assert set(ev) == {'Loss', 'Top-1 Accuracy'} and not model.training


def test_optimizer_and_scheduler_factories():
    from quant.common.initialization import get_lr_scheduler, get_optimizer
    from quant.utils.linear_lr_scheduler import LinearLR
    p = [torch.nn.Parameter(torch.zeros(3))]
    assert isinstance(get_optimizer(p, {'algorithm': 'adam', 'lr': 1e-3}), torch.optim.Adam)
    assert isinstance(get_optimizer(p, {'algorithm': 'adadelta', 'lr': 1.0}), torch.optim.Adadelta)
    with pytest.raises(KeyError):
        get_optimizer(p, {'algorithm': 'lamb'})
    opt = get_optimizer(p, {'algorithm': 'sgd', 'lr': 0.1})
    s = get_lr_scheduler(opt, {'scheduler': 'multi_step_lr', 'milestones': [1, 3], 'gamma': 0.1}, epochs=4, steps_per_epoch=5)
    assert sorted(s.milestones) == [5, 15]
    opt = get_optimizer(p, {'algorithm': 'sgd', 'lr': 0.1})
    s = get_lr_scheduler(opt, {'scheduler': 'lambda_lr', 'lr_lambda': 'lambda step: 0.5 ** step'}, 1, 1)
    opt.step(); s.step()
    assert opt.param_groups[0]['lr'] == pytest.approx(0.05)
    # the yaml carries min_lr as a string ('2e-7'); linear decay over (epochs - 1) * steps_per_epoch batches
    opt = get_optimizer(p, {'algorithm': 'sgd', 'lr': 0.1})
    s = get_lr_scheduler(opt, {'scheduler': 'linear_lr', 'min_lr': '2e-7'}, epochs=3, steps_per_epoch=4)
    assert isinstance(s, LinearLR)
    lrs = []
    for _ in range(10):
        opt.step(); s.step()
        lrs.append(opt.param_groups[0]['lr'])
    assert lrs[0] == pytest.approx(0.1 - 1 / 8 * (0.1 + 2e-7)) and lrs[7] == pytest.approx(2e-7) and lrs[9] == pytest.approx(2e-7)


def test_kd_criterion():
    from quant.utils.kd_criterion import kd_criterion
    s = detgen.normal('kd.s', (6, 10)).requires_grad_()
    t = detgen.normal('kd.t', (6, 10))
    y = torch.arange(6) % 10
    T = 4.0
    want = (torch.nn.functional.kl_div(torch.log_softmax(s / T, 1), torch.softmax(t / T, 1), reduction='none').sum(1) * T * T).mean()
    assert torch.allclose(kd_criterion(s, t, y, T), want)
    assert torch.allclose(kd_criterion(s, t, y, T, teacher_correction=False), want)
    assert float(kd_criterion(t, t, y, T)) == pytest.approx(0.0, abs=1e-6)
    kd_criterion(s, t, y, T).backward()
    assert s.grad is not None and math.isfinite(float(s.grad.abs().sum()))


@pytest.mark.parametrize('scheme,k', [('ls-1', 1), ('ls-2', 2), ('ls-T', 2), ('gf-3', 3), ('fp', 0)])
def test_straight_through_chain_formula(scheme, k):
    """The closed form lsq_ste_backward implements -- t_i = G_{i+1} v_i [|d_i| <= 1], G_i = G_{i+1} - t_i, clamp mask --
    equals autograd through the torch formulation of every quantizer (which the f9_train fixture pins to the reference)."""
    import quant.binary.quantization as Q
    x = (detgen.normal(f'ste.{scheme}', (3, 4, 5, 5), scale=1.4)).requires_grad_()
    g = detgen.normal(f'ste.g.{scheme}', (3, 4, 5, 5))
    alpha = 2.0
    xc = Q.clamp_symmetric(x, alpha)
    if scheme == 'fp':
        xq, scales = xc, []
    elif scheme == 'ls-1':
        v1, xq = Q.quantizer_ls_1(xc); scales = [v1]
    elif scheme == 'ls-2':
        v1, v2, xq = Q.quantizer_ls_2(xc); scales = [v1, v2]
    elif scheme == 'ls-T':
        v1, xq = Q.quantizer_ls_ternary(xc); scales = [v1, v1]
    else:
        vs, xq = Q.quantizer_gf(xc, 3); scales = list(vs)
    xq.backward(g)
    with torch.no_grad():
        xd = x.detach()
        inside = (xd >= -alpha) & (xd <= alpha)
        c = xd.clamp(-alpha, alpha)
        r = torch.zeros_like(c)
        d = []
        for v in scales:
            di = c - r
            d.append(di)
            r = r + v.view(-1, 1, 1, 1) * torch.where(di >= 0, 1.0, -1.0)
        G, acc = g.clone(), torch.zeros_like(g)
        for v, di in zip(reversed(scales), reversed(d)):
            t = torch.where(di.abs() <= 1, G * v.view(-1, 1, 1, 1), torch.zeros_like(G))
            acc += t
            G = G - t
        want = torch.where(inside, acc if k else g, torch.zeros_like(g))
        if k:
            assert torch.equal(r, xq.detach())            # the chain's value IS the quantizer's output
    assert torch.allclose(x.grad, want, rtol=1e-6, atol=1e-7)
