"""a train-mode QuantConv2d step (forward, straight-through backward, weight-scale caching,
eval reuse) against what the REFERENCE computed for the same step (tests/golden/f9_train.npz, written by
make_fixtures.py:f9_train from /root/reference: ste.py:51-66, weight_quantization.py:29-31, :77-79, :97-108)."""

import pytest
import torch

import detgen

TRAIN_PAIRS = [('ls-2', 'ls-1'), ('ls-1', 'ls-1'), ('gf-2', 'ls-1'), ('ls-T', 'ls-1'), ('fp', 'ls-1'), ('ls-1', 'gf-2'),
               ('ls-1', 'ls-2'), ('fp', 'fp')]
CLAMP = {'kind': 'symmetric', 'alpha': 2}


def make_train_conv(xs, ws, device='cpu', clamp=CLAMP):
    from quant.binary.binary_conv import QuantConv2d
    conv = QuantConv2d(xs, ws, 32, 24, 3, clamp, padding=1, bias=True)
    with torch.no_grad():
        conv.weight.copy_(detgen.normal('f9.w', conv.weight.shape, scale=0.3))
        conv.bias.copy_(detgen.normal('f9.b', conv.bias.shape, scale=0.1))
    return conv.to(device).train()


def train_step(conv, device='cpu'):
    x = detgen.normal('f9.x', (3, 32, 10, 10), scale=1.1).to(device).requires_grad_()
    gy = detgen.normal('f9.gy', (3, 24, 10, 10)).to(device)
    y = conv(x)
    y.backward(gy)
    return x, y


def rel(a, b):
    a, b = a.detach().cpu(), b.detach().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize('xs,ws', TRAIN_PAIRS)
def test_train_step_equals_the_reference(golden, xs, ws):
    g = golden('f9_train')
    key = f'{xs}_{ws}'
    conv = make_train_conv(xs, ws)
    x, y = train_step(conv)
    assert torch.equal(y.detach(), g[key + '_y'])                           # same forward, bit for bit
    for name, buf in conv.w_approximate.named_buffers():                    # scales cached by the train-mode forward
        assert torch.equal(buf, g[key + '_w_' + name]), name
    # gradients: same graph (straight-through estimator, |x| <= 1 mask), reductions may associate differently
    assert rel(x.grad, g[key + '_gx']) <= 1e-6
    assert rel(conv.weight.grad, g[key + '_gw']) <= 1e-6
    assert rel(conv.bias.grad, g[key + '_gb']) <= 1e-6
    conv.eval()
    with torch.no_grad():
        assert torch.equal(conv(x.detach()), g[key + '_y_eval'])            # eval reuses the cached scales


def test_fp_fp_is_plain_conv2d():
    """tests/binary/test_binary_conv.py:18-38 of the reference: with both schemes 'fp' (and no clamp) the module IS
    nn.Conv2d."""
    conv = make_train_conv('fp', 'fp', clamp=None)
    plain = torch.nn.Conv2d(32, 24, 3, padding=1, bias=True)
    plain.load_state_dict({'weight': conv.weight.detach(), 'bias': conv.bias.detach()})
    x, y = train_step(conv)
    x2, y2 = train_step(plain)
    assert torch.equal(y, y2) and torch.equal(x.grad, x2.grad) and torch.equal(conv.weight.grad, plain.weight.grad)
