"""CPU restatement of the reference's model forwards (eval mode) -- TEST INFRASTRUCTURE ONLY.

Functional ResNet (regular / xnor blocks) and LeNet-5 forwards driven by a ``state_dict`` and
the yaml ``arch_config`` dictionary, following the reference's ``quant/models/resnet.py``
(:95-101, :180-190, :393-397) and ``quant/models/lenet.py`` (:78-94).  Every quantized
convolution goes through ``oracle.ref_port.quant_conv2d``; everything else is the same stock
torch CPU op the reference's ``nn`` modules call.  Used by tests, by ``smoke()`` and as the
``cpu_baseline`` leg of ``bench.py``; pinned against the reference's logits in
``tests/test_oracle_golden.py``.
"""

from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from oracle import ref_port as P


def _bn(sd, prefix, x, eps=1e-5, affine=True):
    w = sd.get(prefix + '.weight') if affine else None
    b = sd.get(prefix + '.bias') if affine else None
    return F.batch_norm(x, sd[prefix + '.running_mean'], sd[prefix + '.running_var'], w, b, False, 0.1, eps)


def _nonlin(sd, prefix, kind, x):
    if kind == 'relu':
        return F.relu(x)
    if kind == 'prelu':
        return F.prelu(x, sd[prefix + '.weight'])
    return x


def _wscales(sd, prefix) -> List[torch.Tensor]:
    out, i = [], 1
    while f'{prefix}.w_approximate.v{i}' in sd:
        out.append(sd[f'{prefix}.w_approximate.v{i}'])
        i += 1
    return out


def _qconv(sd, prefix, x, cfg, stride, padding, chunk, scales_out):
    d = {} if scales_out is not None else None
    y = P.quant_conv2d(x, sd[prefix + '.weight'], sd.get(prefix + '.bias'), cfg['x_quant'], cfg['w_quant'],
                       _wscales(sd, prefix), cfg.get('clamp'), stride, padding, chunk=chunk, details=d)
    if scales_out is not None:
        scales_out[prefix] = d['act_scales']
    return y


def _shortcut(sd, prefix, x, stride, present):
    if not present:
        return x
    y = F.conv2d(x, sd[prefix + '.0.weight'], sd.get(prefix + '.0.bias'), stride)
    return _bn(sd, prefix + '.1', y)


def xnor_block_forward(sd, p, x, cfg, nl, stride, proj, chunk=0, scales_out=None):
    """XnorBasicBlock.forward in eval mode (quant/models/resnet.py:180-190): BN -> QuantConv2d -> non-linearity with the
    single or the double shortcut; ``p`` = the block's state_dict prefix ('' for a block on its own)."""
    q = (p + '.') if p else ''
    first = _nonlin(sd, q + 'nonlin1', nl[0],
                    _qconv(sd, q + 'conv1', _bn(sd, q + 'bn1', x), cfg, stride, 1, chunk, scales_out))
    if cfg.get('double_shortcut', False):
        first = first + _shortcut(sd, q + 'shortcut', x, stride, proj)
        second = _qconv(sd, q + 'conv2', _bn(sd, q + 'bn2', first), cfg, 1, 1, chunk, scales_out)
        return _nonlin(sd, q + 'nonlin2', nl[1], second) + first
    second = _qconv(sd, q + 'conv2', _bn(sd, q + 'bn2', first), cfg, 1, 1, chunk, scales_out)
    return _nonlin(sd, q + 'nonlin2', nl[1], second + _shortcut(sd, q + 'shortcut', x, stride, proj))


def resnet_forward(sd: Dict[str, torch.Tensor], arch: dict, x: torch.Tensor, chunk: int = 0,
                   scales_out: Optional[dict] = None) -> torch.Tensor:
    l0 = arch['layer0']
    x = F.conv2d(x, sd['conv1.weight'], sd.get('conv1.bias'), l0['stride'], l0['padding'])
    x = F.relu(_bn(sd, 'bn1', x))
    mp = l0['maxpool']
    if mp['type'] == 'maxpool2d':
        x = F.max_pool2d(x, mp['kernel_size'], mp['stride'], mp['padding'])
    width = l0['n_in_channels']
    layers = [arch['layer1'], arch['layer2'], arch['layer3']] + ([arch['layer4']] if arch.get('layer4') else [])
    nl = arch['nonlins']
    idx, in_planes = 1, width
    for li, cfg in enumerate(layers):
        planes = width * (2 ** li)
        for bi in range(arch['num_blocks'][li]):
            stride = (1 if li == 0 else 2) if bi == 0 else 1
            p = f'blocks.{idx}'
            proj = stride != 1 or in_planes != planes
            if arch['block'] == 'xnor':
                x = xnor_block_forward(sd, p, x, cfg, nl, stride, proj, chunk, scales_out)
            else:
                y = _nonlin(sd, p + '.nonlin1', nl[0],
                            _bn(sd, p + '.bn1', _qconv(sd, p + '.conv1', x, cfg, stride, 1, chunk, scales_out)))
                y = _bn(sd, p + '.bn2', _qconv(sd, p + '.conv2', y, cfg, 1, 1, chunk, scales_out))
                x = _nonlin(sd, p + '.nonlin2', nl[1], y + _shortcut(sd, p + '.shortcut', x, stride, proj))
            in_planes = planes
            idx += 1
    x = F.adaptive_avg_pool2d(x, (1, 1)).flatten(1)
    return F.linear(x, sd['linear_classifier.2.weight'], sd['linear_classifier.2.bias'])


def lenet_forward(sd: Dict[str, torch.Tensor], arch: dict, x: torch.Tensor, chunk: int = 0) -> torch.Tensor:
    cfg = {'x_quant': arch.get('x_quant', 'fp'), 'w_quant': arch.get('w_quant', 'fp'), 'clamp': arch.get('clamp')}
    x = F.conv2d(x, sd['conv1.weight'], sd['conv1.bias'])
    x = F.max_pool2d(_bn(sd, 'bn_conv1', F.relu(x), eps=1e-4, affine=False), 2, 2)
    x = F.relu(_qconv(sd, 'conv2', _bn(sd, 'bn_conv2', x, eps=1e-4, affine=False), cfg, 1, 0, chunk, None))
    x = F.max_pool2d(x, 2, 2)
    x = F.relu(F.linear(x.reshape(x.shape[0], -1), sd['fc1.weight'], sd['fc1.bias']))
    return F.log_softmax(F.linear(x, sd['fc2.weight'], sd['fc2.bias']), dim=1)
