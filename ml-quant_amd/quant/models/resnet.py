"""ResNet with quantized 3x3 convolutions (regular and XNOR-style basic blocks).

Constructor arguments, sub-module names (hence ``state_dict`` keys) and error behaviour follow
the reference's ``quant/models/resnet.py`` (``RegularBasicBlock`` :27-101, ``XnorBasicBlock``
:104-190, ``QResNet`` :193-397) so its yaml ``arch_config`` sections and checkpoints drop in.
The stem, the 1x1 shortcuts, batch norms, non-linearities and the classifier are full
precision and stay on stock PyTorch-ROCm ops; every 3x3 convolution of the residual blocks is
a ``QuantConv2d`` and, in eval mode on the GPU, runs on the gfx950 kernels.
"""

import warnings
from typing import Callable, Dict, List, Optional

import torch
import torch.nn as nn

from quant import _hip
from quant.binary import layouts
from quant.binary.binary_conv import QuantConv2d

non_linearity_map = {'relu': nn.ReLU, 'prelu': nn.PReLU, 'identity': nn.Identity}


def _check_nonlins(nonlins: List[str]) -> None:
    if len(nonlins) != 2:
        raise ValueError('There should be 2 non-linearities.')


def _eval_fold_ok(mod: nn.Module, bn: nn.BatchNorm2d, x: torch.Tensor) -> bool:
    return (FUSE_BLOCKS and x.is_cuda and not mod.training and not bn.training and bn.affine
            and bn.track_running_stats and not (torch.is_grad_enabled() and x.requires_grad))


def _folded_conv_bn(owner: nn.Module, conv: nn.Conv2d, bn: nn.BatchNorm2d):
    """(weight, bias) of the single convolution equal to ``bn(conv(x))`` in eval mode; cached on ``owner``
    and rebuilt whenever a parameter or running statistic changes."""
    tensors = [conv.weight, bn.running_mean, bn.running_var, bn.weight, bn.bias] + \
        ([conv.bias] if conv.bias is not None else [])
    stamp = tuple((t._version, t.data_ptr()) for t in tensors)
    hit = owner.__dict__.get('_folded')
    if hit is None or hit[0] != stamp:
        with torch.no_grad():
            scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
            w = conv.weight * scale.view(-1, 1, 1, 1)
            b = bn.bias - bn.running_mean * scale
            if conv.bias is not None:
                b = b + conv.bias * scale
        hit = (stamp, w.contiguous(), b.contiguous())
        owner.__dict__['_folded'] = hit
    return hit[1], hit[2]


def _square_pool(pool: nn.Module):
    """(kernel, stride, pad) when ``pool`` is a plain square floor-mode ``nn.MaxPool2d``, else None."""
    if not isinstance(pool, nn.MaxPool2d) or pool.ceil_mode or pool.return_indices:
        return None
    pair = lambda v: (v, v) if isinstance(v, int) else tuple(v)     # noqa: E731
    k, s, p, d = pair(pool.kernel_size), pair(pool.stride), pair(pool.padding), pair(pool.dilation)
    if k[0] != k[1] or s[0] != s[1] or p[0] != p[1] or d != (1, 1) or 2 * p[0] > k[0]:
        return None
    return k[0], s[0], p[0]


#: set to False to run the stem as MIOpen convolution + the pool / bias / ReLU tail kernel
FUSED_STEM = True
#: bf16 terms per fp32 operand in the fused stem: 3 = fp32-class accuracy (six MFMA passes, 0.70 ms at batch 256),
#: 2 = ~2^-17 per product (three passes, 0.49 ms) -- measurably more sign flips in the first quantizers
STEM_SPLIT = 22     # 22: fp16 hi + scaled fp16 lo, three MFMA passes (|x|, |w| < 65504); 3: three bf16 terms, six passes


def _is_resnet_stem(conv: nn.Conv2d, relu: nn.Module, pool: nn.Module, x: torch.Tensor) -> bool:
    """The ImageNet stem geometry lsq_stem_conv_pool implements: 7x7 / 2 / pad 3 from 3 to 64 channels, ReLU,
    3x3 / 2 / pad 1 max-pool, fp32 NCHW input of even width."""
    if not (isinstance(relu, nn.ReLU) and isinstance(pool, nn.MaxPool2d)):
        return False
    def two(v):
        return tuple(v) if isinstance(v, (tuple, list)) else (v, v)
    return (x.dim() == 4 and x.dtype == torch.float32 and x.shape[1] == 3 and x.shape[3] % 2 == 0 and x.shape[2] >= 8
            and x.shape[3] >= 8 and x.data_ptr() % 8 == 0 and x.is_contiguous()
            and conv.out_channels == 64 and conv.kernel_size == (7, 7) and conv.stride == (2, 2) and conv.padding == (3, 3)
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.weight.dtype == torch.float32
            and two(pool.kernel_size) == (3, 3) and two(pool.stride) == (2, 2) and two(pool.padding) == (1, 1)
            and two(pool.dilation) == (1, 1) and not pool.ceil_mode)


class _Stem(nn.Sequential):
    """``Sequential(conv1, bn1, relu, maxpool)`` of the reference (same keys).  Eval mode on the GPU: batch
    norm folded into the convolution's weights, MIOpen's convolution in channels-last, and the tail
    (max-pool, bias, ReLU -- bias and ReLU commute with the max -- and the layout change to the NCHW tensor
    the quantizer reads) as one kernel, ``lsq_pool_bias_relu_nhwc``."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        conv, bn, relu, pool = self[0], self[1], self[2], self[3]
        if _eval_fold_ok(self, bn, x):
            w, b = _folded_conv_bn(self, conv, bn)
            if FUSED_STEM and _is_resnet_stem(conv, relu, pool, x):
                # conv + bias + ReLU + max-pool in one kernel (csrc/lsq_stem.hip): the 112x112x64 convolution
                # output never goes to HBM, no channels-last copy of the input
                split = STEM_SPLIT
                if split == 22 and not torch.cuda.is_current_stream_capturing() and _hip.stem_overflow_tripped(x.device):
                    # an earlier batch had operands outside the fp16 split's domain (|value| >= 65504): its output held
                    # inf / nan; from here on the bf16 split, which takes any finite input
                    if not self.__dict__.get('_warned_split'):
                        self.__dict__['_warned_split'] = True
                        warnings.warn('lsq_stem_conv_pool: operands at or beyond 65504 seen; the stem now uses the three-term '
                                      'bf16 split (STEM_SPLIT = 3). Outputs of the batches before this warning may hold inf / nan.')
                    split = 3
                return _hip.stem_conv_pool(x, w, b, split)
            # a per-channel bias commutes with max-pooling too: add it on the pooled (4x smaller) tensor.
            # MIOpen's 7x7 stride-2 convolution and the pooling are ~1.3x / ~1.9x faster in channels-last
            # (measured, scripts/stem_bench.py); the bias add writes the NCHW tensor the quantizer reads.
            hit = self.__dict__.get('_folded_cl')
            if hit is None or hit[0] is not w:
                hit = self.__dict__['_folded_cl'] = (w, w.contiguous(memory_format=torch.channels_last))
            y = nn.functional.conv2d(x.contiguous(memory_format=torch.channels_last), hit[1], None,
                                     conv.stride, conv.padding, conv.dilation, conv.groups)
            geom = _square_pool(pool)
            if (geom is not None and isinstance(relu, nn.ReLU) and y.dtype == torch.float32
                    and y.is_contiguous(memory_format=torch.channels_last)):
                # pool + bias + ReLU + NHWC -> NCHW in one HBM pass (csrc/lsq_pool.hip)
                return _hip.pool_bias_relu_nhwc(y, *geom, b, True)
            y = pool(y)
            out = torch.empty(y.shape, dtype=y.dtype, device=y.device)
            torch.add(y, b.view(1, -1, 1, 1), out=out)
            return relu(out)
        return pool(relu(bn(conv(x))))


class _ConvBN(nn.Sequential):
    """``Sequential(conv, bn)`` (same ``state_dict`` keys as the reference) that runs as ONE convolution
    with the batch norm folded into its weights and bias in eval mode on the GPU."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        conv, bn = self[0], self[1]
        if _eval_fold_ok(self, bn, x):
            w, b = _folded_conv_bn(self, conv, bn)
            if conv.kernel_size == (1, 1) and conv.padding == (0, 0) and conv.groups == 1 and x.dim() == 4:
                if (x.dtype == torch.float32 and w.dtype == torch.float32 and conv.stride[0] == conv.stride[1]
                        and conv.in_channels % 64 == 0 and conv.out_channels % 64 == 0
                        and max(x.numel(), x.numel() // conv.in_channels * conv.out_channels) < 2 ** 30):
                    # the strided gather happens while staging the GEMM operand (csrc/lsq_pointwise.hip)
                    return _hip.pointwise_conv(x, w.view(conv.out_channels, conv.in_channels), b, conv.stride[0])
                return self._pointwise(x, w, b, conv.stride)
            return nn.functional.conv2d(x, w, b, conv.stride, conv.padding, conv.dilation, conv.groups)
        return bn(conv(x))

    def _pointwise(self, x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, stride) -> torch.Tensor:
        """The strided 1x1 projection as one gather and one batched fp32 GEMM that writes NCHW directly:
        ``y[n] = [W | b] @ [x[n, :, ::s, ::s] ; 1]`` (the bias rides along as an extra input row of ones).
        MIOpen's path for this shape transposes to NHWC and back and adds the bias in a third kernel
        (measured 444 us vs 250 us per forward for the three projections, scripts/stem_bench.py)."""
        xs = x[:, :, ::stride[0], ::stride[1]]
        n, c, ho, wo = xs.shape
        hit = self.__dict__.get('_pw')
        if hit is None or hit[0] is not w or hit[2].shape != (n, c + 1, ho * wo) or hit[2].device != x.device:
            buf = torch.ones((n, c + 1, ho * wo), dtype=x.dtype, device=x.device)
            hit = self.__dict__['_pw'] = (w, torch.cat([w.view(-1, c), b.view(-1, 1)], 1).contiguous(), buf)
        buf = hit[2]
        buf[:, :c].view(n, c, ho, wo).copy_(xs)
        return torch.matmul(hit[1], buf).view(n, -1, ho, wo)


def _projection(in_planes: int, planes: int, stride: int, bias: bool) -> nn.Sequential:
    """Full-precision down-sampling shortcut (empty when shapes already agree)."""
    if stride == 1 and in_planes == planes:
        return nn.Sequential()
    return _ConvBN(nn.Conv2d(in_planes, planes, kernel_size=1, stride=stride, bias=bias),
                   nn.BatchNorm2d(planes))


def _qconv(x_quant, w_quant, cin, cout, clamp, mode, momentum, stride, bias):
    return QuantConv2d(x_quant, w_quant, cin, cout, 3, clamp, mode, momentum,
                       stride=stride, padding=1, bias=bias)


class RegularBasicBlock(nn.Module):
    """conv-bn-nonlin, conv-bn, add shortcut, nonlin."""

    def __init__(self, in_planes: int, planes: int, x_quant: str, w_quant: str, nonlins: List[str],
                 stride: int = 1, clamp: Optional[Dict] = None, moving_average_mode: str = 'off',
                 moving_average_momentum: float = 0.99) -> None:
        super().__init__()
        _check_nonlins(nonlins)
        mode, mom = moving_average_mode, moving_average_momentum
        self.conv1 = _qconv(x_quant, w_quant, in_planes, planes, clamp, mode, mom, stride, False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.nonlin1 = non_linearity_map[nonlins[0]]()
        self.conv2 = _qconv(x_quant, w_quant, planes, planes, clamp, mode, mom, 1, False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.nonlin2 = non_linearity_map[nonlins[1]]()
        self.shortcut = _projection(in_planes, planes, stride, bias=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = layouts.to_nchw(x)          # (a three-stream tensor from a fused block in front: this block reads NCHW)
        y = self.nonlin1(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y)) + self.shortcut(x)
        return self.nonlin2(y)          # (batch norm follows the conv here: nothing to fold into the quantizer)


class XnorBasicBlock(nn.Module):
    """bn-quantconv-nonlin twice (XNOR-Net ordering), optionally with Bi-Real double shortcuts."""

    def __init__(self, in_planes: int, planes: int, x_quant: str, w_quant: str, nonlins: List[str],
                 stride: int = 1, double_shortcut: bool = False, clamp: Optional[Dict] = None,
                 moving_average_mode: str = 'off', moving_average_momentum: float = 0.99) -> None:
        super().__init__()
        _check_nonlins(nonlins)
        mode, mom = moving_average_mode, moving_average_momentum
        self.double_shortcut = double_shortcut
        self.bn1 = nn.BatchNorm2d(in_planes)
        self.conv1 = _qconv(x_quant, w_quant, in_planes, planes, clamp, mode, mom, stride, True)
        self.nonlin1 = non_linearity_map[nonlins[0]]()
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv2 = _qconv(x_quant, w_quant, planes, planes, clamp, mode, mom, 1, True)
        self.nonlin2 = non_linearity_map[nonlins[1]]()
        self.shortcut = _projection(in_planes, planes, stride, bias=True)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if _fusable(self, x):
            # eval on the GPU: bn -> quantizer and conv -> relu -> (+shortcut) each collapse into one
            # kernel pair (QuantConv2d.fused_forward); same arithmetic as the modular path below
            sc, sc_ready = _shortcut_on_side_stream(self.shortcut, x)
            a1, a2 = _act_args(self.nonlin1), _act_args(self.nonlin2)
            # next_q: who quantizes the result next -- with 1-bit activations that quantizer runs in the producing
            # convolution's epilogue (quant.binary.chain); `chain_next` = the following block's (bn1, conv1), set by QResNet
            nxt = self.__dict__.get('chain_next')
            if self.double_shortcut:
                first = self.conv1.fused_forward(x, self.bn1, res_post=sc, next_q=(self.bn2, self.conv2), res_ready=sc_ready, **a1)
                return self.conv2.fused_forward(first, self.bn2, res_post=first, next_q=nxt, **a2)
            first = self.conv1.fused_forward(x, self.bn1, next_q=(self.bn2, self.conv2), **a1)
            return self.conv2.fused_forward(first, self.bn2, res_pre=sc, next_q=nxt, res_ready=sc_ready, **a2)
        x = layouts.to_nchw(x)          # (a three-stream tensor from a fused block in front: the modules below read NCHW)
        first = self.nonlin1(self.conv1(self.bn1(x)))
        if self.double_shortcut:
            first = first + self.shortcut(x)
            return self.nonlin2(self.conv2(self.bn2(first))) + first
        second = self.conv2(self.bn2(first)) + self.shortcut(x)
        return self.nonlin2(second)


#: the projection shortcut (1x1 stride-2 convolution + batch norm, resnet.py:180-190 of the reference) does not depend on the
#: block's main branch: it runs on a side stream under the block's first quantizer and joins in front of the convolution whose
#: epilogue adds it (False: in line on the caller's stream, as the reference's forward orders it)
SIDE_STREAM_SHORTCUT = False
_SIDE_STREAMS: Dict = {}


def _shortcut_on_side_stream(shortcut: nn.Module, x: torch.Tensor):
    """(shortcut(x), event or None): a projection is queued on the side stream that belongs to the caller's current stream,
    ordered after everything queued so far (``x`` is ready); the returned event marks its end and the convolution that adds
    the result waits for it (``QuantConv2d.fused_forward(res_ready=...)``).  Identity shortcuts and graph capture: in line."""
    if (not SIDE_STREAM_SHORTCUT or len(shortcut) == 0 or not x.is_cuda or torch.cuda.is_current_stream_capturing()):
        return shortcut(x), None
    cur = torch.cuda.current_stream(x.device)
    key = (x.device.index, cur.cuda_stream)
    side = _SIDE_STREAMS.get(key)
    if side is None:
        if len(_SIDE_STREAMS) >= 64:
            _SIDE_STREAMS.clear()
        side = _SIDE_STREAMS[key] = torch.cuda.Stream(device=x.device)
    side.wait_event(cur.record_event())
    with torch.cuda.stream(side):
        sc = shortcut(x)
        ready = side.record_event()
    x.record_stream(side)             # (the allocator must not hand x's block out while the side stream reads it)
    sc.record_stream(cur)             # (allocated on the side stream, consumed on the caller's)
    return sc, ready


def _fusable(block: nn.Module, x: torch.Tensor) -> bool:
    """Fused HIP path: eval mode, CUDA input, ReLU non-linearities, weights binarized (QuantConv2d decides)."""
    return (not block.training and x.is_cuda and FUSE_BLOCKS and _act_args(block.nonlin1) is not None
            and _act_args(block.nonlin2) is not None and block.conv1._wants_hip(x))


def _act_args(nonlin: nn.Module):
    """fused_forward's keyword arguments for one of non_linearity_map's modules (None: not fusable)."""
    if isinstance(nonlin, nn.ReLU):
        return {'relu': True}
    if isinstance(nonlin, nn.PReLU) and nonlin.weight.dtype == torch.float32:
        return {'prelu': nonlin.weight}
    if isinstance(nonlin, nn.Identity):
        return {}
    return None


#: set to False to run residual blocks as separate BN / QuantConv2d / ReLU / add modules on the GPU too
FUSE_BLOCKS = True

_BLOCKS = {'regular': RegularBasicBlock, 'xnor': XnorBasicBlock}


class QResNet(nn.Module):
    """Stem (layer0) + three or four stages of basic blocks + average pool and linear classifier.

    ``layer0`` keys: ``n_in_channels`` (width of stage 1), ``kernel_size``, ``stride``,
    ``padding``, ``bias`` and ``maxpool`` (``{'type': 'identity'}`` or
    ``{'type': 'maxpool2d', 'kernel_size', 'stride', 'padding'}``).  ``layer1`` .. ``layer4``
    are keyword dictionaries for the block class (``x_quant``, ``w_quant``, ``clamp`` and, for
    xnor blocks, ``double_shortcut``); ``layer4`` may be ``None``.
    """

    def __init__(self, loss_fn: Callable[..., torch.Tensor], block: str, layer0: dict, layer1: dict,
                 layer2: dict, layer3: dict, layer4: Optional[dict], nonlins: List[str],
                 num_blocks: List[int], output_classes: int, moving_average_mode: str = 'off',
                 moving_average_momentum: float = 0.99) -> None:
        super().__init__()
        setattr(self, 'loss_fn', loss_fn)
        if block not in _BLOCKS:
            raise ValueError(f'Block {block} is not supported.')
        width = layer0['n_in_channels']
        self.conv1 = nn.Conv2d(3, width, kernel_size=layer0['kernel_size'], stride=layer0['stride'],
                               padding=layer0['padding'], bias=layer0['bias'])
        pool = layer0['maxpool']
        if pool['type'] == 'identity':
            self.maxpool = nn.Identity()
        elif pool['type'] == 'maxpool2d':
            self.maxpool = nn.MaxPool2d(kernel_size=pool['kernel_size'], stride=pool['stride'],
                                        padding=pool['padding'])
        else:
            raise ValueError(f"maxpool type {pool['type']} is not supported.")
        self.bn1 = nn.BatchNorm2d(width)
        self.blocks = nn.ModuleList([_Stem(self.conv1, self.bn1, nn.ReLU(inplace=True), self.maxpool)])

        planes = width
        stages = [(layer1, width, num_blocks[0], 1), (layer2, 2 * width, num_blocks[1], 2),
                  (layer3, 4 * width, num_blocks[2], 2)]
        if layer4 is not None:
            stages.append((layer4, 8 * width, num_blocks[3], 2))
        for config, out_planes, count, stride in stages:
            planes = self._make_layer(_BLOCKS[block], config, planes, out_planes, count, nonlins, stride,
                                      moving_average_mode, moving_average_momentum)
        self.linear_classifier = nn.Sequential(nn.AdaptiveAvgPool2d((1, 1)), nn.Flatten(),
                                               nn.Linear(planes, output_classes))
        # every xnor block knows the block behind it (a plain attribute, not a sub-module): its last convolution can then
        # prepare that block's first quantized input (quant.binary.chain)
        for cur, nxt in zip(self.blocks[1:], self.blocks[2:]):
            if isinstance(cur, XnorBasicBlock) and isinstance(nxt, XnorBasicBlock):
                cur.__dict__['chain_next'] = (nxt.bn1, nxt.conv1, nxt.shortcut)

    def _make_layer(self, block, layer_config: dict, in_planes: int, out_planes: int, num_blocks: int,
                    nonlins: List[str], stride: int, moving_average_mode: str = 'off',
                    moving_average_momentum: float = 0.99) -> int:
        """Append ``num_blocks`` blocks (the first one strided); returns the stage's width."""
        for i in range(num_blocks):
            self.blocks.append(block(in_planes, out_planes, nonlins=nonlins, stride=stride if i == 0 else 1,
                                     moving_average_mode=moving_average_mode,
                                     moving_average_momentum=moving_average_momentum, **layer_config))
            in_planes = out_planes
        return in_planes

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not self.training and x.is_cuda:
            from quant.binary import chain
            with chain.scope(x.device):            # (the row-sum accumulators of chained 1-bit layers: one fill per forward)
                for stage in self.blocks:
                    x = stage(x)
            return self.linear_classifier(layouts.to_nchw(x))
        for stage in self.blocks:
            x = stage(x)
        return self.linear_classifier(x)
