"""Train-mode ``QuantConv2d`` on the gfx950 kernels.

The reference trains through stock autograd: ``QuantConv2d.forward`` (quant/binary/binary_conv.py:161-173) composes the
clamp, the activation quantizer (scales from detached data, signs through ``STESign``, quant/binary/ste.py:21-66), the
weight quantizer in its compute-and-cache mode (quant/binary/weight_quantization.py:29-31, :53-56, :77-79, :103-105)
and ``F.conv2d``.  Here the same step is one ``torch.autograd.Function`` around the C ABI:

forward   the weight scales are computed and cached (``copy_`` into the ``v1..vk`` buffers, as the reference does);
          ``lsq_act_quant`` solves the per-sample activation scales and packs the sign planes; ``lsq_pack_weight`` +
          ``lsq_xnor_conv2d`` (binary activations) or ``lsq_signw_conv2d`` (fp activations) produce the output -- the
          kernels of the inference path.  Saved for backward: the input, both sets of scales.
backward  grad_bias = sum of grad_y;
          grad_xq   = conv_transpose2d(grad_y, w_q): one ``lsq_signw_conv2d`` per weight plane over the flipped,
                      transposed sign weights (packed once per step), the plane's per-output-channel scale folded into
                      its input-channel pre-scale, later planes accumulated through the residual epilogue; stride 2 by
                      zero insertion;
          grad_x    = ``lsq_ste_backward``: straight-through estimator of every sign of the quantizer chain + clamp mask;
          grad_wq   = correlation of x_q (``lsq_quant_values``: the quantizer's value from the saved input and scales) with
                      grad_y -- ``torch.nn.grad.conv2d_weight`` (MIOpen), the one piece without a kernel of its own;
          grad_w    = ``lsq_ste_backward`` over the weight rows.
Geometries the transposed convolution does not take (groups, dilation, strides other than 1 / 2) fall back to the torch
formulation of the module; results are those of the reference's graph within fp32 reassociation (tests: f9_train).
"""

from typing import List, Optional

import torch
import torch.nn.functional as F
from torch.autograd.function import once_differentiable


def supported(conv, x: torch.Tensor) -> bool:
    """Train-mode forward + backward on the kernels: fp32 4-d CUDA input, binary weights, plain geometry -- and the library
    built (without it a training run on a GPU host takes the torch formulation; the eval path, by contrast, raises)."""
    from quant import _hip
    if not _hip.available():
        return False
    if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and conv.weight.dtype == torch.float32):
        return False
    if conv.w_quant == 'fp' or conv.padding_mode != 'zeros' or isinstance(conv.padding, str):
        return False
    if conv.groups != 1 or tuple(conv.dilation) != (1, 1) or conv.stride[0] != conv.stride[1] or conv.stride[0] not in (1, 2):
        return False
    kh, kw = conv.kernel_size
    if conv.padding[0] > kh - 1 or conv.padding[1] > kw - 1:
        return False
    if x.shape[0] > 65535 or conv.out_channels > 65535:
        return False
    return conv._hip_supports(x)


def _weight_residual_planes(weight: torch.Tensor, plane_scales: torch.Tensor) -> List[torch.Tensor]:
    """r_q = w - sum_{p<q} u_p sign(r_p): the tensors whose signs are the weight planes (one per plane scale)."""
    out, resid = [], weight
    for q in range(plane_scales.shape[0]):
        out.append(resid)
        if q + 1 < plane_scales.shape[0]:
            sign = torch.where(resid < 0, -torch.ones_like(resid), torch.ones_like(resid))
            resid = resid - plane_scales[q].view(-1, 1, 1, 1) * sign
    return out


_CONSTANTS = {}


def _constants(c: int, o: int, device):
    """(ones [1, c], zeros [o]) on ``device``: read-only operands of the input-gradient convolutions, made once."""
    key = (c, o, device)
    t = _CONSTANTS.get(key)
    if t is None:
        t = _CONSTANTS[key] = (torch.ones((1, c), dtype=torch.float32, device=device),
                               torch.zeros((o,), dtype=torch.float32, device=device))
    return t


class _QuantConv2dStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, conv):
        from quant import _hip
        x = x.contiguous()
        n, c, h, w = x.shape
        kh, kw = conv.kernel_size
        alpha = conv._alpha()
        geom = _hip.make_geom(n, c, h, w, conv.out_channels, kh, kw, conv.stride, conv.padding, conv.dilation, 1)
        ho, wo = _hip.out_hw(geom)
        # weight scales: computed from the detached weights and cached in the module's buffers (train mode)
        with torch.no_grad():
            conv.w_approximate(weight.detach())
            wscales = conv.w_approximate.plane_scales().to(torch.float32).contiguous()          # [planes, O]
        wbits, wsum = _hip.pack_weight(weight.detach(), geom, wscales)
        y = torch.empty((n, conv.out_channels, ho, wo), dtype=torch.float32, device=x.device)
        b = None if bias is None else bias.detach()
        xq_mod = conv.x_approximate
        if conv.x_quant == 'fp':
            _hip.signw_conv2d(x.detach(), alpha, wbits, wscales, b, geom, y)
            xscales = None
        else:
            k = xq_mod.n_planes
            # the plane workspace is the module's, one per input shape and launch stream as in eval mode (zero halo written
            # once, the kernels rewrite the interior); the scales are saved for backward and stay a tensor of this step
            key = ('train_planes', geom.key()[:4], geom.pad_h, geom.pad_w, k, x.device, _hip.stream_ptr(x.device))
            planes = conv._hip_cache.get(key)
            if planes is None:
                planes = torch.zeros((k * _hip.act_plane_words(geom),), dtype=torch.int64, device=x.device)
                stale = [kk for kk in list(conv._hip_cache) if isinstance(kk, tuple) and kk[0] == 'train_planes']
                for kk in stale[:max(0, len(stale) - 3)]:
                    conv._hip_cache.pop(kk, None)
                conv._hip_cache[key] = planes
            xscales = torch.empty((k, n), dtype=torch.float32, device=x.device)
            forced = xq_mod._forced_scales
            forced = None if forced is None else xq_mod.plane_scales(forced).to(device=x.device, dtype=torch.float32).contiguous()
            _hip.act_quant(x.detach(), geom, xq_mod.hip_scheme, k, conv.act_skip, alpha, planes, xscales, forced)
            # moving averages (activation_quantization.py:72-88): tracked from the batch's mean scales; 'train_and_eval'
            # quantizes with the tracked values
            from quant.binary.activation_quantization import MovingAverageMode
            if forced is None and xq_mod.moving_average_mode != MovingAverageMode.off:
                with torch.no_grad():
                    tracked = xq_mod.moving_avg_module(xscales[:xq_mod.num_scaling_factors].mean(1))
                if xq_mod.moving_average_mode == MovingAverageMode.train_and_eval:
                    forced = xq_mod.plane_scales(tracked.view(-1, 1).expand(-1, n)).to(torch.float32).contiguous()
                    _hip.act_quant(x.detach(), geom, xq_mod.hip_scheme, k, conv.act_skip, alpha, planes, xscales, forced)
            _hip.xnor_conv2d(planes, k, xscales, wbits, wsum, wscales, b, geom, y)
            conv.last_act_scales = xscales
        ctx.conv, ctx.geom, ctx.alpha = conv, geom, alpha
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, weight, wscales, xscales if xscales is not None else x.new_empty(0))
        return y

    @staticmethod
    @once_differentiable                       # (the straight-through kernels have no second derivative: say so instead of returning a wrong one)
    def backward(ctx, gy):
        from quant import _hip
        conv, geom, alpha = ctx.conv, ctx.geom, ctx.alpha
        x, weight, wscales, xscales = ctx.saved_tensors
        xscales = xscales if xscales.numel() else None
        gy = gy.contiguous()
        n, c, h, w = x.shape
        o = conv.out_channels
        kh, kw = conv.kernel_size
        s = conv.stride[0]
        ph, pw = conv.padding
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        gx = gw = gb = None
        if need_b:
            gb = gy.sum(dim=(0, 2, 3))
        if need_x:
            # conv_transpose2d(gy, w_q) as stride-1 sign-weight convolutions of the (zero-inserted) gradient
            if s == 1:
                gin = gy
            else:
                hu, wu = h + 2 * ph - kh + 1, w + 2 * pw - kw + 1          # rows / columns a stride-1 pass would produce
                gin = gy.new_zeros((n, o, hu, wu))
                gin[:, :, ::s, ::s][:, :, :gy.shape[2], :gy.shape[3]] = gy
            tgeom = _hip.make_geom(n, o, gin.shape[2], gin.shape[3], c, kh, kw, (1, 1), (kh - 1 - ph, kw - 1 - pw), (1, 1), 1)
            assert _hip.out_hw(tgeom) == (h, w), (_hip.out_hw(tgeom), h, w)
            ones, zeros = _constants(c, o, x.device)
            gxq = None
            for r, u in zip(_weight_residual_planes(weight.detach(), wscales), wscales):
                wt = r.permute(1, 0, 2, 3).flip(2, 3).contiguous()              # [C, O, KH, KW], taps mirrored
                tbits, _ = _hip.pack_weight(wt, tgeom, ones)
                out = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
                _hip.signw_conv2d(gin, -1.0, tbits, ones, None, tgeom, out, pre=(u.contiguous(), zeros), res_post=gxq)
                gxq = out
            gx = _hip.ste_backward(x, gxq, xscales, alpha)
        if need_w:
            xq = _hip.quant_values(x, xscales, alpha)
            gwq = torch.nn.grad.conv2d_weight(xq, weight.shape, gy, conv.stride, conv.padding, conv.dilation, 1)
            gw = _hip.ste_backward(weight.detach(), gwq, wscales, -1.0)
        return gx, gw, gb, None


def train_step_forward(conv, x: torch.Tensor) -> torch.Tensor:
    """``conv(x)`` in train mode through the kernels, differentiable with respect to x, weight and bias."""
    return _QuantConv2dStep.apply(x, conv.weight, conv.bias, conv)
