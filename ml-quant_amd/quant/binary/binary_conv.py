"""``QuantConv2d``: 2-D convolution of scaled-binary-quantized activations and weights.

Drop-in for the reference's ``quant/binary/binary_conv.py`` (:48-173): same constructor
(positional ``x_quant, w_quant, in_channels, out_channels, kernel_size, clamp,
moving_average_mode, moving_average_momentum`` then ``nn.Conv2d`` kwargs), same attributes
(``x_approximate``, ``w_approximate``, ``clamping_fn``, ``quantized_parameters``), same
``state_dict`` keys and the same ``ValueError`` on bad schemes / clamp kinds.

Dispatch of ``forward``:
  * CUDA (ROCm) tensor, ``eval()`` mode, no gradient wanted for the input -> gfx950 kernels via
    the C ABI (``quant._hip``): clamp + per-sample scale solve + sign packing in one kernel,
    then an XNOR-popcount convolution (binary activations) or a bf16-MFMA convolution
    (fp activations) against weight sign planes packed once per ``eval()`` session.  There is
    no fallback on this branch: a missing library or a failed launch raises.
  * CUDA tensor in ``train()`` mode, plain geometry -> the same kernels for the forward and the kernels of
    ``quant.binary.hip_train`` for the backward (straight-through estimator, transposed sign-weight convolution), one
    ``torch.autograd.Function`` per call;
  * anything else (CPU tensors, grouped / dilated training convolutions) -> the torch formulation in ``quant.binary``.
"""

import re
from collections import defaultdict
from functools import partial
from typing import Any, Callable, Dict, List, Optional, Tuple, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

import quant.binary.activation_quantization as activation_quantization
import quant.binary.quantization as quantization
import quant.binary.weight_quantization as weight_quantization

_SCHEME_RE = re.compile(r'fp|ls-1|ls-2|ls-T|gf-\d+')


class QuantConv2d(nn.Conv2d):
    """``Conv2d(x_quant(clamp(x)), w_quant(w))`` with schemes fp | ls-1 | ls-2 | ls-T | gf-k."""

    #: sub-sampling stride of the activation v1 search (quantizer_ls_2 / ls_ternary default)
    act_skip = 3

    def __init__(self, x_quant: str, w_quant: str, in_channels: int, out_channels: int,
                 kernel_size: Union[int, Tuple[int, int]], clamp: Optional[Dict] = None,
                 moving_average_mode: str = 'off', moving_average_momentum: float = 0.99,
                 **kwargs: Any) -> None:
        super().__init__(in_channels, out_channels, kernel_size, **kwargs)
        self.x_quant, self.w_quant = x_quant, w_quant
        self.x_approximate = self._get_x_quantizer(x_quant, moving_average_mode, moving_average_momentum)
        self.w_approximate = self._get_w_quantizer(w_quant, out_channels)
        self.clamp_config = dict(clamp) if clamp is not None else {'kind': 'identity'}
        self.clamping_fn = self._get_clamper(**self.clamp_config)

        self.quantized_parameters: Dict[str, List[torch.Tensor]] = defaultdict(list)
        if self.bias is not None:
            self.quantized_parameters['fp'].append(self.bias)
        self.quantized_parameters[w_quant].append(self.weight)

        self._hip_cache: Dict[str, Any] = {}          # packed weights, workspaces (never in state_dict)

    # ------------------------------------------------------------------ factories
    @staticmethod
    def _validate_scheme(scheme: str) -> None:
        if not isinstance(scheme, str) or not _SCHEME_RE.fullmatch(scheme):
            raise ValueError(f'Scheme {scheme} is invalid. Please see docs for valid schemes.')

    @staticmethod
    def _get_x_quantizer(scheme: str, moving_average_mode: str = 'off',
                         moving_average_momentum: float = 0.99) -> nn.Module:
        QuantConv2d._validate_scheme(scheme)
        if scheme == 'fp':
            return quantization.QuantizerFP()
        if scheme.startswith('gf-'):
            return activation_quantization.ActivationQuantizerGF(
                int(scheme[3:]), moving_average_mode, moving_average_momentum)
        cls = {'ls-1': activation_quantization.ActivationQuantizerLS1,
               'ls-2': activation_quantization.ActivationQuantizerLS2,
               'ls-T': activation_quantization.ActivationQuantizerLST}[scheme]
        return cls(moving_average_mode, moving_average_momentum)

    @staticmethod
    def _get_w_quantizer(scheme: str, size: int) -> nn.Module:
        QuantConv2d._validate_scheme(scheme)
        if scheme == 'fp':
            return quantization.QuantizerFP()
        if scheme.startswith('gf-'):
            return weight_quantization.WeightQuantizerGF(size, int(scheme[3:]))
        cls = {'ls-1': weight_quantization.WeightQuantizerLS1,
               'ls-2': weight_quantization.WeightQuantizerLS2,
               'ls-T': weight_quantization.WeightQuantizerLST}[scheme]
        return cls(size)

    @staticmethod
    def _get_clamper(kind: str, alpha: float = 2) -> Callable[[torch.Tensor], torch.Tensor]:
        if kind == 'identity':
            return quantization.clamp_identity
        if kind == 'symmetric':
            return partial(quantization.clamp_symmetric, alpha=alpha)
        raise ValueError(f'{kind} is not a valid clamping function.')

    # ------------------------------------------------------------------ forward
    #: train-mode CUDA tensors through the kernels (False: the torch formulation, e.g. to compare the two in tests)
    hip_train = True

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self._wants_hip(x):
            return self._forward_hip(x)
        if self.training and self.hip_train and x.is_cuda:
            from quant.binary import hip_train
            if hip_train.supported(self, x):
                return hip_train.train_step_forward(self, x)
        return self._forward_torch(x)

    def _forward_torch(self, x: torch.Tensor) -> torch.Tensor:
        x_q = self.x_approximate(self.clamping_fn(x))
        w_q = self.w_approximate(self.weight)
        return F.conv2d(x_q, w_q, self.bias, self.stride, self.padding, self.dilation, self.groups)

    def _wants_hip(self, x: torch.Tensor) -> bool:
        if not x.is_cuda or self.training:
            return False
        if torch.is_grad_enabled() and x.requires_grad:
            return False
        if self.padding_mode != 'zeros' or isinstance(self.padding, str) or x.dim() != 4:
            return False
        if self.w_quant == 'fp':          # nothing binary on the weight side: plain conv
            return False
        return self._hip_supports(x)

    def _hip_supports(self, x: torch.Tensor) -> bool:
        """The limits of the kernels (include/lsq_hip.h); anything outside them takes the torch formulation,
        exactly as training and CPU tensors do: fp32 only, at most 8 bit planes, kernels up to 8x8 on the
        XNOR path, at most 2^22 sub-sampled keys per row for the LS-2 / LS-T solve."""
        # (memoised per input dtype and row shape: the rest is fixed at construction; _apply -- .to(), .half() -- clears the cache)
        key = ('sup', x.dtype, x.shape[1], x.shape[2], x.shape[3])
        hit = self._hip_cache.get(key)
        if hit is None:
            if sum(1 for kk in self._hip_cache if isinstance(kk, tuple) and kk[0] == 'sup') >= 16:      # (many image sizes)
                for kk in [kk for kk in self._hip_cache if isinstance(kk, tuple) and kk[0] == 'sup']:
                    del self._hip_cache[kk]
            hit = self._hip_cache[key] = self._hip_supports_uncached(x)
        return hit

    def _hip_supports_uncached(self, x: torch.Tensor) -> bool:
        from quant import _hip
        if x.dtype != torch.float32 or self.weight.dtype != torch.float32:
            return False
        if getattr(self.w_approximate, 'k', 1) > _hip.MAX_PLANES:      # gf-k weights: k planes
            return False
        if self.x_quant != 'fp':
            if getattr(self.x_approximate, 'n_planes', 1) > _hip.MAX_PLANES or max(self.kernel_size) > _hip.MAX_XNOR_KERNEL:
                return False
            if self.x_quant in ('ls-2', 'ls-T'):
                m = x.shape[1] * x.shape[2] * x.shape[3]
                if (m + self.act_skip - 1) // self.act_skip >= _hip.MAX_SOLVER_KEYS:
                    return False
        return True

    def _replicate_for_data_parallel(self):
        replica = super()._replicate_for_data_parallel()
        replica._hip_cache = {}           # packed weights / workspaces live on the replica's own device
        return replica

    def train(self, mode: bool = True):
        if mode:
            self._hip_cache.clear()       # weights (and cached scales) may change
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        self._hip_cache.clear()
        return super()._load_from_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self._hip_cache.clear()
        return super()._apply(fn, *args, **kwargs)

    # ------------------------------------------------------------------ HIP path
    def _alpha(self) -> float:
        if self.clamp_config.get('kind') == 'symmetric':
            return float(self.clamp_config.get('alpha', 2))
        return -1.0

    def _packed_weights(self, geom, _hip):
        wq = self.w_approximate
        bufs = wq.cached_scales()
        w = self._parameters['weight']
        stamp = (w._version, w.data_ptr(), geom.key()[4:]) + tuple((b._version, b.data_ptr()) for b in bufs)
        hit = self._hip_cache.get('w')
        if hit is None or hit[0] != stamp:
            scales = wq.plane_scales().to(torch.float32).contiguous()
            wbits, wsum = _hip.pack_weight(self.weight.detach(), geom, scales)
            # fp activations: the bf16 operand image of the 3x3 fast path, built once with the planes
            wprep = _hip.signw_prepare_weight(wbits, scales.shape[0], geom) if self.x_quant == 'fp' else None
            hit = (stamp, wbits, wsum, scales, wprep)
            self._hip_cache['w'] = hit
        return hit[1], hit[2], hit[3], hit[4]

    def fused_forward(self, x: torch.Tensor, pre_bn: Optional[nn.BatchNorm2d] = None, relu: bool = False,
                      res_pre: Optional[torch.Tensor] = None, res_post: Optional[torch.Tensor] = None,
                      prelu: Optional[torch.Tensor] = None, next_q: Optional[tuple] = None,
                      res_ready: Optional[torch.cuda.Event] = None) -> torch.Tensor:
        """``act(self(pre_bn(x)) + res_pre) + res_post`` -- one residual-block half (quant/models/resnet.py:
        95-100, 182-190); ``act`` = ReLU (``relu=True``), PReLU (``prelu`` = the nn.PReLU weight) or identity.
        On the HIP path the eval-mode batch norm is folded into the quantizer's read and the non-linearity /
        shortcut additions into the convolution's epilogue, so none of them is a separate pass over HBM;
        elsewhere it is the plain composition of the modules."""
        if self._wants_hip(x) and (pre_bn is None or (not pre_bn.training and pre_bn.track_running_stats)):
            # next_q = (batch norm or None, QuantConv2d) that will consume the result: with 1-bit activations on both
            # sides the consumer's quantizer runs in THIS convolution's epilogue (quant.binary.chain)
            # res_ready: the residual operands were produced on another stream (the projection shortcut, models/resnet.py);
            # the quantizer does not need them, the convolution's launch waits for the event
            return self._forward_hip(x, pre_bn, relu, res_pre, res_post, prelu, next_q, res_ready)
        if res_ready is not None:
            torch.cuda.current_stream(x.device).wait_event(res_ready)
        y = self(x if pre_bn is None else pre_bn(x))
        if res_pre is not None:
            y = y + res_pre
        if relu:
            y = torch.relu(y)
        if prelu is not None:
            y = F.prelu(y, prelu)
        return y if res_post is None else y + res_post

    def _folded_bn(self, bn: nn.BatchNorm2d):
        """(scale, shift) with bn(x) = x * scale + shift in eval mode; cached on the BN's buffer versions."""
        # (read straight from the module's dicts: nn.Module.__getattr__ costs more than the rest of this check)
        b, p = bn._buffers, bn._parameters
        mean, var = b['running_mean'], b['running_var']
        stamp = (id(bn), bn.eps, mean._version, mean.data_ptr(), var._version, var.data_ptr())
        if bn.affine:
            gw, gb = p['weight'], p['bias']
            stamp += (gw._version, gw.data_ptr(), gb._version, gb.data_ptr())
        hit = self._hip_cache.get('bn')
        if hit is None or hit[0] != stamp:
            with torch.no_grad():
                inv = torch.rsqrt(bn.running_var.float() + bn.eps)
                scale = inv * bn.weight.float() if bn.affine else inv
                shift = (bn.bias.float() if bn.affine else 0) - bn.running_mean.float() * scale
            hit = (stamp, scale.contiguous(), shift.contiguous())
            self._hip_cache['bn'] = hit
        return hit[1], hit[2]

    def _chain_target(self, next_q, n: int, ho: int, wo: int, device, _hip):
        """``lsq_next_ls1`` for the consumer ``next_q = (bn, conv)`` of this layer's output, or None when the pair cannot
        be chained (other schemes, moving-average scales, no clamp, channel counts)."""
        from quant.binary import chain
        if next_q is None or not chain.ENABLED or self.x_quant != 'ls-1' or self.out_channels % 64:
            return None
        if n * self.out_channels * ho * wo > chain.MAX_ELEMENTS:    # (large layers: the separate HBM-bound sweep is cheaper)
            return None
        bn, conv = next_q[0], next_q[1]
        if not isinstance(conv, QuantConv2d) or conv.x_quant != 'ls-1' or conv.training or conv.w_quant == 'fp':
            return None
        if conv.in_channels != self.out_channels or conv.groups != 1 or isinstance(conv.padding, str) or conv.padding_mode != 'zeros':
            return None
        if conv.weight.device != device or (bn is not None and bn.running_mean.device != device):
            return None                 # (a replica whose ``chain_next`` still names the original module, e.g. nn.DataParallel)
        if conv._alpha() <= 0 or conv.x_approximate.eval_scales(n) is not None:
            return None
        if bn is not None and (bn.training or not bn.track_running_stats):
            return None
        pre = None if bn is None else conv._folded_bn(bn)
        ph, pw = conv.padding
        key = ('pre', n, self.out_channels, ho, wo, ph, pw, device, _hip.stream_ptr(device))
        planes = conv._hip_cache.get(key)
        if planes is None:
            words = n * (self.out_channels // 64) * (ho + 2 * ph) * (wo + 2 * pw)
            planes = torch.zeros((words,), dtype=torch.int64, device=device)          # zero halo; the interior is rewritten
            stale = [kk for kk in list(conv._hip_cache) if isinstance(kk, tuple) and kk[0] == 'pre']
            for kk in stale[:max(0, len(stale) - 3)]:
                conv._hip_cache.pop(kk, None)
            conv._hip_cache[key] = planes
        units = chain.accumulator(n, device)
        nxt = _hip.NextLs1(planes.data_ptr(), units.data_ptr(), None if pre is None else pre[0].data_ptr(),
                           None if pre is None else pre[1].data_ptr(), conv._alpha(), ph, pw)
        keep = (planes, units, pre)
        return nxt, chain.PreQuant(conv, bn, planes, units, (n, self.out_channels, ho, wo), _hip.stream_ptr(device)), keep

    def _reads_split3(self, geom, n: int, _hip) -> bool:
        """Can this call's quantizer read a three-stream input (quant.binary.layouts)?  The solving ls-2 / ls-T kernels under a
        symmetric clamp, sub-sampling stride 3, no given scales."""
        if self.x_quant not in ('ls-2', 'ls-T') or self.act_skip != 3 or self._alpha() <= 0:
            return False
        if self.x_approximate.eval_scales(n) is not None:
            return False
        return bool(_hip.layout_support(geom, self.x_approximate.hip_scheme, 2) & 1)

    def _split3_output(self, next_q, geom, n: int, ho: int, wo: int, device, _hip) -> bool:
        """Should this call leave its output as a three-stream tensor?  Only when every consumer is known to read one:
        ``next_q = (bn, conv[, shortcut])`` -- the quantizer of ``conv`` (the solve then reads 4/3 of the row instead of twice
        all of it), ``conv``'s epilogue (the tensor as a residual operand) and, if given, the projection ``shortcut`` of
        ``conv``'s block."""
        from quant.binary import layouts
        if next_q is None or not layouts.ENABLED or self.x_quant not in ('ls-2', 'ls-T'):
            return False
        if not (_hip.layout_support(geom, self.x_approximate.hip_scheme, 2) & 2):
            return False
        bn, conv = next_q[0], next_q[1]
        if not isinstance(conv, QuantConv2d) or conv.training or conv.w_quant == 'fp' or conv.in_channels != self.out_channels:
            return False
        if conv.groups != 1 or isinstance(conv.padding, str) or conv.padding_mode != 'zeros' or conv.weight.device != device:
            return False
        if bn is not None and (bn.training or not bn.track_running_stats or bn.running_mean.device != device):
            return False
        kh, kw = conv.kernel_size
        cgeom = _hip.make_geom(n, self.out_channels, ho, wo, conv.out_channels, kh, kw, conv.stride, conv.padding, conv.dilation, 1)
        if not conv._reads_split3(cgeom, n, _hip) or not (_hip.layout_support(cgeom, conv.x_approximate.hip_scheme, 2) & 4):
            return False
        if len(next_q) > 2 and next_q[2] is not None and len(next_q[2]) > 0:
            return bool(getattr(next_q[2], 'reads_split3', lambda *_: False)(n, self.out_channels, ho, wo))
        return True

    def _act_planes(self, x, geom, k, n, pre, xq, _hip, x_layout: int = 0):
        """Quantize ``x`` with lsq_act_quant into this module's plane workspace; returns (planes, scales)."""
        # (one workspace per launch stream: two streams through one module must not share planes and scales)
        key = ('act', geom.key()[:4], geom.pad_h, geom.pad_w, self.groups, k, x.device,
               _hip.stream_ptr(x.device))
        ws = self._hip_cache.get(key)
        if ws is None:
            words = _hip.act_plane_words(geom)
            # halo words must be zero; the kernels only ever write the interior
            ws = (torch.zeros((k * words,), dtype=torch.int64, device=x.device),
                  torch.empty((k, n), dtype=torch.float32, device=x.device))
            # one plane workspace per input shape; serving with many batch sizes must not grow without bound
            stale = [kk for kk in list(self._hip_cache) if isinstance(kk, tuple) and kk[0] == 'act']
            for kk in stale[:max(0, len(stale) - 3)]:
                self._hip_cache.pop(kk, None)
            self._hip_cache[key] = ws
        planes, scales = ws
        forced = xq.eval_scales(n)
        if forced is not None:
            forced = forced.to(device=x.device, dtype=torch.float32).contiguous()
        _hip.act_quant(x, geom, xq.hip_scheme, k, self.act_skip, self._alpha(), planes, scales, forced, pre, x_layout)
        return planes, scales

    def _forward_hip(self, x: torch.Tensor, pre_bn: Optional[nn.BatchNorm2d] = None, relu: bool = False,
                     res_pre: Optional[torch.Tensor] = None, res_post: Optional[torch.Tensor] = None,
                     prelu: Optional[torch.Tensor] = None, next_q: Optional[tuple] = None,
                     res_ready: Optional[torch.cuda.Event] = None) -> torch.Tensor:
        from quant import _hip
        from quant.binary import chain, layouts
        handed = chain.pending(x)                  # (an attribute of the tensor object: read before detach())
        # three-stream tensors (quant.binary.layouts) among the operands: the records are attributes of the tensor objects
        x_s3, rp_s3, rq_s3 = layouts.info(x), layouts.info(res_pre), layouts.info(res_post)
        x = x.detach()
        pre = None if pre_bn is None else self._folded_bn(pre_bn)
        n, c, h, w = x.shape
        kh, kw = self.kernel_size
        geom = _hip.make_geom(n, c, h, w, self.out_channels, kh, kw, self.stride, self.padding,
                              self.dilation, self.groups)
        wbits, wsum, wscales, wprep = self._packed_weights(geom, _hip)
        ho, wo = _hip.out_hw(geom)
        x_layout = res_layout = y_layout = _hip.LAYOUT_NCHW
        if x_s3 is not None:
            if self._reads_split3(geom, n, _hip):
                x, x_layout = x_s3.buf, _hip.LAYOUT_SPLIT3
            else:
                x = layouts.unpack(x_s3)
        if rp_s3 is not None or rq_s3 is not None:
            both = all(r is not None for r, t in ((rp_s3, res_pre), (rq_s3, res_post)) if t is not None)
            if both and self.x_quant in ('ls-2', 'ls-T') and (_hip.layout_support(geom, self.x_approximate.hip_scheme, 2) & 4):
                res_pre = None if res_pre is None else rp_s3.buf
                res_post = None if res_post is None else rq_s3.buf
                res_layout = _hip.LAYOUT_SPLIT3
            else:
                res_pre, res_post = layouts.to_nchw(res_pre) if res_pre is not None else None, layouts.to_nchw(res_post) if res_post is not None else None
        if res_layout == _hip.LAYOUT_NCHW:
            res_pre = None if res_pre is None else res_pre.detach().contiguous()
            res_post = None if res_post is None else res_post.detach().contiguous()
        if self._split3_output(next_q, geom, n, ho, wo, x.device, _hip):
            y = layouts.empty(n, self.out_channels, ho, wo, x.device)
            y_layout = _hip.LAYOUT_SPLIT3
        else:
            y = torch.empty((n, self.out_channels, ho, wo), dtype=torch.float32, device=x.device)
        y_dev = layouts.info(y).buf if y_layout else y
        bias = None if self.bias is None else self.bias.detach()
        def join():                      # residual operands from a side stream: in front of the convolution, behind the quantizer
            if res_ready is not None:
                torch.cuda.current_stream(x.device).wait_event(res_ready)
        if self.x_quant == 'fp':
            join()
            _hip.signw_conv2d(x, self._alpha(), wbits, wscales, bias, geom, y, pre, relu, res_pre, res_post, prelu, wprep)
            return y
        xq = self.x_approximate
        k = xq.n_planes
        if self.x_quant == 'ls-1' and chain.ENABLED:
            # chained 1-bit layers: take the planes and row sums the producer's epilogue left for THIS call, and / or leave
            # the consumer's in this call's epilogue
            stream = _hip.stream_ptr(x.device)
            if not (handed is not None and handed.consumer is self and handed.pre_bn is pre_bn and handed.stream == stream
                    and handed.shape == tuple(x.shape) and xq.eval_scales(n) is None and self._alpha() > 0):
                handed = None
            target = self._chain_target(next_q, n, ho, wo, x.device, _hip)
            if handed is not None or target is not None:
                planes_in, scales_in, units_in = None, None, None
                if handed is not None:
                    planes_in, units_in = handed.planes, handed.units
                else:
                    planes_in, scales_in = self._act_planes(x, geom, k, n, pre, xq, _hip)
                join()
                if _hip.xnor_conv2d_chain(planes_in, scales_in, units_in, self._alpha(), wbits, wsum, wscales, bias, geom, y,
                                          relu, res_pre, res_post, prelu, None if target is None else target[0]):
                    self.last_act_scales = scales_in          # (None when the scale came from the producer's row sums)
                    if target is not None:
                        chain.attach(y, target[1], target[2])
                    return y
                if handed is None and scales_in is not None:   # outside the matrix-core kernel: the plain call, same planes
                    _hip.xnor_conv2d(planes_in, k, scales_in, wbits, wsum, wscales, bias, geom, y, relu, res_pre, res_post, prelu)
                    self.last_act_scales = scales_in
                    return y
        planes, scales = self._act_planes(x, geom, k, n, pre, xq, _hip, x_layout)
        join()
        _hip.xnor_conv2d(planes, k, scales, wbits, wsum, wscales, bias, geom, y_dev, relu, res_pre, res_post, prelu, y_layout, res_layout)
        self.last_act_scales = scales
        return y
