"""Chains of 1-bit layers: the next layer's quantizer in the producing convolution's epilogue.

With ``x_quant = 'ls-1'`` the activation quantizer of the reference (``quantizer_ls_1``, quant/binary/quantization.py:35-56)
is a sign and a mean of absolute values -- both can be taken from the values a convolution's epilogue has in registers.
``lsq_xnor_conv2d_chain`` (include/lsq_hip.h) then writes the NEXT ``QuantConv2d``'s bit plane and adds its exact row sums
to an accumulator, and the next call takes its activation scale from that accumulator: the ``lsq_act_quant`` launch between
two quantized convolutions, and its read of the activation tensor, disappear.  This module is the host side of the
hand-over: what a producer attaches to its output tensor (``PreQuant``), and the accumulators of one forward, which are
zeroed with ONE fill (``scope``).
"""

import contextlib
import threading
from typing import Optional

import torch

#: False: every layer runs its own quantizer launch (tests compare the two)
ENABLED = True
#: a consumer whose input has at most this many elements (batch x channels x pixels) is chained: the epilogue's extra ~150
#: VALU instructions per tile sit in a kernel that is matrix-core / VALU-bound while the separate quantizer sweep is
#: HBM-bound, so chaining pays only where a layer is small enough for the LAUNCH to matter.  Measured under graph replay
#: (scripts/ls1_chain.py): CIFAR ResNet-18, batch 100 (6.5 M elements and fewer): 0.555 -> 0.542 ms per forward; ImageNet
#: ResNet-18 ls-1, batch 256: chaining the 7 x 7 layers (6.4 M) 1.942 -> 1.956 ms, the 14 x 14 ones too (12.8 M) 1.991 ms,
#: 28 x 28 too 2.081 ms, every layer 2.194 ms.
MAX_ELEMENTS = 1 << 23

_state = threading.local()
_arenas = {}


class PreQuant:
    """The next layer's 1-bit input as its producer left it: plane words + row sums in units of 2^e."""

    __slots__ = ('consumer', 'pre_bn', 'planes', 'units', 'shape', 'stream', 'version')

    def __init__(self, consumer, pre_bn, planes, units, shape, stream):
        self.consumer, self.pre_bn, self.planes, self.units, self.shape, self.stream = consumer, pre_bn, planes, units, shape, stream
        self.version = None             # ``_version`` of the tensor the record rides on: an in-place edit of it voids the record


@contextlib.contextmanager
def scope(device):
    """One forward of a network: every row-sum accumulator handed out inside comes from one arena that is zeroed by a
    single fill at the first request (outside a scope each accumulator is its own ``torch.zeros``)."""
    prev = getattr(_state, 'arena', None)
    _state.arena = {'device': torch.device(device), 'used': 0, 'buf': None}
    try:
        yield
    finally:
        _state.arena = prev


def accumulator(n: int, device) -> torch.Tensor:
    """``n`` zeroed int64 row sums on ``device``."""
    st = getattr(_state, 'arena', None)
    device = torch.device(device)
    if st is None or st['device'] != device:
        return torch.zeros((n,), dtype=torch.int64, device=device)
    if st['buf'] is None:
        key = (device.index, torch.cuda.current_stream(device).cuda_stream)
        buf = _arenas.get(key)
        need = 64 * n                                   # (more layers than any model here has)
        if buf is None or buf.numel() < need:
            buf = torch.empty((need,), dtype=torch.int64, device=device)
            if len(_arenas) >= 16:
                _arenas.pop(next(iter(_arenas)))
            _arenas[key] = buf
        buf.zero_()                                     # the one fill of this forward
        st['buf'] = buf
    if st['used'] + n > st['buf'].numel():
        return torch.zeros((n,), dtype=torch.int64, device=device)
    out = st['buf'][st['used']:st['used'] + n]
    st['used'] += n
    return out


def attach(y: torch.Tensor, record: PreQuant, keep) -> None:
    """Hand ``record`` to whoever consumes ``y`` next (an attribute of the tensor OBJECT: a view, a copy or a new tensor
    does not carry it, and the consumer then quantizes for itself)."""
    record.version = y._version
    y._lsq_pre = record
    y._lsq_keep = keep


def pending(x: torch.Tensor) -> Optional[PreQuant]:
    """The record a producer left on ``x`` -- unless ``x`` was written to since (a hook, ``x.add_(...)``): the planes and row
    sums describe the values the producer stored, so the consumer must quantize what is there now."""
    rec = getattr(x, '_lsq_pre', None) if ENABLED else None
    if rec is not None and rec.version != x._version:
        return None
    return rec
