"""Three-stream rows (``LSQ_LAYOUT_SPLIT3``, include/lsq_hip.h) on the host side.

Between a quantized convolution and the next layer's quantizer the fused eval path may keep an activation tensor in a
layout in which the sub-sample the v1 search reads (every third element of a sample's row, quantization.py:63) is one
contiguous third of the row: element ``(c, p)`` of the row (flat NCHW index ``e = c * H W + p``) lives in stream
``e % 3 = (c + p) % 3`` at ``(e % 3) * S + c * hp + p // 3`` -- per stream and channel a block of ``hp`` floats that starts on a
128-byte line, ``S = C * hp``.  Such a tensor
never leaves the fused path: it is produced by ``lsq_xnor_conv2d_layout``, read by ``lsq_act_quant_layout`` (the solve then
reads 4/3 of the row instead of twice all of it), by a later convolution's epilogue as a residual operand and by the
projection shortcut, and anything else that receives one calls :func:`to_nchw` first.

The torch tensor that travels between the modules has the NOMINAL shape ``[N, C, H, W]`` (so shape checks keep working) on
top of the ``[N, 3 S]`` buffer and carries a :class:`Split3` record as the attribute ``_lsq_split3``; its element order is
meaningless to torch operators -- the values are the NCHW tensor's, bit for bit, only their addresses differ.
"""

from typing import Optional

import torch

#: keep tensors in the three-stream layout between fused kernels.  OFF by default: measured on the headline network
#: (scripts/split3_ab.py, profiles/r06_split3_*.txt) the solve's first pass turns out to be bound by its per-key work, not by
#: the bytes it reads (29 -> 20 us at 56 x 56 for a third of the bytes, 5 us per launch inside the network), and the
#: convolution that writes the layout pays more than that (stride-3 tiles: +4 ... +20 us per launch) -- 2.37 -> 2.52 ms per
#: step.  Logits are bit-identical either way (tests/test_gpu_round6.py runs both).
ENABLED = False

_ATTR = '_lsq_split3'


class Split3:
    __slots__ = ('buf', 'S', 'shape')

    def __init__(self, buf: torch.Tensor, S: int, shape):
        self.buf, self.S, self.shape = buf, S, tuple(shape)


def stream_floats(c: int, h: int, w: int) -> int:
    """S of a row of c*h*w values, -1 when the shape has no such layout (h*w % 3 != 1)."""
    from quant import _hip
    return int(_hip.lib().lsq_split3_stream_floats(c, h, w))


def empty(n: int, c: int, h: int, w: int, device) -> torch.Tensor:
    """An uninitialised three-stream tensor of nominal shape [n, c, h, w]."""
    S = stream_floats(c, h, w)
    if S <= 0:
        raise ValueError(f'no three-stream layout for rows of {c} x {h} x {w}')
    buf = torch.empty((n, 3 * S), dtype=torch.float32, device=device)
    t = buf[:, :c * h * w].view(n, c, h, w)
    setattr(t, _ATTR, Split3(buf, S, (n, c, h, w)))
    return t


def info(t: Optional[torch.Tensor]) -> Optional[Split3]:
    """The record of a three-stream tensor, None for an ordinary one (read it BEFORE ``detach()``: it is an attribute of
    the tensor object)."""
    return None if t is None else getattr(t, _ATTR, None)


_index_cache = {}


def _index(shape, S: int, device) -> torch.Tensor:
    """Position of every element of a row (in flat NCHW order) inside the row's 3 S floats."""
    _, c, h, w = shape
    key = (c, h, w, S, str(device))
    idx = _index_cache.get(key)
    if idx is None:
        if len(_index_cache) >= 32:
            _index_cache.clear()
        hp = S // c
        ch = torch.arange(c, dtype=torch.int64, device=device).view(-1, 1)
        p = torch.arange(h * w, dtype=torch.int64, device=device).view(1, -1)
        idx = _index_cache[key] = (((ch + p) % 3) * S + ch * hp + p // 3).reshape(-1)
    return idx


def to_nchw(t: torch.Tensor) -> torch.Tensor:
    """An ordinary contiguous NCHW tensor with the values of ``t`` (``t`` itself when it already is one).  The slow way out
    of the fused path: a gather through an index tensor."""
    rec = info(t)
    return t if rec is None else unpack(rec)


def unpack(rec: Split3) -> torch.Tensor:
    n, c, h, w = rec.shape
    return rec.buf[:, _index(rec.shape, rec.S, rec.buf.device)].view(n, c, h, w)


def from_nchw(x: torch.Tensor) -> torch.Tensor:
    """The three-stream copy of an NCHW tensor (tests)."""
    n, c, h, w = x.shape
    t = empty(n, c, h, w, x.device)
    rec = info(t)
    rec.buf.zero_()
    rec.buf[:, _index(rec.shape, rec.S, x.device)] = x.reshape(n, -1).to(torch.float32)
    return t
