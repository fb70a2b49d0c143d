"""ctypes binding of the gfx950 C-ABI library (``include/lsq_hip.h``).

PyTorch is used only as the owner of device memory and streams: every call passes raw
device pointers (``tensor.data_ptr()``) and the current HIP stream.  There is no CPU or
eager fallback here: if the library is missing, or a call fails, an exception is raised.
"""

import contextlib
import ctypes
import warnings
import os
import threading
from typing import Optional, Sequence

import torch

_LIB_PATH = os.environ.get('LSQ_HIP_LIB') or os.path.join(   # (LSQ_HIP_LIB: developer builds, e.g. -DLSQ_PHASE_CLOCKS)
    os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'lib', 'liblsq_hip.so')
_lock = threading.Lock()
_lib = None

ABI_VERSION = 11
LAYOUT_NCHW, LAYOUT_SPLIT3 = 0, 1
SCHEME_LS1, SCHEME_LS2, SCHEME_LST, SCHEME_GF = 1, 2, 3, 4
MAX_PLANES = 8
MAX_XNOR_KERNEL = 8          # lsq_xnor_conv2d: KH, KW <= 8
MAX_SOLVER_KEYS = 1 << 22    # lsq_act_quant LS-2 / LS-T: sub-sampled keys per row
XNOR_MFMA_MAX_OUTPUTS = 1 << 30      # lsq_xnor_conv2d: at this many outputs the popcount kernel serves the call (same bits)
_xnor_limit_warned = set()


class ConvGeom(ctypes.Structure):
    """Mirror of ``lsq_conv_geom``."""

    _fields_ = [(n, ctypes.c_int32) for n in (
        'N', 'C', 'H', 'W', 'O', 'KH', 'KW', 'stride_h', 'stride_w',
        'pad_h', 'pad_w', 'dil_h', 'dil_w', 'groups')]

    def key(self):
        k = self.__dict__.get('_key')             # (make_geom leaves the tuple it was built from: 14 ctypes field reads otherwise)
        return k if k is not None else tuple(getattr(self, n) for n, _ in self._fields_)


class NextLs1(ctypes.Structure):
    """Mirror of ``lsq_next_ls1``: where a chained convolution leaves the next layer's 1-bit input."""

    _fields_ = [('planes', ctypes.c_void_p), ('sum_units', ctypes.c_void_p), ('pre_scale', ctypes.c_void_p),
                ('pre_shift', ctypes.c_void_p), ('clamp_alpha', ctypes.c_float), ('pad_h', ctypes.c_int32),
                ('pad_w', ctypes.c_int32)]


class LsqHipError(RuntimeError):
    """A C-ABI call returned a non-zero code."""


def library_path() -> str:
    return _LIB_PATH


def _declare(lib):
    vp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
    gp = ctypes.POINTER(ConvGeom)
    lib.lsq_abi_version.restype = i32
    lib.lsq_error_string.restype = ctypes.c_char_p
    lib.lsq_error_string.argtypes = [i32]
    lib.lsq_act_plane_words.restype = i64
    lib.lsq_act_plane_words.argtypes = [gp]
    lib.lsq_weight_plane_words.restype = i64
    lib.lsq_weight_plane_words.argtypes = [gp]
    lib.lsq_act_quant.restype = i32
    lib.lsq_act_quant.argtypes = [vp, gp, i32, i32, i32, f32, vp, vp, vp, vp, vp, vp, ctypes.c_size_t, vp]
    lib.lsq_act_quant_layout.restype = i32
    lib.lsq_act_quant_layout.argtypes = [vp, i32, gp, i32, i32, i32, f32, vp, vp, vp, vp, vp, vp, ctypes.c_size_t, vp]
    lib.lsq_split3_stream_floats.restype = i64
    lib.lsq_split3_stream_floats.argtypes = [i64, i64, i64]
    lib.lsq_layout_support.restype = i32
    lib.lsq_layout_support.argtypes = [gp, i32, i32]
    lib.lsq_xnor_conv2d_layout.restype = i32
    lib.lsq_xnor_conv2d_layout.argtypes = [vp, i32, vp, vp, vp, i32, vp, vp, gp, i32, vp, vp, vp, i32, vp, i32, vp]
    lib.lsq_solver_workspace_bytes.restype = i64
    lib.lsq_solver_workspace_bytes.argtypes = [i64]
    lib.lsq_sweep_workspace_bytes.restype = i64
    lib.lsq_sweep_workspace_bytes.argtypes = [i64]
    lib.lsq_solve_rows.restype = i32
    lib.lsq_solve_rows.argtypes = [vp, i64, i64, i32, i32, f32, vp, vp, vp, ctypes.c_size_t, vp]
    lib.lsq_pack_weight.restype = i32
    lib.lsq_pack_weight.argtypes = [vp, gp, i32, vp, vp, vp, vp]
    lib.lsq_xnor_conv2d.restype = i32
    lib.lsq_xnor_conv2d.argtypes = [vp, i32, vp, vp, vp, i32, vp, vp, gp, i32, vp, vp, vp, vp, vp]
    lib.lsq_signw_conv2d.restype = i32
    lib.lsq_signw_conv2d.argtypes = [vp, f32, vp, vp, vp, vp, i32, vp, vp, gp, i32, vp, vp, vp, vp, vp]
    lib.lsq_signw_weight_bytes.restype = i64
    lib.lsq_signw_weight_bytes.argtypes = [gp, i32]
    lib.lsq_signw_prepare_weight.restype = i32
    lib.lsq_signw_prepare_weight.argtypes = [vp, i32, gp, vp, vp]
    lib.lsq_pool_bias_relu_nhwc.restype = i32
    lib.lsq_pool_bias_relu_nhwc.argtypes = [vp, i32, i32, i32, i32, i32, i32, i32, vp, i32, vp, vp]
    lib.lsq_pointwise_conv.restype = i32
    lib.lsq_pointwise_conv.argtypes = [vp, i32, i32, i32, i32, vp, vp, i32, i32, vp, vp]
    for hook in ('lsq_debug_xnor_impl', 'lsq_debug_force_streaming', 'lsq_debug_fused_mode'):     # include/lsq_hip_debug.h
        getattr(lib, hook).restype = i32
        getattr(lib, hook).argtypes = [i32]
    lib.lsq_xnor_conv2d_chain.restype = i32
    lib.lsq_xnor_conv2d_chain.argtypes = [vp, vp, vp, f32, vp, vp, i32, vp, vp, gp, i32, vp, vp, vp, ctypes.POINTER(NextLs1), vp, vp]
    lib.lsq_quant_values.restype = i32
    lib.lsq_quant_values.argtypes = [vp, i64, i64, i32, vp, f32, vp, vp]
    lib.lsq_ste_backward.restype = i32
    lib.lsq_ste_backward.argtypes = [vp, vp, i64, i64, i32, vp, f32, vp, vp]
    lib.lsq_debug_solver_trace.restype = i32
    lib.lsq_debug_solver_trace.argtypes = [vp]
    lib.lsq_stem_conv_pool.restype = i32
    lib.lsq_stem_conv_pool.argtypes = [vp, i32, i32, i32, vp, vp, i32, vp, vp, vp]


def lib():
    """Load (once) and return the C-ABI library; raises if it has not been built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(_LIB_PATH):
                    raise LsqHipError(
                        f'{_LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                        '(or `make -C ml-quant_amd/csrc`). The HIP path has no fallback.')
                handle = ctypes.CDLL(_LIB_PATH)
                _declare(handle)
                if handle.lsq_abi_version() != ABI_VERSION:
                    raise LsqHipError('liblsq_hip.so ABI version mismatch')
                _lib = handle
    return _lib


def available() -> bool:
    return os.path.exists(_LIB_PATH)


def source_fingerprint() -> str:
    """sha256 over the kernel sources (csrc/*.hip, *.h, Makefile; names and contents, sorted): counter profiles record
    it when they are captured and bench.py attaches their figures only while it still matches -- a profile of older
    kernels is reported as stale instead of silently priced against the current ones."""
    import hashlib
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'csrc')
    h = hashlib.sha256()
    for name in sorted(os.listdir(src)):
        if name.endswith(('.hip', '.h')) or name == 'Makefile':
            h.update(name.encode() + b'\0')
            with open(os.path.join(src, name), 'rb') as f:
                h.update(f.read())
    return h.hexdigest()


def check(code: int, what: str) -> None:
    if code != 0:
        msg = lib().lsq_error_string(code).decode()
        raise LsqHipError(f'{what} failed with code {code}: {msg}')


def stream_ptr(device=None) -> int:
    """HIP stream the kernels of ``device`` are launched on (torch's current stream of THAT device)."""
    if _raw_stream is not None:               # (no Stream object built per launch: this runs several times per layer)
        d = torch.device(device) if device is not None else None
        return _raw_stream(torch.cuda.current_device() if d is None or d.index is None else d.index)
    return torch.cuda.current_stream(device).cuda_stream


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


_SAME_DEVICE = contextlib.nullcontext()
_UNTIMED = contextlib.nullcontext()


def _on(t: torch.Tensor):
    """Device guard for a C-ABI call: the library never calls hipSetDevice, so the tensor's device is made
    current around the launch (a model on cuda:1 with cuda:0 current must not launch on cuda:0's stream)."""
    if t.device.index == torch.cuda.current_device():
        return _SAME_DEVICE                    # (the common case: entering torch.cuda.device costs microseconds per launch)
    return torch.cuda.device(t.device)


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def make_geom(n, c, h, w, o, kh, kw, stride, padding, dilation, groups) -> ConvGeom:
    key = (n, c, h, w, o, kh, kw, stride[0], stride[1], padding[0], padding[1], dilation[0], dilation[1], groups)
    g = ConvGeom(*key)
    g._key = key                                   # (valid as long as nobody writes the fields afterwards: nothing here does)
    return g


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise TypeError(f'the HIP path computes in fp32, got {t.dtype}')
    return t if t.is_contiguous() else t.contiguous()


# ---- optional per-kernel timing with HIP events on the launch stream (bench.py's roofline leg)
_timing = None
_timing_only = None
_timing_paused = False


def enable_timing(on: bool = True, only: Optional[Sequence[str]] = None) -> None:
    """Bracket every C-ABI call (or only the entry points named in ``only``) with HIP events on the launch
    stream.  An event pair costs a few microseconds of stream time per call, so benchmarks instrument only
    the kernel they report on inside their timed region."""
    global _timing, _timing_only
    _timing = {} if on else None
    _timing_only = None if only is None else frozenset(only)


def pause_timing(paused: bool = True) -> None:
    """Keep the records but stop (resume) bracketing calls: a benchmark samples some of its timed steps."""
    global _timing_paused
    _timing_paused = bool(paused)


def drain_timing(by_tag: bool = False):
    """{kernel: (launches, total_ms, total_algorithmic_bytes, total_ops, total_survey_bytes)}; call after
    torch.cuda.synchronize().  ops = binary MACs (xnor conv) or bf16 FLOPs (sign-weight conv), 0 for the quantizer;
    survey bytes = SURVEY 8(d)'s count (input once + output once, no residual operands).  ``by_tag``: keys are
    (kernel, tag) with the tag the call site attached (the layer's input channels and height)."""
    out = {}
    for name, recs in (_timing or {}).items():
        groups = {}
        for s, e, b in recs:
            groups.setdefault((name, b[2]) if by_tag else name, []).append((s.elapsed_time(e), b[0], b[1], b[3]))
        for key, rows in groups.items():
            out[key] = (len(rows), sum(r[0] for r in rows), sum(r[1] for r in rows), sum(r[2] for r in rows),
                        sum(r[3] for r in rows))
        recs.clear()
    return out


class _Timed:
    def __init__(self, name, nbytes, ops=0, tag=None, survey_bytes=None):
        # nbytes: every operand the call must move once (residuals of a fused epilogue included);
        # survey_bytes: SURVEY 8(d)'s definition for the path kernels (input read once + output written once)
        self.name, self.nbytes = name, (nbytes, ops, tag, nbytes if survey_bytes is None else survey_bytes)

    def __enter__(self):
        self.on = _timing is not None and not _timing_paused and (_timing_only is None or self.name in _timing_only)
        if self.on:                       # (events record on the current stream of the current device: call inside _on)
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()

    def __exit__(self, *exc):
        if self.on:
            self.e.record()
            _timing.setdefault(self.name, []).append((self.s, self.e, self.nbytes))


_ws_cache = {}
_WS_CACHE_MAX = 16      # (device, stream) entries kept per cache: streams come and go (one per captured graph)


def _remember(cache: dict, key, buf):
    cache.pop(key, None)
    cache[key] = buf                          # (insertion order = age: the oldest entries go first)
    while len(cache) > _WS_CACHE_MAX:
        cache.pop(next(iter(cache)))


_ws_bytes_memo = {}


def _ws_bytes(fn: str, rows: int) -> int:
    key = (fn, rows)
    need = _ws_bytes_memo.get(key)
    if need is None:
        if len(_ws_bytes_memo) > 256:
            _ws_bytes_memo.clear()
        need = _ws_bytes_memo[key] = getattr(lib(), fn)(rows)
    return need


def solver_workspace(rows: int, device) -> torch.Tensor:
    """Scratch for the LS2/LST solve (slot records passed from the sweep to the solve kernel); cached per
    device and grown on demand -- kernels of one stream run in order, so sharing it is safe."""
    need = _ws_bytes('lsq_solver_workspace_bytes', rows)
    device = torch.device(device)
    if device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    key = (device.index, stream_ptr(device))
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < need:
        buf = torch.empty((need,), dtype=torch.uint8, device=device)
        _remember(_ws_cache, key, buf)
    return buf


_sweep_ws_cache = {}

def sweep_workspace(rows: int, device) -> torch.Tensor:
    """Row workspace of the ls-1 / gf-k sweeps (partial sums + arrival counters of rows shared by several workgroups):
    any content (the arrival slots carry a per-launch epoch); cached per (device, stream) like the solver's."""
    need = _ws_bytes('lsq_sweep_workspace_bytes', rows)
    device = torch.device(device)
    if device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    key = (device.index, stream_ptr(device))
    buf = _sweep_ws_cache.get(key)
    if buf is None or buf.numel() < need:
        buf = torch.zeros((need,), dtype=torch.uint8, device=device)
        _remember(_sweep_ws_cache, key, buf)
    return buf


_layout_memo = {}


def layout_support(geom: ConvGeom, scheme: int, kx: int) -> int:
    """Bit mask of the operands of a call with this geometry that may be three-stream tensors (lsq_layout_support)."""
    key = (geom.key(), scheme, kx)
    hit = _layout_memo.get(key)
    if hit is None:
        if len(_layout_memo) > 256:
            _layout_memo.clear()
        hit = _layout_memo[key] = int(lib().lsq_layout_support(ctypes.byref(geom), scheme, kx))
    return hit


def act_quant(x: torch.Tensor, geom: ConvGeom, scheme: int, k: int, skip: int, alpha: float,
              planes: torch.Tensor, scales: torch.Tensor, forced: Optional[torch.Tensor] = None,
              pre: Optional[tuple] = None, x_layout: int = LAYOUT_NCHW) -> None:
    """pre = (scale[C], shift[C]) folds an eval-mode batch norm into the read.  ``x_layout`` = LAYOUT_SPLIT3: ``x`` is the
    [N, 3 S] buffer of a three-stream tensor (quant.binary.layouts)."""
    if x_layout == LAYOUT_NCHW:
        x = _f32c(x)
    elif x.dtype != torch.float32 or not x.is_contiguous():
        raise TypeError('a three-stream operand is passed as its contiguous fp32 [N, 3 S] buffer')
    ws = None
    if forced is None:
        ws = solver_workspace(geom.N, x.device) if scheme in (SCHEME_LS2, SCHEME_LST) else sweep_workspace(geom.N, x.device)
    m = geom.C * geom.H * geom.W
    # (accounting -- x read once + k bit planes written -- only when a benchmark asked for it: the record costs microseconds)
    with _on(x), (_Timed('lsq_act_quant', geom.N * (4 * m + k * m // 8), 0, f'C{geom.C}_H{geom.H}') if _timing is not None else _UNTIMED):
        check(lib().lsq_act_quant_layout(x.data_ptr(), x_layout, ctypes.byref(geom), scheme, k, skip, float(alpha),
                                         None if pre is None else pre[0].data_ptr(), None if pre is None else pre[1].data_ptr(),
                                         ptr(forced), planes.data_ptr(), scales.data_ptr(), ptr(ws),
                                         0 if ws is None else ws.numel(), stream_ptr(x.device)), 'lsq_act_quant')


def solve_rows(rows: torch.Tensor, skip: int, ternary: bool, alpha: float = -1.0):
    """Returns (v12 [2,R] fp32, status [R] int32)."""
    rows = _f32c(rows)
    r, m = rows.shape
    v12 = torch.empty((2, r), dtype=torch.float32, device=rows.device)
    status = torch.empty((r,), dtype=torch.int32, device=rows.device)
    ws = solver_workspace(r, rows.device)
    with _on(rows):
        check(lib().lsq_solve_rows(rows.data_ptr(), r, m, skip, int(ternary), float(alpha), v12.data_ptr(),
                                   status.data_ptr(), ws.data_ptr(), ws.numel(), stream_ptr(rows.device)), 'lsq_solve_rows')
    return v12, status


def pack_weight(w: torch.Tensor, geom: ConvGeom, scales: torch.Tensor):
    """scales [k, O] -> (wbits int64 [k * words], wsum int32 [k, O, taps])."""
    w = _f32c(w)
    scales = _f32c(scales)
    k = scales.shape[0]
    words = lib().lsq_weight_plane_words(ctypes.byref(geom))
    wbits = torch.empty((k * words,), dtype=torch.int64, device=w.device)
    wsum = torch.empty((k, geom.O, geom.KH * geom.KW), dtype=torch.int32, device=w.device)
    with _on(w):
        check(lib().lsq_pack_weight(w.data_ptr(), ctypes.byref(geom), k, scales.data_ptr(), wbits.data_ptr(),
                                    wsum.data_ptr(), stream_ptr(w.device)), 'lsq_pack_weight')
    return wbits, wsum


def act_plane_words(geom: ConvGeom) -> int:
    return lib().lsq_act_plane_words(ctypes.byref(geom))


def out_hw(geom: ConvGeom):
    ho = (geom.H + 2 * geom.pad_h - geom.dil_h * (geom.KH - 1) - 1) // geom.stride_h + 1
    wo = (geom.W + 2 * geom.pad_w - geom.dil_w * (geom.KW - 1) - 1) // geom.stride_w + 1
    return ho, wo


ACT_NONE, ACT_RELU, ACT_PRELU, ACT_PRELU_CHANNEL = 0, 1, 2, 3


def _act(relu: bool, prelu: Optional[torch.Tensor], out_channels: int):
    """(LSQ_ACT_* code, slope pointer) of the fused epilogue: ``prelu`` is an nn.PReLU weight (1 or O slopes)."""
    if prelu is None:
        return (ACT_RELU if relu else ACT_NONE), None
    if relu:
        raise LsqHipError('relu and prelu are exclusive')
    w = _f32c(prelu.detach())
    if w.numel() not in (1, out_channels):
        raise LsqHipError(f'PReLU with {w.numel()} slopes on {out_channels} channels')
    return (ACT_PRELU if w.numel() == 1 else ACT_PRELU_CHANNEL), w


def xnor_conv2d(planes: torch.Tensor, kx: int, xscales: torch.Tensor, wbits: torch.Tensor, wsum: torch.Tensor,
                wscales: torch.Tensor, bias: Optional[torch.Tensor], geom: ConvGeom, y: torch.Tensor,
                relu: bool = False, res_pre: Optional[torch.Tensor] = None,
                res_post: Optional[torch.Tensor] = None, prelu: Optional[torch.Tensor] = None,
                y_layout: int = LAYOUT_NCHW, res_layout: int = LAYOUT_NCHW) -> None:
    """y = act(conv + bias + res_pre) + res_post, act = ReLU (``relu``), PReLU (``prelu`` = its weight) or identity
    (the fused block epilogue is optional).  ``y_layout`` / ``res_layout`` = LAYOUT_SPLIT3: ``y`` / the residuals are the
    [N, 3 S] buffers of three-stream tensors (quant.binary.layouts); ``y.numel()`` below then counts the padding too -- the
    accounting uses the geometry."""
    act, slope = _act(relu, prelu, geom.O)
    m = geom.C * geom.H * geom.W
    ho_, wo_ = out_hw(geom)
    ynum = geom.N * geom.O * ho_ * wo_
    macs = ynum * (geom.C // geom.groups) * geom.KH * geom.KW * kx * wscales.shape[0]
    nres = (res_pre is not None) + (res_post is not None)
    if ynum >= XNOR_MFMA_MAX_OUTPUTS and (geom.KH, geom.KW, geom.groups, geom.dil_h, geom.dil_w) == (3, 3, 1, 1, 1) \
            and geom.C in (64, 128, 256, 512) and geom.O % 32 == 0 and 'outputs' not in _xnor_limit_warned:
        # (csrc/lsq_xnor_mfma.hip indexes its output with 32 bits: a call this large is served by the popcount kernel -- same
        #  bits, about half the speed.  Said once, not silently.)
        _xnor_limit_warned.add('outputs')
        warnings.warn(f'lsq_xnor_conv2d: {ynum} outputs (2^30 or more): this call runs on the popcount kernel instead of the '
                      'int8 matrix-core kernel (same result, about half the speed); split the batch to stay below 2^30 outputs')
    # algorithmic bytes: planes read + fp32 output written + every residual operand of the fused epilogue read
    with _on(y), (_Timed('lsq_xnor_conv2d', geom.N * kx * m // 8 + 4 * ynum * (1 + nres), macs, f'C{geom.C}_H{geom.H}_s{geom.stride_h}',
                         geom.N * kx * m // 8 + 4 * ynum) if _timing is not None else _UNTIMED):
        check(lib().lsq_xnor_conv2d_layout(planes.data_ptr(), kx, xscales.data_ptr(), wbits.data_ptr(), wsum.data_ptr(),
                                           wscales.shape[0], wscales.data_ptr(), ptr(bias), ctypes.byref(geom), act, ptr(slope),
                                           ptr(res_pre), ptr(res_post), res_layout, y.data_ptr(), y_layout,
                                           stream_ptr(y.device)), 'lsq_xnor_conv2d')


E_UNSUPPORTED = -6


def xnor_conv2d_chain(planes: torch.Tensor, xscales: Optional[torch.Tensor], x_units: Optional[torch.Tensor], x_alpha: float,
                      wbits: torch.Tensor, wsum: torch.Tensor, wscales: torch.Tensor, bias: Optional[torch.Tensor],
                      geom: ConvGeom, y: torch.Tensor, relu: bool = False, res_pre: Optional[torch.Tensor] = None,
                      res_post: Optional[torch.Tensor] = None, prelu: Optional[torch.Tensor] = None,
                      nxt: Optional[NextLs1] = None) -> bool:
    """lsq_xnor_conv2d for a CHAIN of 1-bit layers: the activation scale from the row sums ``x_units`` a previous call's
    epilogue left (or ``xscales`` [1, N]), and / or the next layer's quantizer in this call's epilogue (``nxt``).  Returns
    False -- nothing launched -- when the geometry is outside the integer-MFMA kernel (the caller takes the unchained
    calls: same bits)."""
    act, slope = _act(relu, prelu, geom.O)
    m = geom.C * geom.H * geom.W
    nres = (res_pre is not None) + (res_post is not None)
    macs = y.numel() * geom.C * geom.KH * geom.KW * wscales.shape[0]
    extra = 0 if nxt is None else y.numel() // 8
    with _on(y), (_Timed('lsq_xnor_conv2d', geom.N * m // 8 + 4 * y.numel() * (1 + nres) + extra, macs,
                         f'C{geom.C}_H{geom.H}_s{geom.stride_h}', geom.N * m // 8 + 4 * y.numel()) if _timing is not None else _UNTIMED):
        code = lib().lsq_xnor_conv2d_chain(planes.data_ptr(), ptr(xscales), ptr(x_units), float(x_alpha), wbits.data_ptr(),
                                           wsum.data_ptr(), wscales.shape[0], wscales.data_ptr(), ptr(bias), ctypes.byref(geom),
                                           act, ptr(slope), ptr(res_pre), ptr(res_post),
                                           None if nxt is None else ctypes.byref(nxt), y.data_ptr(), stream_ptr(y.device))
    if code == E_UNSUPPORTED:
        return False
    check(code, 'lsq_xnor_conv2d_chain')
    return True


def signw_prepare_weight(wbits: torch.Tensor, planes: int, geom: ConvGeom) -> Optional[torch.Tensor]:
    """The bf16 operand image of lsq_signw_conv2d's 3x3 fast path for the sign planes ``wbits`` (once per eval
    session, next to pack_weight); None when the geometry has no fast path."""
    nbytes = lib().lsq_signw_weight_bytes(ctypes.byref(geom), planes)
    if nbytes <= 0:
        return None
    out = torch.empty((nbytes,), dtype=torch.uint8, device=wbits.device)
    with _on(wbits):
        check(lib().lsq_signw_prepare_weight(wbits.data_ptr(), planes, ctypes.byref(geom), out.data_ptr(),
                                             stream_ptr(wbits.device)), 'lsq_signw_prepare_weight')
    return out


def signw_conv2d(x: torch.Tensor, alpha: float, wbits: torch.Tensor, wscales: torch.Tensor,
                 bias: Optional[torch.Tensor], geom: ConvGeom, y: torch.Tensor, pre: Optional[tuple] = None,
                 relu: bool = False, res_pre: Optional[torch.Tensor] = None,
                 res_post: Optional[torch.Tensor] = None, prelu: Optional[torch.Tensor] = None,
                 wprep: Optional[torch.Tensor] = None) -> None:
    """``wprep``: signw_prepare_weight(wbits, ...) of the same weights and geometry (same result, 3x3 fast path)."""
    x = _f32c(x)
    if wprep is not None:              # (the image depends on channels, out-channels, taps and planes only: any batch / image size)
        want = lib().lsq_signw_weight_bytes(ctypes.byref(geom), wscales.shape[0])
        if wprep.dtype != torch.uint8 or wprep.numel() != want or wprep.device != x.device:
            raise ValueError(f'wprep: expected {want} bytes of signw_prepare_weight output on {x.device} for this geometry, '
                             f'got {wprep.numel()} x {wprep.dtype} on {wprep.device}')
    act, slope = _act(relu, prelu, geom.O)
    flops = 2 * 2 * y.numel() * (geom.C // geom.groups) * geom.KH * geom.KW * wscales.shape[0]   # hi + lo passes
    nres = (res_pre is not None) + (res_post is not None)
    with _on(x), _Timed('lsq_signw_conv2d', 4 * x.numel() + 4 * y.numel() * (1 + nres), flops, f'C{geom.C}_H{geom.H}_s{geom.stride_h}',
                         4 * x.numel() + 4 * y.numel()):     # fp32 input read + fp32 output written (+ residuals read)
        check(lib().lsq_signw_conv2d(x.data_ptr(), float(alpha), None if pre is None else pre[0].data_ptr(),
                                     None if pre is None else pre[1].data_ptr(), wbits.data_ptr(), ptr(wprep), wscales.shape[0],
                                     wscales.data_ptr(), ptr(bias), ctypes.byref(geom), act, ptr(slope), ptr(res_pre),
                                     ptr(res_post), y.data_ptr(), stream_ptr(x.device)), 'lsq_signw_conv2d')


def pool_bias_relu_nhwc(x: torch.Tensor, kernel: int, stride: int, pad: int, bias: Optional[torch.Tensor],
                        relu: bool) -> torch.Tensor:
    """``relu(max_pool2d(x) + bias)`` of a channels-last fp32 tensor ``x`` (logical shape [N, C, H, W]) as
    one kernel that writes a contiguous NCHW tensor."""
    if x.dtype != torch.float32 or x.dim() != 4 or not x.is_contiguous(memory_format=torch.channels_last):
        raise LsqHipError('pool_bias_relu_nhwc needs a channels-last fp32 4-d tensor')
    n, c, h, w = x.shape
    ho, wo = (h + 2 * pad - kernel) // stride + 1, (w + 2 * pad - kernel) // stride + 1
    y = torch.empty((n, c, ho, wo), dtype=torch.float32, device=x.device)
    with _on(x), _Timed('lsq_pool_bias_relu_nhwc', 4 * x.numel() + 4 * y.numel(), 0):
        check(lib().lsq_pool_bias_relu_nhwc(x.data_ptr(), n, c, h, w, kernel, stride, pad,
                                            ptr(None if bias is None else _f32c(bias)), int(relu), y.data_ptr(),
                                            stream_ptr(x.device)), 'lsq_pool_bias_relu_nhwc')
    return y


_stem_guard = {}


def _stem_guard_state(device):
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _stem_guard.get(idx)
    if st is None:
        st = _stem_guard[idx] = {'flag': torch.zeros((1,), dtype=torch.int32, device=torch.device('cuda', idx)),
                                 'host': torch.zeros((1,), dtype=torch.int32).pin_memory(), 'event': None, 'tripped': False,
                                 'calls': 0}
    return st


def stem_overflow_tripped(device) -> bool:
    """True once a FINISHED stem_conv_pool(split=22) call on ``device`` has reported an operand outside the fp16 split's
    domain (|value| >= 65504 or NaN).  Never blocks: the kernel raises a STICKY device flag; its copy to pinned host
    memory rides on the launch stream after the first call and then after every 16th (a copy is a launch of its own:
    4 us per forward if made every time) and is looked at only when its event has completed -- so the report arrives
    up to 16 calls after the offending one (whose output holds inf / nan).  Callers switch to split 3 (any finite
    input) then."""
    st = _stem_guard_state(device)
    ev = st['event']
    if ev is not None and ev.query():
        st['event'] = None
        if int(st['host'][0]) != 0:
            st['tripped'] = True
    return st['tripped']


def stem_overflow_check(device) -> bool:
    """Blocking form of :func:`stem_overflow_tripped`: waits for the device and reads the flag itself -- for the end of an
    evaluation loop and in front of a graph capture, where a report that is 16 calls late would be too late."""
    st = _stem_guard_state(device)
    torch.cuda.synchronize(device)
    st['event'] = None
    if int(st['flag'].item()) != 0:
        st['tripped'] = True
    return st['tripped']


def stem_overflow_flag_raised(device) -> bool:
    """True while the DEVICE flag is up, i.e. a call since the last ``stem_overflow_reset`` saw an out-of-domain operand
    (``stem_overflow_check`` also reports a trip that was dealt with earlier).  Waits for the device."""
    st = _stem_guard_state(device)
    torch.cuda.synchronize(device)
    st['event'] = None
    raised = int(st['flag'].item()) != 0
    st['tripped'] = st['tripped'] or raised
    return raised


def stem_overflow_reset(device, keep_tripped: bool = False) -> None:
    """Forget an earlier report (after the caller has dealt with it); waits for the device.  ``keep_tripped``: lower the
    device flag but keep the host-side memory of it, so that callers stay on the bf16 split (``evaluate`` repeats a pass)."""
    torch.cuda.synchronize(device)
    st = _stem_guard_state(device)
    st['event'], st['tripped'], st['calls'] = None, bool(keep_tripped and st['tripped']), 0
    st['flag'].zero_()
    st['host'].zero_()


def stem_conv_pool(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, split: int = 3) -> torch.Tensor:
    """``max_pool2d(relu(conv2d(x, w, stride=2, padding=3) + bias), 3, 2, 1)`` for a 7x7 convolution 3 -> 64
    channels (batch norm already folded into ``w`` / ``bias``) as ONE kernel, NCHW fp32 in and out.  ``split``: how the
    fp32 operands are fed to the 16-bit matrix cores -- 3: three bf16 terms, six MFMA passes, fp32-class accuracy, any
    finite input; 2: two bf16 terms, three passes, ~2^-17 per product; 22: fp16 leading term + scaled fp16 remainder,
    three passes, fp32-class accuracy (2^-23 per product), operands below 65504 in magnitude (larger ones become
    inf / nan in the output, and the call reports them: ``stem_overflow_tripped``)."""
    x, w, bias = _f32c(x), _f32c(w), _f32c(bias)
    n, c, h, wd = x.shape
    if c != 3 or tuple(w.shape) != (64, 3, 7, 7) or wd % 2 or h < 8 or wd < 8 or x.data_ptr() % 8:
        raise LsqHipError('stem_conv_pool: 7x7 stride-2 convolution from 3 to 64 channels, even width, at least 8 x 8, 8-byte aligned')
    hc, wc = (h - 1) // 2 + 1, (wd - 1) // 2 + 1
    hp, wp = (hc - 1) // 2 + 1, (wc - 1) // 2 + 1
    y = torch.empty((n, 64, hp, wp), dtype=torch.float32, device=x.device)
    flops = 2 * (6 if split == 3 else 3) * n * 64 * hc * wc * 147      # 16-bit MFMA passes issued
    # (no guard while a HIP graph is being captured: the event / pinned-copy bookkeeping is host state of ONE call;
    #  the warm-up forwards in front of a capture run with it)
    guard = _stem_guard_state(x.device) if split == 22 and not torch.cuda.is_current_stream_capturing() else None
    if guard is not None:
        stem_overflow_tripped(x.device)                                   # (collect a finished report before the flag is reused)
    with _on(x), _Timed('lsq_stem_conv_pool', 4 * x.numel() + 4 * y.numel(), flops):
        check(lib().lsq_stem_conv_pool(x.data_ptr(), n, h, wd, w.data_ptr(), bias.data_ptr(), int(split), y.data_ptr(),
                                       None if guard is None else guard['flag'].data_ptr(), stream_ptr(x.device)),
              'lsq_stem_conv_pool')
        if guard is not None:
            guard['calls'] += 1
        if guard is not None and guard['event'] is None and guard['calls'] % 16 == 1:
            guard['host'].copy_(guard['flag'], non_blocking=True)
            guard['event'] = torch.cuda.Event()
            guard['event'].record()
    return y


def pointwise_conv(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], stride: int) -> torch.Tensor:
    """``conv2d(x, w[:, :, None, None], bias, stride)`` for a 1x1 kernel (the projection shortcut, batch norm
    already folded), NCHW fp32, exact fp32 on the matrix cores; C and O multiples of 64."""
    x, w = _f32c(x), _f32c(w)
    n, c, h, wd = x.shape
    o = w.shape[0]
    y = torch.empty((n, o, (h - 1) // stride + 1, (wd - 1) // stride + 1), dtype=torch.float32, device=x.device)
    with _on(x), _Timed('lsq_pointwise_conv', 4 * (y.numel() // o) * c + 4 * y.numel(), 2 * y.numel() * c):
        check(lib().lsq_pointwise_conv(x.data_ptr(), n, c, h, wd, w.data_ptr(), ptr(None if bias is None else _f32c(bias)), o,
                                       int(stride), y.data_ptr(), stream_ptr(x.device)), 'lsq_pointwise_conv')
    return y


def quant_values(x: torch.Tensor, scales: Optional[torch.Tensor], alpha: float) -> torch.Tensor:
    """x_q = sum_i v_i b_i of the quantizer chain for rows x[r] (dim 0) and plane scales [k, rows] (None: the clamp
    alone) -- what the reference's quantizer_* functions return, on the device in one pass."""
    x = _f32c(x)
    rows, m = x.shape[0], x.numel() // max(x.shape[0], 1)
    k = 0 if scales is None else scales.shape[0]
    sc = None if scales is None else _f32c(scales)
    out = torch.empty_like(x)
    with _on(x), _Timed('lsq_quant_values', 8 * x.numel(), 0):
        check(lib().lsq_quant_values(x.data_ptr(), rows, m, k, ptr(sc), float(alpha), out.data_ptr(), stream_ptr(x.device)),
              'lsq_quant_values')
    return out


def ste_backward(x: torch.Tensor, grad_q: torch.Tensor, scales: Optional[torch.Tensor], alpha: float) -> torch.Tensor:
    """Gradient with respect to ``x`` of <grad_q, quantizer(clamp(x))> through the straight-through estimator
    (quant/binary/ste.py:51-66) and the clamp; rows = dim 0, plane scales [k, rows] (None: the clamp alone)."""
    x, grad_q = _f32c(x), _f32c(grad_q)
    if grad_q.shape != x.shape:
        raise ValueError(f'gradient of shape {tuple(grad_q.shape)} for an input of shape {tuple(x.shape)}')
    rows, m = x.shape[0], x.numel() // max(x.shape[0], 1)
    k = 0 if scales is None else scales.shape[0]
    sc = None if scales is None else _f32c(scales)
    out = torch.empty_like(x)
    with _on(x), _Timed('lsq_ste_backward', 12 * x.numel(), 0):
        check(lib().lsq_ste_backward(x.data_ptr(), grad_q.data_ptr(), rows, m, k, ptr(sc), float(alpha), out.data_ptr(),
                                     stream_ptr(x.device)), 'lsq_ste_backward')
    return out


def xnor_impl(mode) -> int:
    """Test / profiling hook (include/lsq_hip_debug.h): 1 / True = every XNOR convolution through the popcount kernel, 0 / False
    = the dispatcher picks the matrix-core kernel where it applies (fp4 operands on the scaled MFMA, the default), 2 = the
    same with the int8 matrix-core kernel of rounds 2-5 (identical results).  Returns the old value."""
    return lib().lsq_debug_xnor_impl(int(mode))


@contextlib.contextmanager
def solver_trace(rows: int, device):
    """Test hook (include/lsq_hip_debug.h): inside the block every LS-2 / LS-T solve stores the sorted position of the
    candidate it chose, one int32 per row, into the yielded tensor (read it after a synchronize)."""
    buf = torch.full((rows,), -2, dtype=torch.int32, device=device)
    lib().lsq_debug_solver_trace(buf.data_ptr())
    try:
        yield buf
    finally:
        torch.cuda.synchronize(device)
        lib().lsq_debug_solver_trace(None)


@contextlib.contextmanager
def debug_switches(xnor_popcount: Optional[bool] = None, force_streaming: Optional[bool] = None,
                   fused_mode: Optional[int] = None):
    """Set the library's test hooks (include/lsq_hip_debug.h) for the duration of a ``with`` block and restore the
    previous values afterwards, whatever happens inside.  Process-wide: meant for single-threaded tests and scripts."""
    handle = lib()
    old = {}
    try:
        if xnor_popcount is not None:
            old['lsq_debug_xnor_impl'] = handle.lsq_debug_xnor_impl(int(xnor_popcount))
        if force_streaming is not None:
            old['lsq_debug_force_streaming'] = handle.lsq_debug_force_streaming(int(bool(force_streaming)))
        if fused_mode is not None:
            old['lsq_debug_fused_mode'] = handle.lsq_debug_fused_mode(int(fused_mode))
        yield
    finally:
        for name, value in old.items():
            getattr(handle, name)(value)
