#!/usr/bin/env python3
"""Developer tool: lsq_xnor_conv2d launch time with / without the fused ReLU and residual."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd'), os.path.join(ROOT, 'scripts')]
import torch  # noqa: E402
from quant import _hip  # noqa: E402
from kbench import SHAPES, timeit  # noqa: E402

n, dev, k = 256, 'cuda:0', 2
tot = [0.0, 0.0, 0.0, 0.0]
for c, h, o, stride, count in SHAPES:
    x = torch.randn(n, c, h, h, device=dev)
    w = torch.randn(o, c, 3, 3, device=dev)
    g = _hip.make_geom(n, c, h, h, o, 3, 3, (stride, stride), (1, 1), (1, 1), 1)
    planes = torch.zeros(k * _hip.act_plane_words(g), dtype=torch.int64, device=dev)
    scales = torch.empty((k, n), device=dev)
    wsc = w.abs().mean(dim=(1, 2, 3)).view(1, -1).contiguous()
    wbits, wsum = _hip.pack_weight(w, g, wsc)
    ho, wo = _hip.out_hw(g)
    y = torch.empty((n, o, ho, wo), device=dev)
    res = torch.randn_like(y)
    bias = torch.zeros(o, device=dev)
    _hip.act_quant(x, g, 2, k, 3, 3.0, planes, scales)
    t0 = timeit(lambda: _hip.xnor_conv2d(planes, k, scales, wbits, wsum, wsc, bias, g, y), 10)
    t1 = timeit(lambda: _hip.xnor_conv2d(planes, k, scales, wbits, wsum, wsc, bias, g, y, relu=True), 10)
    t2 = timeit(lambda: _hip.xnor_conv2d(planes, k, scales, wbits, wsum, wsc, bias, g, y, relu=True, res_pre=res), 10)
    slope = torch.full((o,), 0.25, device=dev)
    t3 = timeit(lambda: _hip.xnor_conv2d(planes, k, scales, wbits, wsum, wsc, bias, g, y, res_pre=res, prelu=slope), 10)
    for i, t in enumerate((t0, t1, t2, t3)):
        tot[i] += t * count
    print(f'C={c:4d} H={h:3d} O={o:4d} s={stride} | plain {t0:6.1f}  +relu {t1:6.1f}  +relu+res {t2:6.1f}  +prelu+res {t3:6.1f}   (x{count})')
print('per forward: plain %.2f ms, +relu %.2f ms, +relu+res %.2f ms, +prelu+res %.2f ms' % tuple(t / 1e3 for t in tot))
