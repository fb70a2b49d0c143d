#!/usr/bin/env python3
"""Developer tool: lsq_signw_conv2d launch time with / without the fused batch norm, ReLU and residual."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd'), os.path.join(ROOT, 'scripts')]
import torch  # noqa: E402
from quant import _hip  # noqa: E402
from kbench import SHAPES, timeit  # noqa: E402

n, dev = 256, 'cuda:0'
for c, h, o, stride, _ in SHAPES:
    w = torch.randn(o, c, 3, 3, device=dev)
    g = _hip.make_geom(n, c, h, h, o, 3, 3, (stride, stride), (1, 1), (1, 1), 1)
    wsc = w.abs().mean(dim=(1, 2, 3)).view(1, -1).contiguous()
    wbits, _ = _hip.pack_weight(w, g, wsc)
    ho, wo = _hip.out_hw(g)
    y = torch.empty((n, o, ho, wo), device=dev)
    res = torch.randn_like(y)
    bias = torch.zeros(o, device=dev)
    pre = (torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev))
    out = []
    for name, x in (('gauss', torch.randn(n, c, h, h, device=dev)),
                    ('relu ', torch.randn(n, c, h, h, device=dev).clamp_(min=0)),
                    ('zeros', torch.zeros(n, c, h, h, device=dev))):
        t0 = timeit(lambda: _hip.signw_conv2d(x, 2.0, wbits, wsc, bias, g, y), 10)
        t1 = timeit(lambda: _hip.signw_conv2d(x, 2.0, wbits, wsc, bias, g, y, pre=pre), 10)
        t2 = timeit(lambda: _hip.signw_conv2d(x, 2.0, wbits, wsc, bias, g, y, pre=pre, relu=True, res_pre=res), 10)
        out.append(f'{name}: plain {t0:6.1f}  +bn {t1:6.1f}  +bn+relu+res {t2:6.1f}')
    print(f'C={c:4d} H={h:3d} O={o:4d} s={stride} | ' + ' | '.join(out))
