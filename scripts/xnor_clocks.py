#!/usr/bin/env python3
"""Developer tool (library built with -DLSQ_XNOR_CLOCKS, LSQ_HIP_LIB pointing at it): shader-clock stamps of wave 0's first
tiles in lsq_xnor_conv2d's matrix-core kernel -- tile start, main loop end, epilogue end."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd'), os.path.join(ROOT, 'scripts')]
import torch  # noqa: E402
from quant import _hip  # noqa: E402
from kbench import SHAPES  # noqa: E402

n, dev, k = 256, 'cuda:0', 2
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
    for _ in range(3):
        _hip.xnor_conv2d(planes, k, scales, wbits, wsum, wsc, bias, g, y, relu=True, res_pre=res)
    torch.cuda.synchronize()
    t = y.view(-1)[:120].view(torch.int64).cpu().view(4, 15)
    for b in range(2):
        r = t[b]
        d = [(int(r[i + 1]) - int(r[i])) for i in range(14) if int(r[i + 1]) and int(r[i])]
        print(f'C={c:4d} H={h:3d} s={stride} wg{b}: ' + ' '.join(f'{v:6d}' for v in d) + '   (main, wait + convert, activation, stores, top of loop; main, ...)')
