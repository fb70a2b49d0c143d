#!/usr/bin/env python3
"""Developer tool: launch lsq_xnor_conv2d on one ResNet-18 layer shape a few times (for rocprofv3 --pmc).

    python scripts/xnor_one.py C H O stride [iters]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd')]
import torch  # noqa: E402
from quant import _hip  # noqa: E402

c, h, o, stride = (int(v) for v in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 3
n, dev, k = 256, 'cuda:0', 2
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
for _ in range(iters):
    _hip.xnor_conv2d(planes, k, scales, wbits, wsum, wsc, bias, g, y, relu=True, res_post=res)
torch.cuda.synchronize()
