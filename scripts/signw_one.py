#!/usr/bin/env python3
"""Developer tool: launch lsq_signw_conv2d on one ResNet-18 layer shape a few times (for rocprofv3 --pmc).

    python scripts/signw_one.py C H O stride [iters]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd')]
import torch  # noqa: E402
from quant import _hip  # noqa: E402

c, h, o, stride = (int(v) for v in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 3
n, dev = 256, 'cuda:0'
x = torch.randn(n, c, h, h, device=dev)
w = torch.randn(o, c, 3, 3, device=dev)
g = _hip.make_geom(n, c, h, h, o, 3, 3, (stride, stride), (1, 1), (1, 1), 1)
wsc = w.abs().mean(dim=(1, 2, 3)).view(1, -1).contiguous()
wbits, _ = _hip.pack_weight(w, g, wsc)
wprep = None if os.environ.get('LSQ_SIGNW_GENERAL') else _hip.signw_prepare_weight(wbits, 1, g)   # fast path unless asked otherwise
ho, wo = _hip.out_hw(g)
y = torch.empty((n, o, ho, wo), device=dev)
bias = torch.zeros(o, device=dev)
for _ in range(iters):
    _hip.signw_conv2d(x, 2.0, wbits, wsc, bias, g, y, wprep=wprep)
torch.cuda.synchronize()
