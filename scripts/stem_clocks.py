#!/usr/bin/env python3
"""Developer tool (library built with -DLSQ_STEM_CLOCKS, LSQ_HIP_LIB pointing at it): shader-clock stamps of wave 0 of four
workgroups of lsq_stem_conv_pool -- per chunk: barrier, convert + barrier, conv rows (MFMAs + horizontal max), barrier,
vertical max + stores."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd')]
import torch  # noqa: E402
from quant import _hip  # noqa: E402

x = torch.randn(256, 3, 224, 224, device='cuda')
w = torch.randn(64, 3, 7, 7, device='cuda') * 0.05
b = torch.randn(64, device='cuda')
for split in (22, 2):
    for _ in range(3):
        y = _hip.stem_conv_pool(x, w, b, split)
    torch.cuda.synchronize()
    t = y.view(-1)[:192].view(torch.int64).cpu().view(4, 24)
    for r in t:
        v = [int(a) for a in r if int(a)]
        d = [v[i + 1] - v[i] for i in range(len(v) - 1)]
        print(f'split {split}: start {d[0]:6d} | ' + ' | '.join(' '.join(f'{c:6d}' for c in d[1 + 5 * k:6 + 5 * k]) for k in range(4)) + f'  total {v[-1] - v[0]}')
