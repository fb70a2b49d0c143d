#!/usr/bin/env python3
"""Developer tool: per-layer statistics of the scale solve inside the real ResNet-18 forward (flagged
level-1 bins, gathered keys, slots that fell back to the block-level path), read from the workspace."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd')]
import numpy as np
import torch
import bench
from quant import _hip
from quant.binary.binary_conv import QuantConv2d

dev = 'cuda:0'
model = bench.build_model(bench.imagenet_arch(), dev)
x = torch.randn(64, 3, 224, 224, device=dev)
WS_ROW = _hip.lib().lsq_solver_workspace_bytes(1)

_orig = _hip.act_quant


def act_quant(x_in, geom, *rest, **kw):
    _orig(x_in, geom, *rest, **kw)
    torch.cuda.synchronize()
    ws = _hip.solver_workspace(64, dev)[:64 * WS_ROW].cpu().numpy().reshape(64, WS_ROW)
    hdr = ws[:, :32].copy().view(np.uint32).reshape(64, 8)
    gathered = ws[:, 24:32].copy().view(np.float64).reshape(64)
    tflag, n, pad = hdr[:, 0], hdr[:, 2], hdr[:, 3]
    print(f'{tuple(x_in.shape)}  n={n[0]}  flagged bins mean {tflag.mean():.1f} max {tflag.max()}  '
          f'gathered mean {gathered.mean():.0f} max {gathered.max():.0f}  slow slots mean {(pad & 0xFFFF).mean():.2f} '
          f'max {(pad & 0xFFFF).max()}  row-pass slots max {(pad >> 16).max()}')


_hip.act_quant = act_quant          # (the fused blocks call the binding directly: no module hooks to hang on)
with torch.no_grad():
    model(x)
