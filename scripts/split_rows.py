#!/usr/bin/env python3
"""Developer tool: lsq_act_quant (ls-2) at small batches on the four ResNet-18 shapes -- rows shared by several workgroups
(three launches of the streaming kernels, round 4: opt-in, lsq_debug_no_row_split(0)) against the single-launch kernel (one
workgroup per row: the default at every batch), us per call;
and the headline network under graph replay at those batches."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd')]
import torch  # noqa: E402
import bench  # noqa: E402
from quant import _hip  # noqa: E402
from quant.common.graph_replay import GraphedForward  # noqa: E402


def call_us(x, no_split):
    n, c, h, w = x.shape
    g = _hip.make_geom(n, c, h, w, c, 3, 3, (1, 1), (1, 1), (1, 1), 1)
    planes = torch.zeros(2 * _hip.act_plane_words(g), dtype=torch.int64, device='cuda')
    scales = torch.empty((2, n), device='cuda')
    with _hip.debug_switches(no_row_split=no_split):
        for _ in range(5):
            _hip.act_quant(x, g, 2, 2, 3, 3.0, planes, scales)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            _hip.act_quant(x, g, 2, 2, 3, 3.0, planes, scales)
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 20


for n in (1, 8, 32, 64):
    row = []
    for c, h in ((64, 56), (128, 28), (256, 14), (512, 7)):
        x = torch.randn(n, c, h, h, device='cuda')
        row.append(f'{c}x{h}: {call_us(x, False):6.1f} / {call_us(x, True):6.1f}')
    print(f'batch {n:3d}  split / single-launch (us):  ' + '   '.join(row), flush=True)
model = bench.build_model(bench.imagenet_arch('ls-2', 3), 'cuda:0')
for n in (1, 8, 32, 64):
    x = torch.randn(n, 3, 224, 224, device='cuda')
    out = []
    for no_split in (False, True):
        with _hip.debug_switches(no_row_split=no_split):
            fwd = GraphedForward(model, x)
        for _ in range(5):
            fwd.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            fwd.replay()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) * 10)
    print(f'batch {n:3d}  headline network, graph replay: split {out[0]:.3f} ms, single-launch {out[1]:.3f} ms', flush=True)
