#!/usr/bin/env python3
"""Developer tool: the CIFAR-100 ResNet-18 ls-1 network (BASELINE configs[1], batch 100) with and without the fused
quantize + convolve launch (lsq_ls1_conv2d): eager and graph-replay time per forward, kernel time per entry point."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd')]
import torch  # noqa: E402
import bench  # noqa: E402
from quant import _hip  # noqa: E402
from quant.binary.binary_conv import QuantConv2d  # noqa: E402
from quant.common.graph_replay import GraphedForward  # noqa: E402

dev = torch.device('cuda', 0)
model = bench.build_model(bench.cifar_arch(), dev)
x = torch.randn(100, 3, 32, 32, generator=torch.Generator().manual_seed(0)).to(dev)
outs = {}
for fuse in (False, True):
    for m in model.modules():
        if isinstance(m, QuantConv2d):
            m.fuse_small = fuse
    with torch.no_grad():
        for _ in range(5):
            y = model(x)
        outs[fuse] = y.clone()
        torch.cuda.synchronize()
        _hip.enable_timing(True)
        model(x)
        torch.cuda.synchronize()
        table = _hip.drain_timing(by_tag=True)
        _hip.enable_timing(False)
        el, mn, med = bench.timed_forward(lambda: model(x), 200, 5)
    g = GraphedForward(model, x)
    gel, gmn, gmed = bench.timed_forward(g.replay, 200, 5)
    per = {}
    for (name, tag), v in table.items():
        per[name] = per.get(name, 0.0) + v[1]
    print(f'fused={fuse}: eager {1e3 * el / 200:.3f} ms ({100 * 200 / el:.0f} img/s), graph replay {1e3 * gel / 200:.3f} ms '
          f'({100 * 200 / gel:.0f} img/s); kernels per step (ms): ' + ', '.join(f'{k} {v:.3f}' for k, v in sorted(per.items())))
    print('   per shape (us): ' + ', '.join(f'{n}:{t} {1e3 * v[1] / v[0]:.1f}' for (n, t), v in sorted(table.items(), key=str) if t))
print('logits equal bit for bit:', bool(torch.equal(outs[False], outs[True])))
