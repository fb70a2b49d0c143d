#!/usr/bin/env python3
"""Developer tool: latency of the headline network at small batches, solving the activation scales per batch
(moving_average_mode off: the benchmark's configuration) and with moving-average scales (eval_only: no solve)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd')]
import torch  # noqa: E402
import bench  # noqa: E402

dev = 'cuda:0'
models = {}
models['solve'] = bench.build_model(bench.imagenet_arch('ls-2', 3), dev)
arch = bench.imagenet_arch('ls-2', 3)
arch.update(moving_average_mode='eval_only', moving_average_momentum=0.0)
m = bench.build_model(arch, dev)
m.train()
with torch.no_grad():
    m(torch.randn(4, 3, 224, 224, device=dev))
models['moving average'] = m.eval()
for n in (1, 8, 32, 256):
    x = torch.randn(n, 3, 224, 224, device=dev)
    row = []
    for name, model in models.items():
        with torch.no_grad():
            for _ in range(10):
                model(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(100):
                model(x)
            torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 10
        from quant.common.graph_replay import GraphedForward
        fwd = GraphedForward(model, x)
        for _ in range(5):
            fwd.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            fwd.replay()
        torch.cuda.synchronize()
        gms = (time.perf_counter() - t0) * 10
        row.append(f'{name}: eager {ms:.3f} ms, graph replay {gms:.3f} ms ({n / gms * 1e3:.0f} images/s)')
    print(f'batch {n:3d}  ' + '   '.join(row), flush=True)
