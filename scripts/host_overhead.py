#!/usr/bin/env python3
"""Developer tool: host-side enqueue time of one forward (no synchronisation) vs GPU time per step."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd')]
import torch  # noqa: E402
import bench  # noqa: E402

model = bench.build_model(bench.imagenet_arch(), 'cuda:0')
x = torch.randn(256, 3, 224, 224, device='cuda:0')
with torch.no_grad():
    for _ in range(5):
        model(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        model(x)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'host enqueue {1e3 * (t1 - t0) / 20:.3f} ms/step, total {1e3 * (t2 - t0) / 20:.3f} ms/step')
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        model(x)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats('tottime').print_stats(14)
