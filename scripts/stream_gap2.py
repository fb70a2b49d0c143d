#!/usr/bin/env python3
"""Developer tool (round 5): host enqueue time against GPU start time of every forward under two streams."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd')]
import torch  # noqa: E402
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dev = torch.device('cuda', 0)
model = bench.build_model(bench.imagenet_arch(), dev)
xs = [torch.randn(256, 3, 224, 224, device=dev) for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
with torch.no_grad():
    for rep in range(2):
        for k in range(2):
            with torch.cuda.stream(streams[k]):
                model(xs[k])
    torch.cuda.synchronize()
    base = torch.cuda.Event(enable_timing=True)
    base.record()
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rows = []
    for i in range(steps):
        k = i % 2
        with torch.cuda.stream(streams[k]):
            h0 = time.perf_counter() - t0
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
            out = model(xs[k])
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            h1 = time.perf_counter() - t0
        rows.append((i, k, h0, h1, e0, e1))
    torch.cuda.synchronize()
    off = None
    for i, k, h0, h1, e0, e1 in rows:
        g0, g1 = base.elapsed_time(e0), base.elapsed_time(e1)
        if off is None:
            off = g0 - h0 * 1e3
        print(f'forward {i:2d} stream {k}: host enqueue {h0 * 1e3:8.3f} .. {h1 * 1e3:8.3f} ms;  GPU start {g0 - off:8.3f} end {g1 - off:8.3f} ms '
              f'(start - host enqueue start = {g0 - off - h0 * 1e3:7.3f} ms)')
