#!/usr/bin/env python3
"""Developer tool: host-side timeline of the two-stream evaluation loop -- how long every submit() call takes on the host and
how far the host runs ahead of the device.  python scripts/host_timeline.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd')]
import torch  # noqa: E402
import bench  # noqa: E402
from quant.common.stream_pipeline import StreamPipeline  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device('cuda', 0)
model = bench.build_model(bench.imagenet_arch('ls-2', 3), dev)
x = torch.randn(256, 3, 224, 224, device=dev)

with torch.no_grad():
    pipe = StreamPipeline(model, dev, 2)
    window = []
    for _ in range(8):
        window.append(pipe.submit(x))
    while window:
        window.pop(0).result()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rows = []
    pend = []
    for k in range(steps):
        a = time.perf_counter()
        p = pipe.submit(x)
        b = time.perf_counter()
        pend.append(p)
        window.append(p)
        if len(window) >= 2:
            window.pop(0).result()
        done = sum(1 for q in pend if q._done is None or q._done.query())
        rows.append((1e3 * (a - t0), 1e3 * (b - a), k + 1 - done))
    torch.cuda.synchronize()
    total = 1e3 * (time.perf_counter() - t0)
print(f'{steps} steps in {total:.2f} ms: {total / steps:.3f} ms per step')
print('step, submit() entered at ms, host time inside submit() ms, forwards submitted and not finished after it')
for k, (a, d, ahead) in enumerate(rows):
    print(f'{k:4d} {a:9.3f} {d:7.3f} {ahead:3d}')

# the same forward, host time only (the device far behind): per-module cost of the enqueue
import cProfile, pstats, io  # noqa: E402
with torch.no_grad():
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(4):
        model(x)
    pr.disable()
    torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45)
print(s.getvalue()[:9000])
