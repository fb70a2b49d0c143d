#!/usr/bin/env python3
"""Developer tool (round 5): why a stream's next forward starts late under two streams.  Each forward is preceded by a tiny
fill kernel on its stream (a marker in the rocprofv3 kernel trace); optional stream priority.
python scripts/stream_gap.py [priority] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd')]
import torch  # noqa: E402
import bench  # noqa: E402

prio = int(sys.argv[1]) if len(sys.argv) > 1 else 0
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device('cuda', 0)
model = bench.build_model(bench.imagenet_arch(), dev)
xs = [torch.randn(256, 3, 224, 224, device=dev) for _ in range(2)]
marks = [torch.zeros(64, device=dev) for _ in range(2)]
streams = [torch.cuda.Stream(priority=prio) for _ in range(2)]
with torch.no_grad():
    for k in range(2):
        model(xs[k])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        k = i % 2
        with torch.cuda.stream(streams[k]):
            marks[k].fill_(float(i))
            out = model(xs[k])
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f'priority {prio}: {dt / steps * 1e3:.3f} ms per batch, host issue {th / steps * 1e3:.3f} ms per batch; '
      f'GPU_MAX_HW_QUEUES={os.environ.get("GPU_MAX_HW_QUEUES")}')
