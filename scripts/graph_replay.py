#!/usr/bin/env python3
"""Developer tool: one eval forward captured in a HIP graph (torch.cuda.CUDAGraph) vs eager launches."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd')]
import torch  # noqa: E402
import bench  # noqa: E402

act = sys.argv[1] if len(sys.argv) > 1 else 'ls-2'
model = bench.build_model(bench.imagenet_arch(act, 3 if act == 'ls-2' else 2), 'cuda:0')
x = torch.randn(256, 3, 224, 224, device='cuda:0')


def timed(fn, n=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


with torch.no_grad():
    for _ in range(5):
        ref = model(x)
    print(f'eager  {timed(lambda: model(x)):.3f} ms/step')
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            model(x)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = model(x)
    g.replay()
    torch.cuda.synchronize()
    print('graph output equals eager:', torch.equal(out, ref))
    print(f'graph  {timed(g.replay):.3f} ms/step')
