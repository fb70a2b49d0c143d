#!/usr/bin/env python3
"""Developer tool: the headline forward as two half batches on two HIP streams (tails and the quantizer's latency
chains of one half overlapping the other half's kernels) against one batch on one stream."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd')]
import torch  # noqa: E402
import bench  # noqa: E402

dev = torch.device('cuda', 0)
model = bench.build_model(bench.imagenet_arch(), dev)
x = torch.randn(256, 3, 224, 224, device=dev)


def timeit(fn, n=60):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    y0 = model(x)
    print(f'one stream, batch 256: {timeit(lambda: model(x)):.3f} ms')
    for parts in (2, 4):
        streams = [torch.cuda.Stream() for _ in range(parts)]
        chunks = x.chunk(parts)
        outs = [None] * parts

        def run():
            cur = torch.cuda.current_stream()
            for s in streams:
                s.wait_stream(cur)
            for i, s in enumerate(streams):
                with torch.cuda.stream(s):
                    outs[i] = model(chunks[i])
            for s in streams:
                cur.wait_stream(s)

        run()
        torch.cuda.synchronize()
        print(f'{parts} streams x batch {256 // parts}: {timeit(run):.3f} ms; equal to the single-stream logits:',
              torch.equal(torch.cat(outs), y0))
