#!/usr/bin/env python3
"""Developer tool: how far behind the submissions the results of quant.common.stream_pipeline are consumed.  The consumer's
wait (caller's stream waits for forward k - lag) sits in FRONT of the input-ready event of the next submission on the caller's
stream: with lag = depth - 1 that event hangs on the forward that is about to finish on the very stream the submission goes to,
and the stream idles for two cross-queue hand-offs after every forward.
python scripts/pipeline_lag.py [ls-2|fp]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd')]
import torch  # noqa: E402
import bench  # noqa: E402
from quant.common.stream_pipeline import StreamPipeline  # noqa: E402

act = sys.argv[1] if len(sys.argv) > 1 else 'ls-2'
dev = torch.device('cuda', 0)
model = bench.build_model(bench.imagenet_arch(act, 3 if act == 'ls-2' else 2), dev)
x = torch.randn(256, 3, 224, 224, device=dev)


def timed(limit, steps=200, reps=4):
    pipe = StreamPipeline(model, dev, 2)

    def run(n):
        window = []
        for _ in range(n):
            window.append(pipe.submit(x))
            if len(window) >= limit:
                window.pop(0).result()
        while window:
            window.pop(0).result()

    run(8)
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(steps)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps * 1e3)
    return best


with torch.no_grad():
    for r in range(2):
        for limit in (2, 3, 4, 1000):
            ms = timed(limit)
            print(f'{act}: results consumed when {limit} forwards are pending: {ms:.3f} ms per batch of 256 ({256 / ms * 1e3:.0f} images/s)', flush=True)
