#!/usr/bin/env python3
"""Developer tool (round 5): how the steps of the headline forward are ISSUED -- streams x micro-batch, the projection
shortcut on a side stream -- against one stream, batch 256.  Every variant processes the same 256 images per step; steps are
queued back to back (no join between steps except the final synchronize), logits compared with the single-stream forward.
python scripts/sched_variants.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd')]
import torch  # noqa: E402
import bench  # noqa: E402
from quant.models import resnet  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device('cuda', 0)
model = bench.build_model(bench.imagenet_arch(), dev)
x = torch.randn(256, 3, 224, 224, device=dev)


def run_variant(nstreams, micro, side, steps=steps):
    """`nstreams` HIP streams; the 256 images of a step are cut into chunks of `micro`; chunk j of every step goes to stream
    j % nstreams (micro == 256: consecutive steps alternate between the streams)."""
    resnet.SIDE_STREAM_SHORTCUT = side
    chunks = list(x.split(micro))
    streams = [torch.cuda.Stream() for _ in range(nstreams)] if nstreams > 1 else [torch.cuda.current_stream()]
    last = {}

    def run(n):
        cur = torch.cuda.current_stream()
        if nstreams > 1:
            for s in streams:
                s.wait_stream(cur)
        j = 0
        for _ in range(n):
            for ci, c in enumerate(chunks):
                with torch.cuda.stream(streams[j % nstreams]):
                    last[ci] = model(c)
                j += 1
        if nstreams > 1:
            for s in streams:
                cur.wait_stream(s)

    run(4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dt / steps * 1e3, t_host / steps * 1e3, torch.cat([last[i] for i in range(len(chunks))])


with torch.no_grad():
    resnet.SIDE_STREAM_SHORTCUT = False
    ref = model(x).clone()
    torch.cuda.synchronize()
    for nstreams, micro, side in [(1, 256, False), (1, 256, True), (1, 128, False), (1, 64, False),
                                  (2, 256, False), (2, 256, True), (2, 128, False), (2, 128, True),
                                  (3, 256, True), (4, 128, True), (4, 64, True), (2, 64, True)]:
        ms, host, out = run_variant(nstreams, micro, side)
        print(f'{nstreams} stream(s) x micro-batch {micro:3d}, shortcut on side stream {side!s:5}: {ms:.3f} ms per 256 images '
              f'({256 / ms * 1e3:.0f} images/s), host issue {host:.3f} ms; logits as on one stream: {torch.equal(out, ref)}', flush=True)
