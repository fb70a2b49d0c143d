#!/usr/bin/env python3
"""Developer tool: CONSECUTIVE full batches of the headline forward on two (three) HIP streams -- the dispatch ramp and the
last round of tiles of one forward's kernels under the other forward's kernels -- against one stream.
python scripts/two_batches.py [ls-2|fp|ls-1]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd')]
import torch  # noqa: E402
import bench  # noqa: E402

act = sys.argv[1] if len(sys.argv) > 1 else 'ls-2'
dev = torch.device('cuda', 0)
model = bench.build_model(bench.imagenet_arch(act, 3 if act == 'ls-2' else 2), dev)
xs = [torch.randn(256, 3, 224, 224, device=dev) for _ in range(3)]
if os.environ.get('LSQ_EXP_NO_FC'):                 # (experiment: the forward ends with the pooling kernel)
    if os.environ['LSQ_EXP_NO_FC'] == 'all':
        model.linear_classifier = torch.nn.Identity()
    else:
        model.linear_classifier[2] = torch.nn.Identity()
if os.environ.get('LSQ_EXP_FUSED_MODE'):           # (experiment: lsq_debug_fused_mode, e.g. 4 = the round-2 solve, kernels without a stack)
    from quant import _hip
    _hip.lib().lsq_debug_fused_mode(int(os.environ['LSQ_EXP_FUSED_MODE']))
if os.environ.get('LSQ_EXP_DUMMY_FIRST'):          # (experiment: a one-workgroup kernel in front of every forward)
    _fwd, _d = model.forward, torch.zeros(64, device=dev)
    model.forward = lambda x: (_d.add_(1.0), _fwd(x))[1]


def timed(nstreams, steps=60):
    streams = [torch.cuda.Stream() for _ in range(nstreams)] if nstreams > 1 else [torch.cuda.current_stream()]
    outs = [None] * nstreams

    def run(n):
        cur = torch.cuda.current_stream()
        if nstreams > 1:
            for s in streams:
                s.wait_stream(cur)
        for i in range(n):
            k = i % nstreams
            with torch.cuda.stream(streams[k]):
                outs[k] = model(xs[k])
        if nstreams > 1:
            for s in streams:
                cur.wait_stream(s)

    run(2 * nstreams)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, [o.clone() for o in outs]


with torch.no_grad():
    ref = [model(x).clone() for x in xs]
    for n in ([int(v) for v in sys.argv[2:]] or (1, 2, 3)):
        ms, outs = timed(n)
        same = all(torch.equal(o, r) for o, r in zip(outs, ref))
        print(f'{act}: {n} stream(s): {ms:.3f} ms per batch of 256 ({256 / ms * 1e3:.0f} images/s); logits as on one stream: {same}')
