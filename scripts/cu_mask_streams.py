#!/usr/bin/env python3
"""Developer tool: consecutive batches on two HIP streams that own DISJOINT halves of the chip (hipExtStreamCreateWithCUMask)
instead of sharing all of it: the HBM-bound 56 x 56 layers of one forward next to the compute- and latency-bound layers of the
other.  python scripts/cu_mask_streams.py"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd')]
import torch  # noqa: E402
import bench  # noqa: E402

hip = ctypes.CDLL('libamdhip64.so')
dev = torch.device('cuda', 0)
model = bench.build_model(bench.imagenet_arch('ls-2', 3), dev)
xs = [torch.randn(256, 3, 224, 224, device=dev) for _ in range(2)]


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[sum(1 << b for b in range(32) if (32 * w + b) in bits) for w in range(8)])
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)


def timed(streams, steps=60):
    outs = [None] * len(streams)

    def run(n):
        cur = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(cur)
        for i in range(n):
            k = i % len(streams)
            with torch.cuda.stream(streams[k]):
                outs[k] = model(xs[k])
        for s in streams:
            cur.wait_stream(s)

    run(2 * len(streams))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, [o.clone() for o in outs]


LAYOUTS = {
    'bits 0-127 | 128-255': (set(range(128)), set(range(128, 256))),
    'even bits | odd bits': (set(range(0, 256, 2)), set(range(1, 256, 2))),
    'bits with (b // 8) even | odd': ({b for b in range(256) if (b // 8) % 2 == 0}, {b for b in range(256) if (b // 8) % 2 == 1}),
    'bits with (b // 32) even | odd': ({b for b in range(256) if (b // 32) % 2 == 0}, {b for b in range(256) if (b // 32) % 2 == 1}),
}
with torch.no_grad():
    ref = [model(x).clone() for x in xs]
    ms, _ = timed([torch.cuda.Stream(), torch.cuda.Stream()])
    print(f'two plain streams: {ms:.3f} ms per batch ({256 / ms * 1e3:.0f} images/s)', flush=True)
    full = masked_stream(set(range(256)))
    ms, _ = timed([full])
    print(f'one stream, all 256 CUs by mask: {ms:.3f} ms per batch', flush=True)
    for name, (ma, mb) in LAYOUTS.items():
        sa, sb = masked_stream(ma), masked_stream(mb)
        one, _ = timed([sa], steps=30)
        ms, outs = timed([sa, sb])
        same = all(torch.equal(o, r) for o, r in zip(outs, ref))
        print(f'{name}: one half alone {one:.3f} ms per batch; two halves {ms:.3f} ms per batch ({256 / ms * 1e3:.0f} images/s); '
              f'logits as on one stream: {same}', flush=True)
