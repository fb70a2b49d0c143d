#!/usr/bin/env python3
"""Per-kernel micro-benchmark over the 16 ResNet-18 QuantConv2d layer shapes.

    python scripts/kbench.py [--batch 256] [--scheme ls-2] [--iters 20]

Prints, per distinct layer shape, the launch time of lsq_act_quant and lsq_xnor_conv2d measured
with HIP events on the launch stream, the algorithmic GB/s of the quantizer and the binary
GMAC/s of the convolution.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd')]
import torch  # noqa: E402
from quant import _hip  # noqa: E402

SHAPES = [  # (C, H, O, stride, count in ResNet-18)
    (64, 56, 64, 1, 4), (64, 56, 128, 2, 1), (128, 28, 128, 1, 3), (128, 28, 256, 2, 1),
    (256, 14, 256, 1, 3), (256, 14, 512, 2, 1), (512, 7, 512, 1, 3)]


PRE = None       # --evict: run in front of every timed launch, outside the events (cold L2 / Infinity Cache, as inside a network)


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if PRE is not None:
            PRE()
        s.record()
        fn()
        e.record()
        evs.append((s, e))
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in evs)
    return ts[len(ts) // 2] * 1e3   # median, us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--scheme', default='ls-2')
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--dist', default='gauss', choices=['gauss', 'relu', 'relu-bn'],
                    help='input distribution: N(0,1); relu(N(0,1)); per-channel affine of relu (what a BN after a ReLU feeds)')
    ap.add_argument('--fold', action='store_true', help='fold a per-channel scale / shift into the quantizer read (as the network does)')
    ap.add_argument('--xnor-popcount', action='store_true', help='every XNOR convolution through the popcount kernel')
    ap.add_argument('--streaming', action='store_true', help='quantizer through the streaming sweeps (test hook lsq_debug_force_streaming)')
    ap.add_argument('--forced', action='store_true', help='quantizer with the caller\'s scales (moving-average inference): no solve')
    ap.add_argument('--evict', type=int, default=0, help='MB streamed through the caches in front of every timed launch (a kernel '
                    'inside a network finds its code and tables neither in L2 nor in the instruction caches)')
    ap.add_argument('--prewarm-code', action='store_true', help='with --evict: the same kernel on 8 samples right in front of the timed launch (its code, tables and weights back in every L2)')
    ap.add_argument('--retouch-weights', action='store_true', help='with --evict --retouch: also the packed weights, tap sums and scales of the convolution')
    ap.add_argument('--retouch', action='store_true', help='with --evict: read the quantizer input / the planes once more after the '
                    'eviction (the producer inside a network has just written them)')
    args = ap.parse_args()
    _hip.xnor_impl(args.xnor_popcount)
    dev = 'cuda:0'
    global PRE
    touch = []
    warm = []
    if args.evict:
        junk = torch.empty(args.evict << 18, device=dev)

        def PRE():
            junk.add_(1.0)
            if args.prewarm_code and warm:
                warm[0]()
            if args.retouch:
                for t in touch:
                    (t.view(-1)[:t.numel() // 4 * 4].view(-1, 4).sum() if t.numel() >= 4096 else t.sum())
    sch = {'ls-1': (1, 1), 'ls-2': (2, 2), 'ls-T': (3, 2), 'gf-2': (4, 2)}[args.scheme]
    n = args.batch
    tot_q = tot_c = tot_f = 0.0
    for c, h, o, stride, count in SHAPES:
        x = torch.randn(n, c, h, h, device=dev)
        if args.dist != 'gauss':
            x = x.clamp(min=0)
        if args.dist == 'relu-bn':
            x = x * (0.5 + torch.rand(1, c, 1, 1, device=dev)) * 1.7 + torch.randn(1, c, 1, 1, device=dev) * 0.5 - 0.7
        w = torch.randn(o, c, 3, 3, device=dev)
        g = _hip.make_geom(n, c, h, h, o, 3, 3, (stride, stride), (1, 1), (1, 1), 1)
        k = sch[1]
        planes = torch.zeros(k * _hip.act_plane_words(g), dtype=torch.int64, device=dev)
        scales = torch.empty((k, n), device=dev)
        wsc = w.abs().mean(dim=(1, 2, 3)).view(1, -1).contiguous()
        wbits, wsum = _hip.pack_weight(w, g, wsc)
        ho, wo = _hip.out_hw(g)
        y = torch.empty((n, o, ho, wo), device=dev)
        bias = torch.zeros(o, device=dev)
        pre = None
        if args.fold:
            pre = ((0.5 + torch.rand(c, device=dev)).contiguous(), (torch.randn(c, device=dev) * 0.3).contiguous())
        forced = (torch.rand((k, n), device=dev) + 0.5).contiguous() if args.forced else None
        g8 = _hip.make_geom(8, c, h, h, o, 3, 3, (stride, stride), (1, 1), (1, 1), 1)
        planes8 = torch.zeros(k * _hip.act_plane_words(g8), dtype=torch.int64, device=dev)
        scales8, y8, x8 = torch.empty((k, 8), device=dev), torch.empty((8, o, ho, wo), device=dev), x[:8].clone()
        touch[:] = [x]
        warm[:] = [lambda: _hip.act_quant(x8, g8, sch[0], k, 3, 3.0, planes8, scales8, pre=pre, forced=None if forced is None else forced[:, :8].contiguous())]
        with _hip.debug_switches(force_streaming=args.streaming):
            tq = timeit(lambda: _hip.act_quant(x, g, sch[0], k, 3, 3.0, planes, scales, pre=pre, forced=forced), args.iters)
        touch[:] = [planes] + ([wbits.view(torch.float32) if wbits.dtype != torch.float32 else wbits, wsum.view(torch.float32), wsc, bias] if args.retouch_weights else [])
        warm[:] = [lambda: _hip.xnor_conv2d(planes8, k, scales8, wbits, wsum, wsc, bias, g8, y8)]
        tc = timeit(lambda: _hip.xnor_conv2d(planes, k, scales, wbits, wsum, wsc, bias, g, y), args.iters)
        warm[:] = []
        wprep = _hip.signw_prepare_weight(wbits, 1, g)      # (the prepared weight image: the fast path, as the modules use it)
        tf = timeit(lambda: _hip.signw_conv2d(x, 2.0, wbits, wsc, bias, g, y, wprep=wprep), args.iters)
        m = c * h * h
        qbytes = n * (4 * m + k * m // 8)
        macs = n * o * ho * wo * c * 9 * k
        print(f'C={c:4d} H={h:3d} O={o:4d} s={stride}  act_quant {tq:8.1f} us  {qbytes / tq / 1e3:7.1f} GB/s alg | '
              f'xnor_conv {tc:8.1f} us  {macs / tc / 1e6:9.1f} T binary-MAC/s | '
              f'signw(mfma) {tf:8.1f} us {2 * 2 * macs / k / tf / 1e6:7.1f} TFLOP/s bf16 (2 passes)   (x{count})')
        tot_f += tf * count
        tot_q += tq * count
        tot_c += tc * count
    print(f'per forward (16 layers, batch {n}): act_quant {tot_q / 1e3:.2f} ms, xnor_conv {tot_c / 1e3:.2f} ms '
          f'=> path-only {n / ((tot_q + tot_c) * 1e-6):.0f} images/s;  fp-act signw conv {tot_f / 1e3:.2f} ms')


if __name__ == '__main__':
    main()
