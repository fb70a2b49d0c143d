#!/usr/bin/env python3
"""Developer tool: lsq_signw_conv2d on the seven ResNet-18 layer shapes, fast path (prepared weights) against the
general kernels: bitwise comparison of the outputs (same arithmetic, same order) and launch times.

    python scripts/signw_layers.py [--batch 256] [--iters 20] [--fused]
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


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
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
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--fused', action='store_true', help='with folded batch norm, PReLU and the residual of the block')
    ap.add_argument('--skip-general', action='store_true')
    args = ap.parse_args()
    dev, n = 'cuda:0', args.batch
    tot_fast = tot_gen = 0.0
    for c, h, o, stride, count in SHAPES:
        torch.manual_seed(c + h)
        x = torch.randn(n, c, h, h, device=dev) * 1.3
        w = torch.randn(o, c, 3, 3, device=dev)
        g = _hip.make_geom(n, c, h, h, o, 3, 3, (stride, stride), (1, 1), (1, 1), 1)
        wsc = w.abs().mean(dim=(1, 2, 3)).view(1, -1).contiguous()
        wbits, _ = _hip.pack_weight(w, g, wsc)
        wprep = _hip.signw_prepare_weight(wbits, 1, g)
        ho, wo = _hip.out_hw(g)
        bias = torch.randn(o, device=dev)
        kw = {}
        if args.fused:
            kw = dict(pre=(torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1),
                      prelu=torch.full((1,), 0.25, device=dev),
                      res_post=torch.randn(n, o, ho, wo, device=dev))
        y_fast = torch.empty((n, o, ho, wo), device=dev)
        y_gen = torch.empty((n, o, ho, wo), device=dev)
        _hip.signw_conv2d(x, 2.0, wbits, wsc, bias, g, y_fast, wprep=wprep, **kw)
        _hip.signw_conv2d(x, 2.0, wbits, wsc, bias, g, y_gen, **kw)
        torch.cuda.synchronize()
        same = torch.equal(y_fast, y_gen)
        err = float((y_fast - y_gen).abs().max() / y_gen.abs().max())
        tf = timeit(lambda: _hip.signw_conv2d(x, 2.0, wbits, wsc, bias, g, y_fast, wprep=wprep, **kw), args.iters)
        tg = 0.0 if args.skip_general else timeit(lambda: _hip.signw_conv2d(x, 2.0, wbits, wsc, bias, g, y_gen, **kw), args.iters)
        flops = 2 * 2 * y_fast.numel() * c * 9
        tot_fast += tf * count
        tot_gen += tg * count
        print(f'C={c:4d} H={h:3d} O={o:4d} s={stride}  fast {tf:7.1f} us {flops / tf / 1e6:7.1f} TF | general {tg:7.1f} us '
              f'| bitwise equal {same} (max rel diff {err:.1e})   (x{count})', flush=True)
    print(f'per forward (16 layers, batch {n}): fast {tot_fast / 1e3:.3f} ms, general {tot_gen / 1e3:.3f} ms', flush=True)


if __name__ == '__main__':
    main()
