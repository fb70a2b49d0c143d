#!/usr/bin/env python3
"""Developer check: the single-launch quantizer (lsq_act_fused.hip) against the streaming three-kernel path
(lsq_act_quant.hip) on the same inputs -- scales and bit planes must be identical bit for bit.

    python scripts/fused_vs_streaming.py [--batch 8]
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd')]
import torch  # noqa: E402
from quant import _hip  # noqa: E402

SHAPES = [(64, 56, 56), (128, 28, 28), (256, 14, 14), (512, 7, 7), (64, 32, 32), (128, 16, 16), (256, 8, 8),
          (512, 4, 4), (20, 12, 12), (96, 10, 6), (64, 12, 20), (192, 9, 9)]


def make(dist, n, c, h, w, gen):
    x = torch.randn(n, c, h, w, generator=gen, device='cuda')
    if dist == 'gauss':
        return x
    if dist == 'relu':
        return x.clamp(min=0)
    if dist == 'relu-bn':
        x = x.clamp(min=0)
        return x * (0.5 + torch.rand(1, c, 1, 1, generator=gen, device='cuda')) * 1.7 + \
            torch.randn(1, c, 1, 1, generator=gen, device='cuda') * 0.5 - 0.7
    if dist == 'saturated':
        return x * 6
    if dist == 'const':
        return torch.full_like(x, 2.0)
    if dist == 'ints':
        return torch.randint(-4, 5, x.shape, generator=gen, device='cuda').float()
    if dist == 'heavy':
        return x * torch.exp(2 * torch.randn(x.shape, generator=gen, device='cuda'))
    if dist == 'tiny':
        return x * 1e-30
    if dist == 'wide':   # many crossing bins: log-uniform over 30 binades
        return torch.exp(torch.rand(x.shape, generator=gen, device='cuda') * 20 - 10) * torch.sign(x)
    raise ValueError(dist)


def run(x, g, scheme, alpha, pre, force, mode=0):
    with _hip.debug_switches(force_streaming=force, fused_mode=mode):
        planes = torch.zeros(2 * _hip.act_plane_words(g), dtype=torch.int64, device='cuda')
        scales = torch.zeros((2, g.N), device='cuda')
        _hip.act_quant(x, g, scheme, 2, 3, alpha, planes, scales, pre=pre)
        torch.cuda.synchronize()
    return planes, scales


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=6)
    args = ap.parse_args()
    gen = torch.Generator(device='cuda')
    gen.manual_seed(7)
    bad = 0
    for (c, h, w) in SHAPES:
        for dist in ('gauss', 'relu', 'relu-bn', 'saturated', 'const', 'ints', 'heavy', 'tiny', 'wide'):
            for scheme in (_hip.SCHEME_LS2, _hip.SCHEME_LST):
                for alpha, fold in ((3.0, False), (2.0, True), (-1.0, False)):
                    x = make(dist, args.batch, c, h, w, gen)
                    g = _hip.make_geom(args.batch, c, h, w, c, 3, 3, (1, 1), (1, 1), (1, 1), 1)
                    pre = None
                    if fold:
                        pre = ((0.5 + torch.rand(c, generator=gen, device='cuda')).contiguous(),
                               (torch.randn(c, generator=gen, device='cuda') * 0.3).contiguous())
                    ps, ss = run(x, g, scheme, alpha, pre, True)
                    for mode in (0, 1, 2):     # ordinary, every bin through the block path, 2048-key list
                        pf, sf = run(x, g, scheme, alpha, pre, False, mode)
                        if not (torch.equal(sf, ss) and torch.equal(pf, ps)):
                            bad += 1
                            dv = (sf - ss).abs().max().item()
                            nb = (pf != ps).sum().item()
                            print(f'MISMATCH mode={mode} C={c} H={h} W={w} {dist} scheme={scheme} alpha={alpha} fold={fold}: '
                                  f'max|dscale|={dv:.3e} differing plane words={nb}  fused={sf[:, :3].tolist()} stream={ss[:, :3].tolist()}')
    print('fused_vs_streaming:', 'ALL EQUAL' if bad == 0 else f'{bad} MISMATCHES')
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
