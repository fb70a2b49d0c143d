#!/usr/bin/env python3
"""Free-running parity at BASELINE sizes (SURVEY section 7 hard part 1).

The reference picks v1 with an fp32 argmin over near-tied candidates; the GPU solves the same problem in exact
arithmetic.  This measures how often and how far the two differ at the four ResNet-18 row lengths, and what the
difference does downstream:

  * rows with GPU v1 == the reference's v1 (oracle/ref_port.py, bit-exact to the reference), max |dv1| / v1;
  * plane-2 bits that flip between the two v1's (fraction of the row);
  * conv output: HIP path free running vs HIP path with the reference's scales injected,
    max |y - y_inj| / max |y_inj| (the injected run is the one held to 1e-4 against the reference).

    python scripts/free_running_parity.py [--rows 16] > profiles/rNN_free_running_rows.txt
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd'), os.path.join(ROOT, 'tests', 'golden')]
import numpy as np  # noqa: E402
import torch  # noqa: E402
from oracle import lsq_exact as E  # noqa: E402
from oracle import ref_port as P  # noqa: E402
from quant.binary.binary_conv import QuantConv2d  # noqa: E402

SHAPES = [(64, 56, 64, 1), (64, 56, 128, 2), (128, 28, 128, 1), (128, 28, 256, 2), (256, 14, 256, 1), (256, 14, 512, 2), (512, 7, 512, 1)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=16)
    args = ap.parse_args()
    dev = 'cuda:0'
    gen = torch.Generator().manual_seed(21)
    print('C   H   O  s | rows  v1 equal  max|dv1|/v1  exact==gpu | flipped plane-2 bits: max per row, fraction | conv max|dy|/max|y| (free vs injected)')
    worst = {'dv1': 0.0, 'flip': 0.0, 'conv': 0.0}
    for c, h, o, stride in SHAPES:
        n = args.rows
        # what a quantizer reads in the network: batch-normed sums of a ReLU branch and a shortcut
        x = (torch.randn(n, c, h, h, generator=gen).clamp(min=0) + 0.7 * torch.randn(n, c, h, h, generator=gen)) \
            * (0.6 + 0.8 * torch.rand(1, c, 1, 1, generator=gen)) + 0.3 * torch.randn(1, c, 1, 1, generator=gen)
        xc = x.clamp(-3, 3)
        rows = xc.reshape(n, -1)
        ref_v1 = P.solve_v1(rows, False, 3, chunk=4).view(-1)
        exact = torch.from_numpy(E.solve_rows(rows.numpy(), False, 3))
        conv = QuantConv2d('ls-2', 'ls-1', c, o, 3, {'kind': 'symmetric', 'alpha': 3}, stride=stride, padding=1, bias=True)
        with torch.no_grad():
            conv.w_approximate.v1.copy_(conv.weight.abs().mean(dim=(1, 2, 3)))
        conv = conv.eval().to(dev)
        with torch.no_grad():
            y_free = conv(x.to(dev)).cpu()
            gpu = conv.last_act_scales.cpu().clone()
            ref_v2 = P.quant_ls2(xc, ref_v1)[1]
            conv.x_approximate._forced_scales = torch.stack([ref_v1, ref_v2])
            y_inj = conv(x.to(dev)).cpu()
            conv.x_approximate._forced_scales = None
        same = int((gpu[0] == ref_v1).sum())
        dv1 = float(((gpu[0] - ref_v1).abs() / ref_v1).max())
        b_gpu = (xc - gpu[0].view(-1, 1, 1, 1) * P.pm1(xc)) >= 0
        b_ref = (xc - ref_v1.view(-1, 1, 1, 1) * P.pm1(xc)) >= 0
        flips = (b_gpu != b_ref).reshape(n, -1).sum(1)
        dconv = float((y_free - y_inj).abs().max() / y_inj.abs().max())
        worst['dv1'] = max(worst['dv1'], dv1)
        worst['flip'] = max(worst['flip'], float(flips.max()) / rows.shape[1])
        worst['conv'] = max(worst['conv'], dconv)
        print(f'{c:3d} {h:3d} {o:3d} {stride} | {n:4d}  {same:4d}/{n:<4d} {dv1:11.3e}  {bool(torch.equal(gpu[0], exact))!s:>10} | '
              f'{int(flips.max()):6d}  {float(flips.sum()) / rows.numel():.2e} | {dconv:.3e}')
    print(f'worst: max|dv1|/v1 {worst["dv1"]:.3e}, flipped fraction of a row {worst["flip"]:.3e}, conv {worst["conv"]:.3e}')


if __name__ == '__main__':
    main()
