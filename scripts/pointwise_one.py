#!/usr/bin/env python3
"""Developer tool: time lsq_pointwise_conv on the three projection shapes of ResNet-18 (batch 256)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd')]
import torch  # noqa: E402
from quant import _hip  # noqa: E402


def main():
    for n, c, h, o in [(256, 64, 56, 128), (256, 128, 28, 256), (256, 256, 14, 512),
                       (100, 64, 32, 128), (100, 128, 16, 256), (100, 256, 8, 512)]:      # ImageNet batch 256, CIFAR batch 100
        x = torch.randn(n, c, h, h, device='cuda')
        w = torch.randn(o, c, device='cuda') * c ** -0.5
        b = torch.randn(o, device='cuda')
        for _ in range(3):
            y = _hip.pointwise_conv(x, w, b, 2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            y = _hip.pointwise_conv(x, w, b, 2)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / 20
        flop = 2.0 * y.numel() * c
        ref = torch.nn.functional.conv2d(x.double(), w.double().view(o, c, 1, 1), b.double(), stride=2)
        err = float((y.double() - ref).abs().max() / ref.abs().max())
        print(f'N={n} C={c:4d} H={h:3d} O={o:4d}: max rel err {err:.1e} {us:7.1f} us  {flop / us / 1e6:6.1f} TFLOP/s fp32  '
              f'{(y.numel() * 4 + x.numel() * 2) / us / 1e3:7.1f} GB/s (y written + the even rows of x)')


if __name__ == '__main__':
    main()
