"""Chained 1-bit layers (quant.binary.chain) against one quantizer launch per layer: ms per forward under graph replay and
the kernel time per entry point, for the CIFAR network (batch 100) and the ImageNet ls-1 / ls-1 network (batch 256), at
several values of chain.MAX_ELEMENTS.  python scripts/ls1_chain.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench  # noqa: E402

from quant import _hip  # noqa: E402
from quant.binary import chain  # noqa: E402
from quant.common.graph_replay import GraphedForward  # noqa: E402


def run(name, arch, shape, settings):
    dev = torch.device('cuda:0')
    model = bench.build_model(arch, dev)
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(1)).to(dev)
    base = None
    for on, px in settings:
        chain.ENABLED, chain.MAX_ELEMENTS = on, px
        with torch.no_grad():
            for _ in range(3):
                y = model(x)
            _hip.enable_timing(True)
            model(x)
            torch.cuda.synchronize()
            kern = {k: round(v[1], 3) for k, v in _hip.drain_timing().items()}
            _hip.enable_timing(False)
            fwd = GraphedForward(model, x)
            for _ in range(5):
                fwd.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                fwd.replay()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 50 * 1e3
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                model(x)
            torch.cuda.synchronize()
            eager = (time.perf_counter() - t0) / 20 * 1e3
        base = y.clone() if base is None else base
        print(f'{name} chain={on} max_elements={px}: graph {ms:.3f} ms ({shape[0] / ms * 1e3:.0f} img/s), eager {eager:.3f} ms, '
              f'equal {torch.equal(y, base)}, kernel ms {kern}', flush=True)


if __name__ == '__main__':
    run('cifar_b100', bench.cifar_arch(), (100, 3, 32, 32), [(False, 0), (True, 1 << 23)])
    run('imagenet_ls1_b256', bench.imagenet_arch('ls-1', 2), (256, 3, 224, 224),
        [(False, 0), (True, 1 << 23), (True, 1 << 24), (True, 1 << 25), (True, 1 << 30)])
