"""Where the HOST time of an eager forward goes (launch-bound configurations: CIFAR ResNet-18 at batch 100 runs ~35 launches
of a few microseconds each).  python scripts/host_profile.py [cifar|imagenet]  ->  cProfile table, sorted by own time."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench  # noqa: E402



def main():
    which = sys.argv[1] if len(sys.argv) > 1 else 'cifar'
    dev = torch.device('cuda:0')
    if which == 'cifar':
        model, shape = bench.build_model(bench.cifar_arch(), dev), (100, 3, 32, 32)
    else:
        model, shape = bench.build_model(bench.imagenet_arch('ls-2', 1), dev), (8, 3, 224, 224)
    x = torch.randn(*shape, device=dev)
    with torch.no_grad():
        for _ in range(5):
            model(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            model(x)
        torch.cuda.synchronize()
        print(f'{which}: {(time.perf_counter() - t0) * 10:.3f} ms per eager forward')
        prof = cProfile.Profile()
        prof.enable()
        for _ in range(100):
            model(x)
        torch.cuda.synchronize()
        prof.disable()
    pstats.Stats(prof).sort_stats('tottime').print_stats(35)


if __name__ == '__main__':
    main()
