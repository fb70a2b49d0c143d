"""Developer tool: does the PHASE between the two streams matter?  Two streams, consecutive forwards alternating; stream 1's first
forward is held back by a GPU-side spin of `delay` ms (torch.cuda._sleep), so that in steady state its forwards run that far
behind stream 0's.  ms per 256 images over 200 forwards."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd')]
import torch
import bench
dev = torch.device('cuda:0')
model = bench.build_model(bench.imagenet_arch('ls-2', 3), dev)
x = torch.randn(256, 3, 224, 224, generator=torch.Generator().manual_seed(0)).to(dev)
s = [torch.cuda.Stream(device=dev) for _ in range(2)]
cycles_per_ms = 2.1e6        # _sleep counts shader clocks
with torch.no_grad():
    for _ in range(5):
        model(x)
    torch.cuda.synchronize()
    for delay in (0.0, 0.5, 1.0, 1.5, 2.0, 0.0, 1.0):
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.cuda.stream(s[1]):
                if delay:
                    torch.cuda._sleep(int(delay * cycles_per_ms))
            n = 200
            for i in range(n):
                with torch.cuda.stream(s[i % 2]):
                    model(x)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print(f'delay {delay:3.1f} ms: {1e3 * dt / n:.4f} ms per forward ({256 * n / dt:,.0f} images/s)')
