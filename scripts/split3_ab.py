"""A/B of the three-stream row layout on the headline network: one-stream forward time and the per-kernel table with
quant.binary.layouts.ENABLED on and off (alternating runs on one box)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd')]
import torch
import bench
from quant import _hip
from quant.binary import layouts

dev = torch.device('cuda:0')
model = bench.build_model(bench.imagenet_arch('ls-2', 3), dev)
x = torch.randn(256, 3, 224, 224, generator=torch.Generator().manual_seed(0)).to(dev)

def run(on, steps=100):
    layouts.ENABLED = on
    with torch.no_grad():
        for _ in range(10):
            model(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            model(x)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        _hip.enable_timing(True)
        for _ in range(3):
            model(x)
        torch.cuda.synchronize()
        tab = _hip.drain_timing(by_tag=True)
        _hip.enable_timing(False)
    rows = {f'{k[0]}:{k[1]}': round(1e3 * v[1] / v[0], 1) for k, v in sorted(tab.items(), key=lambda kv: str(kv[0]))}
    tot = {}
    for k, v in tab.items():
        tot[k[0]] = tot.get(k[0], 0.0) + v[1] / 3
    return ms, rows, {k: round(v, 4) for k, v in tot.items()}

out = []
for rep in range(3):
    for on in (False, True):
        ms, rows, tot = run(on)
        out.append({'split3': on, 'ms_per_step': round(ms, 4), 'kernels_ms': tot, 'avg_launch_us': rows})
        print(json.dumps(out[-1]))
