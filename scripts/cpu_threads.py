#!/usr/bin/env python3
"""Developer tool: the CPU oracle's whole-network forward at several torch thread counts (bench.py's cpu_baseline picks
the faster of 16 and 32; the default of 128 on the MI355X hosts is five times slower)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd')]
import torch  # noqa: E402
import bench  # noqa: E402
from oracle import ref_models  # noqa: E402

arch = bench.imagenet_arch()
model = bench.build_model(arch, torch.device('cpu'))
sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
x = torch.randn(32, 3, 224, 224)
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    with torch.no_grad():
        ref_models.resnet_forward(sd, arch, x[:2])
        t0 = time.perf_counter()
        ref_models.resnet_forward(sd, arch, x)
        dt = time.perf_counter() - t0
    print(nt, 'threads:', round(32 / dt, 2), 'images/s')
