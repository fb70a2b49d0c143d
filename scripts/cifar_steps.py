#!/usr/bin/env python3
"""Developer tool: N eager eval forwards of the CIFAR-100 ls-1 configuration (batch 100), e.g. under
`rocprofv3 --kernel-trace --stats` (scripts/trace_summary.py then gives the kernel time per step)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd')]
import torch  # noqa: E402
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
model = bench.build_model(bench.cifar_arch(), 'cuda:0')
x = torch.randn(100, 3, 32, 32, device='cuda:0')
with torch.no_grad():
    for _ in range(steps):
        model(x)
torch.cuda.synchronize()
