import sys
sys.path[:0] = ['/root/repo', '/root/repo/ml-quant_amd']
import torch
from quant import _hip
x = torch.randn(256, 3, 224, 224, device='cuda'); w = torch.randn(64, 3, 7, 7, device='cuda') * 0.05; b = torch.randn(64, device='cuda')
for split in (3, 2, 22):
    for _ in range(3): _hip.stem_conv_pool(x, w, b, split)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): _hip.stem_conv_pool(x, w, b, split)
    e1.record(); torch.cuda.synchronize()
    print(f'split {split}: {e0.elapsed_time(e1) * 100:.1f} us')
