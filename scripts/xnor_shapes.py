"""Developer tool: lsq_xnor_conv2d on the seven ResNet-18 layer shapes at batch 256 with the network's epilogue (ReLU + one
residual), median of 30 launches with the buffers rotated; LSQ_HIP_LIB selects the build, argv[1] the implementation
(lsq_debug_xnor_impl: 0 fp4 matrix-core kernel, 1 popcount kernel, 2 int8 matrix-core kernel)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd')]
import torch
from quant import _hip as hip
DEV = 'cuda:0'
IMPL = int(sys.argv[1]) if len(sys.argv) > 1 else 0
hip.xnor_impl(IMPL)
SHAPES = [(64, 56, 64, 1, 4), (64, 56, 128, 2, 1), (128, 28, 128, 1, 3), (128, 28, 256, 2, 1), (256, 14, 256, 1, 3), (256, 14, 512, 2, 1), (512, 7, 512, 1, 3)]
tot = 0.0
out = []
for c, h, o, s, cnt in SHAPES:
    n = 256
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(n, c, h, h, generator=g) * 1.2).to(DEV)
    wt = torch.randn(o, c, 3, 3, generator=g).to(DEV)
    geom = hip.make_geom(n, c, h, h, o, 3, 3, (s, s), (1, 1), (1, 1), 1)
    planes = torch.zeros((2 * hip.act_plane_words(geom),), dtype=torch.int64, device=DEV)
    scales = torch.empty((2, n), dtype=torch.float32, device=DEV)
    hip.act_quant(x, geom, hip.SCHEME_LS2, 2, 3, 3.0, planes, scales)
    wsc = wt.abs().mean(dim=(1, 2, 3)).view(1, -1).contiguous()
    wbits, wsum = hip.pack_weight(wt, geom, wsc)
    bias = torch.randn(o, generator=g).to(DEV)
    ho, wo = hip.out_hw(geom)
    NB = 3
    ys = [torch.empty((n, o, ho, wo), device=DEV) for _ in range(NB)]
    rs = [torch.randn(n, o, ho, wo, device=DEV) for _ in range(NB)]
    def run(i):
        hip.xnor_conv2d(planes, 2, scales, wbits, wsum, wsc, bias, geom, ys[i % NB], True, None, rs[i % NB], None)
    for i in range(3):
        run(i)
    torch.cuda.synchronize()
    ts = []
    for i in range(30):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(i); e1.record()
        ts.append((e0, e1))
    torch.cuda.synchronize()
    us = sorted(1e3 * a.elapsed_time(b) for a, b in ts)[15]
    tot += us * cnt
    out.append(f'C{c}_H{h}_s{s} {us:6.1f}')
print(['fp4 ', 'popc', 'int8'][IMPL], '  '.join(out), f' | 16 layers {tot / 1e3:.3f} ms')
