"""Developer tool: the int8-MFMA convolution with operands in the three-stream layout, every combination, one layer shape
at batch 256 in isolation (HIP events, 20 launches each, buffers rotated so that nothing stays in the Infinity Cache)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd')]
import torch
from quant import _hip as hip
from quant.binary import layouts as L

DEV = 'cuda:0'
def case(n, c, h, o, stride):
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(n, c, h, h, generator=g) * 1.2).to(DEV)
    wt = torch.randn(o, c, 3, 3, generator=g).to(DEV)
    geom = hip.make_geom(n, c, h, h, o, 3, 3, (stride, stride), (1, 1), (1, 1), 1)
    planes = torch.zeros((2 * hip.act_plane_words(geom),), dtype=torch.int64, device=DEV)
    scales = torch.empty((2, n), dtype=torch.float32, device=DEV)
    hip.act_quant(x, geom, hip.SCHEME_LS2, 2, 3, 3.0, planes, scales)
    wsc = wt.abs().mean(dim=(1, 2, 3)).view(1, -1).contiguous()
    wbits, wsum = hip.pack_weight(wt, geom, wsc)
    bias = torch.randn(o, generator=g).to(DEV)
    return geom, planes, scales, wbits, wsum, wsc, bias

for shape in [(256, 64, 56, 64, 1), (256, 64, 56, 128, 2), (256, 128, 28, 128, 1)]:
    geom, planes, scales, wbits, wsum, wsc, bias = case(*shape)
    n, c, h, o, s = shape
    ho, wo = hip.out_hw(geom)
    NB = 4                                       # rotate buffers: 4 x (y + res) > Infinity Cache at 56 x 56
    ys = [torch.empty((n, o, ho, wo), device=DEV) for _ in range(NB)]
    rs = [torch.randn(n, o, ho, wo, device=DEV) for _ in range(NB)]
    y3 = [L.info(L.empty(n, o, ho, wo, DEV)).buf for _ in range(NB)]
    r3 = [L.info(L.from_nchw(r)).buf for r in rs]
    for name, yl, rl, res in [('std  no res', 0, 0, False), ('y3   no res', 1, 0, False), ('std  res std', 0, 0, True),
                              ('y3   res std', 1, 0, True), ('std  res s3', 0, 1, True), ('y3   res s3', 1, 1, True)]:
        def run(i):
            y = (y3 if yl else ys)[i % NB]
            r = (r3 if rl else rs)[i % NB] if res else None
            hip.xnor_conv2d(planes, 2, scales, wbits, wsum, wsc, bias, geom, y, True, None, r, None, yl, rl)
        for i in range(4):
            run(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(20):
            run(i)
        e1.record()
        torch.cuda.synchronize()
        print(f'{shape} {name}: {1e3 * e0.elapsed_time(e1) / 20:7.1f} us')
