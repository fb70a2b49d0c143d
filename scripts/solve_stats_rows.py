"""Developer tool: solve statistics (flagged bins, gathered keys, block-path slots) for synthetic relu -> affine rows
(per-channel constants with high multiplicity), the data that stresses the wave-level path."""
import os, sys
ROOT='/root/repo'
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd')]
import numpy as np, torch
from quant import _hip
dev='cuda:0'; n=64
WS_ROW = _hip.lib().lsq_solver_workspace_bytes(1)
for c,h in ((64,56),(128,28),(256,14),(512,7)):
    x = torch.randn(n, c, h, h, device=dev).clamp(min=0)
    x = x * (0.5 + torch.rand(1, c, 1, 1, device=dev)) * 1.7 + torch.randn(1, c, 1, 1, device=dev) * 0.5 - 0.7
    g = _hip.make_geom(n, c, h, h, c, 3, 3, (1,1),(1,1),(1,1),1)
    planes = torch.zeros(2*_hip.act_plane_words(g), dtype=torch.int64, device=dev); scales=torch.empty((2,n),device=dev)
    _hip.act_quant(x, g, 2, 2, 3, 3.0, planes, scales)
    torch.cuda.synchronize()
    ws = _hip.solver_workspace(n, dev)[:n*WS_ROW].cpu().numpy().reshape(n, WS_ROW)
    hdr = ws[:, :32].copy().view(np.uint32).reshape(n, 8); gathered = ws[:, 24:32].copy().view(np.float64).reshape(n)
    tflag, pad = hdr[:,0], hdr[:,3]
    print(c,h,'flagged mean %.1f max %d'%(tflag.mean(), tflag.max()), 'gathered mean %.0f max %.0f'%(gathered.mean(), gathered.max()), 'slow mean %.2f max %d'%((pad&0xFFFF).mean(), (pad&0xFFFF).max()), 'rowpass max', (pad>>16).max())
