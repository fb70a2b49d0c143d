import sys
sys.path[:0]=["tests","tests/golden","ml-quant_amd","."]
import numpy as np, torch
import test_gpu_parity as T
hip = T._hip()
arr, alpha = T._fused_cases()['ties7']
x = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32))
res = {}
for name, force in (('fused', 0), ('streaming', 1)):
    with hip.debug_switches(force_streaming=force, fused_mode=0):
        res[name] = T.run_act_quant(x, 2, 2, alpha, 1, (1, 1))
pf, sf = res['fused']; ps, ss = res['streaming']
print('v1 equal', np.array_equal(sf[0].numpy(), ss[0].numpy()), sf[0].tolist(), 'v2', sf[1].tolist(), ss[1].tolist())
for q in range(2):
    d = pf[q]^ps[q]
    nz = np.argwhere(d!=0)
    print('plane', q, 'words differing', len(nz))
    for (n,g,h,w) in nz[:6]:
        dd=int(d[n,g,h,w]); bits=[b for b in range(64) if (dd>>b)&1]
        print('  n',n,'g',g,'h',h-1,'w',w-1,'bits',bits[:8], 'x', [float(x[n,64*g+b,h-1,w-1]) for b in bits[:8]], 'fused bits', [(int(pf[q][n,g,h,w])>>b)&1 for b in bits[:8]])
