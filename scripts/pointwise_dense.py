import os, sys
sys.path[:0] = ['/root/repo', '/root/repo/ml-quant_amd']
import torch
from quant import _hip as hip
DEV='cuda:0'
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev=[]
    for _ in range(n):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); ev.append((a,b))
    torch.cuda.synchronize()
    return sorted(1e3*a.elapsed_time(b) for a,b in ev)[n//2]
for c,h,o in [(64,56,128),(128,28,256),(256,14,512)]:
    x=[torch.randn(256,c,h,h,device=DEV) for _ in range(3)]
    xd=[xi[:,:,::2,::2].contiguous() for xi in x]
    w=torch.randn(o,c,device=DEV); b=torch.randn(o,device=DEV)
    i=[0]
    def strided():
        i[0]+=1; return hip.pointwise_conv(x[i[0]%3], w, b, 2)
    def dense():
        i[0]+=1; return hip.pointwise_conv(xd[i[0]%3], w, b, 1)
    y1=strided(); y2=hip.pointwise_conv(xd[i[0]%3], w, b, 1)
    print(c,h,o,'strided %.1f us  dense(stride 1 on the quarter tensor) %.1f us  equal %s' % (t(strided), t(dense), torch.equal(y1,y2)))
