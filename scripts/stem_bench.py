"""Developer tool: time the fp32 stem (7x7 s2 conv -> maxpool -> bias -> relu) variants on the GPU."""
import torch, torch.nn.functional as F

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

torch.backends.cudnn.benchmark = True
x = torch.randn(256, 3, 224, 224, device='cuda')
w = torch.randn(64, 3, 7, 7, device='cuda') * 0.05
b = torch.randn(64, device='cuda')
with torch.no_grad():
    y = F.conv2d(x, w, None, 2, 3)
    print('conv nchw            %8.1f us' % t(lambda: F.conv2d(x, w, None, 2, 3)))
    xc, wc = x.contiguous(memory_format=torch.channels_last), w.contiguous(memory_format=torch.channels_last)
    print('conv nhwc            %8.1f us' % t(lambda: F.conv2d(xc, wc, None, 2, 3)))
    print('pool nchw            %8.1f us' % t(lambda: F.max_pool2d(y, 3, 2, 1)))
    yc = y.contiguous(memory_format=torch.channels_last)
    print('pool nhwc            %8.1f us' % t(lambda: F.max_pool2d(yc, 3, 2, 1)))
    p = F.max_pool2d(y, 3, 2, 1)
    print('bias+relu on pooled  %8.1f us' % t(lambda: p.add(b.view(1, -1, 1, 1)).relu_()))
    print('unfold-free pool via 2x separable max %8.1f us' % t(lambda: F.max_pool2d(F.max_pool2d(y, (1, 3), (1, 2), (0, 1)), (3, 1), (2, 1), (1, 0))))
    xh = x.half(); wh = w.half()
    print('conv fp16 (ref only) %8.1f us' % t(lambda: F.conv2d(xh, wh, None, 2, 3)))

with torch.no_grad():
    bv = b.view(1, -1, 1, 1)
    def full_nchw():
        return F.max_pool2d(F.conv2d(x, w, None, 2, 3), 3, 2, 1).add_(bv).relu_()
    def full_nhwc_a():
        p = F.max_pool2d(F.conv2d(x.contiguous(memory_format=torch.channels_last), wc, None, 2, 3), 3, 2, 1)
        return p.add_(bv).relu_().contiguous()
    def full_nhwc_b():
        p = F.max_pool2d(F.conv2d(x.contiguous(memory_format=torch.channels_last), wc, None, 2, 3), 3, 2, 1)
        out = torch.empty(p.shape, device=p.device)
        torch.add(p, bv, out=out)
        return out.relu_()
    def full_nhwc_c():   # NCHW input straight into an NHWC-weight conv (MIOpen picks the layout)
        p = F.max_pool2d(F.conv2d(x, wc, None, 2, 3), 3, 2, 1)
        return p.add_(bv).relu_().contiguous()
    r = full_nchw()
    for name, fn in [('full nchw', full_nchw), ('full nhwc a', full_nhwc_a), ('full nhwc b', full_nhwc_b), ('full nhwc c', full_nhwc_c)]:
        o = fn()
        print('%-14s %8.1f us  contiguous=%s maxdiff=%.2e' % (name, t(fn), o.is_contiguous(), (o - r).abs().max().item()))
