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

# ---- 1x1 stride-2 projection shortcuts (fp32 conv + folded BN bias): MIOpen vs gather + batched GEMM
with torch.no_grad():
    for c, h, o in ((64, 56, 128), (128, 28, 256), (256, 14, 512)):
        xs = torch.randn(256, c, h, h, device='cuda')
        ws = torch.randn(o, c, 1, 1, device='cuda') * c ** -0.5
        bs = torch.randn(o, device='cuda')
        ref = F.conv2d(xs, ws, bs, 2)
        t_ref = t(lambda: F.conv2d(xs, ws, bs, 2))
        n, ho = xs.shape[0], (h + 1) // 2
        buf = torch.ones(n, c + 1, ho * ho, device='cuda')
        waug = torch.cat([ws.view(o, c), bs.view(o, 1)], 1).contiguous()
        def bmm_path():
            buf[:, :c].view(n, c, ho, ho).copy_(xs[:, :, ::2, ::2])
            return torch.matmul(waug, buf).view(n, o, ho, ho)
        def bmm_path2():
            buf[:, :c].view(n, c, ho, ho).copy_(xs[:, :, ::2, ::2])
            return torch.bmm(waug.unsqueeze(0).expand(n, o, c + 1), buf).view(n, o, ho, ho)
        got = bmm_path()
        print('shortcut C=%3d H=%2d O=%3d  miopen %7.1f us   gather+matmul %7.1f us  gather+bmm %7.1f us  gather only %7.1f us  maxdiff %.2e' % (
            c, h, o, t_ref, t(bmm_path), t(bmm_path2),
            t(lambda: buf[:, :c].view(n, c, ho, ho).copy_(xs[:, :, ::2, ::2])), (got - ref).abs().max().item()))
