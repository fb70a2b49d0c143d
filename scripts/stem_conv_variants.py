"""Developer tool: MIOpen variants of the 7x7 stride-2 stem convolution (fp32)."""
import torch, torch.nn.functional as F

def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

x = torch.randn(256, 3, 224, 224, device='cuda')
w = torch.randn(64, 3, 7, 7, device='cuda') * 0.05
for bench in (False, True):
    torch.backends.cudnn.benchmark = bench
    with torch.no_grad():
        xc, wc = x.contiguous(memory_format=torch.channels_last), w.contiguous(memory_format=torch.channels_last)
        print(f'benchmark={bench}: nchw {t(lambda: F.conv2d(x, w, None, 2, 3)):7.1f}  nhwc {t(lambda: F.conv2d(xc, wc, None, 2, 3)):7.1f}', end='')
        x4 = F.pad(x, (0, 0, 0, 0, 0, 1)).contiguous(memory_format=torch.channels_last)
        w4 = F.pad(w, (0, 0, 0, 0, 0, 1)).contiguous(memory_format=torch.channels_last)
        print(f'  nhwc C=4 {t(lambda: F.conv2d(x4, w4, None, 2, 3)):7.1f}', end='')
        x8 = F.pad(x, (0, 0, 0, 0, 0, 5)).contiguous(memory_format=torch.channels_last)
        w8 = F.pad(w, (0, 0, 0, 0, 0, 5)).contiguous(memory_format=torch.channels_last)
        print(f'  nhwc C=8 {t(lambda: F.conv2d(x8, w8, None, 2, 3)):7.1f}', end='')
        # space-to-depth: stride-2 7x7 on 3 channels == stride-1 4x4 on 12 channels (zero taps where k > 6)
        xs = F.pixel_unshuffle(F.pad(x, (3, 3, 3, 3)), 2)                   # [N, 12, 115, 115]
        w8x8 = F.pad(w, (0, 1, 0, 1))                                        # [64, 3, 8, 8]
        ws = w8x8.view(64, 3, 4, 2, 4, 2).permute(0, 1, 3, 5, 2, 4).reshape(64, 12, 4, 4).contiguous()
        ref = F.conv2d(x, w, None, 2, 3)
        got = F.conv2d(xs, ws, None, 1, 0)
        print(f'  s2d nchw {t(lambda: F.conv2d(xs, ws, None, 1, 0)):7.1f} (maxdiff {(got[..., :112, :112] - ref).abs().max().item():.1e}, shape {tuple(got.shape)})', end='')
        xsc, wsc = xs.contiguous(memory_format=torch.channels_last), ws.contiguous(memory_format=torch.channels_last)
        print(f'  s2d nhwc {t(lambda: F.conv2d(xsc, wsc, None, 1, 0)):7.1f}   unshuffle {t(lambda: F.pixel_unshuffle(F.pad(x, (3, 3, 3, 3)), 2)):6.1f}')
