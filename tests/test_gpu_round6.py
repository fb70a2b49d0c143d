"""Round-6 GPU tests: activation tensors in the three-stream row layout (LSQ_LAYOUT_SPLIT3, include/lsq_hip.h) between a
quantized convolution's epilogue and the next layer's quantizer.  The layout changes ADDRESSES only, so every check is
bit-for-bit: the convolution's three-stream output unpacked == its NCHW output, the quantizer's planes and scales from a
three-stream input == those from the NCHW input (and v1 == the exact oracle), and the whole network with the layout on ==
the network with it off (the round-5 data flow and the default, ``quant.binary.layouts.ENABLED``: the layout measured
slower end to end -- DESIGN.md section 4.9)."""

import numpy as np
import pytest
import torch

from oracle import lsq_exact as E

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _hip():
    from quant import _hip
    return _hip


def _layouts():
    from quant.binary import layouts
    return layouts


def _quantize(x, alpha, ternary=False, pre=None, split3=False, mode=0):
    hip, L = _hip(), _layouts()
    n, c, h, w = x.shape
    geom = hip.make_geom(n, c, h, w, 64, 3, 3, (1, 1), (1, 1), (1, 1), 1)
    planes = torch.zeros((2 * hip.act_plane_words(geom),), dtype=torch.int64, device=DEV)
    scales = torch.empty((2, n), dtype=torch.float32, device=DEV)
    xd = x.to(DEV)
    src, layout = xd, hip.LAYOUT_NCHW
    if split3:
        src, layout = L.info(L.from_nchw(xd)).buf, hip.LAYOUT_SPLIT3
    with hip.debug_switches(fused_mode=mode):
        hip.act_quant(src, geom, hip.SCHEME_LST if ternary else hip.SCHEME_LS2, 2, 3, alpha, planes, scales, None, pre, layout)
        torch.cuda.synchronize()
    return planes.cpu(), scales.cpu()


def test_three_stream_round_trip_and_stream_zero_is_the_sub_sample():
    """from_nchw / to_nchw are inverse, and stream 0 holds exactly x.flatten()[::3] (quantization.py:63, skip = 3)."""
    L = _layouts()
    for shape in [(3, 64, 56, 56), (2, 128, 28, 28), (5, 256, 14, 14), (2, 512, 7, 7), (1, 64, 4, 4)]:
        x = torch.randn(*shape, generator=torch.Generator().manual_seed(1)).to(DEV)
        t = L.from_nchw(x)
        rec = L.info(t)
        assert rec.S % 32 == 0 and rec.buf.shape == (shape[0], 3 * rec.S)
        assert torch.equal(L.to_nchw(t), x)
        # stream 0, pads removed, is the sub-sample in order: channel c's block holds its pixels congruent to -c modulo 3
        hp, hw = rec.S // shape[1], shape[2] * shape[3]
        s0 = rec.buf[:, :rec.S].view(shape[0], shape[1], hp)
        got = torch.cat([s0[:, c, :(hw - (-c) % 3 + 2) // 3] for c in range(shape[1])], dim=1)
        assert torch.equal(got, x.reshape(shape[0], -1)[:, ::3])
    assert L.stream_floats(64, 6, 6) == -1          # 36 % 3 == 0: no such layout


@pytest.mark.parametrize('ternary', [False, True])
def test_quantizer_reads_three_stream_rows_bit_for_bit(ternary):
    """lsq_act_quant_layout(SPLIT3) == lsq_act_quant on the same values: planes, v1, v2 bit for bit, v1 == the exact oracle --
    the four ResNet row shapes, with and without the folded batch norm, odd batch sizes, and the forced fall-back of the
    windowed solve (lsq_debug_fused_mode 8: the round-2 body re-reads the sub-sample from stream 0)."""
    rs = np.random.RandomState(6)
    for (n, c, h) in [(3, 64, 56), (5, 128, 28), (4, 256, 14), (7, 512, 7), (2, 64, 28), (2, 64, 14), (3, 128, 7)]:
        x = torch.from_numpy((rs.standard_normal((n, c, h, h)) * 1.3).astype(np.float32))
        for pre in (None, 'bn'):
            p = None
            if pre:
                sc = torch.from_numpy((0.5 + rs.random_sample(c)).astype(np.float32))
                sh = torch.from_numpy((0.3 * rs.standard_normal(c)).astype(np.float32))
                p = (sc.to(DEV), sh.to(DEV))
            want_p, want_s = _quantize(x, 3.0, ternary, p, split3=False)
            for mode in (0, 8):
                got_p, got_s = _quantize(x, 3.0, ternary, p, split3=True, mode=mode)
                assert torch.equal(got_s, want_s), (n, c, h, pre, mode, got_s, want_s)
                assert torch.equal(got_p, want_p), (n, c, h, pre, mode)
            if not pre:
                exact = E.solve_rows(x.clamp(-3, 3).reshape(n, -1).numpy(), ternary, 3)
                assert np.array_equal(want_s[0].numpy(), exact)


def test_quantizer_three_stream_adversarial_rows():
    """The round-5 adversarial rows of the windowed solve (everything below the window, a crossing on its edge, dense bins,
    saturated rows, zeros and ties) through the three-stream kernels: == the NCHW kernels == the exact oracle."""
    from test_gpu_round5 import _window_cases
    for tag, (arr, alpha) in _window_cases().items():
        x = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32))
        n, c, h, w = x.shape
        if (h * w) % 3 != 1:
            continue
        for ternary in (False, True):
            want_p, want_s = _quantize(x, alpha, ternary)
            got_p, got_s = _quantize(x, alpha, ternary, split3=True)
            assert torch.equal(got_s, want_s) and torch.equal(got_p, want_p), (tag, ternary)
            exact = E.solve_rows(x.clamp(-alpha, alpha).reshape(n, -1).numpy(), ternary, 3)
            assert np.array_equal(got_s[0].numpy(), exact), tag


def _conv_case(n, c, h, o, stride, seed):
    """Planes and scales of a random input, packed sign weights: the operands of lsq_xnor_conv2d."""
    hip = _hip()
    g = torch.Generator().manual_seed(seed)
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


@pytest.mark.parametrize('shape', [(3, 64, 56, 64, 1), (5, 64, 56, 128, 2), (4, 128, 28, 128, 1), (3, 128, 28, 256, 2),
                                   (7, 64, 14, 64, 1), (1, 128, 7, 128, 1), (2, 64, 4, 64, 1)])
def test_convolution_writes_and_reads_three_stream_tensors_bit_for_bit(shape):
    """lsq_xnor_conv2d_layout: y as a three-stream tensor, residual operands in either layout -- unpacked, the NCHW call's
    output bit for bit; all four epilogues (none, ReLU + res_post, PReLU + res_pre, both residuals)."""
    hip, L = _hip(), _layouts()
    n, c, h, o, stride = shape
    geom, planes, scales, wbits, wsum, wsc, bias = _conv_case(n, c, h, o, stride, seed=sum(shape))
    assert hip.layout_support(geom, hip.SCHEME_LS2, 2) & 6 == 6
    ho, wo = hip.out_hw(geom)
    g = torch.Generator().manual_seed(9)
    r1 = torch.randn(n, o, ho, wo, generator=g).to(DEV)
    r2 = torch.randn(n, o, ho, wo, generator=g).to(DEV)
    slope = torch.rand(o, generator=g).to(DEV)
    r1s, r2s = L.info(L.from_nchw(r1)).buf, L.info(L.from_nchw(r2)).buf
    for relu, prelu, pre, post in [(False, None, None, None), (True, None, None, r1), (False, slope, r1, None), (True, None, r1, r2)]:
        want = torch.empty((n, o, ho, wo), dtype=torch.float32, device=DEV)
        hip.xnor_conv2d(planes, 2, scales, wbits, wsum, wsc, bias, geom, want, relu, pre, post, prelu)
        for y_l in (0, 1):
            for r_l in ((0, 1) if (pre is not None or post is not None) else (0,)):
                if not y_l and not r_l:
                    continue
                y = L.empty(n, o, ho, wo, DEV) if y_l else torch.empty_like(want)
                ybuf = L.info(y).buf if y_l else y
                ybuf.fill_(float('nan'))
                sel = lambda t, ts: None if t is None else (ts if r_l else t)      # noqa: E731
                hip.xnor_conv2d(planes, 2, scales, wbits, wsum, wsc, bias, geom, ybuf, relu,
                                sel(pre, r1s), sel(post, r1s if post is r1 else r2s), prelu, y_l, r_l)
                torch.cuda.synchronize()
                got = L.to_nchw(y)
                assert torch.equal(got, want), (shape, relu, prelu is not None, y_l, r_l, float((got - want).abs().max()))


def test_three_stream_operands_outside_the_kernels_are_refused():
    """Geometries without a three-stream kernel answer LSQ_E_UNSUPPORTED (the host asks lsq_layout_support first)."""
    hip, L = _hip(), _layouts()
    geom, planes, scales, wbits, wsum, wsc, bias = _conv_case(2, 256, 14, 256, 1, seed=2)
    assert hip.layout_support(geom, hip.SCHEME_LS2, 2) & 6 == 0
    y = L.empty(2, 256, 14, 14, DEV)
    with pytest.raises(hip.LsqHipError):
        hip.xnor_conv2d(planes, 2, scales, wbits, wsum, wsc, bias, geom, L.info(y).buf, False, None, None, None, 1, 0)
    x = torch.randn(2, 64, 6, 6).to(DEV)                # 36 % 3 == 0
    g2 = hip.make_geom(2, 64, 6, 6, 64, 3, 3, (1, 1), (1, 1), (1, 1), 1)
    assert hip.layout_support(g2, hip.SCHEME_LS2, 2) & 1 == 0
    pl = torch.zeros((2 * hip.act_plane_words(g2),), dtype=torch.int64, device=DEV)
    sc = torch.empty((2, 2), dtype=torch.float32, device=DEV)
    with pytest.raises(hip.LsqHipError):
        hip.act_quant(torch.zeros((2, 3 * 64 * 32), device=DEV), g2, hip.SCHEME_LS2, 2, 3, 3.0, pl, sc, None, None, 1)
    # an ls-1 quantizer has no solve: nothing to gain, refused
    g3 = hip.make_geom(2, 64, 14, 14, 64, 3, 3, (1, 1), (1, 1), (1, 1), 1)
    with pytest.raises(hip.LsqHipError):
        hip.act_quant(L.info(L.from_nchw(torch.randn(2, 64, 14, 14).to(DEV))).buf, g3, hip.SCHEME_LS1, 1, 3, 3.0, pl, sc, None, None, 1)


@pytest.mark.parametrize('act', ['ls-2', 'ls-T'])
def test_whole_network_with_three_stream_tensors_is_bit_identical(act):
    """ResNet-18 ImageNet, batch 6 and 33 (pixel counts that are not multiples of 96): logits with the layout on ==
    logits with it off (round 5's data flow), and the layout really is in use (the 56 x 56 and 28 x 28 block outputs)."""
    import bench
    L = _layouts()
    model = bench.build_model(bench.imagenet_arch(act, 3 if act == 'ls-2' else 2), DEV)
    seen = []
    orig = L.empty

    def spy(n, c, h, w, device):
        seen.append((c, h))
        return orig(n, c, h, w, device)

    for batch in (6, 33):
        x = torch.randn(batch, 3, 224, 224, generator=torch.Generator().manual_seed(batch)).to(DEV)
        with torch.no_grad():
            assert L.ENABLED is False                    # (the default: round 5's data flow)
            want = model(x).clone()
            L.ENABLED, L.empty = True, spy
            try:
                got = model(x).clone()
            finally:
                L.ENABLED, L.empty = False, orig
        assert torch.equal(got, want), (act, batch, float((got - want).abs().max()))
    assert (64, 56) in seen and (128, 28) in seen, seen


def test_three_stream_tensor_leaving_the_fused_path_is_unpacked():
    """A block that cannot take the fused path (here: FUSE_BLOCKS switched off between two blocks) receives NCHW values."""
    import bench
    from quant.models import resnet
    L = _layouts()
    model = bench.build_model(bench.imagenet_arch('ls-2', 3), DEV)
    x = torch.randn(4, 3, 224, 224, generator=torch.Generator().manual_seed(4)).to(DEV)
    with torch.no_grad():
        h0 = model.blocks[0](x)
        plain = model.blocks[1](h0)
        L.ENABLED = True
        try:
            h1 = model.blocks[1](h0)                    # fused block: its output is a three-stream tensor
        finally:
            L.ENABLED = False
        assert L.info(h1) is not None and L.info(plain) is None and torch.equal(L.to_nchw(h1), plain)
        resnet.FUSE_BLOCKS = False
        try:
            a = model.blocks[2](h1)                     # modular path: must unpack
            b = model.blocks[2](plain)
        finally:
            resnet.FUSE_BLOCKS = True
        assert torch.equal(a, b)


# ---------------------------------------------------------------------------------------------- the path bench.py times, layer by layer
@pytest.mark.parametrize('act', ['ls-2', 'ls-T'])
def test_full_size_fused_network_every_layer(act):
    """The FUSED batch-256 forward -- what bench.py times -- checked layer by layer on three rows of the batch (round 5's
    test_full_size_whole_network does this on the module-by-module path and compared the two paths by a hand-set cosine):
    every QuantConv2d.fused_forward call is recorded in place, and for each of the 16 layers
      * the solved v1 equals the exact oracle on the very tensor the layer quantized (the eval batch norm folded as ONE fma
        per element, as the kernel's read does),
      * the layer's output is within north_star's 1e-4 of max|y| of an fp64 convolution of that quantized input with the
        GPU's own scales, the block's epilogue (residual before / after the non-linearity, ReLU or PReLU) included."""
    import bench
    from oracle import ref_port as P
    from quant.binary.binary_conv import QuantConv2d
    model = bench.build_model(bench.imagenet_arch(act, 3 if act == 'ls-2' else 2), DEV)
    x = torch.randn(256, 3, 224, 224, generator=torch.Generator().manual_seed(0)).to(DEV)
    rows = [0, 131, 255]
    seen = []
    orig = QuantConv2d.fused_forward

    def spy(self, xin, pre_bn=None, relu=False, res_pre=None, res_post=None, prelu=None, next_q=None, res_ready=None):
        out = orig(self, xin, pre_bn, relu, res_pre, res_post, prelu, next_q, res_ready)
        pick = lambda t: None if t is None else t[rows].clone()        # noqa: E731
        seen.append((self, pre_bn, relu, None if prelu is None else prelu.detach().clone(), pick(xin), pick(res_pre), pick(res_post),
                     pick(out), self.last_act_scales[:, rows].clone()))
        return out
    QuantConv2d.fused_forward = spy
    try:
        with torch.no_grad():
            model(x)
    finally:
        QuantConv2d.fused_forward = orig
    assert len(seen) == 16
    ternary = act == 'ls-T'
    for li, (conv, bn, relu, slope, xin, rpre, rpost, yout, scales) in enumerate(seen):
        alpha = conv._alpha()
        s, t = conv._folded_bn(bn)
        xb = (xin.double() * s.double().view(1, -1, 1, 1) + t.double().view(1, -1, 1, 1)).float()     # one rounding: the kernel's fma
        xc = xb.clamp(-alpha, alpha)
        want = E.solve_rows(xc.reshape(len(rows), -1).cpu().numpy(), ternary, 3)
        assert np.array_equal(scales[0].cpu().numpy(), want), (li, scales[0], want)
        xq = P.quant_lst(xc, scales[0])[1] if ternary else P.quant_ls2(xc, scales[0], scales[1])[2]
        wq = conv.w_approximate.v1.view(-1, 1, 1, 1) * P.pm1(conv.weight)
        ref = torch.nn.functional.conv2d(xq.double(), wq.double(), conv.bias.double(), conv.stride, 1)
        if rpre is not None:
            ref = ref + rpre.double()
        if relu:
            ref = ref.clamp_min(0)
        if slope is not None:
            ref = torch.where(ref > 0, ref, ref * slope.double().view(1, -1, 1, 1))
        if rpost is not None:
            ref = ref + rpost.double()
        err = float((yout.double() - ref).abs().max() / ref.abs().max())
        assert err <= 1e-4, (li, err)


def test_non_finite_rows_stay_contained_in_the_windowed_solve():
    """a row with NaN / +-Inf activations under a symmetric clamp (Inf clamps to alpha, a NaN to -alpha: the
    windowed histogram's index stays inside its 8192 bins -- now also bounded explicitly) must neither fault nor touch its
    neighbours: the finite rows of the batch come out exactly as they do alone, and the Inf row as its clamped self."""
    rs = np.random.RandomState(8)
    for (c, h) in [(64, 56), (128, 28), (512, 7)]:
        x = torch.from_numpy((rs.standard_normal((4, c, h, h)) * 1.5).astype(np.float32))
        bad = x.clone()
        bad[1].view(-1)[::7] = float('nan')
        bad[1].view(-1)[3::11] = float('inf')
        bad[2].view(-1)[5::13] = float('-inf')              # no NaN in row 2: it equals its clamped self
        for ternary in (False, True):
            p_bad, s_bad = _quantize(bad, 3.0, ternary)
            p_ok, s_ok = _quantize(x, 3.0, ternary)
            p_cl, s_cl = _quantize(bad.clamp(-3, 3).nan_to_num(0.0), 3.0, ternary)
            words = p_ok.numel() // 2 // 4
            for row in (0, 3):
                assert s_bad[0, row] == s_ok[0, row] and s_bad[1, row] == s_ok[1, row], (c, h, ternary, row)
                for plane in (0, 1):
                    a = p_bad.view(2, 4, words)[plane, row]
                    assert torch.equal(a, p_ok.view(2, 4, words)[plane, row]), (c, h, ternary, row, plane)
            assert s_bad[0, 2] == s_cl[0, 2] and torch.equal(p_bad.view(2, 4, words)[:, 2], p_cl.view(2, 4, words)[:, 2])


@pytest.mark.parametrize('cin', [64, 128, 256, 512])
def test_fp4_convolution_at_the_extremes_of_its_integer_range(cin):
    """The fp4 matrix-core kernel's accumulator holds (b * s) as an fp32 integer: at most 9 * C = 4608 in magnitude.  All-(+1) and
    all-(-1) activations against all-(+1) weights reach the extremes on interior pixels (and every border pattern's correction on
    the rim): the output is the closed form, and equal to the popcount and int8 kernels bit for bit; fp4 codes of every slot
    (0.5 / 1 / 2 / 0.5 against 4 / 2 / 1 / 4) are exercised by single-channel impulses."""
    hip = _hip()
    n, h, o = 3, 9, 64
    geom = hip.make_geom(n, cin, h, h, o, 3, 3, (1, 1), (1, 1), (1, 1), 1)
    wt = torch.ones(o, cin, 3, 3, device=DEV)
    wt[1::2] = -1.0                                           # odd out-channels: all -1
    wsc = torch.full((1, o), 0.5, device=DEV)
    wbits, wsum = hip.pack_weight(wt, geom, wsc)
    bias = torch.arange(o, dtype=torch.float32, device=DEV)
    planes = torch.zeros((2 * hip.act_plane_words(geom),), dtype=torch.int64, device=DEV)
    scales = torch.empty((2, n), dtype=torch.float32, device=DEV)
    cases = {'plus': torch.full((n, cin, h, h), 2.0), 'minus': torch.full((n, cin, h, h), -2.0)}
    imp = torch.full((n, cin, h, h), -2.0)
    for c in range(0, cin, 5):
        imp[:, c, (c // 5) % h, (c // 3) % h] = 2.0           # single +1 bits in every channel slot
    cases['impulses'] = imp
    for tag, x in cases.items():
        # planes with given scales (v1 = 1, v2 = 0.25): plane 1 = sign(x), plane 2 = sign(x - v1 b1) = sign(x) for |x| = 2
        forced = torch.tensor([[1.0] * n, [0.25] * n], device=DEV)
        hip.act_quant(x.to(DEV), geom, hip.SCHEME_LS2, 2, 3, 3.0, planes, scales, forced)
        outs = []
        for impl in (1, 0, 2):
            with hip.debug_switches(xnor_popcount=impl):
                y = torch.empty((n, o, h, h), device=DEV)
                hip.xnor_conv2d(planes, 2, scales, wbits, wsum, wsc, bias, geom, y)
                outs.append(y.clone())
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), (cin, tag)
        if tag != 'impulses':
            taps = torch.nn.functional.conv2d(torch.ones(1, 1, h, h), torch.ones(1, 1, 3, 3), padding=1).view(h, h).to(DEV)   # taps inside the image
            s1 = 1.0 if tag == 'plus' else -1.0
            sign_o = torch.where(torch.arange(o, device=DEV) % 2 == 0, 1.0, -1.0)
            want = bias.view(1, o, 1, 1) + 0.5 * sign_o.view(1, o, 1, 1) * (1.0 * s1 + 0.25 * s1) * cin * taps.view(1, 1, h, h)
            assert torch.equal(outs[1], want.expand(n, o, h, h).contiguous()), (cin, tag, float((outs[1] - want).abs().max()))
