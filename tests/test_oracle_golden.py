"""Pin the oracle (oracle/ref_port.py, oracle/lsq_exact.py, oracle/radix_select_model.py)
against outputs of the reference itself (tests/golden/*.npz, made by make_fixtures.py)."""

import numpy as np
import pytest
import torch

import detgen
from oracle import lsq_exact as E
from oracle import radix_select_model as RM
from oracle import ref_port as P

torch.set_num_threads(8)

PAIRS = [('ls-1', 'ls-1'), ('ls-2', 'ls-1'), ('ls-T', 'ls-1'), ('gf-2', 'ls-1'),
         ('fp', 'ls-1'), ('fp', 'fp'), ('ls-2', 'ls-2'), ('ls-1', 'gf-2'), ('ls-T', 'ls-T')]


def test_sign_table(golden):
    g = golden('f1_sign')
    assert torch.equal(P.pm1(g['x']), g['sign'])       # sign(+-0) = +1, tests/binary/test_ste.py:13


def test_quantizers_bit_exact(golden):
    g = golden('f24_quantizers')
    x = detgen.normal('f24.x', (4, 64, 14, 14), scale=1.3).clamp(-3, 3)
    v1, xq = P.quant_ls1(x)
    assert torch.equal(v1, g['ls1_v1']) and torch.equal(xq, g['ls1_xq'])
    v1, v2, xq = P.quant_ls2(x)
    assert torch.equal(v1, g['ls2_v1']) and torch.equal(v2, g['ls2_v2']) and torch.equal(xq, g['ls2_xq'])
    v1, v2, _ = P.quant_ls2(x, skip=1)
    assert torch.equal(v1, g['ls2s1_v1']) and torch.equal(v2, g['ls2s1_v2'])
    v1, xq = P.quant_lst(x)
    assert torch.equal(v1, g['lst_v1']) and torch.equal(xq, g['lst_xq'])
    vs, xq = P.quant_gf(x, 2)
    assert torch.equal(vs[0], g['gf2_v1']) and torch.equal(vs[1], g['gf2_v2']) and torch.equal(xq, g['gf2_xq'])
    vs, xq = P.quant_gf(x, 3)
    assert all(torch.equal(a, g[f'gf3_v{i + 1}']) for i, a in enumerate(vs))
    assert np.array_equal(np.packbits((xq.numpy() > 0).astype(np.uint8).reshape(-1)), g.np('gf3_b'))
    _, _, xq = P.quant_ls2(x, g['inj1'], g['inj2'])
    assert torch.equal(xq, g['ls2_inj_xq'])
    assert torch.equal(P.quant_lst(x, g['inj1'])[1], g['lst_inj_xq'])
    assert torch.equal(P.quant_ls1(x, g['inj1'])[1], g['ls1_inj_xq'])
    assert torch.equal(P.quant_ls2(x, g['inj1'])[1], g['ls2_v2_from_inj1'])


def _solver_rows():
    rows = {'long': detgen.normal('f3.long', (4, 25088), scale=1.0).clamp(-3, 3),
            'relu': detgen.normal('f3.relu', (4, 3000)).clamp(min=0),
            'sat': detgen.normal('f3.sat', (4, 3001)).clamp(-0.5, 0.5)}
    for n in (3, 4, 5, 7, 10, 11, 64):
        rows[f'short{n}'] = detgen.normal(f'f3.short{n}', (6, n))
    mix = detgen.uniform('f3.mix', (8, 768))
    mix[1] = 2.0
    mix[5] = -3.0
    mix[6] = detgen.uniform('f3.mix6', (768,), 1.0, 1.2)
    rows['mix'] = mix
    return rows


@pytest.mark.parametrize('ternary', [False, True])
@pytest.mark.parametrize('skip', [1, 3])
def test_solver_internals_bit_exact(golden, ternary, skip):
    """Candidate sets, per-candidate costs, argmin and v1 equal the reference's, bit for bit."""
    g = golden('f3_solver')
    for tag, rows in _solver_rows().items():
        key = f'{tag}_t{int(ternary)}_s{skip}'
        if key + '_raises' in g:
            with pytest.raises(RuntimeError):
                P.solve_v1(rows, ternary, skip)
            continue
        d = {}
        v1 = P.solve_v1(rows, ternary, skip, details=d)
        assert torch.equal(v1.view(-1), g[key + '_v1']), key
        assert torch.equal(d['padded'], g[key + '_cands']), key
        assert torch.equal(d['costs'], g[key + '_costs']), key
        assert [int(c.numel()) for c in d['candidates']] == g[key + '_sizes'].tolist(), key
        # chunked evaluation of the cost tensor changes nothing
        assert torch.equal(P.solve_v1(rows, ternary, skip, chunk=2), v1)


@pytest.mark.parametrize('ternary', [False, True])
@pytest.mark.parametrize('skip', [1, 3])
def test_exact_solver_vs_reference(golden, ternary, skip):
    """The fp64 restatement (what the GPU computes) against the fp32 reference: its choice is one
    of the reference's candidates' neighbours -- v1 within 1e-3 relative and true cost not worse
    than the reference's by more than 1e-5 relative (hard part 1)."""
    g = golden('f3_solver')
    for tag, rows in _solver_rows().items():
        key = f'{tag}_t{int(ternary)}_s{skip}'
        if key + '_raises' in g:
            continue
        ref = g.np(key + '_v1')
        sizes = g.np(key + '_sizes')
        for r in range(rows.shape[0]):
            row = rows[r].numpy()
            d = {}
            mine = float(E.solve_row(row, ternary, skip, d))
            if sizes[r] == 0:
                assert mine == 0.0 and ref[r] == 0.0       # zero padding wins by default
                continue
            c_ref = E.true_cost(row, float(ref[r]), ternary, skip)
            c_mine = E.true_cost(row, mine, ternary, skip)
            assert c_mine <= c_ref * (1 + 1e-5) + 1e-12, (key, r, mine, ref[r])
            assert abs(mine - ref[r]) <= 1e-3 * abs(ref[r]) + 1e-12, (key, r, mine, ref[r])
            # exact candidates are a subset of the reference's (fp32 noise only adds neighbours)
            refpos = set(int(p) for p in g.np(key + '_pos')[r] if p >= 0)
            for p in d['positions']:
                if p >= 0:
                    assert any(abs(p - q) <= 3 for q in refpos), (key, r, p, sorted(refpos))


@pytest.mark.parametrize('ternary', [False, True])
@pytest.mark.parametrize('skip', [1, 3])
def test_radix_select_model_equals_exact(ternary, skip):
    """The host model of the GPU algorithm (3-level radix select, exact integer bin sums) picks
    exactly the element the sort-based fp64 statement picks."""
    rs = np.random.RandomState(7)
    cases = [rs.standard_normal((6, 3000)).astype(np.float32).clip(-3, 3),
             np.maximum(rs.standard_normal((4, 2000)), 0).astype(np.float32),
             rs.standard_normal((4, 2001)).astype(np.float32).clip(-0.5, 0.5),
             np.round(rs.standard_normal((4, 1500)) * 4).astype(np.float32) / 4,
             np.exp(rs.standard_normal((4, 1000)) * 8).astype(np.float32),
             np.full((2, 300), 2.0, dtype=np.float32),
             (rs.pareto(2.0, (2, 1200)) + 1).astype(np.float32)]
    for n in (1, 2, 3, 4, 5, 7, 11):
        cases.append(rs.standard_normal((4, n)).astype(np.float32))
    for rows in cases:
        for row in rows:
            assert E.solve_row(row, ternary, skip) == RM.solve_row_model(row, ternary, skip)


def _ref_conv_inputs(xs, ws, stride, alpha, g, key):
    w = detgen.normal('f5.w.64.64.3', (64, 64, 3, 3), scale=(64 * 9) ** -0.5)
    b = detgen.normal('f5.w.64.64.3.b', (64,), scale=0.1)
    wsc = [g[f'{key}_w_v{i}'] for i in range(1, 9) if f'{key}_w_v{i}' in g]
    return w, b, wsc


@pytest.mark.parametrize('xs,ws', PAIRS)
def test_quant_conv2d_bit_exact(golden, xs, ws):
    g = golden('f5_conv')
    x = detgen.normal('f5.x', (2, 64, 14, 14), scale=1.2)
    for stride in (1, 2):
        for alpha in (2, 3):
            key = f'{xs}_{ws}_s{stride}_a{alpha}'
            if key + '_y' not in g:
                continue
            w, b, wsc = _ref_conv_inputs(xs, ws, stride, alpha, g, key)
            # the cached scales are what a train-mode forward computes
            for mine, ref in zip(P.weight_scales(w, ws), wsc):
                assert torch.equal(mine, ref)
            d = {}
            y = P.quant_conv2d(x, w, b, xs, ws, wsc, {'kind': 'symmetric', 'alpha': alpha}, stride, 1, details=d)
            assert torch.equal(y, g[key + '_y']), key
            if key + '_xv1' in g:
                assert torch.equal(d['act_scales'][0], g[key + '_xv1'])


def test_quant_conv2d_lenet_geometry_and_edge_cases(golden):
    g = golden('f5_conv')
    xl = detgen.normal('f5.xl', (2, 20, 12, 12))
    w = detgen.normal('f5.w.20.50.5', (50, 20, 5, 5), scale=500 ** -0.5)
    b = detgen.normal('f5.w.20.50.5.b', (50,), scale=0.1)
    for xs in ('ls-1', 'ls-2', 'fp', 'gf-2', 'ls-T'):
        y = P.quant_conv2d(xl, w, b, xs, 'ls-1', [g[f'lenet_{xs}_ls-1_w_v1']])
        assert torch.equal(y, g[f'lenet_{xs}_ls-1_y'])
    # never-trained module: zero weight scales -> output is the bias
    x = detgen.normal('f5.x', (2, 64, 14, 14), scale=1.2)
    w = detgen.normal('f5.w.64.64.3', (64, 64, 3, 3), scale=(64 * 9) ** -0.5)
    b = detgen.normal('f5.w.64.64.3.b', (64,), scale=0.1)
    y = P.quant_conv2d(x, w, b, 'ls-1', 'ls-1', [torch.zeros(64)], None, 1, 1)
    assert torch.equal(y, g['untrained_y'])
    # dilation / groups / rectangular kernel
    wg = detgen.normal('f5.w.64.64.(3, 2)', (64, 32, 3, 2), scale=(32 * 6) ** -0.5)
    bg = detgen.normal('f5.w.64.64.(3, 2).b', (64,), scale=0.1)
    y = P.quant_conv2d(x, wg, bg, 'ls-2', 'ls-1', [g['geo_w_v1']], {'kind': 'symmetric', 'alpha': 2},
                       (2, 1), (2, 1), (2, 1), 2)
    assert torch.equal(y, g['geo_y'])


# ------------------------------------------------------------------------------------------------
def _product_model(kind, arch, seed):
    """Parameters come from the product's module tree only for their names/shapes (detgen fills them)."""
    from quant.binary.binary_conv import QuantConv2d
    from quant.models.lenet import QLeNet5
    from quant.models.resnet import QResNet
    cls = QResNet if kind == 'resnet' else QLeNet5
    model = cls(loss_fn=None, **arch)
    detgen.fill_module(model, seed=seed)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, QuantConv2d) and hasattr(m.w_approximate, 'v1'):
                for buf, v in zip(m.w_approximate.cached_scales(), P.weight_scales(m.weight, m.w_quant)):
                    buf.copy_(v)
    return model.eval()


@pytest.mark.parametrize('tag,shape', [('imagenet_ls1w_ls2a', (2, 3, 64, 64)), ('imagenet_ls1w_fpa', (2, 3, 64, 64)),
                                       ('imagenet_ls1w_lsTa', (2, 3, 64, 64)), ('imagenet_ls1w_gf2a', (2, 3, 64, 64)),
                                       ('cifar100_ls1', (4, 3, 32, 32))])
def test_resnet_oracle_logits_bit_exact(golden, tag, shape):
    from oracle import ref_models as RMo
    g = golden('f6_models')
    arch = g.json(tag + '_arch')
    sd = {k: v.detach() for k, v in _product_model('resnet', arch, 1).state_dict().items()}
    scales = {}
    y = RMo.resnet_forward(sd, arch, detgen.normal(tag + '.x', shape), scales_out=scales)
    assert torch.equal(y, g[tag + '_logits']), tag
    for name, sc in scales.items():
        key = f'{tag}_scales_{name}'
        if key in g:
            assert torch.equal(torch.stack(sc), g[key]), key


@pytest.mark.parametrize('tag', ['mnist_ls1w_fpa', 'mnist_ls1', 'mnist_ls1w_ls2a'])
def test_lenet_oracle_bit_exact(golden, tag):
    from oracle import ref_models as RMo
    g = golden('f7_lenet')
    arch = g.json(tag + '_arch')
    sd = {k: v.detach() for k, v in _product_model('lenet', arch, 3).state_dict().items()}
    y = RMo.lenet_forward(sd, arch, detgen.normal(tag + '.x', (64, 1, 28, 28)))
    assert torch.equal(y, g[tag + '_logp']), tag
