"""Round-4 GPU tests: the solver's chosen CANDIDATE (its sorted position, not only its value) pinned on the device."""

import numpy as np
import pytest
import torch

import detgen
from oracle import lsq_exact as E

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _hip():
    from quant import _hip
    return _hip


def _solver_rows():
    """The rows of fixture f3_solver (tests/test_oracle_golden.py builds the same)."""
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


def _check_positions(tag, rows, ternary, skip, v1, pos, g, what):
    """v1 [R] and pos [R] from the device against the sorted sub-sample and the REFERENCE's candidate positions
    (fixture f3_solver `_pos`: optimal.py:78-82 evaluated by the reference itself)."""
    key = f'{tag}_t{int(ternary)}_s{skip}'
    raises = key + '_raises' in g
    for r in range(rows.shape[0]):
        row = rows[r].numpy()
        a = np.sort(E.abs_subsample(row, skip))
        n = a.shape[0]
        d = {}
        exact = E.solve_row(row, ternary, skip, d)
        p, v = int(pos[r]), np.float32(v1[r])
        assert v == exact, (what, key, r, v, exact)
        if not d['positions']:
            assert p == -1 and v == 0.0, (what, key, r, p)                  # no candidate: the zero padding wins
            continue
        pick = d['positions'][int(np.argmin(d['cost_sq']))]
        if pick < 0:
            assert ternary and p == n + 1, (what, key, r, p)                # the appended mean / 2 (optimal.py:114-116)
            continue
        # the device's position is the first element of the chosen key's run in the ascending sub-sample ...
        assert 0 <= p < n and a[p] == v and (p == 0 or a[p - 1] < v), (what, key, r, p, v)
        assert p == int(np.searchsorted(a, v, side='left')) and a[pick] == v, (what, key, r, p, pick)
        if raises:
            continue
        # ... and lies in one of the REFERENCE's candidate clusters: within 3 sorted positions of a position the
        # reference itself flagged (its fp32 means only add neighbours), or inside a run of equal keys that holds one
        refpos = [int(q) for q in g.np(key + '_pos')[r] if q >= 0]
        assert any(abs(p - q) <= 3 or a[q] == v for q in refpos), (what, key, r, p, sorted(refpos))


@pytest.mark.parametrize('ternary', [False, True])
@pytest.mark.parametrize('skip', [1, 3])
def test_chosen_candidate_position_on_device(golden, ternary, skip):
    """lsq_solve_rows (streaming path) and lsq_act_quant (single-launch kernel, its block path and its short key list)
    report through lsq_debug_solver_trace WHICH sorted element they chose: it must be the exact oracle's, the first of
    its run, and inside one of the reference's own candidate clusters."""
    hip = _hip()
    g = golden('f3_solver')
    for tag, rows in _solver_rows().items():
        with hip.solver_trace(rows.shape[0], DEV) as trace:
            v12, _ = hip.solve_rows(rows.to(DEV), skip, ternary)
            torch.cuda.synchronize()
            pos = trace.cpu().numpy().copy()
        _check_positions(tag, rows, ternary, skip, v12[0].cpu().numpy(), pos, g, 'solve_rows')
        m = rows.shape[1]
        if skip != 3 or m % 4 or m < 16:
            continue
        # the same rows as one-channel images through the quantizer entry point: the single-launch kernel
        x = rows.view(rows.shape[0], 1, 4, m // 4).contiguous()
        geom = hip.make_geom(x.shape[0], 1, 4, m // 4, 1, 1, 1, (1, 1), (0, 0), (1, 1), 1)
        words = hip.act_plane_words(geom)
        # (a clamp above every value changes no key and switches the windowed level-1 histogram of round 5 on: mode 0 with it
        #  = windowed solve with its natural fall-backs, mode 8 = every row through the fall-back call)
        loose = max(1.0, 1.5 * float(rows.abs().max()))
        for mode, alpha in ((0, -1.0), (1, -1.0), (2, -1.0), (0, loose), (8, loose), (4, loose)):
            planes = torch.zeros((2 * words,), dtype=torch.int64, device=DEV)
            scales = torch.empty((2, x.shape[0]), dtype=torch.float32, device=DEV)
            with hip.debug_switches(fused_mode=mode), hip.solver_trace(x.shape[0], DEV) as trace:
                hip.act_quant(x.to(DEV), geom, hip.SCHEME_LST if ternary else hip.SCHEME_LS2, 2, 3, alpha, planes, scales)
                torch.cuda.synchronize()
                pos = trace.cpu().numpy().copy()
            _check_positions(tag, rows, ternary, skip, scales[0].cpu().numpy(), pos, g, f'act_quant mode {mode} alpha {alpha}')


def test_trace_is_off_by_default_and_after_the_block():
    hip = _hip()
    rows = detgen.normal('r4.trace', (3, 4096)).clamp(-3, 3)
    with hip.solver_trace(3, DEV) as trace:
        hip.solve_rows(rows.to(DEV), 3, False)
    first = trace.cpu().clone()
    assert (first >= 0).all()
    trace.fill_(-7)
    hip.solve_rows(rows.to(DEV), 3, False)
    torch.cuda.synchronize()
    assert (trace.cpu() == -7).all()                   # the hook is cleared: nothing writes to the old buffer


# ------------------------------------------------------------------------------------------------ training
def _chain(x, scales, alpha):
    """The quantizer chain in torch: value and the straight-through gradient's closed form (tests/test_training.py pins
    this formula to autograd through the reference-equal torch formulation)."""
    inside = (x >= -alpha) & (x <= alpha) if alpha >= 0 else torch.ones_like(x, dtype=torch.bool)
    c = x.clamp(-alpha, alpha) if alpha >= 0 else x
    shape = (-1,) + (1,) * (x.dim() - 1)
    r, d = torch.zeros_like(c), []
    for v in scales:
        di = c - r
        d.append(di)
        r = r + v.view(shape) * torch.where(di >= 0, 1.0, -1.0)

    def grad(g):
        G, acc = g.clone(), torch.zeros_like(g)
        for v, di in zip(reversed(scales), reversed(d)):
            t = torch.where(di.abs() <= 1, G * v.view(shape), torch.zeros_like(G))
            acc = acc + t
            G = G - t
        return torch.where(inside, acc if len(scales) else g, torch.zeros_like(g))
    return (r if len(scales) else c), grad


@pytest.mark.parametrize('shape', [(5, 16, 6, 6), (3, 7, 5, 3), (24, 32, 3, 3), (2, 1, 1, 9)])
@pytest.mark.parametrize('k', [0, 1, 2, 3])
def test_ste_kernels_equal_the_torch_chain(shape, k):
    """lsq_quant_values (bit for bit) and lsq_ste_backward (1e-6) against the chain in torch, for activations (clamp) and
    weight-shaped rows (no clamp), row lengths with and without 16-byte rows, +-0, values on the clamp and on the |d| = 1
    edge of the estimator."""
    hip = _hip()
    x = detgen.normal(f'r4.ste.x.{shape}.{k}', shape, scale=1.3)
    x.view(-1)[::11] = 0.0
    x.view(-1)[3::13] = -0.0
    x.view(-1)[5::17] = 2.0
    x.view(-1)[7::19] = -2.0
    g = detgen.normal(f'r4.ste.g.{shape}.{k}', shape)
    scales = [detgen.uniform(f'r4.ste.v{i}.{shape}', (shape[0],), 0.2, 1.1) for i in range(k)]
    if k:
        x.view(-1)[9::23] = 1.0 + float(scales[0][0])                       # |d_1| = 1 exactly in row 0
    for alpha in (2.0, -1.0):
        want, grad = _chain(x, scales, alpha)
        sc = torch.stack(scales).to(DEV) if k else None
        got = hip.quant_values(x.to(DEV), sc, alpha).cpu()
        assert torch.equal(got, want), (shape, k, alpha)
        gx = hip.ste_backward(x.to(DEV), g.to(DEV), sc, alpha).cpu()
        assert torch.allclose(gx, grad(g), rtol=1e-6, atol=1e-7), (shape, k, alpha, float((gx - grad(g)).abs().max()))
    with pytest.raises(Exception):
        hip.ste_backward(x.to(DEV), g[:1].to(DEV), None, 2.0)


TRAIN_CASES = [('ls-2', 'ls-1', 1, True), ('ls-1', 'ls-1', 2, True), ('gf-2', 'ls-1', 1, False), ('ls-T', 'ls-1', 2, True),
               ('fp', 'ls-1', 1, True), ('ls-1', 'gf-2', 1, True), ('ls-1', 'ls-2', 2, False), ('ls-2', 'ls-T', 1, True)]


@pytest.mark.parametrize('xs,ws,stride,bias', TRAIN_CASES)
def test_train_step_on_the_kernels_equals_the_torch_formulation(xs, ws, stride, bias):
    """One train-mode step of QuantConv2d through quant.binary.hip_train (forward on lsq_act_quant + lsq_xnor_conv2d /
    lsq_signw_conv2d, backward on lsq_signw_conv2d (transposed) + lsq_ste_backward) against the SAME module on the torch
    formulation on the device (autograd through STESign, the graph the f9_train fixture pins to the reference): output,
    the three gradients, the cached weight scales."""
    from quant.binary.binary_conv import QuantConv2d
    clamp = {'kind': 'symmetric', 'alpha': 2}
    convs = []
    for hip_path in (True, False):
        conv = QuantConv2d(xs, ws, 32, 48, 3, clamp, stride=stride, padding=1, bias=bias)
        with torch.no_grad():
            conv.weight.copy_(detgen.normal('r4.train.w', conv.weight.shape, scale=0.3))
            if bias:
                conv.bias.copy_(detgen.normal('r4.train.b', conv.bias.shape, scale=0.1))
        conv.hip_train = hip_path
        convs.append(conv.to(DEV).train())
    out = []
    for conv in convs:
        x = detgen.normal('r4.train.x', (4, 32, 13, 10), scale=1.2).to(DEV).requires_grad_()
        gy_shape = (4, 48, (13 - 1) // stride + 1, (10 - 1) // stride + 1)
        y = conv(x)
        assert tuple(y.shape) == gy_shape
        y.backward(detgen.normal('r4.train.gy', gy_shape).to(DEV))
        out.append((x, y, conv))
    (x1, y1, c1), (x2, y2, c2) = out
    assert type(y1.grad_fn).__name__ == '_QuantConv2dStepBackward' and type(y2.grad_fn).__name__ != '_QuantConv2dStepBackward'

    def rel(a, b):
        a, b = a.detach(), b.detach()
        return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
    assert rel(y1, y2) <= 1e-5, rel(y1, y2)
    assert rel(x1.grad, x2.grad) <= 2e-5, rel(x1.grad, x2.grad)             # (bf16 hi + lo terms of the transposed convolution)
    assert rel(c1.weight.grad, c2.weight.grad) <= 1e-5, rel(c1.weight.grad, c2.weight.grad)
    if bias:
        assert rel(c1.bias.grad, c2.bias.grad) <= 1e-5
    for (n1, b1), (n2, b2) in zip(c1.w_approximate.named_buffers(), c2.w_approximate.named_buffers()):
        assert n1 == n2 and torch.equal(b1, b2) and float(b1.abs().sum()) > 0, n1


@pytest.mark.parametrize('mode', ['eval_only', 'train_and_eval'])
def test_train_step_moving_average_modes_on_the_kernels(mode):
    """activation_quantization.py:72-88 in training: the moving average tracks the batch's mean scales; 'train_and_eval'
    quantizes with the tracked values.  Kernels against the torch formulation over three steps."""
    from quant.binary.binary_conv import QuantConv2d
    convs = []
    for hip_path in (True, False):
        conv = QuantConv2d('ls-2', 'ls-1', 32, 32, 3, {'kind': 'symmetric', 'alpha': 3}, mode, 0.9, padding=1)
        with torch.no_grad():
            conv.weight.copy_(detgen.normal('r4.ma.w', conv.weight.shape, scale=0.3))
            conv.bias.copy_(detgen.normal('r4.ma.b', conv.bias.shape, scale=0.1))
        conv.hip_train = hip_path
        convs.append(conv.to(DEV).train())
    for step in range(3):
        ys = []
        for conv in convs:
            x = detgen.normal(f'r4.ma.x{step}', (3, 32, 8, 8), scale=1.1).to(DEV).requires_grad_()
            y = conv(x)
            y.sum().backward()
            ys.append((y, x.grad))
        assert float((ys[0][0] - ys[1][0]).abs().max()) <= 1e-5 * float(ys[1][0].abs().max())
        assert float((ys[0][1] - ys[1][1]).abs().max()) <= 2e-5 * float(ys[1][1].abs().max())
        ma = [c.x_approximate.moving_avg_module.moving_average for c in convs]
        assert torch.allclose(ma[0], ma[1], rtol=2e-6), (step, ma)


def test_training_loop_on_the_gpu_takes_the_kernels():
    """quant.common.training.train on cuda:0 (training.py:66-152): every quantized layer of a small XNOR ResNet runs its
    step on the kernels, the loss falls, and two steps from the same start match the torch formulation's two steps."""
    import quant.binary.hip_train as HT
    from quant.binary.binary_conv import QuantConv2d
    from quant.common.initialization import get_lr_scheduler, get_optimizer
    from quant.common.metrics import LossMetric
    from quant.common.training import train
    from quant.models.resnet import QResNet
    layer = {'x_quant': 'ls-2', 'w_quant': 'ls-1', 'clamp': {'kind': 'symmetric', 'alpha': 3}, 'double_shortcut': True}
    arch = {'moving_average_mode': 'off', 'moving_average_momentum': 0.99, 'block': 'xnor',
            'layer0': {'n_in_channels': 16, 'kernel_size': 3, 'stride': 1, 'padding': 1, 'bias': False, 'maxpool': {'type': 'identity'}},
            'layer1': layer, 'layer2': layer, 'layer3': layer, 'layer4': layer, 'nonlins': ['relu', 'relu'],
            'num_blocks': [1, 1, 1, 1], 'output_classes': 10}
    g = torch.Generator().manual_seed(5)
    data = torch.randn(64, 3, 16, 16, generator=g)
    target = torch.randint(0, 10, (64,), generator=g)
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(data, target), batch_size=16)
    calls = []
    orig = HT.train_step_forward
    losses = {}
    try:
        HT.train_step_forward = lambda conv, x: (calls.append(conv), orig(conv, x))[1]
        for hip_path in (True, False):
            torch.manual_seed(11)
            model = QResNet(loss_fn=torch.nn.functional.cross_entropy, **arch).to(DEV)
            for m in model.modules():
                if isinstance(m, QuantConv2d):
                    m.hip_train = hip_path
            opt = get_optimizer(model.parameters(), {'algorithm': 'sgd', 'lr': 0.02, 'momentum': 0.9})
            sched = get_lr_scheduler(opt, {'scheduler': 'step_lr', 'step_size': 10, 'gamma': 0.5}, 3, len(loader))
            metrics = {'Loss': LossMetric(model.loss_fn, accumulate=True)}
            losses[hip_path] = [train(model, loader, metrics, opt, sched, torch.device(DEV), e, 100)['Loss'] for e in (1, 2, 3)]
    finally:
        HT.train_step_forward = orig
    assert len(calls) == 8 * 4 * 3                              # 8 quantized layers x 4 batches x 3 epochs, kernels only
    assert losses[True][2] < losses[True][0]
    assert losses[True][0] == pytest.approx(losses[False][0], rel=2e-2)      # (a binarized net amplifies fp32 reassociation)


# ------------------------------------------------------------------------------------------------ chained 1-bit layers
@pytest.mark.parametrize('which,batch', [('cifar', 100), ('cifar', 7), ('imagenet_ls1', 6)])
def test_chained_one_bit_layers_equal_the_unchained_network(which, batch):
    """quant.binary.chain: with ls-1 activations every QuantConv2d after the first gets its bit plane and its exact row sum
    from the epilogue of the convolution in front of it (lsq_xnor_conv2d_chain) -- 15 of the 16 quantizer launches of a
    CIFAR ResNet-18 disappear (layers of more than chain.MAX_ELEMENTS elements keep theirs) -- and the logits are those of the unchained network BIT FOR BIT, eagerly and under graph replay."""
    import bench
    from quant.binary import chain
    from quant.common.graph_replay import GraphedForward
    hip = _hip()
    if which == 'cifar':
        model = bench.build_model(bench.cifar_arch(), DEV)
        x = torch.randn(batch, 3, 32, 32, generator=torch.Generator().manual_seed(1)).to(DEV)
    else:
        model = bench.build_model(bench.imagenet_arch('ls-1', 2), DEV)       # PReLU blocks, 7x7 stem + max-pool
        x = torch.randn(batch, 3, 224, 224, generator=torch.Generator().manual_seed(2)).to(DEV)
    out, calls = {}, {}
    for on in (False, True):
        chain.ENABLED = on
        try:
            with torch.no_grad():
                model(x)
                hip.enable_timing(True)
                out[on] = model(x).clone()
                torch.cuda.synchronize()
                calls[on] = {k: v[0] for k, v in hip.drain_timing().items()}
                hip.enable_timing(False)
        finally:
            chain.ENABLED = True
            hip.enable_timing(False)
    assert calls[False]['lsq_act_quant'] == 16 and calls[False]['lsq_xnor_conv2d'] == 16, calls
    # (layers of more than chain.MAX_ELEMENTS elements would keep their own quantizer launch: none at these batch sizes)
    assert calls[True]['lsq_act_quant'] == 1 and calls[True]['lsq_xnor_conv2d'] == 16, calls
    assert torch.equal(out[True], out[False]), float((out[True] - out[False]).abs().max())
    fwd = GraphedForward(model, x)
    assert torch.equal(fwd.replay(), out[False])
    # and against the oracle's logits for this very model (free-running: no search in ls-1, so 1e-4 of max |logit| + the
    # stem's fp differences amplified -- the bound of the unchained network)
    from oracle import ref_models
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    arch = bench.cifar_arch() if which == 'cifar' else bench.imagenet_arch('ls-1', 2)
    ref = ref_models.resnet_forward(sd, arch, x.cpu()[:4])
    got = out[True].cpu()[:4]
    # (limit derived on the CPU, tests/golden/make_free_limits.py chained_cases: no search and exact integer convolutions in
    #  these networks -- north_star's 1e-4 for the fp32 order of operations of stem and scales, plus twice the oracle network's
    #  own sensitivity to ulp-sized input noise on these four samples)
    import test_gpu_parity as TP
    key = f'chained_{which}_b{batch}'
    err = TP.observe(key, float((got - ref).abs().max()) / float(ref.abs().max()))
    assert err <= TP.FREE_LIMIT[key], (key, err, TP.FREE_LIMIT[key])


def test_lone_row_split_sweep_replayed_in_a_graph():
    """A HIP graph re-issues a captured launch with the SAME arguments -- the arrival slots' epoch among them.  One ls-1
    sweep of few rows (shared by several workgroups each), captured alone and replayed on new data: every replay writes
    every scale (the last arrival releases the slot; with the epoch alone the second replay found a full count)."""
    hip = _hip()
    n, c, h, w = 4, 64, 32, 32
    geom = hip.make_geom(n, c, h, w, 64, 3, 3, (1, 1), (1, 1), (1, 1), 1)
    x = torch.randn(n, c, h, w, device=DEV)
    planes = torch.zeros((hip.act_plane_words(geom),), dtype=torch.int64, device=DEV)
    scales = torch.zeros((1, n), device=DEV)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        hip.act_quant(x, geom, hip.SCHEME_LS1, 1, 3, 2.0, planes, scales)          # (allocates the stream's sweep workspace)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        hip.act_quant(x, geom, hip.SCHEME_LS1, 1, 3, 2.0, planes, scales)
    for seed in range(4):
        x.copy_(torch.randn(n, c, h, w, generator=torch.Generator().manual_seed(seed)).to(DEV) * (seed + 1))
        scales.fill_(-1.0)
        graph.replay()
        want = x.clamp(-2.0, 2.0).abs().double().mean(dim=(1, 2, 3)).float()
        assert torch.allclose(scales[0], want, rtol=1e-6, atol=0), (seed, scales, want)


def test_chained_record_is_void_after_an_in_place_edit():
    """The plane and row sums a producer leaves for its consumer describe the values it STORED: a forward hook that edits the
    tensor in place between two blocks (its ``_version`` moves) makes the consumer quantize what is there now -- the network
    with the hook gives the same logits chained and unchained, bit for bit, and one more quantizer launch runs."""
    import bench
    from quant.binary import chain
    hip = _hip()
    model = bench.build_model(bench.cifar_arch(), DEV)
    x = torch.randn(8, 3, 32, 32, generator=torch.Generator().manual_seed(3)).to(DEV)
    blocks = [m for m in model.modules() if type(m).__name__ == 'XnorBasicBlock']
    handle = blocks[2].register_forward_hook(lambda mod, inp, out: out.add_(0.25))
    out, quant_calls = {}, {}
    try:
        for on in (False, True):
            chain.ENABLED = on
            with torch.no_grad():
                model(x)
                hip.enable_timing(True)
                out[on] = model(x).clone()
                torch.cuda.synchronize()
                quant_calls[on] = hip.drain_timing()['lsq_act_quant'][0]
                hip.enable_timing(False)
    finally:
        chain.ENABLED = True
        hip.enable_timing(False)
        handle.remove()
    assert torch.equal(out[True], out[False])
    assert quant_calls[False] == 16 and quant_calls[True] == 2, quant_calls      # the first layer + the consumer behind the hook


def test_consecutive_batches_on_two_streams_equal_the_single_stream_forward():
    """quant/common/stream_pipeline.py: consecutive batches alternate between two HIP streams (every workspace of the binding
    and of the modules is per launch stream) -- the outputs are those of the plain forward bit for bit, in order, for the solving
    (ls-2) and the chained 1-bit network, and `evaluate` reports the same metrics with it as without."""
    import bench
    from quant.common import training
    from quant.common.metrics import LossMetric, Top1Accuracy
    from quant.common.stream_pipeline import StreamPipeline
    for arch, shape in ((bench.imagenet_arch('ls-2', 3), (6, 3, 224, 224)), (bench.cifar_arch(), (20, 3, 32, 32))):
        model = bench.build_model(arch, DEV)
        xs = [torch.randn(*shape, generator=torch.Generator().manual_seed(s)).to(DEV) for s in range(5)]
        with torch.no_grad():
            want = [model(x).clone() for x in xs]
            pipe = StreamPipeline(model, DEV, 2)
            assert pipe.depth == 2
            got = [y.clone() for y in pipe.map(xs)]
            torch.cuda.synchronize()
        assert len(got) == len(want) and all(torch.equal(g, w) for g, w in zip(got, want))
        targets = [torch.randint(0, want[0].shape[1], (shape[0],), generator=torch.Generator().manual_seed(9 + i)) for i in range(5)]

        class Loader(list):
            dataset = list(range(5 * shape[0]))

        loader = Loader((x.cpu(), t) for x, t in zip(xs, targets))
        res = {}
        for streams in (1, 2):
            old = training.eval_streams
            training.eval_streams = lambda device, sharded=False, n=streams: n
            try:
                metrics = {'loss': LossMetric(torch.nn.functional.cross_entropy, True), 'top1': Top1Accuracy(True)}
                res[streams] = training.evaluate(model, loader, metrics, DEV, 1)
            finally:
                training.eval_streams = old
        assert res[1] == res[2], res
