"""Round-5 GPU tests: the windowed level-1 solve on adversarial rows, the evaluation loop on two streams from a cold model,
the reference's error behaviour of ``opt_v1`` under the strict flag, and the limits at which an entry point changes its kernel
or refuses (the int8 matrix-core convolution at 2^30 outputs, 65 535 rows of the straight-through kernels, 2^22 keys)."""

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


def _quantize(x, alpha, mode, ternary=False):
    hip = _hip()
    n, c, h, w = x.shape
    geom = hip.make_geom(n, c, h, w, 64, 3, 3, (1, 1), (1, 1), (1, 1), 1)
    planes = torch.zeros((2 * hip.act_plane_words(geom),), dtype=torch.int64, device=DEV)
    scales = torch.empty((2, n), dtype=torch.float32, device=DEV)
    with hip.debug_switches(fused_mode=mode):
        hip.act_quant(x.to(DEV), geom, hip.SCHEME_LST if ternary else hip.SCHEME_LS2, 2, 3, alpha, planes, scales)
        torch.cuda.synchronize()
    return planes.cpu(), scales.cpu()


def _window_cases():
    """Rows chosen against the windowed level-1 histogram (csrc/lsq_act_fused.hip, solve under a clamp alpha: 8192 fine bins
    over the 8 / 16 binades below the top of alpha's binade)."""
    rs = np.random.RandomState(5)
    g = lambda *s: rs.standard_normal(s).astype(np.float32)      # noqa: E731
    return {
        # everything far below the window (sigma 1e-6 under alpha 3): the candidates sit in the one-bin-per-binade region
        'tiny': (g(3, 64, 28, 28) * 1e-6, 3.0),
        # a crossing exactly at the window's lower edge: half the keys below 2^-6 (sh = 13) / 2^-14 (sh = 14), half above
        'edge13': (np.where(rs.random_sample((2, 64, 56, 56)) < 0.5, 1.0, 2.0 ** -6 * 0.999).astype(np.float32) * np.sign(g(2, 64, 56, 56)), 3.0),
        'edge14': (np.abs(g(3, 256, 14, 14)) * 2.0 ** -14, 3.0),
        # more keys in ONE fine bin than a task takes (128), all different: 4000 values inside 2^-11 relative of 1.5
        'dense_bin': (np.concatenate([1.5 + rs.random_sample((2, 4000)) * 2.0 ** -11, np.abs(g(2, 64 * 784 - 4000))], axis=1)
                      .astype(np.float32).reshape(2, 64, 28, 28), 3.0),
        # ... and exactly 128 / 129 keys in the bins around the crossings of a two-cluster row
        'cap128': (np.concatenate([np.full((2, 6272 - 300), 0.3), 1.0 + np.arange(300)[None, :] * 2.0 ** -22 * np.ones((2, 1))], axis=1)
                   .astype(np.float32).reshape(2, 128, 7, 7), 2.0),
        # the clamp value in thousands of copies next to other keys of its bin (alpha 2.5 is not a bin edge)
        'saturated': (g(3, 128, 28, 28) * 3.0, 2.5),
        'saturated_mixed': (np.clip(g(2, 64, 56, 56) * 3.0, -2.5, 2.4999), 2.5),
        # a huge and a tiny alpha (window at the top / k0 = 0), subnormal keys
        'alpha_big': (g(2, 64, 16, 16) * 1e30, 3e38),
        'alpha_tiny': (g(2, 64, 16, 16) * 1e-38, 1e-37),
        # ternary extra candidate territory: all keys nearly equal
        'flat': (1.0 + 1e-3 * g(3, 64, 12, 12), 3.0),
        # zeros and exact ties
        'half_zero': (np.maximum(g(4, 256, 14, 14), 0), 3.0),
        'grid': (np.round(g(3, 128, 28, 28) * 8) / 8, 3.0),
    }


@pytest.mark.parametrize('ternary', [False, True])
def test_windowed_solve_on_adversarial_rows(ternary):
    """Mode 0 (windowed histogram with its natural fall-backs) against the exact oracle (v1 bit-equal) and against the
    round-2 solve alone (mode 4) and the forced fall-back (mode 8): planes and both scales bit for bit."""
    for tag, (arr, alpha) in _window_cases().items():
        x = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32))
        n = x.shape[0]
        exact = E.solve_rows(x.clamp(-alpha, alpha).reshape(n, -1).numpy(), ternary, 3)
        p0, s0 = _quantize(x, alpha, 0, ternary)
        assert np.array_equal(s0[0].numpy(), exact), (tag, s0[0], exact)
        for mode in (4, 8):
            p, s = _quantize(x, alpha, mode, ternary)
            assert torch.equal(p, p0) and torch.equal(s, s0), (tag, mode)


def test_evaluate_on_two_streams_from_a_cold_model():
    """`evaluate` starts the two-stream pipeline on a model whose module-level caches (packed weights, folded
    batch norms) are EMPTY -- freshly built, and again after a state_dict load: the first forward of each stream is serialised
    behind the one before it (StreamPipeline.submit), so the metrics equal the one-stream run's."""
    import bench
    from quant.common import training
    from quant.common.metrics import LossMetric, Top1Accuracy
    shape = (8, 3, 224, 224)
    xs = [torch.randn(*shape, generator=torch.Generator().manual_seed(s)) for s in range(4)]
    targets = [torch.randint(0, 1000, (shape[0],), generator=torch.Generator().manual_seed(20 + i)) for i in range(4)]

    class Loader(list):
        dataset = list(range(4 * shape[0]))

    loader = Loader(zip(xs, targets))

    def run(streams, model):
        old = training.eval_streams
        training.eval_streams = lambda device, sharded=False, n=streams: n
        try:
            metrics = {'loss': LossMetric(torch.nn.functional.cross_entropy, True), 'top1': Top1Accuracy(True)}
            return training.evaluate(model, loader, metrics, torch.device(DEV), 1)
        finally:
            training.eval_streams = old

    for attempt in range(3):                                       # (a race would be intermittent)
        cold = bench.build_model(bench.imagenet_arch('ls-2', 3), DEV)          # never ran: every cache is empty
        two = run(2, cold)
        one = run(1, bench.build_model(bench.imagenet_arch('ls-2', 3), DEV))
        assert two == one, (attempt, two, one)
        sd = {k: v.clone() for k, v in cold.state_dict().items()}
        cold.load_state_dict(sd)                                   # clears the packed-weight caches again
        assert run(2, cold) == one, attempt


def test_projection_shortcut_on_a_side_stream_is_bit_identical():
    """quant.models.resnet.SIDE_STREAM_SHORTCUT (off by default: measured slower, scripts/sched_variants.py): the 1x1
    projection on a side stream, joined in front of the convolution that adds it."""
    import bench
    from quant.models import resnet
    model = bench.build_model(bench.imagenet_arch('ls-2', 3), DEV)
    x = torch.randn(6, 3, 224, 224, generator=torch.Generator().manual_seed(3)).to(DEV)
    with torch.no_grad():
        want = model(x).clone()
        resnet.SIDE_STREAM_SHORTCUT = True
        try:
            got = [model(x).clone() for _ in range(3)]
            torch.cuda.synchronize()
        finally:
            resnet.SIDE_STREAM_SHORTCUT = False
    assert all(torch.equal(g, want) for g in got)


@pytest.mark.parametrize('ternary', [False, True])
def test_opt_v1_strict_flag_raises_like_the_reference(golden, ternary):
    """optimal.py:147-151 of the reference: `argmin` over an empty dimension when NO row of the batch has a candidate (fixture
    keys `*_raises` record where the reference itself raised).  Default: zeros; STRICT_NO_CANDIDATE: the same IndexError, on the
    device (solver status read back) and on the host."""
    from quant.binary import optimal
    from test_gpu_round4 import _solver_rows
    g = golden('f3_solver')
    seen = 0
    for tag, rows in _solver_rows().items():
        for skip in (1, 3):
            raises = f'{tag}_t{int(ternary)}_s{skip}_raises' in g
            seen += raises
            for dev in (DEV, 'cpu'):
                optimal.STRICT_NO_CANDIDATE = True
                try:
                    if raises:
                        with pytest.raises(IndexError):
                            optimal.opt_v1(rows.to(dev), ternary, skip)
                    else:
                        optimal.opt_v1(rows.to(dev), ternary, skip)
                finally:
                    optimal.STRICT_NO_CANDIDATE = False
                v = optimal.opt_v1(rows.to(dev), ternary, skip)            # default: never raises; no candidate -> 0
                if raises:
                    assert float(v.abs().sum()) == 0.0
    assert seen >= 1 or ternary                       # (the reference raised for 2-bit rows only: a ternary row of two elements still has mean / 2)


def test_straight_through_kernels_at_their_row_limit():
    """include/lsq_hip.h: lsq_quant_values / lsq_ste_backward take up to 65 535 rows (one workgroup row per grid.y)."""
    hip = _hip()
    for rows, ok in ((65535, True), (65536, False)):
        x = torch.randn(rows, 16, generator=torch.Generator().manual_seed(1)).to(DEV)
        sc = (torch.rand(1, rows, generator=torch.Generator().manual_seed(2)) + 0.5).to(DEV)
        if ok:
            q = hip.quant_values(x, sc, 2.0)
            want = sc[0].view(-1, 1) * torch.where(x.clamp(-2, 2) >= 0, 1.0, -1.0)
            assert torch.equal(q, want)
            gx = hip.ste_backward(x, torch.ones_like(x), sc, 2.0)
            assert gx.shape == x.shape and bool(torch.isfinite(gx).all())
        else:
            with pytest.raises(hip.LsqHipError):
                hip.quant_values(x, sc, 2.0)
            with pytest.raises(hip.LsqHipError):
                hip.ste_backward(x, torch.ones_like(x), sc, 2.0)


def test_solver_at_its_key_limit():
    """2^22 sub-sampled keys per row is where the solve refuses (LSQ_E_TOO_LONG) and QuantConv2d keeps the torch formulation;
    one pixel column less is solved on the device (streaming path), bit-equal to the exact oracle."""
    hip = _hip()
    from quant.binary.binary_conv import QuantConv2d
    assert hip.MAX_SOLVER_KEYS == 1 << 22
    x_ok = torch.randn(1, 64, 384, 511, generator=torch.Generator().manual_seed(4))        # ceil(M / 3) = 4 186 112 keys
    _, s = _quantize(x_ok, 3.0, 0)
    assert np.array_equal(s[0].numpy(), E.solve_rows(x_ok.clamp(-3, 3).reshape(1, -1).numpy(), False, 3))
    x_big = torch.randn(1, 64, 384, 512, generator=torch.Generator().manual_seed(4))       # exactly 2^22 keys
    with pytest.raises(hip.LsqHipError, match='2\\^22'):
        _quantize(x_big, 3.0, 0)
    conv = QuantConv2d('ls-2', 'ls-1', 64, 64, 3, {'kind': 'symmetric', 'alpha': 3}, padding=1).eval().to(DEV)
    assert conv._hip_supports(x_ok.to(DEV)) and not conv._hip_supports(x_big.to(DEV))


def test_xnor_convolution_across_the_matrix_core_kernels_output_limit():
    """csrc/lsq_xnor_mfma.hip takes outputs below 2^30 elements; at 2^30 lsq_xnor_conv2d runs the popcount kernel (same bits:
    the first images equal a small call that the matrix-core kernel serves) and the binding says so once."""
    import warnings
    hip = _hip()
    from quant.binary.binary_conv import QuantConv2d
    conv = QuantConv2d('ls-1', 'ls-1', 64, 64, 3, {'kind': 'symmetric', 'alpha': 2}, padding=1, bias=True)
    detgen.fill_module(conv, seed=2)
    with torch.no_grad():
        conv.w_approximate.v1.copy_(conv.weight.abs().mean(dim=(1, 2, 3)))
    conv.eval().to(DEV)
    n = 4096                                                       # 4096 x 64 x 64 x 64 outputs = 2^30
    x = torch.randn(n, 64, 64, 64, generator=torch.Generator().manual_seed(6)).to(DEV)
    with torch.no_grad():
        small = conv(x[:8]).clone()
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter('always')
            hip._xnor_limit_warned.clear()
            big = conv(x)
            conv(x)
        assert sum('popcount kernel' in str(m.message) for m in w) == 1
        assert torch.equal(big[:8], small)
        below = conv(x[:4095])                                     # 2^30 - 2^18 outputs: the matrix-core kernel
        assert torch.equal(below[:8], small) and torch.equal(below[4000:4095], big[4000:4095])


def test_sharded_step_on_two_streams_over_rccl_with_one_rank():
    """What `evaluate` and bench.py do under ranks since round 5: the local forward of step i on one of two HIP streams, the
    all-gather of its logits on the caller's stream behind the forward's event -- here with the one rank there is (RCCL
    collective forced), five steps: every gathered batch equals the plain forward bit for bit."""
    import os
    import socket
    import torch.distributed as dist
    import bench
    from quant.common.sharded_eval import all_gather_logits, local_forward
    from quant.common.stream_pipeline import StreamPipeline
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    created = not dist.is_initialized()
    if created:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        model = bench.build_model(bench.imagenet_arch(), torch.device(DEV))
        xs = [torch.randn(8, 3, 224, 224, generator=torch.Generator().manual_seed(40 + i)).to(DEV) for i in range(5)]
        with torch.no_grad():
            want = [model(x).clone() for x in xs]
            pipe = StreamPipeline(lambda shard: local_forward(model, shard), DEV, 2)
            outs, window = [], []
            for x in xs:
                window.append(pipe.submit(x))
                if len(window) >= pipe.depth:
                    outs.append(all_gather_logits(window.pop(0).result(), total=8, always_collective=True).clone())
            while window:
                outs.append(all_gather_logits(window.pop(0).result(), total=8, always_collective=True).clone())
            torch.cuda.synchronize()
        assert len(outs) == 5 and all(torch.equal(o, w) for o, w in zip(outs, want))
    finally:
        if created:
            dist.destroy_process_group()


def test_bench_line_under_torchrun_with_one_rank():
    """The driver's launch line with one rank: rank 0's ONE JSON line carries the two-stream headline, the one-stream region,
    both path kernels under fixed roofline keys, the all-gather and the parity object."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from quant.common.rank_launcher import free_port
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', '10', '--warmup', '3',
           '--min-seconds', '0', '--cpu-sample', '0', '--no-configs']
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out['n_gpus'] == 1 and out['pipelined']['streams'] == 2 and out['value'] == out['pipelined']['value'] > 0
    assert out['single_stream']['value'] > 0 and out['steps_timed'] == 10 and out['timed_seconds'] > 0
    assert {'quantizer', 'xnor_conv'} <= set(out['roofline']) and out['roofline']['name'] in ('quantizer', 'xnor_conv')
    for k in ('quantizer', 'xnor_conv'):
        assert 0 < out['roofline'][k]['frac'] < 1
    assert out['allgather']['backend'] == 'nccl' and out['allgather']['world_size_seen_by_backend'] == 1
    assert len(lines[0]) < 8000                                   # (the driver keeps short lines whole)
