#!/usr/bin/env python3
"""Headline benchmark: images/sec of ResNet-18 (ImageNet, ls-1 weights / ls-2 activations) eval forward.

    python bench.py --gpus N --steps K --warmup W

One process per GPU; each rank holds a full replica, takes its own 256 synthetic 3x224x224 images per step (weak
scaling) and the ranks all-gather their logits over RCCL every step.  Rank 0 prints ONE JSON line (see DESIGN.md
"Measurement").  Under torchrun the ranks are torchrun's (RANK / LOCAL_RANK / WORLD_SIZE in the environment); started
plainly with --gpus N > 1 the script starts its N ranks itself (quant/common/rank_launcher.py), each pinned to the
cores of its GPU's NUMA node -- `python bench.py --gpus 8` needs no external launcher.
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'ml-quant_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def _layer(xq, alpha):
    return {'x_quant': xq, 'w_quant': 'ls-1', 'clamp': {'kind': 'symmetric', 'alpha': alpha}, 'double_shortcut': True}


def imagenet_arch(x_quant='ls-2', alpha=3):
    """arch_config of examples/imagenet/imagenet_ls1_weight_<x_quant>_activation_kd.yaml (model section): the ls-2
    file uses ReLU blocks, the fp / ls-T / gf-2 / ls-1 files PReLU ones."""
    return {
        'moving_average_mode': 'off', 'moving_average_momentum': 0.99, 'block': 'xnor',
        'layer0': {'n_in_channels': 64, 'kernel_size': 7, 'stride': 2, 'padding': 3, 'bias': False,
                   'maxpool': {'type': 'maxpool2d', 'kernel_size': 3, 'stride': 2, 'padding': 1}},
        'layer1': _layer(x_quant, alpha), 'layer2': _layer(x_quant, alpha),
        'layer3': _layer(x_quant, alpha), 'layer4': _layer(x_quant, alpha),
        'nonlins': ['relu', 'relu'] if x_quant == 'ls-2' else ['prelu', 'prelu'], 'num_blocks': [2, 2, 2, 2],
        'output_classes': 1000}


def build_model(arch, device):
    """Default nn init under manual_seed(0); weight scales as one train-mode forward would cache them
    (u_o = mean|W_o|, weight_quantization.py:29-31); BatchNorm at its initial running statistics."""
    from quant.binary.binary_conv import QuantConv2d
    from quant.models.resnet import QResNet
    torch.manual_seed(0)
    model = QResNet(loss_fn=torch.nn.functional.cross_entropy, **arch)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, QuantConv2d) and m.w_quant == 'ls-1':
                m.w_approximate.v1.copy_(m.weight.abs().mean(dim=(1, 2, 3)))
    return model.eval().to(device)


def git_blob_hash(path):
    """The id `git hash-object` gives the file: lets a reader check WHICH committed profile a figure was read from."""
    import hashlib
    data = open(path, 'rb').read()
    return hashlib.sha1(b'blob %d\0' % len(data) + data).hexdigest()


def profile_is_current(doc):
    """A counter profile describes the kernels it was captured on: it records the fingerprint of csrc/ at capture time
    (scripts/pmc_traffic.sh, scripts/pmc_sq_table.py) and is used only while that is still the source tree's."""
    from quant import _hip
    return doc.get('csrc_sha256') == _hip.source_fingerprint()


def pmc_traffic_per_launch(entry, act):
    """HBM bytes per C-ABI launch of ``entry`` from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE in separate passes, FETCH_SIZE doubled on gfx950: scripts/pmc_traffic.sh ->
    profiles/*_pmc_hbm_traffic.json), weighted by the kernel launches per step of the committed kernel
    trace (profiles/*_per_step_summary*.csv).  Counters cannot be read from inside this process, so this is the last
    profiled build's figure.  The passes run over bench.py's headline workload (``act`` ls-2) or its --act fp variant and
    belong to THAT workload's main leg only; None for any other workload, when the profiles are absent, or when the
    kernel sources changed after the capture ({'stale': ...} then)."""
    import csv
    import glob
    if act not in ('ls-2', 'fp'):
        return None
    suffix = '_fpact' if act == 'fp' else ''
    prefix = {'lsq_act_quant': 'aq_', 'lsq_xnor_conv2d': 'xnor_', 'lsq_signw_conv2d': 'signw_conv_'}.get(entry)
    tables = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_pmc_hbm_traffic%s.json' % suffix)))
    steps = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_rocprofv3_per_step_summary%s.csv' % suffix)))
    if not prefix or not tables or not steps:
        return None
    doc = json.load(open(tables[-1]))
    source = '%s (git blob %s)' % (os.path.basename(tables[-1]), git_blob_hash(tables[-1])[:12])
    if not profile_is_current(doc):
        return {'stale': source + ' was captured on other kernel sources than this tree\'s (csrc fingerprint differs)'}
    def norm(name):                # "<namespace junk>::<kernel><...> grid=<threads>" -> "<kernel><...>" (None: another kernel)
        name = name.split(' grid=')[0]
        i = name.find(prefix)
        return name[i:] if i >= 0 else None
    per_kernel = {}
    for name, v in doc['kernels'].items():
        if norm(name):
            per_kernel.setdefault(norm(name), []).append(1e6 * (v['hbm_read_MB_corrected'] + v['hbm_write_MB']))
    total = launches = 0.0
    with open(steps[-1]) as f:
        next(f)
        for row in csv.DictReader(f):
            kname = norm(row['kernel'].split('(lsq')[0])                 # (the trace summary cuts names at 40 characters)
            if not kname:
                continue
            match = [k for k in per_kernel if k.startswith(kname) or kname.startswith(k)]
            if match:
                total += float(row['launches_per_step']) * sum(per_kernel[match[0]]) / len(per_kernel[match[0]])
                launches += float(row['launches_per_step'])
    calls = 16.0                                                      # QuantConv2d layers per forward
    return {'bytes_per_launch': total / calls, 'source': source} if launches else None


def cifar_arch():
    """arch_config of examples/cifar100/cifar100_ls1_kd.yaml (model section): 18-layer XNOR ResNet, 3x3 stem, no
    max-pool, ls-1 weights AND activations, clamp alpha = 2."""
    a = imagenet_arch('ls-1', 2)
    a['nonlins'] = ['relu', 'relu']
    a['layer0'] = {'n_in_channels': 64, 'kernel_size': 3, 'stride': 1, 'padding': 1, 'bias': False,
                   'maxpool': {'type': 'identity'}}
    a['output_classes'] = 100
    return a


def build_lenet(device):
    """examples/mnist/mnist_ls1_weight_fp_activation.yaml (BASELINE.json configs[0]): ls-1 weights, fp activations."""
    from quant.models.lenet import QLeNet5
    torch.manual_seed(0)
    model = QLeNet5(loss_fn=torch.nn.functional.nll_loss, x_quant='fp', w_quant='ls-1', clamp={'kind': 'identity'},
                    conv1_filters=20, conv2_filters=50, output_classes=10)
    with torch.no_grad():
        model.conv2.w_approximate.v1.copy_(model.conv2.weight.abs().mean(dim=(1, 2, 3)))
    return model.eval().to(device)


MFMA_I8_PEAK_T = 5000.0      # TOP/s dense int8 = 2 x the bf16 MFMA peak (MI355X_MICROARCH.md)
MFMA_FP4_PEAK_T = 10000.0    # TOP/s dense fp4 / fp6 (block-scaled MFMA, MI355X_MICROARCH.md: ~10 PF dense, 9099 TF measured)
POPCOUNT_PEAK_T = 1258.0      # 256 CU x 128 lanes x 2.4 GHz, v_xor + v_bcnt per 32 binary MACs
HBM_PEAK_GBPS = 8000.0
MFMA_BF16_PEAK_T = 2500.0
PATH_ROOFLINE_IMG_S = {'ls-2': 628e3, 'ls-T': 628e3, 'ls-1': 628e3, 'gf-2': 628e3, 'fp': 628e3}   # 8 TB/s / 12.74 MB


def pmc_mfma_busy(entry, shapes, batch):
    """Counter-based matrix-core utilisation of ``entry``'s kernels from the committed SQ counter passes
    (scripts/capture_profiles.sh -> profiles/*_pmc_sq.json: SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES) per
    layer shape, each kernel alone at batch 256, weighted by the shape's share of the forward).  ``shapes``: the layer
    shape tags the instrumented step of THIS leg launched through ``entry``.  The figure is attached only to a leg that
    launched exactly the profiled shapes at the profiled batch (the ImageNet ResNet-18 legs); None for any other
    workload (LeNet, CIFAR ...), {'stale': ...} when the kernel sources changed after the capture."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_pmc_sq.json')))
    if not files or not shapes:
        return None
    doc = json.load(open(files[-1]))
    rows = [r for r in doc.get('kernels', []) if r.get('entry') == entry]
    if not rows or batch != 256 or set(shapes) != {r['shape'] for r in rows}:
        return None
    source = '%s (git blob %s)' % (os.path.basename(files[-1]), git_blob_hash(files[-1])[:12])
    if not profile_is_current(doc):
        return {'stale': source + ' was captured on other kernel sources than this tree\'s (csrc fingerprint differs)'}
    w = sum(r['count_in_forward'] * r['busy_cu_cycles'] for r in rows)
    busy = sum(r['count_in_forward'] * r['busy_cu_cycles'] * r['mfma_busy_frac'] for r in rows) / max(w, 1e-30)
    return {'mfma_busy_frac': busy, 'source': source,
            'per_shape': {r['shape']: round(r['mfma_busy_frac'], 4) for r in rows}}


def kernel_roofline(name, launches, ms, nbytes, ops, survey_bytes=None, shapes=None, batch=None):
    """The roofline entry of one path kernel: the bound SURVEY 8(d) assigns to it (quantizer: HBM; XNOR conv:
    VALU popcount with HBM second; sign-weight conv: bf16 MFMA with both passes counted).  The HBM view of the
    convolutions comes in both accountings: `frac` counts every operand the call must move once (the residual
    operands of the fused epilogue included), `frac_survey_8d` SURVEY 8(d)'s input-once + output-once bytes."""
    sec = ms * 1e-3
    hbm = {'bound': 'hbm', 'achieved': nbytes / sec / 1e9, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
           'frac': nbytes / sec / 1e9 / HBM_PEAK_GBPS,
           'bytes': 'input read once + output written once + residual operands of the fused epilogue read once'}
    if survey_bytes is not None:
        hbm['achieved_survey_8d'] = survey_bytes / sec / 1e9
        hbm['frac_survey_8d'] = survey_bytes / sec / 1e9 / HBM_PEAK_GBPS
    if name == 'lsq_xnor_conv2d':
        # 3x3 layers over 64..512 channels run on v_mfma_scale_f32_32x32x64_f8f6f4 with fp4 operands (csrc/lsq_xnor_mfma.hip, round 6;
        # int8 MFMA in rounds 2-5): priced against the dense fp4 peak, 2 ops per binary MAC; the popcount kernel (other geometries)
        # against the same peak
        t = 2.0 * ops / sec / 1e12
        r = {'bound': 'mfma', 'achieved': t, 'peak': MFMA_FP4_PEAK_T, 'unit': 'TOP/s', 'frac': t / MFMA_FP4_PEAK_T,
             'frac_of_int8_peak': t / MFMA_I8_PEAK_T,
             'note': 'fp4 (MX, unit scales) MFMA, dense peak 10 POP/s = 2 x int8 (the guide\'s micro-benchmark reaches 9099); achieved = 2 x '
                     'binary MACs; rounds 2-5 priced the int8 kernel against 5 POP/s (frac_of_int8_peak); SURVEY 8(d) popcount figure: '
                     '%.0f T binary-MAC/s = %.2f of 1258' % (t / 2, t / 2 / POPCOUNT_PEAK_T),
             'secondary': hbm}
    elif name == 'lsq_signw_conv2d':
        t = ops / sec / 1e12
        r = {'bound': 'mfma', 'achieved': t, 'peak': MFMA_BF16_PEAK_T, 'unit': 'TFLOP/s', 'frac': t / MFMA_BF16_PEAK_T,
             'note': 'bf16 hi + lo passes both counted (useful fraction = half)', 'secondary': hbm}
    else:
        r = hbm
    if r.get('bound') == 'mfma':
        busy = pmc_mfma_busy(name, shapes, batch)
        r['mfma_busy_frac'] = None                          # (null unless a current profile of exactly these launches exists)
        if busy and 'stale' in busy:
            r['mfma_busy_note'] = busy['stale']
        elif busy:
            r['mfma_busy_frac'] = busy['mfma_busy_frac']
            r['mfma_busy_note'] = ('SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES) of the kernel on its own at the same layer '
                                   'shapes and batch, rocprofv3 --pmc passes in profiles/' + busy['source'] + ', per layer shape: '
                                   + json.dumps(busy['per_shape']))
        else:
            r['mfma_busy_note'] = 'no counter profile of this workload\'s launches is committed'
    r.update(kernel=name, launches=launches, avg_launch_us=1e3 * ms / max(launches, 1))
    return r


def timed_forward(fn, steps, warmup, chunks=10):
    """(images-independent) wall seconds for `steps` calls of fn bracketed by synchronize, plus the per-step
    minimum / median over `chunks` event-bracketed groups of steps (events only between groups)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    per = max(1, steps // chunks)
    groups = [per] * (steps // per) + ([steps % per] if steps % per else [])
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(groups) + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for i, n in enumerate(groups):
        for _ in range(n):
            fn()
        evs[i + 1].record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ms = sorted(evs[i].elapsed_time(evs[i + 1]) / n for i, n in enumerate(groups))
    return elapsed, ms[0], ms[len(ms) // 2]


def shapes_of(by_shape, name):
    return sorted(tag for (k, tag) in by_shape if k == name and tag is not None)


def config_leg(tag, model, x, steps, warmup, workload, cpu_reference_img_s=None, graph=False):
    """One of the other single-GPU configs of BASELINE.json as a short leg of the same process.  `value` is ALWAYS the
    eager forward (what the reference's eager PyTorch figures compare with).  ``graph``: the launch-bound configurations
    (small images / batches: tens of microseconds of GPU work per launch) are ALSO timed as one HIP-graph replay per
    forward (quant/common/graph_replay.py), reported under `graph_replay` with the copy of the input into the graph's
    static buffer inside the timed call -- what a serving caller pays."""
    from quant import _hip
    with torch.no_grad():
        fn = lambda: model(x)      # noqa: E731
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        _hip.enable_timing(True)
        fn()
        torch.cuda.synchronize()
        by_shape = _hip.drain_timing(by_tag=True)
        _hip.enable_timing(False)
        elapsed, ms_min, ms_med = timed_forward(fn, steps, 2)
    table = {}
    for (name, _tag), v in by_shape.items():
        table[name] = tuple(a + b for a, b in zip(table.get(name, (0, 0.0, 0, 0, 0)), v))
    path = {k: v for k, v in table.items() if k in ('lsq_act_quant', 'lsq_xnor_conv2d', 'lsq_signw_conv2d')}
    out = {'workload': workload, 'batch': x.shape[0], 'steps': steps, 'value': x.shape[0] * steps / elapsed, 'unit': 'images/sec',
           'ms_per_step': 1e3 * elapsed / steps, 'ms_per_step_min': ms_min, 'ms_per_step_median': ms_med, 'launch': 'eager'}
    if graph:
        from quant.common.graph_replay import GraphedForward
        with torch.no_grad():
            eager = model(x).clone()
        fwd = GraphedForward(model, x)
        src = x.clone()                                           # a caller's buffer: every timed call copies it in
        same = bool(torch.equal(fwd(src), eager))
        g_elapsed, g_min, g_med = timed_forward(lambda: fwd(src), steps, 2)
        out['graph_replay'] = {'value': x.shape[0] * steps / g_elapsed, 'unit': 'images/sec', 'ms_per_step': 1e3 * g_elapsed / steps,
                               'ms_per_step_min': g_min, 'ms_per_step_median': g_med, 'output_equals_eager': same,
                               'includes': 'device-to-device copy of the input into the graph\'s static buffer + one graph launch'}
    if path:
        dom = max(path, key=lambda k: path[k][1])
        out['roofline'] = kernel_roofline(dom, *path[dom], shapes=shapes_of(by_shape, dom), batch=int(x.shape[0]))
        out['roofline']['measured'] = 'HIP events around every C-ABI call of one instrumented step'
        out['roofline']['traffic'] = None                         # (no counter pass over this leg's workload is committed)
        out['kernels_ms_per_step'] = {k: round(v[1], 4) for k, v in table.items()}
    if cpu_reference_img_s is not None:
        out['reference_cpu_images_per_sec_survey'] = cpu_reference_img_s
    return out


def cpu_baseline(arch, model, sample):
    """The oracle's whole-network forward (same algorithmic structure as the reference: sort + cumsum +
    mask + [N,K,M] cost + fp32 conv; scripts/cpu_oracle_vs_reference.py times the two side by side in the build
    container) on this box's host cores, ONE batch of `sample` images as SURVEY 8(d) states (B = 64).  Thread count:
    the faster of 16 and 32 on a two-image probe -- torch's default of half the logical CPUs (128 here) is five times
    SLOWER on this workload (scripts/cpu_threads.py: 29.5 images/s at 16 threads, 6.0 at 128)."""
    from oracle import ref_models
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(0)
    x = torch.randn(sample, 3, 224, 224, generator=g)
    before = torch.get_num_threads()
    try:
        with torch.no_grad():
            probe = {}
            for nt in sorted({min(16, os.cpu_count() or 1), min(32, os.cpu_count() or 1)}):
                torch.set_num_threads(nt)
                ref_models.resnet_forward(sd, arch, x[:2])              # warm-up (thread pools, allocator)
                t0 = time.perf_counter()
                ref_models.resnet_forward(sd, arch, x[:4])
                probe[nt] = time.perf_counter() - t0
            threads = min(probe, key=probe.get)
            torch.set_num_threads(threads)
            t0 = time.perf_counter()
            ref_models.resnet_forward(sd, arch, x)
            dt = time.perf_counter() - t0
    finally:
        torch.set_num_threads(before)
    return {'value': sample / dt, 'unit': 'images/sec', 'cores': threads, 'kind': 'port',
            'thread_probe_seconds_per_4_images': {str(k): round(v, 4) for k, v in sorted(probe.items())},
            'host_cpu_count': os.cpu_count(),
            'sample': f'one eval forward of a batch of {sample} images of the same workload ({dt:.1f} s) on {threads} threads '
                      f'(the faster of 16 / 32), host cpu_count={os.cpu_count()}'}


class _Clock:
    """Stream events on the GPU, host clock in the CPU plumbing mode (--device cpu: the gloo / launcher test)."""

    def __init__(self, cuda):
        self.cuda = cuda
        if cuda:
            self.ev = torch.cuda.Event(enable_timing=True)
            self.ev.record()
        else:
            self.t = time.perf_counter()

    def ms_until(self, other):
        return self.ev.elapsed_time(other.ev) if self.cuda else 1e3 * (other.t - self.t)


def parity_probe(device):
    """A live check that rides along with the numbers: one eval-mode QuantConv2d (ls-2 x ls-1, 64 -> 64, 3 x 3) on the kernels
    against the CPU oracle -- v1 of the free-running solve bit-equal to the exact oracle, and with the GPU's scales injected
    into the oracle the convolution within north_star's 1e-4 of max|y|.  The free-running deviation from the REFERENCE (its
    fp32 argmin's tie-break) is the committed derivation's, quoted with its source."""
    import glob
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    import detgen
    from oracle import lsq_exact, ref_port
    from quant.binary.binary_conv import QuantConv2d
    clamp = {'kind': 'symmetric', 'alpha': 3}
    conv = QuantConv2d('ls-2', 'ls-1', 64, 64, 3, clamp, padding=1, bias=True)
    detgen.fill_module(conv, seed=11)
    x = detgen.normal('smoke.x', (4, 64, 14, 14), scale=1.2)
    with torch.no_grad():
        conv.w_approximate.v1.copy_(ref_port.weight_scales(conv.weight, 'ls-1')[0])
        conv.eval().to(device)
        y = conv(x.to(device)).cpu()
        v = conv.last_act_scales.clone().cpu()
        y_ref = ref_port.quant_conv2d(x, conv.weight.detach().cpu(), conv.bias.detach().cpu(), 'ls-2', 'ls-1',
                                      [conv.w_approximate.v1.cpu()], clamp, 1, 1, x_scales=[v[0], v[1]])
    err = float((y - y_ref).abs().max() / y_ref.abs().max())
    exact = bool(np.array_equal(v[0].numpy(), lsq_exact.solve_rows(x.clamp(-3, 3).numpy(), False, 3)))
    out = {'injected_scales': {'bound': 1e-4, 'observed_max_rel_err': err, 'met': err < 1e-4},
           'solver_v1_equals_exact_oracle': exact}
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_free_running_parity.json')))
    if files:
        doc = json.load(open(files[-1]))
        keys = ('conv_fixture', 'resnet_logits', 'fused_logits')          # one layer / whole network (ls-T: the worst) / fused blocks
        out['free_running_vs_reference'] = {
            'observed_of_max_abs': {k: round(doc['observed_max'][k], 6) for k in keys if k in doc.get('observed_max', {})},
            'derived_limit': {k: round(doc['limits'][k], 6) for k in keys if k in doc.get('limits', {})},
            'note': 'the reference\'s fp32 argmin decides near-tied candidates by rounding; the kernels equal the exact-arithmetic '
                    'oracle bit for bit; limits derived on the CPU (tests/golden/make_free_limits.py)',
            'source': 'profiles/' + os.path.basename(files[-1])}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--min-seconds', type=float, default=5.0,
                    help='repeat the bracket of --steps timed steps until this much time has been timed (headline region)')
    ap.add_argument('--batch', type=int, default=256, help='images per GPU per step')
    ap.add_argument('--cpu-sample', type=int, default=64,
                    help='batch of the cpu_baseline leg (0 = skip); 64 = the bounded sample SURVEY 8(d) names (B = 64)')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-configs', action='store_true', help='skip the legs of the other single-GPU configs')
    ap.add_argument('--act', default='ls-2', choices=['ls-1', 'ls-2', 'ls-T', 'gf-2', 'fp'],
                    help='activation scheme (default: the headline ls-2 config)')
    ap.add_argument('--device', default='cuda', choices=['cuda', 'cpu'],
                    help='cpu: plumbing check of the launcher and the collective (gloo, torch formulation of the model); never a benchmark')
    ap.add_argument('--image-size', type=int, default=224, help='(plumbing checks only; the metric is quoted at 224)')
    ap.add_argument('--no-pin', action='store_true', help='do not pin the rank to the cores of its GPU\'s NUMA node')
    ap.add_argument('--streams', type=int, default=2,
                    help='HIP streams consecutive steps alternate between (quant.common.stream_pipeline, the product\'s evaluate path); '
                         '1: one stream, every kernel alone on the chip -- what profiled runs pass')
    ap.add_argument('--detail', default=None, help='write the per-kernel / per-layer-shape tables to this JSON file')
    args = ap.parse_args()

    from quant.common import rank_launcher
    if args.gpus > 1 and 'RANK' not in os.environ:
        # started without a launcher: be the launcher (one child per GPU; rank 0's stdout is ours)
        code = rank_launcher.spawn_ranks([os.path.abspath(__file__)] + sys.argv[1:], args.gpus)
        sys.exit(code)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:                         # the environment is what the process group will see: say so and go on
        print(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; running with {world} rank(s)', file=sys.stderr)
    cuda = args.device == 'cuda'
    launcher = 'self' if os.environ.get(rank_launcher.ENV_MARK) else ('torchrun' if 'RANK' in os.environ else 'none')
    pinned = None
    if cuda:
        if local >= torch.cuda.device_count():
            sys.exit(f'bench.py: rank {rank} wants GPU {local} but {torch.cuda.device_count()} GPU(s) are visible')
        torch.cuda.set_device(local)
        device = torch.device('cuda', local)
        if world > 1 and not args.no_pin:
            pinned = rank_launcher.pin_to_gpu_numa(local)
    else:
        device = torch.device('cpu')
        torch.set_num_threads(max(1, min(4, (os.cpu_count() or 1) // max(world, 1))))
    sync = torch.cuda.synchronize if cuda else (lambda: None)
    if 'RANK' in os.environ:                       # launched as one of several ranks (also with one rank): process group
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # RCCL prints a version banner to the process's stdout when its first communicator comes up; stdout is for the
        # ONE JSON line, so file descriptor 1 points at stderr while the group initialises
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            if cuda:
                dist.init_process_group('nccl', device_id=device)
            else:
                dist.init_process_group('gloo')
            dist.barrier()
            sync()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
        world = dist.get_world_size()

    from quant import _hip
    from quant.common.sharded_eval import all_gather_logits, local_forward
    from quant.common.stream_pipeline import StreamPipeline
    arch = imagenet_arch(args.act, 3 if args.act == 'ls-2' else 2)
    model = build_model(arch, device)
    g = torch.Generator(device='cpu').manual_seed(rank)
    x = torch.randn(args.batch, 3, args.image_size, args.image_size, generator=g).to(device)       # resident in HBM before timing
    gathered = torch.empty((world * args.batch, 1000), dtype=torch.float32, device=device) if dist.is_initialized() else None
    in_group = dist.is_initialized()
    roofline = cuda and not args.no_roofline
    nstreams = args.streams if cuda else 1
    PATH_KERNELS = ('lsq_act_quant', 'lsq_xnor_conv2d', 'lsq_signw_conv2d')

    def bracket(run, min_seconds):
        """Repetitions of EXACTLY args.steps steps, each between barrier + synchronize on both sides, until at least
        `min_seconds` have been timed; (seconds, repetitions, ms per step of every repetition), max over ranks per bracket."""
        elapsed, reps, rep_ms = 0.0, 0, []
        while True:
            if world > 1:
                dist.barrier()
            sync()
            t0 = time.perf_counter()
            run(args.steps)
            if world > 1:
                dist.barrier()
            sync()
            dt = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([dt], dtype=torch.float64, device=device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())                           # (every rank sees the same dt: same number of repetitions)
            elapsed += dt
            reps += 1
            rep_ms.append(1e3 * dt / args.steps)
            if elapsed >= min_seconds or reps >= 10000:
                return elapsed, reps, rep_ms

    # ---- the step as the product's evaluation loop issues it (quant.common.training.evaluate): the local forward of step i
    # is queued on one of `nstreams` HIP streams (round robin) while the logits of step i - 1 are all-gathered on the main
    # stream behind that forward's event.  nstreams = 1: everything on the one stream, every kernel alone on the chip.
    pipe = StreamPipeline(lambda shard: local_forward(model, shard), device, nstreams)

    def run_steps(n):
        window = []

        def consume():
            y = window.pop(0).result()
            if in_group:
                all_gather_logits(y, gathered, always_collective=True)

        for k in range(n):
            window.append(pipe.submit(x))
            if len(window) >= pipe.depth:
                consume()
        while window:
            consume()

    with torch.no_grad():
        run_steps(max(args.warmup, 2 * nstreams))
        sync()
        elapsed, reps, rep_ms = bracket(run_steps, args.min_seconds)
    steps_timed = reps * args.steps

    # ---- one stream: the reference point of rounds 1-4 and the region the per-kernel figures come from (each kernel alone on
    # the chip, as in the rocprofv3 summaries under profiles/).  HIP events sit around the launches of the two path kernels in
    # the first step of every group of steps only (an event pair costs a few microseconds of stream time).
    single = None
    table, by_shape, timed_table = {}, {}, {}
    if nstreams > 1 or roofline or world > 1:            # (N > 1: the line always carries the one-stream figure beside the pipelined one)
        one = StreamPipeline(lambda shard: local_forward(model, shard), device, 1)

        def run_single(n, sample_every=0):
            for k in range(n):
                if sample_every:
                    _hip.pause_timing(k % sample_every != 0)
                y = one.submit(x).result()
                if in_group:
                    all_gather_logits(y, gathered, always_collective=True)
            if sample_every:
                _hip.pause_timing(False)

        with torch.no_grad():
            run_single(max(3, args.warmup // 2))
            if roofline:
                sync()
                _hip.enable_timing(True)
                run_single(1)                                     # one fully instrumented (untimed) step: the per-kernel table
                sync()
                by_shape = _hip.drain_timing(by_tag=True)
                for (name, _tag), v in by_shape.items():
                    table[name] = tuple(a + b for a, b in zip(table.get(name, (0, 0.0, 0, 0, 0)), v))
                _hip.enable_timing(True, only=[k for k in PATH_KERNELS if k in table])
            # the one-stream figure is timed on an UNINSTRUMENTED bracket (no event pairs inside: the same footing as the
            # headline region); the per-kernel events then come from a separate, untimed run of kSampleSteps steps in which
            # every kSampleEvery-th step carries them -- a fixed rate, whatever --steps is (round 5 tied it to --steps / 10: under
            # the driver's --steps 20 every second step of the timed region carried 32 event pairs)
            _hip.pause_timing(True)
            s_elapsed, s_reps, s_rep_ms = bracket(lambda n: run_single(n, 0), min(args.min_seconds, 2.0))
            _hip.pause_timing(False)
            if roofline:
                kSampleEvery, kSampleSteps = 20, 100
                run_single(kSampleSteps, kSampleEvery)
                sync()
                timed_table = _hip.drain_timing()
                _hip.enable_timing(False)
        single = {'value': world * args.batch * s_reps * args.steps / s_elapsed, 'unit': 'images/sec', 'steps_timed': s_reps * args.steps,
                  'timed_seconds': s_elapsed, 'ms_per_step': 1e3 * s_elapsed / (s_reps * args.steps),
                  'ms_per_step_best_repetition': min(s_rep_ms)}

    allgather = None
    if in_group:
        # the exchange step alone: [batch, 1000] fp32 logits per rank, events on the launch stream
        logits = torch.randn(args.batch, 1000, device=device)
        for _ in range(5):
            all_gather_logits(logits, gathered, always_collective=True)
        sync()
        ag_reps = 50
        s_ev = _Clock(cuda)
        for _ in range(ag_reps):
            all_gather_logits(logits, gathered, always_collective=True)
        e_ev = _Clock(cuda)
        sync()
        us = 1e3 * s_ev.ms_until(e_ev) / ag_reps
        t = torch.tensor([us], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        us = float(t.item())
        recv = (world - 1) * logits.numel() * 4                   # bytes every rank receives per all-gather
        allgather = {'allgather_us': us, 'bytes_per_rank': logits.numel() * 4, 'bytes_received_per_rank': recv,
                     'allgather_GBps': recv / (us * 1e-6) / 1e9 if world > 1 else 0.0,
                     'per_link_GBps': recv / (us * 1e-6) / 1e9 / max(world - 1, 1) if world > 1 else 0.0,
                     'backend': dist.get_backend(), 'world_size_seen_by_backend': dist.get_world_size(),
                     'note': ('RCCL' if cuda else 'gloo') + ' all_gather of fp32 logits, max over ranks, mean of 50 back-to-back calls'}

    if rank == 0:
        value = world * args.batch * steps_timed / elapsed
        out = {
            'metric': 'images/sec ResNet-18 LS-1w/LS-2a 224x224 eval forward' if args.act == 'ls-2' else
                      f'images/sec ResNet-18 ls-1w/{args.act}-a 224x224 eval forward',
            'value': value, 'unit': 'images/sec',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'steps_timed': steps_timed, 'repetitions': reps, 'timed_seconds': elapsed,
            'ms_per_step': 1e3 * elapsed / steps_timed, 'ms_per_step_best_repetition': min(rep_ms),
            'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'bf16 mfma (hi+lo split) + f32' if args.act == 'fp' else 'fp4 (e2m1, unit block scales) mfma on sign bits: exact integers in the f32 accumulator + f32', 'data': 'synthetic',
            'config': {'workload': f'ResNet-18 ImageNet ls-1 weight / {args.act} activation, '
                                   f'synthetic 3x{args.image_size}x{args.image_size}, batch {args.batch} per GPU, random-init weights',
                       'global_batch': world * args.batch,
                       'parallelism': f'dp{world} (batch-sharded replicas, ' + ('RCCL' if cuda else 'gloo') + ' all-gather of logits)',
                       'issue': (f'consecutive steps alternate between {nstreams} HIP streams, as quant.common.training.evaluate issues '
                                 'consecutive batches; logits bit-identical to the one-stream forward') if nstreams > 1 else 'one HIP stream'},
            'launcher': {'kind': launcher, 'ranks': world, 'numa_pinning_rank0': pinned},
            'path_frac': value / world / PATH_ROOFLINE_IMG_S[args.act],
        }
        if nstreams > 1 or world > 1:
            out['pipelined'] = {'streams': nstreams, 'value': value, 'ms_per_step': out['ms_per_step'],
                                'vs_single_stream': value / single['value'] if single else None}
        out['value_definition'] = ('whole-job images/s of the headline region: consecutive steps on %d HIP stream(s), the product\'s evaluate '
                                   'path (rounds 1-4 reported the one-stream step: `single_stream`, timed on the same footing -- '
                                   'uninstrumented bracket of --steps steps, barrier + synchronize on both sides)' % nstreams)
        if single is not None:
            out['single_stream'] = single
        if not cuda:
            out.update(device='cpu', dtype='f32 (torch formulation)',
                       note='PLUMBING CHECK of the launcher and the gloo collective on the host -- not a measurement of the HIP path')
        detail = {}
        if roofline and table:
            kern = {}
            for k, v in table.items():                                     # the instrumented step
                tv = timed_table.get(k, v)                                 # path kernels: events over the timed region
                e = kernel_roofline(k, *tv, shapes=shapes_of(by_shape, k), batch=args.batch)
                e['ms_per_step'] = v[1]
                e['launches_per_step'] = v[0]
                e['algorithmic_bytes_per_launch'] = v[2] / max(v[0], 1)
                e['traffic'] = e['traffic_ratio'] = None
                p = pmc_traffic_per_launch(k, args.act) if args.batch == 256 and args.image_size == 224 else None
                if p and 'stale' in p:
                    e['traffic_note'] = p['stale']
                elif p:
                    e['traffic'] = p['bytes_per_launch']
                    e['traffic_ratio'] = p['bytes_per_launch'] / e['algorithmic_bytes_per_launch']
                    e['traffic_note'] = 'HBM bytes per launch, rocprofv3 PMC passes over this workload: profiles/' + p['source']
                kern[k] = e
            names = {'lsq_act_quant': 'quantizer', 'lsq_xnor_conv2d': 'xnor_conv', 'lsq_signw_conv2d': 'signw_conv'}
            path = {names[k]: kern[k] for k in kern if k in names}
            slim = lambda e: {kk: vv for kk, vv in e.items() if kk not in ('note', 'bytes', 'mfma_busy_note', 'secondary', 'traffic_note')}   # noqa: E731
            worst = min(path, key=lambda k: path[k]['frac'])                # the path kernel furthest below ITS roofline
            out['roofline'] = dict(slim(path[worst]), name=worst,
                                   measured='one stream (each kernel alone on the chip, as in the rocprofv3 summaries under profiles/): HIP '
                                            'events around every launch of the kernel in every 20th of 100 one-stream steps run behind the '
                                            '(uninstrumented) single_stream region; roofline = the path kernel with the lower fraction, both follow')
            for k, e in path.items():
                out['roofline'][k] = slim(e)
                if 'secondary' in e:
                    out['roofline'][k]['hbm_frac'] = e['secondary']['frac']
            out['roofline']['other_kernels_ms_per_step'] = {k: round(v[1], 4) for k, v in table.items() if k not in names}
            shapes = {}
            for (name, tag), (cnt, ms, nb, ops, _sb) in sorted(by_shape.items(), key=lambda kv: str(kv[0])):
                if tag is None or name not in PATH_KERNELS:
                    continue
                row = {'launches': cnt, 'avg_launch_us': 1e3 * ms / cnt, 'algorithmic_GBps': nb / (ms * 1e-3) / 1e9}
                if ops:
                    row['T_binary_MAC_per_s' if name == 'lsq_xnor_conv2d' else 'TFLOP_per_s'] = ops / (ms * 1e-3) / 1e12
                shapes.setdefault(name, {})[tag] = row
            detail['kernels'] = kern
            detail['by_layer_shape'] = shapes
        if allgather is not None:
            out['allgather'] = allgather
        if cuda and world == 1:
            try:
                out['parity'] = parity_probe(device)
            except Exception as exc:                                       # (the numbers above stand without it; say what happened)
                out['parity'] = {'error': repr(exc)[:200]}
        if cuda and args.cpu_sample > 0 and world == 1:
            out['cpu_baseline'] = cpu_baseline(arch, model, args.cpu_sample)
        if cuda and world == 1 and not args.no_configs and args.act == 'ls-2':
            # the other single-GPU configurations BASELINE.json lists, as short legs of this process (one stream, eager)
            cfg = {}
            # the north_star-literal __popcll XNOR kernel on the headline network, beside the int8-MFMA one above
            old = _hip.xnor_impl(True)
            try:
                cfg['imagenet_ls1w_ls2a_popcount_kernel_b256'] = config_leg(
                    'popc', model, x, 40, 5, 'the headline network with every XNOR convolution on the popcount kernel '
                    '(v_xor + v_bcnt, csrc/lsq_xnor_conv.hip) instead of the int8-MFMA kernel: same bits out')
            finally:
                _hip.xnor_impl(int(old))
            del model
            m = build_model(imagenet_arch('fp', 2), device)
            cfg['imagenet_ls1w_fpa_b256'] = config_leg('fp', m, x, 40, 5, 'ResNet-18 ImageNet ls-1 weight / fp activation '
                                                       '(bf16 MFMA, hi+lo split), synthetic 3x224x224, batch 256', 80.4)
            del m
            m = build_model(cifar_arch(), device)
            xc = torch.randn(100, 3, 32, 32, generator=torch.Generator().manual_seed(0)).to(device)
            cfg['cifar100_ls1_kd_b100'] = config_leg('cifar', m, xc, 100, 5, 'ResNet-18 CIFAR-100 cifar100_ls1_kd (ls-1 weights and '
                                                     'activations, clamp 2), synthetic 3x32x32, batch 100 (yaml test_batch_size)', 192.6, graph=True)
            del m
            # the headline network in the deployment configuration the reference's release notes motivate:
            # moving-average activation scales (eval_only), i.e. NO scale solve at inference -- both planes in one read
            arch_ma = imagenet_arch('ls-2', 3)
            arch_ma.update(moving_average_mode='eval_only', moving_average_momentum=0.0)
            m = build_model(arch_ma, device)
            m.train()
            with torch.no_grad():
                m(x[:4])                              # one calibration step (torch formulation): the scale buffers take the batch's values
            m.eval()
            cfg['imagenet_ls1w_ls2a_moving_average_b256'] = config_leg(
                'ma', m, x, 40, 5, 'ResNet-18 ImageNet ls-1 weight / ls-2 activation with moving-average (eval_only) activation scales: '
                'the quantizer only packs, synthetic 3x224x224, batch 256')
            del m
            m = build_lenet(device)
            xm = torch.randn(64, 1, 28, 28, generator=torch.Generator().manual_seed(0)).to(device)
            cfg['mnist_lenet_ls1w_fpa_b64'] = config_leg('lenet', m, xm, 200, 5, 'LeNet-5 mnist_ls1_weight_fp_activation, synthetic '
                                                         '1x28x28, batch 64', 19048.0, graph=True)
            detail['configs'] = cfg
            # the line carries the legs' headline figures; their kernel tables go to --detail
            out['configs'] = {k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()
                                  if kk in ('batch', 'value', 'ms_per_step', 'launch', 'reference_cpu_images_per_sec_survey')}
                              for k, v in cfg.items()}
            for k, v in cfg.items():
                if v.get('roofline', {}).get('bound') == 'mfma':           # BASELINE config 3 asks for the MFMA utilisation in the line
                    out['configs'][k]['dominant_kernel'] = v['roofline']['kernel']
                    out['configs'][k]['mfma_frac_of_peak'] = round(v['roofline']['frac'], 4)
                    out['configs'][k]['mfma_busy_frac'] = (None if v['roofline'].get('mfma_busy_frac') is None
                                                           else round(v['roofline']['mfma_busy_frac'], 4))
                if 'graph_replay' in v:
                    out['configs'][k]['graph_replay_value'] = round(v['graph_replay']['value'], 1)
            out['configs_note'] = 'one stream, eager eval forward, images/sec, inputs resident in HBM; full tables: --detail'
        if args.detail and detail:
            detail['line'] = out
            os.makedirs(os.path.dirname(os.path.abspath(args.detail)), exist_ok=True)
            with open(args.detail, 'w') as f:
                json.dump(detail, f, indent=1)
        print(json.dumps(out))
        sys.stdout.flush()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
