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
        # 3x3 layers over 64..512 channels run on v_mfma_i32_32x32x32_i8 (csrc/lsq_xnor_mfma.hip): integer MFMA bound,
        # 2 ops per binary MAC; the popcount kernel (other geometries) is priced against the same peak
        t = 2.0 * ops / sec / 1e12
        r = {'bound': 'mfma', 'achieved': t, 'peak': MFMA_I8_PEAK_T, 'unit': 'TOP/s', 'frac': t / MFMA_I8_PEAK_T,
             'note': 'int8 MFMA, dense peak = 2 x bf16 (the guide\'s micro-benchmark reaches 3944); achieved = 2 x binary MACs; '
                     'SURVEY 8(d) popcount figure: %.0f T binary-MAC/s = %.2f of 1258' % (t / 2, t / 2 / POPCOUNT_PEAK_T),
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--min-seconds', type=float, default=1.0,
                    help='repeat the bracket of --steps timed steps until this much time has been timed')
    ap.add_argument('--batch', type=int, default=256, help='images per GPU per step')
    ap.add_argument('--cpu-sample', type=int, default=256, help='batch of the cpu_baseline leg (0 = skip)')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-configs', action='store_true', help='skip the legs of the other single-GPU configs')
    ap.add_argument('--act', default='ls-2', choices=['ls-1', 'ls-2', 'ls-T', 'gf-2', 'fp'],
                    help='activation scheme (default: the headline ls-2 config)')
    ap.add_argument('--device', default='cuda', choices=['cuda', 'cpu'],
                    help='cpu: plumbing check of the launcher and the collective (gloo, torch formulation of the model); never a benchmark')
    ap.add_argument('--image-size', type=int, default=224, help='(plumbing checks only; the metric is quoted at 224)')
    ap.add_argument('--no-pin', action='store_true', help='do not pin the rank to the cores of its GPU\'s NUMA node')
    ap.add_argument('--streams', type=int, default=2,
                    help='HIP streams consecutive steps alternate between in the second timed region (1: only the single-stream region)')
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
    from quant.common.sharded_eval import all_gather_logits, evaluate_sharded
    arch = imagenet_arch(args.act, 3 if args.act == 'ls-2' else 2)
    model = build_model(arch, device)
    g = torch.Generator(device='cpu').manual_seed(rank)
    x = torch.randn(args.batch, 3, args.image_size, args.image_size, generator=g).to(device)       # resident in HBM before timing
    gathered = torch.empty((world * args.batch, 1000), dtype=torch.float32, device=device) if dist.is_initialized() else None
    in_group = dist.is_initialized()
    roofline = cuda and not args.no_roofline

    def step():
        # local forward + all-gather of logits (issued with one rank too when the process is one of a group)
        return evaluate_sharded(model, x, gathered, always_collective=in_group)

    for _ in range(args.warmup):
        step()
    dominant = None
    if roofline:
        # one fully instrumented (untimed) step finds the dominant C-ABI kernel and the per-kernel table;
        # inside the timed region only that kernel is bracketed with HIP events, and only in the first step of every
        # group of steps (an event pair costs a few microseconds of stream time per call: ~0.3 ms per step if every
        # call carried one, ~0.1 ms if every launch of the dominant kernel did)
        sync()
        _hip.enable_timing(True)
        step()
        sync()
        by_shape = _hip.drain_timing(by_tag=True)
        table = {}
        for (name, _tag), v in by_shape.items():
            table[name] = tuple(a + b for a, b in zip(table.get(name, (0, 0.0, 0, 0, 0)), v))
        candidates = {k: v for k, v in table.items() if k in ('lsq_act_quant', 'lsq_xnor_conv2d', 'lsq_signw_conv2d')}
        dominant = max(candidates, key=lambda k: candidates[k][1])
        _hip.enable_timing(True, only=[dominant])
    if world > 1:
        dist.barrier()
    sync()
    # Repetitions of EXACTLY args.steps timed steps, each bracketed by barrier + synchronize on both sides; as many
    # repetitions as it takes to time at least args.min_seconds (a 20-step bracket is 0.06 s: too short for the clocks
    # and the power state to settle).  value = all timed steps / the sum of the brackets' times (max over ranks per
    # bracket).  Events between groups of steps (none inside a step) give the per-step minimum / median.
    per = max(1, args.steps // 10)
    groups = [per] * (args.steps // per) + ([args.steps % per] if args.steps % per else [])
    elapsed, reps, group_ms, rep_ms = 0.0, 0, [], []
    while True:
        if world > 1:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        marks = [_Clock(cuda)]
        for i, n in enumerate(groups):
            for k in range(n):
                _hip.pause_timing(k != 0)
                step()
            marks.append(_Clock(cuda))
        _hip.pause_timing(False)
        if world > 1:
            dist.barrier()
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())                               # (every rank sees the same dt: same number of repetitions)
        elapsed += dt
        reps += 1
        rep_ms.append(1e3 * dt / args.steps)
        group_ms += [marks[i].ms_until(marks[i + 1]) / n for i, n in enumerate(groups)]
        if elapsed >= args.min_seconds or reps >= 10000:
            break
    group_ms.sort()
    steps_timed = reps * args.steps
    timed_table = _hip.drain_timing() if roofline else {}
    _hip.enable_timing(False)

    # ---- second timed region: the same steps, consecutive ones on alternating HIP streams (quant/common/stream_pipeline.py:
    # what quant.common.training.evaluate does with consecutive batches).  Same bracket: EXACTLY args.steps steps between
    # barrier + synchronize, repeated until args.min_seconds have been timed, max over ranks per bracket.  The all-gather
    # of a step's logits stays on the one main stream, ordered after that step's forward.
    piped = None
    if cuda and args.streams > 1:
        from quant.common.stream_pipeline import StreamPipeline
        pipe = StreamPipeline(model, device, args.streams)

        def run_pipelined(n):
            window = []

            def consume():
                y = window.pop(0).result()
                if in_group:
                    all_gather_logits(y, gathered, always_collective=True)

            for _ in range(n):
                window.append(pipe.submit(x))
                if len(window) >= pipe.depth:
                    consume()
            while window:
                consume()

        with torch.no_grad():
            run_pipelined(max(args.warmup, 2 * args.streams))
            p_elapsed, p_reps, p_rep_ms = 0.0, 0, []
            while True:
                if world > 1:
                    dist.barrier()
                sync()
                t0 = time.perf_counter()
                run_pipelined(args.steps)
                if world > 1:
                    dist.barrier()
                sync()
                dt = time.perf_counter() - t0
                if world > 1:
                    t = torch.tensor([dt], dtype=torch.float64, device=device)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    dt = float(t.item())
                p_elapsed += dt
                p_reps += 1
                p_rep_ms.append(1e3 * dt / args.steps)
                if p_elapsed >= args.min_seconds or p_reps >= 10000:
                    break
        piped = {'streams': args.streams, 'steps_timed': p_reps * args.steps, 'repetitions': p_reps, 'timed_seconds': p_elapsed,
                 'ms_per_step': 1e3 * p_elapsed / (p_reps * args.steps), 'ms_per_step_best_repetition': min(p_rep_ms)}

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
                     'note': ('RCCL' if cuda else 'gloo') + ' all_gather of fp32 logits, max over ranks, mean of 50 back-to-back calls; '
                             'per_link = received bytes / (world - 1) point-to-point xGMI links'}

    if rank == 0:
        value = world * args.batch * steps_timed / elapsed
        out = {
            'metric': 'images/sec ResNet-18 LS-1w/LS-2a 224x224 eval forward' if args.act == 'ls-2' else
                      f'images/sec ResNet-18 ls-1w/{args.act}-a 224x224 eval forward',
            'value': value, 'unit': 'images/sec',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'steps_timed': steps_timed, 'repetitions': reps, 'timed_seconds': elapsed,
            'ms_per_step': 1e3 * elapsed / steps_timed, 'ms_per_step_min': group_ms[0], 'ms_per_step_median': group_ms[len(group_ms) // 2],
            'ms_per_step_best_repetition': min(rep_ms),
            'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'bf16 mfma (hi+lo split) + f32' if args.act == 'fp' else 'i8 mfma on sign bits (exact integers) + f32', 'data': 'synthetic',
            'config': {'workload': f'ResNet-18 ImageNet ls-1 weight / {args.act} activation, '
                                   f'synthetic 3x{args.image_size}x{args.image_size}, batch {args.batch} per GPU, random-init weights',
                       'global_batch': world * args.batch, 'parallelism': f'dp{world} (batch-sharded replicas, '
                                                                          'RCCL all-gather of logits)'},
            'launcher': {'kind': launcher, 'ranks': world, 'numa_pinning_rank0': pinned},
            'path_frac': value / world / PATH_ROOFLINE_IMG_S[args.act],
            'path_frac_note': 'images/s per GPU / 628 k images/s = 8 TB/s over the 12.74 MB per image the 16 QuantConv2d '
                              'layers read and write once (SURVEY 8(d)); popcount roofline 376 k images/s (ls-2)',
        }
        if piped is not None:
            # `value` keeps the definition of rounds 1-3: consecutive steps on ONE stream, every kernel alone on the chip -- what
            # `roofline` and the rocprofv3 summaries under profiles/ describe.  The two-stream region is reported BESIDE it, not
            # instead of it: a reader comparing this line with an earlier round's compares like with like.
            pvalue = world * args.batch * piped['steps_timed'] / piped['timed_seconds']
            piped.update(value=pvalue, unit='images/sec', vs_single_stream=pvalue / value,
                         path_frac=pvalue / world / PATH_ROOFLINE_IMG_S[args.act],
                         note=('the same steps with consecutive ones alternating between %d HIP streams (quant/common/stream_pipeline.py, '
                               'what quant.common.training.evaluate does with consecutive batches): the dispatch ramp and the last tiles '
                               'of one step\'s kernels run under the next step\'s kernels; logits bit-identical to the single-stream '
                               'forward.  NOT the definition of `value` in this or any earlier round' % piped['streams']))
            out['pipelined'] = piped
        if not cuda:
            out.update(device='cpu', dtype='f32 (torch formulation)',
                       note='PLUMBING CHECK of the launcher and the gloo collective on the host -- not a measurement of the HIP path')
            out['config']['parallelism'] = f'dp{world} (batch-sharded replicas, gloo all-gather of logits)'
        if roofline:
            out['roofline'] = kernel_roofline(dominant, *timed_table[dominant], shapes=shapes_of(by_shape, dominant),
                                              batch=args.batch)     # events over the timed region
            launches, nbytes = timed_table[dominant][0], timed_table[dominant][2]
            out['roofline']['measured'] = ('HIP events around every launch of this kernel in the first step of every group of '
                                           '%d steps of the timed region (one stream: the kernel alone on the chip, as in the '
                                           'rocprofv3 summaries under profiles/)' % per)
            out['roofline']['traffic'] = None
            pmc = pmc_traffic_per_launch(dominant, args.act) if args.batch == 256 and args.image_size == 224 else None
            if pmc and 'stale' in pmc:
                out['roofline']['traffic_note'] = pmc['stale']
            elif pmc:
                out['roofline']['traffic'] = pmc['bytes_per_launch']
                out['roofline']['traffic_note'] = ('HBM bytes per launch (all kernels of one call), rocprofv3 PMC passes over this very '
                                                   'workload in profiles/' + pmc['source'] + '; algorithmic bytes per launch = %.4g'
                                                   % (nbytes / launches))
            kern = {}
            for k, v in table.items():                                     # the instrumented step before the timed region
                kern[k] = kernel_roofline(k, *v, shapes=shapes_of(by_shape, k), batch=args.batch)
                kern[k]['ms_per_step'] = v[1]
                kern[k]['launches_per_step'] = v[0]
                p = pmc_traffic_per_launch(k, args.act) if args.batch == 256 and args.image_size == 224 else None
                if p and 'stale' not in p:
                    kern[k]['traffic'] = p['bytes_per_launch']
                    kern[k]['algorithmic_bytes_per_launch'] = v[2] / max(v[0], 1)
            # per layer shape: which launches are output-bound (HBM) and which matrix-core-bound
            shapes = {}
            for (name, tag), (cnt, ms, nb, ops, _sb) in sorted(by_shape.items(), key=lambda kv: str(kv[0])):
                if tag is None or name not in ('lsq_act_quant', 'lsq_xnor_conv2d', 'lsq_signw_conv2d'):
                    continue
                row = {'launches': cnt, 'avg_launch_us': 1e3 * ms / cnt, 'algorithmic_GBps': nb / (ms * 1e-3) / 1e9}
                if ops:
                    row['T_binary_MAC_per_s' if name == 'lsq_xnor_conv2d' else 'TFLOP_per_s'] = ops / (ms * 1e-3) / 1e12
                shapes.setdefault(name, {})[tag] = row
            out['roofline']['kernels'] = kern
            out['roofline']['by_layer_shape'] = shapes
            out['roofline']['kernels_measured'] = 'one fully instrumented step after the warm-up'
        if allgather is not None:
            out['allgather'] = allgather
        if cuda and args.cpu_sample > 0 and world == 1:
            out['cpu_baseline'] = cpu_baseline(arch, model, args.cpu_sample)
        if cuda and world == 1 and not args.no_configs and args.act == 'ls-2':
            # the other single-GPU configurations BASELINE.json lists, as short legs of this process
            cfg = {}
            # the north_star-literal __popcll XNOR kernel on the headline network, beside the int8-MFMA one above
            old = _hip.xnor_impl(True)
            try:
                cfg['imagenet_ls1w_ls2a_popcount_kernel_b256'] = config_leg(
                    'popc', model, x, 40, 5, 'the headline network with every XNOR convolution on the popcount kernel '
                    '(v_xor + v_bcnt, csrc/lsq_xnor_conv.hip) instead of the int8-MFMA kernel: same bits out')
            finally:
                _hip.xnor_impl(bool(old))
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
            out['configs'] = cfg
            out['configs_note'] = ('value = images/sec of the EAGER eval forward, inputs resident in HBM (graph_replay, where present, is a '
                                   'separate figure); reference_cpu_images_per_sec_survey = the reference itself on the 8 '
                                   'build-container cores (SURVEY section 6)')
        print(json.dumps(out))
        sys.stdout.flush()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
