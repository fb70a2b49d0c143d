#!/usr/bin/env python3
"""Headline benchmark: images/sec of ResNet-18 (ImageNet, ls-1 weights / ls-2 activations) eval forward.

    python bench.py --gpus N --steps K --warmup W

One process per GPU (torchrun sets RANK/LOCAL_RANK/WORLD_SIZE); each rank holds a full replica,
takes its own 256 synthetic 3x224x224 images per step (weak scaling) and the ranks all-gather
their logits over RCCL every step.  Rank 0 prints ONE JSON line (see DESIGN.md "Measurement").
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
    """arch_config of examples/imagenet/imagenet_ls1_weight_ls2_activation_kd.yaml (model section)."""
    return {
        'moving_average_mode': 'off', 'moving_average_momentum': 0.99, 'block': 'xnor',
        'layer0': {'n_in_channels': 64, 'kernel_size': 7, 'stride': 2, 'padding': 3, 'bias': False,
                   'maxpool': {'type': 'maxpool2d', 'kernel_size': 3, 'stride': 2, 'padding': 1}},
        'layer1': _layer(x_quant, alpha), 'layer2': _layer(x_quant, alpha),
        'layer3': _layer(x_quant, alpha), 'layer4': _layer(x_quant, alpha),
        'nonlins': ['relu', 'relu'], 'num_blocks': [2, 2, 2, 2], 'output_classes': 1000}


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


def pmc_traffic_per_launch(entry):
    """HBM bytes per C-ABI launch of ``entry`` from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE in separate passes, FETCH_SIZE doubled on gfx950: scripts/pmc_traffic.sh ->
    profiles/*_pmc_hbm_traffic.json), weighted by the kernel launches per step of the committed kernel
    trace (profiles/*_per_step_summary*.csv).  Counters cannot be read from inside this process, so this is
    the last profiled build's figure; None when the profiles are absent."""
    import csv
    import glob
    prefix = {'lsq_act_quant': 'aq_', 'lsq_xnor_conv2d': 'xnor_conv_kernel', 'lsq_signw_conv2d': 'signw_conv_'}.get(entry)
    tables = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_hbm_traffic.json')))
    steps = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_per_step_summary%s.csv' % ('_fpact' if entry == 'lsq_signw_conv2d' else ''))))
    if not prefix or not tables or not steps:
        return None
    per_kernel = {}
    for name, v in json.load(open(tables[-1]))['kernels'].items():       # rows: "<kernel> grid=<threads>"
        if name.startswith(prefix):
            per_kernel.setdefault(name.split(' grid=')[0], []).append(1e6 * (v['hbm_read_MB_corrected'] + v['hbm_write_MB']))
    total = launches = 0.0
    with open(steps[-1]) as f:
        next(f)
        for row in csv.DictReader(f):
            kname = row['kernel']
            match = [k for k in per_kernel if kname.startswith(k[:len(kname)]) or k.startswith(kname.split('(')[0])]
            if match:
                total += float(row['launches_per_step']) * sum(per_kernel[match[0]]) / len(per_kernel[match[0]])
                launches += float(row['launches_per_step'])
    calls = 16.0                                                      # QuantConv2d layers per forward
    return {'bytes_per_launch': total / calls, 'source': os.path.basename(tables[-1])} if launches else None


def cpu_baseline(arch, model, sample):
    """The oracle's whole-network forward (same algorithmic structure as the reference: sort + cumsum +
    mask + [N,K,M] cost + fp32 conv) timed on this box's host cores on a bounded sample."""
    from oracle import ref_models
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(0)
    x = torch.randn(sample, 3, 224, 224, generator=g)
    with torch.no_grad():
        ref_models.resnet_forward(sd, arch, x[:2], chunk=16)           # warm-up (thread pools, allocator)
        t0 = time.perf_counter()
        ref_models.resnet_forward(sd, arch, x, chunk=16)
        dt = time.perf_counter() - t0
    return {'value': sample / dt, 'unit': 'images/sec', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'{sample} images of the same workload, one forward ({dt:.1f} s), '
                      f'host cpu_count={os.cpu_count()}'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=256, help='images per GPU per step')
    ap.add_argument('--cpu-sample', type=int, default=48, help='images for the cpu_baseline leg (0 = skip)')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--act', default='ls-2', choices=['ls-1', 'ls-2', 'ls-T', 'gf-2', 'fp'],
                    help='activation scheme (default: the headline ls-2 config)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    if 'RANK' in os.environ:                       # launched by torchrun (also with one rank): RCCL process group
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=device)
        dist.barrier()

    from quant import _hip
    arch = imagenet_arch(args.act, 3 if args.act == 'ls-2' else 2)
    model = build_model(arch, device)
    g = torch.Generator(device='cpu').manual_seed(rank)
    x = torch.randn(args.batch, 3, 224, 224, generator=g).to(device)       # resident in HBM before timing
    gathered = torch.empty((world * args.batch, 1000), dtype=torch.float32, device=device) if world > 1 else None

    from quant.common.sharded_eval import evaluate_sharded

    def step():
        return evaluate_sharded(model, x, gathered)       # local forward + RCCL all-gather of logits

    for _ in range(args.warmup):
        step()
    dominant = None
    if not args.no_roofline:
        # one fully instrumented (untimed) step finds the dominant C-ABI kernel and the per-kernel table;
        # inside the timed region only that kernel is bracketed with HIP events (an event pair costs a few
        # microseconds of stream time per call: ~0.3 ms per step if every call carried one)
        torch.cuda.synchronize()
        _hip.enable_timing(True)
        step()
        torch.cuda.synchronize()
        table = _hip.drain_timing()
        candidates = {k: v for k, v in table.items() if k in ('lsq_act_quant', 'lsq_xnor_conv2d', 'lsq_signw_conv2d')}
        dominant = max(candidates, key=lambda k: candidates[k][1])
        _hip.enable_timing(True, only=[dominant])
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        out = {
            'metric': 'images/sec ResNet-18 LS-1w/LS-2a 224x224 eval forward' if args.act == 'ls-2' else
                      f'images/sec ResNet-18 ls-1w/{args.act}-a 224x224 eval forward',
            'value': world * args.batch * args.steps / elapsed, 'unit': 'images/sec',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'bf16 mfma (hi+lo split) + f32' if args.act == 'fp' else 'u64 popcount + f32', 'data': 'synthetic',
            'config': {'workload': f'ResNet-18 ImageNet ls-1 weight / {args.act} activation, '
                                   f'synthetic 3x224x224, batch {args.batch} per GPU, random-init weights',
                       'global_batch': world * args.batch, 'parallelism': f'dp{world} (batch-sharded replicas, '
                                                                          'RCCL all-gather of logits)'},
        }
        if not args.no_roofline:
            launches, ms, nbytes, ops = _hip.drain_timing()[dominant]      # events over the timed region
            kern = {}
            for k, v in table.items():                                     # the instrumented step before it
                kern[k] = {'launches_per_step': v[0], 'ms_per_step': v[1], 'algorithmic_GBps': v[2] / (v[1] * 1e-3) / 1e9}
                if k == 'lsq_xnor_conv2d':      # VALU popcount: 2 ops / 32 MACs; v_bcnt_u32_b32 measured at half rate
                    kern[k]['T_binary_MAC_per_s'] = v[3] / (v[1] * 1e-3) / 1e12
                    kern[k]['frac_of_valu_popcount_peak_1258T'] = kern[k]['T_binary_MAC_per_s'] / 1258.0
                if k == 'lsq_signw_conv2d':
                    kern[k]['TFLOPs_bf16'] = v[3] / (v[1] * 1e-3) / 1e12
            if dominant == 'lsq_signw_conv2d':
                achieved = ops / (ms * 1e-3) / 1e12
                out['roofline'] = {'bound': 'mfma', 'kernel': dominant, 'achieved': achieved, 'peak': 2500.0,
                                   'unit': 'TFLOP/s', 'frac': achieved / 2500.0, 'traffic': None}
            else:
                achieved = nbytes / (ms * 1e-3) / 1e9
                out['roofline'] = {'bound': 'hbm', 'kernel': dominant, 'achieved': achieved, 'peak': 8000.0,
                                   'unit': 'GB/s', 'frac': achieved / 8000.0, 'traffic': None}
                if dominant == 'lsq_xnor_conv2d':
                    out['roofline']['T_binary_MAC_per_s'] = ops / (ms * 1e-3) / 1e12
            pmc = pmc_traffic_per_launch(dominant)
            if pmc:
                out['roofline']['traffic'] = pmc['bytes_per_launch']
                out['roofline']['traffic_note'] = ('HBM bytes per launch (all kernels of one call), rocprofv3 PMC passes in profiles/'
                                                   + pmc['source'] + '; algorithmic bytes per launch = %.4g' % (nbytes / launches))
            out['roofline'].update(launches=launches, avg_launch_us=1e3 * ms / launches,
                                   measured='HIP events around every launch of this kernel inside the timed region',
                                   kernels=kern, kernels_measured='one fully instrumented step after the warm-up')
        if args.cpu_sample > 0 and world == 1:
            out['cpu_baseline'] = cpu_baseline(arch, model, args.cpu_sample)
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
