#!/usr/bin/env python3
"""Build-container check for bench.py's cpu_baseline leg: the oracle's eval forward
(oracle/ref_models.py, what the GPU box times because the reference cannot travel) against the REFERENCE's own
forward (imported from /root/reference, this container only) on the same model and batch -- same logits, and
wall time within +-10 %.

    python scripts/cpu_oracle_vs_reference.py [--batch 64] [--reps 3] > profiles/rNN_cpu_oracle_vs_reference.txt
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path[:0] = [ROOT]
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--reference', default='/root/reference')
    args = ap.parse_args()
    import bench
    from oracle import ref_models
    arch = bench.imagenet_arch()
    # the reference's model, weights as bench.build_model makes them
    sys.path.insert(0, args.reference)
    for k in [k for k in sys.modules if k == 'quant' or k.startswith('quant.')]:
        del sys.modules[k]
    from quant.models.resnet import QResNet as RefResNet            # the reference's
    from quant.binary.binary_conv import QuantConv2d as RefConv
    torch.manual_seed(0)
    ref = RefResNet(loss_fn=torch.nn.functional.cross_entropy, **arch)
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, RefConv) and hasattr(m.w_approximate, 'v1'):        # ls-1 weights: u_o = mean|W_o|
                m.w_approximate.v1.copy_(m.weight.abs().mean(dim=(1, 2, 3)))
    ref.eval()
    sd = {k: v.detach() for k, v in ref.state_dict().items()}
    x = torch.randn(args.batch, 3, 224, 224, generator=torch.Generator().manual_seed(0))
    print(f'torch {torch.__version__}, threads {torch.get_num_threads()}, cpu_count {os.cpu_count()}, batch {args.batch}')
    with torch.no_grad():
        ref(x[:2]); ref_models.resnet_forward(sd, arch, x[:2])     # noqa: E702  (warm-up)
        tr, to = [], []
        for _ in range(args.reps):
            t0 = time.perf_counter(); yr = ref(x); tr.append(time.perf_counter() - t0)                             # noqa: E702
            t0 = time.perf_counter(); yo = ref_models.resnet_forward(sd, arch, x); to.append(time.perf_counter() - t0)   # noqa: E702
    print(f'logits equal bit for bit: {torch.equal(yr, yo)}')
    for name, t in (('reference', tr), ('oracle', to)):
        print(f'{name:9s}: ' + '  '.join(f'{v:6.2f} s' for v in t) + f'   best {min(t):6.2f} s = {args.batch / min(t):6.2f} images/s')
    print(f'oracle / reference wall time (best of {args.reps}): {min(to) / min(tr):.3f}')


if __name__ == '__main__':
    main()
