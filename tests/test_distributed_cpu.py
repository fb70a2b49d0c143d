"""The N>1 inference path on CPU: world_size 2, gloo backend, batch sharded by rank, logits all-gathered."""

import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    import sys
    for p in (ROOT, os.path.join(ROOT, 'ml-quant_amd'), os.path.join(ROOT, 'tests', 'golden')):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import detgen
    from quant.common.sharded_eval import evaluate_sharded, local_slice
    from quant.models.lenet import QLeNet5
    torch.set_num_threads(2)
    model = QLeNet5(loss_fn=None, x_quant='ls-2', w_quant='ls-1', clamp={'kind': 'symmetric', 'alpha': 2})
    detgen.fill_module(model, seed=4)
    with torch.no_grad():
        model.conv2.w_approximate.v1.copy_(model.conv2.weight.abs().mean(dim=(1, 2, 3)))
    model.eval()
    x = detgen.normal('dist.x', (12, 1, 28, 28))
    gathered = evaluate_sharded(model, x[local_slice(12, rank, world)])
    with torch.no_grad():
        full = model(x)
    torch.save({'gathered': gathered, 'full': full}, os.path.join(out_dir, f'r{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def _tiny_resnet():
    """A QResNet small enough for the CPU suite with the real block structure (xnor blocks, double shortcut,
    a stride-2 stage with its projection shortcut), ls-1 weights / ls-2 activations as the headline config."""
    import detgen
    from quant.binary.binary_conv import QuantConv2d
    from quant.models.resnet import QResNet

    def layer(alpha):
        return {'x_quant': 'ls-2', 'w_quant': 'ls-1', 'clamp': {'kind': 'symmetric', 'alpha': alpha}, 'double_shortcut': True}
    arch = {'moving_average_mode': 'off', 'moving_average_momentum': 0.99, 'block': 'xnor',
            'layer0': {'n_in_channels': 16, 'kernel_size': 3, 'stride': 1, 'padding': 1, 'bias': False,
                       'maxpool': {'type': 'identity'}},
            'layer1': layer(3), 'layer2': layer(3), 'layer3': layer(3), 'layer4': layer(3),
            'nonlins': ['relu', 'relu'], 'num_blocks': [1, 1, 1, 1], 'output_classes': 10}
    model = QResNet(loss_fn=None, **arch)
    detgen.fill_module(model, seed=9)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, QuantConv2d):
                m.w_approximate.v1.copy_(m.weight.abs().mean(dim=(1, 2, 3)))
    return model.eval()


def _worker_resnet_uneven(rank, world, port, out_dir):
    import sys
    for p in (ROOT, os.path.join(ROOT, 'ml-quant_amd'), os.path.join(ROOT, 'tests', 'golden')):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import detgen
    from quant.common.sharded_eval import evaluate_sharded, local_slice
    torch.set_num_threads(2)
    model = _tiny_resnet()
    n = 7                                            # 4 + 3: shards of different size
    x = detgen.normal('dist.rx', (n, 3, 16, 16))
    gathered = evaluate_sharded(model, x[local_slice(n, rank, world)], total=n)
    out = torch.empty((n, 10))
    evaluate_sharded(model, x[local_slice(n, rank, world)], out=out, total=n)
    with torch.no_grad():
        full = model(x)
        local = model(x[local_slice(n, rank, world)])
    torch.save({'gathered': gathered, 'out': out, 'full': full, 'local': local}, os.path.join(out_dir, f'q{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_resnet_uneven_shards_world2_gloo(tmp_path):
    """A QResNet (not only LeNet) through evaluate_sharded with world size 2 and a batch that does not divide
    evenly: shards are padded for the collective and trimmed afterwards."""
    world, port = 2, _free_port()
    mp.spawn(_worker_resnet_uneven, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    ds = [torch.load(os.path.join(str(tmp_path), f'q{r}.pt')) for r in range(world)]
    shards = torch.cat([d['local'] for d in ds], dim=0)
    for d in ds:
        assert d['gathered'].shape == (7, 10)
        # the exchange itself is exact; the fp32 stem / classifier of torch-CPU are not bitwise batch-size invariant,
        # so the comparison with the unsharded forward carries a rounding tolerance
        assert torch.equal(d['gathered'], shards) and torch.equal(d['out'], shards)
        assert torch.allclose(d['gathered'], d['full'], rtol=1e-5, atol=1e-5)


def test_sharded_eval_world2_gloo(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        d = torch.load(os.path.join(str(tmp_path), f'r{r}.pt'))
        # per-sample scales => sharding the batch changes nothing, bit for bit
        assert d['gathered'].shape == (12, 10) and torch.equal(d['gathered'], d['full'])


def test_local_slice_covers_batch():
    from quant.common.sharded_eval import local_slice
    for n, w in ((12, 2), (13, 4), (2048, 8), (3, 8)):
        idx = [i for r in range(w) for i in range(n)[local_slice(n, r, w)]]
        assert idx == list(range(n))


def _worker_small_batch(rank, world, port, out_dir):
    import sys
    for p in (ROOT, os.path.join(ROOT, 'ml-quant_amd'), os.path.join(ROOT, 'tests', 'golden')):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import detgen
    from quant.common.metrics import Top1Accuracy
    from quant.common.training import evaluate
    torch.set_num_threads(2)
    model = _tiny_resnet()
    # a "test set" whose last batch holds ONE sample: rank 1 gets an empty shard of it
    xs = detgen.normal('dist.small', (5, 3, 16, 16))
    ys = torch.tensor([1, 3, 5, 7, 9])

    class Loader(list):
        dataset = list(range(5))
    loader = Loader([(xs[:4], ys[:4]), (xs[4:], ys[4:])])
    metrics = {'Top-1 Accuracy': Top1Accuracy(accumulate=True)}
    out = evaluate(model, loader, metrics, torch.device('cpu'), epoch=1)
    with torch.no_grad():
        full = model(xs)
    want = float((full.argmax(dim=1) == ys).float().mean())
    torch.save({'metric': out['Top-1 Accuracy'], 'want': want}, os.path.join(out_dir, f's{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_evaluate_with_a_batch_smaller_than_the_world(tmp_path):
    """The reference's DataParallel never hands a replica an empty batch; batch-sharded ranks can get one (a last batch
    of one sample on two ranks).  The empty rank must still join the collective, and every rank must report the metric of
    the whole set."""
    world, port = 2, _free_port()
    mp.spawn(_worker_small_batch, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        d = torch.load(os.path.join(str(tmp_path), f's{r}.pt'))
        assert abs(float(d['metric']) - d['want']) < 1e-6 or abs(float(d['metric']) - 100 * d['want']) < 1e-4, d


def _worker_overflow_flag(rank, world, port, out_dir):
    import sys
    for p in (ROOT, os.path.join(ROOT, 'ml-quant_amd'), os.path.join(ROOT, 'tests', 'golden')):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import detgen
    from quant.common import training
    from quant.common.metrics import Top1Accuracy
    torch.set_num_threads(2)
    model = _tiny_resnet()
    xs = detgen.normal('dist.flag', (8, 3, 16, 16))
    ys = torch.arange(8) % 10

    class Loader(list):
        dataset = list(range(8))
    loader = Loader([(xs[:4], ys[:4]), (xs[4:], ys[4:])])
    # the stem's device flag as ONE rank would see it: up after the first pass on rank 1 only, down after the reset
    state = {'raised': False, 'asked': 0, 'resets': 0, 'passes': 0}

    def flag(device):
        state['asked'] += 1
        return state['raised']

    def reset(device):
        state['resets'] += 1
        state['raised'] = False
    training._stem_flag, training._stem_flag_reset = flag, reset
    local_forward = training.local_forward

    def counting_forward(m, shard):
        state['passes'] += 1
        if rank == 1 and state['passes'] == 1:
            state['raised'] = True                    # "the kernel saw an operand beyond 65504" in rank 1's first batch
        return local_forward(m, shard)
    training.local_forward = counting_forward
    out = training.evaluate(model, loader, {'Top-1 Accuracy': Top1Accuracy(accumulate=True)}, torch.device('cpu'), epoch=1)
    torch.save({'metric': out['Top-1 Accuracy'], 'state': state}, os.path.join(out_dir, f'f{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_evaluate_repeats_on_every_rank_when_one_rank_saw_an_overflow(tmp_path):
    """the stem's out-of-domain flag is per device.  When only ONE rank's shard held the offending sample,
    that rank alone used to re-enter evaluate() and issue a second series of all-gathers -- a hang.  The decision is now
    all-reduced: both ranks run the pass twice (4 forwards each: 2 batches x 2 passes) and report the same metric."""
    world, port = 2, _free_port()
    mp.spawn(_worker_overflow_flag, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    ds = [torch.load(os.path.join(str(tmp_path), f'f{r}.pt')) for r in range(world)]
    assert ds[0]['state']['passes'] == ds[1]['state']['passes'] == 4, [d['state'] for d in ds]
    assert ds[0]['state']['resets'] == ds[1]['state']['resets'] == 1
    assert float(ds[0]['metric']) == float(ds[1]['metric'])
