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
