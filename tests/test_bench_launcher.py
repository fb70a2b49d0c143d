"""bench.py's own rank launcher (`python bench.py --gpus N` with no torchrun) and the NUMA pinning helpers, on CPU.

The reference scales with nn.DataParallel inside one process (quant/common/initialization.py:125-127); here the ranks
are processes, and the benchmark must be able to start them itself."""

import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')
SMALL = ['--device', 'cpu', '--batch', '2', '--image-size', '32', '--steps', '2', '--warmup', '1', '--min-seconds', '0']


def _env():
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'LSQ_SELF_LAUNCHED'):
        env.pop(k, None)
    return env


def test_cpulist_and_numa_lookup(tmp_path):
    from quant.common import rank_launcher as rl
    assert rl.parse_cpulist('0-3,8,10-11\n') == [0, 1, 2, 3, 8, 10, 11]
    assert rl.parse_cpulist('') == []
    dev = tmp_path / 'bus' / 'pci' / 'devices' / '0000:05:00.0'
    dev.mkdir(parents=True)
    (dev / 'numa_node').write_text('1\n')
    node = tmp_path / 'devices' / 'system' / 'node' / 'node1'
    node.mkdir(parents=True)
    (node / 'cpulist').write_text('64-127,192-255\n')
    found = rl.pci_numa_cpus('0000:05:00.0', sysfs=str(tmp_path))
    assert found[0] == 1 and len(found[1]) == 128 and found[1][0] == 64 and found[1][-1] == 255
    (dev / 'numa_node').write_text('-1\n')                      # a platform that does not say: no pinning, no error
    assert rl.pci_numa_cpus('0000:05:00.0', sysfs=str(tmp_path)) is None
    assert rl.pci_numa_cpus('0000:99:00.0', sysfs=str(tmp_path)) is None


def test_rank_environment():
    from quant.common import rank_launcher as rl
    env = rl.rank_env(3, 8, 12345, base={'PATH': '/bin'})
    assert (env['RANK'], env['LOCAL_RANK'], env['WORLD_SIZE'], env['MASTER_ADDR'], env['MASTER_PORT']) == ('3', '3', '8', '127.0.0.1', '12345')
    assert env['HSA_ENABLE_IPC_MODE_LEGACY'] == '0' and env[rl.ENV_MARK] == '1' and env['PATH'] == '/bin'


def test_a_failing_rank_takes_the_others_down():
    from quant.common import rank_launcher as rl
    t0 = time.monotonic()
    code = rl.spawn_ranks(['-c', 'import os, sys, time\nsys.exit(3) if os.environ["RANK"] == "1" else time.sleep(120)'], 3)
    assert code == 3 and time.monotonic() - t0 < 30
    assert rl.spawn_ranks(['-c', 'import os; assert os.environ["WORLD_SIZE"] == "2"'], 2) == 0


def _one_json_line(stdout):
    lines = [ln for ln in stdout.strip().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: two ranks, gloo in the CPU plumbing mode, ONE JSON line."""
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2'] + SMALL, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _one_json_line(r.stdout)
    assert out['n_gpus'] == 2 and out['launcher'] == {'kind': 'self', 'ranks': 2, 'numa_pinning_rank0': None}
    assert out['config']['global_batch'] == 4 and out['scaling'] == 'weak' and out['steps'] == 2
    assert out['allgather']['world_size_seen_by_backend'] == 2 and out['allgather']['bytes_received_per_rank'] == 2 * 1000 * 4
    assert out['device'] == 'cpu' and 'PLUMBING' in out['note'] and 'roofline' not in out and 'cpu_baseline' not in out
    assert out['value'] > 0 and out['steps_timed'] == 2
    # 8-GPU readiness: rank 0's line at N > 1 carries the exchange's per-link rate and both issue orders
    assert out['allgather']['per_link_GBps'] > 0 and out['allgather']['backend'] == 'gloo'
    assert out['pipelined']['value'] == out['value'] and out['pipelined']['vs_single_stream'] > 0
    assert out['single_stream']['value'] > 0 and out['single_stream']['steps_timed'] >= 2 and 'value_definition' in out


def test_bench_under_torchrun_and_with_a_world_size_that_differs_from_gpus():
    """The driver's launch line (torch.distributed.run); WORLD_SIZE wins over a stale --gpus instead of an assertion."""
    from quant.common.rank_launcher import free_port
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), BENCH, '--gpus', '4'] + SMALL
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _one_json_line(r.stdout)
    assert out['n_gpus'] == 2 and out['launcher']['kind'] == 'torchrun'
    assert 'WORLD_SIZE=2' in r.stderr


def test_single_rank_needs_no_process_group():
    r = subprocess.run([sys.executable, BENCH, '--gpus', '1'] + SMALL, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _one_json_line(r.stdout)
    assert out['n_gpus'] == 1 and out['launcher']['kind'] == 'none' and 'allgather' not in out


@pytest.mark.gpu
def test_bench_refuses_more_ranks_than_gpus():
    """On a one-GPU box `--gpus 2` must fail with a message, not hang in a collective."""
    import torch
    n = torch.cuda.device_count()
    r = subprocess.run([sys.executable, BENCH, '--gpus', str(n + 1), '--steps', '1', '--warmup', '0', '--cpu-sample', '0', '--no-configs'],
                       env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and 'GPU(s) are visible' in r.stderr
