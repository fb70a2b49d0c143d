"""One process per GPU without an external launcher, each pinned to the CPU cores next to its GPU.

The reference scales over the GPUs of a node with ``nn.DataParallel`` inside ONE process
(``quant/common/initialization.py:125-127``); here every GPU gets its own process (``sharded_eval``).  When a script is
started plainly (``python bench.py --gpus 8``, no ``torchrun``) :func:`spawn_ranks` starts the ranks itself: N copies of
the same command line with ``RANK`` / ``LOCAL_RANK`` / ``WORLD_SIZE`` / ``MASTER_ADDR`` / ``MASTER_PORT`` set, rank 0's
stdout passed through (the one JSON line of a benchmark), the other ranks' stdout sent to stderr.  A rank that fails
takes the others down with it instead of leaving them waiting in a collective.
"""

import os
import socket
import subprocess
import sys
import time
from typing import Dict, List, Optional, Sequence, Tuple

ENV_MARK = 'LSQ_SELF_LAUNCHED'


def free_port() -> int:
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def parse_cpulist(text: str) -> List[int]:
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the format of /sys/devices/system/node/node*/cpulist)."""
    cpus: List[int] = []
    for part in text.strip().split(','):
        if not part:
            continue
        lo, _, hi = part.partition('-')
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def pci_numa_cpus(pci_address: str, sysfs: str = '/sys') -> Optional[Tuple[int, List[int]]]:
    """(NUMA node, its CPUs) of the PCI function ``dddd:bb:dd.f``; None when the platform does not say (node -1)."""
    try:
        with open(os.path.join(sysfs, 'bus', 'pci', 'devices', pci_address.lower(), 'numa_node')) as f:
            node = int(f.read().strip())
        if node < 0:
            return None
        with open(os.path.join(sysfs, 'devices', 'system', 'node', f'node{node}', 'cpulist')) as f:
            cpus = parse_cpulist(f.read())
    except (OSError, ValueError):
        return None
    return (node, cpus) if cpus else None


def gpu_pci_address(index: int) -> Optional[str]:
    """PCI address of HIP device ``index`` as torch reports it (honours HIP_VISIBLE_DEVICES)."""
    import torch
    try:
        p = torch.cuda.get_device_properties(index)
        return '%04x:%02x:%02x.0' % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    except (AttributeError, RuntimeError, AssertionError):
        return None


def pin_to_gpu_numa(index: int, sysfs: str = '/sys') -> Optional[Dict]:
    """Restrict every thread of this process to the cores of GPU ``index``'s NUMA node (threads started later inherit
    the mask).  Returns what was done, or None when the node is unknown or the mask cannot be set."""
    addr = gpu_pci_address(index)
    found = pci_numa_cpus(addr, sysfs) if addr else None
    if not found:
        return None
    node, cpus = found
    allowed = sorted(set(cpus) & set(os.sched_getaffinity(0)))
    if not allowed:
        return None
    try:
        for tid in os.listdir('/proc/self/task'):
            try:
                os.sched_setaffinity(int(tid), allowed)
            except (OSError, ValueError):
                pass
    except OSError:
        return None
    return {'pci': addr, 'numa_node': node, 'cpus': len(allowed), 'first_cpu': allowed[0], 'last_cpu': allowed[-1]}


def rank_env(rank: int, world: int, port: int, base: Optional[Dict[str, str]] = None) -> Dict[str, str]:
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    env[ENV_MARK] = '1'
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')          # dmabuf IPC: what RCCL needs between processes on this driver
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 1) // max(world, 1) // 2)))
    return env


def spawn_ranks(argv: Sequence[str], world: int, timeout: Optional[float] = None, poll: float = 0.05) -> int:
    """Run ``python argv...`` once per rank and wait.  Returns 0 when every rank exits 0, else the first non-zero exit
    code (after the remaining ranks have been terminated)."""
    port = free_port()
    procs = []
    for rank in range(world):
        out = None if rank == 0 else sys.stderr
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=rank_env(rank, world, port), stdout=out))
    deadline = None if timeout is None else time.monotonic() + timeout
    failed = 0
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [c for c in codes if c not in (None, 0)]
            if bad:
                failed = bad[0]
                break
            if all(c == 0 for c in codes):
                return 0
            if deadline is not None and time.monotonic() > deadline:
                failed = 124
                break
            time.sleep(poll)
    finally:
        for p in procs:                                        # exactly the processes started here, by handle
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return failed
