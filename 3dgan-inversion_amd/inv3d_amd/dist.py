"""Multi-GPU layer: independent images sharded over ranks, one tiny stat all-reduce and nothing else.

One process per GPU (`torch.distributed`, backend "nccl" == RCCL over xGMI on ROCm; "gloo" on CPU for tests), launched by
torchrun / `python -m torch.distributed.run`.  Each image is a separate optimisation with its own latent, camera, noise
buffers and optimiser state, so there is no gradient or parameter traffic: the only message is a packed fp32 vector
[sum loss, sum dist, sum psnr, n_active, n_done, max step ms, ...] (<= 16 floats) per logging interval -- pure latency;
ring-vs-tree and the 7 x 153 GB/s xGMI link bound never matter at this size.  The nearest precedent in the reference is
torch_utils/training_stats.py:236-267 (pack moments -> one all_reduce -> unpack)."""
import os
from typing import Dict, List, Sequence

import torch
import torch.distributed as dist

STAT_FIELDS = ('loss', 'dist', 'mse', 'psnr', 'n_active', 'n_done', 'steps')
MAX_FIELDS = ('step_ms',)


def init_from_env(backend: str = None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (no-op for a single process).
    Returns (rank, world_size, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', str(rank)))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


COMM_READY = False       # set by warm_up(): the communicator exists (RCCL creates it lazily, at the first collective)


def warm_up(device) -> None:
    """Run one all-reduce + barrier NOW, on `device`.  RCCL builds its communicator (device allocations, IPC handle exchange, its own
    streams) inside the first collective; that must not happen while a stream is capturing a HIP graph or between a capture's warm-up
    and its replay.  bench.py / InversionCoach.run call this right after init_from_env, before any generator work."""
    global COMM_READY
    if dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.ones(1, dtype=torch.float32, device=device)
        dist.all_reduce(t)
        dist.barrier()
        if torch.device(device).type == 'cuda':
            torch.cuda.synchronize(device)
        assert int(t.item()) == dist.get_world_size(), 'warm-up all-reduce returned a wrong sum'
    COMM_READY = True


def assert_comm_ready():
    """Called in front of a graph capture in a multi-rank job: the first collective must already have run (warm_up)."""
    if dist.is_initialized() and dist.get_world_size() > 1 and not COMM_READY:
        raise RuntimeError('multi-rank job: call inv3d_amd.dist.warm_up(device) after init_from_env and before capturing HIP graphs '
                           '(RCCL creates its communicator inside the first collective)')


def _parse_cpulist(text: str) -> List[int]:
    out = []
    for part in text.strip().split(','):
        if not part:
            continue
        a, _, b = part.partition('-')
        out.extend(range(int(a), int(b or a) + 1))
    return out


def numa_cpus_for_rank(local_rank: int, local_world: int, allowed: Sequence[int], node_cpus: Dict[int, List[int]], gpu_node: Dict[int, int]) -> List[int]:
    """CPUs rank `local_rank` should run on: the allowed CPUs of its GPU's NUMA node, divided evenly among the ranks whose GPUs sit on the
    same node (each rank queues ~200 launches + one graph launch per step: eight ranks on one socket's cores throttle each other);
    without topology information an even contiguous split of the allowed CPUs.  Pure function (tested on CPU)."""
    allowed = sorted(set(allowed))
    node = gpu_node.get(local_rank, -1)
    if node >= 0 and node in node_cpus:
        mine = [c for c in node_cpus[node] if c in set(allowed)]
        peers = sorted(r for r in range(local_world) if gpu_node.get(r, -1) == node)
        if mine and local_rank in peers:
            k, n = peers.index(local_rank), len(peers)
            per = max(1, len(mine) // n)
            chunk = mine[k * per:(k + 1) * per] if k < n - 1 else mine[k * per:]
            return chunk or mine        # (fewer CPUs on the node than peer ranks: share them all rather than pin to nothing)
    per = max(1, len(allowed) // max(1, local_world))
    chunk = allowed[local_rank * per:(local_rank + 1) * per]
    return chunk or allowed


def pin_to_numa(local_rank: int, local_world: int) -> List[int]:
    """sched_setaffinity of this process to numa_cpus_for_rank(...) (EG3D_PIN=0: leave the affinity alone).  Returns the CPU list in effect."""
    if os.environ.get('EG3D_PIN', '1') == '0' or local_world <= 1 or not hasattr(os, 'sched_setaffinity'):
        return sorted(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else []
    node_cpus, gpu_node = {}, {}
    try:
        base = '/sys/devices/system/node'
        for d in os.listdir(base):
            if d.startswith('node') and d[4:].isdigit():
                node_cpus[int(d[4:])] = _parse_cpulist(open(os.path.join(base, d, 'cpulist')).read())
        if torch.cuda.is_available():
            for r in range(min(local_world, torch.cuda.device_count())):
                bdf = getattr(torch.cuda.get_device_properties(r), 'pci_bus_id', None)
                dom = getattr(torch.cuda.get_device_properties(r), 'pci_domain_id', 0)
                dv = getattr(torch.cuda.get_device_properties(r), 'pci_device_id', 0)
                if bdf is not None:
                    path = '/sys/bus/pci/devices/%04x:%02x:%02x.0/numa_node' % (dom, bdf, dv)
                    if os.path.exists(path):
                        gpu_node[r] = int(open(path).read().strip())
    except (OSError, ValueError):
        pass
    cpus = numa_cpus_for_rank(local_rank, local_world, sorted(os.sched_getaffinity(0)), node_cpus, gpu_node)
    try:
        os.sched_setaffinity(0, cpus)
    except OSError:
        pass
    return sorted(os.sched_getaffinity(0))


def shard_images(num_images: int, rank: int, world: int) -> List[int]:
    """Image i -> rank (i mod world): every rank gets floor/ceil(num_images/world) independent inversions."""
    return list(range(rank, num_images, world))


def pack_stats(values: Dict[str, float], device) -> torch.Tensor:
    v = [float(values.get(k, 0.0)) for k in STAT_FIELDS] + [float(values.get(k, 0.0)) for k in MAX_FIELDS]
    return torch.tensor(v, dtype=torch.float32, device=device)


def allreduce_stats(values: Dict[str, float], device) -> Dict[str, float]:
    """Sum the STAT_FIELDS and max the MAX_FIELDS over all ranks with two tiny collectives on one packed vector."""
    t = pack_stats(values, device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        ns = len(STAT_FIELDS)
        s, m = t[:ns].clone(), t[ns:].clone()
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        t = torch.cat([s, m])
    out = t.tolist()
    keys = STAT_FIELDS + MAX_FIELDS
    return {k: out[i] for i, k in enumerate(keys)}


def allreduce_stats_device(t: torch.Tensor) -> torch.Tensor:
    """Per-step variant for the optimisation loop: sum a packed fp32 stat vector that already lives on the device over all ranks,
    in place, with ONE collective and no host read-back (the host keeps queueing the next step; read the tensor at a logging
    interval).  No-op for a single process."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


class StepStatReducer:
    """The per-step packed-stat all-reduce without putting the ranks in lockstep.  A blocking collective per step makes every rank's
    compute stream wait for the slowest rank of THAT step (jitter adds up over 8 ranks) plus the collective's latency.  Here each
    step's vector is copied into a slot of a small ring and reduced with an asynchronous collective: RCCL's stream waits for the copy,
    the compute stream waits for nothing.  A slot is waited on only when it is reused `depth` steps later (long finished by then) or in
    `finish()`, which returns the sum over all pushed steps of the rank-summed vectors."""

    def __init__(self, nfields: int, device, depth: int = 8):
        self.ring = torch.zeros(depth, nfields, dtype=torch.float32, device=device)
        self.total = torch.zeros(nfields, dtype=torch.float32, device=device)
        self.works = [None] * depth
        self.used = [False] * depth
        self.depth, self.i = depth, 0
        self.active = dist.is_initialized() and dist.get_world_size() > 1

    def _retire(self, slot):
        if self.works[slot] is not None:
            self.works[slot].wait()          # nccl: the current stream waits for the collective; gloo: the host does
            self.works[slot] = None
        if self.used[slot]:
            self.total += self.ring[slot]
            self.used[slot] = False

    def push(self, vec: torch.Tensor):
        slot = self.i % self.depth
        self._retire(slot)
        self.ring[slot].copy_(vec)
        self.used[slot] = True
        if self.active:
            self.works[slot] = dist.all_reduce(self.ring[slot], op=dist.ReduceOp.SUM, async_op=True)
        self.i += 1

    def finish(self) -> torch.Tensor:
        for slot in range(self.depth):
            self._retire(slot)
        return self.total


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(x: float, device) -> float:
    t = torch.tensor([x], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
