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


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(x: float, device) -> float:
    t = torch.tensor([x], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
