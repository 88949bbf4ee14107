"""The multi-rank path of bench.py on a single-GPU box: two ranks started by bench.py's own self-launch (torch.distributed.run, loopback
rendezvous), both on GPU 0, process group on gloo (RCCL refuses two ranks on one device) -- everything else is the real thing: one process
per rank, HIP-graph replay of the step, the asynchronous packed-stat reducer, barriers and max-over-ranks timing.  The driver's 8-GPU
scaling run uses the same code with backend nccl (= RCCL over xGMI); the nearest reference precedent is
torch_utils/training_stats.py:236-267 (pack -> one all_reduce -> unpack)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra):
    env = dict(os.environ, **env_extra)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]            # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_two_ranks_one_device_dry_run():
    common = ['--steps', '6', '--warmup', '2', '--no-cpu-baseline', '--no-final-psnr', '--no-roofline', '--no-side-configs']
    two = _run(['--gpus', '2'] + common, dict(EG3D_BENCH_BACKEND='gloo', EG3D_BENCH_ONE_DEVICE='1'))
    assert two['n_gpus'] == 2 and two['config']['world_size'] == 2 and two['steps'] == 6 and two['scaling'] == 'weak'
    assert two['config']['launch'] == 'one HIP graph replay per step'
    assert 'mean loss over ranks and steps' in two['config']['parallelism']            # the reducer's totals reached rank 0
    one = _run(['--gpus', '1'] + common, {})
    assert one['n_gpus'] == 1
    # two ranks time-share ONE device here: the aggregate cannot beat one rank alone by much, and must not collapse either
    # (a blocking per-step collective or a lost graph replay would show up as a several-fold drop)
    ratio = two['value'] / one['value']
    assert 0.5 <= ratio <= 1.35, (two['value'], one['value'])
    assert abs(two['ms_per_step'] * two['value'] / 2 - 1e3) < 1.0                      # value = world x steps / max-over-ranks time


def test_eight_ranks_one_device_dry_run():
    """config C5's launch shape -- eight ranks -- on one device: the communicator warm-up in front of the captures, per-rank CPU pinning,
    eight concurrent graph captures / replays, the asynchronous reducer with eight contributors, one JSON line from rank 0."""
    common = ['--steps', '4', '--warmup', '1', '--no-cpu-baseline', '--no-final-psnr', '--no-roofline', '--no-side-configs']
    r = _run(['--gpus', '8'] + common, dict(EG3D_BENCH_BACKEND='gloo', EG3D_BENCH_ONE_DEVICE='1'))
    assert r['n_gpus'] == 8 and r['config']['world_size'] == 8 and r['steps'] == 4 and r['scaling'] == 'weak'
    assert r['config']['launch'] == 'one HIP graph replay per step'
    assert 'mean loss over ranks and steps' in r['config']['parallelism']
    assert abs(r['ms_per_step'] * r['value'] / 8 - 1e3) < 1.0
    assert r['config']['host_cpus_of_rank0'] >= 1
