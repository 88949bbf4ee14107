"""Multi-process path on CPU (gloo, world_size 2): image sharding + the packed stat all-reduce, i.e. everything that crosses
ranks on the N > 1 path (independent images, one tiny all-reduce and nothing else)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    torch.set_num_threads(2)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from inv3d_amd import dist as D
    r, w, _ = D.init_from_env('gloo')
    assert (r, w) == (rank, world)
    mine = D.shard_images(7, r, w)
    # every rank reports the stats of its own images; the reduced vector must equal the global sums / max
    stats = D.allreduce_stats(dict(loss=float(sum(mine)), dist=1.0, psnr=10.0 * (r + 1), n_active=float(len(mine)), steps=3.0,
                                   step_ms=5.0 + r), torch.device('cpu'))
    D.barrier()
    mx = D.max_over_ranks(1.0 + r, torch.device('cpu'))
    dev_stats = D.allreduce_stats_device(torch.tensor([1.0 + r, 10.0, float(len(mine))]))       # per-step in-place variant
    stats['dev'] = dev_stats.tolist()
    # the asynchronous per-step reducer: 11 steps through a ring of 4 slots
    red = D.StepStatReducer(3, torch.device('cpu'), depth=4)
    for step in range(11):
        red.push(torch.tensor([float(step) * (r + 1), 1.0, float(r)]))
    stats['async'] = red.finish().tolist()
    out.put((rank, mine, stats, mx))
    dist.destroy_process_group()


def test_two_rank_gloo_stat_sync():
    world = 2
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    shards = [r[1] for r in res]
    assert sorted(shards[0] + shards[1]) == list(range(7)) and not set(shards[0]) & set(shards[1])
    for _, _, st, mx in res:
        assert st['loss'] == float(sum(range(7)))
        assert st['n_active'] == 7.0 and st['dist'] == 2.0 and st['steps'] == 6.0 and st['psnr'] == 30.0
        assert st['step_ms'] == 6.0
        assert st['dev'] == [3.0, 20.0, 7.0]
        assert st['async'] == [55.0 * 3, 22.0, 11.0]            # sum over steps of the rank-summed vectors
        assert mx == 2.0


def test_single_process_is_a_noop():
    from inv3d_amd import dist as D
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        os.environ.pop(k, None)
    assert D.init_from_env('gloo') == (0, 1, 0)
    st = D.allreduce_stats(dict(loss=2.5, n_active=1.0), torch.device('cpu'))
    assert st['loss'] == 2.5 and st['n_active'] == 1.0
    assert D.allreduce_stats_device(torch.tensor([4.0])).item() == 4.0
    red = D.StepStatReducer(2, torch.device('cpu'), depth=2)
    for i in range(5):
        red.push(torch.tensor([1.0, float(i)]))
    assert red.finish().tolist() == [5.0, 10.0]
    assert D.shard_images(5, 0, 1) == [0, 1, 2, 3, 4]
    assert D.max_over_ranks(3.0, torch.device('cpu')) == 3.0


def test_bench_self_launch_plumbing():
    """`python bench.py --gpus N` outside a launcher re-launches itself in the driver's own form (torch.distributed.run, one rank per GPU,
    loopback rendezvous) with its arguments passed through unchanged."""
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    argv = ['--gpus', '4', '--steps', '7', '--warmup', '2', '--images-per-gpu', '8']
    cmd = bench.self_launch_command(4, argv, port=23456)
    assert cmd[0] == sys.executable and cmd[1:3] == ['-m', 'torch.distributed.run']
    assert '--nnodes=1' in cmd and '--nproc-per-node=4' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[cmd.index('--master-port') + 1] == '23456'
    k = cmd.index(os.path.join(root, 'bench.py'))
    assert cmd[k + 1:] == argv
    auto = bench.self_launch_command(2, [])
    assert 1024 < int(auto[auto.index('--master-port') + 1]) < 65536


def test_numa_cpu_split_is_a_partition():
    """dist.numa_cpus_for_rank: ranks of one NUMA node share its allowed CPUs without overlap; without topology an even split."""
    from inv3d_amd import dist as D
    node_cpus = {0: list(range(0, 48)) + list(range(96, 144)), 1: list(range(48, 96)) + list(range(144, 192))}
    gpu_node = {0: 0, 1: 0, 2: 0, 3: 0, 4: 1, 5: 1, 6: 1, 7: 1}
    allowed = list(range(192))
    got = [D.numa_cpus_for_rank(r, 8, allowed, node_cpus, gpu_node) for r in range(8)]
    assert all(len(g) == 24 for g in got)
    assert sorted(c for g in got for c in g) == allowed
    for r in range(8):
        assert set(got[r]) <= set(node_cpus[gpu_node[r]])
    even = [D.numa_cpus_for_rank(r, 8, list(range(16)), {}, {}) for r in range(8)]
    assert even == [[2 * r, 2 * r + 1] for r in range(8)]
    assert D.numa_cpus_for_rank(3, 8, [5], {}, {}) == [5]          # fewer CPUs than ranks: never an empty set
    # fewer allowed CPUs on the NUMA node than peer ranks (a cgroup with 2 CPUs of node 0, four ranks there): every rank gets a non-empty set
    # (`A if c else B or mine` bound the `or` to the else branch only -- non-last ranks got [] and stayed unpinned)
    few = [D.numa_cpus_for_rank(r, 8, [0, 1, 50, 51], node_cpus, gpu_node) for r in range(8)]
    assert all(few) and few[0] == [0] and few[1] == [1] and few[2] == [0, 1] and few[3] == [0, 1]
    assert D._parse_cpulist('0-3,8,10-11\n') == [0, 1, 2, 3, 8, 10, 11]


def test_capture_guard_for_the_communicator():
    from inv3d_amd import dist as D
    D.COMM_READY = False
    D.assert_comm_ready()          # single process: nothing to wait for
    D.warm_up('cpu')
    assert D.COMM_READY
