"""world_size-2 gloo test of the N>1 path (instance sharding + the single throughput all-gather).
The data path itself has no collective; on the GPU box the same code runs with backend nccl (= RCCL)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bundletrack_amd import sharding


def test_instances_for_rank_partitions_exactly():
    for n, g in ((256, 8), (33, 4), (3, 8), (0, 2)):
        got = [sharding.instances_for_rank(n, r, g) for r in range(g)]
        flat = sorted(i for part in got for i in part)
        assert flat == list(range(n))
        assert max(len(p) for p in got) - min(len(p) for p in got) <= 1
        for r, part in enumerate(got):
            assert all(i % g == r for i in part)
    with pytest.raises(ValueError):
        sharding.instances_for_rank(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = sharding.init_from_env(backend="gloo")
    mine = sharding.instances_for_rank(10, r, w)
    sharding.barrier()
    seconds = 0.5 + 0.25 * r                      # rank 1 is the slow one
    per_rank = sharding.gather_throughput(seconds, 7.0 * len(mine))
    total, slowest = sharding.aggregate(per_rank)
    q.put((r, mine, per_rank, total, slowest))
    dist.destroy_process_group()


def test_two_rank_gloo_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4, 6, 8] and res[1][1] == [1, 3, 5, 7, 9]
    for r in res:
        assert r[2] == [(0.5, 35.0), (0.75, 35.0)]          # every rank sees every rank's numbers
        assert abs(r[3] - 70.0 / 0.75) < 1e-9 and r[4] == 0.75   # whole-job rate = total work / slowest rank


def test_single_process_needs_no_group():
    assert sharding.gather_throughput(2.0, 14.0) == [(2.0, 14.0)]
    assert sharding.aggregate([(2.0, 14.0)]) == (7.0, 2.0)


def test_rank_cpu_binding_is_disjoint_and_reversible():
    """bench.py binds every rank of a node to its own CPUs (eight launcher threads must not share one quota).  The binding of two local ranks is
    disjoint, never leaves the set the process was allowed, reports what it did, and a single rank is left alone."""
    import os
    from bundletrack_amd import sharding
    if not hasattr(os, "sched_getaffinity"):
        return
    allowed = set(os.sched_getaffinity(0))
    try:
        assert sharding.bind_rank_to_cpus(0, 1)["bound"] is False
        got = []
        for r in range(2):
            os.sched_setaffinity(0, allowed)
            info = sharding.bind_rank_to_cpus(r, 2, gpu_index=None)
            if len(allowed) < 4:
                assert info["bound"] is False
                return
            assert info["bound"] is True and info["cpus"] >= 1
            got.append(set(os.sched_getaffinity(0)))
            assert got[-1] <= allowed
        assert not (got[0] & got[1])
    finally:
        os.sched_setaffinity(0, allowed)


def test_usable_cpus_respects_affinity_and_quota():
    """bench.py sizes its generator pools from sharding.usable_cpus(): never more than the affinity mask, never more than the cgroup quota."""
    import os
    from bundletrack_amd import sharding
    n = sharding.usable_cpus()
    assert 1 <= n <= len(os.sched_getaffinity(0))
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            assert n <= max(1, int(float(q) / float(period) + 0.5))
    except OSError:
        pass
