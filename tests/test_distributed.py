"""The N>1 path on CPU: two processes over gloo agree on the tensor-list partition without
exchanging it, cover the manifest exactly once, and reduce the step time with MAX over ranks --
the only collectives bench.py uses (the data path itself has none)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ggq_pkg import load_package
        import bench
        pkg = load_package()
        # weak scaling as bench.py builds it: world x the per-GPU pool, sharded by the partitioner
        manifest = bench.global_manifest(pkg, pkg.qtypes.Q.Q4_K, pairs=3, world=world)
        mine = pkg.sharding.partition(manifest, world)[rank]
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        cost = sum(pkg.sharding.tensor_cost(manifest[i]) for i in mine)
        costs = [None] * world
        dist.all_gather_object(costs, cost)
        # the timing reduction: every rank ends up with the slowest rank's time
        t = bench.max_over_ranks(10.0 + rank, torch.device("cpu"))
        dist.barrier()
        q.put((rank, gathered, costs, len(manifest), t))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_ranks_partition_and_time_reduction():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=150) for _ in range(world)]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for rank, gathered, costs, n, t in results:
        flat = sorted(i for part in gathered for i in part)
        assert flat == list(range(n))                      # disjoint cover of the tensor list
        assert costs[0] == costs[1]                        # weak scaling: identical per-GPU work
        assert t == 11.0                                   # MAX over ranks
    assert results[0][1] == results[1][1]                  # both ranks computed the same assignment
