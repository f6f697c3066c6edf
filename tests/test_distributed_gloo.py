"""The N > 1 path on CPU: two gloo ranks shard a batch of trajectories, each evaluates its shard,
and the objectives are all-gathered in global trajectory order (the only collective on the path).
The shard evaluation itself is the oracle here (the HIP kernels need a GPU); what is under test is
the partitioning and the gather, which are the same code bench.py and a GPU deployment run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _objective(t):
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    o = O.Problem("S10", "tempest", N=20, Vref=1.0 + 0.1 * t, href=10.0)
    return o.eval(O.perturbed(o, 1000 + t), needG=False)[0][0]


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tol_amd.distributed import gather_objectives, mean_objective, shard_bounds
    lo, hi = shard_bounds(total, rank, world)
    local = torch.tensor([_objective(t) for t in range(lo, hi)], dtype=torch.float64)
    full = gather_objectives(local, total)
    mean = mean_objective(local, total)                       # the optional all-reduce(sum) of SURVEY section 8e
    assert abs(mean.item() - full.mean().item()) <= 1e-12 * abs(full.mean().item())
    trim, work = gather_objectives(local, total, async_op=True)
    work.wait()
    assert torch.equal(trim(), full)
    q.put((rank, lo, hi, full.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 7])
def test_two_ranks_gather_objectives_in_order(total):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.array([_objective(t) for t in range(total)])
    covered = []
    for rank, lo, hi, full in got:
        assert np.array_equal(full, want)        # every rank holds every objective, in global order
        covered += list(range(lo, hi))
    assert sorted(covered) == list(range(total))  # shards tile the batch exactly once


def test_shard_bounds():
    sys.path.insert(0, ROOT)
    from tol_amd.distributed import shard_bounds
    for total in (1, 7, 8, 1024, 8192, 8193):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
