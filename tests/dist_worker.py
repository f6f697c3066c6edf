"""Rank body of tests/test_distributed_gpu.py: evaluate this rank's shard with the HIP path, gather the objectives.
Started by torch.distributed.run as a FRESH process (one per rank; the ranks may share one GPU)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def trajectory(tol_amd, t):
    """Trajectory t of the global batch (the test rebuilds the same table for the oracle)."""
    rng = np.random.default_rng(4000 + t)
    return tol_amd.Trajectory(aircraft=t % 2, Vref=rng.uniform(0, 5), href=rng.uniform(5, 20), radius_goal=100.0,
                              xi=rng.uniform(-50, 50), yi=rng.uniform(-50, 50), zi=rng.uniform(-100, -20))


def decision_vector(x0, t):
    rng = np.random.default_rng(6000 + t)
    x = x0 + 0.05 * rng.uniform(-1, 1, x0.shape) * (1 + np.abs(x0))
    x[0] = abs(x[0]) + 0.01
    return x


def main():
    out, total, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    import tol_amd
    from tol_amd.distributed import gather_objectives, shard_bounds
    dist.init_process_group("gloo")          # two ranks on ONE GPU: objectives staged through host memory
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = shard_bounds(total, rank, world)
    bt = tol_amd.Batch("S10", ("tempest", "skywalker"), ts=N, device=0)
    trajs = [trajectory(tol_amd, t) for t in range(lo, hi)]
    bt.set_trajectories(trajs)
    X = np.stack([decision_vector(bt.x0(i, zi=trajs[i].zi), lo + i) for i in range(hi - lo)])
    dX, dF, dG = bt.alloc(hi - lo)
    dX[:, :bt.n] = torch.from_numpy(X).cuda()
    obj = torch.empty(hi - lo, dtype=torch.float64, device="cuda")
    bt.eval(dX, dF, dG, obj=obj)             # the HIP path (tol_amd/lib/libtolfg.so)
    torch.cuda.synchronize()
    full = gather_objectives(obj.cpu(), total)
    if rank == 0:
        np.savez(out, objectives=full.numpy(), F0=dF[:, 0].cpu().numpy(), lo=lo, hi=hi)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
