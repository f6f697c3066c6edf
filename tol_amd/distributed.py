"""One process per GPU: shard a batch of independent trajectories, gather their objectives.

Trajectories never exchange data while F and G are evaluated, so the only collective on the path is
the final gather of the per-trajectory objectives F[t][0] (BASELINE north star).  With the `nccl`
backend that is one RCCL all-gather over xGMI; the same code runs over `gloo` on CPU tensors, which
is how the N > 1 logic is tested without GPUs.
"""
import torch
import torch.distributed as dist


def shard_bounds(total, rank, world):
    """Contiguous shard [lo, hi) of `total` trajectories; the first total % world ranks get one more."""
    base, extra = divmod(int(total), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_objectives(local_obj, total, group=None, async_op=False):
    """All-gather the per-trajectory objectives of every rank's shard into one vector of length
    `total`, ordered by global trajectory index.  Returns the vector (or (vector, work) when async)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return (local_obj, None) if async_op else local_obj
    rank = dist.get_rank(group)
    lo, hi = shard_bounds(total, rank, world)
    assert local_obj.numel() == hi - lo, "local objectives do not match this rank's shard"
    width = shard_bounds(total, 0, world)[1]            # widest shard
    padded = local_obj
    if local_obj.numel() != width:
        padded = torch.zeros(width, dtype=local_obj.dtype, device=local_obj.device)
        padded[: hi - lo] = local_obj
    out = torch.empty(width * world, dtype=local_obj.dtype, device=local_obj.device)
    work = dist.all_gather_into_tensor(out, padded, group=group, async_op=async_op)

    def trim():
        if total == width * world:
            return out
        parts = []
        for r in range(world):
            a, b = shard_bounds(total, r, world)
            parts.append(out[r * width: r * width + (b - a)])
        return torch.cat(parts)

    if async_op:
        return trim, work
    return trim()


def mean_objective(local_obj, total, group=None):
    """Monte-Carlo mean of the objectives over all `total` trajectories of the job: one all-reduce(sum) of each rank's
    partial sum (the optional second collective; latency-bound, 8 bytes per rank)."""
    s = local_obj.sum(dtype=torch.float64).reshape(1)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(s, op=dist.ReduceOp.SUM, group=group)
    return s / float(total)
