"""SRS-range sharding of one commitment over the GPUs of a node (one process per GPU).

sum_{i<N} c_i SRS_i = sum_g sum_{i in slice g} c_i SRS_i: rank g keeps SRS slice [lo_g, hi_g) resident
(with its window tables) and receives the matching coefficient slice; the only exchange is the
all-gather of one 144-byte blst_p1 partial sum per rank (RCCL over xGMI when the tensors are on the
GPU, gloo in the CPU tests), followed by K-1 complete point additions on every rank
(kzg_g1_sum).  RCCL has no user-defined reduction for curve points, so "reduce" = gather + local
add; the payload is latency-bound.
"""
import numpy as np

from . import G1Point


def shard_range(n, rank, world):
    """Contiguous slice [lo, hi) of n SRS indices owned by `rank` (ceil split, last may be short)."""
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def allgather_partials(partial, device=None, group=None):
    """All-gathers one G1Point per rank through torch.distributed and returns them in rank order."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    # uint64 limbs travel as int64 (same bits); RCCL needs device tensors, gloo takes CPU ones
    mine = torch.from_numpy(partial.p1.view(np.int64).copy())
    if device is not None:
        mine = mine.to(device)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine, group=group)
    return [G1Point(t.cpu().numpy().view(np.uint64)) for t in gathered]


def combine(partials):
    """K-1 complete additions + normalisation = the reduce of the sharded MSM."""
    return G1Point.sum(partials)


def sharded_commit(engine, coeff_slice_limbs, device=None, group=None):
    """Commit this rank's coefficient slice on its SRS slice and reduce across ranks."""
    partial = engine.commit_limbs(coeff_slice_limbs)
    return combine(allgather_partials(partial, device=device, group=group))
