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


_BUFFERS = {}


def _buffers(world, device, count=1):
    """Persistent staging buffers: pinned host in/out plus device in/out (count x 18 int64 per rank)."""
    import torch

    key = (world, str(device), count)
    if key not in _BUFFERS:
        pin = device is not None
        h_in = torch.empty(18 * count, dtype=torch.int64, pin_memory=pin)
        h_out = torch.empty(18 * count * world, dtype=torch.int64, pin_memory=pin)
        if device is not None:
            d_in = torch.empty(18 * count, dtype=torch.int64, device=device)
            d_out = torch.empty(18 * count * world, dtype=torch.int64, device=device)
        else:
            d_in, d_out = h_in, h_out
        _BUFFERS[key] = (h_in, h_out, d_in, d_out)
    return _BUFFERS[key]


def allgather_partials(partial, device=None, group=None):
    """All-gathers one G1Point (144 B) per rank through torch.distributed, rank order preserved.

    With `device` set the exchange runs on the GPU (backend nccl = RCCL over xGMI): one tiny H2D copy,
    one all_gather_into_tensor, one D2H copy, all on persistent buffers.  uint64 limbs travel as int64."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    h_in, h_out, d_in, d_out = _buffers(world, device)
    h_in.numpy()[:] = partial.p1.view(np.int64)
    if device is not None:
        d_in.copy_(h_in, non_blocking=True)
        dist.all_gather_into_tensor(d_out, d_in, group=group)
        h_out.copy_(d_out, non_blocking=True)
        torch.cuda.current_stream(device).synchronize()
    else:
        gathered = [torch.empty_like(h_in) for _ in range(world)]
        dist.all_gather(gathered, h_in, group=group)
        h_out.copy_(torch.cat(gathered))
    arr = h_out.numpy().view(np.uint64).reshape(world, 18)
    return [G1Point(arr[r].copy()) for r in range(world)]


def allgather_partial_batch(partials, device=None, group=None):
    """Batch form: `partials` = this rank's list of B G1Points; returns B lists of `world` points
    (one exchange of B x 144 bytes per rank)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    count = len(partials)
    h_in, h_out, d_in, d_out = _buffers(world, device, count)
    h_in.numpy().reshape(count, 18)[:] = np.stack([p.p1 for p in partials]).view(np.int64)
    if device is not None:
        d_in.copy_(h_in, non_blocking=True)
        dist.all_gather_into_tensor(d_out, d_in, group=group)
        h_out.copy_(d_out, non_blocking=True)
        torch.cuda.current_stream(device).synchronize()
    else:
        gathered = [torch.empty_like(h_in) for _ in range(world)]
        dist.all_gather(gathered, h_in, group=group)
        h_out.copy_(torch.cat(gathered))
    arr = h_out.numpy().view(np.uint64).reshape(world, count, 18)
    return [[G1Point(arr[r, b].copy()) for r in range(world)] for b in range(count)]


def combine(partials):
    """K-1 complete additions + normalisation = the reduce of the sharded MSM."""
    return G1Point.sum(partials)


def sharded_commit(engine, coeff_slice_limbs, device=None, group=None):
    """Commit this rank's coefficient slice on its SRS slice and reduce across ranks."""
    partial = engine.commit_limbs(coeff_slice_limbs)
    return combine(allgather_partials(partial, device=device, group=group))
