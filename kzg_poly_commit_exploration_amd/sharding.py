"""SRS-range sharding of one commitment over the GPUs of a node (one process per GPU).

sum_{i<N} c_i SRS_i = sum_g sum_{i in slice g} c_i SRS_i: rank g keeps SRS slice [lo_g, hi_g) resident
(with its window tables) and receives the matching coefficient slice; the only exchange is the
all-gather of one 144-byte blst_p1 partial sum per rank (RCCL over xGMI when the tensors are on the
GPU, gloo in the CPU tests), followed by K-1 complete point additions on every rank
(kzg_g1_sum).  RCCL has no user-defined reduction for curve points, so "reduce" = gather + local
add; the payload is latency-bound.

Openings shard the same way.  With S[i] = sum_{k>=i} c_k z^(k-i) the quotient is q[j] = S[j+1]
(reference src/polynomial.rs:168-179) and the proof is sum_j q[j] SRS[j]; rank g owns j in [lo_g, hi_g).
Its q[j] only need its own coefficients and ONE carry, C_g = S[hi_g], which enters exactly like an extra
top coefficient: scanning the slice extended by c_ext[len] = C_g yields q[lo_g .. hi_g) and, as its
"evaluation", S[lo_g] = the carry of rank g-1.  So: every rank evaluates its slice at z (device scan, no
carry), the K values H_g = sum_{k in slice} c_k z^(k-lo_g) are all-gathered (32 B each), every rank runs
the K-step recurrence C_{g-1} = H_g + z^(len_g) C_g on the host, then opens its extended slice and the
partial proofs are combined like partial commitments.  No kernel knows about the sharding.
"""
import numpy as np

from . import KZG_ERR_CONSTANT_POLY, KZG_ERR_REMAINDER, R_MODULUS, G1Point, KzgError, Scalar, load_library


def shard_range(n, rank, world):
    """Contiguous slice [lo, hi) of n SRS indices owned by `rank` (ceil split, last may be short)."""
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


_BUFFERS = {}


def _buffers(world, device, count=1):
    """Persistent staging buffers: pinned host in/out plus device in/out (count x 18 int64 per rank, plus ONE
    status word per rank: an engine error on one rank travels with the partials, so every rank leaves the
    collective and raises the same error instead of the healthy ranks blocking in it forever)."""
    import torch

    key = (world, str(device), count)
    if key not in _BUFFERS:
        pin = device is not None
        words = 18 * count + 1
        h_in = torch.empty(words, dtype=torch.int64, pin_memory=pin)
        h_out = torch.empty(words * world, dtype=torch.int64, pin_memory=pin)
        if device is not None:
            d_in = torch.empty(words, dtype=torch.int64, device=device)
            d_out = torch.empty(words * world, dtype=torch.int64, device=device)
        else:
            d_in, d_out = h_in, h_out
        _BUFFERS[key] = (h_in, h_out, d_in, d_out)
    return _BUFFERS[key]


def _raise_first_failure(statuses):
    """Same KzgError on every rank: the status of the lowest failing rank."""
    for r, st in enumerate(statuses):
        if st != 0:
            raise KzgError(int(st), load_library().kzg_strerror(int(st)).decode() + " (rank %d)" % r)


def guarded(fn):
    """Runs a local engine call; returns (result, 0) or (None, status) instead of raising before a collective."""
    try:
        return fn(), 0
    except KzgError as e:
        return None, int(e.status)


_INFINITY = np.zeros(18, dtype=np.uint64)


def allgather_partials(partial, device=None, group=None, status=0):
    """All-gathers one G1Point (144 B) and one status word per rank through torch.distributed, rank order
    preserved.  `status` != 0 (an engine error on this rank; `partial` may then be None) makes EVERY rank raise
    that KzgError after the collective.

    With `device` set the exchange runs on the GPU (backend nccl = RCCL over xGMI): one tiny H2D copy,
    one all_gather_into_tensor, one D2H copy, all on persistent buffers.  uint64 limbs travel as int64."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    h_in, h_out, d_in, d_out = _buffers(world, device)
    h_in.numpy()[:18] = (_INFINITY if partial is None else partial.p1).view(np.int64)
    h_in.numpy()[18] = status
    if device is not None:
        d_in.copy_(h_in, non_blocking=True)
        dist.all_gather_into_tensor(d_out, d_in, group=group)
        h_out.copy_(d_out, non_blocking=True)
        torch.cuda.current_stream(device).synchronize()
    else:
        gathered = [torch.empty_like(h_in) for _ in range(world)]
        dist.all_gather(gathered, h_in, group=group)
        h_out.copy_(torch.cat(gathered))
    raw = h_out.numpy().reshape(world, 19)
    _raise_first_failure(raw[:, 18])
    arr = raw[:, :18].copy().view(np.uint64)
    return [G1Point(arr[r].copy()) for r in range(world)]


def allgather_partial_batch(partials, device=None, group=None, status=0, count=None):
    """Batch form: `partials` = this rank's list of B G1Points; returns B lists of `world` points
    (one exchange of B x 144 bytes + one status word per rank).  With status != 0 pass count=B and partials=None."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    count = len(partials) if count is None else count
    h_in, h_out, d_in, d_out = _buffers(world, device, count)
    if status == 0:
        h_in.numpy()[:18 * count].reshape(count, 18)[:] = np.stack([p.p1 for p in partials]).view(np.int64)
    else:
        h_in.numpy()[:18 * count] = 0
    h_in.numpy()[18 * count] = status
    if device is not None:
        d_in.copy_(h_in, non_blocking=True)
        dist.all_gather_into_tensor(d_out, d_in, group=group)
        h_out.copy_(d_out, non_blocking=True)
        torch.cuda.current_stream(device).synchronize()
    else:
        gathered = [torch.empty_like(h_in) for _ in range(world)]
        dist.all_gather(gathered, h_in, group=group)
        h_out.copy_(torch.cat(gathered))
    raw = h_out.numpy().reshape(world, 18 * count + 1)
    _raise_first_failure(raw[:, 18 * count])
    arr = raw[:, :18 * count].copy().view(np.uint64).reshape(world, count, 18)
    return [[G1Point(arr[r, b].copy()) for r in range(world)] for b in range(count)]


def combine(partials):
    """K-1 complete additions + normalisation = the reduce of the sharded MSM."""
    return G1Point.sum(partials)


def sharded_commit(engine, coeff_slice_limbs, device=None, group=None):
    """Commit this rank's coefficient slice on its SRS slice and reduce across ranks.  An engine error on any rank
    (degree too high for its slice, HIP error, busy slot) is raised on every rank, after the collective."""
    partial, status = guarded(lambda: engine.commit_limbs(coeff_slice_limbs))
    return combine(allgather_partials(partial, device=device, group=group, status=status))


def _allgather_u64(vec, world_group=None, device=None):
    """All-gathers a small uint64 vector per rank (exchange helper for the opening carries)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(world_group)
    mine = torch.from_numpy(np.ascontiguousarray(vec, dtype=np.uint64).view(np.int64).copy())
    if device is not None:
        mine = mine.to(device)
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine, group=world_group)
    return [t.cpu().numpy().view(np.uint64) for t in out]


def opening_carries(h_values, lengths, z):
    """K-step recurrence on the host.  h_values[g] = slice evaluation H_g (int), lengths[g] = slice length.
    Returns (carries, s_at_slice_start): carries[g] = S[hi_g], s_at_slice_start[g] = S[lo_g]."""
    k = len(h_values)
    carries, starts = [0] * k, [0] * k
    c = 0
    for g in range(k - 1, -1, -1):
        carries[g] = c
        c = (h_values[g] + pow(z, lengths[g], R_MODULUS) * c) % R_MODULUS
        starts[g] = c
    return carries, starts


def sharded_open_local(engine, coeff_slice_limbs, carry, z, y_at_slice_start):
    """This rank's partial proof: open the slice extended by the carry as one more top coefficient."""
    ext = np.concatenate([np.ascontiguousarray(coeff_slice_limbs, dtype=np.uint64).reshape(-1, 4),
                          Scalar(carry).limbs().reshape(1, 4)])
    return engine.open_limbs(ext, z, Scalar(y_at_slice_start))


def sharded_open(engine, coeff_slice_limbs, z, y, device=None, group=None):
    """Evaluation::generate_proof (reference src/polynomial.rs:260-269) of a polynomial whose coefficients
    (and SRS) are range-sharded over the ranks of `group`.  Errors are raised identically on every rank."""
    import torch.distributed as dist

    rank = dist.get_rank(group)
    sl = np.ascontiguousarray(coeff_slice_limbs, dtype=np.uint64).reshape(-1, 4)
    h, h_status = guarded(lambda: engine.evaluate_limbs(sl, z)) if len(sl) else (Scalar(0), 0)
    if h is None:
        h = Scalar(0)
    first = 1 if rank == 0 else 0  # index 0 of the polynomial is not a "higher" coefficient
    higher = int(sl[first:].any()) if len(sl) > first else 0
    c0 = sl[0] if (rank == 0 and len(sl)) else np.zeros(4, dtype=np.uint64)
    gathered = _allgather_u64(np.concatenate([h.limbs(), [np.uint64(len(sl)), np.uint64(higher)], c0,
                                              [np.uint64(h_status & 0xFFFFFFFF)]]), group, device)
    _raise_first_failure([int(np.int32(np.uint32(int(g[10])))) for g in gathered])  # an engine error of the evaluation, on every rank
    hs = [Scalar.from_limbs(g[:4]).v for g in gathered]
    lens = [int(g[4]) for g in gathered]
    any_higher = any(int(g[5]) for g in gathered)
    c0_limbs = gathered[0][6:10]
    carries, starts = opening_carries(hs, lens, z.v)
    lib = load_library()
    if not any_higher:  # constant polynomial after truncation (src/polynomial.rs:159-167)
        if Scalar.from_limbs(c0_limbs).v != y.v:
            raise KzgError(KZG_ERR_CONSTANT_POLY, lib.kzg_strerror(KZG_ERR_CONSTANT_POLY).decode())
        return G1Point(np.zeros(18, dtype=np.uint64))
    if starts[0] != y.v:  # P(z) != y (src/polynomial.rs:184-192)
        raise KzgError(KZG_ERR_REMAINDER, lib.kzg_strerror(KZG_ERR_REMAINDER).decode())
    if len(sl):
        partial, status = guarded(lambda: sharded_open_local(engine, sl, carries[rank], z, starts[rank]))
    else:
        partial, status = G1Point(np.zeros(18, dtype=np.uint64)), 0
    return combine(allgather_partials(partial, device=device, group=group, status=status))
