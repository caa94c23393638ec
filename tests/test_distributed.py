"""CPU, world_size 2, gloo: the N > 1 path of bench.py / sharding.py -- SRS-range partition, all-gather
of one blst_p1 partial per rank, K-1 complete additions on every rank (kzg_g1_sum, host side of the
C-ABI).  The per-rank device MSM is stood in for by the oracle (this is a test: the oracle is the
checker and the stand-in, never the product), so the exchange and the combine run exactly as they do
on RCCL, just over gloo."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, out_q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist

    import bigint_twin as T
    import kzg_poly_commit_exploration_amd as K
    import oracle_ctypes as O
    from kzg_poly_commit_exploration_amd.sharding import allgather_partials, combine, shard_range

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_range(n, rank, world)
        # this rank's SRS slice and coefficient slice (what kzg_srs_generate_g1(first=lo) holds on a GPU)
        srs = np.stack([O.srs_g1_at(k, T.BENCH_SECRET_BE) for k in range(lo, hi)]) if hi > lo else O.p1_zeros(0)
        c = O.bench_coefficients(n)[lo:hi]
        rc, partial = O.commit_naive(c, srs) if hi > lo else (0, np.zeros(18, dtype=np.uint64))
        assert rc == 0
        total = combine(allgather_partials(K.G1Point(partial)))
        out_q.put((rank, total.compress().hex()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [101, 2])
def test_two_rank_sharded_commit_over_gloo(golden, n):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = next(b["commit"] for b in golden["bench"] if b["degree"] == n - 1)
    assert got[0] == want and got[1] == want


def _open_worker(rank, world, port, n, out_q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist

    import bigint_twin as T
    import kzg_poly_commit_exploration_amd as K
    import oracle_ctypes as O
    from kzg_poly_commit_exploration_amd import sharding

    class OracleEngine:
        """Stands in for the GPU engine of this rank (tests only): same two calls, oracle arithmetic."""

        def __init__(self, srs):
            self.srs = srs

        def evaluate_limbs(self, c, z):
            return K.Scalar.from_limbs(O.poly_evaluate(c, O.fr_from_int(z.v)))

        def open_limbs(self, c, z, y):
            rc, pf = O.generate_proof(c, O.fr_from_int(z.v), O.fr_from_int(y.v), self.srs)
            if rc != 0:
                raise K.KzgError(rc, "oracle rc %d" % rc)
            return K.G1Point(pf)

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = sharding.shard_range(n, rank, world)
        srs = np.stack([O.srs_g1_at(k, T.BENCH_SECRET_BE) for k in range(lo, hi)])
        c = O.bench_coefficients(n)
        z = K.Scalar(T.bench_input_point(n - 1))
        y = K.Scalar.from_limbs(O.poly_evaluate(c, O.fr_from_int(z.v)))
        proof = sharding.sharded_open(OracleEngine(srs), c[lo:hi], z, y)
        try:
            sharding.sharded_open(OracleEngine(srs), c[lo:hi], z, K.Scalar(y.v + 1))
            wrong = "accepted"
        except K.KzgError as e:
            wrong = e.status
        out_q.put((rank, proof.compress().hex(), wrong))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_opening_over_gloo(golden):
    import torch.multiprocessing as mp

    n = 101
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_open_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = next(b["proof"] for b in golden["bench"] if b["degree"] == n - 1)
    for _, proof, wrong in got:
        assert proof == want
        assert wrong == -3  # KZG_ERR_REMAINDER on every rank


def _failing_worker(rank, world, port, n, out_q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist

    import bigint_twin as T
    import kzg_poly_commit_exploration_amd as K
    import oracle_ctypes as O
    from kzg_poly_commit_exploration_amd import sharding

    class Engine:
        """Oracle stand-in whose rank-1 instance fails like a GPU engine would (degree too high / busy slot)."""

        def __init__(self, srs, fail_with):
            self.srs, self.fail_with = srs, fail_with

        def commit_limbs(self, c):
            if self.fail_with:
                raise K.KzgError(self.fail_with, "injected")
            rc, p = O.commit_naive(c, self.srs)
            assert rc == 0
            return K.G1Point(p)

        def evaluate_limbs(self, c, z):
            return K.Scalar.from_limbs(O.poly_evaluate(c, O.fr_from_int(z.v)))

        def open_limbs(self, c, z, y):
            if self.fail_with:
                raise K.KzgError(self.fail_with, "injected")
            rc, pf = O.generate_proof(c, O.fr_from_int(z.v), O.fr_from_int(y.v), self.srs)
            assert rc == 0
            return K.G1Point(pf)

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = sharding.shard_range(n, rank, world)
        srs = np.stack([O.srs_g1_at(k, T.BENCH_SECRET_BE) for k in range(lo, hi)])
        c = O.bench_coefficients(n)
        z = K.Scalar(T.bench_input_point(n - 1))
        y = K.Scalar.from_limbs(O.poly_evaluate(c, O.fr_from_int(z.v)))
        eng = Engine(srs, K.KZG_ERR_BUSY if rank == 1 else 0)
        seen = []
        for call in (lambda: sharding.sharded_commit(eng, c[lo:hi]), lambda: sharding.sharded_open(eng, c[lo:hi], z, y)):
            try:
                call()
                seen.append("ok")
            except K.KzgError as e:
                seen.append(e.status)
        # the group is still usable afterwards: a healthy collective completes on both ranks
        ok = sharding.sharded_commit(Engine(srs, 0), c[lo:hi]).compress().hex()
        out_q.put((rank, seen, ok))
    finally:
        dist.destroy_process_group()


def test_engine_error_on_one_rank_is_raised_on_every_rank(golden):
    """a KzgError raised on one rank before the all-gather left the other ranks blocked in the collective.
    The status now travels with the partials; both ranks raise the failing rank's error and nobody hangs."""
    import torch.multiprocessing as mp

    n = 101
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = next(b["commit"] for b in golden["bench"] if b["degree"] == n - 1)
    for _, seen, ok in got:
        assert seen == [-8, -8]  # KZG_ERR_BUSY from rank 1, on both ranks, for commit and for open
        assert ok == want


def test_shard_ranges_cover_everything():
    sys.path.insert(0, ROOT)
    from kzg_poly_commit_exploration_amd.sharding import shard_range

    for n in (0, 1, 7, 8, 9, (1 << 20) + 1, (1 << 22) + 1):
        for world in (1, 2, 4, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b and c <= d
