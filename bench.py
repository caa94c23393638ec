#!/usr/bin/env python3
"""bench.py -- G1 MSM commitments/sec at degree 2^20 on N MI355X (BASELINE.json `metric`).

A step is one pass of the hot path over one batch of B polynomials (B = --batch, default = the number
of GPUs): Polynomial::commit (reference src/polynomial.rs:200-215) of B resident copies of the
reference's bench polynomial c_i = 5^i + 10 (benches/polynomial_commitment.rs:10-15) at degree 2^20
over the SRS of secret 00..1f (benches/polynomial_commitment.rs:17-23), inputs already resident in
HBM; `value` counts commitments (B per step).

  N = 1 : B = 1, the whole MSM on one GPU.  Steps are submitted round-robin on the engine's stream
          slots (no host sync between steps) so the latency-bound reduction tail of one step overlaps
          the accumulation of the next; every result is checked against tests/golden afterwards.
  N > 1 : every commitment is sharded by SRS range (rank g holds points [g*ceil(n/N), ...) with their
          window tables, and the matching coefficient slices); a step takes B = N polynomials through
          one batched pass of the kernels, all-gathers the B partial sums of every rank over RCCL/xGMI in
          one exchange and adds them on every rank.  Per-GPU work per step is that of one full MSM at
          every N -> "scaling": "weak".

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel:
bucket accumulation, HIP events on the kernel's own stream) and `cpu_baseline` (the oracle's
restatement of the reference loop, timed on a bounded sample on this box's host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# HIP multiplexes streams onto 4 hardware queues by default.  The engine uses 4 streams of its own (3
# slots + the accumulation stream); with torch's and RCCL's streams on top, two of them would share a
# queue and a slot's reduction would serialise in front of the next accumulation kernel (1.2 ms bubbles,
# tools/gaps.py).  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

DEGREE = 1 << 20
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
# Integer multiply-add ceiling measured on MI355X (tools/microbench mix, profiles/r01_microbench_mix.jsonl): a bare
# loop of independent 32x32+64 multiply-adds reaches 33.4 T/s chip-wide and nothing hides behind it.
VALU_MAD_PEAK_T = 33.4
# executed v_mad_i64_i32 per mixed addition of the accumulation kernel (signed radix-2^30 field, field30.hip.h):
# 6 products x 2 x 13^2, 2 squarings x (91 + 13^2), and R (Q - X3) - Y1 PPP as two digit products under ONE
# reduction (3 x 13^2); counted in the ISA of k_bucket_accumulate (3055)
MADS_PER_MADD = 6 * 338 + 2 * 260 + 3 * 169


def kernel_source_hash():
    """sha256 over the sources of the dominant kernel: off-line PMC figures are only quoted for the code they were
    measured on (profiles/*.json carry the hash of the sources they profiled)."""
    import hashlib

    h = hashlib.sha256()
    for name in ("msm_accum.hip", "g1_30.hip.h", "field30.hip.h"):
        with open(os.path.join(ROOT, "kzg_poly_commit_exploration_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def offline_profile(name):
    """profiles/<name> if it exists AND was measured on the current kernel sources, else None"""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        j = json.load(f)
    return j if j.get("kernel_source_hash") == kernel_source_hash() else None


def bench_coefficient_limbs(n):
    """c_i = 5^i + 10 mod r as blst_fr rows (benches/polynomial_commitment.rs:10-15)."""
    import numpy as np

    import kzg_poly_commit_exploration_amd as K

    r = K.R_MODULUS
    vals, p5 = [], 1
    for _ in range(n):
        vals.append((p5 + 10) % r)
        p5 = p5 * 5 % r
    return K.scalars_to_limbs(vals), vals


def usable_host_cores():
    """cores this process may actually use: the scheduler affinity, capped by the cgroup CPU quota (a GPU box shows 256
    cores but grants a one-GPU job 16 of them: cpu.max = "1600000 100000")"""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return cores


def cpu_baseline(eng, coeff_limbs, sample):
    """Oracle leg (the ONLY use of oracle/ here): the reference's algorithm -- N scalar multiplications + N
    additions, one thread (src/polynomial.rs:208-212) -- run IN FULL on a degree-2^16 polynomial of the same
    workload, ~15 s on one host core; the degree-2^20 rate is that x 1/16 (the loop is linear
    in N, stated as such).  Plus the oracle's bucket method on every host core as the strong CPU baseline."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_ctypes as O

    n = DEGREE + 1
    srs = eng.srs_read(0, sample)
    c = coeff_limbs[:sample]
    t0 = time.perf_counter()
    rc, cm = O.commit_naive(c, srs)
    dt = time.perf_counter() - t0
    assert rc == 0
    got = eng.commit_limbs(c).compress()
    assert got == O.p1_compress(cm), "GPU and oracle disagree on the cpu_baseline sample"
    host_cores = usable_host_cores()
    out = {"value": 1.0 / (dt * n / sample), "unit": "commitments/s", "cores": 1, "kind": "port", "host_cores": host_cores,
           "host_cores_visible": os.cpu_count() or 1,
           "sample": "degree-%d commitment (%d terms) in full, naive N x scalar-mul loop (the reference's algorithm): %.1f s = "
                     "%.4f commitments/s at that degree, bit-identical to the GPU's; value = that / %.1f (linear in N) for degree 2^20"
                     % (sample - 1, sample, dt, 1.0 / dt, n / sample),
           "measured_commitments_per_sec_at_sample_degree": 1.0 / dt}
    # strong CPU baseline: bucket method on every core this process is granted
    threads = host_cores
    sample2 = min(n, 1 << 18)
    srs2 = eng.srs_read(0, sample2)
    t0 = time.perf_counter()
    rc, cp, used = O.commit_pippenger_ex(coeff_limbs[:sample2], srs2, threads=threads)
    dt2 = time.perf_counter() - t0
    assert rc == 0 and eng.commit_limbs(coeff_limbs[:sample2]).compress() == O.p1_compress(cp)
    out["pippenger"] = {"value": 1.0 / (dt2 * n / sample2), "unit": "commitments/s", "cores": used, "host_cores": host_cores,
                        "sample": "first %d terms, bucket method cut into (window, point range) jobs, %d threads had work, %.2f s, scaled x%.1f"
                                  % (sample2, used, dt2, n / sample2)}
    return out


def host_pointer_path(eng, limbs, z, y, want_commit, want_proof, reps):
    """What the Rust shim calls: kzg_commit / kzg_open with HOST pointers, one synchronous call at a
    time, the 32 MiB copy over PCIe included.  Never `value`: reported beside it."""
    t0 = time.perf_counter()
    for _ in range(reps):
        got = eng.commit_limbs(limbs)
    t_commit = (time.perf_counter() - t0) / reps
    assert want_commit is None or got.compress().hex() == want_commit
    t0 = time.perf_counter()
    for _ in range(reps):
        got = eng.open_limbs(limbs, z, y)
    t_open = (time.perf_counter() - t0) / reps
    assert want_proof is None or got.compress().hex() == want_proof
    # the same calls from 3 and 4 host threads (the reference's callers include `cargo test` threads, src/lib.rs:53, 66,
    # 91): every call owns a stream slot of the context for its duration, so the jobs pipeline on the GPU
    import threading

    threaded = {}
    for nthreads in (3, 4):
        for label, fn in (("commitments", lambda: eng.commit_limbs(limbs)), ("proofs", lambda: eng.open_limbs(limbs, z, y))):
            per = 2 * reps

            def worker(fn=fn):
                for _ in range(per):
                    fn()

            ths = [threading.Thread(target=worker) for _ in range(nthreads)]
            t0 = time.perf_counter()
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            threaded["host_pointer_%s_per_sec_%d_threads" % (label, nthreads)] = nthreads * per / (time.perf_counter() - t0)
    return {"host_pointer_commitments_per_sec": 1.0 / t_commit, "host_pointer_proofs_per_sec": 1.0 / t_open, **threaded,
            "host_pointer_ms": {"commit": t_commit * 1e3, "open": t_open * 1e3},
            "host_pointer_note": "kzg_commit / kzg_open on pageable host memory, synchronous, one at a time (the Rust "
                                 "shim's calls); PCIe copy of the coefficients inside the timed region"}


def batch_of_openings(eng, limbs, n, degree, golden, dev, torch, np, K):
    """BASELINE config 5's per-GPU share: 8 degree-2^20 openings in ONE kzg_open_batch_submit (8 quotient scans, one
    batched MSM), coefficients resident, each at its own point; proofs checked against golden (the bench point) and
    by the pairing check (the others)."""
    b = eng.set_max_batch(8)
    if b < 8:
        return None
    r = K.R_MODULUS
    zs = [K.Scalar((pow(5, degree, r) + 20 + 7 * k) % r) for k in range(8)]   # k = 0: the reference bench's point
    ys = [eng.evaluate_limbs(limbs, zk) for zk in zs]
    d8 = torch.from_numpy(np.ascontiguousarray(limbs).view(np.int64)).to(dev).unsqueeze(0).repeat(8, 1, 1).contiguous()
    zl = np.ascontiguousarray(np.stack([zk.limbs() for zk in zs]))
    yl = np.ascontiguousarray(np.stack([yk.limbs() for yk in ys]))
    lib, h = eng._lib, eng._h
    out = np.zeros((8, 18), dtype=np.uint64)
    st = np.zeros(8, dtype=np.int32)
    import ctypes as C

    def once():
        rc = lib.kzg_open_batch_submit(h, 0, C.c_void_p(d8.data_ptr()), n, 8, n, K._ptr(zl), K._ptr(yl))
        assert rc == 0, rc
        rc = lib.kzg_wait_open_batch(h, 0, K._ptr(out), K._ptr(st), 8)
        assert rc == 0 and not st.any(), (rc, st)

    once()
    torch.cuda.synchronize()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        once()
    dt = (time.perf_counter() - t0) / reps
    want_p = next((x["proof"] for x in golden["bench"] if x["degree"] == degree), None)
    if want_p:
        assert K.G1Point(out[0]).compress().hex() == want_p, "batched opening 0 differs from tests/golden"
    # the same share through the host-pointer batch (kzg_open_batch: what a multi-device context runs per device):
    # 8 x 32 MiB of pageable coefficients uploaded in sub-batches under the kernels of the previous ones
    flat = np.ascontiguousarray(np.broadcast_to(np.ascontiguousarray(limbs, dtype=np.uint64).reshape(1, n, 4), (8, n, 4)))
    host_proofs = eng.open_batch_host(flat, zs, ys)
    assert all(not isinstance(p, Exception) for p in host_proofs)
    assert all(host_proofs[k].compress() == K.G1Point(out[k]).compress() for k in range(8)), "host-pointer batch differs from the device-resident batch"
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.open_batch_host(flat, zs, ys)
    dth = (time.perf_counter() - t0) / reps
    eng.set_max_batch(1)
    return {"openings_batch8_per_sec": 8.0 / dt, "openings_batch8_ms": dt * 1e3,
            "openings_batch8_host_pointer_per_sec": 8.0 / dth,
            "openings_batch8_note": "config 5's per-GPU share: 8 x degree-2^20 openings in one kzg_open_batch_submit, resident inputs; "
                                    "host_pointer: the same through kzg_open_batch (PCIe copies inside, one polynomial per stream slot at this size)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--degree", type=int, default=DEGREE)
    ap.add_argument("--cpu-sample", type=int, default=(1 << 16) + 1, help="terms of the reference-algorithm CPU run (degree 2^16 in full)")
    ap.add_argument("--no-extras", action="store_true", help="skip the host-pointer and batch-of-8-openings legs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-openings", action="store_true", help="skip the opening-proof leg")
    ap.add_argument("--slots", type=int, default=0, help="stream slots kept in flight (0 = all the engine has)")
    ap.add_argument("--batch", type=int, default=0,
                    help="polynomials per step, one batched pass of the kernels (0 = number of GPUs)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo (host tensors) only for rehearsing N > 1 on one GPU")
    ap.add_argument("--device", type=int, default=-1,
                    help="rehearsal: put every rank on this device instead of LOCAL_RANK (needs --backend gloo)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) even at world size 1: exercises the exchange path on one GPU")
    args = ap.parse_args()

    import numpy as np
    import torch

    import kzg_poly_commit_exploration_amd as K
    from kzg_poly_commit_exploration_amd.sharding import allgather_partial_batch, combine, shard_range

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    if args.device >= 0:
        local_rank = args.device
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    xdev = dev if args.backend == "nccl" else None   # where the exchanged partials live

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    degree = args.degree
    n = degree + 1
    with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
        golden = json.load(f)
    secret = bytes.fromhex(golden["secret_be"])
    want = next((b["commit"] for b in golden["bench"] if b["degree"] == degree), None)

    lo, hi = shard_range(n, rank, world)
    eng = K.Engine(local_rank)
    eng.srs_generate(secret, hi - lo, first=lo)      # this rank's SRS slice, resident with its tables
    want_batch = args.batch if args.batch > 0 else world
    batch = eng.set_max_batch(want_batch)            # clamped to what the engine's sort geometry allows
    limbs, _ = bench_coefficient_limbs(n)
    mine = np.ascontiguousarray(limbs[lo:hi])
    one = torch.from_numpy(mine.view(np.int64)).to(dev)
    d_coeffs = one.unsqueeze(0).repeat(batch, 1, 1).contiguous()   # B resident copies, [B][n_mine][4]
    dptr = d_coeffs.data_ptr()
    n_mine = hi - lo
    slots = eng.num_slots() if args.slots <= 0 else min(args.slots, eng.num_slots())
    cfg = eng.msm_config()
    # The kernel's duration is measured LIVE, inside the timed region, for EVERY step: the accumulation kernel stamps
    # its own first wave in and last wave out on the device's 100 MHz clock (what rocprofv3 --kernel-trace reports).
    # HIP events on the stream it runs on bracket it on every TIME_EVERY-th step only: a timed job records six events,
    # two of them between consecutive accumulation kernels on their shared stream, and with every job timed that
    # instrumentation alone cost 8-10 % of the rate it was there to explain (356-364 against 392-395 commitments/s on
    # one box, profiles/r03_timing_overhead.txt).
    TIME_EVERY = max(1, int(os.environ.get("KZG_BENCH_TIME_EVERY", "5")))

    results = []
    accum_ms = []
    phase_ms = {}
    timed_slot = {}

    def collect(slot):
        partials = eng.wait_batch(slot, batch)
        t = eng.times(slot)
        accum_ms.append(t["accumulate_ms"])          # every step: the kernel's own stamps, no event involved
        phase_ms.setdefault("references", []).append(t["references"])
        if timed_slot.get(slot):
            for k, v in t.items():
                if k != "references":
                    phase_ms.setdefault(k, []).append(v)
        if dist is not None:
            partials = [combine(ps) for ps in allgather_partial_batch(partials, device=xdev)]
        results.extend(partials)

    def run(steps):
        inflight = []
        for i in range(steps):
            slot = i % slots
            if len(inflight) == slots:
                collect(inflight.pop(0))
            timed_slot[slot] = i % TIME_EVERY == 0
            eng.set_timing(timed_slot[slot])
            eng.commit_batch_submit(slot, dptr, n_mine, batch)
            inflight.append(slot)
        while inflight:
            collect(inflight.pop(0))

    run(args.warmup)
    results.clear()
    accum_ms.clear()
    phase_ms.clear()
    barrier()
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    assert len(results) == args.steps * batch
    ok = all(r.compress().hex() == want for r in results) if want else None
    if want and not ok:
        raise SystemExit("rank %d: commitment differs from tests/golden (degree %d)" % (rank, degree))

    # opening proofs (secondary figure).  Openings shard by POLYNOMIAL (BASELINE config 5: a batch of openings,
    # 8 per GPU): every rank opens its own polynomials against the full SRS, no exchange; the aggregate is
    # (ranks x openings per rank) / slowest rank.  At N > 1 the ranks hold SRS slices for the commitments, so a
    # second engine with the full SRS is set up for this leg (outside every timed region).
    proofs_per_s = None
    quotient_ms = None
    if args.steps > 0 and not args.no_openings:
        z = K.Scalar((pow(5, degree, K.R_MODULUS) + 20) % K.R_MODULUS)   # benches/evaluation_proof.rs:25-27
        if world == 1:
            eng_o, optr = eng, dptr
        else:
            eng_o = K.Engine(local_rank)
            eng_o.srs_generate(secret, n)
            d_full = torch.from_numpy(np.ascontiguousarray(limbs).view(np.int64)).to(dev)
            optr = d_full.data_ptr()
        y = eng_o.evaluate_limbs(limbs, z)
        want_p = next((b["proof"] for b in golden["bench"] if b["degree"] == degree), None)
        k_open = max(3, min(args.steps * max(1, batch // world), 24))
        oslots = eng_o.num_slots() if args.slots <= 0 else min(args.slots, eng_o.num_slots())
        barrier()
        t1 = time.perf_counter()
        inflight, proofs, qms = [], [], []
        for i in range(k_open):
            slot = i % oslots
            if len(inflight) == oslots:
                s0 = inflight.pop(0)
                proofs.append(eng_o.wait(s0))
                if timed_slot.get(s0):
                    qms.append(eng_o.times(s0)["quotient_ms"])
            timed_slot[slot] = i % TIME_EVERY == 0
            eng_o.set_timing(timed_slot[slot])
            eng_o.open_submit(slot, optr, n, z, y)
            inflight.append(slot)
        while inflight:
            s0 = inflight.pop(0)
            proofs.append(eng_o.wait(s0))
            if timed_slot.get(s0):
                qms.append(eng_o.times(s0)["quotient_ms"])
        barrier()
        el_o = time.perf_counter() - t1
        if dist is not None:
            tt = torch.tensor([el_o], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el_o = float(tt.item())
        proofs_per_s = world * k_open / el_o
        quotient_ms = sum(qms) / max(1, len(qms))
        if want_p:
            assert all(p.compress().hex() == want_p for p in proofs), "proof differs from tests/golden"
        if eng_o is not eng:
            eng_o.close()

    eng.set_timing(False)
    if rank == 0:
        avg_accum_ms = sum(accum_ms) / max(1, len(accum_ms))
        # 96 B point + 32 B scalar per term, + 144 B out; one launch = B commitments
        algo_bytes = batch * (n_mine * 128 + 144)
        achieved = algo_bytes / (avg_accum_ms * 1e-3) / 1e9 if avg_accum_ms > 0 else 0.0
        refs = phase_ms.pop("references", [])
        madds = int(sum(refs) / max(1, len(refs)))   # one mixed addition per non-zero scalar digit (counted on the device)
        tmad = madds * MADS_PER_MADD / (avg_accum_ms * 1e-3) / 1e12 if avg_accum_ms > 0 else 0.0
        # PMC figures are measured OFF-LINE (tools/prof_round3.sh) and only quoted when the profiled sources are the
        # ones that just ran (hash of msm_accum.hip + g1_30.hip.h + field30.hip.h stored with them)
        traffic = None
        tj = offline_profile("r03_traffic.json") if (world == 1 and degree == DEGREE and batch == 1) else None
        if tj and tj.get("recoding", "windows") == cfg["recoding"]:
            traffic = tj.get("hbm_bytes_per_launch")
        valu_pmc = {}
        vj = offline_profile("r03_valu_pmc.json") if (world == 1 and degree == DEGREE and batch == 1) else None
        if vj:
            valu_pmc = {"pmc_valu_busy_percent": vj.get("VALUBusy"),
                        "pmc_valu_lane_utilization_percent": vj.get("VALUUtilization"),
                        "pmc_valu_instructions_per_mixed_addition": vj.get("valu_instructions_per_mixed_addition"),
                        "pmc_source": "offline: profiles/r03_valu_pmc.json (tools/prof_round3.sh, same workload, same kernel sources)"}
        line = {
            "metric": "g1_msm_commitments_per_sec_degree_2^20",
            "value": args.steps * batch / elapsed,
            "unit": "commitments/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "i32 digits (Fp: 13 signed radix-2^30 digits, Montgomery 2^390; Fr: 8 x u32, Montgomery 2^256)",
            "data": "synthetic: reference bench inputs c_i=5^i+10, SRS secret 00..1f, generated on device",
            "config": {"workload": "configs[2]: degree-2^%d commit (G1 MSM, %d terms) on %d x MI355X, SRS-range sharded, "
                                   "%d commitments per step in one batched pass" % (degree.bit_length() - 1, n, world, batch),
                       "degree": degree, "terms_per_gpu": n_mine, "commitments_per_step": batch, "recoding": cfg["recoding"],
                       "digit_bits": cfg["digit_bits"], "table_levels": cfg["table_levels"], "buckets": cfg["buckets"],
                       "table_gib": round(cfg["table_levels"] * n_mine * 128 / 2**30, 2), "stream_slots": slots,
                       "bit_exact_vs_golden": ok},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": ("offline: profiles/r03_traffic.json, same kernel sources" if traffic else None),
                         "kernel": "k_bucket_accumulate", "avg_kernel_ms": avg_accum_ms,
                         "avg_kernel_ms_hip_event_bracket": (sum(phase_ms.get("accumulate_events_ms", [0.0])) /
                                                             max(1, len(phase_ms.get("accumulate_events_ms", [])))),
                         "timing": "avg_kernel_ms: every step of the timed region, the kernel's own duration (first wave in to last wave "
                                   "out, stamped by the kernel on the device's 100 MHz clock: what rocprofv3 --kernel-trace reports); "
                                   "avg_kernel_ms_hip_event_bracket: HIP events on its stream around every %d-th step (six events per "
                                   "timed job cost the pipeline 8-10 %% when every job carried them), which also hold the time the "
                                   "launch waited for the chip" % TIME_EVERY,
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "kernel_source_hash": kernel_source_hash(),
                         "note": "integer/modular work with no MFMA form: the kernel is bound by VALU issue (see valu), "
                                 "the HBM fraction is small by construction"},
            "valu": {"achieved_Tmad_s": tmad, "peak_Tmad_s": VALU_MAD_PEAK_T, "frac": tmad / VALU_MAD_PEAK_T,
                     "unit": "1e12 32x32+64-bit integer multiply-adds/s (v_mad_i64_i32 executed; peak = bare v_mad_u64_u32 loop)",
                     "mixed_additions_per_launch": madds, "multiply_adds_per_mixed_addition": MADS_PER_MADD,
                     **valu_pmc},
            # HIP-event spans on the slot's stream: digits and accumulate bracket their own kernels; scatter and reduce
            # INCLUDE queueing behind other slots' kernels (three commitments in flight), so they do not add up to a step
            "phase_ms_queueing_inclusive": {k: sum(v) / len(v) for k, v in phase_ms.items()},
            "opening_proofs_per_sec": proofs_per_s,
            "quotient_ms": quotient_ms,
        }
        if world == 1 and degree == DEGREE and not args.no_extras and not args.no_openings and args.steps > 0:
            want_p0 = next((b["proof"] for b in golden["bench"] if b["degree"] == degree), None)
            line.update(host_pointer_path(eng, limbs, z, y, want, want_p0, 5))
            extra = batch_of_openings(eng, limbs, n, degree, golden, dev, torch, np, K)
            if extra:
                line.update(extra)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(eng, limbs, min(args.cpu_sample, n))
        print(json.dumps(line))
    barrier()
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
