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
# Measured issue costs on MI355X (tools/microbench mix, profiles/r01_microbench_mix.jsonl): a bare v_mad_u64_u32
# loop reaches 33.4 T/s chip-wide (4 waves/SIMD; 30.6 T/s at the kernel's 2 waves/SIMD) and NOTHING hides behind it:
# at 2 waves/SIMD a wave-level v_mad_u64_u32 costs ~5.13 issue cycles and a simple 32-bit VALU instruction ~2.98.
VALU_MAD_PEAK_T = 33.4
MAD_ISSUE_CYCLES, OTHER_ISSUE_CYCLES = 5.13, 2.98
OTHER_VALU_PER_MADD = 3578     # SQ_INSTS_VALU per wave-level mixed addition (6348, profiles/r01_valu_pmc.json) - MADS_PER_MADD
SIMDS, CLOCK_HZ = 1024, 2.4e9
MADS_PER_MADD = 8 * 288 + 2 * 233  # executed v_mad_u64_u32: 8 products x 2 x 12^2, 2 squarings x (89 + 12^2)


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


def cpu_baseline(eng, coeff_limbs, sample):
    """Oracle leg (the ONLY use of oracle/ here): the reference's algorithm -- N scalar
    multiplications + N additions, one thread -- on the first `sample` terms of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_ctypes as O

    srs = eng.srs_read(0, sample)
    c = coeff_limbs[:sample]
    t0 = time.perf_counter()
    rc, cm = O.commit_naive(c, srs)
    dt = time.perf_counter() - t0
    assert rc == 0
    got = eng.commit_limbs(c).compress()
    assert got == O.p1_compress(cm), "GPU and oracle disagree on the cpu_baseline sample"
    n = DEGREE + 1
    per_commit_s = dt * n / sample
    out = {"value": 1.0 / per_commit_s, "unit": "commitments/s", "cores": 1, "kind": "port",
           "sample": "first %d of %d terms, naive N x scalar-mul loop (reference algorithm), %.1f s, scaled x%.1f"
                     % (sample, n, dt, n / sample)}
    # strong CPU baseline: bucket method on all host cores, full size
    cores = os.cpu_count() or 1
    threads = min(cores, 32)
    sample2 = min(n, 1 << 18)
    srs2 = eng.srs_read(0, sample2)
    t0 = time.perf_counter()
    rc, cp = O.commit_pippenger(coeff_limbs[:sample2], srs2, threads=threads)
    dt2 = time.perf_counter() - t0
    assert rc == 0 and eng.commit_limbs(coeff_limbs[:sample2]).compress() == O.p1_compress(cp)
    out["pippenger"] = {"value": 1.0 / (dt2 * n / sample2), "unit": "commitments/s", "cores": threads,
                        "sample": "first %d terms, bucket method, %.1f s, scaled x%.1f" % (sample2, dt2, n / sample2)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--degree", type=int, default=DEGREE)
    ap.add_argument("--cpu-sample", type=int, default=1 << 15)
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
    eng.set_timing(True)

    results = []
    accum_ms = []
    phase_ms = {}

    def collect(slot):
        partials = eng.wait_batch(slot, batch)
        t = eng.times(slot)
        accum_ms.append(t["accumulate_ms"])
        for k, v in t.items():
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
            eng_o.set_timing(True)
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
                qms.append(eng_o.times(s0)["quotient_ms"])
            eng_o.open_submit(slot, optr, n, z, y)
            inflight.append(slot)
        while inflight:
            s0 = inflight.pop(0)
            proofs.append(eng_o.wait(s0))
            qms.append(eng_o.times(s0)["quotient_ms"])
        barrier()
        el_o = time.perf_counter() - t1
        if dist is not None:
            tt = torch.tensor([el_o], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el_o = float(tt.item())
        proofs_per_s = world * k_open / el_o
        quotient_ms = sum(qms) / len(qms)
        if want_p:
            assert all(p.compress().hex() == want_p for p in proofs), "proof differs from tests/golden"
        if eng_o is not eng:
            eng_o.close()

    if rank == 0:
        avg_accum_ms = sum(accum_ms) / max(1, len(accum_ms))
        # 96 B point + 32 B scalar per term, + 144 B out; one launch = B commitments
        algo_bytes = batch * (n_mine * 128 + 144)
        achieved = algo_bytes / (avg_accum_ms * 1e-3) / 1e9 if avg_accum_ms > 0 else 0.0
        refs = phase_ms.pop("references", [])
        madds = int(sum(refs) / max(1, len(refs)))   # one mixed addition per non-zero scalar digit (counted on the device)
        tmad = madds * MADS_PER_MADD / (avg_accum_ms * 1e-3) / 1e12 if avg_accum_ms > 0 else 0.0
        issue_ms = (madds / 64.0) * (MADS_PER_MADD * MAD_ISSUE_CYCLES + OTHER_VALU_PER_MADD * OTHER_ISSUE_CYCLES) \
            / (SIMDS * CLOCK_HZ) * 1e3
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if world == 1 and degree == DEGREE and os.path.exists(tpath):
            # PMC bytes of the same kernel on the same workload, measured off-line by tools/prof_pmc.sh
            with open(tpath) as f:
                tj = json.load(f)
            if tj.get("batch", 1) == batch and tj.get("recoding", "windows") == cfg["recoding"]:
                traffic = tj.get("hbm_bytes_per_launch")
        valu_pmc = {}
        vpath = os.path.join(ROOT, "profiles", "r01_valu_pmc.json")
        if world == 1 and degree == DEGREE and batch == 1 and os.path.exists(vpath):
            with open(vpath) as f:
                vj = json.load(f)
            valu_pmc = {"pmc_valu_busy_percent": vj.get("VALUBusy"),
                        "pmc_valu_lane_utilization_percent": vj.get("VALUUtilization"),
                        "pmc_source": "profiles/r01_valu_pmc.json (tools/prof_valu.sh, same workload)"}
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
            "dtype": "u32 limbs (384-bit Fp / 256-bit Fr Montgomery integers)",
            "data": "synthetic: reference bench inputs c_i=5^i+10, SRS secret 00..1f, generated on device",
            "config": {"workload": "configs[2]: degree-2^%d commit (G1 MSM, %d terms) on %d x MI355X, SRS-range sharded, "
                                   "%d commitments per step in one batched pass" % (degree.bit_length() - 1, n, world, batch),
                       "degree": degree, "terms_per_gpu": n_mine, "commitments_per_step": batch, "recoding": cfg["recoding"],
                       "digit_bits": cfg["digit_bits"], "table_levels": cfg["table_levels"], "buckets": cfg["buckets"],
                       "table_gib": round(cfg["table_levels"] * n_mine * 128 / 2**30, 2), "stream_slots": slots,
                       "bit_exact_vs_golden": ok},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_bucket_accumulate", "avg_kernel_ms": avg_accum_ms,
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "note": "bound by integer VALU issue, not by HBM, by construction: see valu"},
            "valu": {"achieved_Tmad_s": tmad, "peak_Tmad_s": VALU_MAD_PEAK_T, "frac": tmad / VALU_MAD_PEAK_T,
                     "unit": "1e12 v_mad_u64_u32/s", "mixed_additions_per_launch": madds,
                     # time the kernel's own VALU instruction mix needs if the SIMDs issued back to back
                     "issue_model_ms": issue_ms, "issue_model_frac": issue_ms / avg_accum_ms if avg_accum_ms > 0 else 0.0,
                     "note": "bound by total VALU issue, not by the multiply-adds alone: 2770 v_mad_u64_u32 + ~3580 "
                             "other VALU instructions per mixed addition (the v_addc carry per multiply-add is 2770 of them)",
                     **valu_pmc},
            "phase_ms": {k: sum(v) / len(v) for k, v in phase_ms.items()},
            "opening_proofs_per_sec": proofs_per_s,
            "quotient_ms": quotient_ms,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(eng, limbs, min(args.cpu_sample, n))
        print(json.dumps(line))
    barrier()
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
