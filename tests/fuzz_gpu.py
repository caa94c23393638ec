"""Randomised differential fuzz of the C-ABI against the oracle (GPU; not collected by pytest).
    python tests/fuzz_gpu.py [seconds] [seed] [big]     ("big": SRS lengths 2^13 .. 2^17 instead of 1 .. 6000)
Random SRS secrets (degenerate ones included: 0, 1, r-1, small order-revealing values), random lengths, random
coefficient distributions chosen to hit the rare branches (equal points -> doubling inside the mixed addition,
P + (-P), infinity in the table, one-bucket inputs, zero-heavy inputs), commit / open / quotient / batch entry
points, both recodings.  Prints a summary line; exits non-zero on the first mismatch."""
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import kzg_poly_commit_exploration_amd as K  # noqa: E402
import oracle_ctypes as O  # noqa: E402

R = K.R_MODULUS


def gen_coeffs(rnd, n):
    kind = rnd.choice(["uniform", "small", "equal", "sparse", "i128", "near_r", "pow2", "bits", "half", "zeros_tail"])
    if kind == "uniform":
        v = [rnd.randrange(R) for _ in range(n)]
    elif kind == "small":
        v = [rnd.randrange(1 << rnd.choice([1, 4, 12, 33])) for _ in range(n)]
    elif kind == "equal":
        x = rnd.randrange(R)
        v = [x] * n
    elif kind == "sparse":
        v = [rnd.randrange(R) if rnd.random() < 0.03 else 0 for _ in range(n)]
    elif kind == "i128":
        v = [K.Scalar.from_i128(rnd.randrange(-(1 << 127), 1 << 127)).v for _ in range(n)]
    elif kind == "near_r":
        v = [R - 1 - rnd.randrange(1 << 10) for _ in range(n)]
    elif kind == "pow2":
        v = [1 << rnd.randrange(255) for _ in range(n)]
    elif kind == "bits":
        v = [rnd.getrandbits(1) for _ in range(n)]
    elif kind == "half":
        v = [(R - 1) // 2 + rnd.choice([0, 1, 2]) for _ in range(n)]
    else:
        v = [rnd.randrange(R) for _ in range(n)]
        for i in range(rnd.randrange(1, n + 1), n):
            v[i] = 0
    return kind, v


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20261004
    big = len(sys.argv) > 3 and sys.argv[3] == "big"
    rnd = random.Random(seed)
    t_end = time.time() + budget
    stats = {"engines": 0, "commits": 0, "opens": 0, "batches": 0, "errors_expected": 0}
    t_report = time.time() + 30
    while time.time() < t_end:
        if time.time() >= t_report:  # a silent GPU job is taken for hung
            print("fuzz progress", stats, flush=True)
            t_report = time.time() + 30
        sk = rnd.choice(["random"] * 6 + ["zero", "one", "minus_one", "two"])
        s = {"random": rnd.randrange(R), "zero": 0, "one": 1, "minus_one": R - 1, "two": 2}[sk]
        secret = s.to_bytes(32, "big")
        srs_n = rnd.choice([1, 2, 3, 5, 17, 64, 65, 255, 256, 257, 1000, 2049, 4097, rnd.randrange(1, 6000)])
        if big:
            srs_n = rnd.choice([1 << 13, (1 << 14) + 1, 40009, (1 << 16) + 1, rnd.randrange(1 << 13, 1 << 17)])
        if rnd.random() < 0.25:
            os.environ["KZG_MSM_RECODE"] = "naf"
        else:
            os.environ.pop("KZG_MSM_RECODE", None)
        if rnd.random() < 0.3:
            os.environ["KZG_MSM_C"] = str(rnd.randrange(8, 14))
        else:
            os.environ.pop("KZG_MSM_C", None)
        eng = K.SetupArtifactsGenerator(secret).take(srs_n)
        stats["engines"] += 1
        try:
            srs = O.srs_g1(srs_n, secret)
            assert np.array_equal(np.array([O.p1_compress(p) for p in eng.srs_read(0, min(srs_n, 8))]),
                                  np.array([O.p1_compress(p) for p in srs[:min(srs_n, 8)]]))
            for _ in range(rnd.randrange(2, 6)):
                n = srs_n if rnd.random() < 0.4 else rnd.randrange(1, srs_n + 1)
                kind, vals = gen_coeffs(rnd, n)
                c = K.scalars_to_limbs(vals)
                rc, want = (O.commit_naive(c, srs) if n <= 200 else O.commit_pippenger(c, srs, threads=8))
                assert rc == 0
                got = eng.commit_limbs(c)
                if got.compress() != O.p1_compress(want):
                    print("MISMATCH commit", sk, srs_n, n, kind, os.environ.get("KZG_MSM_RECODE"), os.environ.get("KZG_MSM_C"), seed)
                    sys.exit(1)
                stats["commits"] += 1
                # opening at a random point, true and (sometimes) false claimed value
                z = rnd.choice([0, 1, R - 1, rnd.randrange(R)])
                zl = O.fr_from_int(z)
                yl = O.poly_evaluate(c, zl)
                if rnd.random() < 0.2:
                    yl = O.fr_from_int((O.fr_to_int(yl) + 1) % R)
                rc, wantp = O.generate_proof(c, zl, yl, srs)
                try:
                    gotp = eng.open_limbs(c, K.Scalar.from_limbs(zl), K.Scalar.from_limbs(yl))
                    ok = rc == 0 and gotp.compress() == O.p1_compress(wantp)
                except K.KzgError as e:
                    ok = e.status == rc
                    stats["errors_expected"] += 1
                if not ok:
                    print("MISMATCH open", sk, srs_n, n, kind, rc, seed)
                    sys.exit(1)
                stats["opens"] += 1
            if srs_n >= 4 and rnd.random() < 0.5:
                b = eng.set_max_batch(rnd.randrange(2, 6))
                n = rnd.randrange(1, srs_n + 1)
                polys = [K.scalars_to_limbs(gen_coeffs(rnd, n)[1]) for _ in range(b)]
                for p in polys:  # batches take truncated polynomials: keep the top coefficient non-zero
                    if not p[-1].any():
                        p[-1, 0] = 1
                outs = eng.commit_batch_limbs(polys)
                for p, o in zip(polys, outs):
                    rc, want = O.commit_pippenger(p, srs, threads=8)
                    if o.compress() != O.p1_compress(want):
                        print("MISMATCH batch", sk, srs_n, n, seed)
                        sys.exit(1)
                stats["batches"] += 1
        finally:
            eng.close()
    print("fuzz ok", stats, "seed", seed)


if __name__ == "__main__":
    main()
