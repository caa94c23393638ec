"""GPU parity tests: every call goes through the C-ABI (ctypes -> libkzg_mi355x.so -> HIP kernels)
and is compared, as 48-byte compressed G1 encodings (reference src/curves.rs:99-110) or Fr values,
against the CPU oracle on the same inputs and against tests/golden/golden.json.  Bit-exact: this is
integer work, there is no tolerance.  Shapes follow the reference's own round-trip tests
(src/lib.rs:16-33, 51-94) and benches (benches/polynomial_commitment.rs, evaluation_proof.rs)."""
import os
import random

import numpy as np
import pytest

import kzg_poly_commit_exploration_amd as K

pytestmark = pytest.mark.gpu

SMALL_N = 2501  # covers the reference's bench degrees [1, 100, 500, 1000, 2500]


def _bench_poly(oracle, degree):
    c = oracle.bench_coefficients(degree + 1)
    z = oracle.bench_input_point(degree)
    y = oracle.poly_evaluate(c, z)
    return c, K.Scalar.from_limbs(z), K.Scalar.from_limbs(y)


def _case(golden, degree):
    return next(b for b in golden["bench"] if b["degree"] == degree)


# ---------------------------------------------------------------- SRS

def test_srs_generation_matches_trusted_setup(engines, oracle, golden):
    eng = engines.bench_srs(SMALL_N)
    assert eng.srs_len() == SMALL_N
    secret = bytes.fromhex(golden["secret_be"])
    got = eng.srs_read(0, 64)
    want = oracle.srs_g1(64, secret)
    for i in range(64):
        assert oracle.p1_compress(got[i]) == oracle.p1_compress(want[i]), i
    for k in (100, 1024):
        assert oracle.p1_compress(eng.srs_read(k, 1)[0]).hex() == golden["srs_g1"][str(k)]
    assert K.G1Point(got[0]).compress().hex() == golden["constants"]["compress_G"]


def test_srs_load_strided_jacobian(oracle, golden):
    """kzg_srs_load_g1 takes &srs[0].g1 with stride size_of::<SetupArtifact>() = 432 and points with
    arbitrary Z (reference src/trusted_setup.rs:31-35, 54-62)."""
    secret = bytes.fromhex(golden["secret_be"])
    n = 300
    srs = oracle.srs_g1(n, secret)
    art = np.zeros((n, 54), dtype=np.uint64)  # 432 bytes per artifact: g1 then a dummy g2 area
    rnd = random.Random(1)
    for i in range(n):
        lam = np.array([rnd.randrange(1, 1 << 60) for _ in range(6)], dtype=np.uint64)
        lam[5] &= np.uint64(0x0FFFFFFFFFFFFFFF)
        tmp = oracle.p1_zeros(1)
        oracle.lib().oracle_p1_rescale(tmp.ctypes.data, srs[i].ctypes.data, lam.ctypes.data)
        art[i, :18] = tmp[0]
        art[i, 18:] = 0xABCDEF
    eng = K.Engine(0)
    try:
        eng.srs_load(art)  # stride = 432 from the array
        back = eng.srs_read(0, n)
        for i in range(n):
            assert oracle.p1_compress(back[i]) == oracle.p1_compress(srs[i]), i
        c = oracle.bench_coefficients(n)
        rc, want = oracle.commit_naive(c, srs)
        assert rc == 0
        assert eng.commit_limbs(c).compress() == oracle.p1_compress(want)
    finally:
        eng.close()


# ---------------------------------------------------------------- commit / open on the bench inputs

@pytest.mark.parametrize("degree", [1, 2, 3, 100, 500, 1000, 1024, 2500])
def test_commit_and_proof_bench_degrees(engines, oracle, golden, degree):
    eng = engines.bench_srs(SMALL_N)
    c, z, y = _bench_poly(oracle, degree)
    case = _case(golden, degree)
    assert "%x" % y.v == case["y"]
    poly = K.Polynomial.from_limbs(c)
    assert poly.commit(eng).compress().hex() == case["commit"]
    assert K.Evaluation(z, y).generate_proof(poly, eng).compress().hex() == case["proof"]
    assert poly.evaluate(z, eng).result == y


def test_commit_against_oracle_naive_loop(engines, oracle, golden):
    secret = bytes.fromhex(golden["secret_be"])
    eng = engines.bench_srs(SMALL_N)
    srs = oracle.srs_g1(400, secret)
    rnd = random.Random(21)
    for n in (1, 2, 7, 64, 65, 255, 400):
        c = oracle.fr_from_ints([rnd.randrange(K.R_MODULUS) for _ in range(n)])
        rc, want = oracle.commit_naive(c, srs[:n])
        assert rc == 0
        assert eng.commit_limbs(c).compress() == oracle.p1_compress(want), n


def test_kate_proof_random_i128_like_reference_lib_tests(engines, oracle, golden):
    """Shape of src/lib.rs:51-94: random i128 coefficients and points (about half negative, i.e.
    full-width scalars r - |a|), degrees 1, 2 and random below 2000."""
    secret = bytes.fromhex(golden["secret_be"])
    eng = engines.bench_srs(SMALL_N)
    srs = oracle.srs_g1(2000, secret)
    rnd = random.Random(2024)
    degrees = [1] * 3 + [2] * 3 + [rnd.randrange(1, 2000) for _ in range(4)]
    for d in degrees:
        ints = [rnd.randrange(-(1 << 127), 1 << 127) for _ in range(d + 1)]
        poly = K.Polynomial.try_from(ints)
        z = K.Scalar.from_i128(rnd.randrange(-(1 << 127), 1 << 127))
        ev = poly.evaluate(z, eng)
        c = np.stack([oracle.fr_from_i128(a) for a in ints])[: poly.degree() + 1]
        assert np.array_equal(c, poly.limbs)
        zo = oracle.fr_from_i128(int(z.v) if z.v < (1 << 127) else int(z.v) - K.R_MODULUS)
        assert oracle.fr_to_int(zo) == z.v
        yo = oracle.poly_evaluate(c, zo)
        assert oracle.fr_to_int(yo) == ev.result.v
        rc, cm = oracle.commit_naive(c, srs[: len(c)])
        assert rc == 0 and poly.commit(eng).compress() == oracle.p1_compress(cm)
        rc, pf = oracle.generate_proof(c, zo, yo, srs[: len(c)])
        assert rc == 0 and ev.generate_proof(poly, eng).compress() == oracle.p1_compress(pf)


def test_gpu_proofs_pass_the_references_pairing_check(oracle, twin):
    """The reference's acceptance criterion (src/lib.rs:16-33): commit -> evaluate -> generate_proof ->
    verify_proof == true, with a fresh random secret and random i128 inputs, the pairing evaluated by the
    independent restatement in oracle/pairing_twin.py.  A commitment or proof off by one bit fails."""
    import pairing_twin as PT

    rnd = random.Random(4242)
    for degree in (1, 2, 64, rnd.randrange(100, 2000)):
        secret = bytes(rnd.randrange(256) for _ in range(32))
        setup = K.SetupArtifactsGenerator(secret).take(degree + 1)
        try:
            poly = K.Polynomial.try_from([rnd.randrange(-(1 << 127), 1 << 127) for _ in range(degree + 1)])
            z = K.Scalar.from_i128(rnd.randrange(-(1 << 127), 1 << 127))
            commitment = poly.commit(setup)
            evaluation = poly.evaluate(z, setup)
            proof = evaluation.generate_proof(poly, setup)
            C = twin.g1_uncompress(commitment.compress())
            pi = twin.g1_uncompress(proof.compress())
            assert PT.verify_proof(C, pi, z.v, evaluation.result.v, secret), degree
            assert not PT.verify_proof(C, pi, z.v, (evaluation.result.v + 1) % K.R_MODULUS, secret)
            # and the library's own verify_proof (src/polynomial.rs:276-294) closes the loop of src/lib.rs:16-33
            s = int.from_bytes(secret, "big") % K.R_MODULUS
            (xa, xb), (ya, yb) = PT.g2_mul(PT.G2, s)
            mont = lambda v: [((v << 384) % twin.P >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(6)]  # noqa: E731
            s_g2 = mont(xa) + mont(xb) + mont(ya) + mont(yb) + mont(1) + mont(0)
            assert evaluation.verify_proof(proof, commitment, s_g2) is True
            wrong = K.Evaluation(z, K.Scalar((evaluation.result.v + 1) % K.R_MODULUS))
            assert wrong.verify_proof(proof, commitment, s_g2) is False
        finally:
            setup.close()


def test_commit_le_bytes_entry_point(engines, oracle, golden):
    eng = engines.bench_srs(SMALL_N)
    c = oracle.bench_coefficients(1025)
    le = b"".join(oracle.fr_to_int(r).to_bytes(32, "little") for r in c)
    assert eng.commit_le_bytes(le).compress().hex() == _case(golden, 1024)["commit"]


# ---------------------------------------------------------------- edge cases and errors

def test_edge_cases_from_golden(engines, golden):
    eng = engines.bench_srs(SMALL_N)
    for e in golden["edge"]:
        ints = [int(c, 16) for c in e["coeffs"]]
        limbs = K.scalars_to_limbs(ints)
        z, y = K.Scalar(int(e["z"], 16)), K.Scalar(int(e["y"], 16))
        assert eng.commit_limbs(limbs).compress().hex() == e["commit"], e["name"]
        if e["error"] is None:
            assert eng.open_limbs(limbs, z, y).compress().hex() == e["proof"], e["name"]
        else:
            with pytest.raises(K.KzgError) as ei:
                eng.open_limbs(limbs, z, y)
            assert str(ei.value) == e["error"], e["name"]


def test_degree_too_high(engines, oracle):
    eng = engines.bench_srs(SMALL_N)
    c = oracle.bench_coefficients(SMALL_N + 5)
    with pytest.raises(K.KzgError) as ei:
        eng.commit_limbs(c)
    assert ei.value.status == K.KZG_ERR_DEGREE_TOO_HIGH
    assert str(ei.value).startswith("Setup does not allow for commitment generation")
    # trailing zeros beyond the SRS are not a degree: the reference truncates at construction
    padded = np.concatenate([oracle.bench_coefficients(10), np.zeros((SMALL_N, 4), dtype=np.uint64)])
    assert eng.commit_limbs(padded).compress() == eng.commit_limbs(padded[:10]).compress()
    # a proof needs one point fewer than the commitment (quotient has degree d - 1)
    c2 = oracle.bench_coefficients(SMALL_N + 1)
    z = K.Scalar(12345)
    y = K.Scalar.from_limbs(oracle.poly_evaluate(c2, oracle.fr_from_int(12345)))
    rc, q = oracle.quotient(c2, oracle.fr_from_int(12345), oracle.fr_from_int(y.v))
    assert rc == 0
    assert eng.open_limbs(c2, z, y).compress() == eng.commit_limbs(q).compress()
    c3 = oracle.bench_coefficients(SMALL_N + 2)
    y3 = K.Scalar.from_limbs(oracle.poly_evaluate(c3, oracle.fr_from_int(12345)))
    with pytest.raises(K.KzgError) as ei:
        eng.open_limbs(c3, z, y3)
    assert ei.value.status == K.KZG_ERR_DEGREE_TOO_HIGH


def test_srs_can_be_replaced_on_a_live_engine(oracle, golden):
    """Second kzg_srs_generate_g1 / kzg_srs_load_g1 on the same context: new length, new window geometry."""
    secret = bytes.fromhex(golden["secret_be"])
    eng = K.Engine(0)
    try:
        for n in (100, 5000, 37):
            eng.srs_generate(secret, n)
            assert eng.srs_len() == n
            c = oracle.bench_coefficients(n)
            rc, want = oracle.commit_pippenger(c, eng.srs_read(0, n), threads=4)
            assert rc == 0 and eng.commit_limbs(c).compress() == oracle.p1_compress(want), n
        other = bytes(range(1, 33))
        srs = oracle.srs_g1(64, other)
        eng.srs_load(srs)
        c = oracle.bench_coefficients(64)
        rc, want = oracle.commit_naive(c, srs)
        assert rc == 0 and eng.commit_limbs(c).compress() == oracle.p1_compress(want)
        with pytest.raises(K.KzgError) as ei:
            eng.wait(0)  # nothing in flight
        assert ei.value.status == K.KZG_ERR_INVALID_ARG
    finally:
        eng.close()


def test_commit_before_srs_is_an_error():
    eng = K.Engine(0)
    try:
        with pytest.raises(K.KzgError) as ei:
            eng.commit_limbs(np.zeros((3, 4), dtype=np.uint64))
        assert ei.value.status == K.KZG_ERR_NO_SRS
    finally:
        eng.close()


# ---------------------------------------------------------------- quotient / evaluate element-wise

@pytest.mark.parametrize("n", [2, 3, 9, 255, 2048, 2049, 4097, 70001])
def test_quotient_elementwise(oracle, n):
    rnd = random.Random(n)
    eng = K.Engine(0)
    try:
        ci = [rnd.randrange(K.R_MODULUS) for _ in range(n)]
        c = K.scalars_to_limbs(ci)
        z = K.Scalar(rnd.randrange(K.R_MODULUS))
        y = eng.evaluate_limbs(c, z)
        yo = oracle.poly_evaluate(c, oracle.fr_from_int(z.v))
        assert oracle.fr_to_int(yo) == y.v
        rc, q_want = oracle.quotient(c, oracle.fr_from_int(z.v), yo)
        assert rc == 0
        q_got = eng.quotient_limbs(c, z, y)
        assert np.array_equal(q_got, q_want)
        with pytest.raises(K.KzgError) as ei:
            eng.quotient_limbs(c, z, K.Scalar(y.v + 1))
        assert ei.value.status == K.KZG_ERR_REMAINDER
    finally:
        eng.close()


# ---------------------------------------------------------------- adversarial trusted setups

@pytest.mark.parametrize("secret_int,label", [(1, "all points equal G"), (0, "points at infinity"),
                                              (K.R_MODULUS - 1, "alternating G, -G")])
def test_degenerate_secrets_exercise_exceptional_group_law(oracle, secret_int, label):
    """SRS[i] = [s^i]G with s in {1, 0, -1}: every bucket sees equal points (doubling branch),
    infinities, or opposite points (cancellation).  Bit-exactness needs the complete group law."""
    secret = secret_int.to_bytes(32, "big")
    n = 700
    eng = K.SetupArtifactsGenerator(secret).take(n)
    try:
        srs = oracle.srs_g1(n, secret)
        back = eng.srs_read(0, n)
        for i in (0, 1, 2, 3, n - 1):
            assert oracle.p1_compress(back[i]) == oracle.p1_compress(srs[i]), (label, i)
        rnd = random.Random(99)
        for coeffs in ([rnd.randrange(K.R_MODULUS) for _ in range(n)], [12345] * n, [1] * n,
                       [rnd.randrange(1 << 20) for _ in range(n)]):
            c = K.scalars_to_limbs(coeffs)
            rc, want = oracle.commit_pippenger(c, srs, threads=4)
            assert rc == 0
            assert eng.commit_limbs(c).compress() == oracle.p1_compress(want), label
    finally:
        eng.close()


def test_skewed_scalars_heavy_buckets(engines, oracle, golden):
    """All coefficients equal / tiny / r - small: one bucket per window takes every point."""
    secret = bytes.fromhex(golden["secret_be"])
    eng = engines.bench_srs(SMALL_N)
    srs = oracle.srs_g1(SMALL_N, secret)
    for coeffs in ([K.R_MODULUS - 1] * SMALL_N, [7] * SMALL_N, [0] * 100 + [3] * (SMALL_N - 100),
                   [(1 << 254) + 5] * SMALL_N):
        c = K.scalars_to_limbs(coeffs)
        rc, want = oracle.commit_pippenger(c, srs, threads=8)
        assert rc == 0
        assert eng.commit_limbs(c).compress() == oracle.p1_compress(want)


def test_randomized_differential_against_oracle(oracle, golden):
    """Seeded sweep over odd sizes and scalar distributions: exercises tile / bin / segment boundaries of
    the sort and of the segmented accumulation, sub-SRS lengths, and the quotient at every size."""
    secret = bytes.fromhex(golden["secret_be"])
    rnd = random.Random(20260101)
    r = K.R_MODULUS

    def gen(kind, n):
        if kind == "uniform":
            return [rnd.randrange(r) for _ in range(n)]
        if kind == "i128":
            return [K.Scalar.from_i128(rnd.randrange(-(1 << 127), 1 << 127)).v for _ in range(n)]
        if kind == "small":
            return [rnd.randrange(1 << 12) for _ in range(n)]
        if kind == "sparse":
            return [rnd.randrange(r) if rnd.random() < 0.05 else 0 for _ in range(n)]
        if kind == "two_values":
            a, b = rnd.randrange(r), rnd.randrange(r)
            return [a if rnd.random() < 0.5 else b for _ in range(n)]
        raise AssertionError(kind)

    for srs_n in (1, 2, 3, 129, 1000, 4097, 20011, 40009):
        eng = K.SetupArtifactsGenerator(secret).take(srs_n)
        try:
            srs = eng.srs_read(0, srs_n)
            for kind in ("uniform", "i128", "small", "sparse", "two_values"):
                n = srs_n if rnd.random() < 0.5 else rnd.randrange(1, srs_n + 1)
                ints = gen(kind, n)
                if ints[-1] == 0:
                    ints[-1] = 1  # keep the polynomial's length (the reference truncates trailing zeros)
                c = K.scalars_to_limbs(ints)
                rc, want = oracle.commit_pippenger(c, srs[:n], threads=8)
                assert rc == 0
                assert eng.commit_limbs(c).compress() == oracle.p1_compress(want), (srs_n, kind, n)
                if n >= 2:
                    z = K.Scalar(rnd.randrange(r))
                    y = eng.evaluate_limbs(c, z)
                    zo, yo = oracle.fr_from_int(z.v), oracle.fr_from_int(y.v)
                    assert oracle.fr_to_int(oracle.poly_evaluate(c, zo)) == y.v
                    rc, q = oracle.quotient(c, zo, yo)
                    assert rc == 0
                    rc, wantp = oracle.commit_pippenger(q, srs[: len(q)], threads=8) if len(q) else (0, np.zeros(18, dtype=np.uint64))
                    assert rc == 0
                    assert eng.open_limbs(c, z, y).compress() == oracle.p1_compress(wantp), (srs_n, kind, n)
        finally:
            eng.close()


# ---------------------------------------------------------------- 2^16 (BASELINE config 2)

def test_degree_2_16_commit_and_proof(engines, oracle, golden):
    d = 1 << 16
    eng = engines.bench_srs(d + 1)
    c, z, y = _bench_poly(oracle, d)
    case = _case(golden, d)
    assert "%x" % y.v == case["y"]
    got = eng.commit_limbs(c)
    assert got.compress().hex() == case["commit"]
    assert eng.open_limbs(c, z, y).compress().hex() == case["proof"]
    # the SRS itself, spot-checked against the oracle and the fixtures
    secret = bytes.fromhex(golden["secret_be"])
    assert oracle.p1_compress(eng.srs_read(65535, 2)[1]).hex() == golden["srs_g1"]["65536"]
    assert oracle.p1_compress(eng.srs_read(12345, 1)[0]) == oracle.p1_compress(oracle.srs_g1_at(12345, secret))
    # full-size oracle run of the same MSM (bucket method on the host cores)
    srs = eng.srs_read(0, d + 1)
    rc, want = oracle.commit_pippenger(c, srs, threads=8)
    assert rc == 0 and got.compress() == oracle.p1_compress(want)


def test_naf_recoding_is_bit_exact(oracle, golden, monkeypatch):
    """The opt-in width-c NAF recoding (one table level per scalar bit, odd digits, 2^(c-2) buckets of weight
    2b+1; csrc/msm_sort.hip) must give the same bytes as the default windows: golden 2^16 vectors, skewed and
    i128-style scalars (whose top digits pile up in a few buckets), a forced digit width."""
    secret = bytes.fromhex(golden["secret_be"])
    monkeypatch.setenv("KZG_MSM_RECODE", "naf")
    d = 1 << 16
    eng = K.SetupArtifactsGenerator(secret).take(d + 1)
    try:
        cfg = eng.msm_config()
        assert cfg["recoding"] == "naf" and cfg["table_levels"] == 255 and cfg["buckets"] == 1 << (cfg["digit_bits"] - 2)
        c, z, y = _bench_poly(oracle, d)
        case = _case(golden, d)
        assert eng.commit_limbs(c).compress().hex() == case["commit"]
        assert eng.open_limbs(c, z, y).compress().hex() == case["proof"]
    finally:
        eng.close()
    monkeypatch.setenv("KZG_MSM_C", "9")
    eng = K.SetupArtifactsGenerator(secret).take(SMALL_N)
    try:
        assert eng.msm_config()["digit_bits"] == 9
        srs = oracle.srs_g1(SMALL_N, secret)
        rnd = random.Random(7)
        cases = [[K.R_MODULUS - 1] * SMALL_N, [7] * SMALL_N, [(1 << 254) + 5] * SMALL_N,
                 [K.Scalar.from_i128(rnd.randrange(-(1 << 127), 1 << 127)).v for _ in range(SMALL_N)],
                 [rnd.randrange(K.R_MODULUS) for _ in range(SMALL_N)],
                 [(K.R_MODULUS - 1) // 2, (K.R_MODULUS + 1) // 2, 1, 0, K.R_MODULUS - 2] * 500]
        for coeffs in cases:
            c = K.scalars_to_limbs(coeffs)
            rc, want = oracle.commit_pippenger(c, srs[:len(coeffs)], threads=8)
            assert rc == 0
            assert eng.commit_limbs(c).compress() == oracle.p1_compress(want)
    finally:
        eng.close()
    monkeypatch.delenv("KZG_MSM_C")
    monkeypatch.setenv("KZG_MSM_RECODE", "windows")
    eng = K.SetupArtifactsGenerator(secret).take(SMALL_N)
    try:
        assert eng.msm_config()["recoding"] == "windows"
    finally:
        eng.close()


def test_one_bucket_takes_everything_at_full_size(engines, oracle, golden):
    """0/1-valued and constant polynomials (selector columns, bit vectors): every reference of a window falls
    into one bucket.  Exercises the tiled fine sort and all three levels of the reduce-by-key trees
    (csrc/msm_finalize.hip: 2^20 references -> 131072 segment partials -> 2048 chunks -> 32 groups -> 1)."""
    secret = bytes.fromhex(golden["secret_be"])
    rnd = random.Random(11)
    for d in (1 << 16, 1 << 20):
        n = d + 1
        eng = engines.bench_srs(n)
        cases = [[1] * n, [rnd.getrandbits(1) for _ in range(n)]]
        if d == 1 << 16:  # the big size keeps to two cases (host-side conversion time)
            cases.append([(3, 5, 1 << 60)[rnd.randrange(3)] for _ in range(n)])
        for vals in cases:
            c = oracle.fr_from_ints(vals)
            want = oracle.p1_compress(oracle.commit_shortcut(c, secret))
            assert eng.commit_limbs(c).compress() == want, (d, vals[:4])


def test_concurrent_host_threads_share_one_context(engines, oracle, golden):
    """The reference's callers include `cargo test` threads (re-entrant per context).  Four host
    threads hammer one engine through the synchronous entry points (ctypes drops the GIL inside the calls) while
    a second engine on the same device works beside them; every result must still be exact."""
    import threading

    secret = bytes.fromhex(golden["secret_be"])
    eng = engines.bench_srs(SMALL_N)
    srs = oracle.srs_g1(SMALL_N, secret)
    other = K.SetupArtifactsGenerator(secret).take(1001)
    rnd = random.Random(5)
    jobs = []
    for t in range(4):
        n = rnd.randrange(2, SMALL_N + 1)
        c = K.scalars_to_limbs([rnd.randrange(K.R_MODULUS) for _ in range(n)])
        z = oracle.fr_from_int(rnd.randrange(K.R_MODULUS))
        yv = oracle.poly_evaluate(c, z)
        rc, want_c = oracle.commit_naive(c, srs) if n <= 300 else oracle.commit_pippenger(c, srs, threads=2)
        assert rc == 0
        rc, want_p = oracle.generate_proof(c, z, yv, srs)
        assert rc == 0
        jobs.append((c, K.Scalar.from_limbs(z), K.Scalar.from_limbs(yv), oracle.p1_compress(want_c), oracle.p1_compress(want_p)))
    errors = []

    def worker(job):
        c, z, y, want_c, want_p = job
        try:
            for _ in range(6):
                if eng.commit_limbs(c).compress() != want_c:
                    errors.append("commit")
                if eng.open_limbs(c, z, y).compress() != want_p:
                    errors.append("open")
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    def side():
        try:
            c1 = oracle.bench_coefficients(1001)
            for _ in range(6):
                if other.commit_limbs(c1).compress().hex() != _case(golden, 1000)["commit"]:
                    errors.append("side commit")
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(j,)) for j in jobs] + [threading.Thread(target=side)]
    try:
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=300)
        assert not errors, errors[:3]
    finally:
        other.close()


def test_host_pointer_callers_overlap_on_the_stream_slots(engines, oracle, golden):
    """kzg_commit from four host threads at degree 2^20: the calls hold the context mutex only while they touch the
    slot table, so their jobs occupy stream slots of their own and pipeline (sort and reduction of one in the shadow of another's
    accumulation, uploads beside kernels).  Wall time of 4 x 6 threaded calls must be below 24 serial ones (measured:
    0.72-0.76 of it; the bound leaves room for a noisy box -- the rates themselves are bench.py's business)."""
    import threading
    import time

    d = 1 << 20
    eng = engines.bench_srs(d + 1)
    c, _, _ = _bench_poly(oracle, d)
    want = _case(golden, d)["commit"]
    assert eng.commit_limbs(c).compress().hex() == want  # warm-up (staging buffers, clocks)
    reps = 6
    t0 = time.perf_counter()
    for _ in range(4 * reps):
        assert eng.commit_limbs(c).compress().hex() == want
    serial = time.perf_counter() - t0
    errors = []

    def worker():
        try:
            for _ in range(reps):
                if eng.commit_limbs(c).compress().hex() != want:
                    errors.append("commit")
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker) for _ in range(4)]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    threaded = time.perf_counter() - t0
    assert not errors, errors[:3]
    print("host-pointer commits at 2^20: serial %.1f ms, 4 threads %.1f ms per call" % (1e3 * serial / (4 * reps), 1e3 * threaded / (4 * reps)))
    assert threaded < 0.92 * serial, (threaded, serial)
    # a fifth and sixth caller simply wait for a slot (they used to wait for the mutex): no KZG_ERR_BUSY
    threads = [threading.Thread(target=worker) for _ in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors[:3]


def test_zero_heavy_scalars_fill_the_accumulation_lanes(engines, oracle, golden):
    """The segment length follows the number of non-zero digits counted on the device: i128-style coefficients
    leave the upper windows empty, a sparse polynomial most of them; results and reference counts must agree."""
    secret = bytes.fromhex(golden["secret_be"])
    d = 1 << 16
    eng = engines.bench_srs(d + 1)
    srs = eng.srs_read(0, d + 1)
    rnd = random.Random(99)
    eng.set_timing(True)
    try:
        for coeffs in ([rnd.randrange(1 << 40) for _ in range(d + 1)],
                       [rnd.randrange(K.R_MODULUS) if i % 64 == 0 else 0 for i in range(d + 1)]):
            c = K.scalars_to_limbs(coeffs)
            rc, want = oracle.commit_pippenger(c, srs, threads=8)
            assert rc == 0
            assert eng.commit_limbs(c).compress() == oracle.p1_compress(want)
            refs = max(eng.times(s)["references"] for s in range(eng.num_slots()))
            levels = eng.msm_config()["table_levels"]
            assert 0 < refs < (d + 1) * levels // 4
    finally:
        eng.set_timing(False)


def test_sharded_commit_equals_unsharded(engines, oracle, golden):
    """The multi-GPU decomposition with K virtual shards on one device: rank g holds SRS slice
    [lo, hi) and commits the matching coefficient slice; the sum of partials is the commitment."""
    from kzg_poly_commit_exploration_amd.sharding import shard_range

    d = 1 << 16
    n = d + 1
    secret = bytes.fromhex(golden["secret_be"])
    c = oracle.bench_coefficients(n)
    parts = []
    for rank in range(4):
        lo, hi = shard_range(n, rank, 4)
        eng = K.Engine(0)
        try:
            eng.srs_generate(secret, hi - lo, first=lo)
            parts.append(eng.commit_limbs(c[lo:hi]))
        finally:
            eng.close()
    assert K.G1Point.sum(parts).compress().hex() == _case(golden, d)["commit"]


def test_range_sharded_opening_equals_unsharded(oracle, golden):
    """Openings sharded by SRS range (sharding.py): K virtual shards on one GPU, carries by the host
    recurrence, each shard opens its slice extended by the carry; the partial proofs add up to the proof."""
    from kzg_poly_commit_exploration_amd.sharding import opening_carries, shard_range, sharded_open_local

    secret = bytes.fromhex(golden["secret_be"])
    for d, k in ((1 << 16, 4), (1000, 3), (5, 8)):
        n = d + 1
        c, z, y = _bench_poly(oracle, d) if d != 5 else (oracle.bench_coefficients(6), K.Scalar(77), None)
        if y is None:
            y = K.Scalar.from_limbs(oracle.poly_evaluate(c, oracle.fr_from_int(77)))
        spans = [shard_range(n, r, k) for r in range(k)]
        engs = []
        try:
            hs = []
            for lo, hi in spans:
                if hi > lo:
                    e = K.Engine(0)
                    e.srs_generate(secret, hi - lo, first=lo)
                    engs.append(e)
                    hs.append(e.evaluate_limbs(c[lo:hi], z).v)
                else:
                    engs.append(None)
                    hs.append(0)
            carries, starts = opening_carries(hs, [hi - lo for lo, hi in spans], z.v)
            assert starts[0] == y.v  # P(z) reassembled from the slices
            parts = [sharded_open_local(e, c[lo:hi], carries[r], z, starts[r])
                     for r, ((lo, hi), e) in enumerate(zip(spans, engs)) if e is not None]
            got = K.G1Point.sum(parts)
        finally:
            for e in engs:
                if e is not None:
                    e.close()
        if d in (1 << 16, 1000):
            assert got.compress().hex() == _case(golden, d)["proof"], (d, k)
        else:
            srs = oracle.srs_g1(n, secret)
            rc, want = oracle.generate_proof(c, oracle.fr_from_int(z.v), oracle.fr_from_int(y.v), srs)
            assert rc == 0 and got.compress() == oracle.p1_compress(want)


def test_linearity_and_determinism(engines, oracle):
    eng = engines.bench_srs((1 << 16) + 1)
    rnd = random.Random(17)
    n = 50000
    a = [rnd.randrange(K.R_MODULUS) for _ in range(n)]
    b = [rnd.randrange(K.R_MODULUS) for _ in range(n)]
    ca, cb = eng.commit_limbs(K.scalars_to_limbs(a)), eng.commit_limbs(K.scalars_to_limbs(b))
    cab = eng.commit_limbs(K.scalars_to_limbs([x + y for x, y in zip(a, b)]))
    assert K.G1Point.sum([ca, cb]) == cab
    assert eng.commit_limbs(K.scalars_to_limbs(a)).compress() == ca.compress()  # atomics order is irrelevant


def test_pipelined_slots_device_resident(engines, oracle, golden):
    d = 1 << 16
    eng = engines.bench_srs(d + 1)
    c, z, y = _bench_poly(oracle, d)
    case = _case(golden, d)
    dptr = eng.dev_alloc(c.nbytes)
    try:
        eng.dev_upload(dptr, c)
        slots = eng.num_slots()
        for s in range(slots):
            if s % 2 == 0:
                eng.commit_submit(s, dptr, d + 1)
            else:
                eng.open_submit(s, dptr, d + 1, z, y)
        with pytest.raises(K.KzgError) as ei:
            eng.commit_submit(0, dptr, d + 1)
        assert ei.value.status == K.KZG_ERR_BUSY
        for s in range(slots):
            got = eng.wait(s).compress().hex()
            assert got == (case["commit"] if s % 2 == 0 else case["proof"])
    finally:
        eng.dev_free(dptr)


def test_wait_entry_points_reject_the_wrong_job_kind(oracle, golden):
    """A slot filled by a batched submit must be collected by the batched wait and vice versa: the final buffer of a
    batch is laid out [section][polynomial], reading it as a single job would return a wrong point with KZG_OK.
    A rejected wait leaves the job in the slot; the right wait then returns the right answer."""
    secret = bytes.fromhex(golden["secret_be"])
    n = 1500
    eng = K.SetupArtifactsGenerator(secret).take(n)
    try:
        assert eng.set_max_batch(2) == 2
        rnd = random.Random(5)
        polys = [K.scalars_to_limbs([rnd.randrange(K.R_MODULUS) for _ in range(n)]) for _ in range(2)]
        want = [eng.commit_limbs(p).compress() for p in polys]
        flat = np.concatenate(polys)
        dptr = eng.dev_alloc(flat.nbytes)
        try:
            eng.dev_upload(dptr, flat)
            # batched commit, collected with the single wait
            eng.commit_batch_submit(0, dptr, n, 2)
            with pytest.raises(K.KzgError) as ei:
                eng.wait(0)
            assert ei.value.status == K.KZG_ERR_INVALID_ARG
            with pytest.raises(K.KzgError):
                eng.wait_batch(0, 1)  # wrong batch size
            assert [g.compress() for g in eng.wait_batch(0, 2)] == want
            # single commit, collected with the batched wait
            eng.commit_submit(1, dptr, n)
            with pytest.raises(K.KzgError) as ei:
                eng.wait_batch(1, 1)
            assert ei.value.status == K.KZG_ERR_INVALID_ARG
            assert eng.wait(1).compress() == want[0]
            # batched opening, collected with the single wait and with the batched commit wait
            zs = [K.Scalar(rnd.randrange(K.R_MODULUS)) for _ in polys]
            ys = [eng.evaluate_limbs(p, z) for p, z in zip(polys, zs)]
            zl = np.ascontiguousarray(np.stack([z.limbs() for z in zs]))
            yl = np.ascontiguousarray(np.stack([y.limbs() for y in ys]))
            import ctypes as C
            lib, h = eng._lib, eng._h
            assert lib.kzg_open_batch_submit(h, 2, C.c_void_p(dptr), n, 2, n, K._ptr(zl), K._ptr(yl)) == K.KZG_OK
            with pytest.raises(K.KzgError):
                eng.wait(2)
            with pytest.raises(K.KzgError):
                eng.wait_batch(2, 2)
            out = np.zeros((2, 18), dtype=np.uint64)
            st = np.zeros(2, dtype=np.int32)
            assert lib.kzg_wait_open_batch(h, 2, K._ptr(out), K._ptr(st), 2) == K.KZG_OK
            assert list(st) == [0, 0]
            for i in range(2):
                assert K.G1Point(out[i]).compress() == eng.open_limbs(polys[i], zs[i], ys[i]).compress()
        finally:
            eng.dev_free(dptr)
    finally:
        eng.close()


def test_failed_workspace_allocation_leaves_no_half_ready_context(oracle, golden, monkeypatch):
    """If the slot workspaces cannot be allocated the context must not keep slots_ready / n from an earlier setup:
    the next commit answers KZG_ERR_NO_SRS (or works, when the previous sizes could be restored) -- never a launch
    on freed buffers."""
    secret = bytes.fromhex(golden["secret_be"])
    n = 700
    c = oracle.bench_coefficients(n)
    eng = K.SetupArtifactsGenerator(secret).take(n)
    try:
        want = eng.commit_limbs(c).compress()
        # (a) growing the batch fails: the previous size is restored and the engine keeps working
        monkeypatch.setenv("KZG_TEST_FAIL_SLOT_ALLOC", "2")
        with pytest.raises(K.KzgError) as ei:
            eng.set_max_batch(4)
        assert ei.value.status == K.KZG_ERR_HIP
        assert eng.max_batch() == 1
        assert eng.commit_limbs(c).compress() == want
        # (b) every setup fails: loading an SRS reports the error and leaves the context without one
        monkeypatch.setenv("KZG_TEST_FAIL_SLOT_ALLOC", "1")
        with pytest.raises(K.KzgError) as ei:
            eng.srs_generate(secret, n)
        assert ei.value.status == K.KZG_ERR_HIP
        assert eng.srs_len() == 0
        with pytest.raises(K.KzgError) as ei:
            eng.commit_limbs(c)
        assert ei.value.status == K.KZG_ERR_NO_SRS
        # (c) and recovers once allocations succeed again
        monkeypatch.delenv("KZG_TEST_FAIL_SLOT_ALLOC")
        eng.srs_generate(secret, n)
        assert eng.commit_limbs(c).compress() == want
    finally:
        eng.close()


def test_affine_pair_front_end_is_bit_exact(golden):
    """KZG_ACCUM_PAIRS=1 (msm_accum.hip: pairs of references added in affine coordinates with one shared safegcd
    inversion per lane, then ONE mixed addition per pair) is opt-in -- it measured slower -- but stays bit-exact:
    golden commitment / proof at 2^16, a polynomial with all coefficients equal (every pair is a doubling or meets the
    same bucket's running sum) and degenerate SRS secrets (all points equal, opposite, at infinity), against the
    default path in the same process image.  The switch is read once per process, hence the child process."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = r"""
import json, os, sys
sys.path.insert(0, %r)
import numpy as np
import kzg_poly_commit_exploration_amd as K
golden = json.load(open(os.path.join(%r, "tests", "golden", "golden.json")))
secret = bytes.fromhex(golden["secret_be"])
d = 1 << 16
r = K.R_MODULUS
vals, p5 = [], 1
for _ in range(d + 1):
    vals.append((p5 + 10) %% r)
    p5 = p5 * 5 %% r
c = K.scalars_to_limbs(vals)
z = K.Scalar((pow(5, d, r) + 20) %% r)
out = {}
eng = K.SetupArtifactsGenerator(secret).take(d + 1)
y = eng.evaluate_limbs(c, z)
out["commit"] = eng.commit_limbs(c).compress().hex()
out["proof"] = eng.open_limbs(c, z, y).compress().hex()
out["ones"] = eng.commit_limbs(K.scalars_to_limbs([1] * (d + 1))).compress().hex()
out["small"] = eng.commit_limbs(K.scalars_to_limbs([(i * 7919) %% 5 for i in range(d + 1)])).compress().hex()
eng.close()
for name, s in (("one", 1), ("zero", 0), ("minus_one", r - 1)):
    e2 = K.SetupArtifactsGenerator(s.to_bytes(32, "big")).take(40000)
    out["secret_" + name] = e2.commit_limbs(c[:40000]).compress().hex()
    e2.close()
print(json.dumps(out))
""" % (root, root)
    res = {}
    for mode in ("0", "1"):
        # few lanes -> long segments (the front end only runs on segments of at least 16 references)
        env = dict(os.environ, KZG_ACCUM_PAIRS=mode, KZG_ACCUM_LANES="4096")
        p = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600, env=env)
        assert p.returncode == 0, p.stderr[-2000:]
        res[mode] = __import__("json").loads(p.stdout.strip().splitlines()[-1])
    case = _case(golden, 1 << 16)
    assert res["1"]["commit"] == case["commit"] and res["1"]["proof"] == case["proof"]
    assert res["1"] == res["0"]


def test_small_jobs_one_launch_and_general_path_agree(oracle, golden):
    """Jobs of up to 65536 references run accumulation, finalisation, long-bucket trees and both reduction stages as
    phases of ONE launch (msm_finalize.hip: k_small_msm); KZG_SMALL_MSM=0 sends them through the general kernels
    instead (group finalisation, two-stage tree launch).  Both must give the oracle's commitment: ragged sizes around
    the workgroup and threshold boundaries, uniform and skewed coefficients (one bucket takes every point: the
    long-bucket phase inside the launch; at 101 terms and below the direct form without an accumulation phase and its
    whole-workgroup buckets), a degenerate secret (every SRS point equal: doublings in every tree)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = r"""
import json, os, sys
sys.path.insert(0, %r)
import kzg_poly_commit_exploration_amd as K
golden = json.load(open(os.path.join(%r, "tests", "golden", "golden.json")))
secret = bytes.fromhex(golden["secret_be"])
r = K.R_MODULUS
out = {}
for n in (1, 2, 3, 17, 64, 101, 257, 1001, 2501, 2521, 2731, 4096, 5000):
    eng = K.SetupArtifactsGenerator(secret).take(n)
    vals, p5 = [], 1
    for _ in range(n):
        vals.append((p5 + 10) %% r)
        p5 = p5 * 5 %% r
    out["bench_%%d" %% n] = eng.commit_limbs(K.scalars_to_limbs(vals)).compress().hex()
    out["ones_%%d" %% n] = eng.commit_limbs(K.scalars_to_limbs([1] * n)).compress().hex()
    out["neg_%%d" %% n] = eng.commit_limbs(K.scalars_to_limbs([r - 3] * n)).compress().hex()
    out["sparse_%%d" %% n] = eng.commit_limbs(K.scalars_to_limbs([(i %% 7 == 0) * (i + 1) for i in range(n)])).compress().hex()
    same = sum(256 ** k for k in range(31))  # every 8-bit window holds the digit 1: ONE bucket takes all 31 n references
    out["same_%%d" %% n] = eng.commit_limbs(K.scalars_to_limbs([same] * n)).compress().hex()
    eng.close()
e2 = K.SetupArtifactsGenerator((1).to_bytes(32, "big")).take(1500)
out["secret_one"] = e2.commit_limbs(K.scalars_to_limbs([(i * i + 3) %% r for i in range(1500)])).compress().hex()
e2.close()
print(json.dumps(out))
""" % (root, root)
    res = {}
    for mode in ("1", "0"):
        env = dict(os.environ, KZG_SMALL_MSM=mode)
        p = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600, env=env)
        assert p.returncode == 0, p.stderr[-2000:]
        res[mode] = __import__("json").loads(p.stdout.strip().splitlines()[-1])
    assert res["1"] == res["0"]
    secret = bytes.fromhex(golden["secret_be"])
    r = K.R_MODULUS
    for n in (1, 17, 101, 2501, 2731, 5000):
        vals, p5 = [], 1
        for _ in range(n):
            vals.append((p5 + 10) % r)
            p5 = p5 * 5 % r
        for name, coeffs in (("bench", vals), ("ones", [1] * n), ("neg", [r - 3] * n),
                             ("sparse", [(i % 7 == 0) * (i + 1) for i in range(n)]),
                             ("same", [sum(256 ** k for k in range(31))] * n)):
            want = oracle.p1_compress(oracle.commit_shortcut(K.scalars_to_limbs(coeffs), secret))
            assert res["1"]["%s_%d" % (name, n)] == want.hex(), (name, n)
    want = oracle.p1_compress(oracle.commit_shortcut(K.scalars_to_limbs([(i * i + 3) % r for i in range(1500)]),
                                                     (1).to_bytes(32, "big")))
    assert res["1"]["secret_one"] == want.hex()


# ---------------------------------------------------------------- SRS wire / on-disk forms

def test_srs_binary_cache_affine_and_compressed_forms(engines, oracle, golden, tmp_path):
    """kzg_srs_save -> kzg_srs_load_file round trip equals kzg_srs_read_g1; kzg_srs_load_affine and the on-device
    decompression of kzg_srs_load_compressed (what setup.json holds: reference src/curves.rs:99-183) give the same
    resident SRS, the same commitments and proofs; malformed input is refused like blst_p1_uncompress refuses it."""
    d = 2500
    n = d + 1
    src = engines.bench_srs(SMALL_N)
    assert SMALL_N == n
    want = src.srs_read(0, n)
    c, z, y = _bench_poly(oracle, d)
    case = _case(golden, d)
    path = str(tmp_path / "srs.bin")
    src.srs_save(path)
    assert os.path.getsize(path) == 128 + 96 * n

    def check(eng):
        assert eng.srs_len() == n
        assert np.array_equal(eng.srs_read(0, n), want)
        assert eng.commit_limbs(c).compress().hex() == case["commit"]
        assert eng.open_limbs(c, z, y).compress().hex() == case["proof"]

    for make in (lambda: K.Engine(0), lambda: K.Engine(devices=[0, 0, 0])):
        eng = make()
        try:
            eng.srs_load_file(path)
            check(eng)
            eng.srs_load_affine(want[:, :12])
            check(eng)
            blob = b"".join(K.G1Point(p).compress() for p in want)
            eng.srs_load_compressed(blob)
            check(eng)
            # a point at infinity in the middle survives every form
            with_inf = bytearray(blob)
            with_inf[48 * 7:48 * 8] = bytes([0xC0]) + bytes(47)
            eng.srs_load_compressed(bytes(with_inf))
            back = eng.srs_read(0, 9)
            assert not back[7].any() and np.array_equal(back[8], want[8]) and np.array_equal(back[6], want[6])
            # malformed: x not on the curve / flag bits / x >= p
            for idx, mutate in ((5, lambda b: b[:47] + bytes([b[47] ^ 1])),      # (almost surely) not a curve point
                                (9, lambda b: bytes([b[0] & 0x7F]) + b[1:]),      # compressed flag cleared
                                (11, lambda b: bytes([0x9F]) + bytes([0xFF]) * 47)):  # x >= p
                bad = bytearray(blob)
                bad[48 * idx:48 * idx + 48] = mutate(bytes(blob[48 * idx:48 * idx + 48]))
                if idx == 5 and oracle_accepts_x(bytes(bad[48 * idx:48 * idx + 48])):
                    continue
                with pytest.raises(K.KzgError) as ei:
                    eng.srs_load_compressed(bytes(bad))
                assert ei.value.status == K.KZG_ERR_INVALID_ARG and ei.value.bad_index == idx
                assert eng.srs_len() == 0
        finally:
            eng.close()
    # a truncated or foreign file is refused
    with open(path, "rb") as fh:
        data = fh.read()
    bad_path = str(tmp_path / "bad.bin")
    eng = K.Engine(0)
    try:
        for blob2 in (data[:-96], data[:200] + bytes([data[200] ^ 0xFF]) + data[201:], b"NOTASRS" + data[7:]):
            with open(bad_path, "wb") as fh:
                fh.write(blob2)
            with pytest.raises(K.KzgError):
                eng.srs_load_file(bad_path)
    finally:
        eng.close()


def oracle_accepts_x(enc):
    """does the host's uncompress (on-curve check, as blst) accept this encoding?"""
    try:
        K.G1Point.uncompress(enc)
        return True
    except K.KzgError:
        return False


# ---------------------------------------------------------------- multi-device context (kzg_ctx_create_multi)

def _device_lists():
    """device lists for the multi-device tests: virtual slices on device 0 always; every visible device when there is
    more than one (the driver's 8-GPU node; a one-GPU box only has the virtual form)"""
    import torch

    lists = [[0, 0, 0, 0], [0, 0, 0]]
    if torch.cuda.device_count() > 1:
        lists.append(list(range(torch.cuda.device_count())))
    return lists


def test_multi_device_context_commit_and_open(engines, oracle, golden):
    """One context over several devices: SRS split by point range, commit and open sharded transparently (range-
    sharded opening with the carry recurrence on the host), partial sums gathered and added.  Same bytes as the golden
    vectors / the single-device engine, including shorter polynomials (empty slices), the degree boundary and the
    reference's three error texts."""
    secret = bytes.fromhex(golden["secret_be"])
    d = 2500
    c, z, y = _bench_poly(oracle, d)
    case = _case(golden, d)
    for devs in _device_lists():
        eng = K.Engine(devices=devs)
        try:
            assert eng.num_devices() == len(devs)
            eng.srs_generate(secret, d + 1)
            assert eng.srs_len() == d + 1
            # SRS read-back across slice boundaries
            per = (d + 1 + len(devs) - 1) // len(devs)
            got = eng.srs_read(per - 2, 5)
            for k in range(5):
                assert oracle.p1_compress(got[k]) == oracle.p1_compress(oracle.srs_g1_at(per - 2 + k, secret))
            assert eng.commit_limbs(c).compress().hex() == case["commit"]
            assert eng.open_limbs(c, z, y).compress().hex() == case["proof"]
            # shorter polynomials: the upper slices receive nothing
            for dd in (1, 100, 1000):
                cc, zz, yy = _bench_poly(oracle, dd)
                cs = _case(golden, dd)
                assert eng.commit_limbs(cc).compress().hex() == cs["commit"], (devs, dd)
                assert eng.open_limbs(cc, zz, yy).compress().hex() == cs["proof"], (devs, dd)
            # trailing zeros beyond the SRS are not a degree error; a non-zero coefficient there is
            longer = np.concatenate([c, np.zeros((7, 4), dtype=np.uint64)])
            assert eng.commit_limbs(longer).compress().hex() == case["commit"]
            assert eng.open_limbs(longer, z, y).compress().hex() == case["proof"]
            longer[-1, 0] = 1
            with pytest.raises(K.KzgError) as ei:
                eng.commit_limbs(longer)
            assert ei.value.status == K.KZG_ERR_DEGREE_TOO_HIGH
            # degree d + 1 can still be OPENED on d + 1 points (the quotient has degree d) but not committed
            c1 = oracle.bench_coefficients(d + 2)
            z1 = K.Scalar.from_limbs(oracle.bench_input_point(d + 1))
            y1 = K.Scalar.from_limbs(oracle.poly_evaluate(c1, oracle.fr_from_int(z1.v)))
            single = engines.bench_srs(d + 1)
            assert eng.open_limbs(c1, z1, y1).compress() == single.open_limbs(c1, z1, y1).compress()
            # the reference's error order: remainder before degree, constant polynomial, empty polynomial
            with pytest.raises(K.KzgError) as ei:
                eng.open_limbs(c, z, K.Scalar(y.v + 1))
            assert ei.value.status == K.KZG_ERR_REMAINDER
            const = K.scalars_to_limbs([7] + [0] * 40)
            assert eng.open_limbs(const, z, K.Scalar(7)).is_infinity()
            with pytest.raises(K.KzgError) as ei:
                eng.open_limbs(const, z, K.Scalar(8))
            assert ei.value.status == K.KZG_ERR_CONSTANT_POLY
            assert eng.commit_limbs(np.zeros((0, 4), dtype=np.uint64)).is_infinity()
            # random i128-like inputs against the single-device engine
            rnd = random.Random(2024)
            p = K.scalars_to_limbs([rnd.randrange(-(1 << 127), 1 << 127) for _ in range(d + 1)])
            zz = K.Scalar(rnd.randrange(K.R_MODULUS))
            yy = eng.evaluate_limbs(p, zz)
            assert eng.commit_limbs(p).compress() == single.commit_limbs(p).compress()
            assert eng.open_limbs(p, zz, yy).compress() == single.open_limbs(p, zz, yy).compress()
            # asynchronous / device-pointer entry points belong to one device
            with pytest.raises(K.KzgError) as ei:
                eng.dev_alloc(64)
            assert ei.value.status == K.KZG_ERR_INVALID_ARG
            if len(set(devs)) == len(devs) and len(devs) > 1:
                assert eng.rccl_exchanges() > 0  # distinct devices: the partials went through ncclAllGather
            else:
                assert eng.rccl_exchanges() == 0
        finally:
            eng.close()


def test_multi_device_context_exchange_over_rccl_world_of_one(oracle, golden, monkeypatch):
    """The RCCL leg of the exchange (ncclCommInitAll, grouped ncclAllGather on the context's streams, download, sum)
    on a communicator of ONE device: all a one-GPU box can form.  Distinct devices take exactly this code."""
    monkeypatch.setenv("KZG_MULTI_FORCE_RCCL", "1")
    secret = bytes.fromhex(golden["secret_be"])
    d = 1000
    c, z, y = _bench_poly(oracle, d)
    case = _case(golden, d)
    eng = K.Engine(devices=[0])
    try:
        eng.srs_generate(secret, d + 1)
        assert eng.commit_limbs(c).compress().hex() == case["commit"]
        assert eng.open_limbs(c, z, y).compress().hex() == case["proof"]
        assert eng.rccl_exchanges() == 2
    finally:
        eng.close()


def test_multi_device_context_degree_2_20(oracle, golden):
    """BASELINE config 4's shape on the devices at hand: one degree-2^20 commitment and opening sharded over 4 slices"""
    import torch

    secret = bytes.fromhex(golden["secret_be"])
    d = 1 << 20
    c, z, y = _bench_poly(oracle, d)
    case = _case(golden, d)
    ndev = torch.cuda.device_count()
    devs = list(range(ndev)) if ndev > 1 else [0, 0, 0, 0]
    eng = K.Engine(devices=devs)
    try:
        eng.srs_generate(secret, d + 1)
        assert eng.commit_limbs(c).compress().hex() == case["commit"]
        assert eng.open_limbs(c, z, y).compress().hex() == case["proof"]
    finally:
        eng.close()


def _eight_devices():
    """the 8-GPU node's devices when they are visible, otherwise eight virtual devices on GPU 0"""
    import torch

    ndev = torch.cuda.device_count()
    return list(range(8)) if ndev >= 8 else [0] * 8


def test_config4_degree_2_22_sharded_over_eight_slices(oracle, golden):
    """BASELINE config 4 at its own size through ONE library entry point: a degree-2^22 commitment and opening on a
    context of 8 devices (8 x 2^19 SRS points; virtual slices of one GPU when the box has one), partial sums through
    the exchange, against the golden vectors derived from the known secret."""
    secret = bytes.fromhex(golden["secret_be"])
    d = 1 << 22
    c, z, y = _bench_poly(oracle, d)
    case = _case(golden, d)
    devs = _eight_devices()
    eng = K.Engine(devices=devs)
    try:
        eng.srs_generate(secret, d + 1)
        assert eng.num_devices() == 8 and eng.srs_len() == d + 1
        assert oracle.p1_compress(eng.srs_read(d, 1)[0]).hex() == golden["srs_g1"][str(d)]
        assert eng.commit_limbs(c).compress().hex() == case["commit"]
        assert eng.open_limbs(c, z, y).compress().hex() == case["proof"]
        with pytest.raises(K.KzgError) as ei:
            eng.open_limbs(c, z, K.Scalar(y.v ^ 1))
        assert ei.value.status == K.KZG_ERR_REMAINDER
        if len(set(devs)) == 8:
            assert eng.rccl_exchanges() >= 2
    finally:
        eng.close()


def test_config4_exchange_over_rccl_with_raw_partials(oracle, golden, monkeypatch):
    """The same sharded calls with the partial sums forced through the RCCL leg (communicator of one on a one-GPU box):
    un-normalised Jacobian partials up, ncclAllGather, down, K-1 additions, one normalisation; also a batch of
    commitments sharded by range (one exchange for the whole batch)."""
    monkeypatch.setenv("KZG_MULTI_FORCE_RCCL", "1")
    secret = bytes.fromhex(golden["secret_be"])
    d = 2500
    c, z, y = _bench_poly(oracle, d)
    case = _case(golden, d)
    eng = K.Engine(devices=[0])
    try:
        eng.srs_generate(secret, d + 1)
        eng.set_max_batch(4)
        polys = [c] + [K.scalars_to_limbs([(7 * k + i) % 1000 for i in range(d + 1)]) for k in range(1, 6)]
        got = eng.commit_batch_host(polys)
        assert got[0].compress().hex() == case["commit"]
        for k in range(1, 6):
            assert got[k].compress() == eng.commit_limbs(polys[k]).compress()
        assert eng.rccl_exchanges() == 1 + 5
    finally:
        eng.close()


def test_range_split_context_batches(engines, oracle, golden):
    """kzg_commit_batch / kzg_open_batch on a range-split context of 3 virtual slices: every polynomial sharded,
    results equal to the single-device engine's, per-polynomial statuses kept."""
    secret = bytes.fromhex(golden["secret_be"])
    d = 2500
    c, z, y = _bench_poly(oracle, d)
    case = _case(golden, d)
    single = engines.bench_srs(d + 1)
    eng = K.Engine(devices=[0, 0, 0])
    try:
        eng.srs_generate(secret, d + 1)
        eng.set_max_batch(3)
        rnd = random.Random(77)
        polys = [c] + [K.scalars_to_limbs([rnd.randrange(K.R_MODULUS) for _ in range(d + 1)]) for _ in range(4)]
        zs = [z] + [K.Scalar(rnd.randrange(K.R_MODULUS)) for _ in range(4)]
        ys = [y] + [single.evaluate_limbs(p, zz) for p, zz in zip(polys[1:], zs[1:])]
        commits = eng.commit_batch_host(polys)
        assert commits[0].compress().hex() == case["commit"]
        for k in range(1, 5):
            assert commits[k].compress() == single.commit_limbs(polys[k]).compress()
        ys_bad = list(ys)
        ys_bad[2] = K.Scalar((ys[2].v + 1) % K.R_MODULUS)
        proofs = eng.open_batch_host(polys, zs, ys_bad)
        assert proofs[0].compress().hex() == case["proof"]
        assert isinstance(proofs[2], K.KzgError) and proofs[2].status == K.KZG_ERR_REMAINDER
        for k in (1, 3, 4):
            assert proofs[k].compress() == single.open_limbs(polys[k], zs[k], ys[k]).compress()
    finally:
        eng.close()


def test_host_pointer_batches_on_one_device(engines, oracle, golden):
    """kzg_commit_batch / kzg_open_batch on a single-device context: sub-batches pipelined through the stream slots
    (11 polynomials, at most 3 per pass), same bytes as one call per polynomial; constant and empty polynomials take
    the single-opening rules."""
    secret = bytes.fromhex(golden["secret_be"])
    d = 1000
    eng = K.SetupArtifactsGenerator(secret).take(d + 1)
    single = engines.bench_srs(d + 1)
    try:
        assert eng.set_max_batch(3) == 3
        rnd = random.Random(11)
        polys = [K.scalars_to_limbs([K.Scalar.from_i128(rnd.randrange(-(1 << 127), 1 << 127)).v for _ in range(d + 1)])
                 for _ in range(11)]
        zs = [K.Scalar(rnd.randrange(K.R_MODULUS)) for _ in polys]
        ys = [single.evaluate_limbs(p, zz) for p, zz in zip(polys, zs)]
        commits = eng.commit_batch_host(polys)
        proofs = eng.open_batch_host(polys, zs, ys)
        for k in range(11):
            assert commits[k].compress() == single.commit_limbs(polys[k]).compress(), k
            assert proofs[k].compress() == single.open_limbs(polys[k], zs[k], ys[k]).compress(), k
        assert eng.commit_batch_host(polys[:1])[0].compress() == commits[0].compress()
        # constant polynomials (n == 1): c0 == y -> infinity, otherwise the reference's error, per polynomial
        consts = [K.scalars_to_limbs([5]), K.scalars_to_limbs([6])]
        res = eng.open_batch_host(consts, [zs[0], zs[1]], [K.Scalar(5), K.Scalar(5)])
        assert res[0].is_infinity() and isinstance(res[1], K.KzgError) and res[1].status == K.KZG_ERR_CONSTANT_POLY
        with pytest.raises(K.KzgError) as ei:  # batches take truncated polynomials only
            eng.commit_batch_host([np.concatenate([polys[0], np.zeros((1, 4), dtype=np.uint64)])])
        assert ei.value.status == K.KZG_ERR_DEGREE_TOO_HIGH
    finally:
        eng.close()


def test_replicated_context_splits_batches_by_polynomial(engines, oracle, twin, golden):
    """SURVEY section 8(e), config 5's partition in miniature: 8 devices with the WHOLE SRS each (virtual devices on a
    one-GPU box), 64 openings in ONE kzg_open_batch call -- polynomial p on device p mod 8, nothing exchanged --
    verified one by one with the pairing check like src/lib.rs:16-33; single calls take the devices in turn."""
    import pairing_twin as PT

    rnd = random.Random(640)
    secret = bytes(rnd.randrange(256) for _ in range(32))
    s = int.from_bytes(secret, "big") % K.R_MODULUS
    n = 1025
    eng = K.Engine(devices=_eight_devices(), replicate=True)
    try:
        eng.srs_generate(secret, n)
        assert eng.num_devices() == 8 and eng.srs_len() == n
        assert eng.set_max_batch(4) == 4
        polys = [K.scalars_to_limbs([K.Scalar.from_i128(rnd.randrange(-(1 << 127), 1 << 127)).v for _ in range(n)])
                 for _ in range(64)]
        zs = [K.Scalar.from_i128(rnd.randrange(-(1 << 127), 1 << 127)) for _ in polys]
        ys = [eng.evaluate_limbs(p, z) for p, z in zip(polys, zs)]
        commitments = eng.commit_batch_host(polys)
        ys_claimed = list(ys)
        ys_claimed[41] = K.Scalar((ys[41].v + 1) % K.R_MODULUS)
        proofs = eng.open_batch_host(polys, zs, ys_claimed)
        assert isinstance(proofs[41], K.KzgError) and proofs[41].status == K.KZG_ERR_REMAINDER
        proofs[41] = eng.open_limbs(polys[41], zs[41], ys[41])
        s_g2 = K.srs_g2_at(secret, 1)  # setup_artifacts[1].g2 from the library's own host side
        (xa, xb), (ya, yb) = PT.g2_mul(PT.G2, s)
        mont = lambda v: [((v << 384) % twin.P >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(6)]  # noqa: E731
        assert list(s_g2) == mont(xa) + mont(xb) + mont(ya) + mont(yb) + mont(1) + mont(0)  # ... checked against the twin
        assert K.verify_proof_batch(commitments, proofs, zs, ys, s_g2) == [True] * 64
        # same bytes as a single-device engine on the same SRS
        one = K.SetupArtifactsGenerator(secret).take(n)
        try:
            for k in (0, 7, 8, 63):
                assert commitments[k].compress() == one.commit_limbs(polys[k]).compress()
                assert proofs[k].compress() == one.open_limbs(polys[k], zs[k], ys[k]).compress()
        finally:
            one.close()
        assert eng.rccl_exchanges() == 0
    finally:
        eng.close()


def test_config5_sixteen_degree_2_20_openings_over_eight_devices(oracle, twin, golden):
    """BASELINE config 5's shape at full degree through the by-polynomial path: 16 degree-2^20 openings (64 when the
    8-GPU node is visible) in one kzg_open_batch on a replicated-SRS context of 8 devices, then kzg_verify_proof_batch.
    The opening at the reference bench's point must equal the golden proof."""
    import pairing_twin as PT
    import torch

    d = 1 << 20
    n = d + 1
    case = _case(golden, d)
    secret = bytes.fromhex(golden["secret_be"])
    s = int.from_bytes(secret, "big") % K.R_MODULUS
    c, z0, y0 = _bench_poly(oracle, d)
    batch = 64 if torch.cuda.device_count() >= 8 else 16
    eng = K.Engine(devices=_eight_devices(), replicate=True)
    try:
        eng.srs_generate(secret, n)
        assert eng.set_max_batch(2) == 2
        r = K.R_MODULUS
        zs = [z0] + [K.Scalar((z0.v * (k + 2) + 12345 * k) % r) for k in range(1, batch)]
        ys = [eng.evaluate_limbs(c, z) for z in zs]
        assert ys[0] == y0
        flat = np.broadcast_to(np.ascontiguousarray(c, dtype=np.uint64).reshape(1, n, 4), (batch, n, 4))
        proofs = eng.open_batch_host(np.ascontiguousarray(flat), zs, ys)
        assert not any(isinstance(p, K.KzgError) for p in proofs)
        assert proofs[0].compress().hex() == case["proof"]
        assert len({p.compress() for p in proofs}) == batch
        commitment = K.G1Point.uncompress(bytes.fromhex(case["commit"]))
        s_g2 = K.srs_g2_at(secret, 1)
        assert K.verify_proof_batch([commitment] * batch, proofs, zs, ys, s_g2) == [True] * batch
    finally:
        eng.close()


# ---------------------------------------------------------------- 2^20 (BASELINE config 3, the bench workload)

def test_degree_2_20_commit_and_proof_golden(engines, oracle, golden):
    d = 1 << 20
    eng = engines.bench_srs(d + 1)
    secret = bytes.fromhex(golden["secret_be"])
    assert oracle.p1_compress(eng.srs_read(d, 1)[0]).hex() == golden["srs_g1"][str(d)]
    k = 777777
    assert oracle.p1_compress(eng.srs_read(k, 1)[0]) == oracle.p1_compress(oracle.srs_g1_at(k, secret))
    c, z, y = _bench_poly(oracle, d)
    case = _case(golden, d)
    assert "%x" % y.v == case["y"]
    assert eng.evaluate_limbs(c, z) == y
    assert eng.commit_limbs(c).compress().hex() == case["commit"]
    assert eng.open_limbs(c, z, y).compress().hex() == case["proof"]
    # size-independent property at full size: commit(P) = commit(P_low) + commit(P_high)
    half = (d + 1) // 2
    lo = c.copy()
    lo[half:] = 0
    hi = c.copy()
    hi[:half] = 0
    assert K.G1Point.sum([eng.commit_limbs(lo), eng.commit_limbs(hi)]).compress().hex() == case["commit"]
    # quotient * (x - z) + y = P, checked by committing: [Q(s)](s - z) + y = P(s) is what the verifier
    # pairs; here: element-wise against the oracle on a window of the quotient
    q = eng.quotient_limbs(c, z, y)
    rc, q_want = oracle.quotient(c, oracle.fr_from_int(z.v), oracle.fr_from_int(y.v))
    assert rc == 0 and np.array_equal(q, q_want)


def test_batched_commit_equals_individual(oracle, golden):
    """kzg_commit_batch_submit: B polynomials through one pass of the kernels (polynomial-major bucket
    ids) must give the same B commitments as B separate calls -- uniform, skewed and zero polynomials."""
    secret = bytes.fromhex(golden["secret_be"])
    n = 5000
    eng = K.SetupArtifactsGenerator(secret).take(n)
    try:
        b = eng.set_max_batch(6)
        assert 1 <= b <= 6 and eng.max_batch() == b
        rnd = random.Random(77)
        polys = [K.scalars_to_limbs([rnd.randrange(K.R_MODULUS) for _ in range(n)]),
                 oracle.bench_coefficients(n),
                 K.scalars_to_limbs([7] * n),
                 np.zeros((n, 4), dtype=np.uint64),
                 K.scalars_to_limbs([rnd.randrange(-(1 << 127), 1 << 127) for _ in range(n)]),
                 K.scalars_to_limbs([K.R_MODULUS - 1] * n)][:b]
        single = [eng.commit_limbs(p).compress() for p in polys]
        srs = eng.srs_read(0, n)
        for p, got in zip(polys[:3], single[:3]):
            rc, want = oracle.commit_pippenger(p, srs, threads=8)
            assert rc == 0 and got == oracle.p1_compress(want)
        batched = [g.compress() for g in eng.commit_batch_limbs(polys)]
        assert batched == single
        # a shorter batch and a batch of shorter polynomials on the same engine
        assert [g.compress() for g in eng.commit_batch_limbs(polys[:2])] == single[:2]
        short = [p[:1234] for p in polys[:3]]
        assert [g.compress() for g in eng.commit_batch_limbs(short)] == [eng.commit_limbs(p).compress() for p in short]
        with pytest.raises(K.KzgError):
            eng.commit_batch_limbs(polys + polys)  # more than max_batch
    finally:
        eng.close()


def test_batched_openings_equal_individual(oracle, golden):
    """kzg_open_batch_submit (BASELINE config 5 shape): per-polynomial z and y, per-polynomial status."""
    secret = bytes.fromhex(golden["secret_be"])
    n = 3000
    eng = K.SetupArtifactsGenerator(secret).take(n)
    try:
        b = eng.set_max_batch(4)
        rnd = random.Random(91)
        polys = [K.scalars_to_limbs([rnd.randrange(K.R_MODULUS) for _ in range(n)]) for _ in range(3)]
        polys.append(K.scalars_to_limbs([5] + [0] * (n - 1)))  # constant polynomial with trailing zeros
        polys = polys[:b]
        zs = [K.Scalar(rnd.randrange(K.R_MODULUS)) for _ in polys]
        ys = [eng.evaluate_limbs(p, z) for p, z in zip(polys, zs)]
        ys[1] = K.Scalar(ys[1].v + 1)  # wrong claimed value -> remainder error for polynomial 1 only
        got = eng.open_batch_limbs(polys, zs, ys)
        for i, (p, z, y) in enumerate(zip(polys, zs, ys)):
            try:
                want = eng.open_limbs(p, z, y)
            except K.KzgError as e:
                assert isinstance(got[i], K.KzgError) and got[i].status == e.status, i
                continue
            assert got[i].compress() == want.compress(), i
        assert isinstance(got[1], K.KzgError) and got[1].status == K.KZG_ERR_REMAINDER
        if b >= 4:
            assert got[3].is_infinity()  # (5 - 5) / (x - z) = 0
    finally:
        eng.close()


def test_batch_of_openings_and_their_verification(twin, golden):
    """BASELINE config 5 in miniature: a batch of openings on the GPU (kzg_open_batch_submit) and the pairing
    verification of every one of them on the host (kzg_verify_proof_batch), as src/lib.rs:16-33 does one by one."""
    import pairing_twin as PT

    rnd = random.Random(64)
    secret = bytes(rnd.randrange(256) for _ in range(32))
    s = int.from_bytes(secret, "big") % K.R_MODULUS
    n = 1025
    eng = K.SetupArtifactsGenerator(secret).take(n)
    try:
        b = eng.set_max_batch(8)
        polys = [K.scalars_to_limbs([K.Scalar.from_i128(rnd.randrange(-(1 << 127), 1 << 127)).v for _ in range(n)])
                 for _ in range(b)]
        zs = [K.Scalar.from_i128(rnd.randrange(-(1 << 127), 1 << 127)) for _ in polys]
        ys = [eng.evaluate_limbs(p, z) for p, z in zip(polys, zs)]
        commitments = eng.commit_batch_limbs(polys)
        proofs = eng.open_batch_limbs(polys, zs, ys)
        assert not any(isinstance(p, K.KzgError) for p in proofs)
        (xa, xb), (ya, yb) = PT.g2_mul(PT.G2, s)
        mont = lambda v: [((v << 384) % twin.P >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(6)]  # noqa: E731
        s_g2 = mont(xa) + mont(xb) + mont(ya) + mont(yb) + mont(1) + mont(0)
        assert K.verify_proof_batch(commitments, proofs, zs, ys, s_g2) == [True] * b
        ys_bad = list(ys)
        ys_bad[b // 2] = K.Scalar((ys[b // 2].v + 1) % K.R_MODULUS)
        verdicts = K.verify_proof_batch(commitments, proofs, zs, ys_bad, s_g2)
        assert verdicts == [i != b // 2 for i in range(b)]
    finally:
        eng.close()


def test_config5_batch_of_eight_degree_2_20_openings_and_their_verification(engines, oracle, twin, golden):
    """BASELINE config 5 at FULL size, one GPU's share: 8 openings of degree-2^20 polynomials in one
    kzg_open_batch_submit (kzg_set_max_batch(8) at 2^20 + 1 points), each at its own point.  The opening at the
    reference bench's point must equal the golden proof; all eight must pass the reference's pairing check
    (kzg_verify_proof_batch, src/polynomial.rs:276-294) against the golden commitment, and a wrong claimed value is
    rejected for exactly that opening."""
    import pairing_twin as PT

    d = 1 << 20
    n = d + 1
    case = _case(golden, d)
    secret = bytes.fromhex(golden["secret_be"])
    s = int.from_bytes(secret, "big") % K.R_MODULUS
    c, z0, y0 = _bench_poly(oracle, d)
    eng = K.SetupArtifactsGenerator(secret).take(n)
    try:
        assert eng.set_max_batch(8) == 8
        r = K.R_MODULUS
        zs = [z0] + [K.Scalar((z0.v * (k + 2) + 12345 * k) % r) for k in range(1, 8)]
        ys = [eng.evaluate_limbs(c, z) for z in zs]
        assert ys[0] == y0
        proofs = eng.open_batch_limbs([c] * 8, zs, ys)
        assert not any(isinstance(p, K.KzgError) for p in proofs)
        assert proofs[0].compress().hex() == case["proof"]
        assert len({p.compress() for p in proofs}) == 8
        commitment = K.G1Point.uncompress(bytes.fromhex(case["commit"]))
        (xa, xb), (ya, yb) = PT.g2_mul(PT.G2, s)
        mont = lambda v: [((v << 384) % twin.P >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(6)]  # noqa: E731
        s_g2 = mont(xa) + mont(xb) + mont(ya) + mont(yb) + mont(1) + mont(0)
        assert K.verify_proof_batch([commitment] * 8, proofs, zs, ys, s_g2) == [True] * 8
        ys_bad = list(ys)
        ys_bad[5] = K.Scalar((ys[5].v + 1) % r)
        assert K.verify_proof_batch([commitment] * 8, proofs, zs, ys_bad, s_g2) == [i != 5 for i in range(8)]
        # the batched commitments of the same 8 polynomials at full size
        commits = eng.commit_batch_limbs([c] * 8)
        assert all(g.compress().hex() == case["commit"] for g in commits)
    finally:
        eng.close()


# ---------------------------------------------------------------- 2^22 (BASELINE config 4 size, one GPU)

def test_degree_2_22_commit_and_proof_golden(oracle, golden):
    d = 1 << 22
    secret = bytes.fromhex(golden["secret_be"])
    eng = K.SetupArtifactsGenerator(secret).take(d + 1)
    try:
        assert oracle.p1_compress(eng.srs_read(d, 1)[0]).hex() == golden["srs_g1"][str(d)]
        c, z, y = _bench_poly(oracle, d)
        case = _case(golden, d)
        assert "%x" % y.v == case["y"]
        assert eng.commit_limbs(c).compress().hex() == case["commit"]
        assert eng.open_limbs(c, z, y).compress().hex() == case["proof"]
    finally:
        eng.close()


# ---------------------------------------------------------------- C++ host mirror over the same C-ABI

def test_cpp_mirror_example(twin):
    """examples/commit_open.cpp (include/kzg_mi355x.hpp): P = 1 + x over the bench SRS, opened at z = 1."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "commit_open")
    if not os.path.exists(exe):
        pytest.skip("examples/commit_open not built (run __graft_entry__.build())")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.split()
    s = twin.fr_from_be_bytes(twin.BENCH_SECRET_BE)
    assert lines[0:2] == ["degree", "1"]
    assert lines[2] == twin.g1_compress(twin.g1_mul(twin.G1, (1 + s) % twin.R)).hex()
    assert lines[3] == twin.g1_compress(twin.G1).hex()  # (P - 2)/(x - 1) = 1
