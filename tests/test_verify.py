"""kzg_verify_proof (host-side pairing check of the product library; reference src/polynomial.rs:276-294)
against the independent Python pairing twin and on tampered inputs.  No GPU involved."""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import kzg_poly_commit_exploration_amd as K  # noqa: E402

R_FP = 1 << 384


def _fp_mont_limbs(v, P):
    m = v * R_FP % P
    return [(m >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(6)]


def _p2_limbs(pt, P):
    """affine G2 point of the twin ((xa, xb), (ya, yb)) -> blst_p2 (Jacobian, z = 1), Montgomery limbs"""
    (xa, xb), (ya, yb) = pt
    w = _fp_mont_limbs(xa, P) + _fp_mont_limbs(xb, P) + _fp_mont_limbs(ya, P) + _fp_mont_limbs(yb, P) \
        + _fp_mont_limbs(1, P) + _fp_mont_limbs(0, P)
    return np.array(w, dtype=np.uint64)


def _p1(twin, oracle, pt):
    """affine G1 point of the twin (or INF) -> G1Point"""
    if pt is twin.INF:
        return K.G1Point(np.zeros(18, dtype=np.uint64))
    return K.G1Point.uncompress(twin.g1_compress(pt))


def test_verify_proof_accepts_and_rejects_like_the_pairing_twin(twin, oracle):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pairing_twin as PT

    rnd = random.Random(2024)
    P, R = twin.P, twin.R
    for trial in range(3):
        s = rnd.randrange(1, R)
        secret_be = s.to_bytes(32, "big")
        coeffs = [rnd.randrange(R) for _ in range(rnd.randrange(2, 7))]
        z = rnd.randrange(R)
        y = sum(c * pow(z, i, R) for i, c in enumerate(coeffs)) % R
        commitment = twin.commit_shortcut(coeffs, secret_be)
        proof = twin.proof_shortcut(coeffs, z, y, secret_be)
        assert PT.verify_proof(commitment, proof, z, y, secret_be)
        s_g2 = _p2_limbs(PT.g2_mul(PT.G2, s), P)
        c1, p1 = _p1(twin, oracle, commitment), _p1(twin, oracle, proof)
        ev = K.Evaluation(K.Scalar(z), K.Scalar(y))
        assert ev.verify_proof(p1, c1, s_g2) is True
        # the reference's equation rejects every one of these
        assert K.Evaluation(K.Scalar(z), K.Scalar((y + 1) % R)).verify_proof(p1, c1, s_g2) is False
        assert K.Evaluation(K.Scalar((z + 1) % R), K.Scalar(y)).verify_proof(p1, c1, s_g2) is False
        assert ev.verify_proof(c1, p1, s_g2) is False                      # proof and commitment swapped
        other = _p2_limbs(PT.g2_mul(PT.G2, (s + 1) % R), P)
        assert ev.verify_proof(p1, c1, other) is False                     # wrong setup
    # constant polynomial: proof is the point at infinity
    secret_be = (12345).to_bytes(32, "big")
    commitment = twin.commit_shortcut([7], secret_be)
    s_g2 = _p2_limbs(PT.g2_mul(PT.G2, 12345), P)
    inf = K.G1Point(np.zeros(18, dtype=np.uint64))
    assert K.Evaluation(K.Scalar(99), K.Scalar(7)).verify_proof(inf, _p1(twin, oracle, commitment), s_g2) is True
    assert K.Evaluation(K.Scalar(99), K.Scalar(8)).verify_proof(inf, _p1(twin, oracle, commitment), s_g2) is False


def test_verify_proof_batch_matches_single_checks(twin, oracle):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pairing_twin as PT

    rnd = random.Random(77)
    P, R = twin.P, twin.R
    s = rnd.randrange(1, R)
    secret_be = s.to_bytes(32, "big")
    s_g2 = _p2_limbs(PT.g2_mul(PT.G2, s), P)
    cs, ps, zs, ys, want = [], [], [], [], []
    for i in range(6):
        coeffs = [rnd.randrange(R) for _ in range(rnd.randrange(2, 6))]
        z = rnd.randrange(R)
        y = sum(c * pow(z, k, R) for k, c in enumerate(coeffs)) % R
        good = i % 3 != 1
        cs.append(_p1(twin, oracle, twin.commit_shortcut(coeffs, secret_be)))
        ps.append(_p1(twin, oracle, twin.proof_shortcut(coeffs, z, y, secret_be)))
        zs.append(K.Scalar(z))
        ys.append(K.Scalar(y if good else (y + 5) % R))
        want.append(good)
    assert K.verify_proof_batch(cs, ps, zs, ys, s_g2) == want
    assert K.verify_proof_batch([], [], [], [], s_g2) == []


def test_verify_proof_rejects_a_g2_point_off_the_curve(twin):
    P = twin.P
    bad = _p2_limbs(((1, 2), (3, 4)), P)
    g = K.G1Point(np.zeros(18, dtype=np.uint64))
    try:
        K.verify_proof(g, g, K.Scalar(1), K.Scalar(1), bad)
        raise AssertionError("accepted a malformed G2 point")
    except K.KzgError as e:
        assert e.status == K.KZG_ERR_INVALID_ARG


def test_srs_g2_at_matches_the_pairing_twin(twin):
    """kzg_srs_g2_at(secret, k) = [s^k mod r]G2 as a blst_p2 with z = 1 (reference src/trusted_setup.rs:40-53, 64-72):
    index 0 is the generator, index 1 what verify_proof reads (src/polynomial.rs:284); secrets above r are reduced."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pairing_twin as PT

    P, R = twin.P, twin.R
    rnd = random.Random(36)
    secrets = [twin.BENCH_SECRET_BE, (R - 1).to_bytes(32, "big"), (1).to_bytes(32, "big"), bytes([0xFF] * 32)] + \
              [bytes(rnd.randrange(256) for _ in range(32)) for _ in range(3)]
    for secret in secrets:
        s = int.from_bytes(secret, "big") % R
        for k in (0, 1, 2, 7):
            want = _p2_limbs(PT.g2_mul(PT.G2, pow(s, k, R)), P)
            got = K.srs_g2_at(secret, k)
            assert (got == want).all(), (secret.hex(), k)
    # [0]G2: the point at infinity, all-zero z
    assert not K.srs_g2_at(bytes(32), 1)[24:].any()


def test_round_trip_of_lib_rs_without_the_twin_supplying_g2(twin, oracle):
    """src/lib.rs:16-33 with every group element from the product library's host side: [s]G2 from kzg_srs_g2_at,
    the pairing check from kzg_verify_proof (commitment and proof from the known-secret shortcut: no GPU here)."""
    rnd = random.Random(99)
    R = twin.R
    secret = bytes(rnd.randrange(256) for _ in range(32))
    coeffs = [rnd.randrange(R) for _ in range(9)]
    z = rnd.randrange(R)
    y = sum(c * pow(z, i, R) for i, c in enumerate(coeffs)) % R
    c1 = _p1(twin, oracle, twin.commit_shortcut(coeffs, secret))
    p1 = _p1(twin, oracle, twin.proof_shortcut(coeffs, z, y, secret))
    s_g2 = K.srs_g2_at(secret, 1)
    assert K.Evaluation(K.Scalar(z), K.Scalar(y)).verify_proof(p1, c1, s_g2) is True
    assert K.Evaluation(K.Scalar(z), K.Scalar((y + 1) % R)).verify_proof(p1, c1, s_g2) is False
