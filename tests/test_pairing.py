"""CPU: the reference's own acceptance test -- commit, open, then Evaluation::verify_proof must hold
(src/lib.rs:16-33 `run_kate_proof_test`, src/polynomial.rs:276-294) -- run on the ORACLE's outputs with
the independent pairing restatement (oracle/pairing_twin.py).  This is the only kind of pin the reference
itself has for G1 results, so it is applied to the oracle here and to the GPU in test_gpu_parity.py."""
import random

import pytest


@pytest.fixture(scope="module")
def pairing():
    import pairing_twin

    return pairing_twin


def test_pairing_is_bilinear_and_non_degenerate(pairing, twin):
    assert pairing.g2_is_on_curve(pairing.G2)
    a, b = 0xC0FFEE, 0xFACADE
    lhs = [(twin.g1_mul(twin.G1, a), pairing.g2_mul(pairing.G2, b)),
           (twin.g1_neg(twin.g1_mul(twin.G1, a * b % twin.R)), pairing.G2)]
    assert pairing.pairing_product_is_one(lhs)
    lhs[1] = (twin.g1_neg(twin.g1_mul(twin.G1, (a * b + 1) % twin.R)), pairing.G2)
    assert not pairing.pairing_product_is_one(lhs)


@pytest.mark.parametrize("degree", [1, 2, 17, 130])
def test_oracle_proofs_verify_like_reference_lib_tests(oracle, twin, pairing, degree):
    # random i128 coefficients and point, as src/lib.rs:35-41, 51-75
    rnd = random.Random(1000 + degree)
    secret = bytes(rnd.randrange(256) for _ in range(32))
    ints = [rnd.randrange(-(1 << 127), 1 << 127) for _ in range(degree + 1)]
    import numpy as np

    c = np.stack([oracle.fr_from_i128(a) for a in ints])
    zi = rnd.randrange(-(1 << 127), 1 << 127)
    z = oracle.fr_from_i128(zi)
    y = oracle.poly_evaluate(c, z)
    srs = oracle.srs_g1(degree + 1, secret)
    rc, cm = oracle.commit_naive(c, srs)
    assert rc == 0
    rc, pf = oracle.generate_proof(c, z, y, srs)
    assert rc == 0
    C = twin.g1_uncompress(oracle.p1_compress(cm))
    pi = twin.g1_uncompress(oracle.p1_compress(pf))
    zv, yv = oracle.fr_to_int(z), oracle.fr_to_int(y)
    assert pairing.verify_proof(C, pi, zv, yv, secret)
    assert not pairing.verify_proof(C, pi, zv, (yv + 1) % twin.R, secret)
    assert not pairing.verify_proof(C, twin.g1_add(pi, twin.G1), zv, yv, secret)
