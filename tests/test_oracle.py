"""CPU: pins the oracle (C restatement) against the golden vectors, the Python big-int twin and the
reference's own Fr byte-semantics tests.  The reference holds no commitment/proof known answers
("parity unpinned" for G1 outputs), so G1 results are triangulated three ways:
C naive loop == C bucket method == [P(s)]G shortcut == golden.json (made by the independent twin)."""
import random

import numpy as np
import pytest


def test_public_constants(oracle, golden):
    g = oracle.p1_generator()
    assert oracle.p1_compress(g).hex() == golden["constants"]["compress_G"]
    assert oracle.p1_compress(oracle.p1_add(g, g)).hex() == golden["constants"]["compress_2G"]
    assert oracle.p1_compress(oracle.p1_mult(g, 2)).hex() == golden["constants"]["compress_2G"]
    inf = oracle.p1_zeros(1)[0]
    assert oracle.p1_compress(inf).hex() == golden["constants"]["compress_inf"]
    assert oracle.lib().oracle_p1_on_curve(g.ctypes.data) == 1


def test_group_law_self_consistency(oracle, twin):
    # shape of the reference's test_point_addition_and_scalar_multiplication (src/curves.rs:375-406)
    g = oracle.p1_generator()
    two_g = oracle.p1_mult(g, 2)
    assert oracle.p1_equal(two_g, oracle.p1_add(g, g))
    assert not oracle.p1_equal(two_g, g)
    # r * G = infinity, (r-1) G = -G
    assert oracle.lib().oracle_p1_is_inf(oracle.p1_mult(g, twin.R).ctypes.data) == 1
    neg = oracle.p1_mult(g, twin.R - 1)
    assert oracle.lib().oracle_p1_is_inf(oracle.p1_add(neg, g).ctypes.data) == 1
    rnd = random.Random(7)
    for _ in range(8):
        a, b = rnd.randrange(twin.R), rnd.randrange(twin.R)
        lhs = oracle.p1_add(oracle.p1_mult(g, a), oracle.p1_mult(g, b))
        assert oracle.p1_compress(lhs) == twin.g1_compress(twin.g1_mul(twin.G1, (a + b) % twin.R))


def test_compression_round_trip(oracle):
    # reference test_compression_and_serialization (src/curves.rs:408-451)
    g = oracle.p1_generator()
    p = oracle.p1_mult(g, 0xDEADBEEFCAFE)
    c = oracle.p1_compress(p)
    assert oracle.p1_equal(oracle.p1_uncompress(c), p)
    assert oracle.p1_compress(oracle.p1_uncompress(c)) == c


def test_fr_i128_semantics(oracle, twin):
    # reference test_i128_to_scalar_using_le (src/scalar.rs:350-368): a>0 -> a, a<=0 -> r-|a|
    rnd = random.Random(11)
    vals = [0, 1, -1, 5, -5, (1 << 127) - 1, -(1 << 127)] + [rnd.randrange(-(1 << 127), 1 << 127) for _ in range(50)]
    for a in vals:
        fr = oracle.fr_from_i128(a)
        expect = a if a > 0 else (twin.R - (-a)) % twin.R
        assert oracle.fr_to_int(fr) == expect
        assert expect == twin.fr_from_i128(a)


def test_fr_byte_orders_and_arithmetic(oracle, twin):
    # reference test_u128_to_scalar_using_le / _be (src/scalar.rs:370-389), test_pow (:403-414)
    rnd = random.Random(13)
    L = oracle.lib()
    for _ in range(20):
        v = rnd.randrange(1 << 256)
        a = oracle.fr_zeros(1)
        L.oracle_fr_from_be_bytes(a.ctypes.data, v.to_bytes(32, "big"))
        assert oracle.fr_to_int(a[0]) == v % twin.R
        b = oracle.fr_from_int(rnd.randrange(twin.R))
        out = oracle.fr_zeros(1)
        L.oracle_fr_mul(out.ctypes.data, a.ctypes.data, b.ctypes.data)
        assert oracle.fr_to_int(out[0]) == (v % twin.R) * oracle.fr_to_int(b) % twin.R
        L.oracle_fr_sub(out.ctypes.data, a.ctypes.data, b.ctypes.data)
        assert oracle.fr_to_int(out[0]) == ((v % twin.R) - oracle.fr_to_int(b)) % twin.R
        L.oracle_fr_pow(out.ctypes.data, a.ctypes.data, 9)
        assert oracle.fr_to_int(out[0]) == pow(v, 9, twin.R)
    # Montgomery memory image agrees with the twin's
    assert [int(x) for x in oracle.fr_from_int(5)] == twin.fr_to_mont_limbs(5)


def test_srs_against_golden(oracle, golden, twin):
    secret = bytes.fromhex(golden["secret_be"])
    srs = oracle.srs_g1(4, secret)
    for k in range(4):
        assert oracle.p1_compress(srs[k]).hex() == golden["srs_g1"][str(k)]
    for k in (100, 1024, 65535, 65536, 1 << 20, 1 << 22):
        assert oracle.p1_compress(oracle.srs_g1_at(k, secret)).hex() == golden["srs_g1"][str(k)]


@pytest.mark.parametrize("degree", [1, 2, 3, 100, 500, 1000, 1024])
def test_bench_vectors_small(oracle, golden, degree):
    secret = bytes.fromhex(golden["secret_be"])
    case = next(b for b in golden["bench"] if b["degree"] == degree)
    srs = oracle.srs_g1(degree + 1, secret)
    c = oracle.bench_coefficients(degree + 1)
    z = oracle.bench_input_point(degree)
    y = oracle.poly_evaluate(c, z)
    assert "%x" % oracle.fr_to_int(z) == case["z"]
    assert "%x" % oracle.fr_to_int(y) == case["y"]
    rc, cm = oracle.commit_naive(c, srs)
    assert rc == 0 and oracle.p1_compress(cm).hex() == case["commit"]
    rc, cp = oracle.commit_pippenger(c, srs, threads=4)
    assert rc == 0 and oracle.p1_compress(cp).hex() == case["commit"]
    assert oracle.p1_compress(oracle.commit_shortcut(c, secret)).hex() == case["commit"]
    rc, pf = oracle.generate_proof(c, z, y, srs)
    assert rc == 0 and oracle.p1_compress(pf).hex() == case["proof"]


def test_bench_vector_2500_shortcut_and_pippenger(oracle, golden):
    secret = bytes.fromhex(golden["secret_be"])
    case = next(b for b in golden["bench"] if b["degree"] == 2500)
    c = oracle.bench_coefficients(2501)
    assert oracle.p1_compress(oracle.commit_shortcut(c, secret)).hex() == case["commit"]
    srs = oracle.srs_g1(2501, secret)
    rc, cp = oracle.commit_pippenger(c, srs, threads=8)
    assert rc == 0 and oracle.p1_compress(cp).hex() == case["commit"]


def test_large_vectors_by_shortcut(oracle, golden):
    # 2^16: the O(N) Fr route in C must reproduce the twin's fixture
    secret = bytes.fromhex(golden["secret_be"])
    case = next(b for b in golden["bench"] if b["degree"] == 1 << 16)
    c = oracle.bench_coefficients((1 << 16) + 1)
    assert oracle.p1_compress(oracle.commit_shortcut(c, secret)).hex() == case["commit"]
    z = oracle.bench_input_point(1 << 16)
    assert "%x" % oracle.fr_to_int(oracle.poly_evaluate(c, z)) == case["y"]


def _edge_inputs(oracle, e):
    coeffs = oracle.fr_from_ints([int(c, 16) for c in e["coeffs"]]) if e["coeffs"] else oracle.fr_zeros(0)
    return coeffs, oracle.fr_from_int(int(e["z"], 16)), oracle.fr_from_int(int(e["y"], 16))


def test_edge_cases(oracle, golden):
    secret = bytes.fromhex(golden["secret_be"])
    msgs = {oracle.ERR_CONSTANT_POLY: "Unable to divide a constant polynomial",
            oracle.ERR_REMAINDER: "[divide_by_root] Fail to divide the polynomial by a root, constant terms do not add up"}
    for e in golden["edge"]:
        coeffs, z, y = _edge_inputs(oracle, e)
        srs = oracle.srs_g1(max(len(coeffs), 2), secret)
        rc, cm = oracle.commit_naive(coeffs, srs)
        assert rc == 0 and oracle.p1_compress(cm).hex() == e["commit"], e["name"]
        rc, pf = oracle.generate_proof(coeffs, z, y, srs)
        if e["error"] is None:
            assert rc == 0 and oracle.p1_compress(pf).hex() == e["proof"], e["name"]
        else:
            assert msgs[rc] == e["error"], e["name"]


def test_degree_too_high(oracle, golden):
    secret = bytes.fromhex(golden["secret_be"])
    srs = oracle.srs_g1(3, secret)
    c = oracle.fr_from_ints([1, 2, 3, 4])
    rc, _ = oracle.commit_naive(c, srs)
    assert rc == oracle.ERR_DEGREE_TOO_HIGH


def test_quotient_matches_twin(oracle, twin):
    rnd = random.Random(5)
    for n in (2, 3, 17, 200):
        ci = [rnd.randrange(twin.R) for _ in range(n)]
        z = rnd.randrange(twin.R)
        y = twin.poly_evaluate(ci, z)
        rc, q = oracle.quotient(oracle.fr_from_ints(ci), oracle.fr_from_int(z), oracle.fr_from_int(y))
        assert rc == 0
        assert [oracle.fr_to_int(r) for r in q] == twin.poly_divide_by_root(twin.poly_sub(ci, [y]), z)


def test_jacobian_representation_is_irrelevant(oracle, golden):
    # SRS entries come with arbitrary Z (src/trusted_setup.rs:54-62): rescaling must not change results
    secret = bytes.fromhex(golden["secret_be"])
    srs = oracle.srs_g1(9, secret)
    lam = np.array([3, 5, 7, 11, 13, 17], dtype=np.uint64)
    scaled = srs.copy()
    for i in range(len(srs)):
        oracle.lib().oracle_p1_rescale(scaled[i].ctypes.data, srs[i].ctypes.data, lam.ctypes.data)
    c = oracle.bench_coefficients(9)
    assert oracle.p1_compress(oracle.commit_naive(c, srs)[1]) == oracle.p1_compress(oracle.commit_naive(c, scaled)[1])
