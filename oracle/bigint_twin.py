"""Python big-int twin of the reference's commit / open path.  TEST INFRASTRUCTURE ONLY.

This file is part of the ORACLE: it may be imported only from tests/, from
tests/golden/gen_golden.py, from __graft_entry__.smoke() and from bench.py's
cpu_baseline leg -- never from the product package.

Parity status: "parity unpinned" for G1 outputs.  The reference
(VGLoic/kzg-poly-commit-exploration @ 2025-09-19) delegates all arithmetic to
the un-vendored crate blst 0.3.15 (Cargo.toml:10, Cargo.lock:89-92) and holds
no known-answer commitment/proof bytes anywhere.  What
pins this twin instead:
  * public BLS12-381 constants (p, r, generator; reference README.md:30,
    src/scalar.rs:10) and the public ZCash-format encodings of G and 2G,
  * mathematics: the commitment is the unique group element sum c_i [s^i]G1
    and ZCash compression is canonical, so any correct implementation emits the
    bytes blst emits,
  * three-way triangulation (naive loop / this twin / the [P(s)]G shortcut,
    valid because the bench secret is known) against the C restatement in
    oracle/kzg_oracle.c.
The Fr byte semantics ARE pinned by the reference's own tests
(src/scalar.rs:350-389) and are restated in from_i128/to_le_bytes below.

Everything here is plain Python integers: slow, obviously correct, and written
independently of the C oracle so the two can check each other.
"""

# --- public curve constants (reference README.md:30, src/scalar.rs:10) -------------------------
P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
G1X = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
G1Y = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1
B_COEFF = 4

FP_R = 1 << 384  # Montgomery radix of blst_fp (6 x u64)
FR_R = 1 << 256  # Montgomery radix of blst_fr (4 x u64)

INF = None  # affine point at infinity


# --- Fr helpers: reference src/scalar.rs ------------------------------------------------------
def fr_from_i128(a: int) -> int:
    """Scalar::from(i128), src/scalar.rs:27-48: a > 0 -> a ; a <= 0 -> r - |a| (mod r)."""
    assert -(1 << 127) <= a < (1 << 127)
    if a > 0:
        return a % R
    return (R - (-a)) % R


def fr_from_be_bytes(b: bytes) -> int:
    """Scalar::from_be_bytes, src/scalar.rs:66-73 (blst_fr_from_hexascii reduces mod r)."""
    assert len(b) == 32
    return int.from_bytes(b, "big") % R


def fr_from_le_bytes(b: bytes) -> int:
    """Scalar::from_le_bytes, src/scalar.rs:54-61."""
    assert len(b) == 32
    return int.from_bytes(b, "little") % R


def fr_to_le_bytes(a: int) -> bytes:
    """Scalar::to_le_bytes, src/scalar.rs:83-93: canonical integer in [0, r), 32 LE bytes."""
    return (a % R).to_bytes(32, "little")


def fr_to_mont_limbs(a: int):
    """blst_fr in memory: 4 x u64 little-endian limbs of a*2^256 mod r (src/scalar.rs:7-8)."""
    v = (a % R) * FR_R % R
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def fr_from_mont_limbs(l) -> int:
    v = sum(int(x) << (64 * i) for i, x in enumerate(l))
    return v * pow(FR_R, -1, R) % R


def fp_to_mont_limbs(a: int):
    v = (a % P) * FP_R % P
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(6)]


def fp_from_mont_limbs(l) -> int:
    v = sum(int(x) << (64 * i) for i, x in enumerate(l))
    return v * pow(FP_R, -1, P) % P


# --- G1 affine arithmetic (textbook chord-and-tangent; None = infinity) --------------------------
def g1_is_on_curve(pt) -> bool:
    if pt is INF:
        return True
    x, y = pt
    return (y * y - x * x * x - B_COEFF) % P == 0


def g1_neg(pt):
    if pt is INF:
        return INF
    return (pt[0], (-pt[1]) % P)


def g1_add(a, b):
    """Complete affine add: what blst_p1_add_or_double computes (src/curves.rs:79-85)."""
    if a is INF:
        return b
    if b is INF:
        return a
    x1, y1 = a
    x2, y2 = b
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return INF
        lam = 3 * x1 * x1 * pow(2 * y1, -1, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    y3 = (lam * (x1 - x3) - y1) % P
    return (x3, y3)


# Jacobian internals so that scalar multiplication does not need one inversion per step.
def _jac_double(X, Y, Z):
    if Z == 0 or Y == 0:
        return (1, 1, 0)
    A = X * X % P
    Bq = Y * Y % P
    C = Bq * Bq % P
    D = 2 * ((X + Bq) * (X + Bq) - A - C) % P
    E = 3 * A % P
    F = E * E % P
    X3 = (F - 2 * D) % P
    Y3 = (E * (D - X3) - 8 * C) % P
    Z3 = 2 * Y * Z % P
    return (X3, Y3, Z3)


def _jac_add_affine(X1, Y1, Z1, x2, y2):
    if Z1 == 0:
        return (x2, y2, 1)
    Z1Z1 = Z1 * Z1 % P
    U2 = x2 * Z1Z1 % P
    S2 = y2 * Z1 * Z1Z1 % P
    H = (U2 - X1) % P
    r = (S2 - Y1) % P
    if H == 0:
        if r == 0:
            return _jac_double(X1, Y1, Z1)
        return (1, 1, 0)
    HH = H * H % P
    HHH = H * HH % P
    V = X1 * HH % P
    X3 = (r * r - HHH - 2 * V) % P
    Y3 = (r * (V - X3) - Y1 * HHH) % P
    Z3 = Z1 * H % P
    return (X3, Y3, Z3)


def _jac_to_affine(X, Y, Z):
    if Z == 0:
        return INF
    zi = pow(Z, -1, P)
    zi2 = zi * zi % P
    return (X * zi2 % P, Y * zi2 * zi % P)


def g1_mul(pt, k: int):
    """k*pt, k taken as a non-negative integer (blst_p1_mult with nbits=256, src/curves.rs:90-96)."""
    if pt is INF or k == 0:
        return INF
    assert k >= 0
    X, Y, Z = 1, 1, 0
    x2, y2 = pt
    for bit in bin(k)[2:]:
        X, Y, Z = _jac_double(X, Y, Z)
        if bit == "1":
            X, Y, Z = _jac_add_affine(X, Y, Z, x2, y2)
    return _jac_to_affine(X, Y, Z)


G1 = (G1X, G1Y)


def g1_compress(pt) -> bytes:
    """ZCash 48-byte encoding = blst_p1_compress (src/curves.rs:99-110).

    big-endian x; bit 7 of byte 0 = compressed flag; bit 6 = infinity; bit 5 = y is the
    lexicographically larger root (y > (p-1)/2)."""
    if pt is INF:
        return bytes([0xC0]) + bytes(47)
    x, y = pt
    out = bytearray(x.to_bytes(48, "big"))
    out[0] |= 0x80
    if y > (P - 1) // 2:
        out[0] |= 0x20
    return bytes(out)


def g1_uncompress(b: bytes):
    """Inverse of g1_compress (blst_p1_uncompress, src/curves.rs:131); p = 3 mod 4 sqrt."""
    assert len(b) == 48 and b[0] & 0x80
    if b[0] & 0x40:
        return INF
    x = int.from_bytes(bytes([b[0] & 0x1F]) + b[1:], "big")
    y2 = (x * x * x + B_COEFF) % P
    y = pow(y2, (P + 1) // 4, P)
    assert y * y % P == y2, "not on curve"
    if (y > (P - 1) // 2) != bool(b[0] & 0x20):
        y = P - y
    return (x, y)


def g1_to_blst_p1_limbs(pt, z: int = 1):
    """18 x u64 blst_p1 {x,y,z} Montgomery Jacobian limbs; z lets tests build non-trivial Z."""
    if pt is INF:
        return [0] * 18
    x, y = pt
    z %= P
    assert z != 0
    return fp_to_mont_limbs(x * z * z % P) + fp_to_mont_limbs(y * z * z * z % P) + fp_to_mont_limbs(z)


def g1_from_blst_p1_limbs(l):
    X = fp_from_mont_limbs(l[0:6])
    Y = fp_from_mont_limbs(l[6:12])
    Z = fp_from_mont_limbs(l[12:18])
    return _jac_to_affine(X, Y, Z)


# --- trusted setup: reference src/trusted_setup.rs -------------------------------------------------
def srs_scalars(secret_be: bytes, n: int):
    """s^0 .. s^(n-1) mod r; the secret is read big-endian (src/trusted_setup.rs:20-28, :50)."""
    s = fr_from_be_bytes(secret_be)
    out, cur = [], 1
    for _ in range(n):
        out.append(cur)
        cur = cur * s % R
    return out


def srs_g1(secret_be: bytes, n: int):
    """SRS[i].g1 = [s^i mod r] G1 (src/trusted_setup.rs:40-62)."""
    return [g1_mul(G1, k) for k in srs_scalars(secret_be, n)]


# --- polynomial path: reference src/polynomial.rs ---------------------------------------------------
def poly_truncate(coeffs):
    """TryFrom<Vec<Scalar>>, src/polynomial.rs:55-75: drop trailing zeros but keep index 0."""
    last = 0
    for i, v in enumerate(coeffs):
        if v % R != 0:
            last = i
    return [c % R for c in coeffs[: last + 1]]


def poly_from_constant(y):
    """From<Scalar>, src/polynomial.rs:78-89: zero -> [], else [y]."""
    return [] if y % R == 0 else [y % R]


def poly_degree(coeffs):
    return 0 if not coeffs else len(coeffs) - 1


def poly_evaluate(coeffs, x):
    """Polynomial::evaluate, src/polynomial.rs:112-123 (same value; Horner here)."""
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % R
    return acc


def poly_sub(a, b):
    """Polynomial::sub, src/polynomial.rs:128-145."""
    if len(a) > len(b):
        out = list(a)
        for i, rhs in enumerate(b):
            out[i] = (out[i] - rhs) % R
    else:
        out = [(-x) % R for x in b]
        for i, lhs in enumerate(a):
            out[i] = (lhs + out[i]) % R
    return poly_truncate(out)


class DivideError(Exception):
    pass


ERR_CONSTANT = "Unable to divide a constant polynomial"
ERR_REMAINDER = "[divide_by_root] Fail to divide the polynomial by a root, constant terms do not add up"
ERR_DEGREE = "Setup does not allow for commitment generation of the polynomial. The polynomial degree is too high."


def poly_divide_by_root(coeffs, z):
    """Polynomial::divide_by_root, src/polynomial.rs:150-195 (every branch restated)."""
    if not coeffs:
        return []
    if len(coeffs) == 1:
        if coeffs[0] % R == 0:
            return []
        raise DivideError(ERR_CONSTANT)
    d = len(coeffs) - 1
    q_desc = [coeffs[d] % R]
    for i in range(d - 1, 0, -1):
        q_desc.append((coeffs[i] + z * q_desc[d - i - 1]) % R)
    q = q_desc[::-1]
    if (-(z * q[0])) % R != coeffs[0] % R:
        raise DivideError(ERR_REMAINDER)
    return poly_truncate(q)


def commit_naive(coeffs, srs):
    """Polynomial::commit, src/polynomial.rs:200-215: N scalar-muls + N adds."""
    if poly_degree(coeffs) + 1 > len(srs):
        raise DivideError(ERR_DEGREE)
    acc = INF
    for c, pt in zip(coeffs, srs):
        acc = g1_add(acc, g1_mul(pt, c % R))
    return acc


def generate_proof(coeffs, z, y, srs):
    """Evaluation::generate_proof, src/polynomial.rs:260-269."""
    q = poly_divide_by_root(poly_sub(coeffs, poly_from_constant(y)), z)
    return commit_naive(q, srs)


# --- shortcut valid only when the secret is known -----------------------
def commit_shortcut(coeffs, secret_be: bytes):
    s = fr_from_be_bytes(secret_be)
    return g1_mul(G1, poly_evaluate(coeffs, s))


def proof_shortcut(coeffs, z, y, secret_be: bytes):
    q = poly_divide_by_root(poly_sub(coeffs, poly_from_constant(y)), z)
    return commit_shortcut(q, secret_be)


# --- bench input generators: reference benches/polynomial_commitment.rs:10-23, evaluation_proof.rs:25-27
BENCH_SECRET_BE = bytes(range(32))


def bench_coefficients(degree: int):
    out, p5 = [], 1
    for _ in range(degree + 1):
        out.append((p5 + 10) % R)
        p5 = p5 * 5 % R
    return poly_truncate(out)


def bench_input_point(degree: int) -> int:
    return (pow(5, degree, R) + 20) % R
