"""BLS12-381 pairing in plain Python big ints.  TEST INFRASTRUCTURE ONLY (part of the oracle).

Purpose: the reference pins its commit / open path by SELF-CONSISTENCY, not by known answers -- its
tests (src/lib.rs:16-33, 51-94) commit, open and then require Evaluation::verify_proof
(src/polynomial.rs:276-294) to accept:
        e(proof, [s]G2 - [z]G2) == e(commitment - [y]G1, G2).
This module restates that check (bilinear_map = to_affine x2 + Miller loop + final exponentiation,
src/curves.rs:355-371, which the reference delegates to blst) so that tests can run the reference's own
acceptance criterion on the GPU's outputs and on the C oracle's outputs.  It is independent of every
MSM / division implementation in this repository: a commitment or proof that is wrong by one bit
fails the pairing equation.

Construction (textbook, favouring obviousness over speed -- a verification takes a few seconds):
Fp12 = Fp[w] / (w^12 - 2 w^6 + 2); G2 points are mapped to Fp12 coordinates through the sextic twist;
Miller loop over |x| = 0xd201000000010000 with affine line functions; both sides of the equation
share ONE final exponentiation (p^12 - 1) / r.  Any non-degenerate bilinear pairing decides the
equation, so no Frobenius shortcuts or sign conventions are needed.
"""
from bigint_twin import G1, INF, P, R, fr_from_be_bytes, g1_add, g1_mul, g1_neg

# --- Fp2 = Fp[u]/(u^2+1), elements as (a, b) = a + b u -----------------------------------------
def f2_add(x, y): return ((x[0] + y[0]) % P, (x[1] + y[1]) % P)
def f2_sub(x, y): return ((x[0] - y[0]) % P, (x[1] - y[1]) % P)
def f2_mul(x, y): return ((x[0] * y[0] - x[1] * y[1]) % P, (x[0] * y[1] + x[1] * y[0]) % P)
def f2_inv(x):
    d = pow(x[0] * x[0] + x[1] * x[1], -1, P)
    return (x[0] * d % P, (-x[1]) * d % P)
def f2_scalar(x, k): return (x[0] * k % P, x[1] * k % P)

F2_ZERO, F2_ONE = (0, 0), (1, 0)
B2 = (4, 4)  # curve constant of the twist: y^2 = x^3 + 4(u + 1)

# public G2 generator (affine, Fp2 coordinates as (c0, c1))
G2X = (0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
       0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E)
G2Y = (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
       0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE)
G2 = (G2X, G2Y)


def g2_is_on_curve(pt):
    if pt is INF:
        return True
    x, y = pt
    return f2_sub(f2_mul(y, y), f2_add(f2_mul(f2_mul(x, x), x), B2)) == F2_ZERO


def g2_add(a, b):
    if a is INF:
        return b
    if b is INF:
        return a
    (x1, y1), (x2, y2) = a, b
    if x1 == x2:
        if f2_add(y1, y2) == F2_ZERO:
            return INF
        lam = f2_mul(f2_scalar(f2_mul(x1, x1), 3), f2_inv(f2_scalar(y1, 2)))
    else:
        lam = f2_mul(f2_sub(y2, y1), f2_inv(f2_sub(x2, x1)))
    x3 = f2_sub(f2_sub(f2_mul(lam, lam), x1), x2)
    y3 = f2_sub(f2_mul(lam, f2_sub(x1, x3)), y1)
    return (x3, y3)


def g2_neg(a):
    return INF if a is INF else (a[0], ((-a[1][0]) % P, (-a[1][1]) % P))


def g2_mul(pt, k):
    k %= R
    acc = INF
    for bit in bin(k)[2:] if k else "":
        acc = g2_add(acc, acc)
        if bit == "1":
            acc = g2_add(acc, pt)
    return acc


# --- Fp12 as polynomials in w modulo w^12 - 2 w^6 + 2 ----------------------------------------------
class F12:
    __slots__ = ("c",)

    def __init__(self, c):
        self.c = [x % P for x in c]

    @staticmethod
    def one():
        return F12([1] + [0] * 11)

    def __eq__(self, o):
        return self.c == o.c

    def __add__(self, o):
        return F12([a + b for a, b in zip(self.c, o.c)])

    def __sub__(self, o):
        return F12([a - b for a, b in zip(self.c, o.c)])

    def __neg__(self):
        return F12([-a for a in self.c])

    def __mul__(self, o):
        if isinstance(o, int):
            return F12([a * o for a in self.c])
        t = [0] * 23
        for i, a in enumerate(self.c):
            if a:
                for j, b in enumerate(o.c):
                    t[i + j] += a * b
        # reduce: w^12 = 2 w^6 - 2
        for k in range(22, 11, -1):
            v = t[k]
            if v:
                t[k - 6] += 2 * v
                t[k - 12] -= 2 * v
        return F12(t[:12])

    def inv(self):
        """Extended Euclid in Fp[w] against the modulus polynomial."""
        mod = [2, 0, 0, 0, 0, 0, P - 2, 0, 0, 0, 0, 0, 1]
        lm, hm = [1] + [0] * 12, [0] * 13
        low, high = self.c + [0], mod[:]

        def deg(p):
            d = len(p) - 1
            while d and p[d] == 0:
                d -= 1
            return d

        def poly_rounded_div(a, b):
            dega, degb = deg(a), deg(b)
            temp, o = a[:], [0] * len(a)
            binv = pow(b[degb], -1, P)
            for i in range(dega - degb, -1, -1):
                q = temp[degb + i] * binv % P
                o[i] = (o[i] + q) % P
                for c in range(degb + 1):
                    temp[c + i] = (temp[c + i] - q * b[c]) % P
            return o[: deg(o) + 1]

        while deg(low):
            r = poly_rounded_div(high, low)
            r += [0] * (13 - len(r))
            nm, new = hm[:], high[:]
            for i in range(13):
                for j in range(13 - i):
                    nm[i + j] = (nm[i + j] - lm[i] * r[j]) % P
                    new[i + j] = (new[i + j] - low[i] * r[j]) % P
            lm, low, hm, high = nm, new, lm, low
        li = pow(low[0], -1, P)
        return F12([x * li for x in lm[:12]])

    def __truediv__(self, o):
        return self * o.inv()

    def pow(self, e):
        acc, base = F12.one(), self
        while e:
            if e & 1:
                acc = acc * base
            base = base * base
            e >>= 1
        return acc


W = F12([0, 1] + [0] * 10)
_W2_INV = (W * W).inv()
_W3_INV = (W * W * W).inv()


def twist(pt):
    """G2 point (Fp2 coordinates) -> point on the curve over Fp12 (y^2 = x^3 + 4)."""
    if pt is INF:
        return INF
    (xa, xb), (ya, yb) = pt
    nx = F12([(xa - xb) % P] + [0] * 5 + [xb] + [0] * 5)
    ny = F12([(ya - yb) % P] + [0] * 5 + [yb] + [0] * 5)
    return (nx * _W2_INV, ny * _W3_INV)


def cast_g1(pt):
    return INF if pt is INF else (F12([pt[0]] + [0] * 11), F12([pt[1]] + [0] * 11))


def _f12_double(pt):
    x, y = pt
    m = (x * x * 3) / (y * 2)
    nx = m * m - x * 2
    return (nx, m * (x - nx) - y)


def _f12_add(p1, p2):
    if p1 is INF:
        return p2
    if p2 is INF:
        return p1
    (x1, y1), (x2, y2) = p1, p2
    if x1 == x2:
        return _f12_double(p1) if y1 == y2 else INF
    m = (y2 - y1) / (x2 - x1)
    nx = m * m - x1 - x2
    return (nx, m * (x1 - nx) - y1)


def _linefunc(p1, p2, t):
    (x1, y1), (x2, y2), (xt, yt) = p1, p2, t
    if not x1 == x2:
        m = (y2 - y1) / (x2 - x1)
        return m * (xt - x1) - (yt - y1)
    if y1 == y2:
        m = (x1 * x1 * 3) / (y1 * 2)
        return m * (xt - x1) - (yt - y1)
    return xt - x1


ATE_LOOP = 0xD201000000010000
FINAL_EXP = (P ** 12 - 1) // R


def miller_loop(q_g2, p_g1):
    """f_{|x|, Q}(P) without the final exponentiation; 1 if either argument is infinity."""
    if q_g2 is INF or p_g1 is INF:
        return F12.one()
    Q, Pt = twist(q_g2), cast_g1(p_g1)
    Rp, f = Q, F12.one()
    for i in range(ATE_LOOP.bit_length() - 2, -1, -1):
        f = f * f * _linefunc(Rp, Rp, Pt)
        Rp = _f12_double(Rp)
        if ATE_LOOP >> i & 1:
            f = f * _linefunc(Rp, Q, Pt)
            Rp = _f12_add(Rp, Q)
    return f


def pairing(p_g1, q_g2):
    return miller_loop(q_g2, p_g1).pow(FINAL_EXP)


def pairing_product_is_one(pairs):
    """prod e(P_i, Q_i) == 1 with one shared final exponentiation."""
    f = F12.one()
    for p_g1, q_g2 in pairs:
        f = f * miller_loop(q_g2, p_g1)
    return f.pow(FINAL_EXP) == F12.one()


def verify_proof(commitment, proof, z, y, secret_be):
    """Evaluation::verify_proof (reference src/polynomial.rs:276-294) with setup_artifacts[1].g2 = [s]G2
    recomputed from the known test secret (src/trusted_setup.rs:64-72):
        e(proof, [s]G2 - [z]G2) == e(commitment - [y]G1, G2)
    checked as e(proof, [s - z]G2) * e(-(commitment - [y]G1), G2) == 1."""
    s = fr_from_be_bytes(secret_be)
    lhs_g2 = g2_add(g2_mul(G2, s), g2_neg(g2_mul(G2, z % R)))
    rhs_g1 = g1_add(commitment, g1_neg(g1_mul(G1, y % R)))
    return pairing_product_is_one([(proof, lhs_g2), (g1_neg(rhs_g1), G2)])
