"""Host-side checks of csrc/fr30.hip.h (signed radix-2^30 Fr, the arithmetic of the quotient scan and the scalar
recoding) against Python big integers: products, the re-slicing of the blst_fr image into digits and back to the
canonical residue, the host's preparation of a multiplier, and the contract that every column of the multiplier fits
a signed 64-bit accumulator.  CPU only: the header is __host__ __device__ code, compiled here with g++."""
import ctypes
import os
import random
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
B, N = 30, 9
R270 = 1 << 270
I9 = ctypes.c_int32 * 9
U8 = ctypes.c_uint32 * 8
U64x4 = ctypes.c_uint64 * 4


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("r30") / "libr30.so")
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-o", out, os.path.join(ROOT, "tests", "host", "fr30_host.cpp")], check=True)
    L = ctypes.CDLL(out)
    L.r30_mul_max_column.restype = ctypes.c_int64
    return L


def balanced(v):
    d = []
    for _ in range(N - 1):
        r = v & ((1 << B) - 1)
        if r >= 1 << (B - 1):
            r -= 1 << B
        d.append(r)
        v = (v - r) >> B
    d.append(v)
    assert -(1 << 31) <= v < (1 << 31)
    return d


def value(d):
    return sum(int(x) << (B * i) for i, x in enumerate(d))


def limbs(v):
    return [(v >> (32 * i)) & 0xffffffff for i in range(8)]


def test_constants():
    rd = [0x1, -0x4, -0x1a4010, -0x1096ff40, -0x1e27faac, -0x189fdfd9, 0x17d48334, -0x162b3599, 0x73ee]
    assert value(rd) == R and balanced(R) == rd
    ru = [0x1, 0x3ffffffc, 0x3fe5bfef, 0x2f6900bf, 0x21d80553, 0x27602026, 0x17d48333, 0x29d4ca67, 0x73ed]
    assert sum(x << (B * i) for i, x in enumerate(ru)) == R
    assert R % (1 << B) == 1  # so -r^-1 = -1 (mod 2^30)
    one270 = [-0x8d54, 0x23550, -0x21a2ac0, 0x1c22013a, -0x15d0deee, -0x1d3fc534, -0x1dfe7aaf, 0x12b2a695, 0x10dc]
    r2_540 = [0xefe9ec3, 0x1022c0f, 0x12313d61, 0x83bec60, -0xcc084be, 0x4a39dc, -0x17165989, 0x1dd02cf9, -0xc8f]
    assert value(one270) % R == pow(2, 270, R) and value(r2_540) % R == pow(2, 540, R)
    assert all(abs(x) <= 1 << 29 for x in one270[:8] + r2_540[:8])


def test_mul_random_lazy_operands(lib):
    rng = random.Random(9)
    for it in range(4000):
        bits = rng.choice([200, 255, 258, 262])
        a, b = rng.randrange(-(1 << bits), 1 << bits), rng.randrange(-(1 << bits), 1 << bits)
        r = I9()
        lib.r30_mul(I9(*balanced(a)), I9(*balanced(b)), r)
        v = value(list(r))
        assert (v * R270 - a * b) % R == 0
        assert abs(v) <= 0.5001 * R + (abs(a) * abs(b) >> 270) + 1
        assert all(-(1 << 29) <= x < (1 << 29) for x in list(r)[:8])


def test_mul_takes_a_raw_sum_and_columns_fit(lib):
    rng = random.Random(10)
    worst = 0
    for it in range(2000):
        a1, a2 = [balanced(rng.randrange(-(1 << 258), 1 << 258)) for _ in range(2)]
        a = [x + y for x, y in zip(a1, a2)]  # raw sum: digits up to 2^30
        b = balanced(rng.randrange(-(1 << 258), 1 << 258))
        if it % 4 == 0:  # extreme digits
            a = [rng.choice([-(1 << 30), 1 << 30]) for _ in range(8)] + [a[8]]
            b = [rng.choice([-(1 << 29) - 4, (1 << 29) + 4]) for _ in range(8)] + [b[8]]
        r = I9()
        lib.r30_mul(I9(*a), I9(*b), r)
        assert (value(list(r)) * R270 - value(a) * value(b)) % R == 0
        worst = max(worst, lib.r30_mul_max_column(I9(*a), I9(*b)))
    assert worst < (1 << 15)  # |column| < 2^63


def test_from_limbs_and_back(lib):
    rng = random.Random(11)
    for it in range(3000):
        v = rng.choice([0, 1, R - 1, R // 2, (1 << 255) - 1, rng.randrange(R)]) if it < 12 else rng.randrange(R)
        d = I9()
        lib.r30_from_limbs(U8(*limbs(v)), d)
        assert value(list(d)) == v
        assert all(-(1 << 29) - 4 <= x <= (1 << 29) + 4 for x in list(d)[:8])
        out = U8()
        lib.r30_to_limbs(d, out)
        assert list(out) == limbs(v % R)  # (2^255 - 1 is above r: any 256-bit image loads, the store is canonical)


def test_to_limbs_canonicalises_the_range_the_scans_hold(lib):
    rng = random.Random(12)
    cases = [0, -1, 1, R, R - 1, -R + 1, 2 * R - 1, R + 1, -(R // 2), R + R // 2]
    for it in range(4000):
        v = cases[it] if it < len(cases) else rng.randrange(-R + 1, 2 * R)
        d = balanced(v)
        if it % 3 == 0:  # weakly normalised digits too (what fr30_norm leaves)
            j = rng.randrange(8)
            d[j] += 1 << 30
            d[j + 1] -= 1
        out = U8()
        lib.r30_to_limbs(I9(*d), out)
        assert list(out) == limbs(v % R), (it, v)


def test_leaving_the_montgomery_form_and_raw_integers(lib):
    """what the scalar recoding and the SRS generator do: x * 2^256 times the single digit 2^14 is x; a raw 256-bit integer
    times 2^540 mod r is s * 2^270; with that factor on both operands a product keeps it; times the digit 1 drops it"""
    rng = random.Random(14)
    one270 = [-0x8d54, 0x23550, -0x21a2ac0, 0x1c22013a, -0x15d0deee, -0x1d3fc534, -0x1dfe7aaf, 0x12b2a695, 0x10dc]
    r2_540 = [0xefe9ec3, 0x1022c0f, 0x12313d61, 0x83bec60, -0xcc084be, 0x4a39dc, -0x17165989, 0x1dd02cf9, -0xc8f]
    small = lambda c: I9(*([c] + [0] * 8))
    for it in range(500):
        x = rng.choice([0, 1, R - 1]) if it < 3 else rng.randrange(R)
        d, r, out = I9(), I9(), U8()
        lib.r30_from_limbs(U8(*limbs((x << 256) % R)), d)
        lib.r30_mul(d, small(1 << 14), r)
        lib.r30_to_limbs(r, out)
        assert list(out) == limbs(x)
        raw = rng.randrange(1 << 256)  # any 256-bit value
        lib.r30_from_limbs(U8(*limbs(raw)), d)
        lib.r30_mul(d, I9(*one270), r)
        lib.r30_to_limbs(r, out)
        assert list(out) == limbs(raw % R)
        s270 = I9()
        lib.r30_mul(d, I9(*r2_540), s270)
        assert value(list(s270)) % R == (raw << 270) % R
        sq, plain = I9(), I9()
        lib.r30_mul(s270, s270, sq)
        lib.r30_mul(sq, small(1), plain)
        lib.r30_to_limbs(plain, out)
        assert list(out) == limbs(raw * raw % R)


def test_sign_and_magnitude_of_the_recoding(lib):
    """msm_sort.hip's load_scalar: unsigned digits straight into the product, sign and magnitude out; the magnitude times
    the sign is the scalar (mod r) and stays below 2^254"""
    rng = random.Random(15)
    one270 = [-0x8d54, 0x23550, -0x21a2ac0, 0x1c22013a, -0x15d0deee, -0x1d3fc534, -0x1dfe7aaf, 0x12b2a695, 0x10dc]
    small = lambda c: I9(*([c] + [0] * 8))
    edge = [0, 1, R - 1, (R - 1) // 2, (R + 1) // 2, (R - 1) // 2 - 1, (R + 1) // 2 + 1, 5, R - 5]
    for it in range(3000):
        x = edge[it] if it < len(edge) else rng.randrange(R)
        for mont in (True, False):
            image = (x << 256) % R if mont else x + (R if (not mont and it % 7 == 0 and x + R < 1 << 256) else 0)
            d, v, out = I9(), I9(), U8()
            lib.r30_from_limbs_raw(U8(*limbs(image)), d)
            assert all(0 <= t < (1 << 30) for t in list(d)[:8]) and value(list(d)) == image
            lib.r30_mul(d, small(1 << 14) if mont else I9(*one270), v)
            neg = lib.r30_abs_to_limbs(v, out)
            k = sum(w << (32 * i) for i, w in enumerate(out))
            assert k < 1 << 254 and k <= R // 2 + (R >> 31) + 2
            assert ((-k if neg else k) - x) % R == 0
            assert k == abs(value(list(v)))


def test_host_prepared_multiplier_keeps_the_abi_form(lib):
    """x * 2^256 (the ABI's image) times the prepared form of w gives (x w) * 2^256"""
    rng = random.Random(13)
    R256 = 1 << 256
    for it in range(500):
        x, w = rng.randrange(R), rng.randrange(R)
        arg = I9()
        wl = (w * R256) % R
        lib.r30_arg_from_mont256(U64x4(*[(wl >> (64 * i)) & 0xffffffffffffffff for i in range(4)]), arg)
        assert value(list(arg)) % R == (w * R270) % R
        xd = I9()
        lib.r30_from_limbs(U8(*limbs((x * R256) % R)), xd)
        r = I9()
        lib.r30_mul(xd, arg, r)
        out = U8()
        lib.r30_to_limbs(r, out)
        assert list(out) == limbs((x * w * R256) % R)
