"""ctypes loader for oracle/libkzg_oracle.so.  TEST INFRASTRUCTURE ONLY (see kzg_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
numpy arrays carry the blst layouts: Fr = (n,4) uint64, P1 = (n,18) uint64.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

OK, ERR_DEGREE_TOO_HIGH, ERR_CONSTANT_POLY, ERR_REMAINDER = 0, -1, -2, -3


def build():
    """Compile the C restatement (gcc).  Building the checker is not using it."""
    subprocess.run(["make", "-s", "-C", _HERE, "libkzg_oracle.so"], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libkzg_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        vp, sz, u8p = C.c_void_p, C.c_size_t, C.c_char_p
        sig = {
            "oracle_fr_from_le_bytes": (None, [vp, u8p]),
            "oracle_fr_from_be_bytes": (None, [vp, u8p]),
            "oracle_fr_from_i128": (None, [vp, C.c_int64, C.c_uint64]),
            "oracle_fr_to_le_bytes": (None, [vp, vp]),
            "oracle_fr_mul": (None, [vp, vp, vp]),
            "oracle_fr_add": (None, [vp, vp, vp]),
            "oracle_fr_sub": (None, [vp, vp, vp]),
            "oracle_fr_neg": (None, [vp, vp]),
            "oracle_fr_is_zero": (C.c_int, [vp]),
            "oracle_fr_pow": (None, [vp, vp, C.c_uint64]),
            "oracle_p1_generator": (None, [vp]),
            "oracle_p1_add_or_double": (None, [vp, vp, vp]),
            "oracle_p1_double": (None, [vp, vp]),
            "oracle_p1_cneg": (None, [vp, C.c_int]),
            "oracle_p1_mult": (None, [vp, vp, u8p, sz]),
            "oracle_p1_compress": (None, [vp, vp]),
            "oracle_p1_uncompress": (C.c_int, [vp, u8p]),
            "oracle_p1_is_inf": (C.c_int, [vp]),
            "oracle_p1_on_curve": (C.c_int, [vp]),
            "oracle_p1_equal": (C.c_int, [vp, vp]),
            "oracle_p1_to_affine": (None, [vp, vp]),
            "oracle_p1_rescale": (None, [vp, vp, vp]),
            "oracle_srs_g1": (None, [vp, sz, u8p]),
            "oracle_srs_g1_at": (None, [vp, C.c_uint64, u8p]),
            "oracle_poly_truncate": (sz, [vp, sz]),
            "oracle_poly_evaluate": (None, [vp, vp, sz, vp]),
            "oracle_commit_naive": (C.c_int, [vp, vp, sz, vp, sz, sz]),
            "oracle_quotient": (C.c_int, [vp, vp, vp, sz, vp, vp]),
            "oracle_generate_proof": (C.c_int, [vp, vp, sz, vp, vp, vp, sz, sz]),
            "oracle_commit_pippenger": (C.c_int, [vp, vp, sz, vp, sz, sz, C.c_int]),
            "oracle_commit_pippenger_ex": (C.c_int, [vp, vp, sz, vp, sz, sz, C.c_int, C.POINTER(C.c_int)]),
            "oracle_commit_shortcut": (None, [vp, vp, sz, u8p]),
            "oracle_bench_coefficients": (None, [vp, sz]),
            "oracle_bench_input_point": (None, [vp, C.c_uint64]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(_LIB, name)
            fn.restype, fn.argtypes = res, args
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def fr_zeros(n):
    return np.zeros((n, 4), dtype=np.uint64)


def p1_zeros(n):
    return np.zeros((n, 18), dtype=np.uint64)


def fr_from_int(v: int):
    """canonical integer (already reduced or not) -> Montgomery Fr row, via from_le_bytes."""
    out = fr_zeros(1)
    lib().oracle_fr_from_le_bytes(_p(out), (v % (1 << 256)).to_bytes(32, "little"))
    return out[0]


def fr_from_ints(vals):
    out = fr_zeros(len(vals))
    L = lib()
    for i, v in enumerate(vals):
        L.oracle_fr_from_le_bytes(C.c_void_p(out.ctypes.data + 32 * i), (v % (1 << 256)).to_bytes(32, "little"))
    return out


def fr_to_int(row) -> int:
    row = np.ascontiguousarray(row, dtype=np.uint64)
    buf = (C.c_ubyte * 32)()
    lib().oracle_fr_to_le_bytes(C.cast(buf, C.c_void_p), _p(row))
    return int.from_bytes(bytes(buf), "little")


def fr_from_i128(a: int):
    out = fr_zeros(1)
    u = a & ((1 << 128) - 1)
    hi = (u >> 64) & 0xFFFFFFFFFFFFFFFF
    if hi >= 1 << 63:
        hi -= 1 << 64
    lib().oracle_fr_from_i128(_p(out), hi, u & 0xFFFFFFFFFFFFFFFF)
    return out[0]


def p1_compress(row) -> bytes:
    row = np.ascontiguousarray(row, dtype=np.uint64)
    buf = (C.c_ubyte * 48)()
    lib().oracle_p1_compress(C.cast(buf, C.c_void_p), _p(row))
    return bytes(buf)


def p1_uncompress(b: bytes):
    out = p1_zeros(1)
    rc = lib().oracle_p1_uncompress(_p(out), b)
    if rc != 0:
        raise ValueError("oracle_p1_uncompress rc=%d" % rc)
    return out[0]


def p1_generator():
    out = p1_zeros(1)
    lib().oracle_p1_generator(_p(out))
    return out[0]


def p1_add(a, b):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    out = p1_zeros(1)
    lib().oracle_p1_add_or_double(_p(out), _p(a), _p(b))
    return out[0]


def p1_mult(p, k: int):
    p = np.ascontiguousarray(p, dtype=np.uint64)
    out = p1_zeros(1)
    lib().oracle_p1_mult(_p(out), _p(p), (k % (1 << 256)).to_bytes(32, "little"), 256)
    return out[0]


def p1_equal(a, b) -> bool:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    return bool(lib().oracle_p1_equal(_p(a), _p(b)))


def srs_g1(n: int, secret_be: bytes):
    out = p1_zeros(n)
    lib().oracle_srs_g1(_p(out), n, secret_be)
    return out


def srs_g1_at(k: int, secret_be: bytes):
    out = p1_zeros(1)
    lib().oracle_srs_g1_at(_p(out), k, secret_be)
    return out[0]


def bench_coefficients(n: int):
    out = fr_zeros(n)
    lib().oracle_bench_coefficients(_p(out), n)
    return out


def bench_input_point(degree: int):
    out = fr_zeros(1)
    lib().oracle_bench_input_point(_p(out), degree)
    return out[0]


def poly_evaluate(coeffs, x):
    coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64)
    x = np.ascontiguousarray(x, dtype=np.uint64)
    out = fr_zeros(1)
    lib().oracle_poly_evaluate(_p(out), _p(coeffs), len(coeffs), _p(x))
    return out[0]


def commit_naive(coeffs, srs, stride=144):
    coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64)
    out = p1_zeros(1)
    rc = lib().oracle_commit_naive(_p(out), _p(coeffs), len(coeffs), _p(srs), stride, srs.nbytes // stride)
    return rc, out[0]


def commit_pippenger(coeffs, srs, threads=8, stride=144):
    coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64)
    out = p1_zeros(1)
    rc = lib().oracle_commit_pippenger(_p(out), _p(coeffs), len(coeffs), _p(srs), stride, srs.nbytes // stride, threads)
    return rc, out[0]


def commit_pippenger_ex(coeffs, srs, threads=8, stride=144):
    """bucket method over (window, point range) jobs; returns (rc, point, threads that had work)"""
    coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64)
    out = p1_zeros(1)
    used = C.c_int(0)
    rc = lib().oracle_commit_pippenger_ex(_p(out), _p(coeffs), len(coeffs), _p(srs), stride, srs.nbytes // stride, threads,
                                          C.byref(used))
    return rc, out[0], used.value


def commit_shortcut(coeffs, secret_be: bytes):
    coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64)
    out = p1_zeros(1)
    lib().oracle_commit_shortcut(_p(out), _p(coeffs), len(coeffs), secret_be)
    return out[0]


def quotient(coeffs, z, y):
    coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64)
    z = np.ascontiguousarray(z, dtype=np.uint64)
    y = np.ascontiguousarray(y, dtype=np.uint64)
    n = len(coeffs)
    q = fr_zeros(max(n, 1))
    qn = C.c_size_t(0)
    rc = lib().oracle_quotient(_p(q), C.byref(qn), _p(coeffs), n, _p(z), _p(y))
    return rc, q[: qn.value].copy()


def generate_proof(coeffs, z, y, srs, stride=144):
    coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64)
    z = np.ascontiguousarray(z, dtype=np.uint64)
    y = np.ascontiguousarray(y, dtype=np.uint64)
    out = p1_zeros(1)
    rc = lib().oracle_generate_proof(_p(out), _p(coeffs), len(coeffs), _p(z), _p(y), _p(srs), stride, srs.nbytes // stride)
    return rc, out[0]
