"""kzg_poly_commit_exploration_amd -- host-side mirror of the reference's commit / open API over
the MI355X engine (libkzg_mi355x.so, C-ABI in include/kzg_mi355x.h).

The names follow the reference crate (VGLoic/kzg-poly-commit-exploration): Scalar
(src/scalar.rs), G1Point (src/curves.rs), SetupArtifactsGenerator (src/trusted_setup.rs),
Polynomial / Evaluation (src/polynomial.rs) -- same argument meaning, same error messages -- so
the parity tests read like the reference's own tests (src/lib.rs:16-33).  Everything that is
arithmetic on the hot path goes through the C-ABI into HIP kernels; this module only converts
representations (Python ints <-> blst limb layouts), which is what the reference's Rust host code
does around its blst calls.  There is no CPU fallback: importing works without a GPU, creating an
Engine does not.
"""
import ctypes as C
import os

import numpy as np

__all__ = [
    "Engine", "Scalar", "G1Point", "Polynomial", "Evaluation", "SetupArtifactsGenerator", "KzgError",
    "R_MODULUS", "lib_path", "load_library", "ABI_SYMBOLS", "srs_g2_at", "verify_proof", "verify_proof_batch",
]

_HERE = os.path.dirname(os.path.abspath(__file__))
R_MODULUS = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001  # src/scalar.rs:10
_FR_R = 1 << 256
_FR_RINV = pow(_FR_R, -1, R_MODULUS)

KZG_OK = 0
KZG_ERR_DEGREE_TOO_HIGH = -1
KZG_ERR_CONSTANT_POLY = -2
KZG_ERR_REMAINDER = -3
KZG_ERR_INVALID_ARG = -4
KZG_ERR_NO_DEVICE = -5
KZG_ERR_HIP = -6
KZG_ERR_NO_SRS = -7
KZG_ERR_BUSY = -8
KZG_MULTI_REPLICATE_SRS = 1
KZG_ABI_VERSION = 4

# every symbol include/kzg_mi355x.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "kzg_ctx_create", "kzg_ctx_destroy", "kzg_strerror", "kzg_last_error",
    "kzg_srs_load_g1", "kzg_srs_generate_g1", "kzg_srs_read_g1", "kzg_srs_len",
    "kzg_commit", "kzg_commit_le_bytes", "kzg_open", "kzg_quotient", "kzg_evaluate",
    "kzg_srs_load_affine", "kzg_srs_load_compressed", "kzg_srs_save", "kzg_srs_load_file",
    "kzg_ctx_create_multi", "kzg_ctx_create_multi_ex", "kzg_abi_version", "kzg_commit_batch", "kzg_open_batch", "kzg_num_devices", "kzg_rccl_exchanges", "kzg_num_slots", "kzg_commit_submit", "kzg_open_submit", "kzg_wait",
    "kzg_set_max_batch", "kzg_max_batch", "kzg_commit_batch_submit", "kzg_wait_batch",
    "kzg_open_batch_submit", "kzg_wait_open_batch", "kzg_g1_uncompress",
    "kzg_dev_alloc", "kzg_dev_free", "kzg_dev_upload", "kzg_dev_download",
    "kzg_g1_sum", "kzg_g1_compress", "kzg_srs_g2_at", "kzg_verify_proof", "kzg_verify_proof_batch", "kzg_set_timing", "kzg_get_times", "kzg_msm_config",
]


class KzgError(Exception):
    """Carries the reference's anyhow message for the three errors of the path."""

    def __init__(self, status, message):
        super().__init__(message)
        self.status = status


class KernelTimes(C.Structure):
    _fields_ = [(n, C.c_float) for n in
                ("digits_ms", "scatter_ms", "accumulate_ms", "reduce_ms", "quotient_ms", "total_ms")
                ] + [("references", C.c_uint64), ("accumulate_events_ms", C.c_float), ("reserved_", C.c_float)]


def lib_path():
    # KZG_MI355X_LIB: another build of the same library (A/B measurements of kernel variants)
    return os.environ.get("KZG_MI355X_LIB") or os.path.join(_HERE, "libkzg_mi355x.so")


_LIB = None


def load_library():
    """Loads the C-ABI library.  Fails loudly if it has not been built (no fallback exists)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(
            "libkzg_mi355x.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  The product has no CPU path.")
    lib = C.CDLL(path)
    vp, sz, u8p, i = C.c_void_p, C.c_size_t, C.c_char_p, C.c_int
    sig = {
        "kzg_ctx_create": (i, [i, C.POINTER(vp)]),
        "kzg_ctx_create_multi": (i, [vp, i, C.POINTER(vp)]),
        "kzg_ctx_create_multi_ex": (i, [vp, i, C.c_uint, C.POINTER(vp)]),
        "kzg_abi_version": (i, []),
        "kzg_commit_batch": (i, [vp, vp, sz, sz, sz, vp]),
        "kzg_open_batch": (i, [vp, vp, sz, sz, sz, vp, vp, vp, vp]),
        "kzg_num_devices": (i, [vp]),
        "kzg_rccl_exchanges": (C.c_uint64, [vp]),
        "kzg_ctx_destroy": (None, [vp]),
        "kzg_strerror": (C.c_char_p, [i]),
        "kzg_last_error": (C.c_char_p, [vp]),
        "kzg_srs_load_g1": (i, [vp, vp, sz, sz]),
        "kzg_srs_generate_g1": (i, [vp, u8p, C.c_uint64, sz]),
        "kzg_srs_read_g1": (i, [vp, sz, sz, vp]),
        "kzg_srs_load_affine": (i, [vp, vp, sz]),
        "kzg_srs_load_compressed": (i, [vp, vp, sz, C.POINTER(sz)]),
        "kzg_srs_save": (i, [vp, C.c_char_p]),
        "kzg_srs_load_file": (i, [vp, C.c_char_p]),
        "kzg_srs_len": (sz, [vp]),
        "kzg_commit": (i, [vp, vp, sz, vp]),
        "kzg_commit_le_bytes": (i, [vp, vp, sz, vp]),
        "kzg_open": (i, [vp, vp, sz, vp, vp, vp]),
        "kzg_quotient": (i, [vp, vp, sz, vp, vp, vp, C.POINTER(sz)]),
        "kzg_evaluate": (i, [vp, vp, sz, vp, vp]),
        "kzg_num_slots": (i, [vp]),
        "kzg_commit_submit": (i, [vp, i, vp, sz]),
        "kzg_open_submit": (i, [vp, i, vp, sz, vp, vp]),
        "kzg_wait": (i, [vp, i, vp]),
        "kzg_set_max_batch": (i, [vp, sz]),
        "kzg_max_batch": (sz, [vp]),
        "kzg_commit_batch_submit": (i, [vp, i, vp, sz, sz, sz]),
        "kzg_wait_batch": (i, [vp, i, vp, sz]),
        "kzg_open_batch_submit": (i, [vp, i, vp, sz, sz, sz, vp, vp]),
        "kzg_wait_open_batch": (i, [vp, i, vp, vp, sz]),
        "kzg_g1_uncompress": (i, [u8p, vp]),
        "kzg_dev_alloc": (i, [vp, sz, C.POINTER(vp)]),
        "kzg_dev_free": (i, [vp, vp]),
        "kzg_dev_upload": (i, [vp, vp, vp, sz]),
        "kzg_dev_download": (i, [vp, vp, vp, sz]),
        "kzg_g1_sum": (i, [vp, sz, vp]),
        "kzg_g1_compress": (i, [vp, vp]),
        "kzg_set_timing": (i, [vp, i]),
        "kzg_get_times": (i, [vp, i, C.POINTER(KernelTimes)]),
        "kzg_srs_g2_at": (i, [u8p, C.c_uint64, vp]),
        "kzg_verify_proof": (i, [vp, vp, vp, vp, vp, C.POINTER(i)]),
        "kzg_verify_proof_batch": (i, [vp, vp, vp, vp, vp, sz, vp]),
        "kzg_msm_config": (i, [vp, C.POINTER(i), C.POINTER(i), C.POINTER(sz), C.POINTER(i)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _LIB = lib
    return lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


# ---------------------------------------------------------------------------------------------
# Scalar: reference src/scalar.rs.  Value semantics on a canonical integer; `limbs()` is the
# blst_fr memory image (4 x u64, Montgomery) that crosses the C-ABI.
# ---------------------------------------------------------------------------------------------
class Scalar:
    __slots__ = ("v",)

    def __init__(self, v=0):
        self.v = int(v) % R_MODULUS

    @staticmethod
    def from_i128(a):  # src/scalar.rs:27-48: a > 0 -> a ; a <= 0 -> r - |a|
        a = int(a)
        if not -(1 << 127) <= a < (1 << 127):
            raise OverflowError("not an i128")
        return Scalar(a if a > 0 else R_MODULUS - (-a))

    @staticmethod
    def from_le_bytes(b):  # src/scalar.rs:54-61
        assert len(b) == 32
        return Scalar(int.from_bytes(bytes(b), "little"))

    @staticmethod
    def from_be_bytes(b):  # src/scalar.rs:66-73
        assert len(b) == 32
        return Scalar(int.from_bytes(bytes(b), "big"))

    @staticmethod
    def from_limbs(l):
        raw = sum(int(x) << (64 * i) for i, x in enumerate(l))
        return Scalar(raw * _FR_RINV)

    def to_le_bytes(self):  # src/scalar.rs:83-93
        return self.v.to_bytes(32, "little")

    def to_be_bytes(self):  # src/scalar.rs:96-106
        return self.v.to_bytes(32, "big")

    def limbs(self):
        m = self.v * _FR_R % R_MODULUS
        return np.array([(m >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)

    def mul(self, o):
        return Scalar(self.v * o.v)

    def add(self, o):
        return Scalar(self.v + o.v)

    def sub(self, o):
        return Scalar(self.v - o.v)

    def neg(self):
        return Scalar(-self.v)

    def pow(self, n):  # src/scalar.rs:122-187 (same value)
        return Scalar(pow(self.v, int(n), R_MODULUS))

    def is_zero(self):
        return self.v == 0

    def __eq__(self, o):
        return isinstance(o, Scalar) and self.v == o.v

    def __hash__(self):
        return hash(self.v)

    def __repr__(self):
        return "Scalar(%d)" % self.v

    def __str__(self):  # base-10 Display, src/scalar.rs:277-341
        return str(self.v)


def scalars_to_limbs(values):
    """ints (canonical, any sign) -> (n, 4) uint64 Montgomery blst_fr array."""
    out = np.empty((len(values), 4), dtype=np.uint64)
    mask = 0xFFFFFFFFFFFFFFFF
    for i, v in enumerate(values):
        m = (int(v) % R_MODULUS) * _FR_R % R_MODULUS
        out[i, 0] = m & mask
        out[i, 1] = (m >> 64) & mask
        out[i, 2] = (m >> 128) & mask
        out[i, 3] = m >> 192
    return out


def limbs_to_scalars(arr):
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 4)
    out = []
    for row in arr:
        raw = int(row[0]) | (int(row[1]) << 64) | (int(row[2]) << 128) | (int(row[3]) << 192)
        out.append(raw * _FR_RINV % R_MODULUS)
    return out


# ---------------------------------------------------------------------------------------------
# G1Point: reference src/curves.rs:10-17 (a wrapped blst_p1).  Serialisation = 48 compressed bytes
# (src/curves.rs:99-110), which is also how two points are compared for parity.
# ---------------------------------------------------------------------------------------------
class G1Point:
    __slots__ = ("p1",)

    def __init__(self, p1):
        self.p1 = np.ascontiguousarray(p1, dtype=np.uint64).reshape(18)

    def compress(self):
        out = (C.c_ubyte * 48)()
        rc = load_library().kzg_g1_compress(_ptr(self.p1), C.cast(out, C.c_void_p))
        _check(rc)
        return bytes(out)

    @staticmethod
    def uncompress(data):  # Deserialize for G1Point, src/curves.rs:112-183
        out = np.zeros(18, dtype=np.uint64)
        _check(load_library().kzg_g1_uncompress(bytes(data), _ptr(out)))
        return G1Point(out)

    def is_infinity(self):
        return not self.p1[12:18].any()

    def add(self, other):  # src/curves.rs:79-85
        return G1Point.sum([self, other])

    @staticmethod
    def sum(points):
        arr = np.ascontiguousarray(np.stack([p.p1 for p in points]), dtype=np.uint64)
        out = np.zeros(18, dtype=np.uint64)
        _check(load_library().kzg_g1_sum(_ptr(arr), len(points), _ptr(out)))
        return G1Point(out)

    def __eq__(self, o):
        return isinstance(o, G1Point) and self.compress() == o.compress()

    def __repr__(self):
        return "G1Point(%s)" % self.compress().hex()


def _check(rc, ctx=None):
    if rc == KZG_OK:
        return
    lib = load_library()
    msg = lib.kzg_strerror(rc).decode()
    if rc == KZG_ERR_HIP and ctx is not None:
        msg += ": " + lib.kzg_last_error(ctx).decode()
    raise KzgError(rc, msg)


# ---------------------------------------------------------------------------------------------
# Engine: one context = one GPU with a resident SRS (no reference analogue; see kzg_mi355x.h).
# ---------------------------------------------------------------------------------------------
class Engine:
    def __init__(self, device=0, devices=None, replicate=False):
        """device: one HIP device.  devices=[d0, d1, ...]: one context over several devices (kzg_ctx_create_multi):
        the SRS is split by point range and commit / open shard transparently; a device may repeat (virtual slices).
        replicate=True (kzg_ctx_create_multi_ex, KZG_MULTI_REPLICATE_SRS): every device keeps the whole SRS and
        batches are split by polynomial, with nothing to exchange."""
        self._lib = load_library()
        h = C.c_void_p()
        if devices is not None:
            arr = (C.c_int * len(devices))(*[int(d) for d in devices])
            if replicate:
                _check(self._lib.kzg_ctx_create_multi_ex(arr, len(devices), KZG_MULTI_REPLICATE_SRS, C.byref(h)))
            else:
                _check(self._lib.kzg_ctx_create_multi(arr, len(devices), C.byref(h)))
            device = int(devices[0])
        else:
            _check(self._lib.kzg_ctx_create(int(device), C.byref(h)))
        self._h = h
        self.device = device

    def num_devices(self):
        return int(self._lib.kzg_num_devices(self._h))

    def rccl_exchanges(self):
        return int(self._lib.kzg_rccl_exchanges(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.kzg_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- SRS --
    def srs_load(self, p1_array, stride=None):
        """p1_array: (n, k>=18) uint64 rows starting with a blst_p1 (the g1 of a SetupArtifact)."""
        a = np.ascontiguousarray(p1_array, dtype=np.uint64)
        n = a.shape[0]
        stride = a.strides[0] if stride is None else stride
        _check(self._lib.kzg_srs_load_g1(self._h, _ptr(a), stride, n), self._h)

    def srs_load_affine(self, xy_array):
        """(n, 12) uint64: x, y as blst_fp (Montgomery); (0, 0) = infinity"""
        a = np.ascontiguousarray(xy_array, dtype=np.uint64).reshape(-1, 12)
        _check(self._lib.kzg_srs_load_affine(self._h, _ptr(a), a.shape[0]), self._h)

    def srs_load_compressed(self, data):
        """n x 48 bytes (ZCash encoding); raises KzgError with .bad_index on a malformed point"""
        buf = np.frombuffer(bytes(data), dtype=np.uint8)
        assert buf.size % 48 == 0
        bad = C.c_size_t(0)
        rc = self._lib.kzg_srs_load_compressed(self._h, _ptr(buf), buf.size // 48, C.byref(bad))
        if rc != KZG_OK:
            e = KzgError(rc, self._lib.kzg_strerror(rc).decode() + ": " + self._lib.kzg_last_error(self._h).decode())
            e.bad_index = None if bad.value == C.c_size_t(-1).value else int(bad.value)
            raise e

    def srs_save(self, path):
        _check(self._lib.kzg_srs_save(self._h, os.fsencode(path)), self._h)

    def srs_load_file(self, path):
        _check(self._lib.kzg_srs_load_file(self._h, os.fsencode(path)), self._h)

    def srs_generate(self, secret_be, n, first=0):
        _check(self._lib.kzg_srs_generate_g1(self._h, bytes(secret_be), first, n), self._h)

    def srs_read(self, index, count):
        out = np.zeros((count, 18), dtype=np.uint64)
        _check(self._lib.kzg_srs_read_g1(self._h, index, count, _ptr(out)), self._h)
        return out

    def srs_len(self):
        return int(self._lib.kzg_srs_len(self._h))

    def msm_config(self):
        c, w, nb, rec = C.c_int(), C.c_int(), C.c_size_t(), C.c_int()
        _check(self._lib.kzg_msm_config(self._h, C.byref(c), C.byref(w), C.byref(nb), C.byref(rec)))
        return {"recoding": "naf" if rec.value == 1 else "windows", "digit_bits": c.value,
                "table_levels": w.value, "buckets": nb.value}

    # -- hot path, host buffers --
    def commit_limbs(self, coeffs):
        a = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros(18, dtype=np.uint64)
        _check(self._lib.kzg_commit(self._h, _ptr(a), a.shape[0], _ptr(out)), self._h)
        return G1Point(out)

    def commit_le_bytes(self, scalars_le):
        a = np.ascontiguousarray(np.frombuffer(bytes(scalars_le), dtype=np.uint8))
        out = np.zeros(18, dtype=np.uint64)
        _check(self._lib.kzg_commit_le_bytes(self._h, _ptr(a), a.size // 32, _ptr(out)), self._h)
        return G1Point(out)

    def open_limbs(self, coeffs, z, y):
        a = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
        zl, yl = z.limbs(), y.limbs()
        out = np.zeros(18, dtype=np.uint64)
        _check(self._lib.kzg_open(self._h, _ptr(a), a.shape[0], _ptr(zl), _ptr(yl), _ptr(out)), self._h)
        return G1Point(out)

    def quotient_limbs(self, coeffs, z, y):
        a = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
        zl, yl = z.limbs(), y.limbs()
        q = np.zeros((max(a.shape[0], 1), 4), dtype=np.uint64)
        qn = C.c_size_t(0)
        _check(self._lib.kzg_quotient(self._h, _ptr(a), a.shape[0], _ptr(zl), _ptr(yl), _ptr(q), C.byref(qn)), self._h)
        return q[: qn.value].copy()

    def evaluate_limbs(self, coeffs, z):
        a = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
        zl = z.limbs()
        out = np.zeros(4, dtype=np.uint64)
        _check(self._lib.kzg_evaluate(self._h, _ptr(a), a.shape[0], _ptr(zl), _ptr(out)), self._h)
        return Scalar.from_limbs(out)

    # -- device-resident, pipelined --
    def num_slots(self):
        return int(self._lib.kzg_num_slots(self._h))

    def dev_alloc(self, nbytes):
        p = C.c_void_p()
        _check(self._lib.kzg_dev_alloc(self._h, nbytes, C.byref(p)), self._h)
        return p.value

    def dev_free(self, dptr):
        _check(self._lib.kzg_dev_free(self._h, C.c_void_p(dptr)), self._h)

    def dev_upload(self, dptr, array):
        a = np.ascontiguousarray(array)
        _check(self._lib.kzg_dev_upload(self._h, C.c_void_p(dptr), _ptr(a), a.nbytes), self._h)

    def commit_submit(self, slot, dptr, n):
        _check(self._lib.kzg_commit_submit(self._h, slot, C.c_void_p(dptr), n), self._h)

    def open_submit(self, slot, dptr, n, z, y):
        zl, yl = z.limbs(), y.limbs()
        _check(self._lib.kzg_open_submit(self._h, slot, C.c_void_p(dptr), n, _ptr(zl), _ptr(yl)), self._h)

    def wait(self, slot):
        out = np.zeros(18, dtype=np.uint64)
        _check(self._lib.kzg_wait(self._h, slot, _ptr(out)), self._h)
        return G1Point(out)

    def set_max_batch(self, b):
        _check(self._lib.kzg_set_max_batch(self._h, b), self._h)
        return int(self._lib.kzg_max_batch(self._h))

    def max_batch(self):
        return int(self._lib.kzg_max_batch(self._h))

    def commit_batch_submit(self, slot, dptr, n, batch, stride=None):
        _check(self._lib.kzg_commit_batch_submit(self._h, slot, C.c_void_p(dptr), n, batch, n if stride is None else stride),
               self._h)

    def wait_batch(self, slot, batch):
        out = np.zeros((batch, 18), dtype=np.uint64)
        _check(self._lib.kzg_wait_batch(self._h, slot, _ptr(out), batch), self._h)
        return [G1Point(out[i]) for i in range(batch)]

    def open_batch_limbs(self, polys, zs, ys):
        """Batched generate_proof: returns a list of G1Point or KzgError per polynomial."""
        polys = [np.ascontiguousarray(p, dtype=np.uint64).reshape(-1, 4) for p in polys]
        n, b = polys[0].shape[0], len(polys)
        assert all(p.shape[0] == n for p in polys) and len(zs) == b and len(ys) == b
        flat = np.concatenate(polys)
        zl = np.ascontiguousarray(np.stack([z.limbs() for z in zs]))
        yl = np.ascontiguousarray(np.stack([y.limbs() for y in ys]))
        dptr = self.dev_alloc(flat.nbytes)
        try:
            self.dev_upload(dptr, flat)
            _check(self._lib.kzg_open_batch_submit(self._h, 0, C.c_void_p(dptr), n, b, n, _ptr(zl), _ptr(yl)), self._h)
            out = np.zeros((b, 18), dtype=np.uint64)
            st = np.zeros(b, dtype=np.int32)
            _check(self._lib.kzg_wait_open_batch(self._h, 0, _ptr(out), _ptr(st), b), self._h)
        finally:
            self.dev_free(dptr)
        res = []
        for i in range(b):
            if st[i] == KZG_OK:
                res.append(G1Point(out[i]))
            else:
                res.append(KzgError(int(st[i]), self._lib.kzg_strerror(int(st[i])).decode()))
        return res

    # -- host-pointer batches (every kind of context) --
    @staticmethod
    def _stack(polys):
        """list of (n, 4) arrays, or one (batch, n, 4) array -> contiguous (batch, n, 4) uint64"""
        a = np.asarray(polys, dtype=np.uint64) if not isinstance(polys, np.ndarray) else polys
        if a.ndim != 3:
            a = np.stack([np.ascontiguousarray(p, dtype=np.uint64).reshape(-1, 4) for p in polys])
        return np.ascontiguousarray(a, dtype=np.uint64)

    def commit_batch_host(self, polys):
        """kzg_commit_batch: one call for a list of equally long polynomials in host memory."""
        flat = self._stack(polys)
        b, n = flat.shape[0], flat.shape[1]
        out = np.zeros((b, 18), dtype=np.uint64)
        _check(self._lib.kzg_commit_batch(self._h, _ptr(flat), n, b, n, _ptr(out)), self._h)
        return [G1Point(out[i]) for i in range(b)]

    def open_batch_host(self, polys, zs, ys):
        """kzg_open_batch: returns a list of G1Point or KzgError per polynomial."""
        flat = self._stack(polys)
        b, n = flat.shape[0], flat.shape[1]
        assert len(zs) == b and len(ys) == b
        zl = np.ascontiguousarray(np.stack([z.limbs() for z in zs]))
        yl = np.ascontiguousarray(np.stack([y.limbs() for y in ys]))
        out = np.zeros((b, 18), dtype=np.uint64)
        st = np.zeros(b, dtype=np.int32)
        _check(self._lib.kzg_open_batch(self._h, _ptr(flat), n, b, n, _ptr(zl), _ptr(yl), _ptr(out), _ptr(st)), self._h)
        return [G1Point(out[i]) if st[i] == KZG_OK else KzgError(int(st[i]), self._lib.kzg_strerror(int(st[i])).decode())
                for i in range(b)]

    def commit_batch_limbs(self, polys):
        """Commits several coefficient arrays (each (n, 4) uint64, same n) in one batched pass."""
        polys = [np.ascontiguousarray(p, dtype=np.uint64).reshape(-1, 4) for p in polys]
        n = polys[0].shape[0]
        assert all(p.shape[0] == n for p in polys)
        flat = np.concatenate(polys)
        dptr = self.dev_alloc(flat.nbytes)
        try:
            self.dev_upload(dptr, flat)
            self.commit_batch_submit(0, dptr, n, len(polys))
            return self.wait_batch(0, len(polys))
        finally:
            self.dev_free(dptr)

    # -- measurement --
    def set_timing(self, enabled):
        _check(self._lib.kzg_set_timing(self._h, 1 if enabled else 0))

    def times(self, slot):
        t = KernelTimes()
        _check(self._lib.kzg_get_times(self._h, slot, C.byref(t)))
        return {n: getattr(t, n) for n, _ in KernelTimes._fields_}


# ---------------------------------------------------------------------------------------------
# SetupArtifactsGenerator: reference src/trusted_setup.rs:9-29 / 37-78.  `take(n)` materialises the
# first n artifacts -- on the device, G1 side -- and returns the engine that now holds them; the
# reference's `Vec<SetupArtifact>` argument of commit / generate_proof becomes that engine.
# ---------------------------------------------------------------------------------------------
class SetupArtifactsGenerator:
    def __init__(self, secret_be, device=0):
        assert len(secret_be) == 32
        self.secret = bytes(secret_be)
        self.device = device

    def take(self, n, engine=None):
        eng = engine or Engine(self.device)
        eng.srs_generate(self.secret, n)
        return eng


def srs_g2_at(secret_be, index=1):
    """kzg_srs_g2_at: the G2 half of SetupArtifact `index`, [s^index]G2, as a blst_p2 (36 x uint64) -- what
    Evaluation::verify_proof reads from setup_artifacts[1] (src/polynomial.rs:284)."""
    lib = load_library()
    out = np.zeros(36, dtype=np.uint64)
    _check(lib.kzg_srs_g2_at(bytes(secret_be), int(index), _ptr(out)))
    return out


def verify_proof(commitment, proof, z, y, s_g2):
    """kzg_verify_proof: e(proof, [s]G2 - [z]G2) == e(commitment - [y]G1, G2) on the host."""
    lib = load_library()
    g2 = np.ascontiguousarray(s_g2, dtype=np.uint64).reshape(36)
    ok = C.c_int(0)
    zl, yl = z.limbs(), y.limbs()
    _check(lib.kzg_verify_proof(_ptr(commitment.p1), _ptr(proof.p1), _ptr(zl), _ptr(yl), _ptr(g2), C.byref(ok)))
    return bool(ok.value)


def verify_proof_batch(commitments, proofs, zs, ys, s_g2):
    """kzg_verify_proof_batch: one verdict per (commitment, proof, z, y), checks spread over the host cores."""
    lib = load_library()
    n = len(commitments)
    assert len(proofs) == n and len(zs) == n and len(ys) == n
    g2 = np.ascontiguousarray(s_g2, dtype=np.uint64).reshape(36)
    cs = np.ascontiguousarray(np.stack([c.p1 for c in commitments]) if n else np.zeros((0, 18)), dtype=np.uint64)
    ps = np.ascontiguousarray(np.stack([p.p1 for p in proofs]) if n else np.zeros((0, 18)), dtype=np.uint64)
    zl = np.ascontiguousarray(np.stack([z.limbs() for z in zs]) if n else np.zeros((0, 4)), dtype=np.uint64)
    yl = np.ascontiguousarray(np.stack([y.limbs() for y in ys]) if n else np.zeros((0, 4)), dtype=np.uint64)
    ok = np.zeros(max(n, 1), dtype=np.int32)
    _check(lib.kzg_verify_proof_batch(_ptr(cs), _ptr(ps), _ptr(zl), _ptr(yl), _ptr(g2), n, _ptr(ok)))
    return [bool(v) for v in ok[:n]]


# ---------------------------------------------------------------------------------------------
# Polynomial / Evaluation: reference src/polynomial.rs
# ---------------------------------------------------------------------------------------------
class Polynomial:
    def __init__(self, limbs):
        self.limbs = np.ascontiguousarray(limbs, dtype=np.uint64).reshape(-1, 4)

    @staticmethod
    def try_from(values):
        """TryFrom<Vec<i128>> / TryFrom<Vec<Scalar>> (src/polynomial.rs:14-76): trailing zeros are
        dropped, index 0 is kept."""
        if len(values) > 0xFFFFFFFF:
            raise KzgError(KZG_ERR_INVALID_ARG,
                           "Too many coefficients for polynomial, only 2**32 - 1 coefficients is supported. Got %d"
                           % len(values))
        ints = [v.v if isinstance(v, Scalar) else Scalar.from_i128(v).v for v in values]
        last = 0
        for i, v in enumerate(ints):
            if v != 0:
                last = i
        ints = ints[: last + 1] if ints else []
        return Polynomial(scalars_to_limbs(ints))

    @staticmethod
    def from_limbs(limbs):
        """Takes blst_fr rows as they are (applies the same truncation)."""
        a = np.ascontiguousarray(limbs, dtype=np.uint64).reshape(-1, 4)
        nz = np.flatnonzero(a.any(axis=1))
        last = int(nz[-1]) if nz.size else 0
        return Polynomial(a[: last + 1] if a.shape[0] else a)

    def degree(self):  # src/polynomial.rs:93-98
        return 0 if self.limbs.shape[0] == 0 else self.limbs.shape[0] - 1

    def coefficients(self):
        return [Scalar(v) for v in limbs_to_scalars(self.limbs)]

    def commit(self, setup):  # src/polynomial.rs:200-215
        return setup.commit_limbs(self.limbs)

    def evaluate(self, x, setup):  # src/polynomial.rs:112-123
        return Evaluation(x, setup.evaluate_limbs(self.limbs, x))

    def divide_by_root_minus(self, root, y, setup):
        """(self - y).divide_by_root(root): src/polynomial.rs:128-195."""
        return Polynomial(setup.quotient_limbs(self.limbs, root, y))


class Evaluation:
    def __init__(self, point, result):
        self.point, self.result = point, result

    def generate_proof(self, polynomial, setup):  # src/polynomial.rs:260-269
        return setup.open_limbs(polynomial.limbs, self.point, self.result)

    def verify_proof(self, proof, commitment, s_g2):  # src/polynomial.rs:276-294
        """s_g2: setup_artifacts[1].g2 as 36 x u64 (blst_p2).  Host-side pairing check of the library."""
        return verify_proof(commitment, proof, self.point, self.result, s_g2)
