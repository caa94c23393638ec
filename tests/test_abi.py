"""CPU: the C-ABI shared library loads and exports every symbol include/kzg_mi355x.h declares; the
pure-host helpers work; and without a GPU the product refuses to run instead of falling back."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import kzg_poly_commit_exploration_amd as K

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    text = open(os.path.join(ROOT, "include", "kzg_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kzg_[a-z0-9_]+)\s*\(", text)))


def test_header_and_wrapper_agree():
    assert _header_functions() == sorted(K.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = K.load_library()
    for name in _header_functions():
        assert hasattr(lib, name), name


def test_abi_version_and_struct_mirrors():
    lib = K.load_library()
    assert lib.kzg_abi_version() == K.KZG_ABI_VERSION == 4
    text = open(os.path.join(ROOT, "include", "kzg_mi355x.h")).read()
    assert "#define KZG_ABI_VERSION 4" in text
    # kzg_kernel_times: six floats, a u64, two floats (the Python mirror must have the C layout)
    assert C.sizeof(K.KernelTimes) == 40


def test_strerror_carries_reference_messages():
    lib = K.load_library()
    # reference src/polynomial.rs:202-204, :165, :189-191
    assert lib.kzg_strerror(-1).decode().startswith("Setup does not allow for commitment generation")
    assert lib.kzg_strerror(-2).decode() == "Unable to divide a constant polynomial"
    assert lib.kzg_strerror(-3).decode().startswith("[divide_by_root] Fail to divide the polynomial by a root")


def test_host_g1_helpers_against_oracle(oracle, golden):
    g = oracle.p1_generator()
    assert K.G1Point(g).compress().hex() == golden["constants"]["compress_G"]
    inf = np.zeros(18, dtype=np.uint64)
    assert K.G1Point(inf).compress().hex() == golden["constants"]["compress_inf"]
    pts = [oracle.p1_mult(g, k) for k in (1, 2, 3, 12345, K.R_MODULUS - 6)]
    total = K.G1Point.sum([K.G1Point(p) for p in pts])
    assert total.compress() == oracle.p1_compress(oracle.p1_mult(g, 12345))  # 1+2+3-6 = 0
    # complete addition: doubling and cancellation
    assert K.G1Point.sum([K.G1Point(g), K.G1Point(g)]).compress().hex() == golden["constants"]["compress_2G"]
    assert K.G1Point.sum([K.G1Point(pts[0]), K.G1Point(oracle.p1_mult(g, K.R_MODULUS - 1))]).is_infinity()
    # Jacobian inputs with Z != 1
    lam = np.array([9, 8, 7, 6, 5, 4], dtype=np.uint64)
    scaled = oracle.p1_zeros(1)
    oracle.lib().oracle_p1_rescale(scaled.ctypes.data, pts[3].ctypes.data, lam.ctypes.data)
    assert K.G1Point(scaled[0]).compress() == oracle.p1_compress(pts[3])


def test_g1_uncompress_round_trip(oracle, golden):
    g = oracle.p1_generator()
    for k in (1, 2, 3, 0xDEADBEEF, K.R_MODULUS - 1):
        pt = oracle.p1_mult(g, k)
        c = oracle.p1_compress(pt)
        back = K.G1Point.uncompress(c)
        assert back.compress() == c and oracle.p1_equal(back.p1, pt)
    assert K.G1Point.uncompress(bytes.fromhex(golden["constants"]["compress_inf"])).is_infinity()
    with pytest.raises(K.KzgError):
        K.G1Point.uncompress(bytes([0x80]) + bytes(46) + bytes([0x01]))  # x = 1 is not on the curve


def test_no_cpu_fallback():
    """Without a HIP device the engine cannot be created: the product has no CPU path."""
    try:
        import torch

        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip("GPU present")
    with pytest.raises(K.KzgError) as ei:
        K.Engine(0)
    assert ei.value.status == K.KZG_ERR_NO_DEVICE
