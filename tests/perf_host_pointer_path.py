"""Synchronous host-pointer entry points (what the Rust shim calls): kzg_commit / kzg_open at degree 2^20, PCIe copy
included, one call at a time.  GPU; prints one JSON line.  (bench.py's `value` is measured with inputs resident.)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import kzg_poly_commit_exploration_amd as K  # noqa: E402
import oracle_ctypes as O  # noqa: E402  (bench inputs + checker only)


def main():
    n = (1 << 20) + 1
    secret = bytes(range(32))
    eng = K.SetupArtifactsGenerator(secret).take(n)
    c = O.bench_coefficients(n)
    z = K.Scalar((pow(5, n - 1, K.R_MODULUS) + 20) % K.R_MODULUS)
    y = eng.evaluate_limbs(c, z)
    want = O.p1_compress(O.commit_shortcut(c, secret))
    for _ in range(2):
        got = eng.commit_limbs(c)
    assert got.compress() == want
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.commit_limbs(c)
    t_commit = (time.perf_counter() - t0) / reps
    eng.open_limbs(c, z, y)
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.open_limbs(c, z, y)
    t_open = (time.perf_counter() - t0) / reps
    import threading

    threaded = {}
    for nthreads in (2, 3, 4):
        per = 12

        def worker():
            for _ in range(per):
                eng.commit_limbs(c)

        ths = [threading.Thread(target=worker) for _ in range(nthreads)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        threaded["commitments_per_s_%d_threads" % nthreads] = round(nthreads * per / (time.perf_counter() - t0), 1)
    print(json.dumps({"degree": n - 1, "kzg_commit_host_pointer_ms": round(t_commit * 1e3, 3), **threaded,
                      "commitments_per_s_pcie_inclusive": round(1 / t_commit, 1),
                      "kzg_open_host_pointer_ms": round(t_open * 1e3, 3),
                      "opening_proofs_per_s_pcie_inclusive": round(1 / t_open, 1),
                      "note": "pageable numpy buffers, one synchronous call at a time"}))
    eng.close()


if __name__ == "__main__":
    main()
