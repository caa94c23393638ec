#!/usr/bin/env python3
"""kzg_commit from N host threads at degree 2^20 with KZG_HOST_TRACE=1: one engine per thread count so that the trace
printed at close belongs to that count.  python3 tools/host_threads_probe.py [threads ...]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ.setdefault("KZG_HOST_TRACE", "1")
import kzg_poly_commit_exploration_amd as K  # noqa: E402
import oracle_ctypes as O  # noqa: E402  (bench inputs only)

n = (1 << 20) + 1
c = O.bench_coefficients(n)
for nthreads in [int(a) for a in sys.argv[1:]] or [1, 2, 3]:
    eng = K.SetupArtifactsGenerator(bytes(range(32))).take(n)
    eng.commit_limbs(c)
    per = 16

    def worker():
        for _ in range(per):
            eng.commit_limbs(c)

    ths = [threading.Thread(target=worker) for _ in range(nthreads)]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    print("threads %d: %.1f commitments/s (%.2f ms per call per thread)" % (nthreads, nthreads * per / dt, 1e3 * dt / per), flush=True)
    eng.close()
