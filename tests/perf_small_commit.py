"""Launch-bound sizes: commitments/s through kzg_commit_submit / kzg_wait (device-resident coefficients, all slots in
flight) and the latency of one commitment at a time.  GPU; prints JSON lines."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import kzg_poly_commit_exploration_amd as K  # noqa: E402
import oracle_ctypes as O  # noqa: E402  (bench inputs + checker only)


def run(degree, reps=400):
    n = degree + 1
    secret = bytes(range(32))
    eng = K.SetupArtifactsGenerator(secret).take(n)
    c = O.bench_coefficients(n)
    want = O.p1_compress(O.commit_shortcut(c, secret))
    d = eng.dev_alloc(n * 32)
    eng.dev_upload(d, np.ascontiguousarray(c))
    slots = eng.num_slots()
    out = None
    for phase in ("warm", "timed"):
        k = 30 if phase == "warm" else reps
        t0 = time.perf_counter()
        inflight = []
        for i in range(k):
            s = i % slots
            if len(inflight) == slots:
                out = eng.wait(inflight.pop(0))
            eng.commit_submit(s, d, n)
            inflight.append(s)
        while inflight:
            out = eng.wait(inflight.pop(0))
        dt = time.perf_counter() - t0
    assert out.compress() == want
    # one at a time (latency)
    t0 = time.perf_counter()
    for _ in range(100):
        eng.commit_submit(0, d, n)
        eng.wait(0)
    lat = (time.perf_counter() - t0) / 100
    # the reference's other benches at the same degrees: evaluation proof (benches/evaluation_proof.rs) and evaluation
    # (benches/polynomial_evaluation.rs), one at a time, coefficients resident
    z = O.bench_input_point(degree)
    y = O.poly_evaluate(c, z)
    proof = None
    for k in (10, 100):
        t0 = time.perf_counter()
        for _ in range(k):
            eng.open_submit(0, d, n, K.Scalar.from_limbs(z), K.Scalar.from_limbs(y))
            proof = eng.wait(0)
        open_lat = (time.perf_counter() - t0) / k
    rc, q = O.quotient(c, z, y)
    assert rc == 0
    assert proof.compress() == O.p1_compress(O.commit_shortcut(q, secret))
    eng.dev_free(d)
    eng.close()
    return reps / dt, lat * 1e3, open_lat * 1e3


if __name__ == "__main__":
    for degree in (1, 100, 500, 1000, 2500, 16384):  # the reference's bench degrees, and 2^14
        v, lat, open_lat = run(degree)
        print(json.dumps({"degree": degree, "commitments_per_s": round(v, 1), "single_commit_latency_ms": round(lat, 4),
                          "single_opening_proof_latency_ms": round(open_lat, 4)}), flush=True)
