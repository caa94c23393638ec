#!/usr/bin/env python3
"""One job at a time (no overlap), for rocprofv3 --kernel-trace --stats: which kernels make up the latency of a single
commitment / opening.  python3 tools/prof_latency.py <degree> <commit|open> [reps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kzg_poly_commit_exploration_amd as K  # noqa: E402

degree = int(sys.argv[1])
op = sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
n = degree + 1
r = K.R_MODULUS
eng = K.SetupArtifactsGenerator(bytes(range(32))).take(n)
vals, p5 = [], 1
for _ in range(n):
    vals.append((p5 + 10) % r)
    p5 = p5 * 5 % r
c = K.scalars_to_limbs(vals)
z = K.Scalar((pow(5, degree, r) + 20) % r)
y = eng.evaluate_limbs(c, z)
d = eng.dev_alloc(n * 32)
eng.dev_upload(d, np.ascontiguousarray(c))
for phase in ("warm", "timed"):
    k = 5 if phase == "warm" else reps
    t0 = time.perf_counter()
    for _ in range(k):
        if op == "commit":
            eng.commit_submit(0, d, n)
        else:
            eng.open_submit(0, d, n, z, y)
        eng.wait(0)
    dt = (time.perf_counter() - t0) / k
print('{"degree": %d, "op": "%s", "latency_ms": %.4f}' % (degree, op, dt * 1e3))
eng.dev_free(d)
eng.close()
