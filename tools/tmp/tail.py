import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import kzg_poly_commit_exploration_amd as K
r = K.R_MODULUS
for degree in (100, 1000, 2500, 16384, 131072):
    n = degree + 1
    eng = K.SetupArtifactsGenerator(bytes(range(32))).take(n)
    vals, p5 = [], 1
    for _ in range(n):
        vals.append((p5 + 10) % r); p5 = p5 * 5 % r
    c = K.scalars_to_limbs(vals)
    d = eng.dev_alloc(n * 32)
    eng.dev_upload(d, np.ascontiguousarray(c))
    sub, tail, full = [], [], []
    for i in range(60):
        t0 = time.perf_counter(); eng.commit_submit(0, d, n); t1 = time.perf_counter()
        time.sleep(0.004)
        t2 = time.perf_counter(); eng.wait(0); t3 = time.perf_counter()
        sub.append(t1 - t0); tail.append(t3 - t2)
        t0 = time.perf_counter(); eng.commit_submit(0, d, n); eng.wait(0); full.append(time.perf_counter() - t0)
    med = lambda v: sorted(v)[len(v) // 2] * 1e6
    print(f"degree {degree} submit_us {med(sub):.1f} wait_after_done_us {med(tail):.1f} full_us {med(full):.1f}", flush=True)
