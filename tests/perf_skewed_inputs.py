"""Commit latency at degree 2^20 for coefficient distributions other than the bench's pseudo-random ones
(GPU; checker = the oracle's known-secret shortcut [P(s)]G, test infrastructure).  One slot, host wall clock."""
import json
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import kzg_poly_commit_exploration_amd as K  # noqa: E402
import oracle_ctypes as O  # noqa: E402

SECRET = bytes(range(32))


def limbs_from_small(vals):
    """non-negative ints < 2^64 -> Montgomery limbs via the oracle's vectorised conversion"""
    return O.fr_from_ints(vals)


def main():
    n = (1 << 20) + 1
    rnd = random.Random(5)
    eng = K.SetupArtifactsGenerator(SECRET).take(n)
    eng.set_timing(True)
    dists = {
        "bench_pseudo_random": None,
        "all_ones": [1] * n,
        "bits_0_1": [rnd.getrandbits(1) for _ in range(n)],
        "small_lt_2^12": [rnd.randrange(1 << 12) for _ in range(n)],
        "i64": [rnd.randrange(1 << 63) for _ in range(n)],
        "three_values": [(3, 5, 1 << 60)[rnd.randrange(3)] for _ in range(n)],
    }
    for name, vals in dists.items():
        c = O.bench_coefficients(n) if vals is None else limbs_from_small(vals)
        want = O.p1_compress(O.commit_shortcut(c, SECRET))
        d = eng.dev_alloc(n * 32)
        eng.dev_upload(d, np.ascontiguousarray(c))
        for _ in range(2):
            eng.commit_submit(0, d, n)
            got = eng.wait(0)
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            eng.commit_submit(0, d, n)
            got = eng.wait(0)
        ms = (time.perf_counter() - t0) / reps * 1e3
        t = eng.times(0)
        eng.dev_free(d)
        print(json.dumps({"dist": name, "ms_per_commit": round(ms, 3), "ok": got.compress() == want,
                          "refs": t["references"], "phase": {k: round(v, 3) for k, v in t.items() if k.endswith("_ms")}}), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
