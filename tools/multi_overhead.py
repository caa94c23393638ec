#!/usr/bin/env python3
"""What the multi-device context costs on top of its shards, on ONE GPU with 8 virtual slices.
  overhead  KZG_HOST_TRACE=1 makes the context time every sharded call, the slowest device thread of each of its rounds
            and the exchange + sum; call - slowest devices = what the context adds (waking and collecting the persistent
            device threads, the carry recurrence, K-1 additions, one normalisation).  The trace line is printed by the
            library when the context is destroyed; this script wraps it into JSON.
  exchange  the RCCL leg on a communicator of one (KZG_MULTI_FORCE_RCCL=1) against the host gather, at 2^16, best of
            three interleaved rounds (a fresh engine's first calls run at a lower clock).
python3 tools/multi_overhead.py > profiles/rNN_multi_overhead.jsonl"""
import json
import os
import re
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ["KZG_HOST_TRACE"] = "1"
import kzg_poly_commit_exploration_amd as K  # noqa: E402
import oracle_ctypes as O  # noqa: E402  (bench inputs only)

secret = bytes(range(32))
k = 8


def traced_close(eng):
    """closes the engine with stderr redirected, returns the library's trace line"""
    sys.stderr.flush()
    saved = os.dup(2)
    with tempfile.TemporaryFile(mode="w+b") as f:
        os.dup2(f.fileno(), 2)
        try:
            eng.close()
        finally:
            os.dup2(saved, 2)
            os.close(saved)
        f.seek(0)
        text = f.read().decode(errors="replace")
    m = re.search(r"\[kzg multi trace\] (\d+) sharded calls on (\d+) devices, us per call: total ([\d.]+) slowest device of every "
                  r"round ([\d.]+) exchange\+sum ([\d.]+) => context overhead ([\d.]+)", text)
    return None if not m else {"calls": int(m.group(1)), "devices": int(m.group(2)), "call_us": float(m.group(3)),
                               "slowest_device_us": float(m.group(4)), "exchange_and_sum_us": float(m.group(5)),
                               "context_overhead_us": float(m.group(6))}


for lg in (12, 19, 22):
    n = (1 << lg) + 1
    c = O.bench_coefficients(n)
    z = K.Scalar(777)
    for op in ("commit", "open"):
        e = K.Engine(devices=[0] * k)
        e.srs_generate(secret, n)
        y = e.evaluate_limbs(c, z)
        for _ in range(30 if lg < 22 else 10):
            if op == "commit":
                e.commit_limbs(c)
            else:
                e.open_limbs(c, z, y)
        print(json.dumps({"what": "multi-device context, 8 virtual slices of one GPU", "degree": 1 << lg, "op": op,
                          "points_per_slice": (n + k - 1) // k, **(traced_close(e) or {})}), flush=True)


def timed(fn, reps):
    for _ in range(4):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


d2 = 1 << 16
c2 = O.bench_coefficients(d2 + 1)
engines = []
for env in (None, "0", "1"):
    if env is None:
        e = K.Engine(0)
    else:
        os.environ["KZG_MULTI_FORCE_RCCL"] = env
        e = K.Engine(devices=[0])
    e.srs_generate(secret, d2 + 1)
    engines.append(e)
best = [1e9] * 3
for _ in range(3):
    for i, e in enumerate(engines):
        best[i] = min(best[i], timed(lambda e=e: e.commit_limbs(c2), 60))
ex = engines[2].rccl_exchanges()
for e in engines:
    e.close()
print(json.dumps({"what": "exchange leg at degree 2^16, one device", "single_device_commit_ms": round(best[0], 4),
                  "multi_host_gather_commit_ms": round(best[1], 4), "multi_rccl_world_of_one_commit_ms": round(best[2], 4),
                  "rccl_exchanges": ex, "rccl_leg_us": round((best[2] - best[1]) * 1e3, 1),
                  "multi_wrapper_us": round((best[1] - best[0]) * 1e3, 1)}), flush=True)
