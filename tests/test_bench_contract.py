"""GPU: bench.py honours the driver's JSON contract (one line, required keys, roofline + cpu_baseline)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_emits_one_contract_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1",
                          "--cpu-sample", "256"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["value"] > 0 and "workload" in d["config"]
    assert d["config"]["bit_exact_vs_golden"] is True
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # the calls the Rust shim makes (host pointers) and config 5's per-GPU batch are reported beside `value`
    assert d["host_pointer_commitments_per_sec"] > 0 and d["host_pointer_proofs_per_sec"] > 0
    assert d["openings_batch8_per_sec"] > 0 and d["openings_batch8_host_pointer_per_sec"] > 0
    # ... and from several caller threads, which occupy one stream slot each
    assert d["host_pointer_commitments_per_sec_3_threads"] > 0 and d["host_pointer_proofs_per_sec_3_threads"] > 0
    assert r.get("kernel_source_hash")
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] == 1 and c["host_cores"] >= 1
