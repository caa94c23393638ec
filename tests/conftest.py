"""pytest configuration: `gpu` marker, paths, shared fixtures.

`-m "not gpu"`: oracle vs golden vectors, host logic, C-ABI export check, gloo sharding test.
`-m gpu`: parity tests proper -- every one calls through the C-ABI (ctypes) into the HIP library.
"""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    import oracle_ctypes

    oracle_ctypes.lib()
    return oracle_ctypes


@pytest.fixture(scope="session")
def twin():
    import bigint_twin

    return bigint_twin


class _EngineCache:
    """One engine per SRS length: SRS generation + window tables are built once per size."""

    def __init__(self):
        self.by_n = {}

    def bench_srs(self, n):
        import kzg_poly_commit_exploration_amd as K
        import bigint_twin

        if n not in self.by_n:
            self.by_n[n] = K.SetupArtifactsGenerator(bigint_twin.BENCH_SECRET_BE).take(n)
        return self.by_n[n]

    def close(self):
        for e in self.by_n.values():
            e.close()
        self.by_n.clear()


@pytest.fixture(scope="session")
def engines():
    cache = _EngineCache()
    yield cache
    cache.close()
