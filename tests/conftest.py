import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

P = 0xFFFFFFFF00000001


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "primitives.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def orc():
    import oracle_py
    oracle_py.dll()
    return oracle_py


@pytest.fixture(scope="session")
def nlx():
    """The product package; building the HIP library first if it is missing (hipcc cross-compiles)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("nlx_build", os.path.join(ROOT, "near-light-client_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    b.build_lib()
    import nlxpkg
    return nlxpkg.load()


@pytest.fixture(scope="session")
def ctx(nlx):
    c = nlx.Context(0)  # raises loudly if there is no gfx950 device: GPU tests must not fall back
    yield c
    c.close()


def rand_field(rng, shape):
    """uniform-ish canonical field elements"""
    v = rng.integers(0, 2**63, size=shape, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=shape, dtype=np.uint64)
    return np.where(v >= np.uint64(P), v - np.uint64(P), v).astype(np.uint64)
