import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

P = 0xFFFFFFFF00000001


def field_generators(gen_set=None):
    """(set, MULTIPLICATIVE_GROUP_GENERATOR, POWER_OF_TWO_GENERATOR) read from include/nlx_field.h - the one definition;
    NLX_GL_GENERATOR_SET picks the set, exactly as the two library builds do."""
    import re
    gen_set = gen_set or os.environ.get("NLX_GL_GENERATOR_SET", "7")
    with open(os.path.join(ROOT, "include", "nlx_field.h")) as f:
        text = f.read()
    m = re.search(r"NLX_GL_GENERATOR_SET == %s\s*\n#define NLX_GL_MULTIPLICATIVE_GROUP_GENERATOR (\d+)ULL\s*\n"
                  r"#define NLX_GL_POWER_OF_TWO_GENERATOR (\d+)ULL" % gen_set, text)
    assert m, "generator set %s is not defined in include/nlx_field.h" % gen_set
    return gen_set, int(m.group(1)), int(m.group(2))


GEN_SET, GEN, POW2_GEN = field_generators()
GOLDEN_SUFFIX = "" if GEN_SET == "7" else "_gen" + GEN_SET


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "primitives%s.json" % GOLDEN_SUFFIX)) as f:
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
