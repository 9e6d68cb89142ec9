"""CPU side of the generated AIR kernels: the generator is deterministic, what it would write is what csrc/airgen/ holds (the
library was built from the current AIR definitions), and the hash it keys a kernel by is the hash the library computes of the
program a Stark of that AIR hands to nlx_stark_build."""
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.skipif(os.environ.get("NLX_GL_GENERATOR_SET", "7") != "7" or os.environ.get("NLX_NO_AIRGEN") == "1",
                                reason="the generated kernels are built into the library of record only (build.py WITH_AIRGEN)")

PKG = os.path.join(ROOT, "near-light-client_amd")


def test_generated_sources_are_current_and_keyed_by_the_programs_hash(nlx):
    # the generator runs in a child interpreter, as in build.py (it stubs the package's ctypes layer)
    code = ("import sys, json; sys.path.insert(0, %r); import airgen; s = airgen.sources(); "
            "print(json.dumps({k: __import__('hashlib').sha256(v.encode()).hexdigest() for k, v in s.items()}))" % PKG)
    a = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert a.returncode == 0, a.stderr[-2000:]
    import hashlib
    import json
    want = json.loads(a.stdout)
    have = {f: hashlib.sha256(open(os.path.join(PKG, "csrc", "airgen", f), "rb").read()).hexdigest()
            for f in os.listdir(os.path.join(PKG, "csrc", "airgen")) if f.endswith(".hip")}
    assert have == want, "csrc/airgen/ is stale: run python near-light-client_amd/build.py"
    # the key of a kernel = FNV-1a of the canonicalised words of the program the package builds for that AIR
    sys.path.insert(0, PKG)
    import airgen
    S = nlx.stark
    progs = {"sha256_tagged": S.Stark(nlx.sha256_air.sha256_air(tagged=True), 10).program,
             "sha256": S.Stark(nlx.sha256_air.sha256_air(), 10).program,
             "sha512_tagged": S.Stark(nlx.sha512_air.sha512_air(tagged=True), 9).program,
             "ed25519_2p7_tagged": nlx.ed25519_air.Ed25519Stark(7, tagged=True).stark.program}
    for name, words in progs.items():
        text = open(os.path.join(PKG, "csrc", "airgen", "air_%s.hip" % name)).read()
        m = re.search(r"airgen_entry_%s = \{0x([0-9a-f]{16})ull, (\d+)u" % name, text)
        assert m, name
        assert int(m.group(1), 16) == airgen.program_hash(airgen.canonical_words(words)) and int(m.group(2)) == len(words), name
