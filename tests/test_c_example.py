"""include/nlx.h is a plain-C header: a C caller compiles and links against libnlx.so with gcc (CPU), fails
loudly without a GPU, and proves on the GPU."""
import os
import subprocess

import pytest

from conftest import ROOT


def _build(tmp_path, name="prove_example"):
    from conftest import GOLDEN_SUFFIX
    exe = str(tmp_path / name)
    lib_dir = os.path.join(ROOT, "near-light-client_amd")
    cmd = ["gcc", "-std=c11", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", name + ".c"), "-L", lib_dir, "-lnlx" + GOLDEN_SUFFIX, "-lnlx_synth" + GOLDEN_SUFFIX,
           "-Wl,-rpath," + lib_dir, "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True)
    return exe


def test_c_caller_builds_and_fails_loudly_without_gpu(nlx, tmp_path):
    exe = _build(tmp_path)
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    r = subprocess.run([exe, "8"], capture_output=True, text=True)
    assert r.returncode == 1 and "nlx_ctx_create" in r.stderr  # no CPU fallback


@pytest.mark.gpu
def test_c_caller_proves_on_gpu(nlx, tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe, "11"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.startswith("ok: 2^11 rows")


@pytest.mark.gpu
def test_c_caller_sees_return_codes_for_failed_allocations(nlx, tmp_path):
    """A host allocation that fails inside the library (C++ exception, stopped at the ABI), a 2^50-byte device buffer and
    a 65 535-column x 2^27-row commitment: negative return codes, no abort, and the same context proves afterwards."""
    exe = _build(tmp_path)
    r = subprocess.run([exe, "9", "nomem"], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, (r.stdout, r.stderr)
    lines = r.stdout.splitlines()
    assert lines[0].startswith("refused: host allocation -2") and lines[1].startswith("ok: 2^9 rows")


@pytest.mark.gpu
def test_c_caller_proves_a_circuit_with_lookup_tables(nlx, ctx, orc, tmp_path):
    """the descriptor's table arrays from plain C; the proof has the size the Python path's (byte-equal to the oracle's,
    tests/test_gpu_lookup.py) has for the same circuit"""
    exe = _build(tmp_path)
    r = subprocess.run([exe, "11", "lookups"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.startswith("ok: 2^11 rows") and "2 lookup tables" in r.stdout
    syn = nlx.SyntheticCircuit(11, seed=42, num_public_inputs=4, pct_poseidon=20, pct_arithmetic=30, pct_base_sum=5, pct_constant=5,
                               pct_extension=10, num_luts=2, lut_bits=8, num_lookups=300)
    cd = nlx.CircuitData.from_synthetic(ctx, syn)
    got = cd.prove(syn.wires, syn.public_inputs)
    ref = orc.Circuit.from_synthetic(syn)
    assert got == ref.prove(syn.wires, syn.public_inputs)
    assert "proof %d bytes" % len(got) in r.stdout and "circuit digest %016x" % int(cd.circuit_digest[0]) in r.stdout
    cd.close()
    ref.close()


def test_c_stark_caller_builds(nlx, tmp_path):
    _build(tmp_path, "stark_example")


@pytest.mark.gpu
def test_c_stark_caller_matches_python_path(nlx, ctx, tmp_path):
    """examples/stark_example.c (hand-written Fibonacci AIR bytecode, trace in an nlx_buf) produces the same proof
    as the Python assembler + prover for the same AIR, trace and configuration."""
    exe = _build(tmp_path, "stark_example")
    r = subprocess.run([exe, "9"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    S = nlx.stark
    t, pis = S.fibonacci_trace(9, 3, 5)
    # the C example lays its program out by hand and prints it: the program is part of the statement (the transcript opens
    # with its digest), so the Python prover is given the same words for the same AIR
    words = [int(w, 16) for w in r.stdout.split("program:")[1].splitlines()[0].split()]
    st = S.Stark(S.fibonacci_air(), 9, program=words)
    pr = st.build(ctx)
    proof = pr.prove(t, pis)
    fold = 0
    for i in range(0, len(proof) - 7, 8):
        fold = ((fold * 0x100000001B3) ^ int.from_bytes(proof[i:i + 8], "little")) & 0xFFFFFFFFFFFFFFFF
    assert ("proof %d bytes, fold %016x" % (len(proof), fold)) in r.stdout, r.stdout
    pr.close()


def test_c_rounds_caller_builds(nlx, tmp_path):
    _build(tmp_path, "rounds_example")


@pytest.mark.gpu
def test_c_rounds_caller_matches_python_path(nlx, ctx, orc, tmp_path):
    """examples/rounds_example.c (multi-round proof with a round value through the C callback) produces the same proof as
    the Python assembler + prover for the same AIR and column, and its round value is the column's fingerprint."""
    import numpy as np
    from conftest import P
    from test_stark_cpu import fingerprint_air, fingerprint_rounds
    exe = _build(tmp_path, "rounds_example")
    r = subprocess.run([exe, "9"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    S = nlx.stark
    v, x = [], 88172645463325252
    for _ in range(1 << 9):
        x ^= (x << 13) & 0xFFFFFFFFFFFFFFFF
        x ^= x >> 7
        x ^= (x << 17) & 0xFFFFFFFFFFFFFFFF
        v.append(x % P)
    v = np.array(v, dtype=np.uint64)
    words = [int(w, 16) for w in r.stdout.split("program:")[1].splitlines()[0].split()]
    pr = S.Stark(fingerprint_air(S), 9, program=words).build(ctx)
    proof = pr.prove_rounds(fingerprint_rounds(v), [])
    fold = 0
    for i in range(0, len(proof) - 7, 8):
        fold = ((fold * 0x100000001B3) ^ int.from_bytes(proof[i:i + 8], "little")) & 0xFFFFFFFFFFFFFFFF
    gamma, total = orc.stark_values(pr.stark.desc, proof)
    assert ("proof %d bytes, fold %016x, round value %d" % (len(proof), fold, total)) in r.stdout, r.stdout
    acc = 0
    for e in v:
        acc = (acc * gamma + int(e)) % P
    assert acc == total
    pr.close()
