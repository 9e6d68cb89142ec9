"""The straight-line AIR kernels generated at library build time (near-light-client_amd/airgen.py -> csrc/airgen/) against the
register-program interpreter k_air_quotient, which stays the parity reference: the three STARKs of the bench's Sync step
(the mainnet step main_1 -> main_2, under its step tag: exactly the programs the generator is fed) and the untagged SHA-256
STARK of the Verify job's map work, proved once with the generated kernels and once with NLX_AIR_VM=1 - the proof BYTES must be
equal (the quotient's values enter the quotient commitment's cap, hence every later challenge)."""
import os
import types

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("NLX_GL_GENERATOR_SET", "7") != "7" or os.environ.get("NLX_NO_AIRGEN") == "1",
                                                 reason="the generated kernels are built into the library of record only (build.py WITH_AIRGEN)")]


def _setup(nlx):
    import torch
    import bench
    return bench.sync_step_setup(types.SimpleNamespace(log_n=12, gate_mix="nearx"), nlx, torch, 0, 0)


def _close(st):
    for k in ("p256", "p512", "ped", "cd"):
        st[k].close()
    for c in st["ctxs"]:
        c.close()


def _prove_all(nlx, st):
    sa = nlx.sha256_air
    rng = np.random.default_rng(3)
    out = {"sha256": st["p256"].prove(st["sha_msgs"])[0], "sha512": st["p512"].prove(st["sig_msgs"])[0]}
    ed = st["ped"].prove(st["slot_words"])
    out["ed25519"] = ed if isinstance(ed, (bytes, bytearray)) else ed[0]
    # the untagged SHA-256 AIR (the map jobs' STARKs; 2^7 blocks = 2^9 rows)
    sp = sa.Sha256Prover(st["ctxs"][0], 7)
    msgs = [bytes(rng.integers(0, 256, 100, dtype=np.uint8)) for _ in range(40)]
    out["sha256_untagged"] = sp.prove(msgs)[0]
    kinds = {"sha256": st["p256"].prover, "sha512": st["p512"].prover, "ed25519": st["ped"].prover, "sha256_untagged": sp.prover}
    kernel = {k: int(nlx.lib.dll.nlx_stark_quotient_kernel(v.handle)) for k, v in kinds.items()}
    sp.close()
    return out, kernel


def test_generated_air_kernels_give_the_interpreters_proof_bytes(nlx):
    assert "NLX_AIR_VM" not in os.environ
    st = _setup(nlx)
    try:
        gen, kernel = _prove_all(nlx, st)
    finally:
        _close(st)
    assert kernel == {"sha256": 1, "sha512": 1, "ed25519": 1, "sha256_untagged": 1}, "a fixed program lost its generated kernel: %r" % kernel
    os.environ["NLX_AIR_VM"] = "1"
    try:
        st = _setup(nlx)
        try:
            vm, kernel = _prove_all(nlx, st)
        finally:
            _close(st)
    finally:
        del os.environ["NLX_AIR_VM"]
    assert kernel == {"sha256": 0, "sha512": 0, "ed25519": 0, "sha256_untagged": 0}
    for k in gen:
        assert len(gen[k]) == len(vm[k]) and gen[k] == vm[k], "%s: the generated kernel's proof differs from the interpreter's" % k


def test_an_unknown_program_runs_on_the_interpreter(nlx, ctx):
    S = nlx.stark
    st = S.Stark(S.fibonacci_air(), 9).build(ctx)
    assert nlx.lib.dll.nlx_stark_quotient_kernel(st.handle) == 0
    st.close()
