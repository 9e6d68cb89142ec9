"""Byte equality AT THE HEADLINE'S FULL SIZE (BASELINE.json configs[1]: "SyncCircuit prove on 1 x MI355X, bit-exact vs CPU
proof bytes").  The four proofs of the default bench step are built exactly as bench.py builds them (bench.sync_step_setup:
the mainnet step main_1 -> main_2, nearx/src/builder.rs:116-164 one Ed25519 slot per validator, nearx/src/sync.rs:28-44 the
64 public I/O bytes) - the outer plonky2 proof at 2^18 rows / nineteen gate kinds / 64 real public inputs, the SHA-256
STARK (2^8 blocks), the SHA-512 STARK (2^7 blocks), the Ed25519 STARK (2^7 slots, 28 of them inactive validators), all
under the step's tag - proved on the GPU and by the CPU oracle on the same inputs, and the BYTES compared.  About 100 s
of oracle time on the GPU box's 16 host cores; the smaller shapes of test_gpu_prover.py / test_gpu_stark.py stay as they are."""
import os
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same(got, want, what):
    assert len(got) == len(want), "%s: %d bytes against the oracle's %d" % (what, len(got), len(want))
    if got != want:
        a, b = np.frombuffer(got, np.uint8), np.frombuffer(want, np.uint8)
        pytest.fail("%s: proof bytes differ from the oracle's, first at byte %d of %d" % (what, int(np.nonzero(a != b)[0][0]), len(want)))


def test_headline_step_four_proofs_bytes_equal_oracle(nlx, orc):
    import torch
    import bench
    os.environ["OMP_NUM_THREADS"] = str(min(len(os.sched_getaffinity(0)), 16))
    SA, SB, E = nlx.sha256_air, nlx.sha512_air, nlx.ed25519_air
    args = types.SimpleNamespace(log_n=18, gate_mix="nearx")   # bench.py's defaults
    st = bench.sync_step_setup(args, nlx, torch, 0, 0)
    tag = st["step_tag"]
    try:
        assert st["n_validators"] == 100 and st["n_sigs"] == 72 and st["log_slots"] == 7 and st["lb256"] == 8 and st["lb512"] == 7
        # the reference's STARK protocol (whole-row hash_or_noop leaves, every opening observed): the bench's default and the headline's
        for k in ("p256", "p512", "ped"):
            assert st[k].stark.desc.leaf_group_cols == 0 and st[k].stark.desc.openings_group == 0
        # ---- the three STARKs on the GPU (traces generated on the device), then the oracle on the reference traces ----
        got256 = st["p256"].prove(st["sha_msgs"])
        got512 = st["p512"].prove(st["sig_msgs"])
        goted = st["ped"].prove(st["slot_words"])
        blocks, first, digest = SA.blocks_for_messages(st["sha_msgs"], st["lb256"])
        tr, _ = SA.reference_trace(blocks, first)
        want = orc.stark_prove_rounds(st["p256"].stark.desc, SA.cpu_rounds(blocks, first, tr), [int(v) for v in digest] + tag)
        _same(got256[0], want, "SHA-256 STARK, 2^8 blocks")
        blocks, first, digest = SB.blocks_for_messages(st["sig_msgs"], st["lb512"])
        tr, _ = SB.reference_trace(blocks, first)
        want = orc.stark_prove_rounds(st["p512"].stark.desc, SB.cpu_rounds(blocks, first, tr), [int(v) for v in SB.digest_halves(digest)] + tag)
        _same(got512[0], want, "SHA-512 STARK, 2^7 blocks")
        # Ed25519: 2^7 slots = the step's 100 validators (72 signed, 28 inactive) + padding; the oracle proves the trace the
        # device generator wrote (it equals the Python reference bit for bit: tests/test_ed25519_air.py) with its own lookup
        # and binding columns
        ped = st["ped"]
        host = ped.generate_trace(st["slot_words"]).cpu().numpy().view(np.uint64)
        tc = ped.es.table_cols

        def cpu_round1(known):   # known = [alpha0, alpha1, gamma0, gamma1]
            acc, total = E.binding_columns(host, known[2:4])
            cols = np.concatenate([orc.logup_round(host, E.LOOKUPS, 16, host[E.MULT:E.MULT + tc], known[:2], tc),
                                   orc.logup_round(host, E.LOOKUPS9, 9, host[E.MULT9], known[:2]), acc], axis=0)
            return cols, list(total)
        want = orc.stark_prove_rounds(ped.stark.desc, lambda rnd, known: host if rnd == 0 else cpu_round1(known), tag)
        _same(goted if isinstance(goted, (bytes, bytearray)) else goted[0], want, "Ed25519 STARK, 2^7 slots (28 inactive)")
        del host
        # ---- the outer proof: 2^18 rows, nineteen gate kinds, public inputs = the step's 64 I/O bytes ----
        syn, cd = st["syn"], st["cd"]
        n = cd.prove_into(st["wires"], st["pis"].ctypes.data)
        got = cd._buf[:n].tobytes()
        ref = orc.Circuit.from_synthetic(syn)
        assert np.array_equal(cd.constants_sigmas_cap, ref.constants_sigmas_cap())
        want = ref.prove(syn.wires, syn.public_inputs)
        _same(got, want, "outer plonky2 proof, 2^18 rows")
        assert ref.verify(got) == 1
        ref.close()
    finally:
        for k in ("p256", "p512", "ped", "cd"):
            st[k].close()
        for c in st["ctxs"]:
            c.close()
