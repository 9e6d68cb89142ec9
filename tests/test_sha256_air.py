"""SHA-256 compression AIR.  CPU part: the AIR's reference trace computes real SHA-256
(hashlib; and the reference's own pinned header hash), satisfies every constraint row by row, and the
oracle's STARK verifier accepts / rejects as it should.  GPU part: the trace generated on the GPU equals
the reference trace bit for bit and the GPU proof bytes equal the oracle's."""
import hashlib
import os
import struct

import numpy as np
import pytest

from conftest import P, ROOT
from test_stark_cpu import run_program


def _messages():
    return [b"abc", bytes(range(100)), b"", b"x" * 55, b"y" * 56, b"z" * 64]


def test_round_constants_and_padding(nlx):
    SA = nlx.sha256_air
    assert SA.K[:4] == [0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5] and SA.K[63] == 0xc67178f2
    assert len(SA.pad_message(b"x" * 55)) == 1 and len(SA.pad_message(b"x" * 56)) == 2
    assert SA.pad_message(b"abc")[0][0] == 0x61626380 and SA.pad_message(b"abc")[0][15] == 24


def test_reference_trace_is_sha256_and_satisfies_air(nlx):
    SA = nlx.sha256_air
    msgs = _messages()
    blocks, first, digest = SA.blocks_for_messages(msgs, 4)
    assert all(first[:7] == 1) and first.tolist()[7:] == [1, 1, 0, 1, 1, 1, 0, 1, 0]   # seven filler blocks come first
    assert [int(x) for x in digest] == list(struct.unpack(">8I", hashlib.sha256(msgs[-1]).digest()))
    t, hout = SA.reference_trace(blocks, first)
    assert t.shape == (SA.N_COLS, 1024) and int(t.max()) < 2 ** 32
    # the chaining value after each message's last block is hashlib's digest
    b = 7
    for m in msgs:
        nb = len(SA.pad_message(m))
        row = 64 * (b + nb - 1) + 63
        out_cols = [int(t[SA.NEW_A, row])] + [sum(int(t[base + i, row]) << i for i in range(32)) for base in (SA.A, SA.B, SA.C)]
        out_cols += [int(t[SA.NEW_E, row])] + [sum(int(t[base + i, row]) << i for i in range(32)) for base in (SA.E, SA.F, SA.G)]
        got = [(int(t[SA.HIN + k, row]) + out_cols[k]) & 0xFFFFFFFF for k in range(8)]
        assert got == list(struct.unpack(">8I", hashlib.sha256(m).digest())), m
        b += nb
    assert [int(x) for x in hout] == [int(x) for x in digest]
    # every constraint vanishes on every row (all-rows constraints also across the wrap n-1 -> 0)
    air = SA.sha256_air()
    words = air.compile()
    per = [np.array(SA.K, dtype=np.uint64), np.array([0] * 63 + [1], dtype=np.uint64)]
    n = t.shape[1]
    rows = list(range(0, 3)) + [62, 63, 64, 65, 127, 128, 191, 192, 300, n - 2, n - 1]
    for i in rows:
        vals = run_program(words, t[:, i], t[:, (i + 1) % n], digest, periodic=[int(c[i % 64]) for c in per])
        for op, v in vals:
            if (op == 8 and i != 0) or (op == 9 and i != n - 1):
                continue
            assert v == 0, (i, op)


def test_header_hash_through_the_air(nlx):
    """The reference's pinned header hash (test_0.json -> 0x63b87190...98e3, nearx/src/builder.rs:398-417):
    its three SHA-256 calls run through the AIR's trace and come out right."""
    SA = nlx.sha256_air
    io = nlx.nearx_io
    fx = io.load_fixture(os.path.join(ROOT, "tests", "golden", "near", "test_0.json"))
    msgs = io.header_hash_preimages(fx)
    blocks, first, digest = SA.blocks_for_messages(msgs)
    t, hout = SA.reference_trace(blocks, first)
    # messages are [inner_lite, inner_lite_hash || inner_rest_hash, that_hash || prev_hash]; filler blocks come
    # first, so the last row's output chaining value - the AIR's public digest - is the header hash
    row = t.shape[1] - 1
    assert bytes(b"".join(struct.pack(">I", int(x)) for x in digest)).hex().startswith("63b87190")
    words = [int(t[SA.NEW_A, row])] + [sum(int(t[base + i, row]) << i for i in range(32)) for base in (SA.A, SA.B, SA.C)]
    words += [int(t[SA.NEW_E, row])] + [sum(int(t[base + i, row]) << i for i in range(32)) for base in (SA.E, SA.F, SA.G)]
    got = b"".join(struct.pack(">I", (int(t[SA.HIN + k, row]) + words[k]) & 0xFFFFFFFF) for k in range(8))
    assert got.hex() == "63b87190ffbaa36d7dab50f918fe36f70ab26910a0e9d797161e2356561598e3"


def test_oracle_stark_on_sha256(nlx, orc):
    SA, S = nlx.sha256_air, nlx.stark
    blocks, first, digest = SA.blocks_for_messages(_messages()[:3], 2)
    t, _ = SA.reference_trace(blocks, first)
    st = S.Stark(SA.sha256_air(), 8)
    assert st.desc.quotient_degree_factor == 2 and st.desc.n_periodic == 2 and st.desc.period_bits == 6
    proof = orc.stark_prove(st.desc, t, digest)
    assert orc.stark_verify(st.desc, proof) == 1
    for col, row in ((SA.E + 3, 70), (SA.CA, 5), (SA.WIN + 4, 64), (SA.IS_FIRST, 64), (SA.CY + 2, 127), (SA.CY + 5, 255), (SA.HIN, 100)):  # row 127: block 1 chains into block 2
        t2 = t.copy()
        t2[col, row] = (int(t2[col, row]) + 1) % P
        assert orc.stark_verify(st.desc, orc.stark_prove(st.desc, t2, digest)) != 1, (col, row)
    d2 = digest.copy()
    d2[7] ^= np.uint64(1)
    assert orc.stark_verify(st.desc, orc.stark_prove(st.desc, t, d2)) != 1


@pytest.mark.gpu
def test_gpu_trace_equals_reference(nlx, ctx):
    SA = nlx.sha256_air
    msgs = _messages() + [os.urandom(200), os.urandom(119), os.urandom(1)]
    blocks, first, digest = SA.blocks_for_messages(msgs, 4)
    want, hout = SA.reference_trace(blocks, first)
    sp = SA.Sha256Prover(ctx, 4, nlx.StarkConfig(fri_num_queries=10))
    trace, got_digest = sp.generate_trace(blocks, first)
    got = trace.cpu().numpy().view(np.uint64)
    assert np.array_equal(got_digest, digest) and np.array_equal(hout, digest)
    if not np.array_equal(got, want):
        bad = np.argwhere(got != want)[0]
        pytest.fail("GPU trace differs from the reference at column %d row %d" % (bad[0], bad[1]))
    sp.close()


@pytest.mark.gpu
@pytest.mark.parametrize("log_blocks", [0, 2, 5])
def test_gpu_sha256_proof_bytes_equal_oracle(nlx, ctx, orc, log_blocks):
    SA = nlx.sha256_air
    rng = np.random.default_rng(log_blocks)
    msgs = [bytes(rng.integers(0, 256, int(rng.integers(0, 120)), dtype=np.uint8)) for _ in range(max(1, (1 << log_blocks) // 2))]
    msgs = msgs[:1] if log_blocks == 0 else msgs
    if log_blocks == 0:
        msgs = [b"abc"]
    sp = SA.Sha256Prover(ctx, log_blocks)
    proof, digest = sp.prove(msgs)
    blocks, first, want_digest = SA.blocks_for_messages(msgs, log_blocks)
    assert np.array_equal(digest, want_digest)
    t, _ = SA.reference_trace(blocks, first)
    want = orc.stark_prove(sp.stark.desc, t, digest)
    assert len(proof) == len(want)
    if proof != want:
        a, b = np.frombuffer(proof, np.uint8), np.frombuffer(want, np.uint8)
        pytest.fail("SHA-256 STARK proof differs from the oracle, first at byte %d" % int(np.nonzero(a != b)[0][0]))
    assert orc.stark_verify(sp.stark.desc, proof) == 1
    sp.close()


@pytest.mark.gpu
def test_gpu_sha256_1024_blocks_verifies(nlx, ctx, orc):
    """2^10 blocks (65 536 rows x 302 columns): oracle verifier accepts, digest = hashlib."""
    SA = nlx.sha256_air
    rng = np.random.default_rng(7)
    msgs = [bytes(rng.integers(0, 256, 64, dtype=np.uint8)) for _ in range(512)]  # Merkle-node sized: 2 blocks each
    sp = SA.Sha256Prover(ctx, 10)
    proof, digest = sp.prove(msgs)
    assert [int(x) for x in digest] == list(struct.unpack(">8I", hashlib.sha256(msgs[-1]).digest()))
    assert orc.stark_verify(sp.stark.desc, proof) == 1
    sp.close()
