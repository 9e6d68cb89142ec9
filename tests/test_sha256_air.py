"""SHA-256 compression AIR (sixteen rounds per row, four rows per block).  CPU part: the AIR's reference trace computes real SHA-256
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


def _word(t, base, row):
    return sum(int(t[base + i, row]) << i for i in range(32))


def _block_output(SA, t, blk):
    """HIN + the state after round 63, read from the last row of block `blk`."""
    row = 4 * blk + 3
    fin = [_word(t, (15 - k) * SA.SLOT + SA.oA, row) for k in range(4)] + [_word(t, (15 - k) * SA.SLOT + SA.oE, row) for k in range(4)]
    return [(int(t[SA.HIN + k, row]) + fin[k]) & 0xFFFFFFFF for k in range(8)]


def _periodic_values(SA, row):
    q = row % 4
    return [SA.K[16 * q + j] for j in range(16)] + [1 if q == 0 else 0, 1 if q == 3 else 0]


def test_reference_trace_is_sha256_and_satisfies_air(nlx):
    SA = nlx.sha256_air
    msgs = _messages()
    blocks, first, digest = SA.blocks_for_messages(msgs, 4)
    assert all(first[:7] == 1) and first.tolist()[7:] == [1, 1, 0, 1, 1, 1, 0, 1, 0]   # seven filler blocks come first
    assert [int(x) for x in digest] == list(struct.unpack(">8I", hashlib.sha256(msgs[-1]).digest()))
    t, hout = SA.reference_trace(blocks, first)
    assert t.shape == (SA.N_COLS, 64) and int(t.max()) < 2 ** 32
    # the chaining value after each message's last block is hashlib's digest
    b = 7
    for m in msgs:
        nb = len(SA.pad_message(m))
        assert _block_output(SA, t, b + nb - 1) == list(struct.unpack(">8I", hashlib.sha256(m).digest())), m
        b += nb
    assert [int(x) for x in hout] == [int(x) for x in digest]
    # every constraint vanishes on every row (all-rows constraints also across the wrap n-1 -> 0)
    air = SA.sha256_air()
    words = air.compile()
    n = t.shape[1]
    gamma = (0x0123456789abcdef, 0x0fedcba987654321)
    acc, total = SA.binding_columns(blocks, first, gamma)
    assert tuple(total) == SA.fingerprint(blocks, first, gamma)
    full = np.concatenate([t, acc], axis=0)
    values = [int(x) for x in digest] + list(gamma) + list(total)          # public inputs | challenges of round 0 | values of round 1
    for i in list(range(0, 9)) + [27, 28, 35, 36, 39, 40, n - 2, n - 1]:
        vals = run_program(words, full[:, i], full[:, (i + 1) % n], values, periodic=_periodic_values(SA, i), n_public=8)
        assert len(vals) == 2048
        for op, v in vals:
            if (op == 8 and i != 0) or (op == 9 and i != n - 1) or (op == 7 and i == n - 1):
                continue
            assert v == 0, (i, op)
    # the accumulator is bound: another value in a row, or another total, breaks a constraint
    bad = full.copy()
    bad[SA.ACC, 9] = (int(bad[SA.ACC, 9]) + 1) % P
    assert any(v != 0 for _, v in run_program(words, bad[:, 8], bad[:, 9], values, periodic=_periodic_values(SA, 8), n_public=8))
    vals = run_program(words, full[:, n - 1], full[:, 0], values[:10] + [(total[0] + 1) % P, total[1]], periodic=_periodic_values(SA, n - 1), n_public=8)
    assert any(v != 0 for op, v in vals if op == 9)


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
    # first, so the last block's output chaining value - the AIR's public digest - is the header hash
    assert bytes(b"".join(struct.pack(">I", int(x)) for x in digest)).hex().startswith("63b87190")
    got = b"".join(struct.pack(">I", x) for x in _block_output(SA, t, len(blocks) - 1))
    assert got.hex() == "63b87190ffbaa36d7dab50f918fe36f70ab26910a0e9d797161e2356561598e3"


def test_oracle_stark_on_sha256(nlx, orc):
    SA, S = nlx.sha256_air, nlx.stark
    blocks, first, digest = SA.blocks_for_messages(_messages()[:3], 2)     # abc | 100 bytes (2 blocks) | empty
    assert first.tolist() == [1, 1, 0, 1]
    t, _ = SA.reference_trace(blocks, first)
    st = S.Stark(SA.sha256_air(), 4)
    assert st.desc.quotient_degree_factor == 2 and st.desc.n_periodic == 18 and st.desc.period_bits == 2 and st.desc.n_rounds == 2

    def prove(trace, pis):
        return orc.stark_prove_rounds(st.desc, SA.cpu_rounds(blocks, first, trace), pis)
    proof = prove(t, digest)
    assert orc.stark_verify(st.desc, proof) == 1
    vals = orc.stark_values(st.desc, proof)                      # digest | gamma | the fingerprint the proof carries
    assert tuple(vals[10:12]) == SA.fingerprint(blocks, first, vals[8:10])
    other = blocks.copy()
    other[0, 0] ^= 1
    assert tuple(vals[10:12]) != SA.fingerprint(other, first, vals[8:10])
    slot = SA.SLOT
    tampered = [(3 * slot + SA.oE + 3, 5), (SA.oA + 31, 0), (9 * slot + SA.oCA, 6), (2 * slot + SA.oCE + 1, 9), (4 * slot + SA.oW + 4, 5),
                (4 * slot + SA.oSW, 6), (slot + SA.oCW, 13), (SA.PA + 40, 2), (SA.PE + 127, 15), (SA.IS_FIRST, 8), (SA.IS_FIRST, 4),
                (SA.CY + 2, 7), (SA.CY + 5, 15), (SA.HIN, 9), (SA.HIN + 7, 12)]   # row 7: block 1 chains into block 2
    for col, row in tampered:
        t2 = t.copy()
        t2[col, row] = (int(t2[col, row]) + 1) % P
        assert orc.stark_verify(st.desc, prove(t2, digest)) != 1, (col, row)
    d2 = digest.copy()
    d2[7] ^= np.uint64(1)
    assert orc.stark_verify(st.desc, prove(t, d2)) != 1
    # a different message in row 0 of a block (free columns) is a different statement: the digest no longer matches
    t3, _ = SA.reference_trace(SA.blocks_for_messages([b"abd", bytes(range(100)), b"x"], 2)[0], first)
    assert orc.stark_verify(st.desc, prove(t3, digest)) != 1


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
@pytest.mark.parametrize("log_blocks", [2, 3, 6])
def test_gpu_sha256_proof_bytes_equal_oracle(nlx, ctx, orc, log_blocks):
    SA = nlx.sha256_air
    rng = np.random.default_rng(log_blocks)
    msgs = [bytes(rng.integers(0, 256, int(rng.integers(0, 120)), dtype=np.uint8)) for _ in range(max(1, (1 << log_blocks) // 2))]
    if log_blocks == 2:
        msgs = [b"abc"]
    sp = SA.Sha256Prover(ctx, log_blocks)
    proof, digest = sp.prove(msgs)
    blocks, first, want_digest = SA.blocks_for_messages(msgs, log_blocks)
    assert np.array_equal(digest, want_digest)
    t, _ = SA.reference_trace(blocks, first)
    want = orc.stark_prove_rounds(sp.stark.desc, SA.cpu_rounds(blocks, first, t), digest)
    assert len(proof) == len(want)
    if proof != want:
        a, b = np.frombuffer(proof, np.uint8), np.frombuffer(want, np.uint8)
        pytest.fail("SHA-256 STARK proof differs from the oracle, first at byte %d" % int(np.nonzero(a != b)[0][0]))
    assert orc.stark_verify(sp.stark.desc, proof) == 1
    sp.close()


@pytest.mark.gpu
def test_gpu_sha256_1024_blocks_verifies(nlx, ctx, orc):
    """2^10 blocks (4 096 rows x 1 953 columns): oracle verifier accepts, digest = hashlib."""
    SA = nlx.sha256_air
    rng = np.random.default_rng(7)
    msgs = [bytes(rng.integers(0, 256, 64, dtype=np.uint8)) for _ in range(512)]  # Merkle-node sized: 2 blocks each
    sp = SA.Sha256Prover(ctx, 10)
    proof, digest = sp.prove(msgs)
    assert [int(x) for x in digest] == list(struct.unpack(">8I", hashlib.sha256(msgs[-1]).digest()))
    assert orc.stark_verify(sp.stark.desc, proof) == 1
    sp.close()
