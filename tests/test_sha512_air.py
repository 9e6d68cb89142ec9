"""SHA-512 compression AIR (the hash inside every Ed25519 verification).  CPU part: the AIR's
reference trace computes real SHA-512 (hashlib), satisfies every constraint row by row, and the oracle's STARK
verifier accepts / rejects as it should.  GPU part: the trace generated on the GPU equals the reference trace bit
for bit and the GPU proof bytes equal the oracle's."""
import hashlib
import os
import struct

import numpy as np
import pytest

from conftest import P
from test_stark_cpu import run_program


def _messages():
    return [b"abc", bytes(range(150)), b"", b"x" * 111, b"y" * 112, b"z" * 128]


def _word(t, base, row):
    return sum(int(t[base + i, row]) << i for i in range(64))


def _block_output(SB, t, blk):
    """HIN + the state after round 79, read from the last row of block `blk`."""
    row = 4 * blk + 3
    fin = [_word(t, (19 - k) * SB.SLOT + SB.oA, row) for k in range(4)] + [_word(t, (19 - k) * SB.SLOT + SB.oE, row) for k in range(4)]
    hin = [int(t[SB.HIN + 2 * k, row]) | int(t[SB.HIN + 2 * k + 1, row]) << 32 for k in range(8)]
    return [(hin[k] + fin[k]) & SB.M64 for k in range(8)]


def _periodic_values(SB, row):
    q, out = row % 4, []
    for j in range(20):
        out += list(SB._halves(SB.K[20 * q + j]))
    return out + [1 if q == 0 else 0, 1 if q == 3 else 0]


def test_constants_and_padding(nlx):
    SB = nlx.sha512_air
    assert SB.K[1] == 0x7137449123ef65cd and SB.K[78] == 0x5fcb6fab3ad6faec and SB.IV[3] == 0xa54ff53a5f1d36f1
    assert len(SB.pad_message(b"x" * 111)) == 1 and len(SB.pad_message(b"x" * 112)) == 2
    assert SB.pad_message(b"abc")[0][0] == 0x6162638000000000 and SB.pad_message(b"abc")[0][15] == 24
    # an Ed25519 hash input of a NEAR approval: R || A || 41-byte message is one block
    assert len(SB.pad_message(b"r" * 32 + b"a" * 32 + b"m" * 41)) == 1


def test_reference_trace_is_sha512_and_satisfies_air(nlx):
    SB = nlx.sha512_air
    msgs = _messages()
    blocks, first, digest = SB.blocks_for_messages(msgs, 4)
    assert first.tolist() == [1] * 7 + [1, 1, 0, 1, 1, 1, 0, 1, 0]      # seven filler blocks come first
    assert [int(x) for x in digest] == list(struct.unpack(">8Q", hashlib.sha512(msgs[-1]).digest()))
    t, hout = SB.reference_trace(blocks, first)
    assert t.shape == (SB.N_COLS, 64) and int(t.max()) < 2 ** 32
    b = 7
    for m in msgs:
        nb = len(SB.pad_message(m))
        assert _block_output(SB, t, b + nb - 1) == list(struct.unpack(">8Q", hashlib.sha512(m).digest())), m
        b += nb
    assert [int(x) for x in hout] == [int(x) for x in digest]
    air = SB.sha512_air()
    words = air.compile()
    pis = SB.digest_halves(digest)
    n = t.shape[1]
    gamma = (0x0123456789abcdef, 0x0fedcba987654321)
    acc, total = SB.binding_columns(blocks, first, gamma)
    assert tuple(total) == SB.fingerprint(blocks, first, gamma)
    full = np.concatenate([t, acc], axis=0)
    values = [int(x) for x in pis] + list(gamma) + list(total)
    for i in [0, 1, 2, 3, 4, 27, 28, 35, 36, 39, 40, n - 2, n - 1]:
        vals = run_program(words, full[:, i], full[:, (i + 1) % n], values, periodic=_periodic_values(SB, i), n_public=16)
        assert len(vals) == air.num_constraints == 4952
        for op, v in vals:
            if (op == 8 and i != 0) or (op == 9 and i != n - 1) or (op == 7 and i == n - 1):
                continue
            assert v == 0, (i, op)


def test_oracle_stark_on_sha512(nlx, orc):
    SB, S = nlx.sha512_air, nlx.stark
    blocks, first, digest = SB.blocks_for_messages(_messages()[:3], 2)     # abc | 150 bytes (2 blocks) | empty
    assert first.tolist() == [1, 1, 0, 1]
    t, _ = SB.reference_trace(blocks, first)
    st = S.Stark(SB.sha512_air(), 4)
    assert st.desc.quotient_degree_factor == 2 and st.desc.n_periodic == 42 and st.desc.period_bits == 2
    pis = SB.digest_halves(digest)

    def prove(trace, p_):
        return orc.stark_prove_rounds(st.desc, SB.cpu_rounds(blocks, first, trace), p_)
    proof = prove(t, pis)
    assert orc.stark_verify(st.desc, proof) == 1
    vals = orc.stark_values(st.desc, proof)                      # digest halves | gamma | the fingerprint the proof carries
    assert tuple(vals[18:20]) == SB.fingerprint(blocks, first, vals[16:18])
    slot = SB.SLOT
    tampered = [(3 * slot + SB.oE + 40, 5), (SB.oA + 63, 0), (9 * slot + SB.oCA, 6), (9 * slot + SB.oCA + 3, 6), (2 * slot + SB.oCE + 4, 9),
                (4 * slot + SB.oW + 33, 5), (4 * slot + SB.oSW + 1, 6), (slot + SB.oCW + 2, 13), (SB.PA + 70, 2), (SB.PE + 255, 15),
                (SB.IS_FIRST, 8), (SB.CY + 3, 7), (SB.CY + 10, 15), (SB.HIN + 1, 9), (SB.HIN + 14, 12)]   # row 7: block 1 chains into block 2
    for col, row in tampered:
        t2 = t.copy()
        t2[col, row] = (int(t2[col, row]) + 1) % P
        assert orc.stark_verify(st.desc, prove(t2, pis)) != 1, (col, row)
    p2 = pis.copy()
    p2[9] ^= np.uint64(1)
    assert orc.stark_verify(st.desc, prove(t, p2)) != 1


@pytest.mark.gpu
def test_gpu_trace_equals_reference(nlx, ctx):
    SB = nlx.sha512_air
    msgs = _messages() + [os.urandom(300), os.urandom(239), os.urandom(1)]
    blocks, first, digest = SB.blocks_for_messages(msgs, 4)
    want, hout = SB.reference_trace(blocks, first)
    sp = SB.Sha512Prover(ctx, 4, nlx.StarkConfig(fri_num_queries=10))
    trace, got_digest = sp.generate_trace(blocks, first)
    got = trace.cpu().numpy().view(np.uint64)
    assert np.array_equal(got_digest, digest) and np.array_equal(hout, digest)
    if not np.array_equal(got, want):
        bad = np.argwhere(got != want)[0]
        pytest.fail("GPU trace differs from the reference at column %d row %d" % (bad[0], bad[1]))
    sp.close()


@pytest.mark.gpu
@pytest.mark.parametrize("log_blocks", [2, 5])
def test_gpu_sha512_proof_bytes_equal_oracle(nlx, ctx, orc, log_blocks):
    SB = nlx.sha512_air
    rng = np.random.default_rng(log_blocks)
    msgs = [bytes(rng.integers(0, 256, int(rng.integers(0, 240)), dtype=np.uint8)) for _ in range(max(1, (1 << log_blocks) // 2))]
    if log_blocks == 2:
        msgs = [b"abc"]
    sp = SB.Sha512Prover(ctx, log_blocks)
    proof, digest = sp.prove(msgs)
    blocks, first, want_digest = SB.blocks_for_messages(msgs, log_blocks)
    assert np.array_equal(digest, want_digest)
    t, _ = SB.reference_trace(blocks, first)
    want = orc.stark_prove_rounds(sp.stark.desc, SB.cpu_rounds(blocks, first, t), SB.digest_halves(digest))
    assert len(proof) == len(want)
    if proof != want:
        a, b = np.frombuffer(proof, np.uint8), np.frombuffer(want, np.uint8)
        pytest.fail("SHA-512 STARK proof differs from the oracle, first at byte %d" % int(np.nonzero(a != b)[0][0]))
    assert orc.stark_verify(sp.stark.desc, proof) == 1
    sp.close()


@pytest.mark.gpu
def test_gpu_sha512_of_real_approval_signatures(nlx, ctx, orc):
    """The SHA-512 work of the Ed25519 checks of a real Sync step (mainnet fixture main_1.json): one block
    R || A || approval-message per signed approval, proved in one STARK; the digests are hashlib's."""
    import json
    from conftest import ROOT
    SB, NP = nlx.sha512_air, nlx.near_protocol
    with open(os.path.join(ROOT, "tests", "golden", "near", "main_0.json")) as f:
        bps = json.load(f)["body"]["next_bps"]
    with open(os.path.join(ROOT, "tests", "golden", "near", "main_1.json")) as f:
        nxt = json.load(f)["body"]
    msg = NP.reconstruct_approval_message(nxt)
    msgs = []
    for sig, bp in zip(nxt["approvals_after_next"], bps):
        if sig is None:
            continue
        raw_sig, pk = NP._key_bytes(sig, 64), NP._key_bytes(bp["public_key"], 32)
        assert NP.ed25519_verify(pk, msg, raw_sig)
        msgs.append(raw_sig[:32] + pk + msg)
    assert len(msgs) > 32 and all(len(m) == 105 for m in msgs)
    log_blocks = (len(msgs) - 1).bit_length()
    sp = SB.Sha512Prover(ctx, log_blocks)
    proof, digest = sp.prove(msgs)
    assert [int(x) for x in digest] == list(struct.unpack(">8Q", hashlib.sha512(msgs[-1]).digest()))
    assert orc.stark_verify(sp.stark.desc, proof) == 1
    sp.close()


@pytest.mark.gpu
def test_gpu_step_tag_opens_the_transcripts(nlx, ctx, orc):
    """A tagged prover's AIR declares four extra public inputs that no constraint reads; the transcript absorbs them, so
    proofs of one job (same tag) are told apart from another job's: bytes equal the oracle's with the tag, differ between
    tags, and the verifier rejects a proof whose tag was edited afterwards."""
    SB, S = nlx.sha512_air, nlx.stark
    msgs = [b"abc", b"step tag"]
    tag_a, tag_b = S.step_tag(b"step A"), S.step_tag(b"step B")
    assert len(tag_a) == 4 and tag_a != tag_b and max(tag_a) < (1 << 56) and S.step_tag(b"step A") == tag_a
    pa, pb, plain = (SB.Sha512Prover(ctx, 2, nlx.StarkConfig(fri_num_queries=10), step_tag=t) for t in (tag_a, tag_b, None))
    assert pa.stark.desc.num_public_inputs == 20 and plain.stark.desc.num_public_inputs == 16
    proof_a, digest = pa.prove(msgs)
    proof_b, _ = pb.prove(msgs)
    assert proof_a != proof_b and orc.stark_verify(pa.stark.desc, proof_a) == 1 and orc.stark_verify(pb.stark.desc, proof_b) == 1
    blocks, first, _ = SB.blocks_for_messages(msgs, 2)
    tr, _ = SB.reference_trace(blocks, first)
    assert proof_a == orc.stark_prove_rounds(pa.stark.desc, SB.cpu_rounds(blocks, first, tr), [int(v) for v in SB.digest_halves(digest)] + tag_a)
    vals = orc.stark_values(pa.stark.desc, proof_a)
    assert list(vals[16:20]) == tag_a and tuple(vals[22:24]) == SB.fingerprint(blocks, first, vals[20:22])
    # the tag travels in the proof (public inputs): swapping it for the other job's breaks the transcript
    k = proof_a.find(int(tag_a[0]).to_bytes(8, "little"))
    assert k >= 0
    assert orc.stark_verify(pa.stark.desc, proof_a[:k] + int(tag_b[0]).to_bytes(8, "little") + proof_a[k + 8:]) != 1
    for p_ in (pa, pb, plain):
        p_.close()
