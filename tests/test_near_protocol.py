"""Witness-side Sync semantics against the reference's own pinned values: stake totals
(crates/protocol/src/lib.rs:466-497), the epoch-boundary walk (:364-405), the error cases (:407-451, 499-530)
and the SyncCircuit output of the fixture the bench uses."""
import copy
import json
import os

import pytest

from conftest import ROOT

NEAR = os.path.join(ROOT, "tests", "golden", "near")


def load(name):
    with open(os.path.join(NEAR, name)) as f:
        return json.load(f)["body"]


def test_ed25519_rfc8032_vectors(nlx):
    P = nlx.near_protocol
    # RFC 8032 §7.1 TEST 1 and TEST 2
    pk = bytes.fromhex("d75a980182b10ab7d54bfed3c964073a0ee172f3daa62325af021a68f707511a")
    sig = bytes.fromhex("e5564300c360ac729086e2cc806e828a84877f1eb8e5d974d873e065224901555fb8821590a33bacc61e39701cf9b46b"
                        "d25bf5f0595bbe24655141438e7a100b")
    assert P.ed25519_verify(pk, b"", sig)
    assert not P.ed25519_verify(pk, b"x", sig)
    pk2 = bytes.fromhex("3d4017c3e843895a92b70aa74d1b7ebc9c982ccf2ec4968cc0cd55f12af4660c")
    sig2 = bytes.fromhex("92a009a9f0d4cab8720e820b5f642540a2b27b5416503f8fb3762223ebdb69da085ac1e43e15996e458f3613d0f11d8c"
                         "387b2eaeb4302aeeb00d291612bb0c00")
    assert P.ed25519_verify(pk2, bytes.fromhex("72"), sig2)
    bad = bytearray(sig2)
    bad[40] ^= 1
    assert not P.ed25519_verify(pk2, bytes.fromhex("72"), bytes(bad))


def test_stake_totals_match_reference_literals(nlx):
    """test_next_invalid_signatures_stake_isnt_sufficient / _no_approved_stake (mainnet fixtures)."""
    P = nlx.near_protocol
    bps = load("main_0.json")["next_bps"]
    nxt = load("main_1.json")
    msg = P.reconstruct_approval_message(nxt)
    assert len(msg) == 41 and msg[0] == 0
    total, approved, flags = P.validate_signatures(nxt["approvals_after_next"], bps, msg)
    assert (total, approved) == (512915271547861520119028536348929, 345140782903867823005444871054881)
    assert P.ensure_stake_is_sufficient(total, approved)
    assert not P.ensure_stake_is_sufficient(total, total // 3 * 2 - 1)
    assert not P.ensure_stake_is_sufficient(total, total // 3 * 2)       # "<= threshold" fails
    none = [None] * len(nxt["approvals_after_next"])
    assert P.validate_signatures(none, bps, msg)[:2] == (512915271547861520119028536348929, 0)
    # test_next_invalid_signature: a real signature over a bogus message
    assert P.validate_signatures(nxt["approvals_after_next"][:1], bps[:1], b"bogus approval message")[1] == 0
    assert flags[0] is True and flags[1] is False


def test_next_bps_hash(nlx):
    P = nlx.near_protocol
    for name in ("main_1.json", "test_1.json", "test_2.json"):
        b = load(name)
        assert P.next_bps_hash(b["next_bps"]) == nlx.nearx_io.b58decode32(b["inner_lite"]["next_bp_hash"]), name


def test_sync_across_epoch_boundaries(nlx):
    """testnet: test_0 (head, bps) -> test_1 -> test_2, as the reference's test of the same name."""
    P = nlx.near_protocol
    head = load("test_0.json")
    bps = head["next_bps"]
    for name in ("test_1.json", "test_2.json"):
        nxt = load(name)
        out = P.sync(head, bps, nxt)
        assert out["new_head_hash"] == nlx.nearx_io.header_hash(nxt)
        assert out["next_bps"] == nxt["next_bps"] and out["approved"] > out["total"] // 3 * 2
        head, bps = nxt, out["next_bps"]


def test_sync_error_cases(nlx):
    P = nlx.near_protocol
    head = load("main_0.json")
    bps = head["next_bps"]
    nxt = load("main_1.json")
    assert P.sync(head, bps, nxt)["total"] == 512915271547861520119028536348929
    with pytest.raises(P.SyncError, match="BlockAlreadyVerified"):
        P.sync(nxt, bps, head)
    bad = copy.deepcopy(nxt)
    bad["inner_lite"]["epoch_id"] = head["inner_lite"]["prev_state_root"]
    with pytest.raises(P.SyncError, match="BlockNotCurrentOrNextEpoch"):
        P.sync(head, bps, bad)
    bad = copy.deepcopy(nxt)
    bad["next_bps"] = None
    if nxt["inner_lite"]["epoch_id"] == head["inner_lite"]["next_epoch_id"]:
        with pytest.raises(P.SyncError, match="NextBpsInvalid"):
            P.sync(head, bps, bad)
    bad = copy.deepcopy(nxt)
    bad["approvals_after_next"] = [None] * len(bad["approvals_after_next"])
    with pytest.raises(P.SyncError, match="NotEnoughApprovedStake"):
        P.sync(head, bps, bad)
    bad = copy.deepcopy(nxt)
    bad["next_bps"][0]["stake"] = str(int(bad["next_bps"][0]["stake"]) + 1)
    with pytest.raises(P.SyncError, match="NextBpsInvalid"):
        P.sync(head, bps, bad)


@pytest.mark.gpu
def test_gpu_sha256_stark_of_a_real_sync_step(nlx, ctx, orc):
    """The SHA-256 work of one real Sync step (mainnet fixture main_1.json) through the GPU trace generator and
    STARK prover: the proof's public digest is the header's next_bp_hash - a value fixed by NEAR mainnet data."""
    import struct
    P, SA = nlx.near_protocol, nlx.sha256_air
    nxt = load("main_1.json")
    msgs = P.sync_sha256_messages(nxt)
    n_blocks = sum(len(SA.pad_message(m)) for m in msgs)
    log_blocks = (n_blocks - 1).bit_length()
    sp = SA.Sha256Prover(ctx, log_blocks)   # 129 blocks -> 256: the filler messages come first
    proof, digest = sp.prove(msgs)
    got = b"".join(struct.pack(">I", int(x)) for x in digest)
    assert got == nlx.nearx_io.b58decode32(nxt["inner_lite"]["next_bp_hash"])
    assert orc.stark_verify(sp.stark.desc, proof) == 1
    sp.close()


def test_mainnet_walk_to_the_bench_fixture(nlx):
    """main_0 -> main_1 -> main_2: the Sync step whose I/O bytes are the bench's public inputs
    (BASELINE.json configs[0]: "SyncCircuit prove on fixtures/main_2.json")."""
    P, io = nlx.near_protocol, nlx.nearx_io
    head = load("main_0.json")
    bps = head["next_bps"]
    for name in ("main_1.json", "main_2.json"):
        nxt = load(name)
        out = P.sync(head, bps, nxt)
        head = nxt
        bps = out["next_bps"] or bps
    fx = io.load_fixture(os.path.join(NEAR, "main_2.json"))
    sync_in, sync_out = io.sync_io(fx)
    assert sync_out == out["new_head_hash"] and len(sync_in) == 32
    assert sync_in == io.header_hash(load("main_1.json"))   # the trusted hash of step 2 is the head step 1 produced
