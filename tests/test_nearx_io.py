"""CPU tests pinned by the REFERENCE'S OWN artefacts: fixtures copied as data under tests/golden/near/
(fixtures/test_{0,1,2}.json, request inputs of fixtures/{sync,verify}_proof.json) and literals present in
the reference's sources."""
import json
import os

import numpy as np
import pytest

from conftest import ROOT

NEAR = os.path.join(ROOT, "tests", "golden", "near")
# bin/operator/src/succinct/mod.rs:559, nearx/contract/test/NearX.t.sol:16, scripts/forge-script.sh:10
HEADER_HASH_TEST_0 = "63b87190ffbaa36d7dab50f918fe36f70ab26910a0e9d797161e2356561598e3"


def _io(nlx):
    import importlib
    return importlib.import_module("nlx_amd.nearx_io")


def test_header_hash_matches_reference_literal(nlx):
    io = _io(nlx)
    fx = io.load_fixture(os.path.join(NEAR, "test_0.json"))
    assert io.header_hash(fx["body"]).hex() == HEADER_HASH_TEST_0


def test_header_hash_chains_to_next_fixture(nlx):
    """nearx test_header_hash (builder.rs:398-417) compares the circuit's hash with near-primitives';
    here: hash(test_1 header) is the last_block_hash recorded by the following fixture."""
    io = _io(nlx)
    f1 = io.load_fixture(os.path.join(NEAR, "test_1.json"))
    f2 = io.load_fixture(os.path.join(NEAR, "test_2.json"))
    assert io.header_hash(f1["body"]) == io.b58decode32(f2["last_block_hash"])
    inp, out = io.sync_io(f1)
    assert len(inp) == 32 and len(out) == 32 and out == io.b58decode32(f2["last_block_hash"])


def test_request_inputs_have_the_circuit_io_shape(nlx):
    io = _io(nlx)
    req = json.load(open(os.path.join(NEAR, "succinct_inputs.json")))
    sync_in = bytes.fromhex(req["sync_input"][2:])
    assert len(sync_in) == 32                                   # sync.rs:37 evm_read 32 B
    raw = bytes.fromhex(req["verify_input"][2:])
    assert len(raw) == 32 + 128 * 97 == 12448                   # verify.rs:47-55, Mainnet VERIFY_AMT = 128
    header, ids = io.decode_verify_input(raw)
    assert len(ids) == 128
    assert io.encode_verify_input(header, ids) == raw           # codec round trip on the real blob
    is_tx, h, acct = ids[0]
    assert len(h) == 32 and acct and "," not in acct
    # the packed encoding in the contract test (NearX.t.sol:45): zavodil.testnet padded with ','
    enc = io.encode_id(True, bytes.fromhex("2c53bcfe871da28decc45c3437f5864568d91af6d990dbc2662f11ce44c18d79"), "zavodil.testnet")
    assert enc.hex().startswith("012c53bcfe871da28decc45c3437f5864568d91af6d990dbc2662f11ce44c18d797a61766f64696c2e746573746e65742c2c")
    assert len(enc) == 97


def test_public_inputs_from_real_sync_io(nlx):
    """the prover's public inputs for a SyncCircuit-shaped job are the real 64 I/O bytes of the fixture"""
    io = _io(nlx)
    inp, out = io.sync_io(io.load_fixture(os.path.join(NEAR, "test_0.json")))
    pis = io.bytes_to_field_elements(inp + out)
    assert pis.shape == (64,) and pis.dtype == np.uint64 and int(pis.max()) < 256
    syn = nlx.SyntheticCircuit(6, seed=1, num_public_inputs=64)
    syn.set_public_inputs(pis)
    assert np.array_equal(syn.public_inputs, pis)

# nearx/src/builder.rs:642, nearx/src/merkle.rs:95, crates/protocol/src/experimental.rs:337
BLOCK_MERKLE_ROOT = "WWrLWbWHwSmjtTn5oBZPYgRCuCYn6fkYVa4yhPWNK4L"


def _b58_any(io, s):
    n = 0
    for ch in s:
        n = n * 58 + io._B58.index(ch)
    return n.to_bytes(32, "big")


def test_inclusion_proof_blackbox_fixture(nlx):
    """beefy_builder_test_proof_blackbox (nearx/src/builder.rs:637-662): fixtures/old.json verifies
    under the block root literal; any broken link is rejected."""
    io = _io(nlx)
    root = _b58_any(io, BLOCK_MERKLE_ROOT)
    proof = json.load(open(os.path.join(NEAR, "old.json")))
    assert io.inclusion_proof_verify(root, proof) is True
    bad = json.loads(json.dumps(proof))
    bad["block_proof"][3]["direction"] = "Left" if bad["block_proof"][3]["direction"] == "Right" else "Right"
    assert io.inclusion_proof_verify(root, bad) is False
    bad = json.loads(json.dumps(proof))
    bad["outcome_proof"]["outcome"]["gas_burnt"] += 1
    assert io.inclusion_proof_verify(root, bad) is False
    assert io.inclusion_proof_verify(b"\x00" * 32, proof) is False
    # the header of the proven block hashes to the block hash the outcome proof names
    assert io.header_hash(proof["block_header_lite"]) == io.b58decode32(proof["outcome_proof"]["block_hash"])


def test_inclusion_proof_sha256_messages_reproduce_the_check(nlx):
    """the SHA-256 work list of one inclusion proof (what a Verify map job hands to curta_sha256): hashing the messages
    in order reproduces every value `inclusion_proof_verify` compares, for both of the reference's proof fixtures"""
    import hashlib
    io = _io(nlx)
    for name, root in (("old.json", _b58_any(io, BLOCK_MERKLE_ROOT)), ("new.json", None)):
        proof = json.load(open(os.path.join(NEAR, name)))
        msgs, vals = io.inclusion_proof_sha256_messages(proof)
        assert len(msgs) == 3 + 1 + len(proof["outcome_proof"]["outcome"]["logs"]) + 1 + len(proof["outcome_proof"]["proof"]) \
            + 1 + len(proof["outcome_root_proof"]) + len(proof["block_proof"])
        assert hashlib.sha256(msgs[2]).digest() == vals["block_hash"] == io.b58decode32(proof["outcome_proof"]["block_hash"])
        assert vals["outcome_root"] == io.b58decode32(proof["block_header_lite"]["inner_lite"]["outcome_root"])
        assert hashlib.sha256(msgs[-1]).digest() == vals["block_root"]
        if root is not None:
            assert vals["block_root"] == root and io.inclusion_proof_verify(root, proof)
        else:
            assert io.inclusion_proof_verify(vals["block_root"], proof)
        # a chain: every path message contains the digest of the message that produced its child
        assert hashlib.sha256(msgs[0]).digest() == msgs[1][:32]


@pytest.mark.gpu
def test_verify_side_sha256_stark_from_the_proof_fixtures(nlx, ctx, orc):
    """Row f.2: the SHA-256 work of a Verify map job on REAL inputs - the two inclusion proofs of the reference's fixtures
    (fixtures/old.json, new.json) - proved in one GPU STARK whose public digest is the block Merkle root the last proof
    climbs to (old.json: the literal the reference's own test pins, nearx/src/builder.rs:642); the oracle verifier accepts
    it and the round value is the fingerprint of exactly these messages and digests."""
    import struct
    io = _io(nlx)
    SA = nlx.sha256_air
    msgs = []
    for name in ("new.json", "old.json"):
        m, vals = io.inclusion_proof_sha256_messages(json.load(open(os.path.join(NEAR, name))))
        msgs += m
    n_blocks = sum(len(SA.pad_message(m)) for m in msgs)
    lb = max(2, (n_blocks - 1).bit_length())
    pr = SA.Sha256Prover(ctx, lb)
    proof, digest = pr.prove(msgs)
    assert b"".join(struct.pack(">I", int(x)) for x in digest) == _b58_any(io, BLOCK_MERKLE_ROOT)
    assert orc.stark_verify(pr.stark.desc, proof) == 1
    vals = orc.stark_values(pr.stark.desc, proof)     # digest (8) | gamma (2) | the fingerprint the proof carries (2)
    blocks, first, _ = SA.blocks_for_messages(msgs, lb)
    assert tuple(vals[10:12]) == SA.fingerprint(blocks, first, vals[8:10])
    pr.close()
