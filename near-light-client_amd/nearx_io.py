"""Host-side data formats either side of the hot path: the NEAR header hash that
SyncCircuit outputs and the EVM byte encodings of the circuits' inputs / outputs.

Follows the reference directly (sources present in /root/reference):
  * header hash: nearx/src/variables.rs:66-73 (HeaderVariable::hash) and :161-187, 278-312
    (HeaderInnerVariable::encode_borsh / EncodeInner): sha256( sha256( sha256(inner_lite_208B) ||
    inner_rest_hash ) || prev_block_hash ), inner_lite = height LE u64 | epoch_id | next_epoch_id |
    prev_state_root | outcome_root | timestamp LE u64 | next_bp_hash | block_merkle_root.
  * SyncCircuit I/O: 32-byte trusted header hash in, 32-byte new head hash out (nearx/src/sync.rs:37,43).
  * VerifyCircuit input: 32-byte header hash + N x 97-byte ids = 1 flag byte | 32-byte hash | account id
    right-padded with ',' to 64 bytes (nearx/src/variables.rs:686-693, crates/primitives/src/lib.rs:12-22,
    nearx/contract/src/interfaces/INearX.sol:47-69); output N x 33 bytes (verify.rs:94-98).
No GPU work happens here; the prover's public inputs are derived from these bytes.
"""
import hashlib
import json

_B58 = "123456789ABCDEFGHJKLMNPQRSTUVWXYZabcdefghijkmnopqrstuvwxyz"
ACCOUNT_LEN = 64
ID_LEN = 1 + 32 + ACCOUNT_LEN  # 97


def b58decode32(s):
    n = 0
    for ch in s:
        n = n * 58 + _B58.index(ch)
    raw = n.to_bytes(32, "big") if n.bit_length() <= 256 else None
    if raw is None:
        raise ValueError("not a 32-byte base58 value")
    return raw


def inner_lite_bytes(inner):
    ts = inner["timestamp"] if int(inner.get("timestamp", 0)) > 0 else int(inner["timestamp_nanosec"])
    out = int(inner["height"]).to_bytes(8, "little")
    for k in ("epoch_id", "next_epoch_id", "prev_state_root", "outcome_root"):
        out += b58decode32(inner[k])
    out += int(ts).to_bytes(8, "little")
    for k in ("next_bp_hash", "block_merkle_root"):
        out += b58decode32(inner[k])
    assert len(out) == 208
    return out


def header_hash(block_view):
    """CryptoHash of a LightClientBlockView's header."""
    inner = hashlib.sha256(inner_lite_bytes(block_view["inner_lite"])).digest()
    lite_rest = hashlib.sha256(inner + b58decode32(block_view["inner_rest_hash"])).digest()
    return hashlib.sha256(lite_rest + b58decode32(block_view["prev_block_hash"])).digest()


def header_hash_preimages(fixture_or_view):
    """The three SHA-256 preimages behind header_hash, in evaluation order (each hash feeds the next): what
    curta_sha256 is asked to prove for one header (nearx/src/variables.rs:66-73)."""
    view = fixture_or_view.get("body", fixture_or_view)
    m0 = inner_lite_bytes(view["inner_lite"])
    m1 = hashlib.sha256(m0).digest() + b58decode32(view["inner_rest_hash"])
    m2 = hashlib.sha256(m1).digest() + b58decode32(view["prev_block_hash"])
    return [m0, m1, m2]


def load_fixture(path):
    """LightClientFixture { last_block_hash, body } (crates/test-utils/src/lib.rs:11-15)"""
    with open(path) as f:
        return json.load(f)


def sync_io(fixture):
    """(input_bytes, output_bytes) of SyncCircuit for a fixture: trusted hash in, new head hash out."""
    return b58decode32(fixture["last_block_hash"]), header_hash(fixture["body"])


def encode_id(is_transaction, id_hash, account):
    acct = account.encode()
    if len(acct) > ACCOUNT_LEN:
        raise ValueError("account id longer than 64 bytes")
    return bytes([1 if is_transaction else 0]) + id_hash + acct + b"," * (ACCOUNT_LEN - len(acct))


def decode_id(raw):
    if len(raw) != ID_LEN:
        raise ValueError("an id is 97 bytes")
    return bool(raw[0]), raw[1:33], raw[33:].rstrip(b",").decode()


def encode_verify_input(header, ids):
    out = bytes(header)
    for is_tx, h, acct in ids:
        out += encode_id(is_tx, h, acct)
    return out


def decode_verify_input(raw):
    if (len(raw) - 32) % ID_LEN:
        raise ValueError("length must be 32 + 97 k")
    return raw[:32], [decode_id(raw[32 + i * ID_LEN: 32 + (i + 1) * ID_LEN]) for i in range((len(raw) - 32) // ID_LEN)]


def bytes_to_field_elements(raw):
    """one field element per byte (plonky2x ByteVariable-style public I/O)"""
    import numpy as np
    return np.frombuffer(bytes(raw), dtype=np.uint8).astype(np.uint64)


# ---- VerifyCircuit semantics: NEAR transaction / receipt inclusion proof -------------------------
# Follows crates/protocol/src/lib.rs:118-151 (Protocol::inclusion_proof_verify, Proof::Basic) and
# crates/protocol/src/merkle_util.rs:7-30; the circuit restates the same checks in
# nearx/src/builder.rs:344-363 (Verify::verify) with nearx/src/merkle.rs:17-51.
import base64


def _sha(b):
    return hashlib.sha256(b).digest()


def combine_hash(a, b):
    return _sha(a + b)


def compute_root_from_path(path, item_hash):
    h = item_hash
    for uncle in path:
        u = b58decode32(uncle["hash"])
        h = combine_hash(u, h) if uncle["direction"] == "Left" else combine_hash(h, u)
    return h


def _borsh_partial_outcome(outcome):
    """borsh(PartialExecutionOutcome { receipt_ids, gas_burnt, tokens_burnt, executor_id, status })"""
    out = len(outcome["receipt_ids"]).to_bytes(4, "little")
    for r in outcome["receipt_ids"]:
        out += b58decode32(r)
    out += int(outcome["gas_burnt"]).to_bytes(8, "little")
    out += int(outcome["tokens_burnt"]).to_bytes(16, "little")
    ex = outcome["executor_id"].encode()
    out += len(ex).to_bytes(4, "little") + ex
    st = outcome["status"]
    if st == "Unknown" or (isinstance(st, dict) and "Unknown" in st):
        out += b"\x00"
    elif isinstance(st, dict) and "Failure" in st:
        out += b"\x01"
    elif isinstance(st, dict) and "SuccessValue" in st:
        v = base64.b64decode(st["SuccessValue"])
        out += b"\x02" + len(v).to_bytes(4, "little") + v
    elif isinstance(st, dict) and "SuccessReceiptId" in st:
        out += b"\x03" + b58decode32(st["SuccessReceiptId"])
    else:
        raise ValueError("unknown execution status %r" % (st,))
    return out


def outcome_hashes(outcome_proof):
    """ExecutionOutcomeWithIdView::to_hashes: [id, hash(partial outcome), hash(log)...]"""
    hs = [b58decode32(outcome_proof["id"]), _sha(_borsh_partial_outcome(outcome_proof["outcome"]))]
    hs += [_sha(log.encode()) for log in outcome_proof["outcome"]["logs"]]
    return hs


def inclusion_proof_verify(head_block_root, proof):
    """Proof::Basic: True iff the header hashes to the outcome's block, the outcome is under the header's
    outcome_root and the block is under head_block_root."""
    block_hash = header_hash(proof["block_header_lite"])
    block_hash_matches = block_hash == b58decode32(proof["outcome_proof"]["block_hash"])
    hs = outcome_hashes(proof["outcome_proof"])
    outcome_hash = _sha(len(hs).to_bytes(4, "little") + b"".join(hs))
    shard_root = compute_root_from_path(proof["outcome_proof"]["proof"], outcome_hash)
    outcome_root = compute_root_from_path(proof["outcome_root_proof"], _sha(shard_root))
    outcome_verified = outcome_root == b58decode32(proof["block_header_lite"]["inner_lite"]["outcome_root"])
    block_verified = compute_root_from_path(proof["block_proof"], block_hash) == head_block_root
    return block_hash_matches and outcome_verified and block_verified


def inclusion_proof_sha256_messages(proof):
    """Every SHA-256 preimage `Verify::verify` hashes for ONE transaction / receipt inclusion proof, in evaluation order
    (nearx/src/builder.rs:344-363 -> nearx/src/merkle.rs:17-51 `NearMerkleTree::get_root` + `HeaderVariable::hash`,
    each through `curta_sha256`): the proven block's header hash (3), the outcome's hashes, the outcome-root chain and -
    last, so that its digest is the SHA-256 STARK's public output - the block-proof chain, whose final digest must be the
    trusted head's `block_merkle_root`.  Returns (messages, dict(block_hash, outcome_root, block_root)) with the values
    the circuit compares (what `inclusion_proof_verify` checks)."""
    msgs = []

    def sha(b):
        msgs.append(bytes(b))
        return _sha(b)

    def root_from_path(path, item_hash):
        h = item_hash
        for uncle in path:
            u = b58decode32(uncle["hash"])
            h = sha(u + h) if uncle["direction"] == "Left" else sha(h + u)
        return h
    for m in header_hash_preimages(proof["block_header_lite"]):
        block_hash = sha(m)
    op = proof["outcome_proof"]
    hs = [b58decode32(op["id"]), sha(_borsh_partial_outcome(op["outcome"]))]
    hs += [sha(log.encode()) for log in op["outcome"]["logs"]]
    outcome_hash = sha(len(hs).to_bytes(4, "little") + b"".join(hs))
    shard_root = root_from_path(op["proof"], outcome_hash)
    outcome_root = root_from_path(proof["outcome_root_proof"], sha(shard_root))
    block_root = root_from_path(proof["block_proof"], block_hash)
    return msgs, {"block_hash": block_hash, "outcome_root": outcome_root, "block_root": block_root}
