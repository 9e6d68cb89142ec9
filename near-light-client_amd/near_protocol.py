"""Witness-side semantics of SyncCircuit without RPC: the NEAR light-client sync checks
that nearx re-implements in-circuit (nearx/src/builder.rs Sync::sync :265-308) and the reference implements
natively in crates/protocol/src/lib.rs:68-118 (Protocol::sync) - restated here so that the fixtures alone
(tests/golden/near/*.json = the reference's fixtures/) produce the circuit's witness values: the new head,
its hash, the approval message, the per-validator signature checks and the stake totals.

What each piece follows:
  * reconstruct_approval_message   crates/protocol/src/lib.rs:181-206  (ApprovalInner::Endorsement borsh | height+2 LE)
  * validate_signatures            :250-276   (zip(signatures, epoch_bps), Ed25519 over the approval message)
  * ensure_stake_is_sufficient     :299-311   (approved > total / 3 * 2, integer division first)
  * ensure_next_bps_is_valid       :313-329   (sha256(borsh(Vec<ValidatorStakeView>)) == inner_lite.next_bp_hash)
  * ensure_not_already_verified / ensure_epoch_is_current_or_next / ensure_if_next_epoch_contains_next_bps :208-248
Pinned by the reference's own literals (tests/test_near_protocol.py): the stake totals
(512915271547861520119028536348929, 345140782903867823005444871054881) of crates/protocol/src/lib.rs:466-497
and the epoch-boundary walk of :364-405.

Ed25519 verification is the textbook RFC 8032 check in pure Python (a host-side checker of ~100 signatures per
header; the in-circuit version is curta's Ed25519 STARK, not rebuilt here).  No GPU work happens in this module.
"""
import hashlib

from .nearx_io import _B58, b58decode32, header_hash

# ---------------------------------------------------------------------------------------------
# Ed25519 (RFC 8032 §5.1.7 verification, cofactorless equation as ed25519-dalek's `verify`)
# ---------------------------------------------------------------------------------------------
_P = 2 ** 255 - 19
_L = 2 ** 252 + 27742317777372353535851937790883648493
_D = (-121665 * pow(121666, _P - 2, _P)) % _P
_I = pow(2, (_P - 1) // 4, _P)


def _recover_x(y, sign):
    if y >= _P:
        return None
    x2 = (y * y - 1) * pow(_D * y * y + 1, _P - 2, _P) % _P
    if x2 == 0:
        return None if sign else 0
    x = pow(x2, (_P + 3) // 8, _P)
    if (x * x - x2) % _P != 0:
        x = x * _I % _P
    if (x * x - x2) % _P != 0:
        return None
    if (x & 1) != sign:
        x = _P - x
    return x


_GY = 4 * pow(5, _P - 2, _P) % _P
_G = (_recover_x(_GY, 0), _GY, 1, _recover_x(_GY, 0) * _GY % _P)


def _add(p, q):
    a = (p[1] - p[0]) * (q[1] - q[0]) % _P
    b = (p[1] + p[0]) * (q[1] + q[0]) % _P
    c = 2 * p[3] * q[3] * _D % _P
    d = 2 * p[2] * q[2] % _P
    e, f, g, h = b - a, d - c, d + c, b + a
    return (e * f % _P, g * h % _P, f * g % _P, e * h % _P)


def _mul(s, p):
    q = (0, 1, 1, 0)
    while s > 0:
        if s & 1:
            q = _add(q, p)
        p = _add(p, p)
        s >>= 1
    return q


def _decode_point(raw):
    y = int.from_bytes(raw, "little")
    sign = y >> 255
    y &= (1 << 255) - 1
    x = _recover_x(y, sign)
    if x is None:
        return None
    return (x, y, 1, x * y % _P)


def _equal(p, q):
    return (p[0] * q[2] - q[0] * p[2]) % _P == 0 and (p[1] * q[2] - q[1] * p[2]) % _P == 0


def ed25519_verify(public_key, message, signature):
    if len(public_key) != 32 or len(signature) != 64:
        return False
    a = _decode_point(public_key)
    r = _decode_point(signature[:32])
    s = int.from_bytes(signature[32:], "little")
    if a is None or r is None or s >= _L:
        return False
    h = int.from_bytes(hashlib.sha512(signature[:32] + public_key + message).digest(), "little") % _L
    return _equal(_mul(s, _G), _add(r, _mul(h, a)))


# ---------------------------------------------------------------------------------------------
# NEAR encodings
# ---------------------------------------------------------------------------------------------
def b58decode(s):
    n = 0
    for ch in s:
        n = n * 58 + _B58.index(ch)
    pad = len(s) - len(s.lstrip("1"))
    return b"\0" * pad + n.to_bytes((n.bit_length() + 7) // 8, "big")


def _key_bytes(tagged, length):
    kind, _, body = tagged.partition(":")
    if kind != "ed25519":
        raise ValueError("only ed25519 keys / signatures appear in the light-client views")
    raw = b58decode(body)
    raw = b"\0" * (length - len(raw)) + raw
    if len(raw) != length:
        raise ValueError("bad key / signature length")
    return raw


def validator_stake_borsh(v):
    """borsh(ValidatorStakeView::V1 { account_id, public_key, stake }): variant u8 | string | key type u8 | key | u128."""
    acct = v["account_id"].encode()
    return (b"\0" + len(acct).to_bytes(4, "little") + acct + b"\0" + _key_bytes(v["public_key"], 32)
            + int(v["stake"]).to_bytes(16, "little"))


def next_bps_hash(next_bps):
    """CryptoHash::hash_borsh(Vec<ValidatorStakeView>)."""
    return hashlib.sha256(len(next_bps).to_bytes(4, "little") + b"".join(validator_stake_borsh(v) for v in next_bps)).digest()


def reconstruct_approval_message(block_view):
    """ApprovalInner::Endorsement(sha256(next_block_inner_hash || hash(new head))) in borsh, then (height + 2) as
    u64 LE: 41 bytes (Protocol::reconstruct_approval_message; nearx: builder.rs reconstruct_approval_message)."""
    next_block_hash = hashlib.sha256(b58decode32(block_view["next_block_inner_hash"]) + header_hash(block_view)).digest()
    return b"\0" + next_block_hash + (int(block_view["inner_lite"]["height"]) + 2).to_bytes(8, "little")


def validate_signatures(signatures, epoch_bps, approval_message):
    """(total_stake, approved_stake, per-validator flags) over zip(signatures, epoch_bps)."""
    total = approved = 0
    flags = []
    for sig, v in zip(signatures, epoch_bps):
        stake = int(v["stake"])
        total += stake
        ok = sig is not None and ed25519_verify(_key_bytes(v["public_key"], 32), approval_message, _key_bytes(sig, 64))
        if ok:
            approved += stake
        flags.append(ok)
    return total, approved, flags


def ensure_stake_is_sufficient(total, approved):
    return approved > total // 3 * 2


class SyncError(ValueError):
    pass


def sync(head_view, epoch_bps, next_block):
    """Protocol::sync.  head_view / next_block: LightClientBlockView dicts (only the header fields of the head
    are used).  Returns {new_head_hash, approval_message, total, approved, signed, next_bps}."""
    hl, nl = head_view["inner_lite"], next_block["inner_lite"]
    if int(nl["height"]) <= int(hl["height"]):
        raise SyncError("BlockAlreadyVerified")
    if nl["epoch_id"] not in (hl["epoch_id"], hl["next_epoch_id"]):
        raise SyncError("BlockNotCurrentOrNextEpoch")
    if nl["epoch_id"] == hl["next_epoch_id"] and next_block.get("next_bps") is None:
        raise SyncError("NextBpsInvalid")
    msg = reconstruct_approval_message(next_block)
    total, approved, flags = validate_signatures(next_block["approvals_after_next"], epoch_bps, msg)
    if not ensure_stake_is_sufficient(total, approved):
        raise SyncError("NotEnoughApprovedStake")
    nb = next_block.get("next_bps")
    if nb is not None and next_bps_hash(nb) != b58decode32(nl["next_bp_hash"]):
        raise SyncError("NextBpsInvalid")
    return {"new_head_hash": header_hash(next_block), "approval_message": msg, "total": total, "approved": approved,
            "signed": flags, "next_bps": nb}


def sync_sha256_messages(next_block):
    """The SHA-256 preimages one Sync step hashes (what nearx hands to curta_sha256): the new head's three
    header-hash preimages, the next-block-hash combine of the approval message, and - last, so that its
    digest is the SHA-256 STARK's public output - the borsh of next_bps, whose hash must equal
    inner_lite.next_bp_hash (nearx/src/builder.rs ensure_next_bps_is_valid)."""
    from .nearx_io import header_hash_preimages
    msgs = header_hash_preimages(next_block)
    msgs.append(b58decode32(next_block["next_block_inner_hash"]) + header_hash(next_block))
    nb = next_block.get("next_bps")
    if nb is not None:
        msgs.append(len(nb).to_bytes(4, "little") + b"".join(validator_stake_borsh(v) for v in nb))
    return msgs


def approval_statement(epoch_bps, next_block):
    """The Ed25519 side of one Sync step as nearx lays it out (nearx/src/builder.rs:116-164 `validate_signatures<LEN>`): ONE
    SLOT PER VALIDATOR of the epoch, in order - active with its signature over the approval message where the block carries
    one, inactive (`is_active` false, checks off) where it does not - plus the SHA-512 preimages R || A || M of the signed
    ones (what curta_eddsa hashes).  Returns dict(message, slots, signed (validator indices), sig_msgs)."""
    from . import ed25519_air as E
    msg = reconstruct_approval_message(next_block)
    slots, signed, sig_msgs = [], [], []
    for i, (sig, bp) in enumerate(zip(next_block["approvals_after_next"], epoch_bps)):
        if sig is None:
            slots.append(E.inactive_slot())
            continue
        pk, raw = _key_bytes(bp["public_key"], 32), _key_bytes(sig, 64)
        sl = E.slot_from_signature(pk, msg, raw)
        if sl is None:
            raise ValueError("validator %d: malformed key or signature" % i)
        slots.append(sl)
        signed.append(i)
        sig_msgs.append(raw[:32] + pk + msg)
    return dict(message=msg, slots=slots, signed=signed, sig_msgs=sig_msgs)


def slots_with_digests(statement, digest_words):
    """The relying party's view of the Ed25519 slots when the SHA-512 digests come from the SHA-512 STARK's side: signed
    validator k's D is the little-endian integer of digest k's 64 bytes (eight big-endian 64-bit words, as sha512_air's
    block outputs).  With these slots `ed25519_air.fingerprint` must reproduce the Ed25519 proof's round value - the tie
    between the two proofs: the same numbers are absorbed by both fingerprints."""
    slots = list(statement["slots"])
    if len(digest_words) != len(statement["signed"]):
        raise ValueError("one digest per signed validator")
    for i, words in zip(statement["signed"], digest_words):
        d = int.from_bytes(b"".join(int(w).to_bytes(8, "big") for w in words), "little")
        slots[i] = slots[i][:5] + (d,) + slots[i][6:]
    return slots
