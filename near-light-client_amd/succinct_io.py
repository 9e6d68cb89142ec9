"""Request / response codecs either side of the prover: what reaches `build/sync prove input.json`
and what leaves it, so GPU proofs can travel through the reference's own plumbing.

* `ProofRequest` -> `input.json`: plonky2x `backend::function::ProofRequest` as nearx's test harness writes it
  (/root/reference/nearx/src/test_utils.rs:34-60; consumed by /root/reference/scripts/prove-circuit.sh:18 ->
  `Plonky2xFunction::entrypoint`, /root/reference/nearx/src/main.rs:11-25) and as the Succinct platform stores it
  (/root/reference/fixtures/sync_proof.json, verify_proof.json: `proof_request`): a serde enum tagged by `"type"`
  (`req_bytes` / `req_elements`), camelCase fields, `None` fields omitted, byte strings as 0x-hex.
* `ProofResult` -> `output.json`: the same crate's response - tag `res_bytes` / `res_elements`, `proof` and `data.output`.
  The reference holds no output.json; the field set follows plonky2x@4e539f2 (/root/reference/Cargo.lock:4977-4979) from
  memory [U]: `proof` here is 0x-hex of `ProofWithPublicInputs::to_bytes()` (the wrapped-proof form; upstream's unwrapped
  form serialises the proof struct through serde instead).
* Verify output: `C::VERIFY_AMT` x (32-byte id ‖ 1-byte result) in id order
  (/root/reference/nearx/src/verify.rs:94-98; decoded on chain by `decodePackedResults`,
  /root/reference/nearx/contract/src/interfaces/INearX.sol:111-142).
* Gateway call data: `sync(bytes32)` / `verify(bytes32,bytes)` as the relayer submits them
  (/root/reference/nearx/contract/src/NearX.sol:97-105,145-152); the fixtures' `callback_data` are golden vectors for these
  (selector = first four bytes of keccak256 of the signature).
"""
import base64
import json

from . import nearx_io

RESULT_LEN = 33  # bytes32 id ‖ bool


def _hex(b):
    return "0x" + bytes(b).hex()


def _unhex(s):
    if not isinstance(s, str) or not s.startswith("0x"):
        raise ValueError("expected a 0x-prefixed hex string")
    return bytes.fromhex(s[2:])


# ---- ProofRequest (input.json) -------------------------------------------------------------------------------------
def encode_proof_request(input_bytes=None, elements=None, release_id="todo", circuit_id="todo", parent_id=None, files=None):
    """ProofRequest::Bytes (evm I/O circuits: Sync, Verify) or ProofRequest::Elements -> the JSON text of input.json"""
    if (input_bytes is None) == (elements is None):
        raise ValueError("exactly one of input_bytes / elements")
    req = {"type": "req_bytes" if elements is None else "req_elements", "releaseId": release_id}
    if parent_id is not None:
        req["parentId"] = parent_id
    if files is not None:
        req["files"] = list(files)
    if elements is None:
        req["data"] = {"input": _hex(input_bytes)}
    else:
        req["data"] = {"circuitId": circuit_id, "input": [str(int(e)) for e in elements]}
    return json.dumps(req)


def decode_proof_request(text):
    """-> dict(kind 'bytes' | 'elements', input bytes | list[int], release_id, parent_id, files, circuit_id).  Accepts the
    bare request (input.json) and the platform's record that wraps it (`proof_request` of fixtures/*_proof.json)."""
    obj = json.loads(text) if isinstance(text, (str, bytes)) else text
    if "proof_request" in obj:
        obj = obj["proof_request"]
    kind = obj.get("type")
    if kind not in ("req_bytes", "req_elements"):
        raise ValueError("unsupported ProofRequest type %r" % kind)
    data = obj["data"]
    out = {"kind": kind[4:], "release_id": obj.get("releaseId"), "parent_id": obj.get("parentId"),
           "files": obj.get("files"), "circuit_id": data.get("circuitId")}
    out["input"] = _unhex(data["input"]) if kind == "req_bytes" else [int(e) for e in data["input"]]
    return out


# ---- ProofResult (output.json) -------------------------------------------------------------------------------------
def encode_proof_result(proof_bytes, output_bytes=None, elements=None):
    if (output_bytes is None) == (elements is None):
        raise ValueError("exactly one of output_bytes / elements")
    res = {"type": "res_bytes" if elements is None else "res_elements", "proof": _hex(proof_bytes)}
    res["data"] = {"output": _hex(output_bytes)} if elements is None else {"output": [str(int(e)) for e in elements]}
    return json.dumps(res)


def decode_proof_result(text):
    obj = json.loads(text) if isinstance(text, (str, bytes)) else text
    kind = obj.get("type")
    if kind not in ("res_bytes", "res_elements"):
        raise ValueError("unsupported ProofResult type %r" % kind)
    out = {"kind": kind[4:], "proof": _unhex(obj["proof"])}
    out["output"] = _unhex(obj["data"]["output"]) if kind == "res_bytes" else [int(e) for e in obj["data"]["output"]]
    return out


# ---- Verify output: [ProofVerificationResult; N] -------------------------------------------------------------------
def encode_verify_output(results):
    """results: [(id 32 bytes, bool)] in id order -> N x 33 bytes (evm_write CryptoHash, then the bool as one byte)"""
    out = b""
    for rid, ok in results:
        if len(rid) != 32:
            raise ValueError("an id is 32 bytes")
        out += bytes(rid) + (b"\x01" if ok else b"\x00")
    return out


def decode_verify_output(raw):
    if len(raw) % RESULT_LEN:
        raise ValueError("length must be 33 k")
    return [(raw[i:i + 32], raw[i + 32] != 0) for i in range(0, len(raw), RESULT_LEN)]


def default_verify_output(n):
    """the circuit's padding value: zero id, false (nearx/src/verify.rs:61-67)"""
    return [(b"\0" * 32, False)] * n


def merge_verify_outputs(left, right):
    """MergeProofHint::hint (nearx/src/verify.rs:151-183): the reduce step chains the two result arrays, keeps the entries
    whose id is neither all-zero nor all-0xFF, and resizes to N with the default - `Vec::resize`, so a surplus is cut."""
    n = len(left)
    if len(right) != n:
        raise ValueError("both arrays have VERIFY_AMT entries")
    dead = (b"\0" * 32, b"\xff" * 32)
    live = [(bytes(i), bool(r)) for i, r in list(left) + list(right) if bytes(i) not in dead]
    return (live + default_verify_output(n))[:n]


# ---- keccak256 (for the 4-byte function selectors) -----------------------------------------------------------------
_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B, 0x0000000080000001,
       0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
       0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080,
       0x000000000000800A, 0x800000008000000A, 0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]
_M = (1 << 64) - 1


def _rol(x, n):
    n %= 64
    return ((x << n) | (x >> (64 - n))) & _M if n else x


def _keccak_f(a):
    for rc in _RC:
        c = [a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4] for x in range(5)]
        d = [c[(x - 1) % 5] ^ _rol(c[(x + 1) % 5], 1) for x in range(5)]
        a = [[a[x][y] ^ d[x] for y in range(5)] for x in range(5)]
        b = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                b[y][(2 * x + 3 * y) % 5] = _rol(a[x][y], _ROT[x][y])
        a = [[b[x][y] ^ (~b[(x + 1) % 5][y] & _M & b[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        a[0][0] ^= rc
    return a


def keccak256(data):
    """Ethereum's keccak256 (the pre-standard padding 0x01 .. 0x80; hashlib.sha3_256 pads 0x06)"""
    rate = 136
    msg = bytearray(data) + b"\x01" + b"\0" * (-(len(data) + 1) % rate)
    msg[-1] |= 0x80
    a = [[0] * 5 for _ in range(5)]
    for off in range(0, len(msg), rate):
        for i in range(rate // 8):
            a[i % 5][i // 5] ^= int.from_bytes(msg[off + 8 * i: off + 8 * i + 8], "little")
        a = _keccak_f(a)
    return b"".join(a[i % 5][i // 5].to_bytes(8, "little") for i in range(4))


def selector(signature):
    return keccak256(signature.encode())[:4]


# ---- gateway call data ---------------------------------------------------------------------------------------------
def encode_sync_call(trusted_header):
    """NearX.sync(bytes32 trustedHeader)"""
    if len(trusted_header) != 32:
        raise ValueError("a header hash is 32 bytes")
    return selector("sync(bytes32)") + bytes(trusted_header)


def encode_verify_call(trusted_header, packed_ids):
    """NearX.verify(bytes32 trustedHeader, bytes _ids): head = header ‖ offset 0x40; tail = length ‖ ids padded to 32"""
    if len(trusted_header) != 32:
        raise ValueError("a header hash is 32 bytes")
    pad = b"\0" * (-len(packed_ids) % 32)
    return (selector("verify(bytes32,bytes)") + bytes(trusted_header) + (0x40).to_bytes(32, "big") +
            len(packed_ids).to_bytes(32, "big") + bytes(packed_ids) + pad)


def decode_verify_call(data):
    if data[:4] != selector("verify(bytes32,bytes)"):
        raise ValueError("not a verify(bytes32,bytes) call")
    header, off = data[4:36], int.from_bytes(data[36:68], "big")
    ln = int.from_bytes(data[4 + off: 36 + off], "big")
    return header, data[36 + off: 36 + off + ln]


def request_record_input(record):
    """the platform's record of a request (fixtures/*_proof.json: edges.requests[0]): (input bytes, callback data bytes)"""
    r = record["edges"]["requests"][0]
    return base64.b64decode(r["input"]), base64.b64decode(r["callback_data"])


def verify_request_to_ids(input_bytes):
    """a Verify request's input -> (trusted header hash, [(is_transaction, id, account)])"""
    return nearx_io.decode_verify_input(input_bytes)
