"""Extracts the request-side golden data of the reference's platform records (fixtures/sync_proof.json,
fixtures/verify_proof.json of near/near-light-client) into tests/golden/near/succinct_requests.json: the `proof_request`
object exactly as stored, and the relayed request's raw `input` / `callback_data` (base64 in the record, hex here).
Data only; run in the build container where /root/reference exists:  python tests/golden/gen_succinct_requests.py"""
import base64
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/fixtures"


def main():
    out = {"source": "near/near-light-client fixtures/{sync,verify}_proof.json: proof_request; edges.requests[0].{function_id,input,callback_data}"}
    for name in ("sync", "verify"):
        with open(os.path.join(REF, name + "_proof.json")) as f:
            rec = json.load(f)
        r = rec["edges"]["requests"][0]
        out[name] = {"proof_request": rec["proof_request"], "function_id": r["function_id"],
                     "input": "0x" + base64.b64decode(r["input"]).hex(),
                     "callback_data": "0x" + base64.b64decode(r["callback_data"]).hex()}
    path = os.path.join(HERE, "near", "succinct_requests.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
