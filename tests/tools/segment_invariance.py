"""Run on the GPU box: the Ed25519 proof of 2^8 synthetic slots must not depend on how the assembler cuts the AIR
program into segments (segments only split the evaluation; constraints keep their declaration order and powers of
alpha).  `python tests/tools/segment_invariance.py`."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import nlxpkg  # noqa: E402

nlx = nlxpkg.load()
E = nlx.ed25519_air
ctx = nlx.Context(0)
slots = E.synthetic_slots(256)
out = {}
for sn in (1024, 400, 150):
    pr = E.Ed25519Prover(ctx, 8, segment_nodes=sn)
    out[sn] = hashlib.sha256(bytes(pr.prove(slots))).hexdigest()
print(out)
print("invariant:", len(set(out.values())) == 1)
sys.exit(0 if len(set(out.values())) == 1 else 1)
