"""Soak run on one GPU: every prover repeatedly on fixed inputs - proof bytes must not change from run to run
(determinism: SURVEY.md §8b) and the context's device memory must return to the same level (no leak).
    python tools/soak.py [iterations]"""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nlxpkg  # noqa: E402

nlx = nlxpkg.load()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = nlx.Context(0)
E = nlx.ed25519_air
syn = nlx.SyntheticCircuit(13, seed=3, pct_poseidon=20, pct_arithmetic=20, pct_u32=15, pct_extension=10, pct_misc=10)
cd = nlx.CircuitData.from_synthetic(ctx, syn)
p256, p512, ped = nlx.sha256_air.Sha256Prover(ctx, 6), nlx.sha512_air.Sha512Prover(ctx, 5), E.Ed25519Prover(ctx, 8)
msgs = [bytes([i]) * (i * 7 % 200) for i in range(20)]
slots = E.slots_to_words((E.synthetic_slots(16, seed=4) * 16)[:256])
jobs = {"plonky2_2p13": lambda: cd.prove(syn.wires, syn.public_inputs), "sha256_2p6": lambda: p256.prove(msgs)[0],
        "sha512_2p5": lambda: p512.prove(msgs)[0], "ed25519_2p8": lambda: ped.prove(slots)}
first, mem0 = {}, None
t0 = time.time()
for it in range(iters):
    for name, fn in jobs.items():
        h = hashlib.sha256(fn()).hexdigest()
        if first.setdefault(name, h) != h:
            raise SystemExit("proof of %s changed between runs (iteration %d)" % (name, it))
    reserved, in_use = ctx.memory()
    if it == 1:
        mem0 = in_use
    if it > 1 and in_use != mem0:
        raise SystemExit("device memory in use moved from %d to %d bytes (iteration %d)" % (mem0, in_use, it))
print(json.dumps({"iterations": iters, "seconds": round(time.time() - t0, 1), "proofs": first, "in_use_bytes": mem0,
                  "reserved_bytes": reserved}))
