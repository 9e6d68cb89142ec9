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
import numpy as np  # noqa: E402
rs = np.random.RandomState(11)
bn_cols = rs.randint(0, 1 << 60, size=(2, 1 << 14, 4), dtype=np.int64).astype(np.uint64)           # values below r
g1 = np.tile(nlx.bn254_g1_pack([(1, 2), (1368015179489954701390400359078579693043519447331113978918064868415326638035,
                                         9918110051302171585080402603319702774565515993150576347155970296011118125764)]), (1 << 13, 1))
ks = rs.randint(0, 1 << 60, size=(1 << 14, 4), dtype=np.int64).astype(np.uint64)
syn_lk = nlx.SyntheticCircuit(11, seed=8, num_luts=2, lut_bits=9, num_lookups=700)                   # two plonky2 lookup tables
cd_lk = nlx.CircuitData.from_synthetic(ctx, syn_lk)
fr = lambda rows: np.concatenate([rs.randint(0, 1 << 60, size=(rows, 3), dtype=np.int64), rs.randint(0, 1 << 58, size=(rows, 1), dtype=np.int64)],
                                 axis=1).astype(np.uint64)                                           # fr.Element words below r
plonk_polys = {k: fr(1 << 10) for k in ("ql", "qr", "qm", "qo", "qk", "s1", "s2", "s3", "l", "r", "o", "z")}
kzg_poly, kzg_srs = fr(3000), nlx.bn254_g1_multiples(ctx, (1, 2), 2999)
# round 4: commitment rounds in batches, a STARK whose constraints run on a generated kernel (SHA-256 at 2^7 blocks = 2^9 rows), the blinded
# quotient chain, the natural-order transform without a reordering pass
p256b = nlx.sha256_air.Sha256Prover(ctx, 7, nlx.StarkConfig(batch_cols=512))
# the generated kernels exist in the library of record only (build.py WITH_AIRGEN: not in the generator-set-2021 variant, not
# under NLX_NO_AIRGEN / NLX_AIR_VM); elsewhere the same job runs on the interpreter
_generated_expected = os.environ.get("NLX_GL_GENERATOR_SET", "7") == "7" and os.environ.get("NLX_AIR_VM") != "1" and os.environ.get("NLX_NO_AIRGEN") != "1"
assert nlx.lib.dll.nlx_stark_quotient_kernel(p256b.prover.handle) == (1 if _generated_expected else 0)
p512b = nlx.sha512_air.Sha512Prover(ctx, 5, nlx.StarkConfig(batch_cols=512))
gl_cols = rs.randint(0, 1 << 62, size=(2, 1 << 20), dtype=np.int64).astype(np.uint64)
blind = [int(x) for x in rs.randint(1, 1 << 60, size=9)]
jobs = {"plonky2_2p13": lambda: cd.prove(syn.wires, syn.public_inputs), "sha256_2p6": lambda: p256.prove(msgs)[0],
        "sha256_2p7_batches_generated_kernel": lambda: p256b.prove(msgs * 3)[0], "sha512_2p5_batches": lambda: p512b.prove(msgs)[0],
        "bn254_plonk_quotient_2p10_blinded": lambda: nlx.bn254_plonk_quotient(ctx, plonk_polys, 5, 5, 25, 1234, 5678, 91011, blinding=blind)[0].tobytes(),
        "ntt_2p20_natural_order": lambda: nlx.ntt(ctx, gl_cols).tobytes(),
        "plonky2_2p11_lookup_tables": lambda: cd_lk.prove(syn_lk.wires, syn_lk.public_inputs),
        "bn254_plonk_quotient_2p10": lambda: nlx.bn254_plonk_quotient(ctx, plonk_polys, 5, 5, 25, 1234, 5678, 91011)[0].tobytes(),
        "bn254_kzg_open_3000": lambda: b"".join(x.tobytes() for x in nlx.bn254_kzg_open(ctx, kzg_poly, 777, srs=kzg_srs)),
        "sha512_2p5": lambda: p512.prove(msgs)[0], "ed25519_2p8": lambda: ped.prove(slots),
        "bn254_ntt_2p14_coset_dit": lambda: nlx.bn254_ntt(ctx, bn_cols, coset_shift=5, bitrev_in=True).tobytes(),
        "bn254_msm_2p14": lambda: nlx.bn254_msm_g1(ctx, g1, ks).tobytes()}
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
