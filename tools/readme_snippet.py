import sys; sys.path.insert(0, ".")
import nlxpkg; nlx = nlxpkg.load()
ctx = nlx.Context(0)
syn = nlx.SyntheticCircuit(14, seed=1, pct_poseidon=25, pct_arithmetic=20, pct_u32=15)
cd = nlx.CircuitData.from_synthetic(ctx, syn)
proof = cd.prove(syn.wires, syn.public_inputs)
print("plonky2 proof", len(proof))
S = nlx.stark
air = S.Air(2, 3)
air.constraint_transition(air.next(0) - air.local(1))
air.constraint_transition(air.next(1) - air.local(0) - air.local(1))
prover = S.Stark(air, 10).build(ctx)
proof = prover.prove(*S.fibonacci_trace(10)[:1], [0, 1, 0])
print("stark proof", len(proof))
digest_proof, digest = nlx.sha256_air.Sha256Prover(ctx, 4).prove([b"abc", b"hello"])
print("sha256", len(digest_proof))
E = nlx.ed25519_air
pk = bytes.fromhex("d75a980182b10ab7d54bfed3c964073a0ee172f3daa62325af021a68f707511a")
sig = bytes.fromhex("e5564300c360ac729086e2cc806e828a84877f1eb8e5d974d873e065224901555fb8821590a33bacc61e39701cf9b46bd25bf5f0595bbe24655141438e7a100b")
msg = b""
slots = [E.slot_from_signature(pk, msg, sig)] * 255 + [E.inactive_slot()]
ed_proof = E.Ed25519Prover(ctx, 8).prove(slots)
print("ed25519", len(ed_proof))
import numpy as np
pts = nlx.bn254_g1_pack([(1, 2), (1, 2)])
ks = np.array([[3, 0, 0, 0], [4, 0, 0, 0]], dtype=np.uint64)
print("msm 7G", nlx.bn254_g1_unpack(nlx.bn254_msm_g1(ctx, pts, ks)))
