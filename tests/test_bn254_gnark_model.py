"""CPU tests of the gnark-shaped PLONK model (oracle/bn254_py.py gnark_plonk_prove_model / gnark_plonk_verify_trapdoor): the
restated protocol is self-consistent - a blinded proof with public inputs verifies, tampering anywhere in the bytes and a wrong
public input are rejected, point compression round-trips.  (Parity with gnark-produced bytes is unpinned: Go, not in the reference.)"""
import os
import random
import sys

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "oracle"))


def test_gnark_shaped_model_is_self_consistent():
    import bn254_py as bn
    rng = random.Random(11)
    for log_n, n_pi in ((3, 0), (4, 2)):
        n, k1, k2 = 1 << log_n, 5, 25
        p = bn.plonk_witness(log_n, rng, k1, k2, 1, 1)
        p.pop("z")
        pis = [rng.randrange(bn.R) for _ in range(n_pi)]
        p["qk"] = [(a - (pis[i] if i < n_pi else 0)) % bn.R for i, a in enumerate(p["qk"])]
        tau = rng.randrange(1, bn.R)
        srs = bn.kzg_srs(tau, n + 3)
        blind = [rng.randrange(bn.R) for _ in range(9)]
        proof, data = bn.gnark_plonk_prove_model(p, srs, k1, k2, pis, blind)
        assert len(data) == 552 and bn.gnark_proof_bytes(bn.gnark_proof_from_bytes(data)) == data
        vk = {k: bn.msm_g1(bn.ntt(p[k], inverse=True), srs[:n]) for k in ("ql", "qr", "qm", "qo", "qk", "s1", "s2", "s3")}
        assert bn.gnark_plonk_verify_trapdoor(data, vk, n, tau, k1, k2, pis)
        # blinding changes the bytes, not the verdict; no blinding = the unblinded polynomials
        _, plain = bn.gnark_plonk_prove_model(p, srs, k1, k2, pis)
        assert plain != data and bn.gnark_plonk_verify_trapdoor(plain, vk, n, tau, k1, k2, pis)
        for pos in (5, 100, 230, 270, 330, 500, 551):      # a point's x, a claimed value, the shifted opening
            bad = bytearray(data)
            bad[pos] ^= 1
            try:
                ok = bn.gnark_plonk_verify_trapdoor(bytes(bad), vk, n, tau, k1, k2, pis)
            except AssertionError:                          # x is no longer on the curve / a value is not below r
                ok = False
            assert not ok, pos
        if n_pi:
            assert not bn.gnark_plonk_verify_trapdoor(data, vk, n, tau, k1, k2, [pis[0], (pis[1] + 1) % bn.R])
    for pt in (bn.G1, bn.g1_mul(12345, bn.G1), bn.g1_neg(bn.g1_mul(7, bn.G1)), None):
        assert bn.g1_decompress(bn.g1_compress(pt)) == pt
