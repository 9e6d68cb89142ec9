"""CPU tests: the C oracle against the golden vectors (upstream KATs + independent big-int model)."""
import numpy as np

from conftest import GEN, POW2_GEN, P, rand_field


def test_poseidon_kats(orc, golden):
    for kat in golden["poseidon_kat"]:
        out = orc.poseidon_permute(np.array([kat["in"]], dtype=np.uint64))[0]
        assert [int(x) for x in out] == kat["out"], kat["source"]


def test_poseidon_fast_schedule_equals_naive(orc, golden):
    """Poseidon::poseidon (fast partial rounds - what the oracle and the Rust prover run) == Poseidon::poseidon_naive (the
    schedule the known-answer vectors are stated for), on the KATs, on carry-edge states and on random states."""
    rng = np.random.default_rng(77)
    E = 0xFFFFFFFF
    edge = [0, 1, E, E + 1, P - 1, P - 2, (1 << 63), P - E, 2 * E, (1 << 32) + 1, P >> 1, 7]
    states = [k["in"] for k in golden["poseidon_kat"]] + [edge, edge[::-1]] + [list(rand_field(rng, 12)) for _ in range(200)]
    arr = np.array(states, dtype=np.uint64)
    fast, naive = orc.poseidon_permute(arr), orc.poseidon_permute_naive(arr)
    assert np.array_equal(fast, naive)
    for kat, out in zip(golden["poseidon_kat"], naive):
        assert [int(x) for x in out] == kat["out"]


def test_hash_or_noop_lengths(orc, golden):
    for case in golden["hash_or_noop"]:
        out = orc.hash_or_noop(np.array(case["in"], dtype=np.uint64))
        assert [int(x) for x in out] == case["out"], len(case["in"])


def test_two_to_one(orc, golden):
    c = golden["two_to_one"]
    assert [int(x) for x in orc.two_to_one(c["l"], c["r"])] == c["out"]


def test_merkle_levels_and_proofs(orc, golden):
    for key in ("merkle", "merkle_noop"):
        m = golden[key]
        leaves = np.array(m["leaves"], dtype=np.uint64)
        dig, cap = orc.merkle_build(leaves, m["cap_height"])
        flat = [w for lvl in m["levels"] for d in lvl for w in d]
        assert [int(x) for x in dig] == flat
        assert cap.tolist() == m["levels"][-1]
        n = leaves.shape[0]
        for idx in (0, 1, n // 2 + 1, n - 1):
            sib = orc.merkle_prove(dig, n, m["cap_height"], idx)
            assert orc.merkle_verify(leaves[idx], idx, sib, cap, m["cap_height"])
            bad = leaves[idx].copy()
            bad[0] ^= np.uint64(1)
            assert not orc.merkle_verify(bad, idx, sib, cap, m["cap_height"])


def test_commit_matches_horner_model(orc, golden):
    for key in ("commit", "commit_wide"):
        g = golden[key]
        res = orc.commit(np.array(g["values"], dtype=np.uint64), g["rate_bits"], g["cap_height"])
        assert res["coeffs"].tolist() == g["coeffs"]
        assert res["leaves"].tolist() == g["leaves"]
        assert res["cap"].tolist() == g["cap"]
        res2 = orc.commit(np.array(g["coeffs"], dtype=np.uint64), g["rate_bits"], g["cap_height"], from_coeffs=True)
        assert res2["cap"].tolist() == g["cap"]


def test_challenger(orc, golden):
    g = golden["challenger"]
    ch = orc.Challenger()
    outs = []
    for x in g["observe_then_2"]:
        ch.observe(x)
    outs += [ch.challenge(), ch.challenge()]
    for x in g["observe_then_10"]:
        ch.observe(x)
    outs += [ch.challenge() for _ in range(10)]
    assert outs == g["challenges"]


def test_fft_roundtrip_and_definition(orc):
    rng = np.random.default_rng(1)
    for log_n in (0, 1, 2, 5, 10):
        a = rand_field(rng, 1 << log_n)
        v = orc.fft(a)
        assert np.array_equal(orc.fft(v, inverse=True), a)
        shift = GEN
        vc = orc.fft(a, shift=shift)
        assert np.array_equal(orc.fft(vc, inverse=True, shift=shift), a)
        if log_n == 5:  # definition check: v[k] = sum a[j] w^(jk)
            w = pow(POW2_GEN, 1 << (32 - log_n), P)
            for k in (0, 1, 7, 31):
                acc = sum(int(a[j]) * pow(w, j * k, P) for j in range(32)) % P
                assert int(v[k]) == acc


def test_oracle_proof_regression_vectors(nlx, orc):
    """The oracle's own outputs for fixed workloads (tests/golden/oracle_proofs.json, generator beside it): a change
    of transcript order, wire format, workload generator or AIR assembler shows up here, on the CPU."""
    import importlib.util
    import json
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("gen_oracle_proofs", os.path.join(ROOT, "tests", "golden", "gen_oracle_proofs.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    with open(gen.golden_path()) as f:
        want = json.load(f)
    assert gen.cases() == want
