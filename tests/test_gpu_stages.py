"""The fine seam: a whole plonky2 proof assembled stage by stage through the public
stage-level entry points - nlx_commit_from_values, nlx_partial_products_and_zs, nlx_quotient_eval,
nlx_commit_eval_at, nlx_fri_prove, with the host challenger nlx_challenger_* - must give the same bytes as
nlx_prove (and therefore as the oracle)."""
import numpy as np
import pytest

from conftest import POW2_GEN, P

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("log_n,kw", [
    (8, dict(pct_poseidon=20, pct_arithmetic=30, pct_base_sum=5, pct_constant=5)),
    (10, dict(pct_poseidon=10, pct_arithmetic=10, pct_base_sum=5, pct_constant=5, pct_extension=10, pct_misc=20, pct_u32=30)),
    (5, dict(pct_poseidon=20, pct_arithmetic=30, pct_base_sum=5, pct_constant=5)),   # no FRI reduction round
])
def test_stagewise_proof_equals_whole_proof(nlx, ctx, orc, log_n, kw):
    pk = nlx.plonk
    syn = nlx.SyntheticCircuit(log_n, seed=300 + log_n, **kw)
    cd = nlx.CircuitData.from_synthetic(ctx, syn)
    want = cd.prove(syn.wires, syn.public_inputs)
    cfg = syn.config
    nc, npp = cfg.num_challenges, cfg.num_partial_products
    n_cs = syn.num_selectors + cfg.num_constants + cfg.num_routed_wires

    out = bytearray()
    ch = pk.Challenger()
    pih = pk.hash_no_pad(syn.public_inputs)
    # 1. wires
    cw = nlx.PolynomialBatch.from_values(ctx, syn.wires, cfg.rate_bits, cfg.cap_height)
    out += cw.cap.tobytes()
    ch.observe(cd.circuit_digest)
    ch.observe(pih)
    ch.observe(cw.cap)
    betas, gammas = ch.challenges(nc), ch.challenges(nc)
    b2, g2 = np.zeros(2, np.uint64), np.zeros(2, np.uint64)
    b2[:nc], g2[:nc] = betas, gammas
    # 2. Z / partial products
    cz = cd.partial_products_and_zs(syn.wires, b2, g2)
    out += cz.cap.tobytes()
    ch.observe(cz.cap)
    a2 = np.zeros(2, np.uint64)
    a2[:nc] = ch.challenges(nc)
    # 3. quotient
    cq = cd.quotient_eval(cw, cz, b2, g2, a2, pih)
    out += cq.cap.tobytes()
    ch.observe(cq.cap)
    zeta = ch.challenges(2)
    g = pow(POW2_GEN, 1 << (32 - log_n), P)   # primitive 2^log_n-th root of unity
    gzeta = np.array([int(zeta[0]) * g % P, int(zeta[1]) * g % P], dtype=np.uint64)
    # 4. openings
    cs = cd.constants_sigmas_batch()
    assert cs.n_cols == n_cs and np.array_equal(cs.cap, cd.constants_sigmas_cap)
    o_cs, o_w, o_zs, o_q = (b.eval_at(zeta) for b in (cs, cw, cz, cq))
    o_next = cz.eval_at(gzeta)[:nc]
    # OpeningSet wire order: constants, plonk_sigmas, wires, plonk_zs, plonk_zs_next, partial_products, quotient_polys
    out += o_cs.tobytes() + o_w.tobytes() + o_zs[:nc].tobytes() + o_next.tobytes() + o_zs[nc:].tobytes() + o_q.tobytes()
    openings_zeta = np.concatenate([o_cs, o_w, o_zs, o_q])
    ch.observe(openings_zeta)
    ch.observe(o_next)
    # 5. FRI
    fp = pk.FriParams(cfg.fri_arity_bits, cfg.fri_final_poly_bits, cfg.fri_pow_bits, cfg.fri_num_queries)
    out += pk.fri_prove(ctx, [cs, cw, cz, cq], [0, 0, nc, 0], zeta, openings_zeta, o_next, fp, ch)
    out += np.uint64(syn.public_inputs.size).tobytes() + syn.public_inputs.tobytes()   # write_usize(len), then the field vec
    assert len(out) == len(want)
    if bytes(out) != want:
        a, b = np.frombuffer(bytes(out), np.uint8), np.frombuffer(want, np.uint8)
        pytest.fail("stage-wise proof differs from nlx_prove, first at byte %d of %d" % (int(np.nonzero(a != b)[0][0]), len(want)))
    oc = orc.Circuit.from_synthetic(syn)
    assert oc.verify(bytes(out)) == 1
    oc.close()
    for b in (cw, cz, cq, cs):
        b.close()
    cd.close()


def test_stage_calls_reject_mismatched_inputs(nlx, ctx):
    pk = nlx.plonk
    syn = nlx.SyntheticCircuit(7, seed=5)
    cd = nlx.CircuitData.from_synthetic(ctx, syn)
    cfg = syn.config
    cw = nlx.PolynomialBatch.from_values(ctx, syn.wires, cfg.rate_bits, cfg.cap_height)
    z = np.zeros(2, np.uint64)
    with pytest.raises(nlx.NlxError):
        cd.quotient_eval(cw, cw, z, z, z, np.zeros(4, np.uint64))         # zs batch has the wrong width
    fp = pk.FriParams(cfg.fri_arity_bits, cfg.fri_final_poly_bits, cfg.fri_pow_bits, cfg.fri_num_queries)
    with pytest.raises(nlx.NlxError):
        pk.fri_prove(ctx, [cw], [136], z, np.zeros((135, 2), np.uint64), np.zeros((136, 2), np.uint64), fp, pk.Challenger())
    with pytest.raises(nlx.NlxError):
        pk.Challenger().observe(np.array([P], dtype=np.uint64))           # non-canonical element
    cw.close()
    cd.close()
