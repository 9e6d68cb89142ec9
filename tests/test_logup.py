"""Range-check lookups (LogUp in the quadratic extension, near-light-client_amd/logup.py): CPU part against the oracle's
restatement and verifier; GPU part: the round-1 columns computed on the GPU equal the oracle's bit for bit and the
GPU proof bytes equal the oracle's."""
import numpy as np
import pytest

from conftest import P


def range_air(nlx, n_values, bits, fused=True):
    """n_values looked-up columns, then the multiplicity column; round 1 = the lookup columns."""
    S, LU = nlx.stark, nlx.logup
    air = S.Air(n_values + 1 + LU.round_cols(n_values), 0, rounds=[(n_values + 1, 2), (LU.round_cols(n_values), 0)])
    rc = LU.RangeCheck(air, range(n_values), bits, n_values, n_values + 1, fused=fused)
    return air, rc


def make_trace(orc, n_values, bits, db, seed=1):
    rng = np.random.default_rng(seed)
    t0 = np.zeros((n_values + 1, 1 << db), dtype=np.uint64)
    t0[:n_values] = rng.integers(0, 1 << bits, (n_values, 1 << db), dtype=np.uint64)
    t0[n_values] = orc.logup_multiplicities(t0, range(n_values), bits)
    return t0


def rounds_fn(orc, t0, n_values, bits):
    def fn(rnd, chal):
        if rnd == 0:
            return t0
        return orc.logup_round(t0, range(n_values), bits, t0[n_values], chal[:2])
    return fn


@pytest.mark.parametrize("n_values,bits,db", [(5, 6, 7), (4, 8, 8), (1, 4, 5)])
def test_range_check_oracle(nlx, orc, n_values, bits, db):
    S = nlx.stark
    air, rc = range_air(nlx, n_values, bits)
    assert air.constraint_degree == (3 if n_values > 1 else 2) and rc.n_round_cols == 2 * ((n_values + 1) // 2) + 4
    st = S.Stark(air, db, S.StarkConfig(fri_num_queries=20))
    assert st.desc.period_bits == bits and st.desc.n_rounds == 2
    t0 = make_trace(orc, n_values, bits, db)
    assert int(t0[n_values].sum()) == n_values << db
    proof = orc.stark_prove_rounds(st.desc, rounds_fn(orc, t0, n_values, bits), [])
    assert orc.stark_verify(st.desc, proof) == 1
    # a cell outside the table: no multiplicity assignment makes the sums agree
    bad = t0.copy()
    bad[0, 3] = 1 << bits
    with pytest.raises(ValueError):
        orc.logup_multiplicities(bad, range(n_values), bits)
    assert orc.stark_verify(st.desc, orc.stark_prove_rounds(st.desc, rounds_fn(orc, bad, n_values, bits), [])) != 1
    bad[0, 3] = P - 1                                  # "-1"
    assert orc.stark_verify(st.desc, orc.stark_prove_rounds(st.desc, rounds_fn(orc, bad, n_values, bits), [])) != 1
    # a wrong multiplicity
    bad = t0.copy()
    bad[n_values, 2] = (int(bad[n_values, 2]) + 1) % P
    assert orc.stark_verify(st.desc, orc.stark_prove_rounds(st.desc, rounds_fn(orc, bad, n_values, bits), [])) != 1

    # a tampered helper / running sum cell
    def tampered(col, row):
        def fn(rnd, chal):
            out = rounds_fn(orc, t0, n_values, bits)(rnd, chal)
            if rnd == 1:
                out = out.copy()
                out[col, row] = (int(out[col, row]) + 1) % P
            return out
        return fn
    for col in (0, rc.n_round_cols - 4, rc.n_round_cols - 1):
        assert orc.stark_verify(st.desc, orc.stark_prove_rounds(st.desc, tampered(col, 9), [])) != 1


@pytest.mark.parametrize("n_values", [5, 4, 1])
def test_fused_instruction_equals_written_out_constraints(nlx, orc, n_values):
    """NLX_AIR_EMIT_LOGUP computes the same two constraint values as the DSL expressions: same proof bytes from a
    program a fraction of the size; and row by row in the reference interpreter."""
    from test_stark_cpu import run_program
    S = nlx.stark
    bits, db = 6, 7
    t0 = make_trace(orc, n_values, bits, db, seed=9)
    proofs, sizes = [], []
    for fused in (True, False):
        air, rc = range_air(nlx, n_values, bits, fused)
        st = S.Stark(air, db, S.StarkConfig(fri_num_queries=20))
        sizes.append(len(st.program))
        proofs.append(orc.stark_prove_rounds(st.desc, rounds_fn(orc, t0, n_values, bits), []))
        assert orc.stark_verify(st.desc, proofs[-1]) == 1
        alpha = (12345678901234567, 7654321987654321)
        full = np.concatenate([t0, orc.logup_round(t0, range(n_values), bits, t0[n_values], alpha)], axis=0)
        vals = run_program(st.program, full[:, 5], full[:, 6], list(alpha), periodic=[int(c[5 % len(c)]) for c in air._periodic], n_public=0)
        assert len(vals) == air.num_constraints and all(v == 0 for _, v in vals)
        full[0, 5] = (int(full[0, 5]) + 1) % (1 << bits)                   # another in-table value: the helper no longer matches
        vals = run_program(st.program, full[:, 5], full[:, 6], list(alpha), periodic=[int(c[5 % len(c)]) for c in air._periodic], n_public=0)
        assert any(v != 0 for _, v in vals)
    assert proofs[0] == proofs[1] and sizes[0] < sizes[1]


def test_mixed_periods_tile(nlx):
    S = nlx.stark
    air = S.Air(1, 0)
    a = air.periodic([1, 2])
    b = air.periodic([5, 6, 7, 8, 9, 10, 11, 12])
    c = air.periodic([3, 4, 3, 5])
    air.constraint(air.local(0) - a - b - c)
    assert air.period_bits == 3 and [list(map(int, x)) for x in air._periodic] == [[1, 2] * 4, list(range(5, 13)), [3, 4, 3, 5] * 2]
    with pytest.raises(ValueError):
        air.periodic(range(1 << 17))


@pytest.mark.gpu
@pytest.mark.parametrize("n_values,bits,db", [(5, 6, 7), (4, 8, 10), (7, 16, 16), (1, 4, 5)])
def test_gpu_logup_round_equals_oracle(nlx, ctx, orc, n_values, bits, db):
    import torch
    S = nlx.stark
    air, rc = range_air(nlx, n_values, bits)
    t0 = make_trace(orc, n_values, bits, db, seed=db)
    dev = torch.from_numpy(t0.view(np.int64)).to("cuda:%d" % ctx.device)
    dev[n_values].zero_()
    rc.multiplicities(ctx, dev)
    assert np.array_equal(dev.cpu().numpy().view(np.uint64), t0)
    alpha = (0x123456789abcdef1 % P, 0xfedcba9876543210 % P)
    want = orc.logup_round(t0, range(n_values), bits, t0[n_values], alpha)
    out = torch.empty((rc.n_round_cols, 1 << db), dtype=torch.int64, device=dev.device)
    rc.round1(ctx, dev, alpha, out)
    got = out.cpu().numpy().view(np.uint64)
    if not np.array_equal(got, want):
        bad = np.argwhere(got != want)[0]
        pytest.fail("round-1 column %d row %d differs from the oracle" % (bad[0], bad[1]))
    # a cell outside the table is reported, not silently counted
    dev[0, 3] = 1 << bits
    with pytest.raises(nlx.NlxError):
        rc.multiplicities(ctx, dev)
    dev[0, 3] = int(t0[0, 3])
    rc.multiplicities(ctx, dev)                         # (the failed call left the column half-counted)
    # whole proof: GPU prover with GPU-generated round-1 columns == oracle prover with the oracle's
    st = S.Stark(air, db, S.StarkConfig(fri_num_queries=20))
    pr = st.build(ctx)

    def gpu_rounds(rnd, chal):
        if rnd == 0:
            return dev
        return rc.round1(ctx, dev, chal[:2], out)
    proof = pr.prove_rounds(gpu_rounds, [])
    assert proof == orc.stark_prove_rounds(st.desc, rounds_fn(orc, t0, n_values, bits), [])
    assert orc.stark_verify(st.desc, proof) == 1
    pr.close()
