"""Range-check lookups (LogUp in the quadratic extension, near-light-client_amd/logup.py): CPU part against the oracle's
restatement and verifier; GPU part: the round-1 columns computed on the GPU equal the oracle's bit for bit and the
GPU proof bytes equal the oracle's."""
import numpy as np
import pytest

from conftest import P


def range_air(nlx, n_values, bits, fused=True, table_cols=1):
    """n_values looked-up columns, then the multiplicity column(s); round 1 = the lookup columns."""
    S, LU = nlx.stark, nlx.logup
    n0, n1 = n_values + table_cols, LU.round_cols(n_values, table_cols)
    air = S.Air(n0 + n1, 0, rounds=[(n0, 2), (n1, 0)])
    rc = LU.RangeCheck(air, range(n_values), bits, n_values, n0, fused=fused, table_cols=table_cols)
    return air, rc


def make_trace(orc, n_values, bits, db, seed=1, table_cols=1):
    rng = np.random.default_rng(seed)
    t0 = np.zeros((n_values + table_cols, 1 << db), dtype=np.uint64)
    t0[:n_values] = rng.integers(0, 1 << bits, (n_values, 1 << db), dtype=np.uint64)
    t0[n_values:] = orc.logup_multiplicities(t0, range(n_values), bits, table_cols)
    return t0


def rounds_fn(orc, t0, n_values, bits, table_cols=1):
    def fn(rnd, chal):
        if rnd == 0:
            return t0
        return orc.logup_round(t0, range(n_values), bits, t0[n_values:], chal[:2], table_cols)
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
    """NLX_AIR_EMIT_LOGUP computes the same two constraint values as the DSL expressions, row by row in the reference
    interpreter, from a program a fraction of the size.  (Until round 2 the two proofs were byte-identical; the transcript
    now opens with the AIR digest, which covers the program, so they agree up to the first challenge only.)"""
    from test_stark_cpu import run_program
    S = nlx.stark
    bits, db = 6, 7
    t0 = make_trace(orc, n_values, bits, db, seed=9)
    proofs, sizes, rows = [], [], []
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
        rows.append(run_program(st.program, full[:, 6], full[:, 9], list(alpha), periodic=[int(c[6 % len(c)]) for c in air._periodic], n_public=0))
        full[0, 5] = (int(full[0, 5]) + 1) % (1 << bits)                   # another in-table value: the helper no longer matches
        vals = run_program(st.program, full[:, 5], full[:, 6], list(alpha), periodic=[int(c[5 % len(c)]) for c in air._periodic], n_public=0)
        assert any(v != 0 for _, v in vals)
    # same constraint values (kind and value, in order) on a row pair that does NOT satisfy them, same round-0 cap
    assert rows[0] == rows[1] and any(v != 0 for _, v in rows[0])
    assert proofs[0][:512] == proofs[1][:512] and sizes[0] < sizes[1]


@pytest.mark.parametrize("bits,db,table_cols", [(8, 7, 2), (8, 6, 4), (10, 8, 4)])
def test_table_spread_over_several_columns_oracle(nlx, orc, bits, db, table_cols):
    """A table longer than the trace: 2^bits entries in table_cols periodic columns of 2^bits / table_cols rows each."""
    S = nlx.stark
    n_values = 3
    air, rc = range_air(nlx, n_values, bits, table_cols=table_cols)
    st = S.Stark(air, db, S.StarkConfig(fri_num_queries=20))
    assert st.desc.period_bits == bits - table_cols.bit_length() + 1 and rc.n_round_cols == 4 + 2 * table_cols + 2
    t0 = make_trace(orc, n_values, bits, db, seed=3, table_cols=table_cols)
    assert int(t0[n_values:].sum()) == n_values << db and int(t0[:n_values].max()) >= (1 << db)     # values beyond the trace length
    proof = orc.stark_prove_rounds(st.desc, rounds_fn(orc, t0, n_values, bits, table_cols), [])
    assert orc.stark_verify(st.desc, proof) == 1
    bad = t0.copy()
    bad[1, 5] = 1 << bits
    assert orc.stark_verify(st.desc, orc.stark_prove_rounds(st.desc, rounds_fn(orc, bad, n_values, bits, table_cols), [])) != 1
    bad = t0.copy()                                      # a count moved to the same row of another table column
    c = int(np.argmax(t0[n_values:, 2] > 0))
    bad[n_values + c, 2] -= 1
    bad[n_values + (c + 1) % table_cols, 2] += 1
    assert orc.stark_verify(st.desc, orc.stark_prove_rounds(st.desc, rounds_fn(orc, bad, n_values, bits, table_cols), [])) != 1


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
@pytest.mark.parametrize("n_values,bits,db,table_cols", [(5, 6, 7, 1), (4, 8, 10, 1), (7, 16, 16, 1), (1, 4, 5, 1), (3, 8, 6, 4),
                                                         (6, 16, 14, 4)])
def test_gpu_logup_round_equals_oracle(nlx, ctx, orc, n_values, bits, db, table_cols):
    import torch
    S = nlx.stark
    air, rc = range_air(nlx, n_values, bits, table_cols=table_cols)
    t0 = make_trace(orc, n_values, bits, db, seed=db, table_cols=table_cols)
    dev = torch.from_numpy(t0.view(np.int64)).to("cuda:%d" % ctx.device)
    dev[n_values:].zero_()
    rc.multiplicities(ctx, dev)
    assert np.array_equal(dev.cpu().numpy().view(np.uint64), t0)
    alpha = (0x123456789abcdef1 % P, 0xfedcba9876543210 % P)
    want = orc.logup_round(t0, range(n_values), bits, t0[n_values:], alpha, table_cols)
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
    assert proof == orc.stark_prove_rounds(st.desc, rounds_fn(orc, t0, n_values, bits, table_cols), [])
    assert orc.stark_verify(st.desc, proof) == 1
    pr.close()


@pytest.mark.gpu
def test_gpu_witness_calls_reject_bad_arguments(nlx, ctx):
    """The lookup / trace-generation entry points return an error code (never abort) for out-of-range arguments."""
    import ctypes
    import torch
    dll = nlx.lib.dll
    t = torch.zeros((4, 64), dtype=torch.int64, device="cuda:%d" % ctx.device)
    cols = np.array([0, 1], dtype=np.uint32)
    al = np.array([1, 2], dtype=np.uint64)
    out = torch.zeros((6, 64), dtype=torch.int64, device=t.device)

    def mult(**kw):
        a = dict(trace=t.data_ptr(), n_cols=4, log_n=6, cols=cols.ctypes.data, n=2, bits=4, tcols=1, mcol=3)
        a.update(kw)
        return dll.nlx_logup_multiplicities(ctx.handle, a["trace"], a["n_cols"], a["log_n"], a["cols"], a["n"], a["bits"], a["tcols"],
                                            a["mcol"])
    assert mult() == 0
    assert mult(bits=7) < 0 and mult(bits=0) < 0 and mult(bits=17) < 0          # table larger than the trace / out of range
    assert mult(tcols=3) < 0 and mult(tcols=0) < 0 and mult(tcols=2) < 0         # not a power of two / columns run past the trace
    assert mult(mcol=4) < 0 and mult(mcol=1) < 0                                # multiplicity column out of range / looked up
    assert mult(trace=None) < 0 and mult(n=0) < 0
    bad_cols = np.array([0, 9], dtype=np.uint32)
    assert mult(cols=bad_cols.ctypes.data) < 0
    assert b"out of range" in dll.nlx_last_error(ctx.handle)
    assert dll.nlx_logup_round(ctx.handle, t.data_ptr(), 4, 6, cols.ctypes.data, 2, 4, 1, 3, al.ctypes.data, out.data_ptr()) == 0
    assert dll.nlx_logup_round(ctx.handle, t.data_ptr(), 4, 6, cols.ctypes.data, 2, 4, 1, 3, None, out.data_ptr()) < 0
    assert dll.nlx_logup_round(ctx.handle, t.data_ptr(), 4, 6, cols.ctypes.data, 2, 4, 1, 3, al.ctypes.data, None) < 0
    assert dll.nlx_logup_round_cols(2, 1) == 6 and dll.nlx_logup_round_cols(3, 1) == 8 and dll.nlx_logup_round_cols(3, 4) == 14
    words = np.zeros((256, 32), dtype=np.uint64)
    assert dll.nlx_ed25519_trace(ctx.handle, words.ctypes.data, 17, t.data_ptr()) < 0
    assert dll.nlx_ed25519_trace(ctx.handle, None, 8, t.data_ptr()) < 0
    assert dll.nlx_fp25519_chip_trace(ctx.handle, None, words.ctypes.data, 16, t.data_ptr()) < 0
    assert dll.nlx_fp25519_chip_trace(ctx.handle, words.ctypes.data, words.ctypes.data, 3, t.data_ptr()) < 0
    assert dll.nlx_sha512_trace(ctx.handle, words.ctypes.data, words.ctypes.data, 19, t.data_ptr(), None) < 0
    _ = ctypes
