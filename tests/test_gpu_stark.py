"""GPU parity tests for the STARK path (nlx_stark_build / nlx_stark_prove through the C ABI): proof BYTES
must equal the CPU oracle's starky restatement on the same AIR, config, trace and public inputs, and the
oracle's verifier must accept them."""
import numpy as np
import pytest

from conftest import P
from test_stark_cpu import make_case

pytestmark = pytest.mark.gpu

CASES = [
    ("fib", 5, {}),                                   # no FRI reduction round, quotient factor 1 (one of two cosets)
    ("fib", 8, {}),
    ("fib", 12, {}),
    ("fib", 7, dict(num_challenges=1, fri_arity_bits=3, fri_num_queries=20, fri_pow_bits=8, cap_height=2)),
    ("wide16", 6, {}),
    ("wide16", 10, {}),
    ("wide16", 13, {}),                               # two NTT passes
    ("wide64", 12, {}),
    ("wide16", 8, dict(rate_bits=2, fri_arity_bits=2, fri_final_poly_bits=3, fri_num_queries=30)),
    ("deg4", 8, dict(rate_bits=2)),
    ("deg4", 9, dict(rate_bits=3, fri_num_queries=28)),
    ("periodic", 6, {}),
    ("periodic", 9, dict(rate_bits=2)),
    ("periodic", 12, {}),
    ("wide64", 14, {}),                               # 2^14 rows x 64 columns: the largest whole-STARK byte comparison
    # grouped Merkle leaves (nlx_stark_desc.leaf_group_cols): runs of G columns hashed on their own, then the runs' digests
    ("wide64", 9, dict(leaf_group_cols=8)),           # 8 runs of 8: one permutation per run
    ("wide64", 10, dict(leaf_group_cols=24)),         # runs 24 + 24 + 16: a short last run
    ("wide16", 8, dict(leaf_group_cols=12)),          # runs 12 + 4: the last run no longer than a digest is still hashed
    ("wide64", 11, dict(leaf_group_cols=64)),         # the row fits one run: whole-row leaves, but the digest names G
    ("wide320", 7, dict(leaf_group_cols="auto", openings_group="auto")),   # the opt-in variant's shape rule: > 256 columns on <= 2^16 LDE rows -> runs of 128, openings digest
    ("wide320", 9, {}),                               # the default on the same wide trace: whole-row leaves, starky's transcript
    # openings digest (nlx_stark_desc.openings_group): the transcript observes the hash of the openings' run digests
    ("wide64", 9, dict(openings_group=8)),            # 2 (2 x 64 + 2) = 260 values: 32 full runs and a run of 4, zero-padded
    ("wide16", 8, dict(openings_group=24, leaf_group_cols=12)),
    ("fib", 8, dict(openings_group=4096)),            # everything in one padded run
    ("wide320", 8, dict(leaf_group_cols="auto")),     # grouped leaves under starky's transcript (every opening observed)
    # batches (nlx_stark_desc.batch_cols): a round of more than B columns is committed as several PolynomialBatches - plain plonky2
    # batches, each with hash_or_noop leaves over its own row, its own cap and its own FRI oracle
    ("wide320", 8, dict(batch_cols=64)),              # five batches of 64
    ("wide64", 9, dict(batch_cols=24)),               # 24 + 24 + 16
    ("wide16", 8, dict(batch_cols=12)),               # 12 + 4: the last batch's row IS its digest (hash_or_noop)
    ("wide320", 13, dict(batch_cols=128)),            # 128 + 128 + 64 on 2^14 LDE rows: one lane per leaf
    ("wide64", 10, dict(batch_cols=64)),              # the round fits one batch: nothing changes but the statement digest
]


@pytest.mark.parametrize("kind,db,cfg", CASES)
def test_stark_proof_bytes_equal_oracle(nlx, ctx, orc, kind, db, cfg):
    S = nlx.stark
    air, t, pis = make_case(S, kind, db)
    st = S.Stark(air, db, S.StarkConfig(**cfg))
    want = orc.stark_prove(st.desc, t, pis)
    pr = st.build(ctx)
    got = pr.prove(t, pis)
    assert len(got) == len(want)
    if got != want:
        a, b = np.frombuffer(got, np.uint8), np.frombuffer(want, np.uint8)
        first = int(np.nonzero(a != b)[0][0])
        pytest.fail("STARK proof bytes differ from the oracle, first at byte %d of %d" % (first, len(want)))
    assert orc.stark_verify(st.desc, got) == 1
    assert pr.prove(t, pis) == got  # deterministic
    pr.close()


def test_stark_device_resident_trace(nlx, ctx, orc):
    import torch
    S = nlx.stark
    air, t, pis = make_case(S, "wide16", 9)
    st = S.Stark(air, 9)
    pr = st.build(ctx)
    d_t = torch.from_numpy(t.view(np.int64)).to("cuda:0")
    assert pr.prove(d_t, pis) == orc.stark_prove(st.desc, t, pis)
    pr.close()


def test_stark_large_trace_verifies(nlx, ctx, orc):
    """2^16 rows x 64 columns: too slow to compare against a full oracle prove in the test budget on a
    small host; the oracle's verifier (size-independent check) must accept the GPU proof."""
    S = nlx.stark
    air = S.wide_air(64, seed=2)
    t, pis = S.wide_trace(air, 16, seed=3)
    st = S.Stark(air, 16)
    pr = st.build(ctx)
    proof = pr.prove(t, pis)
    assert orc.stark_verify(st.desc, proof) == 1
    bad = bytearray(proof)
    bad[len(bad) // 2] ^= 4
    assert orc.stark_verify(st.desc, bytes(bad)) != 1
    names = [n for n, _ in pr.stage_times()]
    assert names[0] == "commit_trace" and "fri_queries" in names
    pr.close()


def test_stark_unsatisfied_witness_does_not_verify(nlx, ctx, orc):
    S = nlx.stark
    air, t, pis = make_case(S, "wide16", 8)
    st = S.Stark(air, 8)
    pr = st.build(ctx)
    t[5, 17] = (int(t[5, 17]) + 1) % P
    assert orc.stark_verify(st.desc, pr.prove(t, pis)) != 1
    pr.close()


def test_stark_build_rejects_bad_programs(nlx, ctx):
    S = nlx.stark
    air = S.fibonacci_air()

    def build_with(words, **over):
        st = S.Stark(air, 6)
        st.program = np.array(words, dtype=np.uint64)
        st.desc.n_words = len(words)
        st.desc.program = st.program.ctypes.data_as(type(st.desc.program))
        for k, v in over.items():
            setattr(st.desc, k, v)
        return st.build(ctx)

    good = [int(w) for w in air.compile()]
    build_with(good).close()
    with pytest.raises(nlx.NlxError):
        build_with([S.AIR_ADD | 1 << 8 | 2 << 24 | 3 << 40])          # reads unwritten registers
    with pytest.raises(nlx.NlxError):
        build_with([S.AIR_LOCAL | 0 << 8 | 2 << 24])                  # column 2 of a 2-column trace
    with pytest.raises(nlx.NlxError):
        build_with([S.AIR_PUBLIC | 0 << 8 | 3 << 24])                 # public input 3 of 3
    with pytest.raises(nlx.NlxError):
        build_with([S.AIR_LOCAL | 64 << 8])                           # register 64
    with pytest.raises(nlx.NlxError):
        build_with([S.AIR_CONST | 0 << 8])                            # CONST without immediate
    with pytest.raises(nlx.NlxError):
        build_with([20])                                              # unknown opcode
    with pytest.raises(nlx.NlxError):                                 # a register carried across a segment boundary
        build_with([S.AIR_LOCAL | 0 << 8, S.AIR_SEGMENT, S.AIR_EMIT | 0 << 24])
    with pytest.raises(nlx.NlxError):
        build_with([S.AIR_SEGMENT | 1 << 8])                          # operands on a boundary word
    with pytest.raises(nlx.NlxError):
        build_with([S.AIR_SEGMENT] * 256)                             # more than NLX_AIR_MAX_SEGMENTS segments
    build_with([S.AIR_SEGMENT] + good + [S.AIR_SEGMENT, S.AIR_SEGMENT]).close()   # empty segments are legal
    with pytest.raises(nlx.NlxError):
        build_with(good, quotient_degree_factor=4)                    # > 2^rate_bits
    with pytest.raises(nlx.NlxError):
        build_with(good, quotient_degree_factor=3)                    # not a power of two


@pytest.mark.parametrize("kind,db,seg,cfg", [("wide24", 6, 4, {}), ("wide96", 7, 16, {}), ("wide96", 10, 64, dict(num_challenges=1)),
                                             ("fib", 5, 1, {}), ("periodic", 9, 2, dict(rate_bits=2))])
def test_segmented_programs_bytes_equal_oracle(nlx, ctx, orc, kind, db, seg, cfg):
    """The same AIR cut into program segments (NLX_AIR_SEGMENT; the GPU runs them on different waves and adds the partial
    sums): GPU bytes equal the oracle's for the whole and for the cut program.  The two programs' proofs agree up to the
    first challenge only - the transcript opens with the AIR digest, which covers the program words; that the constraint
    VALUES are the same for every segmentation is tests/test_stark_cpu.py::test_program_segments_do_not_change_the_proof."""
    S = nlx.stark
    air, t, pis = make_case(S, kind, db)
    air.segment_nodes = 0
    whole = S.Stark(air, db, S.StarkConfig(**cfg))
    air.segment_nodes = seg
    cut = S.Stark(air, db, S.StarkConfig(**cfg))
    assert len(cut.program) > len(whole.program)
    firsts = []
    for st in (whole, cut):
        want = orc.stark_prove(st.desc, t, pis)
        pr = st.build(ctx)
        got = pr.prove(t, pis)
        assert got == want
        assert orc.stark_verify(st.desc, got) == 1
        firsts.append(got[: 32 << st.desc.cap_height])
        pr.close()
    assert firsts[0] == firsts[1]   # same trace, same first cap


def _random_air(S, rng, n_cols, n_pis, with_periodic):
    """A random constraint DAG over every VM feature.  The trace will NOT satisfy it - byte equality between
    the two provers does not need that (the quotient path is a fixed linear map of the constraint values)."""
    air = S.Air(n_cols, n_pis)
    leaves = [air.local(i) for i in range(n_cols)] + [air.next(i) for i in range(n_cols)]
    leaves += [air.public(i) for i in range(n_pis)]
    if with_periodic:
        leaves += [air.periodic([int(v) for v in rng.integers(0, P, 8, dtype=np.uint64)]) for _ in range(2)]
    if n_cols >= 8:
        leaves += [air.pack(0, 8), air.pack(n_cols - 5, 5, next_row=True), air.pack(1, 1)]
    pool = list(leaves)
    for _ in range(int(rng.integers(6, 14))):
        # one constraint: a few random binary ops, degree kept <= 3
        e = pool[int(rng.integers(0, len(pool)))]
        for _ in range(int(rng.integers(1, 6))):
            o = pool[int(rng.integers(0, len(pool)))]
            k = int(rng.integers(0, 8))
            if k == 6 and e.degree + o.degree <= 2:
                e = air.xor3(e, o, pool[int(rng.integers(0, len(leaves)))]) if e.degree + o.degree + 1 <= 3 else air.ch(e, o, o)
            elif k == 7 and e.degree + o.degree + 1 <= 3:
                e = air.maj(o, e, pool[int(rng.integers(0, len(leaves)))]) if (seed_bit := int(rng.integers(0, 2))) else air.ch(o, e, 3)
            elif k == 0 and e.degree + o.degree <= 3:
                e = e * o
            elif k == 1:
                e = e + o * (1 << int(rng.integers(1, 40)))
            elif k == 2:
                e = e - (1 << int(rng.integers(1, 33))) * o
            elif k == 3:
                e = e - o
            elif k == 4:
                e = e + int(rng.integers(0, P, dtype=np.uint64))
            else:
                e = int(rng.integers(0, P, dtype=np.uint64)) * e + o
        pool.append(e)
        kind = int(rng.integers(0, 5))
        if kind == 0 and e.degree <= 2:
            air.constraint_transition(e)
        elif kind == 1 and e.degree <= 2:
            air.constraint_first_row(e)
        elif kind == 2 and e.degree <= 2:
            air.constraint_last_row(e)
        elif kind == 3:
            air.constraint_boolean(int(rng.integers(0, n_cols)))
        else:
            air.constraint(e)
    return air


@pytest.mark.parametrize("seed", range(12))
def test_stark_random_programs_bytes_equal_oracle(nlx, ctx, orc, seed):
    S = nlx.stark
    rng = np.random.default_rng(1000 + seed)
    n_cols = int(rng.integers(2, 24))
    n_pis = int(rng.integers(0, 4))
    air = _random_air(S, rng, n_cols, n_pis, with_periodic=bool(seed & 1))
    if seed >= 4:
        air.segment_nodes = 3 + seed       # many small segments with different register needs: several launch groups
    db = int(rng.integers(5, 12))
    rate_bits = 1 if air.quotient_degree_factor() <= 2 else 2
    cfg = S.StarkConfig(rate_bits=rate_bits + (seed % 3 == 2), fri_num_queries=12, fri_pow_bits=6,
                        num_challenges=1 + (seed % 2), fri_arity_bits=2 + seed % 3)
    st = S.Stark(air, db, cfg)
    from conftest import rand_field
    t = rand_field(rng, (n_cols, 1 << db))
    pis = rand_field(rng, (n_pis,))
    want = orc.stark_prove(st.desc, t, pis)
    pr = st.build(ctx)
    got = pr.prove(t, pis)
    assert got == want, "seed %d: %d cols, 2^%d rows, %d words" % (seed, n_cols, db, st.desc.n_words)
    pr.close()


def test_stark_batch_prove(nlx, orc):
    """nlx_stark_batch_prove: three workers on distinct contexts, six jobs with different traces; every proof
    equals the oracle's for its trace."""
    import ctypes
    S = nlx.stark
    air = S.wide_air(16, seed=4)
    st = S.Stark(air, 9)
    ctxs = [nlx.Context(0) for _ in range(3)]
    prs = [st.build(c) for c in ctxs]
    traces = [S.wide_trace(air, 9, seed=50 + i) for i in range(6)]
    cap = nlx.lib.dll.nlx_stark_proof_max_bytes(prs[0].handle)
    bufs = [np.zeros(cap, dtype=np.uint8) for _ in traces]
    jobs = (nlx.ProveJob * len(traces))()
    for i, (t, pis) in enumerate(traces):
        jobs[i].wires = t.ctypes.data
        jobs[i].public_inputs = pis.ctypes.data
        jobs[i].proof_out = bufs[i].ctypes.data
        jobs[i].proof_cap = cap
    handles = (ctypes.c_void_p * 3)(*[p.handle for p in prs])
    assert nlx.lib.dll.nlx_stark_batch_prove(handles, 3, jobs, len(traces)) == 0
    for i, (t, pis) in enumerate(traces):
        assert jobs[i].status == 0
        assert bufs[i][:jobs[i].proof_len].tobytes() == orc.stark_prove(st.desc, t, pis), i
    # two workers on the same context are refused
    same = (ctypes.c_void_p * 2)(prs[0].handle, prs[0].handle)
    assert nlx.lib.dll.nlx_stark_batch_prove(same, 2, jobs, 1) != 0
    for p in prs:
        p.close()
    for c in ctxs:
        c.close()


def test_multi_round_logup_bytes_equal_oracle(nlx, ctx, orc):
    """Two commitment rounds with a verifier challenge in between (nlx_stark_prove_rounds): GPU proof bytes equal
    the oracle's, with the round columns handed over as host arrays and as device tensors."""
    import torch
    from test_stark_cpu import logup_air, logup_case, logup_rounds
    S = nlx.stark
    for db, cfg in ((7, dict(fri_num_queries=20)), (10, dict()), (6, dict(rate_bits=2, num_challenges=1, fri_arity_bits=2))):
        st = S.Stark(logup_air(S), db, S.StarkConfig(**cfg))
        v, t, m = logup_case(db)
        rounds = logup_rounds(v, t, m)
        want = orc.stark_prove_rounds(st.desc, rounds, [])
        pr = st.build(ctx)
        got = pr.prove_rounds(rounds)
        assert got == want, db
        assert orc.stark_verify(st.desc, got) == 1
        dev = pr.prove_rounds(lambda r, ch: torch.from_numpy(rounds(r, ch).view(np.int64)).to("cuda:0"))
        assert dev == want
        with pytest.raises(nlx.NlxError):
            pr.prove(np.zeros((6, 1 << db), dtype=np.uint64))          # single-round entry point refuses
        with pytest.raises(ValueError):
            pr.prove_rounds(lambda r, ch: np.zeros((2, 1 << db), dtype=np.uint64))   # wrong shape from the callback
        pr.close()
    # a value outside the table: the GPU proof is rejected by the verifier
    st = S.Stark(logup_air(S), 7, S.StarkConfig(fri_num_queries=20))
    pr = st.build(ctx)
    vb, tb, mb = logup_case(7, bad=True)
    assert orc.stark_verify(st.desc, pr.prove_rounds(logup_rounds(vb, tb, mb))) != 1
    pr.close()


def test_round_values_bytes_equal_oracle(nlx, ctx, orc):
    """A round that sends values (the total of a challenge-dependent accumulator): same proof bytes as the oracle, and
    the value the proof carries is the fingerprint of the committed column."""
    from test_stark_cpu import fingerprint_air, fingerprint_rounds
    S = nlx.stark
    for db in (6, 11):
        st = S.Stark(fingerprint_air(S), db, S.StarkConfig(fri_num_queries=20))
        v = np.random.default_rng(db).integers(0, P, 1 << db, dtype=np.uint64)
        pr = st.build(ctx)
        proof = pr.prove_rounds(fingerprint_rounds(v), [])
        assert proof == orc.stark_prove_rounds(st.desc, fingerprint_rounds(v), [])
        assert orc.stark_verify(st.desc, proof) == 1
        gamma, total = orc.stark_values(st.desc, proof)
        acc = 0
        for x in v:
            acc = (acc * gamma + int(x)) % P
        assert total == acc
        with pytest.raises(ValueError):
            pr.prove_rounds(lambda rnd, known: fingerprint_rounds(v)(rnd, known)[0] if rnd else np.array([v], dtype=np.uint64), [])
        pr.close()
