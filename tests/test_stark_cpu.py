"""CPU tests of the STARK path: the AIR assembler (host logic of the product),
and the oracle's starky restatement - internal consistency (its verifier accepts what its prover emits,
rejects tampered proofs and unsatisfied witnesses).  "Parity unpinned": the reference holds no STARK
proof bytes (oracle/stark.h header)."""
import ctypes

import numpy as np
import pytest

from conftest import P


def run_program(words, local, nxt, pis, periodic=(), n_public=None):
    """Reference interpreter of the AIR register program (python ints): returns [(emit_op, value)]."""
    reg, out, pc = {}, [], 0
    while pc < len(words):
        w = int(words[pc])
        op, dst, a, b, sh = w & 0xFF, (w >> 8) & 0xFFFF, (w >> 24) & 0xFFFF, (w >> 40) & 0xFFFF, (w >> 56) & 0x3F
        if op == 0: reg[dst] = int(local[a])
        elif op == 1: reg[dst] = int(nxt[a])
        elif op == 2: reg[dst] = int(pis[a])
        elif op == 3:
            pc += 1
            reg[dst] = int(words[pc]) % P
        elif op == 4: reg[dst] = (reg[a] + (reg[b] << sh)) % P
        elif op == 5: reg[dst] = (reg[a] - (reg[b] << sh)) % P
        elif op == 6: reg[dst] = (reg[a] * reg[b]) % P
        elif op == 11: reg[dst] = int(periodic[a])
        elif op in (12, 13):
            src = local if op == 12 else nxt
            reg[dst] = sum(int(src[a + i]) << i for i in range(b)) % P
        elif op == 14:
            for i in range(max(b, 1)):
                out.append((10, int(local[a + i]) * (int(local[a + i]) - 1) % P))
        elif op == 15: pass  # LOADV: scheduling hint
        elif op == 19: reg = {}  # SEGMENT: nothing is carried across
        elif op == 21: reg[dst] = (reg[sh] + reg[a] * reg[b]) % P   # MAC
        elif op == 20:           # EMIT_LOGUP: v2 col in dst, v1 col in a, h cols b / b+1, challenge index in sh
            n_pis = len(pis) - 2 if n_public is None else n_public
            a0, a1 = int(pis[n_pis + sh]), int(pis[n_pis + sh + 1])
            h0, h1, v1 = int(local[b]), int(local[b + 1]), int(local[a])
            if dst == 0xFFFF:
                out.append((10, (h0 * (a0 + v1) + 7 * h1 * a1 - 1) % P))
                out.append((10, (h0 * a1 + h1 * (a0 + v1)) % P))
            else:
                v2 = int(local[dst])
                s_ = 2 * a0 + v1 + v2
                u0, u1 = (a0 + v1) * (a0 + v2) + 7 * a1 * a1, a1 * s_
                out.append((10, (h0 * u0 + 7 * h1 * u1 - s_) % P))
                out.append((10, (h0 * u1 + h1 * u0 - 2 * a1) % P))
        elif op in (16, 17, 18):
            x, y, z = reg[a], reg[b], reg[sh]
            if op == 17: reg[dst] = (z + x * (y - z)) % P
            else:
                s2 = (x + y - 2 * x * y) % P
                reg[dst] = (s2 + z - 2 * s2 * z) % P if op == 16 else (x * y + z * s2) % P
        else: out.append((op, reg[a]))
        pc += 1
    return out


def test_air_compile_semantics(nlx):
    S = nlx.stark
    rng = np.random.default_rng(3)
    air = S.Air(6, 2)
    l = [air.local(i) for i in range(6)]
    nx = [air.next(i) for i in range(6)]
    shared = l[0] * l[1] + 7               # shared sub-expression: computed once
    air.constraint_transition(nx[0] - shared)
    air.constraint(shared * l[2] - l[3])
    air.constraint_first_row(l[4] - air.public(1))
    air.constraint_last_row(5 - l[5] * l[5])
    air.constraint_transition((l[0] + l[1]) * (l[2] - 3) * l[3] - nx[1] + air.public(0))
    air.constraint(l[2])                   # a leaf as root
    assert air.num_constraints == 6
    assert air.constraint_degree == 4      # degree-3 product + the transition filter
    assert air.quotient_degree_factor() == 4
    words = air.compile()
    n_mul = sum(1 for w in words if int(w) & 0xFF == 6)
    n_mac = sum(1 for w in words if int(w) & 0xFF == 21)
    assert (n_mul, n_mac) == (4, 1)        # l0*l1 + 7 once (a multiply-add), *l2, l5*l5, two in the last product
    lo = [int(x) for x in rng.integers(0, P, 6, dtype=np.uint64)]
    ne = [int(x) for x in rng.integers(0, P, 6, dtype=np.uint64)]
    pi = [int(x) for x in rng.integers(0, P, 2, dtype=np.uint64)]
    got = run_program(words, lo, ne, pi)
    sh = (lo[0] * lo[1] + 7) % P
    want = [(7, (ne[0] - sh) % P), (10, (sh * lo[2] - lo[3]) % P), (8, (lo[4] - pi[1]) % P), (9, (5 - lo[5] * lo[5]) % P),
            (7, ((lo[0] + lo[1]) * (lo[2] - 3) * lo[3] - ne[1] + pi[0]) % P), (10, lo[2])]
    assert got == want


def test_air_register_pressure(nlx):
    S = nlx.stark
    air = S.Air(128, 0)
    # a long chain keeps few registers live; 100 independent products summed pairwise must still fit
    acc = air.local(0)
    for i in range(1, 128):
        acc = acc * air.local(i) + air.next(i)
    air.constraint(acc)
    words = air.compile()
    assert max((int(w) >> 8) & 0xFFFF for w in words if int(w) & 0xFF <= 6 or int(w) & 0xFF == 21) < S.AIR_MAX_RESIDENT_LEAVES + 4
    # 70 squares summed left to right: post-order evaluation frees operands as it goes
    wide = S.Air(200, 0)
    tot = wide.local(0) * wide.local(0)
    for i in range(1, 70):
        tot = tot + wide.local(i) * wide.local(i)
    wide.constraint(tot)
    assert max((int(w) >> 8) & 0xFFFF for w in wide.compile() if int(w) & 0xFF <= 6) < S.AIR_MAX_RESIDENT_LEAVES + 4
    # 70 values that are all still needed later do not fit 64 registers: refused, not miscompiled
    over = S.Air(200, 0)
    sq = [over.local(i) - over.local(i + 70) for i in range(70)]   # shared by both constraints below
    s1 = sq[0]
    for q in sq[1:]:
        s1 = s1 + q
    over.constraint(s1)
    p1 = sq[0]
    for q in sq[1:]:
        p1 = p1 * q
    over.constraint(p1)
    with pytest.raises(ValueError):
        over.compile()


def test_wide_trace_satisfies_air(nlx):
    S = nlx.stark
    air = S.wide_air(16, seed=5)
    t, pis = S.wide_trace(air, 6, seed=9)
    words = air.compile()
    n = t.shape[1]
    assert t.max() < P
    for i in (0, 1, n // 2, n - 2, n - 1):
        for op, v in run_program(words, t[:, i], t[:, (i + 1) % n], pis):
            if op == 7 and i == n - 1:
                continue  # transition constraints are off on the last row
            if op == 8 and i != 0:
                continue
            if op == 9 and i != n - 1:
                continue
            assert v == 0, (i, op)


def _tamper_offsets(n):
    return (5, 600, 1100, n // 3, n // 2, (2 * n) // 3, n - 100, n - 30, n - 3)


@pytest.mark.parametrize("kind,db,cfg", [
    ("fib", 5, {}),                                   # no FRI reduction round, quotient factor 1
    ("fib", 9, {}),
    ("fib", 7, dict(num_challenges=1, fri_arity_bits=3, fri_num_queries=20, fri_pow_bits=8, cap_height=2)),
    ("wide16", 6, {}),                                # quotient factor 2 = 2^rate_bits
    ("wide16", 10, {}),
    ("wide16", 8, dict(rate_bits=2, fri_arity_bits=2, fri_final_poly_bits=3, fri_num_queries=30)),  # factor 2 < 2^rate_bits
    ("deg4", 8, dict(rate_bits=2)),                   # quotient factor 3 -> 4
    ("deg4", 9, dict(rate_bits=3, fri_num_queries=28)),
    ("periodic", 6, {}),
    ("periodic", 9, dict(rate_bits=2)),
])
def test_stark_prove_verify_roundtrip(nlx, orc, kind, db, cfg):
    S = nlx.stark
    air, t, pis = make_case(S, kind, db)
    st = S.Stark(air, db, S.StarkConfig(**cfg))
    proof = orc.stark_prove(st.desc, t, pis)
    assert orc.stark_verify(st.desc, proof) == 1
    for off in _tamper_offsets(len(proof)):
        bad = bytearray(proof)
        bad[off] ^= 1
        assert orc.stark_verify(st.desc, bytes(bad)) != 1, off
    assert orc.stark_verify(st.desc, proof[:-8]) != 1
    assert orc.stark_verify(st.desc, proof + b"\0" * 8) != 1
    # an unsatisfied witness does not verify
    t2 = t.copy()
    t2[1, 3] = (int(t2[1, 3]) + 1) % P
    assert orc.stark_verify(st.desc, orc.stark_prove(st.desc, t2, pis)) != 1
    # wrong public input
    pis2 = pis.copy()
    pis2[0] = (int(pis2[0]) + 1) % P
    assert orc.stark_verify(st.desc, orc.stark_prove(st.desc, t, pis2)) != 1


def make_case(S, kind, db):
    if kind == "fib":
        air = S.fibonacci_air()
        t, pis = S.fibonacci_trace(db, 3, 5)
    elif kind.startswith("wide"):
        air = S.wide_air(int(kind[4:]), seed=db)
        t, pis = S.wide_trace(air, db, seed=db + 1)
    elif kind == "deg4":
        # x' = x^3 + y, y' = y + 1 : transition degree 3 (+1 for the filter) -> quotient factor 3 -> 4
        air = S.Air(2, 1)
        x, y = air.local(0), air.local(1)
        air.constraint_transition(air.next(0) - (x * x * x + y))
        air.constraint_transition(air.next(1) - (y + 1))
        air.constraint_first_row(x - air.public(0))
        n = 1 << db
        t = np.zeros((2, n), dtype=np.uint64)
        a, b = 11, 2
        for i in range(n):
            t[0, i], t[1, i] = a, b
            a, b = (a * a * a + b) % P, (b + 1) % P
        pis = np.array([11], dtype=np.uint64)
    elif kind == "periodic":
        # periodic columns: x' = x^2 + K[t mod 4] (transition); y counts 0..3 cyclically, reset by a periodic
        # selector - an all-rows constraint that also holds across the wrap from the last row to the first
        air = S.Air(2, 1)
        K = air.periodic([3, 5, 7, 11])
        sel = air.periodic([0, 0, 0, 1])
        x, y = air.local(0), air.local(1)
        air.constraint_transition(air.next(0) - (x * x + K))
        air.constraint((1 - sel) * (air.next(1) - y - 1) + sel * air.next(1))
        air.constraint_first_row(x - air.public(0))
        n = 1 << db
        t = np.zeros((2, n), dtype=np.uint64)
        a = 9
        for i in range(n):
            t[0, i], t[1, i] = a, i % 4
            a = (a * a + [3, 5, 7, 11][i % 4]) % P
        pis = np.array([9], dtype=np.uint64)
    else:
        raise KeyError(kind)
    return air, t, pis


def test_stark_abi_rejects_null(nlx):
    """No GPU here: only the argument checks that run before any device call."""
    d = nlx.lib.dll
    assert d.nlx_stark_build(None, None, None) != 0
    assert d.nlx_stark_proof_max_bytes(None) == 0
    n = ctypes.c_size_t()
    assert d.nlx_stark_prove(None, None, None, None, 0, ctypes.byref(n)) != 0
    k1 = np.ones(1, dtype=np.uint64)
    t = np.zeros((4, 8), dtype=np.uint64)
    pis = np.zeros(2, dtype=np.uint64)
    assert nlx.lib.synth_dll.nlx_synth_stark_trace(3, 3, 1, k1.ctypes.data, t.ctypes.data, pis.ctypes.data) != 0  # n_cols % 4
    assert nlx.lib.synth_dll.nlx_synth_stark_trace(4, 3, 1, k1.ctypes.data, t.ctypes.data, pis.ctypes.data) == 0


def test_air_fused_forms(nlx):
    """Multiplications by powers of two fold into the neighbouring ADD / SUB; pack() and
    constraint_boolean() are single instructions; all agree with the plain formulas."""
    S = nlx.stark
    rng = np.random.default_rng(11)
    air = S.Air(40, 0)
    x, y, z = air.local(32), air.local(33), air.next(34)
    air.constraint(x + y * 8 - 2 * (x * y) + 4 * z)          # three shifted forms, one MUL
    air.constraint(air.pack(0, 32) - air.pack(4, 5, next_row=True) * 3)
    air.constraint_boolean(7)
    air.constraint_transition(air.pack(0, 1) - air.local(0))
    words = air.compile()
    ops = [int(w) & 0xFF for w in words]
    assert ops.count(6) == 2 and ops.count(12) == 2 and ops.count(13) == 1 and ops.count(14) == 1
    assert air.constraint_degree == 2
    lo = [int(v) for v in rng.integers(0, P, 40, dtype=np.uint64)]
    ne = [int(v) for v in rng.integers(0, P, 40, dtype=np.uint64)]
    got = run_program(words, lo, ne, [])
    want = [(10, (lo[32] + 8 * lo[33] - 2 * lo[32] * lo[33] + 4 * ne[34]) % P),
            (10, (sum(lo[i] << i for i in range(32)) - 3 * sum(ne[4 + i] << i for i in range(5))) % P),
            (10, lo[7] * (lo[7] - 1) % P), (7, 0)]
    assert got == want


def test_air_ternary_forms(nlx):
    """xor3 / ch / maj are single instructions, exact on bits and equal to their polynomial on any field element"""
    S = nlx.stark
    rng = np.random.default_rng(13)
    air = S.Air(4, 0)
    x, y, z, w = (air.local(i) for i in range(4))
    air.constraint(air.xor3(x, y, z) - w)
    air.constraint(air.ch(x, y, z) + air.maj(x, y, z) * 5)
    air.constraint(air.xor3(air.xor3(x, y, z), w, 1))
    assert air.constraint_degree == 4   # xor3 of a degree-3 node with a column and a constant
    words = air.compile()
    ops = [int(v) & 0xFF for v in words]
    assert ops.count(16) == 3 and ops.count(17) == 1 and ops.count(18) == 1 and ops.count(6) + ops.count(21) == 1  # the "* 5" (a multiply-add)
    for bits in ((0, 0, 0, 0), (1, 0, 1, 1), (1, 1, 1, 0), (0, 1, 0, 1)):
        got = run_program(words, bits, bits, [])
        bx, by, bz, bw = bits
        assert got[0] == (10, ((bx ^ by ^ bz) - bw) % P)
        assert got[1] == (10, ((by if bx else bz) + 5 * ((bx & by) | (bz & (bx ^ by)))) % P)
        assert got[2] == (10, (bx ^ by ^ bz ^ bw ^ 1) % P)
    lo = [int(v) for v in rng.integers(0, P, 4, dtype=np.uint64)]
    got = run_program(words, lo, lo, [])
    s2 = (lo[0] + lo[1] - 2 * lo[0] * lo[1]) % P
    x3 = (s2 + lo[2] - 2 * s2 * lo[2]) % P
    assert got[0] == (10, (x3 - lo[3]) % P)
    assert got[1] == (10, (lo[2] + lo[0] * (lo[1] - lo[2]) + 5 * (lo[0] * lo[1] + lo[2] * s2)) % P)


def logup_air(S):
    """Two-round AIR: a log-derivative lookup of column v into table column t with multiplicities m.
    Round 0 commits (v, t, m) and draws one challenge alpha; round 1 commits h1 = 1/(alpha + v),
    h2 = m/(alpha + t) and the running sum z.  All-rows constraints (the sum telescopes to zero around the cycle)."""
    air = S.Air(6, 0, rounds=[(3, 1), (3, 0)])
    v, t, m, h1, h2, z = (air.local(i) for i in range(6))
    al = air.challenge(0)
    air.constraint(h1 * (al + v) - 1)
    air.constraint(h2 * (al + t) - m)
    air.constraint(air.next(5) - z - h1 + h2)
    air.constraint_first_row(z)
    return air


def logup_rounds(v, t, m):
    def fn(rnd, chal):
        n = len(v)
        if rnd == 0:
            return np.array([v, t, m], dtype=np.uint64)
        al = chal[0]
        h1 = [pow((al + int(x)) % P, P - 2, P) for x in v]
        h2 = [int(mm) * pow((al + int(x)) % P, P - 2, P) % P for x, mm in zip(t, m)]
        z, acc = [], 0
        for i in range(n):
            z.append(acc)
            acc = (acc + h1[i] - h2[i]) % P
        return np.array([h1, h2, z], dtype=np.uint64)
    return fn


def logup_case(db=7, bad=False, seed=3):
    rng = np.random.default_rng(seed)
    n = 1 << db
    t = np.arange(n, dtype=np.uint64) % np.uint64(64)           # table: 0..63 repeated
    v = rng.integers(0, 64, n, dtype=np.uint64)
    if bad:
        v[5] = 64                                              # not in the table
    m = np.zeros(n, dtype=np.uint64)
    for x in v:
        if int(x) < 64:
            m[int(x)] += 1                                      # multiplicities on the first copy of each table value
    return v, t, m


def test_multi_round_logup_oracle(nlx, orc):
    """Two commitment rounds with a verifier challenge in between (starkyx's round structure): the oracle's
    verifier accepts a correct lookup and rejects a value outside the table, a wrong multiplicity, tampering."""
    S = nlx.stark
    air = logup_air(S)
    assert air.constraint_degree == 2
    st = S.Stark(air, 7, S.StarkConfig(fri_num_queries=20))
    assert st.desc.n_rounds == 2 and list(st.desc.round_cols)[:2] == [3, 3] and st.desc.round_challenges[0] == 1
    v, t, m = logup_case()
    proof = orc.stark_prove_rounds(st.desc, logup_rounds(v, t, m), [])
    assert orc.stark_verify(st.desc, proof) == 1
    for off in (10, 700, 1300, len(proof) // 2, len(proof) - 40):
        bad = bytearray(proof)
        bad[off] ^= 1
        assert orc.stark_verify(st.desc, bytes(bad)) != 1, off
    vb, tb, mb = logup_case(bad=True)
    assert orc.stark_verify(st.desc, orc.stark_prove_rounds(st.desc, logup_rounds(vb, tb, mb), [])) != 1
    m2 = m.copy()
    m2[3] += 1
    assert orc.stark_verify(st.desc, orc.stark_prove_rounds(st.desc, logup_rounds(v, t, m2), [])) != 1
    # a classic single-round prove call refuses a multi-round descriptor
    with pytest.raises(RuntimeError):
        orc.stark_prove(st.desc, np.zeros((6, 128), dtype=np.uint64), [])


def test_program_segments_do_not_change_the_proof(nlx, orc):
    """NLX_AIR_SEGMENT boundaries (no register carried across; shared sub-expressions recomputed per segment) leave
    every constraint value unchanged.  (The proof bytes were equal too until the transcript began with the AIR digest,
    which covers the program words: now only the trace cap - everything before the first challenge - is.)"""
    S = nlx.stark
    for kind, db in (("wide24", 6), ("wide96", 7), ("fib", 5), ("periodic", 6)):
        air, t, pis = make_case(S, kind, db)
        proofs, sizes, rows = [], [], []
        for seg in (0, 1, 16, 1024):
            air.segment_nodes = seg
            st = S.Stark(air, db)
            sizes.append(len(st.program))
            proofs.append(orc.stark_prove(st.desc, t, pis))
            assert orc.stark_verify(st.desc, proofs[-1]) == 1
            row = run_program(st.program, t[:, 3], t[:, 4], pis, periodic=[int(c[3 % len(c)]) for c in air._periodic])
            assert len(row) == air.num_constraints
            # a row pair that is NOT consecutive: non-zero constraint values, equal for every segmentation
            rows.append(run_program(st.program, t[:, 3], t[:, 7], pis, periodic=[int(c[3 % len(c)]) for c in air._periodic]))
        assert all(r == rows[0] for r in rows) and any(v != 0 for _, v in rows[0])
        assert all(p[:512] == proofs[0][:512] for p in proofs) and sizes[1] > sizes[0]


def fingerprint_air(S):
    """Round 0: a column v; challenge gamma; round 1: the Horner accumulator acc(i) = acc(i-1) gamma + v(i), whose last
    value is a ROUND VALUE - sent by the prover, bound by a last-row constraint, compared by whoever relies on the proof
    with the fingerprint of the data they believe v to be."""
    air = S.Air(2, 0, rounds=[(1, 1), (1, 0)], round_values=[0, 1])
    v, acc = air.local(0), air.local(1)
    gamma, total = air.challenge(0), air.round_value(1, 0)
    air.constraint_first_row(acc - v)
    air.constraint_transition(air.next(1) - (acc * gamma + air.next(0)))
    air.constraint_last_row(acc - total)
    return air


def fingerprint_rounds(v, lie=0):
    def fn(rnd, known):
        if rnd == 0:
            return np.array([v], dtype=np.uint64)
        gamma, acc, out = known[0], 0, []
        for x in v:
            acc = (acc * gamma + int(x)) % P
            out.append(acc)
        return np.array([out], dtype=np.uint64), [(acc + lie) % P]
    return fn


def test_round_values_oracle(nlx, orc):
    S = nlx.stark
    air = fingerprint_air(S)
    st = S.Stark(air, 6, S.StarkConfig(fri_num_queries=20))
    assert list(st.desc.round_values) == [0, 1, 0] and ctypes_sizeof(st.desc) == 128   # + batch_cols (round 4)
    rng = np.random.default_rng(5)
    v = rng.integers(0, P, 64, dtype=np.uint64)
    proof = orc.stark_prove_rounds(st.desc, fingerprint_rounds(v), [])
    assert orc.stark_verify(st.desc, proof) == 1
    gamma, total = orc.stark_values(st.desc, proof)       # (challenge of round 0, value of round 1) in values-array order
    acc = 0
    for x in v:
        acc = (acc * gamma + int(x)) % P
    assert total == acc                                    # the relying party's check: the value IS the fingerprint of v
    # a prover that sends another value cannot satisfy the last-row constraint
    assert orc.stark_verify(st.desc, orc.stark_prove_rounds(st.desc, fingerprint_rounds(v, lie=1), [])) != 1
    # the value is part of the transcript: changing it in the proof changes every later challenge
    bad = bytearray(proof)
    bad[-3] ^= 1
    assert orc.stark_verify(st.desc, bytes(bad)) != 1


def ctypes_sizeof(x):
    import ctypes
    return ctypes.sizeof(x)


def test_transcript_binds_public_inputs_and_air(nlx, orc):
    """The transcript opens with the AIR digest and the public inputs (round 1 observed only caps and round values, so
    a public input could be chosen after alpha and zeta were known).  Sharpest case: a public input NO constraint reads -
    the constraint check cannot notice a change, only the transcript can."""
    S = nlx.stark
    air = S.Air(2, 4)  # FibonacciStark + a fourth public input that the program never loads
    air.constraint_first_row(air.local(0) - air.public(0))
    air.constraint_first_row(air.local(1) - air.public(1))
    air.constraint_last_row(air.local(1) - air.public(2))
    air.constraint_transition(air.next(0) - air.local(1))
    air.constraint_transition(air.next(1) - air.local(0) - air.local(1))
    t, pis3 = S.fibonacci_trace(6, 3, 5)
    pis = np.concatenate([pis3, np.array([77], dtype=np.uint64)])
    st = S.Stark(air, 6)
    proof = orc.stark_prove(st.desc, t, pis)
    assert orc.stark_verify(st.desc, proof) == 1
    # tail = u64 count | public inputs: rewrite the unread one after the fact
    assert int.from_bytes(proof[-40:-32], "little") == 4 and int.from_bytes(proof[-8:], "little") == 77
    forged = proof[:-8] + (78).to_bytes(8, "little")
    assert orc.stark_verify(st.desc, forged) != 1
    # the same trace proved for the other value is a different transcript from the first cap's challenges on
    other = orc.stark_prove(st.desc, t, np.concatenate([pis3, np.array([78], dtype=np.uint64)]))
    assert orc.stark_verify(st.desc, other) == 1 and other[:512] == proof[:512] and other[512:1024] != proof[512:1024]
    # and the statement digest covers the program: an AIR differing in one constant rejects the proof of the other
    air2 = S.Air(2, 4)
    air2.constraint_first_row(air2.local(0) - air2.public(0))
    air2.constraint_first_row(air2.local(1) - air2.public(1))
    air2.constraint_last_row(air2.local(1) - air2.public(2))
    air2.constraint_transition(air2.next(0) - air2.local(1))
    air2.constraint_transition((air2.next(1) - air2.local(0) - air2.local(1)) * 2)
    st2 = S.Stark(air2, 6)
    assert orc.stark_air_digest(st.desc) != orc.stark_air_digest(st2.desc)
    assert orc.stark_verify(st2.desc, proof) != 1
    assert orc.stark_verify(st2.desc, orc.stark_prove(st2.desc, t, pis)) == 1


def test_grouped_leaves_are_what_the_header_says(nlx, orc):
    """leaf_group_cols: a row of more than G elements is hash_no_pad(hash_no_pad(run 0) || hash_no_pad(run 1) || ...), shorter rows
    are plonky2's hash_or_noop; the STARK oracle proves and verifies with it, a verifier told another G rejects, and the
    host rule switches it on only for wide short traces"""
    import ctypes
    d = orc.dll()
    d.orc_leaf_digest.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p]
    d.orc_hash_no_pad.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]

    def hash_no_pad(x):
        x = np.ascontiguousarray(x, dtype=np.uint64)
        h = np.zeros(4, dtype=np.uint64)
        d.orc_hash_no_pad(x.ctypes.data, x.size, h.ctypes.data)
        return h
    rng = np.random.default_rng(5)
    row = rng.integers(0, P, 37, dtype=np.uint64)
    out = np.zeros(4, dtype=np.uint64)
    for G in (8, 12, 36):
        d.orc_leaf_digest(row.ctypes.data, row.size, G, out.ctypes.data)
        runs = [hash_no_pad(row[i:i + G]) for i in range(0, row.size, G)]   # a run of ONE element (G = 12, 36) is hashed too
        assert np.array_equal(out, hash_no_pad(np.concatenate(runs)))
    for G in (0, 37, 64):
        d.orc_leaf_digest(row.ctypes.data, row.size, G, out.ctypes.data)
        assert np.array_equal(out, hash_no_pad(row))
    S = nlx.stark
    air, t, pis = make_case(S, "wide16", 6)
    st = S.Stark(air, 6, S.StarkConfig(leaf_group_cols=8))
    plain = S.Stark(air, 6)
    assert st.desc.leaf_group_cols == 8 and plain.desc.leaf_group_cols == 0
    proof = orc.stark_prove(st.desc, t, pis)
    assert orc.stark_verify(st.desc, proof) == 1
    assert orc.stark_verify(plain.desc, proof) != 1 and proof != orc.stark_prove(plain.desc, t, pis)
    assert len(proof) == len(orc.stark_prove(plain.desc, t, pis))       # same layout: opened rows and sibling paths
    # the DEFAULT is starky's tree whatever the shape; the shape rule is the opt-in variant's
    cfg = S.StarkConfig()
    assert cfg.variant == "starky" and cfg.leaf_group_for(9, 4745) == 0 and cfg.openings_group_for(4745) == 0
    cfg = S.StarkConfig.grouped()
    assert cfg.variant == "grouped-leaves"
    assert cfg.leaf_group_for(9, 4745) == 128 and cfg.leaf_group_for(15, 1488) == 128 and cfg.leaf_group_for(16, 1488) == 0 and cfg.leaf_group_for(9, 200) == 0


def test_openings_digest_is_what_the_header_says(nlx, orc):
    """openings_group: the transcript observes hash_no_pad(run digests) of the opened values (local ++ quotient ++ next, zero-padded
    to whole runs) instead of every value; the proof's bytes keep their layout, a verifier told another G (or none) rejects, and
    the host rule switches it on above 256 columns"""
    S = nlx.stark
    air, t, pis = make_case(S, "wide16", 6)
    plain, dig, other = S.Stark(air, 6), S.Stark(air, 6, S.StarkConfig(openings_group=8)), S.Stark(air, 6, S.StarkConfig(openings_group=16))
    assert plain.desc.openings_group == 0 and dig.desc.openings_group == 8
    proof = orc.stark_prove(dig.desc, t, pis)
    assert orc.stark_verify(dig.desc, proof) == 1
    assert orc.stark_verify(plain.desc, proof) != 1 and orc.stark_verify(other.desc, proof) != 1
    base = orc.stark_prove(plain.desc, t, pis)
    assert len(proof) == len(base) and proof != base
    assert not np.array_equal(orc.stark_air_digest(plain.desc), orc.stark_air_digest(dig.desc))
    wide = S.Stark(S.wide_air(320, seed=3), 5)
    assert wide.desc.openings_group == 0 and wide.desc.leaf_group_cols == 0        # the reference's protocol by default
    wide = S.Stark(S.wide_air(320, seed=3), 5, S.StarkConfig.grouped())
    assert wide.desc.openings_group == 64 and wide.desc.leaf_group_cols == 128


def test_batches_are_ordinary_polynomial_batches(nlx, orc):
    """batch_cols: a commitment round of more than B columns is ceil(cols / B) PolynomialBatches - each one's cap is the cap of
    orc.commit over ITS columns alone (hash_or_noop leaves, nothing new), the proof carries the caps in batch order where the
    single cap was, every query opens one more row + path per extra batch, the oracle verifier accepts it and rejects it under
    any other batching, and a statement without batches keeps its digest and its bytes"""
    S = nlx.stark
    air, t, pis = make_case(S, "wide16", 6)
    plain, b12, b8 = S.Stark(air, 6), S.Stark(air, 6, S.StarkConfig(batch_cols=12)), S.Stark(air, 6, S.StarkConfig(batch_cols=8))
    assert plain.desc.batch_cols == 0 and b12.desc.batch_cols == 12
    base = orc.stark_prove(plain.desc, t, pis)
    proof = orc.stark_prove(b12.desc, t, pis)
    assert orc.stark_verify(b12.desc, proof) == 1
    assert orc.stark_verify(plain.desc, proof) != 1 and orc.stark_verify(b8.desc, proof) != 1
    assert not np.array_equal(orc.stark_air_digest(plain.desc), orc.stark_air_digest(b12.desc))
    capb = 32 << 4
    for k, (lo, hi) in enumerate(((0, 12), (12, 16))):          # 12 + 4 columns; the second batch's leaves are the rows themselves
        want = orc.commit(t[lo:hi], 1, 4)["cap"]
        got = np.frombuffer(proof[k * capb:(k + 1) * capb], dtype=np.uint64).reshape(16, 4)
        assert np.array_equal(got, want), "cap of batch %d is PolynomialBatch::from_values of its columns" % k
    log_l = 7
    assert len(proof) == len(base) + capb + 84 * (1 + 32 * (log_l - 4))   # one more cap, one more path per query
    tampered = bytearray(proof)
    tampered[capb + 5] ^= 1                                                # the second batch's cap
    assert orc.stark_verify(b12.desc, bytes(tampered)) != 1
    # a two-round AIR (SHA-256's binding round): every round is batched on its own - round 0 in four batches, round 1 (two columns) in one
    SA = nlx.sha256_air
    blocks, first, digest = SA.blocks_for_messages([b"abc", b"batches"], 2)
    tr, _ = SA.reference_trace(blocks, first)
    st = S.Stark(SA.sha256_air(), 4, S.StarkConfig(batch_cols=512, fri_num_queries=10))
    proof = orc.stark_prove_rounds(st.desc, SA.cpu_rounds(blocks, first, tr), digest)
    assert orc.stark_verify(st.desc, proof) == 1
    assert orc.stark_verify(S.Stark(SA.sha256_air(), 4, S.StarkConfig(fri_num_queries=10)).desc, proof) != 1
    for k in range(4):
        want = orc.commit(tr[512 * k:512 * (k + 1)], 1, 4)["cap"]
        assert np.array_equal(np.frombuffer(proof[k * capb:(k + 1) * capb], dtype=np.uint64).reshape(16, 4), want)
