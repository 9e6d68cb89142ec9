"""CPU tests: the oracle prover/verifier on synthetic nearx-shaped circuits (internal consistency:
the plonky2 verifier restatement accepts what the prover restatement emits, rejects tampering),
plus the host-side pieces of the product that need no GPU (workload generator, fast Poseidon tables)."""
import ctypes

import numpy as np
import pytest

from conftest import POW2_GEN, P


@pytest.mark.parametrize("log_n,kw", [
    (5, dict(pct_poseidon=20, pct_arithmetic=30, pct_base_sum=5, pct_constant=5)),   # no FRI reduction round
    (6, dict(pct_poseidon=0, pct_arithmetic=50, pct_base_sum=10, pct_constant=10)),  # one selector polynomial
    (8, dict(pct_poseidon=25, pct_arithmetic=25, pct_base_sum=5, pct_constant=5)),
    (9, dict(pct_poseidon=10, pct_arithmetic=0, pct_base_sum=0, pct_constant=5)),
    (8, dict(pct_poseidon=15, pct_arithmetic=20, pct_base_sum=5, pct_constant=5, pct_extension=30)),  # 10 gates, 3 selectors
    (7, dict(pct_poseidon=0, pct_arithmetic=0, pct_base_sum=0, pct_constant=5, pct_extension=60)),
    (8, dict(pct_poseidon=10, pct_arithmetic=10, pct_base_sum=5, pct_constant=5, pct_extension=20, pct_misc=30)),  # 14 gates
    (8, dict(pct_poseidon=10, pct_arithmetic=10, pct_base_sum=5, pct_constant=5, pct_extension=10, pct_misc=20, pct_u32=30)),  # all 19 gates
    (7, dict(pct_poseidon=0, pct_arithmetic=0, pct_base_sum=0, pct_constant=5, pct_u32=80)),
])
def test_prove_verify_roundtrip(nlx, orc, log_n, kw):
    syn = nlx.SyntheticCircuit(log_n, seed=log_n, **kw)
    circ = orc.Circuit.from_synthetic(syn)
    proof = circ.prove(syn.wires, syn.public_inputs)
    assert len(proof) > 0
    assert circ.verify(proof) == 1
    # tamper with one byte in each region: caps, openings, query data, final poly / pow witness
    for off in (7, 3 * 512 + 100, len(proof) // 2, len(proof) - 60):
        bad = bytearray(proof)
        bad[off] ^= 1
        assert circ.verify(bytes(bad)) < 1, off
    # truncated / empty
    assert circ.verify(proof[:-8]) < 1
    assert circ.verify(b"") < 1
    circ.close()


def test_unsatisfied_witness_is_rejected(nlx, orc):
    syn = nlx.SyntheticCircuit(7, seed=3)
    circ = orc.Circuit.from_synthetic(syn)
    w = syn.wires.copy()
    w[0, 0] = (int(w[0, 0]) + 1) % P  # row 0 is the PublicInputGate: wire 0 must equal pi_hash[0]
    proof = circ.prove(w, syn.public_inputs)
    assert circ.verify(proof) < 1
    # a broken copy constraint (same gate equations, different routed value) is caught by the permutation argument
    proof2 = circ.prove(syn.wires, syn.public_inputs)
    assert circ.verify(proof2) == 1
    circ.close()


def test_wrong_public_inputs_rejected(nlx, orc):
    syn = nlx.SyntheticCircuit(6, seed=5)
    circ = orc.Circuit.from_synthetic(syn)
    proof = bytearray(circ.prove(syn.wires, syn.public_inputs))
    proof[-8] ^= 1  # last public input
    assert circ.verify(bytes(proof)) < 1
    circ.close()


def test_selector_groups_follow_plonky2(nlx):
    syn = nlx.SyntheticCircuit(5, seed=1)  # all six gates: degrees 0,1,1,2,3,7 -> groups [0,5) and [5,6)
    assert syn.num_gates == 6 and syn.num_selectors == 2
    g = list(syn.gates)
    assert [x.kind for x in g] == [0, 1, 2, 4, 3, 5]
    assert [(x.group_start, x.group_end, x.selector_index) for x in g] == [(0, 5, 0)] * 5 + [(5, 6, 1)]
    syn1 = nlx.SyntheticCircuit(5, seed=1, pct_poseidon=0)  # max degree 3 + 5 gates - 1 <= 8: one selector
    assert syn1.num_selectors == 1
    # selector column holds the gate index on its rows and UNUSED (2^32-1) elsewhere
    sel0, sel1 = syn.constants[0], syn.constants[1]
    assert set(np.unique(sel1).tolist()) <= {5, 0xFFFFFFFF}
    assert ((sel0 == 0xFFFFFFFF) == (sel1 == 5)).all()


def test_sigma_is_a_permutation_respecting_copies(nlx):
    syn = nlx.SyntheticCircuit(7, seed=9)
    n = 1 << 7
    w = pow(POW2_GEN, 1 << (32 - 7), P)
    sub = [pow(w, i, P) for i in range(n)]
    ids = {}
    for j in range(80):
        for i in range(n):
            ids[int(syn.k_is[j]) * sub[i] % P] = (j, i)
    assert len(ids) == 80 * n  # coset shifts give disjoint cosets
    seen = set()
    moved = 0
    for j in range(80):
        for i in range(n):
            tj, ti = ids[int(syn.sigmas[j, i])]
            seen.add((tj, ti))
            assert syn.wires[j, i] == syn.wires[tj, ti]  # sigma only links equal values
            moved += (tj, ti) != (j, i)
    assert len(seen) == 80 * n and moved > 100


def test_fast_poseidon_tables(orc):
    d = orc.dll()
    rng = np.random.default_rng(3)
    for _ in range(20):
        a = (rng.integers(0, 2**63, 12, dtype=np.uint64) * np.uint64(2)) % np.uint64(P)
        b = a.copy()
        d.orc_poseidon_permute(a.ctypes.data_as(orc.u64p))
        d.orc_poseidon_permute_fast(b.ctypes.data_as(orc.u64p))
        assert np.array_equal(a, b)
    # first entries of plonky2's FAST_PARTIAL_* tables (upstream poseidon_goldilocks.rs, recalled)
    class F(ctypes.Structure):
        _fields_ = [("first", ctypes.c_uint64 * 12), ("rc", ctypes.c_uint64 * 22), ("vs", ctypes.c_uint64 * 242),
                    ("w", ctypes.c_uint64 * 242), ("init", ctypes.c_uint64 * 121)]
    d.orc_poseidon_fast_constants.restype = ctypes.POINTER(F)
    f = d.orc_poseidon_fast_constants().contents
    assert f.first[0] == 0x3cc3f892184df408 and f.rc[0] == 0x74cb2e819ae421ab
    assert f.vs[0] == 0x94877900674181c3 and f.w[0] == 0x3d999c961b7c63b0 and f.init[0] == 0x80772dc2645b280b


def test_extension_gates_constrain_their_rows(nlx, orc):
    """each of the four extension-field gates rejects a witness broken on one of its rows"""
    syn = nlx.SyntheticCircuit(8, seed=21, pct_poseidon=10, pct_arithmetic=10, pct_base_sum=5, pct_constant=5,
                               pct_extension=50)
    assert syn.num_gates == 10 and syn.num_selectors == 3
    kinds = [g.kind for g in syn.gates]
    assert kinds == [0, 1, 2, 4, 9, 8, 6, 3, 7, 5]  # sorted by (degree, id) as CircuitBuilder does
    circ = orc.Circuit.from_synthetic(syn)
    good = circ.prove(syn.wires, syn.public_inputs)
    assert circ.verify(good) == 1
    for kind, out_wire in ((6, 6), (7, 4), (8, 0), (9, 1)):
        g = syn.gates[kinds.index(kind)]
        rows = np.nonzero(syn.constants[g.selector_index] == g.index)[0]
        assert rows.size > 0, kind
        w = syn.wires.copy()
        w[out_wire, rows[0]] = (int(w[out_wire, rows[0]]) + 1) % P
        assert circ.verify(circ.prove(w, syn.public_inputs)) < 1, kind
    circ.close()


def test_misc_gates_constrain_their_rows(nlx, orc):
    """PoseidonMdsGate, ExponentiationGate, RandomAccessGate, CosetInterpolationGate: a witness broken on one of
    their rows is rejected"""
    syn = nlx.SyntheticCircuit(8, seed=23, pct_poseidon=5, pct_arithmetic=5, pct_base_sum=5, pct_constant=5,
                               pct_extension=10, pct_misc=50)
    assert syn.num_gates == 14
    kinds = [g.kind for g in syn.gates]
    assert kinds == [0, 1, 10, 2, 4, 9, 8, 6, 3, 7, 11, 12, 13, 5]
    circ = orc.Circuit.from_synthetic(syn)
    assert circ.verify(circ.prove(syn.wires, syn.public_inputs)) == 1
    # outputs, a power bit, an index, a bit wire; coset interpolation: shift, a value, the evaluation value, an
    # intermediate eval / prod, the shifted point
    for kind, wire in ((10, 25), (11, 1), (11, 70), (12, 0), (12, 74), (13, 0), (13, 8), (13, 35), (13, 37), (13, 43), (13, 45)):
        g = syn.gates[kinds.index(kind)]
        rows = np.nonzero(syn.constants[g.selector_index] == g.index)[0]
        assert rows.size > 0, kind
        w = syn.wires.copy()
        w[wire, rows[0]] = (int(w[wire, rows[0]]) + 1) % P
        assert circ.verify(circ.prove(w, syn.public_inputs)) < 1, (kind, wire)
    circ.close()


def test_u32_gates_constrain_their_rows(nlx, orc):
    """plonky2x's u32 gates (U32AddMany, U32Arithmetic, U32Subtraction, U32RangeCheck) and ComparisonGate: the
    generated witness is accepted, a witness broken on one of their rows is rejected"""
    syn = nlx.SyntheticCircuit(8, seed=29, pct_poseidon=5, pct_arithmetic=5, pct_base_sum=5, pct_constant=5, pct_u32=60)
    kinds = [g.kind for g in syn.gates]
    assert kinds == [0, 1, 2, 4, 3, 18, 14, 15, 17, 16, 5]  # (degree, id): Comparison < U32AddMany < U32Arithmetic < U32RangeCheck < U32Subtraction
    circ = orc.Circuit.from_synthetic(syn)
    assert circ.verify(circ.prove(syn.wires, syn.public_inputs)) == 1
    cases = ((14, 3), (14, 4), (14, 25), (14, 41),        # add-many: result, carry, a result limb, a carry limb
             (15, 3), (15, 4), (15, 5), (15, 18), (15, 40),  # arithmetic: low, high, inverse, low limb, high limb
             (16, 3), (16, 4), (16, 30),                  # subtraction: result, borrow, limb
             (17, 0), (17, 7),                            # range check: input, aux limb
             (18, 0), (18, 2), (18, 3), (18, 4), (18, 36), (18, 52), (18, 68), (18, 84))  # comparison wires of every class
    for kind, wire in cases:
        g = syn.gates[kinds.index(kind)]
        rows = np.nonzero(syn.constants[g.selector_index] == g.index)[0]
        assert rows.size > 0, kind
        w = syn.wires.copy()
        w[wire, rows[0]] = (int(w[wire, rows[0]]) + 1) % P
        assert circ.verify(circ.prove(w, syn.public_inputs)) < 1, (kind, wire)
    circ.close()
