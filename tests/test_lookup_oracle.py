"""CPU tests of the ORACLE's lookup argument (plonky2 gates::{lookup, lookup_table}, prover::{set_lookup_wires,
compute_lookup_polys}, vanishing_poly::{check_lookup_constraints, get_lut_poly}; oracle/plonk.c) on the synthetic circuits with
tables (csrc/synth.cpp nlx_synth_circuit_lookups).  plonky2's source is absent here: the restatement is
pinned by its own verifier and by the argument's algebraic properties below, not by upstream vectors - parity unpinned,
as for the rest of the plonky2 path."""
import numpy as np
import pytest

P = (1 << 64) - (1 << 32) + 1

SHAPES = [(9, 1, 6, 100), (10, 2, 8, 333), (11, 1, 10, 80)]   # log_n, tables, log2(entries), lookups per table


def _circuit(nlx, orc, shape, **kw):
    log_n, T, bits, nl = shape
    syn = nlx.SyntheticCircuit(log_n, seed=5, num_public_inputs=4, num_luts=T, lut_bits=bits, num_lookups=nl, **kw)
    return syn, orc.Circuit.from_synthetic(syn)


@pytest.mark.parametrize("shape", SHAPES)
def test_proof_with_tables_verifies_and_tampering_is_caught(nlx, orc, shape):
    syn, c = _circuit(nlx, orc, shape)
    try:
        proof = c.prove(syn.wires, syn.public_inputs)
        assert len(proof) > 0 and c.verify(proof) == 1
        # OpeningSet order: constants, sigmas, wires, zs, zs_next, lookup_zs, ...: flip one bit of the first lookup opening
        off = 3 * 16 * 32 + 16 * (syn.constants.shape[0] + 80 + 135 + 2 + 2)
        bad = bytearray(proof)
        bad[off + 3] ^= 1
        assert c.verify(bytes(bad)) == -2
        # the same circuit without the caller doing anything about multiplicities: the prover's set_lookup_wires did it
        assert int(syn.wires[2::3][:26, syn.lookup_rows[0, 1]:syn.lookup_rows[0, 2] + 1].sum()) == 0
    finally:
        c.close()


def test_a_lookup_outside_the_table_or_with_a_wrong_output_does_not_prove(nlx, orc):
    syn, c = _circuit(nlx, orc, SHAPES[0])
    try:
        w = syn.wires.copy()
        w[0, syn.lookup_rows[0, 0]] = 60000            # input not in the table: set_lookup_wires refuses (upstream panics)
        assert c.prove(w, syn.public_inputs) == b""
        w = syn.wires.copy()
        w[1, syn.lookup_rows[0, 0]] ^= 1               # (input, wrong output): not a table pair
        proof = c.prove(w, syn.public_inputs)
        assert proof == b"" or c.verify(proof) != 1
    finally:
        c.close()


@pytest.mark.parametrize("shape", SHAPES[:2])
def test_set_lookup_wires_and_the_lookup_polynomials(nlx, orc, shape):
    log_n, T, bits, nl = shape
    syn, c = _circuit(nlx, orc, shape)
    try:
        w = c.set_lookup_wires(syn.wires)
        pairs = syn.lut_pairs.reshape(T, 1 << bits, 2).astype(np.uint64)
        for t in range(T):
            last_lu, last_lut, first_lut = (int(x) for x in syn.lookup_rows[t])
            lu = w[:80, last_lu:last_lut]                         # (80, rows): slot i = wires 2i, 2i+1
            inp, out = lu[0::2].T.reshape(-1), lu[1::2].T.reshape(-1)
            # every LookupGate slot (padding included) now holds a pair of the table; multiplicities count them
            index_of = {int(a): i for i, a in enumerate(pairs[t, :, 0])}
            counts = np.zeros(1 << bits, dtype=np.uint64)
            for a, b in zip(inp, out):
                i = index_of[int(a)]
                assert int(pairs[t, i, 1]) == int(b)
                counts[i] += 1
            assert counts.sum() == 40 * (last_lut - last_lu) and counts.sum() >= nl
            lut = w[:78, last_lut:first_lut + 1]
            mult = lut[2::3][:, ::-1].T.reshape(-1)[:1 << bits]   # rows upside down: first entries on first_lut_row
            assert np.array_equal(mult, counts)
            assert np.array_equal(lut[0::3][:, ::-1].T.reshape(-1)[:1 << bits], pairs[t, :, 0])
        # the committed lookup polynomials: per challenge RE, SLDC_0..5 after the 2 * (1 + 9) Zs / partial products
        _, info = c.prove(syn.wires, syn.public_inputs, trace=True)
        zs = info["zs_partial_values"]
        assert zs.shape[0] == 20 + 2 * 7
        for ci in range(2):
            A, B, alpha, delta = (int(x) for x in info["deltas"][4 * ci:4 * ci + 4])
            re, sldc = zs[20 + 7 * ci], zs[21 + 7 * ci:28 + 7 * ci]
            for t in range(T):
                last_lu, last_lut, first_lut = (int(x) for x in syn.lookup_rows[t])
                # get_lut_poly: the pairs as coefficients in delta, first entry highest, zero-padded to whole rows
                rows = first_lut - last_lut + 1
                acc = 0
                for i in range(26 * rows):
                    cf = (int(pairs[t, i, 0]) + B * int(pairs[t, i, 1])) % P if i < (1 << bits) else 0
                    acc = (acc * delta + cf) % P
                assert int(re[last_lut]) == acc and int(re[first_lut + 1]) == 0
                # logUp: sum mult / (alpha - looked) over the table == sum 1 / (alpha - looking) over the lookups, so the
                # running Sum - LDC is back at 0 on the first LookupGate row
                assert int(sldc[5][last_lu]) == 0 and int(sldc[5][last_lut]) != 0
                assert not sldc[:, first_lut + 1].any()
    finally:
        c.close()


def test_circuits_without_tables_are_untouched(nlx, orc):
    """num_luts = 0 leaves descriptor, witness and proof as they were (the golden fixtures pin the bytes; here: the shape)"""
    syn = nlx.SyntheticCircuit(9, seed=3)
    d = syn.desc()
    assert d.num_luts == 0 and not d.lut_sizes and syn.constants.shape[0] == syn.num_selectors + 2
    assert [g.kind for g in syn.gates][0] == 0
