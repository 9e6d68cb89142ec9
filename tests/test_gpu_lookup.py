"""GPU parity tests of the lookup argument (plonky2 LookupGate / LookupTableGate circuits) through the C ABI:
nlx_circuit_build with the descriptor's table arrays, nlx_prove = set_lookup_wires + RE / partial-sum polynomials + the lookup
terms of the quotient + their openings and FRI columns, all on the device (csrc/lookup_arg.hip).  Proof BYTES must equal the CPU
oracle's (tests/test_lookup_oracle.py pins the oracle's side) and the oracle's verifier must accept them."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# log_n, tables, log2(entries), lookups per table, gate mix of the rest of the circuit
SHAPES = [
    (9, 1, 6, 100, {}),                                          # 3 LookupGate rows (the last padded), 3 table rows
    (9, 1, 4, 40, dict(pct_poseidon=0, pct_arithmetic=40)),      # exactly one full LookupGate row: no padding; one table row
    (10, 2, 8, 333, {}),                                         # two tables, the second with permuted inputs
    (12, 1, 10, 80, dict(pct_extension=20, pct_misc=20, pct_u32=20, pct_poseidon=10, pct_arithmetic=10)),   # all 21 gate kinds
    (13, 1, 16, 5000, {}),                                       # a full 16-bit table: 2 521 table rows, 125 LookupGate rows
    (14, 3, 12, 20000, dict(pct_poseidon=20, pct_arithmetic=20, pct_u32=20)),
    (16, 1, 16, 60000, {}),                                      # 2^16 rows with a full 2^16-pair table: the largest byte comparison with tables
]


def _same(got, want, what):
    assert len(got) == len(want), "%s: %d bytes against the oracle's %d" % (what, len(got), len(want))
    if got != want:
        a, b = np.frombuffer(got, np.uint8), np.frombuffer(want, np.uint8)
        pytest.fail("%s: proof bytes differ from the oracle's, first at byte %d of %d" % (what, int(np.nonzero(a != b)[0][0]), len(want)))


@pytest.mark.parametrize("log_n,T,bits,nl,kw", SHAPES)
def test_proof_bytes_with_tables_equal_oracle(nlx, ctx, orc, log_n, T, bits, nl, kw):
    syn = nlx.SyntheticCircuit(log_n, seed=40 + log_n, num_luts=T, lut_bits=bits, num_lookups=nl, **kw)
    ref = orc.Circuit.from_synthetic(syn)
    cd = nlx.CircuitData.from_synthetic(ctx, syn)
    try:
        assert np.array_equal(cd.constants_sigmas_cap, ref.constants_sigmas_cap())
        want = ref.prove(syn.wires, syn.public_inputs)
        assert len(want) > 0
        got = cd.prove(syn.wires, syn.public_inputs)
        _same(got, want, "2^%d rows, %d table(s) of 2^%d, %d lookups each" % (log_n, T, bits, nl))
        assert ref.verify(got) == 1
        assert cd.prove(syn.wires, syn.public_inputs) == got      # the host witness was not touched: same bytes again
    finally:
        cd.close()
        ref.close()


def test_one_challenge_round_with_tables(nlx, ctx, orc):
    """num_challenges = 1: one round of lookup challenges (beta, gamma and two more), one RE / partial-sum set, one alpha sum"""
    cfg = nlx.CircuitConfig(num_challenges=1)
    syn = nlx.SyntheticCircuit(10, seed=77, num_luts=2, lut_bits=7, num_lookups=90, config=cfg)
    ref = orc.Circuit.from_synthetic(syn)
    cd = nlx.CircuitData.from_synthetic(ctx, syn)
    try:
        want = ref.prove(syn.wires, syn.public_inputs)
        got = cd.prove(syn.wires, syn.public_inputs)
        _same(got, want, "one challenge round, two tables")
        assert ref.verify(got) == 1
    finally:
        cd.close()
        ref.close()


def test_device_witness_gets_the_lookup_wires_in_place(nlx, ctx, orc):
    """a device-resident witness is written as prover::set_lookup_wires writes the PartitionWitness: multiplicities on the
    LookupTableGate rows, the table's first pair on the padding slots of the last LookupGate row - and nothing else"""
    import torch
    syn = nlx.SyntheticCircuit(10, seed=9, num_luts=2, lut_bits=7, num_lookups=130)
    ref = orc.Circuit.from_synthetic(syn)
    cd = nlx.CircuitData.from_synthetic(ctx, syn)
    try:
        dev = torch.from_numpy(syn.wires.view(np.int64)).cuda()
        got = cd.prove(dev, syn.public_inputs)
        _same(got, ref.prove(syn.wires, syn.public_inputs), "device witness")
        after = dev.cpu().numpy().view(np.uint64)
        assert np.array_equal(after, ref.set_lookup_wires(syn.wires))
        assert not np.array_equal(after, syn.wires)
        assert cd.prove(dev, syn.public_inputs) == got             # idempotent on an already completed witness
    finally:
        cd.close()
        ref.close()


def test_a_lookup_outside_the_table_is_an_error_and_bad_descriptors_are_refused(nlx, ctx):
    import ctypes
    syn = nlx.SyntheticCircuit(9, seed=5, num_luts=1, lut_bits=6, num_lookups=100)
    cd = nlx.CircuitData.from_synthetic(ctx, syn)
    try:
        w = syn.wires.copy()
        w[0, syn.lookup_rows[0, 0]] = 60000
        with pytest.raises(nlx.NlxError, match="not in its table"):
            cd.prove(w, syn.public_inputs)
        assert len(cd.prove(syn.wires, syn.public_inputs)) > 0      # the circuit is still usable
        # the stage-level calls carry no lookup challenges
        with pytest.raises(nlx.NlxError):
            cd.partial_products_and_zs(syn.wires, [1, 2], [3, 4])
    finally:
        cd.close()
    d = syn.desc()
    rows = syn.lookup_rows.copy()
    rows[0, 2] += 1                                                 # one table row too many for 64 entries
    d.lookup_rows = rows.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
    with pytest.raises(nlx.NlxError):
        nlx.CircuitData(ctx, d, syn.constants, syn.sigmas)
    d = syn.desc()
    d.lut_pairs = None
    with pytest.raises(nlx.NlxError):
        nlx.CircuitData(ctx, d, syn.constants, syn.sigmas)
