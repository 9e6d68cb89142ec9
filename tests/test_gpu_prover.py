"""GPU parity tests for the whole-proof path (nlx_circuit_build / nlx_prove through the C ABI):
proof BYTES must equal the CPU oracle's on the same synthetic circuits, the oracle's verifier
must accept them, and at BASELINE size the proof must verify (no full oracle prove needed)."""
import numpy as np
import pytest

from conftest import GEN, POW2_GEN, P

pytestmark = pytest.mark.gpu

SHAPES = [
    (5, dict(pct_poseidon=20, pct_arithmetic=30, pct_base_sum=5, pct_constant=5)),   # no FRI reduction round
    (6, dict(pct_poseidon=0, pct_arithmetic=50, pct_base_sum=10, pct_constant=10)),  # single selector
    (8, dict(pct_poseidon=25, pct_arithmetic=25, pct_base_sum=5, pct_constant=5)),
    (9, dict(pct_poseidon=10, pct_arithmetic=0, pct_base_sum=0, pct_constant=5)),
    (10, dict(pct_poseidon=30, pct_arithmetic=30, pct_base_sum=5, pct_constant=5)),
    (12, dict(pct_poseidon=20, pct_arithmetic=30, pct_base_sum=5, pct_constant=5)),
    (13, dict(pct_poseidon=20, pct_arithmetic=30, pct_base_sum=5, pct_constant=5)),  # two NTT passes
    (8, dict(pct_poseidon=15, pct_arithmetic=20, pct_base_sum=5, pct_constant=5, pct_extension=30)),  # all 10 gates
    (11, dict(pct_poseidon=10, pct_arithmetic=10, pct_base_sum=5, pct_constant=5, pct_extension=50)),
    (7, dict(pct_poseidon=0, pct_arithmetic=0, pct_base_sum=0, pct_constant=5, pct_extension=60)),
    (9, dict(pct_poseidon=10, pct_arithmetic=10, pct_base_sum=5, pct_constant=5, pct_extension=20, pct_misc=30)),  # all 14 gates
    (10, dict(pct_poseidon=0, pct_arithmetic=0, pct_base_sum=0, pct_constant=5, pct_extension=0, pct_misc=70)),
    (9, dict(pct_poseidon=10, pct_arithmetic=10, pct_base_sum=5, pct_constant=5, pct_extension=10, pct_misc=20, pct_u32=30)),  # all 19 gates
    (10, dict(pct_poseidon=0, pct_arithmetic=0, pct_base_sum=0, pct_constant=5, pct_u32=80)),  # u32 / comparison gates only
    (12, dict(pct_poseidon=20, pct_arithmetic=20, pct_base_sum=5, pct_constant=5, pct_u32=30)),  # nearx-like: hashing + u32 arithmetic
]


@pytest.mark.parametrize("log_n,kw", SHAPES)
def test_proof_bytes_equal_oracle(nlx, ctx, orc, log_n, kw):
    syn = nlx.SyntheticCircuit(log_n, seed=100 + log_n, **kw)
    ref = orc.Circuit.from_synthetic(syn)
    cd = nlx.CircuitData.from_synthetic(ctx, syn)
    assert np.array_equal(cd.constants_sigmas_cap, ref.constants_sigmas_cap())
    assert np.array_equal(cd.circuit_digest, ref.digest())
    want = ref.prove(syn.wires, syn.public_inputs)
    got = cd.prove(syn.wires, syn.public_inputs)
    assert len(got) == len(want)
    if got != want:
        a, b = np.frombuffer(got, np.uint8), np.frombuffer(want, np.uint8)
        first = int(np.nonzero(a != b)[0][0])
        pytest.fail("proof bytes differ from the oracle, first at byte %d of %d" % (first, len(want)))
    assert ref.verify(got) == 1
    # determinism + device-resident witness gives the same bytes
    assert cd.prove(syn.wires, syn.public_inputs) == got
    cd.close()
    ref.close()


def _assert_same_bytes(got, want, what):
    assert len(got) == len(want)
    if got != want:
        a, b = np.frombuffer(got, np.uint8), np.frombuffer(want, np.uint8)
        first = int(np.nonzero(a != b)[0][0])
        pytest.fail("%s: proof bytes differ from the oracle, first at byte %d of %d" % (what, first, len(want)))


def test_bench_shape_bytes_equal_oracle(nlx, ctx, orc):
    """The shape bench.py times by default until round 2 (2^16 rows, the nineteen-gate "nearx" mix,
    standard_recursion_config): whole-proof BYTES equal the oracle prover's (about 10 s of oracle time on 16 cores)."""
    import bench
    syn = nlx.SyntheticCircuit(16, seed=0x6E6C78, **bench.GATE_MIXES["nearx"])
    ref = orc.Circuit.from_synthetic(syn)
    cd = nlx.CircuitData.from_synthetic(ctx, syn)
    want = ref.prove(syn.wires, syn.public_inputs)
    got = cd.prove(syn.wires, syn.public_inputs)
    _assert_same_bytes(got, want, "2^16 nineteen-gate circuit")
    assert ref.verify(got) == 1
    cd.close()
    ref.close()


def test_widest_gate_bytes_equal_oracle(nlx, ctx, orc):
    """ComparisonGate { num_bits 25, num_chunks 25 } emits 132 constraints - more than PoseidonGate's 123 and more than
    the 128 alpha powers the quotient stage reserved for gate constraints until round 2 (it now sizes the table from the
    widest gate of the circuit, as the oracle always did)."""
    syn = nlx.SyntheticCircuit(9, seed=25, pct_poseidon=10, pct_arithmetic=10, pct_base_sum=5, pct_constant=5, pct_u32=50,
                               wide_comparison=True)
    assert any(g.kind == 18 and g.param0 == 25 and g.param1 == 25 for g in syn.gates)
    ref = orc.Circuit.from_synthetic(syn)
    cd = nlx.CircuitData.from_synthetic(ctx, syn)
    want = ref.prove(syn.wires, syn.public_inputs)
    assert ref.verify(want) == 1
    _assert_same_bytes(cd.prove(syn.wires, syn.public_inputs), want, "ComparisonGate(25, 25)")
    cd.close()
    ref.close()


def test_stagewise_against_oracle_trace(nlx, ctx, orc):
    """Localises a mismatch: challenges and intermediate polynomials of the oracle trace vs what the
    GPU commits (Z / partial products and quotient chunks recovered from the proof's openings)."""
    syn = nlx.SyntheticCircuit(8, seed=42)
    ref = orc.Circuit.from_synthetic(syn)
    want, tr = ref.prove(syn.wires, syn.public_inputs, trace=True)
    # Z / partial products: commit the oracle's values on the GPU and compare caps with the proof's
    pb = nlx.PolynomialBatch.from_values(ctx, tr["zs_partial_values"], 3, 4)
    zs_cap = np.frombuffer(want[512:1024], dtype=np.uint64).reshape(16, 4)
    assert np.array_equal(pb.cap, zs_cap)
    pq = nlx.PolynomialBatch.from_coeffs(ctx, tr["quotient_chunk_coeffs"], 3, 4)
    q_cap = np.frombuffer(want[1024:1536], dtype=np.uint64).reshape(16, 4)
    assert np.array_equal(pq.cap, q_cap)
    cd = nlx.CircuitData.from_synthetic(ctx, syn)
    got = cd.prove(syn.wires, syn.public_inputs)
    assert got[:512] == want[:512], "wires cap"
    assert got[512:1024] == want[512:1024], "Z/partial-products cap"
    assert got[1024:1536] == want[1024:1536], "quotient cap"
    assert got == want
    ref.close()


def test_pow_grind_smallest_nonce(nlx, ctx, orc):
    rng = np.random.default_rng(11)
    for bits in (4, 10, 16):
        state = rng.integers(0, P, 12, dtype=np.uint64)
        pos = 3
        nonce = nlx.pow_grind(ctx, state, pos, bits)

        def lz(w):
            st = state.copy()
            st[pos] = w
            out = orc.poseidon_permute(st.reshape(1, 12))[0]
            return 64 - int(out[7]).bit_length()
        assert lz(nonce) >= bits
        lo = max(0, nonce - 2000)
        assert all(lz(w) < bits for w in range(lo, nonce)), "a smaller valid nonce exists"


def test_unsatisfied_witness_rejected_by_verifier(nlx, ctx, orc):
    syn = nlx.SyntheticCircuit(7, seed=3)
    ref = orc.Circuit.from_synthetic(syn)
    cd = nlx.CircuitData.from_synthetic(ctx, syn)
    w = syn.wires.copy()
    w[0, 0] = (int(w[0, 0]) + 1) % P
    bad = cd.prove(w, syn.public_inputs)
    assert ref.verify(bad) < 1
    assert bad == ref.prove(w, syn.public_inputs)  # still the same bytes as the CPU prover
    ref.close()


def test_baseline_size_proof_verifies(nlx, ctx, orc):
    """2^15 rows, full standard_recursion_config: the oracle VERIFIER (cheap) accepts the GPU proof."""
    syn = nlx.SyntheticCircuit(15, seed=7)
    cd = nlx.CircuitData.from_synthetic(ctx, syn)
    proof = cd.prove(syn.wires, syn.public_inputs)
    ref = orc.Circuit.from_synthetic(syn)
    assert np.array_equal(cd.constants_sigmas_cap, ref.constants_sigmas_cap())
    assert ref.verify(proof) == 1
    ref.close()


def test_circuit_build_errors(nlx, ctx):
    syn = nlx.SyntheticCircuit(5, seed=1)
    d = syn.desc()
    d.rate_bits = 5
    with pytest.raises(nlx.NlxError):
        nlx.CircuitData(ctx, d, syn.constants, syn.sigmas)
    d = syn.desc()
    d.gates[0].kind = 99
    with pytest.raises(nlx.NlxError):
        nlx.CircuitData(ctx, d, syn.constants, syn.sigmas)
    d.gates[0].kind = 0


def test_large_proof_2p17_verifies(nlx, ctx, orc):
    """2^17 rows (9 GB-class tables are 2^20; this is the largest the CPU verifier setup handles in
    seconds): three-pass-free NTT path at 2^17, 2^20-point LDE, oracle verifier accepts."""
    syn = nlx.SyntheticCircuit(17, seed=17, pct_poseidon=30, pct_arithmetic=30, pct_base_sum=5, pct_constant=5)
    cd = nlx.CircuitData.from_synthetic(ctx, syn)
    proof = cd.prove(syn.wires, syn.public_inputs)
    ref = orc.Circuit.from_synthetic(syn)
    assert np.array_equal(cd.constants_sigmas_cap, ref.constants_sigmas_cap())
    assert ref.verify(proof) == 1
    ref.close()
    cd.close()


def test_headline_size_2p18_nineteen_gates_verifies(nlx, ctx, orc):
    """The default bench's outer proof - 2^18 rows, all nineteen gate kinds, 64 public inputs (BASELINE.json's metric is
    quoted on this shape): the circuit's constants / sigmas cap equals the oracle's and the oracle's verifier accepts the GPU
    proof; a proof with one opened value changed is rejected."""
    import bench
    syn = nlx.SyntheticCircuit(18, seed=1000, num_public_inputs=64, **bench.GATE_MIXES["nearx"])
    cd = nlx.CircuitData.from_synthetic(ctx, syn)
    proof = cd.prove(syn.wires, syn.public_inputs)
    ref = orc.Circuit.from_synthetic(syn)
    assert np.array_equal(cd.constants_sigmas_cap, ref.constants_sigmas_cap())
    assert ref.verify(proof) == 1
    bad = bytearray(proof)
    bad[len(bad) // 2] ^= 1
    assert ref.verify(bytes(bad)) != 1
    ref.close()
    cd.close()


@pytest.mark.parametrize("cfg", [
    dict(num_challenges=1, cap_height=0, fri_num_queries=5, fri_pow_bits=4, fri_arity_bits=2, fri_final_poly_bits=2),
    dict(cap_height=5, fri_num_queries=40, fri_pow_bits=0),
    dict(fri_arity_bits=3, fri_final_poly_bits=3, cap_height=2, fri_num_queries=11, fri_pow_bits=8),
])
def test_other_configs_match_oracle(nlx, ctx, orc, cfg):
    """the ABI is parametric in CircuitConfig / FriParams: non-default challenge counts, cap heights,
    FRI arities, query counts and PoW bits must stay bit-exact too"""
    config = nlx.CircuitConfig(**cfg)
    syn = nlx.SyntheticCircuit(9, seed=77, config=config, pct_poseidon=20, pct_arithmetic=30, pct_base_sum=5,
                               pct_constant=5, pct_extension=10)
    ref = orc.Circuit.from_synthetic(syn)
    cd = nlx.CircuitData.from_synthetic(ctx, syn)
    want = ref.prove(syn.wires, syn.public_inputs)
    got = cd.prove(syn.wires, syn.public_inputs)
    assert got == want
    assert ref.verify(got) == 1
    cd.close()
    ref.close()


def test_repeated_proving_is_stable(nlx, orc):
    """200 proofs on one context: identical bytes every time, no growth of the device allocation cache,
    and closing the context first is safe for its children."""
    c = nlx.Context(0)
    syn = nlx.SyntheticCircuit(10, seed=4)
    cd = nlx.CircuitData.from_synthetic(c, syn)
    first = cd.prove(syn.wires, syn.public_inputs)
    import torch
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(200):
        assert cd.prove_into(syn.wires, syn.public_inputs.ctypes.data) == len(first)
    assert cd.prove(syn.wires, syn.public_inputs) == first
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 64 << 20, "device memory grew by %d MiB over 200 proofs" % ((free0 - free1) >> 20)
    pb = nlx.PolynomialBatch.from_values(c, np.ones((2, 16), dtype=np.uint64), 3, 2)
    c.close()          # closes cd and pb first
    assert cd.handle is None and pb.handle is None
    cd.close()         # idempotent


def test_2p20_row_proof_sampled_against_the_oracle(nlx, ctx, orc):
    """2^20 rows (the largest proof the bench times; a full oracle prove would take minutes): the oracle VERIFIER accepts the
    GPU proof (every constraint at zeta, every FRI query's Merkle paths and folds), the constants / sigmas cap equals the oracle's
    own commitment, the proof's wires cap is the cap of the stage-level commitment of the same witness, and rows of that
    commitment opened at sampled LDE positions equal the witness polynomials evaluated there by the oracle's Horner loop (the
    coefficients checked by an oracle transform of the sampled columns) and verify against the cap."""
    log_n = 20
    syn = nlx.SyntheticCircuit(log_n, seed=2020, pct_poseidon=20, pct_arithmetic=30, pct_base_sum=5, pct_constant=5)
    cd = nlx.CircuitData.from_synthetic(ctx, syn)
    proof = cd.prove(syn.wires, syn.public_inputs)
    ref = orc.Circuit.from_synthetic(syn)
    try:
        assert np.array_equal(cd.constants_sigmas_cap, ref.constants_sigmas_cap())
        assert ref.verify(proof) == 1
    finally:
        ref.close()
        cd.close()
    pb = nlx.PolynomialBatch.from_values(ctx, syn.wires, 3, 4)
    assert pb.cap.tobytes() == proof[:512], "wires cap of the proof = cap of the stage-level commitment"
    co = pb.coeffs()
    cols = (0, 79, 134)
    for c in cols:
        assert np.array_equal(orc.fft(co[c]), syn.wires[c])
    L = 1 << (log_n + 3)
    idx = np.array([0, L // 2 + 7, L - 1, 1234567], dtype=np.uint64)
    rows, paths = pb.open_rows(idx)
    w = pow(POW2_GEN, 1 << (32 - log_n - 3), P)
    for j, i in enumerate(idx):
        br = int(format(int(i), "0%db" % (log_n + 3))[::-1], 2)
        x = GEN * pow(w, br, P) % P
        for c in cols:
            assert orc.eval_poly_ext(co[c], (x, 0)) == (int(rows[j][c]), 0)
        assert orc.merkle_verify(rows[j], int(i), paths[j], pb.cap, 4)
    pb.close()


def test_batch_prove_concurrent_workers(nlx, orc):
    """nlx_batch_prove: three contexts on one GPU, 7 jobs with different public inputs; every proof
    equals what a single context produces and is accepted by the oracle verifier."""
    ctxs = [nlx.Context(0) for _ in range(3)]
    syn = nlx.SyntheticCircuit(10, seed=31, num_public_inputs=8)
    workers = [nlx.CircuitData.from_synthetic(c, syn) for c in ctxs]
    ref = orc.Circuit.from_synthetic(syn)
    jobs, expect = [], []
    for j in range(7):
        s = nlx.SyntheticCircuit(10, seed=31, num_public_inputs=8)
        s.set_public_inputs(np.arange(8, dtype=np.uint64) + np.uint64(100 * j))
        jobs.append((s.wires, s.public_inputs))
        expect.append(workers[0].prove(s.wires, s.public_inputs))
    proofs = nlx.batch_prove(workers, jobs)
    assert proofs == expect
    assert all(ref.verify(p) == 1 for p in proofs)
    assert len(set(proofs)) == 7
    with pytest.raises(nlx.NlxError):
        nlx.batch_prove([workers[0], workers[0]], jobs[:1])  # same context twice
    # starting a worker thread fails (fault injected: nlx_abi_selftest 3 = the second worker, 4 = the first): the call still
    # returns a code, the workers that did start are joined (a joinable std::thread destroyed during unwinding would be
    # std::terminate) and every job is proved
    try:
        for kind in (3, 4):
            assert nlx.lib.dll.nlx_abi_selftest(kind) == 0
            assert nlx.batch_prove(workers, jobs) == expect
    finally:
        assert nlx.lib.dll.nlx_abi_selftest(5) == 0
    assert nlx.batch_prove(workers, jobs) == expect
    ref.close()
    for c in ctxs:
        c.close()


def test_batch_prove_two_devices(nlx, orc):
    """nlx_batch_prove with one worker per DEVICE (the in-process form of SURVEY 8e's partitioning): proofs equal the
    single-context ones whatever device proved them.  Needs two GPUs (the driver's multi-GPU node; skipped on a one-GPU box)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    ctxs = [nlx.Context(0), nlx.Context(1)]
    syn = nlx.SyntheticCircuit(10, seed=33, num_public_inputs=8)
    workers = [nlx.CircuitData.from_synthetic(c, syn) for c in ctxs]
    jobs, expect = [], []
    for j in range(6):
        s = nlx.SyntheticCircuit(10, seed=33, num_public_inputs=8)
        s.set_public_inputs(np.arange(8, dtype=np.uint64) + np.uint64(7 * j))
        jobs.append((s.wires, s.public_inputs))
        expect.append(workers[j % 2].prove(s.wires, s.public_inputs))
    assert nlx.batch_prove(workers, jobs) == expect
    for c in ctxs:
        c.close()


def test_poseidon_gate_in_kernel_and_in_its_own_kernel_give_the_same_proof(nlx, ctx, orc, monkeypatch):
    """PoseidonGate's partial rounds run in k_quotient_poseidon (the permutation's fused-block schedule, naive formulation) by
    default and as an item of k_quotient (upstream's fast formulation) under NLX_QUOTIENT_POSEIDON_INLINE=1, read at circuit
    build: both give the oracle's bytes.  (Circuits with lookup tables, whose terms reach the sums through the same accumulate
    path, run on the default in tests/test_gpu_lookup.py.)"""
    for log_n, kw in ((10, dict(pct_poseidon=30, pct_arithmetic=30, pct_base_sum=5, pct_constant=5)),
                      (9, dict(pct_poseidon=10, pct_arithmetic=10, pct_base_sum=5, pct_constant=5, pct_extension=10, pct_misc=20, pct_u32=30))):
        syn = nlx.SyntheticCircuit(log_n, seed=500 + log_n, **kw)
        ref = orc.Circuit.from_synthetic(syn)
        want = ref.prove(syn.wires, syn.public_inputs)
        proofs = []
        for inline in ("0", "1"):
            monkeypatch.setenv("NLX_QUOTIENT_POSEIDON_INLINE", inline)
            cd = nlx.CircuitData.from_synthetic(ctx, syn)
            proofs.append(cd.prove(syn.wires, syn.public_inputs))
            cd.close()
        _assert_same_bytes(proofs[0], want, "own kernel")
        _assert_same_bytes(proofs[1], want, "in k_quotient")
        ref.close()
