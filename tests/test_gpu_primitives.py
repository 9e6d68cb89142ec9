"""GPU parity tests (through the C ABI) for the hashing / NTT / commitment primitives:
HIP path vs the CPU oracle on seeded inputs, and vs the committed golden fixtures."""
import numpy as np
import pytest

from conftest import GEN, POW2_GEN, P, rand_field

pytestmark = pytest.mark.gpu


def test_poseidon_kats_gpu(nlx, ctx, golden):
    ins = np.array([k["in"] for k in golden["poseidon_kat"]], dtype=np.uint64)
    out = nlx.poseidon_permute(ctx, ins)
    for k, o in zip(golden["poseidon_kat"], out):
        assert [int(x) for x in o] == k["out"], k["source"]


def test_poseidon_batch_vs_oracle(nlx, ctx, orc):
    rng = np.random.default_rng(7)
    n = 4099  # ragged: not a multiple of the workgroup size
    st = rand_field(rng, (n, 12))
    st[0] = P - 1
    st[1] = 0
    st[2, :6] = 0xFFFFFFFF  # 2^32-1: carry edge of the reduction
    st[3] = 0xFFFFFFFF00000000
    got = nlx.poseidon_permute(ctx, st)
    want = orc.poseidon_permute(st)
    assert np.array_equal(got, want)
    assert (got < np.uint64(P)).all()


def test_hash_rows_lengths(nlx, ctx, orc, golden):
    for case in golden["hash_or_noop"]:
        if not case["in"]:
            continue
        rows = np.array([case["in"]] * 3, dtype=np.uint64)
        out = nlx.hash_rows(ctx, rows)
        assert out[0].tolist() == case["out"] and out[2].tolist() == case["out"]
    rng = np.random.default_rng(8)
    for row_len in (1, 3, 4, 5, 7, 8, 9, 16, 20, 135):
        rows = rand_field(rng, (130, row_len))
        got = nlx.hash_rows(ctx, rows)
        want = np.array([orc.hash_or_noop(r) for r in rows])
        assert np.array_equal(got, want), row_len


def test_merkle_tree(nlx, ctx, orc, golden):
    for key in ("merkle", "merkle_noop"):
        m = golden[key]
        t = nlx.MerkleTree(ctx, np.array(m["leaves"], dtype=np.uint64), m["cap_height"])
        flat = [w for lvl in m["levels"] for d in lvl for w in d]
        assert [int(x) for x in t.digests] == flat
        assert t.cap.tolist() == m["levels"][-1]
    rng = np.random.default_rng(9)
    for (n_leaves, leaf_len, cap_h) in ((1, 5, 0), (2, 9, 1), (1024, 135, 4), (4096, 20, 0), (256, 2, 4)):
        leaves = rand_field(rng, (n_leaves, leaf_len))
        t = nlx.MerkleTree(ctx, leaves, cap_h)
        dig, cap = orc.merkle_build(leaves, cap_h)
        assert np.array_equal(t.digests, dig)
        assert np.array_equal(t.cap, cap)
        idx = n_leaves // 3
        assert orc.merkle_verify(leaves[idx], idx, t.prove(idx), t.cap, cap_h)


@pytest.mark.parametrize("log_n", [0, 1, 3, 8, 12, 13, 14, 16, 17, 18, 19, 21])   # from 2^18 points: natural order out of the last pass itself (no reordering kernel)
def test_ntt_vs_oracle(nlx, ctx, orc, log_n):
    rng = np.random.default_rng(100 + log_n)
    n_cols = 3
    a = rand_field(rng, (n_cols, 1 << log_n))
    fwd = nlx.ntt(ctx, a)
    assert np.array_equal(fwd[1], orc.fft(a[1]))
    assert np.array_equal(nlx.ntt(ctx, fwd, inverse=True), a)
    cf = nlx.ntt(ctx, a, coset_shift=GEN)
    assert np.array_equal(cf[2], orc.fft(a[2], shift=GEN))
    assert np.array_equal(nlx.ntt(ctx, cf, inverse=True, coset_shift=GEN), a)


def test_ntt_linearity_large(nlx, ctx):
    """size-independent property at a size the oracle is too slow for: NTT(a+b) = NTT(a)+NTT(b), roundtrip."""
    rng = np.random.default_rng(5)
    log_n = 20
    a = rand_field(rng, (1, 1 << log_n))
    b = rand_field(rng, (1, 1 << log_n))
    s = ((a.astype(object) + b.astype(object)) % P).astype(np.uint64)
    fa, fb, fs = nlx.ntt(ctx, a), nlx.ntt(ctx, b), nlx.ntt(ctx, s)
    assert np.array_equal(((fa.astype(object) + fb.astype(object)) % P).astype(np.uint64), fs)
    assert np.array_equal(nlx.ntt(ctx, fa, inverse=True), a)
    # value at index 0 is the coefficient sum
    assert int(fa[0, 0]) == int(a.astype(object).sum() % P)


def test_commit_golden(nlx, ctx, golden):
    for key in ("commit", "commit_wide"):
        g = golden[key]
        pb = nlx.PolynomialBatch.from_values(ctx, np.array(g["values"], dtype=np.uint64), g["rate_bits"], g["cap_height"])
        assert pb.cap.tolist() == g["cap"]
        assert pb.coeffs().tolist() == g["coeffs"]
        assert pb.leaves().tolist() == g["leaves"]
        pc = nlx.PolynomialBatch.from_coeffs(ctx, np.array(g["coeffs"], dtype=np.uint64), g["rate_bits"], g["cap_height"])
        assert pc.cap.tolist() == g["cap"]


@pytest.mark.parametrize("shape", [(135, 10, 3, 4), (20, 12, 3, 4), (16, 13, 3, 4), (2, 14, 1, 4), (84, 9, 3, 0),
                                   (5, 4, 3, 4), (1, 0, 3, 1), (7, 15, 1, 2)])
def test_commit_vs_oracle(nlx, ctx, orc, shape):
    n_cols, log_n, rate_bits, cap_h = shape
    rng = np.random.default_rng(sum(shape))
    vals = rand_field(rng, (n_cols, 1 << log_n))
    pb = nlx.PolynomialBatch.from_values(ctx, vals, rate_bits, cap_h)
    ref = orc.commit(vals, rate_bits, cap_h)
    assert np.array_equal(pb.cap, ref["cap"])
    assert np.array_equal(pb.coeffs(), ref["coeffs"])
    assert np.array_equal(pb.digests(), ref["digests"])
    L = 1 << (log_n + rate_bits)
    idx = np.unique(rng.integers(0, L, size=28).astype(np.uint64))
    rows, paths = pb.open_rows(idx)
    for j, i in enumerate(idx):
        assert np.array_equal(rows[j], ref["leaves"][int(i)])
        assert np.array_equal(paths[j], orc.merkle_prove(ref["digests"], L, cap_h, int(i)))
        assert orc.merkle_verify(rows[j], int(i), paths[j], pb.cap, cap_h)
    # openings at an extension point
    zeta = (int(rand_field(rng, 1)[0]), int(rand_field(rng, 1)[0]))
    ev = pb.eval_at(zeta)
    for c in (0, n_cols - 1):
        assert tuple(int(x) for x in ev[c]) == orc.eval_poly_ext(ref["coeffs"][c], zeta)


def test_commit_large_roundtrip_properties(nlx, ctx, orc):
    """BASELINE-size table (135 x 2^16, rate 8): properties that need no full oracle run."""
    rng = np.random.default_rng(77)
    n_cols, log_n, rate_bits, cap_h = 135, 16, 3, 4
    vals = rand_field(rng, (n_cols, 1 << log_n))
    pb = nlx.PolynomialBatch.from_values(ctx, vals, rate_bits, cap_h)
    # (1) coefficients invert back to the values (oracle fft on 2 columns)
    co = pb.coeffs()
    for c in (0, 134):
        assert np.array_equal(orc.fft(co[c]), vals[c])
    # (2) opened rows are the polynomial evaluated at g*w^bitrev(idx) and verify against the cap
    L = 1 << (log_n + rate_bits)
    idx = np.array([0, 1, L // 2 + 3, L - 1, 12345], dtype=np.uint64)
    rows, paths = pb.open_rows(idx)
    w = pow(POW2_GEN, 1 << (32 - log_n - rate_bits), P)
    for j, i in enumerate(idx):
        br = int(format(int(i), "0%db" % (log_n + rate_bits))[::-1], 2)
        x = GEN * pow(w, br, P) % P
        assert orc.eval_poly_ext(co[5], (x, 0)) == (int(rows[j][5]), 0)
        assert orc.merkle_verify(rows[j], int(i), paths[j], pb.cap, cap_h)
    # (3) determinism
    pb2 = nlx.PolynomialBatch.from_values(ctx, vals, rate_bits, cap_h)
    assert np.array_equal(pb.cap, pb2.cap)


def test_error_paths(nlx, ctx):
    with pytest.raises(ValueError):
        nlx.MerkleTree(ctx, np.zeros((3, 5), dtype=np.uint64), 0)
    with pytest.raises(nlx.NlxError):
        nlx.MerkleTree(ctx, np.zeros((4, 5), dtype=np.uint64), 3)
    pb = nlx.PolynomialBatch.from_values(ctx, np.ones((2, 8), dtype=np.uint64), 3, 2)
    with pytest.raises(nlx.NlxError):
        pb.open_rows(np.array([64], dtype=np.uint64))


def _sum_mod_p(a):
    lo = int((a & np.uint64(0xFFFFFFFF)).sum(dtype=np.uint64))
    hi = int((a >> np.uint64(32)).sum(dtype=np.uint64))
    return (lo + (hi << 32)) % P


def test_ntt_2p24_three_passes(nlx, ctx):
    """BASELINE config 5's size (NTT at 2^24, three LDS passes): round trip, DC term, Nyquist term."""
    rng = np.random.default_rng(24)
    log_n = 24
    a = rand_field(rng, (1, 1 << log_n))
    f = nlx.ntt(ctx, a)
    assert int(f[0, 0]) == _sum_mod_p(a[0])                       # value at w^0 = sum of coefficients
    alt = (_sum_mod_p(a[0, 0::2]) - _sum_mod_p(a[0, 1::2])) % P  # value at w^(n/2) = -1: alternating sum
    assert int(f[0, 1 << (log_n - 1)]) == alt
    assert np.array_equal(nlx.ntt(ctx, f, inverse=True), a)
    assert (f < np.uint64(P)).all()


def test_ntt_2p24_sampled_against_horner(nlx, ctx, orc):
    """BASELINE config 5 (2^24-point NTT, a batch of columns): a random sample of outputs of every column equals the
    polynomial evaluated at w^k by the oracle's Horner loop (2^24 multiplications per sample), forward and coset."""
    rng = np.random.default_rng(2424)
    log_n, n_cols = 24, 4
    a = rand_field(rng, (n_cols, 1 << log_n))
    w = pow(POW2_GEN, 1 << (32 - log_n), P)
    f = nlx.ntt(ctx, a)
    fc = nlx.ntt(ctx, a[:2], coset_shift=GEN)
    ks = [0, 1, (1 << log_n) - 1, 1 << 23] + [int(k) for k in rng.integers(0, 1 << log_n, 4)]
    for c in range(n_cols):
        for k in ks[c::2] if c else ks:                      # every sample on column 0, half of them on the others
            assert int(f[c, k]) == orc.eval_poly(a[c], pow(w, k, P)), (c, k)
    for k in ks[:5]:
        assert int(fc[1, k]) == orc.eval_poly(a[1], GEN * pow(w, k, P) % P), k
    assert np.array_equal(nlx.ntt(ctx, fc, inverse=True, coset_shift=GEN), a[:2])


def test_field_core_edge_values(nlx, ctx):
    """GoldilocksField core (incl. the hand-written carry-chain multiply) on every pair of edge values:
    0, 1, 2^32 +-1, p +-1, 2^64-1, ... plus random pairs, against Python big integers."""
    E = 0xFFFFFFFF
    edge = [0, 1, 2, 3, E - 1, E, E + 1, E + 2, 2 * E, (1 << 33) - 1, 1 << 33, (1 << 48), (1 << 63) - 1, 1 << 63,
            (1 << 63) + 1, P - 2, P - 1, P, P + 1, P + E - 1, (1 << 64) - E - 1, (1 << 64) - E, (1 << 64) - 2, (1 << 64) - 1,
            0xFFFFFFFE00000001, 0xFFFFFFFF00000000, 0x00000001FFFFFFFF, 0x0000000100000000, 0xFFFFFFFEFFFFFFFF,
            0x8000000080000000, 0x7FFFFFFF7FFFFFFF, 0xAAAAAAAA55555555]
    rng = np.random.default_rng(3)
    rnd = [int(x) for x in rng.integers(0, 2**63, 64, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, 64, dtype=np.uint64)]
    vals = edge + rnd
    a = np.array([x for x in vals for _ in vals], dtype=np.uint64)
    b = np.array([y for _ in vals for y in vals], dtype=np.uint64)
    out = nlx.field_ops(ctx, a, b)
    for i in range(a.size):
        x, y = int(a[i]), int(b[i])
        xc, yc = x % P, y % P
        assert int(out[0, i]) == xc * yc % P, (hex(x), hex(y), "mul")
        assert int(out[1, i]) == (xc + yc) % P, (hex(x), hex(y), "add")
        assert int(out[2, i]) == (xc - yc) % P, (hex(x), hex(y), "sub")
        assert int(out[4, i]) == x * y % P, (hex(x), hex(y), "mul_loose on unreduced inputs")
        if xc:
            assert int(out[3, i]) * xc % P == 1, (hex(x), "inv")
    assert (out < np.uint64(P)).all()


def test_context_memory_accounting(nlx, orc):
    """nlx_ctx_memory / nlx_ctx_trim: handles account for their bytes, freed blocks are cached until trimmed"""
    c = nlx.Context(0)
    vals = np.zeros((16, 1 << 12), dtype=np.uint64)
    nlx.PolynomialBatch.from_values(c, vals, 3, 4).close()   # builds the twiddle / coset tables, which stay live
    c.trim()
    r0, u0 = c.memory()
    assert r0 == u0 > 0
    pb = nlx.PolynomialBatch.from_values(c, vals, 3, 4)
    r1, u1 = c.memory()
    assert u1 - u0 >= 16 * (1 << 15) * 8          # at least the LDE table
    assert r1 >= u1
    pb.close()
    r2, u2 = c.memory()
    assert u2 == u0 and r2 == r1                  # released to the cache, not to the driver
    c.trim()
    assert c.memory() == (r0, u0)                 # only the tables remain
    c.close()


def test_device_buffer_roundtrip_and_use(nlx, ctx):
    """nlx_buf: upload / download round trip, range checks, and a buffer's device pointer as a prover input"""
    rng = np.random.default_rng(5)
    a = rng.integers(0, P, (3, 1000), dtype=np.uint64)
    b = nlx.DeviceBuffer.from_array(ctx, a)
    assert b.nbytes == a.nbytes and np.array_equal(b.download().reshape(3, 1000), a)
    b.upload(np.arange(10, dtype=np.uint64), offset=80)
    assert np.array_equal(b.download(offset=80, nbytes=80), np.arange(10, dtype=np.uint64))
    with pytest.raises(nlx.NlxError):
        b.upload(np.zeros(8, dtype=np.uint64), offset=a.nbytes - 8)
    with pytest.raises(nlx.NlxError):
        b.download(offset=a.nbytes + 8, nbytes=8)
    b.close()
    syn = nlx.SyntheticCircuit(8, seed=3)
    cd = nlx.CircuitData.from_synthetic(ctx, syn)
    w = nlx.DeviceBuffer.from_array(ctx, syn.wires)
    assert cd.prove(w.ptr, syn.public_inputs) == cd.prove(syn.wires, syn.public_inputs)
    w.close()
    cd.close()


def test_stream_priority_and_cu_mask_keep_results(nlx, orc):
    """nlx_ctx_set_priority / nlx_ctx_set_cu_mask replace the context's stream: scheduling knobs, results unchanged"""
    rng = np.random.default_rng(5)
    vals = rand_field(rng, (9, 1 << 8))
    want = orc.commit(vals, 3, 2)["cap"]
    c = nlx.Context(0)
    c.set_priority(True)
    assert np.array_equal(nlx.PolynomialBatch.from_values(c, vals, 3, 2).cap, want)
    c.set_cu_mask(range(0, 256, 4))          # a quarter of the chip
    assert np.array_equal(nlx.PolynomialBatch.from_values(c, vals, 3, 2).cap, want)
    c.set_priority(False)                    # back to an unrestricted, normal-priority stream
    assert np.array_equal(nlx.PolynomialBatch.from_values(c, vals, 3, 2).cap, want)
    with pytest.raises(nlx.NlxError):
        c.set_cu_mask([])                    # a mask that selects no compute unit
    c.close()


@pytest.mark.parametrize("world,log_n", [(2, 13), (4, 14)])
def test_one_ntt_split_over_ranks(world, log_n):
    """cfg5 / SURVEY 8e: ONE transform split over 2 and 4 ranks (one and two cross-rank levels; all ranks on this GPU over
    gloo - the send / recv pattern, the cross-level kernel and the cyclic output distribution are those of the RCCL run):
    the reassembled result equals the transform done by a single context"""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(29650 + world), os.path.join(ROOT, "tests", "tools", "split_ntt_ranks.py"), str(log_n), "3"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0 and "equals one transform: True" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])
