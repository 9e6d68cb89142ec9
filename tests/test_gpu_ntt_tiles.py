"""Parity of the NTT tile kernels (csrc/ntt_kernels.hip: k_ntt_cols for 2^8 .. 2^11 points, k_ntt_c12 + k_ntt_s<REM> /
k_ntt_strided_reg above) through the C ABI, against the CPU oracle's transform: every column length from 2^7 to 2^21 that
picks a different kernel combination, column counts that do not fill a tile or a block's column loop, both directions, coset
shifts, and the LDE inside PolynomialBatch::from_values (plonky2_field::fft, PolynomialCoeffs::lde - crates pinned at
/root/reference/Cargo.lock:4912-4914) with 1, 2 and 8 cosets."""
import numpy as np
import pytest

from conftest import GEN, POW2_GEN, P, rand_field

pytestmark = pytest.mark.gpu


# 2^7: generic kernel; 2^8 .. 2^11: k_ntt_cols<0 .. 3>; 2^12: c12 alone; 13 .. 16: + register pass of 1 .. 4 levels;
# 17 .. 20: + k_ntt_s<1 .. 4>; 21: c12 + s<1> + register pass (two strided passes)
@pytest.mark.parametrize("log_n", [7, 8, 9, 10, 11, 12, 13, 15, 16, 17, 18, 19, 20, 21])
@pytest.mark.parametrize("n_cols", [1, 5, 17])
def test_forward_inverse_and_coset_against_oracle(nlx, ctx, orc, log_n, n_cols):
    if log_n >= 19 and n_cols == 17:
        pytest.skip("covered by the smaller column counts at this size")
    rng = np.random.default_rng(1000 * log_n + n_cols)
    a = rand_field(rng, (n_cols, 1 << log_n))
    a[0, :4] = [0, P - 1, 1, 0xFFFFFFFF]
    fwd = nlx.ntt(ctx, a)
    check = sorted({0, n_cols // 2, n_cols - 1})
    if log_n <= 16:
        for c in check:
            assert np.array_equal(fwd[c], orc.fft(a[c])), (log_n, n_cols, c)
    else:   # the oracle's transform is slow at these sizes: one column, the others through linearity against it
        assert np.array_equal(fwd[0], orc.fft(a[0]))
        s = ((a[0].astype(object) + a[-1].astype(object)) % P).astype(np.uint64)
        fs = nlx.ntt(ctx, s[None, :])[0]
        assert np.array_equal(fs, ((fwd[0].astype(object) + fwd[-1].astype(object)) % P).astype(np.uint64))
    assert (fwd < np.uint64(P)).all()
    assert np.array_equal(nlx.ntt(ctx, fwd, inverse=True), a)
    shift = GEN if log_n % 2 else 0x123456789ABCDEF
    cf = nlx.ntt(ctx, a, coset_shift=shift)
    if log_n <= 16:
        assert np.array_equal(cf[check[-1]], orc.fft(a[check[-1]], shift=shift))
    assert np.array_equal(nlx.ntt(ctx, cf, inverse=True, coset_shift=shift), a)


@pytest.mark.parametrize("shape", [(1, 8, 3), (15, 8, 1), (16, 8, 3), (17, 8, 2), (33, 9, 3), (7, 9, 1), (9, 10, 3), (4, 10, 2),
                                   (3, 11, 3), (2, 11, 1), (37, 12, 3), (19, 16, 1), (5, 18, 3), (3, 17, 2)])
def test_lde_inside_commit_against_oracle(nlx, ctx, orc, shape):
    """from_values = iNTT (DIF) + coset LDE (DIT) + leaf hashing: cap, coefficients and every digest equal the oracle's for
    column counts around the tile sizes (16 / 8 / 4 / 2 whole columns per tile below 2^12 points)."""
    n_cols, log_n, rate_bits = shape
    rng = np.random.default_rng(7 * n_cols + 100 * log_n + rate_bits)
    vals = rand_field(rng, (n_cols, 1 << log_n))
    pb = nlx.PolynomialBatch.from_values(ctx, vals, rate_bits, 0 if log_n + rate_bits < 4 else 4)
    if log_n <= 16:
        ref = orc.commit(vals, rate_bits, 0 if log_n + rate_bits < 4 else 4)
        assert np.array_equal(pb.cap, ref["cap"])
        assert np.array_equal(pb.coeffs(), ref["coeffs"])
        assert np.array_equal(pb.digests(), ref["digests"])
    else:
        # opened rows equal Horner evaluation of the coefficients at the leaf's point (oracle), and verify against the cap
        co = pb.coeffs()
        assert np.array_equal(nlx.ntt(ctx, co), vals)
        L = 1 << (log_n + rate_bits)
        idx = np.unique(rng.integers(0, L, size=6).astype(np.uint64))
        rows, paths = pb.open_rows(idx)
        w = pow(POW2_GEN, 1 << (32 - log_n - rate_bits), P)
        for j, i in enumerate(idx):
            assert orc.merkle_verify(rows[j], int(i), paths[j], pb.cap, 4)
            br = int(format(int(i), "0%db" % (log_n + rate_bits))[::-1], 2)
            x = GEN * pow(w, br, P) % P
            for c in (0, n_cols - 1):
                assert orc.eval_poly(co[c], x) == int(rows[j][c])
