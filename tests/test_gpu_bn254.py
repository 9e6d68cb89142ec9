"""GPU parity tests for the BN254 scalar-field NTT (row f.4's first piece, nlx_bn254_ntt_batch) against the pure-Python
big-integer model oracle/bn254_py.py: every size from 2^0, both directions, canonical and Montgomery (gnark-crypto
fr.Element) element forms, edge values, and at 2^20 a random sample of outputs against Horner evaluation."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bn():
    import bn254_py
    return bn254_py


@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 4, 5, 6, 7, 9, 10, 12, 13])
def test_bn254_ntt_vs_model(nlx, ctx, bn, log_n):
    rng = random.Random(254 + log_n)
    n = 1 << log_n
    cols = [[rng.randrange(bn.R) for _ in range(n)] for _ in range(3)]
    cols[1][0], cols[1][n - 1] = bn.R - 1, 0                       # edge values
    if n > 2:
        cols[2][1], cols[2][2] = 1, (1 << 255) % bn.R
    got = nlx.bn254_unpack(nlx.bn254_ntt(ctx, nlx.bn254_pack(cols)))
    for c in range(3):
        assert got[c] == bn.ntt(cols[c]), (log_n, c)
    back = nlx.bn254_unpack(nlx.bn254_ntt(ctx, nlx.bn254_pack(got), inverse=True))
    assert back == cols
    # Montgomery form in and out (what a Go caller's []fr.Element holds)
    mont = [[bn.to_montgomery(x) for x in col] for col in cols]
    gm = nlx.bn254_unpack(nlx.bn254_ntt(ctx, nlx.bn254_pack(mont), montgomery=True))
    assert [[bn.from_montgomery(x) for x in col] for col in gm] == got
    gi = nlx.bn254_unpack(nlx.bn254_ntt(ctx, nlx.bn254_pack(gm), inverse=True, montgomery=True))
    assert gi == mont


def test_bn254_ntt_2p20_sampled(nlx, ctx, bn):
    """2^20 points: a random sample of outputs equals the polynomial evaluated at w^k by Horner (big integers)"""
    log_n = 20
    n = 1 << log_n
    rng = np.random.default_rng(2020)
    words = rng.integers(0, 2 ** 62, (1, n, 4), dtype=np.uint64)    # < 2^254 > r possible: reduce the top word so values < r
    words[:, :, 3] &= np.uint64((1 << 60) - 1)
    coeffs = nlx.bn254_unpack(words)[0]
    assert max(coeffs) < bn.R
    out = nlx.bn254_unpack(nlx.bn254_ntt(ctx, words))[0]
    w = bn.root_of_unity(log_n)
    for k in (0, 1, n // 2, n - 1, 123457, 999331):
        assert out[k] == bn.eval_poly(coeffs, pow(w, k, bn.R)), k
    assert nlx.bn254_unpack(nlx.bn254_ntt(ctx, nlx.bn254_pack([out]), inverse=True))[0] == coeffs


def test_bn254_ntt_argument_checks(nlx, ctx):
    with pytest.raises(nlx.NlxError):
        nlx.lib.dll.nlx_bn254_ntt_batch.argtypes  # binding exists
        ctx.check(nlx.lib.dll.nlx_bn254_ntt_batch(ctx.handle, None, 1, 4, 0, 0))
    with pytest.raises(nlx.NlxError):
        a = np.zeros((1, 2, 4), dtype=np.uint64)
        ctx.check(nlx.lib.dll.nlx_bn254_ntt_batch(ctx.handle, a.ctypes.data, 1, 29, 0, 0))   # 2-adicity 28


@pytest.mark.parametrize("log_n", [1, 3, 8, 12, 13])
def test_bn254_ntt_on_a_coset_and_bit_reversed_output(nlx, ctx, bn, log_n):
    """gnark-crypto's two FFT options: OnCoset (forward: evaluations on shift w^k, inverse: back) for the domain's generator 5
    and for a random shift, in both element forms; fft.DIF's bit-reversed output"""
    rng = random.Random(555 + log_n)
    n = 1 << log_n
    col = [rng.randrange(bn.R) for _ in range(n)]
    rev = [int(format(i, "0%db" % log_n)[::-1], 2) for i in range(n)]
    for shift in (5, rng.randrange(2, bn.R)):
        want = bn.ntt([c * pow(shift, j, bn.R) % bn.R for j, c in enumerate(col)])
        got = nlx.bn254_unpack(nlx.bn254_ntt(ctx, nlx.bn254_pack([col]), coset_shift=shift))[0]
        assert got == want, (log_n, shift)
        assert got == [bn.eval_poly(col, shift * pow(bn.root_of_unity(log_n), k, bn.R) % bn.R) for k in range(min(n, 3))] + got[3:]
        back = nlx.bn254_unpack(nlx.bn254_ntt(ctx, nlx.bn254_pack([got]), inverse=True, coset_shift=shift))[0]
        assert back == col
        mont = [[bn.to_montgomery(x) for x in col]]
        gm = nlx.bn254_unpack(nlx.bn254_ntt(ctx, nlx.bn254_pack(mont), montgomery=True, coset_shift=bn.to_montgomery(shift)))[0]
        assert [bn.from_montgomery(x) for x in gm] == want
        gb = nlx.bn254_unpack(nlx.bn254_ntt(ctx, nlx.bn254_pack([want]), inverse=True, coset_shift=shift, bitrev_out=True))[0]
        assert [gb[rev[i]] for i in range(n)] == col
    plain = bn.ntt(col)
    gb = nlx.bn254_unpack(nlx.bn254_ntt(ctx, nlx.bn254_pack([col]), bitrev_out=True))[0]
    assert [gb[rev[i]] for i in range(n)] == plain
    # fft.DIT: bit-reversed order in, natural order out - forward, inverse, on a coset, Montgomery words; and gnark's prover
    # pattern FFTInverse(DIF) -> FFT(DIT, OnCoset) without a reordering in between
    col_br = [col[rev[i]] for i in range(n)]
    assert nlx.bn254_unpack(nlx.bn254_ntt(ctx, nlx.bn254_pack([col_br]), bitrev_in=True))[0] == plain
    plain_br = [plain[rev[i]] for i in range(n)]
    assert nlx.bn254_unpack(nlx.bn254_ntt(ctx, nlx.bn254_pack([plain_br]), inverse=True, bitrev_in=True))[0] == col
    on5 = bn.ntt([c * pow(5, j, bn.R) % bn.R for j, c in enumerate(col)])
    assert nlx.bn254_unpack(nlx.bn254_ntt(ctx, nlx.bn254_pack([col_br]), bitrev_in=True, coset_shift=5))[0] == on5
    on5_br = [on5[rev[i]] for i in range(n)]
    assert nlx.bn254_unpack(nlx.bn254_ntt(ctx, nlx.bn254_pack([on5_br]), inverse=True, bitrev_in=True, coset_shift=5))[0] == col
    mont_br = [[bn.to_montgomery(x) for x in col_br]]
    gm = nlx.bn254_unpack(nlx.bn254_ntt(ctx, nlx.bn254_pack(mont_br), montgomery=True, bitrev_in=True, coset_shift=bn.to_montgomery(5)))[0]
    assert [bn.from_montgomery(x) for x in gm] == on5
    coeffs_br = nlx.bn254_ntt(ctx, nlx.bn254_pack([plain]), inverse=True, bitrev_out=True)     # values -> coefficients, bit-reversed
    assert nlx.bn254_unpack(nlx.bn254_ntt(ctx, coeffs_br, bitrev_in=True, coset_shift=5))[0] == on5
    assert nlx.lib.dll.nlx_bn254_ntt_batch_coset(ctx.handle, nlx.bn254_pack([col]).ctypes.data, 1, log_n, 0, 6, None) < 0      # both orders reversed
    assert nlx.lib.dll.nlx_bn254_ntt_batch_coset(ctx.handle, nlx.bn254_pack([col]).ctypes.data, 1, log_n, 0, 0, np.zeros(4, dtype=np.uint64).ctypes.data) < 0   # shift 0


# ---- the G1 multi-scalar multiplication (row f.4's second piece, nlx_bn254_msm_g1) ----
def _points(bn, count, seed):
    rng = random.Random(seed)
    return [bn.g1_mul(rng.randrange(1, bn.R), bn.G1) for _ in range(count)]


def _scalar_words(ks):
    out = np.zeros((len(ks), 4), dtype=np.uint64)
    for i, k in enumerate(ks):
        for w in range(4):
            out[i, w] = (int(k) >> (64 * w)) & 0xFFFFFFFFFFFFFFFF
    return out


@pytest.mark.parametrize("n", [1, 2, 3, 17, 300, 2048])
def test_bn254_msm_vs_model(nlx, ctx, bn, n):
    """sum_i k_i P_i equals the model's term-by-term sum, for canonical and for Montgomery (fr.Element) scalars"""
    rng = random.Random(1000 + n)
    pts = _points(bn, n, n)
    ks = [rng.randrange(bn.R) for _ in range(n)]
    want = bn.msm_g1(ks, pts)
    got = nlx.bn254_g1_unpack(nlx.bn254_msm_g1(ctx, nlx.bn254_g1_pack(pts), _scalar_words(ks)))
    assert got == want
    mont = _scalar_words([bn.to_montgomery(k) for k in ks])
    assert nlx.bn254_g1_unpack(nlx.bn254_msm_g1(ctx, nlx.bn254_g1_pack(pts), mont, montgomery=True)) == want


def test_bn254_msm_edge_cases(nlx, ctx, bn):
    """zero / one / r - 1 scalars, all-ones digits, the point at infinity, the same point many times (the buckets' doubling
    branch), a point and its negative (the cancelling branch), an empty input, a sum that is the point at infinity"""
    g = bn.G1
    pts = _points(bn, 6, 77)
    p0 = pts[0]
    cases = [
        ([0, 1, bn.R - 1, (1 << 253) - 1, 0xFFFF, 0xFFFF0000], pts),
        ([5, 7, 9, 11], [p0, None, p0, bn.g1_neg(p0)]),                      # infinity in the input; P, P, -P in different buckets
        ([0x1234] * 40, [p0] * 40),                                          # one bucket receives the same point 40 times
        ([0x1234] * 2 + [3], [p0, bn.g1_neg(p0), g]),                        # P + (-P) inside a bucket, then more
        ([0xABCD, 0xABCD], [p0, bn.g1_neg(p0)]),                             # the whole sum is the point at infinity
        ([0, 0, 0], pts[:3]),
        ([bn.R - 1] * 3, [g, g, g]),
    ]
    for ks, ps in cases:
        got = nlx.bn254_g1_unpack(nlx.bn254_msm_g1(ctx, nlx.bn254_g1_pack(ps), _scalar_words(ks)))
        assert got == bn.msm_g1(ks, ps), (ks, ps)
    assert nlx.bn254_g1_unpack(nlx.bn254_msm_g1(ctx, np.zeros((0, 8), dtype=np.uint64), np.zeros((0, 4), dtype=np.uint64))) is None
    assert nlx.lib.dll.nlx_bn254_msm_g1(ctx.handle, None, None, 4, 0, np.zeros(8, dtype=np.uint64).ctypes.data) < 0   # NULL inputs


def test_bn254_msm_2p20_structured(nlx, ctx, bn):
    """2^20 points drawn from 64 distinct ones with random scalars: sum_i k_i P_(i mod 64) = sum_j (sum_(i = j mod 64) k_i) P_j -
    64 scalar multiplications in the model pin a full-size run (and every bucket's doubling branch: its points repeat);
    linearity in the scalars: msm(k) + msm(k') = msm(k + k')"""
    import torch
    n, m = 1 << 20, 64
    base = _points(bn, m, 5)
    packed = np.tile(nlx.bn254_g1_pack(base), (n // m, 1))
    rs = np.random.RandomState(7)
    words = rs.randint(0, 1 << 62, size=(n, 4), dtype=np.int64).astype(np.uint64)   # 254-bit canonical scalars
    words[:, 3] &= np.uint64((1 << 60) - 1)
    ks = [sum(int(words[i, w]) << (64 * w) for w in range(4)) for i in range(n)]
    sums = [sum(ks[j::m]) % bn.R for j in range(m)]
    want = bn.msm_g1(sums, base)
    d_pts = torch.from_numpy(packed.view(np.int64)).to("cuda:%d" % ctx.device)
    d_ks = torch.from_numpy(words.view(np.int64)).to(d_pts.device)
    got = nlx.bn254_g1_unpack(nlx.bn254_msm_g1(ctx, d_pts, d_ks))
    assert got == want
    words2 = rs.randint(0, 1 << 62, size=(n, 4), dtype=np.int64).astype(np.uint64)
    words2[:, 3] &= np.uint64((1 << 60) - 1)
    ks2 = [sum(int(words2[i, w]) << (64 * w) for w in range(4)) for i in range(n)]
    both = _scalar_words([(a + b) % bn.R for a, b in zip(ks, ks2)])
    g2 = nlx.bn254_g1_unpack(nlx.bn254_msm_g1(ctx, d_pts, words2))
    g12 = nlx.bn254_g1_unpack(nlx.bn254_msm_g1(ctx, d_pts, both))
    assert bn.g1_add(got, g2) == g12


def test_bn254_g1_multiples_and_msm_over_distinct_points(nlx, ctx, bn):
    """nlx_bn254_g1_multiples: out[i] = (i + 1) P equals the model's repeated addition (all of the first 700 - a chunk
    boundary included - and samples up to 2^17); an MSM over 2^17 DISTINCT points with random scalars equals
    (sum k_i (i + 1)) P - one scalar multiplication in the model pins the whole run"""
    import torch
    rng = random.Random(4242)
    base = bn.g1_mul(rng.randrange(1, bn.R), bn.G1)
    n = 1 << 17
    dev = "cuda:%d" % ctx.device
    pts = nlx.bn254_g1_multiples(ctx, base, n, device=dev)
    host = pts.cpu().numpy().view(np.uint64)
    acc = None
    for i in range(700):
        acc = bn.g1_add(acc, base)
        assert nlx.bn254_g1_unpack(host[i]) == acc, i
    for i in [rng.randrange(700, n) for _ in range(40)] + [n - 1, 255, 256, 257]:
        assert nlx.bn254_g1_unpack(host[i]) == bn.g1_mul(i + 1, base), i
    small = nlx.bn254_g1_multiples(ctx, base, 5)                       # host output, fewer points than one chunk
    assert [nlx.bn254_g1_unpack(w) for w in small] == [bn.g1_mul(i + 1, base) for i in range(5)]
    rs = np.random.RandomState(17)
    words = rs.randint(0, 1 << 62, size=(n, 4), dtype=np.int64).astype(np.uint64)
    words[:, 3] &= np.uint64((1 << 60) - 1)
    ks = [sum(int(words[i, w]) << (64 * w) for w in range(4)) for i in range(n)]
    total = sum(k * (i + 1) for i, k in enumerate(ks)) % bn.R
    got = nlx.bn254_g1_unpack(nlx.bn254_msm_g1(ctx, pts, torch.from_numpy(words.view(np.int64)).to(dev)))
    assert got == bn.g1_mul(total, base)


# ---- the G2 multi-scalar multiplication (Groth16's B query, nlx_bn254_msm_g2) ----
def test_bn254_msm_g2_vs_model(nlx, ctx, bn):
    """sum_i k_i P_i over G2 (coordinates in Fq2) equals the model's term-by-term sum - random inputs, Montgomery scalars,
    the edge cases of the G1 test (infinity in the input, repeated and opposite points, extreme scalars, an infinite sum) -
    and, at 2^14 points drawn from 16 distinct ones, the regrouped sum; linear in the scalars"""
    import torch
    rng = random.Random(2222)
    g = bn.G2
    pts = [bn.g2_mul(rng.randrange(1, bn.R), g) for _ in range(24)]
    for n in (1, 2, 24):
        ks = [rng.randrange(bn.R) for _ in range(n)]
        want = bn.msm_g2(ks, pts[:n])
        assert nlx.bn254_g2_unpack(nlx.bn254_msm_g2(ctx, nlx.bn254_g2_pack(pts[:n]), _scalar_words(ks))) == want
        mont = _scalar_words([bn.to_montgomery(k) for k in ks])
        assert nlx.bn254_g2_unpack(nlx.bn254_msm_g2(ctx, nlx.bn254_g2_pack(pts[:n]), mont, montgomery=True)) == want
    p0 = pts[0]
    for ks, ps in (([0, 1, bn.R - 1, (1 << 253) - 1, 0xFFFF, 0xFFFF0000], pts[:6]), ([5, 7, 9, 11], [p0, None, p0, bn.g2_neg(p0)]),
                   ([0x1234] * 20, [p0] * 20), ([0x1234] * 2 + [3], [p0, bn.g2_neg(p0), g]), ([0xABCD, 0xABCD], [p0, bn.g2_neg(p0)]),
                   ([0, 0], pts[:2])):
        assert nlx.bn254_g2_unpack(nlx.bn254_msm_g2(ctx, nlx.bn254_g2_pack(ps), _scalar_words(ks))) == bn.msm_g2(ks, ps), ks
    assert nlx.bn254_g2_unpack(nlx.bn254_msm_g2(ctx, np.zeros((0, 16), dtype=np.uint64), np.zeros((0, 4), dtype=np.uint64))) is None
    n, m = 1 << 14, 16
    packed = np.tile(nlx.bn254_g2_pack(pts[:m]), (n // m, 1))
    rs = np.random.RandomState(9)
    words = rs.randint(0, 1 << 62, size=(n, 4), dtype=np.int64).astype(np.uint64)
    words[:, 3] &= np.uint64((1 << 60) - 1)
    ks = [sum(int(words[i, w]) << (64 * w) for w in range(4)) for i in range(n)]
    want = bn.msm_g2([sum(ks[j::m]) % bn.R for j in range(m)], pts[:m])
    d_pts = torch.from_numpy(packed.view(np.int64)).to("cuda:%d" % ctx.device)
    got = nlx.bn254_g2_unpack(nlx.bn254_msm_g2(ctx, d_pts, words))
    assert got == want
    twice = _scalar_words([2 * k % bn.R for k in ks])
    assert nlx.bn254_g2_unpack(nlx.bn254_msm_g2(ctx, d_pts, twice)) == bn.g2_add(got, got)
