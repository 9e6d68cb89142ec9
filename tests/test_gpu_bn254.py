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
