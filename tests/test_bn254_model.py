"""CPU: the pure-Python BN254 model (oracle/bn254_py.py) is an NTT - definition, inverse, Montgomery helpers."""
import random


def test_model_definition_and_inverse():
    import bn254_py as bn
    rng = random.Random(1)
    for log_n in (0, 1, 3, 6):
        n = 1 << log_n
        a = [rng.randrange(bn.R) for _ in range(n)]
        v = bn.ntt(a)
        w = bn.root_of_unity(log_n)
        for k in range(min(n, 5)):
            assert v[k] == sum(a[j] * pow(w, j * k, bn.R) for j in range(n)) % bn.R == bn.eval_poly(a, pow(w, k, bn.R))
        assert bn.ntt(v, inverse=True) == a
    assert bn.from_montgomery(bn.to_montgomery(12345)) == 12345 and bn.to_montgomery(1) == (1 << 256) % bn.R
    assert pow(bn.root_of_unity(28), 1 << 28, bn.R) == 1 and pow(bn.root_of_unity(28), 1 << 27, bn.R) != 1


def test_g1_model():
    """the G1 half of the model: group law, order, EIP-196's doubling of the generator, MSM = the sum of its terms"""
    import bn254_py as bn
    g = bn.G1
    assert (g[1] ** 2 - g[0] ** 3 - 3) % bn.Q == 0
    assert bn.g1_mul(bn.R, g) is None and bn.g1_mul(bn.R - 1, g) == bn.g1_neg(g) and bn.g1_add(g, bn.g1_neg(g)) is None
    assert bn.g1_mul(2, g) == bn.g1_add(g, g) and bn.g1_mul(7, g) == bn.g1_add(bn.g1_mul(3, g), bn.g1_mul(4, g))
    rng = random.Random(2)
    pts = [bn.g1_mul(rng.randrange(bn.R), g) for _ in range(5)]
    ks = [rng.randrange(bn.R) for _ in range(5)]
    for p in pts:
        assert (p[1] ** 2 - p[0] ** 3 - 3) % bn.Q == 0
    # linearity in the scalars, and against the discrete logs
    assert bn.msm_g1(ks, pts) == bn.g1_add(bn.msm_g1(ks[:2], pts[:2]), bn.msm_g1(ks[2:], pts[2:]))
    logs = [rng.randrange(bn.R) for _ in range(4)]
    assert bn.msm_g1(ks[:4], [bn.g1_mul(a, g) for a in logs]) == bn.g1_mul(sum(k * a for k, a in zip(ks, logs)) % bn.R, g)
