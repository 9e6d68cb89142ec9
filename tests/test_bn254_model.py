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
