"""ORACLE - TEST INFRASTRUCTURE ONLY.  Pure-Python (big-integer) model of the BN254 scalar-field NTT, for small sizes and for
sampled checks of large ones.  Follows the published definitions gnark-crypto's ecc/bn254/fr/fft implements (the recursive
wrap is Go, not in /root/reference - SURVEY.md §8 row f.4): r below, 2-adicity 28, the 2^28-th root of unity
5^((r-1)/2^28); FFT: values[k] = sum_j coeffs[j] w_n^(jk), FFTInverse = its inverse (with the 1/n).
Parity status: the root of unity equals gnark-crypto's published constant (asserted below); no Go-produced vector exists here."""
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
ROOT_2_28 = pow(5, (R - 1) >> 28, R)
assert ROOT_2_28 == 19103219067921713944291392827692070036145651957329286315305642004821462161904  # gnark-crypto's fr root of unity
assert pow(ROOT_2_28, 1 << 27, R) == R - 1
MONT_R = (1 << 256) % R


def root_of_unity(log_n):
    return pow(ROOT_2_28, 1 << (28 - log_n), R)


def ntt(coeffs, inverse=False):
    """recursive radix-2, natural order in and out"""
    n = len(coeffs)
    log_n = n.bit_length() - 1
    w = root_of_unity(log_n)
    if inverse:
        w = pow(w, R - 2, R)

    def rec(a, w):
        if len(a) == 1:
            return list(a)
        ev, od = rec(a[0::2], w * w % R), rec(a[1::2], w * w % R)
        out, t, h = [0] * len(a), 1, len(a) // 2
        for k in range(h):
            x = t * od[k] % R
            out[k], out[k + h] = (ev[k] + x) % R, (ev[k] - x) % R
            t = t * w % R
        return out
    out = rec([int(c) % R for c in coeffs], w)
    if inverse:
        ninv = pow(n, R - 2, R)
        out = [x * ninv % R for x in out]
    return out


def eval_poly(coeffs, x):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + int(c)) % R
    return acc


def to_montgomery(x):
    return int(x) * MONT_R % R


def from_montgomery(x):
    return int(x) * pow(MONT_R, R - 2, R) % R
