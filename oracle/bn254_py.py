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


# ---- G1: y^2 = x^3 + 3 over Fq, generator (1, 2), group order R (EIP-196's alt_bn128) ----
Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
G1 = (1, 2)


def g1_add(p, q):
    """affine addition; None = the point at infinity"""
    if p is None:
        return q
    if q is None:
        return p
    x1, y1 = p
    x2, y2 = q
    if x1 == x2:
        if (y1 + y2) % Q == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, Q - 2, Q) % Q
    else:
        lam = (y2 - y1) * pow(x2 - x1, Q - 2, Q) % Q
    x3 = (lam * lam - x1 - x2) % Q
    return x3, (lam * (x1 - x3) - y1) % Q


def g1_neg(p):
    return None if p is None else (p[0], (-p[1]) % Q)


def _jac_dbl(p):
    x, y, z = p
    if z == 0:
        return p
    a, b = x * x % Q, y * y % Q
    c = b * b % Q
    d = 2 * ((x + b) * (x + b) - a - c) % Q
    e = 3 * a % Q
    x3 = (e * e - 2 * d) % Q
    return x3, (e * (d - x3) - 8 * c) % Q, 2 * y * z % Q


def _jac_add_affine(p, q):
    x1, y1, z1 = p
    if z1 == 0:
        return q[0], q[1], 1
    z1z1 = z1 * z1 % Q
    u2, s2 = q[0] * z1z1 % Q, q[1] * z1 * z1z1 % Q
    h, r = (u2 - x1) % Q, (s2 - y1) % Q
    if h == 0:
        return _jac_dbl((q[0], q[1], 1)) if r == 0 else (1, 1, 0)
    hh = h * h % Q
    hhh, v = h * hh % Q, x1 * hh % Q
    x3 = (r * r - hhh - 2 * v) % Q
    return x3, (r * (v - x3) - y1 * hhh) % Q, z1 * h % Q


def g1_mul(k, p):
    """k * p by double-and-add in Jacobian coordinates (one inversion at the end)"""
    k %= R
    if p is None or k == 0:
        return None
    acc = (1, 1, 0)
    for bit in range(k.bit_length() - 1, -1, -1):
        acc = _jac_dbl(acc)
        if (k >> bit) & 1:
            acc = _jac_add_affine(acc, p)
    if acc[2] == 0:
        return None
    zi = pow(acc[2], Q - 2, Q)
    return acc[0] * zi * zi % Q, acc[1] * zi * zi * zi % Q


def msm_g1(scalars, points):
    """sum_i scalars[i] * points[i], term by term"""
    acc = None
    for k, p in zip(scalars, points):
        acc = g1_add(acc, g1_mul(int(k), p))
    return acc


assert g1_add(G1, G1) == (1368015179489954701390400359078579693043519447331113978918064868415326638035,
                          9918110051302171585080402603319702774565515993150576347155970296011118125764)   # EIP-196's 2 G


# ---- G2: y^2 = x^3 + 3 / (9 + u) over Fq2 = Fq[u] / (u^2 + 1), EIP-197's generator ----
def f2_add(a, b):
    return (a[0] + b[0]) % Q, (a[1] + b[1]) % Q


def f2_sub(a, b):
    return (a[0] - b[0]) % Q, (a[1] - b[1]) % Q


def f2_mul(a, b):
    return (a[0] * b[0] - a[1] * b[1]) % Q, (a[0] * b[1] + a[1] * b[0]) % Q


def f2_inv(a):
    d = pow(a[0] * a[0] + a[1] * a[1], Q - 2, Q)
    return a[0] * d % Q, (-a[1]) * d % Q


B2 = f2_mul((3, 0), f2_inv((9, 1)))
G2 = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
       11559732032986387107991004021392285783925812861821192530917403151452391805634),
      (8495653923123431417604973247489272438418190587263600148770280649306958101930,
       4082367875863433681332203403145435568316851327593401208105741076214120093531))
assert f2_mul(G2[1], G2[1]) == f2_add(f2_mul(f2_mul(G2[0], G2[0]), G2[0]), B2)     # the recalled generator is on the twist


def g2_add(p, q):
    """affine addition over Fq2; None = the point at infinity"""
    if p is None:
        return q
    if q is None:
        return p
    (x1, y1), (x2, y2) = p, q
    if x1 == x2:
        if f2_add(y1, y2) == (0, 0):
            return None
        lam = f2_mul(f2_mul((3, 0), f2_mul(x1, x1)), f2_inv(f2_add(y1, y1)))
    else:
        lam = f2_mul(f2_sub(y2, y1), f2_inv(f2_sub(x2, x1)))
    x3 = f2_sub(f2_sub(f2_mul(lam, lam), x1), x2)
    return x3, f2_sub(f2_mul(lam, f2_sub(x1, x3)), y1)


def g2_neg(p):
    return None if p is None else (p[0], ((-p[1][0]) % Q, (-p[1][1]) % Q))


def g2_mul(k, p):
    k %= R
    acc = None
    for bit in range(k.bit_length() - 1, -1, -1):
        acc = g2_add(acc, acc)
        if (k >> bit) & 1:
            acc = g2_add(acc, p)
    return acc


def msm_g2(scalars, points):
    acc = None
    for k, p in zip(scalars, points):
        acc = g2_add(acc, g2_mul(int(k), p))
    return acc
