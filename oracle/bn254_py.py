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


# ---- row f.4, third piece: the PLONK quotient chain and a KZG opening (published protocol, no blinding) ----
def coset_evals(values_on_h, shift, blowup_log=2):
    """values on H = <w_n> -> values on shift * <w_(n << blowup_log)> of the interpolating polynomial"""
    c = ntt(values_on_h, inverse=True)
    n4 = len(c) << blowup_log
    s, out = 1, []
    for j in range(n4):
        out.append((c[j] * s % R) if j < len(c) else 0)
        s = s * shift % R
    return ntt(out)


def plonk_quotient(p, shift, k1, k2, alpha, beta, gamma):
    """p: dict of the thirteen polynomials' values on H (ql qr qm qo qk s1 s2 s3 l r o z, optional pi).  Returns the 4n
    coefficients of t = (gate + alpha perm + alpha^2 L1 (z - 1)) / Z_H computed on the coset shift * <w_4n> (the top n are zero
    exactly for a satisfying witness)."""
    n = len(p["l"])
    log_n = n.bit_length() - 1
    n4 = 4 * n
    ev = {k: coset_evals(v, shift) for k, v in p.items() if v is not None}
    w4 = root_of_unity(log_n + 2)
    ninv = pow(n, R - 2, R)
    t, x = [], shift
    for i in range(n4):
        l, r, o, z, zn = ev["l"][i], ev["r"][i], ev["o"][i], ev["z"][i], ev["z"][(i + 4) % n4]
        gate = (ev["ql"][i] * l + ev["qr"][i] * r + ev["qm"][i] * l * r + ev["qo"][i] * o + ev["qk"][i]) % R
        if "pi" in ev:
            gate = (gate + ev["pi"][i]) % R
        f = (l + beta * x + gamma) * (r + beta * k1 * x + gamma) % R * (o + beta * k2 * x + gamma) % R * z % R
        g = (l + beta * ev["s1"][i] + gamma) * (r + beta * ev["s2"][i] + gamma) % R * (o + beta * ev["s3"][i] + gamma) % R * zn % R
        zh = (pow(x, n, R) - 1) % R
        l1 = zh * pow(n * (x - 1) % R, R - 2, R) % R
        num = (gate + alpha * (f - g) + alpha * alpha % R * l1 % R * (z - 1)) % R
        t.append(num * pow(zh, R - 2, R) % R)
        x = x * w4 % R
    c = ntt(t, inverse=True)
    sinv, s, out = pow(shift, R - 2, R), 1, []
    for j in range(n4):
        out.append(c[j] * s % R)
        s = s * sinv % R
    return out


def kzg_open(coeffs, zeta):
    """(p(zeta), coefficients of (p(X) - p(zeta)) / (X - zeta)) by synthetic division"""
    h, q = 0, []
    for c in reversed(coeffs):
        h = (int(c) + zeta * h) % R
        q.append(h)
    y = q.pop()
    return y, q[::-1]


def plonk_witness(log_n, rng, k1, k2, beta, gamma, satisfied=True):
    """A random satisfying three-wire PLONK instance on n = 2^log_n gates: selectors, wires with copy constraints among
    equal values, the permutation as s1 s2 s3 (identity points w^i, k1 w^i, k2 w^i) and its grand product z."""
    n = 1 << log_n
    w = root_of_unity(log_n)
    pool = [rng.randrange(R) for _ in range(max(2, n // 2))]
    wires = [[pool[rng.randrange(len(pool))] for _ in range(n)] for _ in range(3)]
    ql, qr, qm, qo = ([rng.randrange(R) for _ in range(n)] for _ in range(4))
    qk = [(-(ql[i] * wires[0][i] + qr[i] * wires[1][i] + qm[i] * wires[0][i] * wires[1][i] + qo[i] * wires[2][i])) % R for i in range(n)]
    ident = [[k * pow(w, i, R) % R for i in range(n)] for k in (1, k1, k2)]
    # one cycle per value: every position of a value maps to the next position holding it
    where = {}
    for c in range(3):
        for i in range(n):
            where.setdefault(wires[c][i], []).append((c, i))
    sigma = [[0] * n for _ in range(3)]
    for pos in where.values():
        for a, b in zip(pos, pos[1:] + pos[:1]):
            sigma[a[0]][a[1]] = ident[b[0]][b[1]]
    z, acc = [], 1
    for i in range(n):
        z.append(acc)
        num = den = 1
        for c in range(3):
            num = num * (wires[c][i] + beta * ident[c][i] + gamma) % R
            den = den * (wires[c][i] + beta * sigma[c][i] + gamma) % R
        acc = acc * num % R * pow(den, R - 2, R) % R
    assert acc == 1, "the grand product closes"
    if not satisfied:
        wires[2][rng.randrange(n)] += 1
    return {"ql": ql, "qr": qr, "qm": qm, "qo": qo, "qk": qk, "s1": sigma[0], "s2": sigma[1], "s3": sigma[2],
            "l": wires[0], "r": wires[1], "o": [x % R for x in wires[2]], "z": z}
