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


# ---- a whole (unblinded) PLONK proof over the pieces above: model prover and a trapdoor verifier ----
import hashlib as _hashlib


class PlonkTranscript:
    """SHA-256 chain: absorb 32-byte big-endian integers; a challenge is the chain value read as an integer mod r.  This
    repo's own convention (gnark's fiat-shamir labels and encodings are not reproduced): model and device prover share it."""

    def __init__(self, label=b"nlx-plonk-bn254"):
        self.state = _hashlib.sha256(label).digest()

    def absorb_int(self, x):
        self.state = _hashlib.sha256(self.state + int(x).to_bytes(32, "big")).digest()

    def absorb_point(self, p):
        x, y = (0, 0) if p is None else p
        self.absorb_int(x)
        self.absorb_int(y)

    def challenge(self, label):
        self.state = _hashlib.sha256(self.state + label).digest()
        return int.from_bytes(self.state, "big") % R


def kzg_srs(tau, size):
    """[tau^i] G1, i < size (a test SRS whose trapdoor the test keeps)"""
    out, t = [], 1
    for _ in range(size):
        out.append(g1_mul(t, G1))
        t = t * tau % R
    return out


def plonk_prove_model(p, srs, k1, k2, public_inputs=()):
    """p: plonk_witness()-style dict WITHOUT z (values on H).  Returns the proof as a dict of points / integers."""
    n = len(p["l"])
    log_n = n.bit_length() - 1
    w = root_of_unity(log_n)
    co = {k: ntt(v, inverse=True) for k, v in p.items() if k != "z" and v is not None}
    com = lambda c: msm_g1(c, srs[:len(c)])
    tr = PlonkTranscript()
    tr.absorb_int(n)
    for x in public_inputs:
        tr.absorb_int(x)
    for k in ("ql", "qr", "qm", "qo", "qk", "s1", "s2", "s3"):
        tr.absorb_point(com(co[k]))
    proof = {"a": com(co["l"]), "b": com(co["r"]), "c": com(co["o"])}
    for k in "abc":
        tr.absorb_point(proof[k])
    beta, gamma = tr.challenge(b"beta"), tr.challenge(b"gamma")
    z, acc = [], 1
    for i in range(n):
        z.append(acc)
        x = pow(w, i, R)
        num = (p["l"][i] + beta * x + gamma) * (p["r"][i] + beta * k1 * x + gamma) * (p["o"][i] + beta * k2 * x + gamma) % R
        den = (p["l"][i] + beta * p["s1"][i] + gamma) * (p["r"][i] + beta * p["s2"][i] + gamma) * (p["o"][i] + beta * p["s3"][i] + gamma) % R
        acc = acc * num % R * pow(den, R - 2, R) % R
    co["z"] = ntt(z, inverse=True)
    proof["z"] = com(co["z"])
    tr.absorb_point(proof["z"])
    alpha = tr.challenge(b"alpha")
    t = plonk_quotient(dict(p, z=z), k1, k1, k2, alpha, beta, gamma)
    assert not any(t[3 * n:]), "the witness does not satisfy the circuit"
    chunks = [t[0:n], t[n:2 * n], t[2 * n:3 * n]]
    for name, c in zip(("t_lo", "t_mid", "t_hi"), chunks):
        proof[name] = com(c)
        tr.absorb_point(proof[name])
    zeta = tr.challenge(b"zeta")
    ev = {"a": eval_poly(co["l"], zeta), "b": eval_poly(co["r"], zeta), "c": eval_poly(co["o"], zeta),
          "s1": eval_poly(co["s1"], zeta), "s2": eval_poly(co["s2"], zeta), "zw": eval_poly(co["z"], zeta * w % R)}
    for k in ("a", "b", "c", "s1", "s2", "zw"):
        tr.absorb_int(ev[k])
    v = tr.challenge(b"v")
    terms = plonk_linearisation_scalars(ev, n, zeta, alpha, beta, gamma, k1, k2, v)
    polys = {"qm": co["qm"], "ql": co["ql"], "qr": co["qr"], "qo": co["qo"], "qk": co["qk"], "z": co["z"], "s3": co["s3"],
             "t_lo": chunks[0], "t_mid": chunks[1], "t_hi": chunks[2], "a": co["l"], "b": co["r"], "c": co["o"], "s1": co["s1"], "s2": co["s2"]}
    f = [sum(terms[k] * polys[k][i] for k in terms) % R for i in range(n)]
    _, wz = kzg_open(f, zeta)
    _, wzw = kzg_open(co["z"], zeta * w % R)
    proof["w_zeta"], proof["w_zeta_omega"] = com(wz), com(wzw)
    proof["evals"] = ev
    return proof


def plonk_linearisation_scalars(ev, n, zeta, alpha, beta, gamma, k1, k2, v):
    """coefficients of the polynomials in F(X) = r(X) + v a + v^2 b + v^3 c + v^4 s1 + v^5 s2 (the batched opening at zeta)"""
    zh = (pow(zeta, n, R) - 1) % R
    l1 = zh * pow(n * (zeta - 1) % R, R - 2, R) % R
    a, b, c, s1, s2, zw = (ev[k] for k in ("a", "b", "c", "s1", "s2", "zw"))
    zn = pow(zeta, n, R)
    return {"qm": a * b % R, "ql": a, "qr": b, "qo": c, "qk": 1,
            "z": (alpha * (a + beta * zeta + gamma) % R * (b + beta * k1 * zeta + gamma) % R * (c + beta * k2 * zeta + gamma) + alpha * alpha % R * l1) % R,
            "s3": (-alpha * (a + beta * s1 + gamma) % R * (b + beta * s2 + gamma) % R * beta % R * zw) % R,
            "t_lo": (-zh) % R, "t_mid": (-zh * zn) % R, "t_hi": (-zh * zn % R * zn) % R,
            "a": v, "b": v * v % R, "c": pow(v, 3, R), "s1": pow(v, 4, R), "s2": pow(v, 5, R)}


def plonk_verify_trapdoor(proof, vk, tau, k1, k2, public_inputs=(), pi_at=None):
    """The PLONK verifier's equations with the pairing replaced by the SRS's trapdoor (a test SRS: tau is known):
    [F] - E G = (tau - zeta) [W_zeta] and [z] - zw G = (tau - zeta w) [W_zeta_omega].  vk: commitments of the eight
    preprocessed polynomials and n.  pi_at(zeta) = the public-input polynomial's value (0 without one)."""
    n = vk["n"]
    w = root_of_unity(n.bit_length() - 1)
    tr = PlonkTranscript()
    tr.absorb_int(n)
    for x in public_inputs:
        tr.absorb_int(x)
    for k in ("ql", "qr", "qm", "qo", "qk", "s1", "s2", "s3"):
        tr.absorb_point(vk[k])
    for k in "abc":
        tr.absorb_point(proof[k])
    beta, gamma = tr.challenge(b"beta"), tr.challenge(b"gamma")
    tr.absorb_point(proof["z"])
    alpha = tr.challenge(b"alpha")
    for k in ("t_lo", "t_mid", "t_hi"):
        tr.absorb_point(proof[k])
    zeta = tr.challenge(b"zeta")
    ev = proof["evals"]
    for k in ("a", "b", "c", "s1", "s2", "zw"):
        tr.absorb_int(ev[k])
    v = tr.challenge(b"v")
    sc = plonk_linearisation_scalars(ev, n, zeta, alpha, beta, gamma, k1, k2, v)
    pts = {"qm": vk["qm"], "ql": vk["ql"], "qr": vk["qr"], "qo": vk["qo"], "qk": vk["qk"], "z": proof["z"], "s3": vk["s3"],
           "t_lo": proof["t_lo"], "t_mid": proof["t_mid"], "t_hi": proof["t_hi"], "a": proof["a"], "b": proof["b"], "c": proof["c"],
           "s1": vk["s1"], "s2": vk["s2"]}
    keys = list(sc)
    F = msm_g1([sc[k] for k in keys], [pts[k] for k in keys])
    zh = (pow(zeta, n, R) - 1) % R
    l1 = zh * pow(n * (zeta - 1) % R, R - 2, R) % R
    a, b, c, s1, s2, zw = (ev[k] for k in ("a", "b", "c", "s1", "s2", "zw"))
    pi = pi_at(zeta) if pi_at else 0
    r0 = (pi - l1 * alpha * alpha - alpha * (a + beta * s1 + gamma) % R * (b + beta * s2 + gamma) % R * (c + gamma) % R * zw) % R
    E = (-r0 + v * a + v * v * b + pow(v, 3, R) * c + pow(v, 4, R) * s1 + pow(v, 5, R) * s2) % R
    lhs1 = g1_add(F, g1_neg(g1_mul(E, G1)))
    rhs1 = g1_mul((tau - zeta) % R, proof["w_zeta"]) if proof["w_zeta"] is not None else None
    lhs2 = g1_add(proof["z"], g1_neg(g1_mul(zw, G1)))
    rhs2 = g1_mul((tau - zeta * w) % R, proof["w_zeta_omega"]) if proof["w_zeta_omega"] is not None else None
    return lhs1 == rhs1 and lhs2 == rhs2


def groth16_quotient(a, b, c, shift=5):
    """h = (A B - C) / Z_H from the three polynomials' values on H, on the coset shift * H of the same size (gnark computeH)"""
    n = len(a)
    ev = [coset_evals(v, shift, blowup_log=0) for v in (a, b, c)]
    zhinv = pow((pow(shift, n, R) - 1) % R, R - 2, R)
    t = [(x * y - z) % R * zhinv % R for x, y, z in zip(*ev)]
    co = ntt(t, inverse=True)
    sinv, s, out = pow(shift, R - 2, R), 1, []
    for j in range(n):
        out.append(co[j] * s % R)
        s = s * sinv % R
    return out


# ---- row f.4, fourth piece: the proof in gnark's SHAPE - its fiat-shamir, its blinding, its batched opening, its bytes ----
# gnark (backend/plonk/bn254 Prove / Verify, gnark-crypto fiat-shamir and kzg; Go, not in /root/reference: succinct.json:7-8
# names the entry points that run it) restated from its published structure, from memory [U]: parity with gnark-produced bytes
# is UNPINNED (no Go toolchain, no vector); what is pinned is that model and device prover emit the same bytes and that the
# verifier below (the pairing replaced by the test SRS's trapdoor) accepts them and rejects tampering.
#   transcript  fiatshamir.NewTranscript(sha256, "gamma", "beta", "alpha", "zeta"): challenge_i = SHA-256(name_i ||
#               challenge_{i-1} (its 32 raw bytes, i > 0) || the bytes bound to name_i), read big-endian mod r;
#               gamma <- the verifying key's S1 S2 S3 Ql Qr Qm Qo Qk, the public inputs, then [L] [R] [O]; beta <- nothing;
#               alpha <- [Z]; zeta <- [H1] [H2] [H3]   (points as x || y, 64 bytes; scalars 32 bytes big-endian)
#   blinding    l, r, o += (b0 + b1 X)(X^n - 1), z += (b0 + b1 X + b2 X^2)(X^n - 1)        (getBlindedPolynomial, orders 1, 1, 1, 2)
#   quotient    h of 3 n + 6 coefficients, committed as h1 h2 h3 of n + 2 coefficients each
#   openings    at zeta, ONE batched opening (kzg.BatchOpenSinglePoint) of [foldedH, linearised, l, r, o, s1, s2] with
#               foldedH = h1 + zeta^(n+2) h2 + zeta^(2(n+2)) h3 and the combiner gamma' = SHA-256("gamma" || zeta || the seven
#               digests || the seven claimed values) mod r; at w zeta, z alone
#   bytes       Proof.WriteTo: LRO[3], Z, H[3] as compressed points (32 bytes, flags in the top two bits), the (empty) list of
#               Bsb22 commitments (u32 count), BatchedProof.H, its claimed values (u32 count, 32 bytes each), ZShiftedOpening.H
#               and its claimed value
Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583


def fr_bytes(x):
    return (int(x) % R).to_bytes(32, "big")


def g1_marshal(p):
    """G1Affine.Marshal(): x || y big-endian; the point at infinity is the flag byte 0x40 and zeros"""
    if p is None:
        return bytes([0x40]) + bytes(63)
    return int(p[0]).to_bytes(32, "big") + int(p[1]).to_bytes(32, "big")


def g1_compress(p):
    """G1Affine.Bytes(): x big-endian with 0b10 (y the smaller of y, q - y) or 0b11 (the larger) in the top two bits; 0b01 = infinity"""
    if p is None:
        return bytes([0x40]) + bytes(31)
    b = bytearray(int(p[0]).to_bytes(32, "big"))
    b[0] |= 0xC0 if int(p[1]) > (Q - 1) // 2 else 0x80
    return bytes(b)


def g1_decompress(b):
    flag, x = b[0] >> 6, int.from_bytes(bytes([b[0] & 0x3F]) + b[1:], "big")
    if flag == 1:
        return None
    y = pow((x * x * x + 3) % Q, (Q + 1) // 4, Q)
    assert y * y % Q == (x * x * x + 3) % Q, "not on the curve"
    if (y > (Q - 1) // 2) != (flag == 3):
        y = Q - y
    return (x, y)


class GnarkTranscript:
    def __init__(self, *names):
        self.names, self.bound, self.value = list(names), {k: [] for k in names}, {}

    def bind(self, name, data):
        assert name not in self.value
        self.bound[name].append(bytes(data))

    def challenge(self, name):
        i = self.names.index(name)
        h = _hashlib.sha256(name.encode())
        if i:
            h.update(self.value[self.names[i - 1]])
        for b in self.bound[name]:
            h.update(b)
        self.value[name] = h.digest()
        return int.from_bytes(self.value[name], "big") % R


def blind_coeffs(coeffs, n, b):
    """coeffs (n of them) + (b[0] + b[1] X + ...)(X^n - 1)"""
    out = [int(c) for c in coeffs] + [0] * len(b)
    for i, bi in enumerate(b):
        out[i] = (out[i] - bi) % R
        out[n + i] = (out[n + i] + bi) % R
    return out


def _coset_evals_of_coeffs(c, n4, shift):
    s, out = 1, []
    for j in range(n4):
        out.append((int(c[j]) * s % R) if j < len(c) else 0)
        s = s * shift % R
    return ntt(out)


def gnark_quotient(co, n, shift, k1, k2, alpha, beta, gamma):
    """co: coefficient lists (l r o z blinded: n + 2 / n + 3 long) -> the 4 n coefficients of h on the coset shift <w_4n>"""
    n4, log_n = 4 * n, n.bit_length() - 1
    ev = {k: _coset_evals_of_coeffs(v, n4, shift) for k, v in co.items()}
    w4 = root_of_unity(log_n + 2)
    t, x = [], shift
    for i in range(n4):
        l, r, o, z, zn = ev["l"][i], ev["r"][i], ev["o"][i], ev["z"][i], ev["z"][(i + 4) % n4]
        gate = (ev["ql"][i] * l + ev["qr"][i] * r + ev["qm"][i] * l * r + ev["qo"][i] * o + ev["qk"][i] + (ev["pi"][i] if "pi" in ev else 0)) % R
        f = (l + beta * x + gamma) * (r + beta * k1 * x + gamma) % R * (o + beta * k2 * x + gamma) % R * z % R
        g = (l + beta * ev["s1"][i] + gamma) * (r + beta * ev["s2"][i] + gamma) % R * (o + beta * ev["s3"][i] + gamma) % R * zn % R
        zh = (pow(x, n, R) - 1) % R
        l1 = zh * pow(n * (x - 1) % R, R - 2, R) % R
        t.append((gate + alpha * (f - g) + alpha * alpha % R * l1 % R * (z - 1)) % R * pow(zh, R - 2, R) % R)
        x = x * w4 % R
    c = ntt(t, inverse=True)
    sinv, s, out = pow(shift, R - 2, R), 1, []
    for j in range(n4):
        out.append(c[j] * s % R)
        s = s * sinv % R
    return out


def _bind_vk(fs, vk, public_inputs):
    for k in ("s1", "s2", "s3", "ql", "qr", "qm", "qo", "qk"):
        fs.bind("gamma", g1_marshal(vk[k]))
    for x in public_inputs:
        fs.bind("gamma", fr_bytes(x))


def _batch_gamma(zeta, digests, claimed):
    fs = GnarkTranscript("gamma")
    fs.bind("gamma", fr_bytes(zeta))
    for d in digests:
        fs.bind("gamma", g1_marshal(d))
    for v in claimed:
        fs.bind("gamma", fr_bytes(v))
    return fs.challenge("gamma")


def _lin_scalars(l, r, o, s1, s2, zw, n, zeta, alpha, beta, gamma, k1, k2):
    """the linearised polynomial's coefficients on qm ql qr qo qk z s3"""
    zh = (pow(zeta, n, R) - 1) % R
    l1 = zh * pow(n * (zeta - 1) % R, R - 2, R) % R
    a_ = (l + beta * zeta + gamma) * (r + beta * k1 * zeta + gamma) % R * (o + beta * k2 * zeta + gamma) % R
    b_ = (l + beta * s1 + gamma) * (r + beta * s2 + gamma) % R
    return {"qm": l * r % R, "ql": l, "qr": r, "qo": o, "qk": 1, "z": (alpha * a_ + alpha * alpha % R * l1) % R,
            "s3": (-alpha * b_ % R * beta % R * zw) % R}, l1, b_


def gnark_plonk_prove_model(p, srs, k1, k2, public_inputs=(), blinding=(0,) * 9):
    """p: plonk_witness()-style values on H WITHOUT z; public_inputs: the values of the public-input polynomial on the first
    points of H (added to the gate's constant); blinding: b for l (2), r (2), o (2), z (3).  Returns (proof dict, bytes)."""
    n = len(p["l"])
    log_n = n.bit_length() - 1
    w, u = root_of_unity(log_n), k1
    com = lambda c: msm_g1(c, srs[:len(c)])
    co = {k: ntt(p[k], inverse=True) for k in ("ql", "qr", "qm", "qo", "qk", "s1", "s2", "s3", "l", "r", "o")}
    vk = {k: com(co[k]) for k in ("ql", "qr", "qm", "qo", "qk", "s1", "s2", "s3")}
    pi_vals = [int(x) % R for x in public_inputs] + [0] * (n - len(public_inputs))
    if public_inputs:
        co["pi"] = ntt(pi_vals, inverse=True)
    fs = GnarkTranscript("gamma", "beta", "alpha", "zeta")
    _bind_vk(fs, vk, public_inputs)
    b = [int(x) % R for x in blinding]
    bl = {"l": blind_coeffs(co["l"], n, b[0:2]), "r": blind_coeffs(co["r"], n, b[2:4]), "o": blind_coeffs(co["o"], n, b[4:6])}
    proof = {"lro": [com(bl["l"]), com(bl["r"]), com(bl["o"])]}
    for c in proof["lro"]:
        fs.bind("gamma", g1_marshal(c))
    gamma = fs.challenge("gamma")
    beta = fs.challenge("beta")
    z, acc = [], 1
    for i in range(n):
        z.append(acc)
        x = pow(w, i, R)
        num = (p["l"][i] + beta * x + gamma) * (p["r"][i] + beta * k1 * x + gamma) * (p["o"][i] + beta * k2 * x + gamma) % R
        den = (p["l"][i] + beta * p["s1"][i] + gamma) * (p["r"][i] + beta * p["s2"][i] + gamma) * (p["o"][i] + beta * p["s3"][i] + gamma) % R
        acc = acc * num % R * pow(den, R - 2, R) % R
    assert acc == 1, "the wires do not respect the copy constraints"
    bl["z"] = blind_coeffs(ntt(z, inverse=True), n, b[6:9])
    proof["z"] = com(bl["z"])
    fs.bind("alpha", g1_marshal(proof["z"]))
    alpha = fs.challenge("alpha")
    qco = dict(co)
    qco.update(bl)
    h = gnark_quotient(qco, n, u, k1, k2, alpha, beta, gamma)
    assert not any(h[3 * n + 6:]), "the witness does not satisfy the circuit"
    hs = [h[0:n + 2], h[n + 2:2 * n + 4], h[2 * n + 4:3 * n + 6]]
    proof["h"] = [com(c) for c in hs]
    for c in proof["h"]:
        fs.bind("zeta", g1_marshal(c))
    zeta = fs.challenge("zeta")
    ev = {k: eval_poly(bl[k], zeta) for k in ("l", "r", "o")}
    ev["s1"], ev["s2"] = eval_poly(co["s1"], zeta), eval_poly(co["s2"], zeta)
    zw, zq = kzg_open(bl["z"], zeta * w % R)
    proof["z_shifted"] = {"h": com(zq), "value": zw}
    sc, _, _ = _lin_scalars(ev["l"], ev["r"], ev["o"], ev["s1"], ev["s2"], zw, n, zeta, alpha, beta, gamma, k1, k2)
    src = {"qm": co["qm"], "ql": co["ql"], "qr": co["qr"], "qo": co["qo"], "qk": co["qk"], "z": bl["z"], "s3": co["s3"]}
    m = n + 3
    lin = [sum(sc[k] * (src[k][i] if i < len(src[k]) else 0) for k in sc) % R for i in range(m)]
    zn2 = pow(zeta, n + 2, R)
    folded_h = [(hs[0][i] + zn2 * hs[1][i] + zn2 * zn2 % R * hs[2][i]) % R for i in range(n + 2)]
    polys = [folded_h, lin, bl["l"], bl["r"], bl["o"], co["s1"], co["s2"]]
    folded_h_digest = g1_add(g1_add(proof["h"][0], g1_mul(zn2, proof["h"][1])), g1_mul(zn2 * zn2 % R, proof["h"][2]))
    digests = [folded_h_digest, com(lin)] + proof["lro"] + [vk["s1"], vk["s2"]]
    claimed = [eval_poly(c, zeta) for c in polys]
    gp = _batch_gamma(zeta, digests, claimed)
    folded, g = [0] * m, 1
    for c in polys:
        for i, v in enumerate(c):
            folded[i] = (folded[i] + g * int(v)) % R
        g = g * gp % R
    _, q = kzg_open(folded, zeta)
    proof["batched"] = {"h": com(q), "values": claimed}
    return proof, gnark_proof_bytes(proof)


def gnark_proof_bytes(proof):
    out = b"".join(g1_compress(c) for c in proof["lro"]) + g1_compress(proof["z"]) + b"".join(g1_compress(c) for c in proof["h"])
    out += (0).to_bytes(4, "big")                                     # Bsb22Commitments: none
    out += g1_compress(proof["batched"]["h"]) + len(proof["batched"]["values"]).to_bytes(4, "big")
    out += b"".join(fr_bytes(v) for v in proof["batched"]["values"])
    return out + g1_compress(proof["z_shifted"]["h"]) + fr_bytes(proof["z_shifted"]["value"])


def gnark_proof_from_bytes(data):
    pts = [g1_decompress(data[32 * i:32 * i + 32]) for i in range(7)]
    off = 224
    assert int.from_bytes(data[off:off + 4], "big") == 0
    off += 4
    bh = g1_decompress(data[off:off + 32])
    off += 32
    k = int.from_bytes(data[off:off + 4], "big")
    off += 4
    vals = [int.from_bytes(data[off + 32 * i:off + 32 * i + 32], "big") for i in range(k)]
    off += 32 * k
    zh = g1_decompress(data[off:off + 32])
    zv = int.from_bytes(data[off + 32:off + 64], "big")
    assert off + 64 == len(data) and all(v < R for v in vals + [zv])
    return {"lro": pts[0:3], "z": pts[3], "h": pts[4:7], "batched": {"h": bh, "values": vals}, "z_shifted": {"h": zh, "value": zv}}


def gnark_plonk_verify_trapdoor(data, vk, n, tau, k1, k2, public_inputs=()):
    """gnark's Verify on the proof BYTES with the pairing replaced by the test SRS's trapdoor tau.  vk: the eight commitments."""
    proof = gnark_proof_from_bytes(data)
    w = root_of_unity(n.bit_length() - 1)
    fs = GnarkTranscript("gamma", "beta", "alpha", "zeta")
    _bind_vk(fs, vk, public_inputs)
    for c in proof["lro"]:
        fs.bind("gamma", g1_marshal(c))
    gamma, beta = fs.challenge("gamma"), fs.challenge("beta")
    fs.bind("alpha", g1_marshal(proof["z"]))
    alpha = fs.challenge("alpha")
    for c in proof["h"]:
        fs.bind("zeta", g1_marshal(c))
    zeta = fs.challenge("zeta")
    vals = proof["batched"]["values"]
    if len(vals) != 7:
        return False
    folded_h_zeta, lin_zeta, l, r, o, s1, s2 = vals
    zw = proof["z_shifted"]["value"]
    sc, l1, b_ = _lin_scalars(l, r, o, s1, s2, zw, n, zeta, alpha, beta, gamma, k1, k2)
    zh = (pow(zeta, n, R) - 1) % R
    # PI(zeta) = sum_i PI_i L_i(zeta), L_i(zeta) = w^i Z_H(zeta) / (n (zeta - w^i))
    pi = 0
    for i, x in enumerate(public_inputs):
        wi = pow(w, i, R)
        pi = (pi + int(x) * wi % R * zh % R * pow(n * (zeta - wi) % R, R - 2, R)) % R
    # the identity at zeta: linearised + PI - alpha (l + beta s1 + gamma)(r + beta s2 + gamma)(o + gamma) z(w zeta) - alpha^2 L1 = foldedH Z_H
    if (lin_zeta + pi - alpha * b_ % R * (o + gamma) % R * zw - alpha * alpha % R * l1) % R != folded_h_zeta * zh % R:
        return False
    pts = {"qm": vk["qm"], "ql": vk["ql"], "qr": vk["qr"], "qo": vk["qo"], "qk": vk["qk"], "z": proof["z"], "s3": vk["s3"]}
    keys = list(sc)
    lin_digest = msm_g1([sc[k] for k in keys], [pts[k] for k in keys])
    zn2 = pow(zeta, n + 2, R)
    folded_h_digest = g1_add(g1_add(proof["h"][0], g1_mul(zn2, proof["h"][1])), g1_mul(zn2 * zn2 % R, proof["h"][2]))
    digests = [folded_h_digest, lin_digest] + proof["lro"] + [vk["s1"], vk["s2"]]
    gp = _batch_gamma(zeta, digests, vals)
    acc, g = None, 1
    for d, v in zip(digests, vals):
        term = g1_add(d, g1_neg(g1_mul(v, G1)))
        acc = g1_add(acc, g1_mul(g, term)) if g != 0 else acc
        g = g * gp % R
    rhs = g1_mul((tau - zeta) % R, proof["batched"]["h"]) if proof["batched"]["h"] is not None else None
    if acc != rhs:
        return False
    lhs2 = g1_add(proof["z"], g1_neg(g1_mul(zw, G1)))
    rhs2 = g1_mul((tau - zeta * w) % R, proof["z_shifted"]["h"]) if proof["z_shifted"]["h"] is not None else None
    return lhs2 == rhs2
