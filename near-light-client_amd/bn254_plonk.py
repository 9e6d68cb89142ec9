"""Row f.4: a whole (unblinded) PLONK proof over BN254 on the GPU, assembled from the C-ABI pieces - the host-side mirror of
what gnark's backend/plonk/bn254 `Prove` orchestrates for the recursive wrap (Go, not in /root/reference; succinct.json:7-8
names the entry point that runs it).  The five rounds of the published protocol (Gabizon-Williamson-Ciobotaru), each round's
heavy step one library call on device-resident polynomials:

  1. [a] [b] [c]                      FFTInverse (nlx_bn254_ntt_batch) + three MSMs (nlx_bn254_msm_g1)
  2. beta, gamma -> z, [z]            nlx_bn254_plonk_grand_product + FFTInverse + MSM
  3. alpha -> t, [t_lo] [t_mid] [t_hi]  nlx_bn254_plonk_quotient + three MSMs
  4. zeta -> six evaluations          nlx_bn254_kzg_open (value only)
  5. v -> [W_zeta], [W_zeta_omega]    nlx_bn254_fr_lincomb (the linearisation polynomial and the batch in ONE combination of
                                      fifteen polynomials), nlx_bn254_kzg_open twice (synthetic division + MSM)

What this is not: gnark's byte format, blinding, its fiat-shamir labels (the transcript below is this repo's own SHA-256
chain) or the wrapper circuit - a maintainer wires the same calls into gnark's rounds.  Parity: the
proof's nine points and six scalars equal the big-integer model's (oracle/bn254_py.py plonk_prove_model) and the model's
verifier accepts them (tests/test_gpu_bn254_plonk.py)."""
import hashlib

import numpy as np

from . import batch as B

R = B.BN254_R
_MONT = (1 << 256) % R
_MONT_INV = pow(_MONT, R - 2, R)


def _to_mont(x):
    return int(x) * _MONT % R


def _from_words_mont(w):
    return sum(int(w[k]) << (64 * k) for k in range(4)) * _MONT_INV % R


def root_of_unity(log_n):
    return pow(pow(5, (R - 1) >> 28, R), 1 << (28 - log_n), R)


class Transcript:
    """SHA-256 chain over 32-byte big-endian integers; a challenge is the chain value mod r (this repo's convention)."""

    def __init__(self, label=b"nlx-plonk-bn254"):
        self.state = hashlib.sha256(label).digest()

    def absorb_int(self, x):
        self.state = hashlib.sha256(self.state + int(x).to_bytes(32, "big")).digest()

    def absorb_point(self, words):
        pt = B.bn254_g1_unpack(words)
        x, y = (0, 0) if pt is None else pt
        self.absorb_int(x)
        self.absorb_int(y)

    def challenge(self, label):
        self.state = hashlib.sha256(self.state + label).digest()
        return int.from_bytes(self.state, "big") % R


class ProvingKey:
    """The preprocessed circuit on the device: the eight fixed polynomials by values on H and by coefficients, their
    commitments, the SRS.  Everything is fr.Element / G1Affine words (Montgomery) in torch int64 tensors."""

    NAMES = ("ql", "qr", "qm", "qo", "qk", "s1", "s2", "s3")

    def __init__(self, ctx, values, srs, k1, k2, device="cuda:0"):
        """values: dict name -> n integers (canonical, values on H); srs: (>= n, 8) G1Affine words (array or device tensor)"""
        import torch
        self.ctx, self.device = ctx, device
        self.n = len(values["ql"])
        self.log_n = self.n.bit_length() - 1
        self.k1, self.k2 = int(k1), int(k2)
        self.srs = srs if hasattr(srs, "data_ptr") else torch.from_numpy(np.ascontiguousarray(srs, dtype=np.uint64).view(np.int64)).to(device)
        packed = B.bn254_pack([[_to_mont(x) for x in values[k]] for k in self.NAMES])
        self.values = torch.from_numpy(packed.view(np.int64)).to(device)                      # (8, n, 4)
        self.coeffs = B.bn254_ntt(ctx, self.values.clone(), inverse=True, montgomery=True)     # in place on the clone
        self.commitments = {k: B.bn254_msm_g1(ctx, self.srs[:self.n], self.coeffs[i], montgomery=True) for i, k in enumerate(self.NAMES)}

    def value(self, name):
        return self.values[self.NAMES.index(name)]

    def coeff(self, name):
        return self.coeffs[self.NAMES.index(name)]


def prove(pk, l, r, o, public_inputs=()):
    """l, r, o: the wire values on H - n canonical integers each, or (n, 4) device tensors of fr.Element words (Montgomery).
    Returns the proof: nine G1Affine word arrays and six integers, under the keys of the model's proof dict.
    This is the UNBLINDED paper-shaped proof under this repo's own transcript; it has no public-input polynomial - a statement
    with public inputs is proved with prove_gnark (gnark's transcript, blinding, batched opening and byte format)."""
    import torch
    if len(public_inputs):
        raise ValueError("prove() constrains no public inputs (it has no public-input polynomial): use prove_gnark")
    ctx, n, log_n = pk.ctx, pk.n, pk.log_n
    w = root_of_unity(log_n)
    commit = lambda c: B.bn254_msm_g1(ctx, pk.srs[:c.shape[0]], c, montgomery=True)
    tr = Transcript()
    tr.absorb_int(n)
    for x in public_inputs:
        tr.absorb_int(x)
    for k in pk.NAMES:
        tr.absorb_point(pk.commitments[k])
    # round 1
    if hasattr(l, "data_ptr"):   # already fr.Element words on the device: (n, 4) tensors
        wires = torch.stack([l, r, o])
    else:
        wires = torch.from_numpy(B.bn254_pack([[_to_mont(x) for x in col] for col in (l, r, o)]).view(np.int64)).to(pk.device)   # (3, n, 4)
    wire_coeffs = B.bn254_ntt(ctx, wires.clone(), inverse=True, montgomery=True)
    proof = {"a": commit(wire_coeffs[0]), "b": commit(wire_coeffs[1]), "c": commit(wire_coeffs[2])}
    for k in "abc":
        tr.absorb_point(proof[k])
    beta, gamma = tr.challenge(b"beta"), tr.challenge(b"gamma")
    # round 2
    z = torch.empty((n, 4), dtype=torch.int64, device=pk.device)
    if not grand_product(ctx, log_n, wires[0], wires[1], wires[2], pk.value("s1"), pk.value("s2"), pk.value("s3"), beta, gamma, pk.k1, pk.k2, z):
        raise ValueError("the wires do not respect the circuit's copy constraints (the grand product does not close)")
    z_coeffs = B.bn254_ntt(ctx, z.clone().reshape(1, n, 4), inverse=True, montgomery=True)[0]
    proof["z"] = commit(z_coeffs)
    tr.absorb_point(proof["z"])
    alpha = tr.challenge(b"alpha")
    # round 3
    polys = {k: pk.value(k) for k in pk.NAMES}
    polys.update(l=wires[0], r=wires[1], o=wires[2], z=z)
    t = torch.empty((3, n, 4), dtype=torch.int64, device=pk.device)
    _, ok = B.bn254_plonk_quotient(ctx, polys, *[_to_mont(x) for x in (pk.k1, pk.k1, pk.k2, alpha, beta, gamma)], out=t)
    if not ok:
        raise ValueError("the witness does not satisfy the circuit (the quotient has a fourth chunk)")
    for i, name in enumerate(("t_lo", "t_mid", "t_hi")):
        proof[name] = commit(t[i])
        tr.absorb_point(proof[name])
    zeta = tr.challenge(b"zeta")
    # round 4
    at = lambda c, point: _from_words_mont(B.bn254_kzg_open(ctx, c, _to_mont(point), want_quotient=False)[0])
    ev = {"a": at(wire_coeffs[0], zeta), "b": at(wire_coeffs[1], zeta), "c": at(wire_coeffs[2], zeta),
          "s1": at(pk.coeff("s1"), zeta), "s2": at(pk.coeff("s2"), zeta), "zw": at(z_coeffs, zeta * w % R)}
    for k in ("a", "b", "c", "s1", "s2", "zw"):
        tr.absorb_int(ev[k])
    v = tr.challenge(b"v")
    # round 5: F = r + v a + v^2 b + v^3 c + v^4 s1 + v^5 s2 as ONE combination; its opening at zeta is W_zeta
    sc = linearisation_scalars(ev, n, zeta, alpha, beta, gamma, pk.k1, pk.k2, v)
    terms = {"qm": pk.coeff("qm"), "ql": pk.coeff("ql"), "qr": pk.coeff("qr"), "qo": pk.coeff("qo"), "qk": pk.coeff("qk"), "z": z_coeffs,
             "s3": pk.coeff("s3"), "t_lo": t[0], "t_mid": t[1], "t_hi": t[2], "a": wire_coeffs[0], "b": wire_coeffs[1], "c": wire_coeffs[2],
             "s1": pk.coeff("s1"), "s2": pk.coeff("s2")}
    f = torch.empty((n, 4), dtype=torch.int64, device=pk.device)
    lincomb(ctx, [terms[k] for k in sc], [sc[k] for k in sc], f)
    proof["w_zeta"] = B.bn254_kzg_open(ctx, f, _to_mont(zeta), srs=pk.srs, want_quotient=False)[2]
    proof["w_zeta_omega"] = B.bn254_kzg_open(ctx, z_coeffs, _to_mont(zeta * w % R), srs=pk.srs, want_quotient=False)[2]
    proof["evals"] = ev
    return proof


def linearisation_scalars(ev, n, zeta, alpha, beta, gamma, k1, k2, v):
    """coefficients of the fifteen polynomials in F(X) (the published round 5, constants dropped: they do not reach a quotient)"""
    zh = (pow(zeta, n, R) - 1) % R
    l1 = zh * pow(n * (zeta - 1) % R, R - 2, R) % R
    a, b, c, s1, s2, zw = (ev[k] for k in ("a", "b", "c", "s1", "s2", "zw"))
    zn = pow(zeta, n, R)
    return {"qm": a * b % R, "ql": a, "qr": b, "qo": c, "qk": 1,
            "z": (alpha * (a + beta * zeta + gamma) % R * (b + beta * k1 * zeta + gamma) % R * (c + beta * k2 * zeta + gamma) + alpha * alpha % R * l1) % R,
            "s3": (-alpha * (a + beta * s1 + gamma) % R * (b + beta * s2 + gamma) % R * beta % R * zw) % R,
            "t_lo": (-zh) % R, "t_mid": (-zh * zn) % R, "t_hi": (-zh * zn % R * zn) % R,
            "a": v, "b": v * v % R, "c": pow(v, 3, R), "s1": pow(v, 4, R), "s2": pow(v, 5, R)}


def grand_product(ctx, log_n, l, r, o, s1, s2, s3, beta, gamma, k1, k2, z_out):
    """nlx_bn254_plonk_grand_product on device tensors / host arrays of fr.Element words; scalars: canonical integers.
    Returns whether the product closes."""
    import ctypes
    from ._lib import dll
    keep = [B._fr_words(_to_mont(x)) for x in (beta, gamma, k1, k2)]
    ptr = lambda a: a.data_ptr() if hasattr(a, "data_ptr") else a.ctypes.data
    closes = ctypes.c_int32()
    ctx.check(dll.nlx_bn254_plonk_grand_product(ctx.handle, log_n, ptr(l), ptr(r), ptr(o), ptr(s1), ptr(s2), ptr(s3),
                                                keep[0].ctypes.data, keep[1].ctypes.data, keep[2].ctypes.data, keep[3].ctypes.data,
                                                ptr(z_out), ctypes.byref(closes)))
    return bool(closes.value)


def lincomb(ctx, polys, scalars, out):
    """out = sum_t scalars[t] polys[t] (nlx_bn254_fr_lincomb): polys device tensors / host arrays (m, 4), scalars canonical integers"""
    import ctypes
    from ._lib import dll
    ptr = lambda a: a.data_ptr() if hasattr(a, "data_ptr") else a.ctypes.data
    m = polys[0].shape[0]
    arr = (ctypes.c_void_p * len(polys))(*[ptr(p) for p in polys])
    sc = np.stack([B._fr_words(_to_mont(s)) for s in scalars])
    ctx.check(dll.nlx_bn254_fr_lincomb(ctx.handle, m, len(polys), arr, sc.ctypes.data, ptr(out)))
    return out


def groth16_quotient(ctx, a, b, c, coset_shift=5):
    """nlx_bn254_groth16_quotient: a, b, c = values of A w, B w, C w on H ((n, 4) arrays or device tensors of fr.Element words);
    returns h's n coefficients as a host array (n, 4)."""
    from ._lib import dll
    ptr = lambda x: x.data_ptr() if hasattr(x, "data_ptr") else np.ascontiguousarray(x, dtype=np.uint64).ctypes.data
    n = a.shape[0]
    keep = [x if hasattr(x, "data_ptr") else np.ascontiguousarray(x, dtype=np.uint64) for x in (a, b, c)]
    sh = B._fr_words(_to_mont(coset_shift))
    out = np.zeros((n, 4), dtype=np.uint64)
    ctx.check(dll.nlx_bn254_groth16_quotient(ctx.handle, n.bit_length() - 1, ptr(keep[0]), ptr(keep[1]), ptr(keep[2]), sh.ctypes.data, out.ctypes.data))
    return out


# ---- the proof in gnark's shape: its fiat-shamir, its blinding, its batched opening, its bytes (round 4) ----------------------
# The host-side mirror of gnark backend/plonk/bn254 Prove (Go, not in /root/reference; succinct.json:7-8 names the entry point
# that runs it), restated from its published structure [U: from memory, no gnark-produced vector exists here]; every heavy
# step is one library call on device-resident polynomials.  oracle/bn254_py.py gnark_plonk_prove_model restates the same
# protocol on big integers (tests only); tests/test_gpu_bn254_plonk.py compares the BYTES and runs the model's verifier.
Q = B.BN254_Q


def fr_bytes(x):
    return (int(x) % R).to_bytes(32, "big")


def g1_marshal(words):
    """G1Affine.Marshal(): x || y, big-endian (what the transcript binds)"""
    pt = B.bn254_g1_unpack(words)
    if pt is None:
        return bytes([0x40]) + bytes(63)
    return pt[0].to_bytes(32, "big") + pt[1].to_bytes(32, "big")


def g1_compress(words):
    """G1Affine.Bytes(): x with 0b10 / 0b11 in the top two bits for the smaller / larger y, 0b01 for infinity (Proof.WriteTo)"""
    pt = B.bn254_g1_unpack(words)
    if pt is None:
        return bytes([0x40]) + bytes(31)
    b = bytearray(pt[0].to_bytes(32, "big"))
    b[0] |= 0xC0 if pt[1] > (Q - 1) // 2 else 0x80
    return bytes(b)


class FiatShamir:
    """gnark-crypto fiatshamir.Transcript over SHA-256: named challenges in a fixed order; challenge i hashes its name, the raw
    bytes of challenge i - 1 and whatever was bound to it"""

    def __init__(self, *names):
        self.names, self.bound, self.raw = names, {k: [] for k in names}, {}

    def bind(self, name, data):
        self.bound[name].append(bytes(data))

    def challenge(self, name):
        i = self.names.index(name)
        h = hashlib.sha256(name.encode())
        if i:
            h.update(self.raw[self.names[i - 1]])
        for b in self.bound[name]:
            h.update(b)
        self.raw[name] = h.digest()
        return int.from_bytes(self.raw[name], "big") % R


def _words_to_int(t):
    return sum((int(v) & 0xFFFFFFFFFFFFFFFF) << (64 * k) for k, v in enumerate(t.tolist()))


def _blinded(torch, coeffs, n, b):
    """coefficients (n, 4) on the device + (b[0] + b[1] X + ...)(X^n - 1): a tensor of n + len(b) coefficients; only 2 len(b)
    of them change, patched through the host (fr.Element words are Montgomery residues)"""
    out = torch.zeros((n + len(b), 4), dtype=torch.int64, device=coeffs.device)
    out[:n] = coeffs
    for i, bi in enumerate(b):
        for idx, sign in ((i, -1), (n + i, 1)):
            cur = _words_to_int(out[idx].cpu()) * _MONT_INV % R
            val = _to_mont((cur + sign * bi) % R)
            out[idx] = torch.from_numpy(B._fr_words(val).view(np.int64)).to(coeffs.device)
    return out


def prove_gnark(pk, l, r, o, public_inputs=(), blinding=None):
    """l, r, o: the wire values on H (n canonical integers each).  public_inputs: the values of the public-input polynomial on
    the first points of H (the circuit's qk leaves them out, as gnark's does).  blinding: nine scalars (l 2, r 2, o 2, z 3) -
    random when None; a test passes them to compare bytes with the model.  The SRS must hold n + 3 points.
    Returns the proof's bytes (gnark Proof.WriteTo layout)."""
    import secrets
    import torch
    ctx, n, log_n = pk.ctx, pk.n, pk.log_n
    w, u, dev = root_of_unity(log_n), pk.k1, pk.device
    if pk.srs.shape[0] < n + 3:
        raise ValueError("the SRS must hold n + 3 points (blinded polynomials have up to n + 3 coefficients)")
    b = [secrets.randbelow(R) for _ in range(9)] if blinding is None else [int(x) % R for x in blinding]
    commit = lambda c: B.bn254_msm_g1(ctx, pk.srs[:c.shape[0]], c.contiguous(), montgomery=True)
    at = lambda c, point: _from_words_mont(B.bn254_kzg_open(ctx, c.contiguous(), _to_mont(point), want_quotient=False)[0])
    fs = FiatShamir("gamma", "beta", "alpha", "zeta")
    for k in ("s1", "s2", "s3", "ql", "qr", "qm", "qo", "qk"):
        fs.bind("gamma", g1_marshal(pk.commitments[k]))
    for x in public_inputs:
        fs.bind("gamma", fr_bytes(x))
    # round 1: blinded wires
    wires = torch.from_numpy(B.bn254_pack([[_to_mont(x) for x in col] for col in (l, r, o)]).view(np.int64)).to(dev)   # (3, n, 4)
    wire_coeffs = B.bn254_ntt(ctx, wires.clone(), inverse=True, montgomery=True)
    bl = [_blinded(torch, wire_coeffs[i], n, b[2 * i:2 * i + 2]) for i in range(3)]
    lro = [commit(c) for c in bl]
    for c in lro:
        fs.bind("gamma", g1_marshal(c))
    gamma = fs.challenge("gamma")
    beta = fs.challenge("beta")
    # round 2: the grand product, blinded
    z = torch.empty((n, 4), dtype=torch.int64, device=dev)
    if not grand_product(ctx, log_n, wires[0], wires[1], wires[2], pk.value("s1"), pk.value("s2"), pk.value("s3"), beta, gamma, pk.k1, pk.k2, z):
        raise ValueError("the wires do not respect the circuit's copy constraints (the grand product does not close)")
    blz = _blinded(torch, B.bn254_ntt(ctx, z.clone().reshape(1, n, 4), inverse=True, montgomery=True)[0], n, b[6:9])
    zc = commit(blz)
    fs.bind("alpha", g1_marshal(zc))
    alpha = fs.challenge("alpha")
    # round 3: the quotient of the BLINDED polynomials (nlx_bn254_plonk_quotient patches the coefficients it derives from the
    # values on H), all 4 n coefficients, cut into h1 h2 h3 of n + 2
    polys = {k: pk.value(k) for k in pk.NAMES}
    polys.update(l=wires[0], r=wires[1], o=wires[2], z=z)
    if len(public_inputs):
        pi = [int(x) % R for x in public_inputs] + [0] * (n - len(public_inputs))
        polys["pi"] = torch.from_numpy(B.bn254_pack([[_to_mont(x) for x in pi]])[0].view(np.int64)).to(dev)
    h4 = torch.empty((4 * n, 4), dtype=torch.int64, device=dev)
    _, ok = B.bn254_plonk_quotient(ctx, polys, *[_to_mont(x) for x in (u, pk.k1, pk.k2, alpha, beta, gamma)], out=h4,
                                   blinding=[_to_mont(x) for x in b])
    if not ok:
        raise ValueError("the witness does not satisfy the circuit (the quotient has more than 3 n + 6 coefficients)")
    hs = [h4[0:n + 2], h4[n + 2:2 * n + 4], h4[2 * n + 4:3 * n + 6]]
    hc = [commit(c) for c in hs]
    for c in hc:
        fs.bind("zeta", g1_marshal(c))
    zeta = fs.challenge("zeta")
    # round 4: evaluations, z at w zeta on its own
    lz, rz, oz = (at(c, zeta) for c in bl)
    s1z, s2z = at(pk.coeff("s1"), zeta), at(pk.coeff("s2"), zeta)
    yw, _, zshift = B.bn254_kzg_open(ctx, blz, _to_mont(zeta * w % R), srs=pk.srs, want_quotient=False)
    zw = _from_words_mont(yw)
    # round 5: the linearised polynomial, foldedH, ONE batched opening at zeta
    zh = (pow(zeta, n, R) - 1) % R
    l1 = zh * pow(n * (zeta - 1) % R, R - 2, R) % R
    a_ = (lz + beta * zeta + gamma) * (rz + beta * pk.k1 * zeta + gamma) % R * (oz + beta * pk.k2 * zeta + gamma) % R
    b_ = (lz + beta * s1z + gamma) * (rz + beta * s2z + gamma) % R
    m = n + 3

    def padded(c):
        if c.shape[0] == m:
            return c
        out = torch.zeros((m, 4), dtype=torch.int64, device=dev)
        out[:c.shape[0]] = c
        return out
    lin = torch.empty((m, 4), dtype=torch.int64, device=dev)
    lincomb(ctx, [padded(pk.coeff(k)) for k in ("qm", "ql", "qr", "qo", "qk")] + [blz, padded(pk.coeff("s3"))],
            [lz * rz % R, lz, rz, oz, 1, (alpha * a_ + alpha * alpha % R * l1) % R, (-alpha * b_ % R * beta % R * zw) % R], lin)
    zn2 = pow(zeta, n + 2, R)
    folded_h = torch.empty((n + 2, 4), dtype=torch.int64, device=dev)
    lincomb(ctx, [c.contiguous() for c in hs], [1, zn2, zn2 * zn2 % R], folded_h)
    batch = [padded(folded_h), lin, padded(bl[0]), padded(bl[1]), padded(bl[2]), padded(pk.coeff("s1")), padded(pk.coeff("s2"))]
    digests = [commit(folded_h), commit(lin)] + lro + [pk.commitments["s1"], pk.commitments["s2"]]
    claimed = [at(batch[0], zeta), at(lin, zeta), lz, rz, oz, s1z, s2z]
    fg = FiatShamir("gamma")
    fg.bind("gamma", fr_bytes(zeta))
    for d in digests:
        fg.bind("gamma", g1_marshal(d))
    for v in claimed:
        fg.bind("gamma", fr_bytes(v))
    gp = fg.challenge("gamma")
    folded = torch.empty((m, 4), dtype=torch.int64, device=dev)
    lincomb(ctx, batch, [pow(gp, i, R) for i in range(7)], folded)
    bh = B.bn254_kzg_open(ctx, folded, _to_mont(zeta), srs=pk.srs, want_quotient=False)[2]
    out = b"".join(g1_compress(c) for c in lro) + g1_compress(zc) + b"".join(g1_compress(c) for c in hc)
    out += (0).to_bytes(4, "big")                                    # Bsb22Commitments: none
    out += g1_compress(bh) + (7).to_bytes(4, "big") + b"".join(fr_bytes(v) for v in claimed)
    return out + g1_compress(zshift) + fr_bytes(zw)
