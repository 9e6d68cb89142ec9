"""Host-side mirror of plonky2's PolynomialBatch / MerkleTree / fft helpers over the C ABI.

Names and argument meaning follow plonky2::fri::oracle::PolynomialBatch and
plonky2::hash::merkle_tree::MerkleTree; arrays are numpy uint64
(host) or torch int64/uint64 tensors on the context's device (used in place).
"""
import ctypes

import numpy as np

from ._lib import dll, ptr


def poseidon_permute(ctx, states):
    """plonky2 Poseidon::poseidon on a batch: states (n, 12) uint64, returns a new array."""
    s = np.ascontiguousarray(states, dtype=np.uint64).copy()
    if s.ndim != 2 or s.shape[1] != 12:
        raise ValueError("states must have shape (n, 12)")
    ctx.check(dll.nlx_poseidon_permute_batch(ctx.handle, ptr(s), s.shape[0]))
    return s


def field_ops(ctx, a, b):
    """element-wise a*b, a+b, a-b, 1/a and the raw multiply path on unreduced inputs: (5, n) uint64"""
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    out = np.zeros((5, a.size), dtype=np.uint64)
    ctx.check(dll.nlx_field_ops(ctx.handle, ptr(a), ptr(b), a.size, ptr(out)))
    return out


def hash_rows(ctx, rows):
    """PoseidonHash::hash_or_noop of every row of a row-major (n_rows, row_len) matrix."""
    r = np.ascontiguousarray(rows, dtype=np.uint64)
    out = np.zeros((r.shape[0], 4), dtype=np.uint64)
    ctx.check(dll.nlx_hash_rows(ctx.handle, ptr(r), r.shape[0], r.shape[1], ptr(out)))
    return out


def ntt(ctx, cols, inverse=False, coset_shift=1):
    """fft / ifft / coset_fft / coset_ifft of each row of `cols` ((n_cols, n), natural order)."""
    c = np.ascontiguousarray(cols, dtype=np.uint64).copy()
    n = c.shape[1]
    log_n = n.bit_length() - 1
    if 1 << log_n != n:
        raise ValueError("length must be a power of two")
    ctx.check(dll.nlx_ntt_batch(ctx.handle, ptr(c), c.shape[0], log_n, 1 if inverse else 0, int(coset_shift)))
    return c


class MerkleTree:
    """MerkleTree::new(leaves, cap_height): leaves (n_leaves, leaf_len) row-major."""

    def __init__(self, ctx, leaves, cap_height):
        lv = np.ascontiguousarray(leaves, dtype=np.uint64)
        self.n_leaves, self.leaf_len = lv.shape
        self.cap_height = cap_height
        words = dll.nlx_merkle_digest_words(self.n_leaves, cap_height)
        if words == 0:
            raise ValueError("n_leaves must be a power of two")
        self.digests = np.zeros(words, dtype=np.uint64)
        self.cap = np.zeros((1 << cap_height, 4), dtype=np.uint64)
        ctx.check(dll.nlx_merkle_build(ctx.handle, ptr(lv), self.n_leaves, self.leaf_len, cap_height,
                                       ptr(self.digests), ptr(self.cap)))
        self.leaves = lv

    def prove(self, leaf_index):
        """Sibling digests bottom-up (MerkleTree::prove)."""
        sib, off, lvl, idx = [], 0, self.n_leaves, leaf_index
        while lvl > (1 << self.cap_height):
            s = off + (idx ^ 1) * 4
            sib.append(self.digests[s:s + 4].copy())
            off += lvl * 4
            lvl >>= 1
            idx >>= 1
        return np.array(sib, dtype=np.uint64).reshape(-1, 4)


class PolynomialBatch:
    """Device-resident PolynomialBatch (coefficients + LDE table + Merkle tree in HBM)."""

    def __init__(self, ctx, handle, n_cols, log_n, rate_bits, cap_height, cap):
        self.ctx, self.handle = ctx, handle
        self.n_cols, self.log_n, self.rate_bits, self.cap_height = n_cols, log_n, rate_bits, cap_height
        self.cap = cap
        self._borrowed = False
        ctx._adopt(self)

    @classmethod
    def _make(cls, fn, ctx, data, rate_bits, cap_height):
        if isinstance(data, np.ndarray):
            data = np.ascontiguousarray(data, dtype=np.uint64)
        n_cols, n = data.shape
        log_n = int(n).bit_length() - 1
        if 1 << log_n != n:
            raise ValueError("polynomial length must be a power of two")
        cap = np.zeros((1 << cap_height, 4), dtype=np.uint64)
        h = ctypes.c_void_p()
        ctx.check(fn(ctx.handle, ptr(data), n_cols, log_n, rate_bits, cap_height, ptr(cap), ctypes.byref(h)))
        return cls(ctx, h, n_cols, log_n, rate_bits, cap_height, cap)

    @classmethod
    def _adopt_handle(cls, ctx, handle, n_cols, log_n, rate_bits, cap_height):
        """Wrap a commitment produced by a stage call (nlx_partial_products_and_zs, nlx_quotient_eval)."""
        cap = np.zeros((1 << cap_height, 4), dtype=np.uint64)
        ctx.check(dll.nlx_commit_get_cap(handle, ptr(cap)))
        return cls(ctx, handle, n_cols, log_n, rate_bits, cap_height, cap)

    @classmethod
    def from_values(cls, ctx, values, rate_bits, cap_height):
        """values: (n_cols, n) — one polynomial's subgroup evaluations per row."""
        return cls._make(dll.nlx_commit_from_values, ctx, values, rate_bits, cap_height)

    @classmethod
    def from_coeffs(cls, ctx, coeffs, rate_bits, cap_height):
        return cls._make(dll.nlx_commit_from_coeffs, ctx, coeffs, rate_bits, cap_height)

    @property
    def lde_size(self):
        return 1 << (self.log_n + self.rate_bits)

    def coeffs(self):
        out = np.zeros((self.n_cols, 1 << self.log_n), dtype=np.uint64)
        self.ctx.check(dll.nlx_commit_get_coeffs(self.handle, ptr(out)))
        return out

    def leaves(self):
        out = np.zeros((self.lde_size, self.n_cols), dtype=np.uint64)
        self.ctx.check(dll.nlx_commit_get_leaves(self.handle, ptr(out)))
        return out

    def digests(self):
        out = np.zeros(dll.nlx_merkle_digest_words(self.lde_size, self.cap_height), dtype=np.uint64)
        self.ctx.check(dll.nlx_commit_get_digests(self.handle, ptr(out)))
        return out

    def open_rows(self, indices, with_paths=True):
        idx = np.ascontiguousarray(indices, dtype=np.uint64)
        k = idx.shape[0]
        rows = np.zeros((k, self.n_cols), dtype=np.uint64)
        plen = self.log_n + self.rate_bits - self.cap_height
        paths = np.zeros((k, plen, 4), dtype=np.uint64) if with_paths else None
        self.ctx.check(dll.nlx_commit_open_rows(self.handle, ptr(idx), k, ptr(rows), ptr(paths)))
        return rows, paths

    def eval_at(self, zeta):
        z = np.array(zeta, dtype=np.uint64)
        out = np.zeros((self.n_cols, 2), dtype=np.uint64)
        self.ctx.check(dll.nlx_commit_eval_at(self.handle, ptr(z), ptr(out)))
        return out

    def close(self):
        if self.handle and self.ctx.handle and not self._borrowed:
            dll.nlx_commit_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- row f.4, first piece: the BN254 scalar-field NTT (csrc/bn254.hip, nlx_bn254_ntt_batch) ----
BN254_R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def bn254_pack(values):
    """nested lists of Python ints < r, shape (n_cols, n) -> (n_cols, n, 4) uint64, little-endian words"""
    v = [[int(x) for x in col] for col in values]
    out = np.zeros((len(v), len(v[0]), 4), dtype=np.uint64)
    for c, col in enumerate(v):
        for i, x in enumerate(col):
            for w in range(4):
                out[c, i, w] = (x >> (64 * w)) & 0xFFFFFFFFFFFFFFFF
    return out


def bn254_unpack(arr):
    a = np.asarray(arr, dtype=np.uint64)
    return [[sum(int(a[c, i, w]) << (64 * w) for w in range(4)) for i in range(a.shape[1])] for c in range(a.shape[0])]


def bn254_ntt(ctx, cols, inverse=False, montgomery=False, coset_shift=None, bitrev_out=False, bitrev_in=False):
    """gnark-crypto fft.Domain.FFT / FFTInverse over BN254's scalar field, natural order in and out.  cols: (n_cols, n, 4)
    uint64 (bn254_pack), canonical integers < r, or fr.Element Montgomery words with montgomery=True; or a device tensor of
    that shape (transformed in place and returned).  coset_shift (an integer < r in the form of the data): the transform on
    the coset shift * <w> (fft.OnCoset()); bitrev_out: leave the output in bit-reversed order (fft.DIF); bitrev_in: the
    input is in bit-reversed order (fft.DIT)."""
    if hasattr(cols, "data_ptr"):
        n_cols, n = cols.shape[0], cols.shape[1]
        c = cols
    else:
        c = np.ascontiguousarray(cols, dtype=np.uint64).copy()
        n_cols, n = c.shape[0], c.shape[1]
    log_n = n.bit_length() - 1
    if 1 << log_n != n or c.shape[2] != 4:
        raise ValueError("shape must be (n_cols, 2^k, 4)")
    p = c.data_ptr() if hasattr(c, "data_ptr") else c.ctypes.data
    flags = (1 if montgomery else 0) | (2 if bitrev_out else 0) | (4 if bitrev_in else 0)
    shift = None
    if coset_shift is not None:
        shift = np.array([(int(coset_shift) >> (64 * w)) & 0xFFFFFFFFFFFFFFFF for w in range(4)], dtype=np.uint64)
    ctx.check(dll.nlx_bn254_ntt_batch_coset(ctx.handle, p, n_cols, log_n, 1 if inverse else 0, flags,
                                            shift.ctypes.data if shift is not None else None))
    return c


# ---- row f.4, second piece: the BN254 G1 multi-scalar multiplication (csrc/bn254_msm.hip, nlx_bn254_msm_g1) ----
BN254_Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
_MONT_Q = (1 << 256) % BN254_Q


def bn254_g1_pack(points):
    """[(x, y) | None] (affine integers < q; None = the point at infinity) -> (n, 8) uint64: gnark-crypto's G1Affine layout
    (X, Y in Montgomery form, four little-endian words each; infinity = all zero)"""
    out = np.zeros((len(points), 8), dtype=np.uint64)
    for i, pt in enumerate(points):
        if pt is None:
            continue
        for c, v in enumerate(pt):
            m = int(v) * _MONT_Q % BN254_Q
            for w in range(4):
                out[i, 4 * c + w] = (m >> (64 * w)) & 0xFFFFFFFFFFFFFFFF
    return out


def bn254_g1_unpack(words):
    """8 uint64 words (G1Affine, Montgomery) -> (x, y) integers, or None for the point at infinity"""
    w = [int(v) for v in np.asarray(words, dtype=np.uint64).reshape(8)]
    x, y = (sum(w[4 * c + k] << (64 * k) for k in range(4)) for c in range(2))
    if x == 0 and y == 0:
        return None
    rinv = pow(_MONT_Q, BN254_Q - 2, BN254_Q)
    return x * rinv % BN254_Q, y * rinv % BN254_Q


def bn254_msm_g1(ctx, points, scalars, montgomery=False):
    """gnark-crypto G1Affine.MultiExp: sum_i scalars[i] * points[i].  points: (n, 8) uint64 (bn254_g1_pack) or a device tensor
    of that shape; scalars: (n, 4) uint64 - canonical integers < r, or fr.Element Montgomery words with montgomery=True - or
    a device tensor.  Returns the 8 words of the result (G1Affine, Montgomery; all zero = infinity)."""
    def ptr(a, width):
        if hasattr(a, "data_ptr"):
            if tuple(a.shape)[1:] != (width,) or not a.is_contiguous():
                raise ValueError("expected a contiguous (n, %d) tensor" % width)
            return a, a.data_ptr(), a.shape[0]
        a = np.ascontiguousarray(a, dtype=np.uint64)
        if a.ndim != 2 or a.shape[1] != width:
            raise ValueError("expected shape (n, %d)" % width)
        return a, a.ctypes.data, a.shape[0]
    p_keep, p_ptr, n = ptr(points, 8)
    s_keep, s_ptr, n2 = ptr(scalars, 4)
    if n != n2:
        raise ValueError("points and scalars differ in length")
    out = np.zeros(8, dtype=np.uint64)
    ctx.check(dll.nlx_bn254_msm_g1(ctx.handle, p_ptr if n else None, s_ptr if n else None, n, 1 if montgomery else 0, out.ctypes.data))
    return out


class _PlonkQuotientArgs(ctypes.Structure):
    """nlx_bn254_plonk_quotient_args (include/nlx.h)"""
    _fields_ = [("log_n", ctypes.c_uint32), ("flags", ctypes.c_uint32)] + [(k, ctypes.c_void_p) for k in (
        "ql", "qr", "qm", "qo", "qk", "s1", "s2", "s3", "l", "r", "o", "z", "pi", "coset_shift", "k1", "k2", "alpha", "beta", "gamma", "blinding")]


def _fr_words(x):
    return np.array([(int(x) >> (64 * w)) & 0xFFFFFFFFFFFFFFFF for w in range(4)], dtype=np.uint64)


def bn254_plonk_quotient(ctx, polys, coset_shift, k1, k2, alpha, beta, gamma, out=None, blinding=None):
    """The PLONK prover's quotient chain over BN254's scalar field (nlx_bn254_plonk_quotient).  polys: dict with the values on
    H of ql qr qm qo qk s1 s2 s3 l r o z and optionally pi, each an (n, 4) uint64 array of fr.Element words (Montgomery) or a
    device tensor of that shape; the six scalars: integers in Montgomery form.  Returns (t, ok): t = (3, n, 4) uint64, the
    chunks t_lo, t_mid, t_hi (or `out`, a device tensor of that shape, filled in place); ok = the fourth chunk vanished (the witness
    satisfies the circuit)."""
    names = ("ql", "qr", "qm", "qo", "qk", "s1", "s2", "s3", "l", "r", "o", "z")
    keep, args = [], _PlonkQuotientArgs()
    n = None
    for k in names + ("pi",):
        v = polys.get(k)
        if v is None:
            if k != "pi":
                raise ValueError("missing polynomial %s" % k)
            continue
        if hasattr(v, "data_ptr"):
            shape, ptr = tuple(v.shape), v.data_ptr()
        else:
            v = np.ascontiguousarray(v, dtype=np.uint64)
            shape, ptr = v.shape, v.ctypes.data
        if len(shape) != 2 or shape[1] != 4 or (n is not None and shape[0] != n):
            raise ValueError("%s: expected shape (n, 4)" % k)
        n = shape[0]
        keep.append(v)
        setattr(args, k, ptr)
    args.log_n = n.bit_length() - 1
    if 1 << args.log_n != n:
        raise ValueError("n must be a power of two")
    args.flags = 1
    if blinding is not None:
        # nine scalars (Montgomery integers): l += (b0 + b1 X) Z_H, r: b2 b3, o: b4 b5, z += (b6 + b7 X + b8 X^2) Z_H; the call then
        # returns ALL 4 n coefficients of the quotient (3 n + 6 of them non-zero), shape (4 n, 4)
        if len(blinding) != 9:
            raise ValueError("nine blinding scalars")
        bw = np.stack([_fr_words(x) for x in blinding])
        keep.append(bw)
        args.blinding = bw.ctypes.data
        args.flags = 1 | 0x100
    for k, x in (("coset_shift", coset_shift), ("k1", k1), ("k2", k2), ("alpha", alpha), ("beta", beta), ("gamma", gamma)):
        w = _fr_words(x)
        keep.append(w)
        setattr(args, k, w.ctypes.data)
    if out is None:
        out = np.zeros((3, n, 4) if blinding is None else (4 * n, 4), dtype=np.uint64)
    ok = ctypes.c_int32()
    ctx.check(dll.nlx_bn254_plonk_quotient(ctx.handle, ctypes.byref(args), out.data_ptr() if hasattr(out, "data_ptr") else out.ctypes.data,
                                           ctypes.byref(ok)))
    return out, bool(ok.value)


def bn254_kzg_open(ctx, coeffs, zeta, srs=None, want_quotient=True):
    """One KZG opening (nlx_bn254_kzg_open).  coeffs: (m, 4) uint64 fr.Element words (Montgomery) or a device tensor; zeta: an
    integer in Montgomery form; srs: (>= m - 1, 8) G1Affine words (host array or device tensor) or None.  Returns (y words,
    quotient (m - 1, 4) or None, proof words or None)."""
    if hasattr(coeffs, "data_ptr"):
        m, ptr = coeffs.shape[0], coeffs.data_ptr()
    else:
        coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64)
        m, ptr = coeffs.shape[0], coeffs.ctypes.data
    z = _fr_words(zeta)
    y = np.zeros(4, dtype=np.uint64)
    q = np.zeros((m - 1, 4), dtype=np.uint64) if want_quotient else None
    proof = np.zeros(8, dtype=np.uint64) if srs is not None else None
    srs_ptr = None
    if srs is not None:
        srs_ptr = srs.data_ptr() if hasattr(srs, "data_ptr") else np.ascontiguousarray(srs, dtype=np.uint64).ctypes.data
        if not hasattr(srs, "data_ptr"):
            srs = np.ascontiguousarray(srs, dtype=np.uint64)
            srs_ptr = srs.ctypes.data
    ctx.check(dll.nlx_bn254_kzg_open(ctx.handle, ptr, m, z.ctypes.data, srs_ptr, y.ctypes.data,
                                     q.ctypes.data if q is not None else None, proof.ctypes.data if proof is not None else None))
    return y, q, proof


def bn254_g1_sum(points):
    """The sum of G1Affine points, (n, 8) uint64 words each (host): joins the partial MSMs of several GPUs."""
    a = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 8)
    out = np.zeros(8, dtype=np.uint64)
    if dll.nlx_bn254_g1_sum(a.ctypes.data if len(a) else None, len(a), out.ctypes.data) != 0:
        raise ValueError("nlx_bn254_g1_sum failed")
    return out


def bn254_g1_multiples(ctx, base, n, device=None):
    """[(i + 1) * base for i < n] as G1Affine words, computed on the GPU: (n, 8) uint64 on the host, or an int64 device
    tensor with device="cuda:k" (test and bench data: n distinct curve points)."""
    b = bn254_g1_pack([base])
    if device is not None:
        import torch
        out = torch.empty((n, 8), dtype=torch.int64, device=device)
        ctx.check(dll.nlx_bn254_g1_multiples(ctx.handle, b.ctypes.data, n, out.data_ptr()))
        return out
    out = np.zeros((n, 8), dtype=np.uint64)
    ctx.check(dll.nlx_bn254_g1_multiples(ctx.handle, b.ctypes.data, n, out.ctypes.data))
    return out


# ---- G2 (Groth16's B query): coordinates in Fq2 = Fq[u] / (u^2 + 1), gnark-crypto's G2Affine{X, Y E2{A0, A1}} ----
def bn254_g2_pack(points):
    """[((x0, x1), (y0, y1)) | None] -> (n, 16) uint64: X.A0, X.A1, Y.A0, Y.A1, each a Montgomery fp.Element"""
    out = np.zeros((len(points), 16), dtype=np.uint64)
    for i, pt in enumerate(points):
        if pt is None:
            continue
        for c, v in enumerate((pt[0][0], pt[0][1], pt[1][0], pt[1][1])):
            m = int(v) * _MONT_Q % BN254_Q
            for w in range(4):
                out[i, 4 * c + w] = (m >> (64 * w)) & 0xFFFFFFFFFFFFFFFF
    return out


def bn254_g2_unpack(words):
    """16 uint64 words (G2Affine, Montgomery) -> ((x0, x1), (y0, y1)) integers, or None for the point at infinity"""
    w = [int(v) for v in np.asarray(words, dtype=np.uint64).reshape(16)]
    c = [sum(w[4 * k + j] << (64 * j) for j in range(4)) for k in range(4)]
    if not any(c):
        return None
    rinv = pow(_MONT_Q, BN254_Q - 2, BN254_Q)
    c = [v * rinv % BN254_Q for v in c]
    return (c[0], c[1]), (c[2], c[3])


def bn254_msm_g2(ctx, points, scalars, montgomery=False):
    """gnark-crypto G2Affine.MultiExp: points (n, 16) uint64 (bn254_g2_pack) or a device tensor, scalars as bn254_msm_g1.
    Returns the 16 words of the result."""
    def ptr(a, width):
        if hasattr(a, "data_ptr"):
            if tuple(a.shape)[1:] != (width,) or not a.is_contiguous():
                raise ValueError("expected a contiguous (n, %d) tensor" % width)
            return a, a.data_ptr(), a.shape[0]
        a = np.ascontiguousarray(a, dtype=np.uint64)
        if a.ndim != 2 or a.shape[1] != width:
            raise ValueError("expected shape (n, %d)" % width)
        return a, a.ctypes.data, a.shape[0]
    p_keep, p_ptr, n = ptr(points, 16)
    s_keep, s_ptr, n2 = ptr(scalars, 4)
    if n != n2:
        raise ValueError("points and scalars differ in length")
    out = np.zeros(16, dtype=np.uint64)
    ctx.check(dll.nlx_bn254_msm_g2(ctx.handle, p_ptr if n else None, s_ptr if n else None, n, 1 if montgomery else 0, out.ctypes.data))
    return out


def bn254_g2_sum(points):
    a = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 16)
    out = np.zeros(16, dtype=np.uint64)
    if dll.nlx_bn254_g2_sum(a.ctypes.data if len(a) else None, len(a), out.ctypes.data) != 0:
        raise ValueError("nlx_bn254_g2_sum failed")
    return out
