"""ORACLE - TEST INFRASTRUCTURE ONLY.  ctypes wrapper over oracle/liboracle.so.

May be imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
(never by the product package).  Builds the library with `make` on first use.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# generator pair: include/nlx_field.h; NLX_GL_GENERATOR_SET=2021 builds / loads the other candidate pair
GEN_SET = os.environ.get("NLX_GL_GENERATOR_SET", "7")
_LIBNAME = "liboracle.so" if GEN_SET == "7" else "liboracle_gen%s.so" % GEN_SET
_LIB = os.path.join(_HERE, _LIBNAME)
P = 0xFFFFFFFF00000001


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h", ".inc"))]
    srcs.append(os.path.join(_HERE, "..", "include", "nlx_field.h"))
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(s) for s in srcs):
        subprocess.run(["make", "-C", _HERE, "-B", _LIBNAME, "GEN_SET=" + GEN_SET], check=True, capture_output=True)
    return _LIB


def generators():
    """(MULTIPLICATIVE_GROUP_GENERATOR, POWER_OF_TWO_GENERATOR) the oracle was built with"""
    out = (ctypes.c_uint64 * 2)()
    dll().orc_field_generators(out)
    return int(out[0]), int(out[1])


_dll = None
u64p = ctypes.POINTER(ctypes.c_uint64)


def dll():
    global _dll
    if _dll is None:
        _dll = ctypes.CDLL(build())
        _dll.orc_merkle_digest_words.restype = ctypes.c_size_t
        _dll.orc_merkle_digest_words.argtypes = [ctypes.c_size_t, ctypes.c_uint]
        _dll.orc_merkle_verify.restype = ctypes.c_int
    return _dll


def _p(a):
    return a.ctypes.data_as(u64p) if a is not None else None


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def poseidon_permute(states):
    s = _u64(states).copy().reshape(-1, 12)
    for i in range(s.shape[0]):
        dll().orc_poseidon_permute(_p(s[i]))
    return s


def poseidon_permute_naive(states):
    s = _u64(states).copy().reshape(-1, 12)
    for i in range(s.shape[0]):
        dll().orc_poseidon_permute_naive(_p(s[i]))
    return s


def hash_or_noop(xs):
    x = _u64(xs)
    out = np.zeros(4, dtype=np.uint64)
    dll().orc_hash_or_noop(_p(x), ctypes.c_size_t(x.size), _p(out))
    return out


def two_to_one(l, r):
    out = np.zeros(4, dtype=np.uint64)
    dll().orc_two_to_one(_p(_u64(l)), _p(_u64(r)), _p(out))
    return out


def fft(a, inverse=False, shift=1):
    x = _u64(a).copy()
    log_n = x.size.bit_length() - 1
    d = dll()
    if shift in (0, 1):
        (d.orc_ifft if inverse else d.orc_fft)(_p(x), ctypes.c_uint(log_n))
    else:
        (d.orc_coset_ifft if inverse else d.orc_coset_fft)(_p(x), ctypes.c_uint(log_n), ctypes.c_uint64(shift))
    return x


def merkle_build(leaves, cap_height):
    lv = _u64(leaves)
    n, ln = lv.shape
    words = dll().orc_merkle_digest_words(n, cap_height)
    dig = np.zeros(words, dtype=np.uint64)
    cap = np.zeros((1 << cap_height, 4), dtype=np.uint64)
    dll().orc_merkle_build(_p(lv), ctypes.c_size_t(n), ctypes.c_size_t(ln), ctypes.c_uint(cap_height), _p(dig), _p(cap))
    return dig, cap


def merkle_prove(digests, n_leaves, cap_height, idx):
    plen = (n_leaves.bit_length() - 1) - cap_height
    out = np.zeros((max(plen, 0), 4), dtype=np.uint64)
    dll().orc_merkle_prove(_p(_u64(digests)), ctypes.c_size_t(n_leaves), ctypes.c_uint(cap_height),
                           ctypes.c_size_t(idx), _p(out))
    return out


def merkle_verify(leaf, idx, siblings, cap, cap_height):
    lf, sb, cp = _u64(leaf), _u64(siblings), _u64(cap)
    return bool(dll().orc_merkle_verify(_p(lf), ctypes.c_size_t(lf.size), ctypes.c_size_t(idx), _p(sb),
                                        ctypes.c_uint(sb.size // 4), _p(cp), ctypes.c_uint(cap_height)))


def commit(data, rate_bits, cap_height, from_coeffs=False):
    """returns dict(coeffs, leaves, digests, cap) following PolynomialBatch::from_values/from_coeffs."""
    v = _u64(data)
    n_cols, n = v.shape
    log_n = n.bit_length() - 1
    L = n << rate_bits
    coeffs = np.zeros((n_cols, n), dtype=np.uint64)
    leaves = np.zeros((L, n_cols), dtype=np.uint64)
    dig = np.zeros(dll().orc_merkle_digest_words(L, cap_height), dtype=np.uint64)
    cap = np.zeros((1 << cap_height, 4), dtype=np.uint64)
    if from_coeffs:
        coeffs[:] = v
        dll().orc_commit_from_coeffs(_p(v), ctypes.c_size_t(n_cols), ctypes.c_uint(log_n), ctypes.c_uint(rate_bits),
                                     ctypes.c_uint(cap_height), _p(leaves), _p(dig), _p(cap))
    else:
        dll().orc_commit_from_values(_p(v), ctypes.c_size_t(n_cols), ctypes.c_uint(log_n), ctypes.c_uint(rate_bits),
                                     ctypes.c_uint(cap_height), _p(coeffs), _p(leaves), _p(dig), _p(cap))
    return {"coeffs": coeffs, "leaves": leaves, "digests": dig, "cap": cap}


class Challenger:
    class _S(ctypes.Structure):
        _fields_ = [("state", ctypes.c_uint64 * 12), ("in_buf", ctypes.c_uint64 * 8), ("n_in", ctypes.c_uint),
                    ("out_buf", ctypes.c_uint64 * 8), ("n_out", ctypes.c_uint)]

    def __init__(self):
        self.s = self._S()
        dll().orc_ch_init(ctypes.byref(self.s))
        dll().orc_ch_challenge.restype = ctypes.c_uint64

    def observe(self, x):
        dll().orc_ch_observe(ctypes.byref(self.s), ctypes.c_uint64(int(x)))

    def challenge(self):
        return int(dll().orc_ch_challenge(ctypes.byref(self.s)))


def eval_poly(coeffs, x):
    """Horner evaluation of base-field coefficients at a base-field point, in C (long polynomials)"""
    c = _u64(coeffs)
    d = dll()
    d.orc_eval_poly_base.restype = ctypes.c_uint64
    d.orc_eval_poly_base.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint64]
    return int(d.orc_eval_poly_base(c.ctypes.data, c.size, int(x) % P))


def eval_poly_ext(coeffs, z):
    """Horner evaluation of base-field coefficients at z = (a, b) in F_p[X]/(X^2-7) (python ints)."""
    a, b = 0, 0
    za, zb = int(z[0]), int(z[1])
    for c in reversed([int(x) for x in coeffs]):
        na = (a * za + 7 * b * zb + c) % P
        nb = (a * zb + b * za) % P
        a, b = na, nb
    return a, b


# ---------------- whole-proof oracle (plonk.h) ----------------
class _Gate(ctypes.Structure):
    _fields_ = [(k, ctypes.c_uint32) for k in
                ("kind", "selector_index", "group_start", "group_end", "index", "param0", "param1")]


class _Desc(ctypes.Structure):
    _fields_ = [(k, ctypes.c_uint32) for k in (
        "degree_bits", "num_wires", "num_routed_wires", "num_constants", "num_challenges", "rate_bits",
        "cap_height", "quotient_degree_factor", "num_partial_products", "fri_pow_bits", "fri_num_queries",
        "fri_arity_bits", "fri_final_poly_bits", "num_selectors", "num_gates", "num_public_inputs")] + [
        ("gates", ctypes.POINTER(_Gate)), ("k_is", u64p), ("circuit_digest", ctypes.c_uint64 * 4),
        # lookup tables (plonk.h): all zero = none
        ("num_luts", ctypes.c_uint32), ("pad_", ctypes.c_uint32), ("lut_sizes", ctypes.POINTER(ctypes.c_uint32)),
        ("lut_pairs", ctypes.POINTER(ctypes.c_uint16)), ("lookup_rows", ctypes.POINTER(ctypes.c_uint32)),
        ("lut_num_lookups", ctypes.POINTER(ctypes.c_uint32))]


class _Trace(ctypes.Structure):
    _fields_ = [("betas", ctypes.c_uint64 * 4), ("gammas", ctypes.c_uint64 * 4), ("alphas", ctypes.c_uint64 * 4),
                ("zeta", ctypes.c_uint64 * 2), ("fri_alpha", ctypes.c_uint64 * 2), ("fri_betas", ctypes.c_uint64 * 32),
                ("pow_witness", ctypes.c_uint64), ("n_fri_rounds", ctypes.c_uint32),
                ("query_indices", ctypes.c_uint64 * 128), ("deltas", ctypes.c_uint64 * 16), ("zs_partial_values", u64p),
                ("quotient_chunk_coeffs", u64p), ("fri_final_values", u64p)]


class Circuit:
    """orc_circuit: built from the same flat description the product ABI takes (field-for-field)."""

    def __init__(self, desc_bytes_like, gates_like, k_is, constants, sigmas):
        d = dll()
        d.orc_circuit_build.restype = ctypes.c_void_p
        d.orc_proof_max_bytes.restype = ctypes.c_size_t
        d.orc_proof_max_bytes.argtypes = [ctypes.c_void_p]
        d.orc_prove.restype = ctypes.c_size_t
        d.orc_prove.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        d.orc_prove_traced.restype = ctypes.c_size_t
        d.orc_prove_traced.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_size_t, ctypes.c_void_p]
        d.orc_verify.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        d.orc_circuit_free.argtypes = [ctypes.c_void_p]
        d.orc_circuit_digest.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        d.orc_circuit_constants_sigmas_cap.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self._gates = (_Gate * len(gates_like))()
        for i, g in enumerate(gates_like):
            for f, _ in _Gate._fields_:
                setattr(self._gates[i], f, getattr(g, f))
        self._k = _u64(k_is).copy()
        self.desc = _Desc()
        for f, _ in _Desc._fields_[:16]:
            setattr(self.desc, f, getattr(desc_bytes_like, f))
        self.desc.gates = self._gates
        self.desc.k_is = _p(self._k)
        if getattr(desc_bytes_like, "num_luts", 0):   # the arrays are copied by orc_circuit_build
            self.desc.num_luts = desc_bytes_like.num_luts
            for f in ("lut_sizes", "lut_pairs", "lookup_rows", "lut_num_lookups"):
                setattr(self.desc, f, ctypes.cast(getattr(desc_bytes_like, f), dict(_Desc._fields_)[f]))
        self.h = d.orc_circuit_build(ctypes.byref(self.desc), _p(_u64(constants)), _p(_u64(sigmas)))
        self.max_bytes = d.orc_proof_max_bytes(self.h)

    @classmethod
    def from_synthetic(cls, syn):
        return cls(syn.desc(), list(syn.gates), syn.k_is, syn.constants, syn.sigmas)

    def digest(self):
        out = np.zeros(4, dtype=np.uint64)
        dll().orc_circuit_digest(self.h, out.ctypes.data)
        return out

    def constants_sigmas_cap(self):
        out = np.zeros((1 << self.desc.cap_height, 4), dtype=np.uint64)
        dll().orc_circuit_constants_sigmas_cap(self.h, out.ctypes.data)
        return out

    def prove(self, wires, public_inputs, trace=False):
        w, pi = _u64(wires), _u64(public_inputs)
        buf = np.zeros(self.max_bytes, dtype=np.uint8)
        if not trace:
            ln = dll().orc_prove(self.h, w.ctypes.data, pi.ctypes.data, buf.ctypes.data, buf.size)
            return buf[:ln].tobytes()
        tr = _Trace()
        n = 1 << self.desc.degree_bits
        L = n << self.desc.rate_bits
        nzs = self.desc.num_challenges * (1 + self.desc.num_partial_products)
        if self.desc.num_luts:   # + (RE, SLDC_0..S-1) per challenge
            nzs += self.desc.num_challenges * (1 + -(-(self.desc.num_routed_wires // 2) // (self.desc.quotient_degree_factor - 1)))
        nq = self.desc.num_challenges * self.desc.quotient_degree_factor
        zs = np.zeros((nzs, n), dtype=np.uint64)
        qc = np.zeros((nq, n), dtype=np.uint64)
        fv = np.zeros((L, 2), dtype=np.uint64)
        tr.zs_partial_values, tr.quotient_chunk_coeffs, tr.fri_final_values = _p(zs), _p(qc), _p(fv)
        ln = dll().orc_prove_traced(self.h, w.ctypes.data, pi.ctypes.data, buf.ctypes.data, buf.size, ctypes.byref(tr))
        info = {"betas": list(tr.betas)[:2], "gammas": list(tr.gammas)[:2], "alphas": list(tr.alphas)[:2],
                "zeta": list(tr.zeta), "fri_alpha": list(tr.fri_alpha), "pow_witness": tr.pow_witness,
                "n_fri_rounds": tr.n_fri_rounds, "query_indices": list(tr.query_indices)[:self.desc.fri_num_queries],
                "zs_partial_values": zs, "quotient_chunk_coeffs": qc, "fri_final_values": fv,
                "deltas": list(tr.deltas)[:4 * self.desc.num_challenges]}
        return buf[:ln].tobytes(), info

    def verify(self, proof):
        b = np.frombuffer(proof, dtype=np.uint8)
        return int(dll().orc_verify(self.h, b.ctypes.data, b.size))

    def set_lookup_wires(self, wires):
        """prover::set_lookup_wires on a copy of the witness (multiplicities, padding slots); returns the copy"""
        w = _u64(wires).copy()
        dll().orc_set_lookup_wires.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        if dll().orc_set_lookup_wires(self.h, w.ctypes.data) != 0:
            raise ValueError("a looked-up input is not in its table")
        return w

    def close(self):
        if self.h:
            dll().orc_circuit_free(self.h)
            self.h = None


# ---------------- STARK oracle (stark.h) ----------------
def _stark_sigs():
    d = dll()
    d.orc_stark_proof_max_bytes.restype = ctypes.c_size_t
    d.orc_stark_proof_max_bytes.argtypes = [ctypes.c_void_p]
    d.orc_stark_prove.restype = ctypes.c_size_t
    d.orc_stark_prove.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    d.orc_stark_verify.restype = ctypes.c_int
    d.orc_stark_verify.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    d.orc_stark_air_digest.restype = None
    d.orc_stark_air_digest.argtypes = [ctypes.c_void_p, u64p]
    return d


def stark_prove(desc, trace, public_inputs):
    """desc: a ctypes structure laid out as orc_stark_desc (the package's StarkDesc is).  trace (n_cols, n)."""
    d = _stark_sigs()
    trace = _u64(trace)
    pis = _u64(public_inputs)
    assert trace.shape == (desc.n_cols, 1 << desc.degree_bits) and pis.size == desc.num_public_inputs
    out = np.empty(d.orc_stark_proof_max_bytes(ctypes.addressof(desc)), dtype=np.uint8)
    n = d.orc_stark_prove(ctypes.addressof(desc), trace.ctypes.data, pis.ctypes.data if pis.size else None,
                          out.ctypes.data, out.size)
    if n == 0:
        raise RuntimeError("orc_stark_prove failed (bad descriptor or buffer overflow)")
    return out[:n].tobytes()


def stark_verify(desc, proof):
    d = _stark_sigs()
    buf = np.frombuffer(proof, dtype=np.uint8)
    return int(d.orc_stark_verify(ctypes.addressof(desc), buf.ctypes.data, buf.size))


def stark_air_digest(desc):
    """the statement digest the STARK transcript opens with (orc_stark_air_digest)"""
    d = _stark_sigs()
    out = np.zeros(4, dtype=np.uint64)
    d.orc_stark_air_digest(ctypes.addressof(desc), _p(out))
    return [int(x) for x in out]


ROUND_FN = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint64), ctypes.c_uint32,
                            ctypes.POINTER(ctypes.c_uint64))


def stark_prove_rounds(desc, round_fn, public_inputs):
    """Multi-round STARK (orc_stark_prove_rounds).  round_fn(round, known: list[int]) -> (round_cols, n) uint64
    array: round r's columns, computed from the round values and challenges of the earlier rounds; a round that has
    round values returns (columns, values)."""
    d = _stark_sigs()
    d.orc_stark_prove_rounds.restype = ctypes.c_size_t
    d.orc_stark_prove_rounds.argtypes = [ctypes.c_void_p, ROUND_FN, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_size_t]
    pis = _u64(public_inputs)
    keep = []

    def cb(_user, rnd, ch_ptr, n_ch, values_out):
        chal = [int(ch_ptr[i]) for i in range(n_ch)]
        res = round_fn(rnd, chal)
        n_rv = desc.round_values[rnd] if desc.n_rounds else 0
        if n_rv:                              # a round with round values returns (columns, values)
            res, vals = res
            assert len(vals) == n_rv
            for i, v in enumerate(vals):
                values_out[i] = int(v) % P
        arr = _u64(res)
        assert arr.shape == (desc.round_cols[rnd] if desc.n_rounds else desc.n_cols, 1 << desc.degree_bits)
        keep.append(arr)
        return arr.ctypes.data

    out = np.empty(d.orc_stark_proof_max_bytes(ctypes.addressof(desc)), dtype=np.uint8)
    n = d.orc_stark_prove_rounds(ctypes.addressof(desc), ROUND_FN(cb), None, pis.ctypes.data if pis.size else None,
                                 out.ctypes.data, out.size)
    if n == 0:
        raise RuntimeError("orc_stark_prove_rounds failed (bad descriptor, callback or buffer overflow)")
    return out[:n].tobytes()


def stark_values(desc, proof):
    """orc_stark_values: [public inputs | round values and challenges, round by round] as the verifier derives them."""
    d = dll()
    d.orc_stark_values.restype = ctypes.c_uint32
    d.orc_stark_values.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    buf = np.frombuffer(proof, dtype=np.uint8)
    out = np.zeros(desc.num_public_inputs + 3 * (64 + 16) + 1, dtype=np.uint64)
    n = d.orc_stark_values(ctypes.addressof(desc), buf.ctypes.data, buf.size, out.ctypes.data)
    if n == 0 and desc.num_public_inputs:
        raise ValueError("malformed proof")
    return [int(v) for v in out[:n]]


def logup_multiplicities(trace, cols, table_bits, table_cols=1):
    """orc_logup_multiplicities: trace (n_cols, n) uint64, cols = the looked-up columns.  Returns m: (n,) for one table
    column, (table_cols, n) when the table is spread over several; raises if a cell is outside the table."""
    d = dll()
    t = _u64(trace)
    c = np.ascontiguousarray(cols, dtype=np.uint32)
    n = t.shape[1]
    m = np.zeros((table_cols, n), dtype=np.uint64)
    d.orc_logup_multiplicities.restype = ctypes.c_int
    d.orc_logup_multiplicities.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32,
                                           ctypes.c_uint32, ctypes.c_void_p]
    if not d.orc_logup_multiplicities(t.ctypes.data, n.bit_length() - 1, c.ctypes.data, c.size, table_bits, table_cols, m.ctypes.data):
        raise ValueError("a looked-up cell is outside the table")
    return m[0] if table_cols == 1 else m


def logup_round(trace, cols, table_bits, mult, alpha, table_cols=1):
    """orc_logup_round: the round-1 columns (helpers, g per table column, phi) as a (2 ceil(k / 2) + 2 table_cols + 2, n) array."""
    d = dll()
    t = _u64(trace)
    c = np.ascontiguousarray(cols, dtype=np.uint32)
    m = np.ascontiguousarray(_u64(mult).reshape(table_cols, -1))
    n = t.shape[1]
    al = np.array([int(alpha[0]), int(alpha[1])], dtype=np.uint64)
    out = np.zeros((2 * ((c.size + 1) // 2) + 2 * table_cols + 2, n), dtype=np.uint64)
    d.orc_logup_round.restype = None
    d.orc_logup_round.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    d.orc_logup_round(t.ctypes.data, n.bit_length() - 1, c.ctypes.data, c.size, table_bits, table_cols, m.ctypes.data, al.ctypes.data,
                      out.ctypes.data)
    return out
