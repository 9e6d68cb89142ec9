"""Host-side mirror of the plonky2 circuit / prover interface over the C ABI.

Mirrors plonky2::plonk::circuit_data::{CircuitConfig, CircuitData} and
plonk::prover::prove as reached from nearx/src/test_utils.rs:29,62 (builder.build(),
circuit.prove()).  Real nearx circuits come from the Rust CircuitBuilder; here
`SyntheticCircuit` produces circuits of the same static shape for tests and benchmarks.
"""
import ctypes

import numpy as np

from ._lib import dll, synth_dll, ptr, NlxError

(GATE_NOOP, GATE_CONSTANT, GATE_PUBLIC_INPUT, GATE_ARITHMETIC, GATE_BASE_SUM, GATE_POSEIDON, GATE_ARITHMETIC_EXT,
 GATE_MUL_EXT, GATE_REDUCING, GATE_REDUCING_EXT, GATE_POSEIDON_MDS, GATE_EXPONENTIATION, GATE_RANDOM_ACCESS,
 GATE_COSET_INTERPOLATION, GATE_U32_ADD_MANY, GATE_U32_ARITHMETIC, GATE_U32_SUBTRACTION, GATE_U32_RANGE_CHECK,
 GATE_COMPARISON, GATE_LOOKUP, GATE_LOOKUP_TABLE) = range(21)


class GateDesc(ctypes.Structure):
    _fields_ = [(k, ctypes.c_uint32) for k in
                ("kind", "selector_index", "group_start", "group_end", "index", "param0", "param1")]


class CircuitDesc(ctypes.Structure):
    _fields_ = [(k, ctypes.c_uint32) for k in (
        "degree_bits", "num_wires", "num_routed_wires", "num_constants", "num_challenges", "rate_bits",
        "cap_height", "quotient_degree_factor", "num_partial_products", "fri_pow_bits", "fri_num_queries",
        "fri_arity_bits", "fri_final_poly_bits", "num_selectors", "num_gates", "num_public_inputs")] + [
        ("gates", ctypes.POINTER(GateDesc)), ("k_is", ctypes.POINTER(ctypes.c_uint64)),
        ("circuit_digest", ctypes.c_uint64 * 4),
        # lookup tables (include/nlx.h): a zero tail = none
        ("num_luts", ctypes.c_uint32), ("pad_", ctypes.c_uint32), ("lut_sizes", ctypes.POINTER(ctypes.c_uint32)),
        ("lut_pairs", ctypes.POINTER(ctypes.c_uint16)), ("lookup_rows", ctypes.POINTER(ctypes.c_uint32)),
        ("lut_num_lookups", ctypes.POINTER(ctypes.c_uint32))]


class SynthParams(ctypes.Structure):
    _fields_ = [("log_n", ctypes.c_uint32), ("num_public_inputs", ctypes.c_uint32),
                ("pct_poseidon", ctypes.c_uint32), ("pct_arithmetic", ctypes.c_uint32),
                ("pct_base_sum", ctypes.c_uint32), ("pct_constant", ctypes.c_uint32), ("seed", ctypes.c_uint64),
                ("pct_extension", ctypes.c_uint32), ("pct_misc", ctypes.c_uint32), ("pct_u32", ctypes.c_uint32),
                ("wide_comparison", ctypes.c_uint32), ("num_luts", ctypes.c_uint32), ("lut_bits", ctypes.c_uint32),
                ("num_lookups", ctypes.c_uint32), ("pad_", ctypes.c_uint32)]


class CircuitConfig:
    """CircuitConfig::standard_recursion_config() (plonky2x DefaultParameters)."""

    def __init__(self, **kw):
        self.num_wires = 135
        self.num_routed_wires = 80
        self.num_constants = 2
        self.num_challenges = 2
        self.rate_bits = 3
        self.cap_height = 4
        self.quotient_degree_factor = 8
        self.fri_pow_bits = 16
        self.fri_num_queries = 28
        self.fri_arity_bits = 4
        self.fri_final_poly_bits = 5
        for k, v in kw.items():
            if not hasattr(self, k):
                raise TypeError("unknown config field %s" % k)
            setattr(self, k, v)

    @property
    def num_partial_products(self):
        return (self.num_routed_wires + self.quotient_degree_factor - 1) // self.quotient_degree_factor - 1


class SyntheticCircuit:
    """A satisfiable nearx-shaped circuit + witness (see csrc/synth.cpp)."""

    def __init__(self, log_n, seed=1, num_public_inputs=4, pct_poseidon=20, pct_arithmetic=30, pct_base_sum=5,
                 pct_constant=5, pct_extension=0, pct_misc=0, pct_u32=0, config=None, wide_comparison=False,
                 num_luts=0, lut_bits=8, num_lookups=0):
        self.config = config or CircuitConfig()
        self.log_n = log_n
        sp = SynthParams(log_n, num_public_inputs, pct_poseidon, pct_arithmetic, pct_base_sum, pct_constant, seed,
                         pct_extension, pct_misc, pct_u32, 1 if wide_comparison else 0, num_luts,
                         lut_bits if num_luts else 0, num_lookups if num_luts else 0, 0)
        ng, ns = ctypes.c_uint32(), ctypes.c_uint32()
        synth_dll.nlx_synth_shape(ctypes.byref(sp), ctypes.byref(ng), ctypes.byref(ns))
        n = 1 << log_n
        self.num_selectors, self.num_gates = ns.value, ng.value
        self.gates = (GateDesc * ng.value)()
        self.k_is = np.zeros(80, dtype=np.uint64)
        self.num_luts = num_luts
        self.num_lookup_selectors = 4 + num_luts if num_luts else 0
        self.constants = np.zeros((ns.value + self.num_lookup_selectors + 2, n), dtype=np.uint64)
        self.sigmas = np.zeros((80, n), dtype=np.uint64)
        self.wires = np.zeros((135, n), dtype=np.uint64)
        self.public_inputs = np.zeros(max(num_public_inputs, 1), dtype=np.uint64)[:num_public_inputs]
        pis = ptr(self.public_inputs) if num_public_inputs else None
        if num_luts:
            # the descriptor's table arrays (kept alive here): sizes, (input, output) pairs, LookupWire rows, lookups per table
            self.lut_sizes = np.full(num_luts, 1 << lut_bits, dtype=np.uint32)
            self.lut_pairs = np.zeros((num_luts << lut_bits, 2), dtype=np.uint16)
            self.lookup_rows = np.zeros((num_luts, 3), dtype=np.uint32)
            self.lut_num_lookups = np.full(num_luts, num_lookups, dtype=np.uint32)
            rc = synth_dll.nlx_synth_circuit_lookups(ctypes.byref(sp), self.gates, ptr(self.k_is), ptr(self.constants),
                                                     ptr(self.sigmas), ptr(self.wires), pis, self.lut_pairs.ctypes.data,
                                                     self.lookup_rows.ctypes.data)
        else:
            rc = synth_dll.nlx_synth_circuit(ctypes.byref(sp), self.gates, ptr(self.k_is), ptr(self.constants),
                                             ptr(self.sigmas), ptr(self.wires), pis)
        if rc != 0:
            raise NlxError(rc, "nlx_synth_circuit failed")

    def set_public_inputs(self, public_inputs):
        """Re-target the witness to other public inputs (keeps it satisfying)."""
        pis = np.ascontiguousarray(public_inputs, dtype=np.uint64)
        if pis.size != self.public_inputs.size:
            raise ValueError("the number of public inputs is fixed by the circuit")
        rc = synth_dll.nlx_synth_set_public_inputs(ptr(self.wires), self.log_n, ptr(pis), pis.size)
        if rc != 0:
            raise NlxError(rc, "nlx_synth_set_public_inputs failed")
        self.public_inputs = pis.copy()

    def desc(self):
        c = self.config
        d = CircuitDesc(self.log_n, c.num_wires, c.num_routed_wires, c.num_constants, c.num_challenges, c.rate_bits,
                        c.cap_height, c.quotient_degree_factor, c.num_partial_products, c.fri_pow_bits,
                        c.fri_num_queries, c.fri_arity_bits, c.fri_final_poly_bits, self.num_selectors,
                        self.num_gates, len(self.public_inputs), self.gates,
                        self.k_is.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)))
        if self.num_luts:
            d.num_luts = self.num_luts
            d.lut_sizes = self.lut_sizes.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
            d.lut_pairs = self.lut_pairs.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16))
            d.lookup_rows = self.lookup_rows.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
            d.lut_num_lookups = self.lut_num_lookups.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
        return d


class CircuitData:
    """Prover-side circuit data resident on the GPU (CircuitBuilder::build output)."""

    def __init__(self, ctx, desc, constants, sigmas):
        self.ctx = ctx
        self._desc = desc  # keeps gates / k_is alive
        h = ctypes.c_void_p()
        ctx.check(dll.nlx_circuit_build(ctx.handle, ctypes.byref(desc), ptr(constants), ptr(sigmas), ctypes.byref(h)))
        self.handle = h
        self.cap_height = desc.cap_height
        self.num_wires = desc.num_wires
        self.n = 1 << desc.degree_bits
        self._buf = np.zeros(dll.nlx_proof_max_bytes(h), dtype=np.uint8)
        ctx._adopt(self)

    @classmethod
    def from_synthetic(cls, ctx, syn):
        return cls(ctx, syn.desc(), syn.constants, syn.sigmas)

    @property
    def circuit_digest(self):
        out = np.zeros(4, dtype=np.uint64)
        dll.nlx_circuit_digest(self.handle, ptr(out))
        return out

    @property
    def constants_sigmas_cap(self):
        out = np.zeros((1 << self.cap_height, 4), dtype=np.uint64)
        dll.nlx_circuit_constants_sigmas_cap(self.handle, ptr(out))
        return out

    def prove(self, wires, public_inputs):
        """prove_with_partition_witness + to_bytes: returns the serialized ProofWithPublicInputs."""
        pis = np.ascontiguousarray(public_inputs, dtype=np.uint64)
        ln = ctypes.c_size_t()
        self.ctx.check(dll.nlx_prove(self.handle, ptr(wires), ptr(pis) if pis.size else None,
                                     self._buf.ctypes.data, self._buf.size, ctypes.byref(ln)))
        return self._buf[:ln.value].tobytes()

    # ---- stage-level calls (the fine seam) ----
    def constants_sigmas_batch(self):
        """The circuit's constants + sigmas commitment (FRI oracle 0); borrowed, lives as long as the circuit."""
        from .batch import PolynomialBatch
        d = self._desc
        h = ctypes.c_void_p(dll.nlx_circuit_constants_sigmas(self.handle))
        pb = PolynomialBatch._adopt_handle(self.ctx, h, d.num_selectors + d.num_constants + d.num_routed_wires, d.degree_bits,
                                           d.rate_bits, d.cap_height)
        pb._borrowed = True
        return pb

    def partial_products_and_zs(self, wires, betas, gammas):
        from .batch import PolynomialBatch
        b = np.ascontiguousarray(betas, dtype=np.uint64)
        g = np.ascontiguousarray(gammas, dtype=np.uint64)
        h = ctypes.c_void_p()
        self.ctx.check(dll.nlx_partial_products_and_zs(self.handle, ptr(wires), ptr(b), ptr(g), ctypes.byref(h)))
        d = self._desc
        n_zs = d.num_challenges * (1 + d.num_partial_products)
        return PolynomialBatch._adopt_handle(self.ctx, h, n_zs, d.degree_bits, d.rate_bits, d.cap_height)

    def quotient_eval(self, wires_batch, zs_batch, betas, gammas, alphas, public_inputs_hash):
        from .batch import PolynomialBatch
        arrs = [np.ascontiguousarray(x, dtype=np.uint64) for x in (betas, gammas, alphas, public_inputs_hash)]
        h = ctypes.c_void_p()
        self.ctx.check(dll.nlx_quotient_eval(self.handle, wires_batch.handle, zs_batch.handle, ptr(arrs[0]), ptr(arrs[1]),
                                             ptr(arrs[2]), ptr(arrs[3]), ctypes.byref(h)))
        d = self._desc
        return PolynomialBatch._adopt_handle(self.ctx, h, d.num_challenges * d.quotient_degree_factor, d.degree_bits,
                                             d.rate_bits, d.cap_height)

    def prove_into(self, wires, public_inputs_ptr):
        """Hot-loop variant: no copies of the result; returns the proof length."""
        ln = ctypes.c_size_t()
        self.ctx.check(dll.nlx_prove(self.handle, ptr(wires), public_inputs_ptr, self._buf.ctypes.data,
                                     self._buf.size, ctypes.byref(ln)))
        return ln.value

    def stage_times(self):
        n = ctypes.c_uint32()
        names = (ctypes.c_char_p * 24)()
        ms = (ctypes.c_float * 24)()
        dll.nlx_prove_stage_times(self.handle, ctypes.byref(n), names, ms)
        return [(names[i].decode(), ms[i]) for i in range(n.value)]

    def close(self):
        if self.handle and self.ctx.handle:
            dll.nlx_circuit_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ChallengerState(ctypes.Structure):
    """nlx_challenger (include/nlx.h)"""
    _fields_ = [("state", ctypes.c_uint64 * 12), ("input", ctypes.c_uint64 * 8), ("n_input", ctypes.c_uint32),
                ("pad0", ctypes.c_uint32), ("output", ctypes.c_uint64 * 8), ("n_output", ctypes.c_uint32),
                ("pad1", ctypes.c_uint32)]


class Challenger:
    """plonky2::iop::challenger::Challenger on the host (nlx_challenger_*), for the stage-by-stage flow."""

    def __init__(self):
        self.s = ChallengerState()
        dll.nlx_challenger_init(ctypes.byref(self.s))

    def observe(self, elements):
        e = np.ascontiguousarray(np.asarray(elements, dtype=np.uint64).reshape(-1))
        rc = dll.nlx_challenger_observe(ctypes.byref(self.s), ptr(e) if e.size else None, e.size)
        if rc != 0:
            raise NlxError(rc, "nlx_challenger_observe")

    def challenges(self, n):
        out = np.zeros(n, dtype=np.uint64)
        rc = dll.nlx_challenger_challenge(ctypes.byref(self.s), ptr(out), n)
        if rc != 0:
            raise NlxError(rc, "nlx_challenger_challenge")
        return out


def hash_no_pad(elements):
    e = np.ascontiguousarray(np.asarray(elements, dtype=np.uint64).reshape(-1))
    out = np.zeros(4, dtype=np.uint64)
    rc = dll.nlx_hash_no_pad(ptr(e) if e.size else None, e.size, ptr(out))
    if rc != 0:
        raise NlxError(rc, "nlx_hash_no_pad")
    return out


class FriParams(ctypes.Structure):
    _fields_ = [(k, ctypes.c_uint32) for k in ("arity_bits", "final_poly_bits", "pow_bits", "num_queries")]


def fri_prove(ctx, oracles, n_next, zeta, openings_zeta, openings_next, params, challenger, cap_bytes=1 << 22):
    """nlx_fri_prove: oracles = PolynomialBatch list, n_next[o] = leading columns of oracle o also opened at g*zeta;
    returns the FriProof bytes and advances `challenger`."""
    hs = (ctypes.c_void_p * len(oracles))(*[o.handle for o in oracles])
    nn = (ctypes.c_uint32 * len(oracles))(*[int(x) for x in n_next])
    z = np.ascontiguousarray(zeta, dtype=np.uint64)
    o0 = np.ascontiguousarray(np.asarray(openings_zeta, dtype=np.uint64).reshape(-1))
    o1 = np.ascontiguousarray(np.asarray(openings_next, dtype=np.uint64).reshape(-1))
    buf = np.zeros(cap_bytes, dtype=np.uint8)
    n = ctypes.c_size_t()
    ctx.check(dll.nlx_fri_prove(ctx.handle, hs, len(oracles), nn, ptr(z), ptr(o0), ptr(o1) if o1.size else None,
                                ctypes.byref(params), ctypes.byref(challenger.s), buf.ctypes.data, buf.size, ctypes.byref(n)))
    return buf[:n.value].tobytes()


class ProveJob(ctypes.Structure):
    _fields_ = [("wires", ctypes.c_void_p), ("public_inputs", ctypes.c_void_p), ("proof_out", ctypes.c_void_p),
                ("proof_cap", ctypes.c_size_t), ("proof_len", ctypes.c_size_t), ("status", ctypes.c_int32)]


def batch_prove(workers, jobs):
    """plonky2x LocalProver::batch_prove over nlx_batch_prove.  workers: CircuitData objects of the same
    circuit on distinct contexts; jobs: list of (wires, public_inputs).  Returns the list of proofs."""
    n = len(jobs)
    arr = (ProveJob * n)()
    bufs, keep = [], []
    cap = dll.nlx_proof_max_bytes(workers[0].handle)
    for i, (wires, pis) in enumerate(jobs):
        pis = np.ascontiguousarray(pis, dtype=np.uint64)
        buf = np.zeros(cap, dtype=np.uint8)
        keep.append(pis)
        bufs.append(buf)
        arr[i].wires = ptr(wires)
        arr[i].public_inputs = ptr(pis) if pis.size else None
        arr[i].proof_out = buf.ctypes.data
        arr[i].proof_cap = cap
    handles = (ctypes.c_void_p * len(workers))(*[w.handle for w in workers])
    rc = dll.nlx_batch_prove(handles, len(workers), arr, n)
    if rc != 0:
        bad = next((i for i in range(n) if arr[i].status != 0), None)
        raise NlxError(rc, "nlx_batch_prove: " + ("job %d failed" % bad if bad is not None else
                                                  dll.nlx_last_error(workers[0].ctx.handle).decode()))
    return [bufs[i][:arr[i].proof_len].tobytes() for i in range(n)]


def pow_grind(ctx, state, pos, bits):
    st = np.ascontiguousarray(state, dtype=np.uint64)
    out = ctypes.c_uint64()
    ctx.check(dll.nlx_pow_grind(ctx.handle, ptr(st), pos, bits, ctypes.byref(out)))
    return out.value
