"""ctypes binding of libnlx.so (include/nlx.h).  Fails loudly when the library is missing."""
import ctypes
import os

import numpy as np

GOLDILOCKS_P = 0xFFFFFFFF00000001
_HERE = os.path.dirname(os.path.abspath(__file__))
# NLX_GL_GENERATOR_SET=2021 selects the library built with the other candidate generator pair (include/nlx_field.h)
_GEN_SET = os.environ.get("NLX_GL_GENERATOR_SET", "7")
LIB_PATH = os.path.join(_HERE, "libnlx.so" if _GEN_SET == "7" else "libnlx_gen%s.so" % _GEN_SET)
if os.environ.get("NLX_BUILD_VARIANT"):  # a kernel-tuning build made by build.py with the same variable (experiments only)
    LIB_PATH = LIB_PATH[:-3] + "_" + os.environ["NLX_BUILD_VARIANT"] + ".so"

# If torch is going to be used in this process (bench.py, multi-GPU dispatch) it must load its
# bundled HIP runtime first; libnlx.so then binds to the same libamdhip64.so.7 instance so
# device pointers and streams can be shared.
if os.environ.get("NLX_SKIP_TORCH_PRELOAD") != "1":
    try:  # pragma: no cover - environment dependent
        import torch  # noqa: F401
    except Exception:  # torch absent: the C ABI works on its own
        pass

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "nlx_amd: HIP extension %s not found. Build it with `python near-light-client_amd/build.py` "
        "(or __graft_entry__.build()). There is no CPU fallback." % LIB_PATH)

_dll = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
# the workload generator (synthetic circuits / witnesses - inputs only, include/nlx_synth.h) lives in its own library
SYNTH_LIB_PATH = os.path.join(_HERE, os.path.basename(LIB_PATH).replace("libnlx", "libnlx_synth", 1))
if not os.path.exists(SYNTH_LIB_PATH):
    raise ImportError("nlx_amd: %s not found. Build it with `python near-light-client_amd/build.py`." % SYNTH_LIB_PATH)
_synth_dll = ctypes.CDLL(SYNTH_LIB_PATH)

u64p = ctypes.POINTER(ctypes.c_uint64)
c_void_pp = ctypes.POINTER(ctypes.c_void_p)

# name -> (restype, argtypes); kept in one table so tests can check every symbol of nlx.h
SIGNATURES = {
    "nlx_version": (ctypes.c_uint32, []),
    "nlx_strerror": (ctypes.c_char_p, [ctypes.c_int32]),
    "nlx_abi_selftest": (ctypes.c_int32, [ctypes.c_int32]),
    "nlx_field_generators": (None, [ctypes.POINTER(ctypes.c_uint64)]),
    "nlx_ctx_create": (ctypes.c_int32, [ctypes.c_int, c_void_pp]),
    "nlx_ctx_destroy": (None, [ctypes.c_void_p]),
    "nlx_last_error": (ctypes.c_char_p, [ctypes.c_void_p]),
    "nlx_ctx_set_stream": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p]),
    "nlx_ctx_synchronize": (ctypes.c_int32, [ctypes.c_void_p]),
    "nlx_ctx_set_priority": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int]),
    "nlx_ctx_set_cu_mask": (ctypes.c_int32, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32), ctypes.c_uint32]),
    "nlx_buf_create": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_size_t, c_void_pp]),
    "nlx_buf_destroy": (None, [ctypes.c_void_p]),
    "nlx_buf_device_ptr": (ctypes.c_void_p, [ctypes.c_void_p]),
    "nlx_buf_size": (ctypes.c_size_t, [ctypes.c_void_p]),
    "nlx_buf_upload": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]),
    "nlx_buf_download": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]),
    "nlx_ctx_memory": (ctypes.c_int32, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]),
    "nlx_ctx_trim": (ctypes.c_int32, [ctypes.c_void_p]),
    "nlx_ctx_kernel_timing": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int]),
    "nlx_ctx_kernel_stats": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64),
                                              ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "nlx_ctx_kernel_units": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_double)]),
    "nlx_field_ops": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "nlx_poseidon_permute_batch": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "nlx_hash_rows": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t,
                                       ctypes.c_void_p]),
    "nlx_merkle_digest_words": (ctypes.c_size_t, [ctypes.c_size_t, ctypes.c_uint32]),
    "nlx_merkle_build": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t,
                                          ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]),
    "nlx_ntt_batch": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32,
                                       ctypes.c_int, ctypes.c_uint64]),
    "nlx_ntt_split_level": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_uint32,
                                              ctypes.c_uint32, ctypes.c_uint32]),
    "nlx_bn254_ntt_batch": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_int,
                                              ctypes.c_uint32]),
    "nlx_bn254_ntt_batch_coset": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_int,
                                                    ctypes.c_uint32, ctypes.c_void_p]),
    "nlx_bn254_msm_g1": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32,
                                           ctypes.c_void_p]),
    "nlx_bn254_g1_sum": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]),
    "nlx_bn254_g2_sum": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]),
    "nlx_bn254_msm_g2": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32,
                                           ctypes.c_void_p]),
    "nlx_bn254_g1_multiples": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]),
    "nlx_bn254_plonk_quotient": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32)]),
    "nlx_bn254_plonk_grand_product": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_uint32] + [ctypes.c_void_p] * 11 + [ctypes.POINTER(ctypes.c_int32)]),
    "nlx_bn254_fr_lincomb": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "nlx_bn254_groth16_quotient": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_uint32] + [ctypes.c_void_p] * 5),
    "nlx_bn254_kzg_open": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "nlx_commit_from_values": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32,
                                                ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, c_void_pp]),
    "nlx_commit_from_coeffs": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32,
                                                ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, c_void_pp]),
    "nlx_commit_destroy": (None, [ctypes.c_void_p]),
    "nlx_commit_get_coeffs": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p]),
    "nlx_commit_get_cap": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p]),
    "nlx_commit_open_rows": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                              ctypes.c_void_p]),
    "nlx_commit_eval_at": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "nlx_commit_get_leaves": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p]),
    "nlx_commit_get_digests": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p]),
    "nlx_circuit_build": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, c_void_pp]),
    "nlx_circuit_destroy": (None, [ctypes.c_void_p]),
    "nlx_circuit_digest": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p]),
    "nlx_circuit_constants_sigmas_cap": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p]),
    "nlx_proof_max_bytes": (ctypes.c_size_t, [ctypes.c_void_p]),
    "nlx_prove": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                   ctypes.POINTER(ctypes.c_size_t)]),
    "nlx_circuit_constants_sigmas": (ctypes.c_void_p, [ctypes.c_void_p]),
    "nlx_partial_products_and_zs": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, c_void_pp]),
    "nlx_quotient_eval": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_void_p, c_void_pp]),
    "nlx_challenger_init": (None, [ctypes.c_void_p]),
    "nlx_challenger_observe": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "nlx_challenger_challenge": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "nlx_hash_no_pad": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "nlx_fri_prove": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]),
    "nlx_batch_prove": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_size_t]),
    "nlx_prove_stage_times": (ctypes.c_int32, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32), ctypes.c_void_p,
                                               ctypes.c_void_p]),
    "nlx_pow_grind": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32,
                                       ctypes.POINTER(ctypes.c_uint64)]),
    "nlx_stark_build": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, c_void_pp]),
    "nlx_stark_destroy": (None, [ctypes.c_void_p]),
    "nlx_stark_proof_max_bytes": (ctypes.c_size_t, [ctypes.c_void_p]),
    "nlx_stark_quotient_kernel": (ctypes.c_int32, [ctypes.c_void_p]),
    "nlx_stark_prove": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                         ctypes.POINTER(ctypes.c_size_t)]),
    "nlx_stark_prove_rounds": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]),
    "nlx_stark_stage_times": (ctypes.c_int32, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32), ctypes.c_void_p,
                                               ctypes.c_void_p]),
    "nlx_stark_batch_prove": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_size_t]),
    "nlx_logup_multiplicities": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p,
                                                  ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]),
    "nlx_logup_round_cols": (ctypes.c_uint32, [ctypes.c_uint32, ctypes.c_uint32]),
    "nlx_logup_round": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p,
                                         ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p,
                                         ctypes.c_void_p]),
    "nlx_fp25519_chip_trace": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]),
    "nlx_ed25519_trace": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]),
    "nlx_ed25519_bind_round": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.c_void_p]),
    "nlx_sha256_bind_round": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_void_p]),
    "nlx_sha256_trace": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32,
                                          ctypes.c_void_p, ctypes.c_void_p]),
    "nlx_sha512_bind_round": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_void_p]),
    "nlx_sha512_trace": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32,
                                          ctypes.c_void_p, ctypes.c_void_p]),
}

# include/nlx_synth.h (libnlx_synth.so)
SYNTH_SIGNATURES = {
    "nlx_synth_stark_trace": (ctypes.c_int32, [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_void_p,
                                               ctypes.c_void_p, ctypes.c_void_p]),
    "nlx_synth_shape": (None, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]),
    "nlx_synth_circuit": (ctypes.c_int32, [ctypes.c_void_p] * 7),
    "nlx_synth_circuit_lookups": (ctypes.c_int32, [ctypes.c_void_p] * 9),
    "nlx_synth_set_public_inputs": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32]),
}

for _lib, _sigs in ((_dll, SIGNATURES), (_synth_dll, SYNTH_SIGNATURES)):
    for _name, (_res, _args) in _sigs.items():
        _fn = getattr(_lib, _name)  # AttributeError here = ABI mismatch: fail at import
        _fn.restype = _res
        _fn.argtypes = _args

dll = _dll
synth_dll = _synth_dll


class NlxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("nlx error %d (%s): %s" % (code, dll.nlx_strerror(code).decode(), msg))
        self.code = code


def ptr(a):
    """Raw address of a numpy array, a torch tensor (host or device) or None."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        if a.dtype != np.uint64 or not a.flags["C_CONTIGUOUS"]:
            raise TypeError("expected a C-contiguous uint64 array")
        return a.ctypes.data
    if hasattr(a, "data_ptr"):
        if not a.is_contiguous():
            raise TypeError("expected a contiguous tensor")
        return a.data_ptr()
    if isinstance(a, int):
        return a
    raise TypeError("unsupported buffer type %r" % type(a))


class DeviceBuffer:
    """nlx_buf: a device buffer owned by a Context (for callers without HIP bindings of their own)."""

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        h = ctypes.c_void_p()
        ctx.check(dll.nlx_buf_create(ctx.handle, nbytes, ctypes.byref(h)))
        self.handle = h
        self.nbytes = nbytes
        ctx._adopt(self)

    @classmethod
    def from_array(cls, ctx, a):
        a = np.ascontiguousarray(a)
        b = cls(ctx, a.nbytes)
        b.upload(a)
        return b

    @property
    def ptr(self):
        return dll.nlx_buf_device_ptr(self.handle)

    def upload(self, a, offset=0):
        a = np.ascontiguousarray(a)
        self.ctx.check(dll.nlx_buf_upload(self.handle, offset, a.ctypes.data, a.nbytes))

    def download(self, dtype=np.uint64, offset=0, nbytes=None):
        nbytes = self.nbytes - offset if nbytes is None else nbytes
        out = np.empty(nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        self.ctx.check(dll.nlx_buf_download(self.handle, offset, out.ctypes.data, out.nbytes))
        return out

    def close(self):
        if self.handle and self.ctx.handle:
            dll.nlx_buf_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One per HIP device and host thread (nlx.h threading contract)."""

    def __init__(self, device=0):
        h = ctypes.c_void_p()
        rc = dll.nlx_ctx_create(int(device), ctypes.byref(h))
        if rc != 0:
            raise NlxError(rc, "nlx_ctx_create(device=%d) failed: no usable gfx950 device "
                               "(there is no CPU fallback)" % device)
        self.handle = h
        self.device = device
        # handles created on this context (commitments, circuits); closed before the context so that
        # garbage-collection order at interpreter exit can never free a context under its children
        import weakref
        self._children = weakref.WeakSet()

    def _adopt(self, child):
        self._children.add(child)

    def memory(self):
        """(bytes held from the driver, bytes in use by live handles and tables)"""
        r, u = ctypes.c_size_t(), ctypes.c_size_t()
        self.check(dll.nlx_ctx_memory(self.handle, ctypes.byref(r), ctypes.byref(u)))
        return r.value, u.value

    def trim(self):
        """Return cached free device blocks to the driver."""
        self.check(dll.nlx_ctx_trim(self.handle))

    def check(self, rc):
        if rc != 0:
            raise NlxError(rc, dll.nlx_last_error(self.handle).decode())

    def synchronize(self):
        self.check(dll.nlx_ctx_synchronize(self.handle))

    def set_cu_mask(self, cus):
        """run this context's stream on the given compute units only (iterable of CU indices; nlx_ctx_set_cu_mask)"""
        words = [0] * 8
        for i in cus:
            words[i // 32] |= 1 << (i % 32)
        arr = (ctypes.c_uint32 * 8)(*words)
        self.check(dll.nlx_ctx_set_cu_mask(self.handle, arr, 8))

    def set_priority(self, high=True):
        """scheduling priority of the context's stream (nlx_ctx_set_priority); call before queuing work"""
        self.check(dll.nlx_ctx_set_priority(self.handle, 1 if high else 0))

    def set_stream(self, hip_stream):
        self.check(dll.nlx_ctx_set_stream(self.handle, hip_stream))

    def kernel_timing(self, enable=True):
        self.check(dll.nlx_ctx_kernel_timing(self.handle, 1 if enable else 0))

    def kernel_stats(self, name):
        """(calls, total_ms, algorithmic_bytes) of the named kernel since timing was enabled."""
        n, ms, b = ctypes.c_uint64(), ctypes.c_double(), ctypes.c_double()
        self.check(dll.nlx_ctx_kernel_stats(self.handle, name.encode(), ctypes.byref(n), ctypes.byref(ms), ctypes.byref(b)))
        return n.value, ms.value, b.value

    def kernel_units(self, name):
        """work units other than bytes of the named kernel's samples (Poseidon permutations for the hashing kernels)"""
        u = ctypes.c_double()
        self.check(dll.nlx_ctx_kernel_units(self.handle, name.encode(), ctypes.byref(u)))
        return u.value

    def close(self):
        if self.handle:
            for child in list(self._children):
                try:
                    child.close()
                except Exception:
                    pass
            dll.nlx_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
