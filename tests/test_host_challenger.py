"""The library's HOST transcript (csrc/transcript.hpp Challenger behind nlx_challenger_*; its permutation's linear layer is AVX2 code
on x86-64) against the oracle's Challenger: same challenges for the golden sequence and for random observe / challenge patterns,
including inputs that are not canonical field elements on the wire of the linear layer (carry edges).  Needs no GPU."""
import ctypes

import numpy as np

from conftest import P


def test_host_challenger_equals_oracle(nlx, orc, golden):
    dll = nlx.lib.dll

    class Ch(ctypes.Structure):
        _fields_ = [("state", ctypes.c_uint64 * 12), ("in_buf", ctypes.c_uint64 * 8), ("n_in", ctypes.c_uint32),
                    ("out_buf", ctypes.c_uint64 * 8), ("n_out", ctypes.c_uint32)]
    rng = np.random.default_rng(12)
    edge = np.array([0, 1, P - 1, P - 2, 0xFFFFFFFF, 0x100000000, 0xFFFFFFFE00000001, 0xFFFFFFFF00000000, 0x8000000080000000], dtype=np.uint64)
    for trial in range(40):
        c = Ch()
        dll.nlx_challenger_init(ctypes.byref(c))
        ref = orc.Challenger()
        for _ in range(int(rng.integers(1, 12))):
            k = int(rng.integers(0, 40))
            xs = rng.integers(0, P, k, dtype=np.uint64)
            if k and trial % 3 == 0:
                xs[rng.integers(0, k, max(1, k // 3))] = rng.choice(edge, max(1, k // 3))
            if k:
                assert dll.nlx_challenger_observe(ctypes.byref(c), xs.ctypes.data_as(ctypes.c_void_p), k) == 0
                for x in xs:
                    ref.observe(int(x))
            m = int(rng.integers(0, 11))
            out = np.zeros(max(m, 1), dtype=np.uint64)
            assert dll.nlx_challenger_challenge(ctypes.byref(c), out.ctypes.data_as(ctypes.c_void_p), m) == 0
            assert [int(v) for v in out[:m]] == [ref.challenge() for _ in range(m)]
