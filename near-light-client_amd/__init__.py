"""nlx_amd - MI355X-native prover backend for the nearx plonky2x circuits (host-side mirror).

The compute path is the HIP library `libnlx.so` (C ABI in include/nlx.h).  This package is
the thin host layer above it, mirroring the plonky2 / plonky2x interface names the reference
calls (nearx/src/test_utils.rs:29,62,66): PolynomialBatch, MerkleTree, prove, ...
There is NO CPU fallback: importing `nlx_amd.lib` raises if libnlx.so is missing, and creating
a Context raises if no gfx950 device is usable.
"""
from . import _lib as lib  # noqa: F401  (raises loudly when the HIP extension is absent)
from ._lib import Context, DeviceBuffer, NlxError, GOLDILOCKS_P  # noqa: F401
from .batch import PolynomialBatch, MerkleTree, poseidon_permute, hash_rows, ntt, field_ops, bn254_ntt, bn254_pack, bn254_unpack, BN254_R, bn254_msm_g1, bn254_g1_pack, bn254_g1_unpack, bn254_g1_sum, bn254_g1_multiples, bn254_g2_pack, bn254_g2_unpack, bn254_msm_g2, bn254_g2_sum, BN254_Q, bn254_plonk_quotient, bn254_kzg_open  # noqa: F401
from .plonk import CircuitConfig, CircuitData, SyntheticCircuit, pow_grind, batch_prove, ProveJob  # noqa: F401
from .stark import Air, Stark, StarkConfig, StarkProver, fibonacci_air, fibonacci_trace, wide_air, wide_trace  # noqa: F401
from . import sha256_air, sha512_air, logup, fp25519, ed25519_air, nearx_io, near_protocol, stark, plonk, succinct_io, split_ntt, bn254_plonk  # noqa: F401,E402
