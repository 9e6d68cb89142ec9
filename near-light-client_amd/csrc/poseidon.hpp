// Poseidon-12 permutation over Goldilocks for gfx950 (one sponge state per lane, 12 x u64 in
// VGPRs, round constants read through the scalar cache: every lane of a wave is in the same
// round, so the constant operand is wave-uniform and costs no vector memory traffic).
//
// Replaces plonky2::hash::poseidon::Poseidon::poseidon and hashing::{hash_n_to_m_no_pad, compress}
// (plonky2 0.1.4 @ d2598bd, /root/reference/Cargo.lock:4864-4866).
// Schedule: 4 full rounds, 22 partial rounds, 4 full rounds; S-box x^7;
// MDS = circ(17,15,41,16,2,28,13,13,39,18,34,20) + diag(8,0,...,0).
#pragma once
#include "gl.hpp"
#if defined(__HIP__)
#include "gl32.hpp"
#endif
#include "poseidon_constants.inc"

namespace poseidon {

constexpr int WIDTH = 12;
constexpr int RATE = 8;
constexpr int N_ROUNDS = 30;
constexpr int HALF_FULL = 4;
constexpr int N_PARTIAL = 22;

#if defined(__HIP__)
__constant__ static const uint64_t RC_DEV[360] = NLX_POSEIDON_ROUND_CONSTANTS_INIT;
// Constants of the device schedule, GENERATED (tools/gen_poseidon_blocks.py -> poseidon_blocks.inc, checked there against the
// naive permutation and upstream's known answers before they are written):
//  * LAYER_RCB_DEV: seeds of a single layer's recombination (gl32::mds_layer_mfma): for layer l = 1 .. 29 and output r the low
//    ([l * 24 + r]) and high ([l * 24 + 12 + r]) half of the constant round l adds, zero-extended to the seed's 64-bit scalar
//    operand.  Layers 5 .. 25 are inside the fused blocks (rows unused); layer 26 also settles the offset the blocks carried.
//  * BLOCK_*: the fused partial rounds (gl32::partial_block3): rounds 4 .. 24 run as seven blocks of three rounds.  Only element
//    0 passes an S-box in a partial round, so the constants of elements 1 .. 11 are MOVED, not changed in effect: they ride along
//    as a known offset delta of the state (the kernel holds t - delta, delta[0] = 0) together with what the chain seeds add,
//    and are settled once, in layer 26's twelve constants.  The permutation's outputs are the same field elements.
__constant__ static const uint64_t LAYER_RCB_DEV[31 * 24] = NLX_POSEIDON_LAYER_RCB_INIT;
__constant__ static const uint32_t BLOCK_FRAGMENTS_DEV[NLX_POSEIDON_BLOCK_K * 64 * 4] = NLX_POSEIDON_BLOCK_FRAGMENTS_INIT;
__constant__ static const uint32_t BLOCK_SEED_DEV[16] = NLX_POSEIDON_BLOCK_SEED_INIT;
__constant__ static const uint64_t BLOCK_KAPPA_DEV[NLX_POSEIDON_N_BLOCKS * 2 * NLX_POSEIDON_BLOCK_K] = NLX_POSEIDON_BLOCK_KAPPA_INIT;
static_assert(NLX_POSEIDON_BLOCK_K == 3 && NLX_POSEIDON_BLOCK_DMAX < (1 << 20), "gl32::partial_block3 is written for these");

__device__ __forceinline__ gl32::BlockOperands block_operands(uint32_t lane = threadIdx.x & 63) {
    gl32::BlockOperands op;
#pragma unroll
    for (int p = 0; p < 3; p++) {
        const uint32_t* f = BLOCK_FRAGMENTS_DEV + (p * 64 + lane) * 4;
        op.a[p].x = (int)f[0]; op.a[p].y = (int)f[1]; op.a[p].z = (int)f[2]; op.a[p].w = (int)f[3];
    }
    // lane >> 6 is zero; where the caller made the lane index opaque (k_quotient: see gate_poseidon_mx) it ties the seeds to that
    // point as well, elsewhere it folds away
    const uint32_t zero = lane >> 6;
#pragma unroll
    for (int i = 0; i < 16; i++) op.seed[i] = (int)(BLOCK_SEED_DEV[i] + zero);
    return op;
}
#endif
static const uint64_t RC_HOST[360] = NLX_POSEIDON_ROUND_CONSTANTS_INIT;

GL_HD const uint64_t* rc_table() {
#if defined(__HIP_DEVICE_COMPILE__)
    return RC_DEV;
#else
    return RC_HOST;
#endif
}

GL_HD uint64_t sbox7(uint64_t x) {
    uint64_t x2 = gl::mul_loose(x, x);
    uint64_t x4 = gl::mul_loose(x2, x2);
    uint64_t x3 = gl::mul_loose(x, x2);
    return gl::mul_loose(x3, x4);
}

// Linear layer on loose inputs.  Each lane is split into 32-bit halves; the 12 products by the
// (<= 6-bit) circulant entries accumulate in u64 without overflow (12 * 2^32 * 41 < 2^42), then
// lo + 2^32*hi is folded once: a 96-bit reduction instead of twelve 128-bit ones.
GL_HD uint64_t mds_fold(uint64_t al, uint64_t ah) {
    // value = al + ah * 2^32, with ah < 2^42
    uint64_t l = al + (ah << 32);
    uint64_t h = (ah >> 32) + (l < al ? 1u : 0u);  // < 2^11
    uint64_t t1 = (h << 32) - h;                    // h * (2^32 - 1)
    uint64_t res = l + t1;
    if (res < t1) res += gl::EPS;
    return res;
}
// (Round 4 tried the HOST's layer as AVX2 4-lane multiply-adds over a doubled array of the halves - 72 vpmuludq instead of 288
// scalar multiply-adds: the transcript got SLOWER on the GPU box's EPYC, 2.99 -> 3.59 ms for the 19 000 openings of the SHA-512
// proof; the 32-byte loads straddle the 8-byte stores that built the array, which defeats store forwarding.  Dropped; the scalar
// loop below is what hipcc's host pass vectorises itself.  tests/test_host_challenger.py pins the host transcript either way.)
GL_HD void mds_layer(uint64_t (&s)[12]) {
    constexpr uint32_t C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    uint32_t lo[12], hi[12];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        lo[i] = (uint32_t)s[i];
        hi[i] = (uint32_t)(s[i] >> 32);
    }
#pragma unroll
    for (int r = 0; r < 12; r++) {
        uint64_t al = 0, ah = 0;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            al += (uint64_t)lo[(i + r) % 12] * C[i];
            ah += (uint64_t)hi[(i + r) % 12] * C[i];
        }
        if (r == 0) {
            al += (uint64_t)lo[0] * 8u;
            ah += (uint64_t)hi[0] * 8u;
        }
        s[r] = mds_fold(al, ah);
    }
}

// In-place permutation; input loose, output loose.
GL_HD void permute_loose(uint64_t (&s)[12]) {
    const uint64_t* rc = rc_table();
#if defined(__HIP_DEVICE_COMPILE__)
    // device: {lo, hi} u32 pairs with hand-placed carry chains (gl32.hpp); the linear layer runs on the matrix cores
    // (gl32::mds_layer_mfma: one state per lane, all of a wave's lanes together) and every round's constants seed the
    // recombination of the PREVIOUS layer (only round 0 adds them explicitly); rounds 4 .. 24 run as fused blocks of three
    // (gl32::partial_block3).  CONTRACT: every lane of the wave reaches this call with EXEC all ones (gl32.hpp) - callers clamp
    // spare lanes to the last item instead of returning.
#if defined(NLX_DEBUG)
    if (__builtin_amdgcn_read_exec() != ~0ull) __builtin_trap();
#endif
    const uint64_t* rcb = LAYER_RCB_DEV;
    const gl32::i32x4_t a = gl32::mds_a_fragment();
    const gl32::BlockOperands op = block_operands();
    gl32::F t[12];
#pragma unroll
    for (int i = 0; i < 12; i++) t[i] = gl32::add_const(gl32::from_u64(s[i]), rc[i]);
#pragma unroll 1
    for (int r = 0; r < HALF_FULL; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) t[i] = gl32::sbox7(t[i]);
        gl32::mds_layer_mfma<2>(t, a, rcb + (r + 1) * 24);
    }
#pragma unroll 1
    for (int b = 0; b < NLX_POSEIDON_N_BLOCKS; b++)   // rounds 4 .. 24, three at a time
        gl32::partial_block3<NLX_POSEIDON_BLOCK_GAMMA21>(t, op, BLOCK_KAPPA_DEV + b * 6);
    t[0] = gl32::sbox7(t[0]);                          // round 25
    gl32::mds_layer_mfma<2>(t, a, rcb + (HALF_FULL + N_PARTIAL) * 24);
#pragma unroll 1
    for (int r = HALF_FULL + N_PARTIAL; r < N_ROUNDS - 1; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) t[i] = gl32::sbox7(t[i]);
        gl32::mds_layer_mfma<2>(t, a, rcb + (r + 1) * 24);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) t[i] = gl32::sbox7(t[i]);
    gl32::mds_layer_mfma<0>(t, a, rcb);
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl32::to_u64(t[i]);
#else
#pragma unroll 1
    for (int r = 0; r < HALF_FULL; r++) {
        for (int i = 0; i < 12; i++) s[i] = sbox7(gl::add_loose(s[i], rc[r * 12 + i]));
        mds_layer(s);
    }
    for (int r = HALF_FULL; r < HALF_FULL + N_PARTIAL; r++) {
        for (int i = 0; i < 12; i++) s[i] = gl::add_loose(s[i], rc[r * 12 + i]);
        s[0] = sbox7(s[0]);
        mds_layer(s);
    }
    for (int r = HALF_FULL + N_PARTIAL; r < N_ROUNDS; r++) {
        for (int i = 0; i < 12; i++) s[i] = sbox7(gl::add_loose(s[i], rc[r * 12 + i]));
        mds_layer(s);
    }
#endif
}

GL_HD void permute(uint64_t (&s)[12]) {
    permute_loose(s);
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl::canon(s[i]);
}

}  // namespace poseidon
