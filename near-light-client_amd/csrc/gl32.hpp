// Goldilocks arithmetic on 32-bit limb pairs with hand-placed carry chains (gfx950 inline asm).
//
// Measured on MI355X (tools/ubench, profiles/r02_valu_ubench.txt): at full occupancy v_mad_u64_u32 issues at ~2.0 ns per
// wave-instruction per SIMD like every other integer multiply - and so does every VCC-chained add / subtract / compare
// (v_add_co_u32 + v_addc_co_u32: 1.99 ns each, v_cmp + v_cndmask: 1.82 ns); only carry-free ALU ops reach 1.17 ns.  A
// field multiply therefore costs its INSTRUCTION COUNT x 2 ns, whatever the instructions are, and a multiply-add that
// replaces several carry-chain instructions is a straight win: `r = base + w2 * (2^32 - 1)` is ONE v_mad_u64_u32 (its
// carry-out lands in VCC) instead of four add / subtract-with-carry instructions, and `r += carry ? 2^32 - 1 : 0` is a
// v_cndmask + one more v_mad_u64_u32 (mask * 1 + r) instead of three.  Multiply + reduction: 16 instructions (20 in
// round 1, ~29 from hipcc's u64 code); the fold of an MDS accumulator pair: 4 (9 in round 1).
#pragma once
#include "gl.hpp"

namespace gl32 {

struct F {
    uint32_t lo, hi;  // value = lo + hi * 2^32, any u64 ("loose")
};

__device__ __forceinline__ F from_u64(uint64_t x) { return F{(uint32_t)x, (uint32_t)(x >> 32)}; }
__device__ __forceinline__ uint64_t to_u64(F x) { return ((uint64_t)x.hi << 32) | x.lo; }

// base + w2 * EPS (mod p) -> loose.  2^64 = EPS (mod p): the multiply-add's carry out of 2^64 is worth EPS, and adding it
// cannot carry again (base + w2 * EPS - 2^64 <= 2^64 - 2^33).  Three instructions: multiply-add, carry -> mask, mask * 1 + r.
__device__ __forceinline__ F add_w2_eps(uint64_t base, uint32_t w2) {
    uint64_t r;
    uint32_t t;
    asm("v_mad_u64_u32 %[r], vcc, %[w2], -1, %[base]\n\t"
        "v_cndmask_b32_e64 %[t], 0, -1, vcc\n\t"
        "v_mad_u64_u32 %[r], vcc, %[t], 1, %[r]"
        : [r] "=&v"(r), [t] "=&v"(t)
        : [w2] "v"(w2), [base] "v"(base)
        : "vcc");
    return from_u64(r);
}

// (w3:w2:w1:w0) mod p -> loose
//   r = (w1:w0) - w3 [borrow -> -EPS] + (w2 * EPS) [carry -> +EPS]
__device__ __forceinline__ F reduce128(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
    uint32_t r0, r1, t0;
    asm("v_sub_co_u32 %[r0], vcc, %[w0], %[w3]\n\t"
        "v_subbrev_co_u32 %[r1], vcc, 0, %[w1], vcc\n\t"
        "v_cndmask_b32_e64 %[t0], 0, -1, vcc\n\t"
        "v_sub_co_u32 %[r0], vcc, %[r0], %[t0]\n\t"
        "v_subbrev_co_u32 %[r1], vcc, 0, %[r1], vcc"
        : [r0] "=&v"(r0), [r1] "=&v"(r1), [t0] "=&v"(t0)
        : [w0] "v"(w0), [w1] "v"(w1), [w3] "v"(w3)
        : "vcc");
    return add_w2_eps(((uint64_t)r1 << 32) | r0, w2);
}

// loose * loose -> loose
__device__ __forceinline__ F mul(F a, F b) {
    uint64_t t = (uint64_t)a.lo * b.lo;
    uint64_t m = (uint64_t)a.lo * b.hi;
    uint64_t x = (uint64_t)a.hi * b.hi;
    uint64_t mid;
    uint32_t x1;
    // mid = a.hi * b.lo + m, its carry (weight 2^96) goes straight into the top limb
    asm("v_mad_u64_u32 %[mid], vcc, %[ah], %[bl], %[m]\n\t"
        "v_addc_co_u32 %[x1], vcc, 0, %[xh], vcc"
        : [mid] "=&v"(mid), [x1] "=v"(x1)
        : [ah] "v"(a.hi), [bl] "v"(b.lo), [m] "v"(m), [xh] "v"((uint32_t)(x >> 32))
        : "vcc");
    // limbs w1..w3, then (w1:w0) - w3 [borrow -> -EPS]; the + w2 * EPS [carry -> +EPS] half is add_w2_eps
    uint32_t r0, r1, w1, w2, w3, t0;
    asm("v_add_co_u32 %[w1], vcc, %[th], %[m0]\n\t"
        "v_addc_co_u32 %[w2], vcc, %[x0], %[m1], vcc\n\t"
        "v_addc_co_u32 %[w3], vcc, 0, %[x1], vcc\n\t"
        "v_sub_co_u32 %[r0], vcc, %[w0], %[w3]\n\t"
        "v_subbrev_co_u32 %[r1], vcc, 0, %[w1], vcc\n\t"
        "v_cndmask_b32_e64 %[t0], 0, -1, vcc\n\t"
        "v_sub_co_u32 %[r0], vcc, %[r0], %[t0]\n\t"
        "v_subbrev_co_u32 %[r1], vcc, 0, %[r1], vcc"
        : [r0] "=&v"(r0), [r1] "=&v"(r1), [w1] "=&v"(w1), [w2] "=&v"(w2), [w3] "=&v"(w3), [t0] "=&v"(t0)
        : [w0] "v"((uint32_t)t), [th] "v"((uint32_t)(t >> 32)), [m0] "v"((uint32_t)mid), [x0] "v"((uint32_t)x),
          [m1] "v"((uint32_t)(mid >> 32)), [x1] "v"(x1)
        : "vcc");
    return add_w2_eps(((uint64_t)r1 << 32) | r0, w2);
}

// loose + canonical constant -> loose
__device__ __forceinline__ F add_const(F a, uint64_t c) {
    F r;
    uint32_t t;
    asm("v_add_co_u32 %[r0], vcc, %[a0], %[c0]\n\t"
        "v_addc_co_u32 %[r1], vcc, %[a1], %[c1], vcc\n\t"
        "v_cndmask_b32_e64 %[t], 0, -1, vcc\n\t"
        "v_add_co_u32 %[r0], vcc, %[r0], %[t]\n\t"
        "v_addc_co_u32 %[r1], vcc, 0, %[r1], vcc"
        : [r0] "=&v"(r.lo), [r1] "=&v"(r.hi), [t] "=&v"(t)
        : [a0] "v"(a.lo), [a1] "v"(a.hi), [c0] "s"((uint32_t)c), [c1] "v"((uint32_t)(c >> 32))
        : "vcc");
    return r;
}

// same with a per-lane (vector) constant
__device__ __forceinline__ F add_const_v(F a, uint64_t c) {
    F r;
    uint32_t t;
    asm("v_add_co_u32 %[r0], vcc, %[a0], %[c0]\n\t"
        "v_addc_co_u32 %[r1], vcc, %[a1], %[c1], vcc\n\t"
        "v_cndmask_b32_e64 %[t], 0, -1, vcc\n\t"
        "v_add_co_u32 %[r0], vcc, %[r0], %[t]\n\t"
        "v_addc_co_u32 %[r1], vcc, 0, %[r1], vcc"
        : [r0] "=&v"(r.lo), [r1] "=&v"(r.hi), [t] "=&v"(t)
        : [a0] "v"(a.lo), [a1] "v"(a.hi), [c0] "v"((uint32_t)c), [c1] "v"((uint32_t)(c >> 32))
        : "vcc");
    return r;
}

__device__ __forceinline__ F sbox7(F x) {
    F x2 = mul(x, x);
    F x4 = mul(x2, x2);
    F x3 = mul(x, x2);
    return mul(x3, x4);
}

// value = al + ah * 2^32 (al, ah u64 accumulators below 2^42: twelve products of a 32-bit limb and a constant < 2^6, plus a
// round constant's limb) -> loose.  2^64 = EPS (mod p), so ah's high word folds in first, as a multiply-add that cannot
// carry (al + ah1 * EPS < 2^42 + 2^42); ah's low word then lands on the high word of the sum, and only that addition can
// carry - once, worth EPS, and adding it cannot carry again.  Four instructions (five when the limbs were summed first).
__device__ __forceinline__ F fold_acc(uint64_t al, uint64_t ah) {
    uint64_t t;
    asm("v_mad_u64_u32 %[t], vcc, %[ah1], -1, %[al]" : [t] "=&v"(t) : [ah1] "v"((uint32_t)(ah >> 32)), [al] "v"(al) : "vcc");
    uint32_t r1, m;
    asm("v_add_co_u32 %[r1], vcc, %[t1], %[ah0]\n\t"
        "v_cndmask_b32_e64 %[m], 0, -1, vcc"
        : [r1] "=&v"(r1), [m] "=&v"(m)
        : [t1] "v"((uint32_t)(t >> 32)), [ah0] "v"((uint32_t)ah)
        : "vcc");
    uint64_t r;
    asm("v_mad_u64_u32 %[r], vcc, %[m], 1, %[base]" : [r] "=&v"(r) : [m] "v"(m), [base] "v"(((uint64_t)r1 << 32) | (uint32_t)t) : "vcc");
    return from_u64(r);
}

// Linear layer; `rc_next` (12 canonical constants, wave-uniform, or nullptr) is the NEXT round's
// constant vector: it seeds the accumulators, so the constant addition costs no vector instruction
// (the 64-bit addend operand of the first multiply-add comes straight from scalar registers).
__device__ __forceinline__ void mds_layer(F (&s)[12], const uint64_t* __restrict__ rc_next = nullptr) {
    constexpr uint32_t C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    F o[12];
#pragma unroll
    for (int r = 0; r < 12; r++) {
        uint64_t al = 0, ah = 0;
        if (rc_next) {
            const uint64_t c = rc_next[r];
            al = (uint32_t)c;
            ah = c >> 32;
        }
#pragma unroll
        for (int i = 0; i < 12; i++) {
            al += (uint64_t)s[(i + r) % 12].lo * C[i];
            ah += (uint64_t)s[(i + r) % 12].hi * C[i];
        }
        if (r == 0) {
            al += (uint64_t)s[0].lo * 8u;
            ah += (uint64_t)s[0].hi * 8u;
        }
        o[r] = fold_acc(al, ah);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = o[i];
}


// ---- the linear layer on the MATRIX cores ----------------------------------------------------------------------------
// out[r] = sum_j M[r][j] s_j with M[r][j] = C[(j - r) mod 12] (+ 8 at [0][0]) and 64-bit s_j = sum_b 2^(8b) byte_b(s_j): eight
// products of the constant 12 x 12 matrix with the state's byte planes - ONE v_mfma_i32_32x32x32_i8 each for all 64 states of
// a wave (one state per lane, as everywhere in the hashing kernels).  No lane ever moves data: column n of B is supplied by
// lane n (k < 16) and lane n + 32 (k >= 16), each its OWN state's twelve bytes of the plane, and A is placed so that output r
// of the state in lane n + 32 h lands in row (r & 3) + 8 (r >> 2) + 4 h - which the 32 x 32 accumulator layout (col =
// lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)) hands back to that same lane as register r.  A lane's outputs
// depend on its own inputs only, so lanes that have left the kernel do no harm.  The instruction's bytes are signed: the
// planes are biased by 128 (xor 0x80) and 128 * rowsum * (1 + 2^8 + 2^16 + 2^24) rides in the table that seeds the
// recombination (RCB, with the next round's constants).  Per layer: 24 xor + 48 v_perm + 8 MFMA + 96 v_mad_i64_i32 + the
// 4-instruction folds = ~285 vector instructions where the multiply-accumulate form above needs ~430; the matrix pipe runs
// beside the vector pipe.  Measured (tools/ubench, profiles/r03_poseidon_occupancy.txt): 1.74 -> 2.09 G permutations/s,
// bit-identical outputs.
typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef int i32x16_t __attribute__((ext_vector_type(16)));

// this lane's A operand: A[row = lane & 31][k = 16 (lane >> 5) + j], j = 0 .. 15 (the same k order as the B operand below,
// whatever the hardware's order inside a lane's sixteen bytes is)
__device__ __forceinline__ i32x4_t mds_a_fragment() {
    constexpr int C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    const uint32_t lane = threadIdx.x & 63, rho = lane & 31, h = lane >> 5;
    const uint32_t g = (rho >> 2) & 1, r = (rho & 3) + 4 * (rho >> 3);
    uint32_t w[4] = {0, 0, 0, 0};
    if (h == g && rho < 24) {
#pragma unroll
        for (int j = 0; j < 12; j++) {
            uint32_t m = 0;
#pragma unroll
            for (int rr = 0; rr < 12; rr++)
                if ((uint32_t)rr == r) m = (uint32_t)C[(j - rr + 12) % 12] + ((rr == 0 && j == 0) ? 8u : 0u);
            w[j >> 2] |= m << (8 * (j & 3));
        }
    }
    i32x4_t a;
    a.x = (int)w[0]; a.y = (int)w[1]; a.z = (int)w[2]; a.w = (int)w[3];
    return a;
}
// 4 x 4 byte transpose: p[b] = (x0.byte b, x1.byte b, x2.byte b, x3.byte b)
__device__ __forceinline__ void transpose_bytes4(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t (&p)[4]) {
    const uint32_t t01l = __builtin_amdgcn_perm(x1, x0, 0x05010400u), t01h = __builtin_amdgcn_perm(x1, x0, 0x07030602u);
    const uint32_t t23l = __builtin_amdgcn_perm(x3, x2, 0x05010400u), t23h = __builtin_amdgcn_perm(x3, x2, 0x07030602u);
    p[0] = __builtin_amdgcn_perm(t23l, t01l, 0x05040100u);
    p[1] = __builtin_amdgcn_perm(t23l, t01l, 0x07060302u);
    p[2] = __builtin_amdgcn_perm(t23h, t01h, 0x05040100u);
    p[3] = __builtin_amdgcn_perm(t23h, t01h, 0x07060302u);
}
__device__ __forceinline__ int opaque_sgpr(int v) {   // a wave-uniform constant the compiler must treat as a register, so
    asm("" : "+s"(v));                                 // that d * 2^(8k) + acc stays ONE v_mad_i64_i32 (not shifts and adds)
    return v;
}
// rcb: 24 wave-uniform words: [r] = bias + low half of the NEXT round's constant r, [12 + r] = bias + its high half
__device__ __forceinline__ void mds_layer_mfma(F (&s)[12], const i32x4_t a, const uint64_t* __restrict__ rcb) {
    const int m0 = opaque_sgpr(1), m8 = opaque_sgpr(1 << 8), m16 = opaque_sgpr(1 << 16), m24 = opaque_sgpr(1 << 24);
    i32x16_t zero;
#pragma unroll
    for (int i = 0; i < 16; i++) zero[i] = 0;
    uint64_t acc[2][12];
#pragma unroll
    for (int half = 0; half < 2; half++) {   // planes 0 .. 3 come from the low words, 4 .. 7 from the high words
        uint32_t pl[4][3];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            uint32_t p[4];
            if (half == 0)
                transpose_bytes4(s[4 * q].lo ^ 0x80808080u, s[4 * q + 1].lo ^ 0x80808080u, s[4 * q + 2].lo ^ 0x80808080u,
                                 s[4 * q + 3].lo ^ 0x80808080u, p);
            else
                transpose_bytes4(s[4 * q].hi ^ 0x80808080u, s[4 * q + 1].hi ^ 0x80808080u, s[4 * q + 2].hi ^ 0x80808080u,
                                 s[4 * q + 3].hi ^ 0x80808080u, p);
#pragma unroll
            for (int b = 0; b < 4; b++) pl[b][q] = p[b];
        }
        i32x16_t d[4];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            i32x4_t bf;
            bf.x = (int)pl[b][0]; bf.y = (int)pl[b][1]; bf.z = (int)pl[b][2]; bf.w = 0;
            d[b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bf, zero, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 12; r++) {
            int64_t t = (int64_t)d[0][r] * m0 + (int64_t)rcb[12 * half + r];
            t = (int64_t)d[1][r] * m8 + t;
            t = (int64_t)d[2][r] * m16 + t;
            t = (int64_t)d[3][r] * m24 + t;
            acc[half][r] = (uint64_t)t;
        }
        __builtin_amdgcn_sched_barrier(0);   // the low half's sixteen-register results are dead before the high half's exist
    }
#pragma unroll
    for (int r = 0; r < 12; r++) s[r] = fold_acc(acc[0][r], acc[1][r]);
}

}  // namespace gl32
