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

}  // namespace gl32
