// Goldilocks arithmetic on 32-bit limb pairs with hand-placed carry chains (gfx950 inline asm).
//
// Measured on MI355X (tools/ubench): v_mad_u64_u32 issues at the same rate as every other
// integer multiply (~4.7 cycles per wave-instruction) and plain 32-bit adds at ~2.9, so a field
// multiply is bounded by its four 32x32->64 products plus the carry/borrow glue around them.
// hipcc's u64 code spends ~25 glue instructions per multiply (zero-extending moves, 64-bit
// compares, cndmask pairs); keeping elements as {lo, hi} u32 pairs and chaining carries through
// VCC brings the whole multiply + reduction to 20 instructions.
#pragma once
#include "gl.hpp"

namespace gl32 {

struct F {
    uint32_t lo, hi;  // value = lo + hi * 2^32, any u64 ("loose")
};

__device__ __forceinline__ F from_u64(uint64_t x) { return F{(uint32_t)x, (uint32_t)(x >> 32)}; }
__device__ __forceinline__ uint64_t to_u64(F x) { return ((uint64_t)x.hi << 32) | x.lo; }

// (w3:w2:w1:w0) mod p -> loose
//   r = (w1:w0) - w3 [borrow -> -EPS] + (w2 * EPS) [carry -> +EPS]
__device__ __forceinline__ F reduce128(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
    F r;
    uint32_t t0, t1;
    asm("v_sub_co_u32 %[r0], vcc, %[w0], %[w3]\n\t"
        "v_subbrev_co_u32 %[r1], vcc, 0, %[w1], vcc\n\t"
        "v_cndmask_b32_e64 %[t0], 0, -1, vcc\n\t"
        "v_sub_co_u32 %[r0], vcc, %[r0], %[t0]\n\t"
        "v_subbrev_co_u32 %[r1], vcc, 0, %[r1], vcc\n\t"
        "v_sub_co_u32 %[t0], vcc, 0, %[w2]\n\t"
        "v_subbrev_co_u32 %[t1], vcc, 0, %[w2], vcc\n\t"
        "v_add_co_u32 %[r0], vcc, %[r0], %[t0]\n\t"
        "v_addc_co_u32 %[r1], vcc, %[r1], %[t1], vcc\n\t"
        "v_cndmask_b32_e64 %[t0], 0, -1, vcc\n\t"
        "v_add_co_u32 %[r0], vcc, %[r0], %[t0]\n\t"
        "v_addc_co_u32 %[r1], vcc, 0, %[r1], vcc"
        : [r0] "=&v"(r.lo), [r1] "=&v"(r.hi), [t0] "=&v"(t0), [t1] "=&v"(t1)
        : [w0] "v"(w0), [w1] "v"(w1), [w2] "v"(w2), [w3] "v"(w3)
        : "vcc");
    return r;
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
    // limbs w1..w3 and the reduction r = (w1:w0) - w3 [borrow -> -EPS] + w2*EPS [carry -> +EPS] in ONE
    // block (the compiler pads every asm boundary with an s_nop)
    F r;
    uint32_t w1, w2, w3, t0, t1;
    asm("v_add_co_u32 %[w1], vcc, %[th], %[m0]\n\t"
        "v_addc_co_u32 %[w2], vcc, %[x0], %[m1], vcc\n\t"
        "v_addc_co_u32 %[w3], vcc, 0, %[x1], vcc\n\t"
        "v_sub_co_u32 %[r0], vcc, %[w0], %[w3]\n\t"
        "v_subbrev_co_u32 %[r1], vcc, 0, %[w1], vcc\n\t"
        "v_cndmask_b32_e64 %[t0], 0, -1, vcc\n\t"
        "v_sub_co_u32 %[r0], vcc, %[r0], %[t0]\n\t"
        "v_subbrev_co_u32 %[r1], vcc, 0, %[r1], vcc\n\t"
        "v_sub_co_u32 %[t0], vcc, 0, %[w2]\n\t"
        "v_subbrev_co_u32 %[t1], vcc, 0, %[w2], vcc\n\t"
        "v_add_co_u32 %[r0], vcc, %[r0], %[t0]\n\t"
        "v_addc_co_u32 %[r1], vcc, %[r1], %[t1], vcc\n\t"
        "v_cndmask_b32_e64 %[t0], 0, -1, vcc\n\t"
        "v_add_co_u32 %[r0], vcc, %[r0], %[t0]\n\t"
        "v_addc_co_u32 %[r1], vcc, 0, %[r1], vcc"
        : [r0] "=&v"(r.lo), [r1] "=&v"(r.hi), [w1] "=&v"(w1), [w2] "=&v"(w2), [w3] "=&v"(w3), [t0] "=&v"(t0), [t1] "=&v"(t1)
        : [w0] "v"((uint32_t)t), [th] "v"((uint32_t)(t >> 32)), [m0] "v"((uint32_t)mid), [x0] "v"((uint32_t)x),
          [m1] "v"((uint32_t)(mid >> 32)), [x1] "v"(x1)
        : "vcc");
    return r;
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

// value = al + ah * 2^32 (al, ah u64 accumulators, ah < 2^42) -> loose
__device__ __forceinline__ F fold_acc(uint64_t al, uint64_t ah) {
    // limbs: w0 = al0, w1 = al1 + ah0, w2 = ah1 + carry  (w2 < 2^11), w3 = 0
    uint32_t w1, w2;
    asm("v_add_co_u32 %[w1], vcc, %[al1], %[ah0]\n\t"
        "v_addc_co_u32 %[w2], vcc, 0, %[ah1], vcc"
        : [w1] "=&v"(w1), [w2] "=v"(w2)
        : [al1] "v"((uint32_t)(al >> 32)), [ah0] "v"((uint32_t)ah), [ah1] "v"((uint32_t)(ah >> 32))
        : "vcc");
    // r = (w1:w0) + w2 * EPS, carry -> + EPS
    F r;
    uint32_t t0, t1;
    asm("v_sub_co_u32 %[t0], vcc, 0, %[w2]\n\t"
        "v_subbrev_co_u32 %[t1], vcc, 0, %[w2], vcc\n\t"
        "v_add_co_u32 %[r0], vcc, %[w0], %[t0]\n\t"
        "v_addc_co_u32 %[r1], vcc, %[w1], %[t1], vcc\n\t"
        "v_cndmask_b32_e64 %[t0], 0, -1, vcc\n\t"
        "v_add_co_u32 %[r0], vcc, %[r0], %[t0]\n\t"
        "v_addc_co_u32 %[r1], vcc, 0, %[r1], vcc"
        : [r0] "=&v"(r.lo), [r1] "=&v"(r.hi), [t0] "=&v"(t0), [t1] "=&v"(t1)
        : [w0] "v"((uint32_t)al), [w1] "v"(w1), [w2] "v"(w2)
        : "vcc");
    return r;
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
