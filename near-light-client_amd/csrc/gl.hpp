// Goldilocks field F_p, p = 2^64 - 2^32 + 1, and its quadratic extension F_p[X]/(X^2 - 7),
// for gfx950 device code and for the C++ host orchestration (same source, both sides).
//
// Replaces plonky2_field::goldilocks_field::GoldilocksField and extension::quadratic
// (crate pinned at /root/reference/Cargo.lock:4912-4914; reached from every prover call
// behind nearx/src/test_utils.rs:62).  Written for CDNA4: there is no 64-bit integer MFMA, so
// a field multiply is four v_mad_u64_u32 plus a shift/add reduction that never divides.
//
// Representation: values in memory are canonical (< p).  Inside kernels a value may be any
// u64 congruent to the element ("loose"); functions say which they accept and return.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include "nlx_field.h"

#if defined(__HIP__)
#include <hip/hip_runtime.h>
#define GL_HD __host__ __device__ __forceinline__
#define GL_H __host__ inline   // host half of a function that has a hand-written __device__ overload
#else
#define GL_HD inline
#define GL_H inline
#endif

namespace gl {

constexpr uint64_t P = 0xFFFFFFFF00000001ULL;
constexpr uint64_t EPS = 0xFFFFFFFFULL;  // 2^64 mod p
// the generator pair comes from include/nlx_field.h (one definition for product, oracle and golden model)
constexpr uint64_t GEN = NLX_GL_MULTIPLICATIVE_GROUP_GENERATOR;  // multiplicative generator = coset shift
constexpr uint64_t POW2_GEN = NLX_GL_POWER_OF_TWO_GENERATOR;     // order 2^32
constexpr unsigned TWO_ADICITY = 32;
constexpr uint64_t W = 7;  // X^2 = W in the quadratic extension

GL_HD uint64_t mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

#if defined(__HIP__)
// Device forms with hand-placed carries (overloads of the host forms below: clang resolves __host__ / __device__ per side).  Every vector instruction of this code issues at the same ~2 ns whatever it is
// (gl32.hpp), so the forms below are chosen by COUNT: hipcc's u64 code for the three functions needs 4 / 6 / 6.
// x + EPS carries out of 2^64 exactly when x >= P = 2^64 - EPS, and the sum is then x - P: one multiply-add (-1 * 1 + x,
// carry into an SGPR pair) and two selects.
__device__ __forceinline__ uint64_t canon(uint64_t a) {
    uint64_t t, c;
    asm("v_mad_u64_u32 %[t], %[c], -1, 1, %[a]" : [t] "=&v"(t), [c] "=s"(c) : [a] "v"(a));
    uint32_t r0, r1;
    asm("v_cndmask_b32_e64 %[r0], %[a0], %[t0], %[c]\n\t"
        "v_cndmask_b32_e64 %[r1], %[a1], %[t1], %[c]"
        : [r0] "=&v"(r0), [r1] "=v"(r1)
        : [a0] "v"((uint32_t)a), [a1] "v"((uint32_t)(a >> 32)), [t0] "v"((uint32_t)t), [t1] "v"((uint32_t)(t >> 32)), [c] "s"(c));
    return ((uint64_t)r1 << 32) | r0;
}
// canonical + canonical -> canonical: s = a + b, reduced when the addition carried or s >= P (5 vector instructions)
__device__ __forceinline__ uint64_t add(uint64_t a, uint64_t b) {
    uint32_t s0, s1;
    uint64_t c1, c2, t;
    asm("v_add_co_u32 %[s0], vcc, %[a0], %[b0]\n\t"
        "v_addc_co_u32 %[s1], %[c1], %[a1], %[b1], vcc"
        : [s0] "=&v"(s0), [s1] "=&v"(s1), [c1] "=s"(c1)
        : [a0] "v"((uint32_t)a), [a1] "v"((uint32_t)(a >> 32)), [b0] "v"((uint32_t)b), [b1] "v"((uint32_t)(b >> 32))
        : "vcc");
    const uint64_t s = ((uint64_t)s1 << 32) | s0;
    asm("v_mad_u64_u32 %[t], %[c], -1, 1, %[s]" : [t] "=&v"(t), [c] "=s"(c2) : [s] "v"(s));
    const uint64_t c = c1 | c2;
    uint32_t r0, r1;
    asm("v_cndmask_b32_e64 %[r0], %[s0], %[t0], %[c]\n\t"
        "v_cndmask_b32_e64 %[r1], %[s1], %[t1], %[c]"
        : [r0] "=&v"(r0), [r1] "=v"(r1)
        : [s0] "v"(s0), [s1] "v"(s1), [t0] "v"((uint32_t)t), [t1] "v"((uint32_t)(t >> 32)), [c] "s"(c));
    return ((uint64_t)r1 << 32) | r0;
}
// (any u64) - canonical -> a value congruent to the difference, canonical when a is: a borrow is worth -EPS, and
// a - b + 2^64 >= 2^64 - (P - 1) > EPS, so the correction cannot borrow again (5 vector instructions)
__device__ __forceinline__ uint64_t sub(uint64_t a, uint64_t b) {
    uint32_t d0, d1, m;
    asm("v_sub_co_u32 %[d0], vcc, %[a0], %[b0]\n\t"
        "v_subb_co_u32 %[d1], vcc, %[a1], %[b1], vcc\n\t"
        "v_cndmask_b32_e64 %[m], 0, -1, vcc\n\t"
        "v_sub_co_u32 %[d0], vcc, %[d0], %[m]\n\t"
        "v_subbrev_co_u32 %[d1], vcc, 0, %[d1], vcc"
        : [d0] "=&v"(d0), [d1] "=&v"(d1), [m] "=&v"(m)
        : [a0] "v"((uint32_t)a), [a1] "v"((uint32_t)(a >> 32)), [b0] "v"((uint32_t)b), [b1] "v"((uint32_t)(b >> 32))
        : "vcc");
    return ((uint64_t)d1 << 32) | d0;
}
#endif
GL_H uint64_t canon(uint64_t a) { return a >= P ? a - P : a; }

// canonical + canonical -> canonical
GL_H uint64_t add(uint64_t a, uint64_t b) {
    uint64_t s = a + b;
    return (s < a || s >= P) ? s - P : s;
}
// canonical - canonical -> canonical
GL_H uint64_t sub(uint64_t a, uint64_t b) {
    uint64_t d = a - b;
    return a < b ? d + P : d;
}
GL_HD uint64_t neg(uint64_t a) { return a ? P - a : 0; }

// loose + canonical -> loose (single conditional fix-up; see plonky2 GoldilocksField::add)
#if defined(__HIP__)
// four vector instructions: the carry becomes a mask, and mask * 1 + s adds EPS where it is set
__device__ __forceinline__ uint64_t add_loose(uint64_t a, uint64_t c) {
    uint32_t s0, s1, m;
    asm("v_add_co_u32 %[s0], vcc, %[a0], %[c0]\n\t"
        "v_addc_co_u32 %[s1], vcc, %[a1], %[c1], vcc\n\t"
        "v_cndmask_b32_e64 %[m], 0, -1, vcc"
        : [s0] "=&v"(s0), [s1] "=&v"(s1), [m] "=v"(m)
        : [a0] "v"((uint32_t)a), [a1] "v"((uint32_t)(a >> 32)), [c0] "v"((uint32_t)c), [c1] "v"((uint32_t)(c >> 32))
        : "vcc");
    uint64_t r;
    asm("v_mad_u64_u32 %[r], vcc, %[m], 1, %[s]" : [r] "=&v"(r) : [m] "v"(m), [s] "v"(((uint64_t)s1 << 32) | s0) : "vcc");
    return r;
}
#endif
GL_H uint64_t add_loose(uint64_t a, uint64_t c) {
    uint64_t s = a + c;
    return s < a ? s + EPS : s;
}

// (hi:lo) mod p, any 128-bit input -> loose u64
GL_HD uint64_t reduce128_loose(uint64_t lo, uint64_t hi) {
    uint64_t hi_hi = hi >> 32, hi_lo = hi & EPS;
    uint64_t t0 = lo - hi_hi;
    if (lo < hi_hi) t0 -= EPS;
    uint64_t t1 = (hi_lo << 32) - hi_lo;  // hi_lo * (2^32 - 1)
    uint64_t r = t0 + t1;
    if (r < t1) r += EPS;
    return r;
}
GL_HD uint64_t reduce128(uint64_t lo, uint64_t hi) { return canon(reduce128_loose(lo, hi)); }

#if defined(__HIP_DEVICE_COMPILE__)
// 64x64 -> 128 product and Goldilocks reduction (16 instructions; hipcc's u64 code needs 29).  See gl32.hpp for the measurements
// behind this.  The product is a chain of five multiply-adds none of which can overflow
//   p0 = a0 b0;  p1 = a0 b1 + hi(p0);  p2 = a1 b0 + lo(p1);  p3 = a1 b1 + hi(p1) + hi(p2)   (w0 = lo(p0), w1 = lo(p2), (w3:w2) = p3)
// - the three zero-extended addends cost a v_mov each, which issues at half the price of the add-with-carry instructions the
// four-product form needed (round 4: the Poseidon micro-benchmark 2.56 -> 2.76 G/s on this change alone).
__device__ __forceinline__ uint64_t mul_loose_asm(uint64_t a, uint64_t b) {
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    // plain C++ for the four products: hipcc selects exactly v_mad_u64_u32 with a (value : 0) register pair for each addend and
    // spaces nothing (between asm statements it puts an s_nop); the last addition as x * 1 + c saves the pair
    const uint64_t p0 = (uint64_t)a0 * b0;
    const uint64_t p1 = (uint64_t)a0 * b1 + (p0 >> 32);
    const uint64_t p2 = (uint64_t)a1 * b0 + (uint32_t)p1;
    uint64_t p3 = (uint64_t)a1 * b1 + (p1 >> 32), sink;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %3" : "=v"(p3), "=s"(sink) : "v"((uint32_t)(p2 >> 32)), "v"(p3));
    // r = (w1:w0) - w3 [borrow -> -EPS]
    uint32_t r0, r1, t0;
    const uint32_t w2 = (uint32_t)p3;
    asm("v_sub_co_u32 %[r0], vcc, %[w0], %[w3]\n\t"
        "v_subbrev_co_u32 %[r1], vcc, 0, %[w1], vcc\n\t"
        "v_cndmask_b32_e64 %[t0], 0, -1, vcc\n\t"
        "v_sub_co_u32 %[r0], vcc, %[r0], %[t0]\n\t"
        "v_subbrev_co_u32 %[r1], vcc, 0, %[r1], vcc"
        : [r0] "=&v"(r0), [r1] "=&v"(r1), [t0] "=&v"(t0)
        : [w0] "v"((uint32_t)p0), [w1] "v"((uint32_t)p2), [w3] "v"((uint32_t)(p3 >> 32))
        : "vcc");
    // + w2 * EPS [carry -> +EPS]: ONE multiply-add (carry-out in VCC), carry -> mask, mask * 1 + r.  On gfx950 every
    // VCC-chained add issues as slowly as a multiply (profiles/r02_valu_ubench.txt), so these three replace seven.
    const uint64_t base = ((uint64_t)r1 << 32) | r0;
    uint64_t r;
    uint32_t tm;
    asm("v_mad_u64_u32 %[r], vcc, %[w2], -1, %[base]\n\t"
        "v_cndmask_b32_e64 %[t], 0, -1, vcc\n\t"
        "v_mad_u64_u32 %[r], vcc, %[t], 1, %[r]"
        : [r] "=&v"(r), [t] "=&v"(tm)
        : [w2] "v"(w2), [base] "v"(base)
        : "vcc");
    return r;
}
#endif

// loose * loose -> loose
GL_HD uint64_t mul_loose(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return mul_loose_asm(a, b);
#else
    return reduce128_loose(a * b, mulhi64(a, b));
#endif
}
// any * any -> canonical
GL_HD uint64_t mul(uint64_t a, uint64_t b) { return canon(mul_loose(a, b)); }
GL_HD uint64_t sqr(uint64_t a) { return mul(a, a); }

GL_HD uint64_t pow(uint64_t b, uint64_t e) {
    uint64_t r = 1;
    while (e) {
        if (e & 1) r = mul(r, b);
        b = sqr(b);
        e >>= 1;
    }
    return r;
}
GL_HD uint64_t exp_pow2(uint64_t a, unsigned k) {
    while (k--) a = sqr(a);
    return a;
}
// a^(p-2); addition chain: p-2 = 2^64 - 2^32 - 1  (72 multiplications)
GL_HD uint64_t inv(uint64_t a) {
    // t_k = a^(2^k - 1)
    uint64_t t2 = mul(sqr(a), a);
    uint64_t t3 = mul(sqr(t2), a);
    uint64_t t6 = mul(exp_pow2(t3, 3), t3);
    uint64_t t12 = mul(exp_pow2(t6, 6), t6);
    uint64_t t24 = mul(exp_pow2(t12, 12), t12);
    uint64_t t30 = mul(exp_pow2(t24, 6), t6);
    uint64_t t31 = mul(sqr(t30), a);
    // p - 2 = (2^31 - 1) * 2^33 + (2^32 - 1)   [= 2^64 - 2^33 + 2^32 - 1]
    uint64_t t32 = mul(sqr(t31), a);
    return mul(exp_pow2(t31, 33), t32);
}
GL_HD uint64_t root_of_unity(unsigned n_log) { return exp_pow2(POW2_GEN, TWO_ADICITY - n_log); }

// ---- quadratic extension ----
struct Ext {
    uint64_t a, b;  // a + b X, canonical
};
GL_HD Ext ext(uint64_t a, uint64_t b = 0) { return Ext{a, b}; }
GL_HD Ext add(Ext x, Ext y) { return Ext{add(x.a, y.a), add(x.b, y.b)}; }
GL_HD Ext sub(Ext x, Ext y) { return Ext{sub(x.a, y.a), sub(x.b, y.b)}; }
GL_HD Ext mul(Ext x, Ext y) {
    uint64_t bb = mul(x.b, y.b);
    // 7*bb without a full multiply: 8*bb - bb, done in 128 bits
    uint64_t c0 = add(mul(x.a, y.a), reduce128(bb * W, mulhi64(bb, W)));
    uint64_t c1 = add(mul(x.a, y.b), mul(x.b, y.a));
    return Ext{c0, c1};
}
GL_HD Ext mul(Ext x, uint64_t s) { return Ext{mul(x.a, s), mul(x.b, s)}; }
GL_HD Ext inv(Ext x) {
    uint64_t n = sub(sqr(x.a), mul(W, sqr(x.b)));
    uint64_t ni = inv(n);
    return Ext{mul(x.a, ni), mul(neg(x.b), ni)};
}
GL_HD Ext pow(Ext b, uint64_t e) {
    Ext r{1, 0};
    while (e) {
        if (e & 1) r = mul(r, b);
        b = mul(b, b);
        e >>= 1;
    }
    return r;
}
GL_HD bool eq(Ext x, Ext y) { return x.a == y.a && x.b == y.b; }

GL_HD uint32_t bitrev32(uint32_t x, unsigned bits) {
#if defined(__HIP_DEVICE_COMPILE__)
    return bits ? (__brev(x) >> (32 - bits)) : 0;
#else
    uint32_t r = 0;
    for (unsigned i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
#endif
}

}  // namespace gl
