/* ORACLE - TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
 *
 * CPU restatement (plain C) of the Goldilocks field and its quadratic extension as used by
 * plonky2_field (crate plonky2_field 0.1.1 @ mir-protocol/plonky2 rev d2598bd, pinned at
 * /root/reference/Cargo.lock:4912-4914; source NOT vendored in /root/reference, so this
 * follows the published algorithm: goldilocks_field.rs / extension/quadratic.rs).
 *
 *   p  = 2^64 - 2^32 + 1,  ext2 = F_p[X]/(X^2 - 7).  MULTIPLICATIVE_GROUP_GENERATOR (= coset shift) and
 *   POWER_OF_TWO_GENERATOR (order 2^32) are parameters: include/nlx_field.h holds the one definition
 *   (default 7 / 1753635133440165772; the evidence for each candidate pair is written there).
 *
 * Parity status: "parity unpinned" against Rust-produced proof bytes (none exist in the
 * reference); the Poseidon permutation is pinned by upstream's known-answer
 * vectors (tests/golden/poseidon_kat.json), the field by algebraic identities.
 *
 * All values crossing function boundaries are canonical (< p).
 */
#ifndef NLX_ORACLE_GL_H
#define NLX_ORACLE_GL_H
#include <stdint.h>
#include <stddef.h>
#include "../include/nlx_field.h"

#define GL_P 0xFFFFFFFF00000001ULL
#define GL_EPS 0xFFFFFFFFULL
#define GL_GEN NLX_GL_MULTIPLICATIVE_GROUP_GENERATOR
#define GL_POW2_GEN NLX_GL_POWER_OF_TWO_GENERATOR
#define GL_TWO_ADICITY 32
#define GL_W 7ULL /* ext2 non-residue */

typedef unsigned __int128 u128;

static inline uint64_t gl_canon(uint64_t a) { return a >= GL_P ? a - GL_P : a; }

static inline uint64_t gl_add(uint64_t a, uint64_t b) {
    u128 s = (u128)a + b;
    if (s >= GL_P) s -= GL_P;
    return (uint64_t)s;
}
static inline uint64_t gl_sub(uint64_t a, uint64_t b) { return a >= b ? a - b : a + (GL_P - b); }
static inline uint64_t gl_neg(uint64_t a) { return a ? GL_P - a : 0; }
/* 2^64 = 2^32 - 1 and 2^96 = -1 (mod p): fold hi_hi and hi_lo into lo (plonky2 reduce128) */
static inline uint64_t gl_reduce128(u128 x) {
    uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
    uint64_t hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
    uint64_t t0 = lo - hi_hi;
    if (lo < hi_hi) t0 -= GL_EPS;
    uint64_t t1 = hi_lo * GL_EPS;
    uint64_t r = t0 + t1;
    if (r < t1) r += GL_EPS;
    return r >= GL_P ? r - GL_P : r;
}
static inline uint64_t gl_mul(uint64_t a, uint64_t b) { return gl_reduce128((u128)a * b); }
static inline uint64_t gl_sqr(uint64_t a) { return gl_mul(a, a); }

static inline uint64_t gl_pow(uint64_t b, uint64_t e) {
    uint64_t r = 1;
    while (e) {
        if (e & 1) r = gl_mul(r, b);
        b = gl_sqr(b);
        e >>= 1;
    }
    return r;
}
static inline uint64_t gl_inv(uint64_t a) { return gl_pow(a, GL_P - 2); }
static inline uint64_t gl_exp_pow2(uint64_t a, unsigned k) {
    while (k--) a = gl_sqr(a);
    return a;
}
/* plonky2_field::types::Field::primitive_root_of_unity(n_log) */
static inline uint64_t gl_root_of_unity(unsigned n_log) {
    return gl_exp_pow2(GL_POW2_GEN, GL_TWO_ADICITY - n_log);
}

/* ---- quadratic extension, element = (a0, a1) = a0 + a1*X, X^2 = 7 ---- */
typedef struct { uint64_t a, b; } gl2;

static inline gl2 gl2_make(uint64_t a, uint64_t b) { gl2 r = {a, b}; return r; }
static inline gl2 gl2_from(uint64_t a) { gl2 r = {a, 0}; return r; }
static inline gl2 gl2_add(gl2 x, gl2 y) { return gl2_make(gl_add(x.a, y.a), gl_add(x.b, y.b)); }
static inline gl2 gl2_sub(gl2 x, gl2 y) { return gl2_make(gl_sub(x.a, y.a), gl_sub(x.b, y.b)); }
static inline gl2 gl2_mul(gl2 x, gl2 y) {
    uint64_t c0 = gl_add(gl_mul(x.a, y.a), gl_mul(GL_W, gl_mul(x.b, y.b)));
    uint64_t c1 = gl_add(gl_mul(x.a, y.b), gl_mul(x.b, y.a));
    return gl2_make(c0, c1);
}
static inline gl2 gl2_scale(gl2 x, uint64_t s) { return gl2_make(gl_mul(x.a, s), gl_mul(x.b, s)); }
static inline gl2 gl2_inv(gl2 x) {
    /* 1/(a+bX) = (a-bX)/(a^2 - 7 b^2) */
    uint64_t n = gl_sub(gl_sqr(x.a), gl_mul(GL_W, gl_sqr(x.b)));
    uint64_t ni = gl_inv(n);
    return gl2_make(gl_mul(x.a, ni), gl_mul(gl_neg(x.b), ni));
}
static inline gl2 gl2_pow(gl2 b, uint64_t e) {
    gl2 r = gl2_from(1);
    while (e) {
        if (e & 1) r = gl2_mul(r, b);
        b = gl2_mul(b, b);
        e >>= 1;
    }
    return r;
}
static inline int gl2_eq(gl2 x, gl2 y) { return x.a == y.a && x.b == y.b; }

static inline unsigned gl_log2_strict(size_t n) {
    unsigned l = 0;
    while (((size_t)1 << l) < n) l++;
    return l;
}
static inline size_t gl_bitrev(size_t x, unsigned bits) {
    size_t r = 0;
    for (unsigned i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}
#endif
