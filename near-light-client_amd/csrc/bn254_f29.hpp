// BN254 base field on nine 29-bit limbs - the arithmetic of the MSM's device kernels (bn254_msm.hip).
//
// Why not eight 32-bit limbs (bn254_fp.hpp): gfx950 wants the 64-bit operands of v_mad_u64_u32 in even-aligned register
// pairs, and a CIOS Montgomery product on saturated limbs shifts its accumulator by one 32-bit limb per row - the compiler
// pays for that with register moves (7 294 v_mov_b32 against 2 305 multiply-adds in the first bucket kernel, ~880
// instructions per field product).  With 29-bit limbs a product is < 2^58, eighteen of them fit a 64-bit column, so the
// schoolbook product and the Montgomery reduction are plain `col[i + j] += a[i] * b[j]` into FIXED columns: one multiply-add
// per partial product, no carry handling inside the loops, no shifting (~250 instructions per product).
//
// Representation: value = sum v[i] 2^(29 i), limbs 0..7 < 2^29, limb 8 takes what is left; Montgomery form with R' = 2^261.
// Values are kept loose (no canonical reduction on the fast path).  Bounds, checked by the host test
// (tests/native/bn254_f29_check.cpp) on random and extreme inputs:
//   mul / sqr      inputs < 2^257.5 each (product < 2^515 = 2^254 R')      -> result < 2^254 + q < 2^255   ("tight")
//   add            any two values whose sum is < 2^258                      -> the plain sum, limbs normalised
//   sub<K>(a, b)   a + K q - b for K q >= the bound of b (K = 4: b < 2^255, K = 8: b < 3 * 2^255)
//   tighten        any value < 2^258                                        -> the same residue, < 1.1 q
// The point formulas in bn254_msm.hip state the bound of every intermediate.
#pragma once
#include <cstdint>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define F29_HD __host__ __device__ __forceinline__
#else
#define F29_HD inline   // plain g++ build of the host test
#endif

namespace nlx {
namespace f29 {

constexpr int NL = 9, LB = 29;
constexpr uint32_t MASK = (1u << LB) - 1;

struct Fe {
    uint32_t v[NL];
};

// The two moduli of BN254 on 29-bit limbs: the base field q (curve coordinates, the MSM) and the scalar field r (the NTT).
// p(i): limbs of the modulus; kp<K>(i): limbs of K p for K = 4, 8; one(i): R' mod p; c266(i): 2^266 mod p; NINV = -p^-1 mod
// 2^29; TIGHT = floor(2^268 / p) (both moduli are close to 2^253.6, and the same bounds hold for both).
struct QMod {
    static constexpr uint32_t NINV = 0x4866389u, TIGHT = 21668u;
    F29_HD static uint32_t p(int i) {
        constexpr uint32_t Q[NL] = {0x187cfd47u, 0x010460b6u, 0x1c72a34fu, 0x02d522d0u, 0x1585d978u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
        return Q[i];
    }
    template <int K>
    F29_HD static uint32_t kp(int i) {
        static_assert(K == 4 || K == 8, "K p is tabulated for K = 4 and 8");
        constexpr uint32_t Q4[NL] = {0x01f3f51cu, 0x041182dbu, 0x11ca8d3cu, 0x0b548b43u, 0x161765e0u, 0x0b6d0302u, 0x029b8504u, 0x197098d0u, 0x00c19139u};
        constexpr uint32_t Q8[NL] = {0x03e7ea38u, 0x082305b6u, 0x03951a78u, 0x16a91687u, 0x0c2ecbc0u, 0x16da0605u, 0x05370a08u, 0x12e131a0u, 0x01832273u};
        return K == 4 ? Q4[i] : Q8[i];
    }
    F29_HD static uint32_t one(int i) {
        constexpr uint32_t O[NL] = {0x157ccc21u, 0x141c2758u, 0x185230d3u, 0x014c0419u, 0x0aa36fb9u, 0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};
        return O[i];
    }
    F29_HD static uint32_t c266(int i) {
        constexpr uint32_t C[NL] = {0x13349ca1u, 0x1a5d84a8u, 0x0a3e5cacu, 0x100249e0u, 0x12b951e8u, 0x0e92d304u, 0x14cb95b3u, 0x041b9d3du, 0x00058003u};
        return C[i];
    }
};
struct RMod {
    static constexpr uint32_t NINV = 0xfffffffu, TIGHT = 21668u;
    F29_HD static uint32_t p(int i) {
        constexpr uint32_t R[NL] = {0x10000001u, 0x1f0fac9fu, 0x0e5c2450u, 0x07d090f3u, 0x1585d283u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
        return R[i];
    }
    template <int K>
    F29_HD static uint32_t kp(int i) {
        static_assert(K == 4 || K == 8, "K p is tabulated for K = 4 and 8");
        constexpr uint32_t R4[NL] = {0x00000004u, 0x1c3eb27eu, 0x19709143u, 0x1f4243cdu, 0x16174a0cu, 0x0b6d0302u, 0x029b8504u, 0x197098d0u, 0x00c19139u};
        constexpr uint32_t R8[NL] = {0x00000008u, 0x187d64fcu, 0x12e12287u, 0x1e84879bu, 0x0c2e9419u, 0x16da0605u, 0x05370a08u, 0x12e131a0u, 0x01832273u};
        return K == 4 ? R4[i] : R8[i];
    }
    F29_HD static uint32_t one(int i) {
        constexpr uint32_t O[NL] = {0x0fffff57u, 0x1ea70ab4u, 0x052c068bu, 0x17504f49u, 0x0aa8075bu, 0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};
        return O[i];
    }
    F29_HD static uint32_t c266(int i) {
        constexpr uint32_t C[NL] = {0x0fffead7u, 0x1d5444f4u, 0x04438aa5u, 0x03b4d096u, 0x134c84dau, 0x0e92d304u, 0x14cb95b3u, 0x041b9d3du, 0x00058003u};
        return C[i];
    }
};

F29_HD Fe zero() {
    Fe r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.v[i] = 0;
    return r;
}
template <class M = QMod>
F29_HD Fe one() {   // R' mod p: the Montgomery form of 1
    Fe r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.v[i] = M::one(i);
    return r;
}
F29_HD bool is_zero_exact(const Fe& a) {   // the integer 0 (how the point at infinity is marked), not "0 mod q"
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) o |= a.v[i];
    return o == 0;
}

// Montgomery product a b / 2^261 mod p, operand bounds in the header comment
template <class M = QMod>
F29_HD Fe mul(const Fe& a, const Fe& b) {
    uint64_t col[2 * NL];
#pragma unroll
    for (int k = 0; k < 2 * NL; k++) col[k] = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
#pragma unroll
        for (int j = 0; j < NL; j++) col[i + j] += (uint64_t)a.v[i] * b.v[j];
    }
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const uint32_t m = ((uint32_t)col[i] * M::NINV) & MASK;
#pragma unroll
        for (int j = 0; j < NL; j++) col[i + j] += (uint64_t)m * M::p(j);
        col[i + 1] += col[i] >> LB;   // the low 29 bits of col[i] are now zero
    }
    Fe r;
#pragma unroll
    for (int i = NL; i < 2 * NL - 1; i++) {
        r.v[i - NL] = (uint32_t)col[i] & MASK;
        col[i + 1] += col[i] >> LB;
    }
    r.v[NL - 1] = (uint32_t)col[2 * NL - 1];
    return r;
}
template <class M = QMod>
F29_HD Fe sqr(const Fe& a) { return mul<M>(a, a); }

F29_HD Fe add(const Fe& a, const Fe& b) {
    Fe r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < NL - 1; i++) {
        const uint32_t s = a.v[i] + b.v[i] + c;
        r.v[i] = s & MASK;
        c = s >> LB;
    }
    r.v[NL - 1] = a.v[NL - 1] + b.v[NL - 1] + c;
    return r;
}
F29_HD Fe dbl(const Fe& a) { return add(a, a); }

// a + K p - b, non-negative as long as b <= K p
template <int K, class M = QMod>
F29_HD Fe sub(const Fe& a, const Fe& b) {
    Fe r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < NL - 1; i++) {
        const int32_t s = (int32_t)a.v[i] + (int32_t)M::template kp<K>(i) - (int32_t)b.v[i] + c;
        r.v[i] = (uint32_t)s & MASK;
        c = s >> LB;   // arithmetic shift: floor
    }
    r.v[NL - 1] = (uint32_t)((int32_t)a.v[NL - 1] + (int32_t)M::template kp<K>(NL - 1) - (int32_t)b.v[NL - 1] + c);
    return r;
}

// the same residue below 1.1 q, for any value < 2^258: subtract floor(V / 2^248 * (2^248 / q)) q, the factor rounded down
template <class M = QMod>
F29_HD Fe tighten(const Fe& a) {
    const uint32_t t = ((a.v[NL - 1] >> 16) * M::TIGHT) >> 20;   // <= V / p, short of it by less than 0.03
    Fe r;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < NL - 1; i++) {
        const int64_t s = (int64_t)a.v[i] - (int64_t)((uint64_t)t * M::p(i)) + c;
        r.v[i] = (uint32_t)s & MASK;
        c = s >> LB;
    }
    r.v[NL - 1] = (uint32_t)((int64_t)a.v[NL - 1] - (int64_t)((uint64_t)t * M::p(NL - 1)) + c);
    return r;
}

// a = 0 mod p, for any value < 2^258
template <class M = QMod>
F29_HD bool is_zero_mod(const Fe& a) {
    const Fe t = tighten<M>(a);   // in [0, 1.1 p): zero mod p means 0 or p
    uint32_t z = 0, e = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        z |= t.v[i];
        e |= t.v[i] ^ M::p(i);
    }
    return z == 0 || e == 0;
}

// ---- in and out of the form: 2^256-Montgomery words (gnark-crypto's fp.Element) <-> this ----
F29_HD Fe from_words256(const uint32_t* w /* eight 32-bit limbs of an integer */) {   // plain re-slicing, no arithmetic
    Fe r;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int bit = LB * i, k = bit >> 5, o = bit & 31;
        uint64_t x = (uint64_t)w[k] >> o;
        if (k + 1 < 8) x |= (uint64_t)w[k + 1] << (32 - o);
        r.v[i] = (uint32_t)x & MASK;
    }
    return r;
}
// x 2^256 mod p (an fp.Element / fr.Element read as an integer) -> x 2^261 mod p: one product with 2^266 mod p
template <class M = QMod>
F29_HD Fe from_mont256(const uint32_t* w) {
    Fe c;
#pragma unroll
    for (int i = 0; i < NL; i++) c.v[i] = M::c266(i);
    return mul<M>(from_words256(w), c);
}
// the limbs of a value below 2^256 as eight 32-bit words (plain re-slicing: how values rest in memory between NTT passes)
F29_HD void to_words256(const Fe& a, uint32_t* w) {
#pragma unroll
    for (int k = 0; k < 8; k++) w[k] = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int bit = LB * i, k = bit >> 5, o2 = bit & 31;
        const uint64_t x = (uint64_t)a.v[i] << o2;
        w[k] |= (uint32_t)x;
        if (k + 1 < 8) w[k + 1] |= (uint32_t)(x >> 32);
    }
}
// the canonical representative (< p) of a value below 2^258
template <class M = QMod>
F29_HD Fe canonical(const Fe& a) {
    const Fe t = tighten<M>(a);   // < 1.1 p: at most one p too many
    Fe u;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < NL - 1; i++) {
        const int32_t s = (int32_t)t.v[i] - (int32_t)M::p(i) + c;
        u.v[i] = (uint32_t)s & MASK;
        c = s >> LB;
    }
    const int32_t top = (int32_t)t.v[NL - 1] - (int32_t)M::p(NL - 1) + c;
    u.v[NL - 1] = (uint32_t)top;
    return top < 0 ? t : u;
}
// the canonical integer x < p of a value in this form, as eight 32-bit limbs
template <class M = QMod>
F29_HD void to_canonical256(const Fe& a, uint32_t* w) {
    Fe o = zero();
    o.v[0] = 1;
    Fe t = mul<M>(a, o);   // (A + m p) / R' with m < R': at most p, and p only for A = 0 mod p
    uint32_t e = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) e |= t.v[i] ^ M::p(i);
    if (e == 0) t = zero();
#pragma unroll
    for (int k = 0; k < 8; k++) w[k] = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int bit = LB * i, k = bit >> 5, o2 = bit & 31;
        const uint64_t x = (uint64_t)t.v[i] << o2;
        w[k] |= (uint32_t)x;
        if (k + 1 < 8) w[k + 1] |= (uint32_t)(x >> 32);
    }
}

// ---- field policies for the curve code (bn254_msm.hip): the group law is written once over `F::T` ----
// F1: the base field itself, loose values with the bounds stated at each formula.
struct F1 {
    typedef Fe T;
    static constexpr int WORDS = 8;   // 32-bit words of an element in memory
    F29_HD static T zero() { return f29::zero(); }
    F29_HD static T one() { return f29::one<QMod>(); }
    F29_HD static T mul(const T& a, const T& b) { return f29::mul<QMod>(a, b); }
    F29_HD static T sqr(const T& a) { return f29::mul<QMod>(a, a); }
    F29_HD static T add(const T& a, const T& b) { return f29::add(a, b); }
    F29_HD static T dbl(const T& a) { return f29::add(a, a); }
    template <int K>
    F29_HD static T sub(const T& a, const T& b) { return f29::sub<K, QMod>(a, b); }
    F29_HD static T tighten(const T& a) { return f29::tighten<QMod>(a); }
    F29_HD static bool is_zero_mod(const T& a) { return f29::is_zero_mod<QMod>(a); }
    F29_HD static bool is_zero_exact(const T& a) { return f29::is_zero_exact(a); }
    F29_HD static T from_mont256(const uint32_t* w) { return f29::from_mont256<QMod>(w); }
    F29_HD static void to_words(const T& a, uint32_t* w) { f29::to_words256(a, w); }            // a < 2^256
    F29_HD static T from_words(const uint32_t* w) { return f29::from_words256(w); }
    F29_HD static void to_canonical(const T& a, uint32_t* w) { f29::to_canonical256<QMod>(a, w); }
};
// F2: the quadratic extension Fq[u] / (u^2 + 1) of G2's coordinates.  Every result is tightened (both components below
// 1.1 q), so no bound has to be tracked through the formulas: sums of two components stay below 2.2 q, products of such sums
// far below 2^515, and a + 4 q - b is non-negative for any subtrahend.  Three base-field products per product (Karatsuba).
struct Fe2 {
    Fe c0, c1;
};
struct F2 {
    typedef Fe2 T;
    static constexpr int WORDS = 16;
    F29_HD static Fe tt(const Fe& a) { return f29::tighten<QMod>(a); }
    F29_HD static T zero() { return T{f29::zero(), f29::zero()}; }
    F29_HD static T one() { return T{f29::one<QMod>(), f29::zero()}; }
    F29_HD static T mul(const T& a, const T& b) {
        const Fe t0 = f29::mul<QMod>(a.c0, b.c0), t1 = f29::mul<QMod>(a.c1, b.c1);
        const Fe s = f29::mul<QMod>(f29::add(a.c0, a.c1), f29::add(b.c0, b.c1));
        return T{tt(f29::sub<4, QMod>(t0, t1)), tt(f29::sub<8, QMod>(s, f29::add(t0, t1)))};
    }
    F29_HD static T sqr(const T& a) {   // (a0 + a1)(a0 - a1) + 2 a0 a1 u
        const Fe p = f29::mul<QMod>(f29::add(a.c0, a.c1), f29::sub<4, QMod>(a.c0, a.c1));
        const Fe m = f29::mul<QMod>(a.c0, a.c1);
        return T{tt(p), tt(f29::add(m, m))};
    }
    F29_HD static T add(const T& a, const T& b) { return T{tt(f29::add(a.c0, b.c0)), tt(f29::add(a.c1, b.c1))}; }
    F29_HD static T dbl(const T& a) { return add(a, a); }
    template <int K>
    F29_HD static T sub(const T& a, const T& b) { return T{tt(f29::sub<4, QMod>(a.c0, b.c0)), tt(f29::sub<4, QMod>(a.c1, b.c1))}; }
    F29_HD static T tighten(const T& a) { return T{tt(a.c0), tt(a.c1)}; }
    F29_HD static bool is_zero_mod(const T& a) { return f29::is_zero_mod<QMod>(a.c0) && f29::is_zero_mod<QMod>(a.c1); }
    F29_HD static bool is_zero_exact(const T& a) { return f29::is_zero_exact(a.c0) && f29::is_zero_exact(a.c1); }
    F29_HD static T from_mont256(const uint32_t* w) { return T{tt(f29::from_mont256<QMod>(w)), tt(f29::from_mont256<QMod>(w + 8))}; }
    F29_HD static void to_words(const T& a, uint32_t* w) { f29::to_words256(a.c0, w); f29::to_words256(a.c1, w + 8); }
    F29_HD static T from_words(const uint32_t* w) { return T{f29::from_words256(w), f29::from_words256(w + 8)}; }
    F29_HD static void to_canonical(const T& a, uint32_t* w) { f29::to_canonical256<QMod>(a.c0, w); f29::to_canonical256<QMod>(a.c1, w + 8); }
};

}  // namespace f29
}  // namespace nlx
