// 256-bit Montgomery arithmetic (R = 2^256, eight 32-bit limbs, CIOS) over a modulus picked by a parameter struct - BN254's
// base field Fq (curve coordinates) and scalar field Fr - shared by host and device code of the MSM (bn254_msm.hip;
// SURVEY.md §8 row f.4).  The element layout is gnark-crypto's fp.Element / fr.Element: four little-endian 64-bit words in
// Montgomery form, so a Go caller's []G1Affine and []fr.Element can be handed over as they lie in memory.  The NTT
// (bn254.hip) keeps its own copy of the Fr code with its hand-tuned butterflies.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace nlx {
namespace bnf {

#define BNF_HD __host__ __device__ __forceinline__

template <class P>
struct Fp {
    uint32_t v[8];
};

// q = 21888242871839275222246405745257275088696311157297823662689037894645226208583 (BN254 / alt_bn128 base field)
struct QP {
    static constexpr uint32_t N0INV = 0xe4866389u;  // -q^-1 mod 2^32
    BNF_HD static uint32_t mod(int i) {
        constexpr uint32_t M[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        return M[i];
    }
    BNF_HD static uint32_t one(int i) {  // R mod q
        constexpr uint32_t M[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u, 0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
        return M[i];
    }
    BNF_HD static uint32_t r2(int i) {  // R^2 mod q
        constexpr uint32_t M[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u, 0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
        return M[i];
    }
};
// r = 21888242871839275222246405745257275088548364400416034343698204186575808495617 (the group order)
struct RP {
    static constexpr uint32_t N0INV = 0xefffffffu;
    BNF_HD static uint32_t mod(int i) {
        constexpr uint32_t M[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        return M[i];
    }
    BNF_HD static uint32_t one(int i) {
        constexpr uint32_t M[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
        return M[i];
    }
    BNF_HD static uint32_t r2(int i) {
        constexpr uint32_t M[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
        return M[i];
    }
};

template <class P>
BNF_HD Fp<P> zero() {
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = 0;
    return r;
}
template <class P>
BNF_HD Fp<P> one() {
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = P::one(i);
    return r;
}
template <class P>
BNF_HD bool is_zero(const Fp<P>& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i];
    return o == 0;
}
template <class P>
BNF_HD bool equal(const Fp<P>& a, const Fp<P>& b) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i] ^ b.v[i];
    return o == 0;
}
template <class P>
BNF_HD bool geq_mod(const Fp<P>& a) {  // a >= modulus
#pragma unroll
    for (int i = 7; i >= 0; i--) {
        if (a.v[i] != P::mod(i)) return a.v[i] > P::mod(i);
    }
    return true;
}
template <class P>
BNF_HD void sub_mod_raw(Fp<P>& a) {
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint64_t d = (uint64_t)a.v[i] - P::mod(i) - borrow;
        a.v[i] = (uint32_t)d;
        borrow = (d >> 32) & 1;
    }
}
template <class P>
BNF_HD Fp<P> add(const Fp<P>& a, const Fp<P>& b) {
    Fp<P> r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)a.v[i] + b.v[i];
        r.v[i] = (uint32_t)c;
        c >>= 32;
    }
    if (geq_mod(r)) sub_mod_raw(r);  // both moduli are below 2^254: no carry out of 256 bits
    return r;
}
template <class P>
BNF_HD Fp<P> sub(const Fp<P>& a, const Fp<P>& b) {
    Fp<P> r;
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint64_t d = (uint64_t)a.v[i] - b.v[i] - borrow;
        r.v[i] = (uint32_t)d;
        borrow = (d >> 32) & 1;
    }
    if (borrow) {
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            c += (uint64_t)r.v[i] + P::mod(i);
            r.v[i] = (uint32_t)c;
            c >>= 32;
        }
    }
    return r;
}
template <class P>
BNF_HD Fp<P> dbl(const Fp<P>& a) { return add(a, a); }
template <class P>
BNF_HD Fp<P> neg(const Fp<P>& a) { return is_zero(a) ? a : sub(zero<P>(), a); }

// Montgomery product a b R^-1 mod p (CIOS): every inner step is one 32 x 32 + 64 multiply-add (v_mad_u64_u32)
template <class P>
BNF_HD Fp<P> mul(const Fp<P>& a, const Fp<P>& b) {
    uint32_t t[10];
#pragma unroll
    for (int i = 0; i < 10; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            c += (uint64_t)a.v[j] * b.v[i] + t[j];
            t[j] = (uint32_t)c;
            c >>= 32;
        }
        c += t[8];
        t[8] = (uint32_t)c;
        t[9] = (uint32_t)(c >> 32);
        const uint32_t m = t[0] * P::N0INV;
        c = (uint64_t)m * P::mod(0) + t[0];
        c >>= 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            c += (uint64_t)m * P::mod(j) + t[j];
            t[j - 1] = (uint32_t)c;
            c >>= 32;
        }
        c += t[8];
        t[7] = (uint32_t)c;
        t[8] = t[9] + (uint32_t)(c >> 32);
    }
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = t[i];
    if (t[8] || geq_mod(r)) sub_mod_raw(r);
    return r;
}
template <class P>
BNF_HD Fp<P> sqr(const Fp<P>& a) { return mul(a, a); }

template <class P>
BNF_HD Fp<P> from_mont(const Fp<P>& a) {  // a R^-1: the canonical integer
    Fp<P> o = zero<P>();
    o.v[0] = 1;
    return mul(a, o);
}
template <class P>
BNF_HD Fp<P> to_mont(const Fp<P>& a) {
    Fp<P> r2;
#pragma unroll
    for (int i = 0; i < 8; i++) r2.v[i] = P::r2(i);
    return mul(a, r2);
}
template <class P>
BNF_HD Fp<P> load_words(const uint64_t* w) {  // four little-endian 64-bit words
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        r.v[2 * i] = (uint32_t)w[i];
        r.v[2 * i + 1] = (uint32_t)(w[i] >> 32);
    }
    return r;
}
template <class P>
BNF_HD void store_words(const Fp<P>& a, uint64_t* w) {
#pragma unroll
    for (int i = 0; i < 4; i++) w[i] = (uint64_t)a.v[2 * i] | ((uint64_t)a.v[2 * i + 1] << 32);
}

// a^(p-2) by square-and-multiply (host side of the MSM: one inversion per result)
template <class P>
inline Fp<P> inv_host(const Fp<P>& a) {
    uint32_t e[8];
    for (int i = 0; i < 8; i++) e[i] = P::mod(i);
    e[0] -= 2;  // both moduli end in ...1 / ...7: no borrow
    Fp<P> r = one<P>();
    for (int bit = 255; bit >= 0; bit--) {
        r = sqr(r);
        if ((e[bit >> 5] >> (bit & 31)) & 1) r = mul(r, a);
    }
    return r;
}

}  // namespace bnf
}  // namespace nlx
