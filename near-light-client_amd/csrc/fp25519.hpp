// Witness arithmetic for the mod-(2^255 - 19) multiplication unit of near-light-client_amd/fp25519.py: 16-bit limbs,
// sum of products = c + q p over the integers, signed carries of the two-column groups.  Device-side integer code
// (one lane computes whole units); layout and bounds are documented in fp25519.py.
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FP_HD __host__ __device__
#else
#define FP_HD  // plain C++ build: tests/native/fp25519_host_check.cpp checks this arithmetic on the CPU
#endif

namespace nlx {
namespace fp {

constexpr int LIMBS = 16, Q_LIMBS = 17, N_CARRY = 15;
constexpr int64_t CARRY_OFFSET = (int64_t)1 << 22;

struct Unit {
    uint32_t c[LIMBS];
    uint32_t q[Q_LIMBS];
    uint32_t carry[N_CARRY];  // R = r + 2^22 < 2^23
};

FP_HD inline uint32_t p_limb(int j) { return j == 0 ? 0xFFEDu : (j == 15 ? 0x7FFFu : 0xFFFFu); }

// prod[k] += sum_{i+j=k} a[i] b[j]   (limbs may be a little wider than 16 bits: sums of reduced values)
FP_HD inline void mul_acc(uint64_t prod[32], const uint32_t a[LIMBS], const uint32_t b[LIMBS]) {
    for (int i = 0; i < LIMBS; i++)
        for (int j = 0; j < LIMBS; j++) prod[i + j] += (uint64_t)a[i] * b[j];
}

// Given the column sums of the products (prod[31] = 0 on entry for a single product): canonical c = total mod p,
// q = (total - c) / p, and the carries of the unit's equations.
FP_HD inline void finish(const uint64_t prod[32], Unit& u) {
    // t = the integer total in 16-bit limbs (34 of them cover two full products)
    uint32_t t[34];
    uint64_t cy = 0;
    for (int k = 0; k < 34; k++) {
        const uint64_t v = (k < 32 ? prod[k] : 0) + cy;
        t[k] = (uint32_t)(v & 0xFFFF);
        cy = v >> 16;
    }
    // total = hi * 2^255 + lo;  total = hi * p + (lo + 19 hi)
    uint32_t hi[19];
    for (int k = 0; k < 19; k++) {
        const uint32_t lo_part = k + 15 < 34 ? t[k + 15] >> 15 : 0, hi_part = k + 16 < 34 ? (t[k + 16] << 1) & 0xFFFF : 0;
        hi[k] = lo_part | hi_part;
    }
    uint32_t s[20];
    cy = 0;
    for (int k = 0; k < 20; k++) {
        uint64_t v = cy + (k < 19 ? 19ull * hi[k] : 0);
        if (k < 15) v += t[k];
        if (k == 15) v += t[15] & 0x7FFF;
        s[k] = (uint32_t)(v & 0xFFFF);
        cy = v >> 16;
    }
    // second fold: s = hi2 * 2^255 + lo2, hi2 small
    uint64_t hi2 = (s[15] >> 15) | ((uint64_t)s[16] << 1) | ((uint64_t)s[17] << 17) | ((uint64_t)s[18] << 33);
    uint32_t c[LIMBS];
    cy = 19 * hi2;
    for (int k = 0; k < LIMBS; k++) {
        const uint64_t v = cy + (k == 15 ? (s[15] & 0x7FFF) : s[k]);
        c[k] = (uint32_t)(v & 0xFFFF);
        cy = v >> 16;
    }
    // q = hi + hi2 (+ 1 if c >= p)
    bool ge = c[15] >= 0x7FFF;
    if (ge) {
        if (c[15] == 0x7FFF) {
            for (int k = 14; k >= 1 && ge; k--) ge = c[k] == 0xFFFF;
            ge = ge && c[0] >= 0xFFED;
        }
    }
    if (ge) {  // c -= p  <=>  c += 19 - 2^255
        uint64_t v = (uint64_t)c[0] + 19;
        c[0] = (uint32_t)(v & 0xFFFF);
        uint64_t k2 = v >> 16;
        for (int k = 1; k < LIMBS; k++) {
            v = (uint64_t)c[k] + k2;
            c[k] = (uint32_t)(v & 0xFFFF);
            k2 = v >> 16;
        }
        c[15] &= 0x7FFF;
    }
    cy = hi2 + (ge ? 1 : 0);
    for (int k = 0; k < Q_LIMBS; k++) {
        const uint64_t v = cy + hi[k];
        u.q[k] = (uint32_t)(v & 0xFFFF);
        cy = v >> 16;
    }
    for (int k = 0; k < LIMBS; k++) u.c[k] = c[k];
    // carries: D_k = prod[k] - c[k] - sum q_i p_(k-i);  G_m = D_2m + 2^16 D_2m+1;  G_m + r_(m-1) = 2^32 r_m
    int64_t prev = 0;
    for (int m = 0; m < LIMBS; m++) {
        int64_t d2[2];
        for (int h = 0; h < 2; h++) {
            const int k = 2 * m + h;
            int64_t d = (int64_t)prod[k] - (k < LIMBS ? (int64_t)c[k] : 0);
            const int i0 = k - LIMBS + 1 > 0 ? k - LIMBS + 1 : 0, i1 = k < Q_LIMBS - 1 ? k : Q_LIMBS - 1;
            for (int i = i0; i <= i1; i++) d -= (int64_t)u.q[i] * p_limb(k - i);
            d2[h] = d;
        }
        const int64_t g = d2[0] + d2[1] * 65536 + prev;
        prev = g >> 32;  // exact: g is a multiple of 2^32
        if (m < N_CARRY) u.carry[m] = (uint32_t)(prev + CARRY_OFFSET);
    }
}

}  // namespace fp
}  // namespace nlx
