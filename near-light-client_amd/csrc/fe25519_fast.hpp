// Fast arithmetic mod 2^255 - 19 for the sequential part of the Ed25519 trace generator (k_ed_scan): values only, no
// witness cells.  Ten signed limbs of 26 bits (value = sum l_i 2^(26 i), 260 bits; 2^260 = 608 mod p).  A product is
// a plain 19-column schoolbook in int64 (|column| < 10 * 2^54), carried to 26-bit limbs, the upper ten folded with
// 608 and carried again - 100 multiply-adds against the 256 + quotient + carry chain of the witness unit.  Results
// are only ever compared after freeze() (canonical), so they equal the witness code's canonical values bit for bit;
// tests/native/ed25519_host_check.cpp runs both and compares.
#pragma once
#include "fp25519.hpp"

namespace nlx {
namespace fe {

struct Fe { int64_t l[10]; };  // loosely reduced: |l_i| < 2^27

FP_HD inline Fe from_limbs16(const uint32_t* x) {  // 16 x 16-bit limbs, any value < 2^256
    Fe r;
    FP_UNROLL
    for (int i = 0; i < 10; i++) {
        // bits [26 i, 26 i + 26)
        const int bit = 26 * i, w = bit >> 4, s = bit & 15;
        uint64_t v = 0;
        FP_UNROLL
        for (int k = 0; k < 3; k++)
            if (w + k < 16) v |= (uint64_t)x[w + k] << (16 * k);
        r.l[i] = (int64_t)((v >> s) & 0x3FFFFFF);
    }
    return r;
}

FP_HD inline Fe add(const Fe& a, const Fe& b) { Fe r; FP_UNROLL for (int i = 0; i < 10; i++) r.l[i] = a.l[i] + b.l[i]; return r; }
FP_HD inline Fe sub(const Fe& a, const Fe& b) { Fe r; FP_UNROLL for (int i = 0; i < 10; i++) r.l[i] = a.l[i] - b.l[i]; return r; }
FP_HD inline Fe dbl(const Fe& a) { Fe r; FP_UNROLL for (int i = 0; i < 10; i++) r.l[i] = 2 * a.l[i]; return r; }
FP_HD inline Fe neg(const Fe& a) { Fe r; FP_UNROLL for (int i = 0; i < 10; i++) r.l[i] = -a.l[i]; return r; }

// carry a 10-limb value whose limbs fit int64 comfortably to |l_i| <= 2^25 + small (balanced), folding the top carry
FP_HD inline void carry10(int64_t* l) {
    FP_UNROLL
    for (int pass = 0; pass < 2; pass++) {
        FP_UNROLL
        for (int i = 0; i < 10; i++) {
            const int64_t c = (l[i] + ((int64_t)1 << 25)) >> 26;  // round to nearest: balanced limbs
            l[i] -= c * ((int64_t)1 << 26);
            if (i < 9) l[i + 1] += c;
            else l[0] += 608 * c;
        }
    }
}

FP_HD inline Fe mul(const Fe& a, const Fe& b) {  // inputs |l_i| < 2^28
    int64_t t[20];
    FP_UNROLL
    for (int k = 0; k < 20; k++) t[k] = 0;
    // the limbs fit 32 bits (|l_i| < 2^28): 32 x 32 -> 64-bit multiply-adds, one instruction each on the device
    int32_t a32[10], b32[10];
    FP_UNROLL
    for (int i = 0; i < 10; i++) {
        a32[i] = (int32_t)a.l[i];
        b32[i] = (int32_t)b.l[i];
    }
    FP_UNROLL
    for (int i = 0; i < 10; i++) {
        FP_UNROLL
        for (int j = 0; j < 10; j++) t[i + j] += (int64_t)a32[i] * b32[j];
    }
    // carry the 19 columns to 26-bit limbs (t[19] collects the last carry)
    FP_UNROLL
    for (int k = 0; k < 19; k++) {
        const int64_t c = (t[k] + ((int64_t)1 << 25)) >> 26;
        t[k] -= c * ((int64_t)1 << 26);
        t[k + 1] += c;
    }
    Fe r;
    FP_UNROLL
    for (int i = 0; i < 10; i++) r.l[i] = t[i] + 608 * t[i + 10];
    carry10(r.l);
    return r;
}

// canonical value as 16 x 16-bit limbs
FP_HD inline void freeze(const Fe& a, uint32_t* out) {
    int64_t l[10];
    FP_UNROLL
    for (int i = 0; i < 10; i++) l[i] = a.l[i];
    carry10(l);
    // make every limb non-negative in [0, 2^26): floor carries, top carry folded (twice is enough after carry10)
    FP_UNROLL
    for (int pass = 0; pass < 3; pass++) {
        FP_UNROLL
        for (int i = 0; i < 10; i++) {
            const int64_t c = l[i] >> 26;  // floor
            l[i] -= c * ((int64_t)1 << 26);
            if (i < 9) l[i + 1] += c;
            else l[0] += 608 * c;
        }
    }
    // now 0 <= value < 2^260 with non-negative limbs; reduce bits 255..259: value = hi * 2^255 + lo -> lo + 19 hi
    FP_UNROLL
    for (int pass = 0; pass < 2; pass++) {
        const int64_t hi = l[9] >> 21;  // bit 255 is bit 21 of limb 9 (9 * 26 = 234)
        l[9] &= (1 << 21) - 1;
        l[0] += 19 * hi;
        FP_UNROLL
        for (int i = 0; i < 9; i++) {
            const int64_t c = l[i] >> 26;
            l[i] &= (1 << 26) - 1;
            l[i + 1] += c;
        }
    }
    // value < 2^255 + small; subtract p if value >= p: value + 19 >= 2^255
    int64_t m[10];
    int64_t c = 19;
    FP_UNROLL
    for (int i = 0; i < 10; i++) {
        const int64_t v = l[i] + c;
        m[i] = v & ((1 << 26) - 1);
        c = v >> 26;
    }
    const bool ge = (m[9] >> 21) != 0;  // value + 19 reached 2^255
    if (ge) {
        m[9] &= (1 << 21) - 1;
        FP_UNROLL
        for (int i = 0; i < 10; i++) l[i] = m[i];
    }
    // repack 10 x 26 bits -> 16 x 16 bits
    FP_UNROLL
    for (int w = 0; w < 16; w++) {
        const int bit = 16 * w, i = bit / 26, s = bit % 26;
        uint64_t v = (uint64_t)l[i] >> s;
        if (i + 1 < 10) v |= (uint64_t)l[i + 1] << (26 - s);
        out[w] = (uint32_t)(v & 0xFFFF);
    }
}

}  // namespace fe
}  // namespace nlx
