// What the AIR quotient kernels share: the launch parameters and the two helpers every evaluator of a register program uses -
// the interpreter k_air_quotient (stark.hip) and the straight-line kernels generated from fixed programs (csrc/airgen/,
// near-light-client_amd/airgen.py).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "gl.hpp"
#include "nlx.h"

namespace nlx {

struct AirParams {
    const uint64_t* const* cols; // device: cols[c] = LDE column c ([r][k], L = n << rate_bits) of whichever committed oracle holds it
    const uint64_t* program;    // device, n_words (constants canonical)
    const uint64_t* pis;        // device
    const uint64_t* coset_base; // device: g * w_{n q}^r', r' < q = 2^qdb
    const uint64_t* zh_inv;     // device: 1 / Z_H on quotient coset r'
    const uint64_t* l_inv;      // device: 1 / (n (x - 1)) on the quotient cosets, [r'][k]
    const uint64_t* periodic;   // device: [column][r'][k mod period] = P_a((g w^r')^(n/period) * w_period^k)
    const uint64_t* w_n_table;
    uint64_t* out;              // [challenge][r'][k]
    const uint32_t* seg;        // device: [segment] = {first word, end word} (n_seg > 1)
    const uint64_t* seg_mul;    // device: [segment][challenge] = alpha^(constraints after the segment)
    uint64_t* part;             // [segment][challenge][r'][k] partial sums (n_seg > 1)
    uint32_t n_seg;
    uint64_t alphas[2];
    uint64_t g_inv;             // last = g^-1 (g generates the size-n subgroup)
    uint32_t log_n, rate_bits, qdb, n_words, nc, n_regs, period_bits, n_pis;
};

// x * 2^sh (sh < 64): a 128-bit shift and one reduction instead of a general multiplication
__device__ __forceinline__ uint64_t mul_pow2(uint64_t x, uint32_t sh) {
    if (sh == 0) return x;
    return gl::reduce128(x << sh, x >> (64 - sh));
}

__device__ __forceinline__ uint64_t root_pow_(const uint64_t* __restrict__ half_table, uint32_t e, uint32_t half) {
    return e < half ? half_table[e] : gl::P - half_table[e - half];
}


// A generated evaluator of ONE fixed program (csrc/airgen/air_*.hip, written by airgen.py from the same words the interpreter
// runs): found by the hash of the canonicalised program words when a STARK is built, launched in place of k_air_quotient over
// the same grid (points / block x segments) with the same parameters, results identical word for word.
struct AirGenEntry {
    uint64_t program_hash;   // airgen_program_hash(words)
    uint32_t n_words;
    const char* name;
    void (*launch)(hipStream_t st, unsigned tiles, unsigned n_segments, const AirParams& p);
};
constexpr unsigned AIRGEN_BLOCK = 256;   // points per block of a generated kernel
inline uint64_t airgen_program_hash(const uint64_t* words, size_t n) {   // FNV-1a over the words' bytes, little-endian
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; i++)
        for (int b = 0; b < 8; b++) {
            h ^= (words[i] >> (8 * b)) & 0xFF;
            h *= 0x100000001b3ull;
        }
    return h;
}
#if defined(NLX_NO_AIRGEN)
inline const AirGenEntry* airgen_find(uint64_t, uint32_t) { return nullptr; }   // a build without the generated kernels: the interpreter runs everything
#else
const AirGenEntry* airgen_find(uint64_t program_hash, uint32_t n_words);   // csrc/airgen/registry.hip
#endif

}  // namespace nlx
