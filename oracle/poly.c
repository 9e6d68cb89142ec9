/* ORACLE - TEST INFRASTRUCTURE ONLY (see gl.h header).
 *
 * NTT / LDE / PolynomialBatch commit.  Follows (by module name; source absent):
 *   plonky2_field::fft::{fft_classic, ifft_with_options}, polynomial::{PolynomialCoeffs::lde,
 *     coset_fft, PolynomialValues::coset_ifft}
 *   plonky2_util::{transpose, reverse_index_bits_in_place}
 *   plonky2::fri::oracle::PolynomialBatch::{from_values, from_coeffs, lde_values}
 * Any correct NTT is bit-identical (exact field arithmetic), so the classic in-place
 * bit-reverse + DIT butterfly network is used.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

void orc_field_generators(uint64_t out[2]) {
    out[0] = GL_GEN;
    out[1] = GL_POW2_GEN;
}

/* PolynomialCoeffs::eval in the base field: Horner, coefficients in natural order */
uint64_t orc_eval_poly_base(const uint64_t* coeffs, size_t n, uint64_t x) {
    uint64_t acc = 0;
    for (size_t i = n; i-- > 0;) acc = gl_add(gl_mul(acc, x), coeffs[i]);
    return acc;
}

void orc_fft(uint64_t* a, unsigned log_n) {
    size_t n = (size_t)1 << log_n;
    for (size_t i = 0; i < n; i++) {
        size_t j = gl_bitrev(i, log_n);
        if (i < j) { uint64_t t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    for (unsigned s = 1; s <= log_n; s++) {
        size_t m = (size_t)1 << s, h = m >> 1;
        uint64_t wm = gl_root_of_unity(s);
        for (size_t k = 0; k < n; k += m) {
            uint64_t w = 1;
            for (size_t j = 0; j < h; j++) {
                uint64_t t = gl_mul(w, a[k + j + h]);
                uint64_t u = a[k + j];
                a[k + j] = gl_add(u, t);
                a[k + j + h] = gl_sub(u, t);
                w = gl_mul(w, wm);
            }
        }
    }
}

void orc_ifft(uint64_t* a, unsigned log_n) {
    size_t n = (size_t)1 << log_n;
    orc_fft(a, log_n);
    uint64_t n_inv = gl_inv((uint64_t)n % GL_P);
    /* reverse all but the first, scale by 1/n (fft.rs ifft_with_options) */
    a[0] = gl_mul(a[0], n_inv);
    if (n > 1) a[n / 2] = gl_mul(a[n / 2], n_inv);
    for (size_t i = 1; i < n / 2; i++) {
        size_t j = n - i;
        uint64_t ci = gl_mul(a[j], n_inv), cj = gl_mul(a[i], n_inv);
        a[i] = ci;
        a[j] = cj;
    }
}

void orc_coset_fft(uint64_t* a, unsigned log_n, uint64_t shift) {
    size_t n = (size_t)1 << log_n;
    uint64_t p = 1;
    for (size_t i = 0; i < n; i++) { a[i] = gl_mul(a[i], p); p = gl_mul(p, shift); }
    orc_fft(a, log_n);
}

void orc_coset_ifft(uint64_t* a, unsigned log_n, uint64_t shift) {
    size_t n = (size_t)1 << log_n;
    orc_ifft(a, log_n);
    uint64_t si = gl_inv(shift), p = 1;
    for (size_t i = 0; i < n; i++) { a[i] = gl_mul(a[i], p); p = gl_mul(p, si); }
}

void orc_commit_from_coeffs(const uint64_t* coeffs, size_t n_cols, unsigned log_n, unsigned rate_bits,
                            unsigned cap_height, uint64_t* leaves_out, uint64_t* digests_out,
                            uint64_t* cap_out) {
    orc_commit_from_coeffs_g(coeffs, n_cols, log_n, rate_bits, cap_height, 0, leaves_out, digests_out, cap_out);
}

void orc_commit_from_coeffs_g(const uint64_t* coeffs, size_t n_cols, unsigned log_n, unsigned rate_bits,
                              unsigned cap_height, size_t leaf_group, uint64_t* leaves_out, uint64_t* digests_out,
                              uint64_t* cap_out) {
    size_t n = (size_t)1 << log_n;
    unsigned log_l = log_n + rate_bits;
    size_t L = (size_t)1 << log_l;
    uint64_t* own = NULL;
    if (!leaves_out) leaves_out = own = (uint64_t*)malloc(L * n_cols * 8);
#pragma omp parallel
    {
        uint64_t* tmp = (uint64_t*)malloc(L * 8);
#pragma omp for schedule(dynamic)
        for (size_t c = 0; c < n_cols; c++) {
            memcpy(tmp, coeffs + c * n, n * 8);
            memset(tmp + n, 0, (L - n) * 8); /* PolynomialCoeffs::lde = zero pad */
            orc_coset_fft(tmp, log_l, GL_GEN);
            /* transpose + reverse_index_bits_in_place on the row index */
            for (size_t i = 0; i < L; i++) leaves_out[gl_bitrev(i, log_l) * n_cols + c] = tmp[i];
        }
        free(tmp);
    }
    orc_merkle_build_g(leaves_out, L, n_cols, leaf_group, cap_height, digests_out, cap_out);
    free(own);
}

void orc_commit_from_values(const uint64_t* values, size_t n_cols, unsigned log_n, unsigned rate_bits,
                            unsigned cap_height, uint64_t* coeffs_out, uint64_t* leaves_out,
                            uint64_t* digests_out, uint64_t* cap_out) {
    orc_commit_from_values_g(values, n_cols, log_n, rate_bits, cap_height, 0, coeffs_out, leaves_out, digests_out, cap_out);
}

void orc_commit_from_values_g(const uint64_t* values, size_t n_cols, unsigned log_n, unsigned rate_bits,
                              unsigned cap_height, size_t leaf_group, uint64_t* coeffs_out, uint64_t* leaves_out,
                              uint64_t* digests_out, uint64_t* cap_out) {
    size_t n = (size_t)1 << log_n;
    uint64_t* own = NULL;
    if (!coeffs_out) coeffs_out = own = (uint64_t*)malloc(n * n_cols * 8);
    memcpy(coeffs_out, values, n * n_cols * 8);
#pragma omp parallel for schedule(dynamic)
    for (size_t c = 0; c < n_cols; c++) orc_ifft(coeffs_out + c * n, log_n);
    orc_commit_from_coeffs_g(coeffs_out, n_cols, log_n, rate_bits, cap_height, leaf_group, leaves_out, digests_out, cap_out);
    free(own);
}
