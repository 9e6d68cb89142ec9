// Parameter blocks and launchers of the prover-stage kernels (prover_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/nlx.h"

namespace nlx {

struct GateDev {
    uint32_t kind, selector_index, group_start, group_end, index, param0, param1;
};

struct ZsParams {
    const uint64_t* wires;      // [col][n] subgroup values
    size_t wires_stride;
    const uint64_t* sigmas;     // [routed][n] sigma values on the subgroup
    const uint64_t* k_is;       // device, routed entries
    const uint64_t* w_n_table;  // w_n^e, e < n/2
    uint64_t betas[2], gammas[2];
    uint64_t* out;              // [nc * (1 + npp)][n]: Z columns first, then partial products
    uint32_t log_n, routed, chunk, nc, npp;
};
size_t zs_scratch_words(unsigned log_n, uint32_t nc);
void launch_zs(hipStream_t st, const ZsParams& p, uint64_t* d_scratch);

struct QuotientParams {
    const uint64_t* cs;     // constants+sigmas LDE table [col][L]
    const uint64_t* wires;  // [col][L]
    const uint64_t* zs;     // [col][L]
    const GateDev* gates;   // device
    const uint64_t* k_is;   // device
    const uint64_t* coset_base;  // device: g * w_L^r, r < 2^rate_bits
    const uint64_t* w_n_table;
    const uint64_t* zh_inv;      // device: 1 / Z_H on coset r
    const uint64_t* l0_scaled;   // device: 1 / (n (x - 1)) per LDE point
    const uint64_t* alpha_pows;  // device: [2][alpha_stride]
    uint64_t* out;               // [nc][L]
    uint64_t betas[2], gammas[2], pih[4];
    uint32_t alpha_stride;
    uint32_t log_n, rate_bits, n_gates, n_selectors, n_consts_all, routed, chunk, nc, npp;
    uint32_t num_wires;
    // work split of one tile of 64 points over the kernel's quotient_waves() waves: row w lists wave w's items, terminated by
    // 0xFFFFFFFF; low 16 bits g < n_gates = gate g (bits 16.. = part mask for a gate evaluated in parts: PoseidonGate),
    // n_gates + c = the permutation argument of challenge c (device, host-built)
    const uint32_t* work;
    uint32_t work_stride;
};
void launch_quotient(hipStream_t st, const QuotientParams& p);
size_t quotient_lds_bytes(uint32_t num_wires, uint32_t n_consts_all);
uint32_t quotient_waves();
void launch_l0_table(hipStream_t st, uint64_t* d_out, unsigned log_n, unsigned rate_bits, const uint64_t* d_coset_base,
                     const uint64_t* d_w_n_table);
void launch_quotient_chunks(hipStream_t st, const uint64_t* d_in, uint64_t* d_out, unsigned log_n, unsigned rate_bits,
                            uint32_t nc, const uint64_t* d_w_R_inv_pows, const uint64_t* d_chunk_scale);

struct FriCombineParams {
    const uint64_t* tables[4];  // LDE tables in FRI oracle order
    uint32_t n_cols[4];
    const uint64_t* alpha_pows; // device ext table, sum(n_cols) entries
    const uint64_t* coset_base;
    const uint64_t* w_n_table;
    uint64_t zeta[2], gzeta[2], c0[2], c1[2], alpha_nz[2];
    uint64_t* out;              // L ext values
    uint32_t log_n, rate_bits;
    uint32_t nz[4];             // the g*zeta batch = first nz[o] columns of every table o, in table order
    uint32_t nz_off[4];         // index of table o's first g*zeta polynomial in that batch
};
// scratch: fri_combine_scratch_words(p) words when that is non-zero (small domains of many columns are combined in column
// slices), else may be null
size_t fri_combine_scratch_words(const FriCombineParams& p);
void launch_fri_combine(hipStream_t st, const FriCombineParams& p, uint64_t* scratch);
void launch_fri_leaves(hipStream_t st, const uint64_t* d_values, unsigned log_n, unsigned rate_bits,
                       unsigned arity_bits, uint64_t* d_digests);
void launch_fri_leaves_wide(hipStream_t st, const uint64_t* d_values, unsigned log_n, unsigned rate_bits,
                            unsigned arity_bits, uint64_t* d_digests);
void launch_fri_fold(hipStream_t st, const uint64_t* d_values, uint64_t* d_out, unsigned log_n, unsigned rate_bits,
                     unsigned arity_bits, const uint64_t beta[2], uint64_t shift_inv, const uint64_t* d_w_L_inv_table,
                     const uint64_t* d_w_A_inv_pows);
void launch_fri_final_coeffs(hipStream_t st, const uint64_t* d_values, unsigned log_n, unsigned rate_bits,
                             uint64_t shift, uint64_t* d_out, uint32_t n_out);

struct PowParams {
    uint64_t state[12];
    uint64_t max_rounds;
    uint32_t pos, bits;
};
void launch_pow_grind(hipStream_t st, const PowParams& p, unsigned long long* d_best);
void launch_fri_gather_leaf(hipStream_t st, const uint64_t* d_values, unsigned log_n, unsigned rate_bits,
                            unsigned arity_bits, const uint64_t* d_leaf_idx, uint32_t n_q, uint64_t* d_out,
                            size_t out_stride_words);
void launch_shift_indices(hipStream_t st, const uint64_t* d_in, uint64_t* d_out, uint32_t n, unsigned shift);
void launch_pow_table(hipStream_t st, uint64_t* d_out, uint64_t a0, uint64_t a1, uint32_t count, uint32_t stride);
void launch_ext_pow_table(hipStream_t st, uint64_t* d_out, const uint64_t alpha[2], uint32_t count);

}  // namespace nlx
