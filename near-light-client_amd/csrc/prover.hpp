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
    uint32_t gate_const0;   // first gate-constant column of cs: n_selectors + the lookup selector columns
    uint32_t n_lk_terms;    // lookup terms per challenge round (0 without tables): they sit between the permutation terms and
                            // the gate constraints in the vanishing-term list, so the gates' alpha powers start that much later
    uint32_t accumulate;    // 1: out[] already holds the lookup terms' share of both sums (launch_lookup_terms); it is added in
    uint32_t num_wires;
    // work split of one tile of 64 points over the kernel's quotient_waves() waves: row w lists wave w's items, terminated by
    // 0xFFFFFFFF; low 16 bits g < n_gates = gate g (bits 16.. = part mask for a gate evaluated in parts: PoseidonGate),
    // n_gates + c = the permutation argument of challenge c (device, host-built)
    const uint32_t* work;
    uint32_t work_stride;
};
void launch_quotient(hipStream_t st, const QuotientParams& p);
void launch_quotient_poseidon(hipStream_t st, const QuotientParams& p, uint32_t gate, bool add);   // PoseidonGate part 2, before launch_quotient
size_t quotient_lds_bytes(uint32_t num_wires, uint32_t n_consts_all);
uint32_t quotient_waves();
void launch_l0_table(hipStream_t st, uint64_t* d_out, unsigned log_n, unsigned rate_bits, const uint64_t* d_coset_base,
                     const uint64_t* d_w_n_table);
void launch_quotient_chunks(hipStream_t st, const uint64_t* d_in, uint64_t* d_out, unsigned log_n, unsigned rate_bits,
                            uint32_t nc, const uint64_t* d_w_R_inv_pows, const uint64_t* d_chunk_scale);

// ---- the lookup argument (lookup_arg.hip; plonky2 prover::{set_lookup_wires, compute_lookup_polys},
// vanishing_poly::check_lookup_constraints) ----
struct LookupShape {
    uint32_t num_luts, n_lu_slots, n_lut_slots, lu_degree, lut_degree, n_sldc;
};
struct LookupTableDev {       // one table, device-resident
    uint32_t len, lookups, last_lu, last_lut, first_lut, pad_;
    const uint32_t* pairs;    // len entries: input | output << 16
    const int32_t* idx_of;    // 65 536 entries: table index of an input value, -1 = not in the table
    uint32_t* mult;           // len counters (scratch of set_lookup_wires)
};
// set_lookup_wires on the device witness ([num_wires][n] subgroup values, written in place): multiplicity wires of the
// LookupTableGate rows, padding slots of each table's last LookupGate row.  *d_err (device u32, zeroed here) becomes non-zero
// when a looked-up input is not in its table.  h_tabs: host copy of d_tabs (launch geometry).
void launch_set_lookup_wires(hipStream_t st, const LookupShape& s, const LookupTableDev* d_tabs, const LookupTableDev* h_tabs,
                             uint64_t* d_wires, size_t n, uint32_t* d_mult_all, size_t mult_words, uint32_t* d_err);
// compute_lookup_polys for every challenge round: d_cols = [nc * (1 + S)][n] (RE, SLDC_0..S-1 per round), zero-filled here;
// deltas: 4 per round (A, B, alpha, delta)
void launch_lookup_polys(hipStream_t st, const LookupShape& s, const LookupTableDev* d_tabs, const LookupTableDev* h_tabs,
                         const uint64_t* d_wires, size_t n, uint32_t nc, const uint64_t* d_deltas, uint64_t* d_cols);
struct LookupTermsParams {
    const uint64_t* cs;        // constants+sigmas LDE [col][L]; the lookup selectors start at column sel0
    const uint64_t* wires;     // [col][L]
    const uint64_t* zs;        // [col][L]; round c's lookup polynomials start at column lk0 + c (1 + S)
    const uint64_t* deltas;    // device: 4 per round
    const uint64_t* lut_polys; // device: [nc][num_luts] get_lut_poly values
    const uint64_t* alpha_pows;  // device: [2][alpha_stride]
    uint64_t* out;             // [nc][L]: sum over BOTH rounds' lookup terms times alpha_c^(term index)
    uint32_t alpha_stride, t_lk; // t_lk = index of the first lookup term in the vanishing-term list
    uint32_t log_n, rate_bits, nc, sel0, lk0, n_lk_terms;
    LookupShape s;
};
void launch_lookup_terms(hipStream_t st, const LookupTermsParams& p);

constexpr int FRI_VIEWS = 33;   // FRI_MAX_ORACLES (fri.hpp) + the trailing column group of plonky2's lookup argument
struct FriCombineParams {
    const uint64_t* tables[FRI_VIEWS];  // LDE column groups in the zeta batch's order: the oracles, then (plonky2 with lookup
                                        // tables) the trailing columns of one of them as a group of their own
    uint32_t n_cols[FRI_VIEWS];
    const uint64_t* alpha_pows; // device ext table, sum(n_cols) entries
    const uint64_t* coset_base;
    const uint64_t* w_n_table;
    uint64_t zeta[2], gzeta[2], c0[2], c1[2], alpha_nz[2];
    uint64_t* out;              // L ext values
    uint32_t log_n, rate_bits;
    uint32_t nz[FRI_VIEWS];     // the g*zeta batch = first nz[o] columns of every group o, in group order
    uint32_t nz_off[FRI_VIEWS]; // index of group o's first g*zeta polynomial in that batch
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
