// Internal launcher declarations shared by the HIP translation units and the C-ABI layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace nlx {

// ---- hash_kernels.hip ----
void launch_permute_batch(hipStream_t st, uint64_t* d_states, size_t n);
void launch_hash_leaves_rowmajor(hipStream_t st, const uint64_t* d_rows, uint32_t row_len, size_t n_rows,
                                 uint64_t* d_digests);
const uint64_t* launch_merkle_levels(hipStream_t st, uint64_t* d_digests, size_t n_leaves, unsigned cap_height, uint32_t n_trees = 1,
                                     size_t tree_words = 0);   // n_trees trees of the same shape, tree_words apart: one set of launches
// LDE-table leaf hashing: table is [col][coset r][k] (coset-major natural order);
// the digest of point (r,k) is written at tree position bitrev3(r)*n + bitrev(k).
// batch_cols > 0: ceil(n_cols / batch_cols) batches of columns, each with its own tree (tree_words apart in d_digests)
void launch_hash_lde_leaves(hipStream_t st, const uint64_t* d_lde, size_t col_stride, uint32_t n_cols,
                            unsigned log_n, unsigned rate_bits, uint64_t* d_digests, uint32_t batch_cols = 0, size_t tree_words = 0);
// grouped leaves (n_cols > group > 0): leaf = hash_no_pad of the digests of the row's runs of `group` columns;
// d_group_digests: scratch of 4 * ceil(n_cols / group) columns x L words, alive until the stream has run this
void launch_hash_lde_leaves_grouped(hipStream_t st, const uint64_t* d_lde, size_t col_stride, uint32_t n_cols, uint32_t group,
                                    unsigned log_n, unsigned rate_bits, uint64_t* d_group_digests, uint64_t* d_digests);

// ---- ntt_kernels.hip ----
struct NttTables {
    // fwd[k] / inv[k]: device pointer to w_{2^k}^e (resp. w^-e), e in [0, 2^(k-1)); k in [1, max_log]
    const uint64_t* fwd[33];
    const uint64_t* inv[33];
    unsigned max_log;
};
// values (natural) -> coefficients in bit-reversed order, scaled by 1/n.  src and dst may alias.
void launch_intt_dif(hipStream_t st, const NttTables& tb, const uint64_t* src, size_t src_stride, uint64_t* dst,
                     size_t dst_stride, uint32_t n_cols, unsigned log_n);
// coefficients (bit-reversed) -> values on 2^rate_bits cosets, table dst[col][r][k].
// scale_br: [2^rate_bits][n] table, scale_br[r][j] = (shift * w_L^r)^bitrev(j).
void launch_lde_dit(hipStream_t st, const NttTables& tb, const uint64_t* coeffs_br, size_t src_stride,
                    uint64_t* dst, size_t dst_stride, uint32_t n_cols, unsigned log_n, unsigned rate_bits,
                    const uint64_t* scale_br);
// plain forward transform pieces for the nlx_ntt_batch entry point
void launch_ntt_dif_fwd(hipStream_t st, const NttTables& tb, uint64_t* data, size_t stride, uint32_t n_cols,
                        unsigned log_n, bool inverse, const uint64_t* prescale_nat);
void launch_bitrev_permute(hipStream_t st, const uint64_t* src, uint64_t* dst, size_t stride, uint32_t n_cols,
                           unsigned log_n, const uint64_t* postscale_nat);
// in place and coalesced (tiles through LDS); false (nothing launched) below 2^12 points: use launch_bitrev_permute
bool launch_bitrev_inplace(hipStream_t st, uint64_t* data, size_t stride, uint32_t n_cols, unsigned log_n, const uint64_t* postscale_nat);
// natural -> natural, no reordering pass (2^18 .. 2^28 points; false = size out of range, nothing launched)
bool launch_ntt_dif_natural(hipStream_t st, const NttTables& tb, uint64_t* data, uint64_t* tmp, size_t stride, uint32_t n_cols,
                            unsigned log_n, bool inverse, const uint64_t* prescale_nat, const uint64_t* postscale_nat);
void launch_fill_coset_scale_br(hipStream_t st, uint64_t* d_table, unsigned log_n, unsigned rate_bits,
                                uint64_t shift, bool inverse);
void launch_intt_dif_cosets(hipStream_t st, const NttTables& tb, uint64_t* data, uint32_t n_y, unsigned log_n,
                            unsigned rate_bits, const uint64_t* post_scale_br);
void launch_ntt_split_level(hipStream_t st, uint64_t* mine, const uint64_t* theirs, size_t m, uint32_t n_cols, bool upper,
                            const uint64_t* w_n_table, size_t idx0, unsigned level);
void launch_fill_powers(hipStream_t st, uint64_t* d_table, size_t count, uint64_t base, uint64_t first);

}  // namespace nlx
