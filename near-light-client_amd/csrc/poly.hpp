// Launcher declarations for poly_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace nlx {
size_t eval_scratch_words(uint32_t n_cols, unsigned log_n);
// d_out_ext[c] = p_c(z) for bit-reversed coefficient columns; d_z: 2 words on device
void launch_eval_br(hipStream_t st, const uint64_t* d_coeffs_br, size_t stride, uint32_t n_cols, unsigned log_n,
                    const uint64_t* d_z, uint64_t* d_out_ext, uint64_t* d_scratch, const uint64_t* d_zpow = nullptr);
void launch_gather_rows(hipStream_t st, const uint64_t* d_lde, size_t col_stride, uint32_t n_cols, unsigned log_n,
                        unsigned rate_bits, const uint64_t* d_idx, size_t k, uint64_t* d_rows_out);
void launch_gather_paths(hipStream_t st, const uint64_t* d_digests, unsigned log_leaves, unsigned cap_height,
                         const uint64_t* d_idx, size_t k, uint64_t* d_paths_out);
void launch_table_to_leaves(hipStream_t st, const uint64_t* d_lde, size_t col_stride, uint32_t n_cols,
                            unsigned log_n, unsigned rate_bits, uint64_t* d_leaves);
void launch_field_ops(hipStream_t st, const uint64_t* d_a, const uint64_t* d_b, size_t n, uint64_t* d_out);
}  // namespace nlx
