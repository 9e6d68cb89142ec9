// Poseidon hashing kernels for gfx950: batched permutation, leaf hashing (hash_or_noop) over
// column-major LDE tables and over row-major leaves, and Merkle interior levels.
//
// Replaces plonky2::hash::merkle_tree::MerkleTree::new (leaf hashing + fill_subtree),
// hashing::{hash_n_to_m_no_pad, compress} — SURVEY.md §8a rows a5, a6.
//
// Layout: the LDE table is column-major ([col][row], row index already bit-reversed) so that a
// wave hashing 64 consecutive rows reads 512 contiguous bytes per column; the row-major
// "leaves" matrix of the CPU prover is never materialised.
#include <hip/hip_runtime.h>
#include "poseidon.hpp"
#include "launch.hpp"

namespace nlx {

__global__ __launch_bounds__(256) void k_permute_batch(uint64_t* __restrict__ states, size_t n) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    uint64_t s[12];
    const ulonglong2* src = reinterpret_cast<const ulonglong2*>(states + t * 12);
#pragma unroll
    for (int i = 0; i < 6; i++) {
        ulonglong2 v = src[i];
        s[2 * i] = v.x;
        s[2 * i + 1] = v.y;
    }
    poseidon::permute(s);
    ulonglong2* dst = reinterpret_cast<ulonglong2*>(states + t * 12);
#pragma unroll
    for (int i = 0; i < 6; i++) dst[i] = make_ulonglong2(s[2 * i], s[2 * i + 1]);
}

__device__ __forceinline__ void store_digest(uint64_t* __restrict__ out, size_t idx, const uint64_t (&s)[12]) {
    ulonglong2* dst = reinterpret_cast<ulonglong2*>(out + idx * 4);
    dst[0] = make_ulonglong2(gl::canon(s[0]), gl::canon(s[1]));
    dst[1] = make_ulonglong2(gl::canon(s[2]), gl::canon(s[3]));
}

// Leaf digests of a column-major table: element (c, row) at cols[c * col_stride + row].
// One lane per row; columns are absorbed eight at a time (overwrite-mode sponge, no padding).
__global__ __launch_bounds__(256) void k_hash_leaves_colmajor(const uint64_t* __restrict__ cols, size_t col_stride,
                                                              uint32_t n_cols, size_t n_rows,
                                                              uint64_t* __restrict__ digests) {
    size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    uint64_t s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = 0;
    const uint64_t* p = cols + row;
    if (n_cols <= 4) {  // hash_or_noop: copy, zero padded
        for (uint32_t c = 0; c < n_cols; c++) {
            uint64_t v = p[(size_t)c * col_stride];
            if (c == 0) s[0] = v; else if (c == 1) s[1] = v; else if (c == 2) s[2] = v; else s[3] = v;
        }
        store_digest(digests, row, s);
        return;
    }
    uint32_t c = 0;
    for (; c + 8 <= n_cols; c += 8) {
#pragma unroll
        for (int j = 0; j < 8; j++) s[j] = p[(size_t)(c + j) * col_stride];
        poseidon::permute_loose(s);
    }
    if (c < n_cols) {
        uint32_t rem = n_cols - c;
#pragma unroll
        for (int j = 0; j < 8; j++)
            if ((uint32_t)j < rem) s[j] = p[(size_t)(c + j) * col_stride];
        poseidon::permute_loose(s);
    }
    store_digest(digests, row, s);
}

// Leaf digests of row-major leaves (the MerkleTree::new(leaves, cap_height) calling convention).
__global__ __launch_bounds__(256) void k_hash_leaves_rowmajor(const uint64_t* __restrict__ rows, uint32_t row_len,
                                                              size_t n_rows, uint64_t* __restrict__ digests) {
    size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    uint64_t s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = 0;
    const uint64_t* p = rows + row * (size_t)row_len;
    if (row_len <= 4) {
        for (uint32_t c = 0; c < row_len; c++) {
            uint64_t v = p[c];
            if (c == 0) s[0] = v; else if (c == 1) s[1] = v; else if (c == 2) s[2] = v; else s[3] = v;
        }
        store_digest(digests, row, s);
        return;
    }
    uint32_t c = 0;
    for (; c + 8 <= row_len; c += 8) {
#pragma unroll
        for (int j = 0; j < 8; j++) s[j] = p[c + j];
        poseidon::permute_loose(s);
    }
    if (c < row_len) {
        uint32_t rem = row_len - c;
#pragma unroll
        for (int j = 0; j < 8; j++)
            if ((uint32_t)j < rem) s[j] = p[c + j];
        poseidon::permute_loose(s);
    }
    store_digest(digests, row, s);
}

// One interior level: parent[i] = two_to_one(child[2i], child[2i+1]).
__global__ __launch_bounds__(256) void k_merkle_level(const uint64_t* __restrict__ children,
                                                      uint64_t* __restrict__ parents, size_t n_parents) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_parents) return;
    const ulonglong2* src = reinterpret_cast<const ulonglong2*>(children + i * 8);
    uint64_t s[12];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        ulonglong2 v = src[k];
        s[2 * k] = v.x;
        s[2 * k + 1] = v.y;
    }
    s[8] = s[9] = s[10] = s[11] = 0;
    poseidon::permute_loose(s);
    store_digest(parents, i, s);
}

// ---- host launchers (stream-ordered, no synchronisation) ----
void launch_permute_batch(hipStream_t st, uint64_t* d_states, size_t n) {
    if (!n) return;
    hipLaunchKernelGGL(k_permute_batch, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_states, n);
}

void launch_hash_leaves_colmajor(hipStream_t st, const uint64_t* d_cols, size_t col_stride, uint32_t n_cols,
                                 size_t n_rows, uint64_t* d_digests) {
    if (!n_rows) return;
    hipLaunchKernelGGL(k_hash_leaves_colmajor, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, st, d_cols,
                       col_stride, n_cols, n_rows, d_digests);
}

void launch_hash_leaves_rowmajor(hipStream_t st, const uint64_t* d_rows, uint32_t row_len, size_t n_rows,
                                 uint64_t* d_digests) {
    if (!n_rows) return;
    hipLaunchKernelGGL(k_hash_leaves_rowmajor, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, st, d_rows,
                       row_len, n_rows, d_digests);
}

// digests: level-major, level 0 = n_leaves digests; builds levels down to the cap level.
// Returns a device pointer to the cap level (2^cap_height digests) inside d_digests.
const uint64_t* launch_merkle_levels(hipStream_t st, uint64_t* d_digests, size_t n_leaves, unsigned cap_height) {
    size_t cap = (size_t)1 << cap_height;
    uint64_t* cur = d_digests;
    size_t lvl = n_leaves;
    while (lvl > cap) {
        uint64_t* nxt = cur + lvl * 4;
        size_t half = lvl >> 1;
        hipLaunchKernelGGL(k_merkle_level, dim3((unsigned)((half + 255) / 256)), dim3(256), 0, st, cur, nxt, half);
        cur = nxt;
        lvl = half;
    }
    return cur;
}

}  // namespace nlx

namespace nlx {

// Leaf digests of an LDE table stored coset-major ([col][r][k], value at the point
// g*w_L^(8k+r)).  plonky2 orders Merkle leaves by the bit-reversed LDE index, so the digest of
// (r,k) lands at tree position bitrev_b(r)*n + bitrev_logn(k): a 32-byte scatter per row
// instead of a transposed copy of the whole table.
__global__ __launch_bounds__(256) void k_hash_lde_leaves(const uint64_t* __restrict__ lde, size_t col_stride,
                                                         uint32_t n_cols, unsigned log_n, unsigned rate_bits,
                                                         uint64_t* __restrict__ digests) {
    size_t pos = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >> (log_n + rate_bits)) return;
    uint64_t s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = 0;
    const uint64_t* p = lde + pos;
    const uint32_t r = (uint32_t)(pos >> log_n), k = (uint32_t)(pos & (((size_t)1 << log_n) - 1));
    const size_t leaf = ((size_t)gl::bitrev32(r, rate_bits) << log_n) + gl::bitrev32(k, log_n);
    if (n_cols <= 4) {
        for (uint32_t c = 0; c < n_cols; c++) {
            uint64_t v = p[(size_t)c * col_stride];
            if (c == 0) s[0] = v; else if (c == 1) s[1] = v; else if (c == 2) s[2] = v; else s[3] = v;
        }
        store_digest(digests, leaf, s);
        return;
    }
    uint32_t c = 0;
    for (; c + 8 <= n_cols; c += 8) {
#pragma unroll
        for (int j = 0; j < 8; j++) s[j] = p[(size_t)(c + j) * col_stride];
        poseidon::permute_loose(s);
    }
    if (c < n_cols) {
        uint32_t rem = n_cols - c;
#pragma unroll
        for (int j = 0; j < 8; j++)
            if ((uint32_t)j < rem) s[j] = p[(size_t)(c + j) * col_stride];
        poseidon::permute_loose(s);
    }
    store_digest(digests, leaf, s);
}

void launch_hash_lde_leaves(hipStream_t st, const uint64_t* d_lde, size_t col_stride, uint32_t n_cols,
                            unsigned log_n, unsigned rate_bits, uint64_t* d_digests) {
    size_t rows = (size_t)1 << (log_n + rate_bits);
    hipLaunchKernelGGL(k_hash_lde_leaves, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, d_lde, col_stride,
                       n_cols, log_n, rate_bits, d_digests);
}

}  // namespace nlx
