// Polynomial helper kernels: extension-field evaluation of bit-reversed coefficient columns,
// row / Merkle-path gathers for query openings, de-interleaving.
//
// Replaces plonky2::plonk::proof::OpeningSet::new -> PolynomialCoeffs::eval (SURVEY.md §8a row
// a10) and the row / sibling lookups of MerkleTree::prove + PolynomialBatch::get_lde_values
// used by fri_prover_query_rounds (row a11).
#include <hip/hip_runtime.h>
#include "gl.hpp"
#include "launch.hpp"
#include "poly.hpp"

namespace nlx {

// ---- evaluation at an extension point ----
// Coefficients are stored bit-reversed: position j holds c_{bitrev(j)}.  Then
//   p(z) = E(whole array),  E(block of size 2m) = E(first half) + z^(n/(2m)) * E(second half),
// i.e. a pairwise tree reduction whose level-l multiplier is z^(n / 2^(l+1)).  zpow[k] = z^(2^k).
//
// Stage 1: each workgroup folds a chunk of 2^EVAL_CHUNK_LOG base-field coefficients to one
// extension element.  Stage 2 folds the per-chunk partials.
constexpr unsigned EVAL_CHUNK_LOG = 11;

__device__ __forceinline__ gl::Ext shfl_down_ext(gl::Ext v, int d) {
    gl::Ext r;
    r.a = ((uint64_t)__shfl_down((uint32_t)(v.a >> 32), d, 64) << 32) | __shfl_down((uint32_t)v.a, d, 64);
    r.b = ((uint64_t)__shfl_down((uint32_t)(v.b >> 32), d, 64) << 32) | __shfl_down((uint32_t)v.b, d, 64);
    return r;
}

// One workgroup folds a chunk of up to 2^11 elements: 8 consecutive elements per lane are folded in
// registers (3 levels), 64 lanes by wave shuffles (6 levels, no barrier), the 4 waves through LDS
// (2 levels, one barrier).
template <bool EXT_IN>
__global__ __launch_bounds__(256) void k_eval_fold(const uint64_t* __restrict__ in, size_t in_stride,
                                                   unsigned log_count,  // elements per column = 2^log_count
                                                   unsigned level0,     // tree level of the first fold
                                                   unsigned log_n, const uint64_t* __restrict__ zpow,
                                                   uint64_t* __restrict__ out, size_t out_stride) {
    __shared__ uint64_t wave_part[8];
    const unsigned tid = threadIdx.x;
    const unsigned log_chunk = log_count < EVAL_CHUNK_LOG ? log_count : EVAL_CHUNK_LOG;
    const uint32_t chunk = 1u << log_chunk;
    const size_t elem0 = (size_t)blockIdx.x << log_chunk;
    const uint64_t* col = in + (size_t)blockIdx.y * in_stride;
    auto Y = [&](unsigned l) {  // multiplier of tree level level0 + l
        const unsigned k = log_n - 1 - (level0 + l);
        return gl::Ext{zpow[2 * k], zpow[2 * k + 1]};
    };
    // ---- registers: up to 8 consecutive elements per lane ----
    const unsigned log_per = log_chunk >= 3 ? 3 : log_chunk;
    const uint32_t per = 1u << log_per;
    const uint32_t active = chunk >> log_per;  // lanes holding data (<= 256)
    gl::Ext v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = gl::Ext{0, 0};
    if (tid < active) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if ((uint32_t)i < per) {
                const size_t e = elem0 + (size_t)tid * per + i;
                v[i] = EXT_IN ? gl::Ext{col[2 * e], col[2 * e + 1]} : gl::Ext{col[e], 0};
            }
        }
    }
    unsigned l = 0;
#pragma unroll
    for (int s = 0; s < 3; s++) {
        if ((unsigned)s < log_per) {
            const gl::Ext y = Y(l);
#pragma unroll
            for (int i = 0; i < (4 >> s); i++) v[i] = gl::add(v[2 * i], gl::mul(v[2 * i + 1], y));
            l++;
        }
    }
    gl::Ext acc = v[0];
    // ---- wave: pairwise folds at distance 1, 2, 4, ... (lanes that are not a multiple of 2d carry garbage) ----
    unsigned lanes = active < 64 ? active : 64;
    for (int d = 1; (unsigned)d < lanes; d <<= 1) {
        const gl::Ext other = shfl_down_ext(acc, d);
        acc = gl::add(acc, gl::mul(other, Y(l)));
        l++;
    }
    // ---- workgroup: 4 waves ----
    const uint32_t n_waves = (active + 63) / 64;
    if (n_waves > 1) {
        if ((tid & 63) == 0) {
            wave_part[2 * (tid >> 6)] = acc.a;
            wave_part[2 * (tid >> 6) + 1] = acc.b;
        }
        __syncthreads();
        if (tid == 0) {
            gl::Ext w[4];
            for (uint32_t i = 0; i < 4; i++) w[i] = i < n_waves ? gl::Ext{wave_part[2 * i], wave_part[2 * i + 1]} : gl::Ext{0, 0};
            uint32_t cnt = n_waves;
            while (cnt > 1) {
                const gl::Ext y = Y(l);
                for (uint32_t i = 0; i < cnt / 2; i++) w[i] = gl::add(w[2 * i], gl::mul(w[2 * i + 1], y));
                cnt >>= 1;
                l++;
            }
            acc = w[0];
        }
    }
    if (tid == 0) {
        uint64_t* o = out + (size_t)blockIdx.y * out_stride + 2 * (size_t)blockIdx.x;
        o[0] = acc.a;
        o[1] = acc.b;
    }
}

// zpow[k] = z^(2^k), k in [0, count)
__global__ void k_zpow(const uint64_t* __restrict__ z, unsigned count, uint64_t* __restrict__ zpow) {
    if (threadIdx.x | blockIdx.x) return;
    gl::Ext v{z[0], z[1]};
    for (unsigned k = 0; k < count; k++) {
        zpow[2 * k] = v.a;
        zpow[2 * k + 1] = v.b;
        v = gl::mul(v, v);
    }
}

size_t eval_scratch_words(uint32_t n_cols, unsigned log_n) {
    // zpow (2 * 64) + two partial buffers
    size_t partial = log_n > EVAL_CHUNK_LOG ? ((size_t)1 << (log_n - EVAL_CHUNK_LOG)) : 1;
    return 128 + 2 * (size_t)n_cols * 2 * partial;
}

void launch_eval_br(hipStream_t st, const uint64_t* d_coeffs_br, size_t stride, uint32_t n_cols, unsigned log_n,
                    const uint64_t* d_z, uint64_t* d_out_ext, uint64_t* d_scratch, const uint64_t* d_zpow) {
    if (!n_cols) return;
    const uint64_t* zpow = d_zpow;
    size_t partial = log_n > EVAL_CHUNK_LOG ? ((size_t)1 << (log_n - EVAL_CHUNK_LOG)) : 1;
    uint64_t* bufA = d_scratch + 128;
    uint64_t* bufB = bufA + (size_t)n_cols * 2 * partial;
    if (!zpow) {  // powers z^(2^k) not supplied by the host: one-lane kernel
        hipLaunchKernelGGL(k_zpow, dim3(1), dim3(1), 0, st, d_z, log_n ? log_n : 1, d_scratch);
        zpow = d_scratch;
    }
    if (log_n <= EVAL_CHUNK_LOG) {
        hipLaunchKernelGGL(k_eval_fold<false>, dim3(1, n_cols), dim3(256), 0, st, d_coeffs_br, stride, log_n, 0u, log_n,
                           zpow, d_out_ext, (size_t)2);
        return;
    }
    unsigned log_count = log_n - EVAL_CHUNK_LOG;  // partials per column after stage 1
    hipLaunchKernelGGL(k_eval_fold<false>, dim3(1u << log_count, n_cols), dim3(256), 0, st, d_coeffs_br, stride, log_n,
                       0u, log_n, zpow, bufA, (size_t)2 << log_count);
    unsigned level = EVAL_CHUNK_LOG;
    uint64_t* src = bufA;
    uint64_t* dst = bufB;
    while (log_count > EVAL_CHUNK_LOG) {
        unsigned next = log_count - EVAL_CHUNK_LOG;
        hipLaunchKernelGGL(k_eval_fold<true>, dim3(1u << next, n_cols), dim3(256), 0, st, src, (size_t)2 << log_count,
                           log_count, level, log_n, zpow, dst, (size_t)2 << next);
        level += EVAL_CHUNK_LOG;
        log_count = next;
        uint64_t* t = src; src = dst; dst = t;
    }
    hipLaunchKernelGGL(k_eval_fold<true>, dim3(1, n_cols), dim3(256), 0, st, src, (size_t)2 << log_count, log_count,
                       level, log_n, zpow, d_out_ext, (size_t)2);
}

// ---- query openings ----
// leaf index l (plonky2 order = bit-reversed LDE index) -> position in the coset-major table
__device__ __forceinline__ size_t leaf_to_pos(uint64_t leaf, unsigned log_n, unsigned rate_bits) {
    uint32_t hi = (uint32_t)(leaf >> log_n), lo = (uint32_t)(leaf & (((uint64_t)1 << log_n) - 1));
    return ((size_t)gl::bitrev32(hi, rate_bits) << log_n) + gl::bitrev32(lo, log_n);
}

__global__ void k_gather_rows(const uint64_t* __restrict__ lde, size_t col_stride, uint32_t n_cols, unsigned log_n,
                              unsigned rate_bits, const uint64_t* __restrict__ idx, size_t k,
                              uint64_t* __restrict__ rows_out) {
    size_t q = blockIdx.x;
    if (q >= k) return;
    if (idx[q] >> (log_n + rate_bits)) return;  // device-resident indices cannot be checked on the host: never read out of range
    size_t pos = leaf_to_pos(idx[q], log_n, rate_bits);
    for (uint32_t c = threadIdx.x; c < n_cols; c += blockDim.x) rows_out[q * n_cols + c] = lde[(size_t)c * col_stride + pos];
}
void launch_gather_rows(hipStream_t st, const uint64_t* d_lde, size_t col_stride, uint32_t n_cols, unsigned log_n,
                        unsigned rate_bits, const uint64_t* d_idx, size_t k, uint64_t* d_rows_out) {
    if (!k || !n_cols) return;
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)k), dim3(128), 0, st, d_lde, col_stride, n_cols, log_n, rate_bits,
                       d_idx, k, d_rows_out);
}

// siblings bottom-up from level-major digests
__global__ void k_gather_paths(const uint64_t* __restrict__ digests, unsigned log_leaves, unsigned cap_height,
                               const uint64_t* __restrict__ idx, size_t k, uint64_t* __restrict__ paths_out) {
    size_t q = blockIdx.x;
    unsigned path_len = log_leaves - cap_height;
    unsigned lvl = threadIdx.x >> 2, w = threadIdx.x & 3;
    if (q >= k || lvl >= path_len) return;
    if (idx[q] >> log_leaves) return;  // out-of-range leaf index: leave the output untouched
    // offset of level `lvl` in words: 4 * (L + L/2 + ... ) = 4 * (2L - L >> (lvl-1)) ...
    size_t L = (size_t)1 << log_leaves;
    size_t off = 4 * (2 * L - (2 * L >> lvl));
    size_t node = (idx[q] >> lvl) ^ 1;
    paths_out[(q * path_len + lvl) * 4 + w] = digests[off + node * 4 + w];
}
void launch_gather_paths(hipStream_t st, const uint64_t* d_digests, unsigned log_leaves, unsigned cap_height,
                         const uint64_t* d_idx, size_t k, uint64_t* d_paths_out) {
    if (!k || log_leaves <= cap_height) return;
    hipLaunchKernelGGL(k_gather_paths, dim3((unsigned)k), dim3(128), 0, st, d_digests, log_leaves, cap_height, d_idx,
                       k, d_paths_out);
}

// whole table in plonky2 leaf order, row-major (tests / debugging only)
__global__ void k_table_to_leaves(const uint64_t* __restrict__ lde, size_t col_stride, uint32_t n_cols,
                                  unsigned log_n, unsigned rate_bits, uint64_t* __restrict__ leaves) {
    size_t leaf = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (leaf >> (log_n + rate_bits)) return;
    size_t pos = leaf_to_pos(leaf, log_n, rate_bits);
    for (uint32_t c = 0; c < n_cols; c++) leaves[leaf * n_cols + c] = lde[(size_t)c * col_stride + pos];
}
void launch_table_to_leaves(hipStream_t st, const uint64_t* d_lde, size_t col_stride, uint32_t n_cols,
                            unsigned log_n, unsigned rate_bits, uint64_t* d_leaves) {
    size_t rows = (size_t)1 << (log_n + rate_bits);
    hipLaunchKernelGGL(k_table_to_leaves, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, d_lde, col_stride,
                       n_cols, log_n, rate_bits, d_leaves);
}

// ---- element-wise field operations (self-test entry point nlx_field_ops) ----
// out[0][i] = a*b, out[1][i] = a+b, out[2][i] = a-b, out[3][i] = a^-1 (0 for a = 0),
// out[4][i] = mul_loose(a_raw, b_raw) canonicalised, where a_raw / b_raw are the inputs WITHOUT prior
// canonicalisation (exercises the carry / borrow edges of the hand-written multiply on values >= p).
__global__ void k_field_ops(const uint64_t* __restrict__ a, const uint64_t* __restrict__ b, size_t n,
                            uint64_t* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t ar = a[i], br = b[i];
    const uint64_t x = gl::canon(ar), y = gl::canon(br);
    out[i] = gl::mul(x, y);
    out[n + i] = gl::add(x, y);
    out[2 * n + i] = gl::sub(x, y);
    out[3 * n + i] = x ? gl::inv(x) : 0;
    out[4 * n + i] = gl::canon(gl::mul_loose(ar, br));
}
void launch_field_ops(hipStream_t st, const uint64_t* d_a, const uint64_t* d_b, size_t n, uint64_t* d_out) {
    if (!n) return;
    hipLaunchKernelGGL(k_field_ops, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_a, d_b, n, d_out);
}

}  // namespace nlx
