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

// Stage 1 for columns longer than one chunk, as a weighted sum instead of a tree: inside a chunk the element at
// local offset j carries the weight prod_{l : bit l of j} Y(l) - the same 2^11 extension weights for every chunk and
// every column.  A lane owns the elements j = i * 256 + tid (coalesced 8-byte loads), so its weights factor as
// A[i] * B[tid] and stay in registers while the block walks `cols_per_block` columns: per coefficient two 64 x 64
// multiply-accumulates into unreduced column sums (as GateAcc, prover_kernels.hip) instead of the tree's extension
// multiplication, one reduction per lane and column, the 256 partial sums added through wave shuffles and LDS.
// Field arithmetic is exact, so the partials equal k_eval_fold<false>'s.
struct DotAcc {
    uint64_t a[4];
    uint32_t k[4];
    __device__ __forceinline__ void reset() {
#pragma unroll
        for (int i = 0; i < 4; i++) { a[i] = 0; k[i] = 0; }
    }
    __device__ __forceinline__ void mac(uint64_t c, uint64_t w) {
        const uint32_t c0 = (uint32_t)c, c1 = (uint32_t)(c >> 32), w0 = (uint32_t)w, w1 = (uint32_t)(w >> 32);
        asm("v_mad_u64_u32 %[a0], vcc, %[c0], %[w0], %[a0]\n\t"
            "v_addc_co_u32 %[k0], vcc, 0, %[k0], vcc\n\t"
            "v_mad_u64_u32 %[a1], vcc, %[c0], %[w1], %[a1]\n\t"
            "v_addc_co_u32 %[k1], vcc, 0, %[k1], vcc\n\t"
            "v_mad_u64_u32 %[a2], vcc, %[c1], %[w0], %[a2]\n\t"
            "v_addc_co_u32 %[k2], vcc, 0, %[k2], vcc\n\t"
            "v_mad_u64_u32 %[a3], vcc, %[c1], %[w1], %[a3]\n\t"
            "v_addc_co_u32 %[k3], vcc, 0, %[k3], vcc"
            : [a0] "+v"(a[0]), [a1] "+v"(a[1]), [a2] "+v"(a[2]), [a3] "+v"(a[3]), [k0] "+v"(k[0]), [k1] "+v"(k[1]),
              [k2] "+v"(k[2]), [k3] "+v"(k[3])
            : [c0] "v"(c0), [c1] "v"(c1), [w0] "v"(w0), [w1] "v"(w1)
            : "vcc");
    }
    // A0 + (A1 + A2) 2^32 + (A3 + K0) 2^64 + (K1 + K2) 2^96 + K3 2^128 (mod p): 2^64 = 2^32 - 1, 2^96 = -1, 2^128 = -2^32
    __device__ __forceinline__ uint64_t finish() const {
        const uint64_t m1 = gl::canon(a[1]), m2 = gl::canon(a[2]);
        uint64_t r = gl::canon(a[0]);
        r = gl::add(r, gl::reduce128(m1 << 32, m1 >> 32));
        r = gl::add(r, gl::reduce128(m2 << 32, m2 >> 32));
        r = gl::add(r, gl::mul(gl::canon(a[3]), gl::EPS));
        r = gl::add(r, gl::mul((uint64_t)k[0], gl::EPS));
        r = gl::sub(r, (uint64_t)k[1] + k[2]);
        r = gl::sub(r, (uint64_t)k[3] << 32);
        return r;
    }
};

constexpr uint32_t EVAL_DOT_MAX_COLS = 64;  // columns per block (LDS for their wave partials)

__global__ __launch_bounds__(256) void k_eval_dot(const uint64_t* __restrict__ in, size_t in_stride, uint32_t n_cols,
                                                  uint32_t cols_per_block, unsigned log_n, const uint64_t* __restrict__ zpow,
                                                  uint64_t* __restrict__ out, size_t out_stride) {
    __shared__ uint64_t wave_part[EVAL_DOT_MAX_COLS * 4 * 2];
    const unsigned tid = threadIdx.x;
    const size_t elem0 = (size_t)blockIdx.x << EVAL_CHUNK_LOG;
    auto Y = [&](unsigned l) {
        const unsigned k = log_n - 1 - l;
        return gl::Ext{zpow[2 * k], zpow[2 * k + 1]};
    };
    gl::Ext w[8];
    {
        gl::Ext b{1, 0};
        for (unsigned l = 0; l < 8; l++)
            if ((tid >> l) & 1) b = gl::mul(b, Y(l));
        const gl::Ext y8 = Y(8), y9 = Y(9), y10 = Y(10);
        const gl::Ext y89 = gl::mul(y8, y9);
        w[0] = b;
        w[1] = gl::mul(b, y8);
        w[2] = gl::mul(b, y9);
        w[3] = gl::mul(b, y89);
        w[4] = gl::mul(b, y10);
        w[5] = gl::mul(w[1], y10);
        w[6] = gl::mul(w[2], y10);
        w[7] = gl::mul(w[3], y10);
    }
    const uint32_t col0 = blockIdx.y * cols_per_block;
    const uint32_t col1 = col0 + cols_per_block < n_cols ? col0 + cols_per_block : n_cols;
    for (uint32_t col = col0; col < col1; col++) {
        const uint64_t* src = in + (size_t)col * in_stride + elem0 + tid;
        uint64_t c[8];
#pragma unroll
        for (int i = 0; i < 8; i++) c[i] = src[i * 256];
        DotAcc A, B;
        A.reset();
        B.reset();
#pragma unroll
        for (int i = 0; i < 8; i++) {
            A.mac(c[i], w[i].a);
            B.mac(c[i], w[i].b);
        }
        gl::Ext s{A.finish(), B.finish()};
        for (int d = 32; d >= 1; d >>= 1) s = gl::add(s, shfl_down_ext(s, d));
        if ((tid & 63) == 0) {
            uint64_t* wp = wave_part + ((size_t)(col - col0) * 4 + (tid >> 6)) * 2;
            wp[0] = s.a;
            wp[1] = s.b;
        }
    }
    __syncthreads();
    if (tid < col1 - col0) {
        const uint64_t* wp = wave_part + (size_t)tid * 8;
        gl::Ext s{wp[0], wp[1]};
        for (int i = 1; i < 4; i++) s = gl::add(s, gl::Ext{wp[2 * i], wp[2 * i + 1]});
        uint64_t* o = out + (size_t)(col0 + tid) * out_stride + 2 * (size_t)blockIdx.x;
        o[0] = s.a;
        o[1] = s.b;
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
    {
        // enough blocks to fill the chip, as many columns per block as that leaves (the weights are computed per block)
        uint32_t cpb = EVAL_DOT_MAX_COLS;
        while (cpb > 4 && ((size_t)((n_cols + cpb - 1) / cpb) << log_count) < 2048) cpb >>= 1;
        hipLaunchKernelGGL(k_eval_dot, dim3(1u << log_count, (n_cols + cpb - 1) / cpb), dim3(256), 0, st, d_coeffs_br, stride,
                           n_cols, cpb, log_n, zpow, bufA, (size_t)2 << log_count);
    }
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
