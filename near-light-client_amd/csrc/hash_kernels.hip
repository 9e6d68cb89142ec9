// Poseidon hashing kernels for gfx950: batched permutation, leaf hashing (hash_or_noop) over
// column-major LDE tables and over row-major leaves, and Merkle interior levels.
//
// Replaces plonky2::hash::merkle_tree::MerkleTree::new (leaf hashing + fill_subtree),
// hashing::{hash_n_to_m_no_pad, compress} — SURVEY.md §8a rows a5, a6.
//
// Layout: the LDE table is column-major and coset-major ([col][r][k]) so that a wave
// hashing 64 consecutive points reads 512 contiguous bytes per column; the row-major "leaves"
// matrix of the CPU prover is never materialised - each digest is scattered to its tree position.
#include <hip/hip_runtime.h>
#include "poseidon.hpp"
#include "launch.hpp"

namespace nlx {

__global__ __launch_bounds__(256, 4) void k_permute_batch(uint64_t* __restrict__ states, size_t n) {
    // every lane of a wave stays in the kernel through the permutation: its linear layer is a matrix-core product whose constant
    // operand is spread over all 64 lanes (gl32::mds_layer_mfma); a lane beyond the end redoes the last state and stores nothing
    const size_t t_raw = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = t_raw < n;
    const size_t t = live ? t_raw : n - 1;
    uint64_t s[12];
    const ulonglong2* src = reinterpret_cast<const ulonglong2*>(states + t * 12);
#pragma unroll
    for (int i = 0; i < 6; i++) {
        ulonglong2 v = src[i];
        s[2 * i] = v.x;
        s[2 * i + 1] = v.y;
    }
    poseidon::permute(s);
    if (!live) return;
    ulonglong2* dst = reinterpret_cast<ulonglong2*>(states + t * 12);
#pragma unroll
    for (int i = 0; i < 6; i++) dst[i] = make_ulonglong2(s[2 * i], s[2 * i + 1]);
}

__device__ __forceinline__ void store_digest(uint64_t* __restrict__ out, size_t idx, const uint64_t (&s)[12]) {
    ulonglong2* dst = reinterpret_cast<ulonglong2*>(out + idx * 4);
    dst[0] = make_ulonglong2(gl::canon(s[0]), gl::canon(s[1]));
    dst[1] = make_ulonglong2(gl::canon(s[2]), gl::canon(s[3]));
}

// Leaf digests of row-major leaves (the MerkleTree::new(leaves, cap_height) calling convention).
__global__ __launch_bounds__(256, 4) void k_hash_leaves_rowmajor(const uint64_t* __restrict__ rows, uint32_t row_len,
                                                              size_t n_rows, uint64_t* __restrict__ digests) {
    const size_t row_raw = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = row_raw < n_rows;               // spare lanes redo the last row (see k_permute_batch) and store nothing
    const size_t row = live ? row_raw : n_rows - 1;
    uint64_t s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = 0;
    const uint64_t* p = rows + row * (size_t)row_len;
    if (row_len <= 4) {
        for (uint32_t c = 0; c < row_len; c++) {
            uint64_t v = p[c];
            if (c == 0) s[0] = v; else if (c == 1) s[1] = v; else if (c == 2) s[2] = v; else s[3] = v;
        }
        if (live) store_digest(digests, row, s);
        return;
    }
#pragma unroll 1
    for (uint32_t c = 0; c < row_len; c += 8) {   // one call site of the permutation (see k_hash_lde_leaves)
        const uint32_t rem = row_len - c;
#pragma unroll
        for (int j = 0; j < 8; j++)
            if ((uint32_t)j < rem) s[j] = p[c + j];
        poseidon::permute_loose(s);
    }
    if (live) store_digest(digests, row, s);
}

// One interior level: parent[i] = two_to_one(child[2i], child[2i+1]).
__global__ __launch_bounds__(256, 4) void k_merkle_level(const uint64_t* __restrict__ children,
                                                      uint64_t* __restrict__ parents, size_t n_parents, size_t tree_words) {
    children += (size_t)blockIdx.y * tree_words;   // several trees of one commitment (batches of columns), back to back
    parents += (size_t)blockIdx.y * tree_words;
    const size_t i_raw = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i_raw < n_parents;              // spare lanes redo the last parent (see k_permute_batch) and store nothing
    const size_t i = live ? i_raw : n_parents - 1;
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
    if (live) store_digest(parents, i, s);
}

constexpr size_t MERKLE_WIDE_MAX_PARENTS = (size_t)1 << 14;
constexpr unsigned MERKLE_FUSED_MAX_LEVELS = 6;  // 2^6 children per block: 32 sixteen-lane groups on the first fused level
void launch_merkle_fused(hipStream_t st, const uint64_t* children, size_t n_children, unsigned levels, uint32_t n_trees, size_t tree_words);

// ---- host launchers (stream-ordered, no synchronisation) ----
void launch_permute_batch(hipStream_t st, uint64_t* d_states, size_t n) {
    if (!n) return;
    hipLaunchKernelGGL(k_permute_batch, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_states, n);
}

void launch_hash_leaves_rowmajor(hipStream_t st, const uint64_t* d_rows, uint32_t row_len, size_t n_rows,
                                 uint64_t* d_digests) {
    if (!n_rows) return;
    hipLaunchKernelGGL(k_hash_leaves_rowmajor, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, st, d_rows,
                       row_len, n_rows, d_digests);
}

// digests: level-major, level 0 = n_leaves digests; builds levels down to the cap level.
// Returns a device pointer to the cap level (2^cap_height digests) inside d_digests.
// n_trees > 1: that many trees of the same shape, `tree_words` words apart (the batches of one commitment round): every launch
// covers all of them (blockIdx.y = tree), so ten trees cost the latency of one
const uint64_t* launch_merkle_levels(hipStream_t st, uint64_t* d_digests, size_t n_leaves, unsigned cap_height, uint32_t n_trees, size_t tree_words) {
    size_t cap = (size_t)1 << cap_height;
    uint64_t* cur = d_digests;
    size_t lvl = n_leaves;
    while (lvl > cap) {
        uint64_t* nxt = cur + lvl * 4;
        size_t half = lvl >> 1;
        // below ~2^14 parents a level no longer fills the chip with one lane per permutation: switch to sixteen lanes per
        // permutation (latency ~1/5), and - every such level being latency-bound - run up to six consecutive levels in
        // ONE launch: a block owns a subtree of 2^K children and walks it up in LDS (round 1 launched every level
        // separately: 11 launches of ~29 us for the top of a 2^19-leaf tree, now 2)
        if (half <= MERKLE_WIDE_MAX_PARENTS) {
            unsigned K = 0;
            while (K < MERKLE_FUSED_MAX_LEVELS && (lvl >> K) > cap) K++;
            launch_merkle_fused(st, cur, lvl, K, n_trees, tree_words);
            for (unsigned s = 0; s < K; s++) {
                cur += lvl * 4;
                lvl >>= 1;
            }
            continue;
        }
        hipLaunchKernelGGL(k_merkle_level, dim3((unsigned)((half + 255) / 256), n_trees), dim3(256), 0, st, cur, nxt, half, tree_words);
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
constexpr size_t HASH_LEAVES_WIDE_MAX_ROWS = (size_t)1 << 13;  // measured crossover (4 745 columns): 2^13 rows 16 vs 24 ms, 2^14 rows 31 vs 26 ms

// batch_cols > 0: the table is committed as ceil(n_cols / batch_cols) PolynomialBatches of at most batch_cols columns each
// (blockIdx.y = batch): a batch's leaf is hash_or_noop of ITS columns of the row, its digests go to tree blockIdx.y
// (`tree_words` words apart).  batch_cols = 0: one batch, as always.
__global__ __launch_bounds__(256, 4) void k_hash_lde_leaves(const uint64_t* __restrict__ lde, size_t col_stride,
                                                         uint32_t n_cols, unsigned log_n, unsigned rate_bits,
                                                         uint64_t* __restrict__ digests, uint32_t batch_cols, size_t tree_words) {
    if (batch_cols) {
        const uint32_t c0 = blockIdx.y * batch_cols;
        lde += (size_t)c0 * col_stride;
        n_cols = n_cols - c0 < batch_cols ? n_cols - c0 : batch_cols;
        digests += (size_t)blockIdx.y * tree_words;
    }
    const size_t pos_raw = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = (pos_raw >> (log_n + rate_bits)) == 0;   // spare lanes redo the last point (see k_permute_batch), store nothing
    const size_t pos = live ? pos_raw : ((size_t)1 << (log_n + rate_bits)) - 1;
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
        if (live) store_digest(digests, leaf, s);
        return;
    }
    // ONE call site of the permutation (its code is ~56 KB: a second copy for the ragged last chunk doubled what the waves of a
    // CU pull through the instruction cache); the chunk length is wave-uniform, so the guards are scalar branches
#pragma unroll 1
    for (uint32_t c = 0; c < n_cols; c += 8) {
        const uint32_t rem = n_cols - c;
#pragma unroll
        for (int j = 0; j < 8; j++)
            if ((uint32_t)j < rem) s[j] = p[(size_t)(c + j) * col_stride];
        poseidon::permute_loose(s);
    }
    if (live) store_digest(digests, leaf, s);
}

void launch_hash_lde_leaves_wide(hipStream_t st, const uint64_t* d_lde, size_t col_stride, uint32_t n_cols, unsigned log_n,
                                 unsigned rate_bits, uint64_t* d_digests, uint32_t batch_cols, size_t tree_words);

// Grouped leaves (STARK commitments of wide short traces, include/nlx.h leaf_group_cols): lane (pos, g = blockIdx.y) hashes
// the run of columns [g group, (g + 1) group) of LDE row pos - hash_no_pad, whatever the run's length - and writes the
// digest as four more COLUMNS of a small table laid out like the LDE itself ([4 g + j][pos]).  The leaf digest is then
// the ordinary leaf hash of that table (launch_hash_lde_leaves: hash_no_pad of the 4 K words of a row, scattered to the
// row's tree position).  A 4 745-column trace on 2^10 LDE rows is 594 permutations in sequence on each of 1 024 lanes as one
// leaf per lane; as 38 runs of 128 columns it is 16 permutations on each of 38 912 lanes, then 19 on 1 024 leaves.
__global__ __launch_bounds__(256, 4) void k_hash_lde_groups(const uint64_t* __restrict__ lde, size_t col_stride, uint32_t n_cols,
                                                            uint32_t group, unsigned log_L, uint64_t* __restrict__ out) {
    const size_t pos_raw = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = (pos_raw >> log_L) == 0;   // spare lanes redo the last point (see k_permute_batch), store nothing
    const size_t L = (size_t)1 << log_L, pos = live ? pos_raw : L - 1;
    const uint32_t g = blockIdx.y, c0 = g * group, c1 = c0 + group < n_cols ? c0 + group : n_cols;
    uint64_t s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = 0;
    const uint64_t* p = lde + pos;
#pragma unroll 1
    for (uint32_t c = c0; c < c1; c += 8) {   // one call site of the permutation (see k_hash_lde_leaves)
        const uint32_t rem = c1 - c;
#pragma unroll
        for (int j = 0; j < 8; j++)
            if ((uint32_t)j < rem) s[j] = p[(size_t)(c + j) * col_stride];
        poseidon::permute_loose(s);
    }
    if (!live) return;
#pragma unroll
    for (int j = 0; j < 4; j++) out[((size_t)(4 * g + j) << log_L) + pos] = gl::canon(s[j]);
}

void launch_hash_lde_leaves_grouped(hipStream_t st, const uint64_t* d_lde, size_t col_stride, uint32_t n_cols, uint32_t group,
                                    unsigned log_n, unsigned rate_bits, uint64_t* d_group_digests, uint64_t* d_digests) {
    const size_t rows = (size_t)1 << (log_n + rate_bits);
    const uint32_t K = (n_cols + group - 1) / group;
    hipLaunchKernelGGL(k_hash_lde_groups, dim3((unsigned)((rows + 255) / 256), K), dim3(256), 0, st, d_lde, col_stride, n_cols, group,
                       log_n + rate_bits, d_group_digests);
    launch_hash_lde_leaves(st, d_group_digests, rows, 4 * K, log_n, rate_bits, d_digests, 0, 0);
}

void launch_hash_lde_leaves(hipStream_t st, const uint64_t* d_lde, size_t col_stride, uint32_t n_cols,
                            unsigned log_n, unsigned rate_bits, uint64_t* d_digests, uint32_t batch_cols, size_t tree_words) {
    size_t rows = (size_t)1 << (log_n + rate_bits);
    const uint32_t n_batches = batch_cols ? (n_cols + batch_cols - 1) / batch_cols : 1;
    const uint32_t widest = batch_cols && batch_cols < n_cols ? batch_cols : n_cols;
    // Few rows of many columns (a short wide STARK trace): one lane per leaf leaves most SIMDs idle while every lane
    // walks its ceil(c/8) permutations one after the other; sixteen lanes per leaf cut that chain's latency ~5x.
    // (leaves in flight = rows x batches: with batches a short trace fills the chip with one lane per leaf much sooner)
    if (rows * n_batches <= HASH_LEAVES_WIDE_MAX_ROWS && widest > 16) {
        launch_hash_lde_leaves_wide(st, d_lde, col_stride, n_cols, log_n, rate_bits, d_digests, batch_cols, tree_words);
        return;
    }
    hipLaunchKernelGGL(k_hash_lde_leaves, dim3((unsigned)((rows + 255) / 256), n_batches), dim3(256), 0, st, d_lde, col_stride,
                       n_cols, log_n, rate_bits, d_digests, batch_cols, tree_words);
}

}  // namespace nlx

// =====================================================================================
// Latency-optimised ("wide") Poseidon for small tree levels
// =====================================================================================
// One permutation per 16-lane group, one state element per lane (lanes 12..15 idle).  A level
// with m parents occupies 16*m lanes, so levels too small to fill the chip finish in ~1/5 of
// the one-lane-per-permutation latency (the dependent chain per round is one S-box plus one
// 12-term dot product instead of twelve of each).  The linear layer exchanges elements through
// a per-group LDS slot (1 ds_write_b64 + 12 broadcast ds_read_b64 per round).
namespace nlx {

constexpr unsigned WIDE_GROUPS_PER_BLOCK = 16;  // 256 threads

__device__ __forceinline__ gl32::F permute_wide(gl32::F x, uint32_t j, volatile uint64_t* slot) {
    constexpr uint32_t C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    const uint64_t* rc = poseidon::RC_DEV;
    const uint32_t jj = j < 12 ? j : 0;
    const uint32_t diag = j == 0 ? 8u : 0u;
    // A 16-lane group lives inside ONE wave and LDS operations of a wave execute in order, so no
    // workgroup barrier is needed: a wave-level fence keeps the compiler from reordering the slot
    // write and the reads around it.  Two slots alternate so a round's reads never race the next
    // round's write.
#pragma unroll 1
    for (int r = 0; r < 30; r++) {
        x = gl32::add_const_v(x, rc[r * 12 + jj]);
        const gl32::F x7 = gl32::sbox7(x);
        const bool full = (r < 4) || (r >= 26);
        if (full || j == 0) x = x7;
        volatile uint64_t* sl = slot + (r & 1) * 12;
        if (j < 12) sl[j] = gl32::to_u64(x);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // two accumulator pairs shorten the dependent multiply-add chain
        uint64_t al = (uint64_t)x.lo * diag, ah = (uint64_t)x.hi * diag, bl = 0, bh = 0;
#pragma unroll
        for (int i = 0; i < 12; i += 2) {
            uint32_t s0 = jj + i, s1 = jj + i + 1;
            s0 = s0 >= 12 ? s0 - 12 : s0;
            s1 = s1 >= 12 ? s1 - 12 : s1;
            const uint64_t v0 = sl[s0], v1 = sl[s1];
            al += (uint64_t)(uint32_t)v0 * C[i];
            ah += (uint64_t)(uint32_t)(v0 >> 32) * C[i];
            bl += (uint64_t)(uint32_t)v1 * C[i + 1];
            bh += (uint64_t)(uint32_t)(v1 >> 32) * C[i + 1];
        }
        x = gl32::fold_acc(al + bl, ah + bh);
    }
    return x;
}

// K consecutive tree levels in one launch: block b owns the subtree over children [b 2^K, (b + 1) 2^K) of the level at
// `children` (n_children digests, level-major array: the parent levels follow it back to back) and walks it up in LDS,
// one parent per sixteen-lane group and level; every level is also written to its place in the digest array (Merkle
// paths are read from it later).  2^K <= 2 * FUSED_GROUPS.
constexpr unsigned FUSED_GROUPS = 32;  // 512 threads: two waves per SIMD on the first level, one from the second on

__global__ __launch_bounds__(FUSED_GROUPS * 16) void k_merkle_fused(const uint64_t* __restrict__ children, uint64_t* __restrict__ parents0,
                                                                   size_t n_children, unsigned K, size_t tree_words) {
    children += (size_t)blockIdx.y * tree_words;
    parents0 += (size_t)blockIdx.y * tree_words;
    __shared__ uint64_t slots[FUSED_GROUPS * 24];
    __shared__ uint64_t lv[2][FUSED_GROUPS * 2 * 4];
    const uint32_t j = threadIdx.x & 15, g = threadIdx.x >> 4;
    const uint32_t n_own = 1u << K;
    for (uint32_t i = threadIdx.x; i < n_own * 4; i += FUSED_GROUPS * 16) lv[0][i] = children[((size_t)blockIdx.x << K) * 4 + i];
    __syncthreads();
    uint64_t* outp = parents0;
    size_t level_total = n_children;
    for (unsigned s = 1; s <= K; s++) {
        const uint32_t own = n_own >> s;  // this block's parents on the level
        level_total >>= 1;
        const uint64_t* src = lv[(s - 1) & 1];
        uint64_t* dst = lv[s & 1];
        if (g < own) {
            const uint64_t v = j < 8 ? src[g * 8 + j] : 0;
            const gl32::F x = permute_wide(gl32::from_u64(v), j, slots + g * 24);
            if (j < 4) {
                const uint64_t w = gl::canon(gl32::to_u64(x));
                dst[g * 4 + j] = w;
                outp[((size_t)blockIdx.x * own + g) * 4 + j] = w;
            }
        }
        __syncthreads();
        outp += level_total * 4;
    }
}

// FRI layer leaves, one leaf per 16 lanes (see k_fri_leaves for the index maps)
template <int ARITY_BITS>
__global__ __launch_bounds__(256) void k_fri_leaves_wide(const uint64_t* __restrict__ values, unsigned log_n,
                                                         unsigned rate_bits, uint64_t* __restrict__ digests) {
    __shared__ uint64_t lds[WIDE_GROUPS_PER_BLOCK * 24];
    constexpr int ARITY = 1 << ARITY_BITS;
    const unsigned log_np = log_n - ARITY_BITS;
    const uint32_t j = threadIdx.x & 15, g = threadIdx.x >> 4;
    const size_t jp = (size_t)blockIdx.x * WIDE_GROUPS_PER_BLOCK + g;
    const bool live = (jp >> (log_np + rate_bits)) == 0;
    const size_t np = (size_t)1 << log_np, n = (size_t)1 << log_n;
    const uint32_t r = (uint32_t)(jp >> log_np), kp = (uint32_t)(jp & (np - 1));
    gl32::F x = gl32::from_u64(0);
    // absorb 8 words (4 extension elements, slots m0..m0+3) per permutation; lane j < 8 owns word j
#pragma unroll 1
    for (int m0 = 0; m0 < ARITY; m0 += 4) {
        if (j < 8) {
            const int m = m0 + (int)(j >> 1);
            const uint32_t mm = gl::bitrev32((uint32_t)m, ARITY_BITS);
            uint64_t v = 0;
            if (live) v = values[2 * ((size_t)r * n + kp + (size_t)mm * np) + (j & 1)];
            x = gl32::from_u64(v);  // overwrite-mode absorb
        }
        x = permute_wide(x, j, lds + g * 24);
    }
    if (live && j < 4) {
        const size_t leaf = ((size_t)gl::bitrev32(r, rate_bits) << log_np) + gl::bitrev32(kp, log_np);
        digests[leaf * 4 + j] = gl::canon(gl32::to_u64(x));
    }
}

// LDE leaf digests, one leaf per 16 lanes (index maps as in k_hash_lde_leaves): lane j < 8 of a group owns the j-th
// word of each 8-column chunk.
__global__ __launch_bounds__(256) void k_hash_lde_leaves_wide(const uint64_t* __restrict__ lde, size_t col_stride, uint32_t n_cols,
                                                              unsigned log_n, unsigned rate_bits, uint64_t* __restrict__ digests,
                                                              uint32_t batch_cols, size_t tree_words) {
    if (batch_cols) {   // see k_hash_lde_leaves
        const uint32_t c0 = blockIdx.y * batch_cols;
        lde += (size_t)c0 * col_stride;
        n_cols = n_cols - c0 < batch_cols ? n_cols - c0 : batch_cols;
        digests += (size_t)blockIdx.y * tree_words;
    }
    __shared__ uint64_t lds[WIDE_GROUPS_PER_BLOCK * 24];
    const uint32_t j = threadIdx.x & 15, g = threadIdx.x >> 4;
    const size_t pos = (size_t)blockIdx.x * WIDE_GROUPS_PER_BLOCK + g;
    const bool live = (pos >> (log_n + rate_bits)) == 0;
    const uint64_t* p = lde + pos;
    gl32::F x = gl32::from_u64(0);
    if (n_cols <= 4) {   // hash_or_noop: a row of at most four elements IS its digest, zero-padded (a short last batch)
        if (j < n_cols) x = gl32::from_u64(live ? p[(size_t)j * col_stride] : 0);
    } else {
#pragma unroll 1
        for (uint32_t c = 0; c < n_cols; c += 8) {
            if (j < 8 && c + j < n_cols) x = gl32::from_u64(live ? p[(size_t)(c + j) * col_stride] : 0);  // overwrite-mode absorb
            x = permute_wide(x, j, lds + g * 24);
        }
    }
    if (live && j < 4) {
        const uint32_t r = (uint32_t)(pos >> log_n), k = (uint32_t)(pos & (((size_t)1 << log_n) - 1));
        const size_t leaf = ((size_t)gl::bitrev32(r, rate_bits) << log_n) + gl::bitrev32(k, log_n);
        digests[leaf * 4 + j] = gl::canon(gl32::to_u64(x));
    }
}

void launch_hash_lde_leaves_wide(hipStream_t st, const uint64_t* d_lde, size_t col_stride, uint32_t n_cols, unsigned log_n,
                                 unsigned rate_bits, uint64_t* d_digests, uint32_t batch_cols, size_t tree_words) {
    const size_t rows = (size_t)1 << (log_n + rate_bits);
    const uint32_t n_batches = batch_cols ? (n_cols + batch_cols - 1) / batch_cols : 1;
    hipLaunchKernelGGL(k_hash_lde_leaves_wide, dim3((unsigned)((rows + WIDE_GROUPS_PER_BLOCK - 1) / WIDE_GROUPS_PER_BLOCK), n_batches), dim3(256), 0,
                       st, d_lde, col_stride, n_cols, log_n, rate_bits, d_digests, batch_cols, tree_words);
}

void launch_fri_leaves_wide(hipStream_t st, const uint64_t* d_values, unsigned log_n, unsigned rate_bits,
                            unsigned arity_bits, uint64_t* d_digests) {
    const size_t leaves = (size_t)1 << (log_n - arity_bits + rate_bits);
    const unsigned blocks = (unsigned)((leaves + WIDE_GROUPS_PER_BLOCK - 1) / WIDE_GROUPS_PER_BLOCK);
    if (arity_bits == 4) hipLaunchKernelGGL(k_fri_leaves_wide<4>, dim3(blocks), dim3(256), 0, st, d_values, log_n, rate_bits, d_digests);
    else if (arity_bits == 3) hipLaunchKernelGGL(k_fri_leaves_wide<3>, dim3(blocks), dim3(256), 0, st, d_values, log_n, rate_bits, d_digests);
    else if (arity_bits == 2) hipLaunchKernelGGL(k_fri_leaves_wide<2>, dim3(blocks), dim3(256), 0, st, d_values, log_n, rate_bits, d_digests);
}

void launch_merkle_fused(hipStream_t st, const uint64_t* children, size_t n_children, unsigned levels, uint32_t n_trees, size_t tree_words) {
    // children level at `children`, its parents right behind it (level-major digest array)
    hipLaunchKernelGGL(k_merkle_fused, dim3((unsigned)(n_children >> levels), n_trees), dim3(FUSED_GROUPS * 16), 0, st, children,
                       const_cast<uint64_t*>(children) + n_children * 4, n_children, levels, tree_words);
}

}  // namespace nlx
