// Device-resident polynomial batch commitment (plonky2::fri::oracle::PolynomialBatch).
#pragma once
#include "ctx.hpp"

struct nlx_commit {
    nlx_ctx* ctx = nullptr;
    uint32_t n_cols = 0;
    uint32_t log_n = 0;
    uint32_t rate_bits = 0;
    uint32_t cap_height = 0;
    uint64_t* coeffs_br = nullptr;  // [col][n], coefficient i at position bitrev(i)
    uint64_t* lde = nullptr;        // [col][r][k] = p_col(g * w_L^(8k + r)), L = n << rate_bits
    uint64_t* digests = nullptr;    // level-major; level 0 in plonky2 leaf order
    const uint64_t* cap = nullptr;  // inside digests
    uint64_t* group_digests = nullptr;  // grouped leaves only: the runs' digests, [4 K][L]
    // Batches (STARK commitment rounds, nlx_stark_desc.batch_cols): the columns are n_trees PolynomialBatches of batch_cols columns
    // (the last one what is left), transformed together but each with its OWN Merkle tree (hash_or_noop leaves over its
    // columns) and cap: tree k's level-major digests start tree_words * k words into `digests`.  n_trees = 1: one batch.
    uint32_t n_trees = 1, batch_cols = 0;
    size_t tree_words = 0;
    bool owner = true;                  // false: a view of one batch (commit_view), nothing to free
    size_t n() const { return (size_t)1 << log_n; }
    size_t L() const { return (size_t)1 << (log_n + rate_bits); }
    unsigned log_L() const { return log_n + rate_bits; }
};

namespace nlx {
size_t merkle_digest_words(size_t n_leaves, uint32_t cap_height);
// Input kinds for commit_build
enum class CommitInput { ValuesNatural, CoeffsNatural, CoeffsBitrev };
// d_in: device pointer, [col][n] with column stride in_stride.  Enqueues all work on ctx->stream;
// no synchronisation.  On success *out owns coeffs_br / lde / digests.
// leaf_group: 0 = plonky2 leaves (hash_or_noop of the whole LDE row); G > 0 and n_cols > G: grouped leaves (launch.hpp)
// batch_cols: 0 = one batch; B > 0 and n_cols > B: ceil(n_cols / B) batches with a tree each (see nlx_commit)
int32_t commit_build(nlx_ctx* ctx, const uint64_t* d_in, size_t in_stride, CommitInput kind, uint32_t n_cols,
                     uint32_t log_n, uint32_t rate_bits, uint32_t cap_height, nlx_commit** out, uint32_t leaf_group = 0,
                     uint32_t batch_cols = 0);
// batch k of a commitment as a commitment of its own (non-owning): its columns of the shared tables, its tree, its cap
nlx_commit commit_view(const nlx_commit* c, uint32_t k);
}  // namespace nlx
