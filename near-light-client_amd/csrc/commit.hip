// PolynomialBatch commitment pipeline on device + the primitive C-ABI entry points.
//
// Replaces plonky2::fri::oracle::PolynomialBatch::{from_values, from_coeffs, lde_values},
// MerkleTree::{new, prove} and the fft / hash helpers they call.
// Pipeline per batch (all on one stream, columns stay resident in HBM):
//   values --DIF iNTT--> coeffs(bit-reversed) --scale + DIT x 2^rate cosets--> LDE table
//   --Poseidon leaf hash (scatter to tree order)--> digests --levels--> cap
#include "commit.hpp"
#include "gl.hpp"
#include "poly.hpp"

namespace nlx {

size_t merkle_digest_words(size_t n_leaves, uint32_t cap_height) {
    size_t w = 0, lvl = n_leaves, cap = (size_t)1 << cap_height;
    for (;;) {
        w += lvl * 4;
        if (lvl <= cap) break;
        lvl >>= 1;
    }
    return w;
}

nlx_commit commit_view(const nlx_commit* c, uint32_t k) {
    nlx_commit v = *c;
    v.owner = false;
    v.n_trees = 1;
    v.group_digests = nullptr;
    if (c->n_trees > 1) {
        const uint32_t c0 = k * c->batch_cols;
        v.n_cols = c->n_cols - c0 < c->batch_cols ? c->n_cols - c0 : c->batch_cols;
        v.coeffs_br = c->coeffs_br + (size_t)c0 * c->n();
        v.lde = c->lde + (size_t)c0 * c->L();
        v.digests = c->digests + (size_t)k * c->tree_words;
        v.cap = c->cap + (size_t)k * c->tree_words;
        v.batch_cols = 0;
        v.tree_words = 0;
    }
    return v;
}

int32_t commit_build(nlx_ctx* ctx, const uint64_t* d_in, size_t in_stride, CommitInput kind, uint32_t n_cols,
                     uint32_t log_n, uint32_t rate_bits, uint32_t cap_height, nlx_commit** out, uint32_t leaf_group,
                     uint32_t batch_cols) {
    *out = nullptr;
    if (n_cols == 0 || n_cols > 65535) return ctx->fail(NLX_E_RANGE, "n_cols %u out of range [1, 65535]", n_cols);
    if (log_n + rate_bits > 32) return ctx->fail(NLX_E_RANGE, "log_n + rate_bits > 32");
    if (cap_height > log_n + rate_bits) return ctx->fail(NLX_E_RANGE, "cap_height exceeds tree height");
    int32_t rc = ctx->ensure_tables(log_n);
    if (rc) return rc;
    const uint64_t* scale = nullptr;
    rc = ctx->get_coset_scale(log_n, rate_bits, &scale);
    if (rc) return rc;

    nlx_commit* c = new (std::nothrow) nlx_commit();
    if (!c) return ctx->fail(NLX_E_NOMEM, "host allocation failed");
    c->ctx = ctx;
    c->n_cols = n_cols;
    c->log_n = log_n;
    c->rate_bits = rate_bits;
    c->cap_height = cap_height;
    const size_t n = c->n(), L = c->L();
    c->coeffs_br = (uint64_t*)ctx->alloc((size_t)n_cols * n * 8);
    c->lde = (uint64_t*)ctx->alloc((size_t)n_cols * L * 8);
    const bool batched = batch_cols && n_cols > batch_cols && !(leaf_group && n_cols > leaf_group);
    if (batched) {
        c->n_trees = (n_cols + batch_cols - 1) / batch_cols;
        c->batch_cols = batch_cols;
        c->tree_words = merkle_digest_words(L, cap_height);
    }
    c->digests = (uint64_t*)ctx->alloc(merkle_digest_words(L, cap_height) * 8 * c->n_trees);
    const bool grouped = leaf_group && n_cols > leaf_group;
    const uint32_t n_groups = grouped ? (n_cols + leaf_group - 1) / leaf_group : 0;
    if (grouped) c->group_digests = (uint64_t*)ctx->alloc((size_t)4 * n_groups * L * 8);
    if (!c->coeffs_br || !c->lde || !c->digests || (grouped && !c->group_digests)) {
        ctx->release(c->coeffs_br);
        ctx->release(c->lde);
        ctx->release(c->digests);
        ctx->release(c->group_digests);
        delete c;
        return NLX_E_NOMEM;
    }
    hipStream_t st = ctx->stream;
    switch (kind) {
        case CommitInput::ValuesNatural:
            ctx->begin_kernel("intt", 16.0 * n * n_cols);
            launch_intt_dif(st, ctx->tables, d_in, in_stride, c->coeffs_br, n, n_cols, log_n);
            ctx->end_kernel();
            break;
        case CommitInput::CoeffsNatural:
            if (in_stride == n) {
                launch_bitrev_permute(st, d_in, c->coeffs_br, n, n_cols, log_n, nullptr);
            } else {
                for (uint32_t col = 0; col < n_cols; col++)
                    launch_bitrev_permute(st, d_in + (size_t)col * in_stride, c->coeffs_br + (size_t)col * n, n, 1,
                                          log_n, nullptr);
            }
            break;
        case CommitInput::CoeffsBitrev:
            if (in_stride == n) {
                (void)hipMemcpyAsync(c->coeffs_br, d_in, (size_t)n_cols * n * 8, hipMemcpyDeviceToDevice, st);
            } else {
                (void)hipMemcpy2DAsync(c->coeffs_br, n * 8, d_in, in_stride * 8, n * 8, n_cols,
                                       hipMemcpyDeviceToDevice, st);
            }
            break;
    }
    // algorithmic bytes per SURVEY.md §8(d): LDE 8n + 8L per column; leaf hash 8cL + 32L; tree 64L
    ctx->begin_kernel("lde", (8.0 * n + 8.0 * L) * n_cols);
    launch_lde_dit(st, ctx->tables, c->coeffs_br, n, c->lde, L, n_cols, log_n, rate_bits, scale);
    ctx->end_kernel();
    // units: Poseidon permutations (hash_or_noop: none for rows of <= 4 elements, else one per 8 absorbed)
    if (grouped) {
        // permutations: every run's ceil(len / 8), then ceil(4 K / 8) per leaf over the runs' digests
        const uint32_t last = n_cols - (n_groups - 1) * leaf_group;
        const double perms = (double)L * ((double)(n_groups - 1) * ((leaf_group + 7) / 8) + (last + 7) / 8 + (4 * n_groups + 7) / 8);
        ctx->begin_kernel("hash_lde_leaves", 8.0 * n_cols * L + 32.0 * L, perms);
        launch_hash_lde_leaves_grouped(st, c->lde, L, n_cols, leaf_group, log_n, rate_bits, c->group_digests, c->digests);
        ctx->end_kernel();
    } else if (batched) {
        const uint32_t last = n_cols - (c->n_trees - 1) * batch_cols;
        const double perms = (double)L * ((double)(c->n_trees - 1) * ((batch_cols + 7) / 8) + (last <= 4 ? 0 : (last + 7) / 8));
        ctx->begin_kernel("hash_lde_leaves", 8.0 * n_cols * L + 32.0 * L * c->n_trees, perms);
        launch_hash_lde_leaves(st, c->lde, L, n_cols, log_n, rate_bits, c->digests, batch_cols, c->tree_words);
        ctx->end_kernel();
    } else {
        ctx->begin_kernel("hash_lde_leaves", 8.0 * n_cols * L + 32.0 * L, n_cols <= 4 ? 0.0 : (double)L * ((n_cols + 7) / 8));
        launch_hash_lde_leaves(st, c->lde, L, n_cols, log_n, rate_bits, c->digests);
        ctx->end_kernel();
    }
    ctx->begin_kernel("merkle_levels", 64.0 * L * c->n_trees, ((double)L - (double)((size_t)1 << cap_height)) * c->n_trees);
    c->cap = launch_merkle_levels(st, c->digests, L, cap_height, c->n_trees, c->tree_words);
    ctx->end_kernel();
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        ctx->release(c->coeffs_br);
        ctx->release(c->lde);
        ctx->release(c->digests);
        ctx->release(c->group_digests);
        delete c;
        return ctx->hip_fail(e, "commit_build launch");
    }
    *out = c;
    return NLX_OK;
}

static int32_t copy_out(nlx_ctx* ctx, void* user, const void* dev, size_t bytes) {
    hipError_t e = hipMemcpyAsync(user, dev, bytes, is_device_ptr(user) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                                  ctx->stream);
    if (e != hipSuccess) return ctx->hip_fail(e, "hipMemcpyAsync(out)");
    return NLX_OK;
}

}  // namespace nlx

using namespace nlx;

extern "C" {

int32_t nlx_field_ops(nlx_ctx* ctx, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out) NLX_TRY {
    if (!ctx) return NLX_E_INVAL;
    if (n == 0) return NLX_OK;
    if (!a || !b || !out) return ctx->fail(NLX_E_INVAL, "NULL buffer");
    (void)hipSetDevice(ctx->device);
    Staged sa(ctx, a, n * 8, true, false), sb(ctx, b, n * 8, true, false), so(ctx, out, 5 * n * 8, false, true);
    if (sa.status) return sa.status;
    if (sb.status) return sb.status;
    if (so.status) return so.status;
    launch_field_ops(ctx->stream, sa.as<uint64_t>(), sb.as<uint64_t>(), n, so.as<uint64_t>());
    int32_t rc = so.finish();
    if (rc) return rc;
    NLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return NLX_OK;
} NLX_CATCH(ctx)

int32_t nlx_poseidon_permute_batch(nlx_ctx* ctx, uint64_t* states, size_t n) NLX_TRY {
    if (!ctx) return NLX_E_INVAL;
    if (n == 0) return NLX_OK;
    if (!states) return ctx->fail(NLX_E_INVAL, "states is NULL");
    (void)hipSetDevice(ctx->device);
    Staged s(ctx, states, n * 12 * 8, true, true);
    if (s.status) return s.status;
    launch_permute_batch(ctx->stream, s.as<uint64_t>(), n);
    int32_t rc = s.finish();
    if (rc) return rc;
    NLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return NLX_OK;
} NLX_CATCH(ctx)

int32_t nlx_hash_rows(nlx_ctx* ctx, const uint64_t* rows, size_t n_rows, size_t row_len, uint64_t* digests_out) NLX_TRY {
    if (!ctx) return NLX_E_INVAL;
    if (n_rows == 0) return NLX_OK;
    if (!digests_out || (!rows && row_len)) return ctx->fail(NLX_E_INVAL, "NULL buffer");
    if (row_len > 0xFFFFFFFFull) return ctx->fail(NLX_E_RANGE, "row_len too large");
    (void)hipSetDevice(ctx->device);
    Staged in(ctx, rows, n_rows * row_len * 8, true, false);
    Staged out(ctx, digests_out, n_rows * 32, false, true);
    if (in.status) return in.status;
    if (out.status) return out.status;
    launch_hash_leaves_rowmajor(ctx->stream, in.as<uint64_t>(), (uint32_t)row_len, n_rows, out.as<uint64_t>());
    int32_t rc = out.finish();
    if (rc) return rc;
    NLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return NLX_OK;
} NLX_CATCH(ctx)

size_t nlx_merkle_digest_words(size_t n_leaves, uint32_t cap_height) NLX_TRY {
    if (n_leaves == 0 || (n_leaves & (n_leaves - 1)) || cap_height > 63) return 0;
    return merkle_digest_words(n_leaves, cap_height);
} NLX_CATCH_VALUE(nullptr, 0)

int32_t nlx_merkle_build(nlx_ctx* ctx, const uint64_t* leaves, size_t n_leaves, size_t leaf_len, uint32_t cap_height,
                         uint64_t* digests_out, uint64_t* cap_out) NLX_TRY {
    if (!ctx) return NLX_E_INVAL;
    if (n_leaves == 0 || (n_leaves & (n_leaves - 1))) return ctx->fail(NLX_E_INVAL, "n_leaves must be a power of two");
    if (cap_height > 63 || ((size_t)1 << cap_height) > n_leaves)
        return ctx->fail(NLX_E_RANGE, "cap_height %u exceeds log2(n_leaves)", cap_height);
    if (!cap_out || (!leaves && leaf_len)) return ctx->fail(NLX_E_INVAL, "NULL buffer");
    if (leaf_len > 0xFFFFFFFFull) return ctx->fail(NLX_E_RANGE, "leaf_len too large");
    (void)hipSetDevice(ctx->device);
    size_t words = merkle_digest_words(n_leaves, cap_height);
    Staged in(ctx, leaves, n_leaves * leaf_len * 8, true, false);
    if (in.status) return in.status;
    uint64_t* d_dig = nullptr;
    bool own_dig = false;
    if (digests_out && is_device_ptr(digests_out)) {
        d_dig = digests_out;
    } else {
        d_dig = (uint64_t*)ctx->alloc(words * 8);
        if (!d_dig) return NLX_E_NOMEM;
        own_dig = true;
    }
    launch_hash_leaves_rowmajor(ctx->stream, in.as<uint64_t>(), (uint32_t)leaf_len, n_leaves, d_dig);
    const uint64_t* d_cap = launch_merkle_levels(ctx->stream, d_dig, n_leaves, cap_height);
    int32_t rc = copy_out(ctx, cap_out, d_cap, ((size_t)32) << cap_height);
    if (!rc && own_dig && digests_out) rc = copy_out(ctx, digests_out, d_dig, words * 8);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (own_dig) ctx->release(d_dig);
    if (rc) return rc;
    if (e != hipSuccess) return ctx->hip_fail(e, "hipStreamSynchronize");
    return NLX_OK;
} NLX_CATCH(ctx)

int32_t nlx_ntt_batch(nlx_ctx* ctx, uint64_t* cols, size_t n_cols, uint32_t log_n, int inverse, uint64_t coset_shift) NLX_TRY {
    if (!ctx) return NLX_E_INVAL;
    if (n_cols == 0) return NLX_OK;
    if (!cols) return ctx->fail(NLX_E_INVAL, "cols is NULL");
    if (log_n > 32) return ctx->fail(NLX_E_RANGE, "log_n > 32");
    if (n_cols > 65535) return ctx->fail(NLX_E_RANGE, "n_cols > 65535");
    if (coset_shift >= gl::P) return ctx->fail(NLX_E_INVAL, "coset_shift not canonical");
    (void)hipSetDevice(ctx->device);
    int32_t rc = ctx->ensure_tables(log_n);
    if (rc) return rc;
    const size_t n = (size_t)1 << log_n;
    const bool coset = coset_shift > 1;
    const uint64_t* scale = nullptr;
    if (coset) {
        rc = ctx->get_nat_scale(log_n, inverse ? gl::inv(coset_shift) : coset_shift, &scale);
        if (rc) return rc;
    }
    Staged s(ctx, cols, n_cols * n * 8, true, true);
    if (s.status) return s.status;
    // 2^18 .. 2^28 points: natural -> natural in three (four) passes, the last one writing every value at its natural position
    // through a second buffer (launch_ntt_dif_natural, ntt_kernels.hip) - no reordering pass.  NLX_NTT_REORDER=1 keeps round
    // 3's path (DIF + k_bitrev_tiled) for comparison.
    hipError_t e = hipSuccess;
    {
        const char* keep = getenv("NLX_NTT_REORDER");
        uint64_t* tmp = (log_n >= 18 && log_n <= 28 && !(keep && keep[0] == '1')) ? (uint64_t*)ctx->alloc(n_cols * n * 8) : nullptr;
        if (tmp) {
            ctx->begin_kernel("ntt_transform", 16.0 * n * n_cols);
            const bool ok = launch_ntt_dif_natural(ctx->stream, ctx->tables, s.as<uint64_t>(), tmp, n, (uint32_t)n_cols, log_n, inverse != 0,
                                                   inverse ? nullptr : scale, inverse ? scale : nullptr);
            ctx->end_kernel();
            ctx->release(tmp);   // stream-ordered: the block is handed out again only to work enqueued after these kernels
            if (ok) {
                rc = s.finish();
                if (rc) return rc;
                NLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
                return NLX_OK;
            }
        }
    }
    // natural -> (DIF) -> bit-reversed -> permute back to natural.  Forward: pre-scale by
    // shift^i; inverse: post-scale by shift^-i (and 1/n inside the transform).
    // algorithmic bytes: 16 n per column for the transform (read + write once); the reordering back to
    // natural order is a second read + write that the prover's own pipeline never pays (DIF in, DIT out)
    ctx->begin_kernel("ntt_transform", 16.0 * n * n_cols);
    launch_ntt_dif_fwd(ctx->stream, ctx->tables, s.as<uint64_t>(), n, (uint32_t)n_cols, log_n, inverse != 0,
                       inverse ? nullptr : scale);
    ctx->end_kernel();
    ctx->begin_kernel("ntt_reorder", 16.0 * n * n_cols);
    if (!launch_bitrev_inplace(ctx->stream, s.as<uint64_t>(), n, (uint32_t)n_cols, log_n, inverse ? scale : nullptr)) {
        uint64_t* tmp = (uint64_t*)ctx->alloc(n_cols * n * 8);   // small transforms: gather into a second buffer, copy back
        if (!tmp) { ctx->end_kernel(); return NLX_E_NOMEM; }
        launch_bitrev_permute(ctx->stream, s.as<uint64_t>(), tmp, n, (uint32_t)n_cols, log_n, inverse ? scale : nullptr);
        e = hipMemcpyAsync(s.dev, tmp, n_cols * n * 8, hipMemcpyDeviceToDevice, ctx->stream);
        ctx->release(tmp);
    }
    ctx->end_kernel();
    if (e != hipSuccess) return ctx->hip_fail(e, "hipMemcpyAsync");
    rc = s.finish();
    if (rc) return rc;
    NLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return NLX_OK;
} NLX_CATCH(ctx)

int32_t nlx_ntt_split_level(nlx_ctx* ctx, uint64_t* mine, const uint64_t* theirs, size_t n_cols, uint32_t log_n, uint32_t world_log,
                            uint32_t rank, uint32_t level) NLX_TRY {
    if (!ctx) return NLX_E_INVAL;
    if (n_cols == 0) return NLX_OK;
    if (!mine || !theirs) return ctx->fail(NLX_E_INVAL, "NULL slice");
    if (log_n > 32 || world_log == 0 || world_log > 6 || world_log >= log_n || level >= world_log || (rank >> world_log) || n_cols > 65535)
        return ctx->fail(NLX_E_RANGE, "log_n %u, 2^%u ranks, rank %u, level %u: out of range", log_n, world_log, rank, level);
    if (!is_device_ptr(mine) || !is_device_ptr(theirs)) return ctx->fail(NLX_E_INVAL, "the slices must be device memory");
    (void)hipSetDevice(ctx->device);
    int32_t rc = ctx->ensure_tables(log_n);
    if (rc) return rc;
    const size_t m = (size_t)1 << (log_n - world_log);
    const uint32_t bit = world_log - 1 - level;                 // the bit of the rank this level pairs over
    const bool upper = (rank >> bit) & 1;
    const size_t idx0 = (size_t)(rank & ((1u << bit) - 1)) * m;   // offset of the slice inside the half-block of n / 2^(level + 1)
    launch_ntt_split_level(ctx->stream, mine, theirs, m, (uint32_t)n_cols, upper, ctx->tables.fwd[log_n], idx0, level);
    NLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return ctx->hip_fail(le, "kernel launch");
    return NLX_OK;
} NLX_CATCH(ctx)

static int32_t commit_api(nlx_ctx* ctx, const uint64_t* data, size_t n_cols, uint32_t log_n, uint32_t rate_bits,
                          uint32_t cap_height, uint64_t* cap_out, nlx_commit** out, CommitInput kind) {
    if (!ctx) return NLX_E_INVAL;
    if (!out) return ctx->fail(NLX_E_INVAL, "out is NULL");
    *out = nullptr;
    if (!data) return ctx->fail(NLX_E_INVAL, "input is NULL");
    if (n_cols == 0 || n_cols > 65535) return ctx->fail(NLX_E_RANGE, "n_cols out of range");
    if (log_n > 32) return ctx->fail(NLX_E_RANGE, "log_n > 32");
    (void)hipSetDevice(ctx->device);
    const size_t n = (size_t)1 << log_n;
    Staged in(ctx, data, n_cols * n * 8, true, false);
    if (in.status) return in.status;
    nlx_commit* c = nullptr;
    int32_t rc = commit_build(ctx, in.as<uint64_t>(), n, kind, (uint32_t)n_cols, log_n, rate_bits, cap_height, &c);
    if (rc) return rc;
    if (cap_out) rc = copy_out(ctx, cap_out, c->cap, ((size_t)32) << cap_height);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (!rc && e != hipSuccess) rc = ctx->hip_fail(e, "hipStreamSynchronize");
    if (rc) {
        nlx_commit_destroy(c);
        return rc;
    }
    *out = c;
    return NLX_OK;
}

int32_t nlx_commit_from_values(nlx_ctx* ctx, const uint64_t* values, size_t n_cols, uint32_t log_n,
                               uint32_t rate_bits, uint32_t cap_height, uint64_t* cap_out, nlx_commit** out) NLX_TRY {
    return commit_api(ctx, values, n_cols, log_n, rate_bits, cap_height, cap_out, out, CommitInput::ValuesNatural);
} NLX_CATCH(ctx)

int32_t nlx_commit_from_coeffs(nlx_ctx* ctx, const uint64_t* coeffs, size_t n_cols, uint32_t log_n,
                               uint32_t rate_bits, uint32_t cap_height, uint64_t* cap_out, nlx_commit** out) NLX_TRY {
    return commit_api(ctx, coeffs, n_cols, log_n, rate_bits, cap_height, cap_out, out, CommitInput::CoeffsNatural);
} NLX_CATCH(ctx)

void nlx_commit_destroy(nlx_commit* c) NLX_TRY {
    if (!c) return;
    c->ctx->release(c->coeffs_br);
    c->ctx->release(c->lde);
    c->ctx->release(c->digests);
    c->ctx->release(c->group_digests);
    delete c;
} NLX_CATCH_VOID(nullptr)

int32_t nlx_commit_get_coeffs(nlx_commit* c, uint64_t* coeffs_out) NLX_TRY {
    if (!c || !coeffs_out) return NLX_E_INVAL;
    nlx_ctx* ctx = c->ctx;
    (void)hipSetDevice(ctx->device);
    size_t bytes = (size_t)c->n_cols * c->n() * 8;
    Staged out(ctx, coeffs_out, bytes, false, true);
    if (out.status) return out.status;
    launch_bitrev_permute(ctx->stream, c->coeffs_br, out.as<uint64_t>(), c->n(), c->n_cols, c->log_n, nullptr);
    int32_t rc = out.finish();
    if (rc) return rc;
    NLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return NLX_OK;
} NLX_CATCH(nullptr)

int32_t nlx_commit_get_cap(nlx_commit* c, uint64_t* cap_out) NLX_TRY {
    if (!c || !cap_out) return NLX_E_INVAL;
    nlx_ctx* ctx = c->ctx;
    (void)hipSetDevice(ctx->device);
    int32_t rc = copy_out(ctx, cap_out, c->cap, ((size_t)32) << c->cap_height);
    if (rc) return rc;
    NLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return NLX_OK;
} NLX_CATCH(nullptr)

int32_t nlx_commit_open_rows(nlx_commit* c, const uint64_t* idx, size_t k, uint64_t* rows_out, uint64_t* paths_out) NLX_TRY {
    if (!c) return NLX_E_INVAL;
    nlx_ctx* ctx = c->ctx;
    if (k == 0) return NLX_OK;
    if (!idx || !rows_out) return ctx->fail(NLX_E_INVAL, "NULL buffer");
    (void)hipSetDevice(ctx->device);
    // validate indices on the host when they are host-resident
    if (!is_device_ptr(idx)) {
        for (size_t i = 0; i < k; i++)
            if (idx[i] >= c->L()) return ctx->fail(NLX_E_RANGE, "leaf index %llu out of range", (unsigned long long)idx[i]);
    }
    const unsigned path_len = c->log_L() - c->cap_height;
    Staged sidx(ctx, idx, k * 8, true, false);
    Staged srows(ctx, rows_out, k * c->n_cols * 8, false, true);
    Staged spaths(ctx, paths_out, k * path_len * 32, false, true);
    if (sidx.status) return sidx.status;
    if (srows.status) return srows.status;
    if (spaths.status) return spaths.status;
    launch_gather_rows(ctx->stream, c->lde, c->L(), c->n_cols, c->log_n, c->rate_bits, sidx.as<uint64_t>(), k,
                       srows.as<uint64_t>());
    if (paths_out)
        launch_gather_paths(ctx->stream, c->digests, c->log_L(), c->cap_height, sidx.as<uint64_t>(), k,
                            spaths.as<uint64_t>());
    int32_t rc = srows.finish();
    if (!rc && paths_out) rc = spaths.finish();
    if (rc) return rc;
    NLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return NLX_OK;
} NLX_CATCH(nullptr)

int32_t nlx_commit_eval_at(nlx_commit* c, const uint64_t zeta[2], uint64_t* out_ext) NLX_TRY {
    if (!c) return NLX_E_INVAL;
    nlx_ctx* ctx = c->ctx;
    if (!zeta || !out_ext) return ctx->fail(NLX_E_INVAL, "NULL buffer");
    (void)hipSetDevice(ctx->device);
    Staged sz(ctx, zeta, 16, true, false);
    Staged so(ctx, out_ext, (size_t)c->n_cols * 16, false, true);
    if (sz.status) return sz.status;
    if (so.status) return so.status;
    uint64_t* scratch = (uint64_t*)ctx->alloc(eval_scratch_words(c->n_cols, c->log_n) * 8);
    if (!scratch) return NLX_E_NOMEM;
    launch_eval_br(ctx->stream, c->coeffs_br, c->n(), c->n_cols, c->log_n, sz.as<uint64_t>(), so.as<uint64_t>(), scratch);
    ctx->release(scratch);
    int32_t rc = so.finish();
    if (rc) return rc;
    NLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return NLX_OK;
} NLX_CATCH(nullptr)

int32_t nlx_commit_get_leaves(nlx_commit* c, uint64_t* leaves_out) NLX_TRY {
    if (!c || !leaves_out) return NLX_E_INVAL;
    nlx_ctx* ctx = c->ctx;
    (void)hipSetDevice(ctx->device);
    Staged out(ctx, leaves_out, c->L() * c->n_cols * 8, false, true);
    if (out.status) return out.status;
    launch_table_to_leaves(ctx->stream, c->lde, c->L(), c->n_cols, c->log_n, c->rate_bits, out.as<uint64_t>());
    int32_t rc = out.finish();
    if (rc) return rc;
    NLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return NLX_OK;
} NLX_CATCH(nullptr)

int32_t nlx_commit_get_digests(nlx_commit* c, uint64_t* digests_out) NLX_TRY {
    if (!c || !digests_out) return NLX_E_INVAL;
    nlx_ctx* ctx = c->ctx;
    (void)hipSetDevice(ctx->device);
    int32_t rc = copy_out(ctx, digests_out, c->digests, merkle_digest_words(c->L(), c->cap_height) * 8);
    if (rc) return rc;
    NLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return NLX_OK;
} NLX_CATCH(nullptr)

}  // extern "C"
