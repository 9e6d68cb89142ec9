/* nlx.h - C ABI of the MI355X-native prover backend for the nearx plonky2x circuits.
 *
 * This is the drop-in boundary of SURVEY.md §8(b): the reference has no FFI seam for its
 * prover, so the seam is cut one layer below nearx (nearx/src/test_utils.rs:29,62,66 call
 * CircuitBuilder::build / CircuitBuild::prove / verify; those reach the plonky2 functions
 * named on each entry point below, crates pinned at /root/reference/Cargo.lock:4864-4866,
 * 4912-4974, 4977-4979, 6515-6517).  INTEGRATION.md shows the Rust `extern "C"` stubs a
 * maintainer adds to the patched plonky2 / starkyx / plonky2x crates.
 *
 * Conventions
 *  - Field elements are canonical (< p = 2^64 - 2^32 + 1) little-endian u64.  Inputs that are
 *    not canonical are rejected only where stated; outputs are always canonical.
 *  - Matrices of polynomials are column-major: cols[c * n + i].
 *  - Every data pointer may be a HOST pointer or a HIP DEVICE pointer (e.g. a torch tensor's
 *    data_ptr()); the library inspects it with hipPointerGetAttributes.  Host buffers are
 *    copied over PCIe inside the call; device buffers are used in place.  Output buffers are
 *    caller-owned.  Handles (nlx_ctx, nlx_commit, nlx_circuit) are library-owned and freed
 *    only by their *_destroy call.
 *  - Every call returns 0 on success or a negative NLX_E_* code and never throws or aborts;
 *    nlx_last_error(ctx) describes the most recent failure on that context.
 *  - A context is bound to one HIP device and is NOT thread-safe (one host thread per
 *    context); different contexts are independent.  Calls are synchronous on return.
 *  - Results are deterministic: identical inputs give identical outputs on any device count
 *    or occupancy; the proof-of-work grind returns the SMALLEST valid nonce.
 *  - There is no CPU fallback: if no gfx950 device is usable nlx_ctx_create fails.
 */
#ifndef NLX_H
#define NLX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NLX_OK 0
#define NLX_E_INVAL (-1)
#define NLX_E_NOMEM (-2)
#define NLX_E_HIP (-3)
#define NLX_E_RANGE (-4)
#define NLX_E_UNSUPPORTED (-5)

typedef struct nlx_ctx nlx_ctx;
typedef struct nlx_commit nlx_commit;

/* library / ABI version (major << 16 | minor) */
uint32_t nlx_version(void);
const char* nlx_strerror(int32_t code);

/* ---- context ---- */
int32_t nlx_ctx_create(int device, nlx_ctx** out);
void nlx_ctx_destroy(nlx_ctx* ctx);
const char* nlx_last_error(const nlx_ctx* ctx);
/* Use an existing HIP stream (e.g. torch's current stream) for all work of this context;
 * NULL restores the context's own stream.  The caller keeps ownership of the stream. */
int32_t nlx_ctx_set_stream(nlx_ctx* ctx, void* hip_stream);
int32_t nlx_ctx_synchronize(nlx_ctx* ctx);

/* ---- a5: plonky2::hash::poseidon::Poseidon::poseidon ----
 * states: n x 12 u64, row-major, permuted in place. */
int32_t nlx_poseidon_permute_batch(nlx_ctx* ctx, uint64_t* states, size_t n);

/* ---- a5/a6: PoseidonHash::hash_or_noop over the rows of a row-major matrix ----
 * (the per-leaf step of MerkleTree::new).  digests_out: n_rows x 4. */
int32_t nlx_hash_rows(nlx_ctx* ctx, const uint64_t* rows, size_t n_rows, size_t row_len, uint64_t* digests_out);

/* ---- a6: plonky2::hash::merkle_tree::MerkleTree::new(leaves, cap_height) ----
 * leaves: row-major n_leaves x leaf_len, n_leaves a power of two >= 2^cap_height.
 * digests_out (optional, may be NULL): level-major digests, level 0 = leaf digests
 * (n_leaves x 4), then n_leaves/2, ... down to and including the cap level;
 * nlx_merkle_digest_words() gives its length.  cap_out: 2^cap_height x 4. */
size_t nlx_merkle_digest_words(size_t n_leaves, uint32_t cap_height);
int32_t nlx_merkle_build(nlx_ctx* ctx, const uint64_t* leaves, size_t n_leaves, size_t leaf_len,
                         uint32_t cap_height, uint64_t* digests_out, uint64_t* cap_out);

/* ---- a2: plonky2_field::fft::{fft, ifft}, PolynomialCoeffs::coset_fft, PolynomialValues::coset_ifft ----
 * cols: n_cols x 2^log_n, column-major, transformed in place, natural order in and out.
 * inverse = 0: coefficients -> values on shift*<w>;  inverse = 1: values on shift*<w> -> coefficients.
 * coset_shift = 1 (or 0) means the plain subgroup. */
int32_t nlx_ntt_batch(nlx_ctx* ctx, uint64_t* cols, size_t n_cols, uint32_t log_n, int inverse,
                      uint64_t coset_shift);

/* ---- a3: plonky2::fri::oracle::PolynomialBatch::{from_values, from_coeffs} ----
 * values / coeffs: n_cols x 2^log_n column-major, natural order.  The coset shift is the
 * field's multiplicative generator (plonky2 F::coset_shift()).  blinding / salting is not
 * supported (zero_knowledge = false in the standard recursion config).
 * cap_out: 2^cap_height x 4 words.  *out receives a handle that keeps coefficients, the LDE
 * table and all Merkle digests resident in HBM. */
int32_t nlx_commit_from_values(nlx_ctx* ctx, const uint64_t* values, size_t n_cols, uint32_t log_n,
                               uint32_t rate_bits, uint32_t cap_height, uint64_t* cap_out, nlx_commit** out);
int32_t nlx_commit_from_coeffs(nlx_ctx* ctx, const uint64_t* coeffs, size_t n_cols, uint32_t log_n,
                               uint32_t rate_bits, uint32_t cap_height, uint64_t* cap_out, nlx_commit** out);
void nlx_commit_destroy(nlx_commit* c);

/* PolynomialBatch.polynomials: coefficients, natural order, n_cols x n column-major. */
int32_t nlx_commit_get_coeffs(nlx_commit* c, uint64_t* coeffs_out);
/* MerkleTree.cap */
int32_t nlx_commit_get_cap(nlx_commit* c, uint64_t* cap_out);
/* MerkleTree.leaves[idx[j]] (row of the bit-reversed LDE table, n_cols words) and
 * MerkleTree::prove(idx[j]) (log2(n << rate_bits) - cap_height sibling digests, bottom-up).
 * rows_out: k x n_cols;  paths_out (may be NULL): k x path_len x 4. */
int32_t nlx_commit_open_rows(nlx_commit* c, const uint64_t* idx, size_t k, uint64_t* rows_out,
                             uint64_t* paths_out);
/* a10: PolynomialBatch polynomials evaluated at zeta in the quadratic extension
 * (OpeningSet::new's eval_commitment).  zeta: 2 words; out_ext: n_cols x 2 words. */
int32_t nlx_commit_eval_at(nlx_commit* c, const uint64_t zeta[2], uint64_t* out_ext);
/* whole LDE table in plonky2's leaf order: (n << rate_bits) x n_cols row-major (debug / tests) */
int32_t nlx_commit_get_leaves(nlx_commit* c, uint64_t* leaves_out);
/* level-major digests as in nlx_merkle_build */
int32_t nlx_commit_get_digests(nlx_commit* c, uint64_t* digests_out);

#ifdef __cplusplus
}
#endif
#endif /* NLX_H */
