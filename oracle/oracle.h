/* ORACLE - TEST INFRASTRUCTURE ONLY (see gl.h header).  Public surface of the CPU
 * restatement used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 * Nothing in the product (near-light-client_amd/, include/nlx.h) may link or call this. */
#ifndef NLX_ORACLE_H
#define NLX_ORACLE_H
#include "gl.h"

#ifdef __cplusplus
extern "C" {
#endif

/* the generator pair the oracle was built with (include/nlx_field.h): {MULTIPLICATIVE_GROUP_GENERATOR, POWER_OF_TWO_GENERATOR} */
void orc_field_generators(uint64_t out[2]);

/* ---- plonky2::hash::poseidon (Poseidon::poseidon_naive schedule), hashing.rs ---- */
void orc_poseidon_permute(uint64_t state[12]);        /* Poseidon::poseidon (fast partial rounds) */
void orc_poseidon_permute_naive(uint64_t state[12]);  /* Poseidon::poseidon_naive */
uint64_t orc_poseidon_chain(uint64_t n, int naive);
void orc_hash_no_pad(const uint64_t* in, size_t len, uint64_t out[4]);
void orc_hash_or_noop(const uint64_t* in, size_t len, uint64_t out[4]);
void orc_two_to_one(const uint64_t l[4], const uint64_t r[4], uint64_t out[4]);

/* ---- plonky2_field::fft ---- values[i] = sum_j coeffs[j] w^(ij), natural order both sides */
uint64_t orc_eval_poly_base(const uint64_t* coeffs, size_t n, uint64_t x);
void orc_fft(uint64_t* a, unsigned log_n);
void orc_ifft(uint64_t* a, unsigned log_n);
void orc_coset_fft(uint64_t* a, unsigned log_n, uint64_t shift);
void orc_coset_ifft(uint64_t* a, unsigned log_n, uint64_t shift);

/* ---- plonky2::hash::merkle_tree::MerkleTree ----
 * leaves: row-major n_leaves x leaf_len.  digests_out (optional): level-major, level 0 =
 * leaf digests (n_leaves*4 words), then n_leaves/2, ... down to the cap level (inclusive).
 * cap_out: 2^cap_height digests. */
/* leaves of more than `group` elements hashed in two levels (hash.c orc_leaf_digest); group 0 = plonky2's hash_or_noop */
void orc_leaf_digest(const uint64_t* leaf, size_t leaf_len, size_t group, uint64_t out[4]);
void orc_merkle_build_g(const uint64_t* leaves, size_t n_leaves, size_t leaf_len, size_t group, unsigned cap_height,
                        uint64_t* digests_out, uint64_t* cap_out);
int orc_merkle_verify_g(const uint64_t* leaf, size_t leaf_len, size_t group, size_t leaf_index, const uint64_t* siblings,
                        unsigned n_siblings, const uint64_t* cap, unsigned cap_height);
void orc_commit_from_values_g(const uint64_t* values, size_t n_cols, unsigned log_n, unsigned rate_bits,
                              unsigned cap_height, size_t leaf_group, uint64_t* coeffs_out, uint64_t* leaves_out,
                              uint64_t* digests_out, uint64_t* cap_out);
void orc_commit_from_coeffs_g(const uint64_t* coeffs, size_t n_cols, unsigned log_n, unsigned rate_bits,
                              unsigned cap_height, size_t leaf_group, uint64_t* leaves_out, uint64_t* digests_out,
                              uint64_t* cap_out);
void orc_merkle_build(const uint64_t* leaves, size_t n_leaves, size_t leaf_len, unsigned cap_height,
                      uint64_t* digests_out, uint64_t* cap_out);
/* siblings bottom-up: (log2(n_leaves) - cap_height) digests */
void orc_merkle_prove(const uint64_t* digests, size_t n_leaves, unsigned cap_height, size_t leaf_index,
                      uint64_t* siblings_out);
int orc_merkle_verify(const uint64_t* leaf, size_t leaf_len, size_t leaf_index, const uint64_t* siblings,
                      unsigned n_siblings, const uint64_t* cap, unsigned cap_height);

/* ---- plonky2::iop::challenger::Challenger ---- */
typedef struct {
    uint64_t state[12];
    uint64_t in_buf[8];
    unsigned n_in;
    uint64_t out_buf[8];
    unsigned n_out;
} orc_challenger;
void orc_ch_init(orc_challenger* c);
void orc_ch_observe(orc_challenger* c, uint64_t e);
void orc_ch_observe_many(orc_challenger* c, const uint64_t* e, size_t n);
uint64_t orc_ch_challenge(orc_challenger* c);
gl2 orc_ch_ext_challenge(orc_challenger* c);

/* ---- plonky2::fri::oracle::PolynomialBatch ----
 * values/coeffs: column-major cols[c*n + i].  Outputs (all optional except cap_out):
 *   coeffs_out: column-major n_cols x n
 *   leaves_out: row-major (n<<rate_bits) x n_cols, row index bit-reversed (the Merkle leaves)
 *   digests_out: as orc_merkle_build */
void orc_commit_from_values(const uint64_t* values, size_t n_cols, unsigned log_n, unsigned rate_bits,
                            unsigned cap_height, uint64_t* coeffs_out, uint64_t* leaves_out,
                            uint64_t* digests_out, uint64_t* cap_out);
void orc_commit_from_coeffs(const uint64_t* coeffs, size_t n_cols, unsigned log_n, unsigned rate_bits,
                            unsigned cap_height, uint64_t* leaves_out, uint64_t* digests_out,
                            uint64_t* cap_out);

size_t orc_merkle_digest_words(size_t n_leaves, unsigned cap_height);

#ifdef __cplusplus
}
#endif
#endif
