/* ORACLE - TEST INFRASTRUCTURE ONLY (see gl.h header).
 *
 * Poseidon-12 over Goldilocks, sponge hashing, Merkle tree with cap, Fiat-Shamir challenger.
 * Follows (by module name; source absent from /root/reference):
 *   plonky2::hash::poseidon::{Poseidon::poseidon_naive, constant_layer, sbox_layer, mds_layer,
 *     mds_row_shf}, poseidon_goldilocks::{MDS_MATRIX_CIRC, MDS_MATRIX_DIAG}
 *   plonky2::hash::hashing::{hash_n_to_m_no_pad, compress}, hash_types / HashOut::from_partial
 *   plonky2::hash::merkle_tree::{MerkleTree::new, prove}, merkle_proofs::verify_merkle_proof_to_cap
 *   plonky2::iop::challenger::Challenger
 * Reached from the reference at nearx/src/test_utils.rs:29,62,66 (build / prove / verify).
 */
#include "oracle.h"
#include "poseidon_constants.h"
#include "poseidon_fast_constants.h"
#include <string.h>
#include <stdlib.h>

static const uint64_t RC[360] = NLX_POSEIDON_ROUND_CONSTANTS_INIT;
static const uint64_t MDS_CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
static const uint64_t MDS_DIAG[12] = {8, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

static inline uint64_t sbox7(uint64_t x) {
    uint64_t x2 = gl_sqr(x), x4 = gl_sqr(x2), x3 = gl_mul(x, x2);
    return gl_mul(x3, x4);
}

/* Linear layer.  Same arithmetic as plonky2's mds_layer: each element is split into 32-bit halves,
 * the circulant products accumulate in u64 without overflow (12 * 2^32 * 41 < 2^42) and
 * lo + 2^32 * hi is reduced once.  The doubled arrays make the inner loop contiguous so the
 * compiler can vectorise it (32x32->64 multiplies). */
static void mds_layer(uint64_t s[12]) {
    uint64_t lo2[24], hi2[24], al[12], ah[12];
    for (int i = 0; i < 12; i++) {
        lo2[i] = lo2[i + 12] = s[i] & 0xFFFFFFFFu;
        hi2[i] = hi2[i + 12] = s[i] >> 32;
    }
    for (int r = 0; r < 12; r++) { al[r] = 0; ah[r] = 0; }
    for (int i = 0; i < 12; i++) {
        const uint64_t c = MDS_CIRC[i];
        for (int r = 0; r < 12; r++) {
            al[r] += lo2[i + r] * c;
            ah[r] += hi2[i + r] * c;
        }
    }
    al[0] += lo2[0] * MDS_DIAG[0];
    ah[0] += hi2[0] * MDS_DIAG[0];
    for (int r = 0; r < 12; r++) s[r] = gl_reduce128((u128)al[r] + ((u128)ah[r] << 32));
}

static const uint64_t FAST_FIRST[12] = NLX_POSEIDON_FAST_FIRST_RC_INIT;
static const uint64_t FAST_RC[22] = NLX_POSEIDON_FAST_RC_INIT;
static const uint64_t FAST_VS[22][11] = NLX_POSEIDON_FAST_VS_INIT;
static const uint64_t FAST_W[22][11] = NLX_POSEIDON_FAST_W_HATS_INIT;
static const uint64_t FAST_INIT[11][11] = NLX_POSEIDON_FAST_INITIAL_MATRIX_INIT;

/* sum of <= 12 products of canonical elements, reduced once (each product < 2^128 / 16 is not guaranteed, so the sum is
 * kept as two 128-bit halves: low 64 bits and the rest) */
static inline uint64_t dot_reduce(const uint64_t* a, const uint64_t* b, int n, uint64_t acc0) {
    u128 lo = acc0, hi = 0;
    for (int i = 0; i < n; i++) {
        u128 p = (u128)a[i] * b[i];
        lo += (uint64_t)p;
        hi += (uint64_t)(p >> 64);
    }
    /* value = lo + hi * 2^64 with lo < 2^68, hi < 2^68: fold hi through 2^64 = 2^32 - 1 (mod p) */
    uint64_t r = gl_reduce128(lo);
    uint64_t h = gl_reduce128(hi);
    return gl_add(r, gl_mul(h, GL_EPS));
}

/* plonky2::hash::poseidon::Poseidon::poseidon - the schedule the Rust prover actually runs: 4 full rounds, the 22 partial
 * rounds in their "fast" form (partial_first_constant_layer, mds_partial_layer_init, then per round one S-box, one constant
 * and mds_partial_layer_fast: a 12-term dot product and a rank-one update), 4 full rounds.  Same function as
 * poseidon_naive (tests/test_oracle_golden.py checks fast == naive on the known-answer vectors and random states).
 * Measured here (orc_poseidon_chain, one 2.1 GHz Xeon core): 8.1 us per permutation against 7.7 us for the naive
 * schedule - in scalar C the fast form's 23 full 64 x 64 products per partial round cost what the naive layer's 144
 * small-constant products (which the compiler vectorises) cost; upstream's advantage comes from its AVX2 / NEON kernels. */
void orc_poseidon_permute(uint64_t s[12]) {
    int rc = 0;
    for (int round = 0; round < 4; round++) {
        for (int i = 0; i < 12; i++) s[i] = sbox7(gl_add(s[i], RC[rc++]));
        mds_layer(s);
    }
    for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], FAST_FIRST[i]);
    {
        uint64_t res[12];
        res[0] = s[0];
        for (int c = 1; c < 12; c++) {
            uint64_t col[11];
            for (int r = 1; r < 12; r++) col[r - 1] = FAST_INIT[r - 1][c - 1];
            res[c] = dot_reduce(s + 1, col, 11, 0);
        }
        memcpy(s, res, sizeof res);
    }
    for (int r = 0; r < 22; r++) {
        uint64_t s0 = sbox7(s[0]);
        if (r < 21) s0 = gl_add(s0, FAST_RC[r]);
        const uint64_t d = dot_reduce(s + 1, FAST_W[r], 11, 0);
        for (int i = 1; i < 12; i++) s[i] = gl_add(s[i], gl_mul(s0, FAST_VS[r][i - 1]));
        s[0] = gl_add(d, gl_mul(s0, MDS_CIRC[0] + MDS_DIAG[0]));
    }
    rc = 26 * 12;
    for (int round = 0; round < 4; round++) {
        for (int i = 0; i < 12; i++) s[i] = sbox7(gl_add(s[i], RC[rc++]));
        mds_layer(s);
    }
}

/* Poseidon::poseidon_naive: add all 12 constants, S-box on all lanes (full rounds) or lane 0 (partial rounds), full MDS */
void orc_poseidon_permute_naive(uint64_t s[12]) {
    int rc = 0;
    for (int round = 0; round < 30; round++) {
        for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], RC[rc++]);
        if (round < 4 || round >= 26) {
            for (int i = 0; i < 12; i++) s[i] = sbox7(s[i]);
        } else {
            s[0] = sbox7(s[0]);
        }
        mds_layer(s);
    }
}

/* n chained permutations of one state (timing aid for the cpu_baseline notes; returns a checksum) */
uint64_t orc_poseidon_chain(uint64_t n, int naive) {
    uint64_t s[12] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12};
    for (uint64_t i = 0; i < n; i++) {
        if (naive) orc_poseidon_permute_naive(s);
        else orc_poseidon_permute(s);
    }
    return s[0] ^ s[7];
}

/* hashing.rs hash_n_to_m_no_pad with m = 4: overwrite-mode absorb, rate 8, no padding. */
void orc_hash_no_pad(const uint64_t* in, size_t len, uint64_t out[4]) {
    uint64_t st[12] = {0};
    for (size_t off = 0; off < len; off += 8) {
        size_t k = len - off < 8 ? len - off : 8;
        memcpy(st, in + off, k * 8);
        orc_poseidon_permute(st);
    }
    if (len == 0) { /* no chunk absorbed: squeeze the initial all-zero state */ }
    memcpy(out, st, 32);
}

/* Hasher::hash_or_noop: <= 4 elements are copied (zero padded), otherwise hash_no_pad. */
void orc_hash_or_noop(const uint64_t* in, size_t len, uint64_t out[4]) {
    if (len <= 4) {
        memset(out, 0, 32);
        memcpy(out, in, len * 8);
    } else {
        orc_hash_no_pad(in, len, out);
    }
}

void orc_two_to_one(const uint64_t l[4], const uint64_t r[4], uint64_t out[4]) {
    uint64_t st[12] = {0};
    memcpy(st, l, 32);
    memcpy(st + 4, r, 32);
    orc_poseidon_permute(st);
    memcpy(out, st, 32);
}

/* ---------------- Merkle tree ---------------- */
size_t orc_merkle_digest_words(size_t n_leaves, unsigned cap_height) {
    size_t w = 0, lvl = n_leaves, cap = (size_t)1 << cap_height;
    for (;;) {
        w += lvl * 4;
        if (lvl <= cap) break;
        lvl >>= 1;
    }
    return w;
}

/* Grouped leaves (the STARK side's wide, short traces: stark.h leaf_group_cols): a leaf of more than `group` elements is
 * hashed in two levels - hash_no_pad of every run of `group` elements (the last one may be shorter), then hash_no_pad of
 * the run digests in order - so that the runs of one leaf are independent pieces of work.  group == 0 or a leaf that fits
 * one run: plonky2's hash_or_noop of the whole leaf. */
void orc_leaf_digest(const uint64_t* leaf, size_t leaf_len, size_t group, uint64_t out[4]) {
    if (group == 0 || leaf_len <= group) { orc_hash_or_noop(leaf, leaf_len, out); return; }
    size_t k = (leaf_len + group - 1) / group;
    uint64_t* d = (uint64_t*)malloc(32 * k);
    for (size_t g = 0; g < k; g++) {
        size_t len = leaf_len - g * group < group ? leaf_len - g * group : group;
        orc_hash_no_pad(leaf + g * group, len, d + 4 * g);
    }
    orc_hash_no_pad(d, 4 * k, out);
    free(d);
}

void orc_merkle_build(const uint64_t* leaves, size_t n_leaves, size_t leaf_len, unsigned cap_height,
                      uint64_t* digests_out, uint64_t* cap_out) {
    orc_merkle_build_g(leaves, n_leaves, leaf_len, 0, cap_height, digests_out, cap_out);
}

void orc_merkle_build_g(const uint64_t* leaves, size_t n_leaves, size_t leaf_len, size_t group, unsigned cap_height,
                        uint64_t* digests_out, uint64_t* cap_out) {
    size_t cap = (size_t)1 << cap_height;
    uint64_t* own = NULL;
    if (!digests_out) digests_out = own = (uint64_t*)malloc(orc_merkle_digest_words(n_leaves, cap_height) * 8);
    uint64_t* cur = digests_out;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n_leaves; i++) orc_leaf_digest(leaves + i * leaf_len, leaf_len, group, cur + i * 4);
    size_t lvl = n_leaves;
    while (lvl > cap) {
        uint64_t* nxt = cur + lvl * 4;
        size_t half = lvl >> 1;
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < half; i++) orc_two_to_one(cur + (2 * i) * 4, cur + (2 * i + 1) * 4, nxt + i * 4);
        cur = nxt;
        lvl = half;
    }
    /* n_leaves <= cap: plonky2 asserts cap_height <= log2(n_leaves); the cap is the level itself */
    memcpy(cap_out, cur, lvl * 32);
    free(own);
}

void orc_merkle_prove(const uint64_t* digests, size_t n_leaves, unsigned cap_height, size_t leaf_index,
                      uint64_t* siblings_out) {
    size_t cap = (size_t)1 << cap_height, lvl = n_leaves, idx = leaf_index;
    const uint64_t* cur = digests;
    while (lvl > cap) {
        memcpy(siblings_out, cur + (idx ^ 1) * 4, 32);
        siblings_out += 4;
        cur += lvl * 4;
        lvl >>= 1;
        idx >>= 1;
    }
}

int orc_merkle_verify(const uint64_t* leaf, size_t leaf_len, size_t leaf_index, const uint64_t* siblings,
                      unsigned n_siblings, const uint64_t* cap, unsigned cap_height) {
    return orc_merkle_verify_g(leaf, leaf_len, 0, leaf_index, siblings, n_siblings, cap, cap_height);
}

int orc_merkle_verify_g(const uint64_t* leaf, size_t leaf_len, size_t group, size_t leaf_index, const uint64_t* siblings,
                        unsigned n_siblings, const uint64_t* cap, unsigned cap_height) {
    uint64_t cur[4];
    (void)cap_height;
    orc_leaf_digest(leaf, leaf_len, group, cur);
    size_t idx = leaf_index;
    for (unsigned k = 0; k < n_siblings; k++) {
        uint64_t nxt[4];
        if (idx & 1) orc_two_to_one(siblings + 4 * k, cur, nxt);
        else orc_two_to_one(cur, siblings + 4 * k, nxt);
        memcpy(cur, nxt, 32);
        idx >>= 1;
    }
    return memcmp(cur, cap + idx * 4, 32) == 0;
}

/* ---------------- Challenger ---------------- */
void orc_ch_init(orc_challenger* c) { memset(c, 0, sizeof *c); }

static void ch_duplex(orc_challenger* c) {
    memcpy(c->state, c->in_buf, c->n_in * 8); /* overwrite mode */
    c->n_in = 0;
    orc_poseidon_permute(c->state);
    memcpy(c->out_buf, c->state, 64);
    c->n_out = 8;
}
void orc_ch_observe(orc_challenger* c, uint64_t e) {
    c->n_out = 0; /* any buffered output is now invalid */
    c->in_buf[c->n_in++] = e;
    if (c->n_in == 8) ch_duplex(c);
}
void orc_ch_observe_many(orc_challenger* c, const uint64_t* e, size_t n) {
    for (size_t i = 0; i < n; i++) orc_ch_observe(c, e[i]);
}
uint64_t orc_ch_challenge(orc_challenger* c) {
    if (c->n_in != 0 || c->n_out == 0) ch_duplex(c);
    return c->out_buf[--c->n_out]; /* Vec::pop: from the end */
}
gl2 orc_ch_ext_challenge(orc_challenger* c) {
    uint64_t a = orc_ch_challenge(c);
    uint64_t b = orc_ch_challenge(c);
    return gl2_make(a, b);
}
