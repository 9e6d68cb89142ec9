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

void orc_poseidon_permute(uint64_t s[12]) {
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

void orc_merkle_build(const uint64_t* leaves, size_t n_leaves, size_t leaf_len, unsigned cap_height,
                      uint64_t* digests_out, uint64_t* cap_out) {
    size_t cap = (size_t)1 << cap_height;
    uint64_t* own = NULL;
    if (!digests_out) digests_out = own = (uint64_t*)malloc(orc_merkle_digest_words(n_leaves, cap_height) * 8);
    uint64_t* cur = digests_out;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n_leaves; i++) orc_hash_or_noop(leaves + i * leaf_len, leaf_len, cur + i * 4);
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
    uint64_t cur[4];
    (void)cap_height;
    orc_hash_or_noop(leaf, leaf_len, cur);
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
