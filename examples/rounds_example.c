/* A plain-C caller of the multi-round STARK entry point (include/nlx.h: nlx_stark_prove_rounds): round 0 commits a
 * column v; the prover draws a challenge gamma; round 1 commits the Horner accumulator acc(i) = acc(i-1) gamma + v(i)
 * and SENDS its last value as a round value - the fingerprint of v under gamma, which whoever relies on the proof
 * recomputes.  The round callback is where a caller computes its challenge-dependent columns (lookup / bus
 * accumulators).  Prints a line the test compares with the Python path.
 *   gcc -std=c11 -I include examples/rounds_example.c -L near-light-client_amd -lnlx -o rounds_example */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "nlx.h"

#define P 0xFFFFFFFF00000001ULL
#define CHECK(x)                                                                        \
    do {                                                                                \
        int32_t rc__ = (x);                                                             \
        if (rc__) {                                                                     \
            fprintf(stderr, "%s failed: %d %s\n", #x, rc__, ctx ? nlx_last_error(ctx) : ""); \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

static uint64_t W(uint64_t op, uint64_t dst, uint64_t a, uint64_t b) { return op | dst << 8 | a << 24 | b << 40; }
static uint64_t mulmod(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) % P); }
static uint64_t addmod(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a + b) % P); }

struct rounds {
    size_t n;
    uint64_t* v;    /* round 0: one column */
    uint64_t* acc;  /* round 1: one column */
    uint64_t total;
};

/* nlx_round_fn: `known` = everything after the public inputs in the values array so far (here: gamma, once round 0 is in) */
static const uint64_t* round_fn(void* user, uint32_t round, const uint64_t* known, uint32_t n_known, uint64_t* values_out) {
    struct rounds* r = user;
    if (round == 0) return r->v;
    if (n_known != 1 || !values_out) return NULL;
    uint64_t acc = 0;
    for (size_t i = 0; i < r->n; i++) {
        acc = addmod(mulmod(acc, known[0]), r->v[i]);
        r->acc[i] = acc;
    }
    values_out[0] = r->total = acc;
    return r->acc;
}

int main(int argc, char** argv) {
    const uint32_t log_n = argc > 1 ? (uint32_t)atoi(argv[1]) : 10;
    const size_t n = (size_t)1 << log_n;
    nlx_ctx* ctx = NULL;
    CHECK(nlx_ctx_create(0, &ctx));

    /* columns (v, acc); values array after the (zero) public inputs: [gamma, total]
     *   first row: acc = v;  transitions: acc' = acc gamma + v';  last row: acc = total */
    const uint64_t prog[] = {
        W(NLX_AIR_LOCAL, 0, 1, 0), W(NLX_AIR_LOCAL, 1, 0, 0), W(NLX_AIR_SUB, 2, 0, 1), W(NLX_AIR_EMIT_FIRST, 0, 2, 0),
        W(NLX_AIR_LOCAL, 0, 1, 0), W(NLX_AIR_PUBLIC, 1, 0, 0), W(NLX_AIR_MUL, 2, 0, 1), W(NLX_AIR_NEXT, 3, 0, 0),
        W(NLX_AIR_ADD, 4, 2, 3), W(NLX_AIR_NEXT, 5, 1, 0), W(NLX_AIR_SUB, 6, 5, 4), W(NLX_AIR_EMIT_TRANSITION, 0, 6, 0),
        W(NLX_AIR_LOCAL, 0, 1, 0), W(NLX_AIR_PUBLIC, 1, 1, 0), W(NLX_AIR_SUB, 2, 0, 1), W(NLX_AIR_EMIT_LAST, 0, 2, 0),
    };
    nlx_stark_desc d;
    memset(&d, 0, sizeof d);
    d.degree_bits = log_n; d.n_cols = 2; d.num_challenges = 2; d.rate_bits = 1; d.cap_height = 4;
    d.quotient_degree_factor = 1; d.fri_pow_bits = 16; d.fri_num_queries = 84; d.fri_arity_bits = 4; d.fri_final_poly_bits = 5;
    d.num_public_inputs = 0; d.n_words = (uint32_t)(sizeof prog / sizeof prog[0]); d.program = prog;
    d.n_rounds = 2;
    d.round_cols[0] = 1; d.round_challenges[0] = 1;  /* v, then gamma */
    d.round_cols[1] = 1; d.round_values[1] = 1;      /* acc, sent with its total */
    nlx_stark* stark = NULL;
    CHECK(nlx_stark_build(ctx, &d, &stark));

    struct rounds r = {n, malloc(n * 8), malloc(n * 8), 0};
    uint64_t x = 88172645463325252ULL;
    for (size_t i = 0; i < n; i++) {  /* xorshift64: the test regenerates the same column */
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        r.v[i] = x % P;
    }
    const size_t cap = nlx_stark_proof_max_bytes(stark);
    uint8_t* proof = malloc(cap);
    size_t len = 0;
    CHECK(nlx_stark_prove_rounds(stark, round_fn, &r, NULL, proof, cap, &len));
    uint64_t fold = 0;
    for (size_t i = 0; i + 8 <= len; i += 8) {
        uint64_t w;
        memcpy(&w, proof + i, 8);
        fold = (fold * 0x100000001B3ULL) ^ w;
    }
    printf("program:");  /* part of the statement (the transcript opens with its digest): a second prover needs these words */
    for (uint32_t i = 0; i < d.n_words; i++) printf(" %llx", (unsigned long long)prog[i]);
    printf("\n");
    printf("ok: 2^%u rows, proof %zu bytes, fold %016llx, round value %llu\n", log_n, len, (unsigned long long)fold,
           (unsigned long long)r.total);
    nlx_stark_destroy(stark);
    nlx_ctx_destroy(ctx);
    free(r.v); free(r.acc); free(proof);
    return 0;
}
