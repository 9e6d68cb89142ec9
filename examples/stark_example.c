/* A plain-C caller of the STARK entry points (include/nlx.h): starky's FibonacciStark written directly as an AIR
 * register program, its trace in an nlx_buf, one proof.  Prints a line the test compares with the Python path.
 *   gcc -std=c11 -I include examples/stark_example.c -L near-light-client_amd -lnlx -o stark_example */
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

int main(int argc, char** argv) {
    const uint32_t log_n = argc > 1 ? (uint32_t)atoi(argv[1]) : 10;
    const size_t n = (size_t)1 << log_n;
    nlx_ctx* ctx = NULL;
    CHECK(nlx_ctx_create(0, &ctx));

    /* columns (x0, x1); public inputs (x0[0], x1[0], x1[n-1]):
     *   first row: x0 = pi0, x1 = pi1;  last row: x1 = pi2;  transitions: x0' = x1, x1' = x0 + x1 */
    const uint64_t prog[] = {
        W(NLX_AIR_LOCAL, 0, 0, 0), W(NLX_AIR_PUBLIC, 1, 0, 0), W(NLX_AIR_SUB, 2, 0, 1), W(NLX_AIR_EMIT_FIRST, 0, 2, 0),
        W(NLX_AIR_LOCAL, 0, 1, 0), W(NLX_AIR_PUBLIC, 1, 1, 0), W(NLX_AIR_SUB, 2, 0, 1), W(NLX_AIR_EMIT_FIRST, 0, 2, 0),
        W(NLX_AIR_LOCAL, 0, 1, 0), W(NLX_AIR_PUBLIC, 1, 2, 0), W(NLX_AIR_SUB, 2, 0, 1), W(NLX_AIR_EMIT_LAST, 0, 2, 0),
        W(NLX_AIR_NEXT, 0, 0, 0), W(NLX_AIR_LOCAL, 1, 1, 0), W(NLX_AIR_SUB, 2, 0, 1), W(NLX_AIR_EMIT_TRANSITION, 0, 2, 0),
        W(NLX_AIR_NEXT, 0, 1, 0), W(NLX_AIR_LOCAL, 1, 0, 0), W(NLX_AIR_SUB, 2, 0, 1), W(NLX_AIR_LOCAL, 1, 1, 0),
        W(NLX_AIR_SUB, 3, 2, 1), W(NLX_AIR_EMIT_TRANSITION, 0, 3, 0),
    };
    nlx_stark_desc d;
    memset(&d, 0, sizeof d);
    d.degree_bits = log_n; d.n_cols = 2; d.num_challenges = 2; d.rate_bits = 1; d.cap_height = 4;
    d.quotient_degree_factor = 1; d.fri_pow_bits = 16; d.fri_num_queries = 84; d.fri_arity_bits = 4; d.fri_final_poly_bits = 5;
    d.num_public_inputs = 3; d.n_words = (uint32_t)(sizeof prog / sizeof prog[0]); d.program = prog;
    nlx_stark* stark = NULL;
    CHECK(nlx_stark_build(ctx, &d, &stark));

    uint64_t* trace = malloc(2 * n * 8);
    uint64_t a = 3, b = 5;
    for (size_t i = 0; i < n; i++) {
        trace[i] = a;
        trace[n + i] = b;
        const uint64_t s = (uint64_t)(((unsigned __int128)a + b) % P);
        a = b;
        b = s;
    }
    const uint64_t pis[3] = {trace[0], trace[n], trace[2 * n - 1]};
    nlx_buf* buf = NULL;
    CHECK(nlx_buf_create(ctx, 2 * n * 8, &buf));
    CHECK(nlx_buf_upload(buf, 0, trace, 2 * n * 8));

    const size_t cap = nlx_stark_proof_max_bytes(stark);
    uint8_t* proof = malloc(cap);
    uint8_t* proof2 = malloc(cap);
    size_t len = 0, len2 = 0;
    CHECK(nlx_stark_prove(stark, (const uint64_t*)nlx_buf_device_ptr(buf), pis, proof, cap, &len));  /* device-resident trace */
    CHECK(nlx_stark_prove(stark, trace, pis, proof2, cap, &len2));                                   /* host trace */
    if (len != len2 || memcmp(proof, proof2, len) != 0) {
        fprintf(stderr, "host and device traces gave different proofs\n");
        return 2;
    }
    uint64_t fold = 0;
    for (size_t i = 0; i + 8 <= len; i += 8) {
        uint64_t w;
        memcpy(&w, proof + i, 8);
        fold = fold * 0x100000001B3ULL ^ w;
    }
    printf("program:");  /* the program is part of the statement (the transcript opens with its digest): a second prover needs these words */
    for (uint32_t i = 0; i < d.n_words; i++) printf(" %llx", (unsigned long long)prog[i]);
    printf("\nok: fibonacci stark 2^%u rows, proof %zu bytes, fold %016llx\n", log_n, len, (unsigned long long)fold);
    nlx_buf_destroy(buf);
    nlx_stark_destroy(stark);
    nlx_ctx_destroy(ctx);
    free(trace); free(proof); free(proof2);
    return 0;
}
