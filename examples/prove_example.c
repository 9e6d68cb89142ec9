/* Plain-C use of the nlx C ABI (include/nlx.h): build a synthetic nearx-shaped circuit, prove it on
 * GPU 0, prove it again and check the two proofs are byte-identical.  This is what a cgo / Rust FFI
 * caller does, minus the language binding.
 *
 *   gcc -O2 -I include examples/prove_example.c -L near-light-client_amd -lnlx -lnlx_synth \
 *       -Wl,-rpath,$PWD/near-light-client_amd -o /tmp/prove_example && /tmp/prove_example 12
 *   /tmp/prove_example 12 lookups   the same circuit with two plonky2 lookup tables (LookupGate / LookupTableGate rows): the
 *                                   descriptor's table arrays, a witness whose multiplicities the prover fills in
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "nlx.h"
#include "nlx_synth.h"

#define CHECK(call)                                                                        \
    do {                                                                                   \
        int32_t rc__ = (call);                                                             \
        if (rc__ != NLX_OK) {                                                              \
            fprintf(stderr, "%s failed: %d (%s) %s\n", #call, rc__, nlx_strerror(rc__),    \
                    ctx ? nlx_last_error(ctx) : "");                                       \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

int main(int argc, char** argv) {
    const uint32_t log_n = argc > 1 ? (uint32_t)atoi(argv[1]) : 10;
    const size_t n = (size_t)1 << log_n;
    nlx_ctx* ctx = NULL;
    CHECK(nlx_ctx_create(0, &ctx));

    /* Failures are return codes, never aborts (include/nlx.h): a failed HOST allocation inside the library (a C++ exception
     * there, stopped at the ABI), a device allocation and a commitment far beyond the GPU's memory all come back negative,
     * and the context stays usable for the proof below. */
    if (argc > 2 && strcmp(argv[2], "nomem") == 0) {
        nlx_buf* big = NULL;
        nlx_commit* cm = NULL;
        uint64_t one_col[4] = {1, 2, 3, 4}, cap_out[4 * 16];
        const int32_t r0 = nlx_abi_selftest(0);
        const int32_t r1 = nlx_buf_create(ctx, (size_t)1 << 50, &big);
        const int32_t r2 = nlx_commit_from_values(ctx, one_col, 65535, 27, 3, 4, cap_out, &cm);   /* 65 535 columns x 2^27 rows x 9 tables */
        if (r0 != NLX_E_NOMEM || r1 >= 0 || r2 >= 0 || big != NULL || cm != NULL) {
            fprintf(stderr, "expected negative return codes, got %d %d %d\n", r0, r1, r2);
            return 3;
        }
        printf("refused: host allocation %d, device buffer %d (%s), commitment %d\n", r0, r1, nlx_strerror(r1), r2);
    }

    nlx_synth_params sp;
    memset(&sp, 0, sizeof sp);
    sp.log_n = log_n;
    sp.num_public_inputs = 4;
    sp.pct_poseidon = 20; sp.pct_arithmetic = 30; sp.pct_base_sum = 5; sp.pct_constant = 5; sp.pct_extension = 10;
    sp.seed = 42;
    const int lookups = argc > 2 && strcmp(argv[2], "lookups") == 0;
    if (lookups) { sp.num_luts = 2; sp.lut_bits = 8; sp.num_lookups = 300; }   /* two tables of 256 pairs, 300 lookups each */
    const uint32_t n_lk_sel = lookups ? 4 + sp.num_luts : 0;   /* lookup selector columns between gate selectors and gate constants */
    uint32_t n_gates = 0, n_sel = 0;
    nlx_synth_shape(&sp, &n_gates, &n_sel);

    nlx_gate_desc* gates = calloc(n_gates, sizeof *gates);
    uint64_t* k_is = malloc(80 * 8);
    uint64_t* constants = malloc((n_sel + n_lk_sel + 2) * n * 8);
    uint64_t* sigmas = malloc(80 * n * 8);
    uint64_t* wires = malloc(135 * n * 8);
    uint64_t pis[4];
    uint32_t lut_sizes[2] = {256, 256}, lut_num_lookups[2] = {300, 300}, lookup_rows[2 * 3];
    uint16_t* lut_pairs = malloc(2 * 256 * 2 * sizeof(uint16_t));
    if (lookups) CHECK(nlx_synth_circuit_lookups(&sp, gates, k_is, constants, sigmas, wires, pis, lut_pairs, lookup_rows));
    else CHECK(nlx_synth_circuit(&sp, gates, k_is, constants, sigmas, wires, pis));

    nlx_circuit_desc d;
    memset(&d, 0, sizeof d);
    d.degree_bits = log_n; d.num_wires = 135; d.num_routed_wires = 80; d.num_constants = 2; d.num_challenges = 2;
    d.rate_bits = 3; d.cap_height = 4; d.quotient_degree_factor = 8; d.num_partial_products = 9;
    d.fri_pow_bits = 16; d.fri_num_queries = 28; d.fri_arity_bits = 4; d.fri_final_poly_bits = 5;
    d.num_selectors = n_sel; d.num_gates = n_gates; d.num_public_inputs = 4; d.gates = gates; d.k_is = k_is;
    if (lookups) {   /* CommonCircuitData::luts, ProverOnlyCircuitData::{lookup_rows, lut_to_lookups} - a zero tail means no tables */
        d.num_luts = sp.num_luts; d.lut_sizes = lut_sizes; d.lut_pairs = lut_pairs; d.lookup_rows = lookup_rows;
        d.lut_num_lookups = lut_num_lookups;
    }

    nlx_circuit* circuit = NULL;
    CHECK(nlx_circuit_build(ctx, &d, constants, sigmas, &circuit));
    const size_t cap = nlx_proof_max_bytes(circuit);
    uint8_t* p1 = malloc(cap);
    uint8_t* p2 = malloc(cap);
    size_t l1 = 0, l2 = 0;
    CHECK(nlx_prove(circuit, wires, pis, p1, cap, &l1));
    CHECK(nlx_prove(circuit, wires, pis, p2, cap, &l2));
    if (l1 != l2 || memcmp(p1, p2, l1) != 0) {
        fprintf(stderr, "proofs differ between runs\n");
        return 2;
    }
    uint64_t digest[4];
    CHECK(nlx_circuit_digest(circuit, digest));
    printf("ok: 2^%u rows, %u gates, %u selectors, %u lookup tables, proof %zu bytes, circuit digest %016llx...\n", log_n, n_gates,
           n_sel, d.num_luts, l1, (unsigned long long)digest[0]);
    nlx_circuit_destroy(circuit);
    nlx_ctx_destroy(ctx);
    free(gates); free(k_is); free(constants); free(sigmas); free(wires); free(p1); free(p2); free(lut_pairs);
    return 0;
}
