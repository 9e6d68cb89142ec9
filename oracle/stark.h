/* ORACLE - TEST INFRASTRUCTURE ONLY (see gl.h header).
 *
 * CPU restatement of the starky STARK prover / verifier (plonky2's `starky` crate: prover::prove,
 * compute_quotient_polys, constraint_consumer::ConstraintConsumer, proof::StarkOpeningSet,
 * stark::Stark::fri_instance, verifier::verify_stark_proof, config::StarkConfig::standard_fast_config)
 * - the public ancestor of the reference's un-vendored `starkyx` / curta prover (Cargo.lock:6515-6517,
 * SURVEY.md §8a row a12), whose own source and AIRs (SHA-256, Ed25519) are absent.  The AIR is DATA here:
 * a small register program interpreted identically by this oracle (base field and extension field)
 * and by the HIP kernel, so any AIR can be plugged in without touching the prover.
 *
 * Parity status: "parity unpinned" - no STARK proof bytes exist in the reference; internal consistency
 * (this verifier accepts this prover, rejects tampering) is the pin.
 */
#ifndef NLX_ORACLE_STARK_H
#define NLX_ORACLE_STARK_H
#include "oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- generic FRI over a list of committed batches (plonky2::fri::{oracle::prove_openings, prover, verifier}) ---- */
typedef struct {
    const uint64_t* coeffs;  /* n_cols x n column-major */
    const uint64_t* leaves;  /* L x n_cols, plonky2 leaf order */
    const uint64_t* digests; /* level-major */
    const uint64_t* cap;
    uint32_t n_cols;
} orc_fri_oracle;
typedef struct {
    gl2 point;
    uint32_t n_polys;
    const uint32_t* oracle; /* oracle index of each polynomial */
    const uint32_t* poly;   /* column inside that oracle */
} orc_fri_batch;
typedef struct {
    uint32_t degree_bits, rate_bits, cap_height, pow_bits, n_queries, arity_bits, final_poly_bits;
    uint32_t leaf_group;   /* the oracles' leaves are hashed in runs of this many columns (0: whole rows; hash.c orc_leaf_digest) */
} orc_fri_params;

/* AIR program: 64-bit words, op | dst << 8 | a << 24 | b << 40 (16-bit fields); OP_CONST is followed by
 * one immediate word.  Registers r0..r63.  Same numbering as include/nlx.h (NLX_AIR_*). */
enum {
    ORC_AIR_LOCAL = 0,      /* dst = local_values[a] */
    ORC_AIR_NEXT = 1,       /* dst = next_values[a] */
    ORC_AIR_PUBLIC = 2,     /* dst = public_inputs[a] */
    ORC_AIR_CONST = 3,      /* dst = immediate */
    ORC_AIR_ADD = 4,
    ORC_AIR_SUB = 5,
    ORC_AIR_MUL = 6,
    ORC_AIR_EMIT_TRANSITION = 7, /* constraint_transition(r[a]) : times (x - g^-1) */
    ORC_AIR_EMIT_FIRST = 8,      /* constraint_first_row(r[a]) : times L_0(x) */
    ORC_AIR_EMIT_LAST = 9,       /* constraint_last_row(r[a])  : times L_{n-1}(x) */
    ORC_AIR_EMIT = 10,           /* constraint(r[a]) on every row */
    ORC_AIR_PERIODIC = 11,       /* dst = periodic column a at this row: values[a][row mod 2^period_bits] */
    ORC_AIR_PACK_LOCAL = 12,     /* dst = sum_{i<b} 2^i local_values[a+i]   (b <= 32) */
    ORC_AIR_PACK_NEXT = 13,      /* dst = sum_{i<b} 2^i next_values[a+i] */
    ORC_AIR_EMIT_BOOL = 14,      /* constraint(x * (x - 1)) for x = local_values[a .. a + max(b, 1)), in column order */
    ORC_AIR_LOADV = 15,          /* scheduling hint: the next `dst` words are independent loads (no semantics) */
    /* three-operand forms for bit-valued columns; the third register is in bits 56..61 */
    ORC_AIR_XOR3 = 16,           /* dst = a ^ b ^ c as a polynomial: s = a + b - 2ab, dst = s + c - 2sc */
    ORC_AIR_CH = 17,             /* dst = c + a (b - c)            (choose: a ? b : c) */
    ORC_AIR_MAJ = 18,            /* dst = ab + c (a + b - 2ab)     (majority) */
    ORC_AIR_SEGMENT = 19,        /* no register is carried across this word (registers are cleared here) */
    ORC_AIR_EMIT_LOGUP = 20,     /* the two constraints of a LogUp helper: v2 col in dst (0xFFFF: none), v1 col in a,
                                    h cols b, b+1, challenge index in bits 56..61 */
    ORC_AIR_MAC = 21             /* dst = r[c] + r[a] * r[b], c in bits 56..61 */
};
/* ADD / SUB carry a shift in bits 56..61 of the word: dst = r[a] +- r[b] * 2^shift. */

typedef struct {
    uint32_t degree_bits;
    uint32_t n_cols;
    uint32_t num_challenges;          /* 2 */
    uint32_t rate_bits;               /* 1 */
    uint32_t cap_height;              /* 4 */
    uint32_t quotient_degree_factor;  /* power of two <= 2^rate_bits (max(1, constraint_degree - 1) rounded up) */
    uint32_t fri_pow_bits;            /* 16 */
    uint32_t fri_num_queries;         /* 84 */
    uint32_t fri_arity_bits;          /* 4 */
    uint32_t fri_final_poly_bits;     /* 5 */
    uint32_t num_public_inputs;
    uint32_t n_words;                 /* program length in words */
    const uint64_t* program;
    /* periodic (verifier-computable) columns: column a takes values[a * period + (row mod period)]; as a
     * polynomial it is P_a(x^(n/period)), P_a = interpolation of the values over the period-th roots of unity */
    uint32_t n_periodic;
    uint32_t period_bits;
    const uint64_t* periodic;
    /* rounds of commitment (starkyx's TraceWriter rounds).  0 = classic single-round starky.  Round r commits
     * round_cols[r] columns (sum = n_cols, program column indices run through the rounds in order); after its cap is
     * observed the verifier draws round_challenges[r] base-field challenges, which the program reads as PUBLIC
     * indices num_public_inputs + k in the order drawn. */
    uint32_t n_rounds;
    uint32_t round_cols[3];
    uint32_t round_challenges[3];
    /* Grouped leaves: 0 = every Merkle leaf is plonky2's hash_or_noop of the whole LDE row.  G > 0: a row of more than G
     * columns is hashed as hash_no_pad(hash_no_pad(cols [0, G)) || hash_no_pad(cols [G, 2G)) || ...) - a tree whose bottom
     * level has arity ceil(n_cols / G) over column runs.  Same proof bytes layout (rows and sibling paths), same number of
     * permutations + ceil(K / 2) for the verifier; for the prover the K runs of a leaf are independent work, which is what
     * a trace of thousands of columns on a few thousand rows needs to fill a GPU.  Part of the
     * statement digest when non-zero. */
    uint32_t leaf_group_cols;
    /* round values: round_values[r] field elements the prover sends with round r (bus / accumulator totals that depend on
     * earlier challenges).  They are observed after the round's cap and before its challenges are drawn, are written
     * after the public inputs at the end of the proof, and the program reads them as PUBLIC: the values array is
     * public inputs | values of round 0 | challenges of round 0 | values of round 1 | challenges of round 1 | ... */
    uint32_t round_values[3];
    /* Openings digest: 0 = the transcript observes every opened value (starky's observe_openings: local ++ quotient at zeta,
     * then the next values).  G > 0: it observes FOUR elements instead - that vector, zero-padded to a multiple of G, hashed in
     * runs of G (hash_no_pad each) and the run digests hashed once more.  Same binding, and the long sequential absorption (2 400
     * permutations on one host thread for a 4 745-column trace: 3 of the proof's 8 ms) becomes parallel work.  In the statement
     * digest when non-zero. */
    uint32_t openings_group;
    /* Batches: 0 = a round's columns are one PolynomialBatch.  B > 0: a round of more than B columns is committed as
     * ceil(cols / B) PolynomialBatches of B columns (the last one what is left), in column order - each exactly
     * PolynomialBatch::from_values (hash_or_noop leaves over its own row, its own cap, its own FRI oracle); caps are observed and
     * written in batch order, every query opens each batch's row with its own path; openings and their order do not change.
     * In the statement digest when non-zero.  (include/nlx.h nlx_stark_desc.batch_cols) */
    uint32_t batch_cols;
} orc_stark_desc;
#define ORC_STARK_MAX_ORACLES 31

/* returns round `round`'s columns (round_cols[round] x n, column-major) given everything after the public inputs in the
 * values array so far (`n_known` elements), and writes the round's round_values[round] values to values_out */
typedef const uint64_t* (*orc_round_fn)(void* user, uint32_t round, const uint64_t* known, uint32_t n_known, uint64_t* values_out);
size_t orc_stark_prove_rounds(const orc_stark_desc* d, orc_round_fn fn, void* user, const uint64_t* public_inputs,
                              uint8_t* proof_out, size_t cap_bytes);
size_t orc_stark_proof_max_bytes(const orc_stark_desc* d);
/* trace: n_cols x n column-major.  Returns bytes written (0 on overflow). */
size_t orc_stark_prove(const orc_stark_desc* d, const uint64_t* trace, const uint64_t* public_inputs,
                       uint8_t* proof_out, size_t cap_bytes);
/* the statement digest observed first in the transcript (config, program, periodic columns, round structure) */
void orc_stark_air_digest(const orc_stark_desc* d, uint64_t out[4]);
uint32_t orc_stark_values(const orc_stark_desc* d, const uint8_t* proof, size_t len, uint64_t* out);
int orc_stark_verify(const orc_stark_desc* d, const uint8_t* proof, size_t len);

#ifdef __cplusplus
}
#endif
#endif
