/* ORACLE - TEST INFRASTRUCTURE ONLY (see gl.h header).
 *
 * CPU restatement of the plonky2 prover and verifier for the gate set the synthetic nearx-shaped
 * circuits use.  Follows (by module name; source absent from /root/reference):
 *   plonky2::plonk::prover::{prove_with_partition_witness, wires_permutation_partial_products_and_zs,
 *     compute_quotient_polys}, vanishing_poly::{eval_vanishing_poly, eval_vanishing_poly_base_batch},
 *   plonk_common::{eval_l_0, reduce_with_powers_multi, ZeroPolyOnCoset}, proof::OpeningSet,
 *   gates::{noop, constant, public_input, arithmetic_base, base_sum, poseidon}, gates::selectors,
 *   fri::{oracle::PolynomialBatch::prove_openings, prover::*, verifier::*, reduction_strategies},
 *   util::serialization::Buffer (proof wire format).
 * Reached from the reference at nearx/src/test_utils.rs:62 (prove) and :66 (verify).
 */
#ifndef NLX_ORACLE_PLONK_H
#define NLX_ORACLE_PLONK_H
#include "oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

/* gate kinds (same numbering as include/nlx.h) */
enum {
    ORC_GATE_NOOP = 0,
    ORC_GATE_CONSTANT = 1,     /* param0 = num_consts */
    ORC_GATE_PUBLIC_INPUT = 2,
    ORC_GATE_ARITHMETIC = 3,   /* param0 = num_ops */
    ORC_GATE_BASE_SUM = 4,     /* param0 = base B, param1 = num_limbs */
    ORC_GATE_POSEIDON = 5,
    ORC_GATE_ARITHMETIC_EXT = 6, /* param0 = num_ops (10) */
    ORC_GATE_MUL_EXT = 7,        /* param0 = num_ops (13) */
    ORC_GATE_REDUCING = 8,       /* param0 = num_coeffs (43) */
    ORC_GATE_REDUCING_EXT = 9,   /* param0 = num_coeffs (32) */
    ORC_GATE_POSEIDON_MDS = 10,
    ORC_GATE_EXPONENTIATION = 11, /* param0 = num_power_bits (66) */
    ORC_GATE_RANDOM_ACCESS = 12,  /* param0 = bits, param1 = num_copies | num_extra_constants << 16 */
    ORC_GATE_COSET_INTERPOLATION = 13, /* param0 = subgroup_bits (4), param1 = degree (6) */
    /* plonky2x frontend::num::u32::gates (plonky2-u32): 2-bit limbs throughout */
    ORC_GATE_U32_ADD_MANY = 14,    /* param0 = num_addends, param1 = num_ops */
    ORC_GATE_U32_ARITHMETIC = 15,  /* param0 = num_ops */
    ORC_GATE_U32_SUBTRACTION = 16, /* param0 = num_ops */
    ORC_GATE_U32_RANGE_CHECK = 17, /* param0 = num_input_limbs */
    ORC_GATE_COMPARISON = 18,      /* param0 = num_bits, param1 = num_chunks */
    /* plonky2::gates::{lookup, lookup_table}: no gate constraints of their own - their wires feed the lookup argument
     * (vanishing_poly::check_lookup_constraints).  LookupGate: slot i = (looking_inp 2i, looking_out 2i+1), routed/2 slots;
     * LookupTableGate: slot i = (looked_inp 3i, looked_out 3i+1, multiplicity 3i+2), routed/3 slots. */
    ORC_GATE_LOOKUP = 19,
    ORC_GATE_LOOKUP_TABLE = 20
};

typedef struct {
    uint32_t kind;
    uint32_t selector_index; /* which selector polynomial this gate uses */
    uint32_t group_start;    /* gate-index range [start, end) sharing that selector */
    uint32_t group_end;
    uint32_t index;          /* this gate's index in the sorted gate list (value of the selector on its rows) */
    uint32_t param0, param1;
} orc_gate;

typedef struct {
    uint32_t degree_bits;
    uint32_t num_wires;            /* 135 */
    uint32_t num_routed_wires;     /* 80 */
    uint32_t num_constants;        /* gate constants per row (2), excluding selectors */
    uint32_t num_challenges;       /* 2 */
    uint32_t rate_bits;            /* 3 */
    uint32_t cap_height;           /* 4 */
    uint32_t quotient_degree_factor; /* 8 */
    uint32_t num_partial_products; /* 9 */
    uint32_t fri_pow_bits;         /* 16 */
    uint32_t fri_num_queries;      /* 28 */
    uint32_t fri_arity_bits;       /* 4 */
    uint32_t fri_final_poly_bits;  /* 5 */
    uint32_t num_selectors;
    uint32_t num_gates;
    uint32_t num_public_inputs;
    const orc_gate* gates;
    const uint64_t* k_is;          /* num_routed_wires coset shifts */
    uint64_t circuit_digest[4];
    /* ---- lookup tables (CircuitBuilder::add_all_lookups; CommonCircuitData::luts, ProverOnlyCircuitData::lookup_rows) ----
     * num_luts == 0: no lookup argument, everything below is ignored and the proof is what it was without these fields.
     * With tables the constants matrix holds, between the gate selectors and the gate constants, the 4 + num_luts lookup
     * selector columns (selectors::selectors_lookup: TransSre, TransLdc, InitSre, LastLdc; selector_ends_lookups: one per
     * table), the Zs commitment carries num_challenges * (1 + S) more columns (RE and the S = ceil(routed/2 / (qdf-1))
     * partial sums of every challenge) after the partial products, and 2 * num_challenges more challenges are drawn. */
    uint32_t num_luts;
    uint32_t pad_;
    const uint32_t* lut_sizes;       /* num_luts: entries per table */
    const uint16_t* lut_pairs;       /* the tables' (input, output) pairs, table after table, 2 u16 per entry */
    const uint32_t* lookup_rows;     /* 3 per table: last_lu_row, last_lut_row, first_lut_row (LookupWire) */
    const uint32_t* lut_num_lookups; /* num_luts: lookups made into each table (lut_to_lookups[i].len()); the rest of the
                                        last LookupGate row is padded with the table's first entry by the prover */
} orc_circuit_desc;

typedef struct orc_circuit orc_circuit;

/* CircuitBuilder::build's prover-side products: commits constants (selectors first, then gate
 * constants) and sigma polynomials.  constants: (num_selectors+num_constants) x n column-major,
 * sigmas: num_routed_wires x n column-major.  If desc->circuit_digest is all zero it is computed
 * as hash_no_pad(cap || hash_pad([]) || degree_bits) and stored. */
orc_circuit* orc_circuit_build(const orc_circuit_desc* desc, const uint64_t* constants, const uint64_t* sigmas);
void orc_circuit_free(orc_circuit* c);
void orc_circuit_digest(const orc_circuit* c, uint64_t out[4]);
void orc_circuit_constants_sigmas_cap(const orc_circuit* c, uint64_t* cap_out);

/* prover::set_lookup_wires on a host witness (num_wires x n column-major, in place): the multiplicity wires of every
 * LookupTableGate row and the padding slots of each table's last LookupGate row.  orc_prove applies it to its own copy of
 * the witness, as prove_with_partition_witness does to the PartitionWitness it is handed.  0 = ok, -1 = a looked-up input
 * is not in its table. */
int orc_set_lookup_wires(const orc_circuit* c, uint64_t* wires);

/* upper bound of the serialized proof size in bytes */
size_t orc_proof_max_bytes(const orc_circuit* c);
/* prove_with_partition_witness + ProofWithPublicInputs::to_bytes.  wires: num_wires x n column-major.
 * Returns the number of bytes written, 0 on failure (e.g. unsatisfied witness makes the quotient
 * exceed its degree bound). */
size_t orc_prove(const orc_circuit* c, const uint64_t* wires, const uint64_t* public_inputs, uint8_t* proof_out,
                 size_t cap_bytes);
/* verify(): 1 = accept, <= 0 = reject (negative values name the failing check) */
int orc_verify(const orc_circuit* c, const uint8_t* proof, size_t len);

/* stage-level access for stage-by-stage parity tests */
typedef struct {
    uint64_t betas[4], gammas[4], alphas[4];
    uint64_t zeta[2];
    uint64_t fri_alpha[2];
    uint64_t fri_betas[2 * 16];
    uint64_t pow_witness;
    uint32_t n_fri_rounds;
    uint64_t query_indices[128];
    uint64_t deltas[16];           /* lookup challenges, 4 per challenge round (A, B, alpha, delta); zero without tables */
    /* optional dumps (caller-allocated or NULL) */
    uint64_t* zs_partial_values;   /* (num_challenges*(1+num_partial_products) + num_challenges*(1+S) lookup columns) x n column-major */
    uint64_t* quotient_chunk_coeffs; /* (num_challenges*quotient_degree_factor) x n column-major */
    uint64_t* fri_final_values;    /* L ext values (2 words each), natural LDE order */
} orc_trace;
size_t orc_prove_traced(const orc_circuit* c, const uint64_t* wires, const uint64_t* public_inputs,
                        uint8_t* proof_out, size_t cap_bytes, orc_trace* trace);

/* hash_pad (pad10*1 to a multiple of the sponge width) */
void orc_hash_pad(const uint64_t* in, size_t len, uint64_t out[4]);

/* Poseidon fast-partial-round constants derived from the MDS matrix and round constants
 * (hadeshash calc_equivalent_constants / calc_equivalent_matrices); used by the PoseidonGate. */
typedef struct {
    uint64_t first_round_constant[12];
    uint64_t round_constants[22];       /* last entry unused (0) */
    uint64_t vs[22][11];
    uint64_t w_hats[22][11];
    uint64_t initial_matrix[11][11];
} orc_poseidon_fast;
const orc_poseidon_fast* orc_poseidon_fast_constants(void);
void orc_poseidon_permute_fast(uint64_t state[12]);

#ifdef __cplusplus
}
#endif
#endif
