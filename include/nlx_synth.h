/* nlx_synth.h - workload generation for tests, examples and bench.py: satisfiable synthetic circuits / AIR witnesses of the
 * nearx circuits' static shape.  INPUTS ONLY - nothing here is on the prover path; it lives in its own
 * library (libnlx_synth.so, plain host C++) so that libnlx.so holds the product alone.
 */
#ifndef NLX_SYNTH_H
#define NLX_SYNTH_H
#include "nlx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- synthetic nearx-shaped workload (inputs only) ---- */
typedef struct {
    uint32_t log_n;
    uint32_t num_public_inputs;
    uint32_t pct_poseidon;    /* share of rows (percent) that are PoseidonGate rows */
    uint32_t pct_arithmetic;
    uint32_t pct_base_sum;
    uint32_t pct_constant;    /* remaining rows are NoopGate */
    uint64_t seed;
    uint32_t pct_extension;   /* rows split evenly over ArithmeticExtension / MulExtension / Reducing / ReducingExtension */
    uint32_t pct_misc;        /* rows split evenly over PoseidonMds / Exponentiation / CosetInterpolation / RandomAccess */
    uint32_t pct_u32;         /* rows split evenly over U32AddMany / U32Arithmetic / U32Subtraction / U32RangeCheck / Comparison */
    uint32_t wide_comparison; /* 1: ComparisonGate { num_bits 25, num_chunks 25 } (132 constraints, the widest gate) instead of { 32, 16 } */
    /* lookup tables (plonky2 CircuitBuilder::add_lookup_table_from_pairs / add_lookup_from_index): num_luts tables of
     * 2^lut_bits (input, output) u16 pairs each, num_lookups lookups into every one of them; 0 tables = none of this */
    uint32_t num_luts;        /* 0 .. 8 */
    uint32_t lut_bits;        /* 1 .. 16 */
    uint32_t num_lookups;     /* per table, >= 1 */
    uint32_t pad_;
} nlx_synth_params;
/* number of gates / selector polynomials the generator will emit for these parameters */
void nlx_synth_shape(const nlx_synth_params* sp, uint32_t* n_gates, uint32_t* n_selectors);
/* Fills host buffers: gates[n_gates], k_is[80], constants[(n_selectors+2) x n], sigmas[80 x n],
 * wires[135 x n], public_inputs[num_public_inputs].  The witness satisfies every constraint. */
int32_t nlx_synth_circuit(const nlx_synth_params* sp, nlx_gate_desc* gates, uint64_t* k_is, uint64_t* constants,
                          uint64_t* sigmas, uint64_t* wires, uint64_t* public_inputs);
/* The same with lookup tables (sp->num_luts > 0): constants is (n_selectors + 4 + num_luts + 2) x n (the lookup selector
 * columns between the gate selectors and the gate constants), gates[] starts with num_luts LookupGate and num_luts
 * LookupTableGate entries, lut_pairs[num_luts << lut_bits][2] and lookup_rows[num_luts][3] are the descriptor's arrays
 * (include/nlx.h nlx_circuit_desc).  The witness is the PartitionWitness BEFORE prove: multiplicity wires and the padding
 * slots of each table's last LookupGate row are zero - the prover's set_lookup_wires step fills them. */
int32_t nlx_synth_circuit_lookups(const nlx_synth_params* sp, nlx_gate_desc* gates, uint64_t* k_is, uint64_t* constants,
                                  uint64_t* sigmas, uint64_t* wires, uint64_t* public_inputs, uint16_t* lut_pairs,
                                  uint32_t* lookup_rows);
/* Re-target a generated witness (host buffer, 135 x n column-major) to other public inputs: rewrites the
 * PublicInputGate row so the witness stays satisfying.  Used by the map-reduce workload, where a reduce
 * job's public inputs are its children's digests. */
int32_t nlx_synth_set_public_inputs(uint64_t* wires, uint32_t log_n, const uint64_t* public_inputs, uint32_t count);
/* Synthetic wide-AIR witness (inputs only): n_cols (multiple of 4) x n column-major host buffer, k1 = the
 * n_cols/4 per-group constants of the AIR, public_inputs[2] = first-row values of columns 0 and 1. */
int32_t nlx_synth_stark_trace(uint32_t n_cols, uint32_t log_n, uint64_t seed, const uint64_t* k1, uint64_t* trace,
                              uint64_t* public_inputs);

#ifdef __cplusplus
}
#endif
#endif /* NLX_SYNTH_H */
