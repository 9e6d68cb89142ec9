/* ORACLE - TEST INFRASTRUCTURE ONLY (see gl.h and plonk.h headers).
 *
 * plonky2 prover + verifier restatement (standard_recursion_config shape).  Every step follows
 * SURVEY.md §3.4 and the upstream functions named in plonk.h; comments cite the upstream
 * function each block restates.  Written for clarity and literal fidelity (row-major
 * bit-reversed leaves, natural-order coefficients, coefficient-domain FRI folding), NOT in the
 * layout the HIP path uses - agreement between the two is the parity test.
 */
#include "plonk.h"
#include "poseidon_constants.h"
#include "poseidon_fast_constants.h"
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

static const uint64_t ORC_RC[360] = NLX_POSEIDON_ROUND_CONSTANTS_INIT;
static const uint64_t ORC_MDS_CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
static const uint64_t ORC_MDS_DIAG[12] = {8, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
static const orc_poseidon_fast FAST = {
    NLX_POSEIDON_FAST_FIRST_RC_INIT, NLX_POSEIDON_FAST_RC_INIT, NLX_POSEIDON_FAST_VS_INIT,
    NLX_POSEIDON_FAST_W_HATS_INIT, NLX_POSEIDON_FAST_INITIAL_MATRIX_INIT};
const orc_poseidon_fast* orc_poseidon_fast_constants(void) { return &FAST; }

#define UNUSED_SELECTOR 0xFFFFFFFFULL

/* slot counts of the lookup argument (LookupGate::num_slots = routed / 2, LookupTableGate::num_slots = routed / 3;
 * prover::compute_lookup_polys: S = ceil(lu slots / (max_quotient_degree_factor - 1)) partial sums per row) */
typedef struct {
    uint32_t num_luts, n_lu_slots, n_lut_slots, lu_degree, lut_degree, n_sldc;
} orc_lookup_shape;

/* ---- gate evaluators: base field ---- */
#define FE uint64_t
#define FE_ADD(a, b) gl_add(a, b)
#define FE_SUB(a, b) gl_sub(a, b)
#define FE_MUL(a, b) gl_mul(a, b)
#define FE_U64(c) ((uint64_t)(c))
#define FN(name) name##_base
#include "gates.inc"
#undef FE
#undef FE_ADD
#undef FE_SUB
#undef FE_MUL
#undef FE_U64
#undef FN
/* ---- gate evaluators: quadratic extension ---- */
#define FE gl2
#define FE_ADD(a, b) gl2_add(a, b)
#define FE_SUB(a, b) gl2_sub(a, b)
#define FE_MUL(a, b) gl2_mul(a, b)
#define FE_U64(c) gl2_from((uint64_t)(c))
#define FN(name) name##_ext
#include "gates.inc"
#undef FE
#undef FE_ADD
#undef FE_SUB
#undef FE_MUL
#undef FE_U64
#undef FN

/* Poseidon::poseidon (fast partial rounds) - must equal the naive schedule */
void orc_poseidon_permute_fast(uint64_t s[12]) {
    int rc = 0;
    for (int r = 0; r < 4; r++, rc++) {
        for (int i = 0; i < 12; i++) s[i] = sbox7_base(gl_add(s[i], ORC_RC[rc * 12 + i]));
        mds_layer_base(s);
    }
    for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], FAST.first_round_constant[i]);
    uint64_t res[12] = {s[0]};
    for (int r = 1; r < 12; r++)
        for (int c = 1; c < 12; c++) res[c] = gl_add(res[c], gl_mul(s[r], FAST.initial_matrix[r - 1][c - 1]));
    memcpy(s, res, sizeof res);
    for (int r = 0; r < 22; r++) {
        s[0] = sbox7_base(s[0]);
        if (r < 21) s[0] = gl_add(s[0], FAST.round_constants[r]);
        uint64_t d = gl_mul(s[0], ORC_MDS_CIRC[0] + ORC_MDS_DIAG[0]);
        for (int i = 1; i < 12; i++) d = gl_add(d, gl_mul(s[i], FAST.w_hats[r][i - 1]));
        for (int i = 1; i < 12; i++) s[i] = gl_add(s[i], gl_mul(s[0], FAST.vs[r][i - 1]));
        s[0] = d;
    }
    rc += 22;
    for (int r = 0; r < 4; r++, rc++) {
        for (int i = 0; i < 12; i++) s[i] = sbox7_base(gl_add(s[i], ORC_RC[rc * 12 + i]));
        mds_layer_base(s);
    }
}

/* hashing::hash_pad-style: Hasher::hash_pad = pad with 1, zeros, 1 to a multiple of the width */
void orc_hash_pad(const uint64_t* in, size_t len, uint64_t out[4]) {
    size_t padded = len + 1;
    while ((padded + 1) % 12 != 0) padded++;
    padded++;
    uint64_t* buf = (uint64_t*)calloc(padded, 8);
    if (len) memcpy(buf, in, len * 8);
    buf[len] = 1;
    buf[padded - 1] = 1;
    orc_hash_no_pad(buf, padded, out);
    free(buf);
}

struct orc_circuit {
    orc_circuit_desc d;
    orc_gate* gates;
    uint64_t* k_is;
    size_t n, L;
    unsigned log_n, log_L;
    uint32_t n_cs;          /* constants + sigmas columns */
    uint32_t n_consts_all;  /* selectors + gate constants */
    uint32_t n_zs;          /* num_challenges * (1 + num_partial_products) + n_lk_polys */
    uint32_t n_zpp;         /* num_challenges * (1 + num_partial_products): the Zs and partial products alone */
    uint32_t n_lk_sel;      /* lookup selector columns between the gate selectors and the gate constants: 4 + num_luts or 0 */
    uint32_t n_lk_polys;    /* num_challenges * (1 + S) or 0 */
    uint32_t n_lk_terms;    /* constraints per challenge round: 4 + num_luts + 2 S or 0 */
    orc_lookup_shape lk;
    uint32_t *lut_sizes, *lookup_rows, *lut_num_lookups, *lut_offsets; /* copies of the descriptor's arrays */
    uint16_t* lut_pairs;
    uint32_t n_q;           /* num_challenges * quotient_degree_factor */
    uint32_t max_constraints;
    uint64_t* sigma_values; /* routed x n column-major (prover_data.sigmas) */
    uint64_t* cs_coeffs;    /* n_cs x n */
    uint64_t* cs_leaves;    /* L x n_cs */
    uint64_t* cs_digests;
    uint64_t* cs_cap;
    uint32_t n_fri_rounds;
};

/* FriReductionStrategy::ConstantArityBits(arity_bits, final_poly_bits).reduction_arity_bits */
static uint32_t fri_num_rounds(const orc_circuit_desc* d) {
    uint32_t degree_bits = d->degree_bits, r = 0;
    while (degree_bits > d->fri_final_poly_bits && degree_bits + d->rate_bits >= d->cap_height + d->fri_arity_bits) {
        if (degree_bits < d->fri_arity_bits) break;
        degree_bits -= d->fri_arity_bits;
        r++;
    }
    return r;
}

orc_circuit* orc_circuit_build(const orc_circuit_desc* desc, const uint64_t* constants, const uint64_t* sigmas) {
    orc_circuit* c = (orc_circuit*)calloc(1, sizeof *c);
    c->d = *desc;
    c->gates = (orc_gate*)malloc(sizeof(orc_gate) * desc->num_gates);
    memcpy(c->gates, desc->gates, sizeof(orc_gate) * desc->num_gates);
    c->d.gates = c->gates;
    c->k_is = (uint64_t*)malloc(8 * desc->num_routed_wires);
    memcpy(c->k_is, desc->k_is, 8 * desc->num_routed_wires);
    c->d.k_is = c->k_is;
    c->log_n = desc->degree_bits;
    c->log_L = desc->degree_bits + desc->rate_bits;
    c->n = (size_t)1 << c->log_n;
    c->L = (size_t)1 << c->log_L;
    if (desc->num_luts) {
        uint32_t T = desc->num_luts;
        c->lk.num_luts = T;
        c->lk.n_lu_slots = desc->num_routed_wires / 2;
        c->lk.n_lut_slots = desc->num_routed_wires / 3;
        c->lk.lu_degree = desc->quotient_degree_factor - 1;
        c->lk.n_sldc = (c->lk.n_lu_slots + c->lk.lu_degree - 1) / c->lk.lu_degree;
        c->lk.lut_degree = (c->lk.n_lut_slots + c->lk.n_sldc - 1) / c->lk.n_sldc;
        c->n_lk_sel = 4 + T;
        c->n_lk_polys = desc->num_challenges * (1 + c->lk.n_sldc);
        c->n_lk_terms = 4 + T + 2 * c->lk.n_sldc;
        c->lut_sizes = (uint32_t*)malloc(4 * T);
        c->lut_num_lookups = (uint32_t*)malloc(4 * T);
        c->lookup_rows = (uint32_t*)malloc(12 * T);
        c->lut_offsets = (uint32_t*)malloc(4 * (T + 1));
        memcpy(c->lut_sizes, desc->lut_sizes, 4 * T);
        memcpy(c->lut_num_lookups, desc->lut_num_lookups, 4 * T);
        memcpy(c->lookup_rows, desc->lookup_rows, 12 * T);
        c->lut_offsets[0] = 0;
        for (uint32_t t = 0; t < T; t++) c->lut_offsets[t + 1] = c->lut_offsets[t] + c->lut_sizes[t];
        c->lut_pairs = (uint16_t*)malloc(4 * (size_t)c->lut_offsets[T]);
        memcpy(c->lut_pairs, desc->lut_pairs, 4 * (size_t)c->lut_offsets[T]);
        c->d.lut_sizes = c->lut_sizes; c->d.lut_num_lookups = c->lut_num_lookups;
        c->d.lookup_rows = c->lookup_rows; c->d.lut_pairs = c->lut_pairs;
    }
    c->n_consts_all = desc->num_selectors + c->n_lk_sel + desc->num_constants;
    c->n_cs = c->n_consts_all + desc->num_routed_wires;
    c->n_zpp = desc->num_challenges * (1 + desc->num_partial_products);
    c->n_zs = c->n_zpp + c->n_lk_polys;
    c->n_q = desc->num_challenges * desc->quotient_degree_factor;
    c->max_constraints = 0;
    for (uint32_t g = 0; g < desc->num_gates; g++) {
        uint32_t k = gate_num_constraints_base(&c->gates[g]);
        if (k > c->max_constraints) c->max_constraints = k;
    }
    c->n_fri_rounds = fri_num_rounds(desc);
    size_t n = c->n;
    c->sigma_values = (uint64_t*)malloc(8 * n * desc->num_routed_wires);
    memcpy(c->sigma_values, sigmas, 8 * n * desc->num_routed_wires);
    uint64_t* cs_values = (uint64_t*)malloc(8 * n * c->n_cs);
    memcpy(cs_values, constants, 8 * n * c->n_consts_all);
    memcpy(cs_values + n * c->n_consts_all, sigmas, 8 * n * desc->num_routed_wires);
    c->cs_coeffs = (uint64_t*)malloc(8 * n * c->n_cs);
    c->cs_leaves = (uint64_t*)malloc(8 * c->L * c->n_cs);
    c->cs_digests = (uint64_t*)malloc(8 * orc_merkle_digest_words(c->L, desc->cap_height));
    c->cs_cap = (uint64_t*)malloc(32 << desc->cap_height);
    orc_commit_from_values(cs_values, c->n_cs, c->log_n, desc->rate_bits, desc->cap_height, c->cs_coeffs, c->cs_leaves,
                           c->cs_digests, c->cs_cap);
    free(cs_values);
    int zero = 1;
    for (int i = 0; i < 4; i++) zero &= desc->circuit_digest[i] == 0;
    if (zero) {
        /* CircuitBuilder::build: circuit_digest = hash_no_pad(cap || hash_pad(domain_separator=[]) || degree_bits) */
        size_t capw = (size_t)4 << desc->cap_height;
        uint64_t* parts = (uint64_t*)malloc(8 * (capw + 5));
        memcpy(parts, c->cs_cap, capw * 8);
        orc_hash_pad(NULL, 0, parts + capw);
        parts[capw + 4] = desc->degree_bits;
        orc_hash_no_pad(parts, capw + 5, c->d.circuit_digest);
        free(parts);
    }
    return c;
}

void orc_circuit_free(orc_circuit* c) {
    if (!c) return;
    free(c->gates); free(c->k_is); free(c->sigma_values); free(c->cs_coeffs); free(c->cs_leaves);
    free(c->cs_digests); free(c->cs_cap);
    free(c->lut_sizes); free(c->lookup_rows); free(c->lut_num_lookups); free(c->lut_offsets); free(c->lut_pairs);
    free(c);
}
void orc_circuit_digest(const orc_circuit* c, uint64_t out[4]) { memcpy(out, c->d.circuit_digest, 32); }
void orc_circuit_constants_sigmas_cap(const orc_circuit* c, uint64_t* cap_out) {
    memcpy(cap_out, c->cs_cap, 32 << c->d.cap_height);
}

#include "bytes.h"

/* ---------------- helpers ---------------- */
static gl2 gl2_add_base(gl2 x, uint64_t b) { return gl2_make(gl_add(x.a, b), x.b); }

/* filter: prod_{i in group, i != index}(i - s) * (UNUSED - s if many selectors) */
static uint64_t filter_base(const orc_gate* g, uint64_t s, int many) {
    uint64_t f = 1;
    for (uint32_t i = g->group_start; i < g->group_end; i++)
        if (i != g->index) f = gl_mul(f, gl_sub(i, s));
    if (many) f = gl_mul(f, gl_sub(UNUSED_SELECTOR, s));
    return f;
}
static gl2 filter_ext(const orc_gate* g, gl2 s, int many) {
    gl2 f = gl2_from(1);
    for (uint32_t i = g->group_start; i < g->group_end; i++)
        if (i != g->index) f = gl2_mul(f, gl2_sub(gl2_from(i), s));
    if (many) f = gl2_mul(f, gl2_sub(gl2_from(UNUSED_SELECTOR), s));
    return f;
}

/* ---------------- lookup argument ---------------- */
/* vanishing_poly::get_lut_poly: the table's pairs (inp + B out) as the coefficients of a polynomial in delta, the FIRST entry
 * at the highest power, padded with zero entries to `degree` = slots * rows (what RE reaches on the table's last row). */
static uint64_t get_lut_poly(const orc_circuit* c, uint32_t t, const uint64_t deltas[4]) {
    uint32_t len = c->lut_sizes[t], slots = c->lk.n_lut_slots;
    uint32_t degree = slots * ((len + slots - 1) / slots);
    const uint16_t* pr = c->lut_pairs + 2 * (size_t)c->lut_offsets[t];
    uint64_t acc = 0;
    for (uint32_t i = 0; i < degree; i++) {
        uint64_t coeff = i < len ? gl_add(pr[2 * i], gl_mul(deltas[1], pr[2 * i + 1])) : 0;
        acc = gl_add(gl_mul(acc, deltas[3]), coeff);
    }
    return acc;
}

/* prover::set_lookup_wires */
int orc_set_lookup_wires(const orc_circuit* c, uint64_t* wires) {
    const size_t n = c->n;
    for (uint32_t t = 0; t < c->lk.num_luts; t++) {
        uint32_t len = c->lut_sizes[t], n_lu = c->lk.n_lu_slots, n_lut = c->lk.n_lut_slots;
        uint32_t last_lu = c->lookup_rows[3 * t], last_lut = c->lookup_rows[3 * t + 1], first_lut = c->lookup_rows[3 * t + 2];
        const uint16_t* pr = c->lut_pairs + 2 * (size_t)c->lut_offsets[t];
        int32_t* idx_of = (int32_t*)malloc(4 * 65536);
        uint64_t* mult = (uint64_t*)calloc(len, 8);
        memset(idx_of, 0xFF, 4 * 65536);
        for (uint32_t i = 0; i < len; i++) idx_of[pr[2 * i]] = (int32_t)i; /* HashMap collect: the last entry with that input wins */
        uint32_t lookups = c->lut_num_lookups[t];
        int bad = 0;
        /* lookups fill the LookupGate rows slot after slot from last_lu_row upwards (CircuitBuilder::find_slot) */
        for (uint32_t q = 0; q < lookups; q++) {
            uint32_t row = last_lu + q / n_lu, slot = q % n_lu;
            uint64_t v = wires[(size_t)(2 * slot) * n + row];
            if (row >= last_lut || v > 0xFFFF || idx_of[v] < 0) { bad = 1; break; }
            mult[idx_of[v]]++;
        }
        if (!bad) {
            uint32_t remaining = (n_lu - lookups % n_lu) % n_lu;
            for (uint32_t slot = n_lu - remaining; slot < n_lu; slot++) {
                wires[(size_t)(2 * slot) * n + (last_lut - 1)] = pr[0];
                wires[(size_t)(2 * slot + 1) * n + (last_lut - 1)] = pr[1];
                mult[0]++;
            }
            for (uint32_t e = 0; e < len; e++)
                wires[(size_t)(3 * (e % n_lut) + 2) * n + (first_lut - e / n_lut)] = mult[e];
        }
        free(idx_of); free(mult);
        if (bad) return -1;
    }
    return 0;
}

/* prover::compute_lookup_polys for one challenge round: cols = (1 + S) x n column-major, zeroed here.  RE and the Sum run
 * down the LookupTableGate rows from first_lut_row to last_lut_row, the LDC continues down the LookupGate rows. */
static void compute_lookup_polys(const orc_circuit* c, const uint64_t* wires, const uint64_t deltas[4], uint64_t* cols) {
    const size_t n = c->n;
    const uint32_t S = c->lk.n_sldc, n_lu = c->lk.n_lu_slots, n_lut = c->lk.n_lut_slots;
    memset(cols, 0, 8 * n * (1 + S));
#define WIRE(row, w) wires[(size_t)(w) * n + (row)]
    for (uint32_t t = 0; t < c->lk.num_luts; t++) {
        uint32_t last_lu = c->lookup_rows[3 * t], last_lut = c->lookup_rows[3 * t + 1], first_lut = c->lookup_rows[3 * t + 2];
        for (uint32_t row = first_lut + 1; row-- > last_lut;) {
            uint64_t re = cols[row + 1];
            for (uint32_t i = 0; i < n_lut; i++)
                re = gl_add(gl_mul(re, deltas[3]), gl_add(WIRE(row, 3 * i), gl_mul(deltas[1], WIRE(row, 3 * i + 1))));
            cols[row] = re;
            for (uint32_t p = 0; p < S; p++) {
                uint64_t sum = p ? cols[(size_t)p * n + row] : cols[(size_t)S * n + row + 1];
                for (uint32_t i = p * c->lk.lut_degree; i < (p + 1) * c->lk.lut_degree && i < n_lut; i++) {
                    uint64_t combo = gl_add(WIRE(row, 3 * i), gl_mul(deltas[0], WIRE(row, 3 * i + 1)));
                    sum = gl_add(sum, gl_mul(WIRE(row, 3 * i + 2), gl_inv(gl_sub(deltas[2], combo))));
                }
                cols[(size_t)(p + 1) * n + row] = sum;
            }
        }
        for (uint32_t row = last_lut; row-- > last_lu;) {
            for (uint32_t p = 0; p < S; p++) {
                uint64_t prev = p ? cols[(size_t)p * n + row] : cols[(size_t)S * n + row + 1];
                uint64_t sum = 0;
                for (uint32_t i = p * c->lk.lu_degree; i < (p + 1) * c->lk.lu_degree && i < n_lu; i++) {
                    uint64_t combo = gl_add(WIRE(row, 2 * i), gl_mul(deltas[0], WIRE(row, 2 * i + 1)));
                    sum = gl_add(sum, gl_inv(gl_sub(deltas[2], combo)));
                }
                cols[(size_t)(p + 1) * n + row] = gl_sub(prev, sum);
            }
        }
    }
#undef WIRE
}

size_t orc_proof_max_bytes(const orc_circuit* c) {
    const orc_circuit_desc* d = &c->d;
    size_t capb = (size_t)32 << d->cap_height;
    size_t words = 0;
    words += 2 * (c->n_cs + d->num_wires + c->n_zs + d->num_challenges + c->n_lk_polys + c->n_q);
    size_t bytes = 3 * capb + words * 8 + c->n_fri_rounds * capb;
    size_t per_query = 0;
    uint32_t cols[4] = {c->n_cs, d->num_wires, c->n_zs, c->n_q};
    for (int o = 0; o < 4; o++) per_query += cols[o] * 8 + 1 + 32 * (c->log_L - d->cap_height);
    for (uint32_t r = 0; r < c->n_fri_rounds; r++) per_query += ((size_t)16 << d->fri_arity_bits) + 1 + 32 * c->log_L;
    bytes += per_query * d->fri_num_queries;
    bytes += 16 * ((size_t)1 << d->degree_bits) /* final poly upper bound */ + 8 + 8 + 8 * d->num_public_inputs;
    return bytes + 64;
}

/* ---------------- prover ---------------- */
typedef struct {
    uint64_t* coeffs;  /* n_cols x n */
    uint64_t* leaves;  /* L x n_cols */
    uint64_t* digests;
    uint64_t* cap;
    uint32_t n_cols;
} batch;

static void batch_alloc(const orc_circuit* c, batch* b, uint32_t n_cols) {
    b->n_cols = n_cols;
    b->coeffs = (uint64_t*)malloc(8 * c->n * n_cols);
    b->leaves = (uint64_t*)malloc(8 * c->L * n_cols);
    b->digests = (uint64_t*)malloc(8 * orc_merkle_digest_words(c->L, c->d.cap_height));
    b->cap = (uint64_t*)malloc(32 << c->d.cap_height);
}
static void batch_free(batch* b) { free(b->coeffs); free(b->leaves); free(b->digests); free(b->cap); }

static void observe_cap(orc_challenger* ch, const uint64_t* cap, unsigned cap_height) {
    orc_ch_observe_many(ch, cap, (size_t)4 << cap_height);
}

/* PolynomialCoeffs::eval at an extension point, base coefficients */
static gl2 eval_base_poly_ext(const uint64_t* coeffs, size_t n, gl2 z) {
    gl2 acc = gl2_from(0);
    for (size_t i = n; i-- > 0;) acc = gl2_add_base(gl2_mul(acc, z), coeffs[i]);
    return acc;
}

/* eval_vanishing_poly_base_batch for ONE point.  rows are the opened LDE rows. */
static void vanishing_base(const orc_circuit* c, uint64_t x, size_t i, const uint64_t* cs_row,
                           const uint64_t* wires_row, const uint64_t* zs_row, const uint64_t* zs_next_row,
                           const uint64_t* betas, const uint64_t* gammas, const uint64_t* alphas,
                           const uint64_t* deltas, const uint64_t* lut_polys, const uint64_t* pih, const uint64_t* z_h_evals, uint64_t* tmp_constraints,
                           uint64_t* out /* num_challenges */) {
    const orc_circuit_desc* d = &c->d;
    const uint32_t nc = d->num_challenges, npp = d->num_partial_products, routed = d->num_routed_wires;
    const uint32_t chunk = d->quotient_degree_factor;
    /* gate constraints: constraints[k] = sum_g filter_g * c_{g,k} */
    uint32_t nk = c->max_constraints;
    uint64_t* cons = tmp_constraints;
    uint64_t* gout = tmp_constraints + nk;
    for (uint32_t k = 0; k < nk; k++) cons[k] = 0;
    for (uint32_t g = 0; g < d->num_gates; g++) {
        const orc_gate* gt = &c->gates[g];
        uint32_t k = gate_num_constraints_base(gt);
        if (!k) continue;
        uint64_t f = filter_base(gt, cs_row[gt->selector_index], d->num_selectors > 1);
        gate_eval_base(gt, cs_row + d->num_selectors + c->n_lk_sel, wires_row, pih, gout);
        for (uint32_t j = 0; j < k; j++) cons[j] = gl_add(cons[j], gl_mul(f, gout[j]));
    }
    /* L_0(x) = Z_H(x) / (n (x - 1))  (ZeroPolyOnCoset::eval_l_0) */
    uint64_t zh = z_h_evals[i & ((1u << d->rate_bits) - 1)];
    uint64_t l0 = gl_mul(zh, gl_inv(gl_mul((uint64_t)c->n % GL_P, gl_sub(x, 1))));
    const uint64_t* sig_row = cs_row + c->n_consts_all;
    /* vanishing_terms = [z1 terms (nc)] ++ [partial product terms (nc * (npp+1))] ++ [lookup terms (nc * n_lk_terms)] ++ constraints */
    const uint32_t o_lk = nc + nc * (npp + 1), o_gate = o_lk + nc * c->n_lk_terms;
    uint32_t n_terms = o_gate + nk;
    uint64_t* terms = gout + nk; /* scratch after gate outputs */
    for (uint32_t ci = 0; ci < nc; ci++) {
        uint64_t z_x = zs_row[ci], z_gx = zs_next_row[ci];
        terms[ci] = gl_mul(l0, gl_sub(z_x, 1));
        const uint64_t* pps = zs_row + nc + ci * npp;
        uint64_t acc = z_x;
        uint32_t n_chunks = (routed + chunk - 1) / chunk;
        for (uint32_t q = 0; q < n_chunks; q++) {
            uint64_t num = 1, den = 1;
            for (uint32_t j = q * chunk; j < (q + 1) * chunk && j < routed; j++) {
                uint64_t w = wires_row[j];
                uint64_t s_id = gl_mul(c->k_is[j], x);
                num = gl_mul(num, gl_add(gl_add(w, gl_mul(betas[ci], s_id)), gammas[ci]));
                den = gl_mul(den, gl_add(gl_add(w, gl_mul(betas[ci], sig_row[j])), gammas[ci]));
            }
            uint64_t new_acc = (q + 1 < n_chunks) ? pps[q] : z_gx;
            terms[nc + ci * (npp + 1) + q] = gl_sub(gl_mul(acc, num), gl_mul(new_acc, den));
            acc = new_acc;
        }
    }
    for (uint32_t ci = 0; ci < nc && c->n_lk_terms; ci++)
        lookup_constraints_base(&c->lk, cs_row + d->num_selectors, wires_row, zs_row + c->n_zpp + ci * (1 + c->lk.n_sldc),
                                zs_next_row + c->n_zpp + ci * (1 + c->lk.n_sldc), deltas + 4 * ci,
                                lut_polys + ci * c->lk.num_luts, terms + o_lk + ci * c->n_lk_terms);
    for (uint32_t k = 0; k < nk; k++) terms[o_gate + k] = cons[k];
    uint64_t zh_inv = gl_inv(zh);
    for (uint32_t ci = 0; ci < nc; ci++) {
        uint64_t sum = 0;
        for (uint32_t t = n_terms; t-- > 0;) sum = gl_add(gl_mul(sum, alphas[ci]), terms[t]);
        out[ci] = gl_mul(sum, zh_inv);
    }
}

/* fft of an extension-valued vector: the transform is F_p-linear, do both components */
static void ext_coset_fft(gl2* v, unsigned log_n, uint64_t shift) {
    size_t n = (size_t)1 << log_n;
    uint64_t* a = (uint64_t*)malloc(16 * n);
    uint64_t* b = a + n;
    for (size_t i = 0; i < n; i++) { a[i] = v[i].a; b[i] = v[i].b; }
    orc_coset_fft(a, log_n, shift);
    orc_coset_fft(b, log_n, shift);
    for (size_t i = 0; i < n; i++) { v[i].a = a[i]; v[i].b = b[i]; }
    free(a);
}

size_t orc_prove_traced(const orc_circuit* c, const uint64_t* wires, const uint64_t* public_inputs,
                        uint8_t* proof_out, size_t cap_bytes, orc_trace* tr) {
    const orc_circuit_desc* d = &c->d;
    const size_t n = c->n, L = c->L;
    const uint32_t nc = d->num_challenges, npp = d->num_partial_products, routed = d->num_routed_wires;
    const uint32_t chunk = d->quotient_degree_factor;
    const unsigned cap_h = d->cap_height;
    wbuf w = {proof_out, 0, cap_bytes, 0};
    size_t ret = 0;

    /* 0. set_lookup_wires: on a copy, the caller's witness stays as it was handed over */
    uint64_t* wires_lk = NULL;
    if (d->num_luts) {
        wires_lk = (uint64_t*)malloc(8 * n * d->num_wires);
        memcpy(wires_lk, wires, 8 * n * d->num_wires);
        if (orc_set_lookup_wires(c, wires_lk)) { free(wires_lk); return 0; }
        wires = wires_lk;
    }

    /* 1. public inputs hash */
    uint64_t pih[4];
    orc_hash_no_pad(public_inputs, d->num_public_inputs, pih);

    /* 2. wires commitment */
    batch bw, bz, bq;
    batch_alloc(c, &bw, d->num_wires);
    orc_commit_from_values(wires, d->num_wires, c->log_n, d->rate_bits, cap_h, bw.coeffs, bw.leaves, bw.digests, bw.cap);

    /* 3. challenger: circuit digest, public inputs hash, wires cap -> betas, gammas */
    orc_challenger ch;
    orc_ch_init(&ch);
    orc_ch_observe_many(&ch, d->circuit_digest, 4);
    orc_ch_observe_many(&ch, pih, 4);
    observe_cap(&ch, bw.cap, cap_h);
    uint64_t betas[4], gammas[4], alphas[4];
    for (uint32_t i = 0; i < nc; i++) betas[i] = orc_ch_challenge(&ch);
    for (uint32_t i = 0; i < nc; i++) gammas[i] = orc_ch_challenge(&ch);
    /* lookups: deltas = betas ++ gammas ++ 2 nc more challenges, NUM_COINS_LOOKUP = 4 per round (A, B, alpha, delta) */
    uint64_t deltas[16] = {0}, lut_polys[4 * 16] = {0};
    if (d->num_luts) {
        if (nc > 4 || d->num_luts > 16) { free(wires_lk); batch_free(&bw); return 0; }
        for (uint32_t i = 0; i < nc; i++) { deltas[i] = betas[i]; deltas[nc + i] = gammas[i]; }
        for (uint32_t i = 0; i < 2 * nc; i++) deltas[2 * nc + i] = orc_ch_challenge(&ch);
        for (uint32_t ci = 0; ci < nc; ci++)
            for (uint32_t t = 0; t < d->num_luts; t++) lut_polys[ci * d->num_luts + t] = get_lut_poly(c, t, deltas + 4 * ci);
    }

    /* 4. wires_permutation_partial_products_and_zs: column order [Z_0..Z_{nc-1}, pp(0,·), pp(1,·)..] */
    uint64_t* zs_values = (uint64_t*)malloc(8 * n * c->n_zs);
    {
        /* (i) per row, in parallel: cumulative chunk quotients; (ii) serial running product down the
         * rows (Z_0 = 1, Z_{i+1} = Z_i * rowprod_i); (iii) partial products = Z_i * cumulative quotient */
        uint64_t w_n = gl_root_of_unity(c->log_n);
        uint32_t n_chunks = (routed + chunk - 1) / chunk;
        for (uint32_t ci = 0; ci < nc; ci++) {
#pragma omp parallel for schedule(static)
            for (size_t i = 0; i < n; i++) {
                uint64_t x = gl_pow(w_n, i);
                uint64_t acc = 1;
                for (uint32_t q = 0; q < n_chunks; q++) {
                    uint64_t num = 1, den = 1;
                    for (uint32_t j = q * chunk; j < (q + 1) * chunk && j < routed; j++) {
                        uint64_t wv = wires[(size_t)j * n + i];
                        uint64_t s_id = gl_mul(c->k_is[j], x);
                        num = gl_mul(num, gl_add(gl_add(wv, gl_mul(betas[ci], s_id)), gammas[ci]));
                        den = gl_mul(den, gl_add(gl_add(wv, gl_mul(betas[ci], c->sigma_values[(size_t)j * n + i])), gammas[ci]));
                    }
                    acc = gl_mul(acc, gl_mul(num, gl_inv(den)));
                    if (q + 1 < n_chunks) zs_values[(size_t)(nc + ci * npp + q) * n + i] = acc;
                    else zs_values[(size_t)ci * n + i] = acc;
                }
            }
            uint64_t z_x = 1;
            for (size_t i = 0; i < n; i++) {
                uint64_t rowprod = zs_values[(size_t)ci * n + i];
                zs_values[(size_t)ci * n + i] = z_x;
                z_x = gl_mul(z_x, rowprod);
            }
#pragma omp parallel for schedule(static)
            for (size_t i = 0; i < n; i++)
                for (uint32_t q = 0; q + 1 < n_chunks; q++) {
                    uint64_t* pp = &zs_values[(size_t)(nc + ci * npp + q) * n + i];
                    *pp = gl_mul(*pp, zs_values[(size_t)ci * n + i]);
                }
        }
    }
    /* compute_all_lookup_polys: the rounds' (RE, SLDC_0..S-1) after the partial products in the same commitment */
    for (uint32_t ci = 0; ci < nc && c->n_lk_polys; ci++)
        compute_lookup_polys(c, wires, deltas + 4 * ci, zs_values + (size_t)(c->n_zpp + ci * (1 + c->lk.n_sldc)) * n);
    if (tr && tr->zs_partial_values) memcpy(tr->zs_partial_values, zs_values, 8 * n * c->n_zs);
    batch_alloc(c, &bz, c->n_zs);
    orc_commit_from_values(zs_values, c->n_zs, c->log_n, d->rate_bits, cap_h, bz.coeffs, bz.leaves, bz.digests, bz.cap);
    free(zs_values);
    observe_cap(&ch, bz.cap, cap_h);
    for (uint32_t i = 0; i < nc; i++) alphas[i] = orc_ch_challenge(&ch);

    /* 5. compute_quotient_polys */
    uint64_t* qvals = (uint64_t*)malloc(8 * L * nc); /* column-major nc x L */
    {
        uint32_t rate = 1u << d->rate_bits;
        uint64_t z_h_evals[64];
        uint64_t g_n = gl_exp_pow2(GL_GEN, c->log_n);
        uint64_t w_rate = gl_root_of_unity(d->rate_bits);
        for (uint32_t r = 0; r < rate; r++) z_h_evals[r] = gl_sub(gl_mul(g_n, gl_pow(w_rate, r)), 1);
        uint64_t w_L = gl_root_of_unity(c->log_L);
        size_t next_step = (size_t)1 << d->rate_bits; /* quotient_degree_bits == rate_bits */
#pragma omp parallel
        {
            uint64_t* tmp = (uint64_t*)malloc(8 * (3 * c->max_constraints + 64 + nc * (npp + 2 + c->n_lk_terms)));
#pragma omp for schedule(static)
            for (size_t i = 0; i < L; i++) {
                uint64_t x = gl_mul(GL_GEN, gl_pow(w_L, i));
                size_t li = gl_bitrev(i, c->log_L), ln = gl_bitrev((i + next_step) % L, c->log_L);
                uint64_t out[4];
                vanishing_base(c, x, i, c->cs_leaves + li * c->n_cs, bw.leaves + li * d->num_wires,
                               bz.leaves + li * c->n_zs, bz.leaves + ln * c->n_zs, betas, gammas, alphas, deltas,
                               lut_polys, pih,
                               z_h_evals, tmp, out);
                for (uint32_t ci = 0; ci < nc; ci++) qvals[(size_t)ci * L + i] = out[ci];
            }
            free(tmp);
        }
    }
    /* coset_ifft, split into quotient_degree_factor chunks of n coefficients */
    uint64_t* qchunks = (uint64_t*)malloc(8 * n * c->n_q);
    int degree_ok = 1;
    for (uint32_t ci = 0; ci < nc; ci++) {
        orc_coset_ifft(qvals + (size_t)ci * L, c->log_L, GL_GEN);
        /* trim_to_len(quotient_degree): here quotient_degree = L, nothing to trim */
        memcpy(qchunks + (size_t)ci * chunk * n, qvals + (size_t)ci * L, 8 * n * chunk);
    }
    (void)degree_ok;
    free(qvals);
    if (tr && tr->quotient_chunk_coeffs) memcpy(tr->quotient_chunk_coeffs, qchunks, 8 * n * c->n_q);
    batch_alloc(c, &bq, c->n_q);
    memcpy(bq.coeffs, qchunks, 8 * n * c->n_q);
    orc_commit_from_coeffs(qchunks, c->n_q, c->log_n, d->rate_bits, cap_h, bq.leaves, bq.digests, bq.cap);
    free(qchunks);
    observe_cap(&ch, bq.cap, cap_h);

    /* 6. zeta, openings */
    gl2 zeta = orc_ch_ext_challenge(&ch);
    uint64_t g = gl_root_of_unity(c->log_n);
    gl2 g_zeta = gl2_scale(zeta, g);
    const batch* oracles[4];
    batch bcs = {c->cs_coeffs, c->cs_leaves, c->cs_digests, c->cs_cap, c->n_cs};
    oracles[0] = &bcs; oracles[1] = &bw; oracles[2] = &bz; oracles[3] = &bq;
    /* CommonCircuitData::fri_all_polys: constants_sigmas, wires, Zs ++ partial products, quotient chunks, and LAST the
     * lookup polynomials (columns n_zpp.. of the Zs commitment); fri_next_batch_polys: the Zs, then the lookup polynomials.
     * views[] lists the zeta batch in that order as (coefficients, column count) */
    struct { const uint64_t* coeffs; uint32_t n_cols; } views[5] = {
        {bcs.coeffs, c->n_cs}, {bw.coeffs, d->num_wires}, {bz.coeffs, c->n_zpp}, {bq.coeffs, c->n_q},
        {bz.coeffs + (size_t)c->n_zpp * n, c->n_lk_polys}};
    uint32_t n_open = c->n_cs + d->num_wires + c->n_zs + c->n_q;
    uint32_t n_next = nc + c->n_lk_polys;
    gl2* open_zeta = (gl2*)malloc(sizeof(gl2) * n_open);
    gl2* open_next = (gl2*)malloc(sizeof(gl2) * n_next);
    {
        uint32_t base_k[6] = {0, 0, 0, 0, 0, 0};
        for (int o = 0; o < 5; o++) base_k[o + 1] = base_k[o] + views[o].n_cols;
#pragma omp parallel for schedule(dynamic)
        for (uint32_t k = 0; k < n_open; k++) {
            int o = 0;
            while (k >= base_k[o + 1]) o++;
            open_zeta[k] = eval_base_poly_ext(views[o].coeffs + (size_t)(k - base_k[o]) * n, n, zeta);
        }
        for (uint32_t p = 0; p < n_next; p++)
            open_next[p] = eval_base_poly_ext(bz.coeffs + (size_t)(p < nc ? p : c->n_zpp + (p - nc)) * n, n, g_zeta);
    }
    /* proof: caps + OpeningSet {constants, plonk_sigmas, wires, plonk_zs, plonk_zs_next, partial_products, quotient_polys} */
    w_u64s(&w, bw.cap, (size_t)4 << cap_h);
    w_u64s(&w, bz.cap, (size_t)4 << cap_h);
    w_u64s(&w, bq.cap, (size_t)4 << cap_h);
    {
        uint32_t o_wires = c->n_cs, o_zs = o_wires + d->num_wires, o_pp = o_zs + nc, o_q = o_zs + c->n_zpp, o_lk = o_q + c->n_q;
        w_u64s(&w, (uint64_t*)open_zeta, 2 * c->n_consts_all);                   /* constants */
        w_u64s(&w, (uint64_t*)(open_zeta + c->n_consts_all), 2 * routed);         /* plonk_sigmas */
        w_u64s(&w, (uint64_t*)(open_zeta + o_wires), 2 * d->num_wires);           /* wires */
        w_u64s(&w, (uint64_t*)(open_zeta + o_zs), 2 * nc);                        /* plonk_zs */
        w_u64s(&w, (uint64_t*)open_next, 2 * nc);                                 /* plonk_zs_next */
        w_u64s(&w, (uint64_t*)(open_zeta + o_lk), 2 * c->n_lk_polys);             /* lookup_zs (read_opening_set order) */
        w_u64s(&w, (uint64_t*)(open_next + nc), 2 * c->n_lk_polys);               /* lookup_zs_next */
        w_u64s(&w, (uint64_t*)(open_zeta + o_pp), 2 * nc * npp);                  /* partial_products */
        w_u64s(&w, (uint64_t*)(open_zeta + o_q), 2 * c->n_q);                     /* quotient_polys */
    }
    /* challenger.observe_openings(to_fri_openings): zeta batch in oracle order, then zeta_next batch */
    orc_ch_observe_many(&ch, (uint64_t*)open_zeta, 2 * n_open);
    orc_ch_observe_many(&ch, (uint64_t*)open_next, 2 * n_next);

    /* 7. PolynomialBatch::prove_openings */
    gl2 alpha = orc_ch_ext_challenge(&ch);
    gl2* final_poly = (gl2*)calloc(L, sizeof(gl2)); /* n coefficients, zero padded to L (lde) */
    {
        /* batch 0: all polynomials at zeta */
        gl2* comp = (gl2*)calloc(n, sizeof(gl2));
        gl2* apows = (gl2*)malloc(sizeof(gl2) * (n_open + 1));
        apows[0] = gl2_from(1);
        for (uint32_t k = 0; k < n_open; k++) apows[k + 1] = gl2_mul(apows[k], alpha);
        gl2 apow;
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n; i++) {
            gl2 acc = gl2_from(0);
            uint32_t k = 0;
            for (int o = 0; o < 5; o++)
                for (uint32_t p = 0; p < views[o].n_cols; p++, k++)
                    acc = gl2_add(acc, gl2_scale(apows[k], views[o].coeffs[(size_t)p * n + i]));
            comp[i] = acc;
        }
        free(apows);
        /* divide_by_linear(zeta): q_{i-1} = c_i + z q_i */
        gl2 acc = gl2_from(0);
        for (size_t i = n; i-- > 1;) { acc = gl2_add(gl2_mul(acc, zeta), comp[i]); final_poly[i - 1] = acc; }
        /* batch 1: Zs at g*zeta; alpha.shift_poly(final_poly) multiplies by alpha^(count of batch 1) */
        memset(comp, 0, sizeof(gl2) * n);
        apow = gl2_from(1);
        for (uint32_t p = 0; p < n_next; p++) {
            const uint64_t* co = bz.coeffs + (size_t)(p < nc ? p : c->n_zpp + (p - nc)) * n;
            for (size_t i = 0; i < n; i++) comp[i] = gl2_add(comp[i], gl2_scale(apow, co[i]));
            apow = gl2_mul(apow, alpha);
        }
        for (size_t i = 0; i < n; i++) final_poly[i] = gl2_mul(final_poly[i], apow);
        acc = gl2_from(0);
        for (size_t i = n; i-- > 1;) { acc = gl2_add(gl2_mul(acc, g_zeta), comp[i]); final_poly[i - 1] = gl2_add(final_poly[i - 1], acc); }
        free(comp);
    }
    gl2* coeffs = final_poly;           /* length L (upper part zero) */
    gl2* values = (gl2*)malloc(sizeof(gl2) * L);
    memcpy(values, coeffs, sizeof(gl2) * L);
    ext_coset_fft(values, c->log_L, GL_GEN);
    if (tr && tr->fri_final_values) memcpy(tr->fri_final_values, values, sizeof(gl2) * L);

    /* fri_committed_trees */
    uint32_t R = c->n_fri_rounds;
    uint64_t** tree_leaves = (uint64_t**)calloc(R + 1, sizeof(uint64_t*));
    uint64_t** tree_digests = (uint64_t**)calloc(R + 1, sizeof(uint64_t*));
    size_t* tree_nleaves = (size_t*)calloc(R + 1, sizeof(size_t));
    gl2 fri_betas[16];
    size_t cur_len = L;
    uint64_t shift = GL_GEN;
    const uint32_t arity = 1u << d->fri_arity_bits;
    for (uint32_t r = 0; r < R; r++) {
        unsigned lg = gl_log2_strict(cur_len);
        /* reverse_index_bits_in_place(values); leaves = chunks(arity) flattened */
        size_t n_leaves = cur_len / arity;
        uint64_t* lv = (uint64_t*)malloc(16 * cur_len);
        for (size_t j = 0; j < cur_len; j++) {
            gl2 v = values[gl_bitrev(j, lg)];
            lv[2 * j] = v.a;
            lv[2 * j + 1] = v.b;
        }
        uint64_t* dg = (uint64_t*)malloc(8 * orc_merkle_digest_words(n_leaves, cap_h));
        uint64_t capbuf[4 * 64];
        orc_merkle_build(lv, n_leaves, 2 * arity, cap_h, dg, capbuf);
        tree_leaves[r] = lv; tree_digests[r] = dg; tree_nleaves[r] = n_leaves;
        w_u64s(&w, capbuf, (size_t)4 << cap_h);
        observe_cap(&ch, capbuf, cap_h);
        gl2 beta = orc_ch_ext_challenge(&ch);
        fri_betas[r] = beta;
        /* coeffs = chunks(arity).map(reduce_with_powers(chunk, beta)) */
        size_t new_len = cur_len / arity;
        for (size_t j = 0; j < new_len; j++) {
            gl2 acc = gl2_from(0);
            for (uint32_t t = arity; t-- > 0;) acc = gl2_add(gl2_mul(acc, beta), coeffs[j * arity + t]);
            coeffs[j] = acc;
        }
        cur_len = new_len;
        shift = gl_exp_pow2(shift, d->fri_arity_bits);
        memcpy(values, coeffs, sizeof(gl2) * cur_len);
        ext_coset_fft(values, gl_log2_strict(cur_len), shift);
    }
    size_t final_len = cur_len >> d->rate_bits; /* truncate: removed coefficients are zero */
    for (size_t i = final_len; i < cur_len; i++)
        if (coeffs[i].a || coeffs[i].b) degree_ok = 0;
    orc_ch_observe_many(&ch, (uint64_t*)coeffs, 2 * final_len);

    /* fri_proof_of_work: smallest witness (upstream uses a parallel find_any - see DESIGN.md) */
    uint64_t pow_witness = 0;
    {
        unsigned min_lz = d->fri_pow_bits + (64 - 64); /* F::order().bits() == 64 */
        uint64_t st0[12];
        memcpy(st0, ch.state, sizeof st0);
        memcpy(st0, ch.in_buf, ch.n_in * 8);
        unsigned pos = ch.n_in;
        const uint64_t CH = 1 << 14;
        int found = 0;
        for (uint64_t start = 0; !found; start += CH) {
            uint64_t best = ~0ULL;
#pragma omp parallel for schedule(static) reduction(min : best)
            for (uint64_t cand = start; cand < start + CH; cand++) {
                uint64_t st[12];
                memcpy(st, st0, sizeof st);
                st[pos] = cand;
                orc_poseidon_permute(st);
                uint64_t resp = st[7]; /* squeeze().last() */
                unsigned lz = resp ? (unsigned)__builtin_clzll(resp) : 64;
                if (lz >= min_lz && cand < best) best = cand;
            }
            if (best != ~0ULL) { pow_witness = best; found = 1; }
        }
        orc_ch_observe(&ch, pow_witness);
        (void)orc_ch_challenge(&ch); /* pow_response */
    }

    /* fri_prover_query_rounds */
    uint64_t qidx[128];
    for (uint32_t q = 0; q < d->fri_num_queries; q++) qidx[q] = orc_ch_challenge(&ch) % L;
    for (uint32_t q = 0; q < d->fri_num_queries; q++) {
        size_t x_index = qidx[q];
        for (int o = 0; o < 4; o++) {
            const batch* b = oracles[o];
            w_u64s(&w, b->leaves + x_index * b->n_cols, b->n_cols);
            unsigned plen = c->log_L - cap_h;
            uint64_t sib[64 * 4];
            orc_merkle_prove(b->digests, L, cap_h, x_index, sib);
            w_u8(&w, (uint8_t)plen);
            w_u64s(&w, sib, 4 * plen);
        }
        for (uint32_t r = 0; r < R; r++) {
            size_t leaf = x_index >> d->fri_arity_bits;
            w_u64s(&w, tree_leaves[r] + leaf * 2 * arity, 2 * arity);
            unsigned lg = gl_log2_strict(tree_nleaves[r]);
            unsigned plen = lg > cap_h ? lg - cap_h : 0;
            uint64_t sib[64 * 4];
            orc_merkle_prove(tree_digests[r], tree_nleaves[r], cap_h, leaf, sib);
            w_u8(&w, (uint8_t)plen);
            w_u64s(&w, sib, 4 * plen);
            x_index = leaf;
        }
    }
    w_u64s(&w, (uint64_t*)coeffs, 2 * final_len);
    w_u64s(&w, &pow_witness, 1);
    w_usize(&w, d->num_public_inputs); /* write_proof_with_public_inputs: write_usize(len) then the field vec */
    w_u64s(&w, public_inputs, d->num_public_inputs);

    if (tr) {
        memcpy(tr->betas, betas, sizeof betas);
        memcpy(tr->gammas, gammas, sizeof gammas);
        memcpy(tr->alphas, alphas, sizeof alphas);
        tr->zeta[0] = zeta.a; tr->zeta[1] = zeta.b;
        tr->fri_alpha[0] = alpha.a; tr->fri_alpha[1] = alpha.b;
        for (uint32_t r = 0; r < R; r++) { tr->fri_betas[2 * r] = fri_betas[r].a; tr->fri_betas[2 * r + 1] = fri_betas[r].b; }
        tr->pow_witness = pow_witness;
        tr->n_fri_rounds = R;
        memcpy(tr->query_indices, qidx, 8 * d->fri_num_queries);
        memcpy(tr->deltas, deltas, sizeof deltas);
    }
    ret = (w.overflow || !degree_ok) ? 0 : w.len;

    for (uint32_t r = 0; r < R; r++) { free(tree_leaves[r]); free(tree_digests[r]); }
    free(tree_leaves); free(tree_digests); free(tree_nleaves);
    free(values); free(final_poly); free(open_zeta); free(open_next);
    batch_free(&bw); batch_free(&bz); batch_free(&bq);
    free(wires_lk);
    return ret;
}

size_t orc_prove(const orc_circuit* c, const uint64_t* wires, const uint64_t* public_inputs, uint8_t* proof_out,
                 size_t cap_bytes) {
    return orc_prove_traced(c, wires, public_inputs, proof_out, cap_bytes, NULL);
}

/* ---------------- verifier (plonk::verifier::verify + fri::verifier::verify_fri_proof) ---------------- */
static gl2 reduce_with_powers_ext(const gl2* terms, size_t n, gl2 alpha) {
    gl2 sum = gl2_from(0);
    for (size_t i = n; i-- > 0;) sum = gl2_add(gl2_mul(sum, alpha), terms[i]);
    return sum;
}

int orc_verify(const orc_circuit* c, const uint8_t* proof, size_t len) {
    const orc_circuit_desc* d = &c->d;
    const uint32_t nc = d->num_challenges, npp = d->num_partial_products, routed = d->num_routed_wires;
    const uint32_t chunk = d->quotient_degree_factor;
    const unsigned cap_h = d->cap_height;
    const size_t capw = (size_t)4 << cap_h, L = c->L;
    const uint32_t R = c->n_fri_rounds, arity = 1u << d->fri_arity_bits;
    rbuf r = {proof, len, 0, 0};
    int rc = 1;

    uint64_t* caps = (uint64_t*)malloc(8 * capw * (3 + R));
    r_u64s(&r, caps, 3 * capw);
    const uint64_t *wires_cap = caps, *zs_cap = caps + capw, *q_cap = caps + 2 * capw;
    uint32_t n_open = c->n_cs + d->num_wires + c->n_zs + c->n_q;
    /* read_opening_set: constants, sigmas, wires, zs, zs_next, lookup_zs, lookup_zs_next, partial products, quotient */
    const uint32_t nlk = c->n_lk_polys;
    gl2* o_consts = (gl2*)malloc(sizeof(gl2) * (n_open + nc + nlk));
    gl2* o_sigmas = o_consts + c->n_consts_all;
    gl2* o_wires = o_sigmas + routed;
    gl2* o_zs = o_wires + d->num_wires;
    gl2* o_zs_next = o_zs + nc;
    gl2* o_lk = o_zs_next + nc;
    gl2* o_lk_next = o_lk + nlk;
    gl2* o_pp = o_lk_next + nlk;
    gl2* o_q = o_pp + nc * npp;
    r_u64s(&r, (uint64_t*)o_consts, 2 * (n_open + nc + nlk));
    uint64_t* fri_caps = caps + 3 * capw;
    r_u64s(&r, fri_caps, R * capw);
    /* query rounds are parsed later; find the tail (final poly, pow witness, public inputs) */
    size_t per_query = 0;
    uint32_t cols[4] = {c->n_cs, d->num_wires, c->n_zs, c->n_q};
    for (int o = 0; o < 4; o++) per_query += cols[o] * 8 + 1 + 32 * (c->log_L - cap_h);
    {
        size_t nl = L;
        for (uint32_t k = 0; k < R; k++) {
            nl >>= d->fri_arity_bits;
            unsigned lg = gl_log2_strict(nl);
            per_query += 16 * arity + 1 + 32 * (lg > cap_h ? lg - cap_h : 0);
        }
    }
    size_t q_start = r.pos;
    r.pos += per_query * d->fri_num_queries;
    size_t final_len = ((size_t)1 << (d->degree_bits - R * d->fri_arity_bits));
    gl2* final_poly = (gl2*)malloc(sizeof(gl2) * final_len);
    uint64_t pow_witness;
    r_u64s(&r, (uint64_t*)final_poly, 2 * final_len);
    r_u64s(&r, &pow_witness, 1);
    uint64_t n_pi = 0;
    r_bytes(&r, &n_pi, 8);
    uint64_t* pis = (uint64_t*)malloc(8 * (n_pi + 1));
    if (n_pi != d->num_public_inputs) r.bad = 1;
    else r_u64s(&r, pis, n_pi);
    if (r.bad || r.pos != len) { rc = -1; goto done; }

    {
        /* get_challenges */
        uint64_t pih[4];
        orc_hash_no_pad(pis, n_pi, pih);
        orc_challenger ch;
        orc_ch_init(&ch);
        orc_ch_observe_many(&ch, d->circuit_digest, 4);
        orc_ch_observe_many(&ch, pih, 4);
        orc_ch_observe_many(&ch, wires_cap, capw);
        uint64_t betas[4], gammas[4], alphas[4];
        for (uint32_t i = 0; i < nc; i++) betas[i] = orc_ch_challenge(&ch);
        for (uint32_t i = 0; i < nc; i++) gammas[i] = orc_ch_challenge(&ch);
        uint64_t deltas[16] = {0}, lut_polys[4 * 16] = {0};
        if (d->num_luts) {
            if (nc > 4 || d->num_luts > 16) { rc = -1; goto done; }
            for (uint32_t i = 0; i < nc; i++) { deltas[i] = betas[i]; deltas[nc + i] = gammas[i]; }
            for (uint32_t i = 0; i < 2 * nc; i++) deltas[2 * nc + i] = orc_ch_challenge(&ch);
            for (uint32_t ci = 0; ci < nc; ci++)
                for (uint32_t t = 0; t < d->num_luts; t++) lut_polys[ci * d->num_luts + t] = get_lut_poly(c, t, deltas + 4 * ci);
        }
        orc_ch_observe_many(&ch, zs_cap, capw);
        for (uint32_t i = 0; i < nc; i++) alphas[i] = orc_ch_challenge(&ch);
        orc_ch_observe_many(&ch, q_cap, capw);
        gl2 zeta = orc_ch_ext_challenge(&ch);
        /* observe openings: zeta batch = constants, sigmas, wires, zs, partial products, quotient; then zs_next */
        orc_ch_observe_many(&ch, (uint64_t*)o_consts, 2 * (c->n_cs + d->num_wires + nc));
        orc_ch_observe_many(&ch, (uint64_t*)o_pp, 2 * (nc * npp + c->n_q));
        orc_ch_observe_many(&ch, (uint64_t*)o_lk, 2 * nlk);
        orc_ch_observe_many(&ch, (uint64_t*)o_zs_next, 2 * nc);
        orc_ch_observe_many(&ch, (uint64_t*)o_lk_next, 2 * nlk);
        gl2 fri_alpha = orc_ch_ext_challenge(&ch);
        gl2 fri_betas[16];
        for (uint32_t k = 0; k < R; k++) {
            orc_ch_observe_many(&ch, fri_caps + k * capw, capw);
            fri_betas[k] = orc_ch_ext_challenge(&ch);
        }
        orc_ch_observe_many(&ch, (uint64_t*)final_poly, 2 * final_len);
        orc_ch_observe(&ch, pow_witness);
        uint64_t pow_response = orc_ch_challenge(&ch);
        uint64_t qidx[128];
        for (uint32_t q = 0; q < d->fri_num_queries; q++) qidx[q] = orc_ch_challenge(&ch) % L;

        /* ---- vanishing polynomial identity at zeta ---- */
        uint32_t nk = c->max_constraints;
        gl2* cons = (gl2*)calloc(2 * nk + 1, sizeof(gl2));
        gl2* gout = cons + nk;
        gl2 pih_e[4];
        for (int i = 0; i < 4; i++) pih_e[i] = gl2_from(pih[i]);
        for (uint32_t g = 0; g < d->num_gates; g++) {
            const orc_gate* gt = &c->gates[g];
            uint32_t k = gate_num_constraints_ext(gt);
            if (!k) continue;
            gl2 f = filter_ext(gt, o_consts[gt->selector_index], d->num_selectors > 1);
            gate_eval_ext(gt, o_consts + d->num_selectors + c->n_lk_sel, o_wires, pih_e, gout);
            for (uint32_t j = 0; j < k; j++) cons[j] = gl2_add(cons[j], gl2_mul(f, gout[j]));
        }
        const uint32_t t_lk = nc + nc * (npp + 1), t_gate = t_lk + nc * c->n_lk_terms;
        uint32_t n_terms = t_gate + nk;
        gl2* terms = (gl2*)malloc(sizeof(gl2) * n_terms);
        gl2 zeta_n = zeta;
        for (uint32_t i = 0; i < d->degree_bits; i++) zeta_n = gl2_mul(zeta_n, zeta_n);
        gl2 z_h = gl2_sub(zeta_n, gl2_from(1));
        /* eval_l_0(n, x) = (x^n - 1) / (n (x - 1)) */
        gl2 l0 = gl2_mul(z_h, gl2_inv(gl2_scale(gl2_sub(zeta, gl2_from(1)), (uint64_t)c->n % GL_P)));
        uint32_t n_chunks = (routed + chunk - 1) / chunk;
        for (uint32_t ci = 0; ci < nc; ci++) {
            terms[ci] = gl2_mul(l0, gl2_sub(o_zs[ci], gl2_from(1)));
            gl2 acc = o_zs[ci];
            for (uint32_t q = 0; q < n_chunks; q++) {
                gl2 num = gl2_from(1), den = gl2_from(1);
                for (uint32_t j = q * chunk; j < (q + 1) * chunk && j < routed; j++) {
                    gl2 s_id = gl2_scale(zeta, c->k_is[j]);
                    num = gl2_mul(num, gl2_add_base(gl2_add(o_wires[j], gl2_scale(s_id, betas[ci])), gammas[ci]));
                    den = gl2_mul(den, gl2_add_base(gl2_add(o_wires[j], gl2_scale(o_sigmas[j], betas[ci])), gammas[ci]));
                }
                gl2 new_acc = (q + 1 < n_chunks) ? o_pp[ci * npp + q] : o_zs_next[ci];
                terms[nc + ci * (npp + 1) + q] = gl2_sub(gl2_mul(acc, num), gl2_mul(new_acc, den));
                acc = new_acc;
            }
        }
        for (uint32_t ci = 0; ci < nc && nlk; ci++)
            lookup_constraints_ext(&c->lk, o_consts + d->num_selectors, o_wires, o_lk + ci * (1 + c->lk.n_sldc),
                                   o_lk_next + ci * (1 + c->lk.n_sldc), deltas + 4 * ci, lut_polys + ci * d->num_luts,
                                   terms + t_lk + ci * c->n_lk_terms);
        for (uint32_t k = 0; k < nk; k++) terms[t_gate + k] = cons[k];
        for (uint32_t ci = 0; ci < nc && rc == 1; ci++) {
            gl2 lhs = reduce_with_powers_ext(terms, n_terms, gl2_from(alphas[ci]));
            gl2 rhs = gl2_mul(z_h, reduce_with_powers_ext(o_q + ci * chunk, chunk, zeta_n));
            if (!gl2_eq(lhs, rhs)) rc = -2;
        }
        free(terms);
        free(cons);
        if (rc != 1) goto done;

        /* ---- FRI ---- */
        if (((pow_response >> (64 - d->fri_pow_bits)) != 0) && d->fri_pow_bits) { rc = -3; goto done; }
        /* PrecomputedReducedOpenings */
        gl2 red0 = gl2_from(0), red1 = gl2_from(0);
        {
            /* zeta batch order: constants_sigmas (oracle 0), wires, zs ++ partial products, quotient */
            gl2 apow = gl2_from(1);
            const gl2* seqs[6] = {o_consts, o_wires, o_zs, o_pp, o_q, o_lk};
            uint32_t lens[6] = {c->n_cs, d->num_wires, nc, nc * npp, c->n_q, nlk};
            for (int s = 0; s < 6; s++)
                for (uint32_t i = 0; i < lens[s]; i++) { red0 = gl2_add(red0, gl2_mul(apow, seqs[s][i])); apow = gl2_mul(apow, fri_alpha); }
            apow = gl2_from(1);
            for (uint32_t i = 0; i < nc; i++) { red1 = gl2_add(red1, gl2_mul(apow, o_zs_next[i])); apow = gl2_mul(apow, fri_alpha); }
            for (uint32_t i = 0; i < nlk; i++) { red1 = gl2_add(red1, gl2_mul(apow, o_lk_next[i])); apow = gl2_mul(apow, fri_alpha); }
        }
        gl2 g_zeta = gl2_scale(zeta, gl_root_of_unity(c->log_n));
        gl2 alpha_pow_nc = gl2_pow(fri_alpha, nc + nlk);
        /* fri_combine_initial walks the batch's polynomial list: Zs columns n_zpp.. (the lookup polynomials) come after the
         * quotient chunks in the zeta batch and after the Zs in the zeta_next batch, so the Zs row is kept for later */
        uint64_t* zrow = (uint64_t*)malloc(8 * (c->n_zs + 1));
        const uint64_t* init_caps[4] = {c->cs_cap, wires_cap, zs_cap, q_cap};
        uint64_t* row = (uint64_t*)malloc(8 * (d->num_wires + c->n_cs + c->n_zs + c->n_q + 8));
        rbuf qr = {proof, len, q_start, 0};
        for (uint32_t q = 0; q < d->fri_num_queries && rc == 1; q++) {
            size_t x_index = qidx[q];
            /* fri_verify_initial_proof + fri_combine_initial */
            gl2 sum0 = gl2_from(0), apow = gl2_from(1), sum1 = gl2_from(0);
            for (int o = 0; o < 4; o++) {
                r_u64s(&qr, row, cols[o]);
                uint8_t plen;
                r_bytes(&qr, &plen, 1);
                uint64_t sib[64 * 4];
                if (plen != c->log_L - cap_h) { rc = -4; break; }
                r_u64s(&qr, sib, 4 * plen);
                if (!orc_merkle_verify(row, cols[o], x_index, sib, plen, init_caps[o], cap_h)) { rc = -5; break; }
                uint32_t take = o == 2 ? c->n_zpp : cols[o];
                for (uint32_t i = 0; i < take; i++) { sum0 = gl2_add(sum0, gl2_scale(apow, row[i])); apow = gl2_mul(apow, fri_alpha); }
                if (o == 2) {
                    memcpy(zrow, row, 8 * c->n_zs);
                    gl2 ap = gl2_from(1);
                    for (uint32_t i = 0; i < nc; i++) { sum1 = gl2_add(sum1, gl2_scale(ap, row[i])); ap = gl2_mul(ap, fri_alpha); }
                    for (uint32_t i = 0; i < nlk; i++) { sum1 = gl2_add(sum1, gl2_scale(ap, row[c->n_zpp + i])); ap = gl2_mul(ap, fri_alpha); }
                }
            }
            if (rc != 1) break;
            for (uint32_t i = 0; i < nlk; i++) { sum0 = gl2_add(sum0, gl2_scale(apow, zrow[c->n_zpp + i])); apow = gl2_mul(apow, fri_alpha); }
            uint64_t subgroup_x = gl_mul(GL_GEN, gl_pow(gl_root_of_unity(c->log_L), gl_bitrev(x_index, c->log_L)));
            gl2 sx = gl2_from(subgroup_x);
            gl2 old_eval = gl2_mul(gl2_sub(sum0, red0), gl2_inv(gl2_sub(sx, zeta)));
            old_eval = gl2_mul(old_eval, alpha_pow_nc); /* alpha.shift after batch 1's reduce (count = nc) */
            old_eval = gl2_add(old_eval, gl2_mul(gl2_sub(sum1, red1), gl2_inv(gl2_sub(sx, g_zeta))));
            size_t nl = L;
            for (uint32_t k = 0; k < R; k++) {
                nl >>= d->fri_arity_bits;
                gl2 evals[64];
                r_u64s(&qr, (uint64_t*)evals, 2 * arity);
                uint8_t plen;
                r_bytes(&qr, &plen, 1);
                uint64_t sib[64 * 4];
                unsigned lg = gl_log2_strict(nl);
                if (plen != (lg > cap_h ? lg - cap_h : 0)) { rc = -6; break; }
                r_u64s(&qr, sib, 4 * plen);
                size_t coset_index = x_index >> d->fri_arity_bits;
                size_t within = x_index & (arity - 1);
                if (!gl2_eq(evals[within], old_eval)) { rc = -7; break; }
                /* compute_evaluation: interpolate {(coset_start * g^i, evals_br[i])} and evaluate at beta */
                uint64_t gA = gl_root_of_unity(d->fri_arity_bits);
                size_t rev_within = gl_bitrev(within, d->fri_arity_bits);
                uint64_t coset_start = gl_mul(subgroup_x, gl_pow(gA, arity - rev_within));
                gl2 res = gl2_from(0);
                for (uint32_t i = 0; i < arity; i++) {
                    /* Lagrange basis at beta over points p_j = coset_start * g^j */
                    gl2 yi = evals[gl_bitrev(i, d->fri_arity_bits)];
                    uint64_t pi_ = gl_mul(coset_start, gl_pow(gA, i));
                    gl2 numr = gl2_from(1);
                    uint64_t den = 1;
                    for (uint32_t j = 0; j < arity; j++) {
                        if (j == i) continue;
                        uint64_t pj = gl_mul(coset_start, gl_pow(gA, j));
                        numr = gl2_mul(numr, gl2_sub(fri_betas[k], gl2_from(pj)));
                        den = gl_mul(den, gl_sub(pi_, pj));
                    }
                    res = gl2_add(res, gl2_mul(yi, gl2_scale(numr, gl_inv(den))));
                }
                old_eval = res;
                if (!orc_merkle_verify((uint64_t*)evals, 2 * arity, coset_index, sib, plen, fri_caps + k * capw, cap_h)) { rc = -8; break; }
                subgroup_x = gl_exp_pow2(subgroup_x, d->fri_arity_bits);
                x_index = coset_index;
            }
            if (rc != 1) break;
            gl2 fe = gl2_from(0), sxe = gl2_from(subgroup_x);
            for (size_t i = final_len; i-- > 0;) fe = gl2_add(gl2_mul(fe, sxe), final_poly[i]);
            if (!gl2_eq(fe, old_eval)) rc = -9;
        }
        if (qr.bad) rc = -10;
        free(row); free(zrow);
    }
done:
    free(caps); free(o_consts); free(final_poly); free(pis);
    return rc;
}
