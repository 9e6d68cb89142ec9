// Host orchestration of a whole plonky2 proof on one MI355X (C ABI: nlx_circuit_build, nlx_prove).
//
// Replaces plonky2::plonk::prover::prove_with_partition_witness (after witness generation),
// plonky2::plonk::circuit_builder::CircuitBuilder::build's constants/sigmas commitment and
// util::serialization's ProofWithPublicInputs::to_bytes, reached
// from nearx/src/test_utils.rs:29 (build) and :62 (prove).
//
// Division of labour: every O(n) stage runs on the GPU with tables
// resident in HBM; the Fiat-Shamir transcript is ~120 strictly sequential Poseidon permutations
// over a few KB and runs on the host, as it does in the reference - only
// Merkle caps (512 B), openings (~4.5 KB), the final polynomial and the query answers cross
// PCIe.  It is not a fallback for any device stage.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>
#include "commit.hpp"
#include "gl.hpp"
#include "poly.hpp"
#include "poseidon.hpp"
#include "prover.hpp"
#include "fri.hpp"
#include "transcript.hpp"

using namespace nlx;

namespace {
uint32_t fri_num_rounds(const nlx_circuit_desc& d) {
    return nlx::fri_num_rounds(d.degree_bits, d.rate_bits, d.cap_height, d.fri_arity_bits, d.fri_final_poly_bits);
}

// plonky2 Gate::num_constraints() of every gate kind k_quotient evaluates: the alpha-power table of the quotient stage
// must hold one power per constraint of the widest gate (the circuit's `num_gate_constraints`).
uint32_t gate_num_constraints(const nlx_gate_desc& g) {
    const uint32_t p0 = g.param0, p1 = g.param1;
    switch (g.kind) {
        case NLX_GATE_CONSTANT: case NLX_GATE_ARITHMETIC: return p0;
        case NLX_GATE_PUBLIC_INPUT: return 4;
        case NLX_GATE_BASE_SUM: return 1 + p1;                      // the sum + one range product per limb
        case NLX_GATE_POSEIDON: return 123;                         // 1 swap bit + 4 deltas + 3*12 + 22 + 4*12 + 12 outputs
        case NLX_GATE_ARITHMETIC_EXT: case NLX_GATE_MUL_EXT: case NLX_GATE_REDUCING: case NLX_GATE_REDUCING_EXT:
            return 2 * p0;                                          // D = 2 base constraints per extension constraint
        case NLX_GATE_POSEIDON_MDS: return 24;
        case NLX_GATE_EXPONENTIATION: return p0 + 1;
        case NLX_GATE_RANDOM_ACCESS: return (p1 & 0xFFFF) * (p0 + 2) + (p1 >> 16);
        case NLX_GATE_COSET_INTERPOLATION: return 2 * (2 + 2 * (((1u << p0) - 2) / (p1 - 1)));
        case NLX_GATE_U32_ADD_MANY: return p1 * (3 + 18);
        case NLX_GATE_U32_ARITHMETIC: return p0 * (4 + 32);
        case NLX_GATE_U32_SUBTRACTION: return p0 * (3 + 16);
        case NLX_GATE_U32_RANGE_CHECK: return p0 * 17;
        case NLX_GATE_COMPARISON: return 6 + 5 * p1 + (p0 + p1 - 1) / p1;
        default: return 0;
    }
}

// Cost of one gate's evaluation at one point, in microseconds of k_quotient's time per 2^19 points when every wave of a
// tile evaluates that item alone, less the 234 us an empty item list takes (measured on MI355X with the calibration build,
// tools/quotient_calibrate.sh, standard parameters - round 3's kernel: profiles/r03_quotient_calibrate.txt; scaled linearly in
// the parameter that multiplies the work).  Only the RATIOS matter: they balance the work split.
uint32_t gate_eval_cost(const nlx_gate_desc& g) {
    const uint32_t p0 = g.param0, p1 = g.param1;
    switch (g.kind) {
        case NLX_GATE_ARITHMETIC: return 10 + 6 * p0;                      // 130 for 20 operations
        case NLX_GATE_BASE_SUM: return 10 + 95 * p1 * (1 + p0) / 30;      // 610 for 63 limbs in base 2
        case NLX_GATE_POSEIDON: return 2540;                                // evaluated in three parts, see POSEIDON_PARTS
        case NLX_GATE_ARITHMETIC_EXT: return 10 + 16 * p0;                 // 170
        case NLX_GATE_MUL_EXT: return 10 + 13 * p0;                        // 180
        case NLX_GATE_REDUCING: return 10 + 165 * p0 / 10;                 // 720
        case NLX_GATE_REDUCING_EXT: return 10 + 160 * p0 / 10;             // 520
        case NLX_GATE_POSEIDON_MDS: return 120;
        case NLX_GATE_EXPONENTIATION: return 10 + 80 * p0 / 10;            // 540
        case NLX_GATE_RANDOM_ACCESS: return 10 + 128 * (p1 & 0xFFFF) * (1u << p0) / 16;   // 520
        case NLX_GATE_COSET_INTERPOLATION: return 10 + (34u << p0);        // 550
        case NLX_GATE_U32_ADD_MANY: return 10 + 136 * p1;                  // 690
        case NLX_GATE_U32_ARITHMETIC: return 10 + 250 * p0;                // 760
        case NLX_GATE_U32_SUBTRACTION: return 10 + 132 * p0;               // 800
        case NLX_GATE_U32_RANGE_CHECK: return 10 + 109 * p0;               // 770
        case NLX_GATE_COMPARISON: return 10 + 27 * p1;                     // 440
        default: return 10;
    }
}
// PoseidonGate's three parts (prover_kernels.hip gate_poseidon): part mask, share of the gate's cost in percent
constexpr uint32_t POSEIDON_PARTS[3][2] = {{1, 17}, {2, 59}, {4, 24}};   // measured 440 / 1 490 / 610
// Part 2 (the 22 partial rounds) runs in a kernel of its own before k_quotient (launch_quotient_poseidon: the permutation's
// fused-block schedule, prover_kernels.hip) unless NLX_QUOTIENT_POSEIDON_INLINE=1 keeps it among k_quotient's items.
bool poseidon_part2_separate_default() {
    const char* e = getenv("NLX_QUOTIENT_POSEIDON_INLINE");
    return !(e && e[0] == '1');
}
}  // namespace

struct nlx_circuit {
    nlx_ctx* ctx = nullptr;
    nlx_circuit_desc d{};
    std::vector<nlx_gate_desc> gates;
    std::vector<uint64_t> k_is;
    uint32_t n_consts_all = 0, n_cs = 0, n_zs = 0, n_q = 0, n_fri_rounds = 0, n_terms = 0, max_gate_constraints = 0;
    nlx_commit* cs = nullptr;           // constants + sigmas commitment
    std::vector<uint64_t> cs_cap;       // host copy
    uint64_t* d_sigma_values = nullptr; // [routed][n]
    GateDev* d_gates = nullptr;
    uint32_t* d_work = nullptr;         // k_quotient's work split: [quotient_waves()][work_stride]
    bool poseidon_part2_separate = poseidon_part2_separate_default();
    std::vector<uint32_t> poseidon_gates;   // PoseidonGates whose part 2 runs in k_quotient_poseidon
    uint32_t work_stride = 0;
    uint64_t* d_small = nullptr;        // k_is | coset_base | zh_inv | w_R_inv_pows | chunk_scale | w_A_inv_pows
    uint64_t *d_k_is = nullptr, *d_coset_base = nullptr, *d_zh_inv = nullptr, *d_wR_inv = nullptr,
             *d_chunk_scale = nullptr, *d_wA_inv = nullptr;
    uint64_t* d_l0_scaled = nullptr;
    const uint64_t* d_inv_scale_br = nullptr;  // ctx-owned
    // lookup tables (all empty / zero without them)
    uint32_t n_zpp = 0;        // Zs + partial products columns; n_zs = n_zpp + n_lk_polys
    uint32_t n_lk_sel = 0;     // lookup selector columns (4 + tables), between the gate selectors and the gate constants
    uint32_t n_lk_polys = 0;   // num_challenges * (1 + S)
    uint32_t n_lk_terms = 0;   // 4 + tables + 2 S vanishing terms per challenge round
    LookupShape lk{};
    std::vector<uint32_t> lut_sizes, lut_offsets, lookup_rows, lut_num_lookups;
    std::vector<uint16_t> lut_pairs;
    std::vector<LookupTableDev> h_tabs;
    LookupTableDev* d_tabs = nullptr;
    uint32_t* d_lut_pairs = nullptr;  // all tables, input | output << 16
    int32_t* d_idx_of = nullptr;      // [tables][65536]
    uint32_t* d_mult = nullptr;       // multiplicity counters of all tables, then the error word
    size_t mult_words = 0;
    // stage timing
    hipEvent_t ev[NLX_MAX_STAGES + 1]{};
    const char* stage_names[NLX_MAX_STAGES]{};
    uint32_t n_stages = 0;
    bool timed = false;
    size_t n() const { return (size_t)1 << d.degree_bits; }
    size_t L() const { return (size_t)1 << (d.degree_bits + d.rate_bits); }
};

extern "C" {

int32_t nlx_circuit_build(nlx_ctx* ctx, const nlx_circuit_desc* desc, const uint64_t* constants,
                          const uint64_t* sigmas, nlx_circuit** out) NLX_TRY {
    if (!ctx) return NLX_E_INVAL;
    if (!desc || !constants || !sigmas || !out || !desc->gates || !desc->k_is)
        return ctx->fail(NLX_E_INVAL, "NULL argument");
    *out = nullptr;
    const nlx_circuit_desc& d = *desc;
    if (d.num_challenges < 1 || d.num_challenges > 2) return ctx->fail(NLX_E_UNSUPPORTED, "num_challenges must be 1 or 2");
    if (d.rate_bits < 1 || d.rate_bits > 3) return ctx->fail(NLX_E_UNSUPPORTED, "rate_bits must be in [1, 3]");
    if ((1u << d.rate_bits) != d.quotient_degree_factor)
        return ctx->fail(NLX_E_UNSUPPORTED, "quotient_degree_factor must equal 2^rate_bits");
    if (d.fri_arity_bits < 2 || d.fri_arity_bits > 4) return ctx->fail(NLX_E_UNSUPPORTED, "fri_arity_bits must be in [2, 4]");
    if (d.degree_bits < d.fri_arity_bits || d.degree_bits + d.rate_bits > 30) return ctx->fail(NLX_E_RANGE, "degree_bits out of range");
    if (d.num_routed_wires > d.num_wires || d.num_routed_wires == 0) return ctx->fail(NLX_E_INVAL, "bad wire counts");
    const uint32_t n_chunks = (d.num_routed_wires + d.quotient_degree_factor - 1) / d.quotient_degree_factor;
    if (n_chunks > 10 || n_chunks != d.num_partial_products + 1) return ctx->fail(NLX_E_INVAL, "num_partial_products mismatch");
    if (d.fri_num_queries > 128 || d.cap_height > 6) return ctx->fail(NLX_E_RANGE, "FRI parameters out of range");
    for (uint32_t g = 0; g < d.num_gates; g++) {
        const nlx_gate_desc& gt = d.gates[g];
        if (gt.kind > NLX_GATE_KIND_MAX) return ctx->fail(NLX_E_UNSUPPORTED, "gate kind %u not supported", gt.kind);
        if (gt.kind == NLX_GATE_ARITHMETIC_EXT && (8 * gt.param0 > d.num_wires || d.num_constants < 2)) return ctx->fail(NLX_E_INVAL, "ArithmeticExtensionGate too wide");
        if (gt.kind == NLX_GATE_MUL_EXT && 6 * gt.param0 > d.num_wires) return ctx->fail(NLX_E_INVAL, "MulExtensionGate too wide");
        if (gt.kind == NLX_GATE_REDUCING && (gt.param0 < 1 || 6 + gt.param0 + 2 * (gt.param0 - 1) > d.num_wires)) return ctx->fail(NLX_E_INVAL, "ReducingGate too wide");
        if (gt.kind == NLX_GATE_POSEIDON_MDS && d.num_wires < 48) return ctx->fail(NLX_E_INVAL, "PoseidonMdsGate needs 48 wires");
        if (gt.kind == NLX_GATE_EXPONENTIATION && (gt.param0 < 1 || 2 + 2 * gt.param0 > d.num_wires)) return ctx->fail(NLX_E_INVAL, "ExponentiationGate too wide");
        if (gt.kind == NLX_GATE_U32_ADD_MANY && (gt.param0 < 1 || gt.param0 > 15 || gt.param1 < 1 || (gt.param0 + 3 + 18) * gt.param1 > d.num_wires))
            return ctx->fail(NLX_E_INVAL, "U32AddManyGate does not fit the wires");
        if (gt.kind == NLX_GATE_U32_ARITHMETIC && (gt.param0 < 1 || (6 + 32) * gt.param0 > d.num_wires)) return ctx->fail(NLX_E_INVAL, "U32ArithmeticGate too wide");
        if (gt.kind == NLX_GATE_U32_SUBTRACTION && (gt.param0 < 1 || (5 + 16) * gt.param0 > d.num_wires)) return ctx->fail(NLX_E_INVAL, "U32SubtractionGate too wide");
        if (gt.kind == NLX_GATE_U32_RANGE_CHECK && (gt.param0 < 1 || 17 * gt.param0 > d.num_wires)) return ctx->fail(NLX_E_INVAL, "U32RangeCheckGate too wide");
        if (gt.kind == NLX_GATE_COMPARISON) {
            const uint32_t cb = gt.param1 ? (gt.param0 + gt.param1 - 1) / gt.param1 : 0;
            if (gt.param1 < 1 || cb < 1 || cb > 3 || cb * gt.param1 > 62 || 4 + 5 * gt.param1 + cb + 1 > d.num_wires)
                return ctx->fail(NLX_E_INVAL, "ComparisonGate: chunk_bits must be in [1, 3] and the gate must fit the wires");
        }
        if (gt.kind == NLX_GATE_COSET_INTERPOLATION) {
            const uint32_t np = gt.param0 <= 5 ? 1u << gt.param0 : 0;
            if (np < 4 || gt.param1 < 2 || gt.param1 > np || d.degree_bits < gt.param0 + 1 ||
                1 + 2 * np + 4 + 4 * ((np - 2) / (gt.param1 - 1)) + 2 > d.num_wires)
                return ctx->fail(NLX_E_INVAL, "CosetInterpolationGate does not fit");
        }
        if (gt.kind == NLX_GATE_RANDOM_ACCESS) {
            const uint32_t bits = gt.param0, copies = gt.param1 & 0xFFFF, extra = gt.param1 >> 16;
            if (bits < 1 || bits > 6 || copies < 1 || extra > d.num_constants ||
                (2 + (1u << bits)) * copies + extra + copies * bits > d.num_wires)
                return ctx->fail(NLX_E_INVAL, "RandomAccessGate does not fit the wires");
        }
        if (gt.kind == NLX_GATE_REDUCING_EXT && (gt.param0 < 1 || 6 + 2 * gt.param0 + 2 * (gt.param0 - 1) > d.num_wires)) return ctx->fail(NLX_E_INVAL, "ReducingExtensionGate too wide");
        if (gt.selector_index >= d.num_selectors || gt.group_end > d.num_gates || gt.group_start > gt.group_end)
            return ctx->fail(NLX_E_INVAL, "gate %u: bad selector group", g);
        if (gt.kind == NLX_GATE_POSEIDON && d.num_wires < 135) return ctx->fail(NLX_E_INVAL, "PoseidonGate needs 135 wires");
        if (gt.kind == NLX_GATE_ARITHMETIC && 4 * gt.param0 > d.num_wires) return ctx->fail(NLX_E_INVAL, "ArithmeticGate too wide");
        if (gt.kind == NLX_GATE_BASE_SUM && 1 + gt.param1 > d.num_wires) return ctx->fail(NLX_E_INVAL, "BaseSumGate too wide");
        if (gt.kind == NLX_GATE_CONSTANT && gt.param0 > d.num_constants) return ctx->fail(NLX_E_INVAL, "ConstantGate too wide");
    }
    if (d.num_luts) {
        // CommonCircuitData::luts + ProverOnlyCircuitData::lookup_rows: shapes as CircuitBuilder::add_all_lookups lays them out
        if (d.num_luts > 16) return ctx->fail(NLX_E_RANGE, "at most 16 lookup tables");
        if (!d.lut_sizes || !d.lut_pairs || !d.lookup_rows || !d.lut_num_lookups) return ctx->fail(NLX_E_INVAL, "NULL lookup table array");
        const uint32_t n_lu = d.num_routed_wires / 2, n_lut = d.num_routed_wires / 3;
        if (n_lut < 1 || d.quotient_degree_factor < 2) return ctx->fail(NLX_E_INVAL, "too few routed wires for the lookup gates");
        for (uint32_t t = 0; t < d.num_luts; t++) {
            const uint32_t len = d.lut_sizes[t], lookups = d.lut_num_lookups[t];
            const uint32_t last_lu = d.lookup_rows[3 * t], last_lut = d.lookup_rows[3 * t + 1], first_lut = d.lookup_rows[3 * t + 2];
            if (len < 1 || len > 65536 || lookups < 1) return ctx->fail(NLX_E_RANGE, "table %u: 1 .. 65536 entries and at least one lookup", t);
            if (!(last_lu < last_lut && last_lut <= first_lut) || (uint64_t)first_lut + 1 >= ((uint64_t)1 << d.degree_bits))
                return ctx->fail(NLX_E_INVAL, "table %u: lookup rows out of order or past the circuit", t);
            if (last_lut - last_lu != (lookups + n_lu - 1) / n_lu || first_lut - last_lut + 1 != (len + n_lut - 1) / n_lut)
                return ctx->fail(NLX_E_INVAL, "table %u: row counts do not match its lookups / entries", t);
        }
    }
    (void)hipSetDevice(ctx->device);
    nlx_circuit* c = new (std::nothrow) nlx_circuit();
    if (!c) return ctx->fail(NLX_E_NOMEM, "host allocation failed");
    c->ctx = ctx;
    c->d = d;
    c->gates.assign(d.gates, d.gates + d.num_gates);
    for (const nlx_gate_desc& gt : c->gates) c->max_gate_constraints = std::max(c->max_gate_constraints, gate_num_constraints(gt));
    c->k_is.assign(d.k_is, d.k_is + d.num_routed_wires);
    c->d.gates = c->gates.data();
    c->d.k_is = c->k_is.data();
    if (d.num_luts) {
        const uint32_t T = d.num_luts;
        c->lk.num_luts = T;
        c->lk.n_lu_slots = d.num_routed_wires / 2;
        c->lk.n_lut_slots = d.num_routed_wires / 3;
        c->lk.lu_degree = d.quotient_degree_factor - 1;
        c->lk.n_sldc = (c->lk.n_lu_slots + c->lk.lu_degree - 1) / c->lk.lu_degree;
        c->lk.lut_degree = (c->lk.n_lut_slots + c->lk.n_sldc - 1) / c->lk.n_sldc;
        c->n_lk_sel = 4 + T;
        c->n_lk_polys = d.num_challenges * (1 + c->lk.n_sldc);
        c->n_lk_terms = 4 + T + 2 * c->lk.n_sldc;
        c->lut_sizes.assign(d.lut_sizes, d.lut_sizes + T);
        c->lut_num_lookups.assign(d.lut_num_lookups, d.lut_num_lookups + T);
        c->lookup_rows.assign(d.lookup_rows, d.lookup_rows + 3 * T);
        c->lut_offsets.assign(T + 1, 0);
        for (uint32_t t = 0; t < T; t++) c->lut_offsets[t + 1] = c->lut_offsets[t] + c->lut_sizes[t];
        c->lut_pairs.assign(d.lut_pairs, d.lut_pairs + 2 * (size_t)c->lut_offsets[T]);
        c->d.lut_sizes = c->lut_sizes.data(); c->d.lut_num_lookups = c->lut_num_lookups.data();
        c->d.lookup_rows = c->lookup_rows.data(); c->d.lut_pairs = c->lut_pairs.data();
    }
    c->n_consts_all = d.num_selectors + c->n_lk_sel + d.num_constants;
    c->n_cs = c->n_consts_all + d.num_routed_wires;
    c->n_zpp = d.num_challenges * (1 + d.num_partial_products);
    c->n_zs = c->n_zpp + c->n_lk_polys;
    c->n_q = d.num_challenges * d.quotient_degree_factor;
    c->n_fri_rounds = fri_num_rounds(d);
    const size_t n = c->n(), L = c->L();
    const unsigned log_n = d.degree_bits, log_L = d.degree_bits + d.rate_bits;
    const uint32_t R = 1u << d.rate_bits, A = 1u << d.fri_arity_bits;
    int32_t rc = ctx->ensure_tables(log_L);
    auto fail = [&](int32_t code) {
        nlx_circuit_destroy(c);
        return code;
    };
    if (rc) return fail(rc);
    rc = ctx->get_coset_scale(log_n, d.rate_bits, &c->d_inv_scale_br, true);
    if (rc) return fail(rc);

    // constants ++ sigmas -> one device matrix -> commitment
    {
        uint64_t* d_vals = (uint64_t*)ctx->alloc((size_t)c->n_cs * n * 8);
        if (!d_vals) return fail(NLX_E_NOMEM);
        const size_t cb = (size_t)c->n_consts_all * n * 8, sb = (size_t)d.num_routed_wires * n * 8;
        hipError_t e1 = hipMemcpyAsync(d_vals, constants, cb, is_device_ptr(constants) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream);
        hipError_t e2 = hipMemcpyAsync((uint8_t*)d_vals + cb, sigmas, sb, is_device_ptr(sigmas) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream);
        if (e1 != hipSuccess || e2 != hipSuccess) { ctx->release(d_vals); return fail(ctx->hip_fail(e1 != hipSuccess ? e1 : e2, "hipMemcpyAsync")); }
        rc = commit_build(ctx, d_vals, n, CommitInput::ValuesNatural, c->n_cs, log_n, d.rate_bits, d.cap_height, &c->cs);
        if (rc) { ctx->release(d_vals); return fail(rc); }
        c->d_sigma_values = (uint64_t*)ctx->alloc(sb);
        if (!c->d_sigma_values) { ctx->release(d_vals); return fail(NLX_E_NOMEM); }
        (void)hipMemcpyAsync(c->d_sigma_values, (uint8_t*)d_vals + cb, sb, hipMemcpyDeviceToDevice, ctx->stream);
        ctx->release(d_vals);
    }
    // small tables
    {
        const uint32_t routed = d.num_routed_wires;
        std::vector<uint64_t> small(routed + 3 * R + R + A);
        uint64_t* h_k = small.data();
        uint64_t* h_cb = h_k + routed;
        uint64_t* h_zh = h_cb + R;
        uint64_t* h_wR = h_zh + R;
        uint64_t* h_cs = h_wR + R;
        uint64_t* h_wA = h_cs + R;
        memcpy(h_k, c->k_is.data(), routed * 8);
        const uint64_t w_L = gl::root_of_unity(log_L), w_R = gl::root_of_unity(d.rate_bits);
        const uint64_t g_n = gl::exp_pow2(gl::GEN, log_n);
        const uint64_t g_n_inv = gl::inv(g_n), R_inv = gl::inv((uint64_t)R);
        for (uint32_t r = 0; r < R; r++) {
            h_cb[r] = gl::mul(gl::GEN, gl::pow(w_L, r));
            h_zh[r] = gl::inv(gl::sub(gl::mul(g_n, gl::pow(w_R, r)), 1));  // 1 / (x^n - 1) on coset r
            h_wR[r] = gl::pow(gl::inv(w_R), r);
            h_cs[r] = gl::mul(gl::pow(g_n_inv, r), R_inv);
        }
        const uint64_t w_A_inv = gl::inv(gl::root_of_unity(d.fri_arity_bits));
        for (uint32_t i = 0; i < A; i++) h_wA[i] = gl::pow(w_A_inv, i);
        c->d_small = (uint64_t*)ctx->alloc(small.size() * 8);
        c->d_gates = (GateDev*)ctx->alloc(sizeof(GateDev) * (d.num_gates ? d.num_gates : 1));
        c->d_l0_scaled = (uint64_t*)ctx->alloc(L * 8);
        if (!c->d_small || !c->d_gates || !c->d_l0_scaled) return fail(NLX_E_NOMEM);
        static_assert(sizeof(GateDev) == sizeof(nlx_gate_desc), "gate descriptor layout");
        hipError_t e = hipMemcpy(c->d_small, small.data(), small.size() * 8, hipMemcpyHostToDevice);
        if (e == hipSuccess && d.num_gates) e = hipMemcpy(c->d_gates, c->gates.data(), sizeof(GateDev) * d.num_gates, hipMemcpyHostToDevice);
        if (e != hipSuccess) return fail(ctx->hip_fail(e, "hipMemcpy(tables)"));
        c->d_k_is = c->d_small;
        c->d_coset_base = c->d_k_is + routed;
        c->d_zh_inv = c->d_coset_base + R;
        c->d_wR_inv = c->d_zh_inv + R;
        c->d_chunk_scale = c->d_wR_inv + R;
        c->d_wA_inv = c->d_chunk_scale + R;
        launch_l0_table(ctx->stream, c->d_l0_scaled, log_n, d.rate_bits, c->d_coset_base, ctx->tables.fwd[log_n]);
    }
    if (d.num_luts) {
        // the tables on the device: packed pairs, the input -> index map of set_lookup_wires (HashMap collect: the last entry
        // with an input wins), multiplicity counters
        const uint32_t T = d.num_luts, total = c->lut_offsets[T];
        std::vector<uint32_t> packed(total);
        std::vector<int32_t> idx_of((size_t)T << 16, -1);
        for (uint32_t t = 0; t < T; t++)
            for (uint32_t i = 0; i < c->lut_sizes[t]; i++) {
                const uint16_t* pr = &c->lut_pairs[2 * (size_t)(c->lut_offsets[t] + i)];
                packed[c->lut_offsets[t] + i] = (uint32_t)pr[0] | ((uint32_t)pr[1] << 16);
                idx_of[((size_t)t << 16) + pr[0]] = (int32_t)i;
            }
        c->mult_words = total;
        c->d_lut_pairs = (uint32_t*)ctx->alloc((size_t)total * 4);
        c->d_idx_of = (int32_t*)ctx->alloc(idx_of.size() * 4);
        c->d_mult = (uint32_t*)ctx->alloc(((size_t)total + 1) * 4);
        c->d_tabs = (LookupTableDev*)ctx->alloc(sizeof(LookupTableDev) * T);
        if (!c->d_lut_pairs || !c->d_idx_of || !c->d_mult || !c->d_tabs) return fail(NLX_E_NOMEM);
        c->h_tabs.resize(T);
        for (uint32_t t = 0; t < T; t++) {
            LookupTableDev& lt = c->h_tabs[t];
            lt.len = c->lut_sizes[t]; lt.lookups = c->lut_num_lookups[t];
            lt.last_lu = c->lookup_rows[3 * t]; lt.last_lut = c->lookup_rows[3 * t + 1]; lt.first_lut = c->lookup_rows[3 * t + 2];
            lt.pad_ = 0;
            lt.pairs = c->d_lut_pairs + c->lut_offsets[t];
            lt.idx_of = c->d_idx_of + ((size_t)t << 16);
            lt.mult = c->d_mult + c->lut_offsets[t];
        }
        hipError_t e = hipMemcpy(c->d_lut_pairs, packed.data(), packed.size() * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(c->d_idx_of, idx_of.data(), idx_of.size() * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(c->d_tabs, c->h_tabs.data(), sizeof(LookupTableDev) * T, hipMemcpyHostToDevice);
        if (e != hipSuccess) return fail(ctx->hip_fail(e, "hipMemcpy(lookup tables)"));
    }
    {
        // k_quotient's work split: items = every gate + the permutation argument of each challenge; longest item first into
        // the least loaded wave.  Wave w and wave w + QW/2 of a block share a SIMD, so the bins are then paired heavy with
        // light: what has to balance is the load per SIMD (the same split runs on every tile).
        const uint32_t QWv = quotient_waves();
        if (d.num_gates + d.num_challenges > 0xFFFFu) return fail(ctx->fail(NLX_E_RANGE, "too many gates"));
        std::vector<std::pair<uint32_t, uint32_t>> items;  // (cost, id)
        uint32_t n_words = 0;
        for (uint32_t g = 0; g < d.num_gates; g++) {
            const uint32_t cost = gate_eval_cost(c->gates[g]);
            if (c->gates[g].kind == NLX_GATE_POSEIDON) {
                for (const auto& part : POSEIDON_PARTS) {
                    if (part[0] == 2 && c->poseidon_part2_separate) {
                        c->poseidon_gates.push_back(g);
                        continue;
                    }
                    items.push_back({cost * part[1] / 100, g | (part[0] << 16)});
                }
            } else {
                items.push_back({cost, g});
            }
        }
        for (uint32_t ch = 0; ch < d.num_challenges; ch++) items.push_back({10 + 45 * d.num_routed_wires / 4, d.num_gates + ch});  // 910 for 80 routed wires
        n_words = (uint32_t)items.size();
        std::sort(items.begin(), items.end(), [](const auto& a, const auto& b) { return a.first != b.first ? a.first > b.first : a.second < b.second; });
        std::vector<std::vector<uint32_t>> bins(QWv);
        std::vector<uint64_t> load(QWv, 0);
        for (const auto& it : items) {
            const uint32_t b = (uint32_t)(std::min_element(load.begin(), load.end()) - load.begin());
            bins[b].push_back(it.second);
            load[b] += it.first;
        }
        std::vector<uint32_t> order(QWv);
        for (uint32_t i = 0; i < QWv; i++) order[i] = i;
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return load[a] != load[b] ? load[a] > load[b] : a < b; });
        c->work_stride = n_words + 1;
        std::vector<uint32_t> work((size_t)QWv * c->work_stride, 0xFFFFFFFFu);
        for (uint32_t i = 0; i < QWv; i++) {
            // heaviest bins on waves 0 .. QW/2-1, lightest first on waves QW/2 .. QW-1: SIMD s gets bins order[s] and order[QW-1-s]
            const uint32_t wave = i < QWv / 2 ? i : QWv / 2 + (QWv - 1 - i);
            std::copy(bins[order[i]].begin(), bins[order[i]].end(), work.begin() + (size_t)wave * c->work_stride);
        }
#ifdef NLX_QUOTIENT_CALIBRATE
        // kernel-tuning build only (build.py, NLX_BUILD_VARIANT): every wave evaluates the ONE item named by NLX_Q_ONLY, so
        // that the stage time ranks the items' real costs (tools/quotient_calibrate.sh); the proof is wrong by design
        if (const char* only = getenv("NLX_Q_ONLY")) {
            std::fill(work.begin(), work.end(), 0xFFFFFFFFu);
            for (uint32_t w = 0; w < QWv; w++) work[(size_t)w * c->work_stride] = (uint32_t)atoi(only);
        }
#endif
        c->d_work = (uint32_t*)ctx->alloc(work.size() * 4);
        if (!c->d_work) return fail(NLX_E_NOMEM);
        hipError_t e = hipMemcpy(c->d_work, work.data(), work.size() * 4, hipMemcpyHostToDevice);
        if (e != hipSuccess) return fail(ctx->hip_fail(e, "hipMemcpy(work split)"));
        if (quotient_lds_bytes(d.num_wires, c->n_consts_all) > 160 * 1024)
            return fail(ctx->fail(NLX_E_UNSUPPORTED, "num_wires %u: the quotient kernel's wire tile does not fit the 160 KB of LDS", d.num_wires));
    }
    c->cs_cap.resize((size_t)4 << d.cap_height);
    rc = fetch(ctx, c->cs_cap.data(), c->cs->cap, c->cs_cap.size() * 8);
    if (rc) return fail(rc);
    bool zero = true;
    for (int i = 0; i < 4; i++) zero = zero && d.circuit_digest[i] == 0;
    if (zero) {
        // circuit_digest = hash_no_pad(cap || hash_pad([]) || degree_bits)   (CircuitBuilder::build)
        std::vector<uint64_t> parts(c->cs_cap);
        uint64_t pad[12] = {1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1}, dom[4];
        hash_no_pad_host(pad, 12, dom);
        parts.insert(parts.end(), dom, dom + 4);
        parts.push_back(d.degree_bits);
        hash_no_pad_host(parts.data(), parts.size(), c->d.circuit_digest);
    }
    for (int i = 0; i <= NLX_MAX_STAGES; i++)
        if (hipEventCreate(&c->ev[i]) != hipSuccess) return fail(ctx->fail(NLX_E_HIP, "hipEventCreate failed"));
    *out = c;
    return NLX_OK;
} NLX_CATCH(ctx)

void nlx_circuit_destroy(nlx_circuit* c) NLX_TRY {
    if (!c) return;
    nlx_ctx* ctx = c->ctx;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (c->cs) nlx_commit_destroy(c->cs);
    ctx->release(c->d_sigma_values);
    ctx->release(c->d_gates);
    ctx->release(c->d_work);
    ctx->release(c->d_small);
    ctx->release(c->d_l0_scaled);
    ctx->release(c->d_lut_pairs);
    ctx->release(c->d_idx_of);
    ctx->release(c->d_mult);
    ctx->release(c->d_tabs);
    for (int i = 0; i <= NLX_MAX_STAGES; i++)
        if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    delete c;
} NLX_CATCH_VOID(nullptr)

int32_t nlx_circuit_digest(const nlx_circuit* c, uint64_t out[4]) NLX_TRY {
    if (!c || !out) return NLX_E_INVAL;
    memcpy(out, c->d.circuit_digest, 32);
    return NLX_OK;
} NLX_CATCH(nullptr)

int32_t nlx_circuit_constants_sigmas_cap(const nlx_circuit* c, uint64_t* cap_out) NLX_TRY {
    if (!c || !cap_out) return NLX_E_INVAL;
    memcpy(cap_out, c->cs_cap.data(), c->cs_cap.size() * 8);
    return NLX_OK;
} NLX_CATCH(nullptr)

size_t nlx_proof_max_bytes(const nlx_circuit* c) NLX_TRY {
    if (!c) return 0;
    const nlx_circuit_desc& d = c->d;
    const size_t capb = (size_t)32 << d.cap_height;
    const unsigned log_L = d.degree_bits + d.rate_bits;
    size_t bytes = 3 * capb + 16 * (size_t)(c->n_cs + d.num_wires + c->n_zs + d.num_challenges + c->n_lk_polys + c->n_q) + c->n_fri_rounds * capb;
    size_t per_query = 0;
    const uint32_t cols[4] = {c->n_cs, d.num_wires, c->n_zs, c->n_q};
    for (int o = 0; o < 4; o++) per_query += cols[o] * 8 + 1 + 32 * (size_t)(log_L - d.cap_height);
    per_query += c->n_fri_rounds * ((size_t)16 << d.fri_arity_bits) + c->n_fri_rounds * (1 + 32 * (size_t)log_L);
    bytes += per_query * d.fri_num_queries;
    bytes += ((size_t)16 << (d.degree_bits - c->n_fri_rounds * d.fri_arity_bits)) + 8 + 8 + 8 * (size_t)d.num_public_inputs;
    return bytes + 64;
} NLX_CATCH_VALUE(nullptr, 0)

int32_t nlx_pow_grind(nlx_ctx* ctx, const uint64_t state[12], uint32_t pos, uint32_t bits, uint64_t* nonce_out) NLX_TRY {
    if (!ctx) return NLX_E_INVAL;
    if (!state || !nonce_out) return ctx->fail(NLX_E_INVAL, "NULL argument");
    if (pos >= 8 || bits > 40) return ctx->fail(NLX_E_RANGE, "pos must be < 8 and bits <= 40");
    (void)hipSetDevice(ctx->device);
    unsigned long long* d_best = (unsigned long long*)ctx->alloc(256);
    if (!d_best) return NLX_E_NOMEM;
    PowParams pp{};
    for (int i = 0; i < 12; i++) pp.state[i] = state[i];
    pp.pos = pos;
    pp.bits = bits;
    pp.max_rounds = (uint64_t)1 << 24;
    launch_pow_grind(ctx->stream, pp, d_best);
    uint64_t best = 0;
    int32_t rc = fetch(ctx, &best, d_best, 8);
    ctx->release(d_best);
    if (rc) return rc;
    if (best == ~0ull) return ctx->fail(NLX_E_RANGE, "proof of work: no witness found");
    *nonce_out = best;
    return NLX_OK;
} NLX_CATCH(ctx)

int32_t nlx_prove_stage_times(const nlx_circuit* c, uint32_t* n_stages, const char** names_out, float* ms_out) NLX_TRY {
    if (!c || !n_stages) return NLX_E_INVAL;
    if (!c->timed) { *n_stages = 0; return NLX_OK; }
    *n_stages = c->n_stages;
    for (uint32_t i = 0; i < c->n_stages; i++) {
        if (names_out) names_out[i] = c->stage_names[i];
        if (ms_out) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]) != hipSuccess) ms = -1.f;
            ms_out[i] = ms;
        }
    }
    return NLX_OK;
} NLX_CATCH(nullptr)

}  // extern "C"

namespace {

// a8: Z and partial-product polynomials from the wires' subgroup values (device), committed.
// Temporaries are handed to `defer` (released by the caller after it has synchronised) or, without one,
// released here after a stream synchronisation.
// With lookup tables the same commitment carries, after them, every challenge round's RE and partial-sum polynomials
// (compute_all_lookup_polys); d_deltas = the rounds' (A, B, alpha, delta) on the device.
int32_t zs_stage(nlx_circuit* c, const uint64_t* d_wire_values, const uint64_t betas[2], const uint64_t gammas[2],
                 const uint64_t* d_deltas, nlx_commit** cz, std::vector<void*>* defer) {
    nlx_ctx* ctx = c->ctx;
    const nlx_circuit_desc& d = c->d;
    const unsigned log_n = d.degree_bits;
    const size_t n = c->n();
    uint64_t* d_zs = (uint64_t*)ctx->alloc((size_t)c->n_zs * n * 8);
    uint64_t* d_zs_scratch = (uint64_t*)ctx->alloc(zs_scratch_words(log_n, d.num_challenges) * 8);
    int32_t rc = NLX_OK;
    if (!d_zs || !d_zs_scratch) rc = NLX_E_NOMEM;
    if (!rc) {
        ZsParams zp{};
        zp.wires = d_wire_values;
        zp.wires_stride = n;
        zp.sigmas = c->d_sigma_values;
        zp.k_is = c->d_k_is;
        zp.w_n_table = ctx->tables.fwd[log_n];
        for (int i = 0; i < 2; i++) { zp.betas[i] = betas[i]; zp.gammas[i] = gammas[i]; }
        zp.out = d_zs;
        zp.log_n = log_n; zp.routed = d.num_routed_wires; zp.chunk = d.quotient_degree_factor; zp.nc = d.num_challenges;
        zp.npp = d.num_partial_products;
        launch_zs(ctx->stream, zp, d_zs_scratch);
        if (c->n_lk_polys)
            launch_lookup_polys(ctx->stream, c->lk, c->d_tabs, c->h_tabs.data(), d_wire_values, n, d.num_challenges, d_deltas,
                                d_zs + (size_t)c->n_zpp * n);
        rc = commit_build(ctx, d_zs, n, CommitInput::ValuesNatural, c->n_zs, log_n, d.rate_bits, d.cap_height, cz);
    }
    // commit_build only enqueues: the inputs must outlive the stream work
    if (defer) {
        if (d_zs) defer->push_back(d_zs);
        if (d_zs_scratch) defer->push_back(d_zs_scratch);
    } else {
        (void)hipStreamSynchronize(ctx->stream);
        ctx->release(d_zs);
        ctx->release(d_zs_scratch);
    }
    return rc;
}

// a9: quotient polynomials (compute_quotient_polys) from the three LDE tables, committed from coefficients.
int32_t quotient_stage(nlx_circuit* c, const nlx_commit* cw, const nlx_commit* cz, const uint64_t betas[2],
                       const uint64_t gammas[2], const uint64_t alphas[2], const uint64_t pih[4],
                       const uint64_t* d_deltas, const uint64_t* d_lut_polys, nlx_commit** cq,
                       const std::function<void(const char*)>& stage, std::vector<void*>* defer) {
    nlx_ctx* ctx = c->ctx;
    const nlx_circuit_desc& d = c->d;
    hipStream_t st = ctx->stream;
    const unsigned log_n = d.degree_bits;
    const size_t n = c->n(), L = c->L();
    const uint32_t nc = d.num_challenges, npp = d.num_partial_products;
    // one alpha power per vanishing term (GateAcc reads ap[T0 + k]): Z(1) terms, permutation terms, lookup terms, gate constraints
    c->n_terms = nc + nc * (npp + 1) + nc * c->n_lk_terms + c->max_gate_constraints;
    uint64_t* d_alpha_pows = (uint64_t*)ctx->alloc((size_t)2 * c->n_terms * 8);
    uint64_t* d_qvals = (uint64_t*)ctx->alloc((size_t)nc * L * 8);
    uint64_t* d_qchunks = (uint64_t*)ctx->alloc((size_t)nc * L * 8);
    int32_t rc = NLX_OK;
    if (!d_alpha_pows || !d_qvals || !d_qchunks) rc = NLX_E_NOMEM;
    if (!rc) {
        launch_pow_table(st, d_alpha_pows, alphas[0], alphas[1], c->n_terms, c->n_terms);
        QuotientParams qp{};
        qp.cs = c->cs->lde; qp.wires = cw->lde; qp.zs = cz->lde;
        qp.gates = c->d_gates; qp.k_is = c->d_k_is; qp.coset_base = c->d_coset_base;
        qp.w_n_table = ctx->tables.fwd[log_n];
        qp.zh_inv = c->d_zh_inv; qp.l0_scaled = c->d_l0_scaled; qp.alpha_pows = d_alpha_pows;
        qp.out = d_qvals;
        for (int i = 0; i < 2; i++) { qp.betas[i] = betas[i]; qp.gammas[i] = gammas[i]; }
        for (int i = 0; i < 4; i++) qp.pih[i] = pih[i];
        qp.alpha_stride = c->n_terms;
        qp.num_wires = d.num_wires; qp.work = c->d_work; qp.work_stride = c->work_stride;
        qp.log_n = log_n; qp.rate_bits = d.rate_bits; qp.n_gates = d.num_gates; qp.n_selectors = d.num_selectors;
        qp.n_consts_all = c->n_consts_all; qp.routed = d.num_routed_wires; qp.chunk = d.quotient_degree_factor; qp.nc = nc; qp.npp = npp;
        qp.gate_const0 = d.num_selectors + c->n_lk_sel;
        qp.n_lk_terms = c->n_lk_terms;
        if (c->n_lk_terms) {
            // check_lookup_constraints_batch: the lookup terms' share of both alpha sums goes into d_qvals first, k_quotient adds it
            LookupTermsParams lp{};
            lp.cs = c->cs->lde; lp.wires = cw->lde; lp.zs = cz->lde;
            lp.deltas = d_deltas; lp.lut_polys = d_lut_polys; lp.alpha_pows = d_alpha_pows; lp.out = d_qvals;
            lp.alpha_stride = c->n_terms; lp.t_lk = nc + nc * (npp + 1);
            lp.log_n = log_n; lp.rate_bits = d.rate_bits; lp.nc = nc; lp.sel0 = d.num_selectors; lp.lk0 = c->n_zpp;
            lp.n_lk_terms = c->n_lk_terms; lp.s = c->lk;
            ctx->begin_kernel("lookup_terms", 8.0 * L * (c->n_lk_sel + d.num_routed_wires + 2.0 * c->n_lk_polys + nc));
            launch_lookup_terms(st, lp);
            ctx->end_kernel();
            qp.accumulate = 1;
        }
        ctx->begin_kernel("quotient", 8.0 * L * (c->n_cs + d.num_wires + c->n_zs + nc) + 8.0 * L * nc);
        for (uint32_t g : c->poseidon_gates) {   // their part 2 ahead of the main kernel, which adds the rest
            launch_quotient_poseidon(st, qp, g, qp.accumulate != 0);
            qp.accumulate = 1;
        }
        launch_quotient(st, qp);
        ctx->end_kernel();
        stage("quotient_intt");
        launch_intt_dif_cosets(st, ctx->tables, d_qvals, nc, log_n, d.rate_bits, c->d_inv_scale_br);
        launch_quotient_chunks(st, d_qvals, d_qchunks, log_n, d.rate_bits, nc, c->d_wR_inv, c->d_chunk_scale);
        stage("commit_quotient");
        rc = commit_build(ctx, d_qchunks, n, CommitInput::CoeffsBitrev, c->n_q, log_n, d.rate_bits, d.cap_height, cq);
    }
    if (defer) {
        for (void* q : {(void*)d_alpha_pows, (void*)d_qvals, (void*)d_qchunks})
            if (q) defer->push_back(q);
    } else {
        (void)hipStreamSynchronize(st);
        ctx->release(d_alpha_pows);
        ctx->release(d_qvals);
        ctx->release(d_qchunks);
    }
    return rc;
}

}  // namespace

extern "C" {

int32_t nlx_prove(nlx_circuit* c, const uint64_t* wires, const uint64_t* public_inputs, uint8_t* proof_out,
                  size_t proof_cap, size_t* proof_len) NLX_TRY {
    if (!c) return NLX_E_INVAL;
    nlx_ctx* ctx = c->ctx;
    const nlx_circuit_desc& d = c->d;
    if (!wires || !proof_out || !proof_len || (!public_inputs && d.num_public_inputs))
        return ctx->fail(NLX_E_INVAL, "NULL argument");
    *proof_len = 0;
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    const size_t n = c->n();
    const unsigned log_n = d.degree_bits, cap_h = d.cap_height;
    const uint32_t nc = d.num_challenges, npp = d.num_partial_products;
    const uint32_t NR = c->n_fri_rounds;
    const size_t capw = (size_t)4 << cap_h;
    int32_t rc = NLX_OK;

    // everything allocated here is released at `done`
    std::vector<void*> scratch;
    auto dalloc = [&](size_t bytes) -> uint64_t* {
        void* p = ctx->alloc(bytes);
        if (p) scratch.push_back(p);
        return (uint64_t*)p;
    };
    nlx_commit *cw = nullptr, *cz = nullptr, *cq = nullptr;
    c->n_stages = 0;
    c->timed = false;
    auto stage = [&](const char* name) {
        if (c->n_stages < NLX_MAX_STAGES) {
            (void)hipEventRecord(c->ev[c->n_stages], st);
            c->stage_names[c->n_stages++] = name;
        }
    };
    Writer w{proof_out, 0, proof_cap};
    Challenger ch;
    std::vector<uint64_t> cap(capw);
    uint64_t pih[4];
    std::vector<uint64_t> h_pis(public_inputs, public_inputs + d.num_public_inputs);
    hash_no_pad_host(h_pis.data(), h_pis.size(), pih);

#define CHECK(x) do { rc = (x); if (rc) goto done; } while (0)
#define HIPCHK(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { rc = ctx->hip_fail(e__, #call); goto done; } } while (0)
#define CHECK_ALLOC(p) do { if (!(p)) { rc = NLX_E_NOMEM; goto done; } } while (0)
    {
        // ---- 2. wires commitment ----
        stage("commit_wires");
        Staged sw(ctx, wires, (size_t)d.num_wires * n * 8, true, false);
        CHECK(sw.status);
        // prover::set_lookup_wires, the first thing prove_with_partition_witness does to the witness it is handed: multiplicity
        // wires and padding slots, on the device copy (a device witness is written in place, as upstream's PartitionWitness is)
        if (d.num_luts)
            launch_set_lookup_wires(st, c->lk, c->d_tabs, c->h_tabs.data(), sw.as<uint64_t>(), n, c->d_mult, c->mult_words,
                                    c->d_mult + c->mult_words);
        CHECK(commit_build(ctx, sw.as<uint64_t>(), n, CommitInput::ValuesNatural, d.num_wires, log_n, d.rate_bits, cap_h, &cw));
        CHECK(fetch(ctx, cap.data(), cw->cap, capw * 8));
        if (d.num_luts) {
            uint32_t bad = 0;
            CHECK(fetch(ctx, &bad, c->d_mult + c->mult_words, 4));
            if (bad) { rc = ctx->fail(NLX_E_INVAL, "a looked-up input is not in its table (set_lookup_wires)"); goto done; }
        }
        w.u64s(cap.data(), capw);
        ch.observe(d.circuit_digest, 4);
        ch.observe(pih, 4);
        ch.observe(cap.data(), capw);
        uint64_t betas[2] = {0, 0}, gammas[2] = {0, 0}, alphas[2] = {0, 0};
        for (uint32_t i = 0; i < nc; i++) betas[i] = ch.challenge();
        for (uint32_t i = 0; i < nc; i++) gammas[i] = ch.challenge();
        // lookup challenges: deltas = betas ++ gammas ++ 2 nc more, NUM_COINS_LOOKUP = 4 per round (A, B, alpha, delta)
        // deltas and lut_polys are the SOURCES of asynchronous host-to-device copies: both live until the function returns (the
        // fetch() of the Zs cap below synchronises the stream after the second copy is enqueued; nothing here relies on the
        // runtime staging a pageable source at call time)
        uint64_t deltas[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        uint64_t lut_polys[2 * 16] = {};
        uint64_t* d_deltas = nullptr;   // deltas[4 nc] | lut_polys[nc tables]
        if (d.num_luts) {
            for (uint32_t i = 0; i < nc; i++) { deltas[i] = betas[i]; deltas[nc + i] = gammas[i]; }
            for (uint32_t i = 0; i < 2 * nc; i++) deltas[2 * nc + i] = ch.challenge();
            d_deltas = dalloc((size_t)(8 + 2 * 16) * 8);
            CHECK_ALLOC(d_deltas);
            HIPCHK(hipMemcpyAsync(d_deltas, deltas, sizeof deltas, hipMemcpyHostToDevice, st));
        }

        // ---- 4. partial products and Z ----
        stage("zs_partial_products");
        CHECK(zs_stage(c, sw.as<uint64_t>(), betas, gammas, d_deltas, &cz, &scratch));
        if (d.num_luts) {
            // vanishing_poly::get_lut_poly per (round, table), on the host while the device builds the Zs commitment: the pairs
            // (inp + B out) as coefficients in delta, first entry highest, zero-padded to whole table rows
            for (uint32_t ci = 0; ci < nc; ci++)
                for (uint32_t t = 0; t < d.num_luts; t++) {
                    const uint32_t len = c->lut_sizes[t], slots = c->lk.n_lut_slots;
                    const uint32_t degree = slots * ((len + slots - 1) / slots);
                    const uint16_t* pr = &c->lut_pairs[2 * (size_t)c->lut_offsets[t]];
                    const uint64_t B = deltas[4 * ci + 1], dl = deltas[4 * ci + 3];
                    uint64_t acc = 0;
                    for (uint32_t i = 0; i < len; i++) acc = gl::add(gl::mul(acc, dl), gl::add((uint64_t)pr[2 * i], gl::mul(B, (uint64_t)pr[2 * i + 1])));
                    acc = gl::mul(acc, gl::pow(dl, degree - len));
                    lut_polys[ci * d.num_luts + t] = acc;
                }
            HIPCHK(hipMemcpyAsync(d_deltas + 8, lut_polys, sizeof(uint64_t) * nc * d.num_luts, hipMemcpyHostToDevice, st));
        }
        CHECK(fetch(ctx, cap.data(), cz->cap, capw * 8));
        w.u64s(cap.data(), capw);
        ch.observe(cap.data(), capw);
        for (uint32_t i = 0; i < nc; i++) alphas[i] = ch.challenge();

        // ---- 5. quotient ----
        stage("quotient_eval");
        CHECK(quotient_stage(c, cw, cz, betas, gammas, alphas, pih, d_deltas, d_deltas ? d_deltas + 8 : nullptr, &cq, stage, &scratch));
        CHECK(fetch(ctx, cap.data(), cq->cap, capw * 8));
        w.u64s(cap.data(), capw);
        ch.observe(cap.data(), capw);

        // ---- 6. openings ----
        stage("openings");
        uint64_t zeta[2], gzeta[2];
        ch.ext_challenge(zeta);
        {
            const uint64_t g = gl::root_of_unity(log_n);
            gzeta[0] = gl::mul(zeta[0], g);
            gzeta[1] = gl::mul(zeta[1], g);
        }
        const nlx_commit* oracles[4] = {c->cs, cw, cz, cq};
        const uint32_t n_open = c->n_cs + d.num_wires + c->n_zs + c->n_q;
        uint64_t* d_points = dalloc(2048);
        const uint32_t nlk = c->n_lk_polys, n_next = nc + nlk;
        uint64_t* d_open = dalloc((size_t)(n_open + n_next) * 16);
        uint64_t* d_eval_scratch = dalloc(eval_scratch_words(d.num_wires > c->n_cs ? d.num_wires : c->n_cs, log_n) * 8);
        CHECK_ALLOC(d_points && d_open && d_eval_scratch);
        {
            // zeta^(2^k) and (g zeta)^(2^k), k < log_n, computed on the host (zeta is known here) so the
            // evaluation kernels start immediately
            uint64_t pts[4 + 2 * 2 * 32] = {zeta[0], zeta[1], gzeta[0], gzeta[1]};
            gl::Ext za{zeta[0], zeta[1]}, zb{gzeta[0], gzeta[1]};
            for (unsigned k = 0; k < 32; k++) {
                pts[4 + 2 * k] = za.a; pts[4 + 2 * k + 1] = za.b;
                pts[4 + 64 + 2 * k] = zb.a; pts[4 + 64 + 2 * k + 1] = zb.b;
                if (k + 1 < log_n) { za = gl::mul(za, za); zb = gl::mul(zb, zb); }
            }
            HIPCHK(hipMemcpyAsync(d_points, pts, sizeof pts, hipMemcpyHostToDevice, st));
            uint32_t off = 0;
            for (int o = 0; o < 4; o++) {
                launch_eval_br(st, oracles[o]->coeffs_br, n, oracles[o]->n_cols, log_n, d_points, d_open + 2 * (size_t)off,
                               d_eval_scratch, d_points + 4);
                off += oracles[o]->n_cols;
            }
            launch_eval_br(st, cz->coeffs_br, n, nc, log_n, d_points + 2, d_open + 2 * (size_t)n_open, d_eval_scratch,
                           d_points + 4 + 64);
            if (nlk)  // lookup_zs_next: the lookup polynomials (columns n_zpp.. of the Zs commitment) at g zeta
                launch_eval_br(st, cz->coeffs_br + (size_t)c->n_zpp * n, n, nlk, log_n, d_points + 2,
                               d_open + 2 * (size_t)(n_open + nc), d_eval_scratch, d_points + 4 + 64);
            HIPCHK(hipStreamSynchronize(st));
        }
        std::vector<uint64_t> open((size_t)(n_open + n_next) * 2);
        CHECK(fetch(ctx, open.data(), d_open, open.size() * 8));
        if (nlk) {
            // CommonCircuitData::fri_all_polys lists the lookup polynomials LAST in the zeta batch (after the quotient chunks):
            // move their openings from the middle (commitment order) to the end, then everything below reads FRI order
            uint64_t* zs0 = open.data() + 2 * (size_t)(c->n_cs + d.num_wires);
            std::vector<uint64_t> lkv(zs0 + 2 * (size_t)c->n_zpp, zs0 + 2 * (size_t)c->n_zs);
            memmove(zs0 + 2 * (size_t)c->n_zpp, zs0 + 2 * (size_t)c->n_zs, 2 * (size_t)c->n_q * 8);
            memcpy(zs0 + 2 * (size_t)(c->n_zpp + c->n_q), lkv.data(), lkv.size() * 8);
        }
        {
            // OpeningSet wire order: constants, plonk_sigmas, wires, plonk_zs, plonk_zs_next, partial_products, quotient_polys
            const uint64_t* o_cs = open.data();
            const uint64_t* o_w = o_cs + 2 * (size_t)c->n_cs;
            const uint64_t* o_zs = o_w + 2 * (size_t)d.num_wires;
            const uint64_t* o_pp = o_zs + 2 * (size_t)nc;
            const uint64_t* o_q = o_zs + 2 * (size_t)c->n_zpp;
            const uint64_t* o_lk = o_q + 2 * (size_t)c->n_q;
            const uint64_t* o_next = open.data() + 2 * (size_t)n_open;
            w.u64s(o_cs, 2 * (size_t)c->n_cs);
            w.u64s(o_w, 2 * (size_t)d.num_wires);
            w.u64s(o_zs, 2 * (size_t)nc);
            w.u64s(o_next, 2 * (size_t)nc);
            w.u64s(o_lk, 2 * (size_t)nlk);                    // lookup_zs, lookup_zs_next (read_opening_set order)
            w.u64s(o_next + 2 * (size_t)nc, 2 * (size_t)nlk);
            w.u64s(o_pp, 2 * (size_t)nc * npp);
            w.u64s(o_q, 2 * (size_t)c->n_q);
            ch.observe(open.data(), 2 * (size_t)n_open);
            ch.observe(o_next, 2 * (size_t)n_next);
        }

        // ---- 7. FRI ----
        {
            FriProveArgs fa;
            for (int o = 0; o < 4; o++) fa.oracles[o] = oracles[o];
            fa.n_oracles = 4;
            fa.nz[2] = nc;  // plonk_zs_next: the first nc columns of the Zs / partial-products oracle
            fa.tail_oracle = 2;  // the lookup polynomials: its last nlk columns, a group of their own at the end of both batches
            fa.tail_cols = nlk;
            for (int i = 0; i < 2; i++) { fa.zeta[i] = zeta[i]; fa.gzeta[i] = gzeta[i]; }
            fa.open0 = open.data();
            fa.open1 = open.data() + 2 * (size_t)n_open;
            fa.log_n = log_n; fa.rate_bits = d.rate_bits; fa.cap_height = cap_h; fa.arity_bits = d.fri_arity_bits;
            fa.pow_bits = d.fri_pow_bits; fa.n_queries = d.fri_num_queries; fa.n_rounds = NR;
            fa.d_coset_base = c->d_coset_base;
            fa.d_wA_inv = c->d_wA_inv;
            CHECK(fri_prove(ctx, fa, ch, w, scratch, stage));
        }
        w.usize(d.num_public_inputs);  // write_proof_with_public_inputs: write_usize(len), then the field vec
        w.u64s(h_pis.data(), h_pis.size());
        stage("end");
        c->n_stages--;  // "end" only closes the last interval
        c->timed = true;
        if (w.overflow) { rc = ctx->fail(NLX_E_RANGE, "proof buffer too small (need %zu bytes)", nlx_proof_max_bytes(c)); goto done; }
        *proof_len = w.len;
    }
done:
    {
        hipError_t e = hipStreamSynchronize(st);
        if (!rc && e != hipSuccess) rc = ctx->hip_fail(e, "hipStreamSynchronize");
        hipError_t le = hipGetLastError();
        if (!rc && le != hipSuccess) rc = ctx->hip_fail(le, "kernel launch");
    }
    for (void* p : scratch) ctx->release(p);
    if (cw) nlx_commit_destroy(cw);
    if (cz) nlx_commit_destroy(cz);
    if (cq) nlx_commit_destroy(cq);
#undef CHECK
#undef HIPCHK
#undef CHECK_ALLOC
    return rc;
} NLX_CATCH(nullptr)

// test hook nlx::batch_spawn_fault_after (armed by nlx_abi_selftest kind 3 / 4, ctx.hip): pretend that starting worker thread
// number >= this fails; -1 = off

int32_t nlx_batch_prove(nlx_circuit* const* workers, uint32_t n_workers, nlx_prove_job* jobs, size_t n_jobs) NLX_TRY {
    if (!workers || n_workers == 0 || (!jobs && n_jobs)) return NLX_E_INVAL;
    for (uint32_t w = 0; w < n_workers; w++) {
        if (!workers[w]) return NLX_E_INVAL;
        for (uint32_t v = 0; v < w; v++)
            if (workers[v]->ctx == workers[w]->ctx) return workers[w]->ctx->fail(NLX_E_INVAL, "nlx_batch_prove: workers must use distinct contexts");
    }
    std::atomic<size_t> next{0};
    auto run = [&](nlx_circuit* c) {
        (void)hipSetDevice(c->ctx->device);
        for (;;) {
            const size_t j = next.fetch_add(1);
            if (j >= n_jobs) return;
            nlx_prove_job& job = jobs[j];
            job.proof_len = 0;
            job.status = nlx_prove(c, job.wires, job.public_inputs, job.proof_out, job.proof_cap, &job.proof_len);
        }
    };
    if (n_workers == 1) {
        run(workers[0]);
    } else {
        // a std::thread that is still joinable when it is destroyed calls std::terminate: if creating worker w fails
        // (std::system_error: no more threads), the workers already started are left to drain the queue and are JOINED before the
        // error leaves this function as a return code (NLX_CATCH); the jobs they proved keep their status
        std::vector<std::thread> threads;
        threads.reserve(n_workers);
        bool spawn_failed = false;
        for (uint32_t w = 0; w < n_workers; w++) {
            try {
                if (batch_spawn_fault_after >= 0 && (int)w >= batch_spawn_fault_after) throw std::system_error(std::make_error_code(std::errc::resource_unavailable_try_again));
                threads.emplace_back(run, workers[w]);
            } catch (...) {
                spawn_failed = true;
                break;
            }
        }
        if (spawn_failed && threads.empty()) run(workers[0]);   // nobody started: the calling thread does the work
        for (auto& t : threads) t.join();
        if (spawn_failed) workers[0]->ctx->fail(NLX_OK, "nlx_batch_prove: could not start every worker thread; the jobs were proved by the ones that started");
    }
    for (size_t j = 0; j < n_jobs; j++)
        if (jobs[j].status != NLX_OK) return jobs[j].status;
    return NLX_OK;
} NLX_CATCH(nullptr)


// ---- stage-level entry points (the fine seam of INTEGRATION.md §3) ----

const nlx_commit* nlx_circuit_constants_sigmas(const nlx_circuit* c) NLX_TRY { return c ? c->cs : nullptr; } NLX_CATCH_VALUE(nullptr, nullptr)

int32_t nlx_partial_products_and_zs(nlx_circuit* c, const uint64_t* wires, const uint64_t betas[2], const uint64_t gammas[2],
                                    nlx_commit** zs_out) NLX_TRY {
    if (!c) return NLX_E_INVAL;
    nlx_ctx* ctx = c->ctx;
    if (!wires || !betas || !gammas || !zs_out) return ctx->fail(NLX_E_INVAL, "NULL argument");
    *zs_out = nullptr;
    (void)hipSetDevice(ctx->device);
    if (c->d.num_luts) return ctx->fail(NLX_E_UNSUPPORTED, "circuits with lookup tables are proved through nlx_prove (the stage calls carry no lookup challenges)");
    Staged sw(ctx, wires, (size_t)c->d.num_wires * c->n() * 8, true, false);
    if (sw.status) return sw.status;
    int32_t rc = zs_stage(c, sw.as<uint64_t>(), betas, gammas, nullptr, zs_out, nullptr);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (!rc && e != hipSuccess) rc = ctx->hip_fail(e, "hipStreamSynchronize");
    return rc;
} NLX_CATCH(nullptr)

int32_t nlx_quotient_eval(nlx_circuit* c, const nlx_commit* wires, const nlx_commit* zs, const uint64_t betas[2],
                          const uint64_t gammas[2], const uint64_t alphas[2], const uint64_t public_inputs_hash[4],
                          nlx_commit** quotient_out) NLX_TRY {
    if (!c) return NLX_E_INVAL;
    nlx_ctx* ctx = c->ctx;
    if (!wires || !zs || !betas || !gammas || !alphas || !public_inputs_hash || !quotient_out)
        return ctx->fail(NLX_E_INVAL, "NULL argument");
    *quotient_out = nullptr;
    const nlx_circuit_desc& d = c->d;
    if (d.num_luts) return ctx->fail(NLX_E_UNSUPPORTED, "circuits with lookup tables are proved through nlx_prove (the stage calls carry no lookup challenges)");
    if (wires->ctx != ctx || zs->ctx != ctx) return ctx->fail(NLX_E_INVAL, "commitments belong to another context");
    if (wires->n_cols != d.num_wires || zs->n_cols != c->n_zs || wires->log_n != d.degree_bits || zs->log_n != d.degree_bits ||
        wires->rate_bits != d.rate_bits || zs->rate_bits != d.rate_bits)
        return ctx->fail(NLX_E_INVAL, "commitment shapes do not match the circuit");
    for (int i = 0; i < 4; i++)
        if (public_inputs_hash[i] >= gl::P) return ctx->fail(NLX_E_RANGE, "public inputs hash is not canonical");
    (void)hipSetDevice(ctx->device);
    int32_t rc = quotient_stage(c, wires, zs, betas, gammas, alphas, public_inputs_hash, nullptr, nullptr, quotient_out, [](const char*) {}, nullptr);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (!rc && e != hipSuccess) rc = ctx->hip_fail(e, "hipStreamSynchronize");
    return rc;
} NLX_CATCH(nullptr)

int32_t nlx_fri_prove(nlx_ctx* ctx, const nlx_commit* const* oracles, uint32_t n_oracles, const uint32_t* n_next,
                      const uint64_t zeta[2], const uint64_t* openings_zeta, const uint64_t* openings_next,
                      const nlx_fri_params* params, nlx_challenger* challenger, uint8_t* proof_out, size_t proof_cap,
                      size_t* proof_len) NLX_TRY {
    if (!ctx) return NLX_E_INVAL;
    if (!oracles || !n_next || !zeta || !openings_zeta || !params || !challenger || !proof_out || !proof_len)
        return ctx->fail(NLX_E_INVAL, "NULL argument");
    *proof_len = 0;
    if (n_oracles < 1 || n_oracles > 4) return ctx->fail(NLX_E_RANGE, "1..4 oracles");
    for (uint32_t o = 0; o < n_oracles; o++) {
        if (!oracles[o] || oracles[o]->ctx != ctx) return ctx->fail(NLX_E_INVAL, "oracle %u: NULL or from another context", o);
        if (oracles[o]->log_n != oracles[0]->log_n || oracles[o]->rate_bits != oracles[0]->rate_bits ||
            oracles[o]->cap_height != oracles[0]->cap_height)
            return ctx->fail(NLX_E_INVAL, "oracles must share degree, rate and cap height");
    }
    uint32_t n_next_total = 0;
    for (uint32_t o = 0; o < n_oracles; o++) {
        if (n_next[o] > oracles[o]->n_cols) return ctx->fail(NLX_E_RANGE, "n_next[%u] exceeds the oracle's columns", o);
        n_next_total += n_next[o];
    }
    if (n_next_total && !openings_next) return ctx->fail(NLX_E_INVAL, "NULL openings_next");
    if (params->arity_bits < 2 || params->arity_bits > 4 || params->num_queries == 0 || params->num_queries > 128 ||
        params->pow_bits > 40 || oracles[0]->log_n < params->arity_bits)
        return ctx->fail(NLX_E_RANGE, "FRI parameters out of range");
    if (challenger->n_input > 8 || challenger->n_output > 8) return ctx->fail(NLX_E_INVAL, "challenger buffers hold at most 8 elements");
    (void)hipSetDevice(ctx->device);
    const unsigned log_n = oracles[0]->log_n, rate_bits = oracles[0]->rate_bits;
    int32_t rc = ctx->ensure_tables(log_n + rate_bits);
    if (rc) return rc;
    std::vector<void*> scratch;
    // coset bases g * w_L^r and w_A^-i
    const uint32_t R = 1u << rate_bits, A = 1u << params->arity_bits;
    std::vector<uint64_t> small(4 * R + A);
    coset_tables_host(log_n, rate_bits, small.data());
    const uint64_t w_A_inv = gl::inv(gl::root_of_unity(params->arity_bits));
    for (uint32_t i = 0; i < A; i++) small[4 * R + i] = gl::pow(w_A_inv, i);
    uint64_t* d_small = (uint64_t*)ctx->alloc(small.size() * 8);
    if (!d_small) return NLX_E_NOMEM;
    scratch.push_back(d_small);
    hipError_t e = hipMemcpyAsync(d_small, small.data(), small.size() * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { ctx->release(d_small); return ctx->hip_fail(e, "hipMemcpyAsync(tables)"); }
    FriProveArgs fa;
    for (uint32_t o = 0; o < n_oracles; o++) fa.oracles[o] = oracles[o];
    fa.n_oracles = n_oracles;
    for (uint32_t o = 0; o < n_oracles; o++) fa.nz[o] = n_next[o];
    const uint64_t g = gl::root_of_unity(log_n);
    for (int i = 0; i < 2; i++) { fa.zeta[i] = zeta[i]; fa.gzeta[i] = gl::mul(zeta[i], g); }
    fa.open0 = openings_zeta;
    fa.open1 = openings_next;
    fa.log_n = log_n; fa.rate_bits = rate_bits; fa.cap_height = oracles[0]->cap_height; fa.arity_bits = params->arity_bits;
    fa.pow_bits = params->pow_bits; fa.n_queries = params->num_queries;
    fa.n_rounds = fri_num_rounds(log_n, rate_bits, oracles[0]->cap_height, params->arity_bits, params->final_poly_bits);
    fa.d_coset_base = d_small;
    fa.d_wA_inv = d_small + 4 * R;
    Challenger ch;
    memcpy(ch.state, challenger->state, sizeof ch.state);
    memcpy(ch.in_buf, challenger->input, sizeof ch.in_buf);
    memcpy(ch.out_buf, challenger->output, sizeof ch.out_buf);
    ch.n_in = challenger->n_input;
    ch.n_out = challenger->n_output;
    Writer w{proof_out, 0, proof_cap};
    rc = fri_prove(ctx, fa, ch, w, scratch, [](const char*) {});
    e = hipStreamSynchronize(ctx->stream);
    if (!rc && e != hipSuccess) rc = ctx->hip_fail(e, "hipStreamSynchronize");
    for (void* q : scratch) ctx->release(q);
    if (rc) return rc;
    if (w.overflow) return ctx->fail(NLX_E_RANGE, "proof buffer too small");
    memcpy(challenger->state, ch.state, sizeof ch.state);
    memcpy(challenger->input, ch.in_buf, sizeof ch.in_buf);
    memcpy(challenger->output, ch.out_buf, sizeof ch.out_buf);
    challenger->n_input = ch.n_in;
    challenger->n_output = ch.n_out;
    *proof_len = w.len;
    return NLX_OK;
} NLX_CATCH(ctx)

// plonky2::iop::challenger::Challenger on the host, for callers without their own (tests, the C example)
void nlx_challenger_init(nlx_challenger* c) NLX_TRY { if (c) memset(c, 0, sizeof *c); } NLX_CATCH_VOID(nullptr)
int32_t nlx_challenger_observe(nlx_challenger* c, const uint64_t* elements, size_t n) NLX_TRY {
    if (!c || (!elements && n) || c->n_input > 8 || c->n_output > 8) return NLX_E_INVAL;
    Challenger ch;
    memcpy(ch.state, c->state, sizeof ch.state); memcpy(ch.in_buf, c->input, sizeof ch.in_buf);
    memcpy(ch.out_buf, c->output, sizeof ch.out_buf); ch.n_in = c->n_input; ch.n_out = c->n_output;
    for (size_t i = 0; i < n; i++) {
        if (elements[i] >= gl::P) return NLX_E_RANGE;
        ch.observe(elements[i]);
    }
    memcpy(c->state, ch.state, sizeof ch.state); memcpy(c->input, ch.in_buf, sizeof ch.in_buf);
    memcpy(c->output, ch.out_buf, sizeof ch.out_buf); c->n_input = ch.n_in; c->n_output = ch.n_out;
    return NLX_OK;
} NLX_CATCH(nullptr)
int32_t nlx_challenger_challenge(nlx_challenger* c, uint64_t* out, size_t n) NLX_TRY {
    if (!c || (!out && n) || c->n_input > 8 || c->n_output > 8) return NLX_E_INVAL;
    Challenger ch;
    memcpy(ch.state, c->state, sizeof ch.state); memcpy(ch.in_buf, c->input, sizeof ch.in_buf);
    memcpy(ch.out_buf, c->output, sizeof ch.out_buf); ch.n_in = c->n_input; ch.n_out = c->n_output;
    for (size_t i = 0; i < n; i++) out[i] = ch.challenge();
    memcpy(c->state, ch.state, sizeof ch.state); memcpy(c->input, ch.in_buf, sizeof ch.in_buf);
    memcpy(c->output, ch.out_buf, sizeof ch.out_buf); c->n_input = ch.n_in; c->n_output = ch.n_out;
    return NLX_OK;
} NLX_CATCH(nullptr)
int32_t nlx_hash_no_pad(const uint64_t* elements, size_t n, uint64_t out[4]) NLX_TRY {
    if ((!elements && n) || !out) return NLX_E_INVAL;
    for (size_t i = 0; i < n; i++)
        if (elements[i] >= gl::P) return NLX_E_RANGE;
    hash_no_pad_host(elements, n, out);
    for (int i = 0; i < 4; i++) out[i] = gl::canon(out[i]);
    return NLX_OK;
} NLX_CATCH(nullptr)

}  // extern "C"
