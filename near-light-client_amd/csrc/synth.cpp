// Synthetic nearx-shaped circuit + witness generator (host C++, workload generation only).
//
// The real SyncCircuit / VerifyCircuit witnesses need the Rust plonky2x CircuitBuilder and live
// NEAR RPC data (nearx/src/hint.rs:45-93), neither available here, so the bench and the parity
// tests prove SYNTHETIC circuits of the same static shape: 135 wires / 80 routed,
// standard_recursion_config, a gate mix of PoseidonGate / ArithmeticGate / BaseSumGate /
// ConstantGate / PublicInputGate / NoopGate rows with real copy constraints.  The witness
// satisfies every constraint, so the resulting proof verifies.
//
// This file produces INPUTS (constants, sigmas, wires, public inputs); it is not on the timed
// path and is shared by the HIP prover's tests/bench and the oracle's tests.
#include <cstdint>
#include <cstring>
#include <numeric>
#include <vector>
#include <exception>
#include <new>
#include "../../include/nlx_synth.h"
#include "gl.hpp"
#include "poseidon.hpp"
#include "poseidon_fast_constants.inc"

namespace {

struct Rng {  // SplitMix64
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    uint64_t next() {
        s += 0x9E3779B97F4A7C15ULL;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        return z ^ (z >> 31);
    }
    uint64_t field() { return next() % gl::P; }
    uint32_t below(uint32_t n) { return (uint32_t)(next() % n); }
};

const uint64_t FAST_FIRST[12] = NLX_POSEIDON_FAST_FIRST_RC_INIT;
const uint64_t FAST_RC[22] = NLX_POSEIDON_FAST_RC_INIT;
const uint64_t FAST_VS[22][11] = NLX_POSEIDON_FAST_VS_INIT;
const uint64_t FAST_W[22][11] = NLX_POSEIDON_FAST_W_HATS_INIT;
const uint64_t FAST_INIT[11][11] = NLX_POSEIDON_FAST_INITIAL_MATRIX_INIT;
const uint64_t CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};

uint64_t sbox(uint64_t x) {
    uint64_t x2 = gl::sqr(x), x4 = gl::sqr(x2);
    return gl::mul(gl::mul(x, x2), x4);
}
void mds(uint64_t* s) {
    uint64_t o[12];
    for (int r = 0; r < 12; r++) {
        uint64_t acc = 0;
        for (int i = 0; i < 12; i++) acc = gl::add(acc, gl::mul(s[(i + r) % 12], CIRC[i]));
        if (r == 0) acc = gl::add(acc, gl::mul(s[0], 8));
        o[r] = acc;
    }
    memcpy(s, o, sizeof o);
}

// PoseidonGate witness (plonky2 PoseidonGenerator::run_once): fills the 135 wires of one row
// for the given 12 inputs and swap bit.
void poseidon_gate_row(const uint64_t in[12], uint64_t swap, uint64_t* w /*135*/) {
    const uint64_t* RC = poseidon::RC_HOST;
    for (int i = 0; i < 12; i++) w[i] = in[i];
    w[24] = swap;
    uint64_t st[12];
    for (int i = 0; i < 4; i++) {
        uint64_t delta = gl::mul(swap, gl::sub(in[i + 4], in[i]));
        w[25 + i] = delta;
        st[i] = gl::add(in[i], delta);
        st[i + 4] = gl::sub(in[i + 4], delta);
    }
    for (int i = 8; i < 12; i++) st[i] = in[i];
    int rc = 0;
    for (int r = 0; r < 4; r++, rc++) {
        for (int i = 0; i < 12; i++) st[i] = gl::add(st[i], RC[rc * 12 + i]);
        if (r != 0)
            for (int i = 0; i < 12; i++) w[29 + 12 * (r - 1) + i] = st[i];
        for (int i = 0; i < 12; i++) st[i] = sbox(st[i]);
        mds(st);
    }
    for (int i = 0; i < 12; i++) st[i] = gl::add(st[i], FAST_FIRST[i]);
    {
        uint64_t res[12] = {st[0]};
        for (int r = 1; r < 12; r++)
            for (int c = 1; c < 12; c++) res[c] = gl::add(res[c], gl::mul(st[r], FAST_INIT[r - 1][c - 1]));
        memcpy(st, res, sizeof res);
    }
    for (int r = 0; r < 22; r++) {
        w[65 + r] = st[0];
        st[0] = sbox(st[0]);
        if (r < 21) st[0] = gl::add(st[0], FAST_RC[r]);
        uint64_t d = gl::mul(st[0], 25);
        for (int i = 1; i < 12; i++) d = gl::add(d, gl::mul(st[i], FAST_W[r][i - 1]));
        for (int i = 1; i < 12; i++) st[i] = gl::add(st[i], gl::mul(st[0], FAST_VS[r][i - 1]));
        st[0] = d;
    }
    rc += 22;
    for (int r = 0; r < 4; r++, rc++) {
        for (int i = 0; i < 12; i++) st[i] = gl::add(st[i], RC[rc * 12 + i]);
        for (int i = 0; i < 12; i++) w[87 + 12 * r + i] = st[i];
        for (int i = 0; i < 12; i++) st[i] = sbox(st[i]);
        mds(st);
    }
    for (int i = 0; i < 12; i++) w[12 + i] = st[i];
}

uint32_t gate_degree(uint32_t kind, uint32_t p0, uint32_t p1 = 0) {
    switch (kind) {
        case NLX_GATE_NOOP: return 0;
        case NLX_GATE_LOOKUP: return 0;
        case NLX_GATE_LOOKUP_TABLE: return 0;
        case NLX_GATE_CONSTANT: return 1;
        case NLX_GATE_PUBLIC_INPUT: return 1;
        case NLX_GATE_ARITHMETIC: return 3;
        case NLX_GATE_BASE_SUM: return p0;
        case NLX_GATE_POSEIDON: return 7;
        case NLX_GATE_ARITHMETIC_EXT: return 3;
        case NLX_GATE_MUL_EXT: return 3;
        case NLX_GATE_REDUCING: return 2;
        case NLX_GATE_REDUCING_EXT: return 2;
        case NLX_GATE_POSEIDON_MDS: return 1;
        case NLX_GATE_EXPONENTIATION: return 4;
        case NLX_GATE_RANDOM_ACCESS: return p0 + 1;
        case NLX_GATE_COSET_INTERPOLATION: return 6;  // with_max_degree(4, 8) -> degree 6 (p1)
        case NLX_GATE_U32_ADD_MANY:
        case NLX_GATE_U32_ARITHMETIC:
        case NLX_GATE_U32_SUBTRACTION:
        case NLX_GATE_U32_RANGE_CHECK: return 4;  // 1 << limb_bits
        case NLX_GATE_COMPARISON: return p1 ? 1u << ((p0 + p1 - 1) / p1) : 4;  // 1 << chunk_bits
    }
    return 0;
}

struct Dsu {
    std::vector<uint32_t> p;
    explicit Dsu(size_t n) : p(n) { std::iota(p.begin(), p.end(), 0u); }
    uint32_t find(uint32_t x) {
        while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; }
        return x;
    }
    void unite(uint32_t a, uint32_t b) {
        a = find(a); b = find(b);
        if (a != b) p[b] = a;
    }
};

}  // namespace

// the generator's entry points never throw either (same contract as include/nlx.h; this library has no context to report to)
#define NLX_TRY try
#define NLX_CATCH(ctx) catch (const std::bad_alloc&) { return -2; } catch (...) { return -1; }
#define NLX_CATCH_VOID(ctx) catch (...) { return; }
#define NLX_CATCH_VALUE(ctx, v) catch (...) { return v; }

extern "C" {

// gate list sorted by (degree, id) as plonky2's CircuitBuilder does; the set depends on the mix
static uint32_t build_gate_list(const nlx_synth_params* sp, uint32_t* kinds, uint32_t* p0, uint32_t* p1) {
    uint32_t k = 0;
    auto add = [&](uint32_t kind, uint32_t a, uint32_t b) { kinds[k] = kind; p0[k] = a; p1[k] = b; k++; };
    // degree 0, by id: "LookupGate {..lut_hash}" (one per table) < "LookupTableGate {..}" (one per table) < "NoopGate"
    for (uint32_t t = 0; t < sp->num_luts; t++) add(NLX_GATE_LOOKUP, t, 0);
    for (uint32_t t = 0; t < sp->num_luts; t++) add(NLX_GATE_LOOKUP_TABLE, t, 0);
    add(NLX_GATE_NOOP, 0, 0);                                    // degree 0
    add(NLX_GATE_CONSTANT, 2, 0);                                // degree 1: "ConstantGate" < "PoseidonMdsGate" < "PublicInputGate"
    if (sp->pct_misc) add(NLX_GATE_POSEIDON_MDS, 0, 0);
    add(NLX_GATE_PUBLIC_INPUT, 0, 0);
    if (sp->pct_base_sum) add(NLX_GATE_BASE_SUM, 2, 63);         // degree 2: "BaseSum" < "Comparison" (1-bit chunks) < "ReducingExtension" < "Reducing"
    if (sp->pct_u32 && sp->wide_comparison) add(NLX_GATE_COMPARISON, 25, 25);  // 132 constraints: more than any other gate
    if (sp->pct_extension) add(NLX_GATE_REDUCING_EXT, 32, 0);
    if (sp->pct_extension) add(NLX_GATE_REDUCING, 43, 0);
    if (sp->pct_extension) add(NLX_GATE_ARITHMETIC_EXT, 10, 0);  // degree 3: "ArithmeticExtension" < "ArithmeticGate" < "Mul..."
    if (sp->pct_arithmetic) add(NLX_GATE_ARITHMETIC, 20, 0);
    if (sp->pct_extension) add(NLX_GATE_MUL_EXT, 13, 0);
    if (sp->pct_u32 && !sp->wide_comparison) add(NLX_GATE_COMPARISON, 32, 16);           // degree 4: "ComparisonGate" < "ExponentiationGate" < "U32..."
    if (sp->pct_misc) add(NLX_GATE_EXPONENTIATION, 66, 0);
    if (sp->pct_u32) add(NLX_GATE_U32_ADD_MANY, 2, 5);           // num_ops as plonky2x derives them for 135 / 80 wires
    if (sp->pct_u32) add(NLX_GATE_U32_ARITHMETIC, 3, 0);
    if (sp->pct_u32) add(NLX_GATE_U32_RANGE_CHECK, 7, 0);
    if (sp->pct_u32) add(NLX_GATE_U32_SUBTRACTION, 6, 0);
    if (sp->pct_misc) add(NLX_GATE_RANDOM_ACCESS, 4, 4 | (2u << 16));  // degree 5: bits 4, 4 copies, 2 extra constants
    if (sp->pct_misc) add(NLX_GATE_COSET_INTERPOLATION, 4, 6);   // degree 6: CosetInterpolationGate::with_max_degree(4, 8)
    if (sp->pct_poseidon) add(NLX_GATE_POSEIDON, 0, 0);          // degree 7
    return k;
}

// synthetic table t: 2^lut_bits entries, inputs a permutation of the indices (so the input -> index map is exercised),
// outputs a fixed 16-bit function
static inline void lut_entry(uint32_t t, uint32_t i, uint16_t* inp, uint16_t* out) {
    *inp = (uint16_t)(i ^ (t ? 0x00A5u + t : 0u));
    *out = (uint16_t)((i * i + 7 * t + 3) & 0xFFFF);
}
struct LookupPlan {
    uint32_t T = 0, len = 0, lookups = 0, lu_rows = 0, lut_rows = 0, block = 0;
    explicit LookupPlan(const nlx_synth_params* sp) {
        T = sp->num_luts;
        if (!T) return;
        len = 1u << sp->lut_bits;
        lookups = sp->num_lookups;
        lu_rows = (lookups + 39) / 40;
        lut_rows = (len + 25) / 26;
        block = lu_rows + lut_rows + 1;
    }
    uint32_t end_row() const { return 1 + T * block; }  // first row after the lookup blocks (they start at row 1)
};

void nlx_synth_shape(const nlx_synth_params* sp, uint32_t* n_gates, uint32_t* n_selectors) NLX_TRY {
    uint32_t kinds[40], p0[40], p1[40];
    const uint32_t g = build_gate_list(sp, kinds, p0, p1);
    *n_gates = g;
    // greedy selector groups with max_degree = 8 (gates::selectors::selector_polynomials)
    uint32_t max_deg = gate_degree(kinds[g - 1], p0[g - 1], p1[g - 1]);
    if (max_deg + g - 1 <= 8) { *n_selectors = 1; return; }
    uint32_t sel = 0, start = 0;
    while (start < g) {
        uint32_t size = 0;
        while (start + size < g && size + gate_degree(kinds[start + size], p0[start + size], p1[start + size]) < 8) size++;
        start += size;
        sel++;
    }
    *n_selectors = sel;
} NLX_CATCH_VOID(nullptr)

// Re-target a generated witness to new public inputs: only the PublicInputGate row (row 0, wires
// 0..3 = hash_no_pad(public_inputs)) depends on them.
int32_t nlx_synth_set_public_inputs(uint64_t* wires, uint32_t log_n, const uint64_t* public_inputs, uint32_t count) NLX_TRY {
    if (!wires || (!public_inputs && count)) return NLX_E_INVAL;
    const size_t n = (size_t)1 << log_n;
    uint64_t st[12] = {0};
    for (uint32_t off = 0; off < count; off += 8) {
        uint32_t m = count - off < 8 ? count - off : 8;
        for (uint32_t j = 0; j < m; j++) {
            if (public_inputs[off + j] >= gl::P) return NLX_E_INVAL;
            st[j] = public_inputs[off + j];
        }
        poseidon::permute(st);
    }
    for (int i = 0; i < 4; i++) wires[(size_t)i * n] = st[i];
    return NLX_OK;
} NLX_CATCH(nullptr)

// Witness of the synthetic wide AIR (host mirror: stark.py wide_air): columns in groups of four (a, b, c, d),
//   next.a = a*b + c,  next.b = b*c + k1[g],  next.c = (a + b + c) * d,  d boolean and constant down the trace.
int32_t nlx_synth_stark_trace(uint32_t n_cols, uint32_t log_n, uint64_t seed, const uint64_t* k1, uint64_t* trace,
                              uint64_t* public_inputs) NLX_TRY {
    if (!k1 || !trace || !public_inputs || n_cols == 0 || (n_cols & 3)) return NLX_E_INVAL;
    if (log_n < 1 || log_n > 28) return NLX_E_RANGE;
    const size_t n = (size_t)1 << log_n;
    Rng rng(seed ^ 0x737461726bULL);
    for (uint32_t g = 0; g < n_cols / 4; g++) {
        if (k1[g] >= gl::P) return NLX_E_INVAL;
        uint64_t a = rng.field(), b = rng.field(), c = rng.field();
        const uint64_t d = rng.next() & 1;
        uint64_t *ca = trace + (size_t)(4 * g) * n, *cb = ca + n, *cc = cb + n, *cd = cc + n;
        for (size_t i = 0; i < n; i++) {
            ca[i] = a; cb[i] = b; cc[i] = c; cd[i] = d;
            const uint64_t na = gl::add(gl::mul(a, b), c), nb = gl::add(gl::mul(b, c), k1[g]);
            const uint64_t nc = d ? gl::add(gl::add(a, b), c) : 0;
            a = na; b = nb; c = nc;
        }
    }
    public_inputs[0] = trace[0];
    public_inputs[1] = trace[n];
    return NLX_OK;
} NLX_CATCH(nullptr)

static int32_t synth_circuit(const nlx_synth_params* sp, nlx_gate_desc* gates, uint64_t* k_is, uint64_t* constants,
                             uint64_t* sigmas, uint64_t* wires, uint64_t* public_inputs, uint16_t* lut_pairs,
                             uint32_t* lookup_rows) {
    const uint32_t W = 135, ROUTED = 80, NCONST = 2;
    const uint32_t log_n = sp->log_n;
    if (log_n < 3 || log_n > 26) return NLX_E_RANGE;
    const size_t n = (size_t)1 << log_n;
    Rng rng(sp->seed ^ 0x6e6c78ULL);
    const LookupPlan lp(sp);
    if (lp.T) {
        if (!lut_pairs || !lookup_rows) return NLX_E_INVAL;
        if (lp.T > 8 || sp->lut_bits < 1 || sp->lut_bits > 16 || lp.lookups == 0) return NLX_E_RANGE;
        if ((size_t)lp.end_row() + 2 >= n) return NLX_E_RANGE;
        for (uint32_t t = 0; t < lp.T; t++) {
            for (uint32_t i = 0; i < lp.len; i++) lut_entry(t, i, &lut_pairs[2 * ((size_t)t * lp.len + i)], &lut_pairs[2 * ((size_t)t * lp.len + i) + 1]);
            lookup_rows[3 * t] = 1 + t * lp.block;                              // last_lu_row
            lookup_rows[3 * t + 1] = lookup_rows[3 * t] + lp.lu_rows;           // last_lut_row
            lookup_rows[3 * t + 2] = lookup_rows[3 * t + 1] + lp.lut_rows - 1;  // first_lut_row; the row after it is a NoopGate
        }
    }
    const uint32_t n_lk_sel = lp.T ? 4 + lp.T : 0;  // lookup selector columns sit between the gate selectors and the gate constants

    // ---- gate table + selector groups ----
    uint32_t n_gates, n_sel;
    nlx_synth_shape(sp, &n_gates, &n_sel);
    int g_noop = 0;
    int g_base = -1, g_arith = -1, g_pos = -1, g_aext = -1, g_mext = -1, g_red = -1, g_rext = -1, g_pmds = -1, g_exp = -1,
        g_ra = -1, g_ci = -1, g_cmp = -1, g_uadd = -1, g_uari = -1, g_usub = -1, g_urc = -1, g_const = 1, g_pi = 2;
    {
        uint32_t kinds[40], p0[40], p1[40];
        const uint32_t k = build_gate_list(sp, kinds, p0, p1);
        for (uint32_t g = 0; g < k; g++) {
            gates[g].kind = kinds[g]; gates[g].param0 = p0[g]; gates[g].param1 = p1[g]; gates[g].index = g;
            switch (kinds[g]) {
                case NLX_GATE_NOOP: g_noop = (int)g; break;
                case NLX_GATE_BASE_SUM: g_base = (int)g; break;
                case NLX_GATE_ARITHMETIC: g_arith = (int)g; break;
                case NLX_GATE_POSEIDON: g_pos = (int)g; break;
                case NLX_GATE_ARITHMETIC_EXT: g_aext = (int)g; break;
                case NLX_GATE_MUL_EXT: g_mext = (int)g; break;
                case NLX_GATE_REDUCING: g_red = (int)g; break;
                case NLX_GATE_REDUCING_EXT: g_rext = (int)g; break;
                case NLX_GATE_POSEIDON_MDS: g_pmds = (int)g; break;
                case NLX_GATE_EXPONENTIATION: g_exp = (int)g; break;
                case NLX_GATE_RANDOM_ACCESS: g_ra = (int)g; break;
                case NLX_GATE_COSET_INTERPOLATION: g_ci = (int)g; break;
                case NLX_GATE_COMPARISON: g_cmp = (int)g; break;
                case NLX_GATE_U32_ADD_MANY: g_uadd = (int)g; break;
                case NLX_GATE_U32_ARITHMETIC: g_uari = (int)g; break;
                case NLX_GATE_U32_SUBTRACTION: g_usub = (int)g; break;
                case NLX_GATE_U32_RANGE_CHECK: g_urc = (int)g; break;
                case NLX_GATE_CONSTANT: g_const = (int)g; break;
                case NLX_GATE_PUBLIC_INPUT: g_pi = (int)g; break;
                default: break;
            }
        }
    }
    if (n_sel == 1) {
        for (uint32_t g = 0; g < n_gates; g++) { gates[g].selector_index = 0; gates[g].group_start = 0; gates[g].group_end = n_gates; }
    } else {
        uint32_t start = 0, sel = 0;
        while (start < n_gates) {
            uint32_t size = 0;
            while (start + size < n_gates && size + gate_degree(gates[start + size].kind, gates[start + size].param0, gates[start + size].param1) < 8) size++;
            for (uint32_t g = start; g < start + size; g++) { gates[g].selector_index = sel; gates[g].group_start = start; gates[g].group_end = start + size; }
            start += size;
            sel++;
        }
    }
    // k_is = g^i (plonk_common::get_unique_coset_shifts)
    k_is[0] = 1;
    for (uint32_t j = 1; j < ROUTED; j++) k_is[j] = gl::mul(k_is[j - 1], gl::GEN);

    // ---- rows ----
    std::vector<uint8_t> row_gate(n, 0);
    auto W_at = [&](uint32_t col, size_t row) -> uint64_t& { return wires[(size_t)col * n + row]; };
    auto C_at = [&](uint32_t col, size_t row) -> uint64_t& { return constants[(size_t)col * n + row]; };
    const uint32_t cbase = n_sel + n_lk_sel;  // first gate-constant column
    memset(constants, 0, (size_t)(cbase + NCONST) * n * 8);
    Dsu dsu((size_t)ROUTED * n);
    auto slot = [&](uint32_t col, size_t row) { return (uint32_t)((size_t)col * n + row); };
    std::vector<uint32_t> pool32;  // routed slots holding 32-bit values (inputs of the u32 gates)
    std::vector<uint32_t> pool;  // routed slots whose value later rows may copy
    pool.reserve(n * 4);
    uint64_t prev_pos_out[12];
    size_t prev_pos_row = (size_t)-1;

    // public inputs and their hash at row 0
    for (uint32_t i = 0; i < sp->num_public_inputs; i++) public_inputs[i] = rng.field();
    uint64_t pih[4];
    {
        uint64_t st[12] = {0};
        for (uint32_t off = 0; off < sp->num_public_inputs; off += 8) {
            uint32_t m = sp->num_public_inputs - off < 8 ? sp->num_public_inputs - off : 8;
            for (uint32_t j = 0; j < m; j++) st[j] = public_inputs[off + j];
            poseidon::permute(st);
        }
        memcpy(pih, st, 32);
    }
    for (size_t row = 0; row < n; row++) {
        // random fill first (unconstrained wires stay random)
        for (uint32_t c = 0; c < W; c++) W_at(c, row) = rng.field();
        uint32_t kind = NLX_GATE_NOOP;
        int gidx = g_noop;
        if (row == 0) {
            kind = NLX_GATE_PUBLIC_INPUT; gidx = g_pi;
        } else if (row + 2 >= n) {
            kind = NLX_GATE_NOOP; gidx = g_noop;  // padding rows, as plonky2 pads with NoopGate
        } else if (row < lp.end_row()) {
            // CircuitBuilder::add_all_lookups, one block per table: LookupGate rows, then the table upside down on
            // LookupTableGate rows (first entries on the LAST row), then a NoopGate row
            const uint32_t t = (uint32_t)(row - 1) / lp.block, r = (uint32_t)(row - 1) % lp.block;
            if (r < lp.lu_rows) {
                kind = NLX_GATE_LOOKUP; gidx = (int)t;
                for (uint32_t sl = 0; sl < 40; sl++) {
                    const uint32_t q = r * 40 + sl;
                    uint16_t inp = 0, out = 0;  // slots past the last lookup stay unset (0): the PROVER pads them (set_lookup_wires)
                    if (q < lp.lookups) {
                        // skewed choice: a quarter of the table takes most lookups, some entries none
                        uint32_t e = rng.below(4) ? rng.below((lp.len + 3) / 4) : rng.below(lp.len);
                        lut_entry(t, e, &inp, &out);
                        if ((sl & 7) == 0) pool.push_back(slot(2 * sl + 1, row));  // later rows copy looked-up outputs
                    }
                    W_at(2 * sl, row) = inp;
                    W_at(2 * sl + 1, row) = out;
                }
            } else if (r < lp.lu_rows + lp.lut_rows) {
                kind = NLX_GATE_LOOKUP_TABLE; gidx = (int)(lp.T + t);
                const uint32_t first_lut = lookup_rows[3 * t + 2];
                for (uint32_t sl = 0; sl < 26; sl++) {
                    const uint32_t e = (first_lut - (uint32_t)row) * 26 + sl;
                    uint16_t inp = 0, out = 0;
                    if (e < lp.len) lut_entry(t, e, &inp, &out);
                    W_at(3 * sl, row) = inp;
                    W_at(3 * sl + 1, row) = out;
                    W_at(3 * sl + 2, row) = 0;  // multiplicity: the prover's set_lookup_wires writes it
                }
            }
        } else {
            uint32_t r = rng.below(100);
            uint32_t t = sp->pct_poseidon;
            if (r < t && g_pos >= 0) { kind = NLX_GATE_POSEIDON; gidx = g_pos; }
            else if (r < (t += sp->pct_arithmetic) && g_arith >= 0) { kind = NLX_GATE_ARITHMETIC; gidx = g_arith; }
            else if (r < (t += sp->pct_base_sum) && g_base >= 0) { kind = NLX_GATE_BASE_SUM; gidx = g_base; }
            else if (r < (t += sp->pct_constant)) { kind = NLX_GATE_CONSTANT; gidx = g_const; }
            else if (r < (t += sp->pct_extension) && g_aext >= 0) {
                switch (row & 3) {
                    case 0: kind = NLX_GATE_ARITHMETIC_EXT; gidx = g_aext; break;
                    case 1: kind = NLX_GATE_MUL_EXT; gidx = g_mext; break;
                    case 2: kind = NLX_GATE_REDUCING; gidx = g_red; break;
                    default: kind = NLX_GATE_REDUCING_EXT; gidx = g_rext; break;
                }
            }
            else if (r < (t += sp->pct_misc) && g_pmds >= 0) {
                switch (row % 4) {
                    case 0: kind = NLX_GATE_POSEIDON_MDS; gidx = g_pmds; break;
                    case 1: kind = NLX_GATE_EXPONENTIATION; gidx = g_exp; break;
                    case 2: kind = NLX_GATE_COSET_INTERPOLATION; gidx = g_ci; break;
                    default: kind = NLX_GATE_RANDOM_ACCESS; gidx = g_ra; break;
                }
            }
            else if (r < (t += sp->pct_u32) && g_cmp >= 0) {
                switch (row % 5) {
                    case 0: kind = NLX_GATE_U32_ADD_MANY; gidx = g_uadd; break;
                    case 1: kind = NLX_GATE_U32_ARITHMETIC; gidx = g_uari; break;
                    case 2: kind = NLX_GATE_U32_SUBTRACTION; gidx = g_usub; break;
                    case 3: kind = NLX_GATE_U32_RANGE_CHECK; gidx = g_urc; break;
                    default: kind = NLX_GATE_COMPARISON; gidx = g_cmp; break;
                }
            }
        }
        row_gate[row] = (uint8_t)gidx;
        switch (kind) {
            case NLX_GATE_PUBLIC_INPUT:
                for (int i = 0; i < 4; i++) W_at(i, row) = pih[i];
                break;
            case NLX_GATE_CONSTANT:
                for (int i = 0; i < 2; i++) {
                    uint64_t v = rng.field();
                    C_at(cbase + i, row) = v;
                    W_at(i, row) = v;
                    pool.push_back(slot(i, row));
                }
                break;
            case NLX_GATE_ARITHMETIC: {
                uint64_t c0 = rng.field(), c1 = rng.field();
                C_at(cbase, row) = c0;
                C_at(cbase + 1, row) = c1;
                for (uint32_t op = 0; op < 20; op++) {
                    for (uint32_t q = 0; q < 3; q++) {
                        // half of the operands are copies of earlier routed values (copy constraints)
                        if (!pool.empty() && (rng.next() & 1)) {
                            uint32_t src = pool[rng.below((uint32_t)pool.size())];
                            W_at(4 * op + q, row) = wires[src];
                            dsu.unite(src, slot(4 * op + q, row));
                        }
                    }
                    uint64_t m0 = W_at(4 * op, row), m1 = W_at(4 * op + 1, row), ad = W_at(4 * op + 2, row);
                    W_at(4 * op + 3, row) = gl::add(gl::mul(gl::mul(m0, m1), c0), gl::mul(ad, c1));
                    if ((op & 3) == 0) pool.push_back(slot(4 * op + 3, row));
                }
                break;
            }
            case NLX_GATE_BASE_SUM: {
                uint64_t sum = 0;
                for (uint32_t i = 63; i-- > 0;) {
                    uint64_t bit = rng.next() & 1;
                    W_at(1 + i, row) = bit;
                    sum = gl::add(gl::add(sum, sum), bit);
                }
                W_at(0, row) = sum;
                pool.push_back(slot(0, row));
                break;
            }
            case NLX_GATE_POSEIDON: {
                uint64_t in[12];
                bool chain = prev_pos_row != (size_t)-1 && (rng.next() & 3) != 0;
                for (int i = 0; i < 12; i++) in[i] = chain ? prev_pos_out[i] : rng.field();
                if (!chain && !pool.empty()) {  // absorb a few earlier values
                    for (int i = 0; i < 4; i++) {
                        uint32_t src = pool[rng.below((uint32_t)pool.size())];
                        in[i] = wires[src];
                        dsu.unite(src, slot(i, row));
                    }
                }
                uint64_t tmp[135];
                for (uint32_t c = 0; c < W; c++) tmp[c] = W_at(c, row);
                poseidon_gate_row(in, rng.next() & 1, tmp);
                for (uint32_t c = 0; c < W; c++) W_at(c, row) = tmp[c];
                if (chain)
                    for (int i = 0; i < 12; i++) dsu.unite(slot(12 + i, prev_pos_row), slot(i, row));
                for (int i = 0; i < 12; i++) prev_pos_out[i] = tmp[12 + i];
                prev_pos_row = row;
                pool.push_back(slot(12, row));
                break;
            }
            case NLX_GATE_ARITHMETIC_EXT: {
                const uint64_t c0 = rng.field(), c1 = rng.field();
                C_at(cbase, row) = c0;
                C_at(cbase + 1, row) = c1;
                for (uint32_t op = 0; op < 10; op++) {
                    const uint32_t b = 8 * op;
                    if (!pool.empty() && (rng.next() & 1)) {  // one operand limb copied from an earlier value
                        uint32_t src = pool[rng.below((uint32_t)pool.size())];
                        W_at(b, row) = wires[src];
                        dsu.unite(src, slot(b, row));
                    }
                    const gl::Ext m0{W_at(b, row), W_at(b + 1, row)}, m1{W_at(b + 2, row), W_at(b + 3, row)};
                    const gl::Ext ad{W_at(b + 4, row), W_at(b + 5, row)};
                    const gl::Ext o = gl::add(gl::mul(gl::mul(m0, m1), c0), gl::mul(ad, c1));
                    W_at(b + 6, row) = o.a;
                    W_at(b + 7, row) = o.b;
                    if (op == 0) pool.push_back(slot(b + 6, row));
                }
                break;
            }
            case NLX_GATE_MUL_EXT: {
                const uint64_t c0 = rng.field();
                C_at(cbase, row) = c0;
                for (uint32_t op = 0; op < 13; op++) {
                    const uint32_t b = 6 * op;
                    const gl::Ext m0{W_at(b, row), W_at(b + 1, row)}, m1{W_at(b + 2, row), W_at(b + 3, row)};
                    const gl::Ext o = gl::mul(gl::mul(m0, m1), c0);
                    W_at(b + 4, row) = o.a;
                    W_at(b + 5, row) = o.b;
                    if (op == 0) pool.push_back(slot(b + 4, row));
                }
                break;
            }
            case NLX_GATE_REDUCING:
            case NLX_GATE_REDUCING_EXT: {
                const bool ext = kind == NLX_GATE_REDUCING_EXT;
                const uint32_t nco = ext ? 32 : 43;
                const uint32_t start_coeffs = 6, start_accs = start_coeffs + (ext ? 2 * nco : nco);
                const gl::Ext alpha{W_at(2, row), W_at(3, row)};
                gl::Ext acc{W_at(4, row), W_at(5, row)};
                for (uint32_t i = 0; i < nco; i++) {
                    gl::Ext co = ext ? gl::Ext{W_at(start_coeffs + 2 * i, row), W_at(start_coeffs + 2 * i + 1, row)}
                                     : gl::Ext{W_at(start_coeffs + i, row), 0};
                    acc = gl::add(gl::mul(acc, alpha), co);
                    const uint32_t aw = (i == nco - 1) ? 0 : start_accs + 2 * i;
                    W_at(aw, row) = acc.a;
                    W_at(aw + 1, row) = acc.b;
                }
                pool.push_back(slot(0, row));
                break;
            }
            case NLX_GATE_POSEIDON_MDS: {
                for (int rr = 0; rr < 12; rr++) {
                    gl::Ext c{0, 0};
                    for (int i = 0; i < 12; i++) {
                        const int src = (i + rr) % 12;
                        c = gl::add(c, gl::mul(gl::Ext{W_at(2 * src, row), W_at(2 * src + 1, row)}, CIRC[i]));
                    }
                    if (rr == 0) c = gl::add(c, gl::mul(gl::Ext{W_at(0, row), W_at(1, row)}, (uint64_t)8));
                    W_at(24 + 2 * rr, row) = c.a;
                    W_at(24 + 2 * rr + 1, row) = c.b;
                }
                pool.push_back(slot(24, row));
                break;
            }
            case NLX_GATE_EXPONENTIATION: {
                const uint32_t nb = 66;
                const uint64_t base = W_at(0, row);
                uint64_t cur = 1;
                for (uint32_t i = 0; i < nb; i++) W_at(1 + i, row) = rng.next() & 1;  // power bits, little-endian
                for (uint32_t i = 0; i < nb; i++) {
                    const uint64_t bit = W_at(1 + (nb - 1 - i), row);
                    const uint64_t prev = i ? gl::sqr(cur) : 1;
                    cur = gl::mul(prev, bit ? base : 1);
                    W_at(2 + nb + i, row) = cur;
                }
                W_at(1 + nb, row) = cur;
                pool.push_back(slot(1 + nb, row));
                break;
            }
            case NLX_GATE_RANDOM_ACCESS: {
                const uint32_t bits = 4, copies = 4, extra = 2, vec = 16, routed = (2 + vec) * copies + extra;
                for (uint32_t cpy = 0; cpy < copies; cpy++) {
                    const uint32_t b0 = (2 + vec) * cpy;
                    const uint32_t idx = rng.below(vec);
                    W_at(b0, row) = idx;
                    W_at(b0 + 1, row) = W_at(b0 + 2 + idx, row);
                    for (uint32_t i = 0; i < bits; i++) W_at(routed + cpy * bits + i, row) = (idx >> i) & 1;
                    pool.push_back(slot(b0 + 1, row));
                }
                for (uint32_t i = 0; i < extra; i++) {
                    const uint64_t v = rng.field();
                    C_at(cbase + i, row) = v;
                    W_at((2 + vec) * copies + i, row) = v;
                }
                break;
            }
            case NLX_GATE_U32_ADD_MANY: {
                const uint32_t na = 2, nops = 5, nl = 18;
                for (uint32_t i = 0; i < nops; i++) {
                    const uint32_t b0 = (na + 3) * i, lb = (na + 3) * nops + nl * i;
                    uint64_t sum = 0;
                    for (uint32_t j = 0; j < na; j++) {
                        uint64_t v = rng.next() & 0xFFFFFFFFULL;
                        if (j == 0 && !pool32.empty() && (rng.next() & 1)) {  // a u32 produced earlier in the circuit
                            const uint32_t src = pool32[rng.below((uint32_t)pool32.size())];
                            v = wires[src];
                            dsu.unite(slot(b0 + j, row), src);
                        }
                        W_at(b0 + j, row) = v;
                        sum += v;
                    }
                    const uint64_t cin = rng.next() & 1;
                    W_at(b0 + na, row) = cin;
                    sum += cin;
                    const uint64_t res = sum & 0xFFFFFFFFULL, cy = sum >> 32;
                    W_at(b0 + na + 1, row) = res;
                    W_at(b0 + na + 2, row) = cy;
                    for (uint32_t j = 0; j < 16; j++) W_at(lb + j, row) = (res >> (2 * j)) & 3;
                    for (uint32_t j = 0; j < 2; j++) W_at(lb + 16 + j, row) = (cy >> (2 * j)) & 3;
                    pool32.push_back(slot(b0 + na + 1, row));
                }
                break;
            }
            case NLX_GATE_U32_ARITHMETIC: {
                const uint32_t nops = 3;
                for (uint32_t i = 0; i < nops; i++) {
                    const uint32_t b0 = 6 * i, lb = 6 * nops + 32 * i;
                    uint64_t m0 = rng.next() & 0xFFFFFFFFULL;
                    const uint64_t m1 = rng.next() & 0xFFFFFFFFULL, ad = rng.next() & 0xFFFFFFFFULL;
                    if (!pool32.empty() && (rng.next() & 1)) {
                        const uint32_t src = pool32[rng.below((uint32_t)pool32.size())];
                        m0 = wires[src];
                        dsu.unite(slot(b0, row), src);
                    }
                    const uint64_t out = m0 * m1 + ad;  // < 2^64, and never a non-canonical field element: hi = 2^32-1 forces lo = 0
                    const uint64_t lo = out & 0xFFFFFFFFULL, hi = out >> 32;
                    W_at(b0, row) = m0; W_at(b0 + 1, row) = m1; W_at(b0 + 2, row) = ad;
                    W_at(b0 + 3, row) = lo; W_at(b0 + 4, row) = hi;
                    const uint64_t diff = 0xFFFFFFFFULL - hi;
                    W_at(b0 + 5, row) = diff ? gl::inv(diff) : 0;  // u32::MAX - hi is invertible unless hi = u32::MAX (then lo = 0)
                    for (uint32_t j = 0; j < 32; j++) W_at(lb + j, row) = (out >> (2 * j)) & 3;
                    pool32.push_back(slot(b0 + 3, row));
                    pool32.push_back(slot(b0 + 4, row));
                }
                break;
            }
            case NLX_GATE_U32_SUBTRACTION: {
                const uint32_t nops = 6;
                for (uint32_t i = 0; i < nops; i++) {
                    const uint32_t b0 = 5 * i, lb = 5 * nops + 16 * i;
                    uint64_t x = rng.next() & 0xFFFFFFFFULL;
                    const uint64_t y = rng.next() & 0xFFFFFFFFULL, bin = rng.next() & 1;
                    if (!pool32.empty() && (rng.next() & 1)) {
                        const uint32_t src = pool32[rng.below((uint32_t)pool32.size())];
                        x = wires[src];
                        dsu.unite(slot(b0, row), src);
                    }
                    const bool borrow = x < y + bin;
                    const uint64_t res = (x + (borrow ? (1ULL << 32) : 0)) - y - bin;
                    W_at(b0, row) = x; W_at(b0 + 1, row) = y; W_at(b0 + 2, row) = bin;
                    W_at(b0 + 3, row) = res; W_at(b0 + 4, row) = borrow ? 1 : 0;
                    for (uint32_t j = 0; j < 16; j++) W_at(lb + j, row) = (res >> (2 * j)) & 3;
                    pool32.push_back(slot(b0 + 3, row));
                }
                break;
            }
            case NLX_GATE_U32_RANGE_CHECK: {
                const uint32_t nin = 7;
                for (uint32_t i = 0; i < nin; i++) {
                    uint64_t v = rng.next() & 0xFFFFFFFFULL;
                    if (!pool32.empty() && (rng.next() & 1)) {
                        const uint32_t src = pool32[rng.below((uint32_t)pool32.size())];
                        v = wires[src];
                        dsu.unite(slot(i, row), src);
                    }
                    W_at(i, row) = v;
                    for (uint32_t j = 0; j < 16; j++) W_at(nin + 16 * i + j, row) = (v >> (2 * j)) & 3;
                }
                break;
            }
            case NLX_GATE_COMPARISON: {
                // result = (first <= second) on num_bits-bit inputs in num_chunks chunks (32 bits as 16 chunks of 2 by
                // default; 25 chunks of 1 bit - 132 constraints, the widest gate - with sp->wide_comparison)
                const uint32_t nch = gates[g_cmp].param1, cb = (gates[g_cmp].param0 + nch - 1) / nch;
                const uint64_t in_mask = (1ULL << gates[g_cmp].param0) - 1, ch_mask = (1ULL << cb) - 1;
                uint64_t a = rng.next() & in_mask, b = rng.next() & in_mask;
                if ((rng.next() & 7) == 0) b = a;  // exercise the all-chunks-equal path
                if (!pool32.empty() && (rng.next() & 1)) {
                    const uint32_t src = pool32[rng.below((uint32_t)pool32.size())];
                    if (wires[src] <= in_mask) {
                        a = wires[src];
                        dsu.unite(slot(0, row), src);
                    }
                }
                W_at(0, row) = a; W_at(1, row) = b;
                uint64_t msd = 0;
                for (uint32_t i = 0; i < nch; i++) {
                    const uint64_t f = (a >> (cb * i)) & ch_mask, s2 = (b >> (cb * i)) & ch_mask;
                    const uint64_t diff = gl::sub(s2, f);
                    const uint64_t eq = f == s2 ? 1 : 0;
                    W_at(4 + i, row) = f;
                    W_at(4 + nch + i, row) = s2;
                    W_at(4 + 2 * nch + i, row) = eq ? 1 : gl::inv(diff);  // equality_dummy: diff * dummy = 1 - chunks_equal
                    W_at(4 + 3 * nch + i, row) = eq;
                    const uint64_t iv = eq ? msd : 0;
                    W_at(4 + 4 * nch + i, row) = iv;
                    msd = gl::add(iv, eq ? 0 : diff);
                }
                W_at(3, row) = msd;
                const uint64_t shifted = gl::add(1ULL << cb, msd);  // in [1, 2^(cb+1)): msd in (-2^cb, 2^cb)
                for (uint32_t bb = 0; bb <= cb; bb++) W_at(4 + 5 * nch + bb, row) = (shifted >> bb) & 1;
                W_at(2, row) = (shifted >> cb) & 1;
                pool.push_back(slot(2, row));
                break;
            }
            case NLX_GATE_COSET_INTERPOLATION: {
                // 16 extension values on the coset shift*<g>, interpolated at an extension point (the FRI verifier's
                // compute_evaluation); intermediates every degree-1 = 5 points
                const uint32_t bits = 4, deg = 6, np = 16, sep = 1 + 2 * np, sev = sep + 2, si = sev + 2;
                const uint32_t ni = (np - 2) / (deg - 1), ssh = si + 4 * ni;
                uint64_t shift = rng.field();
                if (shift == 0) shift = 1;
                W_at(0, row) = shift;
                if (!pool.empty() && (rng.next() & 1)) {  // a value copied from earlier in the circuit
                    const uint32_t src = pool[rng.below((uint32_t)pool.size())];
                    W_at(1, row) = wires[src];
                    dsu.unite(slot(1, row), src);
                }
                const gl::Ext point{W_at(sep, row), W_at(sep + 1, row)};
                const gl::Ext pt = gl::mul(point, gl::inv(shift));
                W_at(ssh, row) = pt.a;
                W_at(ssh + 1, row) = pt.b;
                const uint64_t gen = gl::root_of_unity(bits), np_inv = gl::inv((uint64_t)np);
                gl::Ext ev{0, 0}, pr{1, 0};
                uint64_t xj = 1;
                uint32_t j = 0;
                for (uint32_t c = 0; c <= ni; c++) {
                    const uint32_t start = c == 0 ? 0 : 1 + (deg - 1) * c;
                    uint32_t end = c == 0 ? deg : start + deg - 1;
                    end = end > np ? np : end;
                    for (; j < end; j++) {
                        const gl::Ext term{gl::sub(pt.a, xj), pt.b};
                        const gl::Ext vp = gl::mul(gl::Ext{W_at(1 + 2 * j, row), W_at(2 + 2 * j, row)}, pr);
                        ev = gl::add(gl::mul(ev, term), gl::mul(vp, gl::mul(xj, np_inv)));
                        pr = gl::mul(pr, term);
                        xj = gl::mul(xj, gen);
                    }
                    (void)start;
                    if (c < ni) {
                        W_at(si + 2 * c, row) = ev.a; W_at(si + 2 * c + 1, row) = ev.b;
                        W_at(si + 2 * (ni + c), row) = pr.a; W_at(si + 2 * (ni + c) + 1, row) = pr.b;
                    }
                }
                W_at(sev, row) = ev.a;
                W_at(sev + 1, row) = ev.b;
                pool.push_back(slot(sev, row));
                pool.push_back(slot(sev + 1, row));
                break;
            }
            default: break;
        }
        // keep the pool bounded so copies stay "recent" (locality like a real circuit)
        if (pool.size() > 4096) pool.erase(pool.begin(), pool.begin() + 2048);
        if (pool32.size() > 4096) pool32.erase(pool32.begin(), pool32.begin() + 2048);
    }
    // selector columns
    for (size_t row = 0; row < n; row++) {
        uint32_t g = row_gate[row];
        for (uint32_t s = 0; s < n_sel; s++)
            C_at(s, row) = (gates[g].selector_index == s) ? gates[g].index : 0xFFFFFFFFULL;
    }
    // lookup selectors (gates::selectors::selectors_lookup, selector_ends_lookups): TransSre on the table rows, TransLdc on
    // the LookupGate rows, InitSre on the row after the table, LastLdc on the first LookupGate row, one end selector per table
    for (uint32_t t = 0; t < lp.T; t++) {
        const uint32_t last_lu = lookup_rows[3 * t], last_lut = lookup_rows[3 * t + 1], first_lut = lookup_rows[3 * t + 2];
        for (uint32_t row = last_lut; row <= first_lut; row++) C_at(n_sel + 0, row) = 1;
        for (uint32_t row = last_lu; row < last_lut; row++) C_at(n_sel + 1, row) = 1;
        C_at(n_sel + 2, first_lut + 1) = 1;
        C_at(n_sel + 3, last_lu) = 1;
        C_at(n_sel + 4 + t, last_lut) = 1;
    }
    // ---- sigma: each copy class becomes one cycle (next slot in increasing order, wrapping) ----
    const size_t total = (size_t)ROUTED * n;
    std::vector<uint32_t> next_in_class(total), last_seen(total, 0xFFFFFFFFu), first_seen(total, 0xFFFFFFFFu);
    for (size_t s = 0; s < total; s++) {
        uint32_t r = dsu.find((uint32_t)s);
        if (first_seen[r] == 0xFFFFFFFFu) first_seen[r] = (uint32_t)s;
        else next_in_class[last_seen[r]] = (uint32_t)s;
        last_seen[r] = (uint32_t)s;
    }
    for (size_t s = 0; s < total; s++) {
        uint32_t r = dsu.find((uint32_t)s);
        if (last_seen[r] == s) next_in_class[s] = first_seen[r];
    }
    std::vector<uint64_t> subgroup(n);
    {
        uint64_t w = gl::root_of_unity(log_n), x = 1;
        for (size_t i = 0; i < n; i++) { subgroup[i] = x; x = gl::mul(x, w); }
    }
    for (size_t s = 0; s < total; s++) {
        uint32_t t = next_in_class[s];
        sigmas[s] = gl::mul(k_is[t / n], subgroup[t % n]);
    }
    return NLX_OK;
}

int32_t nlx_synth_circuit(const nlx_synth_params* sp, nlx_gate_desc* gates, uint64_t* k_is, uint64_t* constants,
                          uint64_t* sigmas, uint64_t* wires, uint64_t* public_inputs) NLX_TRY {
    if (sp->num_luts) return NLX_E_INVAL;  // circuits with tables come from nlx_synth_circuit_lookups
    return synth_circuit(sp, gates, k_is, constants, sigmas, wires, public_inputs, nullptr, nullptr);
} NLX_CATCH(nullptr)

int32_t nlx_synth_circuit_lookups(const nlx_synth_params* sp, nlx_gate_desc* gates, uint64_t* k_is, uint64_t* constants,
                                  uint64_t* sigmas, uint64_t* wires, uint64_t* public_inputs, uint16_t* lut_pairs,
                                  uint32_t* lookup_rows) NLX_TRY {
    return synth_circuit(sp, gates, k_is, constants, sigmas, wires, public_inputs, lut_pairs, lookup_rows);
} NLX_CATCH(nullptr)

}  // extern "C"
