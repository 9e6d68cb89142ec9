// starky-style STARK prover on one MI355X (C ABI: nlx_stark_build, nlx_stark_prove).
//
// Replaces starky::prover::prove (trace commitment, compute_quotient_polys, StarkOpeningSet, FRI) - the
// public ancestor of the un-vendored starkyx/curta prover that plonky2x runs for nearx's Ed25519 and
// SHA-256 gadgets (nearx/src/builder.rs curta_* calls; Cargo.lock:6515).
//
// MI355X-first choices:
//  * the AIR is data: a register program (NLX_AIR_*) interpreted by k_air_quotient, one lane per point of
//    the quotient coset, with the VM registers in LDS ([reg][lane], conflict-free) so a program of any
//    shape runs without recompiling a kernel and without spilling to scratch;
//  * the quotient domain {g w^(i*step)} is exactly 2^qdb of the LDE table's cosets in the coset-major
//    layout, so "next row" is the neighbouring element of the same coset and the
//    trace LDE is read in place - no second low-degree extension as in the reference flow;
//  * quotient chunks come from per-coset inverse transforms + a 2^qdb-point DFT across cosets, as in the
//    plonky2 path; commitments, openings and FRI are the same device code as nlx_prove.
#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>
#include "commit.hpp"
#include "fri.hpp"
#include "gl.hpp"
#include "poly.hpp"
#include "prover.hpp"
#include "transcript.hpp"
#include "air_vm.hpp"

using namespace nlx;

namespace nlx {

// One lane per point (r', k) of the quotient domain.  Registers live in LDS as regs[reg * blockDim + lane]:
// every access of a wave touches 64 consecutive 8-byte words (no bank conflicts); the program counter, the
// opcode and the operands are wave-uniform, so decode runs on the scalar unit.  blockIdx.y selects the program
// segment (NLX_AIR_SEGMENT): the segments of one point run on different waves and k_air_combine adds them.
__global__ __launch_bounds__(256) void k_air_quotient(AirParams p) {
    extern __shared__ uint64_t regs[];
    const size_t pos = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned log_Q = p.log_n + p.qdb;
    if (pos >> log_Q) return;  // n >= blockDim.x is checked on the host: whole blocks are in or out
    const size_t n = (size_t)1 << p.log_n;
    const uint32_t rq = (uint32_t)(pos >> p.log_n), k = (uint32_t)(pos & (n - 1));
    const uint32_t r = rq << (p.rate_bits - p.qdb);  // LDE coset of quotient coset rq
    const size_t row = ((size_t)r << p.log_n) + k, row_next = ((size_t)r << p.log_n) + ((k + 1) & (n - 1));
    const size_t qrow_next = ((size_t)rq << p.log_n) + ((k + 1) & (n - 1));
    const uint64_t x = gl::mul(p.coset_base[rq], root_pow_(p.w_n_table, k, (uint32_t)(n >> 1)));
    const uint64_t zh = gl::inv(p.zh_inv[rq]);  // wave-uniform; one inversion per lane is noise next to the program
    const uint64_t z_last = gl::sub(x, p.g_inv);
    const uint64_t l_first = gl::mul(zh, p.l_inv[pos]);
    const uint64_t l_last = gl::mul(zh, p.l_inv[qrow_next]);  // 1 / (n (g x - 1)): g x is the next point of the coset
    uint64_t* my = regs + threadIdx.x;
    const uint32_t bd = blockDim.x;
    uint64_t acc0 = 0, acc1 = 0;
    const uint64_t a0 = p.alphas[0], a1 = p.alphas[1];
    const bool two = p.nc > 1;
    const uint32_t sg = blockIdx.y;
    const uint32_t pc_end = p.n_seg > 1 ? p.seg[2 * sg + 1] : p.n_words;
    for (uint32_t pc = p.n_seg > 1 ? p.seg[2 * sg] : 0; pc < pc_end; pc++) {
        const uint64_t w = p.program[pc];
        const uint32_t op = (uint32_t)(w & 0xFF), dst = (uint32_t)((w >> 8) & 0xFFFF);
        const uint32_t a = (uint32_t)((w >> 24) & 0xFFFF), b = (uint32_t)((w >> 40) & 0xFFFF);
        const uint32_t sh = (uint32_t)(w >> 56) & 0x3F;
        uint64_t c;
        switch (op) {
            case NLX_AIR_LOCAL: my[dst * bd] = p.cols[a][row]; continue;
            case NLX_AIR_NEXT: my[dst * bd] = p.cols[a][row_next]; continue;
            case NLX_AIR_PUBLIC: my[dst * bd] = p.pis[a]; continue;
            case NLX_AIR_PERIODIC:
                my[dst * bd] = p.periodic[((((size_t)a << p.qdb) + rq) << p.period_bits) + (k & ((1u << p.period_bits) - 1))];
                continue;
            case NLX_AIR_CONST: my[dst * bd] = p.program[++pc]; continue;
            case NLX_AIR_ADD: my[dst * bd] = gl::add(my[a * bd], mul_pow2(my[b * bd], sh)); continue;
            case NLX_AIR_SUB: my[dst * bd] = gl::sub(my[a * bd], mul_pow2(my[b * bd], sh)); continue;
            case NLX_AIR_MUL: my[dst * bd] = gl::mul(my[a * bd], my[b * bd]); continue;
            case NLX_AIR_MAC: my[dst * bd] = gl::add(my[sh * bd], gl::mul(my[a * bd], my[b * bd])); continue;
            case NLX_AIR_XOR3:
            case NLX_AIR_CH:
            case NLX_AIR_MAJ: {
                const uint64_t x = my[a * bd], y = my[b * bd], z = my[sh * bd];
                uint64_t res;
                if (op == NLX_AIR_CH) {
                    res = gl::add(z, gl::mul(x, gl::sub(y, z)));
                } else {
                    const uint64_t xy = gl::mul(x, y);
                    const uint64_t sx = gl::sub(gl::add(x, y), gl::add(xy, xy));
                    if (op == NLX_AIR_XOR3) {
                        const uint64_t sz = gl::mul(sx, z);
                        res = gl::sub(gl::add(sx, z), gl::add(sz, sz));
                    } else {
                        res = gl::add(xy, gl::mul(z, sx));
                    }
                }
                my[dst * bd] = res;
                continue;
            }
            case NLX_AIR_PACK_LOCAL:
            case NLX_AIR_PACK_NEXT: {
                // b loads in flight at once (wave-uniform trip count), then shift-accumulate
                const size_t rr = op == NLX_AIR_PACK_LOCAL ? row : row_next;
                uint64_t acc = 0;
#pragma unroll 8
                for (uint32_t i = 0; i < b; i++) acc = gl::add(acc, mul_pow2(p.cols[a + i][rr], i));
                my[dst * bd] = acc;
                continue;
            }
            case NLX_AIR_EMIT_BOOL: {
                // x (x - 1) for b consecutive columns: eight loads in flight, constraints emitted in column order
                const uint32_t cnt = b ? b : 1;
                for (uint32_t i0 = 0; i0 < cnt; i0 += 8) {
                    uint64_t v[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) v[i] = i0 + i < cnt ? p.cols[a + i0 + i][row] : 0;
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        if (i0 + i < cnt) {
                            const uint64_t cb = gl::mul(v[i], gl::sub(v[i], 1));
                            acc0 = gl::add(gl::mul(acc0, a0), cb);
                            if (two) acc1 = gl::add(gl::mul(acc1, a1), cb);
                        }
                    }
                }
                continue;
            }
            case NLX_AIR_EMIT_LOGUP: {
                // both coefficients of h (al + v1)(al + v2) - (al + v1) - (al + v2) over F_p[X]/(X^2 - 7), in registers
                const uint64_t al0 = p.pis[p.n_pis + sh], al1 = p.pis[p.n_pis + sh + 1];
                const uint64_t h0 = p.cols[b][row], h1 = p.cols[b + 1][row], v1 = p.cols[a][row];
                uint64_t c0, c1;
                if (dst == 0xFFFF) {
                    const uint64_t d0 = gl::add(al0, v1);
                    c0 = gl::sub(gl::add(gl::mul(h0, d0), mul_pow2(gl::mul(h1, al1), 3)), gl::add(gl::mul(h1, al1), 1));
                    c1 = gl::add(gl::mul(h0, al1), gl::mul(h1, d0));
                } else {
                    const uint64_t v2 = p.cols[dst][row];
                    const uint64_t s2 = gl::add(gl::add(al0, al0), gl::add(v1, v2));
                    const uint64_t a1sq = gl::mul(al1, al1);
                    const uint64_t u0 = gl::add(gl::mul(gl::add(al0, v1), gl::add(al0, v2)), gl::sub(mul_pow2(a1sq, 3), a1sq));
                    const uint64_t u1 = gl::mul(al1, s2);
                    const uint64_t hu = gl::mul(h1, u1);
                    c0 = gl::sub(gl::add(gl::mul(h0, u0), gl::sub(mul_pow2(hu, 3), hu)), s2);
                    c1 = gl::sub(gl::add(gl::mul(h0, u1), gl::mul(h1, u0)), gl::add(al1, al1));
                }
                acc0 = gl::add(gl::mul(gl::add(gl::mul(acc0, a0), c0), a0), c1);
                if (two) acc1 = gl::add(gl::mul(gl::add(gl::mul(acc1, a1), c0), a1), c1);
                continue;
            }
            case NLX_AIR_LOADV: {
                // the next `dst` words are independent loads: issue them all, then write the register file
                uint64_t v[8];
                uint32_t d8[8];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    v[i] = 0;
                    d8[i] = 0;
                    if ((uint32_t)i < dst) {
                        const uint64_t w2 = p.program[pc + 1 + i];
                        const uint32_t op2 = (uint32_t)(w2 & 0xFF), a2 = (uint32_t)((w2 >> 24) & 0xFFFF);
                        d8[i] = (uint32_t)((w2 >> 8) & 0xFFFF);
                        const uint64_t* src = op2 == NLX_AIR_LOCAL ? p.cols[a2] + row
                                            : op2 == NLX_AIR_NEXT ? p.cols[a2] + row_next
                                            : op2 == NLX_AIR_PUBLIC ? p.pis + a2
                                            : p.periodic + ((((size_t)a2 << p.qdb) + rq) << p.period_bits) + (k & ((1u << p.period_bits) - 1));
                        v[i] = *src;
                    }
                }
#pragma unroll
                for (int i = 0; i < 8; i++)
                    if ((uint32_t)i < dst) my[d8[i] * bd] = v[i];
                pc += dst;
                continue;
            }
            case NLX_AIR_EMIT_TRANSITION: c = gl::mul(my[a * bd], z_last); break;
            case NLX_AIR_EMIT_FIRST: c = gl::mul(my[a * bd], l_first); break;
            case NLX_AIR_EMIT_LAST: c = gl::mul(my[a * bd], l_last); break;
            default: c = my[a * bd]; break;  // NLX_AIR_EMIT (the host validated the opcode range)
        }
        acc0 = gl::add(gl::mul(acc0, a0), c);
        if (two) acc1 = gl::add(gl::mul(acc1, a1), c);
    }
    const size_t Q = (size_t)1 << log_Q;
    if (p.n_seg > 1) {
        // this segment's share of sum_i alpha^(N-1-i) c_i: its own Horner sum times alpha^(constraints after it)
        uint64_t* dst = p.part + (((size_t)sg * p.nc) << log_Q) + pos;
        dst[0] = gl::mul(acc0, p.seg_mul[2 * sg]);
        if (two) dst[Q] = gl::mul(acc1, p.seg_mul[2 * sg + 1]);
        return;
    }
    const uint64_t zi = p.zh_inv[rq];
    p.out[pos] = gl::mul(acc0, zi);
    if (two) p.out[Q + pos] = gl::mul(acc1, zi);
}

// out[c][pos] = (sum over segments of part[s][c][pos]) / Z_H(x)
__global__ __launch_bounds__(256) void k_air_combine(const uint64_t* __restrict__ part, const uint64_t* __restrict__ zh_inv,
                                                     uint64_t* __restrict__ out, uint32_t n_seg, uint32_t nc, uint32_t log_n,
                                                     uint32_t log_Q) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // over [challenge][pos]
    if (i >= ((size_t)nc << log_Q)) return;
    const size_t pos = i & (((size_t)1 << log_Q) - 1);
    uint64_t acc = 0;
    for (uint32_t sgm = 0; sgm < n_seg; sgm++) acc = gl::add(acc, part[(((size_t)sgm * nc) << log_Q) + i]);
    out[i] = gl::mul(acc, zh_inv[pos >> log_n]);
}

}  // namespace nlx

// natural order in, natural order out; inverse includes the 1/len factor (host side, build time only)
static void host_ntt(std::vector<uint64_t>& a, unsigned log_len, bool inverse) {
    const size_t len = (size_t)1 << log_len;
    for (size_t i = 0; i < len; i++) {
        size_t j = 0;
        for (unsigned b = 0; b < log_len; b++) j |= ((i >> b) & 1) << (log_len - 1 - b);
        if (j > i) std::swap(a[i], a[j]);
    }
    for (unsigned st = 1; st <= log_len; st++) {
        const size_t half = (size_t)1 << (st - 1);
        uint64_t w_st = gl::root_of_unity(st);
        if (inverse) w_st = gl::inv(w_st);
        for (size_t base = 0; base < len; base += 2 * half) {
            uint64_t w = 1;
            for (size_t k = 0; k < half; k++) {
                const uint64_t u = a[base + k], v = gl::mul(a[base + k + half], w);
                a[base + k] = gl::add(u, v);
                a[base + k + half] = gl::sub(u, v);
                w = gl::mul(w, w_st);
            }
        }
    }
    if (inverse) {
        const uint64_t inv_len = gl::inv((uint64_t)len);
        for (auto& v : a) v = gl::mul(v, inv_len);
    }
}

struct nlx_stark {
    nlx_ctx* ctx = nullptr;
    nlx_stark_desc d{};
    std::vector<uint64_t> program;  // canonicalised copy
    uint32_t qdb = 0, nq = 0, n_regs = 0, n_fri_rounds = 0;
    uint32_t n_rounds = 1, round_cols[3] = {0, 0, 0}, round_challenges[3] = {0, 0, 0}, round_values[3] = {0, 0, 0},
             n_round_challenges = 0;  // n_round_challenges: round values + challenges, i.e. the values array minus public inputs
    uint64_t air_digest[4] = {0, 0, 0, 0};  // the statement digest the transcript opens with (air_digest_host)
    uint64_t* d_program = nullptr;
    const AirGenEntry* gen = nullptr;   // a straight-line kernel generated from exactly this program (csrc/airgen/), or nullptr: the interpreter
    std::vector<uint32_t> seg;        // {first word, end word} per program segment
    std::vector<uint32_t> seg_after;  // constraints emitted after each segment
    std::vector<uint32_t> seg_regs;   // registers each segment uses (table sorted by this)
    std::vector<uint32_t> seg_group;  // first segment of each launch group
    uint32_t* d_seg = nullptr;
    uint64_t* d_small = nullptr;  // FRI coset tables (rate_bits) | quotient coset tables (qdb) | w_A^-i
    uint64_t *d_coset_base = nullptr, *d_q_coset_base = nullptr, *d_q_zh_inv = nullptr, *d_q_wR_inv = nullptr,
             *d_q_chunk_scale = nullptr, *d_wA_inv = nullptr;
    uint64_t* d_l_inv = nullptr;               // [2^qdb][n]
    uint64_t* d_periodic = nullptr;            // [n_periodic][2^qdb][period]
    std::vector<uint64_t> periodic;            // canonicalised host copy
    const uint64_t* d_q_inv_scale_br = nullptr;  // ctx-owned
    hipEvent_t ev[NLX_MAX_STAGES + 1]{};
    const char* stage_names[NLX_MAX_STAGES]{};
    uint32_t n_stages = 0;
    bool timed = false;
};

static size_t stark_proof_max_bytes(const nlx_stark_desc& d, uint32_t n_rounds) {
    const size_t capb = (size_t)32 << d.cap_height;
    const unsigned log_L = d.degree_bits + d.rate_bits;
    uint32_t n_trace_oracles = 0;
    for (uint32_t r = 0; r < (d.n_rounds ? d.n_rounds : 1u); r++) {
        const uint32_t c = d.n_rounds ? d.round_cols[r] : d.n_cols;
        n_trace_oracles += d.batch_cols && c > d.batch_cols ? (c + d.batch_cols - 1) / d.batch_cols : 1;
    }
    const uint32_t nq = d.num_challenges * d.quotient_degree_factor, n_oracles = n_trace_oracles + 1;
    size_t bytes = n_oracles * capb + 16 * (size_t)(2 * d.n_cols + nq) + n_rounds * capb;
    size_t per_query = (size_t)(d.n_cols + nq) * 8 + n_oracles * (1 + 32 * (size_t)log_L) +
                       n_rounds * (((size_t)16 << d.fri_arity_bits) + 1 + 32 * (size_t)log_L);
    bytes += per_query * d.fri_num_queries + ((size_t)16 << d.degree_bits) + 8 + 8 + 8 * (size_t)d.num_public_inputs;
    for (uint32_t r = 0; r < d.n_rounds && r < 3; r++) bytes += 8 * (size_t)d.round_values[r];
    return bytes + 64;
}

// What circuit_digest is to a plonky2 circuit: the whole description (config, program, periodic columns, round
// structure) hashed into four field elements.  hash_no_pad over 32-bit halves (each a canonical field element): 24 shape
// words, then every program word as (lo, hi) - the stored program already has its CONST immediates reduced -, then every
// periodic value (reduced) as (lo, hi).
static void air_digest_host(const nlx_stark_desc& d, const std::vector<uint64_t>& prog, const std::vector<uint64_t>& periodic,
                            uint64_t out[4]) {
    std::vector<uint64_t> v;
    v.reserve(24 + 2 * prog.size() + 2 * periodic.size());
    const uint32_t shape[14] = {d.degree_bits, d.n_cols, d.num_challenges, d.rate_bits, d.cap_height, d.quotient_degree_factor,
                                d.fri_pow_bits, d.fri_num_queries, d.fri_arity_bits, d.fri_final_poly_bits,
                                d.num_public_inputs, d.n_words, d.n_periodic, d.n_periodic ? d.period_bits : 0u};
    for (uint32_t x : shape) v.push_back(x);
    v.push_back(d.n_rounds);
    for (uint32_t r = 0; r < 3; r++) v.push_back(r < d.n_rounds ? d.round_cols[r] : 0);
    for (uint32_t r = 0; r < 3; r++) v.push_back(r < d.n_rounds ? d.round_challenges[r] : 0);
    for (uint32_t r = 0; r < 3; r++) v.push_back(r < d.n_rounds ? d.round_values[r] : 0);
    if (d.leaf_group_cols || d.openings_group) {  // only when used: digests of plain-starky statements stay what they were
        v.push_back(d.leaf_group_cols);
        v.push_back(d.openings_group);
    }
    if (d.batch_cols) {   // likewise: a statement without batches keeps its digest
        v.push_back(0xB47C4u);
        v.push_back(d.batch_cols);
    }
    for (uint64_t w : prog) { v.push_back(w & 0xFFFFFFFFu); v.push_back(w >> 32); }
    for (uint64_t w : periodic) { v.push_back(w & 0xFFFFFFFFu); v.push_back(w >> 32); }
    hash_no_pad_host(v.data(), v.size(), out);
}

extern "C" {

int32_t nlx_stark_build(nlx_ctx* ctx, const nlx_stark_desc* desc, nlx_stark** out) NLX_TRY {
    if (!ctx) return NLX_E_INVAL;
    if (!desc || !out || (!desc->program && desc->n_words)) return ctx->fail(NLX_E_INVAL, "NULL argument");
    *out = nullptr;
    const nlx_stark_desc& d = *desc;
    const uint32_t q = d.quotient_degree_factor;
    if (d.num_challenges < 1 || d.num_challenges > 2) return ctx->fail(NLX_E_UNSUPPORTED, "num_challenges must be 1 or 2");
    if (d.rate_bits < 1 || d.rate_bits > 3) return ctx->fail(NLX_E_UNSUPPORTED, "rate_bits must be in [1, 3]");
    if (!q || (q & (q - 1)) || q > (1u << d.rate_bits))
        return ctx->fail(NLX_E_INVAL, "quotient_degree_factor must be a power of two <= 2^rate_bits");
    if (d.fri_arity_bits < 2 || d.fri_arity_bits > 4) return ctx->fail(NLX_E_UNSUPPORTED, "fri_arity_bits must be in [2, 4]");
    if (d.leaf_group_cols && (d.leaf_group_cols < 8 || d.leaf_group_cols > 4096)) return ctx->fail(NLX_E_RANGE, "leaf_group_cols must be 0 or in [8, 4096]");
    if (d.openings_group && (d.openings_group < 8 || d.openings_group > 4096)) return ctx->fail(NLX_E_RANGE, "openings_group must be 0 or in [8, 4096]");
    if (d.degree_bits < 4 || d.degree_bits < d.fri_arity_bits || d.degree_bits + d.rate_bits > 30)
        return ctx->fail(NLX_E_RANGE, "degree_bits out of range");
    if (d.fri_num_queries > 128 || d.fri_num_queries == 0 || d.cap_height > 6 || d.cap_height > d.degree_bits + d.rate_bits)
        return ctx->fail(NLX_E_RANGE, "FRI parameters out of range");
    if (d.n_cols == 0 || d.n_cols > 8192 || d.num_public_inputs > 4096 || d.n_words > (1u << 20))
        return ctx->fail(NLX_E_RANGE, "AIR shape out of range");
    if (d.n_periodic > NLX_AIR_MAX_PERIODIC || (d.n_periodic && (!d.periodic || d.period_bits > d.degree_bits || d.period_bits > 16)))
        return ctx->fail(NLX_E_RANGE, "periodic columns out of range");
    if (d.n_rounds > 3) return ctx->fail(NLX_E_RANGE, "at most three commitment rounds");
    if (d.batch_cols) {
        if (d.batch_cols < 8 || d.batch_cols > 65535) return ctx->fail(NLX_E_RANGE, "batch_cols must be 0 or 8 .. 65535");
        if (d.leaf_group_cols) return ctx->fail(NLX_E_INVAL, "batch_cols and leaf_group_cols exclude each other");
        uint32_t n_or = 0;
        for (uint32_t r = 0; r < (d.n_rounds ? d.n_rounds : 1u); r++) n_or += ((d.n_rounds ? d.round_cols[r] : d.n_cols) + d.batch_cols - 1) / d.batch_cols;
        if (n_or > NLX_STARK_MAX_ORACLES - 1) return ctx->fail(NLX_E_RANGE, "batch_cols %u makes %u batches (at most %u)", d.batch_cols, n_or, NLX_STARK_MAX_ORACLES - 1);
    }
    uint32_t n_round_challenges = 0;
    if (d.n_rounds) {
        uint32_t tot = 0;
        for (uint32_t r = 0; r < d.n_rounds; r++) {
            if (d.round_cols[r] == 0 || d.round_challenges[r] > 16 || d.round_values[r] > 64)
                return ctx->fail(NLX_E_RANGE, "round %u: columns / challenges / values out of range", r);
            tot += d.round_cols[r];
            n_round_challenges += d.round_challenges[r] + d.round_values[r];  // everything after the public inputs
        }
        if (tot != d.n_cols) return ctx->fail(NLX_E_INVAL, "round_cols must add up to n_cols");
    }
    // program validation: opcodes, operand ranges, no register read before it is written
    std::vector<uint64_t> prog(d.program, d.program + d.n_words);
    uint32_t n_regs = 1, n_emits = 0, cur_regs = 1;
    std::vector<uint32_t> seg_bounds, seg_emits, seg_regs;  // per boundary: word index, constraints emitted before it, registers of the segment it closes
    {
        bool written[NLX_AIR_NUM_REGS] = {false};
        for (uint32_t pc = 0; pc < d.n_words; pc++) {
            const uint64_t w = prog[pc];
            const uint32_t op = (uint32_t)(w & 0xFF), dst = (uint32_t)((w >> 8) & 0xFFFF);
            const uint32_t a = (uint32_t)((w >> 24) & 0xFFFF), b = (uint32_t)((w >> 40) & 0xFFFF);
            if (op > NLX_AIR_MAC) return ctx->fail(NLX_E_INVAL, "AIR word %u: unknown opcode %u", pc, op);
            if (op == NLX_AIR_EMIT_LOGUP) {
                const uint32_t k = (uint32_t)(w >> 56) & 0x3F;
                if ((w >> 62) != 0 || a >= d.n_cols || b + 1 >= d.n_cols || (dst != 0xFFFF && dst >= d.n_cols) || k + 1 >= n_round_challenges)
                    return ctx->fail(NLX_E_INVAL, "AIR word %u: EMIT_LOGUP column / challenge out of range", pc);
                n_emits += 2;
                continue;
            }
            if (op == NLX_AIR_SEGMENT) {
                if (w != NLX_AIR_SEGMENT) return ctx->fail(NLX_E_INVAL, "AIR word %u: operands on a segment boundary", pc);
                if (seg_bounds.size() + 2 > NLX_AIR_MAX_SEGMENTS) return ctx->fail(NLX_E_RANGE, "AIR: too many segments");
                seg_bounds.push_back(pc);
                seg_emits.push_back(n_emits);
                seg_regs.push_back(cur_regs);
                cur_regs = 1;
                for (bool& wr : written) wr = false;
                continue;
            }
            if (op >= NLX_AIR_EMIT_TRANSITION && op <= NLX_AIR_EMIT) n_emits++;
            if (op == NLX_AIR_EMIT_BOOL) n_emits += b ? b : 1;
            if (op == NLX_AIR_LOADV) {
                // a hint: the following words are validated as the ordinary loads they are
                if (dst < 1 || dst > 8 || pc + dst >= d.n_words) return ctx->fail(NLX_E_INVAL, "AIR word %u: LOADV count", pc);
                uint32_t seen_dst[8];
                for (uint32_t i = 0; i < dst; i++) {
                    const uint64_t w2 = prog[pc + 1 + i];
                    const uint32_t op2 = (uint32_t)(w2 & 0xFF), d2 = (uint32_t)((w2 >> 8) & 0xFFFF);
                    if (op2 != NLX_AIR_LOCAL && op2 != NLX_AIR_NEXT && op2 != NLX_AIR_PUBLIC && op2 != NLX_AIR_PERIODIC)
                        return ctx->fail(NLX_E_INVAL, "AIR word %u: LOADV must be followed by plain loads", pc);
                    for (uint32_t j = 0; j < i; j++)
                        if (seen_dst[j] == d2) return ctx->fail(NLX_E_INVAL, "AIR word %u: LOADV destinations must be distinct", pc);
                    seen_dst[i] = d2;
                }
                continue;
            }
            const bool three = (op >= NLX_AIR_XOR3 && op <= NLX_AIR_MAJ) || op == NLX_AIR_MAC;
            const bool writes = op <= NLX_AIR_MUL || (op >= NLX_AIR_PERIODIC && op <= NLX_AIR_PACK_NEXT) || three;
            if (three) {
                const uint32_t c3 = (uint32_t)(w >> 56) & 0x3F;
                if (a >= NLX_AIR_NUM_REGS || b >= NLX_AIR_NUM_REGS || !written[a] || !written[b] || !written[c3])
                    return ctx->fail(NLX_E_INVAL, "AIR word %u: reads an unwritten register", pc);
            }
            if ((op == NLX_AIR_PACK_LOCAL || op == NLX_AIR_PACK_NEXT) && (b < 1 || b > 32 || a + b > d.n_cols))
                return ctx->fail(NLX_E_INVAL, "AIR word %u: PACK range out of the trace", pc);
            if (op == NLX_AIR_EMIT_BOOL && a + (b ? b : 1) > d.n_cols) return ctx->fail(NLX_E_INVAL, "AIR word %u: column out of range", pc);
            if ((w >> 56) != 0 && op != NLX_AIR_ADD && op != NLX_AIR_SUB && !three) return ctx->fail(NLX_E_INVAL, "AIR word %u: shift on a non-ADD/SUB word", pc);
            if ((w >> 62) != 0) return ctx->fail(NLX_E_INVAL, "AIR word %u: reserved bits set", pc);
            if (writes && dst >= NLX_AIR_NUM_REGS) return ctx->fail(NLX_E_INVAL, "AIR word %u: register out of range", pc);
            if ((op == NLX_AIR_LOCAL || op == NLX_AIR_NEXT) && a >= d.n_cols) return ctx->fail(NLX_E_INVAL, "AIR word %u: column out of range", pc);
            if (op == NLX_AIR_PUBLIC && a >= d.num_public_inputs + n_round_challenges)
                return ctx->fail(NLX_E_INVAL, "AIR word %u: public input / challenge out of range", pc);
            if (op == NLX_AIR_PERIODIC && a >= d.n_periodic) return ctx->fail(NLX_E_INVAL, "AIR word %u: periodic column out of range", pc);
            if (op >= NLX_AIR_ADD && op <= NLX_AIR_EMIT) {
                const bool two_src = op <= NLX_AIR_MUL;
                if (a >= NLX_AIR_NUM_REGS || !written[a] || (two_src && (b >= NLX_AIR_NUM_REGS || !written[b])))
                    return ctx->fail(NLX_E_INVAL, "AIR word %u: reads an unwritten register", pc);
            }
            if (op == NLX_AIR_CONST) {
                if (pc + 1 >= d.n_words) return ctx->fail(NLX_E_INVAL, "AIR: CONST without immediate");
                pc++;
                prog[pc] %= gl::P;
            }
            if (writes) {
                written[dst] = true;
                if (dst + 1 > n_regs) n_regs = dst + 1;
                if (dst + 1 > cur_regs) cur_regs = dst + 1;
            }
        }
    }
    (void)hipSetDevice(ctx->device);
    nlx_stark* s = new (std::nothrow) nlx_stark();
    if (!s) return ctx->fail(NLX_E_NOMEM, "host allocation failed");
    s->ctx = ctx;
    s->d = d;
    s->program.swap(prog);
    s->d.program = s->program.data();
    s->n_regs = n_regs;
    {
        // Segment table, sorted by register need: the quotient kernel is launched once per group of segments with a
        // similar register file, because the LDS a launch reserves per wave is that of its hungriest segment (one
        // 64-register segment among 17-register ones tripled the whole kernel's time before this).  The order of the
        // table is free: every segment carries its own alpha power.
        seg_bounds.push_back(d.n_words);
        seg_emits.push_back(n_emits);
        seg_regs.push_back(cur_regs);
        struct SegRow { uint32_t lo, hi, after, regs; };
        std::vector<SegRow> rows;
        uint32_t lo = 0;
        for (size_t i = 0; i < seg_bounds.size(); i++) {
            rows.push_back(SegRow{lo, seg_bounds[i], n_emits - seg_emits[i], seg_regs[i]});
            lo = seg_bounds[i] + 1;
        }
        std::stable_sort(rows.begin(), rows.end(), [](const SegRow& a, const SegRow& b) { return a.regs < b.regs; });
        for (const SegRow& r : rows) {
            s->seg.push_back(r.lo);
            s->seg.push_back(r.hi);
            s->seg_after.push_back(r.after);
            s->seg_regs.push_back(r.regs);
        }
        // groups: a new launch where the register need grows by more than a quarter over the group's first segment
        for (uint32_t i = 0; i < rows.size(); i++)
            if (i == 0 || rows[i].regs > s->seg_regs[s->seg_group.back()] + s->seg_regs[s->seg_group.back()] / 4 + 2) s->seg_group.push_back(i);
    }
    s->n_rounds = d.n_rounds ? d.n_rounds : 1;
    for (uint32_t r = 0; r < s->n_rounds; r++) {
        s->round_cols[r] = d.n_rounds ? d.round_cols[r] : d.n_cols;
        s->round_challenges[r] = d.n_rounds ? d.round_challenges[r] : 0;
        s->round_values[r] = d.n_rounds ? d.round_values[r] : 0;
    }
    s->n_round_challenges = n_round_challenges;
    s->nq = d.num_challenges * q;
    while ((1u << s->qdb) < q) s->qdb++;
    s->n_fri_rounds = fri_num_rounds(d.degree_bits, d.rate_bits, d.cap_height, d.fri_arity_bits, d.fri_final_poly_bits);
    auto fail = [&](int32_t code) {
        nlx_stark_destroy(s);
        return code;
    };
    const unsigned log_n = d.degree_bits;
    int32_t rc = ctx->ensure_tables(log_n + d.rate_bits);
    if (rc) return fail(rc);
    rc = ctx->get_coset_scale(log_n, s->qdb, &s->d_q_inv_scale_br, true);
    if (rc) return fail(rc);
    {
        const uint32_t R = 1u << d.rate_bits, Q = 1u << s->qdb, A = 1u << d.fri_arity_bits;
        std::vector<uint64_t> small(4 * R + 4 * Q + A);
        coset_tables_host(log_n, d.rate_bits, small.data());
        coset_tables_host(log_n, s->qdb, small.data() + 4 * R);
        const uint64_t w_A_inv = gl::inv(gl::root_of_unity(d.fri_arity_bits));
        for (uint32_t i = 0; i < A; i++) small[4 * R + 4 * Q + i] = gl::pow(w_A_inv, i);
        s->d_small = (uint64_t*)ctx->alloc(small.size() * 8);
        s->d_program = (uint64_t*)ctx->alloc((size_t)(d.n_words ? d.n_words : 1) * 8);
        s->d_l_inv = (uint64_t*)ctx->alloc(((size_t)8 << (log_n + s->qdb)));
        s->d_seg = (uint32_t*)ctx->alloc(s->seg.size() * 4);
        if (!s->d_small || !s->d_program || !s->d_l_inv || !s->d_seg) return fail(NLX_E_NOMEM);
        hipError_t e = hipMemcpy(s->d_small, small.data(), small.size() * 8, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(s->d_seg, s->seg.data(), s->seg.size() * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess && d.n_words) e = hipMemcpy(s->d_program, s->program.data(), (size_t)d.n_words * 8, hipMemcpyHostToDevice);
        if (e != hipSuccess) return fail(ctx->hip_fail(e, "hipMemcpy(tables)"));
        s->d_coset_base = s->d_small;
        s->d_q_coset_base = s->d_small + 4 * R;
        s->d_q_zh_inv = s->d_q_coset_base + Q;
        s->d_q_wR_inv = s->d_q_zh_inv + Q;
        s->d_q_chunk_scale = s->d_q_wR_inv + Q;
        s->d_wA_inv = s->d_small + 4 * R + 4 * Q;
        launch_l0_table(ctx->stream, s->d_l_inv, log_n, s->qdb, s->d_q_coset_base, ctx->tables.fwd[log_n]);
        if (d.n_periodic) {
            // P_a = interpolation of column a over the period-th roots of unity, then its values at
            // y = (g w_{nQ}^r')^(n/period) * w_period^k for every quotient coset r' and k < period (host transforms:
            // a period is at most 2^16)
            const uint32_t period = 1u << d.period_bits;
            s->periodic.assign(d.periodic, d.periodic + (size_t)d.n_periodic * period);
            for (auto& v : s->periodic) v %= gl::P;
            s->d.periodic = s->periodic.data();
            std::vector<uint64_t> coeffs(period), shifted(period), table((size_t)d.n_periodic * Q * period);
            for (uint32_t a = 0; a < d.n_periodic; a++) {
                std::copy(s->periodic.begin() + (size_t)a * period, s->periodic.begin() + (size_t)(a + 1) * period, coeffs.begin());
                host_ntt(coeffs, d.period_bits, true);
                for (uint32_t r = 0; r < Q; r++) {
                    // P_a on the coset y0 * <w_period>, y0 = (g w^r')^(n/period): transform of coeffs[m] * y0^m
                    const uint64_t y0 = gl::exp_pow2(small[4 * R + r], log_n - d.period_bits);
                    uint64_t pw = 1;
                    for (uint32_t m = 0; m < period; m++) {
                        shifted[m] = gl::mul(coeffs[m], pw);
                        pw = gl::mul(pw, y0);
                    }
                    host_ntt(shifted, d.period_bits, false);
                    std::copy(shifted.begin(), shifted.end(), table.begin() + ((size_t)a * Q + r) * period);
                }
            }
            s->d_periodic = (uint64_t*)ctx->alloc(table.size() * 8);
            if (!s->d_periodic) return fail(NLX_E_NOMEM);
            hipError_t e2 = hipMemcpy(s->d_periodic, table.data(), table.size() * 8, hipMemcpyHostToDevice);
            if (e2 != hipSuccess) return fail(ctx->hip_fail(e2, "hipMemcpy(periodic)"));
        }
    }
    air_digest_host(s->d, s->program, s->periodic, s->air_digest);
    // a kernel generated from these very words (same hash, same length)?  Two challenges, whole blocks of AIRGEN_BLOCK points and
    // an LDE of at most 2^28 rows only (the generated code addresses a column by a 32-bit byte offset); NLX_AIR_VM=1 keeps the interpreter (the parity reference of the generated code, tests/test_gpu_airgen.py)
    {
        const char* force_vm = getenv("NLX_AIR_VM");
        if (!(force_vm && force_vm[0] == '1') && s->d.num_challenges == 2 && ((size_t)1 << s->d.degree_bits) >= AIRGEN_BLOCK &&
            s->d.degree_bits + s->d.rate_bits <= 28)
            s->gen = airgen_find(airgen_program_hash(s->program.data(), s->program.size()), (uint32_t)s->program.size());
    }
    for (int i = 0; i <= NLX_MAX_STAGES; i++)
        if (hipEventCreate(&s->ev[i]) != hipSuccess) return fail(ctx->fail(NLX_E_HIP, "hipEventCreate failed"));
    {
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return fail(ctx->hip_fail(e, "hipStreamSynchronize"));
    }
    *out = s;
    return NLX_OK;
} NLX_CATCH(ctx)

void nlx_stark_destroy(nlx_stark* s) NLX_TRY {
    if (!s) return;
    nlx_ctx* ctx = s->ctx;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->release(s->d_small);
    ctx->release(s->d_program);
    ctx->release(s->d_seg);
    ctx->release(s->d_l_inv);
    ctx->release(s->d_periodic);
    for (int i = 0; i <= NLX_MAX_STAGES; i++)
        if (s->ev[i]) (void)hipEventDestroy(s->ev[i]);
    delete s;
} NLX_CATCH_VOID(nullptr)

int32_t nlx_stark_quotient_kernel(const nlx_stark* s) NLX_TRY { return s && s->gen ? 1 : 0; } NLX_CATCH(nullptr)

size_t nlx_stark_proof_max_bytes(const nlx_stark* s) NLX_TRY {
    return s ? stark_proof_max_bytes(s->d, s->n_fri_rounds) : 0;
} NLX_CATCH_VALUE(nullptr, 0)

int32_t nlx_stark_stage_times(const nlx_stark* s, uint32_t* n_stages, const char** names_out, float* ms_out) NLX_TRY {
    if (!s || !n_stages) return NLX_E_INVAL;
    if (!s->timed) { *n_stages = 0; return NLX_OK; }
    *n_stages = s->n_stages;
    for (uint32_t i = 0; i < s->n_stages; i++) {
        if (names_out) names_out[i] = s->stage_names[i];
        if (ms_out) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, s->ev[i], s->ev[i + 1]) != hipSuccess) ms = -1.f;
            ms_out[i] = ms;
        }
    }
    return NLX_OK;
} NLX_CATCH(nullptr)

int32_t nlx_stark_prove_rounds(nlx_stark* s, nlx_round_fn round_fn, void* user, const uint64_t* public_inputs,
                               uint8_t* proof_out, size_t proof_cap, size_t* proof_len) NLX_TRY {
    if (!s) return NLX_E_INVAL;
    nlx_ctx* ctx = s->ctx;
    const nlx_stark_desc& d = s->d;
    if (!round_fn || !proof_out || !proof_len || (!public_inputs && d.num_public_inputs))
        return ctx->fail(NLX_E_INVAL, "NULL argument");
    *proof_len = 0;
    for (uint32_t i = 0; i < d.num_public_inputs; i++)
        if (public_inputs[i] >= gl::P) return ctx->fail(NLX_E_RANGE, "public input %u is not canonical", i);
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    const unsigned log_n = d.degree_bits, cap_h = d.cap_height, qdb = s->qdb;
    const size_t n = (size_t)1 << log_n, L = n << d.rate_bits, capw = (size_t)4 << cap_h;
    const uint32_t nc = d.num_challenges, ncols = d.n_cols, nq = s->nq, NRD = s->n_rounds;
    int32_t rc = NLX_OK;
    std::vector<void*> scratch;
    auto dalloc = [&](size_t bytes) -> uint64_t* {
        void* p = ctx->alloc(bytes);
        if (p) scratch.push_back(p);
        return (uint64_t*)p;
    };
    nlx_commit* cr[3] = {nullptr, nullptr, nullptr};
    nlx_commit* cq = nullptr;
    uint32_t col0[4] = {0, 0, 0, 0};
    s->n_stages = 0;
    s->timed = false;
    auto stage = [&](const char* name) {
        if (s->n_stages < NLX_MAX_STAGES) {
            (void)hipEventRecord(s->ev[s->n_stages], st);
            s->stage_names[s->n_stages++] = name;
        }
    };
    Writer w{proof_out, 0, proof_cap};
    Challenger ch;
    // The transcript opens with the statement - the AIR digest, then the public inputs - before any commitment: the
    // public inputs enter the AIR linearly, so a transcript without them would let a prover choose them after alpha and
    // zeta are known (the starky of the pinned era had that gap; plonky2's own prover observes circuit_digest and the
    // public-input hash first, and so does this one).
    ch.observe(s->air_digest, 4);
    ch.observe(public_inputs, d.num_public_inputs);
    std::vector<uint64_t> cap(capw);
    // values readable by NLX_AIR_PUBLIC: the public inputs, then the verifier challenges in the order drawn
    std::vector<uint64_t> values(public_inputs, public_inputs + d.num_public_inputs);
    std::vector<uint64_t> round_vals;
    std::vector<const uint64_t*> h_cols(ncols);
#define CHECK(x) do { rc = (x); if (rc) goto done; } while (0)
#define HIPCHK(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { rc = ctx->hip_fail(e__, #call); goto done; } } while (0)
#define CHECK_ALLOC(p) do { if (!(p)) { rc = NLX_E_NOMEM; goto done; } } while (0)
    {
        // ---- trace commitments, one per round (prover.rs: PolynomialBatch::from_values(trace_poly_values, ..);
        //      starkyx: one TraceWriter round per commitment, challenges drawn in between) ----
        stage("commit_trace");
        for (uint32_t r = 0; r < NRD; r++) {
            const uint32_t rcols = s->round_cols[r];
            const uint32_t n_rv = s->round_values[r];
            uint64_t rv[64];
            const uint64_t* tr_ptr = round_fn(user, r, values.data() + d.num_public_inputs, (uint32_t)(values.size() - d.num_public_inputs),
                                              n_rv ? rv : nullptr);
            if (!tr_ptr) { rc = ctx->fail(NLX_E_INVAL, "round %u: the round callback returned NULL", r); goto done; }
            for (uint32_t k = 0; k < n_rv; k++) rv[k] %= gl::P;
            Staged tr(ctx, tr_ptr, (size_t)rcols * n * 8, true, false);
            CHECK(tr.status);
            // one PolynomialBatch, or (batch_cols) ceil(rcols / batch_cols) of them: transformed together, a tree and a cap each, the
            // caps into the transcript and the proof in batch order
            CHECK(commit_build(ctx, tr.as<uint64_t>(), n, CommitInput::ValuesNatural, rcols, log_n, d.rate_bits, cap_h, &cr[r], d.leaf_group_cols,
                               d.batch_cols));
            for (uint32_t k = 0; k < cr[r]->n_trees; k++) {
                CHECK(fetch(ctx, cap.data(), cr[r]->cap + (size_t)k * cr[r]->tree_words, capw * 8));
                w.u64s(cap.data(), capw);
                ch.observe(cap.data(), capw);
            }
            if (n_rv) {  // the round's values: into the transcript before its challenges, into the proof's tail later
                ch.observe(rv, n_rv);
                values.insert(values.end(), rv, rv + n_rv);
                round_vals.insert(round_vals.end(), rv, rv + n_rv);
            }
            for (uint32_t k = 0; k < s->round_challenges[r]; k++) values.push_back(ch.challenge());
            for (uint32_t c = 0; c < rcols; c++) h_cols[col0[r] + c] = cr[r]->lde + (size_t)c * L;
            col0[r + 1] = col0[r] + rcols;
        }
        uint64_t alphas[2] = {0, 0};
        for (uint32_t i = 0; i < nc; i++) alphas[i] = ch.challenge();

        // ---- compute_quotient_polys ----
        stage("quotient_eval");
        const size_t Q = n << qdb;
        // public inputs ++ round challenges, then each segment's alpha^(constraints after it) per challenge
        const uint32_t n_seg = (uint32_t)s->seg_after.size();
        const size_t n_values = values.size();
        for (uint32_t sg = 0; sg < n_seg; sg++)
            for (uint32_t i = 0; i < 2; i++) values.push_back(gl::pow(alphas[i], s->seg_after[sg]));
        uint64_t* d_pis = dalloc(values.size() * 8);
        const uint64_t** d_cols = (const uint64_t**)dalloc((size_t)ncols * 8);
        uint64_t* d_qvals = dalloc((size_t)nc * Q * 8);
        uint64_t* d_qchunks = dalloc((size_t)nc * Q * 8);
        uint64_t* d_part = n_seg > 1 ? dalloc((size_t)n_seg * nc * Q * 8) : nullptr;
        CHECK_ALLOC(d_pis && d_cols && d_qvals && d_qchunks && (n_seg == 1 || d_part));
        HIPCHK(hipMemcpyAsync(d_pis, values.data(), values.size() * 8, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(d_cols, h_cols.data(), (size_t)ncols * 8, hipMemcpyHostToDevice, st));
        {
            AirParams ap{};
            ap.cols = d_cols; ap.program = s->d_program; ap.pis = d_pis;
            ap.coset_base = s->d_q_coset_base; ap.zh_inv = s->d_q_zh_inv; ap.l_inv = s->d_l_inv;
            ap.periodic = s->d_periodic; ap.period_bits = d.period_bits;
            ap.w_n_table = ctx->tables.fwd[log_n];
            ap.out = d_qvals;
            ap.alphas[0] = alphas[0]; ap.alphas[1] = alphas[1];
            ap.g_inv = gl::inv(gl::root_of_unity(log_n));
            ap.log_n = log_n; ap.rate_bits = d.rate_bits; ap.qdb = qdb; ap.n_words = d.n_words; ap.nc = nc;
            ap.n_regs = s->n_regs; ap.n_pis = d.num_public_inputs;
            ap.seg = s->d_seg; ap.seg_mul = d_pis + n_values; ap.part = d_part; ap.n_seg = n_seg;
            // one wave per block: the LDS register file (n_regs x 64 lanes x 8 B) is the occupancy limiter, and
            // single-wave blocks pack the 160 KB of a CU at the finest granularity
            unsigned bs = 64;
            while (bs > n) bs >>= 1;
            ctx->begin_kernel("air_quotient", 8.0 * Q * (2.0 * ncols + nc));
            if (s->gen) s->gen->launch(st, (unsigned)(Q / AIRGEN_BLOCK), n_seg, ap);   // every segment of every point, straight-line code
            for (size_t gi = 0; !s->gen && gi < s->seg_group.size(); gi++) {
                const uint32_t first = s->seg_group[gi], last = gi + 1 < s->seg_group.size() ? s->seg_group[gi + 1] : n_seg;
                AirParams gp = ap;
                gp.seg = ap.seg + 2 * first;
                gp.seg_mul = ap.seg_mul + 2 * first;
                gp.part = n_seg > 1 ? ap.part + (size_t)first * nc * Q : nullptr;
                const size_t lds = (size_t)bs * s->seg_regs[last - 1] * 8;  // the group's largest register file
                hipLaunchKernelGGL(k_air_quotient, dim3((unsigned)(Q / bs), last - first), dim3(bs), lds, st, gp);
            }
            if (n_seg > 1)
                hipLaunchKernelGGL(k_air_combine, dim3((unsigned)((nc * Q + 255) / 256)), dim3(256), 0, st, d_part, s->d_q_zh_inv,
                                   d_qvals, n_seg, nc, log_n, log_n + qdb);
            ctx->end_kernel();
        }
        stage("quotient_intt");
        launch_intt_dif_cosets(st, ctx->tables, d_qvals, nc, log_n, qdb, s->d_q_inv_scale_br);
        launch_quotient_chunks(st, d_qvals, d_qchunks, log_n, qdb, nc, s->d_q_wR_inv, s->d_q_chunk_scale);
        stage("commit_quotient");
        CHECK(commit_build(ctx, d_qchunks, n, CommitInput::CoeffsBitrev, nq, log_n, d.rate_bits, cap_h, &cq, d.leaf_group_cols));
        CHECK(fetch(ctx, cap.data(), cq->cap, capw * 8));
        w.u64s(cap.data(), capw);
        ch.observe(cap.data(), capw);

        // ---- StarkOpeningSet: local = columns(zeta), next = columns(g zeta), quotient(zeta) ----
        stage("openings");
        uint64_t zeta[2], gzeta[2];
        ch.ext_challenge(zeta);
        {
            const uint64_t g = gl::root_of_unity(log_n);
            gzeta[0] = gl::mul(zeta[0], g);
            gzeta[1] = gl::mul(zeta[1], g);
        }
        const uint32_t n_open = ncols + nq;
        uint32_t widest = nq;
        for (uint32_t r = 0; r < NRD; r++) widest = s->round_cols[r] > widest ? s->round_cols[r] : widest;
        uint64_t* d_points = dalloc(2048);
        // with an openings digest the vector is zero-padded to whole runs and the runs' digests follow it
        const size_t open_words = (size_t)(n_open + ncols) * 2;
        const size_t og = d.openings_group, og_runs = og ? (open_words + og - 1) / og : 0;
        uint64_t* d_open = dalloc((og ? og_runs * og + 4 * og_runs : open_words) * 8);
        uint64_t* d_eval_scratch = dalloc(eval_scratch_words(widest, log_n) * 8);
        CHECK_ALLOC(d_points && d_open && d_eval_scratch);
        {
            uint64_t pts[4 + 2 * 2 * 32] = {zeta[0], zeta[1], gzeta[0], gzeta[1]};
            gl::Ext za{zeta[0], zeta[1]}, zb{gzeta[0], gzeta[1]};
            for (unsigned k = 0; k < 32; k++) {
                pts[4 + 2 * k] = za.a; pts[4 + 2 * k + 1] = za.b;
                pts[4 + 64 + 2 * k] = zb.a; pts[4 + 64 + 2 * k + 1] = zb.b;
                if (k + 1 < log_n) { za = gl::mul(za, za); zb = gl::mul(zb, zb); }
            }
            HIPCHK(hipMemcpyAsync(d_points, pts, sizeof pts, hipMemcpyHostToDevice, st));
            for (uint32_t r = 0; r < NRD; r++) {
                launch_eval_br(st, cr[r]->coeffs_br, n, s->round_cols[r], log_n, d_points, d_open + 2 * (size_t)col0[r],
                               d_eval_scratch, d_points + 4);
                launch_eval_br(st, cr[r]->coeffs_br, n, s->round_cols[r], log_n, d_points + 2,
                               d_open + 2 * (size_t)(n_open + col0[r]), d_eval_scratch, d_points + 4 + 64);
            }
            launch_eval_br(st, cq->coeffs_br, n, nq, log_n, d_points, d_open + 2 * (size_t)ncols, d_eval_scratch, d_points + 4);
            if (og) {
                if (og_runs * og > open_words) HIPCHK(hipMemsetAsync(d_open + open_words, 0, (og_runs * og - open_words) * 8, st));
                launch_hash_leaves_rowmajor(st, d_open, (uint32_t)og, og_runs, d_open + og_runs * og);
            }
        }
        std::vector<uint64_t> open(og ? og_runs * og + 4 * og_runs : open_words);
        CHECK(fetch(ctx, open.data(), d_open, open.size() * 8));
        const uint64_t* o_local = open.data();
        const uint64_t* o_q = o_local + 2 * (size_t)ncols;
        const uint64_t* o_next = open.data() + 2 * (size_t)n_open;
        w.u64s(o_local, 2 * (size_t)ncols);
        w.u64s(o_next, 2 * (size_t)ncols);
        w.u64s(o_q, 2 * (size_t)nq);
        // observe_openings(&openings.to_fri_openings()): zeta batch (local ++ quotient), then the g*zeta batch - every value, or
        // (openings_group) the digest of the runs' digests the device has just made
        if (og) {
            uint64_t dig[4];
            hash_no_pad_host(open.data() + og_runs * og, 4 * og_runs, dig);
            ch.observe(dig, 4);
        } else {
            ch.observe(open.data(), 2 * (size_t)n_open);
            ch.observe(o_next, 2 * (size_t)ncols);
        }

        // ---- FRI: Stark::fri_instance = [zeta: every round's columns ++ quotient], [g zeta: every round's columns] ----
        {
            FriProveArgs fa;
            nlx_commit views[NLX_STARK_MAX_ORACLES];   // a batch as a commitment of its own: its columns, its tree
            uint32_t no = 0;
            for (uint32_t r = 0; r < NRD; r++)
                for (uint32_t k = 0; k < cr[r]->n_trees; k++) {
                    views[no] = commit_view(cr[r], k);
                    fa.oracles[no] = &views[no];
                    fa.nz[no] = views[no].n_cols;
                    no++;
                }
            fa.oracles[no] = cq;
            fa.n_oracles = no + 1;
            for (int i = 0; i < 2; i++) { fa.zeta[i] = zeta[i]; fa.gzeta[i] = gzeta[i]; }
            fa.open0 = open.data();
            fa.open1 = o_next;
            fa.log_n = log_n; fa.rate_bits = d.rate_bits; fa.cap_height = cap_h; fa.arity_bits = d.fri_arity_bits;
            fa.pow_bits = d.fri_pow_bits; fa.n_queries = d.fri_num_queries; fa.n_rounds = s->n_fri_rounds;
            fa.d_coset_base = s->d_coset_base;
            fa.d_wA_inv = s->d_wA_inv;
            CHECK(fri_prove(ctx, fa, ch, w, scratch, stage));
        }
        w.usize(d.num_public_inputs);
        w.u64s(public_inputs, d.num_public_inputs);
        w.u64s(round_vals.data(), round_vals.size());
        stage("end");
        s->n_stages--;
        s->timed = true;
        if (w.overflow) { rc = ctx->fail(NLX_E_RANGE, "proof buffer too small (need %zu bytes)", nlx_stark_proof_max_bytes(s)); goto done; }
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
    for (uint32_t r = 0; r < 3; r++)
        if (cr[r]) nlx_commit_destroy(cr[r]);
    if (cq) nlx_commit_destroy(cq);
#undef CHECK
#undef HIPCHK
#undef CHECK_ALLOC
    return rc;
} NLX_CATCH(nullptr)

static const uint64_t* single_round_fn(void* user, uint32_t round, const uint64_t*, uint32_t, uint64_t*) {
    return round == 0 ? (const uint64_t*)user : nullptr;
}

int32_t nlx_stark_prove(nlx_stark* s, const uint64_t* trace, const uint64_t* public_inputs, uint8_t* proof_out,
                        size_t proof_cap, size_t* proof_len) NLX_TRY {
    if (!s) return NLX_E_INVAL;
    if (!trace) return s->ctx->fail(NLX_E_INVAL, "NULL argument");
    if (s->n_rounds != 1) return s->ctx->fail(NLX_E_INVAL, "a multi-round STARK is proved with nlx_stark_prove_rounds");
    return nlx_stark_prove_rounds(s, single_round_fn, (void*)trace, public_inputs, proof_out, proof_cap, proof_len);
} NLX_CATCH(nullptr)

int32_t nlx_stark_batch_prove(nlx_stark* const* workers, uint32_t n_workers, nlx_prove_job* jobs, size_t n_jobs) NLX_TRY {
    if (!workers || n_workers == 0 || (!jobs && n_jobs)) return NLX_E_INVAL;
    for (uint32_t w = 0; w < n_workers; w++) {
        if (!workers[w]) return NLX_E_INVAL;
        for (uint32_t v = 0; v < w; v++)
            if (workers[v]->ctx == workers[w]->ctx)
                return workers[w]->ctx->fail(NLX_E_INVAL, "nlx_stark_batch_prove: workers must use distinct contexts");
    }
    std::atomic<size_t> next{0};
    auto run = [&](nlx_stark* s) {
        (void)hipSetDevice(s->ctx->device);
        for (;;) {
            const size_t j = next.fetch_add(1);
            if (j >= n_jobs) return;
            nlx_prove_job& job = jobs[j];
            job.proof_len = 0;
            job.status = nlx_stark_prove(s, job.wires, job.public_inputs, job.proof_out, job.proof_cap, &job.proof_len);
        }
    };
    if (n_workers == 1) {
        run(workers[0]);
    } else {
        std::vector<std::thread> threads;
        threads.reserve(n_workers);
        for (uint32_t w = 0; w < n_workers; w++) threads.emplace_back(run, workers[w]);
        for (auto& t : threads) t.join();
    }
    for (size_t j = 0; j < n_jobs; j++)
        if (jobs[j].status != NLX_OK) return jobs[j].status;
    return NLX_OK;
} NLX_CATCH(nullptr)

}  // extern "C"
