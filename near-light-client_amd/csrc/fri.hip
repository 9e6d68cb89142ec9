// Device FRI prover (see fri.hpp).  Value-domain folding: layer r holds the evaluations of the folded
// polynomial on its coset-major domain; nothing is ever transformed back to coefficients except the
// final polynomial.
#include "fri.hpp"
#include "poly.hpp"
#include "prover.hpp"

namespace nlx {

void coset_tables_host(unsigned log_n, unsigned bits, uint64_t* h) {
    const uint32_t R = 1u << bits;
    uint64_t *h_cb = h, *h_zh = h + R, *h_wR = h + 2 * R, *h_cs = h + 3 * R;
    const uint64_t w_L = gl::root_of_unity(log_n + bits), w_R = gl::root_of_unity(bits);
    const uint64_t g_n = gl::exp_pow2(gl::GEN, log_n);
    const uint64_t g_n_inv = gl::inv(g_n), R_inv = gl::inv((uint64_t)R);
    for (uint32_t r = 0; r < R; r++) {
        h_cb[r] = gl::mul(gl::GEN, gl::pow(w_L, r));
        h_zh[r] = gl::inv(gl::sub(gl::mul(g_n, gl::pow(w_R, r)), 1));  // 1 / (x^n - 1) on coset r
        h_wR[r] = gl::pow(gl::inv(w_R), r);
        h_cs[r] = gl::mul(gl::pow(g_n_inv, r), R_inv);
    }
}

#define FRI_CHECK(x) do { int32_t rc__ = (x); if (rc__) return rc__; } while (0)
#define FRI_HIP(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return ctx->hip_fail(e__, #call); } while (0)
#define FRI_ALLOC(p) do { if (!(p)) return NLX_E_NOMEM; } while (0)

int32_t fri_prove(nlx_ctx* ctx, const FriProveArgs& a, Challenger& ch, Writer& w, std::vector<void*>& scratch,
                  const std::function<void(const char*)>& stage) {
    hipStream_t st = ctx->stream;
    const unsigned log_n = a.log_n, log_L = a.log_n + a.rate_bits, cap_h = a.cap_height;
    const size_t L = (size_t)1 << log_L, capw = (size_t)4 << cap_h;
    const uint32_t arity = 1u << a.arity_bits, NR = a.n_rounds, NO = a.n_oracles;
    const nlx_commit* const* oracles = a.oracles;
    uint32_t n_open = 0;
    for (uint32_t o = 0; o < NO; o++) n_open += oracles[o]->n_cols;
    std::vector<uint64_t> cap(capw);
    auto dalloc = [&](size_t bytes) -> uint64_t* {
        void* p = ctx->alloc(bytes);
        if (p) scratch.push_back(p);
        return (uint64_t*)p;
    };

    stage("fri_combine");
    uint64_t fri_alpha[2];
    ch.ext_challenge(fri_alpha);
    uint64_t* d_fri_alpha_pows = dalloc((size_t)n_open * 16);
    uint64_t* d_fri_a = dalloc(L * 16);
    uint64_t* d_fri_b = dalloc((L >> a.arity_bits) * 16 + 256);
    FRI_ALLOC(d_fri_alpha_pows && d_fri_a && d_fri_b);
    launch_ext_pow_table(st, d_fri_alpha_pows, fri_alpha, n_open);
    {
        // reduced openings C0 = sum alpha^i open0_i (zeta batch), C1 = sum alpha^i open1_i (g zeta batch)
        const gl::Ext al{fri_alpha[0], fri_alpha[1]};
        gl::Ext c0{0, 0}, c1{0, 0}, ap{1, 0};
        for (uint32_t i = 0; i < n_open; i++) {
            c0 = gl::add(c0, gl::mul(ap, gl::Ext{a.open0[2 * i], a.open0[2 * i + 1]}));
            ap = gl::mul(ap, al);
        }
        ap = gl::Ext{1, 0};
        uint32_t nz_total = a.tail_cols;
        for (uint32_t o = 0; o < NO; o++) nz_total += a.nz[o];
        for (uint32_t i = 0; i < nz_total; i++) {
            c1 = gl::add(c1, gl::mul(ap, gl::Ext{a.open1[2 * (size_t)i], a.open1[2 * (size_t)i + 1]}));
            ap = gl::mul(ap, al);
        }
        FriCombineParams fp{};
        for (uint32_t o = 0; o < NO; o++) { fp.tables[o] = oracles[o]->lde; fp.n_cols[o] = oracles[o]->n_cols; }
        uint32_t nz_off = 0;
        for (uint32_t o = 0; o < NO; o++) { fp.nz[o] = a.nz[o]; fp.nz_off[o] = nz_off; nz_off += a.nz[o]; }
        if (a.tail_cols) {
            // the trailing column group: cut off its oracle's view, listed last in both batches
            if (a.tail_oracle >= NO) return ctx->fail(NLX_E_INVAL, "FRI: bad trailing column group");
            const nlx_commit* to = oracles[a.tail_oracle];
            if (a.tail_cols > to->n_cols || a.nz[a.tail_oracle] > to->n_cols - a.tail_cols)
                return ctx->fail(NLX_E_INVAL, "FRI: bad trailing column group");
            fp.n_cols[a.tail_oracle] = to->n_cols - a.tail_cols;
            fp.tables[FRI_VIEWS - 1] = to->lde + (size_t)(to->n_cols - a.tail_cols) * L;
            fp.n_cols[FRI_VIEWS - 1] = a.tail_cols;
            fp.nz[FRI_VIEWS - 1] = a.tail_cols;
            fp.nz_off[FRI_VIEWS - 1] = nz_off;
        }
        fp.alpha_pows = d_fri_alpha_pows;
        fp.coset_base = a.d_coset_base;
        fp.w_n_table = ctx->tables.fwd[log_n];
        fp.zeta[0] = a.zeta[0]; fp.zeta[1] = a.zeta[1]; fp.gzeta[0] = a.gzeta[0]; fp.gzeta[1] = a.gzeta[1];
        fp.c0[0] = c0.a; fp.c0[1] = c0.b; fp.c1[0] = c1.a; fp.c1[1] = c1.b;
        fp.alpha_nz[0] = ap.a; fp.alpha_nz[1] = ap.b;  // alpha^nz
        fp.out = d_fri_a;
        fp.log_n = log_n; fp.rate_bits = a.rate_bits;
        uint64_t* d_comb = nullptr;
        if (const size_t words = fri_combine_scratch_words(fp)) {
            d_comb = dalloc(words * 8);
            FRI_ALLOC(d_comb);
        }
        ctx->begin_kernel("fri_combine", 8.0 * L * n_open + 16.0 * L);
        launch_fri_combine(st, fp, d_comb);
        ctx->end_kernel();
    }
    // commit phase: layer values ping-pong between d_fri_a / d_fri_b; digests kept per layer
    stage("fri_commit_phase");
    std::vector<uint64_t*> layer_values(NR + 1), layer_digests(NR);
    std::vector<unsigned> layer_log_n(NR + 1);
    layer_values[0] = d_fri_a;
    layer_log_n[0] = log_n;
    uint64_t shift = gl::GEN;
    for (uint32_t r = 0; r < NR; r++) {
        const unsigned ln = layer_log_n[r];
        const size_t n_leaves = (size_t)1 << (ln - a.arity_bits + a.rate_bits);
        uint64_t* dg = dalloc(merkle_digest_words(n_leaves, cap_h) * 8);
        FRI_ALLOC(dg);
        layer_digests[r] = dg;
        if (n_leaves <= ((size_t)1 << 13)) launch_fri_leaves_wide(st, layer_values[r], ln, a.rate_bits, a.arity_bits, dg);
        else launch_fri_leaves(st, layer_values[r], ln, a.rate_bits, a.arity_bits, dg);
        const uint64_t* d_cap = launch_merkle_levels(st, dg, n_leaves, cap_h);
        FRI_CHECK(fetch(ctx, cap.data(), d_cap, capw * 8));
        w.u64s(cap.data(), capw);
        ch.observe(cap.data(), capw);
        uint64_t beta[2];
        ch.ext_challenge(beta);
        uint64_t* nxt = (r == 0) ? d_fri_b : dalloc(((size_t)16 << (ln - a.arity_bits + a.rate_bits)) + 256);
        FRI_ALLOC(nxt);
        launch_fri_fold(st, layer_values[r], nxt, ln, a.rate_bits, a.arity_bits, beta, gl::inv(shift),
                        ctx->tables.inv[ln + a.rate_bits], a.d_wA_inv);
        layer_values[r + 1] = nxt;
        layer_log_n[r + 1] = ln - a.arity_bits;
        shift = gl::exp_pow2(shift, a.arity_bits);
    }
    // final polynomial
    const uint32_t final_len = 1u << layer_log_n[NR];
    uint64_t* d_final = dalloc((size_t)final_len * 16 + 256);
    FRI_ALLOC(d_final);
    launch_fri_final_coeffs(st, layer_values[NR], layer_log_n[NR], a.rate_bits, shift, d_final, final_len);
    std::vector<uint64_t> final_poly((size_t)final_len * 2);
    FRI_CHECK(fetch(ctx, final_poly.data(), d_final, final_poly.size() * 8));
    ch.observe(final_poly.data(), final_poly.size());

    // proof of work
    stage("fri_pow");
    uint64_t pow_witness = 0;
    {
        PowParams pp{};
        for (int i = 0; i < 12; i++) pp.state[i] = ch.state[i];
        for (unsigned i = 0; i < ch.n_in; i++) pp.state[i] = ch.in_buf[i];
        pp.pos = ch.n_in;
        pp.bits = a.pow_bits;
        pp.max_rounds = (uint64_t)1 << 24;
        unsigned long long* d_best = (unsigned long long*)dalloc(256);
        FRI_ALLOC(d_best);
        launch_pow_grind(st, pp, d_best);
        FRI_CHECK(fetch(ctx, &pow_witness, d_best, 8));
        if (pow_witness == ~0ull) return ctx->fail(NLX_E_RANGE, "proof of work: no witness found");
        ch.observe(pow_witness);
        (void)ch.challenge();
    }
    // query phase
    stage("fri_queries");
    const uint32_t NQ = a.n_queries;
    std::vector<uint64_t> qidx(NQ);
    for (uint32_t q = 0; q < NQ; q++) qidx[q] = ch.challenge() % L;
    {
        const unsigned plen0 = log_L - cap_h;
        // device layout of the answers (words)
        size_t off = 0;
        size_t rows_off[FRI_MAX_ORACLES], paths_off[FRI_MAX_ORACLES];
        for (uint32_t o = 0; o < NO; o++) {
            rows_off[o] = off; off += (size_t)NQ * oracles[o]->n_cols;
            paths_off[o] = off; off += (size_t)NQ * plen0 * 4;
        }
        std::vector<size_t> ev_off(NR), fp_off(NR);
        std::vector<unsigned> fplen(NR);
        for (uint32_t r = 0; r < NR; r++) {
            const unsigned lg = layer_log_n[r] - a.arity_bits + a.rate_bits;
            fplen[r] = lg > cap_h ? lg - cap_h : 0;
            ev_off[r] = off; off += (size_t)NQ * 2 * arity;
            fp_off[r] = off; off += (size_t)NQ * fplen[r] * 4;
        }
        uint64_t* d_ans = dalloc(off * 8 + 256);
        uint64_t* d_idx = dalloc((size_t)(NR + 1) * NQ * 8 + 256);
        FRI_ALLOC(d_ans && d_idx);
        FRI_HIP(hipMemcpyAsync(d_idx, qidx.data(), (size_t)NQ * 8, hipMemcpyHostToDevice, st));
        for (uint32_t o = 0; o < NO; o++) {
            launch_gather_rows(st, oracles[o]->lde, L, oracles[o]->n_cols, log_n, a.rate_bits, d_idx, NQ, d_ans + rows_off[o]);
            launch_gather_paths(st, oracles[o]->digests, log_L, cap_h, d_idx, NQ, d_ans + paths_off[o]);
        }
        unsigned total_shift = 0;
        for (uint32_t r = 0; r < NR; r++) {
            total_shift += a.arity_bits;
            uint64_t* idx_r = d_idx + (size_t)(r + 1) * NQ;
            launch_shift_indices(st, d_idx, idx_r, NQ, total_shift);
            launch_fri_gather_leaf(st, layer_values[r], layer_log_n[r], a.rate_bits, a.arity_bits, idx_r, NQ,
                                   d_ans + ev_off[r], (size_t)2 * arity);
            const unsigned lg = layer_log_n[r] - a.arity_bits + a.rate_bits;
            launch_gather_paths(st, layer_digests[r], lg, cap_h, idx_r, NQ, d_ans + fp_off[r]);
        }
        std::vector<uint64_t> ans(off);
        FRI_CHECK(fetch(ctx, ans.data(), d_ans, off * 8));
        for (uint32_t q = 0; q < NQ; q++) {
            for (uint32_t o = 0; o < NO; o++) {
                const uint32_t ncol = oracles[o]->n_cols;
                w.u64s(ans.data() + rows_off[o] + (size_t)q * ncol, ncol);
                w.u8((uint8_t)plen0);
                w.u64s(ans.data() + paths_off[o] + (size_t)q * plen0 * 4, (size_t)plen0 * 4);
            }
            for (uint32_t r = 0; r < NR; r++) {
                w.u64s(ans.data() + ev_off[r] + (size_t)q * 2 * arity, (size_t)2 * arity);
                w.u8((uint8_t)fplen[r]);
                w.u64s(ans.data() + fp_off[r] + (size_t)q * fplen[r] * 4, (size_t)fplen[r] * 4);
            }
        }
    }
    w.u64s(final_poly.data(), final_poly.size());
    w.u64s(&pow_witness, 1);
    return NLX_OK;
}

}  // namespace nlx
