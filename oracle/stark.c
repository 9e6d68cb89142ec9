/* ORACLE - TEST INFRASTRUCTURE ONLY (see gl.h and stark.h headers).
 *
 * Generic FRI (prove_openings / verify) over arbitrary batches, and the starky-style STARK prover and
 * verifier built on it.  Written the reference's way: row-major bit-reversed leaves, natural-order
 * coefficients, one zero-padded coset FFT per polynomial, coefficient-domain FRI folding.
 */
#include "stark.h"
#include "bytes.h"
#include <stdlib.h>
#include <string.h>

/* ---------------- helpers shared with plonk.c's style ---------------- */
static void observe_cap(orc_challenger* ch, const uint64_t* cap, unsigned cap_height) {
    orc_ch_observe_many(ch, cap, (size_t)4 << cap_height);
}
static gl2 gl2_add_base(gl2 x, uint64_t b) { return gl2_make(gl_add(x.a, b), x.b); }
static gl2 eval_base_poly_ext(const uint64_t* coeffs, size_t n, gl2 z) {
    gl2 acc = gl2_from(0);
    for (size_t i = n; i-- > 0;) acc = gl2_add_base(gl2_mul(acc, z), coeffs[i]);
    return acc;
}
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
static uint32_t fri_num_rounds(const orc_fri_params* p) {
    uint32_t degree_bits = p->degree_bits, r = 0;
    while (degree_bits > p->final_poly_bits && degree_bits + p->rate_bits >= p->cap_height + p->arity_bits) {
        if (degree_bits < p->arity_bits) break;
        degree_bits -= p->arity_bits;
        r++;
    }
    return r;
}

/* ---------------- FRI prover: PolynomialBatch::prove_openings + fri_proof ---------------- */
static void fri_prove(const orc_fri_params* p, const orc_fri_oracle* oracles, uint32_t n_oracles,
                      const orc_fri_batch* batches, uint32_t n_batches, orc_challenger* ch, wbuf* w) {
    const unsigned log_n = p->degree_bits, log_L = log_n + p->rate_bits, cap_h = p->cap_height;
    const size_t n = (size_t)1 << log_n, L = (size_t)1 << log_L;
    const uint32_t arity = 1u << p->arity_bits, R = fri_num_rounds(p);
    gl2 alpha = orc_ch_ext_challenge(ch);
    gl2* final_poly = (gl2*)calloc(L, sizeof(gl2));
    gl2* comp = (gl2*)malloc(sizeof(gl2) * n);
    for (uint32_t b = 0; b < n_batches; b++) {
        /* composition polynomial sum alpha^j f_j, then (F(X) - F(z)) / (X - z); earlier batches are
         * shifted by alpha^(number of polynomials of this batch) (ReducingFactor::shift_poly) */
        const orc_fri_batch* bt = &batches[b];
        gl2 apow = gl2_from(1);
        memset(comp, 0, sizeof(gl2) * n);
        for (uint32_t j = 0; j < bt->n_polys; j++) {
            const uint64_t* co = oracles[bt->oracle[j]].coeffs + (size_t)bt->poly[j] * n;
#pragma omp parallel for schedule(static)
            for (size_t i = 0; i < n; i++) comp[i] = gl2_add(comp[i], gl2_scale(apow, co[i]));
            apow = gl2_mul(apow, alpha);
        }
        if (b > 0)
            for (size_t i = 0; i < n; i++) final_poly[i] = gl2_mul(final_poly[i], apow);
        gl2 acc = gl2_from(0);
        for (size_t i = n; i-- > 1;) {
            acc = gl2_add(gl2_mul(acc, bt->point), comp[i]);
            final_poly[i - 1] = gl2_add(final_poly[i - 1], acc);
        }
    }
    free(comp);
    gl2* coeffs = final_poly;
    gl2* values = (gl2*)malloc(sizeof(gl2) * L);
    memcpy(values, coeffs, sizeof(gl2) * L);
    ext_coset_fft(values, log_L, GL_GEN);

    uint64_t** tree_leaves = (uint64_t**)calloc(R + 1, sizeof(uint64_t*));
    uint64_t** tree_digests = (uint64_t**)calloc(R + 1, sizeof(uint64_t*));
    size_t* tree_nleaves = (size_t*)calloc(R + 1, sizeof(size_t));
    size_t cur_len = L;
    uint64_t shift = GL_GEN;
    for (uint32_t r = 0; r < R; r++) {
        unsigned lg = gl_log2_strict(cur_len);
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
        w_u64s(w, capbuf, (size_t)4 << cap_h);
        observe_cap(ch, capbuf, cap_h);
        gl2 beta = orc_ch_ext_challenge(ch);
        size_t new_len = cur_len / arity;
        for (size_t j = 0; j < new_len; j++) {
            gl2 acc = gl2_from(0);
            for (uint32_t t = arity; t-- > 0;) acc = gl2_add(gl2_mul(acc, beta), coeffs[j * arity + t]);
            coeffs[j] = acc;
        }
        cur_len = new_len;
        shift = gl_exp_pow2(shift, p->arity_bits);
        memcpy(values, coeffs, sizeof(gl2) * cur_len);
        ext_coset_fft(values, gl_log2_strict(cur_len), shift);
    }
    size_t final_len = cur_len >> p->rate_bits;
    orc_ch_observe_many(ch, (uint64_t*)coeffs, 2 * final_len);
    /* proof of work: smallest witness */
    uint64_t pow_witness = 0;
    {
        uint64_t st0[12];
        memcpy(st0, ch->state, sizeof st0);
        memcpy(st0, ch->in_buf, ch->n_in * 8);
        unsigned pos = ch->n_in;
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
                unsigned lz = st[7] ? (unsigned)__builtin_clzll(st[7]) : 64;
                if (lz >= p->pow_bits && cand < best) best = cand;
            }
            if (best != ~0ULL) { pow_witness = best; found = 1; }
        }
        orc_ch_observe(ch, pow_witness);
        (void)orc_ch_challenge(ch);
    }
    for (uint32_t q = 0; q < p->n_queries; q++) {
        size_t x_index = orc_ch_challenge(ch) % L;
        for (uint32_t o = 0; o < n_oracles; o++) {
            const orc_fri_oracle* b = &oracles[o];
            w_u64s(w, b->leaves + x_index * b->n_cols, b->n_cols);
            unsigned plen = log_L - cap_h;
            uint64_t sib[64 * 4];
            orc_merkle_prove(b->digests, L, cap_h, x_index, sib);
            w_u8(w, (uint8_t)plen);
            w_u64s(w, sib, 4 * plen);
        }
        for (uint32_t r = 0; r < R; r++) {
            size_t leaf = x_index >> p->arity_bits;
            w_u64s(w, tree_leaves[r] + leaf * 2 * arity, 2 * arity);
            unsigned lg = gl_log2_strict(tree_nleaves[r]);
            unsigned plen = lg > cap_h ? lg - cap_h : 0;
            uint64_t sib[64 * 4];
            orc_merkle_prove(tree_digests[r], tree_nleaves[r], cap_h, leaf, sib);
            w_u8(w, (uint8_t)plen);
            w_u64s(w, sib, 4 * plen);
            x_index = leaf;
        }
    }
    w_u64s(w, (uint64_t*)coeffs, 2 * final_len);
    w_u64s(w, &pow_witness, 1);
    for (uint32_t r = 0; r < R; r++) { free(tree_leaves[r]); free(tree_digests[r]); }
    free(tree_leaves); free(tree_digests); free(tree_nleaves); free(values); free(final_poly);
}

/* ---------------- FRI verifier: verify_fri_proof ---------------- */
/* opened[b][j] = claimed value of batch b's j-th polynomial at its point.  Reads the FRI part of the
 * proof from r (commit caps, query rounds, final polynomial, pow witness).  1 = accept. */
static int fri_verify(const orc_fri_params* p, const uint64_t* const* caps, const uint32_t* n_cols, uint32_t n_oracles,
                      const orc_fri_batch* batches, uint32_t n_batches, const gl2* const* opened, orc_challenger* ch,
                      rbuf* r) {
    const unsigned log_n = p->degree_bits, log_L = log_n + p->rate_bits, cap_h = p->cap_height;
    const size_t L = (size_t)1 << log_L, capw = (size_t)4 << cap_h;
    const uint32_t arity = 1u << p->arity_bits, R = fri_num_rounds(p);
    int rc = 1;
    gl2 fri_alpha = orc_ch_ext_challenge(ch);
    uint64_t* fri_caps = (uint64_t*)malloc(8 * capw * (R + 1));
    r_u64s(r, fri_caps, R * capw);
    gl2 fri_betas[16];
    for (uint32_t k = 0; k < R; k++) {
        orc_ch_observe_many(ch, fri_caps + k * capw, capw);
        fri_betas[k] = orc_ch_ext_challenge(ch);
    }
    size_t per_query = 0;
    for (uint32_t o = 0; o < n_oracles; o++) per_query += n_cols[o] * 8 + 1 + 32 * (size_t)(log_L - cap_h);
    {
        size_t nl = L;
        for (uint32_t k = 0; k < R; k++) {
            nl >>= p->arity_bits;
            unsigned lg = gl_log2_strict(nl);
            per_query += 16 * arity + 1 + 32 * (lg > cap_h ? lg - cap_h : 0);
        }
    }
    size_t q_start = r->pos;
    r->pos += per_query * p->n_queries;
    size_t final_len = (size_t)1 << (p->degree_bits - R * p->arity_bits);
    gl2* final_poly = (gl2*)malloc(sizeof(gl2) * final_len);
    uint64_t pow_witness = 0;
    r_u64s(r, (uint64_t*)final_poly, 2 * final_len);
    r_u64s(r, &pow_witness, 1);
    if (r->bad) { rc = -1; goto done; }
    orc_ch_observe_many(ch, (uint64_t*)final_poly, 2 * final_len);
    orc_ch_observe(ch, pow_witness);
    {
        uint64_t pow_response = orc_ch_challenge(ch);
        if (p->pow_bits && (pow_response >> (64 - p->pow_bits)) != 0) { rc = -3; goto done; }
    }
    {
        /* reduced openings per batch, and the shift exponents (number of polys of later batches) */
        gl2* red = (gl2*)malloc(sizeof(gl2) * n_batches);
        for (uint32_t b = 0; b < n_batches; b++) {
            gl2 apow = gl2_from(1), acc = gl2_from(0);
            for (uint32_t j = 0; j < batches[b].n_polys; j++) { acc = gl2_add(acc, gl2_mul(apow, opened[b][j])); apow = gl2_mul(apow, fri_alpha); }
            red[b] = acc;
        }
        uint32_t max_cols = 0;
        for (uint32_t o = 0; o < n_oracles; o++) if (n_cols[o] > max_cols) max_cols = n_cols[o];
        uint64_t** rows = (uint64_t**)malloc(sizeof(uint64_t*) * n_oracles);
        for (uint32_t o = 0; o < n_oracles; o++) rows[o] = (uint64_t*)malloc(8 * (n_cols[o] + 1));
        rbuf qr = {r->p, r->len, q_start, 0};
        for (uint32_t q = 0; q < p->n_queries && rc == 1; q++) {
            size_t x_index = orc_ch_challenge(ch) % L;
            for (uint32_t o = 0; o < n_oracles; o++) {
                r_u64s(&qr, rows[o], n_cols[o]);
                uint8_t plen = 0;
                r_bytes(&qr, &plen, 1);
                uint64_t sib[64 * 4];
                if (plen != log_L - cap_h) { rc = -4; break; }
                r_u64s(&qr, sib, 4 * plen);
                if (!orc_merkle_verify_g(rows[o], n_cols[o], p->leaf_group, x_index, sib, plen, caps[o], cap_h)) { rc = -5; break; }
            }
            if (rc != 1) break;
            uint64_t subgroup_x = gl_mul(GL_GEN, gl_pow(gl_root_of_unity(log_L), gl_bitrev(x_index, log_L)));
            gl2 sx = gl2_from(subgroup_x);
            /* fri_combine_initial */
            gl2 sum = gl2_from(0);
            for (uint32_t b = 0; b < n_batches; b++) {
                gl2 apow = gl2_from(1), acc = gl2_from(0);
                for (uint32_t j = 0; j < batches[b].n_polys; j++) {
                    acc = gl2_add(acc, gl2_scale(apow, rows[batches[b].oracle[j]][batches[b].poly[j]]));
                    apow = gl2_mul(apow, fri_alpha);
                }
                sum = gl2_mul(sum, apow); /* alpha.shift: times alpha^(count of this batch) */
                sum = gl2_add(sum, gl2_mul(gl2_sub(acc, red[b]), gl2_inv(gl2_sub(sx, batches[b].point))));
            }
            gl2 old_eval = sum;
            size_t nl = L;
            for (uint32_t k = 0; k < R; k++) {
                nl >>= p->arity_bits;
                gl2 evals[64];
                r_u64s(&qr, (uint64_t*)evals, 2 * arity);
                uint8_t plen = 0;
                r_bytes(&qr, &plen, 1);
                uint64_t sib[64 * 4];
                unsigned lg = gl_log2_strict(nl);
                if (plen != (lg > cap_h ? lg - cap_h : 0)) { rc = -6; break; }
                r_u64s(&qr, sib, 4 * plen);
                size_t coset_index = x_index >> p->arity_bits;
                size_t within = x_index & (arity - 1);
                if (!gl2_eq(evals[within], old_eval)) { rc = -7; break; }
                uint64_t gA = gl_root_of_unity(p->arity_bits);
                size_t rev_within = gl_bitrev(within, p->arity_bits);
                uint64_t coset_start = gl_mul(subgroup_x, gl_pow(gA, arity - rev_within));
                gl2 res = gl2_from(0);
                for (uint32_t i = 0; i < arity; i++) {
                    gl2 yi = evals[gl_bitrev(i, p->arity_bits)];
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
                subgroup_x = gl_exp_pow2(subgroup_x, p->arity_bits);
                x_index = coset_index;
            }
            if (rc != 1) break;
            gl2 fe = gl2_from(0), sxe = gl2_from(subgroup_x);
            for (size_t i = final_len; i-- > 0;) fe = gl2_add(gl2_mul(fe, sxe), final_poly[i]);
            if (!gl2_eq(fe, old_eval)) rc = -9;
        }
        if (qr.bad) rc = -10;
        for (uint32_t o = 0; o < n_oracles; o++) free(rows[o]);
        free(rows);
        free(red);
    }
done:
    free(fri_caps);
    free(final_poly);
    return rc;
}

/* ---------------- AIR program interpreter (base field and extension field) ---------------- */
#define AIR_OP(w) ((uint32_t)((w) & 0xFF))
#define AIR_DST(w) ((uint32_t)(((w) >> 8) & 0xFFFF))
#define AIR_A(w) ((uint32_t)(((w) >> 24) & 0xFFFF))
#define AIR_B(w) ((uint32_t)(((w) >> 40) & 0xFFFF))
#define AIR_SH(w) ((uint32_t)(((w) >> 56) & 0x3F))
#define AIR_REGS 64

/* ConstraintConsumer over the base field: acc_j = acc_j * alpha_j + c */
static void air_eval_base(const orc_stark_desc* d, const uint64_t* local, const uint64_t* next, const uint64_t* pis,
                          const uint64_t* per, uint64_t z_last, uint64_t l_first, uint64_t l_last, const uint64_t* alphas,
                          uint64_t* accs) {
    uint64_t reg[AIR_REGS] = {0};
    for (uint32_t j = 0; j < d->num_challenges; j++) accs[j] = 0;
    for (uint32_t pc = 0; pc < d->n_words; pc++) {
        uint64_t w = d->program[pc];
        uint32_t dst = AIR_DST(w) % AIR_REGS, a = AIR_A(w), b = AIR_B(w);
        uint64_t c = 0;
        int emit = 0;
        switch (AIR_OP(w)) {
            case ORC_AIR_LOCAL: reg[dst] = local[a]; break;
            case ORC_AIR_NEXT: reg[dst] = next[a]; break;
            case ORC_AIR_PUBLIC: reg[dst] = pis[a]; break;
            case ORC_AIR_PERIODIC: reg[dst] = per[a]; break;
            case ORC_AIR_CONST: reg[dst] = d->program[++pc] % GL_P; break;
            case ORC_AIR_ADD: reg[dst] = gl_add(reg[a % AIR_REGS], gl_mul(reg[b % AIR_REGS], 1ULL << AIR_SH(w))); break;
            case ORC_AIR_SUB: reg[dst] = gl_sub(reg[a % AIR_REGS], gl_mul(reg[b % AIR_REGS], 1ULL << AIR_SH(w))); break;
            case ORC_AIR_MUL: reg[dst] = gl_mul(reg[a % AIR_REGS], reg[b % AIR_REGS]); break;
            case ORC_AIR_MAC: reg[dst] = gl_add(reg[AIR_SH(w)], gl_mul(reg[a % AIR_REGS], reg[b % AIR_REGS])); break;
            case ORC_AIR_PACK_LOCAL: case ORC_AIR_PACK_NEXT: {
                const uint64_t* rowv = AIR_OP(w) == ORC_AIR_PACK_LOCAL ? local : next;
                uint64_t acc = 0;
                for (uint32_t i = 0; i < b; i++) acc = gl_add(acc, gl_mul(rowv[a + i], 1ULL << i));
                reg[dst] = acc;
                break;
            }
            case ORC_AIR_EMIT_BOOL:
                for (uint32_t i = 0; i < (b ? b : 1); i++) {
                    const uint64_t cb = gl_mul(local[a + i], gl_sub(local[a + i], 1));
                    for (uint32_t j = 0; j < d->num_challenges; j++) accs[j] = gl_add(gl_mul(accs[j], alphas[j]), cb);
                }
                break;
            case ORC_AIR_XOR3: case ORC_AIR_CH: case ORC_AIR_MAJ: {
                const uint64_t x = reg[a % AIR_REGS], y = reg[b % AIR_REGS], z = reg[AIR_SH(w)];
                if (AIR_OP(w) == ORC_AIR_CH) {
                    reg[dst] = gl_add(z, gl_mul(x, gl_sub(y, z)));
                } else {
                    const uint64_t xy = gl_mul(x, y);
                    const uint64_t sx = gl_sub(gl_add(x, y), gl_add(xy, xy)); /* x ^ y */
                    if (AIR_OP(w) == ORC_AIR_XOR3) {
                        const uint64_t sz = gl_mul(sx, z);
                        reg[dst] = gl_sub(gl_add(sx, z), gl_add(sz, sz));
                    } else {
                        reg[dst] = gl_add(xy, gl_mul(z, sx));
                    }
                }
                break;
            }
            case ORC_AIR_EMIT_TRANSITION: c = gl_mul(reg[a % AIR_REGS], z_last); emit = 1; break;
            case ORC_AIR_EMIT_FIRST: c = gl_mul(reg[a % AIR_REGS], l_first); emit = 1; break;
            case ORC_AIR_EMIT_LAST: c = gl_mul(reg[a % AIR_REGS], l_last); emit = 1; break;
            case ORC_AIR_EMIT: c = reg[a % AIR_REGS]; emit = 1; break;
            case ORC_AIR_SEGMENT: memset(reg, 0, sizeof reg); break;
            case ORC_AIR_EMIT_LOGUP: {
                /* h (al + v1)(al + v2) = (al + v1) + (al + v2) in F_p[X]/(X^2 - 7), al = a0 + a1 X, h = h0 + h1 X */
                const uint64_t a0 = pis[d->num_public_inputs + AIR_SH(w)], a1 = pis[d->num_public_inputs + AIR_SH(w) + 1];
                const uint64_t h0 = local[b], h1 = local[b + 1], v1 = local[a];
                uint64_t c0, c1;
                if (AIR_DST(w) == 0xFFFF) {
                    const uint64_t d0 = gl_add(a0, v1);
                    c0 = gl_sub(gl_add(gl_mul(h0, d0), gl_mul(7, gl_mul(h1, a1))), 1);
                    c1 = gl_add(gl_mul(h0, a1), gl_mul(h1, d0));
                } else {
                    const uint64_t v2 = local[AIR_DST(w)];
                    const uint64_t s = gl_add(gl_add(a0, a0), gl_add(v1, v2));
                    const uint64_t u0 = gl_add(gl_mul(gl_add(a0, v1), gl_add(a0, v2)), gl_mul(7, gl_mul(a1, a1)));
                    const uint64_t u1 = gl_mul(a1, s);
                    c0 = gl_sub(gl_add(gl_mul(h0, u0), gl_mul(7, gl_mul(h1, u1))), s);
                    c1 = gl_sub(gl_add(gl_mul(h0, u1), gl_mul(h1, u0)), gl_add(a1, a1));
                }
                for (uint32_t j = 0; j < d->num_challenges; j++) {
                    accs[j] = gl_add(gl_mul(accs[j], alphas[j]), c0);
                    accs[j] = gl_add(gl_mul(accs[j], alphas[j]), c1);
                }
                break;
            }
            default: break;
        }
        if (emit)
            for (uint32_t j = 0; j < d->num_challenges; j++) accs[j] = gl_add(gl_mul(accs[j], alphas[j]), c);
    }
}
static void air_eval_ext(const orc_stark_desc* d, const gl2* local, const gl2* next, const uint64_t* pis, const gl2* per,
                         gl2 z_last, gl2 l_first, gl2 l_last, const uint64_t* alphas, gl2* accs) {
    gl2 reg[AIR_REGS];
    for (int i = 0; i < AIR_REGS; i++) reg[i] = gl2_from(0);
    for (uint32_t j = 0; j < d->num_challenges; j++) accs[j] = gl2_from(0);
    for (uint32_t pc = 0; pc < d->n_words; pc++) {
        uint64_t w = d->program[pc];
        uint32_t dst = AIR_DST(w) % AIR_REGS, a = AIR_A(w), b = AIR_B(w);
        gl2 c = gl2_from(0);
        int emit = 0;
        switch (AIR_OP(w)) {
            case ORC_AIR_LOCAL: reg[dst] = local[a]; break;
            case ORC_AIR_NEXT: reg[dst] = next[a]; break;
            case ORC_AIR_PUBLIC: reg[dst] = gl2_from(pis[a]); break;
            case ORC_AIR_PERIODIC: reg[dst] = per[a]; break;
            case ORC_AIR_CONST: reg[dst] = gl2_from(d->program[++pc] % GL_P); break;
            case ORC_AIR_ADD: reg[dst] = gl2_add(reg[a % AIR_REGS], gl2_scale(reg[b % AIR_REGS], 1ULL << AIR_SH(w))); break;
            case ORC_AIR_SUB: reg[dst] = gl2_sub(reg[a % AIR_REGS], gl2_scale(reg[b % AIR_REGS], 1ULL << AIR_SH(w))); break;
            case ORC_AIR_MUL: reg[dst] = gl2_mul(reg[a % AIR_REGS], reg[b % AIR_REGS]); break;
            case ORC_AIR_MAC: reg[dst] = gl2_add(reg[AIR_SH(w)], gl2_mul(reg[a % AIR_REGS], reg[b % AIR_REGS])); break;
            case ORC_AIR_PACK_LOCAL: case ORC_AIR_PACK_NEXT: {
                const gl2* rowv = AIR_OP(w) == ORC_AIR_PACK_LOCAL ? local : next;
                gl2 acc = gl2_from(0);
                for (uint32_t i = 0; i < b; i++) acc = gl2_add(acc, gl2_scale(rowv[a + i], 1ULL << i));
                reg[dst] = acc;
                break;
            }
            case ORC_AIR_EMIT_BOOL:
                for (uint32_t i = 0; i < (b ? b : 1); i++) {
                    const gl2 cb = gl2_mul(local[a + i], gl2_sub(local[a + i], gl2_from(1)));
                    for (uint32_t j = 0; j < d->num_challenges; j++) accs[j] = gl2_add(gl2_scale(accs[j], alphas[j]), cb);
                }
                break;
            case ORC_AIR_XOR3: case ORC_AIR_CH: case ORC_AIR_MAJ: {
                const gl2 x = reg[a % AIR_REGS], y = reg[b % AIR_REGS], z = reg[AIR_SH(w)];
                if (AIR_OP(w) == ORC_AIR_CH) {
                    reg[dst] = gl2_add(z, gl2_mul(x, gl2_sub(y, z)));
                } else {
                    const gl2 xy = gl2_mul(x, y);
                    const gl2 sx = gl2_sub(gl2_add(x, y), gl2_add(xy, xy));
                    if (AIR_OP(w) == ORC_AIR_XOR3) {
                        const gl2 sz = gl2_mul(sx, z);
                        reg[dst] = gl2_sub(gl2_add(sx, z), gl2_add(sz, sz));
                    } else {
                        reg[dst] = gl2_add(xy, gl2_mul(z, sx));
                    }
                }
                break;
            }
            case ORC_AIR_EMIT_TRANSITION: c = gl2_mul(reg[a % AIR_REGS], z_last); emit = 1; break;
            case ORC_AIR_EMIT_FIRST: c = gl2_mul(reg[a % AIR_REGS], l_first); emit = 1; break;
            case ORC_AIR_EMIT_LAST: c = gl2_mul(reg[a % AIR_REGS], l_last); emit = 1; break;
            case ORC_AIR_EMIT: c = reg[a % AIR_REGS]; emit = 1; break;
            case ORC_AIR_SEGMENT:
                for (int i = 0; i < AIR_REGS; i++) reg[i] = gl2_from(0);
                break;
            case ORC_AIR_EMIT_LOGUP: {
                const uint64_t a0 = pis[d->num_public_inputs + AIR_SH(w)], a1 = pis[d->num_public_inputs + AIR_SH(w) + 1];
                const gl2 h0 = local[b], h1 = local[b + 1], v1 = local[a];
                gl2 c0, c1;
                if (AIR_DST(w) == 0xFFFF) {
                    const gl2 d0 = gl2_add(gl2_from(a0), v1);
                    c0 = gl2_sub(gl2_add(gl2_mul(h0, d0), gl2_scale(h1, gl_mul(7, a1))), gl2_from(1));
                    c1 = gl2_add(gl2_scale(h0, a1), gl2_mul(h1, d0));
                } else {
                    const gl2 v2 = local[AIR_DST(w)];
                    const gl2 s = gl2_add(gl2_from(gl_add(a0, a0)), gl2_add(v1, v2));
                    const gl2 u0 = gl2_add(gl2_mul(gl2_add(gl2_from(a0), v1), gl2_add(gl2_from(a0), v2)), gl2_from(gl_mul(7, gl_mul(a1, a1))));
                    const gl2 u1 = gl2_scale(s, a1);
                    c0 = gl2_sub(gl2_add(gl2_mul(h0, u0), gl2_scale(gl2_mul(h1, u1), 7)), s);
                    c1 = gl2_sub(gl2_add(gl2_mul(h0, u1), gl2_mul(h1, u0)), gl2_from(gl_add(a1, a1)));
                }
                for (uint32_t j = 0; j < d->num_challenges; j++) {
                    accs[j] = gl2_add(gl2_scale(accs[j], alphas[j]), c0);
                    accs[j] = gl2_add(gl2_scale(accs[j], alphas[j]), c1);
                }
                break;
            }
            default: break;
        }
        if (emit)
            for (uint32_t j = 0; j < d->num_challenges; j++) accs[j] = gl2_add(gl2_scale(accs[j], alphas[j]), c);
    }
}

#define ORC_MAX_PERIODIC 128
/* interpolation of each periodic column over the period-th roots of unity (coefficients, natural order) */
static uint64_t* periodic_coeffs(const orc_stark_desc* d) {
    if (!d->n_periodic) return NULL;
    const size_t period = (size_t)1 << d->period_bits;
    uint64_t* c = (uint64_t*)malloc(8 * period * d->n_periodic);
    for (size_t i = 0; i < period * d->n_periodic; i++) c[i] = d->periodic[i] % GL_P;
    for (uint32_t a = 0; a < d->n_periodic; a++) orc_ifft(c + a * period, d->period_bits);
    return c;
}

/* The AIR digest: what circuit_digest is to a plonky2 circuit - the whole description (config, program, periodic
 * columns, round structure) hashed into four field elements, so that a proof is bound to the statement's constraint set.
 * hash_no_pad over 32-bit halves (every half is a canonical field element): the 24 shape words below, then every
 * program word as (lo, hi) with CONST immediates reduced mod p first, then every periodic value (reduced) as (lo, hi). */
void orc_stark_air_digest(const orc_stark_desc* d, uint64_t out[4]) {
    const size_t n_per = d->n_periodic ? ((size_t)d->n_periodic << d->period_bits) : 0;
    const size_t len = 24 + ((d->leaf_group_cols || d->openings_group) ? 2 : 0) + (d->batch_cols ? 2 : 0) + 2 * (size_t)d->n_words + 2 * n_per;
    uint64_t* v = (uint64_t*)malloc(8 * len);
    size_t k = 0;
    const uint32_t shape[14] = {d->degree_bits, d->n_cols, d->num_challenges, d->rate_bits, d->cap_height,
                                d->quotient_degree_factor, d->fri_pow_bits, d->fri_num_queries, d->fri_arity_bits,
                                d->fri_final_poly_bits, d->num_public_inputs, d->n_words, d->n_periodic,
                                d->n_periodic ? d->period_bits : 0};
    for (int i = 0; i < 14; i++) v[k++] = shape[i];
    v[k++] = d->n_rounds;
    for (int r = 0; r < 3; r++) v[k++] = (uint32_t)r < d->n_rounds ? d->round_cols[r] : 0;
    for (int r = 0; r < 3; r++) v[k++] = (uint32_t)r < d->n_rounds ? d->round_challenges[r] : 0;
    for (int r = 0; r < 3; r++) v[k++] = (uint32_t)r < d->n_rounds ? d->round_values[r] : 0;
    if (d->leaf_group_cols || d->openings_group) {   /* only when used: digests of plain-starky statements stay what they were */
        v[k++] = d->leaf_group_cols;
        v[k++] = d->openings_group;
    }
    if (d->batch_cols) {   /* likewise: a statement without batches keeps its digest */
        v[k++] = 0xB47C4u;
        v[k++] = d->batch_cols;
    }
    for (uint32_t pc = 0; pc < d->n_words; pc++) {
        uint64_t w = d->program[pc];
        v[k++] = w & 0xFFFFFFFFu; v[k++] = w >> 32;
        if (AIR_OP(w) == ORC_AIR_CONST && pc + 1 < d->n_words) {
            w = d->program[++pc] % GL_P;
            v[k++] = w & 0xFFFFFFFFu; v[k++] = w >> 32;
        }
    }
    for (size_t i = 0; i < n_per; i++) { uint64_t w = d->periodic[i] % GL_P; v[k++] = w & 0xFFFFFFFFu; v[k++] = w >> 32; }
    orc_hash_no_pad(v, k, out);
    free(v);
}

/* The transcript opens with the statement: the AIR digest, then the public inputs, before any commitment.  (plonky2's
 * starky of the pinned era observed only the trace cap first - a Fiat-Shamir gap: the public inputs enter the AIR
 * linearly, so a prover could fix them AFTER alpha and zeta were known.  plonky2's own prover observes circuit_digest and
 * the public-input hash first; the STARK side does the same here.) */
static void transcript_start(orc_challenger* ch, const orc_stark_desc* d, const uint64_t* public_inputs) {
    uint64_t dig[4];
    orc_stark_air_digest(d, dig);
    orc_ch_observe_many(ch, dig, 4);
    if (d->num_public_inputs) orc_ch_observe_many(ch, public_inputs, d->num_public_inputs);
}

/* observe_openings(&openings.to_fri_openings()): the zeta batch (local ++ quotient), then the zeta_next batch - every value, or
 * (openings_group = G) the four-element digest of that vector: zero-padded to a multiple of G, runs of G hashed, the run
 * digests hashed */
static void observe_openings(orc_challenger* ch, const orc_stark_desc* d, const gl2* o_local, const gl2* o_q, const gl2* o_next,
                             uint32_t ncols, uint32_t nq) {
    if (!d->openings_group) {
        orc_ch_observe_many(ch, (const uint64_t*)o_local, 2 * ncols);
        orc_ch_observe_many(ch, (const uint64_t*)o_q, 2 * nq);
        orc_ch_observe_many(ch, (const uint64_t*)o_next, 2 * ncols);
        return;
    }
    const size_t G = d->openings_group, len = 2 * (size_t)(2 * ncols + nq), K = (len + G - 1) / G;
    uint64_t* v = (uint64_t*)calloc(K * G + 4 * K, 8);
    uint64_t* dg = v + K * G;
    memcpy(v, o_local, 16 * (size_t)ncols);
    memcpy(v + 2 * (size_t)ncols, o_q, 16 * (size_t)nq);
    memcpy(v + 2 * (size_t)(ncols + nq), o_next, 16 * (size_t)ncols);
    for (size_t k = 0; k < K; k++) orc_hash_no_pad(v + k * G, G, dg + 4 * k);
    uint64_t out[4];
    orc_hash_no_pad(dg, 4 * K, out);
    orc_ch_observe_many(ch, out, 4);
    free(v);
}

/* rounds of commitment: classic starky = one round, no verifier challenges before the alphas */
static uint32_t n_rounds_of(const orc_stark_desc* d) { return d->n_rounds ? d->n_rounds : 1; }
static uint32_t round_cols_of(const orc_stark_desc* d, uint32_t r) { return d->n_rounds ? d->round_cols[r] : d->n_cols; }
/* everything after the public inputs in the values array: round values and challenges of all rounds */
static uint32_t total_round_challenges(const orc_stark_desc* d) {
    uint32_t t = 0;
    for (uint32_t r = 0; r < d->n_rounds; r++) t += d->round_challenges[r] + d->round_values[r];
    return t;
}
static uint32_t total_round_values(const orc_stark_desc* d) {
    uint32_t t = 0;
    for (uint32_t r = 0; r < d->n_rounds; r++) t += d->round_values[r];
    return t;
}

static int desc_ok(const orc_stark_desc* d) {
    uint32_t q = d->quotient_degree_factor;
    if (!q || (q & (q - 1)) || q > (1u << d->rate_bits)) return 0;
    if (d->num_challenges < 1 || d->num_challenges > 2 || d->n_cols == 0) return 0;
    if (d->n_periodic > ORC_MAX_PERIODIC) return 0;
    if (d->n_periodic && (d->period_bits > d->degree_bits || d->period_bits > 16 || !d->periodic)) return 0;
    if (d->n_rounds > 3) return 0;
    if (d->n_rounds) {
        uint32_t tot = 0;
        for (uint32_t r = 0; r < d->n_rounds; r++) {
            if (!d->round_cols[r] || d->round_challenges[r] > 16 || d->round_values[r] > 64) return 0;
            tot += d->round_cols[r];
        }
        if (tot != d->n_cols) return 0;
    }
    if (d->batch_cols) {
        if (d->batch_cols < 8 || d->batch_cols > 65535 || d->leaf_group_cols) return 0;
        uint32_t no = 0;
        for (uint32_t r = 0; r < (d->n_rounds ? d->n_rounds : 1u); r++) {
            const uint32_t rc = d->n_rounds ? d->round_cols[r] : d->n_cols;
            no += (rc + d->batch_cols - 1) / d->batch_cols;
        }
        if (no > ORC_STARK_MAX_ORACLES - 1) return 0;
    }
    const uint32_t n_values = d->num_public_inputs + total_round_challenges(d);
    for (uint32_t pc = 0; pc < d->n_words; pc++) {
        uint64_t w = d->program[pc];
        switch (AIR_OP(w)) {
            case ORC_AIR_LOCAL: case ORC_AIR_NEXT: if (AIR_A(w) >= d->n_cols) return 0; break;
            case ORC_AIR_PUBLIC: if (AIR_A(w) >= n_values) return 0; break;
            case ORC_AIR_PERIODIC: if (AIR_A(w) >= d->n_periodic) return 0; break;
            case ORC_AIR_PACK_LOCAL: case ORC_AIR_PACK_NEXT:
                if (AIR_B(w) < 1 || AIR_B(w) > 32 || AIR_A(w) + AIR_B(w) > d->n_cols) return 0;
                break;
            case ORC_AIR_EMIT_BOOL: if (AIR_A(w) + (AIR_B(w) ? AIR_B(w) : 1) > d->n_cols) return 0; break;
            case ORC_AIR_CONST: if (++pc >= d->n_words) return 0; break;
            case ORC_AIR_EMIT_LOGUP:
                if (AIR_A(w) >= d->n_cols || AIR_B(w) + 1 >= d->n_cols || (AIR_DST(w) != 0xFFFF && AIR_DST(w) >= d->n_cols) ||
                    AIR_SH(w) + 1 >= total_round_challenges(d))
                    return 0;
                break;
            default: if (AIR_OP(w) > ORC_AIR_MAC) return 0;
        }
    }
    return 1;
}

/* batches of a round of rc columns (orc_stark_desc.batch_cols), and the trace oracles of the whole proof */
static uint32_t batches_of(const orc_stark_desc* d, uint32_t rc) {
    return d->batch_cols && rc > d->batch_cols ? (rc + d->batch_cols - 1) / d->batch_cols : 1;
}
static uint32_t batch_width(const orc_stark_desc* d, uint32_t rc, uint32_t k) {
    if (batches_of(d, rc) == 1) return rc;
    const uint32_t c0 = k * d->batch_cols;
    return rc - c0 < d->batch_cols ? rc - c0 : d->batch_cols;
}
static uint32_t n_trace_oracles(const orc_stark_desc* d) {
    uint32_t no = 0;
    for (uint32_t r = 0; r < n_rounds_of(d); r++) no += batches_of(d, round_cols_of(d, r));
    return no;
}

size_t orc_stark_proof_max_bytes(const orc_stark_desc* d) {
    const size_t capb = (size_t)32 << d->cap_height;
    const unsigned log_L = d->degree_bits + d->rate_bits;
    const uint32_t nq = d->num_challenges * d->quotient_degree_factor, NO = n_trace_oracles(d);
    orc_fri_params fp = {d->degree_bits, d->rate_bits, d->cap_height, d->fri_pow_bits, d->fri_num_queries, d->fri_arity_bits, d->fri_final_poly_bits, d->leaf_group_cols};
    uint32_t R = fri_num_rounds(&fp);
    size_t bytes = (NO + 1) * capb + 16 * (size_t)(2 * d->n_cols + nq) + R * capb;
    size_t per_query = (d->n_cols + nq) * 8 + (NO + 1) * (1 + 32 * (size_t)log_L) + R * (((size_t)16 << d->fri_arity_bits) + 1 + 32 * (size_t)log_L);
    bytes += per_query * d->fri_num_queries + ((size_t)16 << d->degree_bits) + 8 + 8 + 8 * (size_t)d->num_public_inputs +
             8 * (size_t)total_round_values(d);
    return bytes + 64;
}

/* Multi-round prover: round r's columns are obtained from `fn` AFTER the challenges of rounds < r are known
 * (starkyx's TraceWriter rounds: lookup / bus accumulators are functions of earlier challenges).  One round with no
 * challenges is exactly starky::prover::prove. */
size_t orc_stark_prove_rounds(const orc_stark_desc* d, orc_round_fn fn, void* user, const uint64_t* public_inputs,
                              uint8_t* proof_out, size_t cap_bytes) {
    if (!desc_ok(d) || !fn) return 0;
    const unsigned log_n = d->degree_bits, log_L = log_n + d->rate_bits, cap_h = d->cap_height;
    const size_t n = (size_t)1 << log_n, L = (size_t)1 << log_L, capw = (size_t)4 << cap_h;
    const uint32_t nc = d->num_challenges, qdf = d->quotient_degree_factor, ncols = d->n_cols, nq = nc * qdf;
    const uint32_t NRD = n_rounds_of(d), n_rch = total_round_challenges(d);
    const unsigned qdb = gl_log2_strict(qdf);
    wbuf w = {proof_out, 0, cap_bytes, 0};
    orc_challenger ch;
    orc_ch_init(&ch);
    /* values readable by PUBLIC: public inputs, then the verifier challenges in the order they are drawn */
    uint64_t* values = (uint64_t*)calloc(d->num_public_inputs + n_rch + 1, 8);
    if (d->num_public_inputs) memcpy(values, public_inputs, 8 * (size_t)d->num_public_inputs);
    transcript_start(&ch, d, values);
    uint32_t n_drawn = 0;
    /* per TRACE ORACLE (a round, or one of its batches): coefficients, row-major leaves, tree, cap, first column and width */
    uint64_t *r_coeffs[ORC_STARK_MAX_ORACLES] = {0}, *r_leaves[ORC_STARK_MAX_ORACLES] = {0}, *r_dig[ORC_STARK_MAX_ORACLES] = {0};
    uint64_t (*r_cap)[4 * 64] = (uint64_t (*)[4 * 64])malloc(sizeof(uint64_t[4 * 64]) * ORC_STARK_MAX_ORACLES);
    uint32_t col0[ORC_STARK_MAX_ORACLES + 1] = {0};
    uint32_t NO = 0;   /* trace oracles so far */
    int ok = 1;
    for (uint32_t r = 0; r < NRD && ok; r++) {
        const uint32_t rc = round_cols_of(d, r);
        uint64_t* rv = values + d->num_public_inputs + n_drawn;  /* this round's values go here */
        const uint32_t n_rv = d->n_rounds ? d->round_values[r] : 0;
        const uint64_t* tr = fn(user, r, values + d->num_public_inputs, n_drawn, n_rv ? rv : NULL);
        if (!tr) { ok = 0; break; }
        for (uint32_t k = 0; k < n_rv; k++) rv[k] %= GL_P;
        for (uint32_t k = 0; k < batches_of(d, rc); k++) {   /* one PolynomialBatch::from_values per batch, caps in batch order */
            const uint32_t bw = batch_width(d, rc, k), o = NO++;
            const uint64_t* btr = tr + (size_t)(col0[o] - (col0[o - k])) * n;   /* the batch's columns of this round (column-major) */
            r_coeffs[o] = (uint64_t*)malloc(8 * n * bw);
            r_leaves[o] = (uint64_t*)malloc(8 * L * bw);
            r_dig[o] = (uint64_t*)malloc(8 * orc_merkle_digest_words(L, cap_h));
            orc_commit_from_values_g(btr, bw, log_n, d->rate_bits, cap_h, d->leaf_group_cols, r_coeffs[o], r_leaves[o], r_dig[o], r_cap[o]);
            w_u64s(&w, r_cap[o], capw);
            observe_cap(&ch, r_cap[o], cap_h);
            col0[o + 1] = col0[o] + bw;
        }
        if (n_rv) orc_ch_observe_many(&ch, rv, n_rv);
        n_drawn += n_rv;
        if (d->n_rounds)
            for (uint32_t k = 0; k < d->round_challenges[r]; k++) values[d->num_public_inputs + n_drawn++] = orc_ch_challenge(&ch);
    }
    if (!ok) {
        for (uint32_t o = 0; o < NO; o++) { free(r_coeffs[o]); free(r_leaves[o]); free(r_dig[o]); }
        free(r_cap);
        free(values);
        return 0;
    }
    uint64_t alphas[4];
    for (uint32_t j = 0; j < nc; j++) alphas[j] = orc_ch_challenge(&ch);
    /* compute_quotient_polys on the coset of size n << qdb (LDE index step = 2^(rate_bits - qdb)) */
    const size_t size = n << qdb, step = (size_t)1 << (d->rate_bits - qdb), next_step = (size_t)1 << qdb;
    const unsigned log_size = log_n + qdb;
    uint64_t* qvals = (uint64_t*)malloc(8 * size * nc);
    /* periodic column a at point i of the quotient domain is P_a(y_i), y_i = x_i^(n/period) = G rho^i with rho of
     * order period << qdb: one coset transform of the zero-padded coefficients gives all of them */
    uint64_t* per_table = NULL;
    const size_t per_len = d->n_periodic ? ((size_t)1 << (d->period_bits + qdb)) : 0;
    if (d->n_periodic) {
        uint64_t* per_coeffs = periodic_coeffs(d);
        const size_t period = (size_t)1 << d->period_bits;
        per_table = (uint64_t*)calloc(per_len * d->n_periodic, 8);
        const uint64_t G = gl_exp_pow2(GL_GEN, log_n - d->period_bits);
        for (uint32_t a = 0; a < d->n_periodic; a++) {
            memcpy(per_table + a * per_len, per_coeffs + a * period, 8 * period);
            orc_coset_fft(per_table + a * per_len, d->period_bits + qdb, G);
        }
        free(per_coeffs);
    }
    {
        const uint64_t g = gl_root_of_unity(log_n), last = gl_inv(g);
        const uint64_t w_s = gl_root_of_unity(log_size);
        const uint64_t n_f = (uint64_t)n % GL_P;
#pragma omp parallel
        {
            uint64_t* loc = (uint64_t*)malloc(16 * (size_t)ncols);
            uint64_t* nxt = loc + ncols;
#pragma omp for schedule(static)
            for (size_t i = 0; i < size; i++) {
                uint64_t x = gl_mul(GL_GEN, gl_pow(w_s, i));
                uint64_t zh = gl_sub(gl_exp_pow2(x, log_n), 1);
                uint64_t l_first = gl_mul(zh, gl_inv(gl_mul(n_f, gl_sub(x, 1))));
                uint64_t l_last = gl_mul(zh, gl_inv(gl_mul(n_f, gl_sub(gl_mul(g, x), 1))));
                size_t li = gl_bitrev(i * step, log_L), ln = gl_bitrev(((i + next_step) % size) * step, log_L);
                for (uint32_t r = 0; r < NO; r++) {
                    const uint32_t rc = col0[r + 1] - col0[r];
                    memcpy(loc + col0[r], r_leaves[r] + li * rc, 8 * (size_t)rc);
                    memcpy(nxt + col0[r], r_leaves[r] + ln * rc, 8 * (size_t)rc);
                }
                uint64_t accs[4], per[ORC_MAX_PERIODIC];
                for (uint32_t a = 0; a < d->n_periodic; a++) per[a] = per_table[a * per_len + (i & (per_len - 1))];
                air_eval_base(d, loc, nxt, values, per, gl_sub(x, last), l_first, l_last, alphas, accs);
                uint64_t zh_inv = gl_inv(zh);
                for (uint32_t j = 0; j < nc; j++) qvals[(size_t)j * size + i] = gl_mul(accs[j], zh_inv);
            }
            free(loc);
        }
    }
    uint64_t* q_coeffs = (uint64_t*)malloc(8 * n * nq);
    for (uint32_t j = 0; j < nc; j++) {
        orc_coset_ifft(qvals + (size_t)j * size, log_size, GL_GEN);
        memcpy(q_coeffs + (size_t)j * qdf * n, qvals + (size_t)j * size, 8 * n * qdf); /* trim_to_len(n * qdf), chunks(n) */
    }
    free(qvals);
    free(per_table);
    uint64_t* q_leaves = (uint64_t*)malloc(8 * L * nq);
    uint64_t* q_dig = (uint64_t*)malloc(8 * orc_merkle_digest_words(L, cap_h));
    uint64_t q_cap[4 * 64];
    orc_commit_from_coeffs_g(q_coeffs, nq, log_n, d->rate_bits, cap_h, d->leaf_group_cols, q_leaves, q_dig, q_cap);
    w_u64s(&w, q_cap, capw);
    observe_cap(&ch, q_cap, cap_h);
    gl2 zeta = orc_ch_ext_challenge(&ch);
    gl2 g_zeta = gl2_scale(zeta, gl_root_of_unity(log_n));
    /* StarkOpeningSet: local and next values of every committed column (round order), quotient at zeta */
    gl2* o_local = (gl2*)malloc(sizeof(gl2) * (2 * ncols + nq));
    gl2* o_next = o_local + ncols;
    gl2* o_q = o_next + ncols;
    for (uint32_t r = 0; r < NO; r++) {
        const uint32_t rc = col0[r + 1] - col0[r];
#pragma omp parallel for schedule(dynamic)
        for (uint32_t c = 0; c < rc; c++) {
            o_local[col0[r] + c] = eval_base_poly_ext(r_coeffs[r] + (size_t)c * n, n, zeta);
            o_next[col0[r] + c] = eval_base_poly_ext(r_coeffs[r] + (size_t)c * n, n, g_zeta);
        }
    }
    for (uint32_t c = 0; c < nq; c++) o_q[c] = eval_base_poly_ext(q_coeffs + (size_t)c * n, n, zeta);
    w_u64s(&w, (uint64_t*)o_local, 2 * ncols);
    w_u64s(&w, (uint64_t*)o_next, 2 * ncols);
    w_u64s(&w, (uint64_t*)o_q, 2 * nq);
    /* observe_openings: zeta batch (local ++ quotient), then zeta_next batch (next) */
    observe_openings(&ch, d, o_local, o_q, o_next, ncols, nq);
    /* fri_instance: batch 0 at zeta = every round's columns ++ quotient, batch 1 at g*zeta = every round's columns */
    orc_fri_oracle oracles[ORC_STARK_MAX_ORACLES + 1];
    for (uint32_t r = 0; r < NO; r++) {
        orc_fri_oracle o = {r_coeffs[r], r_leaves[r], r_dig[r], r_cap[r], col0[r + 1] - col0[r]};
        oracles[r] = o;
    }
    {
        orc_fri_oracle o = {q_coeffs, q_leaves, q_dig, q_cap, nq};
        oracles[NO] = o;
    }
    uint32_t* idx_o = (uint32_t*)malloc(4 * 2 * (ncols + nq));
    uint32_t* idx_p = idx_o + ncols + nq;
    for (uint32_t r = 0; r < NO; r++)
        for (uint32_t c = col0[r]; c < col0[r + 1]; c++) { idx_o[c] = r; idx_p[c] = c - col0[r]; }
    for (uint32_t c = 0; c < nq; c++) { idx_o[ncols + c] = NO; idx_p[ncols + c] = c; }
    orc_fri_batch batches[2] = {{zeta, ncols + nq, idx_o, idx_p}, {g_zeta, ncols, idx_o, idx_p}};
    orc_fri_params fp = {d->degree_bits, d->rate_bits, cap_h, d->fri_pow_bits, d->fri_num_queries, d->fri_arity_bits, d->fri_final_poly_bits, d->leaf_group_cols};
    fri_prove(&fp, oracles, NO + 1, batches, 2, &ch, &w);
    w_usize(&w, d->num_public_inputs);
    w_u64s(&w, public_inputs, d->num_public_inputs);
    {   /* the round values, in round order, after the public inputs */
        uint32_t off = d->num_public_inputs;
        for (uint32_t r = 0; r < d->n_rounds; r++) {
            w_u64s(&w, values + off, d->round_values[r]);
            off += d->round_values[r] + d->round_challenges[r];
        }
    }
    free(idx_o); free(o_local); free(q_coeffs); free(q_leaves); free(q_dig); free(values);
    for (uint32_t r = 0; r < NO; r++) { free(r_coeffs[r]); free(r_leaves[r]); free(r_dig[r]); }
    free(r_cap);
    return w.overflow ? 0 : w.len;
}

static const uint64_t* single_round_fn(void* user, uint32_t round, const uint64_t* challenges, uint32_t n_challenges,
                                       uint64_t* values_out) {
    (void)challenges; (void)n_challenges; (void)values_out;
    return round == 0 ? (const uint64_t*)user : NULL;
}

size_t orc_stark_prove(const orc_stark_desc* d, const uint64_t* trace, const uint64_t* public_inputs,
                       uint8_t* proof_out, size_t cap_bytes) {
    if (d->n_rounds > 1) return 0;
    return orc_stark_prove_rounds(d, single_round_fn, (void*)trace, public_inputs, proof_out, cap_bytes);
}

/* The values array of a proof - public inputs, then per round its round values and the challenges drawn after it - as the
 * verifier derives it from the proof's caps and tail (no checks beyond sizes): what a caller needs to compare a round
 * value with the fingerprint it is supposed to be.  Returns the number of values, 0 on a malformed proof. */
uint32_t orc_stark_values(const orc_stark_desc* d, const uint8_t* proof, size_t len, uint64_t* out) {
    if (!desc_ok(d)) return 0;
    const size_t capw = (size_t)4 << d->cap_height;
    const uint32_t NRD = n_rounds_of(d), n_rv_total = total_round_values(d), NO = n_trace_oracles(d);
    if (len < 8 * (NO + 1) * capw + 8 + 8 * (size_t)(d->num_public_inputs + n_rv_total)) return 0;
    const size_t tail = len - 8 - 8 * (size_t)(d->num_public_inputs + n_rv_total);
    const uint64_t* caps = (const uint64_t*)proof;
    uint64_t rv[3 * 64];
    memcpy(out, proof + tail + 8, 8 * (size_t)d->num_public_inputs);
    memcpy(rv, proof + tail + 8 + 8 * (size_t)d->num_public_inputs, 8 * (size_t)n_rv_total);
    orc_challenger ch;
    orc_ch_init(&ch);
    transcript_start(&ch, d, out);
    uint32_t n = d->num_public_inputs, off = 0, o = 0;
    for (uint32_t rd = 0; rd < NRD; rd++) {
        for (uint32_t k = 0; k < batches_of(d, round_cols_of(d, rd)); k++, o++) {   /* the round's caps, batch by batch */
            uint64_t cap[4 * 64];
            memcpy(cap, caps + o * capw, 8 * capw);
            orc_ch_observe_many(&ch, cap, capw);
        }
        if (!d->n_rounds) continue;
        for (uint32_t k = 0; k < d->round_values[rd]; k++) out[n++] = rv[off + k];
        if (d->round_values[rd]) orc_ch_observe_many(&ch, rv + off, d->round_values[rd]);
        off += d->round_values[rd];
        for (uint32_t k = 0; k < d->round_challenges[rd]; k++) out[n++] = orc_ch_challenge(&ch);
    }
    return n;
}

int orc_stark_verify(const orc_stark_desc* d, const uint8_t* proof, size_t len) {
    if (!desc_ok(d)) return -20;
    const unsigned log_n = d->degree_bits, cap_h = d->cap_height;
    const size_t n = (size_t)1 << log_n, capw = (size_t)4 << cap_h;
    const uint32_t nc = d->num_challenges, qdf = d->quotient_degree_factor, ncols = d->n_cols, nq = nc * qdf;
    const uint32_t NRD = n_rounds_of(d), n_rch = total_round_challenges(d), NO = n_trace_oracles(d);
    rbuf r = {proof, len, 0, 0};
    int rc = 1;
    uint64_t* caps = (uint64_t*)malloc(8 * (NO + 1) * capw);
    r_u64s(&r, caps, (NO + 1) * capw);
    gl2* o_local = (gl2*)malloc(sizeof(gl2) * (2 * ncols + nq));
    gl2* o_next = o_local + ncols;
    gl2* o_q = o_next + ncols;
    r_u64s(&r, (uint64_t*)o_local, 2 * (2 * ncols + nq));
    /* public inputs are at the very end; the verifier challenges follow them in the values array */
    uint64_t* pis = (uint64_t*)calloc(d->num_public_inputs + n_rch + 1, 8);
    const uint32_t n_rv_total = total_round_values(d);
    uint64_t round_vals[3 * 64];
    if (len < 8 + 8 * (size_t)(d->num_public_inputs + n_rv_total)) { rc = -1; goto done; }
    {
        size_t tail = len - 8 - 8 * (size_t)(d->num_public_inputs + n_rv_total);
        uint64_t n_pi;
        memcpy(&n_pi, proof + tail, 8);
        if (n_pi != d->num_public_inputs) { rc = -1; goto done; }
        memcpy(pis, proof + tail + 8, 8 * (size_t)n_pi);
        memcpy(round_vals, proof + tail + 8 + 8 * (size_t)n_pi, 8 * (size_t)n_rv_total);
        for (uint32_t i = 0; i < n_pi; i++) if (pis[i] >= GL_P) { rc = -1; goto done; }
        for (uint32_t i = 0; i < n_rv_total; i++) if (round_vals[i] >= GL_P) { rc = -1; goto done; }
        r.len = tail; /* the FRI reader must consume exactly up to here */
    }
    if (r.bad) { rc = -1; goto done; }
    {
        orc_challenger ch;
        orc_ch_init(&ch);
        transcript_start(&ch, d, pis);
        uint32_t n_drawn = 0, rv_off = 0, oo = 0;
        for (uint32_t rd = 0; rd < NRD; rd++) {
            for (uint32_t k = 0; k < batches_of(d, round_cols_of(d, rd)); k++, oo++) orc_ch_observe_many(&ch, caps + oo * capw, capw);
            if (d->n_rounds) {
                for (uint32_t k = 0; k < d->round_values[rd]; k++) pis[d->num_public_inputs + n_drawn++] = round_vals[rv_off + k];
                if (d->round_values[rd]) orc_ch_observe_many(&ch, round_vals + rv_off, d->round_values[rd]);
                rv_off += d->round_values[rd];
                for (uint32_t k = 0; k < d->round_challenges[rd]; k++) pis[d->num_public_inputs + n_drawn++] = orc_ch_challenge(&ch);
            }
        }
        uint64_t alphas[4];
        for (uint32_t j = 0; j < nc; j++) alphas[j] = orc_ch_challenge(&ch);
        orc_ch_observe_many(&ch, caps + NO * capw, capw);
        gl2 zeta = orc_ch_ext_challenge(&ch);
        observe_openings(&ch, d, o_local, o_q, o_next, ncols, nq);
        /* vanishing polynomial identity at zeta */
        const uint64_t g = gl_root_of_unity(log_n), last = gl_inv(g);
        gl2 zeta_n = zeta;
        for (unsigned i = 0; i < log_n; i++) zeta_n = gl2_mul(zeta_n, zeta_n);
        gl2 z_h = gl2_sub(zeta_n, gl2_from(1));
        gl2 l_first = gl2_mul(z_h, gl2_inv(gl2_scale(gl2_sub(zeta, gl2_from(1)), (uint64_t)n % GL_P)));
        gl2 l_last = gl2_mul(z_h, gl2_inv(gl2_scale(gl2_sub(gl2_scale(zeta, g), gl2_from(1)), (uint64_t)n % GL_P)));
        gl2 accs[4], per[ORC_MAX_PERIODIC];
        if (d->n_periodic) {
            const size_t period = (size_t)1 << d->period_bits;
            uint64_t* pc_ = periodic_coeffs(d);
            gl2 y = zeta;
            for (unsigned i = 0; i < log_n - d->period_bits; i++) y = gl2_mul(y, y);
            for (uint32_t a = 0; a < d->n_periodic; a++) {
                gl2 acc = gl2_from(0);
                for (size_t m = period; m-- > 0;) acc = gl2_add_base(gl2_mul(acc, y), pc_[a * period + m]);
                per[a] = acc;
            }
            free(pc_);
        }
        air_eval_ext(d, o_local, o_next, pis, per, gl2_sub(zeta, gl2_from(last)), l_first, l_last, alphas, accs);
        for (uint32_t j = 0; j < nc && rc == 1; j++) {
            gl2 t = gl2_from(0);
            for (uint32_t k = qdf; k-- > 0;) t = gl2_add(gl2_mul(t, zeta_n), o_q[j * qdf + k]);
            if (!gl2_eq(accs[j], gl2_mul(z_h, t))) rc = -2;
        }
        if (rc != 1) goto done;
        gl2 g_zeta = gl2_scale(zeta, g);
        uint32_t col0[ORC_STARK_MAX_ORACLES + 1] = {0};   /* per trace oracle (a round or one of its batches) */
        {
            uint32_t o = 0;
            for (uint32_t rd = 0; rd < NRD; rd++)
                for (uint32_t k = 0; k < batches_of(d, round_cols_of(d, rd)); k++, o++) col0[o + 1] = col0[o] + batch_width(d, round_cols_of(d, rd), k);
        }
        uint32_t* idx_o = (uint32_t*)malloc(4 * 2 * (ncols + nq));
        uint32_t* idx_p = idx_o + ncols + nq;
        for (uint32_t rd = 0; rd < NO; rd++)
            for (uint32_t c = col0[rd]; c < col0[rd + 1]; c++) { idx_o[c] = rd; idx_p[c] = c - col0[rd]; }
        for (uint32_t c = 0; c < nq; c++) { idx_o[ncols + c] = NO; idx_p[ncols + c] = c; }
        orc_fri_batch batches[2] = {{zeta, ncols + nq, idx_o, idx_p}, {g_zeta, ncols, idx_o, idx_p}};
        gl2* open0 = (gl2*)malloc(sizeof(gl2) * (ncols + nq));
        memcpy(open0, o_local, sizeof(gl2) * ncols);
        memcpy(open0 + ncols, o_q, sizeof(gl2) * nq);
        const gl2* opened[2] = {open0, o_next};
        const uint64_t* cap_ptrs[ORC_STARK_MAX_ORACLES + 1];
        uint32_t n_cols[ORC_STARK_MAX_ORACLES + 1];
        for (uint32_t rd = 0; rd < NO; rd++) { cap_ptrs[rd] = caps + rd * capw; n_cols[rd] = col0[rd + 1] - col0[rd]; }
        cap_ptrs[NO] = caps + NO * capw;
        n_cols[NO] = nq;
        orc_fri_params fp = {d->degree_bits, d->rate_bits, cap_h, d->fri_pow_bits, d->fri_num_queries, d->fri_arity_bits, d->fri_final_poly_bits, d->leaf_group_cols};
        rc = fri_verify(&fp, cap_ptrs, n_cols, NO + 1, batches, 2, opened, &ch, &r);
        if (rc == 1 && r.pos != r.len) rc = -11;
        free(open0);
        free(idx_o);
    }
done:
    free(caps); free(o_local); free(pis);
    return rc;
}
