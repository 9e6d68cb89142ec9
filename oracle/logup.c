/* TEST INFRASTRUCTURE ONLY - CPU restatement of the range-check lookup columns (LogUp in the quadratic extension)
 * that near-light-client_amd/csrc/logup.hip computes on the GPU.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may call this; the product never does.
 *
 * What it restates (parity unpinned: starkyx's lookup/bus argument is not in the reference tree, Cargo.lock:6515; this
 * follows the published log-derivative lookup argument, Haboeck 2022, "Multivariate lookups based on logarithmic
 * derivatives"): for lookup cells v and the table t(i) = i mod 2^bits with multiplicities m,
 *     sum over rows and lookups of 1 / (alpha + v)  =  sum over rows of m / (alpha + t),   alpha in F_p^2.
 * Round-1 columns: one helper h = 1/(alpha+v1) + 1/(alpha+v2) per pair of lookups (a last single one if the count is
 * odd), g = m / (alpha + t), and the running sum phi(0) = 0, phi(i+1) = phi(i) + sum h(i) - g(i).  Extension elements
 * are stored as two base columns (a, b) = a + b X, X^2 = 7. */
#include "gl.h"
#include <stdlib.h>
#include <string.h>

/* multiplicities: the count of value v goes to row v.  Returns 1, or 0 if a looked-up cell is outside the table. */
int orc_logup_multiplicities(const uint64_t* trace, uint32_t log_n, const uint32_t* cols, uint32_t n_lookups,
                             uint32_t table_bits, uint64_t* mult) {
    const size_t n = (size_t)1 << log_n;
    if (table_bits > log_n) return 0;
    memset(mult, 0, 8 * n);
    for (uint32_t l = 0; l < n_lookups; l++) {
        const uint64_t* col = trace + (size_t)cols[l] * n;
        for (size_t i = 0; i < n; i++) {
            if (col[i] >> table_bits) return 0;
            mult[col[i]]++;
        }
    }
    return 1;
}

uint32_t orc_logup_round_cols(uint32_t n_lookups) { return 2 * ((n_lookups + 1) / 2) + 4; }

void orc_logup_round(const uint64_t* trace, uint32_t log_n, const uint32_t* cols, uint32_t n_lookups, uint32_t table_bits,
                     const uint64_t* mult, const uint64_t alpha[2], uint64_t* out) {
    const size_t n = (size_t)1 << log_n;
    const uint32_t H = (n_lookups + 1) / 2;
    const gl2 al = gl2_make(alpha[0] % GL_P, alpha[1] % GL_P);
    uint64_t* g0 = out + (size_t)(2 * H) * n;
    uint64_t* g1 = g0 + n;
    uint64_t* phi0 = g1 + n;
    uint64_t* phi1 = phi0 + n;
    gl2* rowsum = (gl2*)malloc(sizeof(gl2) * n);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        gl2 acc = gl2_from(0);
        for (uint32_t j = 0; j < H; j++) {
            gl2 h = gl2_inv(gl2_add(al, gl2_from(trace[(size_t)cols[2 * j] * n + i] % GL_P)));
            if (2 * j + 1 < n_lookups) h = gl2_add(h, gl2_inv(gl2_add(al, gl2_from(trace[(size_t)cols[2 * j + 1] * n + i] % GL_P))));
            out[(size_t)(2 * j) * n + i] = h.a;
            out[(size_t)(2 * j + 1) * n + i] = h.b;
            acc = gl2_add(acc, h);
        }
        const uint64_t t = (uint64_t)(i & (((size_t)1 << table_bits) - 1));
        const gl2 g = gl2_scale(gl2_inv(gl2_add(al, gl2_from(t))), mult[i] % GL_P);
        g0[i] = g.a;
        g1[i] = g.b;
        rowsum[i] = gl2_sub(acc, g);
    }
    gl2 run = gl2_from(0);
    for (size_t i = 0; i < n; i++) {
        phi0[i] = run.a;
        phi1[i] = run.b;
        run = gl2_add(run, rowsum[i]);
    }
    free(rowsum);
}
