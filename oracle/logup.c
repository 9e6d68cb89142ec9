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

/* The table {0 .. 2^bits - 1} may be spread over `table_cols` (a power of two) periodic columns of period
 * P = 2^bits / table_cols, column j holding j P + (i mod P): a short trace can then carry a table longer than itself.
 * multiplicities: `mult` is table_cols consecutive columns; the count of value v goes to column v / P, row v mod P.
 * Returns 1, or 0 if a looked-up cell is outside the table. */
int orc_logup_multiplicities(const uint64_t* trace, uint32_t log_n, const uint32_t* cols, uint32_t n_lookups,
                             uint32_t table_bits, uint32_t table_cols, uint64_t* mult) {
    const size_t n = (size_t)1 << log_n;
    uint32_t lk = 0;
    while ((1u << lk) < table_cols) lk++;
    if ((1u << lk) != table_cols || lk > table_bits || table_bits - lk > log_n) return 0;
    const uint32_t pb = table_bits - lk;
    memset(mult, 0, 8 * n * table_cols);
    for (uint32_t l = 0; l < n_lookups; l++) {
        const uint64_t* col = trace + (size_t)cols[l] * n;
        for (size_t i = 0; i < n; i++) {
            if (col[i] >> table_bits) return 0;
            mult[(size_t)(col[i] >> pb) * n + (col[i] & (((uint64_t)1 << pb) - 1))]++;
        }
    }
    return 1;
}

uint32_t orc_logup_round_cols(uint32_t n_lookups, uint32_t table_cols) { return 2 * ((n_lookups + 1) / 2) + 2 * table_cols + 2; }

/* out: helpers (2 H columns), g_0 .. g_(table_cols-1) (two columns each), phi (two columns) */
void orc_logup_round(const uint64_t* trace, uint32_t log_n, const uint32_t* cols, uint32_t n_lookups, uint32_t table_bits,
                     uint32_t table_cols, const uint64_t* mult, const uint64_t alpha[2], uint64_t* out) {
    const size_t n = (size_t)1 << log_n;
    const uint32_t H = (n_lookups + 1) / 2;
    uint32_t lk = 0;
    while ((1u << lk) < table_cols) lk++;
    const uint32_t pb = table_bits - lk;
    const gl2 al = gl2_make(alpha[0] % GL_P, alpha[1] % GL_P);
    uint64_t* gcols = out + (size_t)(2 * H) * n;
    uint64_t* phi0 = gcols + (size_t)(2 * table_cols) * n;
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
        for (uint32_t c = 0; c < table_cols; c++) {
            const uint64_t t = ((uint64_t)c << pb) + (uint64_t)(i & (((size_t)1 << pb) - 1));
            const gl2 g = gl2_scale(gl2_inv(gl2_add(al, gl2_from(t))), mult[(size_t)c * n + i] % GL_P);
            gcols[(size_t)(2 * c) * n + i] = g.a;
            gcols[(size_t)(2 * c + 1) * n + i] = g.b;
            acc = gl2_sub(acc, g);
        }
        rowsum[i] = acc;
    }
    gl2 run = gl2_from(0);
    for (size_t i = 0; i < n; i++) {
        phi0[i] = run.a;
        phi1[i] = run.b;
        run = gl2_add(run, rowsum[i]);
    }
    free(rowsum);
}
