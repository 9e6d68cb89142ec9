// The lookup argument of the plonky2 prover on the device (circuits with LookupGate / LookupTableGate rows).
//
// Replaces, inside prove_with_partition_witness (nearx/src/test_utils.rs:62 -> plonky2::plonk::prover):
//   set_lookup_wires          -> k_lk_count + k_lk_write        (multiplicities, padding slots; witness written in place)
//   compute_all_lookup_polys  -> k_lk_row_terms + k_lk_scan     (RE and the S partial sums per challenge round)
//   check_lookup_constraints_batch (inside compute_quotient_polys) -> k_lookup_terms (their share of the vanishing sums)
//
// Layout facts (gates::lookup / gates::lookup_table, CircuitBuilder::add_all_lookups): a LookupGate row holds
// n_lu_slots = routed / 2 slots (input 2i, output 2i+1), a LookupTableGate row n_lut_slots = routed / 3 slots (input 3i,
// output 3i+1, multiplicity 3i+2); per table the LookupGate rows are [last_lu, last_lut), the table rows [last_lut, first_lut]
// with the table UPSIDE DOWN (entry e on row first_lut - e / slots), and row first_lut + 1 is a NoopGate row where every
// lookup polynomial is zero.  RE, Sum and LDC all run from high rows to low rows, so "previous" means row + 1.
//
// All of this is HBM-bound integer work over a few thousand rows (the two witness passes) or one streaming pass over
// 80 wire + 5 selector + 28 Zs columns of the LDE (the quotient share): one lane per row / point, column-major loads
// coalesce across the wave.
#include <algorithm>
#include "gl.hpp"
#include "prover.hpp"

namespace nlx {
namespace {

// ---------------------------------------------------------------------------------------------------------------------
// set_lookup_wires
// ---------------------------------------------------------------------------------------------------------------------
// One lane per LookupGate slot of table blockIdx.y: the first `lookups` slots are real lookups (counted), the rest of the
// last row is padding: it is given the table's first pair and counted on entry 0.
__global__ __launch_bounds__(256) void k_lk_count(LookupShape s, const LookupTableDev* __restrict__ tabs, uint64_t* __restrict__ wires,
                                                  size_t n, uint32_t* __restrict__ err) {
    const LookupTableDev t = tabs[blockIdx.y];
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t slots = (t.last_lut - t.last_lu) * s.n_lu_slots;
    if (q >= slots) return;
    const uint32_t row = t.last_lu + q / s.n_lu_slots, slot = q % s.n_lu_slots;
    if (q < t.lookups) {
        const uint64_t v = wires[(size_t)(2 * slot) * n + row];
        const int32_t idx = v <= 0xFFFF ? t.idx_of[v] : -1;
        if (idx < 0) { atomicOr(err, 1u); return; }
        atomicAdd(&t.mult[idx], 1u);
    } else {
        const uint32_t first = t.pairs[0];
        wires[(size_t)(2 * slot) * n + row] = first & 0xFFFF;
        wires[(size_t)(2 * slot + 1) * n + row] = first >> 16;
        atomicAdd(&t.mult[0], 1u);
    }
}
// One lane per table entry: its multiplicity wire (the padding slots of the last table row keep their zeros).
__global__ __launch_bounds__(256) void k_lk_write(LookupShape s, const LookupTableDev* __restrict__ tabs, uint64_t* __restrict__ wires, size_t n) {
    const LookupTableDev t = tabs[blockIdx.y];
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= t.len) return;
    wires[(size_t)(3 * (e % s.n_lut_slots) + 2) * n + (t.first_lut - e / s.n_lut_slots)] = t.mult[e];
}

// ---------------------------------------------------------------------------------------------------------------------
// compute_lookup_polys
// ---------------------------------------------------------------------------------------------------------------------
// Pass 1, one lane per (row, j), j <= S, of table blockIdx.y % T under round blockIdx.y / T:
//   j < S:  the row's j-th increment of the running sum:  + sum_{i in group j} mult_i / (alpha - looked_i)  on a table row,
//                                                         - sum_{i in group j} 1 / (alpha - looking_i)      on a LookupGate row
//           (one inversion per group: the fractions are added as (num, den) pairs);
//   j == S: the row's own part of RE on a table row: sum_i (inp_i + B out_i) delta^(slots-1-i).
// Written where the finished values go (column 1 + j resp. 0 of the round), pass 2 scans them in place.
__global__ __launch_bounds__(256) void k_lk_row_terms(LookupShape s, const LookupTableDev* __restrict__ tabs,
                                                      const uint64_t* __restrict__ wires, size_t n,
                                                      const uint64_t* __restrict__ deltas, uint64_t* __restrict__ cols) {
    const uint32_t ci = blockIdx.y / s.num_luts;
    const LookupTableDev t = tabs[blockIdx.y % s.num_luts];
    const uint32_t S = s.n_sldc;
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t rows = t.first_lut - t.last_lu + 1;
    if (x >= rows * (S + 1)) return;
    const uint32_t row = t.last_lu + x / (S + 1), j = x % (S + 1);
    const uint64_t dA = deltas[4 * ci], dB = deltas[4 * ci + 1], alpha = deltas[4 * ci + 2], delta = deltas[4 * ci + 3];
    uint64_t* out = cols + (size_t)ci * (1 + S) * n;
    const bool table_row = row >= t.last_lut;
    auto Wr = [&](uint32_t c) { return wires[(size_t)c * n + row]; };
    if (j == S) {
        if (!table_row) return;
        uint64_t re = 0;
        for (uint32_t i = 0; i < s.n_lut_slots; i++)
            re = gl::add(gl::mul(re, delta), gl::add(Wr(3 * i), gl::mul(dB, Wr(3 * i + 1))));
        out[row] = re;
        return;
    }
    uint64_t num = 0, den = 1;
    if (table_row) {
        const uint32_t hi = (j + 1) * s.lut_degree < s.n_lut_slots ? (j + 1) * s.lut_degree : s.n_lut_slots;
        for (uint32_t i = j * s.lut_degree; i < hi; i++) {
            const uint64_t d = gl::sub(alpha, gl::add(Wr(3 * i), gl::mul(dA, Wr(3 * i + 1))));
            num = gl::add(gl::mul(num, d), gl::mul(Wr(3 * i + 2), den));
            den = gl::mul(den, d);
        }
    } else {
        const uint32_t hi = (j + 1) * s.lu_degree < s.n_lu_slots ? (j + 1) * s.lu_degree : s.n_lu_slots;
        for (uint32_t i = j * s.lu_degree; i < hi; i++) {
            const uint64_t d = gl::sub(alpha, gl::add(Wr(2 * i), gl::mul(dA, Wr(2 * i + 1))));
            num = gl::add(gl::mul(num, d), den);
            den = gl::mul(den, d);
        }
    }
    // alpha - combo == 0 has probability 2^-64 per slot; upstream's batch inversion would panic on it, here the row's
    // increment becomes 0 and the proof does not verify
    const uint64_t v = gl::mul(num, gl::inv(den));
    out[(size_t)(1 + j) * n + row] = table_row ? v : gl::sub(0, v);
}

// Pass 2, one block per (round, table): both recurrences in place, each as a three-phase block scan (lane = a contiguous run
// of the sequence, the 256 run totals combined by lane 0).
//   running sum:  sequence index q = (first_lut - row) S + j over all rows of the table's block;  value += increment
//   RE:           sequence index q = first_lut - row over the table rows;  value = value * delta^slots + own part
constexpr uint32_t LK_SCAN_THREADS = 256;
__global__ __launch_bounds__(LK_SCAN_THREADS) void k_lk_scan(LookupShape s, const LookupTableDev* __restrict__ tabs, size_t n,
                                                             const uint64_t* __restrict__ deltas, uint64_t* __restrict__ cols) {
    __shared__ uint64_t sh_a[LK_SCAN_THREADS], sh_b[LK_SCAN_THREADS];
    const uint32_t ci = blockIdx.x / s.num_luts;
    const LookupTableDev t = tabs[blockIdx.x % s.num_luts];
    const uint32_t S = s.n_sldc, tid = threadIdx.x;
    uint64_t* out = cols + (size_t)ci * (1 + S) * n;
    // ---- running sum ----
    {
        const uint32_t M = (t.first_lut - t.last_lu + 1) * S;
        const uint32_t per = (M + LK_SCAN_THREADS - 1) / LK_SCAN_THREADS;
        const uint32_t lo = tid * per < M ? tid * per : M, hi = lo + per < M ? lo + per : M;
        auto at = [&](uint32_t q) -> uint64_t& { return out[(size_t)(1 + q % S) * n + (t.first_lut - q / S)]; };
        uint64_t sum = 0;
        for (uint32_t q = lo; q < hi; q++) sum = gl::add(sum, at(q));
        sh_a[tid] = sum;
        __syncthreads();
        if (tid == 0) {
            uint64_t run = 0;
            for (uint32_t i = 0; i < LK_SCAN_THREADS; i++) { const uint64_t v = sh_a[i]; sh_a[i] = run; run = gl::add(run, v); }
        }
        __syncthreads();
        uint64_t run = sh_a[tid];
        for (uint32_t q = lo; q < hi; q++) { run = gl::add(run, at(q)); at(q) = run; }
    }
    __syncthreads();
    // ---- RE ----
    {
        const uint32_t M = t.first_lut - t.last_lut + 1;
        const uint32_t per = (M + LK_SCAN_THREADS - 1) / LK_SCAN_THREADS;
        const uint32_t lo = tid * per < M ? tid * per : M, hi = lo + per < M ? lo + per : M;
        uint64_t D = 1;   // delta^slots: what one row multiplies the incoming value by
        {
            const uint64_t delta = deltas[4 * ci + 3];
            for (uint32_t i = 0; i < s.n_lut_slots; i++) D = gl::mul(D, delta);
        }
        auto at = [&](uint32_t q) -> uint64_t& { return out[t.first_lut - q]; };
        uint64_t a = 1, b = 0;   // the run as an affine map: in -> in * a + b
        for (uint32_t q = lo; q < hi; q++) { b = gl::add(gl::mul(b, D), at(q)); a = gl::mul(a, D); }
        sh_a[tid] = a;
        sh_b[tid] = b;
        __syncthreads();
        if (tid == 0) {
            uint64_t run = 0;
            for (uint32_t i = 0; i < LK_SCAN_THREADS; i++) {
                const uint64_t in = run;
                run = gl::add(gl::mul(run, sh_a[i]), sh_b[i]);
                sh_b[i] = in;
            }
        }
        __syncthreads();
        uint64_t run = sh_b[tid];
        for (uint32_t q = lo; q < hi; q++) { run = gl::add(gl::mul(run, D), at(q)); at(q) = run; }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// check_lookup_constraints on the LDE domain
// ---------------------------------------------------------------------------------------------------------------------
// One lane per LDE point: the 4 + T + 2 S lookup terms of every challenge round, each multiplied into BOTH alpha sums (every
// alpha reduces the whole vanishing-term list).  Term order of a round (vanishing_poly::check_lookup_constraints):
//   LastLdc SLDC_{S-1} | InitSre SLDC_0 | InitSre RE | end_t (RE - lut_poly_t), t < T | TransSre (RE - Horner(RE_next, row))
//   | per group j: TransSre (lut_prod (SLDC_j - prev) - sum_i mult_i prod_{k != i}) , TransLdc (lu_prod (SLDC_j - prev) + sum_i prod_{k != i})
// sum_i m_i prod_{k != i} d_k is built by the recurrence (num, den) -> (num d + m den, den d): three multiplications per slot.
__global__ __launch_bounds__(256) void k_lookup_terms(LookupTermsParams p) {
    const size_t pos = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned log_L = p.log_n + p.rate_bits;
    if (pos >> log_L) return;
    const size_t n = (size_t)1 << p.log_n, L = (size_t)1 << log_L;
    const size_t pos_next = (pos & ~(n - 1)) | ((pos + 1) & (n - 1));   // same coset, next subgroup element
    const uint32_t S = p.s.n_sldc, T = p.s.num_luts, nc = p.nc;          // nc <= 2 (checked at nlx_circuit_build)
    const uint64_t* ap0 = p.alpha_pows + p.t_lk;
    const uint64_t* ap1 = p.alpha_pows + p.alpha_stride + p.t_lk;
    auto Sel = [&](uint32_t c) { return p.cs[(size_t)(p.sel0 + c) * L + pos]; };
    auto Z = [&](uint32_t ci, uint32_t c) { return p.zs[(size_t)(p.lk0 + ci * (1 + S) + c) * L + pos]; };
    auto Zn = [&](uint32_t ci, uint32_t c) { return p.zs[(size_t)(p.lk0 + ci * (1 + S) + c) * L + pos_next]; };
    uint64_t tot0 = 0, tot1 = 0;
    auto emit_at = [&](uint32_t ci, uint32_t k, uint64_t term) {   // term index inside round ci's list (see the order above)
        const uint32_t at = ci * p.n_lk_terms + k;
        tot0 = gl::add(tot0, gl::mul(term, ap0[at]));
        tot1 = gl::add(tot1, gl::mul(term, ap1[at]));
    };
    const uint64_t s_sre = Sel(0), s_ldc = Sel(1), s_init = Sel(2), s_last = Sel(3);
    // the terms that need no wire
    uint64_t dA[2], dB[2], alpha[2], delta[2], re[2], re_run[2];
    for (uint32_t ci = 0; ci < nc; ci++) {
        dA[ci] = p.deltas[4 * ci]; dB[ci] = p.deltas[4 * ci + 1]; alpha[ci] = p.deltas[4 * ci + 2]; delta[ci] = p.deltas[4 * ci + 3];
        re[ci] = Z(ci, 0);
        re_run[ci] = Zn(ci, 0);
        emit_at(ci, 0, gl::mul(s_last, Z(ci, S)));
        emit_at(ci, 1, gl::mul(s_init, Z(ci, 1)));
        emit_at(ci, 2, gl::mul(s_init, re[ci]));
        for (uint32_t t = 0; t < T; t++) emit_at(ci, 3 + t, gl::mul(Sel(4 + t), gl::sub(re[ci], p.lut_polys[ci * T + t])));
    }
    // ONE pass over the routed wires: wire w is element (w mod 3) of LookupTableGate slot w / 3 AND element (w mod 2) of
    // LookupGate slot w / 2 - both gates' sums advance from the same load, for both challenge rounds.  (The first version
    // walked the wires once per sum and round: 420 loads per point for 80 columns, 7.2 GB fetched per launch at 2^18 rows
    // against 1.9 GB algorithmic - profiles/r03_pmc_new_kernels.txt.)
    const uint32_t n_lut_w = 3 * p.s.n_lut_slots, n_lu_w = 2 * p.s.n_lu_slots, n_w = n_lut_w > n_lu_w ? n_lut_w : n_lu_w;
    uint64_t inp3 = 0, inp2 = 0, combo[2] = {0, 0};
    uint64_t lut_num[2] = {0, 0}, lut_den[2] = {1, 1}, lu_num[2] = {0, 0}, lu_den[2] = {1, 1};
    uint32_t r3 = 0, slot3 = 0, in_group3 = 0, group3 = 0, slot2 = 0, in_group2 = 0, group2 = 0;
#pragma unroll 1
    for (uint32_t w = 0; w < n_w; w++) {
        const uint64_t v = p.wires[(size_t)w * L + pos];
        if (w < n_lut_w) {
            if (r3 == 0) {
                inp3 = v;
            } else if (r3 == 1) {
                for (uint32_t ci = 0; ci < nc; ci++) {
                    combo[ci] = gl::add(inp3, gl::mul(dA[ci], v));
                    re_run[ci] = gl::add(gl::mul(re_run[ci], delta[ci]), gl::add(inp3, gl::mul(dB[ci], v)));
                }
            } else {
                for (uint32_t ci = 0; ci < nc; ci++) {   // (num, den) -> (num d + mult den, den d)
                    const uint64_t d = gl::sub(alpha[ci], combo[ci]);
                    lut_num[ci] = gl::add(gl::mul(lut_num[ci], d), gl::mul(v, lut_den[ci]));
                    lut_den[ci] = gl::mul(lut_den[ci], d);
                }
                slot3++;
                if (++in_group3 == p.s.lut_degree || slot3 == p.s.n_lut_slots) {   // the group's Sum transition
                    for (uint32_t ci = 0; ci < nc; ci++) {
                        const uint64_t prev = group3 ? Z(ci, group3) : Zn(ci, S);   // rows run upside down: row + 1 is "before"
                        const uint64_t step = gl::sub(Z(ci, 1 + group3), prev);
                        emit_at(ci, 4 + T + 2 * group3, gl::mul(s_sre, gl::sub(gl::mul(lut_den[ci], step), lut_num[ci])));
                        lut_num[ci] = 0;
                        lut_den[ci] = 1;
                    }
                    in_group3 = 0;
                    group3++;
                }
            }
            r3 = r3 == 2 ? 0 : r3 + 1;
        }
        if (w < n_lu_w) {
            if ((w & 1) == 0) {
                inp2 = v;
            } else {
                for (uint32_t ci = 0; ci < nc; ci++) {   // (num, den) -> (num d + den, den d)
                    const uint64_t d = gl::sub(alpha[ci], gl::add(inp2, gl::mul(dA[ci], v)));
                    lu_num[ci] = gl::add(gl::mul(lu_num[ci], d), lu_den[ci]);
                    lu_den[ci] = gl::mul(lu_den[ci], d);
                }
                slot2++;
                if (++in_group2 == p.s.lu_degree || slot2 == p.s.n_lu_slots) {     // the group's LDC transition
                    for (uint32_t ci = 0; ci < nc; ci++) {
                        const uint64_t prev = group2 ? Z(ci, group2) : Zn(ci, S);
                        const uint64_t step = gl::sub(Z(ci, 1 + group2), prev);
                        emit_at(ci, 5 + T + 2 * group2, gl::mul(s_ldc, gl::add(gl::mul(lu_den[ci], step), lu_num[ci])));
                        lu_num[ci] = 0;
                        lu_den[ci] = 1;
                    }
                    in_group2 = 0;
                    group2++;
                }
            }
        }
    }
    // groups past the last slot (shapes where S * degree overshoots the slots): empty products, SLDC_j = SLDC_{j-1}
    for (; group3 < S; group3++)
        for (uint32_t ci = 0; ci < nc; ci++)
            emit_at(ci, 4 + T + 2 * group3, gl::mul(s_sre, gl::sub(Z(ci, 1 + group3), group3 ? Z(ci, group3) : Zn(ci, S))));
    for (; group2 < S; group2++)
        for (uint32_t ci = 0; ci < nc; ci++)
            emit_at(ci, 5 + T + 2 * group2, gl::mul(s_ldc, gl::sub(Z(ci, 1 + group2), group2 ? Z(ci, group2) : Zn(ci, S))));
    for (uint32_t ci = 0; ci < nc; ci++) emit_at(ci, 3 + T, gl::mul(s_sre, gl::sub(re[ci], re_run[ci])));   // RE's row transition
    p.out[pos] = tot0;
    if (nc > 1) p.out[L + pos] = tot1;
}

}  // namespace

void launch_set_lookup_wires(hipStream_t st, const LookupShape& s, const LookupTableDev* d_tabs, const LookupTableDev* h_tabs,
                             uint64_t* d_wires, size_t n, uint32_t* d_mult_all, size_t mult_words, uint32_t* d_err) {
    (void)hipMemsetAsync(d_mult_all, 0, mult_words * 4, st);
    (void)hipMemsetAsync(d_err, 0, 4, st);
    uint32_t max_slots = 1, max_len = 1;
    for (uint32_t t = 0; t < s.num_luts; t++) {
        max_slots = std::max(max_slots, (h_tabs[t].last_lut - h_tabs[t].last_lu) * s.n_lu_slots);
        max_len = std::max(max_len, h_tabs[t].len);
    }
    hipLaunchKernelGGL(k_lk_count, dim3((max_slots + 255) / 256, s.num_luts), dim3(256), 0, st, s, d_tabs, d_wires, n, d_err);
    hipLaunchKernelGGL(k_lk_write, dim3((max_len + 255) / 256, s.num_luts), dim3(256), 0, st, s, d_tabs, d_wires, n);
}

void launch_lookup_polys(hipStream_t st, const LookupShape& s, const LookupTableDev* d_tabs, const LookupTableDev* h_tabs,
                         const uint64_t* d_wires, size_t n, uint32_t nc, const uint64_t* d_deltas, uint64_t* d_cols) {
    (void)hipMemsetAsync(d_cols, 0, (size_t)nc * (1 + s.n_sldc) * n * 8, st);
    uint32_t max_items = 1;
    for (uint32_t t = 0; t < s.num_luts; t++)
        max_items = std::max(max_items, (h_tabs[t].first_lut - h_tabs[t].last_lu + 1) * (s.n_sldc + 1));
    hipLaunchKernelGGL(k_lk_row_terms, dim3((max_items + 255) / 256, nc * s.num_luts), dim3(256), 0, st, s, d_tabs, d_wires, n,
                       d_deltas, d_cols);
    hipLaunchKernelGGL(k_lk_scan, dim3(nc * s.num_luts), dim3(LK_SCAN_THREADS), 0, st, s, d_tabs, n, d_deltas, d_cols);
}

void launch_lookup_terms(hipStream_t st, const LookupTermsParams& p) {
    const size_t L = (size_t)1 << (p.log_n + p.rate_bits);
    hipLaunchKernelGGL(k_lookup_terms, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, st, p);
}

}  // namespace nlx
