// Prover-stage kernels for gfx950: permutation-argument partial products / Z (a8), the
// constraint + quotient combiner (a9), FRI batching / folding / layer commitment (a11), the
// proof-of-work grind and the query gathers.
//
// Replaces plonky2::plonk::prover::{wires_permutation_partial_products_and_zs, compute_quotient_polys},
// vanishing_poly::eval_vanishing_poly_base_batch, gates::*::eval_unfiltered_base_batch,
// fri::oracle::PolynomialBatch::prove_openings, fri::prover::{fri_committed_trees,
// fri_proof_of_work, fri_prover_query_rounds}  (SURVEY.md §3.4 steps 4-7, §8a rows a8-a11).
//
// All tables are coset-major natural order ([col][r][k] <-> LDE point g*w_L^(8k+r)), so every
// kernel below reads 512 contiguous bytes per column per wave; "next row" (x*w_n) is k+1 in the
// same coset.  FRI is folded in the VALUE domain (16-point inverse DFT per coset + Horner in
// beta/x): bit-identical to plonky2's coefficient-domain fold followed by a fresh FFT, with no
// transform at all between rounds.
#include <hip/hip_runtime.h>
#include "gl.hpp"
#include "poseidon.hpp"
#include "poseidon_fast_constants.inc"
#include "prover.hpp"

namespace nlx {

__constant__ static const uint64_t FAST_FIRST[12] = NLX_POSEIDON_FAST_FIRST_RC_INIT;
__constant__ static const uint64_t FAST_RC[22] = NLX_POSEIDON_FAST_RC_INIT;
__constant__ static const uint64_t FAST_VS[22][11] = NLX_POSEIDON_FAST_VS_INIT;
__constant__ static const uint64_t FAST_W[22][11] = NLX_POSEIDON_FAST_W_HATS_INIT;
__constant__ static const uint64_t FAST_INIT[11][11] = NLX_POSEIDON_FAST_INITIAL_MATRIX_INIT;

__device__ __forceinline__ uint64_t root_pow(const uint64_t* __restrict__ half_table, uint32_t e, uint32_t half) {
    return e < half ? half_table[e] : gl::P - half_table[e - half];
}

// =====================================================================================
// a8: partial products and Z
// =====================================================================================
// Stage 1: per row i and challenge c, the cumulative chunk quotients
//   cum_q = prod_{q' <= q} prod_{j in chunk q'} (w_j + beta k_j x + gamma) / (w_j + beta sigma_j + gamma)
// cum_0..cum_8 go to the partial-product columns, cum_9 (the row product) to the Z column.
// One lane per row; all denominators of a row (both challenges) share ONE field inversion.
__global__ __launch_bounds__(256) void k_zs_row_products(ZsParams p) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = (size_t)1 << p.log_n;
    if (i >= n) return;
    const uint64_t x = root_pow(p.w_n_table, (uint32_t)i, (uint32_t)(n >> 1));
    constexpr int MAXQ = 10;  // chunks per challenge (ceil(80 / 8)); every array below is indexed by unrolled loop counters
    uint64_t num[2 * MAXQ], den[2 * MAXQ];   // only: they stay in registers (runtime indices had put them in scratch, 496 bytes per lane)
    const uint32_t n_chunks = (p.routed + p.chunk - 1) / p.chunk;
    // every wire / sigma value is loaded ONCE and feeds both challenges (the first version walked the 160 columns once per
    // challenge: PMC 0.86 GB fetched per launch at 2^18 rows for 0.34 GB of columns)
    const uint64_t bx0 = gl::mul(p.betas[0], x), bx1 = gl::mul(p.betas[1], x);
#pragma unroll
    for (int q = 0; q < MAXQ; q++) {
        uint64_t nm0 = 1, dn0 = 1, nm1 = 1, dn1 = 1;
        if ((uint32_t)q < n_chunks) {
            for (uint32_t j = q * p.chunk; j < (q + 1) * p.chunk && j < p.routed; j++) {
                const uint64_t w = p.wires[(size_t)j * p.wires_stride + i];
                const uint64_t sg = p.sigmas[(size_t)j * n + i];
                const uint64_t kj = p.k_is[j];
                nm0 = gl::mul(nm0, gl::add(gl::add(w, gl::mul(bx0, kj)), p.gammas[0]));
                dn0 = gl::mul(dn0, gl::add(gl::add(w, gl::mul(p.betas[0], sg)), p.gammas[0]));
                if (p.nc > 1) {
                    nm1 = gl::mul(nm1, gl::add(gl::add(w, gl::mul(bx1, kj)), p.gammas[1]));
                    dn1 = gl::mul(dn1, gl::add(gl::add(w, gl::mul(p.betas[1], sg)), p.gammas[1]));
                }
            }
        }
        num[q] = nm0;
        den[q] = dn0;
        num[MAXQ + q] = nm1;
        den[MAXQ + q] = dn1;   // unused chunks / the unused challenge hold 1: they pass through the batch inversion unchanged
    }
    // batch inversion of all denominators of this row: pref[t] = den[0] .. den[t-1], one inversion, then backwards
    uint64_t pref[2 * MAXQ];
    uint64_t acc = 1;
#pragma unroll
    for (int t = 0; t < 2 * MAXQ; t++) {
        pref[t] = acc;
        acc = gl::mul(acc, den[t]);
    }
    uint64_t inv = gl::inv(acc);
#pragma unroll
    for (int t = 2 * MAXQ - 1; t >= 0; t--) {
        const uint64_t d = den[t];
        den[t] = gl::mul(inv, pref[t]);  // 1 / den
        inv = gl::mul(inv, d);
    }
#pragma unroll
    for (int c = 0; c < 2; c++) {
        uint64_t cum = 1;
#pragma unroll
        for (int q = 0; q < MAXQ; q++) {
            if ((uint32_t)c < p.nc && (uint32_t)q < n_chunks) {
                cum = gl::mul(cum, gl::mul(num[c * MAXQ + q], den[c * MAXQ + q]));
                if ((uint32_t)q + 1 < n_chunks) p.out[(size_t)(p.nc + c * p.npp + q) * n + i] = cum;
                else p.out[(size_t)c * n + i] = cum;  // row product, turned into Z by the scan
            }
        }
    }
}

// Stage 2: exclusive prefix product down the rows (Z_0 = 1, Z_{i+1} = Z_i * P_i).
// Block-local scan of SCAN_ELEMS elements + block totals; totals are scanned by one block.
constexpr unsigned SCAN_PER_THREAD = 8;
constexpr unsigned SCAN_ELEMS = 256 * SCAN_PER_THREAD;

__device__ __forceinline__ uint64_t wave_scan_incl_mul(uint64_t v) {
    const unsigned lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t lo = __shfl_up((uint32_t)v, off, 64);
        uint32_t hi = __shfl_up((uint32_t)(v >> 32), off, 64);
        uint64_t o = ((uint64_t)hi << 32) | lo;
        if (lane >= (unsigned)off) v = gl::mul(v, o);
    }
    return v;
}

// in-place: data[col][i] <- exclusive prefix product within the block; totals[col][block] <- block product
__global__ __launch_bounds__(256) void k_scan_mul_local(uint64_t* __restrict__ data, size_t stride, size_t count,
                                                        uint64_t* __restrict__ totals, size_t totals_stride) {
    __shared__ uint64_t wave_tot[4];
    uint64_t* col = data + (size_t)blockIdx.y * stride;
    const size_t base = (size_t)blockIdx.x * SCAN_ELEMS + (size_t)threadIdx.x * SCAN_PER_THREAD;
    uint64_t v[SCAN_PER_THREAD];
    uint64_t prod = 1;
#pragma unroll
    for (unsigned k = 0; k < SCAN_PER_THREAD; k++) {
        v[k] = base + k < count ? col[base + k] : 1;
        prod = gl::mul(prod, v[k]);
    }
    uint64_t incl = wave_scan_incl_mul(prod);
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint64_t wave_prefix = 1;
    for (unsigned w2 = 0; w2 < wave; w2++) wave_prefix = gl::mul(wave_prefix, wave_tot[w2]);
    // exclusive prefix of this thread = wave_prefix * (inclusive of previous lane)
    uint32_t plo = __shfl_up((uint32_t)incl, 1, 64), phi = __shfl_up((uint32_t)(incl >> 32), 1, 64);
    uint64_t prev = lane ? (((uint64_t)phi << 32) | plo) : 1;
    uint64_t run = gl::mul(wave_prefix, prev);
#pragma unroll
    for (unsigned k = 0; k < SCAN_PER_THREAD; k++) {
        if (base + k < count) col[base + k] = run;
        run = gl::mul(run, v[k]);
    }
    if (threadIdx.x == 255) totals[(size_t)blockIdx.y * totals_stride + blockIdx.x] = run;
}

// data[col][i] *= prefix[col][i / SCAN_ELEMS]  (prefix = exclusive scan of block totals)
__global__ __launch_bounds__(256) void k_scan_mul_apply(uint64_t* __restrict__ data, size_t stride, size_t count,
                                                        const uint64_t* __restrict__ prefix, size_t prefix_stride) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint64_t* col = data + (size_t)blockIdx.y * stride;
    const uint64_t f = prefix[(size_t)blockIdx.y * prefix_stride + i / SCAN_ELEMS];
    col[i] = gl::mul(col[i], f);
}

// Stage 3: partial products pp_q(i) = Z_i * cum_q(i)
__global__ __launch_bounds__(256) void k_zs_apply(uint64_t* __restrict__ out, unsigned log_n, uint32_t nc, uint32_t npp) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = (size_t)1 << log_n;
    if (i >= n) return;
    const uint32_t c = blockIdx.y / npp, q = blockIdx.y % npp;
    const uint64_t z = out[(size_t)c * n + i];
    uint64_t* pp = out + (size_t)(nc + c * npp + q) * n;
    pp[i] = gl::mul(pp[i], z);
}

void launch_zs(hipStream_t st, const ZsParams& p, uint64_t* d_scratch) {
    const size_t n = (size_t)1 << p.log_n;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_zs_row_products, dim3(blocks), dim3(256), 0, st, p);
    // exclusive scan of the nc Z columns (columns 0..nc-1 of p.out)
    size_t count = n;
    size_t nb = (count + SCAN_ELEMS - 1) / SCAN_ELEMS;
    uint64_t* tot1 = d_scratch;                 // nc x nb
    uint64_t* tot2 = d_scratch + p.nc * nb;     // nc x nb2
    hipLaunchKernelGGL(k_scan_mul_local, dim3((unsigned)nb, p.nc), dim3(256), 0, st, p.out, n, count, tot1, nb);
    if (nb > 1) {
        size_t nb2 = (nb + SCAN_ELEMS - 1) / SCAN_ELEMS;  // 1 for n <= 2^22
        hipLaunchKernelGGL(k_scan_mul_local, dim3((unsigned)nb2, p.nc), dim3(256), 0, st, tot1, nb, nb, tot2, nb2);
        if (nb2 > 1) {
            // n > 2^22 rows: third level (nb2 <= 2048 -> one block)
            uint64_t* tot3 = tot2 + p.nc * nb2;
            hipLaunchKernelGGL(k_scan_mul_local, dim3(1, p.nc), dim3(256), 0, st, tot2, nb2, nb2, tot3, (size_t)1);
            hipLaunchKernelGGL(k_scan_mul_apply, dim3((unsigned)((nb + 255) / 256), p.nc), dim3(256), 0, st, tot1, nb, nb,
                               tot2, nb2);
        }
        hipLaunchKernelGGL(k_scan_mul_apply, dim3(blocks, p.nc), dim3(256), 0, st, p.out, n, count, tot1, nb);
    }
    hipLaunchKernelGGL(k_zs_apply, dim3(blocks, p.nc * p.npp), dim3(256), 0, st, p.out, p.log_n, p.nc, p.npp);
}
size_t zs_scratch_words(unsigned log_n, uint32_t nc) {
    size_t n = (size_t)1 << log_n;
    size_t nb = (n + SCAN_ELEMS - 1) / SCAN_ELEMS;
    size_t nb2 = (nb + SCAN_ELEMS - 1) / SCAN_ELEMS;
    return nc * (nb + nb2 + 2) + 16;
}

// =====================================================================================
// a9: constraint / quotient combiner
// =====================================================================================
// Running sums S_c = sum_k alpha_c^k * constraint_k of the gate being evaluated, for both challenges, with
// LAZY reduction: each 64x64 product is accumulated as four 32x32 partial products into four 64-bit columns
// (+ a carry counter each) - two instructions per partial product - and the 160-bit total is reduced once per
// gate.  A reduced multiply-add costs ~30 instructions; this costs 8 per challenge.
struct GateAcc {
    const uint64_t* __restrict__ ap0;  // alpha_0^(T0 + k), wave-uniform
    const uint64_t* __restrict__ ap1;
    uint64_t a[8];
    uint32_t kc[8];
    uint32_t k;
    uint64_t base[2];   // what stash() folded away so far (canonical)
    __device__ __forceinline__ void reset() {
#pragma unroll
        for (int i = 0; i < 8; i++) { a[i] = 0; kc[i] = 0; }
        k = 0;
        base[0] = base[1] = 0;
    }
    // the 24 registers of the columns -> two canonical sums (PoseidonGate's fused partial rounds need the registers for the
    // matrix pass between two groups of constraints); the constraint counter keeps running
    __device__ __forceinline__ void stash() {
        base[0] = gl::add(base[0], fold_columns(a, kc));
        base[1] = gl::add(base[1], fold_columns(a + 4, kc + 4));
#pragma unroll
        for (int i = 0; i < 8; i++) { a[i] = 0; kc[i] = 0; }
    }
    __device__ __forceinline__ void emit_at(uint32_t idx, uint64_t c) { mac(c, ap0[idx], ap1[idx]); }
    // sums 0 and 1 += c * b0, c * b1 (b0, b1 wave-uniform)
    __device__ __forceinline__ void mac(uint64_t c, uint64_t b0, uint64_t b1) {
        const uint32_t c0 = (uint32_t)c, c1 = (uint32_t)(c >> 32);
        asm("v_mad_u64_u32 %[a0], vcc, %[c0], %[p0], %[a0]\n\t"
            "v_addc_co_u32 %[k0], vcc, 0, %[k0], vcc\n\t"
            "v_mad_u64_u32 %[a1], vcc, %[c0], %[p1], %[a1]\n\t"
            "v_addc_co_u32 %[k1], vcc, 0, %[k1], vcc\n\t"
            "v_mad_u64_u32 %[a2], vcc, %[c1], %[p0], %[a2]\n\t"
            "v_addc_co_u32 %[k2], vcc, 0, %[k2], vcc\n\t"
            "v_mad_u64_u32 %[a3], vcc, %[c1], %[p1], %[a3]\n\t"
            "v_addc_co_u32 %[k3], vcc, 0, %[k3], vcc\n\t"
            "v_mad_u64_u32 %[a4], vcc, %[c0], %[q0], %[a4]\n\t"
            "v_addc_co_u32 %[k4], vcc, 0, %[k4], vcc\n\t"
            "v_mad_u64_u32 %[a5], vcc, %[c0], %[q1], %[a5]\n\t"
            "v_addc_co_u32 %[k5], vcc, 0, %[k5], vcc\n\t"
            "v_mad_u64_u32 %[a6], vcc, %[c1], %[q0], %[a6]\n\t"
            "v_addc_co_u32 %[k6], vcc, 0, %[k6], vcc\n\t"
            "v_mad_u64_u32 %[a7], vcc, %[c1], %[q1], %[a7]\n\t"
            "v_addc_co_u32 %[k7], vcc, 0, %[k7], vcc"
            : [a0] "+v"(a[0]), [a1] "+v"(a[1]), [a2] "+v"(a[2]), [a3] "+v"(a[3]), [a4] "+v"(a[4]), [a5] "+v"(a[5]),
              [a6] "+v"(a[6]), [a7] "+v"(a[7]), [k0] "+v"(kc[0]), [k1] "+v"(kc[1]), [k2] "+v"(kc[2]), [k3] "+v"(kc[3]),
              [k4] "+v"(kc[4]), [k5] "+v"(kc[5]), [k6] "+v"(kc[6]), [k7] "+v"(kc[7])
            : [c0] "v"(c0), [c1] "v"(c1), [p0] "s"((uint32_t)b0), [p1] "s"((uint32_t)(b0 >> 32)), [q0] "s"((uint32_t)b1),
              [q1] "s"((uint32_t)(b1 >> 32))
            : "vcc");
    }
    __device__ __forceinline__ void emit(uint64_t c) { emit_at(k++, c); }
    // sum for challenge ch: A0 + (A1 + A2) 2^32 + A3 2^64 + K0 2^64 + (K1 + K2) 2^96 + K3 2^128 (mod p)
    __device__ __forceinline__ uint64_t finish(int ch) const { return gl::add(fold_columns(a + 4 * ch, kc + 4 * ch), base[ch]); }
    // The four 64-bit columns and their carry counts as ONE 160-bit integer (t4 : t3 : t2 : t1 : t0), reduced with
    // 2^64 = 2^32 - 1, 2^96 = -1, 2^128 = -2^32: (t1:t0) + t2 EPS - t3 - t4 2^32.  ~35 instructions (the first version
    // reduced every column on its own: ~110, twice per item of k_quotient).
    static __device__ __forceinline__ uint64_t fold_columns(const uint64_t* A, const uint32_t* K) {
        uint32_t t1, t2, t3, t4, m0, m1, cm;
        asm("v_add_co_u32 %[m0], vcc, %[a1l], %[a2l]\n\t"
            "v_addc_co_u32 %[m1], vcc, %[a1h], %[a2h], vcc\n\t"
            "v_addc_co_u32 %[cm], vcc, 0, 0, vcc\n\t"
            "v_add_co_u32 %[t1], vcc, %[a0h], %[m0]\n\t"
            "v_addc_co_u32 %[t2], vcc, %[m1], %[a3l], vcc\n\t"
            "v_addc_co_u32 %[t3], vcc, %[cm], %[a3h], vcc\n\t"
            "v_addc_co_u32 %[t4], vcc, 0, %[k3], vcc\n\t"
            "v_add_co_u32 %[t2], vcc, %[t2], %[k0]\n\t"
            "v_addc_co_u32 %[t3], vcc, %[t3], %[k1], vcc\n\t"
            "v_addc_co_u32 %[t4], vcc, 0, %[t4], vcc\n\t"
            "v_add_co_u32 %[t3], vcc, %[t3], %[k2]\n\t"
            "v_addc_co_u32 %[t4], vcc, 0, %[t4], vcc"
            : [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [m0] "=&v"(m0), [m1] "=&v"(m1), [cm] "=&v"(cm)
            : [a0h] "v"((uint32_t)(A[0] >> 32)), [a1l] "v"((uint32_t)A[1]), [a1h] "v"((uint32_t)(A[1] >> 32)), [a2l] "v"((uint32_t)A[2]),
              [a2h] "v"((uint32_t)(A[2] >> 32)), [a3l] "v"((uint32_t)A[3]), [a3h] "v"((uint32_t)(A[3] >> 32)), [k0] "v"(K[0]),
              [k1] "v"(K[1]), [k2] "v"(K[2]), [k3] "v"(K[3])
            : "vcc");
        const uint64_t r = gl::canon(gl32::to_u64(gl32::reduce128((uint32_t)A[0], t1, t2, t3)));
        return gl::sub(r, (uint64_t)t4 << 32);
    }
};

// sum_i x_i c_i for wave-uniform constants c_i, as ONE running 160-bit sum (GateAcc's columns for a single sum): eight
// instructions per product and one reduction, where a reduced multiply-add costs 24
struct DotAcc {
    uint64_t a[4];
    uint32_t kc[4];
    __device__ __forceinline__ void reset() {
#pragma unroll
        for (int i = 0; i < 4; i++) { a[i] = 0; kc[i] = 0; }
    }
    __device__ __forceinline__ void mac(uint64_t x, uint64_t c) {   // x: any u64, c: wave-uniform
        const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32);
        asm("v_mad_u64_u32 %[a0], vcc, %[x0], %[c0], %[a0]\n\t"
            "v_addc_co_u32 %[k0], vcc, 0, %[k0], vcc\n\t"
            "v_mad_u64_u32 %[a1], vcc, %[x0], %[c1], %[a1]\n\t"
            "v_addc_co_u32 %[k1], vcc, 0, %[k1], vcc\n\t"
            "v_mad_u64_u32 %[a2], vcc, %[x1], %[c0], %[a2]\n\t"
            "v_addc_co_u32 %[k2], vcc, 0, %[k2], vcc\n\t"
            "v_mad_u64_u32 %[a3], vcc, %[x1], %[c1], %[a3]\n\t"
            "v_addc_co_u32 %[k3], vcc, 0, %[k3], vcc"
            : [a0] "+v"(a[0]), [a1] "+v"(a[1]), [a2] "+v"(a[2]), [a3] "+v"(a[3]), [k0] "+v"(kc[0]), [k1] "+v"(kc[1]),
              [k2] "+v"(kc[2]), [k3] "+v"(kc[3])
            : [x0] "v"(x0), [x1] "v"(x1), [c0] "s"((uint32_t)c), [c1] "s"((uint32_t)(c >> 32))
            : "vcc");
    }
    __device__ __forceinline__ uint64_t value() const { return GateAcc::fold_columns(a, kc); }   // canonical
};

// sum_j x_j 2^(s_j) kept as a 128-bit integer and reduced once (Horner recombination of range-check limbs
// without a multiplication per limb); the caller keeps the total below 2^128
struct Sum128 {
    uint64_t lo = 0, hi = 0;
    __device__ __forceinline__ void add(uint64_t x, uint32_t sh) {
        const uint64_t tl = x << sh, th = sh ? x >> (64 - sh) : 0;
        lo += tl;
        hi += th + (lo < tl ? 1u : 0u);
    }
    __device__ __forceinline__ uint64_t value() const { return gl::reduce128(lo, hi); }
};

// prod_{x < 4} (l - x): the 2-bit limb range check of the u32 gates, as u (u + 2) with u = l (l - 3)
// (l (l - 3) = l^2 - 3l and (l - 1)(l - 2) = l^2 - 3l + 2): two multiplications instead of three
// (canonical l in, LOOSE result out - any u64 congruent to the product: GateAcc::emit and gl::mul_loose take loose values,
// so the canonicalising compare / select after each multiplication is skipped)
__device__ __forceinline__ uint64_t limb4(uint64_t l) {
    const uint64_t u = gl::mul_loose(l, gl::sub(l, 3));
    return gl::mul_loose(u, gl::add_loose(u, 2));
}

// x^7, any u64 in, LOOSE out (consumers: poseidon::mds_layer, which splits any u64 into halves, and gl::add_loose)
__device__ __forceinline__ uint64_t sbox7c(uint64_t x) {
    uint64_t x2 = gl::mul_loose(x, x), x4 = gl::mul_loose(x2, x2), x3 = gl::mul_loose(x, x2);
    const uint64_t r = gl::mul_loose(x3, x4);
    // one S-box at a time: a wave issues an instruction every four cycles whether or not it depends on the one before, so
    // twelve interleaved S-boxes buy nothing and their temporaries were what k_quotient spilled (20 - 70 registers per part)
    __builtin_amdgcn_sched_barrier(0);
    return r;
}
// The permutation's linear layer on any-u64 inputs, canonical outputs: 32-bit halves, twelve multiply-accumulates per half
// and output, the four-instruction fold of gl32.hpp - one output row at a time (a fence per row: see sbox7c)
__device__ __forceinline__ void mds_canon(uint64_t (&s)[12]) {
    constexpr uint32_t C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    uint64_t o[12];
#pragma unroll
    for (int r = 0; r < 12; r++) {
        uint64_t al = 0, ah = 0;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            al += (uint64_t)(uint32_t)s[(i + r) % 12] * C[i];
            ah += (uint64_t)(uint32_t)(s[(i + r) % 12] >> 32) * C[i];
        }
        if (r == 0) {
            al += (uint64_t)(uint32_t)s[0] * 8u;
            ah += (uint64_t)(uint32_t)(s[0] >> 32) * 8u;
        }
        o[r] = gl::canon(gl32::to_u64(gl32::fold_acc(al, ah)));
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = o[i];
}

// PoseidonGate::eval_unfiltered_base, in three independent parts (bit mask): every round of the gate restarts from WIRES (the
// round's S-box inputs are wires, the constraint ties them to the state computed from the previous round's wires), so the 123
// constraints split wherever the state is reloaded:
//   1: swap bit, deltas, full rounds 0-2 and the check of round 3's inputs            constraints 0 .. 40
//   2: full round 3 from its input wires, the 22 partial rounds, the check of round 26's inputs      41 .. 74
//   4: full rounds 26-29 and the output wires                                                        75 .. 122
// Each part is its own work item of k_quotient; constraint k always meets alpha^k (GateAcc::emit_at).
//
// Upstream evaluates the partial rounds in its fast formulation (sparse matrices, constants pushed onto element 0).  With the
// S-box inputs FREE (they are wires here) the constraints `computed input - wire` and the twelve values after round 25 are the
// same polynomials in the naive formulation - checked numerically against tools/gen_poseidon_fast.py's tables, and by every
// proof-byte comparison with the oracle, which evaluates the fast formulation - so the whole gate runs on the permutation's
// own device schedule (round 4): linear layers on the matrix cores with the next round's constants in the recombination
// (gl32::mds_layer_mfma), rounds 4 .. 24 as seven fused blocks whose S-box hooks emit `computed - wire` and continue from the
// wire (gl32::partial_block3).  NLX_POSEIDON_GATE_FAST_BASIS keeps round 3's evaluation (vector-pipe layers, fast formulation).
template <class WireFn>
struct GateBlockSboxes {
    GateAcc& acc;
    WireFn W;
    uint32_t r0;   // the block's first partial round (0 .. 18): its S-box input is wire 65 + r0, its constraint 41 + r0
    __device__ __forceinline__ gl32::F at(uint32_t r, gl32::F computed) const {
        const uint64_t in = W(65 + r);
        acc.emit_at(41 + r, gl::sub(gl32::to_u64(computed), in));
        const gl32::F u = gl32::sbox7(gl32::from_u64(in));
        __builtin_amdgcn_sched_barrier(0);
        return u;
    }
    __device__ __forceinline__ gl32::F first(gl32::F s0) const { return at(r0, s0); }
    __device__ __forceinline__ gl32::F inner(int i, gl32::F w) const { return at(r0 + (uint32_t)i, w); }
    __device__ __forceinline__ void before_matrix_pass() const { acc.stash(); }
};

template <class WireFn>
__device__ __forceinline__ void gate_poseidon_mx(WireFn W, GateAcc& acc, uint32_t parts) {
    const uint64_t* RC = poseidon::RC_DEV;
    const uint64_t* rcb = poseidon::LAYER_RCB_DEV;   // [l * 24 ..]: the constants round l adds, as the recombination's seeds
    // The matrix operands depend on the lane only: left to itself hipcc hoists them out of k_quotient's item loop - ~32 registers
    // pinned under EVERY gate (the kernel's empty-list time went 0.23 -> 0.38 ms per 2^19 points, every item a few percent
    // slower).  The lane index is tied to the item here (an empty asm with the part mask as input), so they are built per item.
    uint32_t lane = threadIdx.x & 63;
    asm("" : "+v"(lane) : "s"(parts));
    const gl32::i32x4_t afrag = gl32::mds_a_fragment(lane);
    gl32::F t[12];
    // S-boxes one at a time (a fence each): interleaved they buy nothing and their temporaries were what k_quotient spilled
    auto sbox_of_wires = [&](uint32_t w0) {
#pragma unroll
        for (int i = 0; i < 12; i++) {
            t[i] = gl32::sbox7(gl32::from_u64(W(w0 + i)));
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto check_against_wires = [&](uint32_t c0, uint32_t w0) {   // constraints c0 + i: computed state - wire w0 + i
#pragma unroll
        for (int i = 0; i < 12; i++) {
            acc.emit_at(c0 + i, gl::sub(gl32::to_u64(t[i]), W(w0 + i)));
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if (parts & 1u) {
        const uint64_t swap = W(24);
        acc.emit_at(0, gl::mul(swap, gl::sub(swap, 1)));
        uint64_t st[12];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint64_t lhs = W(i), rhs = W(i + 4), delta = W(25 + i);
            acc.emit_at(1 + i, gl::sub(gl::mul(swap, gl::sub(rhs, lhs)), delta));
            st[i] = gl::add(lhs, delta);
            st[i + 4] = gl::sub(rhs, delta);
        }
#pragma unroll
        for (int i = 8; i < 12; i++) st[i] = W(i);
        // rounds 0 .. 2 in full, each followed by the check of the next round's S-box inputs (wires), from which the next
        // round restarts; round 3 itself belongs to part 2
#pragma unroll
        for (int i = 0; i < 12; i++) {
            t[i] = gl32::sbox7(gl32::from_u64(gl::add_loose(st[i], RC[i])));
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll 1
        for (int r = 0; r < 3; r++) {
            if (r) sbox_of_wires(29 + 12 * (r - 1));
            gl32::mds_layer_mfma<2>(t, afrag, rcb + (r + 1) * 24);
            check_against_wires(5 + 12 * r, 29 + 12 * r);
        }
    }
    if (parts & 2u) {
        const gl32::BlockOperands op = poseidon::block_operands(lane);
        sbox_of_wires(29 + 24);                              // round 3's S-box inputs are wires
        gl32::mds_layer_mfma<2>(t, afrag, rcb + 4 * 24);
#pragma unroll 1
        for (uint32_t b = 0; b < NLX_POSEIDON_N_BLOCKS; b++)   // partial rounds 0 .. 20 (rounds 4 .. 24)
            gl32::partial_block3<NLX_POSEIDON_BLOCK_GAMMA21>(t, op, poseidon::BLOCK_KAPPA_DEV + b * 6, GateBlockSboxes<WireFn>{acc, W, 3 * b});
        t[0] = GateBlockSboxes<WireFn>{acc, W, 21}.first(t[0]);   // partial round 21 (round 25)
        gl32::mds_layer_mfma<2>(t, afrag, rcb + 26 * 24);
        check_against_wires(63, 87);
    }
    if (parts & 4u) {
#pragma unroll 1
        for (int r = 0; r < 4; r++) {
            sbox_of_wires(87 + 12 * r);
            if (r < 3) {
                gl32::mds_layer_mfma<2>(t, afrag, rcb + (27 + r) * 24);
                check_against_wires(63 + 12 * (r + 1), 87 + 12 * (r + 1));
            } else {
                gl32::mds_layer_mfma<0>(t, afrag, rcb);
                check_against_wires(111, 12);
            }
        }
    }
}

// round 3's evaluation (vector-pipe layers, upstream's fast formulation of the partial rounds): kept for the parts NLX_PGATE_MX
// leaves to it, and as the ablation reference
template <class WireFn>
__device__ __forceinline__ void gate_poseidon_fast(WireFn W, GateAcc& acc, uint32_t parts) {
    const uint64_t* RC = poseidon::RC_DEV;
    uint64_t st[12];
    if (parts & 1u) {
        const uint64_t swap = W(24);
        acc.emit_at(0, gl::mul(swap, gl::sub(swap, 1)));
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint64_t lhs = W(i), rhs = W(i + 4), delta = W(25 + i);
            acc.emit_at(1 + i, gl::sub(gl::mul(swap, gl::sub(rhs, lhs)), delta));
            st[i] = gl::add(lhs, delta);
            st[i + 4] = gl::sub(rhs, delta);
        }
#pragma unroll
        for (int i = 8; i < 12; i++) st[i] = W(i);
        // rounds 0 .. 2 in full, each followed by the check of the next round's S-box inputs (wires), from which the next
        // round restarts; round 3 itself belongs to part 2
#pragma unroll
        for (int i = 0; i < 12; i++) st[i] = gl::add_loose(st[i], RC[i]);
#pragma unroll 1
        for (int r = 0; r < 3; r++) {
#pragma unroll
            for (int i = 0; i < 12; i++) st[i] = sbox7c(st[i]);
            mds_canon(st);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 12; i++) {
                const uint64_t in = W(29 + 12 * r + i);
                acc.emit_at(5 + 12 * r + i, gl::sub(gl::add(st[i], RC[(r + 1) * 12 + i]), in));
                st[i] = in;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (parts & 2u) {
#pragma unroll
        for (int i = 0; i < 12; i++) st[i] = sbox7c(W(29 + 24 + i));   // round 3's S-box inputs are wires
        mds_canon(st);
#pragma unroll
        for (int i = 0; i < 12; i++) st[i] = gl::add(st[i], FAST_FIRST[i]);
        {
            // st[1..11] <- M^T st[1..11] (mds_partial_layer_init): each output column is ONE running 160-bit sum over the
            // eleven products by wave-uniform constants, reduced once
            uint64_t res[12];
            res[0] = st[0];
#pragma unroll
            for (int c = 1; c < 12; c++) {
                DotAcc dot;
                dot.reset();
#pragma unroll
                for (int r = 1; r < 12; r++) dot.mac(st[r], FAST_INIT[r - 1][c - 1]);
                res[c] = dot.value();
                __builtin_amdgcn_sched_barrier(0);   // one column at a time (see sbox7c)
            }
#pragma unroll
            for (int i = 0; i < 12; i++) st[i] = res[i];
        }
#pragma unroll 1
        for (int r = 0; r < 22; r++) {
            const uint64_t in = W(65 + r);
            acc.emit_at(41 + r, gl::sub(st[0], in));
            uint64_t s0 = sbox7c(in);
            if (r < 21) s0 = gl::add_loose(s0, FAST_RC[r]);
            // new st[0] = 25 s0 + sum_i st[i] w_hat[i]: one running sum, one reduction; st[i] += s0 v[i]
            DotAcc dot;
            dot.reset();
            dot.mac(s0, 25);
#pragma unroll
            for (int i = 1; i < 12; i++) {
                dot.mac(st[i], FAST_W[r][i - 1]);
                st[i] = gl::add(st[i], gl::mul(s0, FAST_VS[r][i - 1]));
                __builtin_amdgcn_sched_barrier(0);
            }
            st[0] = dot.value();
        }
#pragma unroll
        for (int i = 0; i < 12; i++)
            acc.emit_at(63 + i, gl::sub(gl::add(st[i], RC[26 * 12 + i]), W(87 + i)));
    }
    if (parts & 4u) {
#pragma unroll 1
        for (int r = 0; r < 4; r++) {
            // the check of this round's S-box inputs against the state the previous round left, THEN the round from its
            // input wires (read a second time from LDS): the old state is dead before the new one is built
            if (r != 0) {
#pragma unroll
                for (int i = 0; i < 12; i++) {
                    acc.emit_at(63 + 12 * r + i, gl::sub(gl::add(st[i], RC[(26 + r) * 12 + i]), W(87 + 12 * r + i)));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int i = 0; i < 12; i++) st[i] = sbox7c(W(87 + 12 * r + i));
            mds_canon(st);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 12; i++) acc.emit_at(111 + i, gl::sub(st[i], W(12 + i)));
    }
}

// Inside k_quotient the gate's parts run in round 3's form.  The matrix-core schedule was tried there and lost (round 4): parts 1
// and 3 - seven full rounds - measured 440 -> 720 and 610 -> 1 040 us per 2^19 points in the calibration build
// (profiles/r04_quotient_calibrate_poseidon_gate_mx7_v1.txt); part 2 measured 1 490 -> 990 there but made the KERNEL slower, 6.8 ->
// 8.6 ms at 2^18 rows: the block needs ~122 registers beside GateAcc's 28 at this kernel's 128, the handful of spilled registers
// it reloads per block are scratch round trips, and in k_quotient a tile's eight waves run different items, so one wave's
// memory latency is the tile's critical path (with every wave on the same item the latency hides).  Part 2 therefore has a
// kernel of its own, k_quotient_poseidon below, where every wave runs it and 168 registers are allowed; NLX_PGATE_MX selects
// parts for the in-kernel experiment.
#ifndef NLX_PGATE_MX
#define NLX_PGATE_MX 0u
#endif
template <class WireFn>
__device__ __forceinline__ void gate_poseidon(WireFn W, GateAcc& acc, uint32_t parts) {
    if (parts & NLX_PGATE_MX) gate_poseidon_mx(W, acc, parts & NLX_PGATE_MX);
    if (parts & ~NLX_PGATE_MX) gate_poseidon_fast(W, acc, parts & ~NLX_PGATE_MX);
}

// One gate's unfiltered constraints at this lane's point, folded into `acc` with the alpha powers (Gate::eval_unfiltered_base).
// W(c): wire c of the point; CC(c): constants column c (selectors first) of the point; n = 2^log_n.
template <class WireFn, class ConstFn>
__device__ __forceinline__ void eval_gate(const GateDev& gd, const QuotientParams& p, WireFn W, ConstFn CC, GateAcc& acc, size_t n,
                                          uint32_t parts) {
    switch (gd.kind) {
        case NLX_GATE_CONSTANT:
            for (uint32_t i = 0; i < gd.param0; i++) acc.emit(gl::sub(CC(p.gate_const0 + i), W(i)));
            break;
        case NLX_GATE_PUBLIC_INPUT:
            for (uint32_t i = 0; i < 4; i++) acc.emit(gl::sub(W(i), p.pih[i]));
            break;
        case NLX_GATE_ARITHMETIC: {
            const uint64_t c0 = CC(p.gate_const0), c1 = CC(p.gate_const0 + 1);
            for (uint32_t i = 0; i < gd.param0; i++) {
                const uint64_t m0 = W(4 * i), m1 = W(4 * i + 1), ad = W(4 * i + 2), o = W(4 * i + 3);
                acc.emit(gl::sub(o, gl::add(gl::mul(gl::mul(m0, m1), c0), gl::mul(ad, c1))));
            }
            break;
        }
        case NLX_GATE_BASE_SUM: {
            const uint32_t B = gd.param0, nl = gd.param1;
            uint64_t sum = 0;
            for (uint32_t i = nl; i-- > 0;) sum = gl::add(gl::mul(sum, (uint64_t)B), W(1 + i));
            acc.emit(gl::sub(sum, W(0)));
            for (uint32_t i = 0; i < nl; i++) {
                const uint64_t limb = W(1 + i);
                uint64_t prod = 1;
                for (uint32_t t = 0; t < B; t++) prod = gl::mul(prod, gl::sub(limb, (uint64_t)t));
                acc.emit(prod);
            }
            break;
        }
        case NLX_GATE_POSEIDON:
            gate_poseidon(W, acc, parts);
            break;
        case NLX_GATE_ARITHMETIC_EXT: {
            const uint64_t c0 = CC(p.gate_const0), c1 = CC(p.gate_const0 + 1);
            for (uint32_t i = 0; i < gd.param0; i++) {
                const gl::Ext m0{W(8 * i), W(8 * i + 1)}, m1{W(8 * i + 2), W(8 * i + 3)};
                const gl::Ext pr = gl::mul(m0, m1);
                acc.emit(gl::sub(W(8 * i + 6), gl::add(gl::mul(pr.a, c0), gl::mul(W(8 * i + 4), c1))));
                acc.emit(gl::sub(W(8 * i + 7), gl::add(gl::mul(pr.b, c0), gl::mul(W(8 * i + 5), c1))));
            }
            break;
        }
        case NLX_GATE_MUL_EXT: {
            const uint64_t c0 = CC(p.gate_const0);
            for (uint32_t i = 0; i < gd.param0; i++) {
                const gl::Ext m0{W(6 * i), W(6 * i + 1)}, m1{W(6 * i + 2), W(6 * i + 3)};
                const gl::Ext pr = gl::mul(m0, m1);
                acc.emit(gl::sub(W(6 * i + 4), gl::mul(pr.a, c0)));
                acc.emit(gl::sub(W(6 * i + 5), gl::mul(pr.b, c0)));
            }
            break;
        }
        case NLX_GATE_REDUCING:
        case NLX_GATE_REDUCING_EXT: {
            const uint32_t nco = gd.param0;
            const bool ext = gd.kind == NLX_GATE_REDUCING_EXT;
            const uint32_t start_coeffs = 6, start_accs = start_coeffs + (ext ? 2 * nco : nco);
            const gl::Ext alpha{W(2), W(3)};
            gl::Ext a{W(4), W(5)};
            for (uint32_t i = 0; i < nco; i++) {
                const uint32_t aw = (i == nco - 1) ? 0 : start_accs + 2 * i;  // last accumulator = output wires
                const gl::Ext nxt{W(aw), W(aw + 1)};
                const gl::Ext pr = gl::mul(a, alpha);
                if (ext) {
                    acc.emit(gl::sub(gl::add(pr.a, W(start_coeffs + 2 * i)), nxt.a));
                    acc.emit(gl::sub(gl::add(pr.b, W(start_coeffs + 2 * i + 1)), nxt.b));
                } else {
                    acc.emit(gl::sub(gl::add(pr.a, W(start_coeffs + i)), nxt.a));
                    acc.emit(gl::sub(pr.b, nxt.b));
                }
                a = nxt;
            }
            break;
        }
        case NLX_GATE_POSEIDON_MDS: {
            // outputs = MDS * inputs on both components of the extension algebra: the linear layer of the
            // permutation itself (32-bit halves, multiply-accumulate, one reduction per output)
#pragma unroll 1
            for (uint32_t comp = 0; comp < 2; comp++) {   // one component at a time: both at once need 114 registers
                uint64_t st[12];
#pragma unroll
                for (int i = 0; i < 12; i++) st[i] = W(2 * i + comp);
                mds_canon(st);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rr = 0; rr < 12; rr++) {
                    acc.emit_at(2 * rr + comp, gl::sub(W(24 + 2 * rr + comp), st[rr]));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            acc.k += 24;
            break;
        }
        case NLX_GATE_EXPONENTIATION: {
            const uint32_t nb = gd.param0;
            const uint64_t base = W(0);
            uint64_t prev_iv = 1;
            for (uint32_t i = 0; i < nb; i++) {
                const uint64_t prev = i ? gl::mul(prev_iv, prev_iv) : 1;
                const uint64_t bit = W(1 + (nb - 1 - i));
                const uint64_t sel = gl::add(gl::mul(bit, base), gl::sub(1, bit));
                const uint64_t iv = W(2 + nb + i);
                acc.emit(gl::sub(gl::mul(prev, sel), iv));
                prev_iv = iv;
            }
            acc.emit(gl::sub(W(1 + nb), prev_iv));
            break;
        }
        case NLX_GATE_U32_ADD_MANY: {
            const uint32_t na = gd.param0, nops = gd.param1, nl = 18, nrl = 16;
            for (uint32_t i = 0; i < nops; i++) {
                const uint32_t b0 = (na + 3) * i, lb = (na + 3) * nops + nl * i;
                uint64_t computed = W(b0 + na);
                for (uint32_t j = 0; j < na; j++) computed = gl::add(computed, W(b0 + j));
                const uint64_t res = W(b0 + na + 1), cy = W(b0 + na + 2);
                acc.emit(gl::sub(gl::add(gl::mul(cy, 1ULL << 32), res), computed));
                Sum128 cr, cc;
                for (uint32_t j = nl; j-- > 0;) {
                    const uint64_t l = W(lb + j);
                    acc.emit(limb4(l));
                    if (j < nrl) cr.add(l, 2 * j);
                    else cc.add(l, 2 * (j - nrl));
                }
                acc.emit(gl::sub(cr.value(), res));
                acc.emit(gl::sub(cc.value(), cy));
            }
            break;
        }
        case NLX_GATE_U32_ARITHMETIC: {
            const uint32_t nops = gd.param0;
            for (uint32_t i = 0; i < nops; i++) {
                const uint32_t b0 = 6 * i, lb = 6 * nops + 32 * i;
                const uint64_t computed = gl::add(gl::mul(W(b0), W(b0 + 1)), W(b0 + 2));
                const uint64_t lo = W(b0 + 3), hi = W(b0 + 4), inv = W(b0 + 5);
                const uint64_t hi_not_max = gl::sub(gl::mul(inv, gl::sub(0xFFFFFFFFULL, hi)), 1);
                acc.emit(gl::mul(hi_not_max, lo));
                acc.emit(gl::sub(gl::add(gl::mul(hi, 1ULL << 32), lo), computed));
                Sum128 cl, ch;
                for (uint32_t j = 32; j-- > 0;) {
                    const uint64_t l = W(lb + j);
                    acc.emit(limb4(l));
                    if (j < 16) cl.add(l, 2 * j);
                    else ch.add(l, 2 * (j - 16));
                }
                acc.emit(gl::sub(cl.value(), lo));
                acc.emit(gl::sub(ch.value(), hi));
            }
            break;
        }
        case NLX_GATE_U32_SUBTRACTION: {
            const uint32_t nops = gd.param0;
            for (uint32_t i = 0; i < nops; i++) {
                const uint32_t b0 = 5 * i, lb = 5 * nops + 16 * i;
                const uint64_t initial = gl::sub(gl::sub(W(b0), W(b0 + 1)), W(b0 + 2));
                const uint64_t res = W(b0 + 3), bo = W(b0 + 4);
                acc.emit(gl::sub(res, gl::add(initial, gl::mul(bo, 1ULL << 32))));
                Sum128 c;
                for (uint32_t j = 16; j-- > 0;) {
                    const uint64_t l = W(lb + j);
                    acc.emit(limb4(l));
                    c.add(l, 2 * j);
                }
                acc.emit(gl::sub(c.value(), res));
                acc.emit(gl::mul(bo, gl::sub(1, bo)));
            }
            break;
        }
        case NLX_GATE_U32_RANGE_CHECK: {
            const uint32_t nin = gd.param0;
            for (uint32_t i = 0; i < nin; i++) {
                const uint32_t ab = nin + 16 * i;
                Sum128 sum;
                for (uint32_t j = 0; j < 16; j++) sum.add(W(ab + j), 2 * j);
                acc.emit(gl::sub(sum.value(), W(i)));
                for (uint32_t j = 0; j < 16; j++) acc.emit(limb4(W(ab + j)));
            }
            break;
        }
        case NLX_GATE_COMPARISON: {
            const uint32_t nbits = gd.param0, nch = gd.param1, cb = (nbits + nch - 1) / nch;
            const uint32_t fc = 4, sc = 4 + nch, eqd = 4 + 2 * nch, ceq = 4 + 3 * nch, iv = 4 + 4 * nch, msb = 4 + 5 * nch;
            Sum128 fcomb, scomb;  // nch * cb = num_bits <= 62 (checked at circuit build)
            for (uint32_t i = 0; i < nch; i++) {
                fcomb.add(W(fc + i), cb * i);
                scomb.add(W(sc + i), cb * i);
            }
            acc.emit(gl::sub(fcomb.value(), W(0)));
            acc.emit(gl::sub(scomb.value(), W(1)));
            uint64_t msd = 0;
            for (uint32_t i = 0; i < nch; i++) {
                const uint64_t f = W(fc + i), s2 = W(sc + i);
                uint64_t p1, p2;
                if (cb == 2) {
                    p1 = limb4(f);
                    p2 = limb4(s2);
                } else {
                    p1 = f;
                    p2 = s2;
                    for (uint32_t x2 = 1; x2 < (1u << cb); x2++) {
                        p1 = gl::mul(p1, gl::sub(f, (uint64_t)x2));
                        p2 = gl::mul(p2, gl::sub(s2, (uint64_t)x2));
                    }
                }
                acc.emit(p1);
                acc.emit(p2);
                const uint64_t diff = gl::sub(s2, f), e = W(ceq + i), ivv = W(iv + i);
                acc.emit(gl::sub(gl::mul(diff, W(eqd + i)), gl::sub(1, e)));
                acc.emit(gl::mul(e, diff));
                acc.emit(gl::sub(ivv, gl::mul(e, msd)));
                msd = gl::add(ivv, gl::mul(gl::sub(1, e), diff));
            }
            acc.emit(gl::sub(W(3), msd));
            uint64_t bc = 0;
            for (uint32_t b = 0; b <= cb; b++) {
                const uint64_t bit = W(msb + b);
                acc.emit(gl::mul(bit, gl::sub(1, bit)));
            }
            for (uint32_t b = cb + 1; b-- > 0;) bc = gl::add(gl::add(bc, bc), W(msb + b));
            acc.emit(gl::sub(gl::add(1ULL << cb, W(3)), bc));
            acc.emit(gl::sub(W(2), W(msb + cb)));
            break;
        }
        case NLX_GATE_COSET_INTERPOLATION: {
            const uint32_t bits = gd.param0, deg = gd.param1, np = 1u << bits;
            const uint32_t sep = 1 + 2 * np, sev = sep + 2, si = sev + 2, ni = (np - 2) / (deg - 1), ssh = si + 4 * ni;
            const uint64_t shift = W(0);
            const gl::Ext pt{W(ssh), W(ssh + 1)};
            acc.emit(gl::sub(W(sep), gl::mul(pt.a, shift)));
            acc.emit(gl::sub(W(sep + 1), gl::mul(pt.b, shift)));
            // domain x_j = w_np^j comes from the w_n table; on a multiplicative subgroup the barycentric
            // weight 1 / prod_{i != j} (x_j - x_i) is x_j / np
            uint64_t np_inv = 1;
            for (uint32_t i = 0; i < bits; i++) np_inv = gl::mul(np_inv, 0x7FFFFFFF80000001ULL);  // 2^-1
            const uint32_t stride = (uint32_t)(n >> bits), half = (uint32_t)(n >> 1);
            gl::Ext ev{0, 0}, pr{1, 0};
            for (uint32_t c = 0; c <= ni; c++) {
                const uint32_t start = c == 0 ? 0 : 1 + (deg - 1) * c;
                uint32_t end = c == 0 ? deg : start + deg - 1;
                end = end > np ? np : end;
                for (uint32_t j = start; j < end; j++) {
                    const uint64_t xj = root_pow(p.w_n_table, j * stride, half);
                    const gl::Ext term{gl::sub(pt.a, xj), pt.b};
                    const gl::Ext vp = gl::mul(gl::Ext{W(1 + 2 * j), W(2 + 2 * j)}, pr);
                    ev = gl::add(gl::mul(ev, term), gl::mul(vp, gl::mul(xj, np_inv)));
                    pr = gl::mul(pr, term);
                }
                if (c < ni) {
                    const gl::Ext ie{W(si + 2 * c), W(si + 2 * c + 1)}, ip{W(si + 2 * (ni + c)), W(si + 2 * (ni + c) + 1)};
                    acc.emit(gl::sub(ie.a, ev.a));
                    acc.emit(gl::sub(ie.b, ev.b));
                    acc.emit(gl::sub(ip.a, pr.a));
                    acc.emit(gl::sub(ip.b, pr.b));
                    ev = ie;
                    pr = ip;
                }
            }
            acc.emit(gl::sub(W(sev), ev.a));
            acc.emit(gl::sub(W(sev + 1), ev.b));
            break;
        }
        case NLX_GATE_RANDOM_ACCESS: {
            const uint32_t bits = gd.param0, copies = gd.param1 & 0xFFFF, extra = gd.param1 >> 16;
            const uint32_t vec = 1u << bits, rt = (2 + vec) * copies + extra;
            for (uint32_t cpy = 0; cpy < copies; cpy++) {
                const uint32_t b0 = (2 + vec) * cpy, bw = rt + cpy * bits;
                for (uint32_t i = 0; i < bits; i++) {
                    const uint64_t b = W(bw + i);
                    acc.emit(gl::mul(b, gl::sub(b, 1)));
                }
                uint64_t rec = 0;
                for (uint32_t i = bits; i-- > 0;) rec = gl::add(gl::add(rec, rec), W(bw + i));
                acc.emit(gl::sub(rec, W(b0)));
                // fold the list by the index bits; level i consumes bit i.  The selected element is
                // computed by a recursive descent so that no list array lives in registers:
                // value(level, j) = value(level-1, 2j) + bit * (value(level-1, 2j+1) - value(level-1, 2j))
                // evaluated iteratively over the 2^bits leaves with a small stack.
                uint64_t stack[7];
                uint32_t depth_of[7];
                int sp = 0;
                for (uint32_t leaf = 0; leaf < vec; leaf++) {
                    uint64_t v = W(b0 + 2 + leaf);
                    uint32_t lvl = 0;
                    while (sp > 0 && depth_of[sp - 1] == lvl) {
                        const uint64_t left = stack[--sp];
                        const uint64_t b = W(bw + lvl);
                        v = gl::add(left, gl::mul(b, gl::sub(v, left)));
                        lvl++;
                    }
                    stack[sp] = v;
                    depth_of[sp] = lvl;
                    sp++;
                }
                acc.emit(gl::sub(stack[0], W(b0 + 1)));
            }
            for (uint32_t i = 0; i < extra; i++) acc.emit(gl::sub(CC(p.gate_const0 + i), W((2 + vec) * copies + i)));
            break;
        }
        default: break;  // NoopGate
    }
}

// k_quotient: compute_quotient_polys' inner loop (eval_vanishing_poly_base_batch) on the LDE domain.
//
// A block is QW waves that share ONE tile of 64 consecutive LDE points (same coset).  The tile's wire row and constants row
// are staged ONCE in LDS ([column][64 points], 512 contiguous bytes per column, conflict-free) - in round 1 every gate re-read
// its wires from global memory: 2 310 loads per point, 10 GB of traffic per launch against 1 GB of distinct columns (PMC).
// The work of a point - every gate of the circuit plus the permutation argument of each challenge - is split over the waves
// by a host-built, cost-balanced item list (QuotientParams::work): all gates are evaluated at every point whatever their
// selector says, so the split is the same for every tile and control flow stays wave-uniform.  The waves' partial sums meet
// in LDS.  Sigma and Z columns are used once per point and are read straight from global memory.
#ifndef NLX_QW
#define NLX_QW 8        // waves per tile: 2 tiles x 8 waves per CU = 4 waves per SIMD
#endif
#ifndef NLX_QMINW
#define NLX_QMINW 4     // minimum waves per SIMD the register allocation must allow
#endif
constexpr int QW = NLX_QW;

__global__ __launch_bounds__(64 * QW, NLX_QMINW) void k_quotient(QuotientParams p) {
    extern __shared__ uint64_t q_lds[];
    // the wave index is the same in all 64 lanes, but hipcc cannot know that of threadIdx.x >> 6: without readfirstlane the item
    // list is read with vector loads and EVERYTHING decoded from it - the gate switch, every gate's loops, every column index -
    // is compiled as divergent control flow on vector registers (120 exec-masked branches, 111 spilled registers)
    const uint32_t lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned log_L = p.log_n + p.rate_bits;
    const size_t n = (size_t)1 << p.log_n, L = (size_t)1 << log_L;
    // The lane's LDE point.  Its coordinates are needed by the tile load, by the two permutation-argument items and by the
    // epilogue; kept in registers across the whole item loop they were what the kernel still spilled at entry (17 words per
    // lane and tile = 1.2 GB per launch at 2^18 rows), so each of those places recomputes them from the lane index - made
    // opaque to the compiler there, or it would merge the copies back into one long-lived value.
    struct Point { size_t pos; uint32_t r, k; bool live; };
    auto point_of = [&](uint32_t ln) {
        const size_t pos_raw = (size_t)blockIdx.x * 64 + ln;
        Point q;
        q.live = pos_raw < L;                      // a domain of fewer than 64 points: the spare lanes redo the last point
        q.pos = q.live ? pos_raw : L - 1;
        q.r = (uint32_t)(q.pos >> p.log_n);
        q.k = (uint32_t)(q.pos & (n - 1));
        return q;
    };
    // NOT volatile: hipcc treats a volatile asm as a possible store, and every wave-uniform table read behind it (alpha
    // powers, round constants) then goes through vector memory instead of the scalar cache (k_quotient 7.3 -> 9.3 ms).  The
    // second operand ties the copy to the place it is made at (the item index), so it is neither hoisted nor merged.
    auto opaque = [](uint32_t v, uint32_t where) {
        asm("" : "+v"(v) : "s"(where));
        return v;
    };
    const size_t pos = point_of(lane).pos;
    uint64_t* lw = q_lds;                                  // [num_wires][64]
    uint64_t* lc = q_lds + (size_t)p.num_wires * 64;       // [n_consts_all][64]
    for (uint32_t c = wv; c < p.num_wires; c += QW) lw[c * 64 + lane] = p.wires[(size_t)c * L + pos];
    for (uint32_t c = wv; c < p.n_consts_all; c += QW) lc[c * 64 + lane] = p.cs[(size_t)c * L + pos];
    __syncthreads();
    auto W = [&](uint32_t c) { return lw[c * 64 + lane]; };
    auto CC = [&](uint32_t c) { return lc[c * 64 + lane]; };

    const uint32_t nc = p.nc, npp = p.npp;
    const uint32_t T0 = nc + nc * (npp + 1) + nc * p.n_lk_terms;  // first gate constraint's index in the vanishing-term list
    const uint64_t* ap0 = p.alpha_pows;
    const uint64_t* ap1 = p.alpha_pows + p.alpha_stride;
    uint64_t tot0 = 0, tot1 = 0;  // sum over all terms EXCEPT the L_0 terms (divided by Z_H later)
    uint64_t l0a = 0, l0b = 0;    // sum_i (Z_i - 1) alpha_c^i, multiplied by L_0(x)/Z_H(x) below
    GateAcc acc;
    acc.ap0 = ap0 + T0;
    acc.ap1 = ap1 + T0;
    const uint32_t* work = p.work + (size_t)wv * p.work_stride;
    for (uint32_t wi = 0;; wi++) {
        const uint32_t word = __builtin_amdgcn_readfirstlane(work[wi]);  // wave-uniform: item in the low half, for a gate evaluated in parts the part mask above it
        if (word == 0xFFFFFFFFu) break;
        const uint32_t item = word & 0xFFFFu, parts = (word >> 16) ? (word >> 16) : 7u;
        if (item < p.n_gates) {
            // ---- one gate (or one part of it): filter x sum_k alpha^k c_k ----
            const GateDev gd = p.gates[item];
            const uint64_t s = CC(gd.selector_index);
            uint64_t f = 1;
            for (uint32_t i = gd.group_start; i < gd.group_end; i++)
                if (i != gd.index) f = gl::mul(f, gl::sub((uint64_t)i, s));
            if (p.n_selectors > 1) f = gl::mul(f, gl::sub(0xFFFFFFFFULL, s));
            acc.reset();
            eval_gate(gd, p, W, CC, acc, n, parts);
            tot0 = gl::add(tot0, gl::mul(f, acc.finish(0)));
            tot1 = gl::add(tot1, gl::mul(f, acc.finish(1)));
        } else {
            // ---- the permutation argument of challenge c ----
            // vanishing_terms = [L_0 (Z_i - 1)]_i ++ [partial-product checks]_i ++ gate constraints, and EVERY
            // alpha reduces the whole list, so each term feeds both sums.
            const uint32_t c = item - p.n_gates;
            const uint32_t n_chunks = (p.routed + p.chunk - 1) / p.chunk;
            const Point pt = point_of(opaque(lane, wi));
            const size_t pos = pt.pos;
            auto ZS = [&](uint32_t col) { return p.zs[(size_t)col * L + pos]; };
            const size_t pos_next = ((size_t)pt.r << p.log_n) + ((pt.k + 1) & (n - 1));
            const uint64_t x = gl::mul(p.coset_base[pt.r], root_pow(p.w_n_table, pt.k, (uint32_t)(n >> 1)));
            const uint64_t beta = p.betas[c], gamma = p.gammas[c];
            const uint64_t bx = gl::mul(beta, x);
            const uint64_t z_x = ZS(c);
            const uint64_t z_gx = p.zs[(size_t)c * L + pos_next];
            const uint64_t zm1 = gl::sub(z_x, 1);
            l0a = gl::add(l0a, gl::mul(zm1, ap0[c]));
            l0b = gl::add(l0b, gl::mul(zm1, ap1[c]));
            uint64_t accv = z_x;
#pragma unroll 1
            for (uint32_t q = 0; q < n_chunks; q++) {
                // numerator / denominator products in loose form (any u64 congruent to the value): w + gamma is shared and
                // canonical, a product plus it needs ONE carry fix-up (gl::add_loose), and nothing here is compared
                uint64_t nm = 1, dn = 1;
                for (uint32_t j = q * p.chunk; j < (q + 1) * p.chunk && j < p.routed; j++) {
                    const uint64_t wg = gl::add(W(j), gamma);
                    const uint64_t sg = p.cs[(size_t)(p.n_consts_all + j) * L + pos];
                    nm = gl::mul_loose(nm, gl::add_loose(gl::mul_loose(bx, p.k_is[j]), wg));
                    dn = gl::mul_loose(dn, gl::add_loose(gl::mul_loose(beta, sg), wg));
                }
                const uint64_t new_acc = (q + 1 < n_chunks) ? ZS(nc + c * npp + q) : z_gx;
                const uint64_t term = gl::sub(gl::mul(accv, nm), gl::mul(new_acc, dn));
                const uint32_t t = nc + c * (npp + 1) + q;
                tot0 = gl::add(tot0, gl::mul(term, ap0[t]));
                tot1 = gl::add(tot1, gl::mul(term, ap1[t]));
                accv = new_acc;
            }
        }
    }
    // ---- the waves' partial sums meet in LDS (the tile is dead once every wave is past its last item) ----
    __syncthreads();
    uint64_t* red = q_lds;  // [QW][4][64]
    red[(wv * 4 + 0) * 64 + lane] = tot0;
    red[(wv * 4 + 1) * 64 + lane] = tot1;
    red[(wv * 4 + 2) * 64 + lane] = l0a;
    red[(wv * 4 + 3) * 64 + lane] = l0b;
    __syncthreads();
    if (wv != 0) return;
    for (uint32_t w2 = 1; w2 < QW; w2++) {
        tot0 = gl::add(tot0, red[(w2 * 4 + 0) * 64 + lane]);
        tot1 = gl::add(tot1, red[(w2 * 4 + 1) * 64 + lane]);
        l0a = gl::add(l0a, red[(w2 * 4 + 2) * 64 + lane]);
        l0b = gl::add(l0b, red[(w2 * 4 + 3) * 64 + lane]);
    }
    // quotient = (L_0 terms + rest) / Z_H(x);  L_0(x)/Z_H(x) = 1 / (n (x - 1)) = l0_scaled[pos]
    const Point pe = point_of(opaque(lane, 0xFFFFFFFFu));
    const uint64_t zh_inv = p.zh_inv[pe.r];
    const uint64_t l0s = p.l0_scaled[pe.pos];
    if (!pe.live) return;
    if (p.accumulate) {  // circuits with lookup tables: k_lookup_terms left those terms' share of the sums here
        tot0 = gl::add(tot0, p.out[pe.pos]);
        if (nc > 1) tot1 = gl::add(tot1, p.out[L + pe.pos]);
    }
    p.out[pe.pos] = gl::add(gl::mul(tot0, zh_inv), gl::mul(l0a, l0s));
    if (nc > 1) p.out[L + pe.pos] = gl::add(gl::mul(tot1, zh_inv), gl::mul(l0b, l0s));
}

size_t quotient_lds_bytes(uint32_t num_wires, uint32_t n_consts_all) {
    const size_t tile = (size_t)(num_wires + n_consts_all) * 64 * 8, red = (size_t)QW * 4 * 64 * 8;
    return tile > red ? tile : red;
}
uint32_t quotient_waves() { return QW; }

void launch_quotient(hipStream_t st, const QuotientParams& p) {
    const size_t L = (size_t)1 << (p.log_n + p.rate_bits);
    const size_t lds = quotient_lds_bytes(p.num_wires, p.n_consts_all);
    // more than the default 64 KB of dynamic LDS: the attribute is per function and device, setting it again is free
    (void)hipFuncSetAttribute((const void*)k_quotient, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_quotient, dim3((unsigned)((L + 63) / 64)), dim3(64 * QW), lds, st, p);
}

// PoseidonGate's part 2 (the 22 partial rounds: constraints 41 .. 74) for every point, one lane per point, on the permutation's
// fused-block schedule (gate_poseidon_mx): filter x sum_k alpha^k c_k for both challenges goes into (add = 0) or onto (add = 1)
// the sums k_quotient completes (QuotientParams::accumulate, as the lookup terms').  Wires and the selector are read straight
// from the LDE tables (46 columns, coalesced); every lane stays active (the matrix instructions read all 64), spare lanes redo the
// last point and store nothing.
__global__ __launch_bounds__(256, 3) void k_quotient_poseidon(QuotientParams p, uint32_t gate, uint32_t add) {
    const unsigned log_L = p.log_n + p.rate_bits;
    const size_t L = (size_t)1 << log_L;
    const size_t pos_raw = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = pos_raw < L;
    const size_t pos = live ? pos_raw : L - 1;
    auto W = [&](uint32_t c) { return p.wires[(size_t)c * L + pos]; };
    const GateDev gd = p.gates[gate];
    const uint64_t s = p.cs[(size_t)gd.selector_index * L + pos];
    uint64_t f = 1;
    for (uint32_t i = gd.group_start; i < gd.group_end; i++)
        if (i != gd.index) f = gl::mul(f, gl::sub((uint64_t)i, s));
    if (p.n_selectors > 1) f = gl::mul(f, gl::sub(0xFFFFFFFFULL, s));
    const uint32_t T0 = p.nc + p.nc * (p.npp + 1) + p.nc * p.n_lk_terms;
    GateAcc acc;
    acc.ap0 = p.alpha_pows + T0;
    acc.ap1 = p.alpha_pows + p.alpha_stride + T0;
    acc.reset();
    gate_poseidon_mx(W, acc, 2u);
    uint64_t t0 = gl::mul(f, acc.finish(0)), t1 = gl::mul(f, acc.finish(1));
    if (!live) return;
    if (add) {
        t0 = gl::add(t0, p.out[pos]);
        if (p.nc > 1) t1 = gl::add(t1, p.out[L + pos]);
    }
    p.out[pos] = t0;
    if (p.nc > 1) p.out[L + pos] = t1;
}
void launch_quotient_poseidon(hipStream_t st, const QuotientParams& p, uint32_t gate, bool add) {
    const size_t L = (size_t)1 << (p.log_n + p.rate_bits);
    hipLaunchKernelGGL(k_quotient_poseidon, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, st, p, gate, add ? 1u : 0u);
}

// l0_scaled[pos] = 1 / (n * (x_pos - 1)), batch-inverted 8 per thread
__global__ __launch_bounds__(256) void k_l0_table(uint64_t* __restrict__ out, unsigned log_n, unsigned rate_bits,
                                                  const uint64_t* __restrict__ coset_base,
                                                  const uint64_t* __restrict__ w_n_table) {
    const size_t pos = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >> (log_n + rate_bits)) return;
    const size_t n = (size_t)1 << log_n;
    const uint32_t r = (uint32_t)(pos >> log_n), k = (uint32_t)(pos & (n - 1));
    const uint64_t x = gl::mul(coset_base[r], root_pow(w_n_table, k, (uint32_t)(n >> 1)));
    out[pos] = gl::inv(gl::mul((uint64_t)(n % gl::P), gl::sub(x, 1)));
}
void launch_l0_table(hipStream_t st, uint64_t* d_out, unsigned log_n, unsigned rate_bits, const uint64_t* d_coset_base,
                     const uint64_t* d_w_n_table) {
    const size_t L = (size_t)1 << (log_n + rate_bits);
    hipLaunchKernelGGL(k_l0_table, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, st, d_out, log_n, rate_bits,
                       d_coset_base, d_w_n_table);
}

// After the per-coset inverse transforms (A_r[m] in bit-reversed m order, already multiplied by
// (g w_L^r)^-m), the 2^rate coefficient chunks are an inverse DFT across the cosets:
//   chunk_c[m] = g^(-n c) / R * sum_r w_R^(-c r) A_r[m]
// in:  [challenge][r][j]   out: [challenge * R + c][j]   (j = bit-reversed m; R = 2^rate_bits <= 8)
__global__ __launch_bounds__(256) void k_quotient_chunks(const uint64_t* __restrict__ in, uint64_t* __restrict__ out,
                                                         unsigned log_n, unsigned rate_bits,
                                                         const uint64_t* __restrict__ w_R_inv_pows,  // R entries
                                                         const uint64_t* __restrict__ chunk_scale) { // R entries
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = (size_t)1 << log_n;
    if (j >= n) return;
    const uint32_t R = 1u << rate_bits;
    const uint64_t* src = in + ((size_t)blockIdx.y << (log_n + rate_bits));
    uint64_t* dst = out + ((size_t)blockIdx.y << (log_n + rate_bits));
    uint64_t a[8];
    for (uint32_t r = 0; r < R; r++) a[r] = src[(size_t)r * n + j];
    for (uint32_t c = 0; c < R; c++) {
        uint64_t s = 0;
        for (uint32_t r = 0; r < R; r++) s = gl::add(s, gl::mul(a[r], w_R_inv_pows[(c * r) & (R - 1)]));
        dst[(size_t)c * n + j] = gl::mul(s, chunk_scale[c]);
    }
}
void launch_quotient_chunks(hipStream_t st, const uint64_t* d_in, uint64_t* d_out, unsigned log_n, unsigned rate_bits,
                            uint32_t nc, const uint64_t* d_w_R_inv_pows, const uint64_t* d_chunk_scale) {
    const size_t n = (size_t)1 << log_n;
    hipLaunchKernelGGL(k_quotient_chunks, dim3((unsigned)((n + 255) / 256), nc), dim3(256), 0, st, d_in, d_out, log_n,
                       rate_bits, d_w_R_inv_pows, d_chunk_scale);
}

// =====================================================================================
// a11: FRI
// =====================================================================================
// F(x) = alpha^nz * (sum_i alpha^i p_i(x) - C0) / (x - zeta) + (sum_{i<nz} alpha^i z_i(x) - C1) / (x - g zeta)
// over every LDE point; p_i runs over the (up to four) oracles in FRI order, z_i over the first nz columns
// of every oracle o with nz[o] > 0 (plonk_zs_next for plonky2, every committed trace column for a STARK).
// The two column sums of one point over the global column range [lo, hi): s0 = sum alpha^idx v, s1 = sum alpha^zi v
// over the columns opened at g zeta.  Each term is a base value times a wave-uniform extension constant: two
// multiply-accumulates into GateAcc's unreduced column sums (16 instructions instead of ~110 for two reduced
// products and two additions), one reduction at the end.  Where a column's two powers coincide (zi == idx: every
// trace column of a STARK) its products are accumulated once and shared by both sums.
__device__ __forceinline__ void fri_column_sums(const FriCombineParams& p, size_t pos, size_t L, uint32_t lo, uint32_t hi,
                                                gl::Ext& s0, gl::Ext& s1) {
    GateAcc both, only0, only1;
    both.reset();
    only0.reset();
    only1.reset();
    uint32_t first = 0;  // global index of the table's first column
    for (int o = 0; o < FRI_VIEWS; o++) {
        const uint64_t* tab = p.tables[o];
        const uint32_t nco = p.n_cols[o];
        const uint32_t c_lo = lo > first ? lo - first : 0, c_hi = hi > first ? (hi - first < nco ? hi - first : nco) : 0;
        const uint32_t nz = p.nz[o], zoff = p.nz_off[o];
        const bool shared = zoff == first;  // wave-uniform
        const uint64_t* col = tab + pos;
        const uint64_t* ap = p.alpha_pows + 2 * (size_t)first;
        const uint64_t* az = p.alpha_pows + 2 * (size_t)zoff;
        // columns opened at both points, then the rest; four loads in flight per lane
        const uint32_t z_hi = c_hi < nz ? c_hi : nz;
        uint32_t c = c_lo;
        if (shared) {
            for (; c + 4 <= z_hi; c += 4) {
                uint64_t v[4];
#pragma unroll
                for (int t = 0; t < 4; t++) v[t] = col[(size_t)(c + t) * L];
#pragma unroll
                for (int t = 0; t < 4; t++) both.mac(v[t], ap[2 * (c + t)], ap[2 * (c + t) + 1]);
            }
            for (; c < z_hi; c++) both.mac(col[(size_t)c * L], ap[2 * c], ap[2 * c + 1]);
        } else {
            for (; c < z_hi; c++) {
                const uint64_t v = col[(size_t)c * L];
                only0.mac(v, ap[2 * c], ap[2 * c + 1]);
                only1.mac(v, az[2 * c], az[2 * c + 1]);
            }
        }
        for (; c + 4 <= c_hi; c += 4) {
            uint64_t v[4];
#pragma unroll
            for (int t = 0; t < 4; t++) v[t] = col[(size_t)(c + t) * L];
#pragma unroll
            for (int t = 0; t < 4; t++) only0.mac(v[t], ap[2 * (c + t)], ap[2 * (c + t) + 1]);
        }
        for (; c < c_hi; c++) only0.mac(col[(size_t)c * L], ap[2 * c], ap[2 * c + 1]);
        first += nco;
    }
    const gl::Ext b{both.finish(0), both.finish(1)};
    s0 = gl::add(b, gl::Ext{only0.finish(0), only0.finish(1)});
    s1 = gl::add(b, gl::Ext{only1.finish(0), only1.finish(1)});
}

__global__ __launch_bounds__(256) void k_fri_combine(FriCombineParams p) {
    const size_t pos = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned log_L = p.log_n + p.rate_bits;
    if (pos >> log_L) return;
    const size_t n = (size_t)1 << p.log_n, L = (size_t)1 << log_L;
    const uint32_t r = (uint32_t)(pos >> p.log_n), k = (uint32_t)(pos & (n - 1));
    const uint64_t x = gl::mul(p.coset_base[r], root_pow(p.w_n_table, k, (uint32_t)(n >> 1)));
    gl::Ext s0, s1;
    fri_column_sums(p, pos, L, 0u, 0xFFFFFFFFu, s0, s1);
    const gl::Ext zeta{p.zeta[0], p.zeta[1]}, gzeta{p.gzeta[0], p.gzeta[1]};
    const gl::Ext d0 = gl::sub(gl::ext(x), zeta), d1 = gl::sub(gl::ext(x), gzeta);
    const gl::Ext inv01 = gl::inv(gl::mul(d0, d1));
    const gl::Ext i0 = gl::mul(inv01, d1), i1 = gl::mul(inv01, d0);
    gl::Ext q0 = gl::mul(gl::sub(s0, gl::Ext{p.c0[0], p.c0[1]}), i0);
    q0 = gl::mul(q0, gl::Ext{p.alpha_nz[0], p.alpha_nz[1]});
    const gl::Ext q1 = gl::mul(gl::sub(s1, gl::Ext{p.c1[0], p.c1[1]}), i1);
    const gl::Ext res = gl::add(q0, q1);
    reinterpret_cast<ulonglong2*>(p.out)[pos] = make_ulonglong2(res.a, res.b);
}
// Small domains (a wide STARK trace of a few hundred rows: 4 745 columns x 1 024 points): one lane per point walks
// thousands of columns alone while most of the chip idles.  The columns are then cut into `slices` ranges
// (blockIdx.y), each lane sums its range into a partial (s0, s1), and a second kernel adds the partials and divides.
__global__ __launch_bounds__(256) void k_fri_combine_partial(FriCombineParams p, uint32_t per_slice, uint64_t* __restrict__ part) {
    const size_t pos = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned log_L = p.log_n + p.rate_bits;
    if (pos >> log_L) return;
    const size_t L = (size_t)1 << log_L;
    const uint32_t lo = blockIdx.y * per_slice, hi = lo + per_slice;
    gl::Ext s0, s1;
    fri_column_sums(p, pos, L, lo, hi, s0, s1);
    uint64_t* dst = part + (((size_t)blockIdx.y << log_L) + pos) * 4;
    dst[0] = s0.a; dst[1] = s0.b; dst[2] = s1.a; dst[3] = s1.b;
}
__global__ __launch_bounds__(256) void k_fri_combine_finish(FriCombineParams p, uint32_t slices, const uint64_t* __restrict__ part) {
    const size_t pos = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned log_L = p.log_n + p.rate_bits;
    if (pos >> log_L) return;
    const size_t n = (size_t)1 << p.log_n;
    gl::Ext s0{0, 0}, s1{0, 0};
    for (uint32_t sl = 0; sl < slices; sl++) {
        const uint64_t* src = part + (((size_t)sl << log_L) + pos) * 4;
        s0 = gl::add(s0, gl::Ext{src[0], src[1]});
        s1 = gl::add(s1, gl::Ext{src[2], src[3]});
    }
    const uint32_t r = (uint32_t)(pos >> p.log_n), k = (uint32_t)(pos & (n - 1));
    const uint64_t x = gl::mul(p.coset_base[r], root_pow(p.w_n_table, k, (uint32_t)(n >> 1)));
    const gl::Ext zeta{p.zeta[0], p.zeta[1]}, gzeta{p.gzeta[0], p.gzeta[1]};
    const gl::Ext d0 = gl::sub(gl::ext(x), zeta), d1 = gl::sub(gl::ext(x), gzeta);
    const gl::Ext inv01 = gl::inv(gl::mul(d0, d1));
    const gl::Ext i0 = gl::mul(inv01, d1), i1 = gl::mul(inv01, d0);
    gl::Ext q0 = gl::mul(gl::sub(s0, gl::Ext{p.c0[0], p.c0[1]}), i0);
    q0 = gl::mul(q0, gl::Ext{p.alpha_nz[0], p.alpha_nz[1]});
    const gl::Ext q1 = gl::mul(gl::sub(s1, gl::Ext{p.c1[0], p.c1[1]}), i1);
    const gl::Ext res = gl::add(q0, q1);
    reinterpret_cast<ulonglong2*>(p.out)[pos] = make_ulonglong2(res.a, res.b);
}

// number of column slices for this shape (1 = the single-kernel path) and the scratch it needs
uint32_t fri_combine_slices(const FriCombineParams& p) {
    const size_t L = (size_t)1 << (p.log_n + p.rate_bits);
    uint32_t cols = 0;
    for (int o = 0; o < FRI_VIEWS; o++) cols += p.n_cols[o];
    if (L >= ((size_t)1 << 15) || cols < 256) return 1;
    uint32_t want = (uint32_t)((((size_t)1 << 17) + L - 1) / L);  // aim for ~2^17 lanes
    const uint32_t max_slices = cols / 64 ? cols / 64 : 1;        // at least 64 columns per slice
    if (want > max_slices) want = max_slices;
    return want > 64 ? 64 : want;
}
size_t fri_combine_scratch_words(const FriCombineParams& p) {
    const uint32_t s = fri_combine_slices(p);
    return s > 1 ? ((size_t)s << (p.log_n + p.rate_bits)) * 4 : 0;
}
void launch_fri_combine(hipStream_t st, const FriCombineParams& p, uint64_t* scratch) {
    const size_t L = (size_t)1 << (p.log_n + p.rate_bits);
    const uint32_t slices = scratch ? fri_combine_slices(p) : 1;
    if (slices <= 1) {
        hipLaunchKernelGGL(k_fri_combine, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, st, p);
        return;
    }
    uint32_t cols = 0;
    for (int o = 0; o < FRI_VIEWS; o++) cols += p.n_cols[o];
    const uint32_t per_slice = (cols + slices - 1) / slices;
    hipLaunchKernelGGL(k_fri_combine_partial, dim3((unsigned)((L + 255) / 256), slices), dim3(256), 0, st, p, per_slice, scratch);
    hipLaunchKernelGGL(k_fri_combine_finish, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, st, p, slices, scratch);
}

// Layer leaf digests.  values: ext (2 words), coset-major with sub-domain size n = 2^log_n
// (L = n << rate_bits points).  Leaf of folded index j' = (r, k') holds the `arity` points
// (r, k' + mm * n/arity), in-leaf slot m = bitrev(mm); its tree position is
// bitrev(r) * n' + bitrev(k').
template <int ARITY_BITS>
__global__ __launch_bounds__(256, 4) void k_fri_leaves(const uint64_t* __restrict__ values, unsigned log_n,
                                                    unsigned rate_bits, uint64_t* __restrict__ digests) {
    constexpr int ARITY = 1 << ARITY_BITS;
    const unsigned log_np = log_n - ARITY_BITS;
    const size_t jp_raw = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = (jp_raw >> (log_np + rate_bits)) == 0;   // every lane stays through the permutation (its linear layer is a
    const size_t jp = live ? jp_raw : ((size_t)1 << (log_np + rate_bits)) - 1;   // wave-wide matrix product); spare lanes store nothing
    const size_t np = (size_t)1 << log_np, n = (size_t)1 << log_n;
    const uint32_t r = (uint32_t)(jp >> log_np), kp = (uint32_t)(jp & (np - 1));
    const ulonglong2* v = reinterpret_cast<const ulonglong2*>(values) + ((size_t)r * n + kp);
    uint64_t s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = 0;
    // absorb 8 words = 4 ext elements per permutation, in slot order m = 0..ARITY-1 (the permutation's own
    // rolled round loops keep this loop from being unrolled; the compiler decides)
    for (int m0 = 0; m0 < ARITY; m0 += 4) {
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int m = m0 + t;
            const int mm = (int)(__brev((unsigned)m) >> (32 - ARITY_BITS));
            const ulonglong2 e = v[(size_t)mm * np];
            s[2 * t] = e.x;
            s[2 * t + 1] = e.y;
        }
        poseidon::permute_loose(s);
    }
    if (!live) return;
    const size_t leaf = ((size_t)gl::bitrev32(r, rate_bits) << log_np) + gl::bitrev32(kp, log_np);
    ulonglong2* dst = reinterpret_cast<ulonglong2*>(digests + leaf * 4);
    dst[0] = make_ulonglong2(gl::canon(s[0]), gl::canon(s[1]));
    dst[1] = make_ulonglong2(gl::canon(s[2]), gl::canon(s[3]));
}
void launch_fri_leaves(hipStream_t st, const uint64_t* d_values, unsigned log_n, unsigned rate_bits,
                       unsigned arity_bits, uint64_t* d_digests) {
    const size_t leaves = (size_t)1 << (log_n - arity_bits + rate_bits);
    const unsigned blocks = (unsigned)((leaves + 255) / 256);
    if (arity_bits == 4) hipLaunchKernelGGL(k_fri_leaves<4>, dim3(blocks), dim3(256), 0, st, d_values, log_n, rate_bits, d_digests);
    else if (arity_bits == 3) hipLaunchKernelGGL(k_fri_leaves<3>, dim3(blocks), dim3(256), 0, st, d_values, log_n, rate_bits, d_digests);
    else if (arity_bits == 2) hipLaunchKernelGGL(k_fri_leaves<2>, dim3(blocks), dim3(256), 0, st, d_values, log_n, rate_bits, d_digests);
}

// Fold: out[j'] = sum_t (beta/x)^t * (1/A) sum_mm w_A^(-t mm) v[mm],  x = shift * w_L^(i'), i' = 8k' + r
template <int ARITY_BITS>
__global__ __launch_bounds__(256) void k_fri_fold(const uint64_t* __restrict__ values, uint64_t* __restrict__ out,
                                                  unsigned log_n, unsigned rate_bits, uint64_t beta_a, uint64_t beta_b,
                                                  uint64_t shift_inv, const uint64_t* __restrict__ w_L_inv_table,
                                                  const uint64_t* __restrict__ w_A_inv_pows, uint64_t arity_inv) {
    constexpr int ARITY = 1 << ARITY_BITS;
    const unsigned log_np = log_n - ARITY_BITS;
    const unsigned log_L = log_n + rate_bits;
    const size_t jp_raw = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = (jp_raw >> (log_np + rate_bits)) == 0;   // every lane stays through the permutation (its linear layer is a
    const size_t jp = live ? jp_raw : ((size_t)1 << (log_np + rate_bits)) - 1;   // wave-wide matrix product); spare lanes store nothing
    const size_t np = (size_t)1 << log_np, n = (size_t)1 << log_n;
    const uint32_t r = (uint32_t)(jp >> log_np), kp = (uint32_t)(jp & (np - 1));
    const ulonglong2* v = reinterpret_cast<const ulonglong2*>(values) + ((size_t)r * n + kp);
    gl::Ext a[ARITY];
#pragma unroll
    for (int mm = 0; mm < ARITY; mm++) {
        const ulonglong2 e = v[(size_t)mm * np];
        a[mm] = gl::Ext{e.x, e.y};
    }
    // x^-1 = shift^-1 * w_L^-(i'), i' = (kp << rate_bits) + r  (old-layer index of the mm = 0 point)
    const uint32_t ip = (kp << rate_bits) + r;
    const uint64_t x_inv = gl::mul(shift_inv, root_pow(w_L_inv_table, ip, 1u << (log_L - 1)));
    const gl::Ext bx = gl::mul(gl::Ext{beta_a, beta_b}, x_inv);
    // naive inverse DFT fused with Horner in (beta/x): res = sum_t bx^t * (1/A) * sum_mm w^(-t mm) a[mm]
    gl::Ext res{0, 0};
#pragma unroll 1
    for (int t = ARITY - 1; t >= 0; t--) {
        gl::Ext c{0, 0};
#pragma unroll
        for (int mm = 0; mm < ARITY; mm++) c = gl::add(c, gl::mul(a[mm], w_A_inv_pows[(t * mm) & (ARITY - 1)]));
        res = gl::add(gl::mul(res, bx), c);
    }
    res = gl::mul(res, arity_inv);
    reinterpret_cast<ulonglong2*>(out)[jp] = make_ulonglong2(res.a, res.b);
}
void launch_fri_fold(hipStream_t st, const uint64_t* d_values, uint64_t* d_out, unsigned log_n, unsigned rate_bits,
                     unsigned arity_bits, const uint64_t beta[2], uint64_t shift_inv, const uint64_t* d_w_L_inv_table,
                     const uint64_t* d_w_A_inv_pows) {
    const size_t outs = (size_t)1 << (log_n - arity_bits + rate_bits);
    const unsigned blocks = (unsigned)((outs + 255) / 256);
    const uint64_t ainv = gl::inv((uint64_t)1 << arity_bits);
    if (arity_bits == 4) hipLaunchKernelGGL(k_fri_fold<4>, dim3(blocks), dim3(256), 0, st, d_values, d_out, log_n, rate_bits, beta[0], beta[1], shift_inv, d_w_L_inv_table, d_w_A_inv_pows, ainv);
    else if (arity_bits == 3) hipLaunchKernelGGL(k_fri_fold<3>, dim3(blocks), dim3(256), 0, st, d_values, d_out, log_n, rate_bits, beta[0], beta[1], shift_inv, d_w_L_inv_table, d_w_A_inv_pows, ainv);
    else if (arity_bits == 2) hipLaunchKernelGGL(k_fri_fold<2>, dim3(blocks), dim3(256), 0, st, d_values, d_out, log_n, rate_bits, beta[0], beta[1], shift_inv, d_w_L_inv_table, d_w_A_inv_pows, ainv);
}

// Final polynomial: coefficients of the degree < n_f polynomial whose values on
// shift * <w_Lf> are given (coset-major).  Direct O(Lf * n_f) evaluation of the inverse transform;
// Lf <= 2^(final_poly_bits + arity_bits - 1 + rate_bits) is a few hundred points.
__global__ void k_fri_final_coeffs(const uint64_t* __restrict__ values, unsigned log_n, unsigned rate_bits,
                                   uint64_t shift_inv, uint64_t w_L_inv, uint64_t L_inv, uint64_t* __restrict__ out,
                                   uint32_t n_out) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_out) return;
    const unsigned log_L = log_n + rate_bits;
    const size_t L = (size_t)1 << log_L, n = (size_t)1 << log_n;
    // c_j = shift^-j / L * sum_i v_i w_L^(-i j)
    const uint64_t wj = gl::pow(w_L_inv, j);
    uint64_t wij = 1;
    gl::Ext acc{0, 0};
    for (size_t i = 0; i < L; i++) {
        const size_t pos = ((i & ((1u << rate_bits) - 1)) * n) + (i >> rate_bits);
        const gl::Ext v{values[2 * pos], values[2 * pos + 1]};
        acc = gl::add(acc, gl::mul(v, wij));
        wij = gl::mul(wij, wj);
    }
    acc = gl::mul(acc, gl::mul(gl::pow(shift_inv, j), L_inv));
    out[2 * j] = acc.a;
    out[2 * j + 1] = acc.b;
}
void launch_fri_final_coeffs(hipStream_t st, const uint64_t* d_values, unsigned log_n, unsigned rate_bits,
                             uint64_t shift, uint64_t* d_out, uint32_t n_out) {
    const unsigned log_L = log_n + rate_bits;
    hipLaunchKernelGGL(k_fri_final_coeffs, dim3((n_out + 63) / 64), dim3(64), 0, st, d_values, log_n, rate_bits,
                       gl::inv(shift), gl::inv(gl::root_of_unity(log_L)), gl::inv((uint64_t)1 << log_L), d_out, n_out);
}

// ---- proof of work: smallest nonce whose duplex output word 7 has >= bits leading zeros ----
// 2^16 candidates per round: with 16 PoW bits a round succeeds with probability 1 - 1/e and costs one
// permutation latency (a 2^18 round is throughput-bound and takes ~4x longer)
constexpr uint32_t POW_GRID = 256, POW_BLOCK = 256;
__global__ __launch_bounds__(POW_BLOCK, 4) void k_pow_grind(PowParams p, unsigned long long* __restrict__ best) {
    const uint64_t stride = (uint64_t)POW_GRID * POW_BLOCK;
    const uint64_t gid = (uint64_t)blockIdx.x * POW_BLOCK + threadIdx.x;
    for (uint64_t round = 0; round < p.max_rounds; round++) {
        const uint64_t cand = round * stride + gid;
        // a smaller nonce found in an earlier round (or earlier in this one) ends the search:
        // every candidate below it has been or is being tested by a lane that cannot skip it.
        // (a wave leaves only as a whole: the permutation's linear layer is a matrix-core product over all 64 lanes)
        const unsigned long long b = __hip_atomic_load(best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__builtin_amdgcn_readfirstlane((uint32_t)(b < round * stride))) return;   // any lane's view will do: best only falls
        const bool valid = cand < gl::P;
        if (!__any((int)valid)) return;
        uint64_t s[12];
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = p.state[i];
#pragma unroll
        for (int i = 0; i < 8; i++)
            if ((uint32_t)i == p.pos) s[i] = cand;
        poseidon::permute(s);
        const uint64_t resp = s[7];
        const unsigned lz = resp ? (unsigned)__clzll((long long)resp) : 64u;
        if (valid && lz >= p.bits) atomicMin(best, (unsigned long long)cand);
    }
}
void launch_pow_grind(hipStream_t st, const PowParams& p, unsigned long long* d_best) {
    (void)hipMemsetAsync(d_best, 0xFF, 8, st);
    hipLaunchKernelGGL(k_pow_grind, dim3(POW_GRID), dim3(POW_BLOCK), 0, st, p, d_best);
}

// ---- query gathers ----
// FRI layer openings: for query q and layer values (coset-major, sub-domain 2^log_n), the leaf
// with tree index leaf_idx[q]: `arity` ext values in slot order.
__global__ void k_fri_gather_leaf(const uint64_t* __restrict__ values, unsigned log_n, unsigned rate_bits,
                                  unsigned arity_bits, const uint64_t* __restrict__ leaf_idx, uint64_t* __restrict__ out,
                                  size_t out_stride_words) {
    const uint32_t q = blockIdx.x;
    const uint32_t m = threadIdx.x;
    const uint32_t arity = 1u << arity_bits;
    if (m >= arity) return;
    const unsigned log_np = log_n - arity_bits;
    const size_t np = (size_t)1 << log_np, n = (size_t)1 << log_n;
    const uint64_t leaf = leaf_idx[q];
    const uint32_t r = gl::bitrev32((uint32_t)(leaf >> log_np), rate_bits);
    const uint32_t kp = gl::bitrev32((uint32_t)(leaf & (np - 1)), log_np);
    const uint32_t mm = gl::bitrev32(m, arity_bits);
    const size_t pos = (size_t)r * n + kp + (size_t)mm * np;
    uint64_t* o = out + (size_t)q * out_stride_words + 2 * m;
    o[0] = values[2 * pos];
    o[1] = values[2 * pos + 1];
}
void launch_fri_gather_leaf(hipStream_t st, const uint64_t* d_values, unsigned log_n, unsigned rate_bits,
                            unsigned arity_bits, const uint64_t* d_leaf_idx, uint32_t n_q, uint64_t* d_out,
                            size_t out_stride_words) {
    hipLaunchKernelGGL(k_fri_gather_leaf, dim3(n_q), dim3(64), 0, st, d_values, log_n, rate_bits, arity_bits,
                       d_leaf_idx, d_out, out_stride_words);
}

// idx_out[q] = idx_in[q] >> shift
__global__ void k_shift_indices(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, uint32_t n, unsigned shift) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) out[q] = in[q] >> shift;
}
void launch_shift_indices(hipStream_t st, const uint64_t* d_in, uint64_t* d_out, uint32_t n, unsigned shift) {
    hipLaunchKernelGGL(k_shift_indices, dim3((n + 63) / 64), dim3(64), 0, st, d_in, d_out, n, shift);
}

// alpha power tables: out[c][t] = alpha_c^t (base field)
__global__ void k_pow_table(uint64_t* __restrict__ out, uint64_t a0, uint64_t a1, uint32_t count, uint32_t stride) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    out[t] = gl::pow(a0, t);
    out[stride + t] = gl::pow(a1, t);
}
void launch_pow_table(hipStream_t st, uint64_t* d_out, uint64_t a0, uint64_t a1, uint32_t count, uint32_t stride) {
    hipLaunchKernelGGL(k_pow_table, dim3((count + 63) / 64), dim3(64), 0, st, d_out, a0, a1, count, stride);
}
// ext power table: out[t] = alpha^t (2 words each)
__global__ void k_ext_pow_table(uint64_t* __restrict__ out, uint64_t a, uint64_t b, uint32_t count) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const gl::Ext v = gl::pow(gl::Ext{a, b}, t);
    out[2 * t] = v.a;
    out[2 * t + 1] = v.b;
}
void launch_ext_pow_table(hipStream_t st, uint64_t* d_out, const uint64_t alpha[2], uint32_t count) {
    hipLaunchKernelGGL(k_ext_pow_table, dim3((count + 63) / 64), dim3(64), 0, st, d_out, alpha[0], alpha[1], count);
}

}  // namespace nlx
