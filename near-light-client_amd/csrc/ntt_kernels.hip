// Goldilocks NTT / LDE kernels for gfx950.
//
// Replaces plonky2_field::fft::{fft_classic, ifft_with_options}, PolynomialCoeffs::{lde, coset_fft}
// and the transpose + reverse_index_bits_in_place of PolynomialBatch::lde_values
// (pinned crates at /root/reference/Cargo.lock:4912-4974).
//
// MI355X-first design:
//  * No bit-reversal or transpose pass exists anywhere.  The inverse transform runs
//    decimation-in-frequency (natural values -> bit-reversed coefficients), the forward
//    low-degree extension runs decimation-in-time (bit-reversed coefficients -> natural
//    values), so coefficients simply LIVE in bit-reversed order in HBM.
//  * An LDE of rate 2^b is 2^b independent size-n transforms of pre-scaled coefficients
//    (coset r uses scale (g*w_L^r)^i), not one zero-padded size-(n<<b) transform: 3 fewer
//    butterfly levels and no zero traffic at rate 8.  The table is stored coset-major,
//    [col][r][k] = p_col(g * w_L^(8k+r)).
//  * Each pass stages a tile of 4096 field elements (32 KiB) in LDS, runs up to 12 radix-2
//    levels there, and touches HBM once per pass with >=128-byte contiguous segments.
#include <hip/hip_runtime.h>
#include "gl.hpp"
#include "gl32.hpp"
#include "launch.hpp"

namespace nlx {

constexpr unsigned TILE_LOG = 12;
constexpr unsigned TILE = 1u << TILE_LOG;
constexpr unsigned NTT_THREADS = 256;
constexpr unsigned STRIDED_BITS_MAX = 8;  // 256-point strided sub-transform, 16-element (128 B) rows

struct PassParams {
    const uint64_t* src;   // read base (column 0)
    uint64_t* dst;         // write base (column 0)
    size_t src_stride;     // elements between columns
    size_t dst_stride;
    size_t src_z_stride;   // elements between blockIdx.z slices (cosets) on the read side
    size_t dst_z_stride;
    const uint64_t* tw;    // w_N^e, e in [0, N/2), N = 2^log_N = sub-problem size of this pass
    const uint64_t* scale; // optional per-element factor applied on load (indexed like src within a z slice)
    size_t scale_z_stride;
    uint64_t final_scale;  // multiplied into every output (1 = none)
    const uint64_t* post_scale;  // optional per-element factor applied on store (indexed like dst within a z slice)
    size_t post_scale_z_stride;
    unsigned log_n;        // column length
    unsigned log_N;        // sub-problem size at this pass (A * M)
    unsigned log_A;        // transform length inside the tile
    unsigned log_T;        // consecutive elements per tile row (strided pass) ; 0 for contiguous
    unsigned log_Q;        // sub-problems per tile (contiguous pass); 0 for strided
    uint32_t n_cols;       // tile kernels: columns in the batch; a block walks cols_per_block of them at one tile position
    uint32_t cols_per_block;
};

__device__ __forceinline__ uint64_t tw_full(const uint64_t* __restrict__ tw, uint32_t e, uint32_t half_N) {
    // w_N^e for e in [0, N): the table holds the first half, the second half is its negation
    return e < half_N ? tw[e] : gl::P - tw[e - half_N];
}

// ---- in-register radix-2^g sub-transforms; the constant twiddles (powers of w_16) are powers of TWO ----
// 2 has order 192 in F_p (2^96 = -1), so the subgroup of order 64 is <8> and every w_16^e is +-2^s: a multiplication by it
// is a shift and a short reduction instead of a 64 x 64 product.  W16_LOG2 = log_2 of the library's w_16 (set 7:
// POW2_GEN^(2^28) = 2^156; set 2021: 2^12), its inverse 2^(192 - that); run_passes checks the claim against
// gl::root_of_unity on the host before the first launch.
#if NLX_GL_GENERATOR_SET == 7
constexpr int W16_LOG2 = 156;
#else
constexpr int W16_LOG2 = 12;
#endif

// One wave issues a vector instruction every four cycles whether or not it depends on the one before (MI355X_MICROARCH.md,
// cycle constants), so interleaving independent butterflies buys nothing - but hipcc's scheduler does it, and the sixteen
// products it keeps in flight cost the tile kernels ~100 spilled registers.  A fence after every butterfly / product keeps
// the live set at the data plus one operation's temporaries.
#define NLX_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

// x * 2^s mod p for canonical x, 0 <= s < 96 -> canonical
__device__ __forceinline__ uint64_t mul_pow2(uint64_t x, int s) {
    if (s == 0) return x;
    if (s < 32) {  // (x2 : x1 : x0) = x << s, x2 < 2^s:  (x1:x0) + x2 * EPS
        return gl::canon(gl32::to_u64(gl32::add_w2_eps(x << s, (uint32_t)(x >> (64 - s)))));
    }
    if (s == 32) return gl::canon(gl32::to_u64(gl32::add_w2_eps(x << 32, (uint32_t)(x >> 32))));
    if (s < 64) {  // y = x << (s - 32) = (y2:y1:y0); y * 2^32 = the 128-bit limbs (y2 : y1 : y0 : 0)
        const int t = s - 32;
        const uint64_t y = x << t;
        return gl::canon(gl32::to_u64(gl32::reduce128(0u, (uint32_t)y, (uint32_t)(y >> 32), (uint32_t)(x >> (64 - t)))));
    }
    // y = x << (s - 64) = (y2:y1:y0); y * 2^64 = y0 * EPS - y1 - y2 * 2^32  (2^96 = -1): both terms are canonical
    const int t = s - 64;
    const uint64_t y = t ? x << t : x;
    const uint32_t y2 = t ? (uint32_t)(x >> (64 - t)) : 0u;
    return gl::sub((uint64_t)(uint32_t)y * 0xFFFFFFFFull, ((uint64_t)y2 << 32) | (y >> 32));
}

// DIF: natural in, bit-reversed out.  DIT: bit-reversed in, natural out.  x points at 2^g canonical values.
template <int g, bool INV>
__device__ __forceinline__ void dft_dif(uint64_t* x) {
    constexpr int G = 1 << g;
    constexpr int L = INV ? 192 - W16_LOG2 : W16_LOG2;
#pragma unroll
    for (int t = 0; t < g; t++) {
        const int half = G >> (t + 1);
#pragma unroll
        for (int b = 0; b < G; b += 2 * half) {
#pragma unroll
            for (int u = 0; u < half; u++) {
                const uint64_t a = x[b + u], c = x[b + u + half];
                const int e = (u << t) * (16 / G);  // w_{2 half}^u = w_G^(u << t) = w_16^e = +-2^s
                const int S = (e * L) % 192;
                x[b + u] = gl::add(a, c);
                x[b + u + half] = mul_pow2(S >= 96 ? gl::sub(c, a) : gl::sub(a, c), S % 96);
                NLX_SCHED_FENCE();
            }
        }
    }
}
template <int g, bool INV>
__device__ __forceinline__ void dft_dit(uint64_t* x) {
    constexpr int G = 1 << g;
    constexpr int L = INV ? 192 - W16_LOG2 : W16_LOG2;
#pragma unroll
    for (int t = 0; t < g; t++) {
        const int half = 1 << t;
#pragma unroll
        for (int b = 0; b < G; b += 2 * half) {
#pragma unroll
            for (int u = 0; u < half; u++) {
                const uint64_t a = x[b + u];
                const int e = (u << (g - 1 - t)) * (16 / G);
                const int S = (e * L) % 192;
                const uint64_t c = mul_pow2(x[b + u + half], S % 96);
                x[b + u] = S >= 96 ? gl::sub(a, c) : gl::add(a, c);
                x[b + u + half] = S >= 96 ? gl::add(a, c) : gl::sub(a, c);
                NLX_SCHED_FENCE();
            }
        }
    }
}

__device__ __forceinline__ uint32_t lds_pad(uint32_t i) { return i + (i >> 4); }  // breaks power-of-two strides

// One radix-2^g group over the tile in LDS.  Blocks of size B = 2^(log_P + g) along j1; the thread
// owning (block c, offset lo, column jt) transforms the 2^g elements m*P + lo.
// twl[e] = w_A^e (e < A/2) is the tile's root table in LDS: the 15 per-thread twiddle gathers of a
// group hit LDS banks instead of L1 (measured: global gathers made the pass 2.5x slower than its
// instruction count).
template <bool DIT, bool INV, int g>
__device__ __forceinline__ void ntt_group(uint64_t* __restrict__ lds, const uint64_t* __restrict__ twl,
                                          const PassParams& p, unsigned log_P, unsigned log_T, uint32_t tile_elems) {
    constexpr int G = 1 << g;
    const unsigned log_B = log_P + g;
    const unsigned up = p.log_A - log_B;  // w_B^e = w_A^(e << up)
    const uint32_t half_A = 1u << (p.log_A - 1);
    const uint32_t n_items = tile_elems >> g;
    for (uint32_t w = threadIdx.x; w < n_items; w += NTT_THREADS) {
        const uint32_t jt = w & ((1u << log_T) - 1);
        const uint32_t lo = (w >> log_T) & ((1u << log_P) - 1);
        const uint32_t c = w >> (log_T + log_P);
        uint64_t x[G];
        uint32_t idx[G];
#pragma unroll
        for (int m = 0; m < G; m++) {
            idx[m] = lds_pad((((((c << g) + m) << log_P) | lo) << log_T) | jt);
            x[m] = lds[idx[m]];
        }
        if (DIT) {
            if (log_P != 0) {
#pragma unroll
                for (int m = 1; m < G; m++) {
                    const uint32_t i1 = __brev((uint32_t)m) >> (32 - g);
                    const uint32_t e = (lo * i1) << up;
                    x[m] = gl::mul(x[m], tw_full(twl, e, half_A));
                }
            }
            dft_dit<g, INV>(x);
        } else {
            dft_dif<g, INV>(x);
            if (log_P != 0) {
#pragma unroll
                for (int m = 1; m < G; m++) {
                    const uint32_t k1 = __brev((uint32_t)m) >> (32 - g);
                    const uint32_t e = (lo * k1) << up;
                    x[m] = gl::mul(x[m], tw_full(twl, e, half_A));
                }
            }
        }
#pragma unroll
        for (int m = 0; m < G; m++) lds[idx[m]] = x[m];
    }
}

// One pass of a decimation-in-frequency (DIT = false) or decimation-in-time (DIT = true)
// transform over a tile held in LDS, log_A butterfly levels done as radix-16 register groups.
//   strided pass   : element (j1, jt) of tile (q, t) lives at q*N + j1*M + t*T + jt
//   contiguous pass: element (qq, j1) of tile `tile` lives at (tile*Q + qq)*A + j1   (M = T = 1)
template <bool DIT, bool INV>
__global__ __launch_bounds__(NTT_THREADS) void k_ntt_pass(PassParams p) {
    __shared__ uint64_t lds[TILE + TILE / 16];
    __shared__ uint64_t twl[TILE / 2];
    const unsigned tid = threadIdx.x;
    const unsigned log_A = p.log_A, log_T = p.log_T, log_Q = p.log_Q;
    const uint32_t A = 1u << log_A, T = 1u << log_T;
    const unsigned log_M = p.log_N - log_A;
    const uint32_t tile_elems = 1u << (log_A + log_T + log_Q);
    const uint32_t half_N = p.log_N ? 1u << (p.log_N - 1) : 0u;
    const bool strided = log_M != 0;

    const uint64_t* __restrict__ src = p.src + (size_t)blockIdx.y * p.src_stride + (size_t)blockIdx.z * p.src_z_stride;
    uint64_t* __restrict__ dst = p.dst + (size_t)blockIdx.y * p.dst_stride + (size_t)blockIdx.z * p.dst_z_stride;
    const uint64_t* __restrict__ scale = p.scale ? p.scale + (size_t)blockIdx.z * p.scale_z_stride : nullptr;
    const uint64_t* __restrict__ tw = p.tw;

    size_t base;
    uint32_t j0_base;
    if (strided) {
        const uint32_t tiles_per_sub = 1u << (log_M - log_T);
        const uint32_t q = blockIdx.x >> (log_M - log_T);
        const uint32_t t = blockIdx.x & (tiles_per_sub - 1);
        j0_base = t << log_T;
        base = ((size_t)q << p.log_N) + j0_base;
    } else {
        j0_base = 0;
        base = (size_t)blockIdx.x << (log_A + log_Q);
    }

    // tile root table: w_A^e = w_N^(e << log_M)
    if (log_A >= 1)
        for (uint32_t e = tid; e < (A >> 1); e += NTT_THREADS) twl[e] = tw[(size_t)e << log_M];
    // ---- load (optional pre-scale; DIT strided passes apply the inter-pass twiddle here) ----
    if (!strided && tile_elems >= 2 * NTT_THREADS) {
        // contiguous tile: 16-byte accesses (two consecutive elements per lane); 8-byte accesses reach
        // only 0.5-0.7x of the 16-byte streaming rate on this part
        const ulonglong2* __restrict__ src2 = reinterpret_cast<const ulonglong2*>(src + base);
        const ulonglong2* __restrict__ sc2 = scale ? reinterpret_cast<const ulonglong2*>(scale + base) : nullptr;
        for (uint32_t i2 = tid; i2 < (tile_elems >> 1); i2 += NTT_THREADS) {
            ulonglong2 v = src2[i2];
            if (sc2) {
                const ulonglong2 sc = sc2[i2];
                v.x = gl::mul(v.x, sc.x);
                v.y = gl::mul(v.y, sc.y);
            }
            const uint32_t li = lds_pad(2 * i2);  // 2*i2 is even: its neighbour shares the 16-element pad group
            lds[li] = v.x;
            lds[li + 1] = v.y;
        }
    } else {
        for (uint32_t idx = tid; idx < tile_elems; idx += NTT_THREADS) {
            const uint32_t jt = idx & (T - 1);
            const uint32_t j1 = (idx >> log_T) & (A - 1);
            const size_t g = strided ? base + ((size_t)j1 << log_M) + jt : base + idx;
            uint64_t v = src[g];
            if (scale) v = gl::mul(v, scale[g]);
            if (DIT && strided) {
                const uint32_t i1 = gl::bitrev32(j1, log_A);
                const uint32_t e = i1 * (j0_base + jt);
                if (e) v = gl::mul(v, tw_full(tw, e, half_N));
            }
            lds[lds_pad(idx)] = v;
        }
    }
    __syncthreads();

    // ---- butterfly levels along j1, four at a time ----
    unsigned done = 0;
    while (done < log_A) {
        const unsigned rem = log_A - done;
        const unsigned g = rem >= 4 ? 4 : rem;
        // DIF: blocks shrink (B = A >> done); DIT: blocks grow (P = 1 << done)
        const unsigned log_P = DIT ? done : (log_A - done - g);
        switch (g) {
            case 4: ntt_group<DIT, INV, 4>(lds, twl, p, log_P, log_T, tile_elems); break;
            case 3: ntt_group<DIT, INV, 3>(lds, twl, p, log_P, log_T, tile_elems); break;
            case 2: ntt_group<DIT, INV, 2>(lds, twl, p, log_P, log_T, tile_elems); break;
            default: ntt_group<DIT, INV, 1>(lds, twl, p, log_P, log_T, tile_elems); break;
        }
        done += g;
        __syncthreads();
    }

    // ---- store (DIF strided passes apply the inter-pass twiddle; optional scales) ----
    if (!strided && tile_elems >= 2 * NTT_THREADS) {
        ulonglong2* __restrict__ dst2 = reinterpret_cast<ulonglong2*>(dst + base);
        const ulonglong2* __restrict__ ps2 =
            p.post_scale ? reinterpret_cast<const ulonglong2*>(p.post_scale + (size_t)blockIdx.z * p.post_scale_z_stride + base) : nullptr;
        for (uint32_t i2 = tid; i2 < (tile_elems >> 1); i2 += NTT_THREADS) {
            const uint32_t li = lds_pad(2 * i2);
            ulonglong2 v = make_ulonglong2(lds[li], lds[li + 1]);
            if (p.final_scale != 1) {
                v.x = gl::mul(v.x, p.final_scale);
                v.y = gl::mul(v.y, p.final_scale);
            }
            if (ps2) {
                const ulonglong2 sc = ps2[i2];
                v.x = gl::mul(v.x, sc.x);
                v.y = gl::mul(v.y, sc.y);
            }
            dst2[i2] = v;
        }
        return;
    }
    for (uint32_t idx = tid; idx < tile_elems; idx += NTT_THREADS) {
        const uint32_t jt = idx & (T - 1);
        const uint32_t j1 = (idx >> log_T) & (A - 1);
        const size_t g = strided ? base + ((size_t)j1 << log_M) + jt : base + idx;
        uint64_t v = lds[lds_pad(idx)];
        if (!DIT && strided) {
            const uint32_t k1 = gl::bitrev32(j1, log_A);
            const uint32_t e = k1 * (j0_base + jt);
            if (e) v = gl::mul(v, tw_full(tw, e, half_N));
        }
        if (p.final_scale != 1) v = gl::mul(v, p.final_scale);
        if (p.post_scale) v = gl::mul(v, p.post_scale[(size_t)blockIdx.z * p.post_scale_z_stride + g]);
        dst[g] = v;
    }
}

// Strided pass with a transform length of at most 16 (log_A <= 4): no LDS at all.  One lane owns one
// column j0 of one sub-problem, loads its A elements with A wave-coalesced loads (lanes = consecutive
// j0), forms the inter-pass twiddles w_N^(j0 * i) by repeated multiplication from ONE coalesced table
// read (instead of A gathered reads), transforms in registers and stores back in place.
template <bool DIT, bool INV, int g>
__global__ __launch_bounds__(256) void k_ntt_strided_reg(PassParams p) {
    constexpr int G = 1 << g;
    const unsigned log_M = p.log_N - g;
    const size_t col_elems = (size_t)1 << p.log_n;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // index over (q, j0)
    if (t >= (col_elems >> g)) return;
    const uint32_t j0 = (uint32_t)(t & (((size_t)1 << log_M) - 1));
    const size_t q = t >> log_M;
    const size_t base = (q << p.log_N) + j0;
    const uint64_t* __restrict__ src = p.src + (size_t)blockIdx.y * p.src_stride + (size_t)blockIdx.z * p.src_z_stride;
    uint64_t* __restrict__ dst = p.dst + (size_t)blockIdx.y * p.dst_stride + (size_t)blockIdx.z * p.dst_z_stride;
    uint64_t x[G];
#pragma unroll
    for (int m = 0; m < G; m++) x[m] = src[base + ((size_t)m << log_M)];
    if (p.scale) {
        const uint64_t* __restrict__ sc = p.scale + (size_t)blockIdx.z * p.scale_z_stride;
#pragma unroll
        for (int m = 0; m < G; m++) x[m] = gl::mul(x[m], sc[base + ((size_t)m << log_M)]);
    }
    // pw[i] = w_N^(j0 * i), i < G  (j0 < M <= N/2: direct table entry)
    uint64_t pw[G];
    pw[0] = 1;
    pw[1] = p.tw[j0];
#pragma unroll
    for (int i = 2; i < G; i++) pw[i] = (i & 1) ? gl::mul(pw[i - 1], pw[1]) : gl::mul(pw[i / 2], pw[i / 2]);
    if (DIT) {
#pragma unroll
        for (int m = 1; m < G; m++) x[m] = gl::mul(x[m], pw[__brev((unsigned)m) >> (32 - g)]);
        dft_dit<g, INV>(x);
    } else {
        dft_dif<g, INV>(x);
#pragma unroll
        for (int m = 1; m < G; m++) x[m] = gl::mul(x[m], pw[__brev((unsigned)m) >> (32 - g)]);
    }
    const uint64_t* __restrict__ ps = p.post_scale ? p.post_scale + (size_t)blockIdx.z * p.post_scale_z_stride : nullptr;
#pragma unroll
    for (int m = 0; m < G; m++) {
        const size_t gi = base + ((size_t)m << log_M);
        uint64_t v = x[m];
        if (p.final_scale != 1) v = gl::mul(v, p.final_scale);
        if (ps) v = gl::mul(v, ps[gi]);
        dst[gi] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Tile kernels (column length >= 2^12): a block of 256 lanes owns ONE tile position of 4096 elements - sixteen per lane in
// every radix round - and walks `cols_per_block` columns of the batch through it.  What depends on the position only is
// made once per block and kept in registers / LDS across the columns: the inter-pass twiddles w_N^(k1 j0) (sixteen per
// lane, a geometric sequence built from TWO table reads; the generic kernel gathers one value per element and pass from a
// table of N/2 entries - 64 MB at 2^24), and the rounds' own twiddles.  The first round reads global memory and the last
// one writes it wherever the round's lane order is the memory order, so an element makes one LDS round trip per exchange
// between rounds instead of one per round plus a staging copy at either end.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t brev4(uint32_t m) { return __brev(m) >> 28; }
__device__ __forceinline__ uint64_t mul_f(uint64_t a, uint64_t b) {
    const uint64_t r = gl::mul(a, b);
    NLX_SCHED_FENCE();
    return r;
}
// Global memory of the tile kernels goes through buffer instructions: a wave-uniform descriptor per column (SGPRs), ONE
// lane byte offset per round (a VGPR) and the row of the element as a scalar offset - sixteen accesses cost one address
// register.  With 64-bit flat addresses hipcc hoists sixteen address pairs per round out of the column loop (64 VGPRs).
// Columns of the tile kernels are at most 2^28 elements, so byte offsets fit 32 bits.
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t buf_t;
__device__ __forceinline__ buf_t make_buf(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0xFFFFFFFFu, 0x00020000);
}
__device__ __forceinline__ uint64_t buf_ld(buf_t r, uint32_t voff, uint32_t soff) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return (uint64_t)v.x | ((uint64_t)v.y << 32);
}
__device__ __forceinline__ void buf_st(buf_t r, uint32_t voff, uint32_t soff, uint64_t x) {
    u32x2 v;
    v.x = (uint32_t)x;
    v.y = (uint32_t)(x >> 32);
    __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, 0);
}
__device__ __forceinline__ ulonglong2 buf_ld16(buf_t r, uint32_t voff, uint32_t soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_ulonglong2((uint64_t)v.x | ((uint64_t)v.y << 32), (uint64_t)v.z | ((uint64_t)v.w << 32));
}
__device__ __forceinline__ void buf_st16(buf_t r, uint32_t voff, uint32_t soff, ulonglong2 x) {
    u32x4 v;
    v.x = (uint32_t)x.x;
    v.y = (uint32_t)(x.x >> 32);
    v.z = (uint32_t)x.y;
    v.w = (uint32_t)(x.y >> 32);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, 0);
}

// Contiguous pass, 12 levels = three radix-16 rounds over the tile [base, base + 4096).  DIF (strides 256, 16, 1): direct
// load, the store goes through LDS (round 3 leaves a lane with 16 consecutive elements).  DIT (strides 1, 16, 256): the
// load goes through LDS, direct store.  p.tw = w_4096^e, e < 2048.  LDS positions are padded by one element per sixteen
// (lds_pad), written out here as lane base + m x constant so that the sixteen accesses of a round share one address register.
template <bool DIT, bool INV>
__global__ __launch_bounds__(256, 4) void k_ntt_c12(PassParams p) {
    __shared__ uint64_t lds[TILE + TILE / 16];
    __shared__ uint64_t w2[256];   // w_256^(lo * brev4(m)) at [lo][m]
    const uint32_t tid = threadIdx.x;
    const size_t base = (size_t)blockIdx.x << TILE_LOG;
    const uint64_t* __restrict__ tw = p.tw;
    // round of stride 256: lane `tid` multiplies element m by w_4096^(tid * brev4(m)) - the same for every column
    uint64_t W1[16];
#pragma unroll
    for (int m = 1; m < 16; m++) W1[m] = tw_full(tw, tid * brev4(m), TILE / 2);
    w2[tid] = tw_full(tw, 16u * (tid >> 4) * brev4(tid & 15), TILE / 2);
    __syncthreads();
    const uint64_t* __restrict__ w2l = w2 + (tid & 15) * 16;
    uint64_t* __restrict__ l256 = lds + tid + (tid >> 4);                      // + 272 m : element 256 m + tid
    uint64_t* __restrict__ l16 = lds + (tid >> 4) * 272 + (tid & 15);          // + 17 m  : element 256 c + 16 m + lo
    uint64_t* __restrict__ l1 = lds + tid * 17;                                // + m     : element 16 tid + m
    const bool has_scale = p.scale != nullptr, has_post = p.post_scale != nullptr;
    const buf_t scale = make_buf(has_scale ? p.scale + (size_t)blockIdx.z * p.scale_z_stride + base : nullptr);
    const buf_t post = make_buf(has_post ? p.post_scale + (size_t)blockIdx.z * p.post_scale_z_stride + base : nullptr);
    const uint32_t col_end = min(p.n_cols, (blockIdx.y + 1) * p.cols_per_block);
    for (uint32_t col = blockIdx.y * p.cols_per_block; col < col_end; col++) {
        const buf_t src = make_buf(p.src + (size_t)col * p.src_stride + (size_t)blockIdx.z * p.src_z_stride + base);
        const buf_t dst = make_buf(p.dst + (size_t)col * p.dst_stride + (size_t)blockIdx.z * p.dst_z_stride + base);
        uint64_t x[16];
        if (!DIT) {
#pragma unroll
            for (int m = 0; m < 16; m++) x[m] = buf_ld(src, tid * 8, m * 2048);
            if (has_scale) {
#pragma unroll
                for (int m = 0; m < 16; m++) x[m] = mul_f(x[m], buf_ld(scale, tid * 8, m * 2048));
            }
            dft_dif<4, INV>(x);
#pragma unroll
            for (int m = 1; m < 16; m++) x[m] = mul_f(x[m], W1[m]);
#pragma unroll
            for (int m = 0; m < 16; m++) l256[272 * m] = x[m];
            __syncthreads();
#pragma unroll
            for (int m = 0; m < 16; m++) x[m] = l16[17 * m];
            dft_dif<4, INV>(x);
#pragma unroll
            for (int m = 1; m < 16; m++) x[m] = mul_f(x[m], w2l[m]);
#pragma unroll
            for (int m = 0; m < 16; m++) l16[17 * m] = x[m];
            __syncthreads();
#pragma unroll
            for (int m = 0; m < 16; m++) x[m] = l1[m];
            dft_dif<4, INV>(x);
            if (p.final_scale != 1) {
#pragma unroll
                for (int m = 0; m < 16; m++) x[m] = mul_f(x[m], p.final_scale);
            }
#pragma unroll
            for (int m = 0; m < 16; m++) l1[m] = x[m];
            __syncthreads();
            // pairs (2 i2, 2 i2 + 1), i2 = tid + 256 i: lds_pad(2 i2) = 2 tid + (tid >> 3) + 544 i
            const uint64_t* __restrict__ lp = lds + 2 * tid + (tid >> 3);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                ulonglong2 v = make_ulonglong2(lp[544 * i], lp[544 * i + 1]);
                if (has_post) {
                    const ulonglong2 sc = buf_ld16(post, tid * 16, i * 4096);
                    v.x = mul_f(v.x, sc.x);
                    v.y = mul_f(v.y, sc.y);
                }
                buf_st16(dst, tid * 16, i * 4096, v);
            }
            __syncthreads();
        } else {
            uint64_t* __restrict__ lp = lds + 2 * tid + (tid >> 3);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                ulonglong2 v = buf_ld16(src, tid * 16, i * 4096);
                if (has_scale) {
                    const ulonglong2 sc = buf_ld16(scale, tid * 16, i * 4096);
                    v.x = mul_f(v.x, sc.x);
                    v.y = mul_f(v.y, sc.y);
                }
                lp[544 * i] = v.x;
                lp[544 * i + 1] = v.y;
            }
            __syncthreads();
#pragma unroll
            for (int m = 0; m < 16; m++) x[m] = l1[m];
            dft_dit<4, INV>(x);
#pragma unroll
            for (int m = 0; m < 16; m++) l1[m] = x[m];
            __syncthreads();
#pragma unroll
            for (int m = 0; m < 16; m++) x[m] = l16[17 * m];
#pragma unroll
            for (int m = 1; m < 16; m++) x[m] = mul_f(x[m], w2l[m]);
            dft_dit<4, INV>(x);
#pragma unroll
            for (int m = 0; m < 16; m++) l16[17 * m] = x[m];
            __syncthreads();
#pragma unroll
            for (int m = 0; m < 16; m++) x[m] = l256[272 * m];
#pragma unroll
            for (int m = 1; m < 16; m++) x[m] = mul_f(x[m], W1[m]);
            dft_dit<4, INV>(x);
            if (p.final_scale != 1) {
#pragma unroll
                for (int m = 0; m < 16; m++) x[m] = mul_f(x[m], p.final_scale);
            }
            if (has_post) {
#pragma unroll
                for (int m = 0; m < 16; m++) x[m] = mul_f(x[m], buf_ld(post, tid * 8, m * 2048));
            }
#pragma unroll
            for (int m = 0; m < 16; m++) buf_st(dst, tid * 8, m * 2048, x[m]);
            __syncthreads();
        }
    }
}

// Short columns (2^8 .. 2^11 points: the wide STARK traces of a Sync step - thousands of columns of 2^9 .. 2^11 rows): the
// same three rounds on a tile of 16 >> REM1 WHOLE COLUMNS of 2^(8 + REM1) points, the tile index being (column : 4 - REM1
// bits)(point : 8 + REM1 bits).  The round of stride 256 is a radix-2^REM1 transform per column (none for 2^8 points), the
// rounds of stride 16 and 1 are c12's.  The generic kernel gave such a column a block of its own with 15 of 16 lanes idle
// (1.6 ms per launch on the SHA-512 trace, 11 ms per Sync step).  p.tw = w_N^e, e < N/2, N = 2^(8 + REM1).
template <bool DIT, bool INV, int REM1>
__global__ __launch_bounds__(256, 4) void k_ntt_cols(PassParams p) {
    constexpr unsigned K = 8 + REM1, N = 1u << K, G = 1u << REM1, C = 16u >> REM1;
    __shared__ uint64_t lds[TILE + TILE / 16];
    __shared__ uint64_t w2[256];   // w_256^(lo * brev4(m)) at [lo][m]
    const uint32_t tid = threadIdx.x;
    const uint64_t* __restrict__ tw = p.tw;
    uint64_t W1[G];                // w_N^(tid * bitrev_REM1(jt))
#pragma unroll
    for (unsigned jt = 1; jt < G; jt++) W1[jt] = tw_full(tw, tid * (__brev(jt) >> (32 - (REM1 ? REM1 : 1))), N / 2);
    w2[tid] = tw_full(tw, G * (tid >> 4) * brev4(tid & 15), N / 2);   // w_256 = w_N^G
    __syncthreads();
    const uint64_t* __restrict__ w2l = w2 + (tid & 15) * 16;
    uint64_t* __restrict__ l256 = lds + tid + (tid >> 4);                      // + 272 m : element 256 m + tid
    uint64_t* __restrict__ l16 = lds + (tid >> 4) * 272 + (tid & 15);          // + 17 m  : element 256 c + 16 m + lo
    uint64_t* __restrict__ l1 = lds + tid * 17;                                // + m     : element 16 tid + m
    const bool has_scale = p.scale != nullptr, has_post = p.post_scale != nullptr;
    const buf_t scale = make_buf(has_scale ? p.scale + (size_t)blockIdx.z * p.scale_z_stride : nullptr);
    const buf_t post = make_buf(has_post ? p.post_scale + (size_t)blockIdx.z * p.post_scale_z_stride : nullptr);
    const uint32_t src_cs = (uint32_t)p.src_stride * 8, dst_cs = (uint32_t)p.dst_stride * 8;   // bytes between columns
    const uint32_t groups = (p.n_cols + C - 1) / C;
    const uint32_t grp_end = min(groups, (blockIdx.x + 1) * p.cols_per_block);
    for (uint32_t grp = blockIdx.x * p.cols_per_block; grp < grp_end; grp++) {
        const uint32_t col0 = grp * C, n_valid = min(C, p.n_cols - col0);     // columns col0 .. col0 + n_valid - 1
        const buf_t src = make_buf(p.src + (size_t)col0 * p.src_stride + (size_t)blockIdx.z * p.src_z_stride);
        const buf_t dst = make_buf(p.dst + (size_t)col0 * p.dst_stride + (size_t)blockIdx.z * p.dst_z_stride);
        uint64_t x[16];
        // pairs of consecutive tile elements for the staged side: element 2 (tid + 256 i) = column cc, point jj
        auto pair_col = [&](int i) { return (2 * (tid + 256 * i)) >> K; };
        auto pair_pt = [&](int i) { return (2 * (tid + 256 * i)) & (N - 1); };
        uint64_t* __restrict__ lp = lds + 2 * tid + (tid >> 3);                // + 544 i (lds_pad of the pair's first element)
        if (!DIT) {
            if (REM1 > 0) {
                // round of stride 256: register m = (column c, jt), point jt 256 + tid
#pragma unroll
                for (unsigned c = 0; c < C; c++)
#pragma unroll
                    for (unsigned jt = 0; jt < G; jt++)
                        x[c * G + jt] = c < n_valid ? buf_ld(src, tid * 8, c * src_cs + jt * 2048) : 0;
                if (has_scale) {
#pragma unroll
                    for (unsigned c = 0; c < C; c++)
#pragma unroll
                        for (unsigned jt = 0; jt < G; jt++) x[c * G + jt] = mul_f(x[c * G + jt], buf_ld(scale, tid * 8, jt * 2048));
                }
#pragma unroll
                for (unsigned c = 0; c < C; c++) {
                    dft_dif<REM1, INV>(x + c * G);
#pragma unroll
                    for (unsigned jt = 1; jt < G; jt++) x[c * G + jt] = mul_f(x[c * G + jt], W1[jt]);
                }
#pragma unroll
                for (int m = 0; m < 16; m++) l256[272 * m] = x[m];
                __syncthreads();
#pragma unroll
                for (int m = 0; m < 16; m++) x[m] = l16[17 * m];
            } else {
                // 2^8 points: the round of stride 16 reads global memory itself: lane (column c = tid >> 4, lo), point 16 m + lo
                const uint32_t c = tid >> 4, off = c * src_cs + (tid & 15) * 8;
#pragma unroll
                for (int m = 0; m < 16; m++) x[m] = c < n_valid ? buf_ld(src, off, m * 128) : 0;
                if (has_scale) {
#pragma unroll
                    for (int m = 0; m < 16; m++) x[m] = mul_f(x[m], buf_ld(scale, (tid & 15) * 8, m * 128));
                }
            }
            dft_dif<4, INV>(x);
#pragma unroll
            for (int m = 1; m < 16; m++) x[m] = mul_f(x[m], w2l[m]);
#pragma unroll
            for (int m = 0; m < 16; m++) l16[17 * m] = x[m];
            __syncthreads();
#pragma unroll
            for (int m = 0; m < 16; m++) x[m] = l1[m];
            dft_dif<4, INV>(x);
            if (p.final_scale != 1) {
#pragma unroll
                for (int m = 0; m < 16; m++) x[m] = mul_f(x[m], p.final_scale);
            }
#pragma unroll
            for (int m = 0; m < 16; m++) l1[m] = x[m];
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const uint32_t cc = pair_col(i), jj = pair_pt(i);
                ulonglong2 v = make_ulonglong2(lp[544 * i], lp[544 * i + 1]);
                if (has_post) {
                    const ulonglong2 sc = buf_ld16(post, jj * 8, 0);
                    v.x = mul_f(v.x, sc.x);
                    v.y = mul_f(v.y, sc.y);
                }
                if (cc < n_valid) buf_st16(dst, cc * dst_cs + jj * 8, 0, v);
            }
            __syncthreads();
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const uint32_t cc = pair_col(i), jj = pair_pt(i);
                ulonglong2 v = make_ulonglong2(0, 0);
                if (cc < n_valid) v = buf_ld16(src, cc * src_cs + jj * 8, 0);
                if (has_scale) {
                    const ulonglong2 sc = buf_ld16(scale, jj * 8, 0);
                    v.x = mul_f(v.x, sc.x);
                    v.y = mul_f(v.y, sc.y);
                }
                lp[544 * i] = v.x;
                lp[544 * i + 1] = v.y;
            }
            __syncthreads();
#pragma unroll
            for (int m = 0; m < 16; m++) x[m] = l1[m];
            dft_dit<4, INV>(x);
#pragma unroll
            for (int m = 0; m < 16; m++) l1[m] = x[m];
            __syncthreads();
#pragma unroll
            for (int m = 0; m < 16; m++) x[m] = l16[17 * m];
#pragma unroll
            for (int m = 1; m < 16; m++) x[m] = mul_f(x[m], w2l[m]);
            dft_dit<4, INV>(x);
            if (REM1 > 0) {
#pragma unroll
                for (int m = 0; m < 16; m++) l16[17 * m] = x[m];
                __syncthreads();
#pragma unroll
                for (int m = 0; m < 16; m++) x[m] = l256[272 * m];
#pragma unroll
                for (unsigned c = 0; c < C; c++) {
#pragma unroll
                    for (unsigned jt = 1; jt < G; jt++) x[c * G + jt] = mul_f(x[c * G + jt], W1[jt]);
                    dft_dit<REM1, INV>(x + c * G);
                }
            }
            if (p.final_scale != 1) {
#pragma unroll
                for (int m = 0; m < 16; m++) x[m] = mul_f(x[m], p.final_scale);
            }
            if (REM1 > 0) {
                if (has_post) {
#pragma unroll
                    for (unsigned c = 0; c < C; c++)
#pragma unroll
                        for (unsigned jt = 0; jt < G; jt++) x[c * G + jt] = mul_f(x[c * G + jt], buf_ld(post, tid * 8, jt * 2048));
                }
#pragma unroll
                for (unsigned c = 0; c < C; c++)
#pragma unroll
                    for (unsigned jt = 0; jt < G; jt++)
                        if (c < n_valid) buf_st(dst, tid * 8, c * dst_cs + jt * 2048, x[c * G + jt]);
            } else {
                const uint32_t c = tid >> 4, off = c * dst_cs + (tid & 15) * 8;
                if (has_post) {
#pragma unroll
                    for (int m = 0; m < 16; m++) x[m] = mul_f(x[m], buf_ld(post, (tid & 15) * 8, m * 128));
                }
                if (c < n_valid) {
#pragma unroll
                    for (int m = 0; m < 16; m++) buf_st(dst, off, m * 128, x[m]);
                }
            }
            __syncthreads();
        }
    }
}

// Strided pass of 4 + REM levels (REM = 1 .. 4): element (j1, jt) of the tile lives at q N + j1 M + j0_base + jt with
// A = 2^(4 + REM) values of j1 and T = 4096 / A consecutive jt (rows of >= 128 bytes).  Two rounds: a radix-2^REM round at
// stride 16 and a radix-16 round at stride 1; the latter carries the inter-pass twiddle w_N^(bitrev_A(j1) (j0_base + jt))
// - DIF: last, multiplied into the store; DIT: first, multiplied into the load - and every round's lanes are consecutive
// jt, so both ends touch global memory directly: one LDS exchange per element and pass.
template <bool DIT, bool INV, int REM>
__global__ __launch_bounds__(256, 4) void k_ntt_s(PassParams p) {
    constexpr unsigned LOG_A = 4 + REM, LOG_T = TILE_LOG - LOG_A, T = 1u << LOG_T, G1 = 1u << REM, ITEMS = 16 >> REM;
    constexpr unsigned ROW = T + T / 16;   // LDS elements per padded row of T
    __shared__ uint64_t lds[TILE + TILE / 16];
    __shared__ uint64_t wa[16 << REM];   // w_A^(lo * bitrev_REM(m)) at [lo][m]
    const uint32_t tid = threadIdx.x;
    const unsigned log_M = p.log_N - LOG_A;
    const uint32_t half_N = 1u << (p.log_N - 1);
    const uint32_t q = blockIdx.x >> (log_M - LOG_T), t = blockIdx.x & ((1u << (log_M - LOG_T)) - 1);
    const uint32_t j0_base = t << LOG_T;
    const size_t base = ((size_t)q << p.log_N) + j0_base;
    const uint64_t* __restrict__ tw = p.tw;
    for (uint32_t i = tid; i < (16u << REM); i += 256)
        wa[i] = tw_full(tw, ((i >> REM) * (__brev(i & (G1 - 1)) >> (32 - REM))) << log_M, half_N);
    // radix-16 round: lane = (c, jt), elements j1 = 16 c + m; bitrev_A(j1) = brev4(m) 2^REM + bitrev_REM(c)
    const uint32_t jt4 = tid & (T - 1), c4 = tid >> LOG_T;
    uint64_t F[16];
    {
        const uint32_t j0 = j0_base + jt4;
        const uint64_t p0 = tw[(size_t)j0 * (__brev(c4) >> (32 - REM))], p1 = tw[(size_t)j0 << REM];
        uint64_t f = p0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            F[brev4(i)] = f;
            if (i < 15) f = mul_f(f, p1);
        }
    }
    __syncthreads();
    // addresses: a wave-uniform row pointer (scalar arithmetic: + m rows) plus ONE lane byte offset per round; padded LDS
    // positions as lane base + m x constant (an instruction's immediate offset)
    const uint32_t mrow = 8u << log_M;                               // bytes between j1 and j1 + 1
    const uint32_t g4 = (((c4 * 16) << log_M) + jt4) * 8;             // radix-16 round: row m
    uint64_t* __restrict__ l4 = lds + c4 * 16 * ROW + jt4 + (jt4 >> 4);   // + m ROW
    const bool has_scale = p.scale != nullptr, has_post = p.post_scale != nullptr;
    const buf_t scale = make_buf(has_scale ? p.scale + (size_t)blockIdx.z * p.scale_z_stride + base : nullptr);
    const buf_t post = make_buf(has_post ? p.post_scale + (size_t)blockIdx.z * p.post_scale_z_stride + base : nullptr);
    const uint32_t col_end = min(p.n_cols, (blockIdx.y + 1) * p.cols_per_block);
    for (uint32_t col = blockIdx.y * p.cols_per_block; col < col_end; col++) {
        const buf_t src = make_buf(p.src + (size_t)col * p.src_stride + (size_t)blockIdx.z * p.src_z_stride + base);
        const buf_t dst = make_buf(p.dst + (size_t)col * p.dst_stride + (size_t)blockIdx.z * p.dst_z_stride + base);
        uint64_t x[16];
        if (!DIT) {
            // radix-2^REM round at stride 16: item w = (lo, jt), elements j1 = 16 m + lo
#pragma unroll
            for (unsigned it = 0; it < ITEMS; it++) {
                const uint32_t w = tid + 256 * it, jt = w & (T - 1), lo = w >> LOG_T;
                const uint32_t g1 = ((lo << log_M) + jt) * 8;
#pragma unroll
                for (unsigned m = 0; m < G1; m++) x[it * G1 + m] = buf_ld(src, g1, m * 16 * mrow);
            }
            if (has_scale) {
#pragma unroll
                for (unsigned it = 0; it < ITEMS; it++) {
                    const uint32_t w = tid + 256 * it, jt = w & (T - 1), lo = w >> LOG_T;
                    const uint32_t g1 = ((lo << log_M) + jt) * 8;
#pragma unroll
                    for (unsigned m = 0; m < G1; m++) x[it * G1 + m] = mul_f(x[it * G1 + m], buf_ld(scale, g1, m * 16 * mrow));
                }
            }
#pragma unroll
            for (unsigned it = 0; it < ITEMS; it++) {
                const uint32_t w = tid + 256 * it, jt = w & (T - 1), lo = w >> LOG_T;
                uint64_t* __restrict__ l1 = lds + lo * ROW + jt + (jt >> 4);   // + 16 m ROW
                dft_dif<REM, INV>(x + it * G1);
#pragma unroll
                for (unsigned m = 1; m < G1; m++) x[it * G1 + m] = mul_f(x[it * G1 + m], wa[lo * G1 + m]);
#pragma unroll
                for (unsigned m = 0; m < G1; m++) l1[16 * ROW * m] = x[it * G1 + m];
            }
            __syncthreads();
#pragma unroll
            for (int m = 0; m < 16; m++) x[m] = l4[ROW * m];
            dft_dif<4, INV>(x);
#pragma unroll
            for (int m = 0; m < 16; m++) x[m] = mul_f(x[m], F[m]);
            if (p.final_scale != 1) {
#pragma unroll
                for (int m = 0; m < 16; m++) x[m] = mul_f(x[m], p.final_scale);
            }
            if (has_post) {
#pragma unroll
                for (int m = 0; m < 16; m++) x[m] = mul_f(x[m], buf_ld(post, g4, m * mrow));
            }
#pragma unroll
            for (int m = 0; m < 16; m++) buf_st(dst, g4, m * mrow, x[m]);
            __syncthreads();
        } else {
#pragma unroll
            for (int m = 0; m < 16; m++) x[m] = buf_ld(src, g4, m * mrow);
            if (has_scale) {
#pragma unroll
                for (int m = 0; m < 16; m++) x[m] = mul_f(x[m], buf_ld(scale, g4, m * mrow));
            }
#pragma unroll
            for (int m = 0; m < 16; m++) x[m] = mul_f(x[m], F[m]);
            dft_dit<4, INV>(x);
#pragma unroll
            for (int m = 0; m < 16; m++) l4[ROW * m] = x[m];
            __syncthreads();
#pragma unroll
            for (unsigned it = 0; it < ITEMS; it++) {
                const uint32_t w = tid + 256 * it, jt = w & (T - 1), lo = w >> LOG_T;
                const uint64_t* __restrict__ l1 = lds + lo * ROW + jt + (jt >> 4);
#pragma unroll
                for (unsigned m = 0; m < G1; m++) x[it * G1 + m] = l1[16 * ROW * m];
            }
#pragma unroll
            for (unsigned it = 0; it < ITEMS; it++) {
                const uint32_t w = tid + 256 * it, lo = w >> LOG_T;
#pragma unroll
                for (unsigned m = 1; m < G1; m++) x[it * G1 + m] = mul_f(x[it * G1 + m], wa[lo * G1 + m]);
                dft_dit<REM, INV>(x + it * G1);
            }
            if (p.final_scale != 1) {
#pragma unroll
                for (int m = 0; m < 16; m++) x[m] = mul_f(x[m], p.final_scale);
            }
#pragma unroll
            for (unsigned it = 0; it < ITEMS; it++) {
                const uint32_t w = tid + 256 * it, jt = w & (T - 1), lo = w >> LOG_T;
                const uint32_t g1 = ((lo << log_M) + jt) * 8;
                if (has_post) {
#pragma unroll
                    for (unsigned m = 0; m < G1; m++) x[it * G1 + m] = mul_f(x[it * G1 + m], buf_ld(post, g1, m * 16 * mrow));
                }
#pragma unroll
                for (unsigned m = 0; m < G1; m++) buf_st(dst, g1, m * 16 * mrow, x[it * G1 + m]);
            }
            __syncthreads();
        }
    }
}

// Pass planning: contiguous pass of `c` bits, then strided passes of <= STRIDED_BITS_MAX bits.
struct Plan {
    unsigned n_pass;
    unsigned log_A[8];
};
static Plan make_plan(unsigned log_n) {
    Plan pl{};
    unsigned c = log_n < TILE_LOG ? log_n : TILE_LOG;
    unsigned rem = log_n - c;
    unsigned n_strided = (rem + STRIDED_BITS_MAX - 1) / STRIDED_BITS_MAX;
    pl.n_pass = 0;
    pl.log_A[pl.n_pass++] = c;  // index 0 = contiguous pass
    for (unsigned i = 0; i < n_strided; i++) {
        unsigned bits = (rem + (n_strided - i) - 1) / (n_strided - i);  // balanced split
        pl.log_A[pl.n_pass++] = bits;
        rem -= bits;
    }
    return pl;
}

// the constant twiddles of the register sub-transforms are compiled in as powers of two: check them once per process
static bool w16_is_the_compiled_power_of_two() {
    static const bool ok = [] {
        const uint64_t w16 = gl::root_of_unity(4);
        return gl::pow(2, W16_LOG2) == w16 && gl::pow(2, 192 - W16_LOG2) == gl::inv(w16);
    }();
    return ok;
}

// columns a block walks at one tile position: as many as leave the grid a few blocks per CU slot (1024 resident blocks)
static uint32_t pick_cols_per_block(size_t tiles_times_z, uint32_t n_cols) {
    uint32_t cpb = 16;
    while (cpb > 1 && tiles_times_z * ((n_cols + cpb - 1) / cpb) < 4096) cpb >>= 1;
    return cpb < n_cols ? cpb : n_cols;
}

template <bool DIT, bool INV>
static void run_passes_t(hipStream_t st, const NttTables& tb, const uint64_t* src, size_t src_stride, size_t src_z_stride,
                         uint64_t* dst, size_t dst_stride, size_t dst_z_stride, uint32_t n_cols, uint32_t n_z, unsigned log_n,
                         const uint64_t* scale, size_t scale_z_stride, uint64_t final_scale, const uint64_t* post_scale,
                         size_t post_scale_z_stride) {
    if (!w16_is_the_compiled_power_of_two()) abort();   // a build with the wrong NLX_GL_GENERATOR_SET table: never at run time
    Plan pl = make_plan(log_n);
    const uint64_t* const* roots = INV ? tb.inv : tb.fwd;
    // DIF order: strided passes from the largest sub-problem down, contiguous pass last.
    // DIT order: contiguous pass first, strided passes with growing sub-problem.
    // Sub-problem sizes: contiguous N = 2^c; strided pass i (i = 1..) N_i = 2^(c + bits_1 + ... + bits_i).
    unsigned logN_of[8];
    unsigned acc = pl.log_A[0];
    logN_of[0] = acc;
    for (unsigned i = 1; i < pl.n_pass; i++) {
        acc += pl.log_A[i];
        logN_of[i] = acc;
    }
    for (unsigned step = 0; step < pl.n_pass; step++) {
        unsigned i = DIT ? step : (pl.n_pass - 1 - step);
        const bool first = step == 0, last = step + 1 == pl.n_pass;
        PassParams p{};
        p.src = first ? src : dst;
        p.src_stride = first ? src_stride : dst_stride;
        p.src_z_stride = first ? src_z_stride : dst_z_stride;
        p.dst = dst;
        p.dst_stride = dst_stride;
        p.dst_z_stride = dst_z_stride;
        p.log_n = log_n;
        p.log_N = logN_of[i];
        p.log_A = pl.log_A[i];
        p.tw = p.log_N >= 1 ? roots[p.log_N] : nullptr;
        p.scale = first ? scale : nullptr;
        p.scale_z_stride = scale_z_stride;
        p.final_scale = last ? final_scale : 1;
        p.post_scale = last ? post_scale : nullptr;
        p.post_scale_z_stride = post_scale_z_stride;
        p.n_cols = n_cols;
        p.cols_per_block = 1;
        unsigned tiles;
        if (i == 0) {  // contiguous
            p.log_T = 0;
            // pack several sub-problems into a tile when the transform is shorter than the tile
            unsigned log_subs = log_n - p.log_A;  // sub-problems per column
            unsigned q = TILE_LOG - p.log_A;
            p.log_Q = q < log_subs ? q : log_subs;
            tiles = 1u << (log_n - p.log_A - p.log_Q);
        } else {
            p.log_Q = 0;
            unsigned log_M = p.log_N - p.log_A;
            unsigned t = TILE_LOG - p.log_A;
            p.log_T = t < log_M ? t : log_M;
            tiles = 1u << (log_n - p.log_A - p.log_T);
        }
        if (i != 0 && p.log_A >= 1 && p.log_A <= 4) {
            // short strided transform: register-only kernel, one lane per (sub-problem, column)
            const size_t threads = ((size_t)1 << log_n) >> p.log_A;
            const dim3 grid((unsigned)((threads + 255) / 256), n_cols, n_z);
            switch (p.log_A) {
                case 4: hipLaunchKernelGGL((k_ntt_strided_reg<DIT, INV, 4>), grid, dim3(256), 0, st, p); break;
                case 3: hipLaunchKernelGGL((k_ntt_strided_reg<DIT, INV, 3>), grid, dim3(256), 0, st, p); break;
                case 2: hipLaunchKernelGGL((k_ntt_strided_reg<DIT, INV, 2>), grid, dim3(256), 0, st, p); break;
                default: hipLaunchKernelGGL((k_ntt_strided_reg<DIT, INV, 1>), grid, dim3(256), 0, st, p); break;
            }
            continue;
        }
        if (i == 0 && pl.n_pass == 1 && log_n >= 8 && log_n < TILE_LOG && src_stride < ((size_t)1 << 24) && dst_stride < ((size_t)1 << 24)) {
            // short columns (one contiguous pass): tiles of 16 >> (log_n - 8) whole columns; a block walks cols_per_block tiles
            const uint32_t C = 16u >> (log_n - 8), groups = (n_cols + C - 1) / C;
            p.cols_per_block = 1;
            while (p.cols_per_block < 8 && (size_t)((groups + 2 * p.cols_per_block - 1) / (2 * p.cols_per_block)) * n_z >= 2048) p.cols_per_block *= 2;
            const dim3 grid((groups + p.cols_per_block - 1) / p.cols_per_block, 1, n_z);
            switch (log_n) {
                case 8: hipLaunchKernelGGL((k_ntt_cols<DIT, INV, 0>), grid, dim3(256), 0, st, p); break;
                case 9: hipLaunchKernelGGL((k_ntt_cols<DIT, INV, 1>), grid, dim3(256), 0, st, p); break;
                case 10: hipLaunchKernelGGL((k_ntt_cols<DIT, INV, 2>), grid, dim3(256), 0, st, p); break;
                default: hipLaunchKernelGGL((k_ntt_cols<DIT, INV, 3>), grid, dim3(256), 0, st, p); break;
            }
            continue;
        }
        if (log_n >= TILE_LOG && log_n <= 28) {
            // tile kernels (32-bit byte offsets inside a column): the contiguous pass has 12 levels, a strided one 5 .. 8 over tiles of 2^(12 - log_A) columns
            p.cols_per_block = pick_cols_per_block((size_t)tiles * n_z, n_cols);
            const dim3 grid(tiles, (n_cols + p.cols_per_block - 1) / p.cols_per_block, n_z);
            if (i == 0) {
                hipLaunchKernelGGL((k_ntt_c12<DIT, INV>), grid, dim3(256), 0, st, p);
            } else {
                switch (p.log_A) {
                    case 5: hipLaunchKernelGGL((k_ntt_s<DIT, INV, 1>), grid, dim3(256), 0, st, p); break;
                    case 6: hipLaunchKernelGGL((k_ntt_s<DIT, INV, 2>), grid, dim3(256), 0, st, p); break;
                    case 7: hipLaunchKernelGGL((k_ntt_s<DIT, INV, 3>), grid, dim3(256), 0, st, p); break;
                    default: hipLaunchKernelGGL((k_ntt_s<DIT, INV, 4>), grid, dim3(256), 0, st, p); break;
                }
            }
            continue;
        }
        hipLaunchKernelGGL((k_ntt_pass<DIT, INV>), dim3(tiles, n_cols, n_z), dim3(NTT_THREADS), 0, st, p);
    }
}

template <bool DIT>
static void run_passes(hipStream_t st, const NttTables& tb, bool inverse_roots, const uint64_t* src,
                       size_t src_stride, size_t src_z_stride, uint64_t* dst, size_t dst_stride,
                       size_t dst_z_stride, uint32_t n_cols, uint32_t n_z, unsigned log_n, const uint64_t* scale,
                       size_t scale_z_stride, uint64_t final_scale, const uint64_t* post_scale = nullptr,
                       size_t post_scale_z_stride = 0) {
    if (inverse_roots)
        run_passes_t<DIT, true>(st, tb, src, src_stride, src_z_stride, dst, dst_stride, dst_z_stride, n_cols, n_z, log_n, scale,
                                scale_z_stride, final_scale, post_scale, post_scale_z_stride);
    else
        run_passes_t<DIT, false>(st, tb, src, src_stride, src_z_stride, dst, dst_stride, dst_z_stride, n_cols, n_z, log_n, scale,
                                 scale_z_stride, final_scale, post_scale, post_scale_z_stride);
}

void launch_intt_dif(hipStream_t st, const NttTables& tb, const uint64_t* src, size_t src_stride, uint64_t* dst,
                     size_t dst_stride, uint32_t n_cols, unsigned log_n) {
    if (!n_cols) return;
    uint64_t n_inv = gl::inv((uint64_t)1 << log_n);
    run_passes<false>(st, tb, true, src, src_stride, 0, dst, dst_stride, 0, n_cols, 1, log_n, nullptr, 0, n_inv);
}

void launch_lde_dit(hipStream_t st, const NttTables& tb, const uint64_t* coeffs_br, size_t src_stride,
                    uint64_t* dst, size_t dst_stride, uint32_t n_cols, unsigned log_n, unsigned rate_bits,
                    const uint64_t* scale_br) {
    if (!n_cols) return;
    const size_t n = (size_t)1 << log_n;
    run_passes<true>(st, tb, false, coeffs_br, src_stride, 0, dst, dst_stride, n, n_cols, 1u << rate_bits, log_n,
                     scale_br, n, 1);
}

// Per-coset inverse transform of an LDE-shaped table [y][r][k] (in place): natural values ->
// bit-reversed coefficients, scaled by 1/n and by post_scale_br[r][j] (the inverse coset powers).
void launch_intt_dif_cosets(hipStream_t st, const NttTables& tb, uint64_t* data, uint32_t n_y, unsigned log_n,
                            unsigned rate_bits, const uint64_t* post_scale_br) {
    if (!n_y) return;
    const size_t n = (size_t)1 << log_n;
    uint64_t n_inv = gl::inv((uint64_t)1 << log_n);
    run_passes<false>(st, tb, true, data, n << rate_bits, n, data, n << rate_bits, n, n_y, 1u << rate_bits, log_n,
                      nullptr, 0, n_inv, post_scale_br, n);
}

void launch_ntt_dif_fwd(hipStream_t st, const NttTables& tb, uint64_t* data, size_t stride, uint32_t n_cols,
                        unsigned log_n, bool inverse, const uint64_t* prescale_nat) {
    if (!n_cols) return;
    uint64_t fs = inverse ? gl::inv((uint64_t)1 << log_n) : 1;
    run_passes<false>(st, tb, inverse, data, stride, 0, data, stride, 0, n_cols, 1, log_n, prescale_nat, 0, fs);
}

__global__ void k_bitrev_permute(const uint64_t* __restrict__ src, uint64_t* __restrict__ dst, size_t stride,
                                 unsigned log_n, const uint64_t* __restrict__ postscale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >> log_n) return;
    const uint64_t* s = src + (size_t)blockIdx.y * stride;
    uint64_t* d = dst + (size_t)blockIdx.y * stride;
    uint64_t v = s[gl::bitrev32((uint32_t)i, log_n)];
    if (postscale) v = gl::mul(v, postscale[i]);
    d[i] = v;
}
void launch_bitrev_permute(hipStream_t st, const uint64_t* src, uint64_t* dst, size_t stride, uint32_t n_cols,
                           unsigned log_n, const uint64_t* postscale_nat) {
    if (!n_cols) return;
    size_t n = (size_t)1 << log_n;
    hipLaunchKernelGGL(k_bitrev_permute, dim3((unsigned)((n + 255) / 256), n_cols), dim3(256), 0, st, src, dst, stride,
                       log_n, postscale_nat);
}

// The same permutation IN PLACE and coalesced, for log_n >= 2 K: an index is (hi : K bits)(mid : m bits)(lo : K bits) and its
// reversal (rev lo)(rev mid)(rev hi), so the 2^K x 2^K tile `mid` (rows hi, 2^K contiguous elements each) goes, transposed
// with both coordinates bit-reversed, onto tile rev(mid).  A block stages the two tiles of a pair in LDS and writes each
// where the other was: every global access is a 512-byte row.  The gather version above reads 8 bytes per 64-byte line and
// needs a second buffer and a copy back: 6.0 ms of the 14.3 ms of a 16 x 2^24 batch.
constexpr int BR_K = 6, BR_T = 1 << BR_K;
// 16-byte accesses (a lane owns two adjacent elements of a row: 32 lanes per 512-byte row, eight rows per sweep) and every
// load of both tiles in flight before the first LDS write: sixteen independent 16-byte loads per lane.
__global__ __launch_bounds__(256) void k_bitrev_tiled(uint64_t* __restrict__ data, size_t stride, unsigned log_n,
                                                      const uint64_t* __restrict__ postscale) {
    __shared__ uint64_t ta[BR_T][BR_T + 1], tb[BR_T][BR_T + 1];
    const unsigned m = log_n - 2 * BR_K;
    const uint32_t mid = blockIdx.x, rmid = m ? gl::bitrev32(mid, m) : 0;
    if (rmid < mid) return;   // the pair's other block does the work
    uint64_t* d = data + (size_t)blockIdx.y * stride;
    const uint32_t l2 = threadIdx.x & 31, row0 = threadIdx.x >> 5;   // elements 2 l2, 2 l2 + 1 of rows row0 + 8 k
    const bool self = rmid == mid;
    constexpr int SWEEPS = BR_T / 8;
    ulonglong2 va[SWEEPS], vb[SWEEPS];
#pragma unroll
    for (int k = 0; k < SWEEPS; k++) {
        const size_t hi = row0 + 8 * k;
        va[k] = *reinterpret_cast<const ulonglong2*>(d + ((hi << (m + BR_K)) | ((size_t)mid << BR_K) | (2 * l2)));
        if (!self) vb[k] = *reinterpret_cast<const ulonglong2*>(d + ((hi << (m + BR_K)) | ((size_t)rmid << BR_K) | (2 * l2)));
    }
#pragma unroll
    for (int k = 0; k < SWEEPS; k++) {
        const uint32_t hi = row0 + 8 * k;
        ta[hi][2 * l2] = va[k].x; ta[hi][2 * l2 + 1] = va[k].y;
        if (!self) { tb[hi][2 * l2] = vb[k].x; tb[hi][2 * l2 + 1] = vb[k].y; }
    }
    __syncthreads();
    const uint32_t r0 = gl::bitrev32(2 * l2, BR_K);   // rev(2 l2 + 1) = r0 + 32
#pragma unroll
    for (int k = 0; k < SWEEPS; k++) {
        const uint32_t hi = row0 + 8 * k, rhi = gl::bitrev32(hi, BR_K);
        // destination (hi, rmid, lo) takes source (rev lo, mid, rev hi); destination (hi, mid, lo) the same from tile rmid
        const size_t i1 = ((size_t)hi << (m + BR_K)) | ((size_t)rmid << BR_K) | (2 * l2);
        ulonglong2 v = make_ulonglong2(ta[r0][rhi], ta[r0 + BR_T / 2][rhi]);
        if (postscale) { v.x = gl::mul(v.x, postscale[i1]); v.y = gl::mul(v.y, postscale[i1 + 1]); }
        *reinterpret_cast<ulonglong2*>(d + i1) = v;
        if (!self) {
            const size_t i2 = ((size_t)hi << (m + BR_K)) | ((size_t)mid << BR_K) | (2 * l2);
            ulonglong2 w = make_ulonglong2(tb[r0][rhi], tb[r0 + BR_T / 2][rhi]);
            if (postscale) { w.x = gl::mul(w.x, postscale[i2]); w.y = gl::mul(w.y, postscale[i2 + 1]); }
            *reinterpret_cast<ulonglong2*>(d + i2) = w;
        }
    }
}
bool launch_bitrev_inplace(hipStream_t st, uint64_t* data, size_t stride, uint32_t n_cols, unsigned log_n, const uint64_t* postscale_nat) {
    if (log_n < 2 * BR_K || !n_cols) return false;
    hipLaunchKernelGGL(k_bitrev_tiled, dim3(1u << (log_n - 2 * BR_K), n_cols), dim3(256), 0, st, data, stride, log_n, postscale_nat);
    return true;
}

// ---- natural order in, natural order out, no reordering pass (round 4; nlx_ntt_batch, BASELINE.json configs[4]) ----------------
// A decimation in frequency leaves X[k] at position bitrev(k); round 3 ran k_bitrev_tiled afterwards - a quarter of a 16 x 2^24
// call, there only to un-permute what the last pass had just written.  Here the LAST pass writes every value at its natural
// position.  That needs the pass's tile to hold runs of consecutive OUTPUT addresses, and an output address's low bits are the
// reversed HIGH bits of the position: the contiguous pass is cut to 8 levels (rows of 256 elements) and a block takes the
// sixteen rows whose positions differ in their top four bits - position (u : 4)(h' : log_n - 12)(c : 8) goes to address
// (rev8 c)(rev h')(rev4 u), so for every c the sixteen u are one 128-byte line.  Reads: sixteen runs of 2 KB; writes: 256
// lines of 128 bytes per tile; the strided passes above it take 8 levels each instead of 6 (k_ntt_s<4>), the last of them
// writing into a second buffer so that this pass can scatter back into the caller's (no copy).
template <bool INV>
__global__ __launch_bounds__(256, 4) void k_ntt_c8_nat(PassParams p) {
    constexpr unsigned ROW = 273;                         // 256 + one pad per sixteen + 1: rows start on different banks
    __shared__ uint64_t lds[16 * ROW];
    __shared__ uint64_t w2[256];                          // w_256^(lo * brev4(m)) at [lo][m]
    const uint32_t tid = threadIdx.x;
    const unsigned hb = p.log_n - 12, top = p.log_n - 8;
    const uint32_t hp = blockIdx.x, rhp = hb ? gl::bitrev32(hp, hb) : 0;
    w2[tid] = tw_full(p.tw, (tid >> 4) * brev4(tid & 15), 128);
    __syncthreads();
    // round of stride 16: lane (row ua, lo) holds c = 16 m + lo; round of stride 1: lane (chi, u' = rev4 u) holds c = 16 chi + m
    const uint32_t ua = tid >> 4, lo = tid & 15, chi = tid >> 4, up = tid & 15, u2 = brev4(up);
    const uint32_t g1 = (((((ua << hb) | hp) << 8) + lo)) * 8;                     // + 128 m bytes
    const uint32_t g2 = (((brev4(chi) << top) | (rhp << 4) | up)) * 8;              // + (16 brev4(m) << top) * 8 bytes
    uint64_t* __restrict__ l1 = lds + ua * ROW + lo;                              // + 17 m
    const uint64_t* __restrict__ l2 = lds + u2 * ROW + 17 * chi;                   // + m
    const uint64_t* __restrict__ w2l = w2 + lo * 16;
    const bool has_scale = p.scale != nullptr, has_post = p.post_scale != nullptr;
    const buf_t scale = make_buf(p.scale), post = make_buf(p.post_scale);
    const uint32_t col_end = min(p.n_cols, (blockIdx.y + 1) * p.cols_per_block);
    for (uint32_t col = blockIdx.y * p.cols_per_block; col < col_end; col++) {
        const buf_t src = make_buf(p.src + (size_t)col * p.src_stride);
        const buf_t dst = make_buf(p.dst + (size_t)col * p.dst_stride);
        uint64_t x[16];
#pragma unroll
        for (int m = 0; m < 16; m++) x[m] = buf_ld(src, g1, m * 128);
        if (has_scale) {
#pragma unroll
            for (int m = 0; m < 16; m++) x[m] = mul_f(x[m], buf_ld(scale, g1, m * 128));
        }
        dft_dif<4, INV>(x);
#pragma unroll
        for (int m = 1; m < 16; m++) x[m] = mul_f(x[m], w2l[m]);
#pragma unroll
        for (int m = 0; m < 16; m++) l1[17 * m] = x[m];
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 16; m++) x[m] = l2[m];
        dft_dif<4, INV>(x);
        if (p.final_scale != 1) {
#pragma unroll
            for (int m = 0; m < 16; m++) x[m] = mul_f(x[m], p.final_scale);
        }
#pragma unroll
        for (int m = 0; m < 16; m++) {
            const uint32_t so = ((16u * (__brev((uint32_t)m) >> 28)) << top) * 8;   // wave-uniform: the scalar offset
            if (has_post) x[m] = mul_f(x[m], buf_ld(post, g2, so));
            buf_st(dst, g2, so, x[m]);
        }
        __syncthreads();
    }
}

// Natural -> natural transform of n_cols columns of 2^log_n points (18 <= log_n <= 28) through `tmp` (same shape as data).
// Returns false when the size is outside that range (the caller then runs the DIF passes and the reordering kernel).
bool launch_ntt_dif_natural(hipStream_t st, const NttTables& tb, uint64_t* data, uint64_t* tmp, size_t stride, uint32_t n_cols,
                            unsigned log_n, bool inverse, const uint64_t* prescale_nat, const uint64_t* postscale_nat) {
    if (log_n < 18 || log_n > 28 || !n_cols) return false;
    if (!w16_is_the_compiled_power_of_two()) abort();
    const uint64_t* const* roots = inverse ? tb.inv : tb.fwd;
    const unsigned rem = log_n - 8, n_strided = (rem + STRIDED_BITS_MAX - 1) / STRIDED_BITS_MAX;
    unsigned bits[4], logN[4], left = rem, acc = 8;
    for (unsigned i = 0; i < n_strided; i++) {           // i = 0: the pass next to the contiguous one (smallest sub-problem)
        bits[i] = (left + (n_strided - i) - 1) / (n_strided - i);
        left -= bits[i];
        acc += bits[i];
        logN[i] = acc;
    }
    for (unsigned step = 0; step < n_strided; step++) {  // DIF: the largest sub-problem first
        const unsigned i = n_strided - 1 - step;
        const bool first = step == 0, last_strided = step + 1 == n_strided;
        PassParams p{};
        p.src = first ? data : data;
        p.dst = last_strided ? tmp : data;
        p.src_stride = stride;
        p.dst_stride = stride;
        p.log_n = log_n;
        p.log_N = logN[i];
        p.log_A = bits[i];
        p.tw = roots[p.log_N];
        p.scale = first ? prescale_nat : nullptr;
        p.final_scale = 1;
        p.n_cols = n_cols;
        const unsigned log_M = p.log_N - p.log_A, t = TILE_LOG - p.log_A;
        p.log_T = t < log_M ? t : log_M;
        const unsigned tiles = 1u << (log_n - p.log_A - p.log_T);
        p.cols_per_block = pick_cols_per_block(tiles, n_cols);
        const dim3 grid(tiles, (n_cols + p.cols_per_block - 1) / p.cols_per_block, 1);
#define NLX_NAT_S(REM)                                                                                                  \
    do {                                                                                                                \
        if (inverse) hipLaunchKernelGGL((k_ntt_s<false, true, REM>), grid, dim3(256), 0, st, p);                        \
        else hipLaunchKernelGGL((k_ntt_s<false, false, REM>), grid, dim3(256), 0, st, p);                               \
    } while (0)
        switch (p.log_A) {
            case 5: NLX_NAT_S(1); break;
            case 6: NLX_NAT_S(2); break;
            case 7: NLX_NAT_S(3); break;
            default: NLX_NAT_S(4); break;
        }
#undef NLX_NAT_S
    }
    PassParams p{};
    p.src = tmp;
    p.dst = data;
    p.src_stride = stride;
    p.dst_stride = stride;
    p.log_n = log_n;
    p.log_N = 8;
    p.log_A = 8;
    p.tw = roots[8];
    p.final_scale = inverse ? gl::inv((uint64_t)1 << log_n) : 1;
    p.post_scale = postscale_nat;
    p.n_cols = n_cols;
    const unsigned tiles = 1u << (log_n - 12);
    p.cols_per_block = pick_cols_per_block(tiles, n_cols);
    const dim3 grid(tiles, (n_cols + p.cols_per_block - 1) / p.cols_per_block, 1);
    if (inverse) hipLaunchKernelGGL((k_ntt_c8_nat<true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((k_ntt_c8_nat<false>), grid, dim3(256), 0, st, p);
    return true;
}

// One cross-rank level of a decimation in frequency whose input is split over several GPUs (nlx_ntt_split_level): this
// rank's slice against its partner's.  Lower partner: a' = a + b (a = mine, b = theirs).  Upper partner: b' = (a - b) w,
// a = theirs, b = mine, w = w_n^(idx0 + q) 2^level read from the size-n table (idx0 = the slice's offset inside the level's
// half-block).
__global__ __launch_bounds__(256) void k_ntt_split_level(uint64_t* __restrict__ mine, const uint64_t* __restrict__ theirs, size_t m,
                                                         int upper, const uint64_t* __restrict__ w_n_table, size_t idx0, unsigned level) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= m) return;
    const size_t at = (size_t)blockIdx.y * m + q;
    const uint64_t x = mine[at], y = theirs[at];
    mine[at] = upper ? gl::mul(gl::sub(y, x), w_n_table[(idx0 + q) << level]) : gl::add(x, y);
}
void launch_ntt_split_level(hipStream_t st, uint64_t* mine, const uint64_t* theirs, size_t m, uint32_t n_cols, bool upper,
                            const uint64_t* w_n_table, size_t idx0, unsigned level) {
    if (!n_cols || !m) return;
    hipLaunchKernelGGL(k_ntt_split_level, dim3((unsigned)((m + 255) / 256), n_cols), dim3(256), 0, st, mine, theirs, m, upper ? 1 : 0,
                       w_n_table, idx0, level);
}

// table[e] = root^e for e in [0, 2^log_size)
__global__ void k_fill_powers(uint64_t* __restrict__ table, size_t count, uint64_t base, uint64_t first) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    table[i] = gl::mul(first, gl::pow(base, i));
}
void launch_fill_powers(hipStream_t st, uint64_t* d_table, size_t count, uint64_t base, uint64_t first) {
    if (!count) return;
    hipLaunchKernelGGL(k_fill_powers, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, d_table, count, base,
                       first);
}
// scale_br[r][j] = (shift * w_L^r)^bitrev_n(j), L = n << rate_bits
__global__ void k_fill_coset_scale_br(uint64_t* __restrict__ table, unsigned log_n, unsigned rate_bits,
                                      uint64_t shift, uint64_t w_L) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t n = (size_t)1 << log_n;
    if (i >= (n << rate_bits)) return;
    uint32_t r = (uint32_t)(i >> log_n), j = (uint32_t)(i & (n - 1));
    uint64_t base = gl::mul(shift, gl::pow(w_L, r));
    table[i] = gl::pow(base, gl::bitrev32(j, log_n));
}
void launch_fill_coset_scale_br(hipStream_t st, uint64_t* d_table, unsigned log_n, unsigned rate_bits,
                                uint64_t shift, bool inverse) {
    size_t total = (size_t)1 << (log_n + rate_bits);
    uint64_t w_L = gl::root_of_unity(log_n + rate_bits);
    if (inverse) {
        shift = gl::inv(shift);
        w_L = gl::inv(w_L);
    }
    hipLaunchKernelGGL(k_fill_coset_scale_br, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_table, log_n,
                       rate_bits, shift, w_L);
}

}  // namespace nlx
