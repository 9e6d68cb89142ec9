// BN254 scalar field (Fr) arithmetic and NTT for gfx950 - the first piece of SURVEY.md §8 row f.4 (the recursive wrap:
// plonky2x's BN128-Poseidon wrapper -> gnark Groth16 / PLONK over BN254; BASELINE.json configs[4] names its 2^24-point NTT).
// The wrap is not in /root/reference (succinct.json:7-8 only names the platform entry point that can run it; gnark is
// Go, un-vendored), so this follows the published definitions:
//   r = 21888242871839275222246405745257275088548364400416034343698204186575808495617 (2-adicity 28),
//   the 2^28-th root of unity gnark-crypto's ecc/bn254/fr/fft uses
//   (19103219067921713944291392827692070036145651957329286315305642004821462161904 = 5^((r-1)/2^28)),
//   elements in Montgomery form (R = 2^256) as four little-endian 64-bit words - gnark-crypto's fr.Element layout -
//   so a Go caller's []fr.Element can be handed over as it lies in memory (nlx.h: nlx_bn254_ntt_batch).
// Replaces gnark-crypto fft.Domain.FFT / FFTInverse (DIF, natural order out after the bit-reversal the caller would do).
//
// Multiplication: the butterflies compute on nine 29-bit limbs (bn254_f29.hpp, modulus r: Montgomery form with R' = 2^261,
// one v_mad_u64_u32 per partial product into fixed 64-bit columns, loose values with stated bounds) - the CIOS product on
// eight 32-bit limbs of the first version cost ~800 instructions, most of them register moves (gfx950 wants the 64-bit
// operands of the multiply-add in even-aligned pairs and a CIOS accumulator shifts by one limb per row).  Elements rest
// in memory as 32 bytes in every pass (values below 2^256, re-sliced on load and store); the first pass multiplies by the
// constant that takes the caller's form (fr.Element or canonical) into the kernels' form, the reordering pass by the one
// that takes it back (with the 1/n).  The eight-limb code below remains for the host (twiddle tables, constants).  A
// 256-bit product is ~160 multiply-adds against Goldilocks' 4, so unlike the Goldilocks transforms this one is bound by
// the integer-VALU issue rate outright; the passes fuse three butterfly levels per trip through HBM (radix 8 in registers).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <vector>
#include "bn254_f29.hpp"
#include "ctx.hpp"
#include "../../include/nlx.h"

namespace nlx {
namespace bn {

struct Fr {
    uint32_t v[8];
};

#define BN_HD __host__ __device__ __forceinline__

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __constant__ static const uint32_t D_MOD[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                                                         0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
#endif
static const uint32_t H_MOD[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
static const uint32_t H_ONE[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};  // R mod r
static const uint32_t H_R2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};   // R^2 mod r
static const uint32_t H_ROOT28[8] = {0x80d13d9cu, 0x636e7355u, 0x2445ffd6u, 0xa22bf374u, 0x1eb203d8u, 0x56452ac0u, 0x2963f9e7u, 0x1860ef94u};  // w_{2^28} R mod r
constexpr uint32_t N0INV = 0xefffffffu;  // -r^-1 mod 2^32

BN_HD const uint32_t* modulus() {
#if defined(__HIP_DEVICE_COMPILE__)
    return D_MOD;
#else
    return H_MOD;
#endif
}

// a >= b as 256-bit integers
BN_HD bool geq(const Fr& a, const uint32_t* b) {
#pragma unroll
    for (int i = 7; i >= 0; i--) {
        if (a.v[i] != b[i]) return a.v[i] > b[i];
    }
    return true;
}
BN_HD void sub_mod_raw(Fr& a, const uint32_t* b) {  // a -= b (no borrow out expected)
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint64_t d = (uint64_t)a.v[i] - b[i] - borrow;
        a.v[i] = (uint32_t)d;
        borrow = (d >> 32) & 1;
    }
}
BN_HD Fr add(const Fr& a, const Fr& b) {
    Fr r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)a.v[i] + b.v[i];
        r.v[i] = (uint32_t)c;
        c >>= 32;
    }
    // r < 2 * modulus < 2^255: no carry out of 256 bits
    if (geq(r, modulus())) sub_mod_raw(r, modulus());
    return r;
}
BN_HD Fr sub(const Fr& a, const Fr& b) {
    Fr r;
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint64_t d = (uint64_t)a.v[i] - b.v[i] - borrow;
        r.v[i] = (uint32_t)d;
        borrow = (d >> 32) & 1;
    }
    if (borrow) {
        uint64_t c = 0;
        const uint32_t* m = modulus();
#pragma unroll
        for (int i = 0; i < 8; i++) {
            c += (uint64_t)r.v[i] + m[i];
            r.v[i] = (uint32_t)c;
            c >>= 32;
        }
    }
    return r;
}

// Montgomery product a b R^-1 mod r (CIOS, 32-bit limbs): t has nine limbs; every inner step is one 32 x 32 + 64 multiply-add
BN_HD Fr mul(const Fr& a, const Fr& b) {
    const uint32_t* m = modulus();
    uint32_t t[10];
#pragma unroll
    for (int i = 0; i < 10; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            c += (uint64_t)a.v[j] * b.v[i] + t[j];
            t[j] = (uint32_t)c;
            c >>= 32;
        }
        c += t[8];
        t[8] = (uint32_t)c;
        t[9] = (uint32_t)(c >> 32);
        const uint32_t q = t[0] * N0INV;
        c = (uint64_t)q * m[0] + t[0];
        c >>= 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            c += (uint64_t)q * m[j] + t[j];
            t[j - 1] = (uint32_t)c;
            c >>= 32;
        }
        c += t[8];
        t[7] = (uint32_t)c;
        t[8] = t[9] + (uint32_t)(c >> 32);
    }
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = t[i];
    if (t[8] || geq(r, m)) sub_mod_raw(r, m);
    return r;
}

inline Fr from_limbs(const uint32_t* l) {
    Fr r;
    memcpy(r.v, l, 32);
    return r;
}
inline Fr h_pow(Fr b, uint64_t e) {  // host: Montgomery-form power
    Fr r = from_limbs(H_ONE);
    while (e) {
        if (e & 1) r = mul(r, b);
        b = mul(b, b);
        e >>= 1;
    }
    return r;
}
inline Fr h_inv(const Fr& a) {  // a^(r-2), exponent as 256-bit
    Fr e = from_limbs(H_MOD);
    e.v[0] -= 2;  // r - 2 (the low limb 0xf0000001 does not borrow)
    Fr r = from_limbs(H_ONE), b = a;
    for (int i = 0; i < 256; i++) {
        if ((e.v[i / 32] >> (i % 32)) & 1) r = mul(r, b);
        b = mul(b, b);
    }
    return r;
}

// ---- kernels ----
// data layout: column-major, element = 8 little-endian 32-bit limbs = gnark-crypto's [4]uint64; 128-bit loads / stores
__device__ __forceinline__ Fr load(const Fr* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    const uint4 a = q[0], b = q[1];
    Fr r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}
__device__ __forceinline__ void store(Fr* p, const Fr& r) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
    q[1] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
}

// ---- the kernels' element: nine 29-bit limbs, modulus r (bn254_f29.hpp) ----
using f29::Fe;
typedef f29::RMod RM;
static const uint32_t H_K261[8] = {0x8fffff57u, 0x2fd4e156u, 0xa494b01au, 0x75bba827u, 0x819caa80u, 0x5301fa84u, 0x563d4475u, 0x0dc83629u};  // 2^261 mod r (plain)
static const uint32_t H_C522[f29::NL] = {0x05b69bd4u, 0x06170a5au, 0x020cddceu, 0x1db6310bu, 0x0e54d0ffu, 0x1cf855e3u, 0x1c15e103u, 0x07d09161u, 0x000a054au};  // 2^522 mod r
__device__ __forceinline__ Fe load29(const Fr* p) {
    const Fr r = load(p);
    return f29::from_words256(r.v);
}
__device__ __forceinline__ void store29(Fr* p, const Fe& a) {   // a < 2^256
    Fr r;
    f29::to_words256(a, r.v);
    store(p, r);
}
// What a pass multiplies element e of a column by: a constant (k), or - for a transform on a coset - the e-th power of the
// shift times that constant, from a two-level table: lo[e & 4095] * hi[e >> 12] (32-byte words; hi[0] is the Montgomery
// one, so entries below 4096 need no product).  The first pass uses it to take the caller's form into the kernels' (and
// the coefficients onto the coset), the last one to take it back (with the 1/n and the inverse powers).
struct Scale {
    Fe k;
    const Fr* lo;
    const Fr* hi;
    int on;
    int rev_bits;   // > 0: the data is in bit-reversed order there - the power's exponent is the reversal of the position
};
__device__ __forceinline__ Fe scaled(const Fe& x, const Scale& sc, size_t e) {
    if (!sc.lo) return f29::mul<RM>(x, sc.k);
    if (sc.rev_bits) e = __brevll(e) >> (64 - sc.rev_bits);
    Fe f = load29(sc.lo + (e & 4095u));
    if (e >> 12) f = f29::mul<RM>(f, load29(sc.hi + (e >> 12)));
    return f29::mul<RM>(x, f);
}
typedef Scale InScale;

// T[e] = lo[e & 4095] * hi[e >> 12]: the full table w_n^e, e < n/2, built once per (size, direction) on the device from
// two small host-made tables.  A two-level table read inside the butterflies would cost a second 256-bit product per
// butterfly - as much as the butterfly itself (measured: 97 ms against this version's time at 16 x 2^24).
// The table holds the canonical integers w^e 2^261 mod r (the kernels' Montgomery form) as 32-byte words.
__global__ __launch_bounds__(256) void k_bn_fill_twiddles(Fr* __restrict__ out, size_t count, const Fr* __restrict__ lo,
                                                          const Fr* __restrict__ hi, Fr k261) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count) return;
    const Fr a = load(lo + (e & 4095u));
    const Fr w = e < 4096u ? a : mul(a, load(hi + (e >> 12)));   // w^e 2^256 (eight-limb Montgomery form)
    store(out + e, mul(w, k261));                                  // (w^e 2^256) (2^261) / 2^256
}

__device__ __forceinline__ Fe twiddle(const Fr* __restrict__ table, const Fr* __restrict__, uint32_t e) { return load29(table + e); }

// One DIF level on a thread's register tile: partners are HALF apart (in units of h_last); HALF is a template parameter so
// that every x[] index is static and the tile stays in registers (a runtime `half` put the whole tile into scratch memory:
// the first version ran at a third of this one's speed).
// Bounds: values enter a pass below 2^255; a sum is tightened (< 1.1 r) except on the first fused level (< 2^256 there, which
// a - b + 8 r still accepts as b); differences go through a product (< 2^255) or are tightened.
template <int R, int HALF, bool TIGHT>
__device__ __forceinline__ void dif_level(Fe (&x)[R], size_t h_last, size_t off, unsigned shift, const Fr* __restrict__ tw_lo,
                                          const Fr* __restrict__ tw_hi) {
#pragma unroll
    for (int k = 0; k < R; k++) {
        if ((k & HALF) == 0) {
            // position of x[k] inside its 2h block: (k mod 2 HALF) * h_last + off, and k mod 2 HALF < HALF here
            const uint32_t idx = (uint32_t)((size_t)(k & (HALF - 1)) * h_last + off);
            const uint32_t e = idx << shift;   // w_{2h}^idx = w_n^(idx n / 2h)
            const Fe a = x[k], b = x[k + HALF];
            x[k] = TIGHT ? f29::tighten<RM>(f29::add(a, b)) : f29::add(a, b);
            const Fe d = f29::sub<8, RM>(a, b);
            x[k + HALF] = e ? f29::mul<RM>(d, twiddle(tw_lo, tw_hi, e)) : f29::tighten<RM>(d);
        }
    }
}

// LEVELS (1..3) consecutive DIF levels per trip through HBM.  Level with half-size h pairs (i, i + h): a' = a + b,
// b' = (a - b) w_{2h}^(i mod h), w_{2h} = w_n^(n / 2h).  A thread owns the 2^LEVELS elements base + k * h_last
// (h_last = the half-size of the last fused level), i.e. one butterfly network of the fused levels.
template <int LEVELS>
__global__ __launch_bounds__(256) void k_bn_dif(Fr* __restrict__ data, unsigned log_n, unsigned log_h_first,
                                                const Fr* __restrict__ tw_lo, const Fr* __restrict__ tw_hi, uint32_t n_cols, InScale in,
                                                Scale out) {
    constexpr int R = 1 << LEVELS;
    const size_t n = (size_t)1 << log_n;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // butterfly-network index within a column
    const size_t per_col = n >> LEVELS;
    if (t >= per_col * n_cols) return;
    const size_t col = t / per_col, u = t % per_col;
    const unsigned log_h_last = log_h_first - (LEVELS - 1);
    const size_t h_last = (size_t)1 << log_h_last;
    // u = (block index among groups of 2 h_first) * h_last + offset below h_last
    const size_t off = u & (h_last - 1), grp = u >> log_h_last;
    Fr* base = data + col * n + (grp << (log_h_first + 1)) + off;
    Fe x[R];
    const size_t e0 = (grp << (log_h_first + 1)) + off;   // index of x[0] within its column
#pragma unroll
    for (int k = 0; k < R; k++) {
        x[k] = load29(base + (size_t)k * h_last);
        if (in.on) x[k] = scaled(x[k], in, e0 + (size_t)k * h_last);
    }
    // level l has half-size 2^(log_h_first - l): twiddle exponent shift = log_n - 1 - (log_h_first - l)
    const unsigned sh0 = log_n - 1 - log_h_first;
    if constexpr (LEVELS == 3) {
        // (unreachable: three levels run in k_bn_dif3, whose tile is eight named registers - the compiler kept this
        // array form in scratch memory even with static indices)
        dif_level<R, 4, false>(x, h_last, off, sh0, tw_lo, tw_hi);
        dif_level<R, 2, true>(x, h_last, off, sh0 + 1, tw_lo, tw_hi);
        dif_level<R, 1, true>(x, h_last, off, sh0 + 2, tw_lo, tw_hi);
    } else if constexpr (LEVELS == 2) {
        dif_level<R, 2, false>(x, h_last, off, sh0, tw_lo, tw_hi);
        dif_level<R, 1, true>(x, h_last, off, sh0 + 1, tw_lo, tw_hi);
    } else {
        dif_level<R, 1, true>(x, h_last, off, sh0, tw_lo, tw_hi);
    }
#pragma unroll
    for (int k = 0; k < R; k++) {
        if (out.on) x[k] = f29::canonical<RM>(scaled(x[k], out, e0 + (size_t)k * h_last));   // the last pass also leaves the kernels' form
        store29(base + (size_t)k * h_last, x[k]);
    }
}

// Three fused levels with the tile in eight NAMED registers (radix 8).
#define BN_BF(A, B, IDX, SH, TIGHT)                                                       \
    {                                                                                     \
        const uint32_t e_ = (uint32_t)(IDX) << (SH);                                      \
        const Fe a_ = A, b_ = B;                                                          \
        A = TIGHT ? f29::tighten<RM>(f29::add(a_, b_)) : f29::add(a_, b_);                \
        const Fe d_ = f29::sub<8, RM>(a_, b_);                                            \
        B = e_ ? f29::mul<RM>(d_, twiddle(tw_lo, tw_hi, e_)) : f29::tighten<RM>(d_);      \
    }
#ifndef NLX_BN_MINW
#define NLX_BN_MINW 2   // waves per SIMD the radix-8 kernels' register allocation must allow (tuning builds: NLX_EXTRA_FLAGS)
#endif
__global__ __launch_bounds__(256, NLX_BN_MINW) void k_bn_dif3(Fr* __restrict__ data, unsigned log_n, unsigned log_h_first,
                                                 const Fr* __restrict__ tw_lo, const Fr* __restrict__ tw_hi, uint32_t n_cols, InScale in,
                                                 Scale out) {
    const size_t n = (size_t)1 << log_n;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t per_col = n >> 3;
    if (t >= per_col * n_cols) return;
    const size_t col = t / per_col, u = t % per_col;
    const unsigned log_h_last = log_h_first - 2;
    const size_t h = (size_t)1 << log_h_last;
    const size_t off = u & (h - 1), grp = u >> log_h_last;
    Fr* base = data + col * n + (grp << (log_h_first + 1)) + off;
    Fe x0 = load29(base), x1 = load29(base + h), x2 = load29(base + 2 * h), x3 = load29(base + 3 * h), x4 = load29(base + 4 * h),
       x5 = load29(base + 5 * h), x6 = load29(base + 6 * h), x7 = load29(base + 7 * h);
    const size_t e0 = (grp << (log_h_first + 1)) + off;   // index of x0 within its column
    if (in.on) {
        x0 = scaled(x0, in, e0); x1 = scaled(x1, in, e0 + h); x2 = scaled(x2, in, e0 + 2 * h); x3 = scaled(x3, in, e0 + 3 * h);
        x4 = scaled(x4, in, e0 + 4 * h); x5 = scaled(x5, in, e0 + 5 * h); x6 = scaled(x6, in, e0 + 6 * h); x7 = scaled(x7, in, e0 + 7 * h);
    }
    const unsigned s0 = log_n - 1 - log_h_first;
    BN_BF(x0, x4, off, s0, false) BN_BF(x1, x5, h + off, s0, false) BN_BF(x2, x6, 2 * h + off, s0, false) BN_BF(x3, x7, 3 * h + off, s0, false)
    BN_BF(x0, x2, off, s0 + 1, true) BN_BF(x1, x3, h + off, s0 + 1, true) BN_BF(x4, x6, off, s0 + 1, true) BN_BF(x5, x7, h + off, s0 + 1, true)
    BN_BF(x0, x1, off, s0 + 2, true) BN_BF(x2, x3, off, s0 + 2, true) BN_BF(x4, x5, off, s0 + 2, true) BN_BF(x6, x7, off, s0 + 2, true)
    if (out.on) {   // the last pass also leaves the kernels' form
        x0 = f29::canonical<RM>(scaled(x0, out, e0)); x1 = f29::canonical<RM>(scaled(x1, out, e0 + h));
        x2 = f29::canonical<RM>(scaled(x2, out, e0 + 2 * h)); x3 = f29::canonical<RM>(scaled(x3, out, e0 + 3 * h));
        x4 = f29::canonical<RM>(scaled(x4, out, e0 + 4 * h)); x5 = f29::canonical<RM>(scaled(x5, out, e0 + 5 * h));
        x6 = f29::canonical<RM>(scaled(x6, out, e0 + 6 * h)); x7 = f29::canonical<RM>(scaled(x7, out, e0 + 7 * h));
    }
    store29(base, x0); store29(base + h, x1); store29(base + 2 * h, x2); store29(base + 3 * h, x3);
    store29(base + 4 * h, x4); store29(base + 5 * h, x5); store29(base + 6 * h, x6); store29(base + 7 * h, x7);
}
#undef BN_BF

// ---- decimation in time: bit-reversed order in, natural order out (gnark-crypto's fft.DIT).  Level with half-size h pairs
// (i, i + h): a' = a + w b, b' = a - w b, w = w_{2h}^(i mod h); the half-sizes grow from 1 to n / 2, LEVELS per pass.  A
// thread owns the 2^LEVELS elements base + k h0 (h0 = the half-size of the pass's first level).  Bounds: w b is a product
// (< 2^255; for w = 1, b tightened), so with a + 4 r - w b a value grows by less than 2^255.6 per level: below 2^257.5
// after three, inside what a product and `tighten` accept; every stored value is tightened. ----
#define BN_BT(A, B, IDX, SH)                                                              \
    {                                                                                     \
        const uint32_t e_ = (uint32_t)(IDX) << (SH);                                      \
        const Fe a_ = A;                                                                  \
        const Fe t_ = e_ ? f29::mul<RM>(B, twiddle(tw_lo, tw_hi, e_)) : f29::tighten<RM>(B); \
        A = f29::add(a_, t_);                                                             \
        B = f29::sub<4, RM>(a_, t_);                                                      \
    }
template <int LEVELS>
__global__ __launch_bounds__(256) void k_bn_dit(Fr* __restrict__ data, unsigned log_n, unsigned log_h0,
                                                const Fr* __restrict__ tw_lo, const Fr* __restrict__ tw_hi, uint32_t n_cols, InScale in,
                                                Scale out) {
    const size_t n = (size_t)1 << log_n;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t per_col = n >> LEVELS;
    if (t >= per_col * n_cols) return;
    const size_t col = t / per_col, u = t % per_col;
    const size_t h = (size_t)1 << log_h0;
    const size_t off = u & (h - 1), grp = u >> log_h0;
    const size_t e0 = (grp << (log_h0 + LEVELS)) + off;   // index of x0 within its column
    Fr* base = data + col * n + e0;
    const unsigned s0 = log_n - 1 - log_h0;
    Fe x0 = load29(base), x1 = load29(base + h), x2, x3, x4, x5, x6, x7;
    if constexpr (LEVELS >= 2) { x2 = load29(base + 2 * h); x3 = load29(base + 3 * h); }
    if constexpr (LEVELS >= 3) { x4 = load29(base + 4 * h); x5 = load29(base + 5 * h); x6 = load29(base + 6 * h); x7 = load29(base + 7 * h); }
    if (in.on) {
        x0 = scaled(x0, in, e0); x1 = scaled(x1, in, e0 + h);
        if constexpr (LEVELS >= 2) { x2 = scaled(x2, in, e0 + 2 * h); x3 = scaled(x3, in, e0 + 3 * h); }
        if constexpr (LEVELS >= 3) { x4 = scaled(x4, in, e0 + 4 * h); x5 = scaled(x5, in, e0 + 5 * h); x6 = scaled(x6, in, e0 + 6 * h); x7 = scaled(x7, in, e0 + 7 * h); }
    }
    BN_BT(x0, x1, off, s0)
    if constexpr (LEVELS >= 2) {
        BN_BT(x2, x3, off, s0)
        BN_BT(x0, x2, off, s0 - 1) BN_BT(x1, x3, h + off, s0 - 1)
    }
    if constexpr (LEVELS >= 3) {
        BN_BT(x4, x5, off, s0) BN_BT(x6, x7, off, s0)
        BN_BT(x4, x6, off, s0 - 1) BN_BT(x5, x7, h + off, s0 - 1)
        BN_BT(x0, x4, off, s0 - 2) BN_BT(x1, x5, h + off, s0 - 2) BN_BT(x2, x6, 2 * h + off, s0 - 2) BN_BT(x3, x7, 3 * h + off, s0 - 2)
    }
    auto fin = [&](const Fe& x, size_t e) { return out.on ? f29::canonical<RM>(scaled(x, out, e)) : f29::tighten<RM>(x); };
    store29(base, fin(x0, e0)); store29(base + h, fin(x1, e0 + h));
    if constexpr (LEVELS >= 2) { store29(base + 2 * h, fin(x2, e0 + 2 * h)); store29(base + 3 * h, fin(x3, e0 + 3 * h)); }
    if constexpr (LEVELS >= 3) {
        store29(base + 4 * h, fin(x4, e0 + 4 * h)); store29(base + 5 * h, fin(x5, e0 + 5 * h));
        store29(base + 6 * h, fin(x6, e0 + 6 * h)); store29(base + 7 * h, fin(x7, e0 + 7 * h));
    }
}
#undef BN_BT

// data[bitrev(i)] <- data[i] * k, canonical, IN PLACE: the thread of the smaller index of each pair (i, bitrev(i)) swaps the
// two (the bit-reversal back to natural order, fused with the product that takes the kernels' form back to the caller's -
// and carries the 1/n of the inverse transform).  The first version wrote a second buffer and copied it back: twice the
// traffic and 8.6 GB of scratch at 16 x 2^24.
__global__ __launch_bounds__(256) void k_bn_bitrev(Fr* __restrict__ data, unsigned log_n, uint32_t n_cols, Scale k) {
    const size_t n = (size_t)1 << log_n;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * n_cols) return;
    const size_t col = t >> log_n, i = t & (n - 1);
    const size_t j = log_n ? (__brevll(i) >> (64 - log_n)) : 0;
    if (j < i) return;
    Fr* base = data + col * n;
    const Fe a = f29::canonical<RM>(scaled(load29(base + i), k, j));   // position i holds natural index j = bitrev(i)
    if (j == i) {
        store29(base + i, a);
        return;
    }
    const Fe b = f29::canonical<RM>(scaled(load29(base + j), k, i));
    store29(base + j, a);
    store29(base + i, b);
}

// The same, coalesced, for log_n >= 10: an index is (hi : 5 bits)(mid)(lo : 5 bits) and its reversal (rev lo)(rev mid)(rev hi),
// so the 32 x 32 tile `mid` (rows hi, 32 contiguous elements = 1 KB each) goes, transposed with both coordinates
// bit-reversed, onto tile rev(mid).  A block stages the two tiles of a pair in LDS and writes each where the other was
// (the pairwise kernel above touches 32 bytes per access at bit-reversed addresses: 11.5 ms at 16 x 2^24).
constexpr int BR_K = 5, BR_T = 1 << BR_K;
__global__ __launch_bounds__(256) void k_bn_bitrev_tiled(Fr* __restrict__ data, unsigned log_n, Scale k) {
    __shared__ Fr ta[BR_T][BR_T + 1], tb[BR_T][BR_T + 1];
    const unsigned m = log_n - 2 * BR_K;
    const uint32_t mid = blockIdx.x, rmid = m ? (uint32_t)(__brev(mid) >> (32 - m)) : 0;
    if (rmid < mid) return;   // the pair's other block does the work
    Fr* d = data + ((size_t)blockIdx.y << log_n);
    const uint32_t lo = threadIdx.x & (BR_T - 1), row0 = threadIdx.x >> BR_K;   // 8 rows per sweep
    const bool self = rmid == mid;
    for (uint32_t hi = row0; hi < BR_T; hi += 256 / BR_T) {
        ta[hi][lo] = load(d + (((size_t)hi << (m + BR_K)) | ((size_t)mid << BR_K) | lo));
        if (!self) tb[hi][lo] = load(d + (((size_t)hi << (m + BR_K)) | ((size_t)rmid << BR_K) | lo));
    }
    __syncthreads();
    const uint32_t rlo = __brev(lo) >> (32 - BR_K);
    for (uint32_t hi = row0; hi < BR_T; hi += 256 / BR_T) {
        const uint32_t rhi = __brev(hi) >> (32 - BR_K);
        // destination (hi, rmid, lo) - a natural index - takes source (rev lo, mid, rev hi); (hi, mid, lo) the same from tile rmid
        const size_t i1 = ((size_t)hi << (m + BR_K)) | ((size_t)rmid << BR_K) | lo;
        store29(d + i1, f29::canonical<RM>(scaled(f29::from_words256(ta[rlo][rhi].v), k, i1)));
        if (!self) {
            const size_t i2 = ((size_t)hi << (m + BR_K)) | ((size_t)mid << BR_K) | lo;
            store29(d + i2, f29::canonical<RM>(scaled(f29::from_words256(tb[rlo][rhi].v), k, i2)));
        }
    }
}

}  // namespace bn
}  // namespace nlx

using namespace nlx;
using nlx::bn::Fr;

extern "C" {

int32_t nlx_bn254_ntt_batch_coset(nlx_ctx* ctx, uint64_t* cols, size_t n_cols, uint32_t log_n, int inverse, uint32_t flags,
                                  const uint64_t* coset_shift) NLX_TRY {
    if (!ctx) return NLX_E_INVAL;
    if (n_cols == 0) return NLX_OK;
    if (!cols) return ctx->fail(NLX_E_INVAL, "cols is NULL");
    if (log_n > 28) return ctx->fail(NLX_E_RANGE, "BN254 Fr has 2-adicity 28");
    if (n_cols > 65535 || (flags & ~(NLX_BN254_MONTGOMERY | NLX_BN254_BITREV_OUT | NLX_BN254_BITREV_IN))) return ctx->fail(NLX_E_RANGE, "n_cols > 65535 or unknown flag");
    if ((flags & NLX_BN254_BITREV_OUT) && (flags & NLX_BN254_BITREV_IN)) return ctx->fail(NLX_E_UNSUPPORTED, "bit-reversed order on both sides: no decimation produces it");
    if (log_n == 0) return NLX_OK;   // the transform of one point is that point, on any coset
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    const size_t n = (size_t)1 << log_n, count = n * n_cols;
    const bool mont_io = flags & NLX_BN254_MONTGOMERY;
    // twiddle tables for w = w_n (or its inverse): lo[i] = w^i, hi[i] = w^(4096 i), Montgomery form
    // (the table lives in the context, like the Goldilocks ones, and goes with it: 32 n / 2 bytes, 268 MB at n = 2^24)
    struct { Fr* d_lo; Fr* d_hi; } tb;
    void*& slot = ctx->bn254_tables[log_n * 2 + (inverse ? 1 : 0)];
    tb.d_lo = (Fr*)slot;
    tb.d_hi = nullptr;
    if (!tb.d_lo) {
        Fr w = bn::h_pow(bn::from_limbs(bn::H_ROOT28), (uint64_t)1 << (28 - log_n));
        if (inverse) w = bn::h_inv(w);
        const size_t half = std::max<size_t>(n >> 1, 1), n_hi = std::max<size_t>((half + 4095) >> 12, 1);
        std::vector<Fr> lo(4096), hi(n_hi);
        lo[0] = bn::from_limbs(bn::H_ONE);
        for (size_t i = 1; i < 4096; i++) lo[i] = bn::mul(lo[i - 1], w);
        const Fr w4096 = bn::mul(lo[4095], w);
        hi[0] = lo[0];
        for (size_t i = 1; i < n_hi; i++) hi[i] = bn::mul(hi[i - 1], w4096);
        Fr* d_lo = (Fr*)ctx->alloc(4096 * sizeof(Fr));
        Fr* d_hi = (Fr*)ctx->alloc(n_hi * sizeof(Fr));
        tb.d_lo = (Fr*)ctx->alloc(half * sizeof(Fr));
        // a table that could not be built completely is not kept: every block goes back to the context's allocator
        auto drop = [&]() {
            ctx->release(d_lo);
            ctx->release(d_hi);
            ctx->release(tb.d_lo);
            tb.d_lo = nullptr;
        };
        if (!d_lo || !d_hi || !tb.d_lo) {
            drop();
            return ctx->fail(NLX_E_NOMEM, "BN254 twiddle table for 2^%u points (%zu bytes)", log_n, half * sizeof(Fr));
        }
        hipError_t te = hipMemcpyAsync(d_lo, lo.data(), 4096 * sizeof(Fr), hipMemcpyHostToDevice, st);
        if (te == hipSuccess) te = hipMemcpyAsync(d_hi, hi.data(), n_hi * sizeof(Fr), hipMemcpyHostToDevice, st);
        if (te == hipSuccess) {
            hipLaunchKernelGGL(bn::k_bn_fill_twiddles, dim3((unsigned)((half + 255) / 256)), dim3(256), 0, st, tb.d_lo, half, d_lo, d_hi,
                               bn::from_limbs(bn::H_K261));
            te = hipStreamSynchronize(st);
        }
        if (te != hipSuccess) {
            drop();
            return ctx->hip_fail(te, "BN254 twiddle table");
        }
        ctx->release(d_lo);
        ctx->release(d_hi);
        slot = tb.d_lo;
    }
    nlx::Staged s(ctx, cols, count * 32, true, true);
    if (s.status) return s.status;
    Fr* d = s.as<Fr>();
    const unsigned blocks1 = (unsigned)((count + 255) / 256);
    // DIF: natural order in, bit-reversed out; three levels per pass while they last
    ctx->begin_kernel("bn254_ntt_transform", 64.0 * count);  // algorithmic bytes: 32 B read + 32 B written per element
    // the first pass also takes the caller's form into the kernels' (x 2^261 mod r): fr.Element words (x 2^256) times 2^266 /
    // 2^261, canonical integers times 2^522 / 2^261
    bn::InScale first{};
    first.on = 1;
    for (int i = 0; i < f29::NL; i++) first.k.v[i] = mont_io ? f29::RMod::c266(i) : bn::H_C522[i];
    bn::InScale none{};
    // On a coset (forward: coefficient j times shift^j first; inverse: coefficient j times shift^-j last) the constant of
    // that pass moves into a two-level power table, built on the host with the eight-limb code: lo[j] = plain(s^j C),
    // hi[m] = plain(s^(4096 m) 2^261), so that the kernels' product lo hi / 2^261 is plain(s^(j + 4096 m) C).
    Fr* d_pow = nullptr;
    auto power_table = [&](Fr base_mont /* eight-limb Montgomery form of s */, const Fr& c_plain, bn::Scale& sc) -> int32_t {
        const size_t n_hi = std::max<size_t>((n + 4095) >> 12, 1);
        std::vector<Fr> t(4096 + n_hi);
        Fr pw = bn::from_limbs(bn::H_ONE);                                   // s^0
        for (size_t j = 0; j < 4096; j++) {
            t[j] = bn::mul(pw, c_plain);                                     // s^j 2^256 C / 2^256
            pw = bn::mul(pw, base_mont);
        }
        const Fr step = pw, k261 = bn::from_limbs(bn::H_K261);              // s^4096
        pw = bn::from_limbs(bn::H_ONE);
        for (size_t m = 0; m < n_hi; m++) {
            t[4096 + m] = bn::mul(pw, k261);
            pw = bn::mul(pw, step);
        }
        d_pow = (Fr*)ctx->alloc(t.size() * sizeof(Fr));
        if (!d_pow) return ctx->fail(NLX_E_NOMEM, "BN254 coset power table");
        hipError_t e = hipMemcpyAsync(d_pow, t.data(), t.size() * sizeof(Fr), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);                    // t is a local
        if (e != hipSuccess) return ctx->hip_fail(e, "hipMemcpyAsync(coset powers)");
        sc.lo = d_pow;
        sc.hi = d_pow + 4096;
        return NLX_OK;
    };
    Fr shift_mont{};
    if (coset_shift) {
        if (nlx::is_device_ptr(coset_shift)) return ctx->fail(NLX_E_INVAL, "coset_shift is read on the host: pass a host pointer");
        for (int i = 0; i < 4; i++) { shift_mont.v[2 * i] = (uint32_t)coset_shift[i]; shift_mont.v[2 * i + 1] = (uint32_t)(coset_shift[i] >> 32); }
        if (!mont_io) {
            // canonical integers must be below r (compare from the top limb down)
            bool below = false;
            for (int i = 7; i >= 0; i--) {
                if (shift_mont.v[i] != bn::H_MOD[i]) { below = shift_mont.v[i] < bn::H_MOD[i]; break; }
            }
            if (!below) return ctx->fail(NLX_E_INVAL, "coset shift is not below the modulus");
            shift_mont = bn::mul(shift_mont, bn::from_limbs(bn::H_R2));
        }
        bool zero = true;
        for (int i = 0; i < 8; i++) zero = zero && shift_mont.v[i] == 0;
        if (zero) return ctx->fail(NLX_E_INVAL, "coset shift is zero");
    }
    if (coset_shift && !inverse) {
        Fr c_in{};   // plain 2^266 or 2^522 mod r, from the kernels' limbs
        f29::to_words256(first.k, c_in.v);
        const int32_t prc = power_table(shift_mont, c_in, first);
        if (prc) { if (d_pow) ctx->release(d_pow); return prc; }
    }
    // the product that leaves the kernels' form: an element is y 2^261, and y 2^261 c / 2^261 = y c, so c is the PLAIN integer
    // (2^256 if the caller wants fr.Element words) (1/n if inverse) mod r.  The eight-limb Montgomery form of z is the plain
    // integer z 2^256 mod r, which the host code below produces directly.  It rides on the reordering pass, or - when there
    // is none (bit-reversed output, or the decimation in time) - on the last transform pass.
    Fr c = bn::from_limbs(bn::H_ONE);                        // plain 2^256 mod r
    if (inverse) {
        Fr nn{};
        nn.v[0] = (uint32_t)n; nn.v[1] = (uint32_t)((uint64_t)n >> 32);
        c = bn::h_inv(bn::mul(nn, bn::from_limbs(bn::H_R2)));   // Montgomery form of 1/n = plain 2^256 / n
    }
    if (!mont_io) {
        Fr one{};
        one.v[0] = 1;
        c = bn::mul(c, one);                                  // / 2^256: plain 1/n (or 1)
    }
    bn::Scale last{};
    last.on = 1;
    last.k = f29::from_words256(c.v);
    if (coset_shift && inverse) {
        const int32_t prc = power_table(bn::h_inv(shift_mont), c, last);
        if (prc) { if (d_pow) ctx->release(d_pow); return prc; }
    }
    const bool dit = flags & NLX_BN254_BITREV_IN, rev_out = flags & NLX_BN254_BITREV_OUT;
    first.rev_bits = dit ? (int)log_n : 0;    // where the input is bit-reversed, position p holds coefficient bitrev(p)
    last.rev_bits = rev_out ? (int)log_n : 0;
    const bool fused_out = dit || rev_out;
    if (!dit) {
        // decimation in frequency: half-sizes n/2 down to 1, three levels per pass while they last
        int lvl = (int)log_n - 1;  // log2 of the current level's half-size
        bool is_first = true;
        while (lvl >= 0) {
            const int take = lvl >= 2 ? 3 : lvl + 1;
            const size_t nets = (n >> take) * n_cols;
            const unsigned blocks = (unsigned)((nets + 255) / 256);
            const bn::InScale in = is_first ? first : none;
            const bn::Scale out = (fused_out && lvl - take < 0) ? last : none;
            if (take == 3) hipLaunchKernelGGL(bn::k_bn_dif3, dim3(blocks), dim3(256), 0, st, d, log_n, (unsigned)lvl, tb.d_lo, tb.d_hi, (uint32_t)n_cols, in, out);
            else if (take == 2) hipLaunchKernelGGL(bn::k_bn_dif<2>, dim3(blocks), dim3(256), 0, st, d, log_n, (unsigned)lvl, tb.d_lo, tb.d_hi, (uint32_t)n_cols, in, out);
            else hipLaunchKernelGGL(bn::k_bn_dif<1>, dim3(blocks), dim3(256), 0, st, d, log_n, (unsigned)lvl, tb.d_lo, tb.d_hi, (uint32_t)n_cols, in, out);
            is_first = false;
            lvl -= take;
        }
    } else {
        // decimation in time: half-sizes 1 up to n/2; the short pass (log_n mod 3 levels) goes first
        unsigned lh = 0;
        while (lh < log_n) {
            const unsigned take = (lh == 0 && log_n % 3) ? log_n % 3 : 3;
            const size_t nets = (n >> take) * n_cols;
            const unsigned blocks = (unsigned)((nets + 255) / 256);
            const bn::InScale in = lh == 0 ? first : none;
            const bn::Scale out = lh + take >= log_n ? last : none;
            if (take == 3) hipLaunchKernelGGL(bn::k_bn_dit<3>, dim3(blocks), dim3(256), 0, st, d, log_n, lh, tb.d_lo, tb.d_hi, (uint32_t)n_cols, in, out);
            else if (take == 2) hipLaunchKernelGGL(bn::k_bn_dit<2>, dim3(blocks), dim3(256), 0, st, d, log_n, lh, tb.d_lo, tb.d_hi, (uint32_t)n_cols, in, out);
            else hipLaunchKernelGGL(bn::k_bn_dit<1>, dim3(blocks), dim3(256), 0, st, d, log_n, lh, tb.d_lo, tb.d_hi, (uint32_t)n_cols, in, out);
            lh += take;
        }
    }
    ctx->end_kernel();
    if (!fused_out) {   // back to natural order, with the product that leaves the kernels' form
        ctx->begin_kernel("bn254_ntt_reorder", 64.0 * count);
        if (log_n >= 2 * bn::BR_K) hipLaunchKernelGGL(bn::k_bn_bitrev_tiled, dim3(1u << (log_n - 2 * bn::BR_K), (unsigned)n_cols), dim3(256), 0, st, d, log_n, last);
        else hipLaunchKernelGGL(bn::k_bn_bitrev, dim3(blocks1), dim3(256), 0, st, d, log_n, (uint32_t)n_cols, last);
        ctx->end_kernel();
    }
    int32_t rc = s.finish();
    hipError_t es = hipStreamSynchronize(st);
    if (d_pow) ctx->release(d_pow);
    if (rc) return rc;
    if (es != hipSuccess) return ctx->hip_fail(es, "hipStreamSynchronize");
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return ctx->hip_fail(le, "kernel launch");
    return NLX_OK;
} NLX_CATCH(ctx)

int32_t nlx_bn254_ntt_batch(nlx_ctx* ctx, uint64_t* cols, size_t n_cols, uint32_t log_n, int inverse, uint32_t flags) NLX_TRY {
    return nlx_bn254_ntt_batch_coset(ctx, cols, n_cols, log_n, inverse, flags, nullptr);
} NLX_CATCH(ctx)

}  // extern "C"
