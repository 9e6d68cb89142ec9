// Multi-scalar multiplication over BN254 G1 for gfx950 - the second piece of SURVEY.md §8 row f.4 (the recursive wrap: a
// gnark PLONK / Groth16 prover over BN254 commits to its polynomials with KZG, i.e. one G1 MSM of the circuit's size per
// polynomial; BASELINE.json configs[4] puts that size at 2^24).  The wrap is not in /root/reference (succinct.json:7-8 only
// names the platform entry point; gnark-crypto is Go, un-vendored): this follows the published definitions - the curve
// y^2 = x^3 + 3 over Fq, its group order r - and gnark-crypto's memory layout (G1Affine = X, Y as fp.Element: four
// little-endian 64-bit words each, Montgomery form, the point at infinity as (0, 0); scalars as fr.Element), so that
// ecc/bn254 G1Affine.MultiExp(points, scalars) can be replaced by one call on the caller's slices.
//
// Bucket method (Pippenger) with 16-bit unsigned windows, laid out for the GPU:
//   k_msm_digits   one lane per scalar: out of Montgomery form, sixteen 16-bit digits -> one (digit, index) pair per window;
//   hipcub         radix sort of each window's pairs by digit (16-bit keys); k_msm_ranges reads every bucket's first and
//                  last position off the sorted keys;
//   k_msm_buckets  one lane per bucket (16 x 65 535 of them): the sum of its points by mixed Jacobian additions, points
//                  fetched through the sorted indices - at 2^24 points a bucket holds ~256, so neighbouring lanes run
//                  loops of similar length;
//   k_msm_reduce   per window sum_b b * B_b: 2 048 lanes x 32 buckets each by running sums, the chunk offsets by a 16-bit
//                  double-and-add, 256 partial results at a time through LDS;
//   host           the blocks' partial sums, sum_w 2^(16 w) W_w (240 doublings) and the one inversion for the affine result.
// Bound: integer VALU - a mixed addition is 11 field products.  The kernels compute on nine 29-bit limbs (bn254_f29.hpp:
// fixed 64-bit columns, one multiply-add per partial product, ~250 instructions per product against ~880 for the CIOS
// product on 32-bit limbs, whose shifting accumulator costs gfx950 a register move per multiply-add); the points are
// converted once per call, the host's short tail stays on 32-bit limbs (bn254_fp.hpp).
#include <hipcub/hipcub.hpp>
#include <vector>
#include "bn254_f29.hpp"
#include "bn254_fp.hpp"
#include "ctx.hpp"
#include "transcript.hpp"
#include "../../include/nlx.h"

namespace nlx {
namespace msm {

using namespace bnf;
typedef Fp<QP> Fq;
typedef Fp<RP> Fr;

constexpr int WINDOW_BITS = 16, N_WINDOWS = 16, N_BUCKETS = 1 << WINDOW_BITS;
// The top window's digit has only 14 bits (scalars are below r < 2^254): its 12 388 buckets would hold five times the
// points of any other and their lanes would run on alone at the end of the bucket kernel (63 ms, against 28 ms of issue
// time).  Its sort key is therefore digit * 4 + (point index mod 4): four sub-buckets per digit, which the window
// reduction adds up before weighing them.
constexpr int TOP_SUB_BITS = 2;

struct Affine { Fq x, y; };       // (0, 0) = the point at infinity (gnark-crypto's convention; not on the curve)
struct Jac { Fq x, y, z; };       // z = 0: the point at infinity

BNF_HD bool is_inf(const Affine& p) { return is_zero(p.x) && is_zero(p.y); }
BNF_HD Jac jac_inf() { return Jac{one<QP>(), one<QP>(), zero<QP>()}; }
BNF_HD Jac from_affine(const Affine& p) { return is_inf(p) ? jac_inf() : Jac{p.x, p.y, one<QP>()}; }

// dbl-2009-l (a = 0): 2M + 5S
BNF_HD Jac jdbl(const Jac& p) {
    if (is_zero(p.z)) return p;
    const Fq a = sqr(p.x), b = sqr(p.y), c = sqr(b);
    Fq d = sub(sub(sqr(add(p.x, b)), a), c);
    d = dbl(d);
    const Fq e = add(dbl(a), a), f = sqr(e);
    Jac r;
    r.x = sub(f, dbl(d));
    r.y = sub(mul(e, sub(d, r.x)), dbl(dbl(dbl(c))));
    r.z = dbl(mul(p.y, p.z));
    return r;
}
// madd-2007-bl: Jacobian + affine, 7M + 4S; the exceptional cases (infinity on either side, equal or opposite points) are
// real here: a bucket may well receive the same point twice
BNF_HD Jac jmadd(const Jac& p, const Affine& q) {
    if (is_inf(q)) return p;
    if (is_zero(p.z)) return Jac{q.x, q.y, one<QP>()};
    const Fq z1z1 = sqr(p.z), u2 = mul(q.x, z1z1), s2 = mul(mul(q.y, p.z), z1z1);
    const Fq h = sub(u2, p.x);
    Fq r = sub(s2, p.y);
    if (is_zero(h)) {
        if (is_zero(r)) return jdbl(Jac{q.x, q.y, one<QP>()});
        return jac_inf();
    }
    r = dbl(r);
    const Fq hh = sqr(h), i = dbl(dbl(hh)), j = mul(h, i), v = mul(p.x, i);
    Jac o;
    o.x = sub(sub(sqr(r), j), dbl(v));
    o.y = sub(mul(r, sub(v, o.x)), dbl(mul(p.y, j)));
    o.z = sub(sub(sqr(add(p.z, h)), z1z1), hh);
    return o;
}
// add-2007-bl: 11M + 5S
BNF_HD Jac jadd(const Jac& p, const Jac& q) {
    if (is_zero(p.z)) return q;
    if (is_zero(q.z)) return p;
    const Fq z1z1 = sqr(p.z), z2z2 = sqr(q.z);
    const Fq u1 = mul(p.x, z2z2), u2 = mul(q.x, z1z1);
    const Fq s1 = mul(mul(p.y, q.z), z2z2), s2 = mul(mul(q.y, p.z), z1z1);
    const Fq h = sub(u2, u1);
    Fq r = sub(s2, s1);
    if (is_zero(h)) {
        if (is_zero(r)) return jdbl(p);
        return jac_inf();
    }
    r = dbl(r);
    const Fq i = sqr(dbl(h)), j = mul(h, i), v = mul(u1, i);
    Jac o;
    o.x = sub(sub(sqr(r), j), dbl(v));
    o.y = sub(mul(r, sub(v, o.x)), dbl(mul(s1, j)));
    o.z = mul(sub(sub(sqr(add(p.z, q.z)), z1z1), z2z2), h);
    return o;
}
BNF_HD Jac jneg(const Jac& p) { return Jac{p.x, neg(p.y), p.z}; }

// ---- the same group law on the device's field representation (bn254_f29.hpp).  Coordinates of stored points are "tight"
// (< 2^255); the bound of every intermediate is noted where it is not a product (products are < 2^255 whenever the two
// operands multiply to less than 2^515) ----
using f29::Fe;
struct Aff29 { Fe x, y; };        // both exactly zero: the point at infinity
// How the converted points rest in memory: both coordinates (tight, below 2^255) re-sliced into eight 32-bit words each -
// 64 bytes, one aligned access per gathered point (as 72 bytes of limbs a point straddled two 128-byte lines: the bucket
// kernel fetched 67 GB for 2^28 gathered points, PMC FETCH_SIZE).
struct __attribute__((aligned(64))) AffPacked { uint32_t x[8], y[8]; };
__device__ __forceinline__ Aff29 unpack(const AffPacked& p) { return Aff29{f29::from_words256(p.x), f29::from_words256(p.y)}; }
struct Jac29 { Fe x, y, z; };     // z exactly zero: the point at infinity
struct JacWords { uint32_t x[8], y[8], z[8]; };   // canonical integers, for the host's tail

__device__ __forceinline__ bool is_inf(const Aff29& p) { return f29::is_zero_exact(p.x) && f29::is_zero_exact(p.y); }
__device__ __forceinline__ Jac29 inf29() { return Jac29{f29::one(), f29::one(), f29::zero()}; }

__device__ __noinline__ Jac29 jdbl29(const Jac29& p) {   // rare in the bucket kernel (a bucket receiving its own sum): kept out of line
    using namespace f29;
    if (is_zero_exact(p.z)) return p;
    const Fe a = sqr(p.x), b = sqr(p.y), c = sqr(b);
    const Fe d = dbl(tighten(sub<8>(sqr(add(p.x, b)), add(a, c))));   // (x + b)^2 - a - c: the subtrahend < 2^256; d < 2.2 q
    const Fe e = add(dbl(a), a);                                      // 3 a < 3 * 2^255
    Jac29 r;
    r.x = tighten(sub<8>(sqr(e), dbl(d)));                            // 2 d < 2^256
    const Fe t = mul(e, sub<4>(d, r.x));                              // d - x3 + 4 q < 2^256.3, times e < 2^256.6
    const Fe c4 = dbl(dbl(tighten(c)));                               // 4 c < 4.4 q: subtracted twice (8 c would pass 8 q)
    r.y = tighten(sub<8>(tighten(sub<8>(t, c4)), c4));
    r.z = tighten(dbl(mul(p.y, p.z)));
    return r;
}
// The mixed addition in place, common case only: returns 0 and leaves acc + q in acc, or - touching nothing - 1 if q is
// acc's own point (the sum is a doubling) or 2 if it is its negative (the sum is the point at infinity).  The rare cases
// are the caller's, OUTSIDE this function: with the doubling called from in here the result struct of the whole addition
// lived in scratch memory - 112 bytes written and read back per addition, 30 GB per 2^24-point MSM (PMC WRITE_SIZE).
__device__ __forceinline__ int jmadd29_common(Jac29& acc, const Aff29& q) {
    using namespace f29;
    const Fe z1z1 = sqr(acc.z), u2 = mul(q.x, z1z1), s2 = mul(mul(q.y, acc.z), z1z1);
    const Fe h = sub<4>(u2, acc.x), r0 = sub<4>(s2, acc.y);           // < 2^255 + 4 q = 2^256.3
    if (is_zero_mod(h)) return is_zero_mod(r0) ? 1 : 2;
    const Fe r = dbl(r0);                                             // < 2^257.3
    const Fe hh = sqr(h), i = dbl(dbl(hh)), j = mul(h, i), v = mul(acc.x, i);   // i < 2^257
    const Fe x3 = tighten(sub<8>(sqr(r), add(j, dbl(v))));            // j + 2 v < 3 * 2^255 <= 8 q
    const Fe y3 = tighten(sub<8>(mul(r, sub<4>(v, x3)), dbl(mul(acc.y, j))));
    acc.z = tighten(sub<8>(sqr(add(acc.z, h)), add(z1z1, hh)));       // z + h < 2^256.8
    acc.x = x3;
    acc.y = y3;
    return 0;
}
__device__ __forceinline__ Jac29 jmadd29(const Jac29& p, const Aff29& q) {
    if (is_inf(q)) return p;
    if (f29::is_zero_exact(p.z)) return Jac29{q.x, q.y, f29::one()};
    Jac29 o = p;
    const int st = jmadd29_common(o, q);
    if (st == 1) return jdbl29(Jac29{q.x, q.y, f29::one()});
    if (st == 2) return inf29();
    return o;
}
__device__ __forceinline__ Jac29 jadd29(const Jac29& p, const Jac29& q) {
    using namespace f29;
    if (is_zero_exact(p.z)) return q;
    if (is_zero_exact(q.z)) return p;
    const Fe z1z1 = sqr(p.z), z2z2 = sqr(q.z);
    const Fe u1 = mul(p.x, z2z2), u2 = mul(q.x, z1z1);
    const Fe s1 = mul(mul(p.y, q.z), z2z2), s2 = mul(mul(q.y, p.z), z1z1);
    const Fe h = sub<4>(u2, u1), r0 = sub<4>(s2, s1);
    if (is_zero_mod(h)) {
        if (is_zero_mod(r0)) return jdbl29(p);
        return inf29();
    }
    const Fe r = dbl(r0);
    const Fe i = sqr(dbl(h)), j = mul(h, i), v = mul(u1, i);          // 2 h < 2^257.3
    Jac29 o;
    o.x = tighten(sub<8>(sqr(r), add(j, dbl(v))));
    o.y = tighten(sub<8>(mul(r, sub<4>(v, o.x)), dbl(mul(s1, j))));
    o.z = mul(sub<8>(sqr(add(p.z, q.z)), add(z1z1, z2z2)), h);        // (< 2^257.1) (< 2^256.3)
    return o;
}
__device__ __forceinline__ Jac29 jneg29(const Jac29& p) { return Jac29{p.x, f29::tighten(f29::sub<4>(f29::zero(), p.y)), p.z}; }

// gnark-crypto G1Affine words -> the device form, once per point
__global__ __launch_bounds__(256) void k_msm_convert(const uint64_t* __restrict__ points, size_t n, AffPacked* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t w[16];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint64_t x = points[8 * i + k];
        w[2 * k] = (uint32_t)x;
        w[2 * k + 1] = (uint32_t)(x >> 32);
    }
    AffPacked p;
    f29::to_words256(f29::from_mont256(w), p.x);       // (0, 0) stays exactly (0, 0)
    f29::to_words256(f29::from_mont256(w + 8), p.y);
    out[i] = p;
}

// ---- digits ----
__global__ __launch_bounds__(256) void k_msm_digits(const uint64_t* __restrict__ scalars, size_t n, int montgomery,
                                                    uint16_t* __restrict__ keys /* [window][n] */) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr s = load_words<RP>(scalars + 4 * i);
    if (montgomery) s = from_mont(s);
    else if (geq_mod(s)) sub_mod_raw(s);   // a non-reduced word string: fold once (callers hand reduced values)
#pragma unroll
    for (int w = 0; w < N_WINDOWS; w++) {
        uint32_t d = (s.v[w >> 1] >> (16 * (w & 1))) & 0xFFFFu;
        if (w == N_WINDOWS - 1) d = (d << TOP_SUB_BITS) | ((uint32_t)i & ((1u << TOP_SUB_BITS) - 1));   // d < 2^14 there
        keys[(size_t)w * n + i] = (uint16_t)d;
    }
}
__global__ __launch_bounds__(256) void k_msm_iota(uint32_t* __restrict__ v, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = (uint32_t)i;
}

// where every bucket of one window starts and ends in its sorted pairs: position p opens bucket key[p] if it differs from
// its left neighbour and closes it if it differs from its right one (empty buckets keep lo = hi = 0) - no atomics: a
// histogram of 2^28 digits on 2^20 counters (the top window's digits have 12 bits) cost 12 ms of contention
__global__ __launch_bounds__(256) void k_msm_ranges(const uint16_t* __restrict__ sorted_keys, size_t n, uint32_t base /* w * n */,
                                                    uint32_t* __restrict__ lo, uint32_t* __restrict__ hi /* this window's 65536 */) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint16_t k = sorted_keys[p];
    if (p == 0 || sorted_keys[p - 1] != k) lo[k] = base + (uint32_t)p;
    if (p + 1 == n || sorted_keys[p + 1] != k) hi[k] = base + (uint32_t)p + 1;
}

// ---- bucket sums: lane (w, b) adds the points whose window-w digit is b ----
#ifndef NLX_MSM_MINW
#define NLX_MSM_MINW 3   // waves per SIMD the register allocation must allow (tuning builds: build.py NLX_EXTRA_FLAGS)
#endif
__global__ __launch_bounds__(64, NLX_MSM_MINW) void k_msm_buckets(const AffPacked* __restrict__ points, const uint32_t* __restrict__ sorted /* [window][n] */,
                                                    const uint32_t* __restrict__ range_lo, const uint32_t* __restrict__ range_hi /* [window][65536] */,
                                                    Jac29* __restrict__ buckets /* [window][65536] */) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;   // = w * 65536 + b
    if (t >= (uint32_t)N_WINDOWS * N_BUCKETS) return;
    Jac29 acc = inf29();
    const uint32_t digit = (t >> WINDOW_BITS) == N_WINDOWS - 1 ? (t & (N_BUCKETS - 1)) >> TOP_SUB_BITS : t & (N_BUCKETS - 1);
    if (digit != 0) {   // digit 0 weighs nothing
        const uint32_t lo = range_lo[t], hi = range_hi[t];   // positions in the window-major sorted array
#pragma unroll 1
        for (uint32_t p = lo; p < hi; p++) {
            const Aff29 q = unpack(points[sorted[p]]);
            if (is_inf(q)) continue;
            if (f29::is_zero_exact(acc.z)) {
                acc = Jac29{q.x, q.y, f29::one()};
                continue;
            }
            const int st = jmadd29_common(acc, q);   // acc stays in registers on this path
            if (st == 1) acc = jdbl29(Jac29{q.x, q.y, f29::one()});
            else if (st == 2) acc = inf29();
        }
    }
    buckets[t] = acc;
}

// ---- window sums: W_w = sum_b b B_b.  Lane c of window w owns buckets 32 c .. 32 c + 31; a block is 256 such lanes, a
// window RED_BLOCKS blocks whose partial sums the host adds.  The running-sum chain per lane is 64 additions + 16
// double-and-add steps + 8 tree levels: with 256 buckets per lane (one block per window) this kernel took 14 ms ----
constexpr int RED_LANES = 256, RED_CHUNK = 32, RED_BLOCKS = N_BUCKETS / (RED_LANES * RED_CHUNK);
__global__ __launch_bounds__(RED_LANES) void k_msm_reduce(const Jac29* __restrict__ buckets, JacWords* __restrict__ window_sums /* [window][RED_BLOCKS] */) {
    __shared__ Jac29 part[RED_LANES];
    const uint32_t w = blockIdx.x / RED_BLOCKS, c = (blockIdx.x % RED_BLOCKS) * RED_LANES + threadIdx.x, base = c * RED_CHUNK;
    const Jac29* b = buckets + (size_t)w * N_BUCKETS + base;
    Jac29 running = inf29(), local = inf29();
    const int sub = w == N_WINDOWS - 1 ? TOP_SUB_BITS : 0;   // buckets per digit = 2^sub (a chunk holds whole digits)
#pragma unroll 1
    for (int j = RED_CHUNK - 1; j >= 0; j--) {   // running = sum_{j' >= j} B, local = sum_d (d + 1) (digit (base >> sub) + d's buckets)
        running = jadd29(running, b[j]);
        if ((j & ((1 << sub) - 1)) == 0) local = jadd29(local, running);
    }
    // sum_d ((base >> sub) + d) B_d = local + ((base >> sub) - 1) * running; for the first chunk that is local - running
    Jac29 shifted = inf29();
    if (c == 0) {
        shifted = jneg29(running);
    } else {
        const uint32_t k = (base >> sub) - 1;
#pragma unroll 1
        for (int bit = 15; bit >= 0; bit--) {
            shifted = jdbl29(shifted);
            if ((k >> bit) & 1) shifted = jadd29(shifted, running);
        }
    }
    const uint32_t l = threadIdx.x;
    part[l] = jadd29(local, shifted);
    __syncthreads();
#pragma unroll 1
    for (int stride = RED_LANES / 2; stride > 0; stride >>= 1) {
        if (l < (uint32_t)stride) part[l] = jadd29(part[l], part[l + stride]);
        __syncthreads();
    }
    if (l == 0) {   // canonical integers for the host (which works on 32-bit limbs, bn254_fp.hpp)
        JacWords o;
        f29::to_canonical256(part[0].x, o.x);
        f29::to_canonical256(part[0].y, o.y);
        if (f29::is_zero_exact(part[0].z)) {
            for (int k = 0; k < 8; k++) o.z[k] = 0;
        } else {
            f29::to_canonical256(part[0].z, o.z);
        }
        window_sums[blockIdx.x] = o;
    }
}

// ---- test / bench data: out[i] = (i + 1) P for i < n as G1Affine words - n DISTINCT curve points (an SRS's worth of
// gather targets; a big-integer model produces a few thousand per second).  One lane per chunk of GEN_CHUNK consecutive
// multiples: the chunk's first point by double-and-add, the rest by mixed additions of P, kept in Jacobian form in a
// scratch array; then ONE inversion per lane (Montgomery's trick over the chunk's Z coordinates) takes them to affine. ----
constexpr uint32_t GEN_CHUNK = 256;
__device__ __forceinline__ Fe inv29(const Fe& a) {   // a^(q - 2), a != 0 mod q
    constexpr uint32_t E[8] = {0xd87cfd45u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    Fe r = f29::one();
#pragma unroll 1
    for (int bit = 253; bit >= 0; bit--) {
        r = f29::sqr(r);
        if ((E[bit >> 5] >> (bit & 31)) & 1) r = f29::mul(r, a);
    }
    return r;
}
__global__ __launch_bounds__(64) void k_g1_multiples(AffPacked base_packed, uint64_t n, Jac29* __restrict__ jac /* [n] */,
                                                     Fe* __restrict__ prefix /* [n] */, uint64_t* __restrict__ out /* [n][8] */) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t first = t * GEN_CHUNK;
    if (first >= n) return;
    const uint32_t count = (uint32_t)(n - first < GEN_CHUNK ? n - first : GEN_CHUNK);
    const Aff29 base = unpack(base_packed);
    // (first + 1) P
    Jac29 acc = inf29();
    const uint64_t k = first + 1;
#pragma unroll 1
    for (int bit = 63 - __clzll((long long)k); bit >= 0; bit--) {
        acc = jdbl29(acc);
        if ((k >> bit) & 1) acc = jmadd29(acc, base);
    }
    Fe run = f29::one();
#pragma unroll 1
    for (uint32_t j = 0; j < count; j++) {
        if (j) acc = jmadd29(acc, base);
        jac[first + j] = acc;
        prefix[first + j] = run;                      // product of the Z's before this one
        run = f29::mul(run, acc.z);                   // (no multiple below the group order is the point at infinity: Z != 0)
    }
    Fe inv = inv29(run);
    Fe k256;                                          // 2^256 mod q (plain): takes x 2^261 to the caller's x 2^256
    {
        constexpr uint32_t K[f29::NL] = {0x058f0d9du, 0x1aea1c6eu, 0x11c2cf74u, 0x11d651ebu, 0x1462c0a7u, 0x11b7bc3cu, 0x1cbd99bau, 0x183340fbu, 0x000e0a77u};
#pragma unroll
        for (int i = 0; i < f29::NL; i++) k256.v[i] = K[i];
    }
#pragma unroll 1
    for (uint32_t j = count; j-- > 0;) {
        const Jac29 p = jac[first + j];
        const Fe zi = f29::mul(inv, prefix[first + j]);   // 1 / Z_j
        inv = f29::mul(inv, p.z);
        const Fe zi2 = f29::sqr(zi);
        const Fe x = f29::canonical(f29::mul(f29::mul(p.x, zi2), k256)), y = f29::canonical(f29::mul(f29::mul(p.y, f29::mul(zi2, zi)), k256));
        uint32_t w[16];
        f29::to_words256(x, w);
        f29::to_words256(y, w + 8);
#pragma unroll
        for (int i = 0; i < 8; i++) out[(first + j) * 8 + i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
    }
}

}  // namespace msm
}  // namespace nlx

using namespace nlx;

extern "C" int32_t nlx_bn254_msm_g1(nlx_ctx* ctx, const uint64_t* points, const uint64_t* scalars, uint64_t n, uint32_t flags,
                                    uint64_t out[8]) {
    using namespace nlx::msm;
    if (!ctx) return NLX_E_INVAL;
    if (!out || (n && (!points || !scalars))) return ctx->fail(NLX_E_INVAL, "NULL argument");
    if (flags & ~(uint32_t)NLX_BN254_MONTGOMERY) return ctx->fail(NLX_E_RANGE, "unknown flag");
    if (n > ((uint64_t)1 << 27)) return ctx->fail(NLX_E_RANGE, "at most 2^27 points per call (32-bit positions of 16 n pairs)");
    for (int i = 0; i < 8; i++) out[i] = 0;
    if (n == 0) return NLX_OK;
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    Staged sp(ctx, points, (size_t)n * 64, true, false);
    if (sp.status) return sp.status;
    Staged ss(ctx, scalars, (size_t)n * 32, true, false);
    if (ss.status) return ss.status;
    const size_t pairs = (size_t)n * N_WINDOWS, n_hist = (size_t)N_WINDOWS * N_BUCKETS;
    uint16_t* d_keys = (uint16_t*)ctx->alloc(pairs * 2);
    uint16_t* d_keys_sorted = (uint16_t*)ctx->alloc((size_t)n * 2);
    uint32_t* d_iota = (uint32_t*)ctx->alloc((size_t)n * 4);
    uint32_t* d_sorted = (uint32_t*)ctx->alloc(pairs * 4);
    uint32_t* d_lo = (uint32_t*)ctx->alloc(n_hist * 2 * 4);   // range_lo | range_hi
    uint32_t* d_hi = d_lo ? d_lo + n_hist : nullptr;
    Jac29* d_buckets = (Jac29*)ctx->alloc(n_hist * sizeof(Jac29));
    AffPacked* d_pts = (AffPacked*)ctx->alloc((size_t)n * sizeof(AffPacked));   // the points in the kernels' field representation
    constexpr int N_WSUM = N_WINDOWS * RED_BLOCKS;
    JacWords* d_wsum = (JacWords*)ctx->alloc(N_WSUM * sizeof(JacWords));
    size_t tmp_bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_keys, d_keys_sorted, d_iota, d_sorted, (int)n, 0, WINDOW_BITS, st);
    void* d_tmp = ctx->alloc(tmp_bytes ? tmp_bytes : 16);
    auto release_all = [&]() {
        for (void* p : {(void*)d_keys, (void*)d_keys_sorted, (void*)d_iota, (void*)d_sorted, (void*)d_lo, (void*)d_buckets, (void*)d_pts,
                        (void*)d_wsum, d_tmp})
            if (p) ctx->release(p);
    };
    if (!d_keys || !d_keys_sorted || !d_iota || !d_sorted || !d_lo || !d_buckets || !d_pts || !d_wsum || !d_tmp) {
        release_all();
        return NLX_E_NOMEM;
    }
    int32_t rc = NLX_OK;
    auto hip_ok = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && !rc) rc = ctx->hip_fail(e, what);
        return e == hipSuccess;
    };
    const unsigned blocks_n = (unsigned)((n + 255) / 256);
    hip_ok(hipMemsetAsync(d_lo, 0, n_hist * 2 * 4, st), "hipMemsetAsync");
    // algorithmic bytes of the whole job: every point and scalar once
    ctx->begin_kernel("bn254_msm_g1", 96.0 * (double)n, n);
    hipLaunchKernelGGL(k_msm_digits, dim3(blocks_n), dim3(256), 0, st, ss.as<uint64_t>(), (size_t)n,
                       (flags & NLX_BN254_MONTGOMERY) ? 1 : 0, d_keys);
    hipLaunchKernelGGL(k_msm_iota, dim3(blocks_n), dim3(256), 0, st, d_iota, (size_t)n);
    hipLaunchKernelGGL(k_msm_convert, dim3(blocks_n), dim3(256), 0, st, sp.as<uint64_t>(), (size_t)n, d_pts);
    for (int w = 0; w < N_WINDOWS && !rc; w++) {
        size_t tb = tmp_bytes;
        hip_ok(hipcub::DeviceRadixSort::SortPairs(d_tmp, tb, d_keys + (size_t)w * n, d_keys_sorted, d_iota, d_sorted + (size_t)w * n,
                                                  (int)n, 0, WINDOW_BITS, st), "hipcub::SortPairs");
        hipLaunchKernelGGL(k_msm_ranges, dim3(blocks_n), dim3(256), 0, st, d_keys_sorted, (size_t)n, (uint32_t)((size_t)w * n),
                           d_lo + (size_t)w * N_BUCKETS, d_hi + (size_t)w * N_BUCKETS);
    }
    if (!rc) {
        hipLaunchKernelGGL(k_msm_buckets, dim3((unsigned)(n_hist / 64)), dim3(64), 0, st, d_pts, d_sorted, d_lo, d_hi, d_buckets);
        hipLaunchKernelGGL(k_msm_reduce, dim3(N_WSUM), dim3(RED_LANES), 0, st, d_buckets, d_wsum);
    }
    ctx->end_kernel();
    JacWords words[N_WSUM];
    Jac part[N_WSUM], wsum[N_WINDOWS];
    if (!rc) rc = fetch(ctx, words, d_wsum, sizeof(words));
    hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize");
    hip_ok(hipGetLastError(), "kernel launch");
    release_all();
    if (rc) return rc;
    // the blocks' partial sums per window, sum_w 2^(16 w) W_w, then to affine
    for (int k = 0; k < N_WSUM; k++) {
        Fq x, y, z;
        for (int l = 0; l < 8; l++) { x.v[l] = words[k].x[l]; y.v[l] = words[k].y[l]; z.v[l] = words[k].z[l]; }
        part[k] = Jac{to_mont(x), to_mont(y), to_mont(z)};   // z = 0 stays 0: the point at infinity
    }
    for (int w = 0; w < N_WINDOWS; w++) {
        wsum[w] = part[w * RED_BLOCKS];
        for (int k = 1; k < RED_BLOCKS; k++) wsum[w] = jadd(wsum[w], part[w * RED_BLOCKS + k]);
    }
    Jac acc = wsum[N_WINDOWS - 1];
    for (int w = N_WINDOWS - 2; w >= 0; w--) {
        for (int k = 0; k < WINDOW_BITS; k++) acc = jdbl(acc);
        acc = jadd(acc, wsum[w]);
    }
    if (is_zero(acc.z)) return NLX_OK;   // infinity: (0, 0)
    const Fq zi = inv_host(acc.z), zi2 = sqr(zi);
    store_words(mul(acc.x, zi2), out);
    store_words(mul(acc.y, mul(zi2, zi)), out + 4);
    return NLX_OK;
}

// Sum of n G1Affine points on the host (gnark-crypto layouts as above): what joins the partial results of an MSM whose points
// were split over several GPUs - one addition per rank.
extern "C" int32_t nlx_bn254_g1_sum(const uint64_t* points, uint64_t n, uint64_t out[8]) {
    using namespace nlx::msm;
    if (!out || (n && !points)) return NLX_E_INVAL;
    Jac acc = jac_inf();
    for (uint64_t i = 0; i < n; i++) {
        Affine p{load_words<QP>(points + 8 * i), load_words<QP>(points + 8 * i + 4)};
        acc = jmadd(acc, p);
    }
    for (int i = 0; i < 8; i++) out[i] = 0;
    if (is_zero(acc.z)) return NLX_OK;
    const Fq zi = inv_host(acc.z), zi2 = sqr(zi);
    store_words(mul(acc.x, zi2), out);
    store_words(mul(acc.y, mul(zi2, zi)), out + 4);
    return NLX_OK;
}

// out[i] = (i + 1) P, i < n, as G1Affine words (Montgomery), on the device: n distinct curve points for tests and benches.
extern "C" int32_t nlx_bn254_g1_multiples(nlx_ctx* ctx, const uint64_t base[8], uint64_t n, uint64_t* out) {
    using namespace nlx::msm;
    if (!ctx) return NLX_E_INVAL;
    if (!base || (n && !out)) return ctx->fail(NLX_E_INVAL, "NULL argument");
    if (n > ((uint64_t)1 << 27)) return ctx->fail(NLX_E_RANGE, "at most 2^27 points");
    if (n == 0) return NLX_OK;
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    Staged so(ctx, out, (size_t)n * 64, false, true);
    if (so.status) return so.status;
    uint64_t* d_base = (uint64_t*)ctx->alloc(64 + sizeof(AffPacked));
    Jac29* d_jac = (Jac29*)ctx->alloc((size_t)n * sizeof(Jac29));
    Fe* d_prefix = (Fe*)ctx->alloc((size_t)n * sizeof(Fe));
    int32_t rc = NLX_OK;
    if (!d_base || !d_jac || !d_prefix) rc = NLX_E_NOMEM;
    AffPacked packed{};
    if (!rc) {
        hipError_t e = hipMemcpyAsync(d_base, base, 64, hipMemcpyHostToDevice, st);
        AffPacked* d_packed = (AffPacked*)(d_base + 8);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_msm_convert, dim3(1), dim3(256), 0, st, d_base, (size_t)1, d_packed);
            e = hipMemcpyAsync(&packed, d_packed, sizeof(AffPacked), hipMemcpyDeviceToHost, st);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) rc = ctx->hip_fail(e, "nlx_bn254_g1_multiples");
    }
    if (!rc) {
        const uint64_t lanes = (n + GEN_CHUNK - 1) / GEN_CHUNK;
        hipLaunchKernelGGL(k_g1_multiples, dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, st, packed, n, d_jac, d_prefix, so.as<uint64_t>());
        rc = so.finish();
        hipError_t e = hipStreamSynchronize(st);
        if (!rc && e != hipSuccess) rc = ctx->hip_fail(e, "hipStreamSynchronize");
        hipError_t le = hipGetLastError();
        if (!rc && le != hipSuccess) rc = ctx->hip_fail(le, "kernel launch");
    }
    for (void* p : {(void*)d_base, (void*)d_jac, (void*)d_prefix})
        if (p) ctx->release(p);
    return rc;
}
