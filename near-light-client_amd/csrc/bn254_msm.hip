// Multi-scalar multiplication over BN254 G1 (and G2) for gfx950 - the second piece of SURVEY.md §8 row f.4 (the recursive wrap: a
// gnark PLONK / Groth16 prover over BN254 commits to its polynomials with KZG, i.e. one G1 MSM of the circuit's size per
// polynomial; BASELINE.json configs[4] puts that size at 2^24).  The wrap is not in /root/reference (succinct.json:7-8 only
// names the platform entry point; gnark-crypto is Go, un-vendored): this follows the published definitions - the curve
// y^2 = x^3 + 3 over Fq, its group order r - and gnark-crypto's memory layout (G1Affine = X, Y as fp.Element: four
// little-endian 64-bit words each, Montgomery form, the point at infinity as (0, 0); scalars as fr.Element), so that
// ecc/bn254 G1Affine.MultiExp(points, scalars) can be replaced by one call on the caller's slices.  Groth16's B query is a G2
// MSM: the same pipeline over Fq2 = Fq[u] / (u^2 + 1) (G2Affine = X, Y as E2{A0, A1}; nlx_bn254_msm_g2) - the group law
// below is written once over a field policy (bn254_f29.hpp F1 / F2 on the device, H1 / H2 on the host).
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

// ---- host side (the short tail of an MSM, nlx_bn254_g1_sum): eight 32-bit limbs, bn254_fp.hpp ----
struct H1 {   // Fq
    typedef Fq T;
    static constexpr int WORDS64 = 4;
    static T zero() { return bnf::zero<QP>(); }
    static T one() { return bnf::one<QP>(); }
    static bool is_zero(const T& a) { return bnf::is_zero(a); }
    static T add(const T& a, const T& b) { return bnf::add(a, b); }
    static T sub(const T& a, const T& b) { return bnf::sub(a, b); }
    static T mul(const T& a, const T& b) { return bnf::mul(a, b); }
    static T sqr(const T& a) { return bnf::mul(a, a); }
    static T neg(const T& a) { return bnf::neg(a); }
    static T inv(const T& a) { return bnf::inv_host(a); }
    static T load(const uint64_t* w) { return load_words<QP>(w); }
    static void store(const T& a, uint64_t* w) { store_words(a, w); }
    static T from_canonical(const uint32_t* w) {   // plain integer limbs -> Montgomery form
        Fq x;
        for (int l = 0; l < 8; l++) x.v[l] = w[l];
        return to_mont(x);
    }
};
struct H2 {   // Fq2 = Fq[u] / (u^2 + 1); gnark-crypto's E2{A0, A1}
    struct T { Fq c0, c1; };
    static constexpr int WORDS64 = 8;
    static T zero() { return T{bnf::zero<QP>(), bnf::zero<QP>()}; }
    static T one() { return T{bnf::one<QP>(), bnf::zero<QP>()}; }
    static bool is_zero(const T& a) { return bnf::is_zero(a.c0) && bnf::is_zero(a.c1); }
    static T add(const T& a, const T& b) { return T{bnf::add(a.c0, b.c0), bnf::add(a.c1, b.c1)}; }
    static T sub(const T& a, const T& b) { return T{bnf::sub(a.c0, b.c0), bnf::sub(a.c1, b.c1)}; }
    static T mul(const T& a, const T& b) {
        const Fq t0 = bnf::mul(a.c0, b.c0), t1 = bnf::mul(a.c1, b.c1);
        const Fq s = bnf::mul(bnf::add(a.c0, a.c1), bnf::add(b.c0, b.c1));
        return T{bnf::sub(t0, t1), bnf::sub(bnf::sub(s, t0), t1)};
    }
    static T sqr(const T& a) { return mul(a, a); }
    static T neg(const T& a) { return T{bnf::neg(a.c0), bnf::neg(a.c1)}; }
    static T inv(const T& a) {   // conj(a) / (a0^2 + a1^2)
        const Fq d = bnf::inv_host(bnf::add(bnf::mul(a.c0, a.c0), bnf::mul(a.c1, a.c1)));
        return T{bnf::mul(a.c0, d), bnf::neg(bnf::mul(a.c1, d))};
    }
    static T load(const uint64_t* w) { return T{load_words<QP>(w), load_words<QP>(w + 4)}; }
    static void store(const T& a, uint64_t* w) { store_words(a.c0, w); store_words(a.c1, w + 4); }
    static T from_canonical(const uint32_t* w) { return T{H1::from_canonical(w), H1::from_canonical(w + 8)}; }
};
template <class H> struct AffineH { typename H::T x, y; };   // (0, 0) = the point at infinity (gnark-crypto's convention)
template <class H> struct JacH { typename H::T x, y, z; };   // z = 0: the point at infinity
template <class H> inline typename H::T hdbl(const typename H::T& a) { return H::add(a, a); }
template <class H> inline JacH<H> hinf() { return JacH<H>{H::one(), H::one(), H::zero()}; }
// dbl-2009-l (a = 0): 2M + 5S
template <class H> inline JacH<H> hjdbl(const JacH<H>& p) {
    typedef typename H::T T;
    if (H::is_zero(p.z)) return p;
    const T a = H::sqr(p.x), b = H::sqr(p.y), c = H::sqr(b);
    const T d = hdbl<H>(H::sub(H::sub(H::sqr(H::add(p.x, b)), a), c));
    const T e = H::add(hdbl<H>(a), a), f = H::sqr(e);
    JacH<H> r;
    r.x = H::sub(f, hdbl<H>(d));
    r.y = H::sub(H::mul(e, H::sub(d, r.x)), hdbl<H>(hdbl<H>(hdbl<H>(c))));
    r.z = hdbl<H>(H::mul(p.y, p.z));
    return r;
}
// add-2007-bl: 11M + 5S; equal and opposite points are real cases
template <class H> inline JacH<H> hjadd(const JacH<H>& p, const JacH<H>& q) {
    typedef typename H::T T;
    if (H::is_zero(p.z)) return q;
    if (H::is_zero(q.z)) return p;
    const T z1z1 = H::sqr(p.z), z2z2 = H::sqr(q.z);
    const T u1 = H::mul(p.x, z2z2), u2 = H::mul(q.x, z1z1);
    const T s1 = H::mul(H::mul(p.y, q.z), z2z2), s2 = H::mul(H::mul(q.y, p.z), z1z1);
    const T h = H::sub(u2, u1);
    T r = H::sub(s2, s1);
    if (H::is_zero(h)) return H::is_zero(r) ? hjdbl<H>(p) : hinf<H>();
    r = hdbl<H>(r);
    const T i = H::sqr(hdbl<H>(h)), j = H::mul(h, i), v = H::mul(u1, i);
    JacH<H> o;
    o.x = H::sub(H::sub(H::sqr(r), j), hdbl<H>(v));
    o.y = H::sub(H::mul(r, H::sub(v, o.x)), hdbl<H>(H::mul(s1, j)));
    o.z = H::mul(H::sub(H::sub(H::sqr(H::add(p.z, q.z)), z1z1), z2z2), h);
    return o;
}
template <class H> inline JacH<H> hfrom_affine(const AffineH<H>& p) {
    return (H::is_zero(p.x) && H::is_zero(p.y)) ? hinf<H>() : JacH<H>{p.x, p.y, H::one()};
}
// Jacobian -> affine words (all zero for the point at infinity)
template <class H> inline void hstore_affine(const JacH<H>& p, uint64_t* out) {
    for (int i = 0; i < 2 * H::WORDS64; i++) out[i] = 0;
    if (H::is_zero(p.z)) return;
    const typename H::T zi = H::inv(p.z), zi2 = H::sqr(zi);
    H::store(H::mul(p.x, zi2), out);
    H::store(H::mul(p.y, H::mul(zi2, zi)), out + H::WORDS64);
}

// ---- the group law on the device's field representation (bn254_f29.hpp), written over a field policy F: F1 = Fq (loose
// values; the bound of every intermediate is noted where it is not a product - products are < 2^255 whenever the two
// operands multiply to less than 2^515), F2 = Fq2 (every result tightened, nothing to track).  Coordinates of stored points
// are "tight" (< 2^255). ----
using f29::F1;
using f29::F2;
using f29::Fe;
template <class F> struct AffT { typename F::T x, y; };        // both exactly zero: the point at infinity
template <class F> struct JacT { typename F::T x, y, z; };     // z exactly zero: the point at infinity
template <class F> struct JacWordsT { uint32_t x[F::WORDS], y[F::WORDS], z[F::WORDS]; };   // canonical integers, for the host's tail
// How the converted points rest in memory: the coordinates (tight, below 2^255) re-sliced into eight 32-bit words per base
// field element - 64 bytes per G1 point, one aligned access per gathered point (as 72 bytes of limbs a point straddled two
// 128-byte lines: the bucket kernel fetched 67 GB for 2^28 gathered points, PMC FETCH_SIZE); 128 bytes per G2 point.
template <class F> struct __attribute__((aligned(64))) PackedT { uint32_t x[F::WORDS], y[F::WORDS]; };
template <class F> __device__ __forceinline__ AffT<F> unpack(const PackedT<F>& p) { return AffT<F>{F::from_words(p.x), F::from_words(p.y)}; }
typedef AffT<F1> Aff29;
typedef JacT<F1> Jac29;
typedef PackedT<F1> AffPacked;

template <class F> __device__ __forceinline__ bool is_inf(const AffT<F>& p) { return F::is_zero_exact(p.x) && F::is_zero_exact(p.y); }
template <class F> __device__ __forceinline__ JacT<F> inf29() { return JacT<F>{F::one(), F::one(), F::zero()}; }

template <class F>
__device__ __noinline__ JacT<F> jdbl29(const JacT<F>& p) {   // rare in the bucket kernel (a bucket receiving its own sum): kept out of line
    typedef typename F::T T;
    if (F::is_zero_exact(p.z)) return p;
    const T a = F::sqr(p.x), b = F::sqr(p.y), c = F::sqr(b);
    const T d = F::dbl(F::tighten(F::template sub<8>(F::sqr(F::add(p.x, b)), F::add(a, c))));   // (x + b)^2 - a - c: the subtrahend < 2^256; d < 2.2 q
    const T e = F::add(F::dbl(a), a);                                      // 3 a < 3 * 2^255
    JacT<F> r;
    r.x = F::tighten(F::template sub<8>(F::sqr(e), F::dbl(d)));            // 2 d < 2^256
    const T t = F::mul(e, F::template sub<4>(d, r.x));                     // d - x3 + 4 q < 2^256.3, times e < 2^256.6
    const T c4 = F::dbl(F::dbl(F::tighten(c)));                            // 4 c < 4.4 q: subtracted twice (8 c would pass 8 q)
    r.y = F::tighten(F::template sub<8>(F::tighten(F::template sub<8>(t, c4)), c4));
    r.z = F::tighten(F::dbl(F::mul(p.y, p.z)));
    return r;
}
// The mixed addition in place, common case only: returns 0 and leaves acc + q in acc, or - touching nothing - 1 if q is
// acc's own point (the sum is a doubling) or 2 if it is its negative (the sum is the point at infinity).  The rare cases
// are the caller's, OUTSIDE this function: with the doubling called from in here the result struct of the whole addition
// lived in scratch memory - 112 bytes written and read back per addition, 30 GB per 2^24-point MSM (PMC WRITE_SIZE).
template <class F>
__device__ __forceinline__ int jmadd29_common(JacT<F>& acc, const AffT<F>& q) {
    typedef typename F::T T;
    const T z1z1 = F::sqr(acc.z), u2 = F::mul(q.x, z1z1), s2 = F::mul(F::mul(q.y, acc.z), z1z1);
    const T h = F::template sub<4>(u2, acc.x), r0 = F::template sub<4>(s2, acc.y);   // < 2^255 + 4 q = 2^256.3
    if (F::is_zero_mod(h)) return F::is_zero_mod(r0) ? 1 : 2;
    const T r = F::dbl(r0);                                                           // < 2^257.3
    const T hh = F::sqr(h), i = F::dbl(F::dbl(hh)), j = F::mul(h, i), v = F::mul(acc.x, i);   // i < 2^257
    const T x3 = F::tighten(F::template sub<8>(F::sqr(r), F::add(j, F::dbl(v))));     // j + 2 v < 3 * 2^255 <= 8 q
    const T y3 = F::tighten(F::template sub<8>(F::mul(r, F::template sub<4>(v, x3)), F::dbl(F::mul(acc.y, j))));
    acc.z = F::tighten(F::template sub<8>(F::sqr(F::add(acc.z, h)), F::add(z1z1, hh)));   // z + h < 2^256.8
    acc.x = x3;
    acc.y = y3;
    return 0;
}
template <class F>
__device__ __forceinline__ JacT<F> jmadd29(const JacT<F>& p, const AffT<F>& q) {
    if (is_inf(q)) return p;
    if (F::is_zero_exact(p.z)) return JacT<F>{q.x, q.y, F::one()};
    JacT<F> o = p;
    const int st = jmadd29_common(o, q);
    if (st == 1) return jdbl29(JacT<F>{q.x, q.y, F::one()});
    if (st == 2) return inf29<F>();
    return o;
}
template <class F>
__device__ __forceinline__ JacT<F> jadd29(const JacT<F>& p, const JacT<F>& q) {
    typedef typename F::T T;
    if (F::is_zero_exact(p.z)) return q;
    if (F::is_zero_exact(q.z)) return p;
    const T z1z1 = F::sqr(p.z), z2z2 = F::sqr(q.z);
    const T u1 = F::mul(p.x, z2z2), u2 = F::mul(q.x, z1z1);
    const T s1 = F::mul(F::mul(p.y, q.z), z2z2), s2 = F::mul(F::mul(q.y, p.z), z1z1);
    const T h = F::template sub<4>(u2, u1), r0 = F::template sub<4>(s2, s1);
    if (F::is_zero_mod(h)) {
        if (F::is_zero_mod(r0)) return jdbl29(p);
        return inf29<F>();
    }
    const T r = F::dbl(r0);
    const T i = F::sqr(F::dbl(h)), j = F::mul(h, i), v = F::mul(u1, i);          // 2 h < 2^257.3
    JacT<F> o;
    o.x = F::tighten(F::template sub<8>(F::sqr(r), F::add(j, F::dbl(v))));
    o.y = F::tighten(F::template sub<8>(F::mul(r, F::template sub<4>(v, o.x)), F::dbl(F::mul(s1, j))));
    o.z = F::mul(F::template sub<8>(F::sqr(F::add(p.z, q.z)), F::add(z1z1, z2z2)), h);        // (< 2^257.1) (< 2^256.3)
    return o;
}
template <class F>
__device__ __forceinline__ JacT<F> jneg29(const JacT<F>& p) { return JacT<F>{p.x, F::tighten(F::template sub<4>(F::zero(), p.y)), p.z}; }

// gnark-crypto G1Affine / G2Affine words -> the device form, once per point
template <class F>
__global__ __launch_bounds__(256) void k_msm_convert(const uint64_t* __restrict__ points, size_t n, PackedT<F>* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int W = F::WORDS;   // 32-bit words per coordinate
    uint32_t w[2 * W];
#pragma unroll
    for (int k = 0; k < W; k++) {
        const uint64_t x = points[(size_t)W * i + k];
        w[2 * k] = (uint32_t)x;
        w[2 * k + 1] = (uint32_t)(x >> 32);
    }
    PackedT<F> p;
    F::to_words(F::from_mont256(w), p.x);       // (0, 0) stays exactly (0, 0)
    F::to_words(F::from_mont256(w + W), p.y);
    out[i] = p;
}

// ---- digits ----
__global__ __launch_bounds__(256) void k_msm_digits(const uint64_t* __restrict__ scalars, size_t n, int montgomery,
                                                    uint16_t* __restrict__ keys /* [window][n] */) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr s = load_words<RP>(scalars + 4 * i);
    if (montgomery) s = from_mont(s);
    else {
        // a non-reduced word string: 2^256 / r < 6, so at most five subtractions reduce it fully (one was not enough for
        // s >= 2 r: the top window's digit then exceeded 14 bits and the key was truncated - a wrong point with NLX_OK)
#pragma unroll 1
        for (int k = 0; k < 5 && geq_mod(s); k++) sub_mod_raw(s);
    }
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
template <class F>
__device__ __forceinline__ void bucket_sum(const PackedT<F>* __restrict__ points, const uint32_t* __restrict__ sorted,
                                           const uint32_t* __restrict__ range_lo, const uint32_t* __restrict__ range_hi,
                                           JacT<F>* __restrict__ buckets) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;   // = w * 65536 + b
    if (t >= (uint32_t)N_WINDOWS * N_BUCKETS) return;
    JacT<F> acc = inf29<F>();
    const uint32_t digit = (t >> WINDOW_BITS) == N_WINDOWS - 1 ? (t & (N_BUCKETS - 1)) >> TOP_SUB_BITS : t & (N_BUCKETS - 1);
    if (digit != 0) {   // digit 0 weighs nothing
        const uint32_t lo = range_lo[t], hi = range_hi[t];   // positions in the window-major sorted array
#pragma unroll 1
        for (uint32_t p = lo; p < hi; p++) {
            const AffT<F> q = unpack(points[sorted[p]]);
            if (is_inf(q)) continue;
            if (F::is_zero_exact(acc.z)) {
                acc = JacT<F>{q.x, q.y, F::one()};
                continue;
            }
            const int st = jmadd29_common(acc, q);   // acc stays in registers on this path
            if (st == 1) acc = jdbl29(JacT<F>{q.x, q.y, F::one()});
            else if (st == 2) acc = inf29<F>();
        }
    }
    buckets[t] = acc;
}
__global__ __launch_bounds__(64, NLX_MSM_MINW) void k_msm_buckets(const PackedT<F1>* __restrict__ points, const uint32_t* __restrict__ sorted /* [window][n] */,
                                                                  const uint32_t* __restrict__ range_lo, const uint32_t* __restrict__ range_hi /* [window][65536] */,
                                                                  JacT<F1>* __restrict__ buckets /* [window][65536] */) {
    bucket_sum<F1>(points, sorted, range_lo, range_hi, buckets);
}
__global__ __launch_bounds__(64) void k_msm_buckets_g2(const PackedT<F2>* __restrict__ points, const uint32_t* __restrict__ sorted,
                                                       const uint32_t* __restrict__ range_lo, const uint32_t* __restrict__ range_hi,
                                                       JacT<F2>* __restrict__ buckets) {
    bucket_sum<F2>(points, sorted, range_lo, range_hi, buckets);
}

// ---- window sums: W_w = sum_b b B_b.  Lane c of window w owns buckets 32 c .. 32 c + 31; a block is RED_LANES such lanes, a
// window RED_BLOCKS blocks whose partial sums the host adds.  The running-sum chain per lane is 64 additions + 16
// double-and-add steps + the tree's levels: with 256 buckets per lane (one block per window) this kernel took 14 ms ----
constexpr int RED_CHUNK = 32;
template <class F, int RED_LANES>
__global__ __launch_bounds__(RED_LANES) void k_msm_reduce(const JacT<F>* __restrict__ buckets, JacWordsT<F>* __restrict__ window_sums /* [window][blocks] */) {
    __shared__ JacT<F> part[RED_LANES];
    constexpr int RED_BLOCKS = N_BUCKETS / (RED_LANES * RED_CHUNK);
    const uint32_t w = blockIdx.x / RED_BLOCKS, c = (blockIdx.x % RED_BLOCKS) * RED_LANES + threadIdx.x, base = c * RED_CHUNK;
    const JacT<F>* b = buckets + (size_t)w * N_BUCKETS + base;
    JacT<F> running = inf29<F>(), local = inf29<F>();
    const int sub = w == N_WINDOWS - 1 ? TOP_SUB_BITS : 0;   // buckets per digit = 2^sub (a chunk holds whole digits)
#pragma unroll 1
    for (int j = RED_CHUNK - 1; j >= 0; j--) {   // running = sum_{j' >= j} B, local = sum_d (d + 1) (digit (base >> sub) + d's buckets)
        running = jadd29(running, b[j]);
        if ((j & ((1 << sub) - 1)) == 0) local = jadd29(local, running);
    }
    // sum_d ((base >> sub) + d) B_d = local + ((base >> sub) - 1) * running; for the first chunk that is local - running
    JacT<F> shifted = inf29<F>();
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
        JacWordsT<F> o;
        F::to_canonical(part[0].x, o.x);
        F::to_canonical(part[0].y, o.y);
        if (F::is_zero_exact(part[0].z)) {
            for (int k = 0; k < F::WORDS; k++) o.z[k] = 0;
        } else {
            F::to_canonical(part[0].z, o.z);
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
    Jac29 acc = inf29<F1>();
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

namespace {

// One MSM: F = the device field policy (bn254_f29.hpp), H = the host's, LANES = lanes per window-reduction block (the
// block's partial sums live in LDS: 108 bytes per G1 point, 216 per G2 point).
template <class F, class H, int LANES, class BucketKernel>
int32_t msm_run(nlx_ctx* ctx, const uint64_t* points, const uint64_t* scalars, uint64_t n, uint32_t flags, uint64_t* out,
                BucketKernel bucket_kernel, const char* sample_name) {
    using namespace nlx::msm;
    constexpr int OUT_WORDS = 2 * H::WORDS64;
    if (!ctx) return NLX_E_INVAL;
    if (!out || (n && (!points || !scalars))) return ctx->fail(NLX_E_INVAL, "NULL argument");
    if (flags & ~(uint32_t)NLX_BN254_MONTGOMERY) return ctx->fail(NLX_E_RANGE, "unknown flag");
    if (n > ((uint64_t)1 << 27)) return ctx->fail(NLX_E_RANGE, "at most 2^27 points per call (32-bit positions of 16 n pairs)");
    for (int i = 0; i < OUT_WORDS; i++) out[i] = 0;
    if (n == 0) return NLX_OK;
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    Staged sp(ctx, points, (size_t)n * OUT_WORDS * 8, true, false);
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
    JacT<F>* d_buckets = (JacT<F>*)ctx->alloc(n_hist * sizeof(JacT<F>));
    PackedT<F>* d_pts = (PackedT<F>*)ctx->alloc((size_t)n * sizeof(PackedT<F>));   // the points in the kernels' field representation
    constexpr int RED_BLOCKS = N_BUCKETS / (LANES * RED_CHUNK), N_WSUM = N_WINDOWS * RED_BLOCKS;
    JacWordsT<F>* d_wsum = (JacWordsT<F>*)ctx->alloc(N_WSUM * sizeof(JacWordsT<F>));
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
        return ctx->fail(NLX_E_NOMEM, "MSM of %llu points: device memory for the digit keys / sorted indices / buckets", (unsigned long long)n);
    }
    int32_t rc = NLX_OK;
    auto hip_ok = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && !rc) rc = ctx->hip_fail(e, what);
        return e == hipSuccess;
    };
    const unsigned blocks_n = (unsigned)((n + 255) / 256);
    hip_ok(hipMemsetAsync(d_lo, 0, n_hist * 2 * 4, st), "hipMemsetAsync");
    // algorithmic bytes of the whole job: every point and scalar once
    ctx->begin_kernel(sample_name, (32.0 + OUT_WORDS * 8.0) * (double)n, n);
    hipLaunchKernelGGL(k_msm_digits, dim3(blocks_n), dim3(256), 0, st, ss.as<uint64_t>(), (size_t)n,
                       (flags & NLX_BN254_MONTGOMERY) ? 1 : 0, d_keys);
    hipLaunchKernelGGL(k_msm_iota, dim3(blocks_n), dim3(256), 0, st, d_iota, (size_t)n);
    hipLaunchKernelGGL(k_msm_convert<F>, dim3(blocks_n), dim3(256), 0, st, sp.as<uint64_t>(), (size_t)n, d_pts);
    for (int w = 0; w < N_WINDOWS && !rc; w++) {
        size_t tb = tmp_bytes;
        hip_ok(hipcub::DeviceRadixSort::SortPairs(d_tmp, tb, d_keys + (size_t)w * n, d_keys_sorted, d_iota, d_sorted + (size_t)w * n,
                                                  (int)n, 0, WINDOW_BITS, st), "hipcub::SortPairs");
        hipLaunchKernelGGL(k_msm_ranges, dim3(blocks_n), dim3(256), 0, st, d_keys_sorted, (size_t)n, (uint32_t)((size_t)w * n),
                           d_lo + (size_t)w * N_BUCKETS, d_hi + (size_t)w * N_BUCKETS);
    }
    if (!rc) {
        hipLaunchKernelGGL(bucket_kernel, dim3((unsigned)(n_hist / 64)), dim3(64), 0, st, d_pts, d_sorted, d_lo, d_hi, d_buckets);
        hipLaunchKernelGGL((k_msm_reduce<F, LANES>), dim3(N_WSUM), dim3(LANES), 0, st, d_buckets, d_wsum);
    }
    ctx->end_kernel();
    std::vector<JacWordsT<F>> words(N_WSUM);
    if (!rc) rc = fetch(ctx, words.data(), d_wsum, (size_t)N_WSUM * sizeof(JacWordsT<F>));
    hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize");
    hip_ok(hipGetLastError(), "kernel launch");
    release_all();
    if (rc) return rc;
    // the blocks' partial sums per window, sum_w 2^(16 w) W_w, then to affine
    std::vector<JacH<H>> part(N_WSUM), wsum(N_WINDOWS);
    for (int k = 0; k < N_WSUM; k++)
        part[k] = JacH<H>{H::from_canonical(words[k].x), H::from_canonical(words[k].y), H::from_canonical(words[k].z)};   // z = 0 stays 0
    for (int w = 0; w < N_WINDOWS; w++) {
        wsum[w] = part[w * RED_BLOCKS];
        for (int k = 1; k < RED_BLOCKS; k++) wsum[w] = hjadd<H>(wsum[w], part[w * RED_BLOCKS + k]);
    }
    JacH<H> acc = wsum[N_WINDOWS - 1];
    for (int w = N_WINDOWS - 2; w >= 0; w--) {
        for (int k = 0; k < WINDOW_BITS; k++) acc = hjdbl<H>(acc);
        acc = hjadd<H>(acc, wsum[w]);
    }
    hstore_affine<H>(acc, out);
    return NLX_OK;
}

template <class H>
int32_t affine_sum(const uint64_t* points, uint64_t n, uint64_t* out) {
    using namespace nlx::msm;
    if (!out || (n && !points)) return NLX_E_INVAL;
    JacH<H> acc = hinf<H>();
    for (uint64_t i = 0; i < n; i++) {
        const AffineH<H> p{H::load(points + 2 * H::WORDS64 * i), H::load(points + 2 * H::WORDS64 * i + H::WORDS64)};
        acc = hjadd<H>(acc, hfrom_affine<H>(p));
    }
    hstore_affine<H>(acc, out);
    return NLX_OK;
}

}  // namespace

extern "C" int32_t nlx_bn254_msm_g1(nlx_ctx* ctx, const uint64_t* points, const uint64_t* scalars, uint64_t n, uint32_t flags,
                                    uint64_t out[8]) NLX_TRY {
    return msm_run<f29::F1, msm::H1, 256>(ctx, points, scalars, n, flags, out, msm::k_msm_buckets, "bn254_msm_g1");
} NLX_CATCH(ctx)
extern "C" int32_t nlx_bn254_msm_g2(nlx_ctx* ctx, const uint64_t* points, const uint64_t* scalars, uint64_t n, uint32_t flags,
                                    uint64_t out[16]) NLX_TRY {
    return msm_run<f29::F2, msm::H2, 128>(ctx, points, scalars, n, flags, out, msm::k_msm_buckets_g2, "bn254_msm_g2");
} NLX_CATCH(ctx)

// Sum of n affine points on the host (gnark-crypto layouts as above): what joins the partial results of an MSM whose points
// were split over several GPUs - one addition per rank.
extern "C" int32_t nlx_bn254_g1_sum(const uint64_t* points, uint64_t n, uint64_t out[8]) NLX_TRY { return affine_sum<msm::H1>(points, n, out); } NLX_CATCH(nullptr)
extern "C" int32_t nlx_bn254_g2_sum(const uint64_t* points, uint64_t n, uint64_t out[16]) NLX_TRY { return affine_sum<msm::H2>(points, n, out); } NLX_CATCH(nullptr)

// out[i] = (i + 1) P, i < n, as G1Affine words (Montgomery), on the device: n distinct curve points for tests and benches.
extern "C" int32_t nlx_bn254_g1_multiples(nlx_ctx* ctx, const uint64_t base[8], uint64_t n, uint64_t* out) NLX_TRY {
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
            hipLaunchKernelGGL(k_msm_convert<f29::F1>, dim3(1), dim3(256), 0, st, d_base, (size_t)1, d_packed);
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
} NLX_CATCH(ctx)
