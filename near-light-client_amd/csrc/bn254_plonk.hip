// Row f.4, third piece: the PLONK prover's quotient chain and one KZG opening over BN254's scalar field, on the pieces that
// exist (nlx_bn254_ntt_batch_coset: FFTInverse(DIF) / FFT(DIT, OnCoset); nlx_bn254_msm_g1).
//
// What the recursive wrap's prover (gnark backend/plonk/bn254 `Prove`, reached through succinct.json:7-8's entry point; Go,
// not in /root/reference) does between its commitments, restated from the published protocol (Gabizon-Williamson-Ciobotaru,
// "PLONK", the three-wire arithmetisation gnark implements), NOT from gnark's source - no blinding, no gnark byte format:
//   1. the thirteen polynomials it knows by their values on H (selectors ql qr qm qo qk, permutation s1 s2 s3, wires l r o,
//      grand product z, public-input polynomial) go to coefficients:           FFTInverse(DIF), no reordering
//   2. and from there to the coset shift * <w_4n> of four times the size:       FFT(DIT, OnCoset)
//   3. one pointwise pass builds the quotient's values there
//        t = [ ql l + qr r + qm l r + qo o + qk + pi
//              + alpha ( (l + beta x + gamma)(r + beta k1 x + gamma)(o + beta k2 x + gamma) z
//                        - (l + beta s1 + gamma)(r + beta s2 + gamma)(o + beta s3 + gamma) z(w x) )
//              + alpha^2 L1(x) (z - 1) ] / Z_H(x)
//   4. FFTInverse(OnCoset) gives t's coefficients: three chunks of n (the fourth is zero exactly when the witness satisfies
//      the circuit - reported, not assumed)
//   5. KZG: commitments are nlx_bn254_msm_g1 over the SRS; an opening at zeta is the evaluation, the synthetic division
//      (p(X) - p(zeta)) / (X - zeta) and one more MSM.
//
// Arithmetic: bn254_f29.hpp's nine 29-bit limbs (the NTT's and the MSM's element), every sum / difference tightened.  Elements
// rest in memory as gnark-crypto's fr.Element (Montgomery, R = 2^256: "D-form", x 2^256); the kernels' products divide by
// R' = 2^261, so uniform constants and the domain tables are kept in "I-form" (x 2^261): D x I -> D, I x I -> I, and only a
// product of two data values needs the factor 2^5 put back (x32).
#include <algorithm>
#include <cstring>
#include <vector>
#include "ctx.hpp"
#include "bn254_f29.hpp"

namespace nlx {
namespace bnp {

using f29::Fe;
typedef f29::RMod RM;

__device__ __forceinline__ Fe ld(const uint64_t* p, size_t i) {
    const uint4* q = reinterpret_cast<const uint4*>(p + 4 * i);
    const uint4 a = q[0], b = q[1];
    const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    return f29::from_words256(w);
}
__device__ __forceinline__ void st(uint64_t* p, size_t i, const Fe& v) {   // canonical: what gnark keeps in memory
    const Fe c = f29::canonical<RM>(v);
    uint32_t w[8];
    f29::to_words256(c, w);
    uint4* q = reinterpret_cast<uint4*>(p + 4 * i);
    q[0] = make_uint4(w[0], w[1], w[2], w[3]);
    q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
__device__ __forceinline__ Fe mul(const Fe& a, const Fe& b) { return f29::mul<RM>(a, b); }
__device__ __forceinline__ Fe add(const Fe& a, const Fe& b) { return f29::tighten<RM>(f29::add(a, b)); }
__device__ __forceinline__ Fe sub(const Fe& a, const Fe& b) { return f29::tighten<RM>(f29::sub<4, RM>(a, b)); }
template <int K>
__device__ __forceinline__ Fe shl(const Fe& a) {   // a 2^K as an integer (the caller keeps it below 2^258)
    Fe r;
    r.v[0] = (a.v[0] << K) & f29::MASK;
#pragma unroll
    for (int i = 1; i < f29::NL - 1; i++) r.v[i] = ((a.v[i] << K) | (a.v[i - 1] >> (f29::LB - K))) & f29::MASK;
    r.v[f29::NL - 1] = (a.v[f29::NL - 1] << K) | (a.v[f29::NL - 2] >> (f29::LB - K));
    return r;
}
// x 2^5: D x D products come out as x y 2^251; a value below 1.1 r times 4, tightened, times 8, tightened
__device__ __forceinline__ Fe x32(const Fe& a) {
    const Fe t = f29::tighten<RM>(shl<2>(f29::tighten<RM>(a)));
    return f29::tighten<RM>(shl<3>(t));
}
__device__ __forceinline__ Fe mul_dd(const Fe& a, const Fe& b) { return x32(mul(a, b)); }   // D x D -> D
__device__ __forceinline__ Fe one_i() { return f29::one<RM>(); }
__device__ __forceinline__ Fe one_d() {   // 2^256 mod r
    constexpr uint32_t C[f29::NL] = {0x0ffffffbu, 0x04b1a0e2u, 0x18334a6bu, 0x18ed2b3eu, 0x1462e36fu, 0x11b7bc3cu, 0x1cbd99bau, 0x183340fbu, 0x000e0a77u};
    Fe r;
#pragma unroll
    for (int i = 0; i < f29::NL; i++) r.v[i] = C[i];
    return r;
}
__device__ __forceinline__ Fe root28_i() {   // gnark-crypto's 2^28-th root of unity 5^((r-1)/2^28), I-form
    constexpr uint32_t C[f29::NL] = {0x1a27b370u, 0x1d788b88u, 0x0a3c6e0bu, 0x1fd3f9dau, 0x0f541c23u, 0x1e4ddf15u, 0x093d0e83u, 0x0ae32ca7u, 0x0005d90bu};
    Fe r;
#pragma unroll
    for (int i = 0; i < f29::NL; i++) r.v[i] = C[i];
    return r;
}
__device__ Fe pow_i(Fe b, uint64_t e) {   // I-form power
    Fe r = one_i();
    while (e) {
        if (e & 1) r = mul(r, b);
        b = mul(b, b);
        e >>= 1;
    }
    return r;
}
__device__ Fe inv_i(const Fe& a) {   // a^(r - 2), I-form in and out
    const uint64_t E[4] = {0x43e1f593efffffffull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
    Fe r = one_i(), b = a;
#pragma unroll 1
    for (int w = 0; w < 4; w++) {
        uint64_t e = E[w];
#pragma unroll 1
        for (int i = 0; i < 64; i++) {
            if (e & 1) r = mul(r, b);
            b = mul(b, b);
            e >>= 1;
        }
    }
    return r;
}

// the uniform values of one quotient call, made once on the device
struct Consts {
    Fe w4n, shift, alpha, alpha2, beta_d, beta_k1_d, beta_k2_d, gamma_d, n_i;   // _d: D-form, the rest I-form
    Fe zh_inv[4];    // 1 / (x^n - 1) on the coset: x^n = shift^n i^(k mod 4)
};
struct QuotientParams {
    const uint64_t* ev;   // [13 or 14][4n] evaluations on the coset, natural order: ql qr qm qo qk s1 s2 s3 l r o z (pi)
    const uint64_t* x;    // [4n] the points, I-form
    const uint64_t* linv; // [4n] 1 / (n (x - 1)), I-form
    const Consts* k;
    uint64_t* t;          // [4n]
    uint32_t log_n, has_pi;
};

// bad: set to 1 when Z_H vanishes on the coset (the shift lies in the size-4n subgroup): inv_i(0) = 0 would otherwise give a
// silently wrong quotient
__global__ void k_plonk_consts(Consts* out, uint32_t log_n, const uint64_t* in /* shift k1 k2 alpha beta gamma, D-form words */, uint32_t* bad) {
    if (threadIdx.x || blockIdx.x) return;
    Consts k;
    Fe w = root28_i();
    for (uint32_t i = log_n + 2; i < 28; i++) w = mul(w, w);
    k.w4n = w;
    k.shift = x32(ld(in, 0));
    const Fe k1 = x32(ld(in, 1)), k2 = x32(ld(in, 2));
    k.alpha = x32(ld(in, 3));
    k.alpha2 = mul(k.alpha, k.alpha);
    k.beta_d = ld(in, 4);
    k.beta_k1_d = mul(k.beta_d, k1);
    k.beta_k2_d = mul(k.beta_d, k2);
    k.gamma_d = ld(in, 5);
    Fe n = one_i();
    for (uint32_t i = 0; i < log_n; i++) n = add(n, n);
    k.n_i = n;
    const Fe sn = pow_i(k.shift, (uint64_t)1 << log_n), j = pow_i(w, (uint64_t)1 << log_n);   // j: a primitive 4th root of unity
    Fe jk = one_i();
    for (int q = 0; q < 4; q++) {
        const Fe den = sub(mul(sn, jk), one_i());
        if (f29::is_zero_mod<RM>(den)) *bad = 1;
        k.zh_inv[q] = inv_i(den);
        jk = mul(jk, j);
    }
    *out = k;
}

// One lane per run of DOMAIN_RUN points: x_i = shift w^i (one power, then a product per point) and 1 / (n (x_i - 1)) by
// Montgomery's batch inversion over the run (prefix products parked in the output, one inversion per run).
constexpr uint32_t DOMAIN_RUN = 64;
__global__ __launch_bounds__(64) void k_plonk_domain(const Consts* __restrict__ kp, uint32_t log_n, uint64_t* __restrict__ x_out,
                                                     uint64_t* __restrict__ linv_out) {
    const size_t N4 = (size_t)4 << log_n;
    const size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * DOMAIN_RUN;
    if (i0 >= N4) return;
    const size_t i1 = i0 + DOMAIN_RUN < N4 ? i0 + DOMAIN_RUN : N4;
    const Fe w = kp->w4n, n = kp->n_i, one = one_i();
    Fe x = mul(kp->shift, pow_i(w, i0)), acc = one;
#pragma unroll 1
    for (size_t i = i0; i < i1; i++) {
        st(x_out, i, x);
        acc = mul(acc, mul(n, sub(x, one)));
        st(linv_out, i, acc);
        x = mul(x, w);
    }
    Fe inv = inv_i(acc);
#pragma unroll 1
    for (size_t i = i1; i-- > i0;) {
        const Fe v = mul(n, sub(ld(x_out, i), one));
        const Fe prev = i > i0 ? ld(linv_out, i - 1) : one;
        st(linv_out, i, mul(inv, prev));
        inv = mul(inv, v);
    }
}

// Blinding in coefficient form: p(X) += (b_0 + b_1 X + ...)(X^n - 1), i.e. coefficient i loses b_i and coefficient n + i gains it.
// The coefficients lie in the 4n-array in bit-reversed order (what the coset FFT (DIT) reads): one lane per touched coefficient.
// ev: [poly][4n] elements; job t = (poly, i, sign) for the nine blinding scalars b (D-form words, in order l l r r o o z z z).
__global__ void k_plonk_blind(uint64_t* __restrict__ ev, const uint64_t* __restrict__ b, uint32_t log_n) {
    const uint32_t t = threadIdx.x;
    if (t >= 18) return;
    const uint32_t j = t >> 1, add_side = t & 1;                   // scalar j, the - b_i (0) or the + b_i (1) side
    const uint32_t poly = j < 6 ? 8 + (j >> 1) : 11, i = j < 6 ? (j & 1) : j - 6;
    const size_t N4 = (size_t)4 << log_n;
    const uint32_t idx = (add_side ? ((uint32_t)1 << log_n) : 0u) + i;
    const uint32_t pos = __brev(idx) >> (32 - (log_n + 2));
    uint64_t* col = ev + (size_t)poly * N4 * 4;
    const Fe cur = ld(col, pos), bj = ld(b, j);
    st(col, pos, add_side ? add(cur, bj) : sub(cur, bj));
}

__global__ __launch_bounds__(256) void k_plonk_quotient(QuotientParams p) {
    const size_t N4 = (size_t)4 << p.log_n;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N4) return;
    const Consts& k = *p.k;
    auto E = [&](int poly, size_t at) { return ld(p.ev + (size_t)poly * 4 * N4, at); };   // canonical words: already tight
    const Fe l = E(8, i), r = E(9, i), o = E(10, i);
    // gate: ql l + qr r + qm l r + qo o + qk (+ pi); every product of two data values is put back into D-form
    Fe gate = mul_dd(E(0, i), l);
    gate = add(gate, mul_dd(E(1, i), r));
    gate = add(gate, mul_dd(E(2, i), mul_dd(l, r)));
    gate = add(gate, mul_dd(E(3, i), o));
    gate = add(gate, E(4, i));
    if (p.has_pi) gate = add(gate, E(12, i));
    // permutation: the identity side on x, k1 x, k2 x and the sigma side on s1, s2, s3; z at the next point of the size-n
    // subgroup = four positions further on the size-4n coset
    const Fe x = ld(p.x, i), g = k.gamma_d;
    const Fe z = E(11, i), zn = E(11, (i + 4) & (N4 - 1));
    Fe f = mul_dd(add(add(l, mul(k.beta_d, x)), g), add(add(r, mul(k.beta_k1_d, x)), g));
    f = mul_dd(f, add(add(o, mul(k.beta_k2_d, x)), g));
    f = mul_dd(f, z);
    const Fe bi = x32(k.beta_d);   // beta, I-form: times a data value stays data
    Fe h = mul_dd(add(add(l, mul(E(5, i), bi)), g), add(add(r, mul(E(6, i), bi)), g));
    h = mul_dd(h, add(add(o, mul(E(7, i), bi)), g));
    h = mul_dd(h, zn);
    Fe t = add(gate, mul(sub(f, h), k.alpha));
    t = mul(t, k.zh_inv[i & 3]);
    // alpha^2 L1(x) (z - 1) / Z_H(x) = alpha^2 (z - 1) / (n (x - 1))
    t = add(t, mul(mul(sub(z, one_d()), ld(p.linv, i)), k.alpha2));
    st(p.t, i, t);
}

__global__ __launch_bounds__(256) void k_any_nonzero(const uint64_t* __restrict__ v, size_t words, uint32_t* __restrict__ flag) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool nz = i < words && v[i] != 0;
    if (__any(nz) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);   // one atomic per wave that saw anything
}

// ---- Horner scan: H_j = e_j + z H_{j-1} over a sequence in place (the synthetic division of a KZG opening) ----
// Three phases per level, runs of SCAN_RUN elements per lane: the runs' own Horner values, the same scan over those with
// z^SCAN_RUN, then every run again from its incoming value.  `rev`: the sequence is the array read backwards.
constexpr uint32_t SCAN_RUN = 64;
__global__ __launch_bounds__(64) void k_horner_local(const uint64_t* __restrict__ seq, size_t len, int rev, const uint64_t* __restrict__ zi,
                                                     uint64_t* __restrict__ runs) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, j0 = t * SCAN_RUN;
    if (j0 >= len) return;
    const size_t j1 = j0 + SCAN_RUN < len ? j0 + SCAN_RUN : len;
    const Fe z = ld(zi, 0);
    Fe h = f29::zero();
#pragma unroll 1
    for (size_t j = j0; j < j1; j++) h = add(ld(seq, rev ? len - 1 - j : j), mul(h, z));
    st(runs, t, h);
}
// the incoming value of run t is the finished value of run t - 1 one level up; a short last run of the level above has
// already been multiplied by the right power there, because every run but the last is full
__global__ __launch_bounds__(64) void k_horner_final(uint64_t* __restrict__ seq, size_t len, int rev, const uint64_t* __restrict__ zi,
                                                     const uint64_t* __restrict__ runs_done) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, j0 = t * SCAN_RUN;
    if (j0 >= len) return;
    const size_t j1 = j0 + SCAN_RUN < len ? j0 + SCAN_RUN : len;
    const Fe z = ld(zi, 0);
    Fe h = (t && runs_done) ? ld(runs_done, t - 1) : f29::zero();
#pragma unroll 1
    for (size_t j = j0; j < j1; j++) {
        const size_t at = rev ? len - 1 - j : j;
        h = add(ld(seq, at), mul(h, z));
        st(seq, at, h);
    }
}
// z (D-form words) -> zpow[k] = z^(SCAN_RUN^k) in I-form, k < levels
__global__ void k_horner_powers(const uint64_t* __restrict__ z_d, uint64_t* __restrict__ zpow, uint32_t levels) {
    if (threadIdx.x || blockIdx.x) return;
    Fe z = x32(ld(z_d, 0));
    for (uint32_t k = 0; k < levels; k++) {
        st(zpow, k, z);
        for (uint32_t s = 1; s < SCAN_RUN; s <<= 1) z = mul(z, z);
    }
}

// ---- the permutation's grand product: z_0 = 1, z_{i+1} = z_i prod_j (w_j + beta id_j + gamma) / (w_j + beta sigma_j + gamma) ----
// k_gp_ratios: one lane per run of GP_RUN rows - the rows' ratios with ONE inversion per run (Montgomery's trick: the
// denominators' prefix products parked in the output), in I-form, written one place on (slot i + 1 = ratio_i, slot 0 = 1), so
// that an inclusive product scan leaves z in place; the last row's ratio goes to `last` (z_{n-1} ratio_{n-1} must be 1).
constexpr uint32_t GP_RUN = 64;
struct GrandProductParams {
    const uint64_t *l, *r, *o, *s1, *s2, *s3;
    const uint64_t* sc;   // device: beta, gamma, k1, k2 (D-form words)
    uint64_t* z;          // n elements
    uint64_t* last;       // 1 element: ratio_{n-1}, I-form
    uint32_t log_n;
};
__global__ __launch_bounds__(64) void k_gp_ratios(GrandProductParams p) {
    const size_t n = (size_t)1 << p.log_n;
    const size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * GP_RUN;
    if (i0 >= n) return;
    const size_t i1 = i0 + GP_RUN < n ? i0 + GP_RUN : n;
    const Fe beta = ld(p.sc, 0), gamma = ld(p.sc, 1), beta_i = x32(beta);
    const Fe bk1 = mul(beta, x32(ld(p.sc, 2))), bk2 = mul(beta, x32(ld(p.sc, 3)));   // D x I -> D
    Fe w = root28_i();
    for (uint32_t k = p.log_n; k < 28; k++) w = mul(w, w);
    auto den_of = [&](size_t i) {   // I-form
        Fe d = mul_dd(add(add(ld(p.l, i), mul(ld(p.s1, i), beta_i)), gamma), add(add(ld(p.r, i), mul(ld(p.s2, i), beta_i)), gamma));
        return x32(mul_dd(d, add(add(ld(p.o, i), mul(ld(p.s3, i), beta_i)), gamma)));
    };
    Fe acc = one_i();
#pragma unroll 1
    for (size_t i = i0; i < i1; i++) {
        acc = mul(acc, den_of(i));
        if (i + 1 < i1) st(p.z, i + 1, acc);   // prefix product up to row i, parked where row i's ratio will go
    }
    Fe inv = inv_i(acc);
    Fe x = pow_i(w, i1 - 1);                   // the run's last point, then downwards by w^-1 = w^(n-1)
    const Fe winv = pow_i(w, n - 1);
#pragma unroll 1
    for (size_t i = i1; i-- > i0;) {
        const Fe prev = i > i0 ? ld(p.z, i) : one_i();   // prefix product up to row i - 1
        const Fe dinv = mul(inv, prev);
        inv = mul(inv, den_of(i));
        Fe num = mul_dd(add(add(ld(p.l, i), mul(beta, x)), gamma), add(add(ld(p.r, i), mul(bk1, x)), gamma));
        num = mul_dd(num, add(add(ld(p.o, i), mul(bk2, x)), gamma));
        const Fe ratio = mul(num, dinv);       // D x I -> D; the scan wants I-form
        if (i + 1 < n) st(p.z, i + 1, x32(ratio));
        else st(p.last, 0, x32(ratio));
        x = mul(x, winv);
    }
    if (i0 == 0) st(p.z, 0, one_i());
}
// inclusive product scan in place (I-form), runs of SCAN_RUN per lane; to_d: the finished values leave in D-form
__global__ __launch_bounds__(64) void k_mulscan_local(const uint64_t* __restrict__ seq, size_t len, uint64_t* __restrict__ runs) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, j0 = t * SCAN_RUN;
    if (j0 >= len) return;
    const size_t j1 = j0 + SCAN_RUN < len ? j0 + SCAN_RUN : len;
    Fe h = one_i();
#pragma unroll 1
    for (size_t j = j0; j < j1; j++) h = mul(h, ld(seq, j));
    st(runs, t, h);
}
__global__ __launch_bounds__(64) void k_mulscan_final(uint64_t* __restrict__ seq, size_t len, const uint64_t* __restrict__ runs_done, int to_d) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, j0 = t * SCAN_RUN;
    if (j0 >= len) return;
    const size_t j1 = j0 + SCAN_RUN < len ? j0 + SCAN_RUN : len;
    Fe h = (t && runs_done) ? ld(runs_done, t - 1) : one_i();
    const Fe c = one_d();   // the integer 2^256 mod r: I x plain -> D
#pragma unroll 1
    for (size_t j = j0; j < j1; j++) {
        h = mul(h, ld(seq, j));
        st(seq, j, to_d ? mul(h, c) : h);
    }
}
// closes = (z_{n-1} ratio_{n-1} == 1): what a consistent permutation gives
__global__ void k_gp_closes(const uint64_t* __restrict__ z_d, size_t n, const uint64_t* __restrict__ last_i, uint32_t* __restrict__ flag) {
    if (threadIdx.x || blockIdx.x) return;
    const Fe v = f29::canonical<RM>(mul(ld(z_d, n - 1), ld(last_i, 0)));   // D x I -> D
    const Fe o = f29::canonical<RM>(one_d());
    uint32_t diff = 0;
    for (int i = 0; i < f29::NL; i++) diff |= v.v[i] ^ o.v[i];
    *flag = diff ? 1u : 0u;
}

// Groth16's quotient on the coset of the SAME size: h = (a b - c) / (x^n - 1), the divisor one constant there
__global__ __launch_bounds__(256) void k_g16_pointwise(const uint64_t* __restrict__ a, const uint64_t* __restrict__ b, const uint64_t* __restrict__ c,
                                                       const uint64_t* __restrict__ zhinv_i, uint64_t* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    st(out, i, mul(sub(mul_dd(ld(a, i), ld(b, i)), ld(c, i)), ld(zhinv_i, 0)));
}
__global__ void k_g16_const(const uint64_t* __restrict__ shift_d, uint32_t log_n, uint64_t* __restrict__ zhinv_i, uint32_t* bad) {
    if (threadIdx.x || blockIdx.x) return;
    const Fe den = sub(pow_i(x32(ld(shift_d, 0)), (uint64_t)1 << log_n), one_i());
    *bad = f29::is_zero_mod<RM>(den) ? 1u : 0u;   // the shift lies in H: x^n - 1 vanishes on the whole coset
    st(zhinv_i, 0, inv_i(den));
}

// out[i] = sum_t scalars[t] polys[t][i]: linearisation and batching polynomials of the last round
__global__ void k_to_iform(uint64_t* __restrict__ sc, uint32_t count) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < count) st(sc, t, x32(ld(sc, t)));
}
constexpr uint32_t LINCOMB_MAX_TERMS = 16;
struct LincombParams {
    const uint64_t* polys[LINCOMB_MAX_TERMS];
    const uint64_t* sc_i;   // device: the scalars, I-form
    uint64_t* out;
    size_t m;
    uint32_t n_terms;
};
__global__ __launch_bounds__(256) void k_lincomb(LincombParams p) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.m) return;
    Fe acc = f29::zero();
    for (uint32_t t = 0; t < p.n_terms; t++) acc = add(acc, mul(ld(p.polys[t], i), ld(p.sc_i, t)));
    st(p.out, i, acc);
}

}  // namespace bnp
}  // namespace nlx

using namespace nlx;

namespace {
// H_j in place over seq (len elements, optionally read backwards); d_zpow: the powers z^(64^k), device, 4 words each
int32_t horner_scan(nlx_ctx* ctx, uint64_t* seq, size_t len, int rev, const uint64_t* d_zpow, uint32_t level, std::vector<void*>& tmp) {
    hipStream_t st = ctx->stream;
    const size_t runs = (len + bnp::SCAN_RUN - 1) / bnp::SCAN_RUN;
    if (runs <= 1) {
        hipLaunchKernelGGL(bnp::k_horner_final, dim3(1), dim3(64), 0, st, seq, len, rev, d_zpow + 4 * level, (const uint64_t*)nullptr);
        return NLX_OK;
    }
    uint64_t* d_runs = (uint64_t*)ctx->alloc(runs * 32);
    if (!d_runs) return NLX_E_NOMEM;
    tmp.push_back(d_runs);
    const unsigned blocks = (unsigned)((runs + 63) / 64);
    hipLaunchKernelGGL(bnp::k_horner_local, dim3(blocks), dim3(64), 0, st, seq, len, rev, d_zpow + 4 * level, d_runs);
    const int32_t rc = horner_scan(ctx, d_runs, runs, 0, d_zpow, level + 1, tmp);
    if (rc) return rc;
    hipLaunchKernelGGL(bnp::k_horner_final, dim3(blocks), dim3(64), 0, st, seq, len, rev, d_zpow + 4 * level, d_runs);
    return NLX_OK;
}

// inclusive product scan of seq (I-form) in place; to_d on the outermost level only
int32_t mul_scan(nlx_ctx* ctx, uint64_t* seq, size_t len, int to_d, std::vector<void*>& tmp) {
    hipStream_t st = ctx->stream;
    const size_t runs = (len + bnp::SCAN_RUN - 1) / bnp::SCAN_RUN;
    if (runs <= 1) {
        hipLaunchKernelGGL(bnp::k_mulscan_final, dim3(1), dim3(64), 0, st, seq, len, (const uint64_t*)nullptr, to_d);
        return NLX_OK;
    }
    uint64_t* d_runs = (uint64_t*)ctx->alloc(runs * 32);
    if (!d_runs) return NLX_E_NOMEM;
    tmp.push_back(d_runs);
    const unsigned blocks = (unsigned)((runs + 63) / 64);
    hipLaunchKernelGGL(bnp::k_mulscan_local, dim3(blocks), dim3(64), 0, st, seq, len, d_runs);
    const int32_t rc = mul_scan(ctx, d_runs, runs, 0, tmp);
    if (rc) return rc;
    hipLaunchKernelGGL(bnp::k_mulscan_final, dim3(blocks), dim3(64), 0, st, seq, len, d_runs, to_d);
    return NLX_OK;
}
}  // namespace

extern "C" {

int32_t nlx_bn254_plonk_grand_product(nlx_ctx* ctx, uint32_t log_n, const uint64_t* l, const uint64_t* r, const uint64_t* o,
                                      const uint64_t* s1, const uint64_t* s2, const uint64_t* s3, const uint64_t beta[4],
                                      const uint64_t gamma[4], const uint64_t k1[4], const uint64_t k2[4], uint64_t* z_out,
                                      int32_t* closes) NLX_TRY {
    if (!ctx) return NLX_E_INVAL;
    if (!l || !r || !o || !s1 || !s2 || !s3 || !beta || !gamma || !k1 || !k2 || !z_out) return ctx->fail(NLX_E_INVAL, "NULL argument");
    if (log_n < 1 || log_n > 28) return ctx->fail(NLX_E_RANGE, "log_n must be in [1, 28]");
    if (is_device_ptr(beta) || is_device_ptr(gamma) || is_device_ptr(k1) || is_device_ptr(k2)) return ctx->fail(NLX_E_INVAL, "the challenges and shifts are host values");
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    const size_t n = (size_t)1 << log_n;
    std::vector<void*> tmp;
    auto done = [&](int32_t code) {
        (void)hipStreamSynchronize(st);
        for (void* p : tmp) ctx->release(p);
        return code;
    };
    const uint64_t* in[6] = {l, r, o, s1, s2, s3};
    const uint64_t* dev[6];
    for (int i = 0; i < 6; i++) {
        if (is_device_ptr(in[i])) { dev[i] = in[i]; continue; }
        uint64_t* d = (uint64_t*)ctx->alloc(n * 32);
        if (!d) return done(NLX_E_NOMEM);
        tmp.push_back(d);
        hipError_t e = hipMemcpyAsync(d, in[i], n * 32, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return done(ctx->hip_fail(e, "hipMemcpyAsync"));
        dev[i] = d;
    }
    uint64_t* d_z = is_device_ptr(z_out) ? z_out : (uint64_t*)ctx->alloc(n * 32);
    uint64_t* d_small = (uint64_t*)ctx->alloc(4 * 32 + 32 + 64);
    if (d_z && d_z != z_out) tmp.push_back(d_z);
    if (d_small) tmp.push_back(d_small);
    if (!d_z || !d_small) return done(NLX_E_NOMEM);
    uint64_t h[16];
    memcpy(h, beta, 32); memcpy(h + 4, gamma, 32); memcpy(h + 8, k1, 32); memcpy(h + 12, k2, 32);
    hipError_t e = hipMemcpyAsync(d_small, h, sizeof h, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return done(ctx->hip_fail(e, "hipMemcpyAsync"));
    uint32_t* d_flag = (uint32_t*)(d_small + 20);
    bnp::GrandProductParams gp{dev[0], dev[1], dev[2], dev[3], dev[4], dev[5], d_small, d_z, d_small + 16, log_n};
    hipLaunchKernelGGL(bnp::k_gp_ratios, dim3((unsigned)(((n + bnp::GP_RUN - 1) / bnp::GP_RUN + 63) / 64)), dim3(64), 0, st, gp);
    int32_t rc = mul_scan(ctx, d_z, n, 1, tmp);
    if (rc) return done(rc);
    hipLaunchKernelGGL(bnp::k_gp_closes, dim3(1), dim3(1), 0, st, d_z, n, d_small + 16, d_flag);
    uint32_t flag = 0;
    e = hipMemcpyAsync(&flag, d_flag, 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && d_z != z_out) e = hipMemcpyAsync(z_out, d_z, n * 32, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return done(ctx->hip_fail(e, "copy out"));
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return done(ctx->hip_fail(le, "kernel launch"));
    if (closes) *closes = flag ? 0 : 1;
    return done(NLX_OK);
} NLX_CATCH(ctx)

// a host scalar handed over as fr.Element words (Montgomery residue): any value below r is a residue, anything else is not an
// fr.Element and the 29-bit-limb arithmetic's bounds would not hold for it
static bool fr_words_below_r(const uint64_t* w) {
    static const uint64_t R[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
    for (int i = 3; i >= 0; i--) {
        if (w[i] < R[i]) return true;
        if (w[i] > R[i]) return false;
    }
    return false;
}

int32_t nlx_bn254_groth16_quotient(nlx_ctx* ctx, uint32_t log_n, const uint64_t* a, const uint64_t* b, const uint64_t* c,
                                   const uint64_t coset_shift[4], uint64_t* h_out) NLX_TRY {
    if (!ctx) return NLX_E_INVAL;
    if (!a || !b || !c || !coset_shift || !h_out) return ctx->fail(NLX_E_INVAL, "NULL argument");
    if (log_n < 1 || log_n > 28) return ctx->fail(NLX_E_RANGE, "log_n must be in [1, 28]");
    if (is_device_ptr(coset_shift)) return ctx->fail(NLX_E_INVAL, "the coset shift is a host value");
    if (!fr_words_below_r(coset_shift)) return ctx->fail(NLX_E_RANGE, "the coset shift is not below r");
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    const size_t n = (size_t)1 << log_n;
    std::vector<void*> tmp;
    auto done = [&](int32_t code) {
        (void)hipStreamSynchronize(st);
        for (void* p : tmp) ctx->release(p);
        return code;
    };
    uint64_t* d_abc = (uint64_t*)ctx->alloc(3 * n * 32);
    uint64_t* d_small = (uint64_t*)ctx->alloc(96);   // shift | 1 / (shift^n - 1) | flag
    if (d_abc) tmp.push_back(d_abc);
    if (d_small) tmp.push_back(d_small);
    if (!d_abc || !d_small) return done(NLX_E_NOMEM);
    const uint64_t* in[3] = {a, b, c};
    for (int i = 0; i < 3; i++) {
        hipError_t e = hipMemcpyAsync(d_abc + (size_t)i * n * 4, in[i], n * 32, is_device_ptr(in[i]) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return done(ctx->hip_fail(e, "hipMemcpyAsync"));
    }
    hipError_t e = hipMemcpy(d_small, coset_shift, 32, hipMemcpyHostToDevice);
    if (e != hipSuccess) return done(ctx->hip_fail(e, "hipMemcpy"));
    hipLaunchKernelGGL(bnp::k_g16_const, dim3(1), dim3(1), 0, st, d_small, log_n, d_small + 4, (uint32_t*)(d_small + 8));
    {
        uint32_t bad = 0;
        e = hipMemcpyAsync(&bad, d_small + 8, 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) return done(ctx->hip_fail(e, "hipMemcpyAsync"));
        if (bad) return done(ctx->fail(NLX_E_INVAL, "coset shift lies in the evaluation subgroup (x^n - 1 vanishes on the coset)"));
    }
    // FFTInverse(DIF) -> FFT(DIT, OnCoset): no reordering pass; then a b - c over the coset's constant x^n - 1
    int32_t rc = nlx_bn254_ntt_batch_coset(ctx, d_abc, 3, log_n, 1, NLX_BN254_MONTGOMERY | NLX_BN254_BITREV_OUT, nullptr);
    if (!rc) rc = nlx_bn254_ntt_batch_coset(ctx, d_abc, 3, log_n, 0, NLX_BN254_MONTGOMERY | NLX_BN254_BITREV_IN, coset_shift);
    if (rc) return done(rc);
    hipLaunchKernelGGL(bnp::k_g16_pointwise, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_abc, d_abc + n * 4, d_abc + 2 * n * 4,
                       d_small + 4, d_abc, n);
    rc = nlx_bn254_ntt_batch_coset(ctx, d_abc, 1, log_n, 1, NLX_BN254_MONTGOMERY, coset_shift);
    if (rc) return done(rc);
    e = hipMemcpyAsync(h_out, d_abc, n * 32, is_device_ptr(h_out) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return done(ctx->hip_fail(e, "copy out"));
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return done(ctx->hip_fail(le, "kernel launch"));
    return done(NLX_OK);
} NLX_CATCH(ctx)

int32_t nlx_bn254_fr_lincomb(nlx_ctx* ctx, uint64_t m, uint32_t n_terms, const uint64_t* const* polys, const uint64_t* scalars,
                             uint64_t* out) NLX_TRY {
    if (!ctx) return NLX_E_INVAL;
    if (!polys || !scalars || !out) return ctx->fail(NLX_E_INVAL, "NULL argument");
    if (n_terms < 1 || n_terms > bnp::LINCOMB_MAX_TERMS || m < 1 || m > ((uint64_t)1 << 28)) return ctx->fail(NLX_E_RANGE, "1 .. 16 terms of 1 .. 2^28 elements");
    if (is_device_ptr(polys) || is_device_ptr(scalars)) return ctx->fail(NLX_E_INVAL, "the pointer array and the scalars are host arrays");
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    std::vector<void*> tmp;
    auto done = [&](int32_t code) {
        (void)hipStreamSynchronize(st);
        for (void* p : tmp) ctx->release(p);
        return code;
    };
    bnp::LincombParams lp{};
    for (uint32_t t = 0; t < n_terms; t++) {
        if (!polys[t]) return done(ctx->fail(NLX_E_INVAL, "NULL polynomial"));
        if (is_device_ptr(polys[t])) { lp.polys[t] = polys[t]; continue; }
        uint64_t* d = (uint64_t*)ctx->alloc(m * 32);
        if (!d) return done(NLX_E_NOMEM);
        tmp.push_back(d);
        hipError_t e = hipMemcpyAsync(d, polys[t], m * 32, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return done(ctx->hip_fail(e, "hipMemcpyAsync"));
        lp.polys[t] = d;
    }
    uint64_t* d_sc = (uint64_t*)ctx->alloc((size_t)n_terms * 32);
    uint64_t* d_out = is_device_ptr(out) ? out : (uint64_t*)ctx->alloc(m * 32);
    if (d_sc) tmp.push_back(d_sc);
    if (d_out && d_out != out) tmp.push_back(d_out);
    if (!d_sc || !d_out) return done(NLX_E_NOMEM);
    hipError_t e = hipMemcpy(d_sc, scalars, (size_t)n_terms * 32, hipMemcpyHostToDevice);
    if (e != hipSuccess) return done(ctx->hip_fail(e, "hipMemcpy"));
    hipLaunchKernelGGL(bnp::k_to_iform, dim3(1), dim3(64), 0, st, d_sc, n_terms);
    lp.sc_i = d_sc; lp.out = d_out; lp.m = m; lp.n_terms = n_terms;
    hipLaunchKernelGGL(bnp::k_lincomb, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, lp);
    if (d_out != out) e = hipMemcpyAsync(out, d_out, m * 32, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return done(ctx->hip_fail(e, "copy out"));
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return done(ctx->hip_fail(le, "kernel launch"));
    return done(NLX_OK);
} NLX_CATCH(ctx)

int32_t nlx_bn254_plonk_quotient(nlx_ctx* ctx, const nlx_bn254_plonk_quotient_args* a, uint64_t* t_out, int32_t* high_chunk_is_zero) NLX_TRY {
    if (!ctx) return NLX_E_INVAL;
    if (!a || !t_out) return ctx->fail(NLX_E_INVAL, "NULL argument");
    const uint64_t* polys[13] = {a->ql, a->qr, a->qm, a->qo, a->qk, a->s1, a->s2, a->s3, a->l, a->r, a->o, a->z, a->pi};
    for (int i = 0; i < 12; i++)
        if (!polys[i]) return ctx->fail(NLX_E_INVAL, "NULL polynomial");
    if (a->log_n < 2 || a->log_n > 26) return ctx->fail(NLX_E_RANGE, "log_n must be in [2, 26] (the coset has four times the points)");
    const bool blinded = (a->flags & NLX_BN254_PLONK_BLINDED) != 0;
    if ((a->flags & ~NLX_BN254_PLONK_BLINDED) != NLX_BN254_MONTGOMERY) return ctx->fail(NLX_E_UNSUPPORTED, "elements must be fr.Element words (flags = NLX_BN254_MONTGOMERY)");
    if (blinded) {
        if (!a->blinding || is_device_ptr(a->blinding)) return ctx->fail(NLX_E_INVAL, "the blinding scalars are host values (nine elements of four words)");
        if (a->log_n < 3) return ctx->fail(NLX_E_RANGE, "a blinded quotient needs log_n >= 3 (3 n + 6 coefficients on 4 n points)");
        for (int i = 0; i < 9; i++)
            if (!fr_words_below_r(a->blinding + 4 * i)) return ctx->fail(NLX_E_RANGE, "a blinding scalar is not below r");
    }
    {
        const uint64_t* sc[6] = {a->coset_shift, a->k1, a->k2, a->alpha, a->beta, a->gamma};
        for (const uint64_t* q : sc)
            if (!q || is_device_ptr(q)) return ctx->fail(NLX_E_INVAL, "the challenges and shifts are host values (four words each)");
        for (const uint64_t* q : sc)
            if (!fr_words_below_r(q)) return ctx->fail(NLX_E_RANGE, "a challenge or shift is not below r (fr.Element words are residues)");
    }
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    const uint32_t log_n = a->log_n, P = a->pi ? 13 : 12;
    const size_t n = (size_t)1 << log_n, N4 = 4 * n;
    std::vector<void*> tmp;
    auto dalloc = [&](size_t bytes) -> uint64_t* {
        void* p = ctx->alloc(bytes);
        if (p) tmp.push_back(p);
        return (uint64_t*)p;
    };
    int32_t rc = NLX_OK;
    uint64_t* d_in = dalloc((size_t)P * n * 32);
    uint64_t* d_ev = dalloc((size_t)P * N4 * 32);
    uint64_t* d_x = dalloc(N4 * 32);
    uint64_t* d_linv = dalloc(N4 * 32);
    uint64_t* d_t = dalloc(N4 * 32);
    uint64_t* d_small = dalloc(6 * 32 + sizeof(bnp::Consts) + 64 + 9 * 32);
    auto done = [&](int32_t code) {
        (void)hipStreamSynchronize(st);
        for (void* p : tmp) ctx->release(p);
        return code;
    };
    if (!d_in || !d_ev || !d_x || !d_linv || !d_t || !d_small) return done(NLX_E_NOMEM);
    bnp::Consts* d_k = (bnp::Consts*)(d_small + 6 * 4);
    uint32_t* d_flag = (uint32_t*)((char*)d_k + sizeof(bnp::Consts));
    uint64_t* d_blind = (uint64_t*)((char*)d_flag + 64);
    {
        uint64_t h[6 * 4];
        const uint64_t* src[6] = {a->coset_shift, a->k1, a->k2, a->alpha, a->beta, a->gamma};
        for (int i = 0; i < 6; i++) memcpy(h + 4 * i, src[i], 32);
        hipError_t e = hipMemcpyAsync(d_small, h, sizeof h, hipMemcpyHostToDevice, st);
        if (e == hipSuccess && blinded) e = hipMemcpyAsync(d_blind, a->blinding, 9 * 32, hipMemcpyHostToDevice, st);   // caller-owned: outlives the sync below
        if (e == hipSuccess) e = hipMemsetAsync(d_flag, 0, 8, st);   // [0] high chunk non-zero, [1] Z_H vanishes on the coset
        if (e == hipSuccess) e = hipStreamSynchronize(st);   // h leaves scope
        if (e != hipSuccess) return done(ctx->hip_fail(e, "hipMemcpyAsync"));
    }
    for (uint32_t i = 0; i < P; i++) {
        hipError_t e = hipMemcpyAsync(d_in + (size_t)i * n * 4, polys[i], n * 32,
                                      is_device_ptr(polys[i]) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return done(ctx->hip_fail(e, "hipMemcpyAsync"));
    }
    hipLaunchKernelGGL(bnp::k_plonk_consts, dim3(1), dim3(1), 0, st, d_k, log_n, d_small, d_flag + 1);
    hipLaunchKernelGGL(bnp::k_plonk_domain, dim3((unsigned)(((N4 + bnp::DOMAIN_RUN - 1) / bnp::DOMAIN_RUN + 63) / 64)), dim3(64), 0, st, d_k, log_n, d_x, d_linv);
    // 1. FFTInverse(DIF): values on H -> coefficients in bit-reversed order
    rc = nlx_bn254_ntt_batch_coset(ctx, d_in, P, log_n, 1, NLX_BN254_MONTGOMERY | NLX_BN254_BITREV_OUT, nullptr);
    if (rc) return done(rc);
    // 2. zero-padded to 4n in bit-reversed order (coefficient at position p of n sits at 4 p of 4n), then FFT(DIT, OnCoset)
    {
        hipError_t e = hipMemsetAsync(d_ev, 0, (size_t)P * N4 * 32, st);
        for (uint32_t i = 0; i < P && e == hipSuccess; i++)
            e = hipMemcpy2DAsync(d_ev + (size_t)i * N4 * 4, 128, d_in + (size_t)i * n * 4, 32, 32, n, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return done(ctx->hip_fail(e, "hipMemcpy2DAsync"));
    }
    if (blinded) hipLaunchKernelGGL(bnp::k_plonk_blind, dim3(1), dim3(64), 0, st, d_ev, d_blind, log_n);
    rc = nlx_bn254_ntt_batch_coset(ctx, d_ev, P, log_n + 2, 0, NLX_BN254_MONTGOMERY | NLX_BN254_BITREV_IN, a->coset_shift);
    if (rc) return done(rc);
    // 3. the quotient's values on the coset
    bnp::QuotientParams qp{d_ev, d_x, d_linv, d_k, d_t, log_n, a->pi ? 1u : 0u};
    ctx->begin_kernel("plonk_quotient", 32.0 * N4 * (P + 4));
    hipLaunchKernelGGL(bnp::k_plonk_quotient, dim3((unsigned)((N4 + 255) / 256)), dim3(256), 0, st, qp);
    ctx->end_kernel();
    // 4. back to coefficients
    rc = nlx_bn254_ntt_batch_coset(ctx, d_t, 1, log_n + 2, 1, NLX_BN254_MONTGOMERY, a->coset_shift);
    if (rc) return done(rc);
    // coefficients that must vanish: 3 n .. 4 n - 1 (3 n + 6 .. with blinding: the blinded wires raise the quotient's degree by six)
    const size_t t_keep = blinded ? 3 * n + 6 : 3 * n;
    hipLaunchKernelGGL(bnp::k_any_nonzero, dim3((unsigned)(((N4 - t_keep) * 4 + 255) / 256)), dim3(256), 0, st, d_t + t_keep * 4, (N4 - t_keep) * 4, d_flag);
    uint32_t flags[2] = {0, 0};
    hipError_t e = hipMemcpyAsync(flags, d_flag, 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return done(ctx->hip_fail(e, "copy out"));
    if (flags[1]) return done(ctx->fail(NLX_E_INVAL, "coset shift lies in the evaluation subgroup (x^n - 1 vanishes on the coset)"));
    e = hipMemcpyAsync(t_out, d_t, (blinded ? N4 : 3 * n) * 32, is_device_ptr(t_out) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return done(ctx->hip_fail(e, "copy out"));
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return done(ctx->hip_fail(le, "kernel launch"));
    const uint32_t flag = flags[0];
    if (high_chunk_is_zero) *high_chunk_is_zero = flag ? 0 : 1;
    return done(NLX_OK);
} NLX_CATCH(ctx)

int32_t nlx_bn254_kzg_open(nlx_ctx* ctx, const uint64_t* coeffs, uint64_t m, const uint64_t zeta[4], const uint64_t* srs,
                           uint64_t y_out[4], uint64_t* quotient_out, uint64_t proof_out[8]) NLX_TRY {
    if (!ctx) return NLX_E_INVAL;
    if (!coeffs || !zeta || !y_out) return ctx->fail(NLX_E_INVAL, "NULL argument");
    if (m < 2 || m > ((uint64_t)1 << 28)) return ctx->fail(NLX_E_RANGE, "2 <= coefficients <= 2^28");
    if (proof_out && !srs) return ctx->fail(NLX_E_INVAL, "an opening proof needs the SRS");
    if (is_device_ptr(zeta) || is_device_ptr(y_out)) return ctx->fail(NLX_E_INVAL, "the point and the value are host words");
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    std::vector<void*> tmp;
    auto done = [&](int32_t code) {
        (void)hipStreamSynchronize(st);
        for (void* p : tmp) ctx->release(p);
        return code;
    };
    uint64_t* d_h = (uint64_t*)ctx->alloc(m * 32);
    uint64_t* d_z = (uint64_t*)ctx->alloc(32 + 8 * 32);
    if (d_h) tmp.push_back(d_h);
    if (d_z) tmp.push_back(d_z);
    if (!d_h || !d_z) return done(NLX_E_NOMEM);
    hipError_t e = hipMemcpyAsync(d_h, coeffs, m * 32, is_device_ptr(coeffs) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpy(d_z, zeta, 32, hipMemcpyHostToDevice);
    if (e != hipSuccess) return done(ctx->hip_fail(e, "hipMemcpy"));
    uint32_t levels = 1;
    for (uint64_t len = m; len > bnp::SCAN_RUN; len = (len + bnp::SCAN_RUN - 1) / bnp::SCAN_RUN) levels++;
    hipLaunchKernelGGL(bnp::k_horner_powers, dim3(1), dim3(1), 0, st, d_z, d_z + 4, levels);
    // H_j = p_{m-1-j} + zeta H_{j-1}: the array read from the top coefficient down; afterwards position i holds h_i with
    // h_0 = p(zeta) and q_{i-1} = h_i: the quotient is the array shifted down by one
    int32_t rc = horner_scan(ctx, d_h, m, 1, d_z + 4, 0, tmp);
    if (rc) return done(rc);
    e = hipMemcpyAsync(y_out, d_h, 32, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && quotient_out)
        e = hipMemcpyAsync(quotient_out, d_h + 4, (m - 1) * 32, is_device_ptr(quotient_out) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return done(ctx->hip_fail(e, "copy out"));
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return done(ctx->hip_fail(le, "kernel launch"));
    if (proof_out) {
        rc = nlx_bn254_msm_g1(ctx, srs, d_h + 4, m - 1, NLX_BN254_MONTGOMERY, proof_out);
        if (rc) return done(rc);
    }
    return done(NLX_OK);
} NLX_CATCH(ctx)

}  // extern "C"
