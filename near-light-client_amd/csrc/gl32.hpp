// Goldilocks arithmetic on 32-bit limb pairs with hand-placed carry chains (gfx950 inline asm).
//
// Measured on MI355X (tools/ubench, profiles/r02_valu_ubench.txt): at full occupancy v_mad_u64_u32 issues at ~2.0 ns per
// wave-instruction per SIMD like every other integer multiply - and so does every VCC-chained add / subtract / compare
// (v_add_co_u32 + v_addc_co_u32: 1.99 ns each, v_cmp + v_cndmask: 1.82 ns); only carry-free ALU ops reach 1.17 ns.  A
// field multiply therefore costs its INSTRUCTION COUNT x 2 ns, whatever the instructions are, and a multiply-add that
// replaces several carry-chain instructions is a straight win: `r = base + w2 * (2^32 - 1)` is ONE v_mad_u64_u32 (its
// carry-out lands in VCC) instead of four add / subtract-with-carry instructions, and `r += carry ? 2^32 - 1 : 0` is a
// v_cndmask + one more v_mad_u64_u32 (mask * 1 + r) instead of three.  Multiply + reduction: 16 instructions (20 in
// round 1, ~29 from hipcc's u64 code); the fold of an MDS accumulator pair: 4 (9 in round 1).
#pragma once
#include "gl.hpp"
#include "poseidon_blocks.inc"   // generated tables of the fused partial rounds (tools/gen_poseidon_blocks.py)

namespace gl32 {

struct F {
    uint32_t lo, hi;  // value = lo + hi * 2^32, any u64 ("loose")
};

__device__ __forceinline__ F from_u64(uint64_t x) { return F{(uint32_t)x, (uint32_t)(x >> 32)}; }
__device__ __forceinline__ uint64_t to_u64(F x) { return ((uint64_t)x.hi << 32) | x.lo; }

// base + w2 * EPS (mod p) -> loose.  2^64 = EPS (mod p): the multiply-add's carry out of 2^64 is worth EPS, and adding it
// cannot carry again (base + w2 * EPS - 2^64 <= 2^64 - 2^33).  Three instructions: multiply-add, carry -> mask, mask * 1 + r.
__device__ __forceinline__ F add_w2_eps(uint64_t base, uint32_t w2) {
    uint64_t r;
    uint32_t t;
    asm("v_mad_u64_u32 %[r], vcc, %[w2], -1, %[base]\n\t"
        "v_cndmask_b32_e64 %[t], 0, -1, vcc\n\t"
        "v_mad_u64_u32 %[r], vcc, %[t], 1, %[r]"
        : [r] "=&v"(r), [t] "=&v"(t)
        : [w2] "v"(w2), [base] "v"(base)
        : "vcc");
    return from_u64(r);
}

// (w3:w2:w1:w0) mod p -> loose
//   r = (w1:w0) - w3 [borrow -> -EPS] + (w2 * EPS) [carry -> +EPS]
__device__ __forceinline__ F reduce128(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
    uint32_t r0, r1, t0;
    asm("v_sub_co_u32 %[r0], vcc, %[w0], %[w3]\n\t"
        "v_subbrev_co_u32 %[r1], vcc, 0, %[w1], vcc\n\t"
        "v_cndmask_b32_e64 %[t0], 0, -1, vcc\n\t"
        "v_sub_co_u32 %[r0], vcc, %[r0], %[t0]\n\t"
        "v_subbrev_co_u32 %[r1], vcc, 0, %[r1], vcc"
        : [r0] "=&v"(r0), [r1] "=&v"(r1), [t0] "=&v"(t0)
        : [w0] "v"(w0), [w1] "v"(w1), [w3] "v"(w3)
        : "vcc");
    return add_w2_eps(((uint64_t)r1 << 32) | r0, w2);
}

__device__ __forceinline__ uint64_t mad_u64_one(uint32_t a, uint64_t c);

// loose * loose -> loose.  The 128-bit product as a chain of five multiply-adds none of which can overflow
//   p0 = al bl;  p1 = al bh + hi(p0);  p2 = ah bl + lo(p1);  p3 = ah bh + hi(p1) + hi(p2)   (w0 = lo(p0), w1 = lo(p2), (w3:w2) = p3)
// instead of four products and four carry-chain additions: the three zero-extended addends cost a v_mov each, which issues at
// half the price of an add-with-carry (profiles/r04_valu_issue_rates_v1.txt).
__device__ __forceinline__ F mul(F a, F b) {
#ifdef NLX_GL32_MUL_CARRY_CHAINS
    uint64_t t = (uint64_t)a.lo * b.lo;
    uint64_t m = (uint64_t)a.lo * b.hi;
    uint64_t x = (uint64_t)a.hi * b.hi;
    uint64_t mid;
    uint32_t x1;
    // mid = a.hi * b.lo + m, its carry (weight 2^96) goes straight into the top limb
    asm("v_mad_u64_u32 %[mid], vcc, %[ah], %[bl], %[m]\n\t"
        "v_addc_co_u32 %[x1], vcc, 0, %[xh], vcc"
        : [mid] "=&v"(mid), [x1] "=v"(x1)
        : [ah] "v"(a.hi), [bl] "v"(b.lo), [m] "v"(m), [xh] "v"((uint32_t)(x >> 32))
        : "vcc");
    // limbs w1..w3, then (w1:w0) - w3 [borrow -> -EPS]; the + w2 * EPS [carry -> +EPS] half is add_w2_eps
    uint32_t r0, r1, w1, w2, w3, t0;
    asm("v_add_co_u32 %[w1], vcc, %[th], %[m0]\n\t"
        "v_addc_co_u32 %[w2], vcc, %[x0], %[m1], vcc\n\t"
        "v_addc_co_u32 %[w3], vcc, 0, %[x1], vcc\n\t"
        "v_sub_co_u32 %[r0], vcc, %[w0], %[w3]\n\t"
        "v_subbrev_co_u32 %[r1], vcc, 0, %[w1], vcc\n\t"
        "v_cndmask_b32_e64 %[t0], 0, -1, vcc\n\t"
        "v_sub_co_u32 %[r0], vcc, %[r0], %[t0]\n\t"
        "v_subbrev_co_u32 %[r1], vcc, 0, %[r1], vcc"
        : [r0] "=&v"(r0), [r1] "=&v"(r1), [w1] "=&v"(w1), [w2] "=&v"(w2), [w3] "=&v"(w3), [t0] "=&v"(t0)
        : [w0] "v"((uint32_t)t), [th] "v"((uint32_t)(t >> 32)), [m0] "v"((uint32_t)mid), [x0] "v"((uint32_t)x),
          [m1] "v"((uint32_t)(mid >> 32)), [x1] "v"(x1)
        : "vcc");
    return add_w2_eps(((uint64_t)r1 << 32) | r0, w2);
#else
    // plain C++ for the four products: hipcc selects exactly v_mad_u64_u32 with a (value : 0) register pair for each addend
    // and spaces nothing (between asm statements it puts an s_nop); the last addition as x * 1 + c saves the pair
    const uint64_t p0 = (uint64_t)a.lo * b.lo;
    const uint64_t p1 = (uint64_t)a.lo * b.hi + (p0 >> 32);
    const uint64_t p2 = (uint64_t)a.hi * b.lo + (uint32_t)p1;
    uint64_t p3 = (uint64_t)a.hi * b.hi + (p1 >> 32);
    p3 = mad_u64_one((uint32_t)(p2 >> 32), p3);
    return reduce128((uint32_t)p0, (uint32_t)p2, (uint32_t)p3, (uint32_t)(p3 >> 32));
#endif
}

// loose + canonical constant -> loose
__device__ __forceinline__ F add_const(F a, uint64_t c) {
    F r;
    uint32_t t;
    asm("v_add_co_u32 %[r0], vcc, %[a0], %[c0]\n\t"
        "v_addc_co_u32 %[r1], vcc, %[a1], %[c1], vcc\n\t"
        "v_cndmask_b32_e64 %[t], 0, -1, vcc\n\t"
        "v_add_co_u32 %[r0], vcc, %[r0], %[t]\n\t"
        "v_addc_co_u32 %[r1], vcc, 0, %[r1], vcc"
        : [r0] "=&v"(r.lo), [r1] "=&v"(r.hi), [t] "=&v"(t)
        : [a0] "v"(a.lo), [a1] "v"(a.hi), [c0] "s"((uint32_t)c), [c1] "v"((uint32_t)(c >> 32))
        : "vcc");
    return r;
}

// same with a per-lane (vector) constant
__device__ __forceinline__ F add_const_v(F a, uint64_t c) {
    F r;
    uint32_t t;
    asm("v_add_co_u32 %[r0], vcc, %[a0], %[c0]\n\t"
        "v_addc_co_u32 %[r1], vcc, %[a1], %[c1], vcc\n\t"
        "v_cndmask_b32_e64 %[t], 0, -1, vcc\n\t"
        "v_add_co_u32 %[r0], vcc, %[r0], %[t]\n\t"
        "v_addc_co_u32 %[r1], vcc, 0, %[r1], vcc"
        : [r0] "=&v"(r.lo), [r1] "=&v"(r.hi), [t] "=&v"(t)
        : [a0] "v"(a.lo), [a1] "v"(a.hi), [c0] "v"((uint32_t)c), [c1] "v"((uint32_t)(c >> 32))
        : "vcc");
    return r;
}

__device__ __forceinline__ F sbox7(F x) {
    F x2 = mul(x, x);
    F x4 = mul(x2, x2);
    F x3 = mul(x, x2);
    return mul(x3, x4);
}

// value = al + ah * 2^32 (al, ah u64 accumulators below 2^42: twelve products of a 32-bit limb and a constant < 2^6, plus a
// round constant's limb) -> loose.  2^64 = EPS (mod p), so ah's high word folds in first, as a multiply-add that cannot
// carry (al + ah1 * EPS < 2^42 + 2^42); ah's low word then lands on the high word of the sum, and only that addition can
// carry - once, worth EPS, and adding it cannot carry again.  Four instructions (five when the limbs were summed first).
__device__ __forceinline__ F fold_acc(uint64_t al, uint64_t ah) {
    uint64_t t;
    asm("v_mad_u64_u32 %[t], vcc, %[ah1], -1, %[al]" : [t] "=&v"(t) : [ah1] "v"((uint32_t)(ah >> 32)), [al] "v"(al) : "vcc");
    uint32_t r1, m;
    asm("v_add_co_u32 %[r1], vcc, %[t1], %[ah0]\n\t"
        "v_cndmask_b32_e64 %[m], 0, -1, vcc"
        : [r1] "=v"(r1), [m] "=v"(m)   // no early clobber: r1 may take over t's dying high register (no v_mov to re-form the pair)
        : [t1] "v"((uint32_t)(t >> 32)), [ah0] "v"((uint32_t)ah)
        : "vcc");
    uint64_t r;
    asm("v_mad_u64_u32 %[r], vcc, %[m], 1, %[base]" : [r] "=&v"(r) : [m] "v"(m), [base] "v"(((uint64_t)r1 << 32) | (uint32_t)t) : "vcc");
    return from_u64(r);
}

// Linear layer; `rc_next` (12 canonical constants, wave-uniform, or nullptr) is the NEXT round's
// constant vector: it seeds the accumulators, so the constant addition costs no vector instruction
// (the 64-bit addend operand of the first multiply-add comes straight from scalar registers).
__device__ __forceinline__ void mds_layer(F (&s)[12], const uint64_t* __restrict__ rc_next = nullptr) {
    constexpr uint32_t C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    F o[12];
#pragma unroll
    for (int r = 0; r < 12; r++) {
        uint64_t al = 0, ah = 0;
        if (rc_next) {
            const uint64_t c = rc_next[r];
            al = (uint32_t)c;
            ah = c >> 32;
        }
#pragma unroll
        for (int i = 0; i < 12; i++) {
            al += (uint64_t)s[(i + r) % 12].lo * C[i];
            ah += (uint64_t)s[(i + r) % 12].hi * C[i];
        }
        if (r == 0) {
            al += (uint64_t)s[0].lo * 8u;
            ah += (uint64_t)s[0].hi * 8u;
        }
        o[r] = fold_acc(al, ah);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = o[i];
}


// ---- the linear layer on the MATRIX cores ----------------------------------------------------------------------------
// out[r] = sum_j M[r][j] s_j with M[r][j] = C[(j - r) mod 12] (+ 8 at [0][0]) and 64-bit s_j = sum_b 2^(8b) byte_b(s_j): eight
// products of the constant 12 x 12 matrix with the state's byte planes - ONE v_mfma_i32_32x32x32_i8 each for all 64 states of
// a wave (one state per lane, as everywhere in the hashing kernels).  No lane ever moves data: column n of B is supplied by
// lane n (k < 16) and lane n + 32 (k >= 16), each its OWN state's twelve bytes of the plane, and A is placed so that output r
// of the state in lane n + 32 h lands in row (r & 3) + 8 (r >> 2) + 4 h - which the 32 x 32 accumulator layout (col =
// lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)) hands back to that same lane as register r.
//
// CONTRACT: the constant operand A is spread over ALL 64 lanes of the wave (and must be zero in the lanes that hold no row),
// and the matrix cores read every lane's registers whatever EXEC says: every lane of a wave must reach the permutation, with
// EXEC all ones.  Callers clamp spare lanes to the last item and store nothing for them; no caller may return early.
//
// The instruction's bytes are signed: the planes are biased by 128 (xor 0x80), and the correction 128 * rowsum(r) is added BY
// THE MATRIX CORES (round 4): a state uses twelve of its sixteen K slots, three of the spare ones hold the constant -128 in B
// and -128, -128, (r == 0 ? -8 : 0) in A: 2 * 16384 + 1024 [r == 0] = 128 * 256 (+ 128 * 8).  Every D_b[r] is therefore the
// exact unsigned sum  sum_j M[r][j] byte_b(s_j)  in [0, 67 320] and the recombination needs no sign handling:
//   x = D_0 + 2^8 D_1, y = D_2 + 2^8 D_3 per 32-bit half (v_lshl_add_u32, full rate), then
//   value = xL + 2^16 yL + 2^32 xH + 2^48 yH:  T = yL * 2^16 + (xH : xL) is ONE v_mad_u64_u32 on the register PAIR, and
//   fold_pair adds 2^48 yH mod p (2^64 = EPS: five multiply / carry class instructions).
// An output that also takes a round constant keeps the two-accumulator form (seeded from scalar registers) and fold_acc.
// Per output 10 vector instructions of which 6 are multiply / carry class (12 and 8 with a constant); round 3 had 14 (8 + 2 + 4),
// all of them multiply / carry class, plus 2 moves.  Per layer: 24 xor + 48 v_perm + 8 MFMA + ~130.
typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef int i32x16_t __attribute__((ext_vector_type(16)));

// Rows 12 .. 15 of a result are never read.  Left at that, hipcc treats those four registers of the sixteen-register result as
// dead from the start: it has overlapped them with the SAME instruction's B operand and handed them to other values while the
// instruction was in flight (seen in k_quotient, where PoseidonGate's three parts share a function: wrong sums that came and
// went with the register allocation).  An empty asm statement that takes the whole result keeps all sixteen registers allocated
// to it up to that point, at no instruction.  It stands right after the last matrix instruction of a group is issued: that covers
// the instruction's own operands and everything scheduled before it; what follows is compiler-generated code reading the
// results (it waits for them), and any write into a register the matrix cores have yet to write is spaced by the hazard
// recogniser as long as it is not inline asm - the first asm statement comes after every result of the group was read.
__device__ __forceinline__ void keep_whole(const i32x16_t& d) { asm volatile("" : : "v"(d)); }

// this lane's A operand: A[row = lane & 31][k = 16 (lane >> 5) + j], j = 0 .. 15 (the same k order as the B operand below,
// whatever the hardware's order inside a lane's sixteen bytes is); slots 12 .. 14 carry the bias correction (see above)
__device__ __forceinline__ i32x4_t mds_a_fragment(uint32_t lane = threadIdx.x & 63) {
    constexpr int C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    const uint32_t rho = lane & 31, h = lane >> 5;
    const uint32_t g = (rho >> 2) & 1, r = (rho & 3) + 4 * (rho >> 3);
    uint32_t w[4] = {0, 0, 0, 0};
    if (h == g && rho < 24) {
#pragma unroll
        for (int j = 0; j < 12; j++) {
            uint32_t m = 0;
#pragma unroll
            for (int rr = 0; rr < 12; rr++)
                if ((uint32_t)rr == r) m = (uint32_t)C[(j - rr + 12) % 12] + ((rr == 0 && j == 0) ? 8u : 0u);
            w[j >> 2] |= m << (8 * (j & 3));
        }
        w[3] = r == 0 ? 0x00F88080u : 0x00008080u;
    }
    i32x4_t a;
    a.x = (int)w[0]; a.y = (int)w[1]; a.z = (int)w[2]; a.w = (int)w[3];
    return a;
}
constexpr uint32_t MDS_B_BIAS_SLOTS = 0x00808080u;   // the B operand's fourth dword: -128 in K slots 12 .. 14, slot 15 unused

// 4 x 4 byte transpose: p[b] = (x0.byte b, x1.byte b, x2.byte b, x3.byte b)
__device__ __forceinline__ void transpose_bytes4(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t (&p)[4]) {
    const uint32_t t01l = __builtin_amdgcn_perm(x1, x0, 0x05010400u), t01h = __builtin_amdgcn_perm(x1, x0, 0x07030602u);
    const uint32_t t23l = __builtin_amdgcn_perm(x3, x2, 0x05010400u), t23h = __builtin_amdgcn_perm(x3, x2, 0x07030602u);
    p[0] = __builtin_amdgcn_perm(t23l, t01l, 0x05040100u);
    p[1] = __builtin_amdgcn_perm(t23l, t01l, 0x07060302u);
    p[2] = __builtin_amdgcn_perm(t23h, t01h, 0x05040100u);
    p[3] = __builtin_amdgcn_perm(t23h, t01h, 0x07060302u);
}
// a * b + c as ONE v_mad_u64_u32 (the compiler would expand a power-of-two b into shifts and carry chains); the carry-out
// goes to a scalar pair nobody reads, so VCC stays free for the neighbours
__device__ __forceinline__ uint64_t mad_u64(uint32_t a, uint32_t b, uint64_t c) {
    uint64_t r, sink;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(sink) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint64_t mad_u64_sc(uint32_t a, uint32_t b, uint64_t c_uniform) {   // c from scalar registers
    uint64_t r, sink;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(sink) : "v"(a), "v"(b), "s"(c_uniform));
    return r;
}
__device__ __forceinline__ uint64_t mad_u64_one(uint32_t a, uint64_t c) {   // a + c
    uint64_t r, sink;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %3" : "=v"(r), "=s"(sink) : "v"(a), "v"(c));
    return r;
}
// t + 2^48 yh (mod p) -> loose, for t < 2^59 and yh < 2^26.  z = yh * 2^16 = z0 + 2^32 z1 (one multiply, z1 < 2^10) and
// 2^48 yh = 2^32 z0 + 2^64 z1; 2^64 = EPS, so z1 folds in by a multiply-add that cannot carry and z0 lands on the high word,
// the only addition that can carry - once, worth EPS, and adding it cannot carry again.
__device__ __forceinline__ F fold_pair(uint64_t t, uint32_t yh, uint32_t k16) {
    uint64_t z, u, sink, sink2;
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(z), "=s"(sink) : "v"(yh), "v"(k16));
    asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(u), "=s"(sink2) : "v"((uint32_t)(z >> 32)), "v"(t));
    uint32_t r1, m;
    asm("v_add_co_u32 %0, vcc, %2, %3\n\t"
        "v_cndmask_b32_e64 %1, 0, -1, vcc"
        : "=v"(r1), "=v"(m)
        : "v"((uint32_t)(u >> 32)), "v"((uint32_t)z)
        : "vcc");
    return from_u64(mad_u64_one(m, ((uint64_t)r1 << 32) | (uint32_t)u));
}

// RC: 0 = no round constant follows this layer, 1 = a constant for output 0 only (the partial rounds, poseidon.hpp), 2 = twelve.
// rcb: 24 wave-uniform words: [r] = low half of the NEXT round's constant r, [12 + r] = its high half (zero-extended)
template <int RC>
__device__ __forceinline__ void mds_layer_mfma(F (&s)[12], const i32x4_t a, const uint64_t* __restrict__ rcb) {
    uint32_t k16 = 65536u, bw = MDS_B_BIAS_SLOTS;
    asm("" : "+v"(k16));   // a register the compiler cannot see through: yL * k16 + pair stays ONE v_mad_u64_u32
    asm("" : "+v"(bw));    // likewise: a B operand assembled around a literal costs five moves per MFMA instead of one
    i32x16_t zero;
#pragma unroll
    for (int i = 0; i < 16; i++) zero[i] = 0;
    uint32_t x[2][12], y[2][12];
#pragma unroll
    for (int half = 0; half < 2; half++) {   // planes 0 .. 3 come from the low words, 4 .. 7 from the high words
        uint32_t pl[4][3];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            uint32_t p[4];
            if (half == 0)
                transpose_bytes4(s[4 * q].lo ^ 0x80808080u, s[4 * q + 1].lo ^ 0x80808080u, s[4 * q + 2].lo ^ 0x80808080u,
                                 s[4 * q + 3].lo ^ 0x80808080u, p);
            else
                transpose_bytes4(s[4 * q].hi ^ 0x80808080u, s[4 * q + 1].hi ^ 0x80808080u, s[4 * q + 2].hi ^ 0x80808080u,
                                 s[4 * q + 3].hi ^ 0x80808080u, p);
#pragma unroll
            for (int b = 0; b < 4; b++) pl[b][q] = p[b];
        }
        i32x16_t d[4];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            i32x4_t bf;
            bf.x = (int)pl[b][0]; bf.y = (int)pl[b][1]; bf.z = (int)pl[b][2]; bf.w = (int)bw;
            d[b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bf, zero, 0, 0, 0);
        }
#pragma unroll
        for (int b = 0; b < 4; b++) keep_whole(d[b]);   // after the LAST instruction is issued, before the first result is read
#pragma unroll
        for (int r = 0; r < 12; r++) {
            x[half][r] = ((uint32_t)d[1][r] << 8) + (uint32_t)d[0][r];
            y[half][r] = ((uint32_t)d[3][r] << 8) + (uint32_t)d[2][r];
        }
        __builtin_amdgcn_sched_barrier(0);   // the low half's sixteen-register results are dead before the high half's exist
    }
#pragma unroll
    for (int r = 0; r < 12; r++) {
        if (RC == 2 || (RC == 1 && r == 0)) {
            const uint64_t al = mad_u64_one(x[0][r], mad_u64_sc(y[0][r], k16, rcb[r]));
            const uint64_t ah = mad_u64_one(x[1][r], mad_u64_sc(y[1][r], k16, rcb[12 + r]));
            s[r] = fold_acc(al, ah);
        } else {
            const uint64_t t = mad_u64(y[0][r], k16, ((uint64_t)x[1][r] << 32) | x[0][r]);
            s[r] = fold_pair(t, y[1][r], k16);
        }
    }
}

// ---- Fused partial rounds (round 4; tables and derivation: tools/gen_poseidon_blocks.py -> poseidon_blocks.inc) ----------------
// Three partial rounds are ONE pass of the state through the byte planes: the split (24 xor + 48 v_perm) and the recombination
// happen once per block instead of once per round.  The block's matrix [M Q^2 | M Q e0 | M e0] has 22-bit entries: three balanced
// base-256 digit matrices A_0 .. A_2, and digit p times byte plane b accumulates into output plane b + p INSIDE the matrix cores
// (the C operand chains the instructions of one output plane), ten output planes.  The two S-box inputs inside the block are
// twelve-term dot products with the small rows m0 and m0 Q on the vector pipe (24 multiply-adds each; as rows 12 / 13 of a first
// matrix pass they cost sixteen more matrix instructions and nine waits for a chain's result: measured slower, 2.50 against
// 2.6 G/s).  Every chain starts from the seed vector, which keeps
// the plane sums non-negative whatever the (signed) digits and the biased bytes are; what the seeds add in total is a constant
// per row that the generated constants absorb.  Plane sums stay below NLX_POSEIDON_BLOCK_DMAX (< 2^20), so x = d + 2^8 d' fits
// 32 bits and the folds below cannot carry where they assume so (checked by the generator, `bounds`).
struct BlockOperands {
    i32x4_t a[3];    // this lane's A fragments of the three digit matrices
    i32x16_t seed;   // chain seeds, one per output row
};

template <int G>
__device__ __forceinline__ uint64_t mad_u64_imm(uint32_t a, uint64_t c) {   // a * G + c, G an inline constant
    uint64_t r, sink;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(sink) : "v"(a), "n"(G), "v"(c));
    return r;
}
// t + 2^48 yh + 2^64 x2 (mod p) -> loose: fold_pair with the planes above 2^64 (x2 < 2^29) added to the word that EPS multiplies
__device__ __forceinline__ F fold_pair_x(uint64_t t, uint32_t yh, uint32_t x2, uint32_t k16) {
    uint64_t z, u, sink, sink2;
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(z), "=s"(sink) : "v"(yh), "v"(k16));
    const uint32_t e = (uint32_t)(z >> 32) + x2;
    asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(u), "=s"(sink2) : "v"(e), "v"(t));
    uint32_t r1, m;
    asm("v_add_co_u32 %0, vcc, %2, %3\n\t"
        "v_cndmask_b32_e64 %1, 0, -1, vcc"
        : "=v"(r1), "=v"(m)
        : "v"((uint32_t)(u >> 32)), "v"((uint32_t)z)
        : "vcc");
    return from_u64(mad_u64_one(m, ((uint64_t)r1 << 32) | (uint32_t)u));
}
__device__ __forceinline__ uint64_t mad_u64_sm(uint32_t a, uint32_t m_uniform, uint64_t c) {   // a * m + c, m from a scalar register
    uint64_t r, sink;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(sink) : "v"(a), "s"(m_uniform), "v"(c));
    return r;
}
// sum_j rho[j] z_j + (klo + 2^32 khi) as the two accumulators fold_acc takes (value = al + 2^32 ah; rho[j] < 2^15: no overflow)
template <typename RHO>
__device__ __forceinline__ void dot12(const F (&z)[12], const RHO& rho, uint64_t klo, uint64_t khi, uint64_t& al, uint64_t& ah) {
    al = klo;
    ah = khi;
#pragma unroll
    for (int j = 0; j < 12; j++) {
        al = mad_u64_sm(z[j].lo, rho.v[j], al);
        ah = mad_u64_sm(z[j].hi, rho.v[j], ah);
    }
}
struct BlockRho1 { static constexpr uint32_t v[12] = NLX_POSEIDON_BLOCK_RHO1_INIT; };
struct BlockRho2 { static constexpr uint32_t v[12] = NLX_POSEIDON_BLOCK_RHO2_INIT; };

// What happens at the three S-boxes of a block.  The permutation applies them; PoseidonGate (prover_kernels.hip) is handed the
// computed S-box input, emits `input - wire` and continues from the wire's S-box - the same linear maps either way.
struct BlockSboxes {
    __device__ __forceinline__ F first(F s0) const { return sbox7(s0); }            // round a: element 0 of the state
    __device__ __forceinline__ F inner(int, F w) const { return sbox7(w); }         // rounds a + 1, a + 2: w_1, w_2
    __device__ __forceinline__ void before_matrix_pass() const {}
};

// kap: this block's six wave-uniform words (low, high) x (output 0's constant, S-box input 1's, S-box input 2's); GAMMA21 = M[0][0]
template <int GAMMA21, class Sboxes = BlockSboxes>
__device__ __forceinline__ void partial_block3(F (&s)[12], const BlockOperands& op, const uint64_t* __restrict__ kap,
                                               Sboxes sb = Sboxes{}) {
    uint32_t k16 = 65536u;
    asm("" : "+v"(k16));
    s[0] = sb.first(s[0]);
    // the S-box inputs of the block's second and third round, and their outputs
    uint64_t al, ah;
    dot12(s, BlockRho1{}, kap[2], kap[3], al, ah);
    const F u1 = sb.inner(1, fold_acc(al, ah));
    dot12(s, BlockRho2{}, kap[4], kap[5], al, ah);
    al = mad_u64_imm<GAMMA21>(u1.lo, al);
    ah = mad_u64_imm<GAMMA21>(u1.hi, ah);
    const F u2 = sb.inner(2, fold_acc(al, ah));
    sb.before_matrix_pass();
    uint32_t pl[8][3];
#pragma unroll
    for (int half = 0; half < 2; half++)
#pragma unroll
        for (int q = 0; q < 3; q++) {
            uint32_t p[4];
            if (half == 0)
                transpose_bytes4(s[4 * q].lo ^ 0x80808080u, s[4 * q + 1].lo ^ 0x80808080u, s[4 * q + 2].lo ^ 0x80808080u,
                                 s[4 * q + 3].lo ^ 0x80808080u, p);
            else
                transpose_bytes4(s[4 * q].hi ^ 0x80808080u, s[4 * q + 1].hi ^ 0x80808080u, s[4 * q + 2].hi ^ 0x80808080u,
                                 s[4 * q + 3].hi ^ 0x80808080u, p);
#pragma unroll
            for (int b = 0; b < 4; b++) pl[4 * half + b][q] = p[b];
        }
    // main pass: slots 12 / 13 carry u1 / u2
    uint32_t w3[8];
    {
        const uint32_t a1l = u1.lo ^ 0x80808080u, a1h = u1.hi ^ 0x80808080u, a2l = u2.lo ^ 0x80808080u, a2h = u2.hi ^ 0x80808080u;
        w3[0] = __builtin_amdgcn_perm(a2l, a1l, 0x00000400u); w3[1] = __builtin_amdgcn_perm(a2l, a1l, 0x00000501u);
        w3[2] = __builtin_amdgcn_perm(a2l, a1l, 0x00000602u); w3[3] = __builtin_amdgcn_perm(a2l, a1l, 0x00000703u);
        w3[4] = __builtin_amdgcn_perm(a2h, a1h, 0x00000400u); w3[5] = __builtin_amdgcn_perm(a2h, a1h, 0x00000501u);
        w3[6] = __builtin_amdgcn_perm(a2h, a1h, 0x00000602u); w3[7] = __builtin_amdgcn_perm(a2h, a1h, 0x00000703u);
    }
    // a pair of output planes at a time: x = d + 2^8 d' (x0, y0, x1, y1, x2); what can be folded is folded as soon as its
    // planes exist, so at most three twelve-register terms are alive beside the pair in flight
    uint32_t x0[12], y0[12], y1[12];
    uint64_t tt[12];   // [0]: the low accumulator of the seeded form; [r >= 1]: t = x0 + 2^16 y0 + 2^32 x1
    uint64_t ah0 = 0;
    auto chain = [&](int q) {
        i32x16_t d = op.seed;
#pragma unroll
        for (int p = 0; p < 3; p++)
            if (q - p >= 0 && q - p <= 7) {
                i32x4_t bf;
                bf.x = (int)pl[q - p][0]; bf.y = (int)pl[q - p][1]; bf.z = (int)pl[q - p][2]; bf.w = (int)w3[q - p];
                d = __builtin_amdgcn_mfma_i32_32x32x32_i8(op.a[p], bf, d, 0, 0, 0);
            }
        return d;
    };
#pragma unroll
    for (int pr = 0; pr < 5; pr++) {
        i32x16_t d[2];
        d[0] = chain(2 * pr);
        d[1] = chain(2 * pr + 1);
        keep_whole(d[0]);
        keep_whole(d[1]);
#pragma unroll
        for (int r = 0; r < 12; r++) {
            // the compiler's own shift-add: the matrix cores' results must NOT go straight into inline asm - the hazard recogniser
            // that spaces a vector read from the matrix instruction that wrote the register does not look inside asm statements
            // (seen: wrong sums).  The empty asm only stops the re-association of x2 + z1 into v_lshlrev + v_add3.
            uint32_t x = ((uint32_t)d[1][r] << 8) + (uint32_t)d[0][r];
            asm("" : "+v"(x));
            if (pr == 0) x0[r] = x;
            if (pr == 1) {
                y0[r] = x;
                if (r == 0) tt[0] = mad_u64_one(x0[0], mad_u64_sc(x, k16, kap[0]));
            }
            if (pr == 2) {
                if (r == 0) ah0 = x;
                else tt[r] = mad_u64(y0[r], k16, ((uint64_t)x << 32) | x0[r]);
            }
            if (pr == 3) {
                y1[r] = x;
                if (r == 0) ah0 = mad_u64_one((uint32_t)ah0, mad_u64_sc(x, k16, kap[1]));
            }
            if (pr == 4) {
                if (r == 0) s[0] = fold_acc(tt[0], ah0 + ((uint64_t)x << 32));
                else s[r] = fold_pair_x(tt[r], y1[r], x, k16);
            }
        }
#ifndef NLX_PB_NO_BARRIER
        __builtin_amdgcn_sched_barrier(0);   // a pair of planes is consumed before the next pair's sixteen-register results exist
#endif
    }
}

}  // namespace gl32
