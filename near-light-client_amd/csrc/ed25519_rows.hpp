// Witness generation for the Ed25519 verification AIR (near-light-client_amd/ed25519_air.py; SURVEY.md §8a row a12,
// curta_eddsa_verify_sigs_conditional at nearx/src/builder.rs:152).  One trace row = one bit of both scalars: a point
// doubling and the addition of 0, B, -A or B - A (Shamir's trick), as 15 multiplication units mod 2^255 - 19
// (fp25519.hpp) plus the auxiliary unit's step of the per-slot program.  Plain C++ / device code: the same function
// runs in the sequential per-slot scan (no output), in the row-parallel emitter, and on the host in the tests.
#pragma once
#include "fe25519_fast.hpp"
#include "fp25519.hpp"

namespace nlx {
namespace ed {

constexpr int ROWS = 256, N_MAIN = 15, UNIT_CELLS = fp::UNIT_CELLS;
// round-0 column map (ed25519_air.py)
constexpr uint32_t cSIN = 0, cSB = 48, cHB = 49, cSA = 50, cHA = 51, cAX = 52, cAY = 68, cRX = 84, cRY = 100, cNT = 116,
                   cSW = 132, cHW = 148, cCHK = 164, cSAA = 168, cSBB = 184, cSCC = 200, cSX3 = 216, cSY3 = 232, cSZ3 = 248,
                   cSPT = 264, cP2 = 280, cMAIN = 344, cAUX_A = cMAIN + N_MAIN * UNIT_CELLS, cAUX_B = cAUX_A + 16,
                   cAUX_E = cAUX_B + 16, cAUX_F = cAUX_E + 16, cAUX = cAUX_F + 16, cACT = cAUX + UNIT_CELLS, cDW = cACT + 1,
                   cQW = cDW + 32, cCHKQ = cQW + 17, cCLO = cCHKQ + 2, cDH = cCLO + 1, cDS = cDH + 1, cCHI = cDS + 1,
                   cBH = cCHI + 1, cBS = cBH + 1, cSGA = cBS + 1, cSGR = cSGA + 1, cCXY = cSGR + 1, cKXA = cCXY + 4, cKXR = cKXA + 1,
                   cBXY = cKXR + 1, cMULT9 = cBXY + 4,
                   cMULT = cMULT9 + 1, N_COLS0 = cMULT + 1;  // further multiplicity columns of a spread table follow (caller's)
constexpr int SLOT_WORDS = 32, MODL_ROWS = 32;
enum { U_A, U_B, U_ZZ, U_E, U_X2, U_Y2, U_T2, U_Z2, U_PA, U_PB, U_PC, U_PD, U_X4, U_Y4, U_Z4 };
enum { STEP_YCMP = 0, STEP_A_U = 1, STEP_A_NT = 2, STEP_A_U2 = 3, STEP_A_V = 4, STEP_A_CHK = 5, STEP_R_U = 6, STEP_R_V = 7,
       STEP_R_CHK = 8, STEP_S_AA = 9, STEP_S_BB = 10, STEP_S_CC = 11, STEP_S_X = 12, STEP_S_Y = 13, STEP_S_Z = 14, STEP_S_T = 15,
       STEP_S_PT = 16, STEP_XCMP = 255 };

typedef int32_t limbs_t[16];

// curve constants as 16-bit limbs, little-endian: d, 2d, the base point's triple (y - x, y + x, 2 d x y), p - 1, x_B y_B
static constexpr uint16_t K_LIMBS[7][16] = {
    {0x78a3, 0x1359, 0x4dca, 0x75eb, 0xd8ab, 0x4141, 0x0a4d, 0x0070, 0xe898, 0x7779, 0x4079, 0x8cc7, 0xfe73, 0x2b6f, 0x6cee, 0x5203},
    {0xf159, 0x26b2, 0x9b94, 0xebd6, 0xb156, 0x8283, 0x149a, 0x00e0, 0xd130, 0xeef3, 0x80f2, 0x198e, 0xfce7, 0x56df, 0xd9dc, 0x2406},
    {0x913e, 0xd740, 0x3905, 0x9d10, 0xbeb3, 0xd140, 0x9f05, 0xfd39, 0x8a09, 0x688f, 0x8434, 0xa5c1, 0x1267, 0x98f8, 0x2f92, 0x44fd},
    {0x3b85, 0xf58c, 0x93c6, 0x2fbc, 0x0e19, 0xfb8c, 0x2dc6, 0xcf93, 0x42c2, 0x643d, 0x4898, 0x270b, 0xba65, 0x33d4, 0x9d3a, 0x07cf},
    {0xaa68, 0x877a, 0x1205, 0xabc9, 0xc49e, 0xccaa, 0xe823, 0x26d9, 0x598c, 0xdd43, 0x7dcb, 0x5a1b, 0x65a8, 0x9f0c, 0x7b68, 0x6f11},
    {0xffec, 0xffff, 0xffff, 0xffff, 0xffff, 0xffff, 0xffff, 0xffff, 0xffff, 0xffff, 0xffff, 0xffff, 0xffff, 0xffff, 0xffff, 0x7fff},
    {0xdda3, 0xa5b7, 0x8ab3, 0x6dde, 0x52f5, 0x7751, 0x9f80, 0x20f0, 0xe37d, 0x64ab, 0x4e8e, 0x66ea, 0x7665, 0xd78b, 0x5f0f, 0x6787}};
FP_HD inline void const_limbs(int which, limbs_t out) {
    for (int i = 0; i < 16; i++) out[i] = K_LIMBS[which][i];
}
enum { K_D = 0, K_D2 = 1, K_B_YMX = 2, K_B_YPX = 3, K_B_T2D = 4, K_M1 = 5, K_B_T = 6 };

// one product unit: c = a b (canonical), quotient, carries.  Not inlined on the device: a row calls it 16 times, and one
// copy with fully unrolled loops (every array in registers) is what keeps the generator out of scratch memory.
FP_HD FP_NOINLINE inline void mul1(const limbs_t a, const limbs_t b, fp::Unit& u, const uint32_t* c_fixed = nullptr) {
    int64_t prod[32];
    FP_UNROLL
    for (int k = 0; k < 32; k++) prod[k] = 0;
    fp::mul_acc(prod, a, b);
    fp::finish(prod, u, c_fixed);
}
FP_HD inline void vsub(const uint32_t* a, const uint32_t* b, limbs_t o) { for (int i = 0; i < 16; i++) o[i] = (int32_t)a[i] - (int32_t)b[i]; }
FP_HD inline void vadd(const uint32_t* a, const uint32_t* b, limbs_t o) { for (int i = 0; i < 16; i++) o[i] = (int32_t)a[i] + (int32_t)b[i]; }
FP_HD inline void vcopy(const uint32_t* a, limbs_t o) { for (int i = 0; i < 16; i++) o[i] = (int32_t)a[i]; }

struct Point { uint32_t x[16], y[16], z[16]; };  // reduced limbs

// The slot's constant data: limbs of A, R, the scalars, and what the auxiliary program derives from A: 2dxy and the
// point B - A with the three products of its addition (all canonical).
struct Slot {
    uint32_t ax[16], ay[16], rx[16], ry[16], nt[16], sw[16], hw[16];
    uint32_t saa[16], sbb[16], scc[16], sx3[16], sy3[16], sz3[16], st3[16], spt[16];
    // the reduction h = D mod L and the two comparisons (ed25519_air.py modl_witness)
    uint32_t dw[32], qw[17], carry[MODL_ROWS + 1], dh[16], ds[16];
    uint32_t bh, bs;      // bit j: the carry into limb j of h + (L - 1 - h) / S + (L - 1 - S)
    uint32_t active;      // 1: the slot's three checks are on
    uint32_t s_in_range;  // 0: S >= L or a coordinate >= p (no witness for the comparisons)
    uint32_t cxy[4][16];  // p - 1 - v for v = A.x, A.y, R.x, R.y
    uint32_t bxy[4];      // bit j: the carry into limb j of v + (p - 1 - v)
};

// the group order L = 2^252 + 27742317777372353535851937790883648493, 16-bit limbs
static constexpr uint16_t L_LIMBS[16] = {0xd3ed, 0x5cf5, 0x631a, 0x5812, 0x9cd6, 0xa2f7, 0xf9de, 0x14de, 0, 0, 0, 0, 0, 0, 0, 0x1000};

// h = D mod L, q = D div L and the carries of q L + h = D position by position; L - 1 - h, L - 1 - S with their carry bits.
// Long division by 16-bit digits: L = 2^252 + c with c < 2^125, so floor(R / 2^252) is the quotient digit or one more.
FP_HD inline void modl_witness(Slot& s) {
    uint32_t rem[18];
    for (int i = 0; i < 18; i++) rem[i] = 0;
    for (int i = 0; i < 17; i++) s.qw[i] = 0;
    for (int k = 31; k >= 0; k--) {
        for (int i = 17; i > 0; i--) rem[i] = rem[i - 1];   // rem = rem * 2^16 + D_k  (rem < L, so 17 limbs + the new one)
        rem[0] = s.dw[k];
        uint32_t digit = ((rem[16] << 16) | rem[15]) >> 12;   // floor(rem / 2^252); rem < 2^269
        // rem -= digit * L
        int64_t borrow = 0;
        for (int i = 0; i < 18; i++) {
            const int64_t v = (int64_t)rem[i] - (i < 16 ? (int64_t)digit * L_LIMBS[i] : 0) + borrow;
            rem[i] = (uint32_t)(v & 0xFFFF);
            borrow = v >> 16;   // arithmetic shift: floor
        }
        if (borrow < 0) {       // one too many: add L back
            digit--;
            uint32_t c = 0;
            for (int i = 0; i < 18; i++) {
                const uint32_t v = rem[i] + (i < 16 ? L_LIMBS[i] : 0u) + c;
                rem[i] = v & 0xFFFF;
                c = v >> 16;
            }
        }
        if (k < 17) s.qw[k] = digit;   // D < 2^512 and L > 2^252: the digits above position 16 are zero
    }
    for (int i = 0; i < 16; i++) s.hw[i] = rem[i];
    uint64_t c = 0;
    s.carry[0] = 0;
    for (int k = 0; k < MODL_ROWS; k++) {
        uint64_t tot = c + (k < 16 ? s.hw[k] : 0u);
        for (int i = 0; i < 17; i++)
            if (k - i >= 0 && k - i < 16) tot += (uint64_t)s.qw[i] * L_LIMBS[k - i];
        c = (tot - s.dw[k]) >> 16;   // tot = D_k (mod 2^16) and tot >= D_k
        s.carry[k + 1] = (uint32_t)c;
    }
    // complements: x + (L - 1 - x) = L - 1 limb by limb
    s.s_in_range = 1;
    for (int which = 0; which < 2; which++) {
        const uint32_t* x = which ? s.sw : s.hw;
        uint32_t* comp = which ? s.ds : s.dh;
        int32_t borrow = 0;
        for (int i = 0; i < 16; i++) {   // comp = L - 1 - x
            const int32_t v = (int32_t)L_LIMBS[i] - (i == 0 ? 1 : 0) - (int32_t)x[i] + borrow;
            comp[i] = (uint32_t)(v & 0xFFFF);
            borrow = v >> 16;
        }
        if (borrow < 0 && which) s.s_in_range = 0;
        uint32_t bits = 0, cb = 0;
        for (int i = 0; i < 16; i++) {
            const uint32_t lm1 = (uint32_t)L_LIMBS[i] - (i == 0 ? 1u : 0u);
            cb = (x[i] + comp[i] + cb - lm1) >> 16 & 1u;
            if (i < 15) bits |= cb << (i + 1);
        }
        (which ? s.bs : s.bh) = bits;
    }
    // canonical coordinates: v + (p - 1 - v) = p - 1, p - 1 = 2^255 - 20
    for (int v = 0; v < 4; v++) {
        const uint32_t* x = v == 0 ? s.ax : (v == 1 ? s.ay : (v == 2 ? s.rx : s.ry));
        int32_t borrow = 0;
        for (int i = 0; i < 16; i++) {
            const int32_t t = (int32_t)K_LIMBS[K_M1][i] - (int32_t)x[i] + borrow;
            s.cxy[v][i] = (uint32_t)(t & 0xFFFF);
            borrow = t >> 16;
        }
        if (borrow < 0) s.s_in_range = 0;
        uint32_t bits = 0, cb = 0;
        for (int i = 0; i < 16; i++) {
            cb = (x[i] + s.cxy[v][i] + cb - (uint32_t)K_LIMBS[K_M1][i]) >> 16 & 1u;
            if (i < 15) bits |= cb << (i + 1);
        }
        s.bxy[v] = bits;
    }
}

// the linear combinations the B - A addition multiplies: E = BB - AA, H = BB + AA, F = CC' + 2, G = 2 - CC'
FP_HD inline void sum_operands(const Slot& s, limbs_t e, limbs_t f, limbs_t g, limbs_t h) {
    for (int i = 0; i < 16; i++) {
        const int32_t two = i == 0 ? 2 : 0;
        e[i] = (int32_t)s.sbb[i] - (int32_t)s.saa[i];
        h[i] = (int32_t)s.sbb[i] + (int32_t)s.saa[i];
        f[i] = (int32_t)s.scc[i] + two;
        g[i] = two - (int32_t)s.scc[i];
    }
}

FP_HD inline void slot_from_words(const uint64_t* w /* ax ay rx ry s: 5 x 4 words, D: 8 words, active, 3 spare */, Slot& s) {
    uint32_t* dst[5] = {s.ax, s.ay, s.rx, s.ry, s.sw};
    for (int v = 0; v < 5; v++)
        for (int i = 0; i < 16; i++) dst[v][i] = (uint32_t)((w[4 * v + (i >> 2)] >> (16 * (i & 3))) & 0xFFFF);
    for (int i = 0; i < 32; i++) s.dw[i] = (uint32_t)((w[20 + (i >> 2)] >> (16 * (i & 3))) & 0xFFFF);
    s.active = (uint32_t)(w[28] & 1);
    modl_witness(s);
    limbs_t a, b, k;
    fp::Unit u;
    auto store = [&](uint32_t* out) { for (int i = 0; i < 16; i++) out[i] = u.c[i]; };
    vcopy(s.ax, a);
    vcopy(s.ay, b);
    mul1(a, b, u);
    const_limbs(K_D2, k);
    vcopy(u.c, a);
    mul1(k, a, u);
    store(s.nt);
    vadd(s.ay, s.ax, a);
    const_limbs(K_B_YMX, k);
    mul1(a, k, u);
    store(s.saa);
    vsub(s.ay, s.ax, a);
    const_limbs(K_B_YPX, k);
    mul1(a, k, u);
    store(s.sbb);
    vcopy(s.nt, a);
    const_limbs(K_B_T, k);
    mul1(a, k, u);
    store(s.scc);
    limbs_t e, f, g, h;
    sum_operands(s, e, f, g, h);
    mul1(e, f, u);
    store(s.sx3);
    mul1(g, h, u);
    store(s.sy3);
    mul1(f, g, u);
    store(s.sz3);
    mul1(e, h, u);
    store(s.st3);
    const_limbs(K_D2, k);
    vcopy(s.st3, a);
    mul1(k, a, u);
    store(s.spt);
}

// the row's addend (y - x, y + x, 2dt, 2z): neutral, B, -A or B - A by the two bits (signed limbs)
FP_HD inline void addend(const Slot& s, int sbit, int hbit, limbs_t ymx, limbs_t ypx, limbs_t t2d, limbs_t z2) {
    for (int i = 0; i < 16; i++) {
        const int32_t one = i == 0, two = i == 0 ? 2 : 0;
        if (sbit && hbit) {
            ymx[i] = (int32_t)s.sy3[i] - (int32_t)s.sx3[i];
            ypx[i] = (int32_t)s.sy3[i] + (int32_t)s.sx3[i];
            t2d[i] = (int32_t)s.spt[i];
            z2[i] = 2 * (int32_t)s.sz3[i];
        } else if (hbit) {  // -A = (-x, y)
            ymx[i] = (int32_t)s.ay[i] + (int32_t)s.ax[i];
            ypx[i] = (int32_t)s.ay[i] - (int32_t)s.ax[i];
            t2d[i] = -(int32_t)s.nt[i];
            z2[i] = two;
        } else if (sbit) {
            ymx[i] = K_LIMBS[K_B_YMX][i];
            ypx[i] = K_LIMBS[K_B_YPX][i];
            t2d[i] = K_LIMBS[K_B_T2D][i];
            z2[i] = two;
        } else {
            ymx[i] = ypx[i] = one;
            t2d[i] = 0;
            z2[i] = two;
        }
    }
}

// One row: in -> 2 in + (0 | B | -A | B - A).  Sink::unit(index, Unit) receives the fifteen units in column order.
template <class Sink>
FP_HD inline void row_main(Sink& sink, const Point& in, int sbit, int hbit, const Slot& s, Point& out) {
    fp::Unit a_, b_, zz, e1, u;
    limbs_t l0;
    vcopy(in.x, l0);
    mul1(l0, l0, a_);
    sink.unit(U_A, a_);
    vcopy(in.y, l0);
    mul1(l0, l0, b_);
    sink.unit(U_B, b_);
    vcopy(in.z, l0);
    mul1(l0, l0, zz);
    sink.unit(U_ZZ, zz);
    vadd(in.x, in.y, l0);
    mul1(l0, l0, e1);
    sink.unit(U_E, e1);
    limbs_t e, f, g, h;
    for (int i = 0; i < 16; i++) {
        const int32_t av = (int32_t)a_.c[i], bv = (int32_t)b_.c[i];
        e[i] = (int32_t)e1.c[i] - av - bv;
        g[i] = bv - av;
        f[i] = g[i] - 2 * (int32_t)zz.c[i];
        h[i] = -(av + bv);
    }
    uint32_t x2[16], y2[16], t2[16], z2[16];
    mul1(e, f, u);
    sink.unit(U_X2, u);
    for (int i = 0; i < 16; i++) x2[i] = u.c[i];
    mul1(g, h, u);
    sink.unit(U_Y2, u);
    for (int i = 0; i < 16; i++) y2[i] = u.c[i];
    mul1(e, h, u);
    sink.unit(U_T2, u);
    for (int i = 0; i < 16; i++) t2[i] = u.c[i];
    mul1(f, g, u);
    sink.unit(U_Z2, u);
    for (int i = 0; i < 16; i++) z2[i] = u.c[i];
    limbs_t ymx, ypx, t2d, pz2;
    addend(s, sbit, hbit, ymx, ypx, t2d, pz2);
    fp::Unit pa, pb, pc, pd;
    vsub(y2, x2, l0);
    mul1(l0, ymx, pa);
    sink.unit(U_PA, pa);
    vadd(y2, x2, l0);
    mul1(l0, ypx, pb);
    sink.unit(U_PB, pb);
    vcopy(t2, l0);
    mul1(l0, t2d, pc);
    sink.unit(U_PC, pc);
    vcopy(z2, l0);
    mul1(l0, pz2, pd);
    sink.unit(U_PD, pd);
    for (int i = 0; i < 16; i++) {
        e[i] = (int32_t)pb.c[i] - (int32_t)pa.c[i];
        f[i] = (int32_t)pd.c[i] - (int32_t)pc.c[i];
        g[i] = (int32_t)pd.c[i] + (int32_t)pc.c[i];
        h[i] = (int32_t)pb.c[i] + (int32_t)pa.c[i];
    }
    mul1(e, f, u);
    sink.unit(U_X4, u);
    for (int i = 0; i < 16; i++) out.x[i] = u.c[i];
    mul1(g, h, u);
    sink.unit(U_Y4, u);
    for (int i = 0; i < 16; i++) out.y[i] = u.c[i];
    mul1(f, g, u);
    sink.unit(U_Z4, u);
    for (int i = 0; i < 16; i++) out.z[i] = u.c[i];
}

// The auxiliary unit of row r (0..255) of a slot: operands (a, b, e, f) and the unit a b + e e - f f = c.
// `out` is this row's result point; prev_* are the previous slot's R_y and final point (row 0's comparison).
struct AuxRow {
    limbs_t a, b, e, f;
    fp::Unit u;
};
// prev_active: the previous slot's flag (row 0 holds ITS Y comparison).  The three checks of an inactive slot run with a
// free result (the canonical product) instead of the value the check demands.
FP_HD inline bool row_aux(int r, const Slot& s, const Point& out, const uint32_t* prev_ry, const Point* prev_final, bool prev_active,
                          AuxRow& x) {
    for (int i = 0; i < 16; i++) x.a[i] = x.b[i] = x.e[i] = x.f[i] = 0;
    const uint32_t* fixed = nullptr;
    uint32_t cfix[16];
    fp::Unit t;
    limbs_t l0, l1, se, sf, sg, sh;
    auto xy_of = [&](const uint32_t* px, const uint32_t* py, fp::Unit& o) {
        vcopy(px, l0);
        vcopy(py, l1);
        mul1(l0, l1, o);
    };
    auto set2 = [&](const limbs_t a, const limbs_t b) {
        for (int i = 0; i < 16; i++) { x.a[i] = a[i]; x.b[i] = b[i]; }
    };
    switch (r) {
        case STEP_YCMP:
            vcopy(prev_ry, x.a);
            vcopy(prev_final->z, x.b);
            for (int i = 0; i < 16; i++) cfix[i] = prev_final->y[i];
            if (prev_active) fixed = cfix;
            break;
        case STEP_A_U: case STEP_A_U2: vcopy(s.ax, x.a); vcopy(s.ay, x.b); break;
        case STEP_A_NT: xy_of(s.ax, s.ay, t); const_limbs(K_D2, x.a); vcopy(t.c, x.b); break;
        case STEP_A_V: xy_of(s.ax, s.ay, t); vcopy(t.c, x.a); vcopy(t.c, x.b); break;
        case STEP_R_V: xy_of(s.rx, s.ry, t); vcopy(t.c, x.a); vcopy(t.c, x.b); break;
        case STEP_R_U: vcopy(s.rx, x.a); vcopy(s.ry, x.b); break;
        case STEP_A_CHK: case STEP_R_CHK: {
            const uint32_t* px = r == STEP_A_CHK ? s.ax : s.rx;
            const uint32_t* py = r == STEP_A_CHK ? s.ay : s.ry;
            xy_of(px, py, t);
            vcopy(t.c, l0);
            mul1(l0, l0, t);  // v = u u
            const_limbs(K_D, x.a);
            vcopy(t.c, x.b);
            vcopy(px, x.e);
            vcopy(py, x.f);
            limbs_t m1;
            const_limbs(K_M1, m1);
            for (int i = 0; i < 16; i++) cfix[i] = (uint32_t)m1[i];
            if (s.active) fixed = cfix;
            break;
        }
        case STEP_S_AA: vadd(s.ay, s.ax, x.a); const_limbs(K_B_YMX, x.b); break;
        case STEP_S_BB: vsub(s.ay, s.ax, x.a); const_limbs(K_B_YPX, x.b); break;
        case STEP_S_CC: vcopy(s.nt, x.a); const_limbs(K_B_T, x.b); break;
        case STEP_S_X: sum_operands(s, se, sf, sg, sh); set2(se, sf); break;
        case STEP_S_Y: sum_operands(s, se, sf, sg, sh); set2(sg, sh); break;
        case STEP_S_Z: sum_operands(s, se, sf, sg, sh); set2(sf, sg); break;
        case STEP_S_T: sum_operands(s, se, sf, sg, sh); set2(se, sh); break;
        case STEP_S_PT: const_limbs(K_D2, x.a); vcopy(s.st3, x.b); break;
        case STEP_XCMP:
            vcopy(s.rx, x.a);
            vcopy(out.z, x.b);
            for (int i = 0; i < 16; i++) cfix[i] = out.x[i];
            if (s.active) fixed = cfix;
            break;
        default: break;
    }
    int64_t prod[32];
    for (int k = 0; k < 32; k++) prod[k] = 0;
    fp::mul_acc(prod, x.a, x.b);
    fp::mul_acc(prod, x.e, x.e);
    fp::mul_acc(prod, x.f, x.f, -1);
    return fp::finish(prod, x.u, fixed);  // false: a curve equation or the final comparison does not hold
}

struct NoSink {
    FP_HD void unit(int, const fp::Unit&) {}
};

// The same row on values only (fe25519_fast.hpp): what the sequential scan needs - the canonical point after the row.
struct FastSlot {
    fe::Fe ymx[4], ypx[4], t2d[4], z2[4];  // the addend for (s_bit + 2 h_bit)
};
FP_HD inline void fast_slot(const Slot& s, FastSlot& f) {
    for (int sel = 0; sel < 4; sel++) {
        limbs_t ymx, ypx, t2d, z2;
        addend(s, sel & 1, sel >> 1, ymx, ypx, t2d, z2);
        for (int v = 0; v < 4; v++) {
            const int32_t* src = v == 0 ? ymx : (v == 1 ? ypx : (v == 2 ? t2d : z2));
            // signed limbs -> a non-negative representative: add p limb-wise where needed is not necessary, fe::Fe limbs
            // are signed; regroup 16-bit signed limbs into 26-bit signed limbs through the value's two's-complement-free sum
            fe::Fe r;
            for (int i = 0; i < 10; i++) r.l[i] = 0;
            for (int i = 0; i < 16; i++) {
                // limb i sits at bit 16 i = 26 q + o
                const int q = (16 * i) / 26, o = (16 * i) % 26;
                r.l[q] += (int64_t)src[i] * ((int64_t)1 << o);
            }
            fe::carry10(r.l);
            (v == 0 ? f.ymx : (v == 1 ? f.ypx : (v == 2 ? f.t2d : f.z2)))[sel] = r;
        }
    }
}
struct FastPoint { fe::Fe x, y, z; };
FP_HD inline void fast_row(FastPoint& q, bool sbit, bool hbit, const FastSlot& s) {
    const fe::Fe a = fe::mul(q.x, q.x), b = fe::mul(q.y, q.y), zz = fe::mul(q.z, q.z);
    const fe::Fe xy = fe::add(q.x, q.y);
    const fe::Fe e1 = fe::mul(xy, xy);
    const fe::Fe e = fe::sub(fe::sub(e1, a), b), g = fe::sub(b, a), f = fe::sub(g, fe::dbl(zz)), h = fe::neg(fe::add(a, b));
    const fe::Fe x2 = fe::mul(e, f), y2 = fe::mul(g, h), t2 = fe::mul(e, h), z2 = fe::mul(f, g);
    const int sel = (sbit ? 1 : 0) + (hbit ? 2 : 0);
    const fe::Fe ymx_in = fe::sub(y2, x2), ypx_in = fe::add(y2, x2);
    fe::Fe pa, pb, pc, pd;
    if (sel == 0) {  // the neutral addend (1, 1, 0, 2)
        pa = ymx_in;
        pb = ypx_in;
        for (int i = 0; i < 10; i++) pc.l[i] = 0;
        pd = fe::dbl(z2);
    } else {
        pa = fe::mul(ymx_in, s.ymx[sel]);
        pb = fe::mul(ypx_in, s.ypx[sel]);
        pc = fe::mul(t2, s.t2d[sel]);
        pd = fe::mul(z2, s.z2[sel]);
    }
    const fe::Fe e_ = fe::sub(pb, pa), f_ = fe::sub(pd, pc), g_ = fe::add(pd, pc), h_ = fe::add(pb, pa);
    q.x = fe::mul(e_, f_);
    q.y = fe::mul(g_, h_);
    q.z = fe::mul(f_, g_);
}
FP_HD inline void fast_store(const FastPoint& q, Point& p) {
    fe::freeze(q.x, p.x);
    fe::freeze(q.y, p.y);
    fe::freeze(q.z, p.z);
}

// All round-0 cells of row r of a slot (multiplicity columns zero) through put(column, value).  `in` is the row's
// input point; the row's result is returned in `out`.  Returns false if the row's auxiliary check (curve equation of
// A or R, final comparison) is not satisfied by the slot's data - the signature does not verify.
template <class Put>
struct EmitSink {
    Put& put;
    FP_HD void unit(int index, const fp::Unit& u) { cells(cMAIN + (uint32_t)index * UNIT_CELLS, u); }
    FP_HD void cells(uint32_t base, const fp::Unit& u) { u.cells(base, put); }
};
FP_HD inline uint64_t gl_signed(int32_t v) { return v >= 0 ? (uint64_t)v : 0xFFFFFFFF00000001ull - (uint64_t)(-(int64_t)v); }

template <class Put>
FP_HD inline bool emit_row(int r, const Slot& s, const Point& in, const uint32_t* prev_ry, const Point* prev_final, bool prev_active,
                           Put& put, Point& out) {
    const int bit = ROWS - 1 - r;
    const int sbit = (int)((s.sw[bit >> 4] >> (bit & 15)) & 1), hbit = (int)((s.hw[bit >> 4] >> (bit & 15)) & 1);
    // limb accumulators: the bits of this limb seen so far, MSB first
    const int limb = bit >> 4, seen = 16 - (bit & 15);
    put(cSB, (uint64_t)sbit);
    put(cHB, (uint64_t)hbit);
    put(cSA, (uint64_t)(s.sw[limb] >> (16 - seen)));
    put(cHA, (uint64_t)(s.hw[limb] >> (16 - seen)));
    for (int i = 0; i < 16; i++) {
        put(cSIN + i, in.x[i]);
        put(cSIN + 16 + i, in.y[i]);
        put(cSIN + 32 + i, in.z[i]);
        put(cAX + i, s.ax[i]);
        put(cAY + i, s.ay[i]);
        put(cRX + i, s.rx[i]);
        put(cRY + i, s.ry[i]);
        put(cNT + i, s.nt[i]);
        put(cSW + i, s.sw[i]);
        put(cHW + i, s.hw[i]);
        put(cSAA + i, s.saa[i]);
        put(cSBB + i, s.sbb[i]);
        put(cSCC + i, s.scc[i]);
        put(cSX3 + i, s.sx3[i]);
        put(cSY3 + i, s.sy3[i]);
        put(cSZ3 + i, s.sz3[i]);
        put(cSPT + i, s.spt[i]);
    }
    {
        // the row that closes block j carries limb j of A and R through the looked-up cells (their range check)
        const bool close = (r & 15) == 15;
        const int j = 15 - (r >> 4);
        put(cCHK, close ? s.ax[j] : 0u);
        put(cCHK + 1, close ? s.ay[j] : 0u);
        put(cCHK + 2, close ? s.rx[j] : 0u);
        put(cCHK + 3, close ? s.ry[j] : 0u);
        put(cCHKQ, close ? s.qw[j] : 0u);
        put(cCHKQ + 1, r == ROWS - 1 ? s.qw[16] : 0u);
    }
    {
        // the reduction mod L: per-slot D and q, the carry into position r, the complements' limb r and carry bits
        put(cACT, s.active);
        for (int i = 0; i < 32; i++) put(cDW + i, s.dw[i]);
        for (int i = 0; i < 17; i++) put(cQW + i, s.qw[i]);
        const uint32_t c = r <= MODL_ROWS ? s.carry[r] : 0u;
        put(cCLO, c & 0xFFFF);
        put(cCHI, c >> 16);
        put(cDH, r < 16 ? s.dh[r] : 0u);
        put(cDS, r < 16 ? s.ds[r] : 0u);
        put(cBH, r < 16 ? (s.bh >> r) & 1u : 0u);
        put(cBS, r < 16 ? (s.bs >> r) & 1u : 0u);
        // the encodings' sign bits, the coordinates' complements to p - 1, limb 0 of x halved (parity)
        put(cSGA, s.ax[0] & 1u);
        put(cSGR, s.rx[0] & 1u);
        for (int v = 0; v < 4; v++) {
            put(cCXY + v, r < 16 ? s.cxy[v][r] : 0u);
            put(cBXY + v, r < 16 ? (s.bxy[v] >> r) & 1u : 0u);
        }
        put(cKXA, r == 0 ? s.ax[0] >> 1 : 0u);
        put(cKXR, r == 0 ? s.rx[0] >> 1 : 0u);
    }
    {
        limbs_t ymx, ypx, t2d, z2;
        addend(s, sbit, hbit, ymx, ypx, t2d, z2);
        for (int i = 0; i < 16; i++) {
            put(cP2 + i, gl_signed(ymx[i]));
            put(cP2 + 16 + i, gl_signed(ypx[i]));
            put(cP2 + 32 + i, gl_signed(t2d[i]));
            put(cP2 + 48 + i, gl_signed(z2[i]));
        }
    }
    EmitSink<Put> sink{put};
    row_main(sink, in, sbit, hbit, s, out);
    AuxRow ax;
    const bool ok = row_aux(r, s, out, prev_ry, prev_final, prev_active, ax);   // S >= L (s.s_in_range == 0) is the caller's to report
    for (int i = 0; i < 16; i++) {
        put(cAUX_A + i, gl_signed(ax.a[i]));
        put(cAUX_B + i, gl_signed(ax.b[i]));
        put(cAUX_E + i, gl_signed(ax.e[i]));
        put(cAUX_F + i, gl_signed(ax.f[i]));
    }
    sink.cells(cAUX, ax.u);
    put(cMULT, 0);
    put(cMULT9, 0);
    return ok;
}

}  // namespace ed
}  // namespace nlx
