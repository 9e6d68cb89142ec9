// Witness generation for the Ed25519 verification AIR (near-light-client_amd/ed25519_air.py; SURVEY.md §8a row a12,
// curta_eddsa_verify_sigs_conditional at nearx/src/builder.rs:152).  One trace row = one bit of both scalars: a point
// doubling, a conditional addition of the base point and one of -A, as 21 multiplication units mod 2^255 - 19
// (fp25519.hpp) plus the auxiliary unit's step of the per-slot program.  Plain C++ / device code: the same function
// runs in the sequential per-slot scan (no output), in the row-parallel emitter, and on the host in the tests.
#pragma once
#include "fe25519_fast.hpp"
#include "fp25519.hpp"

namespace nlx {
namespace ed {

constexpr int ROWS = 256, N_MAIN = 21, UNIT_CELLS = fp::UNIT_CELLS;
// round-0 column map (ed25519_air.py)
constexpr uint32_t cSIN = 0, cSB = 48, cHB = 49, cSA = 50, cHA = 51, cAX = 52, cAY = 68, cRX = 84, cRY = 100, cNT = 116,
                   cSW = 132, cHW = 148, cCHK = 164, cMAIN = 168, cAUX_A = cMAIN + N_MAIN * UNIT_CELLS, cAUX_B = cAUX_A + 16,
                   cAUX_E = cAUX_B + 16, cAUX_F = cAUX_E + 16, cAUX = cAUX_F + 16, cMULT = cAUX + UNIT_CELLS,
                   cMULT9 = cMULT + 1, N_COLS0 = cMULT9 + 1;
enum { U_A, U_B, U_ZZ, U_E, U_X2, U_Y2, U_T2, U_Z2, U_BA, U_BB, U_BC, U_X3, U_Y3, U_T3, U_Z3, U_AA, U_AB, U_AC, U_X4, U_Y4, U_Z4 };
enum { STEP_YCMP = 0, STEP_A_U = 1, STEP_A_NT = 2, STEP_A_U2 = 3, STEP_A_V = 4, STEP_A_CHK = 5, STEP_R_U = 6, STEP_R_V = 7,
       STEP_R_CHK = 8, STEP_XCMP = 255 };

typedef int32_t limbs_t[16];

// curve constants as 16-bit limbs, little-endian: d, 2d, the base point's triple (y - x, y + x, 2 d x y), p - 1
static constexpr uint16_t K_LIMBS[6][16] = {
    {0x78a3, 0x1359, 0x4dca, 0x75eb, 0xd8ab, 0x4141, 0x0a4d, 0x0070, 0xe898, 0x7779, 0x4079, 0x8cc7, 0xfe73, 0x2b6f, 0x6cee, 0x5203},
    {0xf159, 0x26b2, 0x9b94, 0xebd6, 0xb156, 0x8283, 0x149a, 0x00e0, 0xd130, 0xeef3, 0x80f2, 0x198e, 0xfce7, 0x56df, 0xd9dc, 0x2406},
    {0x913e, 0xd740, 0x3905, 0x9d10, 0xbeb3, 0xd140, 0x9f05, 0xfd39, 0x8a09, 0x688f, 0x8434, 0xa5c1, 0x1267, 0x98f8, 0x2f92, 0x44fd},
    {0x3b85, 0xf58c, 0x93c6, 0x2fbc, 0x0e19, 0xfb8c, 0x2dc6, 0xcf93, 0x42c2, 0x643d, 0x4898, 0x270b, 0xba65, 0x33d4, 0x9d3a, 0x07cf},
    {0xaa68, 0x877a, 0x1205, 0xabc9, 0xc49e, 0xccaa, 0xe823, 0x26d9, 0x598c, 0xdd43, 0x7dcb, 0x5a1b, 0x65a8, 0x9f0c, 0x7b68, 0x6f11},
    {0xffec, 0xffff, 0xffff, 0xffff, 0xffff, 0xffff, 0xffff, 0xffff, 0xffff, 0xffff, 0xffff, 0xffff, 0xffff, 0xffff, 0xffff, 0x7fff}};
FP_HD inline void const_limbs(int which, limbs_t out) {
    for (int i = 0; i < 16; i++) out[i] = K_LIMBS[which][i];
}

enum { K_D = 0, K_D2 = 1, K_B_YMX = 2, K_B_YPX = 3, K_B_T2D = 4, K_M1 = 5 };

// one product unit: c = a b (canonical), quotient, carries.  Not inlined on the device: a row calls it 22 times, and one
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

// The slot's constant data: limbs of A, R, 2dxy of A, the scalars.
struct Slot {
    uint32_t ax[16], ay[16], rx[16], ry[16], nt[16], sw[16], hw[16];
};

FP_HD inline void slot_from_words(const uint64_t* w /* ax ay rx ry s h: 6 x 4 words */, Slot& s) {
    uint32_t* dst[6] = {s.ax, s.ay, s.rx, s.ry, s.sw, s.hw};
    for (int v = 0; v < 6; v++)
        for (int i = 0; i < 16; i++) dst[v][i] = (uint32_t)((w[4 * v + (i >> 2)] >> (16 * (i & 3))) & 0xFFFF);
    limbs_t a, b, k;
    fp::Unit u;
    vcopy(s.ax, a);
    vcopy(s.ay, b);
    mul1(a, b, u);
    const_limbs(K_D2, k);
    vcopy(u.c, a);
    mul1(k, a, u);
    for (int i = 0; i < 16; i++) s.nt[i] = u.c[i];
}

// Madd with the precomputed triple (ymx, ypx, t2d) of the addend (signed limbs).  Sink::unit(index, Unit) receives
// the seven (six without T) units in the order of the column map.
template <class Sink>
FP_HD inline void madd(Sink& sink, int first_unit, const uint32_t* x, const uint32_t* y, const uint32_t* z, const uint32_t* t,
                       const limbs_t ymx, const limbs_t ypx, const limbs_t t2d, bool want_t, Point& out, uint32_t* t_out) {
    fp::Unit aa, bb, cc, u;
    limbs_t l0;
    vsub(y, x, l0);
    mul1(l0, ymx, aa);
    sink.unit(first_unit, aa);
    vadd(y, x, l0);
    mul1(l0, ypx, bb);
    sink.unit(first_unit + 1, bb);
    vcopy(t, l0);
    mul1(l0, t2d, cc);
    sink.unit(first_unit + 2, cc);
    limbs_t e, f, g, h;
    for (int i = 0; i < 16; i++) {
        const int32_t dd = 2 * (int32_t)z[i];
        e[i] = (int32_t)bb.c[i] - (int32_t)aa.c[i];
        f[i] = dd - (int32_t)cc.c[i];
        g[i] = dd + (int32_t)cc.c[i];
        h[i] = (int32_t)bb.c[i] + (int32_t)aa.c[i];
    }
    mul1(e, f, u);
    sink.unit(first_unit + 3, u);
    for (int i = 0; i < 16; i++) out.x[i] = u.c[i];
    mul1(g, h, u);
    sink.unit(first_unit + 4, u);
    for (int i = 0; i < 16; i++) out.y[i] = u.c[i];
    int nxt = first_unit + 5;
    if (want_t) {
        mul1(e, h, u);
        sink.unit(nxt++, u);
        for (int i = 0; i < 16; i++) t_out[i] = u.c[i];
    }
    mul1(f, g, u);
    sink.unit(nxt, u);
    for (int i = 0; i < 16; i++) out.z[i] = u.c[i];
}

// One row: in -> 2 in + sbit B + hbit (-A).
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
    Point p2, p3;
    uint32_t t2[16], t3[16];
    mul1(e, f, u);
    sink.unit(U_X2, u);
    for (int i = 0; i < 16; i++) p2.x[i] = u.c[i];
    mul1(g, h, u);
    sink.unit(U_Y2, u);
    for (int i = 0; i < 16; i++) p2.y[i] = u.c[i];
    mul1(e, h, u);
    sink.unit(U_T2, u);
    for (int i = 0; i < 16; i++) t2[i] = u.c[i];
    mul1(f, g, u);
    sink.unit(U_Z2, u);
    for (int i = 0; i < 16; i++) p2.z[i] = u.c[i];
    limbs_t ymx, ypx, t2d;
    if (sbit) {
        const_limbs(K_B_YMX, ymx);
        const_limbs(K_B_YPX, ypx);
        const_limbs(K_B_T2D, t2d);
    } else {
        for (int i = 0; i < 16; i++) { ymx[i] = ypx[i] = i == 0; t2d[i] = 0; }
    }
    madd(sink, U_BA, p2.x, p2.y, p2.z, t2, ymx, ypx, t2d, true, p3, t3);
    if (hbit) {  // -A = (-x, y): (y + x, y - x, -2dxy)
        for (int i = 0; i < 16; i++) {
            ymx[i] = (int32_t)s.ay[i] + (int32_t)s.ax[i];
            ypx[i] = (int32_t)s.ay[i] - (int32_t)s.ax[i];
            t2d[i] = -(int32_t)s.nt[i];
        }
    } else {
        for (int i = 0; i < 16; i++) { ymx[i] = ypx[i] = i == 0; t2d[i] = 0; }
    }
    madd(sink, U_AA, p3.x, p3.y, p3.z, t3, ymx, ypx, t2d, false, out, nullptr);
}

// The auxiliary unit of row r (0..255) of a slot: operands (a, b, e, f) and the unit a b + e e - f f = c.
// `out` is this row's result point; prev_* are the previous slot's R_y and final point (row 0's comparison).
struct AuxRow {
    limbs_t a, b, e, f;
    fp::Unit u;
};
FP_HD inline bool row_aux(int r, const Slot& s, const Point& out, const uint32_t* prev_ry, const Point* prev_final, AuxRow& x) {
    for (int i = 0; i < 16; i++) x.a[i] = x.b[i] = x.e[i] = x.f[i] = 0;
    const uint32_t* fixed = nullptr;
    uint32_t cfix[16];
    fp::Unit t;
    limbs_t l0, l1;
    auto xy_of = [&](const uint32_t* px, const uint32_t* py, fp::Unit& o) {
        vcopy(px, l0);
        vcopy(py, l1);
        mul1(l0, l1, o);
    };
    switch (r) {
        case STEP_YCMP:
            vcopy(prev_ry, x.a);
            vcopy(prev_final->z, x.b);
            for (int i = 0; i < 16; i++) cfix[i] = prev_final->y[i];
            fixed = cfix;
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
            fixed = cfix;
            break;
        }
        case STEP_XCMP:
            vcopy(s.rx, x.a);
            vcopy(out.z, x.b);
            for (int i = 0; i < 16; i++) cfix[i] = out.x[i];
            fixed = cfix;
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
    fe::Fe a_ymx, a_ypx, a_t2d;  // the triple of -A: (y + x, y - x, -2dxy)
    fe::Fe b_ymx, b_ypx, b_t2d;
};
FP_HD inline void fast_slot(const Slot& s, FastSlot& f) {
    const fe::Fe x = fe::from_limbs16(s.ax), y = fe::from_limbs16(s.ay), nt = fe::from_limbs16(s.nt);
    f.a_ymx = fe::add(y, x);
    f.a_ypx = fe::sub(y, x);
    f.a_t2d = fe::neg(nt);
    uint32_t k[16];
    for (int i = 0; i < 16; i++) k[i] = K_LIMBS[K_B_YMX][i];
    f.b_ymx = fe::from_limbs16(k);
    for (int i = 0; i < 16; i++) k[i] = K_LIMBS[K_B_YPX][i];
    f.b_ypx = fe::from_limbs16(k);
    for (int i = 0; i < 16; i++) k[i] = K_LIMBS[K_B_T2D][i];
    f.b_t2d = fe::from_limbs16(k);
}
struct FastPoint { fe::Fe x, y, z; };
FP_HD inline void fast_madd(fe::Fe& x, fe::Fe& y, fe::Fe& z, fe::Fe& t, bool bit, const fe::Fe& ymx, const fe::Fe& ypx, const fe::Fe& t2d,
                            bool want_t) {
    // a zero bit adds the neutral triple (1, 1, 0): A = y - x, B = y + x, C = 0
    const fe::Fe ymx_in = fe::sub(y, x), ypx_in = fe::add(y, x);
    const fe::Fe aa = bit ? fe::mul(ymx_in, ymx) : ymx_in, bb = bit ? fe::mul(ypx_in, ypx) : ypx_in;
    fe::Fe cc;
    if (bit) cc = fe::mul(t, t2d);
    else for (int i = 0; i < 10; i++) cc.l[i] = 0;
    const fe::Fe dd = fe::dbl(z);
    const fe::Fe e = fe::sub(bb, aa), f = fe::sub(dd, cc), g = fe::add(dd, cc), h = fe::add(bb, aa);
    x = fe::mul(e, f);
    y = fe::mul(g, h);
    if (want_t) t = fe::mul(e, h);
    z = fe::mul(f, g);
}
FP_HD inline void fast_row(FastPoint& q, bool sbit, bool hbit, const FastSlot& s) {
    const fe::Fe a = fe::mul(q.x, q.x), b = fe::mul(q.y, q.y), zz = fe::mul(q.z, q.z);
    const fe::Fe xy = fe::add(q.x, q.y);
    const fe::Fe e1 = fe::mul(xy, xy);
    const fe::Fe e = fe::sub(fe::sub(e1, a), b), g = fe::sub(b, a), f = fe::sub(g, fe::dbl(zz)), h = fe::neg(fe::add(a, b));
    fe::Fe x = fe::mul(e, f), y = fe::mul(g, h), t = fe::mul(e, h), z = fe::mul(f, g);
    fast_madd(x, y, z, t, sbit, s.b_ymx, s.b_ypx, s.b_t2d, true);
    fast_madd(x, y, z, t, hbit, s.a_ymx, s.a_ypx, s.a_t2d, false);
    q.x = x;
    q.y = y;
    q.z = z;
}
FP_HD inline void fast_store(const FastPoint& q, Point& p) {
    fe::freeze(q.x, p.x);
    fe::freeze(q.y, p.y);
    fe::freeze(q.z, p.z);
}

// All round-0 cells of row r of a slot (multiplicity column zero) through put(column, value).  `in` is the row's input
// point; the row's result is returned in `out`.  Returns false if the row's auxiliary check (curve equation of A or R,
// final comparison) is not satisfied by the slot's data - the signature does not verify.
template <class Put>
struct EmitSink {
    Put& put;
    FP_HD void unit(int index, const fp::Unit& u) { cells(cMAIN + (uint32_t)index * UNIT_CELLS, u); }
    FP_HD void cells(uint32_t base, const fp::Unit& u) { u.cells(base, put); }
};
FP_HD inline uint64_t gl_signed(int32_t v) { return v >= 0 ? (uint64_t)v : 0xFFFFFFFF00000001ull - (uint64_t)(-(int64_t)v); }

template <class Put>
FP_HD inline bool emit_row(int r, const Slot& s, const Point& in, const uint32_t* prev_ry, const Point* prev_final, Put& put,
                           Point& out) {
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
    }
    {
        // the row that closes block j carries limb j of A and R through the looked-up cells (their range check)
        const bool close = (r & 15) == 15;
        const int j = 15 - (r >> 4);
        put(cCHK, close ? s.ax[j] : 0u);
        put(cCHK + 1, close ? s.ay[j] : 0u);
        put(cCHK + 2, close ? s.rx[j] : 0u);
        put(cCHK + 3, close ? s.ry[j] : 0u);
    }
    EmitSink<Put> sink{put};
    row_main(sink, in, sbit, hbit, s, out);
    AuxRow ax;
    const bool ok = row_aux(r, s, out, prev_ry, prev_final, ax);
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
