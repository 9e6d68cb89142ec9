"""Ed25519 signature verification as an AIR (`curta_eddsa_verify_sigs_conditional`,
nearx/src/builder.rs:152 - the largest share of a Sync proof; curta's own tables are not in the reference tree,
Cargo.lock:6515, so this is an independent construction on the pieces of this package: fp25519 units, logup range
checks, the multi-round prover).

Statement per signature slot (256 consecutive rows), for the slot's A = (AX, AY), R = (RX, RY) (affine, 16-bit limbs),
S (16 limbs), D (32 limbs: the 512-bit SHA-512 digest of R || A || M read as a little-endian integer - sha512_air.py
proves the digest itself) and the flag `active`:

    h = D mod L   (0 <= h < L, D = q L + h limb by limb with range-checked q and carries),   S < L,   and, IF active:
    A and R are on the twisted Edwards curve -x^2 + y^2 = 1 + d x^2 y^2 over F_p, p = 2^255 - 19, and [S] B + [h] (-A) = R.

An inactive slot (`curta_eddsa_verify_sigs_conditional`'s is_active = false, nearx/src/builder.rs:137-158: a validator that
did not sign, or the dummy key) runs the same rows with its three checks switched off.  Row r of a slot handles bit
255 - r of both scalars (Shamir's trick):

    Q <- 2 Q                                      8 units   (dbl-2008-hwcd, a = -1)
    Q <- Q + (0 | B | -A | B - A)                 7 units   (add-2008-hwcd-3 with the addend in the precomputed form
                                                             (y-x, y+x, 2dt, 2z), selected by the two bits into
                                                             64 operand cells; the result's T is not needed)

15 multiplication units per row plus one auxiliary unit that runs a per-slot program in otherwise idle rows:
2d*x*y of A, the point B - A (one addition, its intermediate and final values kept in per-slot columns), the curve
equation of A and of R, and the final comparison RX * Z = X, RY * Z = Y.  Everything a unit reads is either a
range-checked cell, a constant, or a cell tied by a degree <= 3 constraint to one of those; all unit results, quotients
and carries (and, once per slot, the limbs of A and R) go through the 2^16-table lookup (the carries' high parts through
a 2^9 table).  Constraint degree 3, two commitment rounds (the second is logup.py's columns for the two tables).

The reduction mod L (rows 0 .. 31 of a slot, one 16-bit position of D = q L + h per row; rows 0 .. 15 also carry the two
comparisons h <= L - 1 and S <= L - 1 as additions h + d = L - 1 with range-checked d): q sits in 17 per-slot columns, the
carries in two looked-up cells per row, everything of degree 2.

Binding to public data: a fifth and sixth challenge gamma (drawn with the lookup challenge, after round 0) and a
round-1 accumulator column fold every slot's 96 limbs (the 32-byte encodings of A and R - y with the parity of x on top,
both coordinates proved canonical, so decompression is part of the statement -, S, D low half, D high half, active) into
one Horner fingerprint in the quadratic extension; its total is a ROUND VALUE of the proof (stark.py).  Whoever relies on the proof
recomputes `fingerprint(slots, gamma)` from the tuples it believes were verified - public keys, signatures, SHA-512
digests, activity flags - and compares; h never leaves the proof.  The check is the cofactorless one (ed25519-dalek's
`verify`: [S]B - [h]A == R).
"""
import numpy as np

from . import fp25519 as fp
from . import logup
from .stark import STEP_TAG_LEN, Air, Stark

P = fp.P25519
D = (-121665 * pow(121666, P - 2, P)) % P
D2 = 2 * D % P
L_ORDER = (1 << 252) + 27742317777372353535851937790883648493
BY = 4 * pow(5, P - 2, P) % P


def _recover_x(y, sign):
    x2 = (y * y - 1) * pow(D * y * y + 1, P - 2, P) % P
    x = pow(x2, (P + 3) // 8, P)
    if (x * x - x2) % P:
        x = x * pow(2, (P - 1) // 4, P) % P
    if (x * x - x2) % P:
        return None
    if (x & 1) != sign:
        x = P - x
    return x


BX = _recover_x(BY, 0)
ROWS = 256                       # rows per signature slot
UNIT = fp.UNIT_CELLS             # 63 range-checked cells per unit


class _Layout:
    """Round-0 column allocation; `lookups16` / `lookups9` collect the cells checked against the two range tables."""

    def __init__(self):
        self.n = 0
        self.lookups16, self.lookups9 = [], []

    def take(self, count, lookup=False):
        base = self.n
        self.n += count
        if lookup:
            self.lookups16 += list(range(base, base + count))
        return base

    def take_unit(self):
        """c[16] q[17] lo[15] in the 2^16 table, hi[15] in the 2^9 table"""
        base = self.n
        self.n += fp.UNIT_CELLS
        self.lookups16 += list(range(base, base + fp.UNIT_CELLS16))
        self.lookups9 += list(range(base + fp.UNIT_CELLS16, base + fp.UNIT_CELLS))
        return base


LAY = _Layout()
SIN = LAY.take(48)                               # the row's input point X, Y, Z
SB, HB, SA, HA = (LAY.take(1) for _ in range(4))  # scalar bits and their 16-bit limb accumulators
AX, AY, RX, RY = (LAY.take(16) for _ in range(4))
NT, SW, HW = (LAY.take(16) for _ in range(3))     # 2d x y of A; limbs of S and h
CHK = LAY.take(4, True)                          # on the last row of limb j's block: limb j of AX, AY, RX, RY (range check)
# B - A: the three products of its addition, then its X, Y, Z and 2d T (per-slot columns tied to auxiliary results)
SAA, SBB, SCC, SX3, SY3, SZ3, SPT = (LAY.take(16) for _ in range(7))
P2 = LAY.take(64)                                # the row's addend (y-x, y+x, 2dt, 2z), selected by the two bits
N_MAIN = 15
MAIN = [LAY.take_unit() for _ in range(N_MAIN)]
AUX_A, AUX_B, AUX_E, AUX_F = (LAY.take(16) for _ in range(4))
AUX = LAY.take_unit()
ACT = LAY.take(1)                                # per slot: 1 = the signature is checked, 0 = the three checks are off
DW = LAY.take(32)                                # per slot: the 512-bit digest D, 16-bit limbs
QW = LAY.take(17)                                # per slot: q = D div L
CHKQ = LAY.take(2, True)                         # range check of q: limb j on the row that closes block j; limb 16 on the slot's last row
CLO = LAY.take(1, True)                          # row k < 32: carry into position k of q L + h = D (low 16 bits) ...
DH, DS = LAY.take(1, True), LAY.take(1, True)    # row j < 16: limb j of L - 1 - h and of L - 1 - S
CHI = LAY.take(1)                                # ... and its high part (2^9 table)
LAY.lookups9.append(CHI)
BH, BS = LAY.take(1), LAY.take(1)                # row j <= 16: carry bit into limb j of h + (L - 1 - h) and S + (L - 1 - S)
SGA, SGR = LAY.take(1), LAY.take(1)              # per slot: the sign bits of the encodings of A and R = the parity of A.x, R.x
CXY = LAY.take(4, True)                          # row j < 16: limb j of p - 1 - v for v = A.x, A.y, R.x, R.y (canonical coordinates)
KXA, KXR = LAY.take(1, True), LAY.take(1, True)  # row 0: (limb 0 of A.x, R.x) div 2
BXY = LAY.take(4)                                # row j <= 16: the carry bits of those four additions
MULT9, MULT = LAY.take(1), LAY.take(1)     # multiplicities of the 2^9 table, then (last: it may grow) of the 2^16 table
LOOKUPS, LOOKUPS9 = list(LAY.lookups16), list(LAY.lookups9)


def layout(table_cols=1):
    """Column counts when the 2^16 table is spread over `table_cols` columns (proofs of fewer than 2^8 slots: a trace of
    2^15 rows carries the table in two columns of 2^15, and so on): the multiplicity columns MULT .. MULT + table_cols - 1
    close round 0; round 1 = lookup columns of the 2^16 table, of the 2^9 table, the fingerprint accumulator."""
    n0 = LAY.n + table_cols - 1
    n1a, n1b = logup.round_cols(len(LOOKUPS), table_cols), logup.round_cols(len(LOOKUPS9))
    return dict(n_cols0=n0, n_cols1a=n1a, n_cols1b=n1b, acc=n0 + n1a + n1b, n_cols1=n1a + n1b + 2)


N_COLS0, N_COLS1A, N_COLS1B, ACC, N_COLS1 = (layout()[k] for k in ("n_cols0", "n_cols1a", "n_cols1b", "acc", "n_cols1"))
# what the fingerprint absorbs per limb index j, limb 15 first: the 32-byte ENCODINGS of A and R (y with the sign of x in
# bit 255: (y column base, sign column)), S, the two halves of D, then `active` (limb 0 only)
BOUND = ((AY, SGA), (RY, SGR), (SW, None), (DW, None), (DW + 16, None))
N_BOUND = len(BOUND) + 1                   # values per limb index
PM1_LIMBS = [((P - 1) >> (16 * i)) & 0xFFFF for i in range(16)]
L_LIMBS = [(L_ORDER >> (16 * i)) & 0xFFFF for i in range(16)]
LM1_LIMBS = [((L_ORDER - 1) >> (16 * i)) & 0xFFFF for i in range(16)]
MODL_ROWS = 32                             # positions of D = q L + h, one per row from the slot's first row
# main unit indices
(U_A, U_B, U_ZZ, U_E, U_X2, U_Y2, U_T2, U_Z2, U_PA, U_PB, U_PC, U_PD, U_X4, U_Y4, U_Z4) = range(N_MAIN)
AUX_STEPS = {"ycmp": 0, "a_u": 1, "a_nt": 2, "a_u2": 3, "a_v": 4, "a_chk": 5, "r_u": 6, "r_v": 7, "r_chk": 8,
             "s_aa": 9, "s_bb": 10, "s_cc": 11, "s_x": 12, "s_y": 13, "s_z": 14, "s_t": 15, "s_pt": 16, "xcmp": 255}
B_YMX, B_YPX, B_T2D, B_T = (BY - BX) % P, (BY + BX) % P, D2 * BX * BY % P, BX * BY % P


class _Vec:
    """16 limb expressions (or integers) with a bound on their magnitude.  Lazy: `limbs` builds the expressions
    afresh at every use, so two units reading the same linear combination (E, F, G, H of an addition) do not share
    expression nodes - shared nodes would stay in VM registers from their first use to their last, 64 of them across
    the four products of an addition, and the register file is what limits the quotient kernel's occupancy."""

    def __init__(self, make, bound):
        self._make, self.bound = (make if callable(make) else (lambda v=list(make): list(v))), bound

    @property
    def limbs(self):
        return self._make()

    def __add__(self, o):
        return _Vec(lambda: [a + b for a, b in zip(self.limbs, o.limbs)], self.bound + o.bound)

    def __sub__(self, o):
        return _Vec(lambda: [a - b for a, b in zip(self.limbs, o.limbs)], self.bound + o.bound)

    def scale(self, k):
        return _Vec(lambda: [a * k for a in self.limbs], self.bound * abs(k))


def ed25519_air(max_resident_leaves=None, table_cols=1, segment_nodes=None, tagged=False):
    lay = layout(table_cols)
    ACC = lay["acc"]  # noqa: N806  (shadows the module constant, which is the table_cols = 1 value)
    air = Air(lay["n_cols0"] + lay["n_cols1"], STEP_TAG_LEN if tagged else 0, rounds=[(lay["n_cols0"], 4), (lay["n_cols1"], 0)], round_values=[0, 2])
    if max_resident_leaves is not None:
        air.max_resident_leaves = max_resident_leaves
    # 400-node segments put the units whose operands are plain cells into their own low-register launch groups
    # (15-16 registers instead of ~40): quotient 30.9 -> 26.5 ms at 2^10 slots (profiles/)
    air.segment_nodes = 400 if segment_nodes is None else segment_nodes
    L, N = air.local, air.next  # noqa: N806

    def cells(base, nxt=False):
        return _Vec([(N if nxt else L)(base + i) for i in range(16)], 1 << 16)

    def unit_cells(base):
        q = [L(base + 16 + i) for i in range(17)]
        carries = [(L(base + 33 + m), L(base + 48 + m)) for m in range(fp.N_CARRY)]
        return q, carries

    def unit(base, products, c=None):
        """products: [(vec a, vec b, sign)]; result cells at `base` unless c (a _Vec of cells / constants) is given."""
        q, carries = unit_cells(base)
        out = cells(base) if c is None else c
        fp.mul_unit_constraints(air, [(a.limbs, b.limbs, sg, a.bound * b.bound) for a, b, sg in products], out.limbs, q, carries)
        return cells(base)

    def periodic_rows(rows):
        return air.periodic([1 if r in rows else 0 for r in range(ROWS)])

    is_last = periodic_rows({ROWS - 1})
    limb_end = periodic_rows({r for r in range(ROWS) if r % 16 == 15})      # also: "the next row starts a limb"
    block = [periodic_rows(set(range(16 * (15 - j), 16 * (15 - j) + 16))) for j in range(16)]   # rows of limb j (MSB first)
    step = {name: periodic_rows({r}) for name, r in AUX_STEPS.items()}

    # ---- scalars: bits, limb accumulators, limbs ----
    sb, hb = L(SB), L(HB)
    air.constraint(sb * (sb - 1))
    air.constraint(hb * (hb - 1))
    for acc, bit, words in ((SA, SB, SW), (HA, HB, HW)):
        air.constraint(N(acc) - (1 - limb_end) * L(acc) * 2 - N(bit))
        cur = block[0] * L(words)
        for j in range(1, 16):
            cur = cur + block[j] * L(words + j)
        air.constraint(limb_end * (L(acc) - cur))
    # ---- the limbs of A and R are range-checked once per slot: limb j shows up in a looked-up cell on the row that
    # closes block j (the other rows of those four cells are free in-range values) ----
    for k, base in enumerate((AX, AY, RX, RY)):
        cur = block[0] * L(base)
        for j in range(1, 16):
            cur = cur + block[j] * L(base + j)
        air.constraint(limb_end * (L(CHK + k) - cur))
    # ---- per-slot columns stay constant inside a slot ----
    for base in (AX, AY, RX, RY, NT, SW, HW, SAA, SBB, SCC, SX3, SY3, SZ3, SPT):
        for i in range(16):
            air.constraint((1 - is_last) * (N(base + i) - L(base + i)))
    for col in [ACT] + list(range(DW, DW + 32)) + list(range(QW, QW + 17)):
        air.constraint((1 - is_last) * (N(col) - L(col)))
    act = L(ACT)
    air.constraint(act * (act - 1))

    # ---- h = D mod L: position k of  q L + h = D  on row k of the slot (k < 32), carries through two looked-up cells;
    # h <= L - 1 and S <= L - 1 as additions with a range-checked complement on rows 0 .. 15 ----
    e_row = [periodic_rows({k}) for k in range(MODL_ROWS)]
    m32, m16 = periodic_rows(set(range(MODL_ROWS))), periodic_rows(set(range(16)))
    lm1 = air.periodic([LM1_LIMBS[r] if r < 16 else 0 for r in range(ROWS)])
    conv = None
    for i in range(17):
        coeff = air.periodic([L_LIMBS[r - i] if (r < MODL_ROWS and 0 <= r - i < 16) else 0 for r in range(ROWS)])
        term = L(QW + i) * coeff
        conv = term if conv is None else conv + term
    hsel, ssel, dsel = e_row[0] * L(HW), e_row[0] * L(SW), e_row[0] * L(DW)
    for k in range(1, MODL_ROWS):
        if k < 16:
            hsel, ssel = hsel + e_row[k] * L(HW + k), ssel + e_row[k] * L(SW + k)
        dsel = dsel + e_row[k] * L(DW + k)
    carry_in, carry_out = L(CLO) + L(CHI) * 65536, N(CLO) + N(CHI) * 65536
    air.constraint(conv + hsel - dsel + m32 * (carry_in - carry_out * 65536))
    for cell in (CLO, CHI):
        air.constraint(e_row[0] * L(cell))                 # nothing is carried into position 0 ...
        air.constraint(e_row[MODL_ROWS - 1] * N(cell))     # ... or out of position 31 (q L + h = D exactly, D < 2^512)
    for sel, comp, bit in ((hsel, DH, BH), (ssel, DS, BS)):
        b = L(bit)
        air.constraint(b * (b - 1))
        air.constraint(sel - lm1 + m16 * (L(comp) + b - N(bit) * 65536))
        air.constraint(e_row[0] * b)
        air.constraint(e_row[15] * N(bit))
    # canonical coordinates (v <= p - 1, the same complement addition) and the encodings' sign bits (parity of x): what the
    # fingerprint binds is the 32-byte encoding of A and of R, so "decompression" is part of the statement
    pm1 = air.periodic([PM1_LIMBS[r] if r < 16 else 0 for r in range(ROWS)])
    for v, base in enumerate((AX, AY, RX, RY)):
        vsel = e_row[0] * L(base)
        for k in range(1, 16):
            vsel = vsel + e_row[k] * L(base + k)
        b = L(BXY + v)
        air.constraint(b * (b - 1))
        air.constraint(vsel - pm1 + m16 * (L(CXY + v) + b - N(BXY + v) * 65536))
        air.constraint(e_row[0] * b)
        air.constraint(e_row[15] * N(BXY + v))
    for sign, half, base in ((SGA, KXA, AX), (SGR, KXR, RX)):
        sg = L(sign)
        air.constraint(sg * (sg - 1))
        air.constraint((1 - is_last) * (N(sign) - sg))
        air.constraint(e_row[0] * (L(base) - L(half) * 2 - sg))
    # q's limbs pass through looked-up cells once per slot, like the limbs of A and R
    cur = block[0] * L(QW)
    for j in range(1, 16):
        cur = cur + block[j] * L(QW + j)
    air.constraint(limb_end * (L(CHKQ) - cur))
    air.constraint(is_last * (L(CHKQ + 1) - L(QW + 16)))

    # ---- doubling of the input point ----
    x1, y1, z1 = cells(SIN), cells(SIN + 16), cells(SIN + 32)
    a_ = unit(MAIN[U_A], [(x1, x1, 1)])
    b_ = unit(MAIN[U_B], [(y1, y1, 1)])
    zz = unit(MAIN[U_ZZ], [(z1, z1, 1)])
    xy = x1 + y1
    e1 = unit(MAIN[U_E], [(xy, xy, 1)])
    e = e1 - a_ - b_
    g = b_ - a_
    f = g - zz.scale(2)
    h = (a_ + b_).scale(-1)
    x2 = unit(MAIN[U_X2], [(e, f, 1)])
    y2 = unit(MAIN[U_Y2], [(g, h, 1)])
    t2 = unit(MAIN[U_T2], [(e, h, 1)])
    z2 = unit(MAIN[U_Z2], [(f, g, 1)])

    # ---- the addend: 0, B, -A or B - A by the two bits, in the form (y - x, y + x, 2 d t, 2 z) ----
    ax, ay, nt = cells(AX), cells(AY), cells(NT)
    sx3, sy3, sz3, spt = cells(SX3), cells(SY3), cells(SZ3), cells(SPT)
    w_ = sb * hb
    c_o, c_b, c_n = 1 - sb - hb + w_, sb - w_, hb - w_
    forms = (  # (neutral, B, -A = (-x, y), B - A) for each of the four coordinates
        (fp.to_limbs(1), fp.to_limbs(B_YMX), (ay + ax).limbs, (sy3 - sx3).limbs),
        (fp.to_limbs(1), fp.to_limbs(B_YPX), (ay - ax).limbs, (sy3 + sx3).limbs),
        (fp.to_limbs(0), fp.to_limbs(B_T2D), nt.scale(-1).limbs, spt.limbs),
        (fp.to_limbs(2), fp.to_limbs(2), fp.to_limbs(2), sz3.scale(2).limbs))
    for v, (o_l, b_l, n_l, s_l) in enumerate(forms):
        for i in range(16):
            sel = w_ * s_l[i] + c_n * n_l[i]
            if b_l[i]:
                sel = sel + c_b * b_l[i]
            if o_l[i]:
                sel = sel + c_o * o_l[i]
            air.constraint(L(P2 + 16 * v + i) - sel)
    p_ymx, p_ypx, p_t2d, p_z2 = (_Vec([L(P2 + 16 * v + i) for i in range(16)], 1 << 17) for v in range(4))
    pa = unit(MAIN[U_PA], [(y2 - x2, p_ymx, 1)])
    pb = unit(MAIN[U_PB], [(y2 + x2, p_ypx, 1)])
    pc = unit(MAIN[U_PC], [(t2, p_t2d, 1)])
    pd = unit(MAIN[U_PD], [(z2, p_z2, 1)])
    e_, f_, g_, h_ = pb - pa, pd - pc, pd + pc, pb + pa
    x4 = unit(MAIN[U_X4], [(e_, f_, 1)])
    y4 = unit(MAIN[U_Y4], [(g_, h_, 1)])
    z4 = unit(MAIN[U_Z4], [(f_, g_, 1)])
    # ---- the next row starts from this row's result, a new slot from the neutral element (0, 1, 1) ----
    for k, vec in enumerate((x4, y4, z4)):
        for i in range(16):
            neutral = 1 if (k > 0 and i == 0) else 0
            air.constraint(N(SIN + 16 * k + i) - (vec.limbs[i] + is_last * (neutral - vec.limbs[i])))

    # ---- the auxiliary unit:  A * B + E * E - F * F = C ----
    ua, ub, ue, uf, uc = cells(AUX_A), cells(AUX_B), cells(AUX_E), cells(AUX_F), cells(AUX)
    unit(AUX, [(ua, ub, 1), (ue, ue, 1), (uf, uf, -1)])
    rx, ry = cells(RX), cells(RY)
    d_l, d2_l, m1_l = fp.to_limbs(D), fp.to_limbs(D2), fp.to_limbs(P - 1)

    def tie(sel, lhs, rhs, gated=False):
        """sel * (lhs - rhs) = 0 limb by limb; gated: only in an active slot (the three checks of the statement)"""
        for i in range(16):
            r = rhs[i] if isinstance(rhs, list) else rhs.limbs[i]
            air.constraint((sel * act if gated else sel) * (lhs.limbs[i] - r))

    nxt_a, nxt_b, nxt_c = cells(AUX_A, True), cells(AUX_B, True), cells(AUX, True)
    for pre, u_, v_, chk, px, py in (("a", "a_u2", "a_v", "a_chk", ax, ay), ("r", "r_u", "r_v", "r_chk", rx, ry)):
        tie(step[u_], ua, px)                     # u = x y
        tie(step[u_], ub, py)
        tie(step[u_], nxt_a, uc)                  # v = u u   (the next row reads this row's result)
        tie(step[u_], nxt_b, uc)
        tie(step[v_], nxt_b, uc)                  # check: d v + x x - y y = -1
        tie(step[chk], ua, d_l)
        tie(step[chk], ue, px)
        tie(step[chk], uf, py)
        tie(step[chk], uc, m1_l, gated=True)
    tie(step["a_u"], ua, ax)                      # u = x y, then NT = 2d u
    tie(step["a_u"], ub, ay)
    tie(step["a_u"], nxt_b, uc)
    tie(step["a_nt"], ua, d2_l)
    tie(step["a_nt"], nt, uc)
    # B - A = B + (-A) by the mixed addition with B's constants, -A = (-x, y, 1, -xy):
    #   AA = (y + x) B_ymx, BB = (y - x) B_ypx, CC = -(2dxy)(x_B y_B) = -NT * T_B, D = 2; E = BB - AA, F = 2 - CC,
    #   G = 2 + CC, H = BB + AA; X = E F, Y = G H, Z = F G, T = E H, and the addend's third coordinate 2d T
    saa, sbb, scc = cells(SAA), cells(SBB), cells(SCC)
    two = _Vec(fp.to_limbs(2), 2)
    e_s, h_s = sbb - saa, sbb + saa
    f_s, g_s = scc + two, two - scc                # SCC holds NT * T_B = -CC
    for name, a_src, b_src, dst in (("s_aa", ay + ax, fp.to_limbs(B_YMX), saa), ("s_bb", ay - ax, fp.to_limbs(B_YPX), sbb),
                                    ("s_cc", nt, fp.to_limbs(B_T), scc), ("s_x", e_s, f_s, sx3), ("s_y", g_s, h_s, sy3),
                                    ("s_z", f_s, g_s, sz3), ("s_t", e_s, h_s, None)):
        tie(step[name], ua, a_src)
        tie(step[name], ub, b_src)
        if dst is not None:
            tie(step[name], dst, uc)
    tie(step["s_t"], nxt_b, uc)                   # 2d T: the next row multiplies this row's result by 2d
    tie(step["s_pt"], ua, d2_l)
    tie(step["s_pt"], spt, uc)
    tie(step["xcmp"], ua, rx)                     # RX * Z = X on the slot's last row ...
    tie(step["xcmp"], ub, z4)
    tie(step["xcmp"], uc, x4, gated=True)
    tie(step["xcmp"], nxt_a, ry)                  # ... RY * Z = Y on the row after it
    tie(step["xcmp"], nxt_b, z4)
    tie(step["xcmp"], nxt_c, y4, gated=True)
    no_sq = 1 - step["a_chk"] - step["r_chk"]
    for i in range(16):
        air.constraint(no_sq * L(AUX_E + i))
        air.constraint(no_sq * L(AUX_F + i))

    # ---- fingerprint of the slots' data: acc' = acc gamma^6 + sum_k gamma^(5-k) limb_k on the row that closes block j
    # (limb j of enc(A), enc(R), S, D low, D high; then `active` for j = 0 and nothing for the other j), unchanged
    # elsewhere; starts at 0, and the total after the last row is the proof's round value ----
    def ext_mul(x, y):
        return x[0] * y[0] + x[1] * y[1] * logup.W, x[0] * y[1] + x[1] * y[0]
    gam = [(air.challenge(2), air.challenge(3))]
    for _ in range(N_BOUND - 1):
        gam.append(ext_mul(gam[-1], gam[0]))                  # gam[k] = gamma^(k+1)
    closing = [periodic_rows({16 * (15 - j) + 15}) for j in range(16)]
    acc = (L(ACC), L(ACC + 1))

    def word(j):
        """sum_k gamma^(5-k) value_(k,j) as an extension element: limb j of the five vectors with weights gamma^5 .. gamma,
        the flag (limb 0 only) with weight 1"""
        w0, w1 = (act if j == 0 else None), None
        for k, (base, sign) in enumerate(BOUND):
            g0, g1 = gam[len(BOUND) - 1 - k]
            limb = L(base + j)
            if sign is not None and j == 15:
                limb = limb + L(sign) * 32768
            w0 = g0 * limb if w0 is None else w0 + g0 * limb
            w1 = g1 * limb if w1 is None else w1 + g1 * limb
        return w0, w1

    absorbed = [None, None]                                    # sum_j closing_j word(j): degree 2
    for j in range(16):
        for c, wc in enumerate(word(j)):
            term = closing[j] * wc
            absorbed[c] = term if absorbed[c] is None else absorbed[c] + term
    g_all = gam[N_BOUND - 1]                                   # gamma^6
    grown = ext_mul(acc, (g_all[0] - 1, g_all[1]))             # acc (gamma^6 - 1)
    total = ext_mul(acc, g_all)
    last_word = word(0)                                        # the last row of the trace closes block 0 of the last slot
    for c in range(2):
        air.constraint_first_row(acc[c])
        air.constraint_transition(N(ACC + c) - (acc[c] + limb_end * grown[c] + absorbed[c]))
        air.constraint_last_row(total[c] + last_word[c] - air.round_value(1, c))
    rc = logup.RangeCheck(air, LOOKUPS, 16, MULT, lay["n_cols0"], table_cols=table_cols)
    rc9 = logup.RangeCheck(air, LOOKUPS9, fp.CARRY_HI_BITS, MULT9, lay["n_cols0"] + lay["n_cols1a"])
    return air, (rc, rc9)


# ---------------------------------------------------------------------------------------------
# plain-Python witness (tests only; the product path generates the trace on the GPU)
# ---------------------------------------------------------------------------------------------
def _vadd(a, b):
    return [x + y for x, y in zip(a, b)]


def _vsub(a, b):
    return [x - y for x, y in zip(a, b)]


def _vscale(a, k):
    return [x * k for x in a]


def _canon_mul(a, b):
    """canonical limbs of a * b for limb vectors (signed limbs allowed)"""
    return fp.to_limbs(fp.from_limbs(a) * fp.from_limbs(b) % P)


def slot_constants(ax, ay):
    """The per-slot values the auxiliary program computes: 2dxy, and B - A with its intermediate products."""
    axl, ayl = fp.to_limbs(ax), fp.to_limbs(ay)
    nt = fp.to_limbs(D2 * ax * ay % P)
    saa = _canon_mul(_vadd(ayl, axl), fp.to_limbs(B_YMX))
    sbb = _canon_mul(_vsub(ayl, axl), fp.to_limbs(B_YPX))
    scc = _canon_mul(nt, fp.to_limbs(B_T))
    two = fp.to_limbs(2)
    e_s, h_s, f_s, g_s = _vsub(sbb, saa), _vadd(sbb, saa), _vadd(scc, two), _vsub(two, scc)
    sx3, sy3, sz3, st3 = _canon_mul(e_s, f_s), _canon_mul(g_s, h_s), _canon_mul(f_s, g_s), _canon_mul(e_s, h_s)
    spt = _canon_mul(fp.to_limbs(D2), st3)
    return dict(nt=nt, saa=saa, sbb=sbb, scc=scc, sx3=sx3, sy3=sy3, sz3=sz3, st3=st3, spt=spt, e=e_s, f=f_s, g=g_s, h=h_s)


def modl_witness(s, d):
    """What the mod-L rows hold for the scalar S and the digest D: h = D mod L, the limbs of q = D div L, the carry into
    every position of q L + h = D (33 values, first and last 0), and for x in (h, S) the limbs of L - 1 - x with the carry
    bits of x + (L - 1 - x) (17 values, first and last 0).  Raises AssertionError if S >= L or D >= 2^512."""
    assert 0 <= d < (1 << 512) and 0 <= s < L_ORDER
    q, h = divmod(d, L_ORDER)
    qw = [(q >> (16 * i)) & 0xFFFF for i in range(17)]
    hw, dw = fp.to_limbs(h), [(d >> (16 * k)) & 0xFFFF for k in range(32)]
    carries = [0]
    for k in range(MODL_ROWS):
        tot = sum(qw[i] * L_LIMBS[k - i] for i in range(17) if 0 <= k - i < 16) + (hw[k] if k < 16 else 0) + carries[-1] - dw[k]
        assert tot % 65536 == 0 and tot >= 0
        carries.append(tot >> 16)
    assert carries[-1] == 0 and max(carries) < (1 << 25)
    comps = []
    for x in (h, s):
        xl, cl, bits = fp.to_limbs(x), fp.to_limbs(L_ORDER - 1 - x), [0]
        for j in range(16):
            tot = xl[j] + cl[j] + bits[-1] - LM1_LIMBS[j]
            assert tot in (0, 65536)
            bits.append(tot >> 16)
        assert bits[-1] == 0
        comps.append((cl, bits))
    return dict(h=h, qw=qw, dw=dw, carries=carries, dh=comps[0][0], bh=comps[0][1], ds=comps[1][0], bs=comps[1][1])


def reference_slot(ax, ay, rx, ry, s, d, active=1):
    """The 256 round-0 rows of one slot (multiplicity columns zero): (N_COLS0, 256) uint64 in the Goldilocks field.
    Raises AssertionError if the statement is false (a unit has no witness)."""
    gl = 0xFFFFFFFF00000001
    t = np.zeros((N_COLS0, ROWS), dtype=np.uint64)
    ml = modl_witness(s, d)
    h = ml["h"]
    t[ACT, :] = 1 if active else 0
    for k in range(32):
        t[DW + k, :] = ml["dw"][k]
    for i in range(17):
        t[QW + i, :] = ml["qw"][i]
    for k in range(MODL_ROWS + 1):
        t[CLO, k], t[CHI, k] = ml["carries"][k] & 0xFFFF, ml["carries"][k] >> 16
    for j in range(17):
        t[BH, j], t[BS, j] = ml["bh"][j], ml["bs"][j]
        if j < 16:
            t[DH, j], t[DS, j] = ml["dh"][j], ml["ds"][j]
            t[CHKQ, 16 * (15 - j) + 15] = ml["qw"][j]
    t[CHKQ + 1, ROWS - 1] = ml["qw"][16]
    for v, x in enumerate((ax, ay, rx, ry)):
        assert 0 <= x < P                                      # canonical coordinates only
        xl, cl, carry = fp.to_limbs(x), fp.to_limbs(P - 1 - x), 0
        for j in range(16):
            t[CXY + v, j], t[BXY + v, j] = cl[j], carry
            carry = (xl[j] + cl[j] + carry - PM1_LIMBS[j]) >> 16
        assert carry == 0
    t[SGA, :], t[SGR, :] = ax & 1, rx & 1
    t[KXA, 0], t[KXR, 0] = (ax & 0xFFFF) >> 1, (rx & 0xFFFF) >> 1

    def put_unit(base, row, products, c=None):
        cl, ql, carries = fp.mul_unit_witness(products, c=c)
        t[base:base + fp.UNIT_CELLS, row] = fp.unit_cell_values(cl, ql, carries)
        return cl

    def put_vec(base, row, limbs):
        t[base:base + 16, row] = [v % gl for v in limbs]

    axl, ayl, rxl, ryl = (fp.to_limbs(v) for v in (ax, ay, rx, ry))
    k = slot_constants(ax, ay)
    one, two, zero = fp.to_limbs(1), fp.to_limbs(2), [0] * 16
    addends = {  # (s_bit, h_bit) -> (y - x, y + x, 2 d t, 2 z)
        (0, 0): (one, one, zero, two),
        (1, 0): (fp.to_limbs(B_YMX), fp.to_limbs(B_YPX), fp.to_limbs(B_T2D), two),
        (0, 1): (_vadd(ayl, axl), _vsub(ayl, axl), _vscale(k["nt"], -1), two),
        (1, 1): (_vsub(k["sy3"], k["sx3"]), _vadd(k["sy3"], k["sx3"]), k["spt"], _vscale(k["sz3"], 2))}
    carried = ((AX, axl), (AY, ayl), (RX, rxl), (RY, ryl), (NT, k["nt"]), (SW, fp.to_limbs(s)), (HW, fp.to_limbs(h)),
               (SAA, k["saa"]), (SBB, k["sbb"]), (SCC, k["scc"]), (SX3, k["sx3"]), (SY3, k["sy3"]), (SZ3, k["sz3"]), (SPT, k["spt"]))
    q = (zero, one, one)                                   # X, Y, Z limbs
    sa = ha = 0
    aux_prev_c = None
    for r in range(ROWS):
        bit = ROWS - 1 - r
        sbit, hbit = (s >> bit) & 1, (h >> bit) & 1
        sa = (0 if r % 16 == 0 else 2 * sa) + sbit
        ha = (0 if r % 16 == 0 else 2 * ha) + hbit
        t[SB, r], t[HB, r], t[SA, r], t[HA, r] = sbit, hbit, sa, ha
        for base, limbs in carried:
            put_vec(base, r, limbs)
        if r % 16 == 15:
            j = 15 - r // 16
            t[CHK:CHK + 4, r] = [axl[j], ayl[j], rxl[j], ryl[j]]
        x1, y1, z1 = q
        put_vec(SIN, r, x1)
        put_vec(SIN + 16, r, y1)
        put_vec(SIN + 32, r, z1)
        a_ = put_unit(MAIN[U_A], r, [(x1, x1)])
        b_ = put_unit(MAIN[U_B], r, [(y1, y1)])
        zz = put_unit(MAIN[U_ZZ], r, [(z1, z1)])
        xy = _vadd(x1, y1)
        e1 = put_unit(MAIN[U_E], r, [(xy, xy)])
        e = _vsub(_vsub(e1, a_), b_)
        g = _vsub(b_, a_)
        f = _vsub(g, _vscale(zz, 2))
        hh = _vscale(_vadd(a_, b_), -1)
        x2 = put_unit(MAIN[U_X2], r, [(e, f)])
        y2 = put_unit(MAIN[U_Y2], r, [(g, hh)])
        t2 = put_unit(MAIN[U_T2], r, [(e, hh)])
        z2 = put_unit(MAIN[U_Z2], r, [(f, g)])
        p2 = addends[(sbit, hbit)]
        for v in range(4):
            put_vec(P2 + 16 * v, r, p2[v])
        pa = put_unit(MAIN[U_PA], r, [(_vsub(y2, x2), p2[0])])
        pb = put_unit(MAIN[U_PB], r, [(_vadd(y2, x2), p2[1])])
        pc = put_unit(MAIN[U_PC], r, [(t2, p2[2])])
        pd = put_unit(MAIN[U_PD], r, [(z2, p2[3])])
        e_, f_, g_, h_ = _vsub(pb, pa), _vsub(pd, pc), _vadd(pd, pc), _vadd(pb, pa)
        x4 = put_unit(MAIN[U_X4], r, [(e_, f_)])
        y4 = put_unit(MAIN[U_Y4], r, [(g_, h_)])
        z4 = put_unit(MAIN[U_Z4], r, [(f_, g_)])
        # auxiliary program
        ua = ub = ue = uf = zero
        c_fixed = None
        st = {v: n for n, v in AUX_STEPS.items()}.get(r)
        if st in ("a_u", "a_u2"):
            ua, ub = axl, ayl
        elif st == "a_nt":
            ua, ub = fp.to_limbs(D2), aux_prev_c
        elif st in ("a_v", "r_v"):
            ua = ub = aux_prev_c
        elif st == "a_chk":
            ua, ub, ue, uf, c_fixed = fp.to_limbs(D), aux_prev_c, axl, ayl, (P - 1 if active else None)
        elif st == "r_u":
            ua, ub = rxl, ryl
        elif st == "r_chk":
            ua, ub, ue, uf, c_fixed = fp.to_limbs(D), aux_prev_c, rxl, ryl, (P - 1 if active else None)
        elif st == "s_aa":
            ua, ub = _vadd(ayl, axl), fp.to_limbs(B_YMX)
        elif st == "s_bb":
            ua, ub = _vsub(ayl, axl), fp.to_limbs(B_YPX)
        elif st == "s_cc":
            ua, ub = k["nt"], fp.to_limbs(B_T)
        elif st == "s_x":
            ua, ub = k["e"], k["f"]
        elif st == "s_y":
            ua, ub = k["g"], k["h"]
        elif st == "s_z":
            ua, ub = k["f"], k["g"]
        elif st == "s_t":
            ua, ub = k["e"], k["h"]
        elif st == "s_pt":
            ua, ub = fp.to_limbs(D2), aux_prev_c
        elif st == "xcmp":
            ua, ub, c_fixed = rxl, z4, (fp.from_limbs(x4) if active else None)
        for base, limbs in ((AUX_A, ua), (AUX_B, ub), (AUX_E, ue), (AUX_F, uf)):
            put_vec(base, r, limbs)
        if st != "ycmp":
            aux_prev_c = put_unit(AUX, r, [(ua, ub, 1), (ue, ue, 1), (uf, uf, -1)], c=c_fixed)
        q = (x4, y4, z4)
    return t, q


def reference_trace(slots):
    """Round-0 trace of a list of slots (ax, ay, rx, ry, s, d[, active]): the Y comparison of slot k sits in row 0 of
    slot k + 1 (cyclically)."""
    parts, finals = [], []
    for sl in slots:
        tr, q = reference_slot(*sl)
        parts.append(tr)
        finals.append(q)
    for k, sl in enumerate(slots):
        prev = (k - 1) % len(slots)
        ry, (x4, y4, z4) = fp.to_limbs(slots[prev][3]), finals[prev]
        tr = parts[k]
        tr[AUX_A:AUX_A + 16, 0] = ry
        tr[AUX_B:AUX_B + 16, 0] = z4
        prev_active = len(slots[prev]) < 7 or slots[prev][6]
        tr[AUX:AUX + fp.UNIT_CELLS, 0] = fp.unit_cell_values(*fp.mul_unit_witness([(ry, z4, 1)], c=fp.from_limbs(y4) if prev_active else None))
    return np.concatenate(parts, axis=1)


def slot_from_signature(public_key, message, signature, active=1):
    """(ax, ay, rx, ry, s, d, active) for an RFC 8032 signature - d = SHA-512(R || A || M) as a little-endian integer -
    or None if the encodings are invalid."""
    import hashlib
    if len(public_key) != 32 or len(signature) != 64:
        return None

    def decode(b):
        y = int.from_bytes(b, "little")
        sign, y = y >> 255, y & ((1 << 255) - 1)
        if y >= P:
            return None
        x = _recover_x(y, sign)
        return None if x is None else (x, y)

    a, r = decode(public_key), decode(signature[:32])
    s = int.from_bytes(signature[32:], "little")
    if a is None or r is None or s >= L_ORDER:
        return None
    d = int.from_bytes(hashlib.sha512(signature[:32] + public_key + message).digest(), "little")
    return a[0], a[1], r[0], r[1], s, d, active


def inactive_slot():
    """A slot whose checks are switched off (a validator that did not sign): the neutral point, zero scalars."""
    return 0, 1, 0, 1, 0, 0, 0


def synthetic_slots(count, seed=1):
    """`count` true statements [S]B = R + [D mod L]A with random keys, nonces and 512-bit D (the AIR does not tie D to a
    hash - sha512_air.py does - so any D with S = r + D a mod L makes a valid slot): the bench's workload."""
    import random
    from .near_protocol import _G, _mul
    rnd = random.Random(seed)

    def affine(pt):
        zi = pow(pt[2], P - 2, P)
        return pt[0] * zi % P, pt[1] * zi % P
    out = []
    for _ in range(count):
        a, r = (rnd.randrange(1, L_ORDER) for _ in range(2))
        d = rnd.randrange(1 << 512)
        (ax, ay), (rx, ry) = affine(_mul(a, _G)), affine(_mul(r, _G))
        out.append((ax, ay, rx, ry, (r + d * a) % L_ORDER, d, 1))
    return out


def _ext_mul(x, y):
    gl = 0xFFFFFFFF00000001
    return (x[0] * y[0] + logup.W * x[1] * y[1]) % gl, (x[0] * y[1] + x[1] * y[0]) % gl


def bound_values(slot):
    """The six 256-bit values of a slot the fingerprint absorbs: the RFC 8032 encodings of A and R (y with the parity of x in
    bit 255 - the public key and the first half of the signature as little-endian integers), S, D mod 2^256, D div 2^256,
    active."""
    ax, ay, rx, ry, s, d = slot[:6]
    return (ay | ((ax & 1) << 255), ry | ((rx & 1) << 255), s, d & ((1 << 256) - 1), d >> 256,
            (1 if len(slot) < 7 or slot[6] else 0))


def public_slot(public_key, signature, digest, active=1):
    """The same six values from what a relying party holds - 32-byte key, 64-byte signature, 64-byte SHA-512 digest - with
    no curve arithmetic: `fingerprint` accepts these tuples as well as full slots."""
    return PublicSlot((int.from_bytes(public_key, "little"), int.from_bytes(signature[:32], "little"),
                       int.from_bytes(signature[32:], "little"), int.from_bytes(digest[:32], "little"),
                       int.from_bytes(digest[32:], "little"), 1 if active else 0))


class PublicSlot(tuple):
    """six bound values, already in the fingerprint's form"""


def fingerprint(slots, gamma):
    """What the proof's round value must be for these slots: Horner in F_p^2 over, slot by slot, limb 15 down to limb 0
    of (enc A, enc R, S, D low, D high, active) - the relying party's side of the binding."""
    acc = (0, 0)
    for sl in slots:
        limbs = [fp.to_limbs(v) for v in (sl if isinstance(sl, PublicSlot) else bound_values(sl))]
        for j in range(15, -1, -1):
            for k in range(N_BOUND):
                acc = _ext_mul(acc, gamma)
                acc = ((acc[0] + limbs[k][j]) % 0xFFFFFFFF00000001, acc[1])
    return acc


def binding_columns(t0, gamma):
    """Round-1 accumulator columns (2, n) and the total, from a round-0 trace (plain Python; tests and the oracle path)."""
    gl = 0xFFFFFFFF00000001
    n = t0.shape[1]
    out = np.zeros((2, n), dtype=np.uint64)
    acc = (0, 0)
    for r in range(n):
        out[0, r], out[1, r] = acc
        if r % 16 == 15:
            j = 15 - (r % ROWS) // 16
            for k in range(N_BOUND):
                acc = _ext_mul(acc, gamma)
                if k < len(BOUND):
                    base, sign = BOUND[k]
                    v = int(t0[base + j, r]) + (int(t0[sign, r]) << 15 if sign is not None and j == 15 else 0)
                else:
                    v = int(t0[ACT, r]) if j == 0 else 0
                acc = ((acc[0] + v) % gl, acc[1])
    return out, acc


SLOT_WORDS = 32


def slots_to_words(slots):
    """[(ax, ay, rx, ry, s, d[, active])] -> (n, 32) uint64: the input format of nlx_ed25519_trace (five 256-bit values,
    the 512-bit digest, the flag, three spare words)"""
    out = np.zeros((len(slots), SLOT_WORDS), dtype=np.uint64)
    for k, sl in enumerate(slots):
        ax, ay, rx, ry, s, d = sl[:6]
        for v, x in enumerate((ax, ay, rx, ry, s, d & ((1 << 256) - 1), d >> 256)):
            for w in range(4):
                out[k, 4 * v + w] = (int(x) >> (64 * w)) & 0xFFFFFFFFFFFFFFFF
        out[k, 28] = 1 if len(sl) < 7 or sl[6] else 0
    return out


class Ed25519Stark:
    """The AIR compiled for 2^log_slots signature slots (256 rows each)."""

    def __init__(self, log_slots, config=None, max_resident_leaves=None, segment_nodes=None, tagged=False):
        if log_slots < 4:
            raise ValueError("at least 2^4 slots per proof")
        self.log_slots = log_slots
        # below 2^8 slots the trace is shorter than the 2^16-entry range table: spread the table over several columns
        self.table_cols = 1 << max(0, 8 - log_slots)
        self.layout = layout(self.table_cols)
        self.air, self.range_checks = ed25519_air(max_resident_leaves, self.table_cols, segment_nodes, tagged)
        self.stark = Stark(self.air, log_slots + 8, config)


class Ed25519Prover:
    """Proves 2^log_slots Ed25519 verifications on one GPU: trace generation (nlx_ed25519_trace), multiplicities and
    lookup columns (nlx_logup_*), two-round STARK (nlx_stark_prove_rounds) - nothing of the trace touches the host."""

    def __init__(self, ctx, log_slots, config=None, max_resident_leaves=None, segment_nodes=None, step_tag=None):
        self.ctx = ctx
        self.step_tag = None if step_tag is None else [int(v) for v in step_tag]   # see stark.step_tag: public inputs 0..3
        self.es = Ed25519Stark(log_slots, config, max_resident_leaves, segment_nodes, tagged=self.step_tag is not None)
        self.stark = self.es.stark
        self.prover = self.stark.build(ctx)
        self._t0 = self._t1 = None

    def generate_trace(self, slots):
        """slots: list of 2^log_slots (ax, ay, rx, ry, s, d[, active]) or their words.  Returns the device round-0 trace
        with multiplicities; raises NlxError if an active slot's statement is false (its comparison unit has no in-range
        witness) or any S >= L."""
        import torch
        from ._lib import dll
        n_slots = 1 << self.es.log_slots
        words = slots if isinstance(slots, np.ndarray) else slots_to_words(slots)
        if words.shape != (n_slots, SLOT_WORDS):
            raise ValueError("expected 2^%d slots" % self.es.log_slots)
        n = n_slots * ROWS
        if self._t0 is None:
            dev = "cuda:%d" % self.ctx.device
            self._t0 = torch.empty((self.es.layout["n_cols0"], n), dtype=torch.int64, device=dev)
            self._t1 = torch.empty((self.es.layout["n_cols1"], n), dtype=torch.int64, device=dev)
        words = np.ascontiguousarray(words, dtype=np.uint64)
        self.ctx.check(dll.nlx_ed25519_trace(self.ctx.handle, words.ctypes.data, self.es.log_slots, self._t0.data_ptr()))
        for rc in self.es.range_checks:
            rc.multiplicities(self.ctx, self._t0)
        return self._t0

    def round1(self, known):
        """Round 1 for the challenges known = [alpha0, alpha1, gamma0, gamma1]: the lookup columns of both tables and the
        binding accumulator into the device round-1 buffer; returns (columns, [total0, total1])."""
        from ._lib import dll
        alpha, gamma = known[:2], np.array([int(known[2]), int(known[3])], dtype=np.uint64)
        rc16, rc9 = self.es.range_checks
        n1a, n1b = self.es.layout["n_cols1a"], self.es.layout["n_cols1b"]
        rc16.round1(self.ctx, self._t0, alpha, self._t1[:n1a])
        rc9.round1(self.ctx, self._t0, alpha, self._t1[n1a:n1a + n1b])
        total = np.zeros(2, dtype=np.uint64)
        self.ctx.check(dll.nlx_ed25519_bind_round(self.ctx.handle, self._t0.data_ptr(), self.es.log_slots, gamma.ctypes.data,
                                                  self._t1[n1a + n1b:].data_ptr(), total.ctypes.data))
        self.last_total = (int(total[0]), int(total[1]))
        return self._t1, [int(total[0]), int(total[1])]

    def prove(self, slots):
        t0 = self.generate_trace(slots)
        return self.prover.prove_rounds(lambda rnd, known: t0 if rnd == 0 else self.round1(known), self.step_tag or [])

    def close(self):
        self.prover.close()
        self._t0 = self._t1 = None
