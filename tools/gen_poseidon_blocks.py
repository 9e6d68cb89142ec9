#!/usr/bin/env python3
"""Tables for the device permutation's FUSED PARTIAL ROUNDS (csrc/poseidon.hpp, csrc/gl32.hpp `partial_block3`).

In a partial round only element 0 passes the S-box, so K consecutive partial rounds are ONE linear map of the twelve inputs plus
K - 1 scalars (the S-box outputs inside the block), and the S-box inputs inside the block are linear in the same data:

    z   = state after round a's S-box (z_0 = u_a)                      Q = P M,  P = diag(0, 1, .., 1),  m0 = row 0 of M
    w_i = m0 Q^(i-1) z + sum_{t<i} (m0 Q^(i-1-t) e0) u_t + kappa_i     u_i = w_i^7                        (i = 1 .. K-1)
    x   = M Q^(K-1) z + sum_t (M Q^(K-1-t) e0) u_t  (+ constants)      = the state before round a+K's S-box

The matrix cores evaluate  [M Q^(K-1) | columns for u_1 .. u_(K-1)]  on byte planes (the S-box inputs w_i are 12-term dot products
with the small rows  rho_i = m0 Q^(i-1)  on the vector pipe: a first matrix pass for them was built and measured slower - the
wave waits for every chain's result with nothing else to do).  The matrix entries are up to 8 K bits wide, so they are split into K balanced base-256 digits in [-128, 127] (the
instruction's bytes are signed) and the product of digit p with the state's byte plane b accumulates - inside the matrix cores,
through the C operand - into output plane b + p.  The planes of the state are biased by 128 (xor 0x80): instead of the exact
correction per plane, every chain starts from ONE seed vector that keeps all plane sums non-negative, and what that adds in
total is a constant per row (E below), which the round constants absorb together with the partial rounds' own constants
(the delta recursion of round 4: constants of elements 1 .. 11 ride along as a known offset of the state and are settled in the
constants of round 26).

Everything here is checked before anything is written: a Python model of the device schedule on these very tables (integer plane
arithmetic, the same digit / seed / constant tables) against the naive permutation and upstream's known answers.

Usage: python tools/gen_poseidon_blocks.py   (rewrites csrc/poseidon_blocks.inc)
"""
import json
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_poseidon_constants import round_constants

P = 0xFFFFFFFF00000001
T, RF_HALF, RP, NR = 12, 4, 22, 30
CIRC = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]
K = 3                                   # rounds per block
FIRST = RF_HALF                         # the blocks cover rounds FIRST .. FIRST + K * N_BLOCKS - 1
N_BLOCKS = (RP - 1) // K                # round 25 (the last partial round) stays a single layer: its constants are twelve again
SLOTS = 16                              # K slots of one state in the instruction (12 state elements, K - 1 scalars, rest zero)
ROWS = 16                               # output rows per state (12 outputs, K - 1 S-box inputs, rest idle)


def mds():
    return [[CIRC[(c - r) % T] + (8 if r == c == 0 else 0) for c in range(T)] for r in range(T)]


def matmul(a, b):
    return [[sum(a[i][x] * b[x][j] for x in range(len(b))) for j in range(len(b[0]))] for i in range(len(a))]


def matvec(a, v, mod=None):
    out = [sum(a[i][j] * v[j] for j in range(len(v))) for i in range(len(a))]
    return [x % mod for x in out] if mod else out


def balanced_digits(v, n):
    """v = sum d_p 256^p with d_p in [-128, 127]"""
    out = []
    for _ in range(n):
        d = ((v + 128) % 256) - 128
        out.append(d)
        v = (v - d) // 256
    assert v == 0, "entry does not fit the digits"
    return out


def block_matrix():
    """ROWS x SLOTS integer matrix of one block: rows 0 .. 11 = [M Q^(K-1) | M Q^(K-1-t) e0 (t = 1 .. K-1)], row 11 + i = m0 Q^(i-1)"""
    M = mds()
    Pm = [[(1 if i == j and i != 0 else 0) for j in range(T)] for i in range(T)]
    Q = matmul(Pm, M)
    Qpow = [[[1 if i == j else 0 for j in range(T)] for i in range(T)]]
    for _ in range(K):
        Qpow.append(matmul(Qpow[-1], Q))
    A = [[0] * SLOTS for _ in range(ROWS)]
    N = matmul(M, Qpow[K - 1])
    for r in range(T):
        for j in range(T):
            A[r][j] = N[r][j]
        for t in range(1, K):
            A[r][T + t - 1] = matmul(M, Qpow[K - 1 - t])[r][0]
    m0 = [M[0]]
    rho = [[0] * T] + [matmul(m0, Qpow[i - 1])[0] for i in range(1, K)]   # rho[i]: S-box input i = rho[i] . z + ...
    gamma = [[0] * K for _ in range(K)]   # gamma[i][t] = m0 Q^(i-1-t) e0, t < i
    for i in range(1, K):
        for t in range(1, i):
            gamma[i][t] = matmul(m0, Qpow[i - 1 - t])[0][0]
    return A, gamma, M, Q, rho


def tables():
    A, gamma, M, Q, rho = block_matrix()
    digits = [[[0] * SLOTS for _ in range(ROWS)] for _ in range(K)]
    for r in range(ROWS):
        for j in range(SLOTS):
            d = balanced_digits(A[r][j], K)
            for p in range(K):
                digits[p][r][j] = d[p]
    # one seed per row: the largest amount any chain's raw sum can go negative (signed bytes in [-128, 127] against the digits)
    def neg(r, ps):
        return sum((128 * a if a > 0 else 127 * -a) for p in ps for a in digits[p][r])
    def pos(r, ps):
        return sum((127 * a if a > 0 else 128 * -a) for p in ps for a in digits[p][r])
    seed = [0] * ROWS
    for r in range(T):
        seed[r] = neg(r, range(K))
    dmax_main = max(seed[r] + pos(r, range(K)) for r in range(T))
    # E[r] = what the seeds add in total: sum_q 256^q (seed - 128 * rowsum of the digits that take part in plane q)
    def excess(r, nd):
        e = 0
        for q in range(8 + nd - 1):
            s = sum(sum(digits[p][r]) for p in range(nd) if 0 <= q - p <= 7)
            e += (seed[r] - 128 * s) << (8 * q)
        return e
    E = [excess(r, K) for r in range(T)]
    return dict(A=A, gamma=gamma, M=M, Q=Q, rho=rho, digits=digits, seed=seed, E=E, dmax_main=dmax_main)


def constants(tb):
    """Per block: kappa_i (seed of S-box input i's recombination, i = 1 .. K-1) and kappa_0 (output 0's); then the twelve constants
    of the layers the blocks do not cover (rcb: layer l's = what round l adds, with the carried offset settled at round 26)."""
    rc = round_constants()
    c = [rc[i * T:(i + 1) * T] for i in range(NR)]
    M, Q, E = tb["M"], tb["Q"], tb["E"]
    delta = [0] * T                       # true state = held state + delta (delta[0] = 0 whenever an S-box is applied)
    blocks = []
    a = FIRST
    for _ in range(N_BLOCKS):
        d = list(delta)
        kap = [0] * K
        for i in range(1, K):
            kap[i] = (sum(M[0][j] * d[j] for j in range(T)) + c[a + i][0]) % P
            d = [(x + (c[a + i][j] if j else 0)) % P for j, x in enumerate(matvec(Q, d, P))]
        g = [(x + c[a + K][j]) % P for j, x in enumerate(matvec(M, d, P))]
        kap[0] = (g[0] - E[0]) % P
        delta = [0] + [(g[j] - E[j]) % P for j in range(1, T)]
        blocks.append(kap)
        a += K
    # single layers: layer l (after round l - 1's S-boxes) adds round l's constants
    layer = {}
    for l in range(1, NR):
        layer[l] = list(c[l])
    assert a == RF_HALF + RP - 1          # round 25 is the single partial round left: its layer settles delta
    md = matvec(M, delta, P)
    layer[a + 1] = [(c[a + 1][j] + md[j]) % P for j in range(T)]
    return blocks, layer


# ---- the model of the device schedule --------------------------------------------------------------------------------------
def sbox(x):
    return pow(x, 7, P)


def naive(state):
    rc = round_constants()
    M = mds()
    s = list(state)
    for r in range(NR):
        s = [(x + rc[r * T + i]) % P for i, x in enumerate(s)]
        if r < RF_HALF or r >= RF_HALF + RP:
            s = [sbox(x) for x in s]
        else:
            s[0] = sbox(s[0])
        s = matvec(M, s, P)
    return s


def planes_of(vals64):
    """8 byte planes (biased to signed bytes) of canonical-or-loose 64-bit values"""
    return [[((v >> (8 * b)) & 0xFF) - 128 for v in vals64] for b in range(8)]


def loose(x, rnd):
    """any 64-bit representative of x, as the device may hold"""
    if x < (1 << 64) - P and rnd.random() < 0.5:
        return x + P
    return x


def block_model(tb, kap, s, rnd):
    digits, seed, gamma = tb["digits"], tb["seed"], tb["gamma"]
    z = list(s)
    z[0] = sbox(z[0])
    zl = [loose(v, rnd) for v in z]
    pl = planes_of(zl)                                    # pl[b][j], j < 12
    def chains(nd, slots):                                # slots[b] = the 16 signed bytes of plane b
        out = []
        for q in range(8 + nd - 1):
            row = []
            for r in range(ROWS):
                acc = seed[r]
                for p in range(nd):
                    if 0 <= q - p <= 7:
                        acc += sum(digits[p][r][j] * slots[q - p][j] for j in range(SLOTS))
                row.append(acc)
            out.append(row)
        return out
    junk = [[rnd.randrange(-128, 128) for _ in range(4)] for _ in range(8)]   # slots 14, 15 hold whatever the registers held
    dmax = 0
    u = [0] * K
    for i in range(1, K):
        w = (sum(tb["rho"][i][j] * zl[j] for j in range(T)) + kap[i] + sum(gamma[i][t] * u[t] for t in range(1, i))) % P
        u[i] = sbox(w)
    ul = [loose(v, rnd) for v in u]
    upl = planes_of(ul[1:])
    main = chains(K, [pl[b] + upl[b] + junk[b][:SLOTS - T - (K - 1)] for b in range(8)])
    out = []
    for r in range(T):
        assert all(main[q][r] >= 0 for q in range(len(main)))
        dmax = max(dmax, max(main[q][r] for q in range(len(main))))
        out.append((sum(main[q][r] << (8 * q) for q in range(len(main))) + (kap[0] if r == 0 else 0)) % P)
    return out, dmax


def model(tb, blocks, layer, state, rnd):
    rc = round_constants()
    M = tb["M"]
    s = [(x + rc[i]) % P for i, x in enumerate(state)]
    for r in range(RF_HALF):
        s = [sbox(x) for x in s]
        s = [(x + layer[r + 1][i]) % P for i, x in enumerate(matvec(M, s, P))]
    dmax = 0
    for b in range(N_BLOCKS):
        s, dm = block_model(tb, blocks[b], s, rnd)
        dmax = max(dmax, dm)
    r = FIRST + K * N_BLOCKS
    s[0] = sbox(s[0])
    s = [(x + layer[r + 1][i]) % P for i, x in enumerate(matvec(M, s, P))]
    for r in range(RF_HALF + RP, NR):
        s = [sbox(x) for x in s]
        s = matvec(M, s, P)
        if r + 1 < NR:
            s = [(x + layer[r + 1][i]) % P for i, x in enumerate(s)]
    return s, dmax


def check(tb, blocks, layer):
    rnd = random.Random(20260404)
    here = os.path.dirname(os.path.abspath(__file__))
    kat = os.path.join(here, "..", "tests", "golden", "primitives.json")   # upstream's known answers (tests/test_oracle_golden.py)
    cases = []
    if os.path.exists(kat):
        for v in json.load(open(kat))["poseidon_kat"]:
            cases.append(([int(x) for x in v["in"]], [int(x) for x in v["out"]]))
    edge = [[0] * T, [P - 1] * T, list(range(T)), [(1 << 64) % P] * T]
    for st in edge + [[rnd.randrange(P) for _ in range(T)] for _ in range(40)]:
        cases.append((st, naive(st)))
    worst = 0
    for st, want in cases:
        assert naive(st) == want, "naive permutation disagrees with the known answer"
        got, dm = model(tb, blocks, layer, st, rnd)
        assert got == want, "block schedule disagrees with the naive permutation"
        worst = max(worst, dm)
    return len(cases), worst


def bounds(tb):
    """the recombinations' no-overflow conditions (gl32.hpp), from the worst plane sums the tables allow"""
    dmax, npl = tb["dmax_main"], 8 + K - 1
    x = dmax * 257                       # d + (d << 8)
    assert x < 1 << 32
    x2 = dmax * (1 + (256 if npl > 9 else 0)) if npl > 8 else 0
    t = x + (x << 32) + (x << 16)
    z1 = (x << 16) >> 32
    assert t + (z1 + x2) * 0xFFFFFFFF < 1 << 64      # fold_pair_x: u = t + (z1 + x2) * EPS cannot carry
    # seeded form: al = x + 2^16 y + const_lo, ah likewise + 2^32 x2; fold_acc: al + ah_hi * EPS < 2^64
    al = x + (x << 16) + (1 << 32)
    ah_hi = ((x + (x << 16) + (1 << 32)) >> 32) + x2
    assert ah_hi < 1 << 32 and al + ah_hi * 0xFFFFFFFF < 1 << 64
    # the S-box inputs: al = sum rho_j lo_j + gamma u_lo + const_lo, ah likewise; fold_acc as above
    for i in range(1, K):
        acc = (sum(tb["rho"][i]) + sum(tb["gamma"][i]) + 1) * 0xFFFFFFFF
        assert acc < 1 << 63 and acc + (acc >> 32) * 0xFFFFFFFF < 1 << 64
    assert max(max(g) for g in tb["gamma"]) < 64        # an inline constant of the multiply-add


def fragment_words(tb):
    """per digit matrix and lane: the A operand's four dwords (row = lane & 31, K block = lane >> 5; a state's rows are placed so
    that no lane moves data: gl32.hpp mds_a_fragment)"""
    out = []
    for p in range(K):
        lanes = []
        for lane in range(64):
            rho, h = lane & 31, lane >> 5
            g, r = (rho >> 2) & 1, (rho & 3) + 4 * (rho >> 3)
            w = [0, 0, 0, 0]
            if h == g:
                for j in range(SLOTS):
                    w[j >> 2] |= (tb["digits"][p][r][j] & 0xFF) << (8 * (j & 3))
            lanes.append(w)
        out.append(lanes)
    return out


def emit(tb, blocks, layer, path):
    fr = fragment_words(tb)
    L = ["// GENERATED by tools/gen_poseidon_blocks.py - do not edit.  Fused partial rounds: K = %d rounds per block, %d blocks" % (K, N_BLOCKS),
         "// (rounds %d .. %d), checked against the naive permutation and upstream's known answers before it was written." % (FIRST, FIRST + K * N_BLOCKS - 1),
         "#define NLX_POSEIDON_BLOCK_K %d" % K,
         "#define NLX_POSEIDON_N_BLOCKS %d" % N_BLOCKS,
         "#define NLX_POSEIDON_BLOCK_DMAX %d" % tb["dmax_main"],
         "// A operand of digit matrix p for lane l: four dwords at [(p * 64 + l) * 4]",
         "#define NLX_POSEIDON_BLOCK_FRAGMENTS_INIT { \\"]
    for p in range(K):
        for lane in range(64):
            L.append("    " + ", ".join("0x%08xu" % w for w in fr[p][lane]) + ", \\")
    L[-1] = L[-1][:-3] + " }"
    L.append("// chain seeds per output row (rows 12 .. 15 idle)")
    L.append("#define NLX_POSEIDON_BLOCK_SEED_INIT { " + ", ".join("%du" % s for s in tb["seed"]) + " }")
    L.append("// S-box input i = rho_i . z + gamma[i][t] u_t (t < i) + kappa_i")
    for i in range(1, K):
        L.append("#define NLX_POSEIDON_BLOCK_RHO%d_INIT { " % i + ", ".join("%du" % x for x in tb["rho"][i]) + " }")
    L.append("#define NLX_POSEIDON_BLOCK_GAMMA21 %du" % tb["gamma"][2][1] if K >= 3 else "#define NLX_POSEIDON_BLOCK_GAMMA21 0u")
    L.append("// per block: [2 i] / [2 i + 1] = low / high half of kappa_i (i = 0: output 0's constant, i >= 1: S-box input i's)")
    L.append("#define NLX_POSEIDON_BLOCK_KAPPA_INIT { \\")
    for kap in blocks:
        L.append("    " + ", ".join("0x%08xull, 0x%08xull" % (k & 0xFFFFFFFF, k >> 32) for k in kap) + ", \\")
    L[-1] = L[-1][:-3] + " }"
    L.append("// single layers: for layer l = 1 .. 29, [l * 24 + r] / [l * 24 + 12 + r] = low / high half of the constant output r takes")
    L.append("// (layers inside the blocks: unused, zero; layer %d settles the offset the blocks carried)" % (FIRST + K * N_BLOCKS + 1))
    L.append("#define NLX_POSEIDON_LAYER_RCB_INIT { \\")
    covered = set(range(FIRST + 1, FIRST + K * N_BLOCKS + 1))
    row = ["0x0ull"] * 24
    L.append("    " + ", ".join(row) + ", \\")
    for l in range(1, NR):
        cs = [0] * T if l in covered else layer[l]
        L.append("    " + ", ".join(["0x%08xull" % (x & 0xFFFFFFFF) for x in cs] + ["0x%08xull" % (x >> 32) for x in cs]) + ", \\")
    L.append("    " + ", ".join(row) + " }")
    with open(path, "w") as f:
        f.write("\n".join(L) + "\n")


def build():
    tb = tables()
    bounds(tb)
    blocks, layer = constants(tb)
    n, worst = check(tb, blocks, layer)
    assert worst <= tb["dmax_main"]
    return tb, blocks, layer, n, worst


if __name__ == "__main__":
    tb, blocks, layer, n, worst = build()
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(here, "..", "near-light-client_amd", "csrc", "poseidon_blocks.inc")
    emit(tb, blocks, layer, out)
    print("K = %d, %d blocks; %d cases equal the naive permutation; plane sums <= %d (bound %d); wrote %s"
          % (K, N_BLOCKS, n, worst, tb["dmax_main"], os.path.normpath(out)))
