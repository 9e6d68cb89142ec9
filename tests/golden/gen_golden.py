#!/usr/bin/env python3
"""Generate tests/golden/primitives.json from an independent pure-Python (big-int) model.

The model restates the published plonky2 algorithms with no code shared with oracle/ or the
HIP library: Poseidon-12 (naive schedule), overwrite-mode sponge, hash_or_noop, two_to_one,
Merkle cap, LDE by direct Horner evaluation at g*w^i, Challenger.  The first two Poseidon
vectors are upstream's own known-answer tests (plonky2 poseidon_goldilocks.rs test_vectors,
quoted in SURVEY.md §8c); the third is SURVEY's [V] vector.

Run:  python tests/golden/gen_golden.py   (rewrites primitives.json; deterministic)
      NLX_GL_GENERATOR_SET=2021 python tests/golden/gen_golden.py   (primitives_gen2021.json: the other
      candidate generator pair of include/nlx_field.h; only the LDE / commit / field sections differ)
"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tools"))
from gen_poseidon_constants import round_constants  # constants generator (ChaCha8 recipe)

P = 0xFFFFFFFF00000001


def field_generators(gen_set):
    """the pair of include/nlx_field.h for the given set (the model shares only this definition with the C code)"""
    import re
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "..", "..", "include", "nlx_field.h")) as f:
        text = f.read()
    m = re.search(r"NLX_GL_GENERATOR_SET == %s\s*\n#define NLX_GL_MULTIPLICATIVE_GROUP_GENERATOR (\d+)ULL\s*\n"
                  r"#define NLX_GL_POWER_OF_TWO_GENERATOR (\d+)ULL" % gen_set, text)
    return int(m.group(1)), int(m.group(2))


GEN_SET = os.environ.get("NLX_GL_GENERATOR_SET", "7")
GEN, POW2_GEN = field_generators(GEN_SET)
RC = round_constants()
CIRC = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]
DIAG = [8] + [0] * 11


def permute(s):
    s = list(s)
    for r in range(30):
        s = [(x + RC[r * 12 + i]) % P for i, x in enumerate(s)]
        if r < 4 or r >= 26:
            s = [pow(x, 7, P) for x in s]
        else:
            s[0] = pow(s[0], 7, P)
        s = [(sum(s[(i + r_) % 12] * CIRC[i] for i in range(12)) + s[r_] * DIAG[r_]) % P for r_ in range(12)]
    return s


def hash_no_pad(xs):
    st = [0] * 12
    for off in range(0, len(xs), 8):
        chunk = xs[off:off + 8]
        st[:len(chunk)] = chunk
        st = permute(st)
    return st[:4]


def hash_or_noop(xs):
    if len(xs) <= 4:
        return list(xs) + [0] * (4 - len(xs))
    return hash_no_pad(xs)


def two_to_one(l, r):
    return permute(list(l) + list(r) + [0] * 4)[:4]


def merkle_cap(leaves, cap_height):
    lvl = [hash_or_noop(l) for l in leaves]
    while len(lvl) > (1 << cap_height):
        lvl = [two_to_one(lvl[2 * i], lvl[2 * i + 1]) for i in range(len(lvl) // 2)]
    return lvl


def merkle_levels(leaves, cap_height):
    lvls = [[hash_or_noop(l) for l in leaves]]
    while len(lvls[-1]) > (1 << cap_height):
        p = lvls[-1]
        lvls.append([two_to_one(p[2 * i], p[2 * i + 1]) for i in range(len(p) // 2)])
    return lvls


def root_of_unity(log_n):
    return pow(POW2_GEN, 1 << (32 - log_n), P)


def bitrev(x, bits):
    return int(format(x, "0%db" % bits)[::-1], 2) if bits else 0


def horner(coeffs, x):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % P
    return acc


def lde_leaves(coeff_cols, log_n, rate_bits):
    """rows of PolynomialBatch's Merkle leaves: leaf[l][c] = p_c(g * w_L^bitrev(l))."""
    log_l = log_n + rate_bits
    w = root_of_unity(log_l)
    rows = []
    for l in range(1 << log_l):
        x = GEN * pow(w, bitrev(l, log_l), P) % P
        rows.append([horner(c, x) for c in coeff_cols])
    return rows


class Challenger:
    def __init__(self):
        self.state = [0] * 12
        self.inp = []
        self.out = []

    def _duplex(self):
        self.state[:len(self.inp)] = self.inp
        self.inp = []
        self.state = permute(self.state)
        self.out = self.state[:8]

    def observe(self, x):
        self.out = []
        self.inp.append(x)
        if len(self.inp) == 8:
            self._duplex()

    def challenge(self):
        if self.inp or not self.out:
            self._duplex()
        return self.out.pop()


class Lcg:
    """deterministic test data (SplitMix64), values reduced into [0, p)."""

    def __init__(self, seed):
        self.s = seed & (2**64 - 1)

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & (2**64 - 1)
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
        return (z ^ (z >> 31)) % P


def main():
    g = {}
    neg1 = P - 1
    g["poseidon_kat"] = [
        {"in": [0] * 12, "out": permute([0] * 12), "source": "plonky2 poseidon_goldilocks.rs test_vectors #1"},
        {"in": list(range(12)), "out": permute(list(range(12))), "source": "plonky2 test_vectors #2"},
        {"in": [neg1] * 12, "out": permute([neg1] * 12), "source": "plonky2 test_vectors #3 (first 4 words in SURVEY.md)"},
    ]
    # upstream / SURVEY literals, asserted here so a wrong model can never be written out
    assert g["poseidon_kat"][0]["out"] == [
        0x3C18A9786CB0B359, 0xC4055E3364A246C3, 0x7953DB0AB48808F4, 0xC71603F33A1144CA,
        0xD7709673896996DC, 0x46A84E87642F44ED, 0xD032648251EE0B3C, 0x1C687363B207DF62,
        0xDF8565563E8045FE, 0x40F5B37FF4254DAE, 0xD070F637B431067C, 0x1792B1C4342109D7]
    assert g["poseidon_kat"][1]["out"] == [
        0xD64E1E3EFC5B8E9E, 0x53666633020AAA47, 0xD40285597C6A8825, 0x613A4F81E81231D2,
        0x414754BFEBD051F0, 0xCB1F8980294A023F, 0x6EB2A9E4D54A9D0F, 0x1902BC3AF467E056,
        0xF045D5EAFDC6021F, 0xE4150F77CAAA3BE5, 0xC9BFD01D39B50CCE, 0x5C0A27FCB0E1459B]
    assert g["poseidon_kat"][2]["out"][:4] == [
        0xBE0085CFC57A8357, 0xD95AF71847D05C09, 0xCF55A13D33C1C953, 0x95803A74F4530E82]
    rng = Lcg(0x6E6C78)
    rnd = [rng.next() for _ in range(12)]
    g["poseidon_kat"].append({"in": rnd, "out": permute(rnd), "source": "SplitMix64(0x6e6c78) mod p"})

    g["hash_or_noop"] = []
    for ln in (0, 1, 4, 5, 8, 9, 16, 17, 135):
        xs = [rng.next() for _ in range(ln)]
        g["hash_or_noop"].append({"in": xs, "out": hash_or_noop(xs)})
    l, r = [rng.next() for _ in range(4)], [rng.next() for _ in range(4)]
    g["two_to_one"] = {"l": l, "r": r, "out": two_to_one(l, r)}

    leaves = [[rng.next() for _ in range(7)] for _ in range(64)]
    g["merkle"] = {"leaves": leaves, "cap_height": 2, "levels": merkle_levels(leaves, 2)}
    leaves3 = [[rng.next() for _ in range(3)] for _ in range(16)]  # noop leaves (<= 4 elements)
    g["merkle_noop"] = {"leaves": leaves3, "cap_height": 4, "levels": merkle_levels(leaves3, 4)}

    # LDE / commit: 3 polynomials of degree < 8, rate 8, cap height 2
    coeffs = [[rng.next() for _ in range(8)] for _ in range(3)]
    w8 = root_of_unity(3)
    values = [[horner(c, pow(w8, i, P)) for i in range(8)] for c in coeffs]
    lv = lde_leaves(coeffs, 3, 3)
    g["commit"] = {"log_n": 3, "rate_bits": 3, "cap_height": 2, "coeffs": coeffs, "values": values,
                   "leaves": lv, "cap": merkle_cap(lv, 2)}
    # wide batch: 9 polynomials (two absorb chunks), log_n = 4, rate 2 (STARK shape), cap 1
    coeffs9 = [[rng.next() for _ in range(16)] for _ in range(9)]
    w16 = root_of_unity(4)
    values9 = [[horner(c, pow(w16, i, P)) for i in range(16)] for c in coeffs9]
    lv9 = lde_leaves(coeffs9, 4, 1)
    g["commit_wide"] = {"log_n": 4, "rate_bits": 1, "cap_height": 1, "coeffs": coeffs9, "values": values9,
                        "leaves": lv9, "cap": merkle_cap(lv9, 1)}

    ch = Challenger()
    obs = [rng.next() for _ in range(11)]
    outs = []
    for x in obs[:3]:
        ch.observe(x)
    outs.append(ch.challenge())
    outs.append(ch.challenge())
    for x in obs[3:]:
        ch.observe(x)
    outs += [ch.challenge() for _ in range(10)]
    g["challenger"] = {"observe_then_2": obs[:3], "observe_then_10": obs[3:], "challenges": outs}

    g["field"] = {"p": P, "generator": GEN, "pow2_generator": POW2_GEN,
                  "gen_order_check": pow(GEN, (P - 1) >> 32, P), "root_2_8": root_of_unity(8)}
    assert g["field"]["gen_order_check"] == POW2_GEN
    assert all(pow(GEN, (P - 1) // q, P) != 1 for q in (2, 3, 5, 17, 257, 65537)), "not a generator of F_p^*"
    g["field"]["generator_set"] = GEN_SET
    name = "primitives.json" if GEN_SET == "7" else "primitives_gen%s.json" % GEN_SET
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), name)
    with open(path, "w") as f:
        json.dump(g, f, indent=0, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
