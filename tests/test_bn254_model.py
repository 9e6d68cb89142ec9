"""CPU: the pure-Python BN254 model (oracle/bn254_py.py) is an NTT - definition, inverse, Montgomery helpers."""
import random


def test_model_definition_and_inverse():
    import bn254_py as bn
    rng = random.Random(1)
    for log_n in (0, 1, 3, 6):
        n = 1 << log_n
        a = [rng.randrange(bn.R) for _ in range(n)]
        v = bn.ntt(a)
        w = bn.root_of_unity(log_n)
        for k in range(min(n, 5)):
            assert v[k] == sum(a[j] * pow(w, j * k, bn.R) for j in range(n)) % bn.R == bn.eval_poly(a, pow(w, k, bn.R))
        assert bn.ntt(v, inverse=True) == a
    assert bn.from_montgomery(bn.to_montgomery(12345)) == 12345 and bn.to_montgomery(1) == (1 << 256) % bn.R
    assert pow(bn.root_of_unity(28), 1 << 28, bn.R) == 1 and pow(bn.root_of_unity(28), 1 << 27, bn.R) != 1


def test_g1_model():
    """the G1 half of the model: group law, order, EIP-196's doubling of the generator, MSM = the sum of its terms"""
    import bn254_py as bn
    g = bn.G1
    assert (g[1] ** 2 - g[0] ** 3 - 3) % bn.Q == 0
    assert bn.g1_mul(bn.R, g) is None and bn.g1_mul(bn.R - 1, g) == bn.g1_neg(g) and bn.g1_add(g, bn.g1_neg(g)) is None
    assert bn.g1_mul(2, g) == bn.g1_add(g, g) and bn.g1_mul(7, g) == bn.g1_add(bn.g1_mul(3, g), bn.g1_mul(4, g))
    rng = random.Random(2)
    pts = [bn.g1_mul(rng.randrange(bn.R), g) for _ in range(5)]
    ks = [rng.randrange(bn.R) for _ in range(5)]
    for p in pts:
        assert (p[1] ** 2 - p[0] ** 3 - 3) % bn.Q == 0
    # linearity in the scalars, and against the discrete logs
    assert bn.msm_g1(ks, pts) == bn.g1_add(bn.msm_g1(ks[:2], pts[:2]), bn.msm_g1(ks[2:], pts[2:]))
    logs = [rng.randrange(bn.R) for _ in range(4)]
    assert bn.msm_g1(ks[:4], [bn.g1_mul(a, g) for a in logs]) == bn.g1_mul(sum(k * a for k, a in zip(ks, logs)) % bn.R, g)


def test_f29_field_code_built_for_the_host(tmp_path):
    """csrc/bn254_f29.hpp (the MSM kernels' arithmetic on nine 29-bit limbs), compiled with g++: every operation equals the
    big-integer computation and stays inside the bounds its header states - on random operands and on the extremes of the
    allowed ranges"""
    import os
    import subprocess
    import bn254_py as bn
    from conftest import ROOT
    exe = str(tmp_path / "f29check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-Wno-unknown-pragmas", "-fsanitize=undefined", "-fno-sanitize-recover=all",
                    "-I", os.path.join(ROOT, "near-light-client_amd", "csrc"), os.path.join(ROOT, "tests", "native", "bn254_f29_check.cpp"), "-o", exe],
                   check=True, capture_output=True)
    rp = 1 << 261
    rng = random.Random(29)
    cases = []
    for name, q in (("q", bn.Q), ("r", bn.R)):
        tight = lambda: rng.randrange(1 << 255)                              # noqa: E731
        mul_in = lambda: rng.choice([rng.randrange(int(2 ** 257.5)), int(2 ** 257.5) - 1, 0, 1, q, q - 1, 2 * q])   # noqa: E731,B023
        for _ in range(200):
            cases.append((name, "mul", mul_in(), mul_in()))
            cases.append((name, "add", rng.randrange(1 << 257), rng.randrange(1 << 257)))
            cases.append((name, "sub4", rng.randrange(1 << 257), rng.choice([tight(), (1 << 255) - 1, 0])))
            cases.append((name, "sub8", rng.randrange(1 << 257), rng.choice([rng.randrange(3 << 255), (3 << 255) - 1, 0])))
            cases.append((name, "tighten", rng.choice([rng.randrange(1 << 258), (1 << 258) - 1, 0, q, 21 * q, q - 1, 7 * q + 5]), 0))
            cases.append((name, "canonical", rng.choice([rng.randrange(1 << 258), (1 << 258) - 1, 0, q, 21 * q, q - 1, 7 * q + 5, 2 * q - 1]), 0))
            cases.append((name, "iszero", rng.choice([rng.randrange(1 << 258), 0, q, 5 * q, 21 * q, q + 1, 13 * q - 1]), 0))
            cases.append((name, "frommont", rng.choice([rng.randrange(q), 0, q - 1, (1 << 256) % q]), 0))
            cases.append((name, "words", rng.choice([rng.randrange(1 << 256), (1 << 256) - 1, 0]), 0))
            cases.append((name, "tocanon", rng.choice([tight(), 0, q, 2 * q, rp % q]), 0))
    text = "".join("%s %s %x %x\n" % c for c in cases)
    out = subprocess.run([exe], input=text, text=True, capture_output=True, check=True).stdout.split("\n")
    assert len(out) >= len(cases)
    for (name, op, a, b), line in zip(cases, out):
        q = bn.Q if name == "q" else bn.R
        val, normalised = line.split()
        got = int(val, 16)
        assert normalised == "1", (op, a, b)
        if op == "mul":
            assert got % q == a * b * pow(rp, -1, q) % q and got < (1 << 254) + q, (op, hex(a), hex(b))
        elif op == "add":
            assert got == a + b
        elif op == "sub4":
            assert got == a + 4 * q - b
        elif op == "sub8":
            assert got == a + 8 * q - b
        elif op == "tighten":
            assert got % q == a % q and got < 1.1 * q
        elif op == "canonical":
            assert got == a % q
        elif op == "iszero":
            assert got == (1 if a % q == 0 else 0)
        elif op == "frommont":
            assert got % q == a * 32 % q and got < (1 << 255)            # x 2^256 -> x 2^261
        elif op == "words":
            assert got == a
        elif op == "tocanon":
            assert got == a * pow(rp, -1, q) % q


def test_g1_sum_on_the_host(nlx):
    """nlx_bn254_g1_sum (host code of the MSM's tail, no GPU): sums of G1Affine words equal the model's, infinity included"""
    import bn254_py as bn
    rng = random.Random(3)
    pts = [bn.g1_mul(rng.randrange(1, bn.R), bn.G1) for _ in range(6)]
    for chosen in (pts, pts[:1], [], [pts[0], bn.g1_neg(pts[0])], [pts[1], pts[1], None, pts[2]]):
        want = None
        for p in chosen:
            want = bn.g1_add(want, p)
        assert nlx.bn254_g1_unpack(nlx.bn254_g1_sum(nlx.bn254_g1_pack(chosen))) == want


def test_g2_model_and_host_sum(nlx):
    """the G2 half of the model (EIP-197's generator: on the twist, of order r) and nlx_bn254_g2_sum (host code, no GPU)"""
    import bn254_py as bn
    g = bn.G2
    assert bn.g2_mul(bn.R, g) is None and bn.g2_mul(bn.R - 1, g) == bn.g2_neg(g) and bn.g2_add(g, bn.g2_neg(g)) is None
    assert bn.g2_mul(7, g) == bn.g2_add(bn.g2_mul(3, g), bn.g2_mul(4, g))
    rng = random.Random(4)
    pts = [bn.g2_mul(rng.randrange(1, bn.R), g) for _ in range(4)]
    for chosen in (pts, pts[:1], [], [pts[0], bn.g2_neg(pts[0])], [pts[1], pts[1], None, pts[2]]):
        want = None
        for p in chosen:
            want = bn.g2_add(want, p)
        assert nlx.bn254_g2_unpack(nlx.bn254_g2_sum(nlx.bn254_g2_pack(chosen))) == want


def test_plonk_quotient_model_and_kzg_division():
    """the big-integer model behind tests/test_gpu_bn254_plonk.py: a satisfying three-wire instance divides by Z_H (the quotient
    has degree < 3n), a broken gate or a broken copy does not; synthetic division satisfies p = q (X - zeta) + p(zeta)"""
    import random
    import bn254_py as bn
    rng = random.Random(21)
    k1, k2 = 5, 25
    alpha, beta, gamma = (rng.randrange(bn.R) for _ in range(3))
    for log_n in (2, 4):
        n = 1 << log_n
        p = bn.plonk_witness(log_n, rng, k1, k2, beta, gamma)
        t = bn.plonk_quotient(p, 5, k1, k2, alpha, beta, gamma)
        assert not any(t[3 * n:]) and any(t[:3 * n])
        bad = bn.plonk_witness(log_n, rng, k1, k2, beta, gamma, satisfied=False)
        assert any(bn.plonk_quotient(bad, 5, k1, k2, alpha, beta, gamma)[3 * n:])
        swapped = dict(p, s1=p["s2"], s2=p["s1"])             # a different permutation: z no longer matches it
        assert any(bn.plonk_quotient(swapped, 5, k1, k2, alpha, beta, gamma)[3 * n:])
    coeffs = [rng.randrange(bn.R) for _ in range(37)]
    zeta = rng.randrange(bn.R)
    y, q = bn.kzg_open(coeffs, zeta)
    assert y == bn.eval_poly(coeffs, zeta) and len(q) == 36
    back = [0] * 37
    for i, qi in enumerate(q):
        back[i + 1] = (back[i + 1] + qi) % bn.R
        back[i] = (back[i] - zeta * qi) % bn.R
    back[0] = (back[0] + y) % bn.R
    assert back == coeffs
