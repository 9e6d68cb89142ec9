"""The Goldilocks generator pair is ONE definition (include/nlx_field.h) shared by product, oracle and golden model.
These CPU tests check the definition's arithmetic, the evidence DESIGN.md §2 cites for the default, and that every
consumer really derives from it.  Run the whole suite under the other candidate with NLX_GL_GENERATOR_SET=2021."""
import ctypes

from conftest import GEN, GEN_SET, POW2_GEN, P, field_generators

PRIME_FACTORS_OF_P_MINUS_1 = (2, 3, 5, 17, 257, 65537)  # p - 1 = 2^32 * 3 * 5 * 17 * 257 * 65537


def test_both_sets_are_self_consistent():
    assert (P - 1) == (1 << 32) * 3 * 5 * 17 * 257 * 65537
    for gen_set in ("7", "2021"):
        _, g, w = field_generators(gen_set)
        assert all(pow(g, (P - 1) // q, P) != 1 for q in PRIME_FACTORS_OF_P_MINUS_1), "not a generator of F_p^*"
        assert pow(g, (P - 1) >> 32, P) == w, "POWER_OF_TWO_GENERATOR must be g^((p-1)/2^32)"
        assert pow(w, 1 << 31, P) == P - 1, "order exactly 2^32"


def test_default_is_the_smallest_primitive_root():
    """upstream's comment on the constant is `Sage: GF(p).multiplicative_generator()`, which returns the smallest one"""
    assert field_generators("7")[1] == 7
    for g in range(2, 7):
        assert any(pow(g, (P - 1) // q, P) == 1 for q in PRIME_FACTORS_OF_P_MINUS_1), "%d is a primitive root" % g


def test_default_matches_upstream_extension_generator():
    """plonky2_field's quadratic extension carries EXT_POWER_OF_TWO_GENERATOR = [0, 15659105665374529263] (order 2^33,
    X^2 = 7): its square 7 * b^2 is an element of order 2^32 of the base field and upstream defines the base field's
    POWER_OF_TWO_GENERATOR as exactly that square.  It equals set 7's value and not set 2021's."""
    b = 15659105665374529263
    sq = 7 * b * b % P
    assert sq == field_generators("7")[2]
    assert sq != field_generators("2021")[2]
    assert pow(sq, 1 << 31, P) == P - 1


def test_upstream_extension_generators_fit_set_7_only():
    """Two more constants of plonky2_field's quadratic extension, recalled independently of each other:
    EXT_MULTIPLICATIVE_GROUP_GENERATOR = [18081566051660590251, 16121475356294670766] and
    EXT_POWER_OF_TWO_GENERATOR = [0, 15659105665374529263].  Under X^2 = 7 the first one generates all of F_{p^2}^*
    (order p^2 - 1 = 2^33 * 3 * 5 * 7 * 17 * 179 * 257 * 65537 * 7361031152998637: no proper divisor kills it) and its
    (p^2 - 1) / 2^33-th power is EXACTLY the second one - so the three recollections (W = 7 and the two generators) agree
    with each other, and the second one's square is set 7's base-field POWER_OF_TWO_GENERATOR (previous test)."""
    w = 7

    def mul(a, b):
        return ((a[0] * b[0] + w * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)

    def power(a, e):
        r = (1, 0)
        while e:
            if e & 1:
                r = mul(r, a)
            a = mul(a, a)
            e >>= 1
        return r
    factors = (2, 3, 5, 7, 17, 179, 257, 65537, 7361031152998637)
    order = P * P - 1
    rest = order
    for q in factors:
        while rest % q == 0:
            rest //= q
    assert rest == 1                                            # the factorisation of p^2 - 1 is complete
    g = (18081566051660590251, 16121475356294670766)
    assert power(g, order) == (1, 0) and all(power(g, order // q) != (1, 0) for q in factors)
    assert power(g, order >> 33) == (0, 15659105665374529263)
    assert pow(7, (P - 1) // 2, P) == P - 1                     # 7 is a non-residue: X^2 - 7 is irreducible


def test_oracle_and_golden_use_the_header_pair(orc, golden):
    assert orc.generators() == (GEN, POW2_GEN)
    assert golden["field"]["generator"] == GEN and golden["field"]["pow2_generator"] == POW2_GEN
    assert golden["field"].get("generator_set", "7") == GEN_SET
    assert golden["field"]["root_2_8"] == pow(POW2_GEN, 1 << 24, P)


def test_library_exports_the_header_pair(nlx):
    out = (ctypes.c_uint64 * 2)()
    nlx.lib.dll.nlx_field_generators(out)  # no context needed: a caller checks this before creating one
    assert (int(out[0]), int(out[1])) == (GEN, POW2_GEN)


def test_synthetic_circuit_k_is_are_powers_of_the_generator(nlx):
    """plonk::permutation_argument / get_unique_coset_shifts: k_i = g^i"""
    syn = nlx.SyntheticCircuit(6, seed=3)
    assert [int(x) for x in syn.k_is[:4]] == [pow(GEN, i, P) for i in range(4)]
