/* nlx_field.h - the ONE definition of the Goldilocks generator pair.
 *
 * Everything that depends on a choice of generators derives from these two numbers and from
 * nothing else: the NTT root tables (primitive_root_of_unity(k) = POWER_OF_TWO_GENERATOR^(2^(32-k))),
 * the LDE / FRI coset shift (= MULTIPLICATIVE_GROUP_GENERATOR, plonky2 `F::coset_shift()`),
 * the permutation argument's k_i (`get_unique_coset_shifts`: powers of the generator) and the
 * verifier's subgroup_x.  Included by the product (near-light-client_amd/csrc/gl.hpp), by the test
 * oracle (oracle/gl.h) and parsed by the Python golden-vector model (tests/golden/gen_golden.py).
 *
 * Which pair does plonky2_field @ d2598bd (/root/reference/Cargo.lock:4912-4914) use?  Its source is
 * not vendored, so the pair cannot be read from /root/reference.  Evidence:
 *   set 7     g = 7, g_{2^32} = 7^((p-1)/2^32) = 1753635133440165772.  7 is the smallest primitive
 *             root of p ("Sage: GF(p).multiplicative_generator()", the comment upstream carries), and
 *             upstream's quadratic-extension constant EXT_POWER_OF_TWO_GENERATOR = [0, 15659105665374529263]
 *             (order 2^33) squares to 7 * 15659105665374529263^2 = 1753635133440165772 - i.e. to THIS
 *             set's g_{2^32}, not to the other one's; upstream's EXT_MULTIPLICATIVE_GROUP_GENERATOR
 *             [18081566051660590251, 16121475356294670766] generates F_{p^2}^* under X^2 = 7 and its (p^2-1)/2^33-th power
 *             is exactly that [0, 15659105665374529263] (tests/test_field_params.py checks the arithmetic).
 *   set 2021  g = 14293326489335486720, g_{2^32} = 7277203076849721926: self-consistent, carried by
 *             early-2021 plonky2, and the pair round 1 of this repository was built with.
 * Default: set 7.  Build the other one with -DNLX_GL_GENERATOR_SET=2021 (build.py and oracle_py.py do
 * that when the environment variable NLX_GL_GENERATOR_SET=2021 is set; the libraries are then named
 * libnlx_gen2021.so / liboracle_gen2021.so so both builds can sit side by side).
 */
#ifndef NLX_FIELD_H
#define NLX_FIELD_H

#ifndef NLX_GL_GENERATOR_SET
#define NLX_GL_GENERATOR_SET 7
#endif

#if NLX_GL_GENERATOR_SET == 7
#define NLX_GL_MULTIPLICATIVE_GROUP_GENERATOR 7ULL
#define NLX_GL_POWER_OF_TWO_GENERATOR 1753635133440165772ULL
#elif NLX_GL_GENERATOR_SET == 2021
#define NLX_GL_MULTIPLICATIVE_GROUP_GENERATOR 14293326489335486720ULL
#define NLX_GL_POWER_OF_TWO_GENERATOR 7277203076849721926ULL
#else
#error "NLX_GL_GENERATOR_SET must be 7 or 2021"
#endif

#endif
