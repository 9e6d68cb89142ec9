/* nlx.h - C ABI of the MI355X-native prover backend for the nearx plonky2x circuits.
 *
 * This is the drop-in boundary of SURVEY.md §8(b): the reference has no FFI seam for its
 * prover, so the seam is cut one layer below nearx (nearx/src/test_utils.rs:29,62,66 call
 * CircuitBuilder::build / CircuitBuild::prove / verify; those reach the plonky2 functions
 * named on each entry point below, crates pinned at /root/reference/Cargo.lock:4864-4866,
 * 4912-4974, 4977-4979, 6515-6517).  INTEGRATION.md shows the Rust `extern "C"` stubs a
 * maintainer adds to the patched plonky2 / starkyx / plonky2x crates.
 *
 * Conventions
 *  - Field elements are canonical (< p = 2^64 - 2^32 + 1) little-endian u64.  Inputs that are
 *    not canonical are rejected only where stated; outputs are always canonical.
 *  - Matrices of polynomials are column-major: cols[c * n + i].
 *  - Every data pointer may be a HOST pointer or a HIP DEVICE pointer (e.g. a torch tensor's
 *    data_ptr()); the library inspects it with hipPointerGetAttributes.  Host buffers are
 *    copied over PCIe inside the call; device buffers are used in place.  Output buffers are
 *    caller-owned.  Handles (nlx_ctx, nlx_commit, nlx_circuit) are library-owned and freed
 *    only by their *_destroy call.
 *  - Every call returns 0 on success or a negative NLX_E_* code and never throws or aborts;
 *    nlx_last_error(ctx) describes the most recent failure on that context.
 *  - A context is bound to one HIP device and is NOT thread-safe (one host thread per
 *    context); different contexts are independent.  Calls are synchronous on return.
 *  - Results are deterministic: identical inputs give identical outputs on any device count
 *    or occupancy; the proof-of-work grind returns the SMALLEST valid nonce.
 *  - There is no CPU fallback: if no gfx950 device is usable nlx_ctx_create fails.
 */
#ifndef NLX_H
#define NLX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NLX_OK 0
#define NLX_E_INVAL (-1)
#define NLX_E_NOMEM (-2)
#define NLX_E_HIP (-3)
#define NLX_E_RANGE (-4)
#define NLX_E_UNSUPPORTED (-5)

typedef struct nlx_ctx nlx_ctx;
typedef struct nlx_commit nlx_commit;

/* library / ABI version (major << 16 | minor) */
uint32_t nlx_version(void);
const char* nlx_strerror(int32_t code);
/* The Goldilocks generator pair this library was built with (include/nlx_field.h):
 * out[0] = MULTIPLICATIVE_GROUP_GENERATOR (coset shift, k_i base), out[1] = POWER_OF_TWO_GENERATOR (order 2^32).
 * plonky2_field::goldilocks_field::GoldilocksField::{MULTIPLICATIVE_GROUP_GENERATOR, POWER_OF_TWO_GENERATOR}
 * (crate pinned at /root/reference/Cargo.lock:4912-4914); a caller checks it against its own constants once. */
void nlx_field_generators(uint64_t out[2]);
/* Self-test of the "never throws" rule on the caller's platform: raises a C++ exception INSIDE the library - kind 0: a
 * real failed host allocation (a std::vector larger than the address space), 1: std::runtime_error, 2: a non-standard
 * object - and returns what the entry points' guard makes of it: NLX_E_NOMEM for kind 0, NLX_E_INVAL otherwise
 * (NLX_E_RANGE for an unknown kind).  Kinds 3 / 4 / 5 arm / disarm a fault in nlx_batch_prove's start-up (3: starting the
 * second worker thread fails, 4: the first, 5: off; they return NLX_OK): the batch must still return a code, with the
 * workers that did start joined and every job proved (tests/test_gpu_mapreduce.py).  Needs no context and no GPU.  Every extern "C" definition of the library is a
 * function-try-block with this guard (csrc/ctx.hpp NLX_TRY / NLX_CATCH), so a std::bad_alloc in the host-side
 * orchestration reaches a Rust / Go caller as a return code (crates/protocol/src/prelude.rs:1 maps
 * such codes to anyhow::Error at the caller), never as an unwind through foreign frames. */
int32_t nlx_abi_selftest(int32_t kind);

/* ---- context ---- */
int32_t nlx_ctx_create(int device, nlx_ctx** out);
void nlx_ctx_destroy(nlx_ctx* ctx);
const char* nlx_last_error(const nlx_ctx* ctx);
/* Use an existing HIP stream (e.g. torch's current stream) for all work of this context;
 * NULL restores the context's own stream.  The caller keeps ownership of the stream. */
int32_t nlx_ctx_set_stream(nlx_ctx* ctx, void* hip_stream);
int32_t nlx_ctx_synchronize(nlx_ctx* ctx);
/* Scheduling priority of the context's own stream: high = 1 puts its kernels ahead of other streams' when workgroup slots
 * free up.  For small latency-bound proofs (the curta STARKs of a Sync step) sharing a GPU with a large throughput-bound one
 * (the outer proof): the short kernels no longer queue behind the long ones.  Call it before queuing work. */
int32_t nlx_ctx_set_priority(nlx_ctx* ctx, int high);
/* Restrict the context's own stream to a subset of the device's compute units (bit i of the mask = CU i; MI355X has 256).
 * Two contexts with disjoint masks partition the chip: a chain of short latency-bound kernels (a small STARK) then runs
 * beside a long throughput-bound proof instead of queueing behind its millisecond-long workgroups.  Call it before queuing
 * work; a later nlx_ctx_set_priority undoes it. */
int32_t nlx_ctx_set_cu_mask(nlx_ctx* ctx, const uint32_t* mask, uint32_t n_words);
/* Device buffers for callers without HIP bindings of their own (SURVEY.md §8b `nlx_buf`): every entry point that
 * takes a "host or device" pointer accepts nlx_buf_device_ptr(buf) (+ an offset).  A buffer belongs to its context
 * and must be destroyed before it; upload / download are synchronous. */
typedef struct nlx_buf nlx_buf;
int32_t nlx_buf_create(nlx_ctx* ctx, size_t bytes, nlx_buf** out);
void nlx_buf_destroy(nlx_buf* buf);
void* nlx_buf_device_ptr(const nlx_buf* buf);
size_t nlx_buf_size(const nlx_buf* buf);
int32_t nlx_buf_upload(nlx_buf* buf, size_t offset, const void* src, size_t bytes);
int32_t nlx_buf_download(nlx_buf* buf, size_t offset, void* dst, size_t bytes);
/* The context keeps freed device blocks for reuse (hipMalloc / hipFree synchronise the device).
 * nlx_ctx_memory reports the bytes it holds from the driver and the part currently in use by live handles and
 * tables; nlx_ctx_trim returns the unused part to the driver (it synchronises the stream first). */
int32_t nlx_ctx_memory(const nlx_ctx* ctx, size_t* reserved_bytes, size_t* in_use_bytes);
int32_t nlx_ctx_trim(nlx_ctx* ctx);
/* Per-kernel device timing for measurement (bench.py's roofline): when enabled, the library brackets
 * its main kernels ("intt", "lde", "hash_lde_leaves", "merkle_levels", "quotient", "air_quotient", "fri_combine"; nlx_ntt_batch:
 * "ntt_transform", "ntt_reorder")
 * with HIP events on the context's stream.  nlx_ctx_kernel_timing(ctx, x) also clears the samples. */
int32_t nlx_ctx_kernel_timing(nlx_ctx* ctx, int enable);
int32_t nlx_ctx_kernel_stats(nlx_ctx* ctx, const char* name, uint64_t* calls, double* total_ms, double* alg_bytes);
/* work units of the same samples that are not bytes: Poseidon permutations for "hash_lde_leaves" and "merkle_levels"
 * (the kernels whose bound is the integer-VALU issue rate, not HBM); 0 for the others */
int32_t nlx_ctx_kernel_units(nlx_ctx* ctx, const char* name, double* units);

/* ---- a1: GoldilocksField arithmetic, element-wise (device self-test of the field core) ----
 * a, b: n arbitrary u64 (values >= p are reduced first, except for row 4).  out: 5 x n:
 * row 0 a*b, row 1 a+b, row 2 a-b, row 3 a^-1 (0 if a = 0), row 4 the raw multiply path applied to
 * the UNREDUCED inputs (carry/borrow edges), all canonical. */
int32_t nlx_field_ops(nlx_ctx* ctx, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out);

/* ---- a5: plonky2::hash::poseidon::Poseidon::poseidon ----
 * states: n x 12 u64, row-major, permuted in place. */
int32_t nlx_poseidon_permute_batch(nlx_ctx* ctx, uint64_t* states, size_t n);

/* ---- a5/a6: PoseidonHash::hash_or_noop over the rows of a row-major matrix ----
 * (the per-leaf step of MerkleTree::new).  digests_out: n_rows x 4. */
int32_t nlx_hash_rows(nlx_ctx* ctx, const uint64_t* rows, size_t n_rows, size_t row_len, uint64_t* digests_out);

/* ---- a6: plonky2::hash::merkle_tree::MerkleTree::new(leaves, cap_height) ----
 * leaves: row-major n_leaves x leaf_len, n_leaves a power of two >= 2^cap_height.
 * digests_out (optional, may be NULL): level-major digests, level 0 = leaf digests
 * (n_leaves x 4), then n_leaves/2, ... down to and including the cap level;
 * nlx_merkle_digest_words() gives its length.  cap_out: 2^cap_height x 4. */
size_t nlx_merkle_digest_words(size_t n_leaves, uint32_t cap_height);
int32_t nlx_merkle_build(nlx_ctx* ctx, const uint64_t* leaves, size_t n_leaves, size_t leaf_len,
                         uint32_t cap_height, uint64_t* digests_out, uint64_t* cap_out);

/* ---- a2: plonky2_field::fft::{fft, ifft}, PolynomialCoeffs::coset_fft, PolynomialValues::coset_ifft ----
 * cols: n_cols x 2^log_n, column-major, transformed in place, natural order in and out.
 * inverse = 0: coefficients -> values on shift*<w>;  inverse = 1: values on shift*<w> -> coefficients.
 * coset_shift = 1 (or 0) means the plain subgroup. */
int32_t nlx_ntt_batch(nlx_ctx* ctx, uint64_t* cols, size_t n_cols, uint32_t log_n, int inverse,
                      uint64_t coset_shift);

/* cfg5 / SURVEY 8e: ONE forward transform of 2^log_n points per column split over G = 2^world_log GPUs.  Rank r holds the
 * contiguous slice [r m, (r + 1) m), m = 2^log_n / G, of every column (n_cols x m, column-major, device memory).  The first
 * world_log levels of a decimation in frequency pair element j with j + n / 2^(level + 1) - on another rank: for level =
 * 0 .. world_log - 1 the caller exchanges slices with rank r XOR (G >> (level + 1)) (RCCL send / recv) and this call computes
 * the rank's half of the level in place on `mine` from the partner's slice `theirs`.  After the last cross level every slice
 * is an independent transform of m points: nlx_ntt_batch(mine, n_cols, log_n - world_log, 0, 1) leaves rank r holding
 * X[k] for k = bitrev_G(r) (mod G) at local index k div G.  (near-light-client_amd/split_ntt.py drives it.) */
int32_t nlx_ntt_split_level(nlx_ctx* ctx, uint64_t* mine, const uint64_t* theirs, size_t n_cols, uint32_t log_n, uint32_t world_log,
                            uint32_t rank, uint32_t level);

/* ---- f.4 (first piece): the BN254 scalar-field NTT of the recursive wrap (gnark-crypto ecc/bn254/fr/fft Domain.FFT /
 * FFTInverse; the wrap itself is not in /root/reference - succinct.json:7-8 names the entry point that can run it).
 * cols: n_cols x 2^log_n elements, column-major, transformed in place, natural order in and out; an element is 32 bytes =
 * four little-endian 64-bit words.  flags = NLX_BN254_MONTGOMERY: elements are in Montgomery form (R = 2^256), i.e.
 * gnark-crypto's fr.Element exactly as it lies in memory; flags = 0: canonical integers < r.  log_n <= 28 (Fr's 2-adicity);
 * the root of unity is gnark-crypto's (5^((r-1)/2^28)).  inverse = 1 also multiplies by 1/n. */
#define NLX_BN254_MONTGOMERY 1u
int32_t nlx_bn254_ntt_batch(nlx_ctx* ctx, uint64_t* cols, size_t n_cols, uint32_t log_n, int inverse, uint32_t flags);
/* The same with gnark-crypto's two FFT options.  coset_shift != NULL (four words in the form of the data): the transform on
 * the coset shift * <w> - fft.OnCoset() with the domain's FrMultiplicativeGen: forward evaluates on shift w^k (coefficient j
 * is multiplied by shift^j first), inverse interpolates from there (coefficient j is multiplied by shift^-j last).
 * NLX_BN254_BITREV_OUT: natural order in, bit-reversed order out - what fft.DIF leaves; NLX_BN254_BITREV_IN: bit-reversed
 * order in, natural order out - fft.DIT (decimation in time).  Neither has a reordering pass: a prover that alternates
 * them (FFTInverse(DIF) -> pointwise work -> FFT(DIT, OnCoset)), as gnark's does, never reorders.  Not both at once.
 * coset_shift is read on the HOST (a device pointer is refused); with flags = 0 it must be a canonical integer below r. */
#define NLX_BN254_BITREV_OUT 2u
#define NLX_BN254_BITREV_IN 4u
int32_t nlx_bn254_ntt_batch_coset(nlx_ctx* ctx, uint64_t* cols, size_t n_cols, uint32_t log_n, int inverse, uint32_t flags,
                                  const uint64_t* coset_shift);
/* ---- f.4 (second piece): the G1 multi-scalar multiplication of the wrap's KZG commitments (gnark-crypto ecc/bn254
 * G1Affine.MultiExp(points, scalars, config)).  points: n x 8 little-endian 64-bit words = gnark-crypto's G1Affine as it lies in
 * memory (X then Y, each an fp.Element in Montgomery form, R = 2^256; the point at infinity is (0, 0)); scalars: n x 4 words,
 * fr.Element in Montgomery form with flags = NLX_BN254_MONTGOMERY, canonical integers < r with flags = 0.  out: sum_i
 * scalars[i] * points[i] as a G1Affine (8 words, Montgomery; (0, 0) for the point at infinity).  points / scalars may be host
 * or device pointers; n <= 2^27.  Points are taken to be on the curve (as MultiExp does). */
int32_t nlx_bn254_msm_g1(nlx_ctx* ctx, const uint64_t* points, const uint64_t* scalars, uint64_t n, uint32_t flags, uint64_t out[8]);
/* The same over G2 (Groth16's B query): points n x 16 words = gnark-crypto's G2Affine (X then Y, each an E2{A0, A1} of two
 * fp.Element, Montgomery; infinity = all zero), the coordinates in Fq2 = Fq[u] / (u^2 + 1); out: a G2Affine, 16 words. */
int32_t nlx_bn254_msm_g2(nlx_ctx* ctx, const uint64_t* points, const uint64_t* scalars, uint64_t n, uint32_t flags, uint64_t out[16]);
/* The sum of n G1Affine points (same layout; host pointers, computed on the host): joins the partial results of an MSM whose
 * points were split over several GPUs - one addition per rank. */
int32_t nlx_bn254_g1_sum(const uint64_t* points, uint64_t n, uint64_t out[8]);
int32_t nlx_bn254_g2_sum(const uint64_t* points, uint64_t n, uint64_t out[16]);   /* the same for G2Affine points */
/* Test / bench data on the device: out[i] = (i + 1) * base for i < n as G1Affine words - n distinct curve points (an SRS's
 * worth of gather targets for the MSM; a big-integer model makes a few thousand per second).  out: host or device, n x 8
 * words; n <= 2^27; base must not be the point at infinity. */
int32_t nlx_bn254_g1_multiples(nlx_ctx* ctx, const uint64_t base[8], uint64_t n, uint64_t* out);

/* ---- f.4 (third piece): the PLONK prover's quotient chain and a KZG opening on those kernels.  What the wrap's prover
 * (gnark backend/plonk/bn254 Prove; Go, not in /root/reference - succinct.json:7-8 names the entry point that runs it) does
 * between its commitments, restated from the published protocol (three-wire PLONK as gnark arithmetises it), not from
 * gnark's source: no blinding terms, no gnark byte format.  Every element is an fr.Element as it lies in memory (four words,
 * Montgomery): flags must be NLX_BN254_MONTGOMERY.
 *
 * nlx_bn254_plonk_quotient: the thirteen polynomials given by their values on H = <w_n> (host or device pointers, n x 4 words
 * each, natural order; pi may be NULL) -> coefficients (FFTInverse, DIF) -> values on the coset coset_shift * <w_4n> (FFT,
 * DIT, OnCoset) -> there, pointwise,
 *     t = [ ql l + qr r + qm l r + qo o + qk + pi
 *           + alpha ( (l + beta x + gamma)(r + beta k1 x + gamma)(o + beta k2 x + gamma) z
 *                     - (l + beta s1 + gamma)(r + beta s2 + gamma)(o + beta s3 + gamma) z(w_n x) )
 *           + alpha^2 L1(x) (z - 1) ] / (x^n - 1),           L1(x) = (x^n - 1) / (n (x - 1))
 * -> coefficients (FFTInverse, OnCoset).  t_out: the three chunks t_lo, t_mid, t_hi of n coefficients each (3 n x 4 words,
 * natural order, host or device).  *high_chunk_is_zero (may be NULL): 1 if coefficients 3n .. 4n-1 all vanish - they do
 * exactly when the witness satisfies gates and copy constraints, and a prover must not commit to t otherwise.
 * coset_shift, k1, k2, alpha, beta, gamma: host, four words each, in the form of the data.  2 <= log_n <= 26. */
typedef struct {
    uint32_t log_n;
    uint32_t flags;                       /* NLX_BN254_MONTGOMERY */
    const uint64_t *ql, *qr, *qm, *qo, *qk;   /* selectors */
    const uint64_t *s1, *s2, *s3;             /* the permutation, as polynomials: s_j(w^i) = the point of H, k1 H or k2 H that wire j of gate i maps to */
    const uint64_t *l, *r, *o;                /* wires */
    const uint64_t *z;                        /* the permutation's grand product, z(w^0) = 1 */
    const uint64_t *pi;                       /* public-input polynomial (values on H) or NULL */
    const uint64_t *coset_shift, *k1, *k2;    /* gnark: the domain's FrMultiplicativeGen u, then k1 = u, k2 = u^2 */
    const uint64_t *alpha, *beta, *gamma;
    /* Blinding (flags |= NLX_BN254_PLONK_BLINDED; gnark: getBlindedPolynomial, order 1 for l, r, o and order 2 for z): host, nine
     * elements of four words in the form of the data - b[0], b[1] for l, b[2], b[3] for r, b[4], b[5] for o, b[6], b[7], b[8] for
     * z: the polynomial that enters the quotient is l(X) + (b[0] + b[1] X)(X^n - 1) ... z(X) + (b[6] + b[7] X + b[8] X^2)(X^n - 1)
     * (same values on H; the caller commits to and opens the blinded polynomials).  The quotient then has 3 n + 6 coefficients:
     * t_out receives ALL 4 n of them (4 n x 4 words) and *high_chunk_is_zero tells whether coefficients 3 n + 6 .. 4 n - 1 vanish;
     * gnark's h1, h2, h3 are t[0 : n + 2], t[n + 2 : 2 n + 4], t[2 n + 4 : 3 n + 6].  log_n >= 3.  NULL / flag clear: as before. */
    const uint64_t *blinding;
} nlx_bn254_plonk_quotient_args;
#define NLX_BN254_PLONK_BLINDED 0x100u
int32_t nlx_bn254_plonk_quotient(nlx_ctx* ctx, const nlx_bn254_plonk_quotient_args* args, uint64_t* t_out, int32_t* high_chunk_is_zero);
/* The permutation's grand product (gnark computeZ / the paper's round 2): z(w^0) = 1,
 * z(w^(i+1)) = z(w^i) prod_j (w_j(i) + beta id_j(i) + gamma) / (w_j(i) + beta s_j(i) + gamma), id = (w^i, k1 w^i, k2 w^i).
 * One inversion per 64 rows (batch inversion), the running product as a multi-level scan.  All inputs n x 4 words on H in
 * natural order, host or device; beta, gamma, k1, k2 host; z_out: n x 4 words, host or device.  *closes (may be NULL): 1 if
 * the product returns to 1 after the last row - it does exactly when the wires respect the permutation. */
int32_t nlx_bn254_plonk_grand_product(nlx_ctx* ctx, uint32_t log_n, const uint64_t* l, const uint64_t* r, const uint64_t* o,
                                      const uint64_t* s1, const uint64_t* s2, const uint64_t* s3, const uint64_t beta[4],
                                      const uint64_t gamma[4], const uint64_t k1[4], const uint64_t k2[4], uint64_t* z_out,
                                      int32_t* closes);
/* out[i] = sum_t scalars[t] * polys[t][i], i < m: the linearisation polynomial and the batched opening polynomial of the last
 * round.  polys: host array of n_terms (<= 16) pointers, each to m x 4 words (host or device); scalars: host, n_terms x 4 words;
 * out: m x 4 words, host or device (may alias one of the inputs). */
int32_t nlx_bn254_fr_lincomb(nlx_ctx* ctx, uint64_t m, uint32_t n_terms, const uint64_t* const* polys, const uint64_t* scalars,
                             uint64_t* out);
/* Groth16's quotient (gnark backend/groth16/bn254 computeH): a, b, c = the values of A w, B w, C w on H (n x 4 words each,
 * natural order, host or device) -> h = (a b - c) / (x^n - 1) as n coefficients (the top one is zero): three FFTInverse(DIF),
 * three FFT(DIT, OnCoset) on the coset of the SAME size, one pointwise pass (the divisor is one constant there), one
 * FFTInverse(OnCoset).  The proof's points are then nlx_bn254_msm_g1 / _g2 over the proving key's queries. */
int32_t nlx_bn254_groth16_quotient(nlx_ctx* ctx, uint32_t log_n, const uint64_t* a, const uint64_t* b, const uint64_t* c,
                                   const uint64_t coset_shift[4], uint64_t* h_out);
/* One KZG opening (gnark-crypto kzg.Open): coeffs = m coefficients (natural order, host or device), zeta = the point (host).
 * y_out = p(zeta); quotient_out (may be NULL; host or device, m - 1 coefficients) = (p(X) - p(zeta)) / (X - zeta) - the
 * running Horner values, computed as a parallel scan; proof_out (may be NULL) = its commitment sum_i q_i srs[i] over the first
 * m - 1 points of the SRS (G1Affine words as for nlx_bn254_msm_g1; host or device).  2 <= m <= 2^28. */
int32_t nlx_bn254_kzg_open(nlx_ctx* ctx, const uint64_t* coeffs, uint64_t m, const uint64_t zeta[4], const uint64_t* srs,
                           uint64_t y_out[4], uint64_t* quotient_out, uint64_t proof_out[8]);

/* ---- a3: plonky2::fri::oracle::PolynomialBatch::{from_values, from_coeffs} ----
 * values / coeffs: n_cols x 2^log_n column-major, natural order.  The coset shift is the
 * field's multiplicative generator (plonky2 F::coset_shift()).  blinding / salting is not
 * supported (zero_knowledge = false in the standard recursion config).
 * cap_out: 2^cap_height x 4 words.  *out receives a handle that keeps coefficients, the LDE
 * table and all Merkle digests resident in HBM. */
int32_t nlx_commit_from_values(nlx_ctx* ctx, const uint64_t* values, size_t n_cols, uint32_t log_n,
                               uint32_t rate_bits, uint32_t cap_height, uint64_t* cap_out, nlx_commit** out);
int32_t nlx_commit_from_coeffs(nlx_ctx* ctx, const uint64_t* coeffs, size_t n_cols, uint32_t log_n,
                               uint32_t rate_bits, uint32_t cap_height, uint64_t* cap_out, nlx_commit** out);
void nlx_commit_destroy(nlx_commit* c);

/* PolynomialBatch.polynomials: coefficients, natural order, n_cols x n column-major. */
int32_t nlx_commit_get_coeffs(nlx_commit* c, uint64_t* coeffs_out);
/* MerkleTree.cap */
int32_t nlx_commit_get_cap(nlx_commit* c, uint64_t* cap_out);
/* MerkleTree.leaves[idx[j]] (row of the bit-reversed LDE table, n_cols words) and
 * MerkleTree::prove(idx[j]) (log2(n << rate_bits) - cap_height sibling digests, bottom-up).
 * rows_out: k x n_cols;  paths_out (may be NULL): k x path_len x 4. */
int32_t nlx_commit_open_rows(nlx_commit* c, const uint64_t* idx, size_t k, uint64_t* rows_out,
                             uint64_t* paths_out);
/* a10: PolynomialBatch polynomials evaluated at zeta in the quadratic extension
 * (OpeningSet::new's eval_commitment).  zeta: 2 words; out_ext: n_cols x 2 words. */
int32_t nlx_commit_eval_at(nlx_commit* c, const uint64_t zeta[2], uint64_t* out_ext);
/* whole LDE table in plonky2's leaf order: (n << rate_bits) x n_cols row-major (debug / tests) */
int32_t nlx_commit_get_leaves(nlx_commit* c, uint64_t* leaves_out);
/* level-major digests as in nlx_merkle_build */
int32_t nlx_commit_get_digests(nlx_commit* c, uint64_t* digests_out);

/* ===================================================================================
 * Whole-proof interface: plonky2::plonk::circuit_data / prover (a14) and its stages
 * (a8 partial products, a9 quotient, a10 openings, a11 FRI), SURVEY.md §3.4.
 * =================================================================================== */

/* gate kinds understood by the constraint combiner (plonky2::gates::*) */
#define NLX_GATE_NOOP 0          /* gates::noop::NoopGate */
#define NLX_GATE_CONSTANT 1      /* gates::constant::ConstantGate { num_consts = param0 } */
#define NLX_GATE_PUBLIC_INPUT 2  /* gates::public_input::PublicInputGate */
#define NLX_GATE_ARITHMETIC 3    /* gates::arithmetic_base::ArithmeticGate { num_ops = param0 } */
#define NLX_GATE_BASE_SUM 4      /* gates::base_sum::BaseSumGate<B = param0> { num_limbs = param1 } */
#define NLX_GATE_POSEIDON 5      /* gates::poseidon::PoseidonGate */
#define NLX_GATE_ARITHMETIC_EXT 6 /* gates::arithmetic_extension::ArithmeticExtensionGate<2> { num_ops = param0 } */
#define NLX_GATE_MUL_EXT 7        /* gates::multiplication_extension::MulExtensionGate<2> { num_ops = param0 } */
#define NLX_GATE_REDUCING 8       /* gates::reducing::ReducingGate<2> { num_coeffs = param0 } */
#define NLX_GATE_REDUCING_EXT 9   /* gates::reducing_extension::ReducingExtensionGate<2> { num_coeffs = param0 } */
#define NLX_GATE_POSEIDON_MDS 10   /* gates::poseidon_mds::PoseidonMdsGate */
#define NLX_GATE_EXPONENTIATION 11 /* gates::exponentiation::ExponentiationGate { num_power_bits = param0 } */
#define NLX_GATE_RANDOM_ACCESS 12  /* gates::random_access::RandomAccessGate { bits = param0, num_copies = param1 & 0xFFFF,
                                      num_extra_constants = param1 >> 16 } */
#define NLX_GATE_COSET_INTERPOLATION 13 /* gates::coset_interpolation::CosetInterpolationGate<2> { subgroup_bits = param0,
                                           degree = param1 } - the FRI-verifier gate of every recursive (reduce) proof */
/* plonky2x frontend::num::u32::gates (from plonky2-u32), 2-bit range-check limbs: the u32 / u64 arithmetic and
 * comparisons of nearx's header, height and stake logic (nearx/src/builder.rs ensure_height / ensure_stake ...) */
#define NLX_GATE_U32_ADD_MANY 14    /* U32AddManyGate { num_addends = param0, num_ops = param1 } */
#define NLX_GATE_U32_ARITHMETIC 15  /* U32ArithmeticGate { num_ops = param0 } */
#define NLX_GATE_U32_SUBTRACTION 16 /* U32SubtractionGate { num_ops = param0 } */
#define NLX_GATE_U32_RANGE_CHECK 17 /* U32RangeCheckGate { num_input_limbs = param0 } */
#define NLX_GATE_COMPARISON 18      /* ComparisonGate { num_bits = param0, num_chunks = param1 } */
/* gates::lookup::LookupGate / gates::lookup_table::LookupTableGate of table param0: no gate constraints; their wires feed
 * the lookup argument (vanishing_poly::check_lookup_constraints).  LookupGate slot i = wires (2i, 2i+1) = (input, output),
 * num_routed_wires / 2 slots; LookupTableGate slot i = wires (3i, 3i+1, 3i+2) = (input, output, multiplicity),
 * num_routed_wires / 3 slots. */
#define NLX_GATE_LOOKUP 19
#define NLX_GATE_LOOKUP_TABLE 20
#define NLX_GATE_KIND_MAX 20

typedef struct {
    uint32_t kind;
    uint32_t selector_index; /* selector polynomial used by this gate (gates::selectors::SelectorsInfo) */
    uint32_t group_start;    /* gate-index range [start, end) that shares the selector */
    uint32_t group_end;
    uint32_t index;          /* position in the sorted gate list = selector value on the gate's rows */
    uint32_t param0, param1;
} nlx_gate_desc;

/* CommonCircuitData + CircuitConfig + FriParams, flattened (standard_recursion_config values in
 * comments).  Same field order as the oracle's orc_circuit_desc. */
typedef struct {
    uint32_t degree_bits;
    uint32_t num_wires;              /* 135 */
    uint32_t num_routed_wires;       /* 80 */
    uint32_t num_constants;          /* 2 (gate constants per row, selectors excluded) */
    uint32_t num_challenges;         /* 2 */
    uint32_t rate_bits;              /* 3 */
    uint32_t cap_height;             /* 4 */
    uint32_t quotient_degree_factor; /* 8 */
    uint32_t num_partial_products;   /* 9 */
    uint32_t fri_pow_bits;           /* 16 */
    uint32_t fri_num_queries;        /* 28 */
    uint32_t fri_arity_bits;         /* 4 */
    uint32_t fri_final_poly_bits;    /* 5 */
    uint32_t num_selectors;
    uint32_t num_gates;
    uint32_t num_public_inputs;
    const nlx_gate_desc* gates;
    const uint64_t* k_is;            /* num_routed_wires coset shifts (host pointer) */
    uint64_t circuit_digest[4];      /* all zero: computed by nlx_circuit_build as plonky2 does */
    /* ---- lookup tables (CommonCircuitData::luts, ProverOnlyCircuitData::{lookup_rows, lut_to_lookups}) ----
     * num_luts == 0 (a zero-initialised tail): no lookup argument, the rest is ignored.  With tables:
     *  - the constants matrix carries 4 + num_luts lookup selector columns between the gate selectors and the gate
     *    constants (gates::selectors::selectors_lookup: TransSre, TransLdc, InitSre, LastLdc; selector_ends_lookups);
     *  - nlx_prove first does prover::set_lookup_wires ON THE DEVICE WITNESS IT IS HANDED (multiplicity wires of the
     *    LookupTableGate rows, padding slots of each table's last LookupGate row - the caller's buffer is written);
     *  - 2 * num_challenges more challenges are drawn after the gammas, the Zs commitment carries the
     *    num_challenges * (1 + S) lookup polynomials (RE, SLDC_0..S-1; S = ceil(num_routed_wires / 2 / (qdf - 1)))
     *    after the partial products, they are opened at zeta and g * zeta, and the vanishing polynomial carries
     *    check_lookup_constraints' 4 + num_luts + 2 S terms per challenge between the permutation and the gate terms.
     * All four arrays are host pointers, copied by nlx_circuit_build. */
    uint32_t num_luts;               /* 0 .. 16 */
    uint32_t pad_;
    const uint32_t* lut_sizes;       /* num_luts: entries per table (1 .. 65536) */
    const uint16_t* lut_pairs;       /* the tables' (input, output) pairs, table after table, 2 x u16 per entry */
    const uint32_t* lookup_rows;     /* 3 per table: last_lu_row, last_lut_row, first_lut_row (LookupWire); the LookupGate rows
                                        are [last_lu_row, last_lut_row), the LookupTableGate rows [last_lut_row, first_lut_row],
                                        row first_lut_row + 1 is a NoopGate row */
    const uint32_t* lut_num_lookups; /* num_luts: lookups made into each table (lut_to_lookups[t].len()) */
} nlx_circuit_desc;

typedef struct nlx_circuit nlx_circuit;

/* CircuitBuilder::build's prover data: commits the constants (selectors first, then gate
 * constants; (num_selectors + num_constants) x n) and the sigma polynomials (num_routed_wires x n),
 * both column-major subgroup evaluations, and keeps everything the prover needs resident in HBM. */
int32_t nlx_circuit_build(nlx_ctx* ctx, const nlx_circuit_desc* desc, const uint64_t* constants,
                          const uint64_t* sigmas, nlx_circuit** out);
void nlx_circuit_destroy(nlx_circuit* c);
int32_t nlx_circuit_digest(const nlx_circuit* c, uint64_t out[4]);
int32_t nlx_circuit_constants_sigmas_cap(const nlx_circuit* c, uint64_t* cap_out);
size_t nlx_proof_max_bytes(const nlx_circuit* c);

/* prove_with_partition_witness + ProofWithPublicInputs::to_bytes.
 * wires: num_wires x n column-major (host or device).  proof_out: host buffer of proof_cap bytes.
 * Fails with NLX_E_INVAL if the witness does not satisfy the circuit (quotient degree check). */
int32_t nlx_prove(nlx_circuit* c, const uint64_t* wires, const uint64_t* public_inputs, uint8_t* proof_out,
                  size_t proof_cap, size_t* proof_len);

/* ---- stage-level entry points: the fine seam for callers that keep plonky2's own
 * prove_with_partition_witness loop and replace it stage by stage (SURVEY.md §8b call list) ---- */

/* The circuit's constants + sigmas commitment (FRI oracle 0), borrowed: owned by the circuit, do not destroy. */
const nlx_commit* nlx_circuit_constants_sigmas(const nlx_circuit* c);
/* a8 wires_permutation_partial_products_and_zs + PolynomialBatch::from_values: wires = num_wires x n subgroup
 * values (host or device); the commitment holds num_challenges Z columns followed by the partial products. */
int32_t nlx_partial_products_and_zs(nlx_circuit* c, const uint64_t* wires, const uint64_t betas[2], const uint64_t gammas[2],
                                    nlx_commit** zs_out);
/* a9 compute_quotient_polys + PolynomialBatch::from_coeffs: the num_challenges * quotient_degree_factor quotient
 * chunks, committed.  public_inputs_hash = hash_no_pad(public inputs) (nlx_hash_no_pad). */
int32_t nlx_quotient_eval(nlx_circuit* c, const nlx_commit* wires, const nlx_commit* zs, const uint64_t betas[2],
                          const uint64_t gammas[2], const uint64_t alphas[2], const uint64_t public_inputs_hash[4],
                          nlx_commit** quotient_out);

/* plonky2::iop::challenger::Challenger state: sponge_state, input_buffer, output_buffer (popped from the end). */
typedef struct {
    uint64_t state[12];
    uint64_t input[8];
    uint32_t n_input;
    uint32_t pad0;
    uint64_t output[8];
    uint32_t n_output;
    uint32_t pad1;
} nlx_challenger;
void nlx_challenger_init(nlx_challenger* c);
int32_t nlx_challenger_observe(nlx_challenger* c, const uint64_t* elements, size_t n);
int32_t nlx_challenger_challenge(nlx_challenger* c, uint64_t* out, size_t n);
/* PoseidonHash::hash_no_pad on the host (public-inputs hash, circuit digest) */
int32_t nlx_hash_no_pad(const uint64_t* elements, size_t n, uint64_t out[4]);

typedef struct {
    uint32_t arity_bits;       /* FriReductionStrategy::ConstantArityBits(arity_bits, final_poly_bits) */
    uint32_t final_poly_bits;
    uint32_t pow_bits;         /* proof_of_work_bits */
    uint32_t num_queries;      /* num_query_rounds */
} nlx_fri_params;
/* a11 PolynomialBatch::prove_openings / fri_proof for an instance of the shape every caller on this path has:
 * batch 0 = every column of every oracle (in order) opened at zeta, batch 1 = the first n_next[o] columns of
 * every oracle o (in order) opened at g * zeta (plonky2: oracles = constants_sigmas, wires, zs_partial_products,
 * quotient and n_next = {0, 0, num_challenges, 0}; a STARK: n_next = all columns of every trace oracle).  openings_* are the extension values (2 words each)
 * already observed by the caller's challenger; the challenger is advanced exactly as upstream's
 * (fri alpha, commit-phase caps and betas, final polynomial, proof of work, query indices).  Writes the FriProof
 * bytes: commit_phase_merkle_caps, query_round_proofs, final_poly, pow_witness. */
int32_t nlx_fri_prove(nlx_ctx* ctx, const nlx_commit* const* oracles, uint32_t n_oracles, const uint32_t* n_next,
                      const uint64_t zeta[2], const uint64_t* openings_zeta, const uint64_t* openings_next,
                      const nlx_fri_params* params, nlx_challenger* challenger, uint8_t* proof_out, size_t proof_cap,
                      size_t* proof_len);

/* a13: plonky2x LocalProver::batch_prove.  Proves n_jobs independent jobs with n_workers concurrent
 * workers.  workers[i] are circuits built from the SAME description on DISTINCT contexts (one stream +
 * one host thread each; contexts may be on the same GPU - overlapping one proof's latency-bound phases
 * with another's throughput-bound kernels - or on different GPUs).  Jobs are taken in order by whichever
 * worker is free; each job's outcome is written to its own status / proof_len.  Returns NLX_OK if every
 * job succeeded, else the first failing job's code. */
typedef struct {
    const uint64_t* wires;          /* num_wires x n column-major, host or device (device of the worker that runs it) */
    const uint64_t* public_inputs;
    uint8_t* proof_out;             /* host buffer */
    size_t proof_cap;
    size_t proof_len;               /* out */
    int32_t status;                 /* out */
} nlx_prove_job;
int32_t nlx_batch_prove(nlx_circuit* const* workers, uint32_t n_workers, nlx_prove_job* jobs, size_t n_jobs);

/* Per-stage device time of the most recent nlx_prove on this circuit, in milliseconds
 * (HIP events on the context's stream).  names_out receives static strings. */
#define NLX_MAX_STAGES 24
int32_t nlx_prove_stage_times(const nlx_circuit* c, uint32_t* n_stages, const char** names_out, float* ms_out);

/* a11 fri_proof_of_work: smallest nonce w such that the Poseidon duplex of `state` with w written
 * at `pos` has at least `bits` leading zero bits in output word 7. */
int32_t nlx_pow_grind(nlx_ctx* ctx, const uint64_t state[12], uint32_t pos, uint32_t bits, uint64_t* nonce_out);

/* ---- a12: starky-style STARK prover ----
 * Replaces starky::prover::prove / compute_quotient_polys / StarkOpeningSet::new / Stark::fri_instance -
 * the public ancestor of the un-vendored starkyx (curta) prover that plonky2x runs for nearx's
 * curta_eddsa_verify / curta_sha256 calls (nearx/src/builder.rs; Cargo.lock:6515).  The AIR is data: a
 * register program replacing Stark::eval_packed_generic + ConstraintConsumer, interpreted per point of
 * the quotient coset.  Words are op | dst << 8 | a << 24 | b << 40 (16-bit fields); NLX_AIR_CONST is
 * followed by one immediate word.  Registers r0 .. r63; a register must be written before it is read.
 * Each EMIT feeds every challenge's accumulator: acc_j = acc_j * alpha_j + c. */
#define NLX_AIR_LOCAL 0            /* r[dst] = local_values[a] */
#define NLX_AIR_NEXT 1             /* r[dst] = next_values[a] */
#define NLX_AIR_PUBLIC 2           /* r[dst] = public_inputs[a] */
#define NLX_AIR_CONST 3            /* r[dst] = immediate (next word) */
#define NLX_AIR_ADD 4              /* r[dst] = r[a] + r[b] * 2^sh, sh = word bits 56..61 (0 = plain add) */
#define NLX_AIR_SUB 5              /* r[dst] = r[a] - r[b] * 2^sh */
#define NLX_AIR_MUL 6
#define NLX_AIR_EMIT_TRANSITION 7  /* ConstraintConsumer::constraint_transition(r[a]): times (x - g^-1) */
#define NLX_AIR_EMIT_FIRST 8       /* constraint_first_row(r[a]): times L_0(x) */
#define NLX_AIR_EMIT_LAST 9        /* constraint_last_row(r[a]): times L_{n-1}(x) */
#define NLX_AIR_EMIT 10            /* constraint(r[a]) on every row */
#define NLX_AIR_PERIODIC 11        /* r[dst] = periodic column a at this row (values[a][row mod period]) */
#define NLX_AIR_PACK_LOCAL 12      /* r[dst] = sum_{i<b} 2^i local_values[a+i], 1 <= b <= 32 (bits -> word) */
#define NLX_AIR_PACK_NEXT 13       /* r[dst] = sum_{i<b} 2^i next_values[a+i] */
#define NLX_AIR_EMIT_BOOL 14       /* constraint(x * (x - 1)) for x = local_values[a .. a + max(b, 1)), in column order */
#define NLX_AIR_LOADV 15           /* scheduling hint: the next `dst` (<= 8) words are independent LOCAL / NEXT / PUBLIC /
                                      PERIODIC loads with distinct destinations; the kernel issues them together */
/* three-operand forms for bit-valued columns (bitwise hash AIRs); the third register index is in bits 56..61 */
#define NLX_AIR_XOR3 16            /* r[dst] = a ^ b ^ c as a polynomial: s = a + b - 2ab, s + c - 2sc */
#define NLX_AIR_CH 17              /* r[dst] = c + a (b - c) */
#define NLX_AIR_MAJ 18             /* r[dst] = ab + c (a + b - 2ab) */
/* A segment boundary: no register value is carried across this word (the builder rejects a read of a register not
 * written since the last boundary).  A sequential interpreter may ignore it; the GPU evaluates the segments of a
 * long program as independent work items - (quotient points) x (segments) waves instead of (quotient points) -
 * and adds the segments' partial sums with the alpha powers their position in the program implies, so a wide AIR
 * on a short trace still fills the chip.  At most NLX_AIR_MAX_SEGMENTS - 1 boundaries per program. */
#define NLX_AIR_SEGMENT 19
/* The two base-field constraints of one LogUp helper (near-light-client_amd/logup.py) as one instruction:
 *   h (alpha + v1)(alpha + v2) = (alpha + v1) + (alpha + v2)   in F_p[X]/(X^2 - 7),
 * v1 = local[a], v2 = local[dst] (dst = 0xFFFF: a single lookup, h (alpha + v1) = 1), h = local[b] + local[b+1] X,
 * alpha = challenge k + challenge k+1 X with k in word bits 56..61.  Emits the X^0 then the X^1 coefficient. */
#define NLX_AIR_EMIT_LOGUP 20
#define NLX_AIR_MAC 21             /* r[dst] = r[c] + r[a] * r[b], c in word bits 56..61: the inner step of every limb convolution */
#define NLX_AIR_MAX_SEGMENTS 256
#define NLX_AIR_NUM_REGS 64
#define NLX_AIR_MAX_PERIODIC 128

typedef struct {
    uint32_t degree_bits;
    uint32_t n_cols;                  /* Stark::COLUMNS */
    uint32_t num_challenges;          /* StarkConfig::standard_fast_config: 2 */
    uint32_t rate_bits;               /* 1 */
    uint32_t cap_height;              /* 4 */
    uint32_t quotient_degree_factor;  /* Stark::quotient_degree_factor() rounded up to a power of two, <= 2^rate_bits */
    uint32_t fri_pow_bits;            /* 16 */
    uint32_t fri_num_queries;         /* 84 */
    uint32_t fri_arity_bits;          /* 4 */
    uint32_t fri_final_poly_bits;     /* 5 */
    uint32_t num_public_inputs;       /* Stark::PUBLIC_INPUTS */
    uint32_t n_words;
    const uint64_t* program;          /* host */
    /* Periodic columns (round constants, round selectors): known to the verifier, not committed.  Column a
     * has the value periodic[a * period + (row mod period)], period = 2^period_bits <= n; as a polynomial it
     * is P_a(x^(n/period)) with P_a the interpolation over the period-th roots of unity (degree < n, counts
     * as degree 1 in the constraint degree). */
    uint32_t n_periodic;              /* <= NLX_AIR_MAX_PERIODIC */
    uint32_t period_bits;
    const uint64_t* periodic;         /* host, n_periodic x period */
    /* Rounds of commitment (starkyx's TraceWriter rounds: lookup / bus accumulators are functions of challenges
     * drawn after the main trace is committed).  0 = classic single-round starky.  Round r commits round_cols[r]
     * columns (sum = n_cols; program column indices run through the rounds in order); once its Merkle cap is in the
     * transcript the verifier draws round_challenges[r] base-field challenges, which the program reads as
     * NLX_AIR_PUBLIC indices num_public_inputs + k in the order drawn.  The alphas follow the last round. */
    uint32_t n_rounds;                /* 0..3 */
    uint32_t round_cols[3];
    uint32_t round_challenges[3];
    /* Grouped Merkle leaves.  0 = every leaf is plonky2's hash_or_noop of the whole LDE row (starky's MerkleTree).  G > 0:
     * a row of more than G columns is hashed as hash_no_pad(hash_no_pad(cols [0, G)) || hash_no_pad(cols [G, 2G)) || ...) -
     * the tree's bottom level has arity ceil(n_cols / G) over column runs.  The proof's layout does not change (opened rows
     * and sibling paths), the verifier spends ceil(K / 2) more permutations per opened row; for the prover the K runs of a
     * leaf are independent work, which a trace of thousands of columns on a few thousand rows needs to fill the GPU
     * (host rule: stark.py StarkConfig.leaf_group_cols).  Applies to every commitment of the proof
     * (narrower ones are unaffected) and is part of the statement digest when non-zero.  8 <= G <= 4096. */
    uint32_t leaf_group_cols;
    /* Round values: round_values[r] (<= 64) field elements the prover sends with round r - totals of bus / accumulator
     * columns that depend on earlier challenges.  They enter the transcript after the round's cap and before its
     * challenges are drawn, travel after the public inputs at the end of the proof, and the program reads them as
     * NLX_AIR_PUBLIC: the values array is  public inputs | values of round 0 | challenges of round 0 | values of round 1 |
     * challenges of round 1 | ...  The proof only shows that the constraints hold for the values it carries; whoever
     * relies on the proof compares them with what they should be (e.g. the fingerprint of the claimed data). */
    uint32_t round_values[3];
    /* Openings digest.  0 = the transcript observes every opened value (starky's observe_openings: local ++ quotient at zeta,
     * then the next values).  G > 0: it observes FOUR elements instead - that vector, zero-padded to a multiple of G, hashed
     * in runs of G (hash_no_pad each) and the run digests hashed once more.  The binding is the same; the sequential absorption
     * of 19 000 values on one host thread (2 400 permutations: 3 of the 8 ms of the Sync step's SHA-512 proof) becomes one
     * small device launch and 150 host permutations.  Part of the statement digest when non-zero.
     * 8 <= G <= 4096; a patch that must stay compatible with an unpatched starky verifier passes 0. */
    uint32_t openings_group;
    /* Batches.  0 = a round's columns are ONE PolynomialBatch (one Merkle tree, one cap).  B > 0: a round of more than B columns
     * is committed as ceil(cols / B) PolynomialBatches of B columns (the last one what is left), in column order - each exactly
     * plonky2's PolynomialBatch::from_values: leaf = hash_or_noop(that batch's row), its own 2^cap_height-entry cap, its own FRI
     * oracle.  Nothing new for a verifier: the transcript observes the caps in order, the proof carries them in order, every
     * query opens each batch's row with its own Merkle path, the openings and their order do not change.  Why: a leaf of a
     * 4 745-column row is 594 permutations in SEQUENCE (a sponge), and a trace of 2^10 LDE rows has 1 024 of them - the GPU
     * idles while every lane walks its chain; ten batches are ten independent sponges per row.  The cost is on the verifier's
     * side: ceil(cols / B) - 1 more Merkle paths per query and round.  Part of the statement digest when non-zero.
     * 8 <= B <= 65535; at most NLX_STARK_MAX_ORACLES batches + 1 (the quotient) in all. */
    uint32_t batch_cols;
} nlx_stark_desc;
#define NLX_STARK_MAX_ORACLES 31
typedef struct nlx_stark nlx_stark;

/* Validates the program and keeps it and the coset tables resident on the device. */
int32_t nlx_stark_build(nlx_ctx* ctx, const nlx_stark_desc* desc, nlx_stark** out);
void nlx_stark_destroy(nlx_stark* s);
size_t nlx_stark_proof_max_bytes(const nlx_stark* s);
/* Which kernel evaluates this STARK's constraints on the quotient coset (starky::prover::compute_quotient_polys ->
 * Stark::eval_packed_generic; for the reference's circuits the curta AIRs behind /root/reference/nearx/src/builder.rs:152,220,316):
 * 1 = a straight-line kernel generated at library build time from exactly this program (near-light-client_amd/airgen.py: the
 * fixed AIRs of a Sync step and of the Verify job's map STARKs), 0 = the register-program interpreter (any program).  Same
 * results either way; the environment variable NLX_AIR_VM=1, read when the STARK is built, keeps the interpreter. */
int32_t nlx_stark_quotient_kernel(const nlx_stark* s);
/* starky::prover::prove + StarkProofWithPublicInputs serialisation (wire format in DESIGN.md):
 * trace: n_cols x n column-major (host or device), every value canonical. */
int32_t nlx_stark_prove(nlx_stark* s, const uint64_t* trace, const uint64_t* public_inputs, uint8_t* proof_out,
                        size_t proof_cap, size_t* proof_len);
/* Multi-round proving: round_fn(user, r, known, n_known, values_out) returns round r's columns (round_cols[r] x n,
 * column-major, host or device pointer, valid until the next callback or the end of the call) given `known` - the
 * round values and challenges of the earlier rounds, in values-array order - and writes the round's round_values[r]
 * values to values_out (NULL when the round sends none).  Proof bytes: one cap per round, the quotient cap, local / next values of every column in round
 * order, quotient values, FriProof (one oracle per round + the quotient oracle), public inputs, round values. */
typedef const uint64_t* (*nlx_round_fn)(void* user, uint32_t round, const uint64_t* known, uint32_t n_known, uint64_t* values_out);
int32_t nlx_stark_prove_rounds(nlx_stark* s, nlx_round_fn round_fn, void* user, const uint64_t* public_inputs,
                               uint8_t* proof_out, size_t proof_cap, size_t* proof_len);
int32_t nlx_stark_stage_times(const nlx_stark* s, uint32_t* n_stages, const char** names_out, float* ms_out);
/* As nlx_batch_prove, for STARK jobs: workers = provers built from the SAME description on DISTINCT contexts;
 * a job's `wires` field is its trace (n_cols x n, host or device). */
int32_t nlx_stark_batch_prove(nlx_stark* const* workers, uint32_t n_workers, nlx_prove_job* jobs, size_t n_jobs);
/* a12: range-check lookups for multi-round AIRs - the log-derivative argument with the challenge in the quadratic
 * extension (starkyx's lookup / bus accumulators are the reason it commits in rounds; its own constraints are not in
 * the reference tree, Cargo.lock:6515).  Constraint side: near-light-client_amd/logup.py.  Witness side, on the device:
 *   nlx_logup_multiplicities: trace is the round-0 buffer (n_cols x 2^log_n, column-major, host or device); the cells of
 *     the n_lookups columns `cols` must lie in [0, 2^table_bits) (else NLX_E_RANGE: the witness is wrong); column
 *     mult_col is overwritten with the multiplicities (the count of value v in row v).
 *   nlx_logup_round: for alpha = alpha[0] + alpha[1] X writes the nlx_logup_round_cols(n_lookups) round-1 columns into
 *     out: helpers h_j = 1/(alpha+v_2j) + 1/(alpha+v_2j+1) (two base columns each), g = m/(alpha+t) with t(i) = i mod
 *     2^table_bits, and the running sum phi(0) = 0, phi(i+1) = phi(i) + sum_j h_j(i) - g(i).
 * table_cols (a power of two, normally 1): the table may be spread over that many periodic columns of period
 * P = 2^table_bits / table_cols, column c holding c P + (i mod P), so that a trace shorter than the table can carry it;
 * mult_col is then the first of table_cols consecutive multiplicity columns (value v counts in column v / P, row v mod P)
 * and the round has one g per table column: helpers | g_0 .. g_(table_cols-1) | phi.
 * Both may be called from inside an nlx_round_fn callback on the same context. */
int32_t nlx_logup_multiplicities(nlx_ctx* ctx, uint64_t* trace, uint32_t n_cols, uint32_t log_n, const uint32_t* cols,
                                 uint32_t n_lookups, uint32_t table_bits, uint32_t table_cols, uint32_t mult_col);
uint32_t nlx_logup_round_cols(uint32_t n_lookups, uint32_t table_cols);
int32_t nlx_logup_round(nlx_ctx* ctx, const uint64_t* trace, uint32_t n_cols, uint32_t log_n, const uint32_t* cols,
                        uint32_t n_lookups, uint32_t table_bits, uint32_t table_cols, uint32_t mult_col, const uint64_t alpha[2],
                        uint64_t* out);
/* a12: the multiplication unit mod 2^255 - 19 (the field under the Ed25519 verifications of
 * curta_eddsa_verify_sigs_conditional, nearx/src/builder.rs:152) as a stand-alone chip: one a * b = c (mod p) per row.
 * Constraints: near-light-client_amd/fp25519.py::FpMulChip.  a, b: 2^log_rows operands of four little-endian 64-bit
 * words each (any value < 2^256).  Writes the NLX_FP25519_CHIP_COLS x 2^log_rows round-0 trace (16-bit limbs of a, b,
 * the canonical c, the quotient, the carries' low and high parts; the two multiplicity columns - of the 2^16 table and of
 * the 2^9 table of the carries' high parts - zeroed: fill them with nlx_logup_multiplicities). */
#define NLX_FP25519_CHIP_COLS 97
int32_t nlx_fp25519_chip_trace(nlx_ctx* ctx, const uint64_t* a, const uint64_t* b, uint32_t log_rows, uint64_t* trace_out);
/* a12 / f.1: trace generation on the GPU for the Ed25519 verification AIR (constraints and column layout:
 * near-light-client_amd/ed25519_air.py; caller in the reference: curta_eddsa_verify_sigs_conditional,
 * nearx/src/builder.rs:152).  slots: 2^log_slots signatures of NLX_ED25519_SLOT_WORDS = 32 little-endian 64-bit words
 * each: five 256-bit numbers A.x, A.y, R.x, R.y (affine, reduced), S; the 512-bit D = SHA-512(R || A || M) read as a
 * little-endian integer (8 words; the AIR reduces it mod L itself); word 28 = the slot's `active` flag (0: a validator
 * that did not sign - the rows are generated with the three checks off); three spare words.  Writes the NLX_ED25519_COLS0 x
 * (256 << log_slots) round-0 trace (host or device), the two multiplicity columns (2^9 table of the carries' high parts,
 * then the 2^16 table - last, so that a proof of fewer than 2^8 slots can append the further multiplicity columns of a
 * table spread over several columns) zeroed (nlx_logup_multiplicities fills them).  Returns NLX_E_INVAL naming the first
 * slot whose statement is false (an active slot whose signature does not verify or whose A / R is off the curve; any slot
 * with S >= L or a coordinate >= p): no trace satisfies the AIR for it. */
#define NLX_ED25519_SLOT_WORDS 32
#define NLX_ED25519_COLS0 1488
int32_t nlx_ed25519_trace(nlx_ctx* ctx, const uint64_t* slots, uint32_t log_slots, uint64_t* trace_out);
/* The AIR's binding accumulator (round 1, two base columns = one column over F_p^2) for the challenge gamma = gamma[0] +
 * gamma[1] X: the running Horner fingerprint of every slot's 96 limbs (limb 15 first; the 32-byte encodings of A and R - y with the
 * parity of x in bit 255 -, S, D mod 2^256, D div 2^256, active), each row
 * holding what was absorbed before it.  trace: the round-0 trace (host or device); acc_out: 2 x (256 << log_slots);
 * total_out: the fingerprint of all slots - the round value the proof sends, which the relying party recomputes from
 * the tuples it believes were verified (near-light-client_amd/ed25519_air.py::fingerprint). */
int32_t nlx_ed25519_bind_round(nlx_ctx* ctx, const uint64_t* trace, uint32_t log_slots, const uint64_t gamma[2],
                               uint64_t* acc_out, uint64_t total_out[2]);
/* f.1: trace generation on the GPU for the SHA-256 compression AIR (column layout and constraints:
 * near-light-client_amd/sha256_air.py; callers in the reference: curta_sha256 at nearx/src/merkle.rs:49,
 * nearx/src/variables.rs:71-72).  blocks: 2^log_blocks padded 512-bit blocks as 16 big-endian-decoded words
 * each; is_first[b] != 0 where block b starts a new message (block 0 always does).  Writes the
 * NLX_SHA256_COLS x (4 << log_blocks) column-major trace (host or device buffer) and, if digest_out is
 * not NULL, the eight words of the last block's output chaining value (the AIR's public inputs). */
#define NLX_SHA256_COLS 1953
int32_t nlx_sha256_trace(nlx_ctx* ctx, const uint32_t* blocks, const uint8_t* is_first, uint32_t log_blocks,
                         uint64_t* trace_out, uint64_t digest_out[8]);
/* The SHA-256 AIR's binding accumulator (round 1, one column over F_p^2 = two base columns) for the challenge gamma:
 * the running Horner fingerprint of every block's (message-start flag, 16 message words) - absorbed on the block's first
 * row - and 8 output chaining words - on its last row -, each row holding what was absorbed before it.  trace: the
 * round-0 trace; acc_out: 2 x (4 << log_blocks); total_out: the round value the proof sends
 * (near-light-client_amd/sha256_air.py::fingerprint is the relying party's side). */
int32_t nlx_sha256_bind_round(nlx_ctx* ctx, const uint64_t* trace, uint32_t log_blocks, const uint64_t gamma[2],
                              uint64_t* acc_out, uint64_t total_out[2]);
/* The SHA-512 sibling (column layout and constraints: near-light-client_amd/sha512_air.py; caller in the reference:
 * the SHA-512 of R || A || M inside curta_eddsa_verify_sigs_conditional, nearx/src/builder.rs:152).  blocks:
 * 2^log_blocks padded 1024-bit blocks as 16 big-endian-decoded 64-bit words each.  Writes the NLX_SHA512_COLS x
 * (4 << log_blocks) column-major trace and, if digest_out is not NULL, the eight 64-bit words of the last block's
 * output chaining value (the AIR's sixteen public inputs are their (low, high) 32-bit halves, in that order). */
#define NLX_SHA512_COLS 4745
int32_t nlx_sha512_trace(nlx_ctx* ctx, const uint64_t* blocks, const uint8_t* is_first, uint32_t log_blocks,
                         uint64_t* trace_out, uint64_t digest_out[8]);
/* Round 1 of the SHA-512 AIR: the binding accumulator for gamma (as nlx_sha256_bind_round; 64-bit words are absorbed as
 * (low, high) halves: 33 elements on a block's first row, 16 on its last). */
int32_t nlx_sha512_bind_round(nlx_ctx* ctx, const uint64_t* trace, uint32_t log_blocks, const uint64_t gamma[2],
                              uint64_t* acc_out, uint64_t total_out[2]);
#ifdef __cplusplus
}
#endif
#endif /* NLX_H */
