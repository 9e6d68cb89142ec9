// Device FRI prover shared by the plonky2 proof (prover.hip) and the STARK proof (stark.hip):
// plonky2::fri::oracle::PolynomialBatch::prove_openings + fri::prover::fri_proof, value-domain version
// (DESIGN.md §FRI).  The FRI instance has the shape both callers need:
//   batch 0: every column of every oracle, in oracle order, opened at zeta
//   batch 1: the first nz[o] columns of every oracle o (in oracle order), opened at g * zeta
// (+ an optional trailing column group of one oracle that closes both batches: FriProveArgs::tail_cols)
#pragma once
#include <functional>
#include <vector>
#include "commit.hpp"
#include "transcript.hpp"

namespace nlx {

constexpr int FRI_MAX_ORACLES = 32;   // plonky2: 4; a STARK whose rounds are committed in batches (nlx_stark_desc.batch_cols): up to 31 + the quotient

struct FriProveArgs {
    const nlx_commit* oracles[FRI_MAX_ORACLES] = {};
    uint32_t n_oracles = 0;
    uint32_t nz[FRI_MAX_ORACLES] = {};  // columns of each oracle that are also opened at g * zeta
    // plonky2 with lookup tables (CommonCircuitData::fri_all_polys / fri_next_batch_polys): the last tail_cols columns of
    // oracle tail_oracle are listed AFTER every oracle in the zeta batch, and after the nz[] columns in the g * zeta batch
    // (all of them are opened at both points).  open0 / open1 are in that order.
    uint32_t tail_oracle = 0, tail_cols = 0;
    uint64_t zeta[2], gzeta[2];
    const uint64_t* open0 = nullptr;  // host: ext openings of batch 0 (2 words each), oracle order
    const uint64_t* open1 = nullptr;  // host: ext openings of batch 1
    unsigned log_n = 0, rate_bits = 0, cap_height = 0, arity_bits = 0, pow_bits = 0;
    uint32_t n_queries = 0, n_rounds = 0;
    const uint64_t* d_coset_base = nullptr;  // device: g * w_L^r, r < 2^rate_bits
    const uint64_t* d_wA_inv = nullptr;      // device: w_A^-i, i < 2^arity_bits
};

// Draws fri_alpha, runs combine / commit phase / proof of work / queries and appends the FriProof
// (commit_phase_merkle_caps, query_round_proofs, final_poly, pow_witness) to `w`.  Device buffers are
// pushed onto `scratch` (the caller releases them after synchronising); `stage` marks timing stages.
int32_t fri_prove(nlx_ctx* ctx, const FriProveArgs& a, Challenger& ch, Writer& w, std::vector<void*>& scratch,
                  const std::function<void(const char*)>& stage);

// Small device tables used above (and by the quotient kernels): fills `h` with
// coset_base[R] | zh_inv[R] | w_R_inv_pows[R] | chunk_scale[R] for the 2^bits cosets of the size-n subgroup
// inside the size-(n << bits) one.
void coset_tables_host(unsigned log_n, unsigned bits, uint64_t* h);

}  // namespace nlx
