// Trace generation for the SHA-256 compression AIR ("AIR evaluators + trace generation on
// GPU"; callers nearx/src/merkle.rs:43-50, nearx/src/variables.rs:71-72 via curta_sha256).  Column layout:
// near-light-client_amd/sha256_air.py (NLX_SHA256_COLS columns, sixteen rounds per row, four rows per block).
//
// Two kernels.  k_sha_chain: one lane per message start walks that message's blocks and records every
// block's input chaining value.  k_sha_trace: one lane per trace row (below).
#include "ctx.hpp"
#include "gl.hpp"
#include "transcript.hpp"
#include <vector>

namespace nlx {

namespace sha {
__constant__ static const uint32_t K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
__constant__ static const uint32_t IV[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a,
                                            0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};

__device__ __forceinline__ uint32_t rotr(uint32_t x, int r) { return (x >> r) | (x << (32 - r)); }
__device__ __forceinline__ uint32_t s0(uint32_t x) { return rotr(x, 7) ^ rotr(x, 18) ^ (x >> 3); }
__device__ __forceinline__ uint32_t s1(uint32_t x) { return rotr(x, 17) ^ rotr(x, 19) ^ (x >> 10); }
__device__ __forceinline__ uint32_t S0(uint32_t x) { return rotr(x, 2) ^ rotr(x, 13) ^ rotr(x, 22); }
__device__ __forceinline__ uint32_t S1(uint32_t x) { return rotr(x, 6) ^ rotr(x, 11) ^ rotr(x, 25); }

// one block: h <- h + compress(h, m)
__device__ void compress(uint32_t h[8], const uint32_t* __restrict__ m) {
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = m[i];
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll 1
    for (int r0 = 0; r0 < 64; r0 += 16) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const uint32_t t1 = hh + S1(e) + ((e & f) ^ (~e & g)) + K[r0 + i] + w[i];
            const uint32_t t2 = S0(a) + ((a & b) ^ (a & c) ^ (b & c));
            w[i] = w[i] + s0(w[(i + 1) & 15]) + w[(i + 9) & 15] + s1(w[(i + 14) & 15]);
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}
}  // namespace sha

__global__ __launch_bounds__(64) void k_sha_chain(const uint32_t* __restrict__ blocks, const uint8_t* __restrict__ is_first,
                                                  uint32_t n_blocks, uint32_t* __restrict__ hin) {
    const uint32_t b0 = blockIdx.x * blockDim.x + threadIdx.x;
    if (b0 >= n_blocks || !(is_first[b0] || b0 == 0)) return;
    uint32_t h[8];
#pragma unroll
    for (int k = 0; k < 8; k++) h[k] = sha::IV[k];
    for (uint32_t b = b0; b < n_blocks && (b == b0 || !is_first[b]); b++) {
#pragma unroll
        for (int k = 0; k < 8; k++) hin[(size_t)b * 8 + k] = h[k];
        sha::compress(h, blocks + (size_t)b * 16);
        if (b == n_blocks - 1) {  // output chaining value of the last block = digest of the last message
#pragma unroll
            for (int k = 0; k < 8; k++) hin[(size_t)n_blocks * 8 + k] = h[k];
        }
    }
}

// Column layout (sha256_air.py): sixteen round slots of 105 columns, then the row's start state and block data
enum : uint32_t { kSLOT = 105, oA = 0, oE = 32, oW = 64, oCA = 96, oCE = 99, oCW = 102, oSW = 104, cPA = 16 * kSLOT,
                  cPE = cPA + 128, cHIN = cPE + 128, cCY = cHIN + 8, cIS_FIRST = cCY + 8 };
static_assert(cIS_FIRST + 1 == NLX_SHA256_COLS, "column map");

// One lane per trace row (four rows per block, sixteen rounds per row).  Every lane replays its block's 64 rounds
// (a few hundred integer instructions, 4x redundant per block - noise next to the 15.6 KB it then writes) and
// emits its row; a wave's 64 lanes are 64 consecutive rows, so each of the 1 953 column stores is 512 contiguous
// bytes.
__global__ __launch_bounds__(256) void k_sha_trace(const uint32_t* __restrict__ blocks, const uint8_t* __restrict__ is_first,
                                                   const uint32_t* __restrict__ hin, uint32_t n_blocks,
                                                   uint64_t* __restrict__ trace) {
    const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = (size_t)n_blocks << 2;
    if (row >= n) return;
    const uint32_t blk = (uint32_t)(row >> 2), q = (uint32_t)(row & 3);
    const uint32_t prev = blk ? blk - 1 : n_blocks - 1;  // the schedule recurrence of row 0 looks back cyclically
    uint32_t w[80];  // w[16 + t] = W_t of this block for t = 0..63; w[0..15] = W_48..63 of the previous block
    {
        uint32_t wp[16];
#pragma unroll
        for (int i = 0; i < 16; i++) wp[i] = blocks[(size_t)prev * 16 + i];
#pragma unroll 1
        for (int t = 16; t < 64; t++) {
            const uint32_t v = wp[t & 15] + sha::s0(wp[(t + 1) & 15]) + wp[(t + 9) & 15] + sha::s1(wp[(t + 14) & 15]);
            wp[t & 15] = v;
        }
#pragma unroll
        for (int i = 0; i < 16; i++) w[i] = wp[(48 + i) & 15];
#pragma unroll
        for (int i = 0; i < 16; i++) w[16 + i] = blocks[(size_t)blk * 16 + i];
#pragma unroll 1
        for (int t = 16; t < 64; t++) w[16 + t] = w[t] + sha::s0(w[t + 1]) + w[t + 9] + sha::s1(w[t + 14]);
    }
    uint32_t h0[8];
#pragma unroll
    for (int k = 0; k < 8; k++) h0[k] = hin[(size_t)blk * 8 + k];
    // a[t + 4], e[t + 4] for t = -4..63, with the carries of every round
    uint32_t a[68], e[68];
    uint8_t ca[64], ce[64];
    a[3] = h0[0]; a[2] = h0[1]; a[1] = h0[2]; a[0] = h0[3];
    e[3] = h0[4]; e[2] = h0[5]; e[1] = h0[6]; e[0] = h0[7];
#pragma unroll 1
    for (int r = 0; r < 64; r++) {
        const uint32_t a1 = a[r + 3], a2 = a[r + 2], a3 = a[r + 1], a4 = a[r];
        const uint32_t e1 = e[r + 3], e2 = e[r + 2], e3 = e[r + 1], e4 = e[r];
        const uint64_t t1 = (uint64_t)e4 + sha::S1(e1) + ((e1 & e2) ^ (~e1 & e3)) + sha::K[r] + w[16 + r];
        const uint64_t sa = t1 + sha::S0(a1) + ((a1 & a2) ^ (a1 & a3) ^ (a2 & a3));
        const uint64_t se = (uint64_t)a4 + t1;
        a[r + 4] = (uint32_t)sa;
        e[r + 4] = (uint32_t)se;
        ca[r] = (uint8_t)(sa >> 32);
        ce[r] = (uint8_t)(se >> 32);
    }
    auto put = [&](uint32_t col, uint64_t v) { trace[(size_t)col * n + row] = v; };
    auto put_bits = [&](uint32_t base, uint32_t v, int cnt) {
        for (int i = 0; i < cnt; i++) put(base + i, (v >> i) & 1u);
    };
#pragma unroll 1
    for (uint32_t j = 0; j < 16; j++) {
        const uint32_t r = 16 * q + j, base = j * kSLOT;
        put_bits(base + oA, a[r + 4], 32);
        put_bits(base + oE, e[r + 4], 32);
        put_bits(base + oW, w[16 + r], 32);
        put_bits(base + oCA, ca[r], 3);
        put_bits(base + oCE, ce[r], 3);
        // w[16 + r + k] is W_{r + k}, k >= -16 (k < 0 in row 0 reaches the previous block's W_48..63)
        const uint64_t sw = (uint64_t)sha::s1(w[16 + r - 2]) + w[16 + r - 7] + sha::s0(w[16 + r - 15]) + w[16 + r - 16];
        put(base + oSW, (uint32_t)sw);
        put_bits(base + oCW, (uint32_t)(sw >> 32), 2);
    }
#pragma unroll 1
    for (uint32_t k = 0; k < 4; k++) {
        put_bits(cPA + 32 * k, a[16 * q + 3 - k], 32);
        put_bits(cPE + 32 * k, e[16 * q + 3 - k], 32);
    }
    const uint32_t fin[8] = {a[67], a[66], a[65], a[64], e[67], e[66], e[65], e[64]};
#pragma unroll
    for (int k = 0; k < 8; k++) {
        put(cHIN + k, h0[k]);
        put(cCY + k, q == 3 ? (uint32_t)(((uint64_t)h0[k] + fin[k]) >> 32) : 0u);
    }
    put(cIS_FIRST, (q == 0 && (is_first[blk] || blk == 0)) ? 1u : 0u);
}

// ---- binding accumulator (round 1 of the AIR): Horner fingerprint in F_p^2 of every block's (message-start flag, 16
// message words) and 8 output chaining words, read back from the trace ----
__device__ __forceinline__ uint64_t packed_word(const uint64_t* __restrict__ trace, size_t n, size_t row, uint32_t base) {
    uint64_t v = 0;
#pragma unroll 8
    for (int i = 0; i < 32; i++) v |= trace[(size_t)(base + i) * n + row] << i;
    return v;
}
// the 25 elements of block b, absorbed into acc: head (17) from its first row, tail (8) from its last
__device__ __forceinline__ gl::Ext sha_absorb_head(gl::Ext acc, gl::Ext gamma, const uint64_t* __restrict__ trace, size_t n, size_t row0) {
    acc = gl::mul(acc, gamma);
    acc.a = gl::add(acc.a, trace[(size_t)cIS_FIRST * n + row0]);
    for (uint32_t j = 0; j < 16; j++) {
        acc = gl::mul(acc, gamma);
        acc.a = gl::add(acc.a, packed_word(trace, n, row0, j * kSLOT + oW));
    }
    return acc;
}
__device__ __forceinline__ gl::Ext sha_absorb_tail(gl::Ext acc, gl::Ext gamma, const uint64_t* __restrict__ trace, size_t n, size_t row3) {
    for (uint32_t k = 0; k < 8; k++) {
        const uint32_t base = (15 - (k & 3)) * kSLOT + (k < 4 ? oA : oE);
        const uint64_t ho = (trace[(size_t)(cHIN + k) * n + row3] + packed_word(trace, n, row3, base)) & 0xFFFFFFFFull;
        acc = gl::mul(acc, gamma);
        acc.a = gl::add(acc.a, ho);
    }
    return acc;
}
__global__ __launch_bounds__(64) void k_sha_bind_block(const uint64_t* __restrict__ trace, uint32_t n_blocks, gl::Ext gamma,
                                                       gl::Ext* __restrict__ block_fp) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    const size_t n = (size_t)n_blocks << 2;
    gl::Ext acc = sha_absorb_head(gl::Ext{0, 0}, gamma, trace, n, (size_t)b * 4);
    block_fp[b] = sha_absorb_tail(acc, gamma, trace, n, (size_t)b * 4 + 3);
}
__global__ __launch_bounds__(64) void k_sha_bind_rows(const uint64_t* __restrict__ trace, uint32_t n_blocks, gl::Ext gamma,
                                                      const gl::Ext* __restrict__ start, uint64_t* __restrict__ out) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    const size_t n = (size_t)n_blocks << 2, row0 = (size_t)b * 4;
    const gl::Ext s0 = start[b], s1 = sha_absorb_head(s0, gamma, trace, n, row0);
    out[row0] = s0.a;
    out[n + row0] = s0.b;
    for (int q = 1; q < 4; q++) {  // rows 1..3 hold the value after the block's first row was absorbed
        out[row0 + q] = s1.a;
        out[n + row0 + q] = s1.b;
    }
}

}  // namespace nlx

using namespace nlx;

extern "C" int32_t nlx_sha256_bind_round(nlx_ctx* ctx, const uint64_t* trace, uint32_t log_blocks, const uint64_t gamma[2],
                                         uint64_t* acc_out, uint64_t total_out[2]) NLX_TRY {
    if (!ctx) return NLX_E_INVAL;
    if (!trace || !gamma || !acc_out || !total_out) return ctx->fail(NLX_E_INVAL, "NULL argument");
    if (log_blocks > 20) return ctx->fail(NLX_E_RANGE, "log_blocks must be <= 20");
    (void)hipSetDevice(ctx->device);
    const uint32_t n_blocks = 1u << log_blocks;
    const size_t n = (size_t)n_blocks << 2;
    Staged tr(ctx, trace, (size_t)NLX_SHA256_COLS * n * 8, true, false);
    if (tr.status) return tr.status;
    Staged so(ctx, acc_out, 2 * n * 8, false, true);
    if (so.status) return so.status;
    gl::Ext* d_fp = (gl::Ext*)ctx->alloc((size_t)n_blocks * sizeof(gl::Ext));
    if (!d_fp) return NLX_E_NOMEM;
    const gl::Ext g{gamma[0] % gl::P, gamma[1] % gl::P};
    hipStream_t st = ctx->stream;
    hipLaunchKernelGGL(k_sha_bind_block, dim3((n_blocks + 63) / 64), dim3(64), 0, st, tr.as<uint64_t>(), n_blocks, g, d_fp);
    std::vector<gl::Ext> fp_h(n_blocks), start(n_blocks);
    int32_t rc = fetch(ctx, fp_h.data(), d_fp, (size_t)n_blocks * sizeof(gl::Ext));
    if (!rc) {
        const gl::Ext g25 = gl::pow(g, 25);  // a block absorbs 17 + 8 elements
        gl::Ext acc{0, 0};
        for (uint32_t b = 0; b < n_blocks; b++) {
            start[b] = acc;
            acc = gl::add(gl::mul(acc, g25), fp_h[b]);
        }
        total_out[0] = acc.a;
        total_out[1] = acc.b;
        hipError_t e = hipMemcpyAsync(d_fp, start.data(), (size_t)n_blocks * sizeof(gl::Ext), hipMemcpyHostToDevice, st);
        if (e != hipSuccess) rc = ctx->hip_fail(e, "hipMemcpyAsync");
    }
    if (!rc) {
        hipLaunchKernelGGL(k_sha_bind_rows, dim3((n_blocks + 63) / 64), dim3(64), 0, st, tr.as<uint64_t>(), n_blocks, g, d_fp,
                           so.as<uint64_t>());
        rc = so.finish();
    }
    hipError_t e = hipStreamSynchronize(st);
    ctx->release(d_fp);
    if (!rc && e != hipSuccess) rc = ctx->hip_fail(e, "hipStreamSynchronize");
    hipError_t le = hipGetLastError();
    if (!rc && le != hipSuccess) rc = ctx->hip_fail(le, "kernel launch");
    return rc;
} NLX_CATCH(ctx)


extern "C" int32_t nlx_sha256_trace(nlx_ctx* ctx, const uint32_t* blocks, const uint8_t* is_first, uint32_t log_blocks,
                                    uint64_t* trace_out, uint64_t digest_out[8]) NLX_TRY {
    if (!ctx) return NLX_E_INVAL;
    if (!blocks || !is_first || !trace_out) return ctx->fail(NLX_E_INVAL, "NULL argument");
    if (log_blocks > 20) return ctx->fail(NLX_E_RANGE, "log_blocks must be <= 20");
    (void)hipSetDevice(ctx->device);
    const uint32_t n_blocks = 1u << log_blocks;
    const size_t n = (size_t)n_blocks << 2;  // four rows per block
    Staged sb(ctx, blocks, (size_t)n_blocks * 64, true, false);
    if (sb.status) return sb.status;
    Staged sf(ctx, is_first, n_blocks, true, false);
    if (sf.status) return sf.status;
    Staged st(ctx, trace_out, (size_t)NLX_SHA256_COLS * n * 8, false, true);
    if (st.status) return st.status;
    uint32_t* d_hin = (uint32_t*)ctx->alloc((size_t)(n_blocks + 1) * 32);
    if (!d_hin) return NLX_E_NOMEM;
    hipLaunchKernelGGL(k_sha_chain, dim3((n_blocks + 63) / 64), dim3(64), 0, ctx->stream, sb.as<uint32_t>(),
                       sf.as<uint8_t>(), n_blocks, d_hin);
    hipLaunchKernelGGL(k_sha_trace, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, sb.as<uint32_t>(),
                       sf.as<uint8_t>(), d_hin, n_blocks, st.as<uint64_t>());
    int32_t rc = st.finish();
    if (!rc && digest_out) {
        uint32_t dg[8];
        rc = fetch(ctx, dg, d_hin + (size_t)n_blocks * 8, 32);
        for (int k = 0; k < 8; k++) digest_out[k] = dg[k];
    }
    hipError_t e = hipStreamSynchronize(ctx->stream);
    ctx->release(d_hin);
    if (!rc && e != hipSuccess) rc = ctx->hip_fail(e, "hipStreamSynchronize");
    hipError_t le = hipGetLastError();
    if (!rc && le != hipSuccess) rc = ctx->hip_fail(le, "kernel launch");
    return rc;
} NLX_CATCH(ctx)
