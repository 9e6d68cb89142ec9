// Host-side Fiat-Shamir transcript, proof byte writer and the pinned device->host fetch shared by the
// plonky2 prover (prover.hip), the generic FRI prover (fri.hip) and the STARK prover (stark.hip).
#pragma once
#include <cstring>
#include <stdint.h>
#include <stddef.h>
#include "ctx.hpp"
#include "gl.hpp"
#include "poseidon.hpp"

namespace nlx {

// plonky2::iop::challenger::Challenger (host)
struct Challenger {
    uint64_t state[12] = {0};
    uint64_t in_buf[8];
    unsigned n_in = 0;
    uint64_t out_buf[8];
    unsigned n_out = 0;
    void duplex() {
        for (unsigned i = 0; i < n_in; i++) state[i] = in_buf[i];
        n_in = 0;
        poseidon::permute(state);
        for (int i = 0; i < 8; i++) out_buf[i] = state[i];
        n_out = 8;
    }
    void observe(uint64_t e) {
        n_out = 0;
        in_buf[n_in++] = e;
        if (n_in == 8) duplex();
    }
    void observe(const uint64_t* e, size_t n) {
        for (size_t i = 0; i < n; i++) observe(e[i]);
    }
    uint64_t challenge() {
        if (n_in != 0 || n_out == 0) duplex();
        return out_buf[--n_out];
    }
    void ext_challenge(uint64_t out[2]) {
        out[0] = challenge();
        out[1] = challenge();
    }
};

inline void hash_no_pad_host(const uint64_t* in, size_t len, uint64_t out[4]) {
    uint64_t st[12] = {0};
    for (size_t off = 0; off < len; off += 8) {
        size_t k = len - off < 8 ? len - off : 8;
        for (size_t j = 0; j < k; j++) st[j] = in[off + j];
        poseidon::permute(st);
    }
    memcpy(out, st, 32);
}

// plonky2::util::serialization::Buffer (write side), little-endian
struct Writer {
    uint8_t* p;
    size_t len = 0, cap;
    bool overflow = false;
    void bytes(const void* src, size_t n) {
        if (len + n > cap) { overflow = true; return; }
        memcpy(p + len, src, n);
        len += n;
    }
    void u64s(const uint64_t* v, size_t n) { bytes(v, n * 8); }
    void u8(uint8_t v) { bytes(&v, 1); }
    void usize(uint64_t v) { bytes(&v, 8); }  // Write::write_usize: 8 little-endian bytes
};

// plonky2::fri::reduction_strategies::FriReductionStrategy::ConstantArityBits(arity_bits, final_poly_bits)
inline uint32_t fri_num_rounds(uint32_t degree_bits, uint32_t rate_bits, uint32_t cap_height, uint32_t arity_bits,
                               uint32_t final_poly_bits) {
    uint32_t r = 0;
    while (degree_bits > final_poly_bits && degree_bits + rate_bits >= cap_height + arity_bits) {
        if (degree_bits < arity_bits) break;
        degree_bits -= arity_bits;
        r++;
    }
    return r;
}

inline int32_t ensure_pinned(nlx_ctx* ctx, size_t bytes) {
    if (ctx->pinned_bytes >= bytes) return NLX_OK;
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    ctx->pinned = nullptr;
    ctx->pinned_bytes = 0;
    hipError_t e = hipHostMalloc(&ctx->pinned, bytes, hipHostMallocDefault);
    if (e != hipSuccess) return ctx->hip_fail(e, "hipHostMalloc");
    ctx->pinned_bytes = bytes;
    return NLX_OK;
}

// device -> host through the pinned staging buffer, synchronous
inline int32_t fetch(nlx_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes) {
    int32_t rc = ensure_pinned(ctx, bytes < (1u << 20) ? (1u << 20) : bytes);
    if (rc) return rc;
    NLX_HIP(ctx, hipMemcpyAsync(ctx->pinned, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    NLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(host_dst, ctx->pinned, bytes);
    return NLX_OK;
}

}  // namespace nlx
