// Trace generation for the mod-(2^255 - 19) multiplication chip (near-light-client_amd/fp25519.py::FpMulChip; SURVEY.md
// §8a row a12 - the field arithmetic under curta_eddsa_verify_sigs_conditional, nearx/src/builder.rs:152): one lane per
// row computes c = a b mod p, the quotient and the carries in 16-bit limbs and writes its 97 cells; a wave's 64
// lanes are 64 consecutive rows, so every column store is 512 contiguous bytes.
#include "ctx.hpp"
#include "fp25519.hpp"
#include "transcript.hpp"

namespace nlx {

// chip columns: a[16] b[16] | unit cells c[16] q[17] lo[15] hi[15] | the two multiplicity columns
enum : uint32_t { cA = 0, cB = 16, cC = 32, cMULT16 = 95, cMULT9 = 96 };
static_assert(cC + fp::UNIT_CELLS == cMULT16 && cMULT9 + 1 == NLX_FP25519_CHIP_COLS, "column map");

__global__ __launch_bounds__(256) void k_fp25519_chip_trace(const uint64_t* __restrict__ a, const uint64_t* __restrict__ b,
                                                            uint32_t log_n, uint64_t* __restrict__ trace) {
    const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >> log_n) return;
    int32_t al[16], bl[16];
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const uint64_t x = a[row * 4 + w], y = b[row * 4 + w];
#pragma unroll
        for (int h = 0; h < 4; h++) {
            al[4 * w + h] = (int32_t)((x >> (16 * h)) & 0xFFFF);
            bl[4 * w + h] = (int32_t)((y >> (16 * h)) & 0xFFFF);
        }
    }
    int64_t prod[32];
    for (int k = 0; k < 32; k++) prod[k] = 0;
    fp::mul_acc(prod, al, bl);
    fp::Unit u;
    fp::finish(prod, u);
    auto put = [&](uint32_t col, uint64_t v) { trace[((size_t)col << log_n) + row] = v; };
    for (int i = 0; i < 16; i++) {
        put(cA + i, (uint64_t)al[i]);
        put(cB + i, (uint64_t)bl[i]);
    }
    u.cells(cC, put);
    put(cMULT16, 0);
    put(cMULT9, 0);
}

}  // namespace nlx

using namespace nlx;

extern "C" int32_t nlx_fp25519_chip_trace(nlx_ctx* ctx, const uint64_t* a, const uint64_t* b, uint32_t log_rows,
                                          uint64_t* trace_out) NLX_TRY {
    if (!ctx) return NLX_E_INVAL;
    if (!a || !b || !trace_out) return ctx->fail(NLX_E_INVAL, "NULL argument");
    if (log_rows < 4 || log_rows > 24) return ctx->fail(NLX_E_RANGE, "log_rows must be in [4, 24]");
    (void)hipSetDevice(ctx->device);
    const size_t n = (size_t)1 << log_rows;
    Staged sa(ctx, a, n * 32, true, false);
    if (sa.status) return sa.status;
    Staged sb(ctx, b, n * 32, true, false);
    if (sb.status) return sb.status;
    Staged st(ctx, trace_out, (size_t)NLX_FP25519_CHIP_COLS * n * 8, false, true);
    if (st.status) return st.status;
    hipLaunchKernelGGL(k_fp25519_chip_trace, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, sa.as<uint64_t>(),
                       sb.as<uint64_t>(), log_rows, st.as<uint64_t>());
    int32_t rc = st.finish();
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (!rc && e != hipSuccess) rc = ctx->hip_fail(e, "hipStreamSynchronize");
    hipError_t le = hipGetLastError();
    if (!rc && le != hipSuccess) rc = ctx->hip_fail(le, "kernel launch");
    return rc;
} NLX_CATCH(ctx)
