// Trace generation for the Ed25519 verification AIR (near-light-client_amd/ed25519_air.py; SURVEY.md §8a row a12 /
// §8f.1: curta_eddsa_verify_sigs_conditional, nearx/src/builder.rs:152).  Two kernels over the row code of
// ed25519_rows.hpp:
//   k_ed_scan  one lane per slot walks its 256 rows (the point after each row depends on the one before) and
//              records the input point of every row - the only sequential part, on values only (fe25519_fast.hpp:
//              26-bit limbs, no witness cells; canonical results, so identical to what the row emitter recomputes);
//   k_ed_rows  one lane per row recomputes its 16 units from that input point (results are canonical, so the
//              recomputation is bit-identical) and writes its 1 488 cells; a wave's 64 lanes are 64 consecutive rows,
//              so every column store is 512 contiguous bytes.
#include "ctx.hpp"
#include "ed25519_rows.hpp"
#include "gl.hpp"
#include "transcript.hpp"
#include <vector>

namespace nlx {

static_assert(ed::N_COLS0 == NLX_ED25519_COLS0, "column map");

__global__ __launch_bounds__(64) void k_ed_scan(const uint64_t* __restrict__ words, uint32_t n_slots, ed::Slot* __restrict__ slots,
                                                ed::Point* __restrict__ in, ed::Point* __restrict__ fin) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_slots) return;
    ed::Slot s;
    ed::slot_from_words(words + (size_t)k * ed::SLOT_WORDS, s);
    slots[k] = s;
    ed::FastSlot fs;
    ed::fast_slot(s, fs);
    ed::FastPoint fq;
    {
        uint32_t zero[16], one[16];
        for (int i = 0; i < 16; i++) { zero[i] = 0; one[i] = i == 0; }
        fq.x = fe::from_limbs16(zero);
        fq.y = fe::from_limbs16(one);
        fq.z = fe::from_limbs16(one);
    }
    ed::Point q;
#pragma unroll 1
    for (int r = 0; r < ed::ROWS; r++) {
        ed::fast_store(fq, q);
        in[(size_t)k * ed::ROWS + r] = q;
        const int bit = ed::ROWS - 1 - r;
        ed::fast_row(fq, ((s.sw[bit >> 4] >> (bit & 15)) & 1) != 0, ((s.hw[bit >> 4] >> (bit & 15)) & 1) != 0, fs);
    }
    ed::fast_store(fq, q);
    fin[k] = q;
}

struct TracePut {
    uint64_t* trace;
    size_t n, row;
    __device__ __forceinline__ void operator()(uint32_t col, uint64_t v) { trace[(size_t)col * n + row] = v; }
};

__global__ __launch_bounds__(64) void k_ed_rows(const ed::Slot* __restrict__ slots, const ed::Point* __restrict__ in,
                                                const ed::Point* __restrict__ fin, uint32_t n_slots, uint64_t* __restrict__ trace,
                                                uint32_t* __restrict__ bad_slot) {
    const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = (size_t)n_slots * ed::ROWS;
    if (row >= n) return;
    const uint32_t k = (uint32_t)(row / ed::ROWS), prev = k ? k - 1 : n_slots - 1;
    const int r = (int)(row % ed::ROWS);
    const ed::Slot s = slots[k];
    const ed::Point p = in[row];
    ed::Point prev_fin;
    uint32_t prev_ry[16];
    bool prev_active = true;
    if (r == ed::STEP_YCMP) {
        prev_fin = fin[prev];
        for (int i = 0; i < 16; i++) prev_ry[i] = slots[prev].ry[i];
        prev_active = slots[prev].active != 0;
    }
    TracePut put{trace, n, row};
    ed::Point o;
    // row 0 checks the previous slot's Y comparison
    if (!ed::emit_row(r, s, p, prev_ry, &prev_fin, prev_active, put, o)) atomicMin(bad_slot, r == ed::STEP_YCMP ? prev : k);
    if (r == 1 && !s.s_in_range) atomicMin(bad_slot, k);   // S >= L: the comparison rows have no witness
}

// ---- binding accumulator (round 1): Horner fingerprint in F_p^2 of every slot's limbs, limb 15 first, in the order
// enc(A), enc(R) (y with the sign bit of x on top of limb 15), S, D low half, D high half, active (a one-limb value) ----
constexpr int ED_BOUND = 6;   // values absorbed per limb index
__device__ __forceinline__ gl::Ext absorb_limb(gl::Ext acc, gl::Ext gamma, const uint64_t* __restrict__ trace, size_t n, size_t row, int j) {
    const uint32_t base[ED_BOUND - 1] = {ed::cAY, ed::cRY, ed::cSW, ed::cDW, ed::cDW + 16};
    const uint32_t sign[2] = {ed::cSGA, ed::cSGR};
#pragma unroll
    for (int k = 0; k < ED_BOUND - 1; k++) {
        acc = gl::mul(acc, gamma);
        uint64_t v = trace[(size_t)(base[k] + j) * n + row];
        if (k < 2 && j == 15) v += trace[(size_t)sign[k] * n + row] << 15;   // both small: no reduction needed before the add
        acc.a = gl::add(acc.a, v);
    }
    acc = gl::mul(acc, gamma);
    if (j == 0) acc.a = gl::add(acc.a, trace[(size_t)ed::cACT * n + row]);
    return acc;
}

// one lane per slot: the slot's own fingerprint (starting from 0)
__global__ __launch_bounds__(64) void k_ed_bind_slot(const uint64_t* __restrict__ trace, uint32_t n_slots, gl::Ext gamma,
                                                     gl::Ext* __restrict__ slot_fp) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_slots) return;
    const size_t n = (size_t)n_slots * ed::ROWS, row = (size_t)k * ed::ROWS;
    gl::Ext acc{0, 0};
    for (int j = 15; j >= 0; j--) acc = absorb_limb(acc, gamma, trace, n, row, j);
    slot_fp[k] = acc;
}

// one lane per slot: the accumulator column from the slot's start value (exclusive: a row holds what was absorbed before it)
__global__ __launch_bounds__(64) void k_ed_bind_rows(const uint64_t* __restrict__ trace, uint32_t n_slots, gl::Ext gamma,
                                                     const gl::Ext* __restrict__ start, uint64_t* __restrict__ out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_slots) return;
    const size_t n = (size_t)n_slots * ed::ROWS, row0 = (size_t)k * ed::ROWS;
    gl::Ext acc = start[k];
    for (int r = 0; r < ed::ROWS; r++) {
        out[row0 + r] = acc.a;
        out[n + row0 + r] = acc.b;
        if ((r & 15) == 15) acc = absorb_limb(acc, gamma, trace, n, row0 + r, 15 - (r >> 4));
    }
}

}  // namespace nlx

using namespace nlx;

extern "C" int32_t nlx_ed25519_bind_round(nlx_ctx* ctx, const uint64_t* trace, uint32_t log_slots, const uint64_t gamma[2],
                                          uint64_t* acc_out, uint64_t total_out[2]) NLX_TRY {
    if (!ctx) return NLX_E_INVAL;
    if (!trace || !gamma || !acc_out || !total_out) return ctx->fail(NLX_E_INVAL, "NULL argument");
    if (log_slots > 16) return ctx->fail(NLX_E_RANGE, "log_slots must be <= 16");
    (void)hipSetDevice(ctx->device);
    const uint32_t n_slots = 1u << log_slots;
    const size_t n = (size_t)n_slots * ed::ROWS;
    Staged tr(ctx, trace, (size_t)ed::N_COLS0 * n * 8, true, false);
    if (tr.status) return tr.status;
    Staged so(ctx, acc_out, 2 * n * 8, false, true);
    if (so.status) return so.status;
    gl::Ext* d_fp = (gl::Ext*)ctx->alloc((size_t)n_slots * sizeof(gl::Ext));
    if (!d_fp) return NLX_E_NOMEM;
    const gl::Ext g{gamma[0] % gl::P, gamma[1] % gl::P};
    hipStream_t st = ctx->stream;
    hipLaunchKernelGGL(k_ed_bind_slot, dim3((n_slots + 63) / 64), dim3(64), 0, st, tr.as<uint64_t>(), n_slots, g, d_fp);
    std::vector<gl::Ext> fp_h(n_slots), start(n_slots);
    int32_t rc = fetch(ctx, fp_h.data(), d_fp, (size_t)n_slots * sizeof(gl::Ext));
    if (!rc) {
        // the slots' start values: acc_(s+1) = acc_s gamma^96 + fp_s (a few thousand extension multiplications, on the host)
        const gl::Ext g_slot = gl::pow(g, 16 * ED_BOUND);
        gl::Ext acc{0, 0};
        for (uint32_t k = 0; k < n_slots; k++) {
            start[k] = acc;
            acc = gl::add(gl::mul(acc, g_slot), fp_h[k]);
        }
        total_out[0] = acc.a;
        total_out[1] = acc.b;
        hipError_t e = hipMemcpyAsync(d_fp, start.data(), (size_t)n_slots * sizeof(gl::Ext), hipMemcpyHostToDevice, st);
        if (e != hipSuccess) rc = ctx->hip_fail(e, "hipMemcpyAsync");
    }
    if (!rc) {
        hipLaunchKernelGGL(k_ed_bind_rows, dim3((n_slots + 63) / 64), dim3(64), 0, st, tr.as<uint64_t>(), n_slots, g, d_fp,
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


extern "C" int32_t nlx_ed25519_trace(nlx_ctx* ctx, const uint64_t* slots, uint32_t log_slots, uint64_t* trace_out) NLX_TRY {
    if (!ctx) return NLX_E_INVAL;
    if (!slots || !trace_out) return ctx->fail(NLX_E_INVAL, "NULL argument");
    if (log_slots > 16) return ctx->fail(NLX_E_RANGE, "log_slots must be <= 16");
    (void)hipSetDevice(ctx->device);
    const uint32_t n_slots = 1u << log_slots;
    const size_t n = (size_t)n_slots * ed::ROWS;
    Staged sw(ctx, slots, (size_t)n_slots * ed::SLOT_WORDS * 8, true, false);
    if (sw.status) return sw.status;
    Staged st(ctx, trace_out, (size_t)ed::N_COLS0 * n * 8, false, true);
    if (st.status) return st.status;
    const size_t bytes = (size_t)n_slots * sizeof(ed::Slot) + (n + n_slots) * sizeof(ed::Point) + 16;
    char* d = (char*)ctx->alloc(bytes);
    if (!d) return NLX_E_NOMEM;
    ed::Slot* d_slots = (ed::Slot*)d;
    ed::Point* d_in = (ed::Point*)(d + (size_t)n_slots * sizeof(ed::Slot));
    ed::Point* d_fin = d_in + n;
    uint32_t* d_bad = (uint32_t*)(d_fin + n_slots);
    uint32_t bad = 0xFFFFFFFFu;
    hipError_t e0 = hipMemcpyAsync(d_bad, &bad, 4, hipMemcpyHostToDevice, ctx->stream);
    if (e0 != hipSuccess) { ctx->release(d); return ctx->hip_fail(e0, "hipMemcpyAsync"); }
    hipLaunchKernelGGL(k_ed_scan, dim3((n_slots + 63) / 64), dim3(64), 0, ctx->stream, sw.as<uint64_t>(), n_slots, d_slots, d_in,
                       d_fin);
    hipLaunchKernelGGL(k_ed_rows, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, d_slots, d_in, d_fin, n_slots,
                       st.as<uint64_t>(), d_bad);
    e0 = hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream);
    int32_t rc = st.finish();
    hipError_t e = hipStreamSynchronize(ctx->stream);
    ctx->release(d);
    if (!rc && e != hipSuccess) rc = ctx->hip_fail(e, "hipStreamSynchronize");
    hipError_t le = hipGetLastError();
    if (!rc && le != hipSuccess) rc = ctx->hip_fail(le, "kernel launch");
    if (!rc && e0 != hipSuccess) rc = ctx->hip_fail(e0, "hipMemcpyAsync");
    if (!rc && bad != 0xFFFFFFFFu)
        rc = ctx->fail(NLX_E_INVAL, "slot %u: the statement is false (the signature does not verify, A / R is not on the curve, "
                                    "S >= L or a coordinate >= p); the trace was written but cannot satisfy the AIR", bad);
    return rc;
} NLX_CATCH(ctx)
