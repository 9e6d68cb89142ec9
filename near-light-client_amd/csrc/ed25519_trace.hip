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

// ---- the scan on FOUR lanes per slot ------------------------------------------------------------------------------------
// A row of the double-and-add chain is fifteen field products in four dependent levels (4 squares; 4 products of the
// doubling; 4 products with the addend; 3 products of the addition): k_ed_scan walks them one after the other on one lane per
// slot - 128 lanes for the Sync step's 128 slots, 3.7 ms of latency.  Here a quad of lanes owns a slot: lane j computes
// product j of every level (operands picked by lane-index selects, no divergent control flow), the four results travel by
// DPP quad broadcasts (ten 26-bit limbs each), and lanes 0..2 canonicalise and store one coordinate of the row's input point
// each.  The state is replicated in the quad; values are the same residues as k_ed_scan's (canonical after freeze), so the
// trace is bit-identical.
namespace quad {
struct F10 { int32_t l[10]; };   // fe::Fe's limbs after a carry fit 32 bits (|l| <= 2^25 + small)

template <int SRC>
__device__ __forceinline__ F10 bcast(const F10& v) {   // lane SRC of the quad's value, in every lane of the quad
    F10 r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.l[i] = __builtin_amdgcn_mov_dpp(v.l[i], SRC * 0x55, 0xF, 0xF, true);
    return r;
}
__device__ __forceinline__ F10 pick(uint32_t j, const F10& a, const F10& b, const F10& c, const F10& d) {
    F10 r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.l[i] = j == 0 ? a.l[i] : (j == 1 ? b.l[i] : (j == 2 ? c.l[i] : d.l[i]));
    return r;
}
__device__ __forceinline__ F10 add(const F10& a, const F10& b) { F10 r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.l[i] = a.l[i] + b.l[i]; return r; }
__device__ __forceinline__ F10 sub(const F10& a, const F10& b) { F10 r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.l[i] = a.l[i] - b.l[i]; return r; }
__device__ __forceinline__ fe::Fe wide(const F10& a) { fe::Fe r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.l[i] = a.l[i]; return r; }
__device__ __forceinline__ F10 narrow(const fe::Fe& a) { F10 r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.l[i] = (int32_t)a.l[i]; return r; }
// operands: sums of at most four carried values (|l| < 2^28, fe::mul's bound); result carried
__device__ __forceinline__ F10 mul(const F10& a, const F10& b) { return narrow(fe::mul(wide(a), wide(b))); }
}  // namespace quad

constexpr uint32_t SCAN4_SLOTS_PER_BLOCK = 16;   // 64 lanes
__global__ __launch_bounds__(64) void k_ed_scan4(const uint64_t* __restrict__ words, uint32_t n_slots, ed::Slot* __restrict__ slots,
                                                 ed::Point* __restrict__ in, ed::Point* __restrict__ fin) {
    using quad::F10;
    __shared__ int32_t tab[64][4][10];   // this lane's column of the slot's addend table: [sel][limb]
    const uint32_t k_raw = blockIdx.x * SCAN4_SLOTS_PER_BLOCK + (threadIdx.x >> 2), j = threadIdx.x & 3;
    const bool live = k_raw < n_slots;               // spare quads redo the last slot and store nothing (whole quads: DPP needs its lanes)
    const uint32_t k = live ? k_raw : n_slots - 1;
    ed::Slot s;
    ed::slot_from_words(words + (size_t)k * ed::SLOT_WORDS, s);
    if (live && j == 0) slots[k] = s;
    {
        ed::FastSlot fs;
        ed::fast_slot(s, fs);
        for (int sel = 0; sel < 4; sel++) {
            const fe::Fe& v = j == 0 ? fs.ymx[sel] : (j == 1 ? fs.ypx[sel] : (j == 2 ? fs.t2d[sel] : fs.z2[sel]));
            for (int i = 0; i < 10; i++) tab[threadIdx.x][sel][i] = (int32_t)v.l[i];
        }
    }
    F10 x, y, z;
#pragma unroll
    for (int i = 0; i < 10; i++) { x.l[i] = 0; y.l[i] = i == 0; z.l[i] = i == 0; }
    uint32_t* out_words = reinterpret_cast<uint32_t*>(in + (size_t)k * ed::ROWS) + 16 * j;   // coordinate j of the slot's first row
#pragma unroll 1
    for (int r = 0; r <= ed::ROWS; r++) {
        // the row's input point (after the last row: the slot's final point): lanes 0..2 freeze and store one coordinate each
        if (live && j < 3) {
            uint32_t c16[16];
            fe::freeze(quad::wide(quad::pick(j, x, y, z, z)), c16);
            uint32_t* dst = r < ed::ROWS ? out_words + (size_t)r * 48 : reinterpret_cast<uint32_t*>(fin + k) + 16 * j;
#pragma unroll
            for (int i = 0; i < 16; i += 4) *reinterpret_cast<uint4*>(dst + i) = make_uint4(c16[i], c16[i + 1], c16[i + 2], c16[i + 3]);
        }
        if (r == ed::ROWS) break;
        const int bit = ed::ROWS - 1 - r;
        const uint32_t sel = ((s.sw[bit >> 4] >> (bit & 15)) & 1) + 2 * ((s.hw[bit >> 4] >> (bit & 15)) & 1);
        // level 1: a = x^2, b = y^2, zz = z^2, e1 = (x + y)^2
        const F10 sq_in = quad::pick(j, x, y, z, quad::add(x, y));
        const F10 sq = quad::mul(sq_in, sq_in);
        const F10 a = quad::bcast<0>(sq), b = quad::bcast<1>(sq), zz = quad::bcast<2>(sq), e1 = quad::bcast<3>(sq);
        const F10 e = quad::sub(quad::sub(e1, a), b), g = quad::sub(b, a), f = quad::sub(g, quad::add(zz, zz)), h = quad::sub(quad::sub(g, b), b);   // h = -(a + b)
        // level 2: x2 = e f, y2 = g h, t2 = e h, z2 = f g
        const F10 m2 = quad::mul(quad::pick(j, e, g, e, f), quad::pick(j, f, h, h, g));
        const F10 x2 = quad::bcast<0>(m2), y2 = quad::bcast<1>(m2);
        // level 3: the addend (its four products with the doubled point), sel = 0 being the neutral element's (1, 1, 0, 2)
        F10 t;
#pragma unroll
        for (int i = 0; i < 10; i++) t.l[i] = tab[threadIdx.x][sel][i];
        const F10 m3 = quad::mul(quad::pick(j, quad::sub(y2, x2), quad::add(y2, x2), m2, m2), t);
        const F10 pa = quad::bcast<0>(m3), pb = quad::bcast<1>(m3), pc = quad::bcast<2>(m3), pd = quad::bcast<3>(m3);
        const F10 e_ = quad::sub(pb, pa), f_ = quad::sub(pd, pc), g_ = quad::add(pd, pc), h_ = quad::add(pb, pa);
        // level 4: x = e_ f_, y = g_ h_, z = f_ g_ (lane 3 repeats z)
        const F10 m4 = quad::mul(quad::pick(j, e_, g_, f_, f_), quad::pick(j, f_, h_, g_, g_));
        x = quad::bcast<0>(m4);
        y = quad::bcast<1>(m4);
        z = quad::bcast<2>(m4);
    }
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
    // four lanes per slot (k_ed_scan4); k_ed_scan, one lane per slot, is the same walk written sequentially (kept: the reference
    // the quad version is checked against in tests/native and the form the row emitter's recomputation mirrors)
    hipLaunchKernelGGL(k_ed_scan4, dim3((n_slots + SCAN4_SLOTS_PER_BLOCK - 1) / SCAN4_SLOTS_PER_BLOCK), dim3(64), 0, ctx->stream,
                       sw.as<uint64_t>(), n_slots, d_slots, d_in, d_fin);
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
