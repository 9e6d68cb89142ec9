// Range-check lookups for multi-round AIRs (starkyx commits in rounds so that lookup
// accumulators can depend on verifier challenges; the argument itself - curta's is not in the reference tree - is
// the log-derivative lookup with the challenge in the quadratic extension, see near-light-client_amd/logup.py).
//
// Witness side, all on the device: multiplicities of the looked-up cells (LDS histograms per block, then one
// global add per non-zero bin), then for a challenge alpha the round-1 columns: one helper h = 1/(alpha+v1) + 1/(alpha+v2) per pair of
// lookups, g = m/(alpha+t) and the running sum phi (a two-level additive scan).  One extension inversion per
// eight looked-up cells (Montgomery's trick over a lane's helpers; the inversion is a base-field inversion of the norm).
#include "ctx.hpp"
#include "gl.hpp"
#include "transcript.hpp"

namespace nlx {

// Multiplicities: a histogram of n_lookups * n cells over up to 2^16 bins.  Many cells are far from uniform (the high
// parts of carries, constant columns), so global atomics straight from the cells serialise on a few addresses (154 ms
// for the Ed25519 trace).  Each block instead counts a contiguous chunk of cells in an LDS histogram - one pass per
// 2^14 bins (64 KB of 32-bit counters) - and adds its non-zero bins to the global column once per pass.
constexpr uint32_t COUNT_BINS = 1u << 14;
constexpr uint32_t COUNT_THREADS = 1024;

__global__ __launch_bounds__(COUNT_THREADS) void k_logup_count(const uint64_t* __restrict__ trace, const uint32_t* __restrict__ cols,
                                                               uint32_t log_n, uint32_t n_lookups, uint32_t table_bits, uint32_t period_bits,
                                                               unsigned long long* __restrict__ mult, uint32_t* __restrict__ err) {
    __shared__ uint32_t bins[COUNT_BINS];
    const size_t total = (size_t)n_lookups << log_n;
    const size_t chunk = (total + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * chunk, hi = lo + chunk < total ? lo + chunk : total;
    const uint32_t n_pass = table_bits > 14 ? 1u << (table_bits - 14) : 1u;
    const size_t n_mask = ((size_t)1 << log_n) - 1;
    for (uint32_t pass = 0; pass < n_pass; pass++) {
        for (uint32_t i = threadIdx.x; i < COUNT_BINS; i += COUNT_THREADS) bins[i] = 0;
        __syncthreads();
        for (size_t idx = lo + threadIdx.x; idx < hi; idx += COUNT_THREADS) {
            const uint64_t v = trace[((size_t)cols[idx >> log_n] << log_n) + (idx & n_mask)];
            if (v >> table_bits) {
                if (pass == 0) atomicOr(err, 1u);
            } else if ((uint32_t)(v >> 14) == pass) {
                atomicAdd(&bins[(uint32_t)v & (COUNT_BINS - 1)], 1u);
            }
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < COUNT_BINS; i += COUNT_THREADS) {
            const uint32_t c = bins[i];
            if (c) {
                // value v = pass 2^14 + i lives in table column v >> period_bits, row v mod 2^period_bits
                const uint32_t v = (pass << 14) + i;
                atomicAdd(mult + ((size_t)(v >> period_bits) << log_n) + (v & ((1u << period_bits) - 1)), (unsigned long long)c);
            }
        }
        __syncthreads();
    }
}

struct LogupParams {
    const uint64_t* trace;
    const uint32_t* cols;
    uint64_t* out;  // [2 H + 4][n]
    uint64_t alpha[2];
    uint32_t log_n, n_lookups, table_bits, mult_col, table_cols, period_bits;
};

// One lane per (row, group of HELPERS_PER_LANE helpers).  h = 1/d1 + 1/d2 = (d1 + d2) / (d1 d2), and the denominators
// of a lane's helpers are inverted together (Montgomery's trick): one extension inversion - a base-field inversion
// of the norm, ~75 multiplications - per eight lookups instead of one per lookup.
constexpr uint32_t HELPERS_PER_LANE = 4;

__global__ __launch_bounds__(256) void k_logup_helpers(LogupParams p) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >> p.log_n) return;
    const uint32_t H = (p.n_lookups + 1) / 2, j0 = blockIdx.y * HELPERS_PER_LANE;
    const gl::Ext al{p.alpha[0], p.alpha[1]};
    gl::Ext num[HELPERS_PER_LANE], den[HELPERS_PER_LANE], pre[HELPERS_PER_LANE];
    gl::Ext run{1, 0};
#pragma unroll
    for (uint32_t t = 0; t < HELPERS_PER_LANE; t++) {
        const uint32_t j = j0 + t;
        num[t] = gl::Ext{1, 0};
        den[t] = gl::Ext{1, 0};
        if (j < H) {
            const gl::Ext d1 = gl::add(al, gl::ext(p.trace[((size_t)p.cols[2 * j] << p.log_n) + i]));
            if (2 * j + 1 < p.n_lookups) {
                const gl::Ext d2 = gl::add(al, gl::ext(p.trace[((size_t)p.cols[2 * j + 1] << p.log_n) + i]));
                num[t] = gl::add(d1, d2);
                den[t] = gl::mul(d1, d2);
            } else {
                den[t] = d1;
            }
        }
        pre[t] = run;  // product of the denominators before t
        run = gl::mul(run, den[t]);
    }
    gl::Ext inv = gl::inv(run);
#pragma unroll
    for (int t = HELPERS_PER_LANE - 1; t >= 0; t--) {
        const uint32_t j = j0 + t;
        const gl::Ext h = gl::mul(num[t], gl::mul(inv, pre[t]));  // num / den
        inv = gl::mul(inv, den[t]);
        if (j < H) {
            p.out[((size_t)(2 * j) << p.log_n) + i] = h.a;
            p.out[((size_t)(2 * j + 1) << p.log_n) + i] = h.b;
        }
    }
}

// one lane per row: g_c = m_c / (alpha + t_c) for every table column c, and the row's contribution sum h - sum g into the
// phi columns (scanned next)
__global__ __launch_bounds__(256) void k_logup_rowsum(LogupParams p) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >> p.log_n) return;
    const uint32_t H = (p.n_lookups + 1) / 2;
    gl::Ext acc{0, 0};
    for (uint32_t j = 0; j < H; j++)
        acc = gl::add(acc, gl::Ext{p.out[((size_t)(2 * j) << p.log_n) + i], p.out[((size_t)(2 * j + 1) << p.log_n) + i]});
    uint64_t* gcol = p.out + ((size_t)(2 * H) << p.log_n);
    for (uint32_t c = 0; c < p.table_cols; c++) {
        const uint64_t t = ((uint64_t)c << p.period_bits) + (uint64_t)(i & (((size_t)1 << p.period_bits) - 1));
        const uint64_t m = p.trace[((size_t)(p.mult_col + c) << p.log_n) + i];
        const gl::Ext g = gl::mul(gl::inv(gl::add(gl::Ext{p.alpha[0], p.alpha[1]}, gl::ext(t))), m);
        gcol[((size_t)(2 * c) << p.log_n) + i] = g.a;
        gcol[((size_t)(2 * c + 1) << p.log_n) + i] = g.b;
        acc = gl::sub(acc, g);
    }
    gcol[((size_t)(2 * p.table_cols) << p.log_n) + i] = acc.a;
    gcol[((size_t)(2 * p.table_cols + 1) << p.log_n) + i] = acc.b;
}

// ---- additive exclusive scan in F_p (blockIdx.y = column) ----
constexpr unsigned ADD_SCAN_PER_THREAD = 8;
constexpr unsigned ADD_SCAN_ELEMS = 256 * ADD_SCAN_PER_THREAD;

__device__ __forceinline__ uint64_t shfl_up64(uint64_t v, int off) {
    const uint32_t lo = __shfl_up((uint32_t)v, off, 64), hi = __shfl_up((uint32_t)(v >> 32), off, 64);
    return ((uint64_t)hi << 32) | lo;
}

// in place: data[col][i] <- exclusive prefix sum within the block; totals[col][block] <- block sum
__global__ __launch_bounds__(256) void k_scan_add_local(uint64_t* __restrict__ data, size_t stride, size_t count,
                                                        uint64_t* __restrict__ totals, size_t totals_stride) {
    __shared__ uint64_t wave_tot[4];
    uint64_t* col = data + (size_t)blockIdx.y * stride;
    const size_t base = (size_t)blockIdx.x * ADD_SCAN_ELEMS + (size_t)threadIdx.x * ADD_SCAN_PER_THREAD;
    uint64_t v[ADD_SCAN_PER_THREAD];
    uint64_t sum = 0;
#pragma unroll
    for (unsigned k = 0; k < ADD_SCAN_PER_THREAD; k++) {
        v[k] = base + k < count ? col[base + k] : 0;
        sum = gl::add(sum, v[k]);
    }
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint64_t o = shfl_up64(incl, off);
        if (lane >= (unsigned)off) incl = gl::add(incl, o);
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint64_t run = 0;
    for (unsigned w2 = 0; w2 < wave; w2++) run = gl::add(run, wave_tot[w2]);
    const uint64_t prev = shfl_up64(incl, 1);
    if (lane) run = gl::add(run, prev);
#pragma unroll
    for (unsigned k = 0; k < ADD_SCAN_PER_THREAD; k++) {
        if (base + k < count) col[base + k] = run;
        run = gl::add(run, v[k]);
    }
    if (threadIdx.x == 255) totals[(size_t)blockIdx.y * totals_stride + blockIdx.x] = run;
}

__global__ __launch_bounds__(256) void k_scan_add_apply(uint64_t* __restrict__ data, size_t stride, size_t count,
                                                        const uint64_t* __restrict__ prefix, size_t prefix_stride) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint64_t* col = data + (size_t)blockIdx.y * stride;
    col[i] = gl::add(col[i], prefix[(size_t)blockIdx.y * prefix_stride + i / ADD_SCAN_ELEMS]);
}

// exclusive scan of `ncol` columns of `count` elements (column stride `stride`); scratch >= add_scan_scratch_words
static void launch_add_scan(hipStream_t st, uint64_t* data, size_t stride, size_t count, uint32_t ncol, uint64_t* scratch) {
    const size_t nb = (count + ADD_SCAN_ELEMS - 1) / ADD_SCAN_ELEMS;
    uint64_t* tot = scratch;  // ncol x nb
    hipLaunchKernelGGL(k_scan_add_local, dim3((unsigned)nb, ncol), dim3(256), 0, st, data, stride, count, tot, nb);
    if (nb > 1) {
        launch_add_scan(st, tot, nb, nb, ncol, scratch + (size_t)ncol * nb);
        hipLaunchKernelGGL(k_scan_add_apply, dim3((unsigned)((count + 255) / 256), ncol), dim3(256), 0, st, data, stride, count,
                           tot, nb);
    }
}
static size_t add_scan_scratch_words(size_t count, uint32_t ncol) {
    size_t words = 0;
    while (true) {
        const size_t nb = (count + ADD_SCAN_ELEMS - 1) / ADD_SCAN_ELEMS;
        words += (size_t)ncol * nb;
        if (nb <= 1) break;
        count = nb;
    }
    return words + 16;
}

}  // namespace nlx

using namespace nlx;

static int32_t logup_check(nlx_ctx* ctx, const void* trace, uint32_t n_cols, uint32_t log_n, const uint32_t* cols,
                           uint32_t n_lookups, uint32_t table_bits, uint32_t table_cols, uint32_t mult_col, uint32_t* period_bits) {
    if (!trace || !cols) return ctx->fail(NLX_E_INVAL, "NULL argument");
    uint32_t lk = 0;
    while ((1u << lk) < table_cols) lk++;
    if (table_cols == 0 || (1u << lk) != table_cols || lk > 8) return ctx->fail(NLX_E_RANGE, "table_cols must be a power of two <= 256");
    if (log_n < 1 || log_n > 26 || table_bits < 1 || table_bits > 16 || lk > table_bits || table_bits - lk > log_n)
        return ctx->fail(NLX_E_RANGE, "need 1 <= table_bits <= 16, 2^table_bits / table_cols <= 2^log_n and log_n <= 26");
    if (n_lookups == 0 || n_lookups > 8192 || n_cols > 8192 || mult_col + table_cols > n_cols)
        return ctx->fail(NLX_E_RANGE, "lookup count / column index out of range");
    for (uint32_t l = 0; l < n_lookups; l++)
        if (cols[l] >= n_cols || (cols[l] >= mult_col && cols[l] < mult_col + table_cols))
            return ctx->fail(NLX_E_RANGE, "lookup %u: column out of range", l);
    *period_bits = table_bits - lk;
    return NLX_OK;
}

extern "C" int32_t nlx_logup_multiplicities(nlx_ctx* ctx, uint64_t* trace, uint32_t n_cols, uint32_t log_n, const uint32_t* cols,
                                            uint32_t n_lookups, uint32_t table_bits, uint32_t table_cols, uint32_t mult_col) NLX_TRY {
    if (!ctx) return NLX_E_INVAL;
    uint32_t pb = 0;
    int32_t rc = logup_check(ctx, trace, n_cols, log_n, cols, n_lookups, table_bits, table_cols, mult_col, &pb);
    if (rc) return rc;
    (void)hipSetDevice(ctx->device);
    const size_t n = (size_t)1 << log_n;
    Staged tr(ctx, trace, (size_t)n_cols * n * 8, true, true);
    if (tr.status) return tr.status;
    uint32_t* d_cols = (uint32_t*)ctx->alloc((size_t)n_lookups * 4 + 16);
    if (!d_cols) return NLX_E_NOMEM;
    uint32_t* d_err = d_cols + n_lookups;
    hipStream_t st = ctx->stream;
    uint64_t* mult = tr.as<uint64_t>() + (size_t)mult_col * n;
    hipError_t e = hipMemcpyAsync(d_cols, cols, (size_t)n_lookups * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(d_err, 0, 4, st);
    if (e == hipSuccess) e = hipMemsetAsync(mult, 0, n * 8 * table_cols, st);
    uint32_t err = 0;
    if (e == hipSuccess) {
        const size_t total = (size_t)n_lookups << log_n;
        const unsigned blocks = (unsigned)(total < ((size_t)512 << 12) ? (total + 4095) / 4096 : 512);  // >= 4096 cells per block
        hipLaunchKernelGGL(k_logup_count, dim3(blocks), dim3(COUNT_THREADS), 0, st, tr.as<uint64_t>(), d_cols, log_n, n_lookups,
                           table_bits, pb, (unsigned long long*)mult, d_err);
        e = hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) rc = tr.finish();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    ctx->release(d_cols);
    if (e != hipSuccess) return ctx->hip_fail(e, "nlx_logup_multiplicities");
    if (!rc && err) rc = ctx->fail(NLX_E_RANGE, "a looked-up cell is outside the table [0, 2^%u)", table_bits);
    return rc;
} NLX_CATCH(ctx)

extern "C" uint32_t nlx_logup_round_cols(uint32_t n_lookups, uint32_t table_cols) NLX_TRY { return 2 * ((n_lookups + 1) / 2) + 2 * table_cols + 2; } NLX_CATCH_VALUE(nullptr, 0)

extern "C" int32_t nlx_logup_round(nlx_ctx* ctx, const uint64_t* trace, uint32_t n_cols, uint32_t log_n, const uint32_t* cols,
                                   uint32_t n_lookups, uint32_t table_bits, uint32_t table_cols, uint32_t mult_col,
                                   const uint64_t alpha[2], uint64_t* out) NLX_TRY {
    if (!ctx) return NLX_E_INVAL;
    uint32_t pb = 0;
    int32_t rc = logup_check(ctx, trace, n_cols, log_n, cols, n_lookups, table_bits, table_cols, mult_col, &pb);
    if (rc) return rc;
    if (!alpha || !out) return ctx->fail(NLX_E_INVAL, "NULL argument");
    (void)hipSetDevice(ctx->device);
    const size_t n = (size_t)1 << log_n;
    const uint32_t H = (n_lookups + 1) / 2, n_out = 2 * H + 2 * table_cols + 2;
    Staged tr(ctx, trace, (size_t)n_cols * n * 8, true, false);
    if (tr.status) return tr.status;
    Staged so(ctx, out, (size_t)n_out * n * 8, false, true);
    if (so.status) return so.status;
    const size_t scan_words = add_scan_scratch_words(n, 2);
    uint64_t* d_scratch = (uint64_t*)ctx->alloc(scan_words * 8 + (size_t)n_lookups * 4);
    if (!d_scratch) return NLX_E_NOMEM;
    uint32_t* d_cols = (uint32_t*)(d_scratch + scan_words);
    hipStream_t st = ctx->stream;
    hipError_t e = hipMemcpyAsync(d_cols, cols, (size_t)n_lookups * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        LogupParams p{};
        p.trace = tr.as<uint64_t>(); p.cols = d_cols; p.out = so.as<uint64_t>();
        p.alpha[0] = alpha[0] % gl::P; p.alpha[1] = alpha[1] % gl::P;
        p.log_n = log_n; p.n_lookups = n_lookups; p.table_bits = table_bits; p.mult_col = mult_col;
        p.table_cols = table_cols; p.period_bits = pb;
        const unsigned blocks = (unsigned)((n + 255) / 256);
        hipLaunchKernelGGL(k_logup_helpers, dim3(blocks, (H + HELPERS_PER_LANE - 1) / HELPERS_PER_LANE), dim3(256), 0, st, p);
        hipLaunchKernelGGL(k_logup_rowsum, dim3(blocks), dim3(256), 0, st, p);
        launch_add_scan(st, p.out + (size_t)(2 * H + 2 * table_cols) * n, n, n, 2, d_scratch);
        rc = so.finish();
        e = hipStreamSynchronize(st);
    }
    ctx->release(d_scratch);
    if (e != hipSuccess) return ctx->hip_fail(e, "nlx_logup_round");
    hipError_t le = hipGetLastError();
    if (!rc && le != hipSuccess) rc = ctx->hip_fail(le, "kernel launch");
    return rc;
} NLX_CATCH(ctx)
