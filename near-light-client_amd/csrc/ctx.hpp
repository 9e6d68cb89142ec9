// Per-device context: stream, caching device allocator, twiddle / coset tables, error slot.
// One context per HIP device and per host thread (SURVEY.md §8b threading contract).
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <exception>
#include <map>
#include <new>
#include <string>
#include <vector>
#include "../../include/nlx.h"
#include "launch.hpp"

struct nlx_ctx {
    int device = -1;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;  // stream in use (own_stream or a borrowed one)
    std::string err;

    // caching allocator: freed blocks are kept by size and reused (hipMalloc/hipFree synchronise)
    std::multimap<size_t, void*> free_blocks;
    std::map<void*, size_t> live_blocks;
    size_t bytes_reserved = 0;

    // twiddle tables, built lazily up to the largest transform seen
    nlx::NttTables tables{};
    std::vector<void*> table_allocs;
    // coset scale tables keyed by (log_n << 8 | rate_bits)
    std::map<uint32_t, uint64_t*> coset_scale;
    // shift^i (natural order) tables for nlx_ntt_batch keyed by (log_n, shift)
    std::map<std::pair<uint32_t, uint64_t>, uint64_t*> nat_scale;
    std::vector<std::pair<uint32_t, uint64_t>> nat_scale_order;   // oldest first: at most four tables are kept
    // BN254 Fr twiddle table (csrc/bn254.hip) keyed by 2 log_n + inverse: w_n^e for e < n/2
    std::map<uint32_t, void*> bn254_tables;

    // Optional per-kernel device timing (HIP events on `stream` around selected launches); used by
    // bench.py to report the dominant kernel's average duration from inside the timed region.
    struct KernelSample { const char* name; double alg_bytes; double units; hipEvent_t e0, e1; };
    bool kernel_timing = false;
    std::vector<KernelSample> samples;
    std::vector<hipEvent_t> event_pool;
    hipEvent_t get_event();
    void begin_kernel(const char* name, double alg_bytes, double units = 0.0);  // units: work items other than bytes (Poseidon permutations)
    void end_kernel();

    // pinned staging buffer for small device->host reads
    void* pinned = nullptr;
    size_t pinned_bytes = 0;

    int32_t fail(int32_t code, const char* fmt, ...) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
        return code;
    }
    int32_t hip_fail(hipError_t e, const char* what) {
        return fail(e == hipErrorOutOfMemory ? NLX_E_NOMEM : NLX_E_HIP, "%s: %s", what, hipGetErrorString(e));
    }

    void* alloc(size_t bytes);
    void release(void* p);
    void trim();
    int32_t ensure_tables(unsigned log_n);
    int32_t get_coset_scale(unsigned log_n, unsigned rate_bits, const uint64_t** out, bool inverse = false);
    int32_t get_nat_scale(unsigned log_n, uint64_t shift, const uint64_t** out);

};

// ---- the ABI never throws or aborts (include/nlx.h, SURVEY.md §8b: Rust / Go callers map return codes to their own errors) ----
// Host code behind the entry points uses std::vector / map / string / thread; every extern "C" definition is therefore a
// function-try-block:   int32_t nlx_foo(nlx_ctx* ctx, ...) NLX_TRY { ... } NLX_CATCH(ctx)
// std::bad_alloc -> NLX_E_NOMEM, anything else -> NLX_E_INVAL, with nlx_last_error set when the call has a context.
namespace nlx {
extern std::atomic<int> batch_spawn_fault_after;   // ctx.hip
inline void on_exception(nlx_ctx* ctx, const char* what) noexcept {
    if (!ctx) return;
    try {
        ctx->err = what;
    } catch (...) {
    }
}
}  // namespace nlx
#define NLX_TRY try
#define NLX_CATCH_BODY(ctx, ret_nomem, ret_other)                                                          \
    catch (const std::bad_alloc&) {                                                                       \
        nlx::on_exception(ctx, "out of host memory (std::bad_alloc)");                                     \
        return ret_nomem;                                                                                 \
    }                                                                                                     \
    catch (const std::exception& e__) {                                                                   \
        nlx::on_exception(ctx, e__.what());                                                               \
        return ret_other;                                                                                 \
    }                                                                                                     \
    catch (...) {                                                                                         \
        nlx::on_exception(ctx, "unknown C++ exception");                                                  \
        return ret_other;                                                                                 \
    }
#define NLX_CATCH(ctx) NLX_CATCH_BODY(ctx, NLX_E_NOMEM, NLX_E_INVAL)
#define NLX_CATCH_VOID(ctx) NLX_CATCH_BODY(ctx, , )
#define NLX_CATCH_VALUE(ctx, v) NLX_CATCH_BODY(ctx, v, v)

#define NLX_HIP(ctx, call)                                   \
    do {                                                     \
        hipError_t e__ = (call);                             \
        if (e__ != hipSuccess) return (ctx)->hip_fail(e__, #call); \
    } while (0)

namespace nlx {
bool is_device_ptr(const void* p);
// RAII device view of a caller buffer: device pointers are used in place, host pointers are
// staged through a context allocation (copied in on construction if `in`, out on finish()).
struct Staged {
    nlx_ctx* ctx;
    void* user;
    void* dev = nullptr;
    size_t bytes;
    bool owned = false;
    bool out;
    int32_t status = 0;
    Staged(nlx_ctx* c, const void* user_ptr, size_t nbytes, bool copy_in, bool copy_out);
    ~Staged();
    int32_t finish();  // enqueue copy-out (if any); caller synchronises
    template <class T> T* as() { return reinterpret_cast<T*>(dev); }
};
}  // namespace nlx
