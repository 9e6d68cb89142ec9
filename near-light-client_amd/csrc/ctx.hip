// Context implementation: device selection (gfx950 only, no CPU fallback), caching allocator,
// lazily built twiddle / coset-scale tables.
#include "ctx.hpp"
#include <stdexcept>
#include <cstring>
#include <new>
#include "gl.hpp"

void* nlx_ctx::alloc(size_t bytes) {
    if (bytes == 0) bytes = 256;
    bytes = (bytes + 255) & ~(size_t)255;
    auto it = free_blocks.lower_bound(bytes);
    // reuse a cached block if it wastes < 25 %
    if (it != free_blocks.end() && it->first <= bytes + bytes / 4) {
        void* p = it->second;
        live_blocks[p] = it->first;
        free_blocks.erase(it);
        return p;
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        trim();
        e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            hip_fail(e, "hipMalloc");
            return nullptr;
        }
    }
    bytes_reserved += bytes;
    live_blocks[p] = bytes;
    return p;
}

void nlx_ctx::release(void* p) {
    if (!p) return;
    auto it = live_blocks.find(p);
    if (it == live_blocks.end()) return;
    free_blocks.emplace(it->second, p);
    live_blocks.erase(it);
}

void nlx_ctx::trim() {
    if (free_blocks.empty()) return;
    (void)hipStreamSynchronize(stream);
    for (auto& kv : free_blocks) {
        (void)hipFree(kv.second);
        bytes_reserved -= kv.first;
    }
    free_blocks.clear();
}

int32_t nlx_ctx::ensure_tables(unsigned log_n) {
    if (log_n > 32) return fail(NLX_E_RANGE, "log_n %u exceeds the field's two-adicity (32)", log_n);
    for (unsigned k = tables.max_log + 1; k <= log_n; k++) {
        size_t half = (size_t)1 << (k - 1);
        uint64_t* f = (uint64_t*)alloc(half * 8);
        uint64_t* i = f ? (uint64_t*)alloc(half * 8) : nullptr;
        if (!f || !i) {
            release(f);   // a half-built level is not kept: the table stays complete up to max_log
            return NLX_E_NOMEM;
        }
        uint64_t w = gl::root_of_unity(k);
        nlx::launch_fill_powers(stream, f, half, w, 1);
        nlx::launch_fill_powers(stream, i, half, gl::inv(w), 1);
        tables.fwd[k] = f;
        tables.inv[k] = i;
        tables.max_log = k;
    }
    return NLX_OK;
}

int32_t nlx_ctx::get_coset_scale(unsigned log_n, unsigned rate_bits, const uint64_t** out, bool inverse) {
    uint32_t key = (log_n << 8) | rate_bits | (inverse ? 0x80000000u : 0u);
    auto it = coset_scale.find(key);
    if (it == coset_scale.end()) {
        uint64_t* t = (uint64_t*)alloc(((size_t)8 << (log_n + rate_bits)));
        if (!t) return NLX_E_NOMEM;
        nlx::launch_fill_coset_scale_br(stream, t, log_n, rate_bits, gl::GEN, inverse);
        it = coset_scale.emplace(key, t).first;
    }
    *out = it->second;
    return NLX_OK;
}

int32_t nlx_ctx::get_nat_scale(unsigned log_n, uint64_t shift, const uint64_t** out) {
    // a bounded cache (the four most recently built tables): a caller that walks random coset shifts at 2^24 points would
    // otherwise grow the context by 128 MB per call.  An evicted table goes back to the context's allocator; work already
    // queued on the stream that reads it is ordered before any reuse of the block (one stream per context).
    auto key = std::make_pair((uint32_t)log_n, shift);
    auto it = nat_scale.find(key);
    if (it == nat_scale.end()) {
        while (nat_scale_order.size() >= 4) {
            auto old = nat_scale.find(nat_scale_order.front());
            nat_scale_order.erase(nat_scale_order.begin());
            if (old != nat_scale.end()) {
                release(old->second);
                nat_scale.erase(old);
            }
        }
        uint64_t* t = (uint64_t*)alloc((size_t)8 << log_n);
        if (!t) return NLX_E_NOMEM;
        nlx::launch_fill_powers(stream, t, (size_t)1 << log_n, shift, 1);
        it = nat_scale.emplace(key, t).first;
        nat_scale_order.push_back(key);
    }
    *out = it->second;
    return NLX_OK;
}

hipEvent_t nlx_ctx::get_event() {
    if (!event_pool.empty()) {
        hipEvent_t e = event_pool.back();
        event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
void nlx_ctx::begin_kernel(const char* name, double alg_bytes, double units) {
    if (!kernel_timing) return;
    KernelSample ks{name, alg_bytes, units, get_event(), get_event()};
    (void)hipEventRecord(ks.e0, stream);
    samples.push_back(ks);
}
void nlx_ctx::end_kernel() {
    if (!kernel_timing || samples.empty()) return;
    (void)hipEventRecord(samples.back().e1, stream);
}

namespace nlx {
std::atomic<int> batch_spawn_fault_after{-1};   // fault injection for nlx_batch_prove (prover.hip), armed by nlx_abi_selftest

bool is_device_ptr(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // plain host memory: not an error for us
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

Staged::Staged(nlx_ctx* c, const void* user_ptr, size_t nbytes, bool copy_in, bool copy_out)
    : ctx(c), user(const_cast<void*>(user_ptr)), bytes(nbytes), out(copy_out) {
    if (!user_ptr || nbytes == 0) {
        dev = user;
        return;
    }
    if (is_device_ptr(user_ptr)) {
        dev = user;
        return;
    }
    dev = ctx->alloc(nbytes);
    if (!dev) {
        status = NLX_E_NOMEM;
        return;
    }
    owned = true;
    if (copy_in) {
        hipError_t e = hipMemcpyAsync(dev, user, nbytes, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) status = ctx->hip_fail(e, "hipMemcpyAsync(H2D)");
    }
}

int32_t Staged::finish() {
    if (owned && out && status == 0) {
        hipError_t e = hipMemcpyAsync(user, dev, bytes, hipMemcpyDeviceToHost, ctx->stream);
        if (e != hipSuccess) status = ctx->hip_fail(e, "hipMemcpyAsync(D2H)");
    }
    return status;
}

Staged::~Staged() {
    if (owned) {
        // the block may still be in use by queued work on ctx->stream; the allocator only hands
        // it to later work on the same stream, which is ordered after it.
        ctx->release(dev);
    }
}

}  // namespace nlx

extern "C" {

uint32_t nlx_version(void) NLX_TRY { return (0u << 16) | 3u; } NLX_CATCH_VALUE(nullptr, 0)

void nlx_field_generators(uint64_t out[2]) NLX_TRY {
    out[0] = gl::GEN;
    out[1] = gl::POW2_GEN;
} NLX_CATCH_VOID(nullptr)

int32_t nlx_abi_selftest(int32_t kind) NLX_TRY {
    if (kind == 0) {
        std::vector<uint64_t> v;
        v.resize(v.max_size() / 2);   // std::bad_alloc (or std::length_error on a platform whose max_size is smaller)
        return v[v.size() - 1] != 0 ? NLX_E_HIP : NLX_OK;   // not reached
    }
    if (kind == 1) throw std::runtime_error("nlx_abi_selftest");
    if (kind == 2) throw 42;
    if (kind >= 3 && kind <= 5) {   // fault injection for nlx_batch_prove's thread start-up: 3 = the second worker fails, 4 = the first, 5 = off
        nlx::batch_spawn_fault_after = kind == 3 ? 1 : kind == 4 ? 0 : -1;
        return NLX_OK;
    }
    return NLX_E_RANGE;
} catch (const std::length_error&) {
    return NLX_E_NOMEM;
} NLX_CATCH(nullptr)

const char* nlx_strerror(int32_t code) NLX_TRY {
    switch (code) {
        case NLX_OK: return "ok";
        case NLX_E_INVAL: return "invalid argument";
        case NLX_E_NOMEM: return "out of device memory";
        case NLX_E_HIP: return "HIP runtime error";
        case NLX_E_RANGE: return "argument out of range";
        case NLX_E_UNSUPPORTED: return "unsupported";
        default: return "unknown error";
    }
} NLX_CATCH_VALUE(nullptr, "exception inside the library")

int32_t nlx_ctx_create(int device, nlx_ctx** out) NLX_TRY {
    if (!out) return NLX_E_INVAL;
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return NLX_E_HIP;  // no CPU fallback by design
    }
    if (device < 0 || device >= count) return NLX_E_RANGE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return NLX_E_HIP;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return NLX_E_UNSUPPORTED;  // kernels are built for gfx950 only
    if (hipSetDevice(device) != hipSuccess) return NLX_E_HIP;
    nlx_ctx* c = new (std::nothrow) nlx_ctx();
    if (!c) return NLX_E_NOMEM;
    c->device = device;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return NLX_E_HIP;
    }
    c->stream = c->own_stream;
    *out = c;
    return NLX_OK;
} NLX_CATCH(nullptr)

int32_t nlx_ctx_set_priority(nlx_ctx* c, int high) NLX_TRY {
    if (!c) return NLX_E_INVAL;
    (void)hipSetDevice(c->device);
    if (c->stream != c->own_stream) return c->fail(NLX_E_INVAL, "the context runs on a caller-provided stream (nlx_ctx_set_stream)");
    NLX_HIP(c, hipStreamSynchronize(c->own_stream));
    int least = 0, greatest = 0;
    NLX_HIP(c, hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t s = nullptr;
    NLX_HIP(c, hipStreamCreateWithPriority(&s, hipStreamNonBlocking, high ? greatest : least));
    (void)hipStreamDestroy(c->own_stream);
    c->own_stream = c->stream = s;
    return NLX_OK;
} NLX_CATCH(c)

int32_t nlx_ctx_set_cu_mask(nlx_ctx* c, const uint32_t* mask, uint32_t n_words) NLX_TRY {
    if (!c) return NLX_E_INVAL;
    if (!mask || n_words == 0 || n_words > 32) return c->fail(NLX_E_INVAL, "CU mask: 1..32 words");
    uint32_t any = 0;
    for (uint32_t i = 0; i < n_words; i++) any |= mask[i];
    if (!any) return c->fail(NLX_E_INVAL, "CU mask selects no compute unit");
    (void)hipSetDevice(c->device);
    if (c->stream != c->own_stream) return c->fail(NLX_E_INVAL, "the context runs on a caller-provided stream (nlx_ctx_set_stream)");
    NLX_HIP(c, hipStreamSynchronize(c->own_stream));
    hipStream_t s = nullptr;
    NLX_HIP(c, hipExtStreamCreateWithCUMask(&s, n_words, mask));
    (void)hipStreamDestroy(c->own_stream);
    c->own_stream = c->stream = s;
    return NLX_OK;
} NLX_CATCH(c)

void nlx_ctx_destroy(nlx_ctx* c) NLX_TRY {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto& kv : c->live_blocks) (void)hipFree(kv.first);
    for (auto& kv : c->free_blocks) (void)hipFree(kv.second);
    if (c->pinned) (void)hipHostFree(c->pinned);
    for (auto& ks : c->samples) { (void)hipEventDestroy(ks.e0); (void)hipEventDestroy(ks.e1); }
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
} NLX_CATCH_VOID(nullptr)

const char* nlx_last_error(const nlx_ctx* c) NLX_TRY { return c ? c->err.c_str() : "null context"; } NLX_CATCH_VALUE(nullptr, "exception inside the library")

int32_t nlx_ctx_set_stream(nlx_ctx* c, void* hip_stream) NLX_TRY {
    if (!c) return NLX_E_INVAL;
    (void)hipSetDevice(c->device);
    NLX_HIP(c, hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return NLX_OK;
} NLX_CATCH(c)

int32_t nlx_ctx_kernel_timing(nlx_ctx* c, int enable) NLX_TRY {
    if (!c) return NLX_E_INVAL;
    (void)hipSetDevice(c->device);
    NLX_HIP(c, hipStreamSynchronize(c->stream));
    for (auto& ks : c->samples) {
        c->event_pool.push_back(ks.e0);
        c->event_pool.push_back(ks.e1);
    }
    c->samples.clear();
    c->kernel_timing = enable != 0;
    return NLX_OK;
} NLX_CATCH(c)

int32_t nlx_ctx_kernel_stats(nlx_ctx* c, const char* name, uint64_t* calls, double* total_ms, double* alg_bytes) NLX_TRY {
    if (!c || !name) return NLX_E_INVAL;
    (void)hipSetDevice(c->device);
    NLX_HIP(c, hipStreamSynchronize(c->stream));
    uint64_t n = 0;
    double ms = 0, bytes = 0;
    for (auto& ks : c->samples) {
        if (strcmp(ks.name, name) != 0) continue;
        float t = 0;
        if (hipEventElapsedTime(&t, ks.e0, ks.e1) != hipSuccess) continue;
        n++;
        ms += t;
        bytes += ks.alg_bytes;
    }
    if (calls) *calls = n;
    if (total_ms) *total_ms = ms;
    if (alg_bytes) *alg_bytes = bytes;
    return NLX_OK;
} NLX_CATCH(c)

int32_t nlx_ctx_kernel_units(nlx_ctx* c, const char* name, double* units) NLX_TRY {
    if (!c || !name || !units) return NLX_E_INVAL;
    double u = 0;
    for (auto& ks : c->samples)
        if (strcmp(ks.name, name) == 0) u += ks.units;
    *units = u;
    return NLX_OK;
} NLX_CATCH(c)

// ---- nlx_buf: device buffers for callers without their own HIP bindings ----
struct nlx_buf {
    nlx_ctx* ctx;
    void* dev;
    size_t bytes;
};

int32_t nlx_buf_create(nlx_ctx* c, size_t bytes, nlx_buf** out) NLX_TRY {
    if (!c) return NLX_E_INVAL;
    if (!out || bytes == 0) return c->fail(NLX_E_INVAL, "nlx_buf_create: NULL out or zero size");
    *out = nullptr;
    (void)hipSetDevice(c->device);
    void* p = c->alloc(bytes);
    if (!p) return NLX_E_NOMEM;
    nlx_buf* b = new (std::nothrow) nlx_buf{c, p, bytes};
    if (!b) { c->release(p); return c->fail(NLX_E_NOMEM, "host allocation failed"); }
    *out = b;
    return NLX_OK;
} NLX_CATCH(c)

void nlx_buf_destroy(nlx_buf* b) NLX_TRY {
    if (!b) return;
    (void)hipSetDevice(b->ctx->device);
    (void)hipStreamSynchronize(b->ctx->stream);
    b->ctx->release(b->dev);
    delete b;
} NLX_CATCH_VOID(nullptr)

void* nlx_buf_device_ptr(const nlx_buf* b) NLX_TRY { return b ? b->dev : nullptr; } NLX_CATCH_VALUE(nullptr, nullptr)
size_t nlx_buf_size(const nlx_buf* b) NLX_TRY { return b ? b->bytes : 0; } NLX_CATCH_VALUE(nullptr, 0)

int32_t nlx_buf_upload(nlx_buf* b, size_t offset, const void* src, size_t bytes) NLX_TRY {
    if (!b) return NLX_E_INVAL;
    nlx_ctx* c = b->ctx;
    if (!src && bytes) return c->fail(NLX_E_INVAL, "NULL source");
    if (offset > b->bytes || bytes > b->bytes - offset) return c->fail(NLX_E_RANGE, "upload exceeds the buffer");
    (void)hipSetDevice(c->device);
    NLX_HIP(c, hipMemcpyAsync((uint8_t*)b->dev + offset, src, bytes, hipMemcpyHostToDevice, c->stream));
    NLX_HIP(c, hipStreamSynchronize(c->stream));
    return NLX_OK;
} NLX_CATCH(nullptr)

int32_t nlx_buf_download(nlx_buf* b, size_t offset, void* dst, size_t bytes) NLX_TRY {
    if (!b) return NLX_E_INVAL;
    nlx_ctx* c = b->ctx;
    if (!dst && bytes) return c->fail(NLX_E_INVAL, "NULL destination");
    if (offset > b->bytes || bytes > b->bytes - offset) return c->fail(NLX_E_RANGE, "download exceeds the buffer");
    (void)hipSetDevice(c->device);
    NLX_HIP(c, hipMemcpyAsync(dst, (const uint8_t*)b->dev + offset, bytes, hipMemcpyDeviceToHost, c->stream));
    NLX_HIP(c, hipStreamSynchronize(c->stream));
    return NLX_OK;
} NLX_CATCH(nullptr)

int32_t nlx_ctx_trim(nlx_ctx* c) NLX_TRY {
    if (!c) return NLX_E_INVAL;
    (void)hipSetDevice(c->device);
    // the BN254 twiddle tables (up to 268 MB each at 2^24 points) are rebuilt on demand: give them back as well
    (void)hipStreamSynchronize(c->stream);
    for (auto& kv : c->bn254_tables) c->release(kv.second);
    c->bn254_tables.clear();
    c->trim();
    return NLX_OK;
} NLX_CATCH(c)

int32_t nlx_ctx_memory(const nlx_ctx* c, size_t* reserved_bytes, size_t* in_use_bytes) NLX_TRY {
    if (!c) return NLX_E_INVAL;
    size_t used = 0;
    for (const auto& kv : c->live_blocks) used += kv.second;
    if (reserved_bytes) *reserved_bytes = c->bytes_reserved;
    if (in_use_bytes) *in_use_bytes = used;
    return NLX_OK;
} NLX_CATCH(nullptr)

int32_t nlx_ctx_synchronize(nlx_ctx* c) NLX_TRY {
    if (!c) return NLX_E_INVAL;
    NLX_HIP(c, hipStreamSynchronize(c->stream));
    return NLX_OK;
} NLX_CATCH(c)

}  // extern "C"
