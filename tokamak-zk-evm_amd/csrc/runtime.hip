// runtime.hip — device runtime entry points of the C ABI (include/tkmk.h): the work-alike of
// icicle_runtime::{Device, DeviceVec, IcicleStream} that the reference uses around every MSM / NTT
// (e.g. packages/backend/libs/src/bivariate_polynomial/mod.rs:446-457, libs/src/utils/mod.rs:78-110).
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"

tkmk_error tk_map_hip_error(hipError_t e) {
    switch (e) {
        case hipSuccess: return TKMK_SUCCESS;
        case hipErrorOutOfMemory: return TKMK_ERR_OUT_OF_MEMORY;
        case hipErrorInvalidDevice: return TKMK_ERR_INVALID_DEVICE;
        case hipErrorNoDevice: return TKMK_ERR_NO_DEVICE;
        case hipErrorInvalidValue: return TKMK_ERR_INVALID_ARGUMENT;
        case hipErrorInvalidDevicePointer: return TKMK_ERR_INVALID_POINTER;
        default: return TKMK_ERR_UNKNOWN;
    }
}

static std::once_flag g_dev_once;
static tkmk_error g_dev_status = TKMK_ERR_NO_DEVICE;
// The library's state — scratch arenas, the allocation cache, the NTT domain and its tables, the MSM pipeline's streams — lives on
// ONE device: the one that is current when the first entry point runs (one process per GPU: DESIGN.md section 6).  ICICLE lets a
// process hop between devices with set_device; here a later tkmk_set_device to ANOTHER device is refused instead of handing
// device-0 scratch to kernels running on device 1 (without peer access that is a memory fault).
static std::atomic<int> g_bound_device{-1};

// the decision tkmk_set_device takes, as a pure function (tests/test_abi.py exercises it without a GPU through tkmk_diag_device_switch)
tkmk_error tk_device_switch_verdict(int bound_device, int requested_device, int device_count) {
    if (device_count <= 0) return TKMK_ERR_NO_DEVICE;
    if (requested_device < 0 || requested_device >= device_count) return TKMK_ERR_INVALID_DEVICE;
    if (bound_device >= 0 && bound_device != requested_device) return TKMK_ERR_INVALID_DEVICE;   // state exists on another device
    return TKMK_SUCCESS;
}
TK_API tkmk_error tkmk_diag_device_switch(int bound_device, int requested_device, int device_count) {
    return tk_device_switch_verdict(bound_device, requested_device, device_count);
}

tkmk_error tk_require_device() {
    std::call_once(g_dev_once, [] {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
            g_dev_status = TKMK_ERR_NO_DEVICE;
            return;
        }
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            g_dev_status = TKMK_ERR_INVALID_DEVICE;
            return;
        }
        // the code objects in this library are gfx950 only
        const char *arch = prop.gcnArchName;
        bool ok = arch[0] == 'g' && arch[1] == 'f' && arch[2] == 'x' && arch[3] == '9' && arch[4] == '5' && arch[5] == '0';
        g_dev_status = ok ? TKMK_SUCCESS : TKMK_ERR_INVALID_DEVICE;
        if (ok) g_bound_device.store(dev);
    });
    // HIP's current device is per host thread and starts at 0: a thread the host started after binding (a rank's helper thread, the
    // virtual ranks of the loopback transport) must work on the library's device, not on device 0 of a multi-GPU node
    if (g_dev_status == TKMK_SUCCESS) {
        static thread_local bool checked = false;
        if (!checked) {
            const int bound = g_bound_device.load();
            int cur = -1;
            if (bound >= 0 && hipGetDevice(&cur) == hipSuccess && cur != bound) (void)hipSetDevice(bound);
            checked = true;
        }
    }
    return g_dev_status;
}

struct tk_arena {
    struct chunk {
        char *base;
        size_t size;
    };
    std::vector<chunk> chunks;
    size_t cur = 0, off = 0;  // bump position: chunk index, offset within it
    size_t used_peak = 0, used_now = 0;
    int depth = 0;
    hipStream_t stream = nullptr;
};
static std::mutex g_arena_mu;
static std::map<hipStream_t, tk_arena *> g_arenas;

tk_arena *tk_arena_for(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_arena_mu);
    auto it = g_arenas.find(s);
    if (it != g_arenas.end()) return it->second;
    tk_arena *a = new tk_arena();
    a->stream = s;
    g_arenas[s] = a;
    return a;
}

static tkmk_error arena_alloc(tk_arena *a, size_t bytes, void **out) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes == 0) bytes = 256;
    while (a->cur < a->chunks.size()) {
        tk_arena::chunk &c = a->chunks[a->cur];
        if (a->off + bytes <= c.size) {
            *out = c.base + a->off;
            a->off += bytes;
            a->used_now += bytes;
            if (a->used_now > a->used_peak) a->used_peak = a->used_now;
            return TKMK_SUCCESS;
        }
        a->cur++;
        a->off = 0;
    }
    // grow: a new chunk (rare; the frame pop consolidates to one chunk of the peak size)
    size_t sz = bytes > ((size_t)64 << 20) ? bytes : ((size_t)64 << 20);
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, sz);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return e == hipErrorOutOfMemory ? TKMK_ERR_OUT_OF_MEMORY : TKMK_ERR_ALLOCATION_FAILED;
    }
    a->chunks.push_back({(char *)p, sz});
    a->cur = a->chunks.size() - 1;
    a->off = bytes;
    a->used_now += bytes;
    if (a->used_now > a->used_peak) a->used_peak = a->used_now;
    *out = p;
    return TKMK_SUCCESS;
}

tk_frame::tk_frame(hipStream_t s) : a(tk_arena_for(s)) {
    saved_chunk = a->cur;
    saved_off = a->off;
    saved_used = a->used_now;
    a->depth++;
}
tk_frame::~tk_frame() {
    a->depth--;
    if (a->depth > 0) {
        a->cur = saved_chunk;
        a->off = saved_off;
        a->used_now = saved_used;
        return;
    }
    a->cur = 0;
    a->off = 0;
    a->used_now = 0;
    if (a->chunks.size() > 1) {
        // consolidate so the next call of this size bump-allocates from one block
        (void)hipStreamSynchronize(a->stream);
        size_t want = a->used_peak + ((size_t)16 << 20);
        for (auto &c : a->chunks) (void)hipFree(c.base);
        a->chunks.clear();
        void *p = nullptr;
        if (hipMalloc(&p, want) == hipSuccess) a->chunks.push_back({(char *)p, want});
        else (void)hipGetLastError();
    }
}

tkmk_error tk_scratch::alloc(size_t bytes, hipStream_t stream) { return arena_alloc(tk_arena_for(stream), bytes, &p); }

tkmk_error tk_staged::in(const void *src, size_t bytes, bool on_device, hipStream_t s) {
    if (!src && bytes) return TKMK_ERR_INVALID_POINTER;
    if (on_device) {
        dev = const_cast<void *>(src);
        return TKMK_SUCCESS;
    }
    TK_TRY(own.alloc(bytes, s));
    dev = own.p;
    TK_HIP(hipMemcpyAsync(dev, src, bytes, hipMemcpyHostToDevice, s));
    return TKMK_SUCCESS;
}
tkmk_error tk_staged::out(void *dst, size_t bytes, bool on_device, hipStream_t s) {
    if (!dst && bytes) return TKMK_ERR_INVALID_POINTER;
    if (on_device) {
        dev = dst;
        return TKMK_SUCCESS;
    }
    TK_TRY(own.alloc(bytes, s));
    dev = own.p;
    return TKMK_SUCCESS;
}
tkmk_error tk_staged::copy_back(void *dst, size_t bytes, bool on_device, hipStream_t s) {
    if (on_device) return TKMK_SUCCESS;
    TK_HIP(hipMemcpyAsync(dst, dev, bytes, hipMemcpyDeviceToHost, s));
    return TKMK_SUCCESS;
}

// frees every per-stream scratch arena (they are grow-only otherwise: a 2^24-point MSM leaves ~7 GiB parked)
void tk_alloc_cache_release();
size_t tk_alloc_cache_bytes();
TK_API tkmk_error tkmk_release_scratch(void) {
    std::lock_guard<std::mutex> lk(g_arena_mu);
    (void)hipDeviceSynchronize();
    tk_alloc_cache_release();
    for (auto &kv : g_arenas) {
        tk_arena *a = kv.second;
        if (a->depth > 0) return TKMK_ERR_INVALID_ARGUMENT;  // a call is in progress on that stream
        for (auto &c : a->chunks) (void)hipFree(c.base);
        a->chunks.clear();
        a->cur = a->off = 0;
        a->used_peak = a->used_now = 0;
    }
    return TKMK_SUCCESS;
}

TK_API tkmk_error tkmk_device_count(int *count) {
    if (!count) return TKMK_ERR_INVALID_POINTER;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *count = n;
    return TKMK_SUCCESS;
}
TK_API tkmk_error tkmk_set_device(int device_id) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return TKMK_ERR_NO_DEVICE;
    TK_TRY(tk_device_switch_verdict(g_bound_device.load(), device_id, n));
    TK_HIP(hipSetDevice(device_id));
    return tk_require_device();   // binds the library to device_id if this is the first entry point of the process
}
TK_API tkmk_error tkmk_get_available_memory(size_t *total, size_t *free_bytes) {
    TK_TRY(tk_require_device());
    size_t f = 0, t = 0;
    TK_HIP(hipMemGetInfo(&f, &t));
    if (total) *total = t;
    if (free_bytes) *free_bytes = f + tk_alloc_cache_bytes();   // cached blocks are available to the next tkmk_malloc
    return TKMK_SUCCESS;
}
// ---- caller-visible device allocations (DeviceVec::device_malloc / Drop in the reference) ----
// hipMalloc + hipFree of a large block cost ~0.13 ms and hipFree synchronises the device; the polynomial layer above the
// ABI allocates and drops a 256 MiB temporary per operation (as the reference does).  Freed blocks are therefore kept in
// size-class free lists and handed out again; a block is reused only after the event recorded at its release has
// completed.  (hipMallocAsync's pool is not used: recycled pool memory read stale data across XCD L2s on this ROCm — see
// tools/coherence_test.hip.)  tkmk_release_scratch() returns everything to the driver.
namespace {
// a parked block may be handed out again once everything queued before its release has finished: the event on the releasing
// stream, and — for tkmk_free, which like Drop for DeviceVec names no stream — one event on every stream the caller created with
// tkmk_stream_create (they are non-blocking: the NULL stream's event does not order their work).  The library's own pipeline
// streams have drained when an entry point returns.
struct cached_block {
    void *p;
    std::vector<hipEvent_t> evs;
};
std::mutex g_streams_mu;
std::vector<hipStream_t> g_user_streams;
std::mutex g_alloc_mu;
std::map<void *, size_t> g_live;                       // block -> class size
std::map<size_t, std::vector<cached_block>> g_cache;   // class size -> free blocks
size_t g_cached_bytes = 0;

size_t alloc_class(size_t bytes) {
    if (bytes == 0) bytes = 16;
    const size_t g = bytes < ((size_t)1 << 20) ? (size_t)4096 : ((size_t)2 << 20);
    return (bytes + g - 1) / g * g;
}
size_t alloc_cache_cap() {
    static const size_t cap = [] {
        const char *e = getenv("TKMK_ALLOC_CACHE_GB");
        return (size_t)(e ? atoi(e) : 64) << 30;
    }();
    return cap;
}
void alloc_cache_flush_locked() {
    for (auto &kv : g_cache)
        for (auto &b : kv.second) {
            for (hipEvent_t e : b.evs) {
                (void)hipEventSynchronize(e);
                (void)hipEventDestroy(e);
            }
            (void)hipFree(b.p);
        }
    g_cache.clear();
    g_cached_bytes = 0;
}
tkmk_error cached_malloc(void **ptr, size_t bytes) {
    const size_t cls = alloc_class(bytes);
    cached_block blk{nullptr, {}};
    {
        std::lock_guard<std::mutex> lk(g_alloc_mu);
        auto it = g_cache.find(cls);
        if (it != g_cache.end() && !it->second.empty()) {
            blk = it->second.back();
            it->second.pop_back();
            g_cached_bytes -= cls;
            g_live[blk.p] = cls;
        }
    }
    if (blk.p) {
        for (hipEvent_t e : blk.evs) {
            (void)hipEventSynchronize(e);   // work queued before the release has finished
            (void)hipEventDestroy(e);
        }
        *ptr = blk.p;
        return TKMK_SUCCESS;
    }
    hipError_t e = hipMalloc(ptr, cls);
    if (e != hipSuccess) {   // give the cached blocks back and try once more
        (void)hipGetLastError();
        {
            std::lock_guard<std::mutex> lk(g_alloc_mu);
            alloc_cache_flush_locked();
        }
        e = hipMalloc(ptr, cls);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *ptr = nullptr;
        return e == hipErrorOutOfMemory ? TKMK_ERR_OUT_OF_MEMORY : TKMK_ERR_ALLOCATION_FAILED;
    }
    std::lock_guard<std::mutex> lk(g_alloc_mu);
    g_live[*ptr] = cls;
    return TKMK_SUCCESS;
}
tkmk_error cached_free(void *ptr, hipStream_t s, bool every_caller_stream) {
    size_t cls = 0;
    {
        std::lock_guard<std::mutex> lk(g_alloc_mu);
        auto it = g_live.find(ptr);
        if (it != g_live.end()) {
            cls = it->second;
            g_live.erase(it);
        }
    }
    if (!cls || cls > alloc_cache_cap()) {   // not one of ours (or too large to keep): plain release
        if (hipFree(ptr) != hipSuccess) {
            (void)hipGetLastError();
            return TKMK_ERR_DEALLOCATION_FAILED;
        }
        return TKMK_SUCCESS;
    }
    cached_block blk{ptr, {}};
    std::vector<hipStream_t> streams = {s};
    if (every_caller_stream) {
        std::lock_guard<std::mutex> lk(g_streams_mu);
        streams.insert(streams.end(), g_user_streams.begin(), g_user_streams.end());
    }
    bool ok = true;
    for (hipStream_t st : streams) {
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess || hipEventRecord(e, st) != hipSuccess) {
            (void)hipGetLastError();
            if (e) (void)hipEventDestroy(e);
            ok = false;
            break;
        }
        blk.evs.push_back(e);
    }
    if (!ok) {   // cannot order the release: fall back to the synchronising free
        for (hipEvent_t e : blk.evs) (void)hipEventDestroy(e);
        return hipFree(ptr) == hipSuccess ? TKMK_SUCCESS : TKMK_ERR_DEALLOCATION_FAILED;
    }
    std::lock_guard<std::mutex> lk(g_alloc_mu);
    if (g_cached_bytes + cls > alloc_cache_cap()) alloc_cache_flush_locked();
    g_cache[cls].push_back(blk);
    g_cached_bytes += cls;
    return TKMK_SUCCESS;
}
}  // namespace

void tk_alloc_cache_release() {
    std::lock_guard<std::mutex> lk(g_alloc_mu);
    alloc_cache_flush_locked();
}
size_t tk_alloc_cache_bytes() {
    std::lock_guard<std::mutex> lk(g_alloc_mu);
    return g_cached_bytes;
}

TK_API tkmk_error tkmk_malloc(void **ptr, size_t bytes) {
    if (!ptr) return TKMK_ERR_INVALID_POINTER;
    TK_TRY(tk_require_device());
    return cached_malloc(ptr, bytes);
}
TK_API tkmk_error tkmk_malloc_async(void **ptr, size_t bytes, tkmk_stream s) {
    (void)s;  // a cached block is ready when it is handed out; a fresh one comes from hipMalloc
    if (!ptr) return TKMK_ERR_INVALID_POINTER;
    TK_TRY(tk_require_device());
    return cached_malloc(ptr, bytes);
}
TK_API tkmk_error tkmk_free(void *ptr) {
    if (!ptr) return TKMK_SUCCESS;
    return cached_free(ptr, nullptr, true);
}
TK_API tkmk_error tkmk_free_async(void *ptr, tkmk_stream s) {
    if (!ptr) return TKMK_SUCCESS;
    return cached_free(ptr, tk_stream(s), false);   // the caller names the stream that last used the block
}
static tkmk_error copy(void *dst, const void *src, size_t bytes, hipMemcpyKind k) {
    if (bytes == 0) return TKMK_SUCCESS;
    if (!dst || !src) return TKMK_ERR_INVALID_POINTER;
    TK_TRY(tk_require_device());
    if (hipMemcpy(dst, src, bytes, k) != hipSuccess) {
        (void)hipGetLastError();
        return TKMK_ERR_COPY_FAILED;
    }
    return TKMK_SUCCESS;
}
static tkmk_error copy_async(void *dst, const void *src, size_t bytes, hipMemcpyKind k, tkmk_stream s) {
    if (bytes == 0) return TKMK_SUCCESS;
    if (!dst || !src) return TKMK_ERR_INVALID_POINTER;
    TK_TRY(tk_require_device());
    if (hipMemcpyAsync(dst, src, bytes, k, tk_stream(s)) != hipSuccess) {
        (void)hipGetLastError();
        return TKMK_ERR_COPY_FAILED;
    }
    return TKMK_SUCCESS;
}
TK_API tkmk_error tkmk_memcpy_h2d(void *dst, const void *src, size_t bytes) { return copy(dst, src, bytes, hipMemcpyHostToDevice); }
TK_API tkmk_error tkmk_memcpy_d2h(void *dst, const void *src, size_t bytes) { return copy(dst, src, bytes, hipMemcpyDeviceToHost); }
TK_API tkmk_error tkmk_memcpy_d2d(void *dst, const void *src, size_t bytes) { return copy(dst, src, bytes, hipMemcpyDeviceToDevice); }
TK_API tkmk_error tkmk_memcpy_h2d_async(void *dst, const void *src, size_t bytes, tkmk_stream s) {
    return copy_async(dst, src, bytes, hipMemcpyHostToDevice, s);
}
TK_API tkmk_error tkmk_memcpy_d2h_async(void *dst, const void *src, size_t bytes, tkmk_stream s) {
    return copy_async(dst, src, bytes, hipMemcpyDeviceToHost, s);
}
TK_API tkmk_error tkmk_memcpy_2d_d2d(void *dst, size_t dst_pitch, const void *src, size_t src_pitch, size_t width_bytes, size_t rows) {
    if (width_bytes == 0 || rows == 0) return TKMK_SUCCESS;
    if (!dst || !src) return TKMK_ERR_INVALID_POINTER;
    if (dst_pitch < width_bytes || src_pitch < width_bytes) return TKMK_ERR_INVALID_ARGUMENT;
    TK_TRY(tk_require_device());
    if (hipMemcpy2D(dst, dst_pitch, src, src_pitch, width_bytes, rows, hipMemcpyDeviceToDevice) != hipSuccess) {
        (void)hipGetLastError();
        return TKMK_ERR_COPY_FAILED;
    }
    return TKMK_SUCCESS;
}
TK_API tkmk_error tkmk_memset(void *ptr, int value, size_t bytes) {
    if (bytes == 0) return TKMK_SUCCESS;
    if (!ptr) return TKMK_ERR_INVALID_POINTER;
    TK_TRY(tk_require_device());
    TK_HIP(hipMemset(ptr, value, bytes));
    return TKMK_SUCCESS;
}
TK_API tkmk_error tkmk_stream_create(tkmk_stream *s) {
    if (!s) return TKMK_ERR_INVALID_POINTER;
    TK_TRY(tk_require_device());
    hipStream_t h;
    if (hipStreamCreateWithFlags(&h, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        return TKMK_ERR_STREAM_CREATION_FAILED;
    }
    {
        std::lock_guard<std::mutex> lk(g_streams_mu);
        g_user_streams.push_back(h);
    }
    *s = (tkmk_stream)h;
    return TKMK_SUCCESS;
}
TK_API tkmk_error tkmk_stream_synchronize(tkmk_stream s) {
    TK_TRY(tk_require_device());
    if (hipStreamSynchronize(tk_stream(s)) != hipSuccess) {
        (void)hipGetLastError();
        return TKMK_ERR_SYNCHRONIZATION_FAILED;
    }
    return TKMK_SUCCESS;
}
// ---- background streams: MSM batches issued on them yield registers to everything else (csrc/msm_impl.inc) ----
static std::mutex g_background_mu;
static std::vector<hipStream_t> g_background;
bool tk_stream_is_background(hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_background_mu);
    return std::find(g_background.begin(), g_background.end(), s) != g_background.end();
}
TK_API tkmk_error tkmk_stream_set_background(tkmk_stream stream, int on) {
    if (!stream) return TKMK_ERR_INVALID_ARGUMENT;   // the default stream is the foreground by definition
    std::lock_guard<std::mutex> lk(g_background_mu);
    auto it = std::find(g_background.begin(), g_background.end(), tk_stream(stream));
    if (on && it == g_background.end()) g_background.push_back(tk_stream(stream));
    if (!on && it != g_background.end()) g_background.erase(it);
    return TKMK_SUCCESS;
}
namespace tk_msm_bls12_381 { void drop_pipe_set(hipStream_t caller); }
namespace tk_msm_bn254 { void drop_pipe_set(hipStream_t caller); }
TK_API tkmk_error tkmk_stream_destroy(tkmk_stream s) {
    if (!s) return TKMK_SUCCESS;
    tk_msm_bls12_381::drop_pipe_set(tk_stream(s));   // the MSM pipeline sets that served batches issued on this stream
    tk_msm_bn254::drop_pipe_set(tk_stream(s));
    (void)tkmk_stream_set_background(s, 0);
    {
        std::lock_guard<std::mutex> lk(g_streams_mu);
        for (size_t i = 0; i < g_user_streams.size(); i++)
            if (g_user_streams[i] == tk_stream(s)) {
                g_user_streams.erase(g_user_streams.begin() + i);
                break;
            }
    }
    (void)hipStreamSynchronize(tk_stream(s));   // parked blocks may hold events recorded on this stream
    if (hipStreamDestroy(tk_stream(s)) != hipSuccess) {
        (void)hipGetLastError();
        return TKMK_ERR_STREAM_DESTRUCTION_FAILED;
    }
    return TKMK_SUCCESS;
}
TK_API tkmk_error tkmk_device_synchronize(void) {
    TK_TRY(tk_require_device());
    if (hipDeviceSynchronize() != hipSuccess) {
        (void)hipGetLastError();
        return TKMK_ERR_SYNCHRONIZATION_FAILED;
    }
    return TKMK_SUCCESS;
}
TK_API const char *tkmk_error_string(tkmk_error e) {
    switch (e) {
        case TKMK_SUCCESS: return "success";
        case TKMK_ERR_INVALID_DEVICE: return "invalid device (libtkmk_hip.so carries gfx950 code objects only)";
        case TKMK_ERR_OUT_OF_MEMORY: return "out of device memory";
        case TKMK_ERR_INVALID_POINTER: return "invalid pointer";
        case TKMK_ERR_ALLOCATION_FAILED: return "allocation failed";
        case TKMK_ERR_DEALLOCATION_FAILED: return "deallocation failed";
        case TKMK_ERR_COPY_FAILED: return "copy failed";
        case TKMK_ERR_SYNCHRONIZATION_FAILED: return "synchronization failed";
        case TKMK_ERR_STREAM_CREATION_FAILED: return "stream creation failed";
        case TKMK_ERR_STREAM_DESTRUCTION_FAILED: return "stream destruction failed";
        case TKMK_ERR_API_NOT_IMPLEMENTED: return "not implemented";
        case TKMK_ERR_INVALID_ARGUMENT: return "invalid argument";
        case TKMK_ERR_NO_DEVICE: return "no HIP device (there is no CPU fallback)";
        default: return "unknown error";
    }
}
TK_API int tkmk_is_hip_build(void) { return 1; }

// HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default), and streams that share a queue run one
// after the other.  The resident prover keeps up to three commit batches in flight (the main thread's, the binding helper's and the early
// commits' — three pipeline streams each, csrc/msm_impl.inc): on 4 queues an early commit was seen waiting 11 ms behind an unrelated
// binding job that shared its queue (rocprofv3 kernel trace, production shape).  8 queues measured best (production shape 68.6 -> 65.7 ms,
// configs[3] 203.9 -> 202.2 ms; 16 no better).  Set when this library is loaded, before the HIP runtime reads its environment, and only
// if the process has not chosen a value itself.
__attribute__((constructor)) static void tk_default_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

// ---- event profiler ----
// Sections are bracketed by events recorded on the launch stream; nothing waits for them when they are recorded (a
// synchronisation inside the pipelined MSM entry would serialise exactly the overlap being measured).  finish() parks the
// events; tkmk_profile_get / _collect drain them after the device has gone idle.
static bool g_prof_on = false;
static std::mutex g_prof_mu;
static std::map<std::string, std::pair<double, int>> g_prof;  // name -> (sum ms, count)
static std::vector<std::vector<std::pair<std::string, hipEvent_t>>> g_prof_pending;

tk_prof::tk_prof(hipStream_t stream) : s(stream), on(g_prof_on) {
    if (on) mark("");
}
tk_prof::~tk_prof() {
    for (auto &e : ev) (void)hipEventDestroy(e.second);
}
void tk_prof::mark(const char *name) {
    if (!on) return;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, s);
    ev.emplace_back(name, e);
}
static void prof_account_and_destroy(std::vector<std::pair<std::string, hipEvent_t>> &list) {
    for (size_t i = 1; i < list.size(); i++) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, list[i - 1].second, list[i].second) != hipSuccess) continue;
        auto &slot = g_prof[list[i].first];
        slot.first += ms;
        slot.second += 1;
    }
    for (auto &e : list) (void)hipEventDestroy(e.second);
    list.clear();
}
// lists whose last event has completed are accounted without waiting (hipEventQuery); past a cap the oldest are waited for, so
// a resident process that profiles for hours and never polls holds a bounded number of events
static constexpr size_t PROF_PENDING_CAP = 256;
static void prof_drain_locked(bool wait_all) {
    size_t kept = 0;
    for (size_t k = 0; k < g_prof_pending.size(); k++) {
        auto &list = g_prof_pending[k];
        if (list.empty()) continue;
        bool must = wait_all || g_prof_pending.size() - k > PROF_PENDING_CAP;
        if (must) (void)hipEventSynchronize(list.back().second);
        if (must || hipEventQuery(list.back().second) == hipSuccess) {
            prof_account_and_destroy(list);
        } else {
            if (kept != k) g_prof_pending[kept] = std::move(list);
            kept++;
        }
    }
    g_prof_pending.resize(kept);
}
void tk_prof::finish() {
    if (!on || ev.size() < 2) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_pending.push_back(std::move(ev));
    ev.clear();
    if (g_prof_pending.size() >= 32) prof_drain_locked(false);
}
TK_API tkmk_error tkmk_profile_enable(int on) {
    g_prof_on = on != 0;
    if (!g_prof_on) {       // switching off accounts what is pending: nothing stays parked in a process that stops profiling
        std::lock_guard<std::mutex> lk(g_prof_mu);
        prof_drain_locked(true);
    }
    return TKMK_SUCCESS;
}
TK_API tkmk_error tkmk_profile_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    prof_drain_locked(true);
    g_prof.clear();
    return TKMK_SUCCESS;
}
TK_API tkmk_error tkmk_profile_get(const char *name, double *sum_ms, int *count) {
    if (!name || !sum_ms || !count) return TKMK_ERR_INVALID_POINTER;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    prof_drain_locked(true);
    auto it = g_prof.find(name);
    *sum_ms = it == g_prof.end() ? 0.0 : it->second.first;
    *count = it == g_prof.end() ? 0 : it->second.second;
    return TKMK_SUCCESS;
}

// ---- pipeline width of the multi-MSM entries ----
static std::atomic<int> g_msm_streams{0};   // 0 = not resolved yet
uint32_t tk_msm_pipeline_streams() {
    int v = g_msm_streams.load(std::memory_order_relaxed);
    if (v == 0) {
        const char *e = getenv("TKMK_MSM_STREAMS");
        v = e ? atoi(e) : 3;
        v = v < 1 ? 1 : v > 8 ? 8 : v;
        g_msm_streams.store(v, std::memory_order_relaxed);
    }
    return (uint32_t)v;
}
TK_API tkmk_error tkmk_msm_set_pipeline_streams(int n) {
    if (n < 0 || n > 8) return TKMK_ERR_INVALID_ARGUMENT;
    g_msm_streams.store(n, std::memory_order_relaxed);   // 0: back to TKMK_MSM_STREAMS / the default
    return TKMK_SUCCESS;
}
TK_API int tkmk_msm_get_pipeline_streams(void) { return (int)tk_msm_pipeline_streams(); }

// ---- work counters: what the library was asked to do since the last reset (bench.py turns them into algorithmic bytes:
// 128 B per MSM point, 64 B per NTT element) ----
static std::atomic<uint64_t> g_stat[TK_STAT_COUNT];
void tk_stat_add(int which, uint64_t v) {
    if (which >= 0 && which < TK_STAT_COUNT) g_stat[which].fetch_add(v, std::memory_order_relaxed);
}
// bucket additions actually issued (the length of the sorted lists: zero digits of the scalars drop out): counted on the device by a
// one-thread kernel after every sort, into one resident 64-bit cell; read back on request only
static unsigned long long *g_dev_entries = nullptr;
static std::mutex g_dev_entries_mu;
unsigned long long *tk_stat_device_entries() {
    std::lock_guard<std::mutex> lk(g_dev_entries_mu);
    if (!g_dev_entries) {
        if (hipMalloc((void **)&g_dev_entries, sizeof(unsigned long long)) != hipSuccess) return nullptr;
        (void)hipMemset(g_dev_entries, 0, sizeof(unsigned long long));
    }
    return g_dev_entries;
}
TK_API tkmk_error tkmk_stats_reset(void) {
    for (auto &c : g_stat) c.store(0);
    std::lock_guard<std::mutex> lk(g_dev_entries_mu);
    if (g_dev_entries) {
        TK_HIP(hipDeviceSynchronize());
        TK_HIP(hipMemset(g_dev_entries, 0, sizeof(unsigned long long)));
    }
    return TKMK_SUCCESS;
}
TK_API tkmk_error tkmk_stats_get(const char *name, uint64_t *value) {
    if (!name || !value) return TKMK_ERR_INVALID_POINTER;
    if (std::string(name) == "msm.bucket_additions") {
        std::lock_guard<std::mutex> lk(g_dev_entries_mu);
        unsigned long long v = 0;
        if (g_dev_entries) {
            TK_HIP(hipDeviceSynchronize());
            TK_HIP(hipMemcpy(&v, g_dev_entries, sizeof v, hipMemcpyDeviceToHost));
        }
        *value = v;
        return TKMK_SUCCESS;
    }
    static const char *names[TK_STAT_COUNT] = {"msm.points", "msm.calls", "ntt.elements", "ntt.calls", "poly.elements"};
    for (int i = 0; i < TK_STAT_COUNT; i++)
        if (std::string(name) == names[i]) {
            *value = g_stat[i].load();
            return TKMK_SUCCESS;
        }
    return TKMK_ERR_INVALID_ARGUMENT;
}
