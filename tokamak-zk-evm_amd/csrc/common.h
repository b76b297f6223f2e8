// common.h — host-side plumbing shared by the HIP translation units of libtkmk_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <utility>
#include <vector>

#include "../../include/tkmk.h"
#include "ec.h"

#define TK_API extern "C" __attribute__((visibility("default")))

tkmk_error tk_map_hip_error(hipError_t e);

#define TK_HIP(call)                                         \
    do {                                                     \
        hipError_t _e = (call);                              \
        if (_e != hipSuccess) return tk_map_hip_error(_e);   \
    } while (0)

#define TK_TRY(call)                          \
    do {                                      \
        tkmk_error _t = (call);               \
        if (_t != TKMK_SUCCESS) return _t;    \
    } while (0)

// Makes sure a gfx950 device is current; TKMK_ERR_NO_DEVICE otherwise (there is no CPU fallback).
tkmk_error tk_require_device();

static inline hipStream_t tk_stream(tkmk_stream s) { return (hipStream_t)s; }
bool tk_stream_is_background(hipStream_t s);   // runtime.hip: tkmk_stream_set_background

// Device scratch comes from a per-stream, grow-only arena of plain hipMalloc memory, bump-allocated inside
// a tk_frame (one per C-ABI call) and recycled when the frame ends.  Launchers therefore never call
// hipMalloc/hipFree between kernels (cdna_hip_programming.md Guideline 9).  hipMallocAsync/hipFreeAsync
// are deliberately NOT used: on ROCm 7.2 / gfx950 memory that the stream-ordered pool frees and hands out
// again is read stale across XCD L2s by the next kernels (tools/coherence_test.hip reproduces it; plain
// hipMalloc memory reused across kernels is coherent at kernel boundaries).
struct tk_arena;
tk_arena *tk_arena_for(hipStream_t s);
struct tk_frame {
    tk_arena *a;
    size_t saved_chunk, saved_off, saved_used;
    explicit tk_frame(hipStream_t s);
    ~tk_frame();
    tk_frame(const tk_frame &) = delete;
    tk_frame &operator=(const tk_frame &) = delete;
};
// internal streams the pipelined MSM entries spread independent jobs over (runtime.hip; 1 = every kernel of a batch in issue order
// on one stream: the serialised form the profiling passes use)
uint32_t tk_msm_pipeline_streams();
struct tk_scratch {
    void *p = nullptr;
    tkmk_error alloc(size_t bytes, hipStream_t stream);  // valid until the enclosing tk_frame ends
    template <class T>
    T *as() const { return (T *)p; }
};

// Stages a caller buffer on the device when it is a host pointer; no-op (aliases) when it is already
// a device pointer.  copy_back() returns device results to a host destination.
struct tk_staged {
    void *dev = nullptr;
    tk_scratch own;
    tkmk_error in(const void *src, size_t bytes, bool on_device, hipStream_t s);     // for inputs
    tkmk_error out(void *dst, size_t bytes, bool on_device, hipStream_t s);          // for outputs
    tkmk_error copy_back(void *dst, size_t bytes, bool on_device, hipStream_t s);
};

// Per-kernel timing with HIP events recorded on the launch stream (bench.py's roofline leg).  Off unless
// tkmk_profile_enable(1); when on, tk_prof::mark(name) brackets the kernels launched since the last mark.
struct tk_prof {
    hipStream_t s;
    bool on;
    std::vector<std::pair<std::string, hipEvent_t>> ev;
    explicit tk_prof(hipStream_t stream);
    ~tk_prof();
    void mark(const char *name);  // ends the section `name` (sections start at the previous mark)
    void finish();                // synchronises and publishes the section times
};

// work counters behind tkmk_stats_get (runtime.hip)
enum { TK_STAT_MSM_POINTS = 0, TK_STAT_MSM_CALLS, TK_STAT_NTT_ELEMENTS, TK_STAT_NTT_CALLS, TK_STAT_POLY_ELEMENTS, TK_STAT_COUNT };
void tk_stat_add(int which, uint64_t v);
unsigned long long *tk_stat_device_entries();   // device cell counting the bucket additions issued (sorted-list lengths); nullptr on failure

static inline unsigned tk_div_up(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }
static inline int tk_log2_exact(uint64_t n) {
    if (n == 0 || (n & (n - 1))) return -1;
    int l = 0;
    while ((1ull << l) < n) l++;
    return l;
}

// ---- 16-byte vectorised global access for field elements (2 or 3 x dwordx4 per element) ----
template <class E>
__device__ __forceinline__ E tk_load(const E *p) {
    static_assert(sizeof(E) % 16 == 0, "");
    E r;
    const uint4 *s = reinterpret_cast<const uint4 *>(p);
    uint4 *d = reinterpret_cast<uint4 *>(&r);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(E) / 16); i++) d[i] = s[i];
    return r;
}
template <class E>
__device__ __forceinline__ void tk_store(E *p, const E &v) {
    uint4 *d = reinterpret_cast<uint4 *>(p);
    const uint4 *s = reinterpret_cast<const uint4 *>(&v);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(E) / 16); i++) d[i] = s[i];
}
