// diag.hip — micro-benchmarks of the arithmetic primitives (diagnostics only; not on the prover path).
// They answer SURVEY.md §8d's open question: the sustained v_mad_u64_u32 rate on gfx950, which is the
// roofline that actually bounds MSM / NTT (integer VALU issue, not HBM, not MFMA).
#include "common.h"

// kind 0: Fr Montgomery products, 1: Fq products, 2: raw v_mad_u64_u32, 3: G1 XYZZ mixed additions,
// 4: Fr add/sub pairs, 5: Fq sqr
template <int KIND>
__global__ __launch_bounds__(256) void k_diag(uint32_t iters, uint32_t *sink) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (KIND == 0) {
        fr_t a = Fr::one(), b = Fr::r2();
        a.l[0] ^= t;
        a = Fr::canon(a);
        for (uint32_t i = 0; i < iters; i++) a = Fr::mul(a, b);
        if (a.l[0] == 0x12345) sink[t] = a.l[1];
    } else if (KIND == 1) {
        fq_t a = Fq::one(), b = Fq::r2();
        a.l[0] ^= t;
        for (uint32_t i = 0; i < iters; i++) a = Fq::mul(a, b);
        if (a.l[0] == 0x12345) sink[t] = a.l[1];
    } else if (KIND == 2) {
        uint64_t acc[8];
        uint32_t x = t | 1, y = t * 2654435761u + 1;
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] = k + t;
        for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
            for (int k = 0; k < 8; k++)
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[k]) : "v"(x), "v"(y) : "vcc");
        }
        uint64_t s = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) s ^= acc[k];
        if (s == 0x12345) sink[t] = (uint32_t)s;
    } else if (KIND == 3) {
        g1_affine_t q;
        q.x = Fq::one();
        q.y = Fq::r2();
        q.x.l[0] ^= t;
        g1_xyzz_t acc = G1::from_affine(q);
        q.y.l[1] ^= 0x55;
        for (uint32_t i = 0; i < iters; i++) {
            acc = G1::add_mixed(acc, q);
            q.x.l[2] += 1;  // keep the compiler from hoisting
        }
        if (acc.x.l[0] == 0x12345) sink[t] = acc.y.l[1];
    } else if (KIND == 4) {
        fr_t a = Fr::one(), b = Fr::r2();
        a.l[0] ^= t;
        a = Fr::canon(a);
        for (uint32_t i = 0; i < iters; i++) {
            fr_t s = Fr::add(a, b);
            b = Fr::sub(a, b);
            a = s;
        }
        if (a.l[0] == 0x12345) sink[t] = b.l[1];
    } else if (KIND == 5) {
        fq_t a = Fq::one();
        a.l[0] ^= t;
        for (uint32_t i = 0; i < iters; i++) a = Fq::sqr(a);
        if (a.l[0] == 0x12345) sink[t] = a.l[1];
    }
}

// Runs `reps` launches of (blocks x 256) threads, `iters` operations per thread; returns mean ms per launch.
TK_API tkmk_error tkmk_diag_bench(int kind, uint32_t iters, uint32_t blocks, int reps, float *ms_out) {
    if (!ms_out || reps < 1) return TKMK_ERR_INVALID_ARGUMENT;
    TK_TRY(tk_require_device());
    uint32_t *sink = nullptr;
    TK_HIP(hipMalloc((void **)&sink, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1;
    TK_HIP(hipEventCreate(&e0));
    TK_HIP(hipEventCreate(&e1));
    auto launch = [&]() {
        switch (kind) {
            case 0: hipLaunchKernelGGL(k_diag<0>, blocks, 256, 0, 0, iters, sink); break;
            case 1: hipLaunchKernelGGL(k_diag<1>, blocks, 256, 0, 0, iters, sink); break;
            case 2: hipLaunchKernelGGL(k_diag<2>, blocks, 256, 0, 0, iters, sink); break;
            case 3: hipLaunchKernelGGL(k_diag<3>, blocks, 256, 0, 0, iters, sink); break;
            case 4: hipLaunchKernelGGL(k_diag<4>, blocks, 256, 0, 0, iters, sink); break;
            default: hipLaunchKernelGGL(k_diag<5>, blocks, 256, 0, 0, iters, sink); break;
        }
    };
    launch();
    TK_HIP(hipDeviceSynchronize());
    TK_HIP(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; r++) launch();
    TK_HIP(hipEventRecord(e1, 0));
    TK_HIP(hipEventSynchronize(e1));
    float ms = 0;
    TK_HIP(hipEventElapsedTime(&ms, e0, e1));
    *ms_out = ms / reps;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(sink);
    return TKMK_SUCCESS;
}
