// diag.hip — micro-benchmarks of the arithmetic primitives (diagnostics only; not on the prover path).
// They answer SURVEY.md §8d's open question: the sustained v_mad_u64_u32 rate on gfx950, which is the
// roofline that actually bounds MSM / NTT (integer VALU issue, not HBM, not MFMA).
#include "common.h"
#include "ffu.h"

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
    } else if (KIND == 6) {   // Fr product in the tightest unsaturated form (9 x 29 bits, 162 multiply-adds, no carries)
        using FU = ffu<bls12_381_fr9_params>;
        typename FU::E a = FU::one(), b = FU::one();
        a.l[0] ^= t & 0xffff;
        b.l[1] ^= 0x1234;
        for (uint32_t i = 0; i < iters; i++) a = FU::mul(a, b);
        if (a.l[0] == 0x12345) sink[t] = a.l[1];
    } else if (KIND == 7) {   // one butterfly's worth in that form: product + add + sub (limb-wise, one carry sweep each)
        using FU = ffu<bls12_381_fr9_params>;
        typename FU::E a = FU::one(), b = FU::one(), w = FU::one();
        a.l[0] ^= t & 0xffff;
        w.l[1] ^= 0x1234;
        for (uint32_t i = 0; i < iters; i++) {
            typename FU::E m = FU::mul(b, w);
            typename FU::E s = FU::add(a, m);
            b = FU::template sub<2>(a, m);
            a = FU::cond_sub_p(FU::cond_sub_p(s));   // keep the chain bounded like a real pass would every few stages
            b = FU::cond_sub_p(FU::cond_sub_p(FU::cond_sub_p(b)));
        }
        if (a.l[0] == 0x12345) sink[t] = b.l[1];
    } else if (KIND == 8) {   // the same butterfly on the saturated form (what k_ntt_pass runs)
        fr_t a = Fr::one(), b = Fr::r2(), w = Fr::r2();
        a.l[0] ^= t;
        a = Fr::canon(a);
        for (uint32_t i = 0; i < iters; i++) {
            fr_t m = Fr::mul(b, w);
            fr_t s = Fr::add(a, m);
            b = Fr::sub(a, m);
            a = s;
        }
        if (a.l[0] == 0x12345) sink[t] = b.l[1];
    }
}

// Raw issue-rate probes: 8 independent chains of one instruction each (inline asm so nothing is folded).
#define PROBE_KERNEL(NAME, TYPE, ASM_LINE)                                                            \
    __global__ __launch_bounds__(256) void NAME(uint32_t iters, uint32_t *sink) {                     \
        uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;                                           \
        TYPE acc[8];                                                                                  \
        uint32_t x = t | 1, y = t * 2654435761u + 1;                                                  \
        _Pragma("unroll") for (int k = 0; k < 8; k++) acc[k] = k + t;                                 \
        for (uint32_t i = 0; i < iters; i++) {                                                        \
            _Pragma("unroll") for (int k = 0; k < 8; k++)                                             \
                asm volatile(ASM_LINE : "+v"(acc[k]) : "v"(x), "v"(y) : "vcc");                       \
        }                                                                                             \
        TYPE s = 0;                                                                                   \
        _Pragma("unroll") for (int k = 0; k < 8; k++) s ^= acc[k];                                    \
        if (s == 0x12345) sink[t] = (uint32_t)s;                                                      \
    }
PROBE_KERNEL(k_probe_add_co, uint32_t, "v_add_co_u32 %0, vcc, %1, %0")
PROBE_KERNEL(k_probe_addc, uint32_t, "v_addc_co_u32 %0, vcc, %1, %0, vcc")
PROBE_KERNEL(k_probe_add, uint32_t, "v_add_u32 %0, %1, %0")
PROBE_KERNEL(k_probe_add3, uint32_t, "v_add3_u32 %0, %1, %2, %0")
PROBE_KERNEL(k_probe_mov, uint32_t, "v_mov_b32 %0, %1")
PROBE_KERNEL(k_probe_cndmask, uint32_t, "v_cndmask_b32 %0, %1, %0, vcc")
PROBE_KERNEL(k_probe_mul_lo, uint32_t, "v_mul_lo_u32 %0, %1, %0")
PROBE_KERNEL(k_probe_mul_hi, uint32_t, "v_mul_hi_u32 %0, %1, %0")
PROBE_KERNEL(k_probe_mad24, uint32_t, "v_mad_u32_u24 %0, %1, %2, %0")
PROBE_KERNEL(k_probe_lshl_add_u64, uint64_t, "v_lshl_add_u64 %0, %0, 0, %0")
PROBE_KERNEL(k_probe_mad_u64, uint64_t, "v_mad_u64_u32 %0, vcc, %1, %2, %0")
PROBE_KERNEL(k_probe_fma64, uint64_t, "v_fma_f64 %0, %0, %0, %0")
PROBE_KERNEL(k_probe_mad_i32_i24, uint32_t, "v_mad_i32_i24 %0, %1, %2, %0")
PROBE_KERNEL(k_probe_alignbit, uint32_t, "v_alignbit_b32 %0, %1, %0, 3")
PROBE_KERNEL(k_probe_mad_u32_u16, uint32_t, "v_mad_u32_u16 %0, %1, %2, %0")
// cndmask variants: VOP2 with VCC written once per iteration by a compare; VOP3 with an SGPR-pair mask
__global__ __launch_bounds__(256) void k_probe_cndmask_real(uint32_t iters, uint32_t *sink) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc[8];
    uint32_t x = t | 1;
#pragma unroll
    for (int k = 0; k < 8; k++) acc[k] = k + t;
    for (uint32_t i = 0; i < iters; i++) {
        asm volatile("v_cmp_lt_u32 vcc, %8, %0\n\t"
                     "v_cndmask_b32 %0, %8, %0, vcc\n\tv_cndmask_b32 %1, %8, %1, vcc\n\tv_cndmask_b32 %2, %8, %2, vcc\n\t"
                     "v_cndmask_b32 %3, %8, %3, vcc\n\tv_cndmask_b32 %4, %8, %4, vcc\n\tv_cndmask_b32 %5, %8, %5, vcc\n\t"
                     "v_cndmask_b32 %6, %8, %6, vcc\n\tv_cndmask_b32 %7, %8, %7, vcc"
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
                     : "v"(x) : "vcc");
    }
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) s ^= acc[k];
    if (s == 0x12345) sink[t] = s;
}
__global__ __launch_bounds__(256) void k_probe_and_addc(uint32_t iters, uint32_t *sink) {
    // the mask-and-add alternative: 8 x (v_and with mask, v_addc)
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc[8];
    uint32_t x = t | 1, m = 0u - (t & 1);
#pragma unroll
    for (int k = 0; k < 8; k++) acc[k] = k + t;
    for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            uint32_t tmp;
            asm volatile("v_and_b32 %1, %2, %3\n\tv_addc_co_u32 %0, vcc, %1, %0, vcc" : "+v"(acc[k]), "=&v"(tmp) : "v"(x), "v"(m) : "vcc");
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) s ^= acc[k];
    if (s == 0x12345) sink[t] = s;
}
PROBE_KERNEL(k_probe_sub, uint32_t, "v_sub_u32 %0, %1, %0")
PROBE_KERNEL(k_probe_and, uint32_t, "v_and_b32 %0, %1, %0")
PROBE_KERNEL(k_probe_lshr, uint32_t, "v_lshrrev_b32 %0, 29, %0")
PROBE_KERNEL(k_probe_lshr64, uint64_t, "v_lshrrev_b64 %0, 29, %0")
PROBE_KERNEL(k_probe_and_or, uint32_t, "v_and_or_b32 %0, %1, %2, %0")
PROBE_KERNEL(k_probe_bfe, uint32_t, "v_bfe_u32 %0, %0, 3, 29")
PROBE_KERNEL(k_probe_lshl_add, uint32_t, "v_lshl_add_u32 %0, %1, 3, %0")
PROBE_KERNEL(k_probe_mul_u32_u24, uint32_t, "v_mul_u32_u24 %0, %1, %0")

// the product step of the Montgomery multiplier: mad into a 64-bit pair + addc into a third word
__global__ __launch_bounds__(256) void k_probe_mad_pair(uint32_t iters, uint32_t *sink) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t acc[8];
    uint32_t c2[8];
    uint32_t x = t | 1, y = t * 2654435761u + 1;
#pragma unroll
    for (int k = 0; k < 8; k++) acc[k] = k + t, c2[k] = k;
    for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++)
            asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc[k]), "+v"(c2[k]) : "v"(x), "v"(y) : "vcc");
    }
    uint64_t s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) s ^= acc[k] + c2[k];
    if (s == 0x12345) sink[t] = (uint32_t)s;
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
            case 5: hipLaunchKernelGGL(k_diag<5>, blocks, 256, 0, 0, iters, sink); break;
            case 6: hipLaunchKernelGGL(k_diag<6>, blocks, 256, 0, 0, iters, sink); break;
            case 7: hipLaunchKernelGGL(k_diag<7>, blocks, 256, 0, 0, iters, sink); break;
            case 8: hipLaunchKernelGGL(k_diag<8>, blocks, 256, 0, 0, iters, sink); break;
            case 100: hipLaunchKernelGGL(k_probe_add_co, blocks, 256, 0, 0, iters, sink); break;
            case 101: hipLaunchKernelGGL(k_probe_addc, blocks, 256, 0, 0, iters, sink); break;
            case 102: hipLaunchKernelGGL(k_probe_add, blocks, 256, 0, 0, iters, sink); break;
            case 103: hipLaunchKernelGGL(k_probe_add3, blocks, 256, 0, 0, iters, sink); break;
            case 104: hipLaunchKernelGGL(k_probe_mov, blocks, 256, 0, 0, iters, sink); break;
            case 105: hipLaunchKernelGGL(k_probe_cndmask, blocks, 256, 0, 0, iters, sink); break;
            case 106: hipLaunchKernelGGL(k_probe_mul_lo, blocks, 256, 0, 0, iters, sink); break;
            case 107: hipLaunchKernelGGL(k_probe_mul_hi, blocks, 256, 0, 0, iters, sink); break;
            case 108: hipLaunchKernelGGL(k_probe_mad24, blocks, 256, 0, 0, iters, sink); break;
            case 109: hipLaunchKernelGGL(k_probe_lshl_add_u64, blocks, 256, 0, 0, iters, sink); break;
            case 110: hipLaunchKernelGGL(k_probe_mad_u64, blocks, 256, 0, 0, iters, sink); break;
            case 111: hipLaunchKernelGGL(k_probe_mad_pair, blocks, 256, 0, 0, iters, sink); break;
            case 112: hipLaunchKernelGGL(k_probe_fma64, blocks, 256, 0, 0, iters, sink); break;
            case 113: hipLaunchKernelGGL(k_probe_mad_i32_i24, blocks, 256, 0, 0, iters, sink); break;
            case 114: hipLaunchKernelGGL(k_probe_alignbit, blocks, 256, 0, 0, iters, sink); break;
            case 115: hipLaunchKernelGGL(k_probe_mad_u32_u16, blocks, 256, 0, 0, iters, sink); break;
            case 116: hipLaunchKernelGGL(k_probe_cndmask_real, blocks, 256, 0, 0, iters, sink); break;
            case 117: hipLaunchKernelGGL(k_probe_and_addc, blocks, 256, 0, 0, iters, sink); break;
            case 118: hipLaunchKernelGGL(k_probe_sub, blocks, 256, 0, 0, iters, sink); break;
            case 119: hipLaunchKernelGGL(k_probe_and, blocks, 256, 0, 0, iters, sink); break;
            case 120: hipLaunchKernelGGL(k_probe_lshr, blocks, 256, 0, 0, iters, sink); break;
            case 121: hipLaunchKernelGGL(k_probe_lshr64, blocks, 256, 0, 0, iters, sink); break;
            case 122: hipLaunchKernelGGL(k_probe_and_or, blocks, 256, 0, 0, iters, sink); break;
            case 123: hipLaunchKernelGGL(k_probe_bfe, blocks, 256, 0, 0, iters, sink); break;
            case 124: hipLaunchKernelGGL(k_probe_lshl_add, blocks, 256, 0, 0, iters, sink); break;
            case 125: hipLaunchKernelGGL(k_probe_mul_u32_u24, blocks, 256, 0, 0, iters, sink); break;
            default: break;
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

// out[i] = a[i] * b[i] (plain in / plain out) through the device Montgomery product — lets the gpu test tier
// compare the hand-scheduled product against the oracle on edge values.  field 0 = Fr, 1 = Fq.
template <class F>
__global__ __launch_bounds__(256) void k_diag_mul(const typename F::E *a, const typename F::E *b, typename F::E *out, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    tk_store(out + i, F::mul(F::to_mont(F::canon(tk_load(a + i))), F::canon(tk_load(b + i))));
}
TK_API tkmk_error tkmk_diag_field_mul(int field, const void *a_dev, const void *b_dev, void *out_dev, uint64_t n) {
    TK_TRY(tk_require_device());
    if (n == 0) return TKMK_SUCCESS;
    if (!a_dev || !b_dev || !out_dev) return TKMK_ERR_INVALID_POINTER;
    if (field == 0)
        hipLaunchKernelGGL(k_diag_mul<Fr>, tk_div_up(n, 256), 256, 0, 0, (const fr_t *)a_dev, (const fr_t *)b_dev, (fr_t *)out_dev, n);
    else
        hipLaunchKernelGGL(k_diag_mul<Fq>, tk_div_up(n, 256), 256, 0, 0, (const fq_t *)a_dev, (const fq_t *)b_dev, (fq_t *)out_dev, n);
    TK_HIP(hipGetLastError());
    TK_HIP(hipDeviceSynchronize());
    return TKMK_SUCCESS;
}

// ---- known-bytes gather probe: calibrates rocprofv3's FETCH_SIZE for the access pattern of k_accumulate_chunks ----
// One lane loads one whole row (row_vecs x 16 bytes with consecutive dwordx4 loads, exactly as tk_load<g1_affine_t> does) at a
// caller-given row index and folds it into a checksum; n rows are gathered in total.  The number of useful bytes (n * row bytes)
// and of 64-byte sectors / 128-byte lines touched is known from the index list, so the counter's unit for THIS pattern can be
// read off (MI355X_MICROARCH.md: "other access widths are uncalibrated: calibrate on a known byte count in your own access
// pattern").  Also reports the launch time (gather bandwidth).
template <int VECS>
__global__ __launch_bounds__(256) void k_gather_probe(const uint4 *__restrict__ table, const uint32_t *__restrict__ idx, uint64_t n,
                                                     uint32_t *__restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 *row = table + (uint64_t)idx[i] * VECS;
    uint4 v[VECS];
#pragma unroll
    for (int k = 0; k < VECS; k++) v[k] = row[k];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < VECS; k++) s ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    if (s == 0x9e3779b9u) out[0] = s;   // keeps the loads alive; practically never true
}
TK_API tkmk_error tkmk_diag_gather_probe(const void *table_dev, uint32_t row_bytes, const uint32_t *idx_dev, uint64_t n, int reps, float *ms_out) {
    if (!table_dev || !idx_dev || !ms_out || reps < 1) return TKMK_ERR_INVALID_POINTER;
    if (row_bytes != 96 && row_bytes != 64 && row_bytes != 128) return TKMK_ERR_INVALID_ARGUMENT;
    TK_TRY(tk_require_device());
    uint32_t *out = nullptr;
    TK_HIP(hipMalloc((void **)&out, 4));
    hipEvent_t e0, e1;
    TK_HIP(hipEventCreate(&e0));
    TK_HIP(hipEventCreate(&e1));
    auto launch = [&]() {
        if (row_bytes == 96) hipLaunchKernelGGL(k_gather_probe<6>, tk_div_up(n, 256), 256, 0, 0, (const uint4 *)table_dev, idx_dev, n, out);
        else if (row_bytes == 64) hipLaunchKernelGGL(k_gather_probe<4>, tk_div_up(n, 256), 256, 0, 0, (const uint4 *)table_dev, idx_dev, n, out);
        else hipLaunchKernelGGL(k_gather_probe<8>, tk_div_up(n, 256), 256, 0, 0, (const uint4 *)table_dev, idx_dev, n, out);
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
    (void)hipFree(out);
    return TKMK_SUCCESS;
}
