// Diagnostic (not on the product path): what does a ONE-PASS bucket scatter through global atomics cost on gfx950, against the
// two-pass LDS-staged sort of csrc/msm_impl.inc (19 ms of a configs[3] proof, 1.15 TB/s of its ~30 bytes per entry)?
//   pass 1  count[d]++            (atomic without return) over all entries
//   scan    over the 2^19 counters
//   pass 2  pos = cursor[d]++     (atomic with return), sorted[pos] = entry
// with and without wave-level aggregation of equal digits (match-any through a 64-lane compare loop on the leader's digit) — the
// aggregation is what keeps a witness-like distribution (one bucket holding a third of the entries) from serialising on one address.
// Digits: uniform over 2^19 buckets (dense quotient scalars) and skewed (30 % in bucket 0, 30 % in 255 buckets, rest uniform).
// build: hipcc -O3 --offload-arch=gfx950 tools/atomic_scatter_probe.hip -o tools/atomic_scatter_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_gen(uint32_t *dig, uint64_t n, uint32_t B, int skew) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t x = i * 0x9E3779B97F4A7C15ull + 0x1234567;
    x ^= x >> 31; x *= 0xD6E8FEB86659FD93ull; x ^= x >> 29; x *= 0xD6E8FEB86659FD93ull; x ^= x >> 32;
    uint32_t d = (uint32_t)x & (B - 1);
    if (skew) {
        uint32_t r = (uint32_t)(x >> 40) % 10;
        if (r < 3) d = 0;
        else if (r < 6) d = 1 + ((uint32_t)(x >> 20) & 0xff);
    }
    dig[i] = d;
}
__global__ void k_count(const uint32_t *__restrict__ dig, uint64_t n, uint32_t *__restrict__ cnt) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(&cnt[dig[i]], 1u);
}
// wave-aggregated: lanes with the same digit as the first active lane elect it to add their count; repeat until all lanes served
__global__ void k_count_agg(const uint32_t *__restrict__ dig, uint64_t n, uint32_t *__restrict__ cnt) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool live = i < n;
    uint32_t d = live ? dig[i] : 0xffffffffu;
    while (true) {
        uint64_t todo = __ballot(live);
        if (!todo) break;
        int leader = __ffsll((long long)todo) - 1;
        uint32_t dl = __shfl(d, leader);
        uint64_t same = __ballot(live && d == dl);
        if (live && d == dl) {
            if ((int)(threadIdx.x & 63) == leader) atomicAdd(&cnt[dl], (uint32_t)__popcll(same));
            live = false;
        }
    }
}
__global__ void k_scatter(const uint32_t *__restrict__ dig, uint64_t n, uint32_t *__restrict__ cur, uint32_t *__restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[atomicAdd(&cur[dig[i]], 1u)] = (uint32_t)i;
}
__global__ void k_scatter_agg(const uint32_t *__restrict__ dig, uint64_t n, uint32_t *__restrict__ cur, uint32_t *__restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool live = i < n;
    uint32_t d = live ? dig[i] : 0xffffffffu;
    const int lane = threadIdx.x & 63;
    while (true) {
        uint64_t todo = __ballot(live);
        if (!todo) break;
        int leader = __ffsll((long long)todo) - 1;
        uint32_t dl = __shfl(d, leader);
        uint64_t same = __ballot(live && d == dl);
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&cur[dl], (uint32_t)__popcll(same));
        base = __shfl(base, leader);
        if (live && d == dl) {
            out[base + (uint32_t)__popcll(same & ((1ull << lane) - 1))] = (uint32_t)i;
            live = false;
        }
    }
}
__global__ void k_scan(const uint32_t *cnt, uint32_t *cur, uint32_t B) {   // one workgroup, serial per thread block of bins (diagnostic only)
    __shared__ uint32_t part[1024];
    uint32_t per = B / 1024, t = threadIdx.x, s = 0;
    for (uint32_t k = 0; k < per; k++) s += cnt[t * per + k];
    part[t] = s;
    __syncthreads();
    if (t == 0) { uint32_t a = 0; for (int k = 0; k < 1024; k++) { uint32_t v = part[k]; part[k] = a; a += v; } }
    __syncthreads();
    uint32_t a = part[t];
    for (uint32_t k = 0; k < per; k++) { cur[t * per + k] = a; a += cnt[t * per + k]; }
}

int main() {
    const uint32_t B = 1u << 19;
    uint32_t *cnt, *cur;
    CK(hipMalloc(&cnt, B * 4)); CK(hipMalloc(&cur, B * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (uint64_t n : {54525952ull, 218103808ull}) {
        uint32_t *dig, *out;
        CK(hipMalloc(&dig, n * 4)); CK(hipMalloc(&out, n * 4));
        for (int skew = 0; skew < 2; skew++) {
            hipLaunchKernelGGL(k_gen, (unsigned)((n + 255) / 256), 256, 0, 0, dig, n, B, skew);
            for (int agg = 0; agg < 2; agg++) {
                float ms_c = 0, ms_s = 0;
                for (int rep = 0; rep < 3; rep++) {
                    CK(hipMemset(cnt, 0, B * 4));
                    CK(hipEventRecord(e0, 0));
                    if (agg) hipLaunchKernelGGL(k_count_agg, (unsigned)((n + 255) / 256), 256, 0, 0, dig, n, cnt);
                    else hipLaunchKernelGGL(k_count, (unsigned)((n + 255) / 256), 256, 0, 0, dig, n, cnt);
                    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                    float m; CK(hipEventElapsedTime(&m, e0, e1)); if (rep) ms_c += m / 2;
                    hipLaunchKernelGGL(k_scan, 1, 1024, 0, 0, cnt, cur, B);
                    CK(hipEventRecord(e0, 0));
                    if (agg) hipLaunchKernelGGL(k_scatter_agg, (unsigned)((n + 255) / 256), 256, 0, 0, dig, n, cur, out);
                    else hipLaunchKernelGGL(k_scatter, (unsigned)((n + 255) / 256), 256, 0, 0, dig, n, cur, out);
                    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&m, e0, e1)); if (rep) ms_s += m / 2;
                }
                // spot check: every position written exactly once (sum of out == n(n-1)/2 mod 2^64 is enough for a diagnostic)
                printf("{\"entries\": %llu, \"digits\": \"%s\", \"wave_aggregated\": %d, \"count_ms\": %.3f, \"scatter_ms\": %.3f, \"entries_per_s_e9\": %.2f}\n",
                       (unsigned long long)n, skew ? "skewed" : "uniform", agg, ms_c, ms_s, n / ((ms_c + ms_s) * 1e-3) / 1e9);
                fflush(stdout);
            }
        }
        CK(hipFree(dig)); CK(hipFree(out));
    }
    return 0;
}
