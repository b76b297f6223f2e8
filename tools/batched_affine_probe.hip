// Diagnostic (not on the product path): what does BATCHED-AFFINE bucket accumulation cost on gfx950, measured, against the XYZZ
// mixed-addition kernel of csrc/msm_impl.inc (k_accumulate_chunks: 8M + 2S per addition, 96 bytes gathered per addition, 6.5 x 10^9
// additions/s on the 2^24-point commit)?
//
// Affine addition R = P + Q needs lambda = (y2 - y1) / (x2 - x1): 1 inversion + 2M + 1S.  The inversions of a whole level of a pairwise
// tree reduction are shared by Montgomery's trick — prefix products, ONE inversion, back-substitution: 3 more products per pair — so a
// pair costs 5M + 1S (+ its share of the one inversion) instead of 8M + 2S.  The price is memory: the pairs of a level are independent,
// but the prefix products have to wait for the inversion, so a level is two passes over its operands with the prefix products parked in
// between; and every level writes its sums back as affine points for the next one.
//
//   pass A  per pair: d = x2 - x1, running product over the lane's RUN consecutive pairs, prefix written (48 B / pair); lane total out
//   pass I  the lane totals are inverted together (the same trick one level up; here: serial per workgroup + one field inversion —
//           timed apart, it is 1 / RUN of the work and parallelises the same way)
//   pass B  per pair, backwards through the lane's run: 1 / d from the inverted total and the parked prefix; lambda, x3, y3; sum written
//
// Operands: level 0 GATHERS both points of a pair from a table by index (random rows of a table far larger than the Infinity Cache, as
// the sorted bucket lists address the 21 GB commit table) — in pass A and again in pass B; levels >= 1 read the previous level's sums
// in order.  A bucket of m entries costs m - 1 additions over log2(m) levels: half of all additions are level-0 additions.
// Field arithmetic: the library's saturated 12 x 32-bit Montgomery Fq (csrc/ff.h); the accumulate kernel's unsaturated 14 x 29-bit
// products are 1.3 x faster per product — the summary scales the arithmetic-bound figures by that, optimistically.
// Correctness: a sample of sums is recomputed pair by pair with its own inversion.
// build: hipcc -O3 --offload-arch=gfx950 -I tokamak-zk-evm_amd/csrc tools/batched_affine_probe.hip -o tools/batched_affine_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <vector>

#include "ec.h"

using Fq = ff<bls12_381_fq_params>;
using fq_t = Fq::E;
struct pt_t {
    fq_t x, y;
};
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define RUN 16   // consecutive pairs per lane

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 31; x *= 0xD6E8FEB86659FD93ull; x ^= x >> 29; x *= 0xD6E8FEB86659FD93ull; x ^= x >> 32;
    return x;
}
// table rows: pseudo-random field elements as coordinates (the arithmetic does not care whether (x, y) lies on the curve; distinct x
// keep every denominator non-zero)
__global__ void k_fill(pt_t *t, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    pt_t p;
    for (int j = 0; j < 12; j++) {
        p.x.l[j] = (uint32_t)mix(i * 24 + j + 1);
        p.y.l[j] = (uint32_t)mix(i * 24 + 12 + j + 1);
    }
    p.x.l[11] &= 0x0fffffffu, p.y.l[11] &= 0x0fffffffu;   // below p
    t[i] = p;
}
__global__ void k_index(uint32_t *idx, uint64_t n, uint32_t rows) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) idx[i] = (uint32_t)(mix(i + 0x9E3779B97F4A7C15ull) % rows);
}

// operands of pair i: level 0 -> table[idx[2 i]], table[idx[2 i + 1]]; level >= 1 -> in[2 i], in[2 i + 1]
template <bool GATHER>
__device__ __forceinline__ void operands(const pt_t *__restrict__ src, const uint32_t *__restrict__ idx, uint64_t i, pt_t &p, pt_t &q) {
    if (GATHER) p = src[idx[2 * i]], q = src[idx[2 * i + 1]];
    else p = src[2 * i], q = src[2 * i + 1];
}

template <bool GATHER>
__global__ __launch_bounds__(256) void k_pass_a(const pt_t *__restrict__ src, const uint32_t *__restrict__ idx, uint64_t pairs, fq_t *__restrict__ prefix,
                                                fq_t *__restrict__ totals) {
    const uint64_t lane = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, first = lane * RUN;
    if (first >= pairs) return;
    fq_t run = Fq::one();
    for (int j = 0; j < RUN && first + j < pairs; j++) {
        pt_t p, q;
        operands<GATHER>(src, idx, first + j, p, q);
        run = Fq::mul(run, Fq::sub(q.x, p.x));
        prefix[first + j] = run;
    }
    totals[lane] = run;
}
// the totals of `lanes` runs -> their inverses, in place: each lane of ONE workgroup walks a slice serially, the slice totals are
// inverted by lane 0 one by one (a real implementation recurses with pass A / pass B instead; this is 1 / RUN of the level's elements)
__global__ __launch_bounds__(256) void k_pass_i(fq_t *__restrict__ totals, uint64_t lanes, fq_t *__restrict__ scratch) {
    __shared__ fq_t slice_total[256];
    const uint32_t t = threadIdx.x;
    const uint64_t per = (lanes + 255) / 256, lo = t * per, hi = lo + per < lanes ? lo + per : lanes;
    fq_t run = Fq::one();
    for (uint64_t k = lo; k < hi; k++) {
        scratch[k] = run;              // product of the slice's earlier totals
        run = Fq::mul(run, totals[k]);
    }
    slice_total[t] = run;
    __syncthreads();
    if (t == 0) {                      // 256 values: prefix, one inversion, back-substitution
        fq_t pre[256], acc = Fq::one();
        for (int k = 0; k < 256; k++) pre[k] = acc, acc = Fq::mul(acc, slice_total[k]);
        fq_t inv = Fq::inv(acc);
        for (int k = 255; k >= 0; k--) {
            fq_t mine = Fq::mul(inv, pre[k]);
            inv = Fq::mul(inv, slice_total[k]);
            slice_total[k] = mine;
        }
    }
    __syncthreads();
    fq_t inv = slice_total[t];         // inverse of this slice's total
    for (uint64_t k = hi; k-- > lo;) {
        fq_t mine = Fq::mul(inv, scratch[k]);
        inv = Fq::mul(inv, totals[k]);
        totals[k] = mine;
    }
}
template <bool GATHER>
__global__ __launch_bounds__(256) void k_pass_b(const pt_t *__restrict__ src, const uint32_t *__restrict__ idx, uint64_t pairs, const fq_t *__restrict__ prefix,
                                                const fq_t *__restrict__ inv_totals, pt_t *__restrict__ out) {
    const uint64_t lane = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, first = lane * RUN;
    if (first >= pairs) return;
    const int cnt = pairs - first < RUN ? (int)(pairs - first) : RUN;
    fq_t inv = inv_totals[lane];       // 1 / (d_0 ... d_{cnt-1})
    for (int j = cnt - 1; j >= 0; j--) {
        pt_t p, q;
        operands<GATHER>(src, idx, first + j, p, q);
        const fq_t d = Fq::sub(q.x, p.x);
        const fq_t dinv = j ? Fq::mul(inv, prefix[first + j - 1]) : inv;   // 1 / d_j
        inv = Fq::mul(inv, d);
        const fq_t lam = Fq::mul(Fq::sub(q.y, p.y), dinv);
        pt_t r;
        r.x = Fq::sub(Fq::sub(Fq::sqr(lam), p.x), q.x);
        r.y = Fq::sub(Fq::mul(lam, Fq::sub(p.x, r.x)), p.y);
        out[first + j] = r;
    }
}
// reference for the check: one pair, its own inversion
template <bool GATHER>
__global__ void k_check(const pt_t *__restrict__ src, const uint32_t *__restrict__ idx, uint64_t pairs, const pt_t *__restrict__ got, uint32_t samples,
                        uint32_t *__restrict__ bad) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= samples) return;
    const uint64_t i = mix(s + 77) % pairs;
    pt_t p, q;
    operands<GATHER>(src, idx, i, p, q);
    const fq_t lam = Fq::mul(Fq::sub(q.y, p.y), Fq::inv(Fq::sub(q.x, p.x)));
    pt_t r;
    r.x = Fq::sub(Fq::sub(Fq::sqr(lam), p.x), q.x);
    r.y = Fq::sub(Fq::mul(lam, Fq::sub(p.x, r.x)), p.y);
    if (!Fq::eq(Fq::canon(r.x), Fq::canon(got[i].x)) || !Fq::eq(Fq::canon(r.y), Fq::canon(got[i].y))) atomicAdd(bad, 1u);
}

template <bool GATHER>
static int level(const char *name, const pt_t *src, const uint32_t *idx, uint64_t pairs, fq_t *prefix, fq_t *totals, fq_t *scratch, pt_t *out, uint32_t *bad,
                 double *ms_ab_out) {
    const uint64_t lanes = (pairs + RUN - 1) / RUN;
    const unsigned grid = (unsigned)((lanes + 255) / 256);
    hipEvent_t e[4];
    for (auto &x : e) CK(hipEventCreate(&x));
    float best_a = 1e30f, best_i = 1e30f, best_b = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        CK(hipEventRecord(e[0]));
        hipLaunchKernelGGL(k_pass_a<GATHER>, grid, 256, 0, 0, src, idx, pairs, prefix, totals);
        CK(hipEventRecord(e[1]));
        hipLaunchKernelGGL(k_pass_i, 1, 256, 0, 0, totals, lanes, scratch);
        CK(hipEventRecord(e[2]));
        hipLaunchKernelGGL(k_pass_b<GATHER>, grid, 256, 0, 0, src, idx, pairs, (const fq_t *)prefix, (const fq_t *)totals, out);
        CK(hipEventRecord(e[3]));
        CK(hipDeviceSynchronize());
        float a, i, b;
        CK(hipEventElapsedTime(&a, e[0], e[1]));
        CK(hipEventElapsedTime(&i, e[1], e[2]));
        CK(hipEventElapsedTime(&b, e[2], e[3]));
        if (rep) best_a = a < best_a ? a : best_a, best_i = i < best_i ? i : best_i, best_b = b < best_b ? b : best_b;
    }
    CK(hipMemset(bad, 0, 4));
    hipLaunchKernelGGL(k_check<GATHER>, 16, 256, 0, 0, src, idx, pairs, (const pt_t *)out, 4096u, bad);
    uint32_t nbad = 0;
    CK(hipMemcpy(&nbad, bad, 4, hipMemcpyDeviceToHost));
    const double ab = best_a + best_b;
    // bytes: pass A reads 2 points and writes a prefix; pass B reads 2 points and a prefix and writes a point (+ 4-byte indices when gathering)
    const double bytes = (double)pairs * (2 * 96 + 48 + 2 * 96 + 48 + 96 + (GATHER ? 16 : 0));
    printf("{\"level\": \"%s\", \"pairs\": %llu, \"pass_a_ms\": %.3f, \"pass_b_ms\": %.3f, \"pass_i_ms_one_workgroup\": %.3f, \"additions_per_s_e9_passes_a_b\": %.3f, "
           "\"bytes_moved_per_addition\": %.0f, \"GBps_passes_a_b\": %.0f, \"mismatches_in_4096_samples\": %u}\n",
           name, (unsigned long long)pairs, best_a, best_b, best_i, pairs / (ab * 1e-3) / 1e9, bytes / pairs, bytes / (ab * 1e-3) / 1e9, nbad);
    *ms_ab_out = ab;
    return nbad ? 2 : 0;
}

int main(int argc, char **argv) {
    const uint64_t pairs = argc > 1 ? strtoull(argv[1], nullptr, 0) : (1ull << 23);        // level 0: 2^24 entries
    const uint32_t rows = argc > 2 ? (uint32_t)strtoull(argv[2], nullptr, 0) : (1u << 26);    // 2^26 rows x 96 B = 6.4 GB table
    pt_t *table, *l0, *l1;
    uint32_t *idx, *bad;
    fq_t *prefix, *totals, *scratch;
    CK(hipMalloc(&table, (size_t)rows * sizeof(pt_t)));
    CK(hipMalloc(&idx, pairs * 2 * 4));
    CK(hipMalloc(&l0, pairs * sizeof(pt_t)));
    CK(hipMalloc(&l1, pairs / 2 * sizeof(pt_t)));
    CK(hipMalloc(&prefix, pairs * sizeof(fq_t)));
    CK(hipMalloc(&totals, (pairs / RUN + 1) * sizeof(fq_t)));
    CK(hipMalloc(&scratch, (pairs / RUN + 1) * sizeof(fq_t)));
    CK(hipMalloc(&bad, 4));
    hipLaunchKernelGGL(k_fill, (unsigned)(((uint64_t)rows + 255) / 256), 256, 0, 0, table, (uint64_t)rows);
    hipLaunchKernelGGL(k_index, (unsigned)((pairs * 2 + 255) / 256), 256, 0, 0, idx, pairs * 2, rows);
    CK(hipDeviceSynchronize());
    double ms0 = 0, ms1 = 0;
    int rc = level<true>("0: both operands gathered from the table by index", table, idx, pairs, prefix, totals, scratch, l0, bad, &ms0);
    rc |= level<false>("1+: operands are the previous level's sums, in order", l0, nullptr, pairs / 2, prefix, totals, scratch, l1, bad, &ms1);
    // a bucket list of 2 * pairs entries reduced to sums of 2^k entries each: level 0 has `pairs` additions, level j has pairs / 2^j
    const double rate0 = pairs / (ms0 * 1e-3), rate1 = (pairs / 2) / (ms1 * 1e-3);
    const double tree = 2.0 / (1.0 / rate0 + 1.0 / rate1);   // half of the additions at level 0's rate, half at the later levels'
    printf("{\"summary\": \"whole tree: half of the additions are level-0 additions\", \"additions_per_s_e9\": %.3f, \"with_unsaturated_products_x1.3_at_best_e9\": %.3f, "
           "\"xyzz_kernel_additions_per_s_e9\": 6.5, \"run_length\": %d, \"table_rows\": %u, \"note\": \"pass I excluded (1 / RUN of the elements; recursion)\"}\n",
           tree / 1e9, 1.3 * tree / 1e9, RUN, rows);
    return rc;
}
