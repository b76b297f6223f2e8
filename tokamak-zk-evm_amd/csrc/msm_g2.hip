// msm_g2.hip — G2 multi-scalar multiplication for BLS12-381 behind bls12_381_g2_msm (include/tkmk.h): Pippenger over the twist
// y^2 = x^3 + 4(1 + u) on Fp2 = Fq[u] / (u^2 + 1).
//
// The reference has NO G2 MSM call site: G2 appears as nine scalar multiplications in Sigma2::gen
// (packages/backend/libs/src/group_structures/mod.rs:752-777) and in the verifier's pairings.  BASELINE.json's north_star names
// "G1/G2", so the entry exists, with ICICLE's msm signature for the G2 curve (icicle_bls12_381::curve::G2CurveCfg), and Sigma2::gen
// can run through it as a batch of one-point MSMs.  It is not on the prover's hot path and is built for correctness and balance,
// not for the last 20 %: saturated 12-limb Montgomery arithmetic in a 3-product Karatsuba Fp2, the group law of csrc/ec.h
// instantiated over Fp2, windows of at most 12 bits (2048 buckets), a bucket sort with global atomics, and the chunked
// accumulate / fragment combine / segment reduction scheme of the G1 kernel (csrc/msm_impl.inc) so that skewed scalars (one giant
// bucket) stay parallel.  Bound: integer VALU, like G1 (an Fp2 product is 3 Fq products).
#include <cstring>
#include <vector>

#include "common.h"

// ---- Fp2 over a base field class F (csrc/ff.h interface), Montgomery form in both components ----
template <class F>
struct fp2 {
    struct E {
        typename F::E c0, c1;
    };
    static FF_HD E zero() { return E{F::zero(), F::zero()}; }
    static FF_HD E one() { return E{F::one(), F::zero()}; }
    static FF_HD bool is_zero(const E &a) { return F::is_zero(a.c0) && F::is_zero(a.c1); }
    static FF_HD bool eq(const E &a, const E &b) { return F::eq(a.c0, b.c0) && F::eq(a.c1, b.c1); }
    static FF_HD E neg(const E &a) { return E{F::neg(a.c0), F::neg(a.c1)}; }
    static FF_HD E add(const E &a, const E &b) { return E{F::add(a.c0, b.c0), F::add(a.c1, b.c1)}; }
    static FF_HD E sub(const E &a, const E &b) { return E{F::sub(a.c0, b.c0), F::sub(a.c1, b.c1)}; }
    static FF_HD E dbl(const E &a) { return E{F::dbl(a.c0), F::dbl(a.c1)}; }
    static FF_HD E mul(const E &a, const E &b) {   // Karatsuba: 3 base-field products
        typename F::E t0 = F::mul(a.c0, b.c0), t1 = F::mul(a.c1, b.c1);
        typename F::E m = F::mul(F::add(a.c0, a.c1), F::add(b.c0, b.c1));
        return E{F::sub(t0, t1), F::sub(F::sub(m, t0), t1)};
    }
    static FF_HD E sqr(const E &a) {                // (a0 + a1)(a0 - a1) + 2 a0 a1 u
        typename F::E p = F::mul(a.c0, a.c1);
        return E{F::mul(F::add(a.c0, a.c1), F::sub(a.c0, a.c1)), F::dbl(p)};
    }
    static FF_HD E inv(const E &a) {                // conj(a) / (a0^2 + a1^2)
        typename F::E n = F::inv(F::add(F::sqr(a.c0), F::sqr(a.c1)));
        return E{F::mul(a.c0, n), F::mul(F::neg(a.c1), n)};
    }
};
using Fq2 = fp2<Fq>;
using fq2_t = Fq2::E;
using G2 = ec<Fq2>;
using g2_affine_t = affine_t<Fq2>;   // 192 B
using g2_xyzz_t = xyzz_t<Fq2>;       // 384 B
static_assert(sizeof(g2_affine_t) == 192 && sizeof(g2_xyzz_t) == 384 && sizeof(tkmk_g2_affine) == 192 && sizeof(tkmk_g2_projective) == 288, "G2 layout");

struct g2_plan_t {
    uint32_t n, c, W, B, bits, chunk;
};
#define G2_SEG 8
#define G2_PARTS 4
#define G2_BIG 32

// plain (or Montgomery) affine records -> Montgomery; (0,0) stays infinity
__global__ __launch_bounds__(128) void k_g2_prepare(const g2_affine_t *__restrict__ in, g2_affine_t *__restrict__ out, uint64_t n, int in_mont) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    g2_affine_t p = tk_load(in + i);
    fq_t *c[4] = {&p.x.c0, &p.x.c1, &p.y.c0, &p.y.c1};
    for (int k = 0; k < 4; k++) {
        *c[k] = Fq::canon(*c[k]);
        if (!in_mont) *c[k] = Fq::to_mont(*c[k]);
    }
    tk_store(out + i, p);
}
// signed digits, [w][i] records: bit 31 = negative, low bits |d| in [1, B]; 0 = skip.  Also counts the buckets.
__global__ __launch_bounds__(256) void k_g2_digits(const fr_t *__restrict__ scalars, uint32_t *__restrict__ dig, uint32_t *__restrict__ counts,
                                                  g2_plan_t pl, int scalars_mont) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pl.n) return;
    fr_t s = Fr::canon(tk_load(scalars + i));
    if (scalars_mont) s = Fr::from_mont(s);
    uint32_t carry = 0;
    const uint32_t mask = (1u << pl.c) - 1u;
    for (uint32_t w = 0; w < pl.W; w++) {
        uint32_t lo = w * pl.c, li = lo >> 5, sh = lo & 31, raw = 0;
        if (li < 8) {
            raw = s.l[li] >> sh;
            if (sh + pl.c > 32 && li + 1 < 8) raw |= s.l[li + 1] << (32 - sh);
        }
        raw &= mask;
        if (lo + pl.c > pl.bits) {
            uint32_t keep = pl.bits > lo ? pl.bits - lo : 0;
            raw &= keep >= 32 ? 0xffffffffu : ((1u << keep) - 1u);
        }
        uint32_t v = raw + carry, rec;
        if (v > pl.B) {
            rec = ((1u << pl.c) - v) | 0x80000000u;
            carry = 1;
        } else {
            rec = v;
            carry = 0;
        }
        dig[(uint64_t)w * pl.n + i] = rec;
        if (rec & 0x7fffffffu) atomicAdd(&counts[(uint64_t)w * pl.B + (rec & 0x7fffffffu) - 1], 1u);
    }
}
// per window: bstart[w][b] = exclusive prefix of counts (b in [0, B]); cursors reset to the starts.  One block per window, B <= 2048.
__global__ __launch_bounds__(1024) void k_g2_scan(const uint32_t *__restrict__ counts, uint32_t *__restrict__ bstart, uint32_t *__restrict__ cursor, g2_plan_t pl) {
    __shared__ uint32_t part[1024];
    const uint32_t w = blockIdx.x, t = threadIdx.x;
    const uint32_t per = (pl.B + 1023) / 1024;   // 1 or 2
    uint32_t v[2] = {0, 0}, sum = 0;
    for (uint32_t k = 0; k < per; k++) {
        uint32_t b = t * per + k;
        v[k] = b < pl.B ? counts[(uint64_t)w * pl.B + b] : 0;
        sum += v[k];
    }
    part[t] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        uint32_t x = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += x;
        __syncthreads();
    }
    uint32_t run = part[t] - sum;
    for (uint32_t k = 0; k < per; k++) {
        uint32_t b = t * per + k;
        if (b < pl.B) {
            bstart[(uint64_t)w * (pl.B + 1) + b] = run;
            cursor[(uint64_t)w * pl.B + b] = run;
            run += v[k];
        }
    }
    if (t == 1023) bstart[(uint64_t)w * (pl.B + 1) + pl.B] = part[1023];
}
__global__ __launch_bounds__(256) void k_g2_fill(const uint32_t *__restrict__ dig, uint32_t *__restrict__ cursor, uint32_t *__restrict__ sorted, g2_plan_t pl) {
    uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (uint64_t)pl.W * pl.n) return;
    uint32_t w = (uint32_t)(e / pl.n), i = (uint32_t)(e - (uint64_t)w * pl.n);
    uint32_t rec = dig[e], r = rec & 0x7fffffffu;
    if (!r) return;
    uint32_t pos = atomicAdd(&cursor[(uint64_t)w * pl.B + r - 1], 1u);
    sorted[(uint64_t)w * pl.n + pos] = i | (rec & 0x80000000u);
}
__device__ __forceinline__ uint32_t g2_bucket_of(const uint32_t *__restrict__ bs, uint32_t B, uint32_t pos) {
    uint32_t lo = 0, hi = B;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (bs[mid] <= pos) lo = mid;
        else hi = mid;
    }
    return lo;
}
// one lane per chunk of the window's sorted list (the scheme of k_accumulate_chunks in msm_impl.inc): a chunk intersects at most one
// HEAD fragment, any number of whole buckets, one TAIL fragment
__global__ __launch_bounds__(64) void k_g2_accumulate(const g2_affine_t *__restrict__ bases, const uint32_t *__restrict__ sorted,
                                                     const uint32_t *__restrict__ bstart, g2_xyzz_t *__restrict__ buckets,
                                                     g2_xyzz_t *__restrict__ frag_head, g2_xyzz_t *__restrict__ frag_tail, g2_plan_t pl, uint32_t cpw) {
    uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= pl.W * cpw) return;
    uint32_t w = gid / cpw, t = gid - w * cpw;
    const uint32_t *bs = bstart + (uint64_t)w * (pl.B + 1);
    const uint32_t total = bs[pl.B];
    uint32_t lo = t * pl.chunk;
    if (lo >= total) return;
    uint32_t hi = lo + pl.chunk < total ? lo + pl.chunk : total;
    const uint32_t *s = sorted + (uint64_t)w * pl.n;
    uint32_t b = g2_bucket_of(bs, pl.B, lo);
    uint32_t bend = bs[b + 1];
    while (bend <= lo) bend = bs[++b + 1];
    const bool first_starts_here = bs[b] == lo;
    uint32_t seg_lo = lo;
    g2_xyzz_t acc = G2::inf();
    for (uint32_t k = lo; k < hi; k++) {
        uint32_t rec = s[k];
        g2_affine_t q = tk_load(bases + (rec & 0x7fffffffu));
        if (!G2::is_inf(q)) {
            if (rec & 0x80000000u) q = G2::neg(q);
            acc = G2::add_mixed(acc, q);
        }
        if (k + 1 == bend || k + 1 == hi) {
            bool whole = (seg_lo != lo || first_starts_here) && k + 1 == bend;
            g2_xyzz_t *dst = whole ? buckets + (uint64_t)w * pl.B + b : seg_lo == lo ? frag_head + (uint64_t)w * cpw + t : frag_tail + (uint64_t)w * cpw + t;
            tk_store(dst, acc);
            acc = G2::inf();
            seg_lo = k + 1;
            if (k + 1 < hi) {
                do {
                    b++;
                    bend = bs[b + 1];
                } while (bend <= k + 1);
            }
        }
    }
}
// one lane per (window, bucket): sum the bucket's chunk fragments; buckets with more than G2_BIG fragments are queued for k_g2_combine_big
__global__ __launch_bounds__(64) void k_g2_combine(const uint32_t *__restrict__ bstart, g2_xyzz_t *__restrict__ buckets, const g2_xyzz_t *__restrict__ frag_head,
                                                  const g2_xyzz_t *__restrict__ frag_tail, g2_plan_t pl, uint32_t cpw, uint32_t *__restrict__ big, uint32_t big_cap) {
    uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= pl.W * pl.B) return;
    uint32_t w = gid / pl.B, b = gid - w * pl.B;
    const uint32_t *bs = bstart + (uint64_t)w * (pl.B + 1);
    uint32_t sb = bs[b], eb = bs[b + 1];
    if (eb == sb) return;   // buckets[] was zeroed = infinity
    uint32_t t0 = sb / pl.chunk, t1 = (eb - 1) / pl.chunk;
    if (t0 == t1) return;   // wholly inside one chunk: already written
    if (t1 - t0 > G2_BIG) {
        uint32_t slot = atomicAdd(&big[0], 1u);
        if (slot < big_cap) {
            big[1 + slot] = gid;
            return;
        }
    }
    const g2_xyzz_t *fh = frag_head + (uint64_t)w * cpw, *ft = frag_tail + (uint64_t)w * cpw;
    g2_xyzz_t acc = tk_load(sb == t0 * pl.chunk ? fh + t0 : ft + t0);
    for (uint32_t t = t0 + 1; t <= t1; t++) acc = G2::add(acc, tk_load(fh + t));
    tk_store(buckets + gid, acc);
}
// queued big buckets, one workgroup each (grid-stride): 64 lanes sum strided fragments, then an LDS tree
__global__ __launch_bounds__(64) void k_g2_combine_big(const uint32_t *__restrict__ bstart, g2_xyzz_t *__restrict__ buckets, const g2_xyzz_t *__restrict__ frag_head,
                                                      const g2_xyzz_t *__restrict__ frag_tail, g2_plan_t pl, uint32_t cpw, const uint32_t *__restrict__ big, uint32_t big_cap) {
    __shared__ g2_xyzz_t sh[64];
    uint32_t count = big[0] < big_cap ? big[0] : big_cap;
    for (uint32_t i = blockIdx.x; i < count; i += gridDim.x) {
        uint32_t gid = big[1 + i], w = gid / pl.B, b = gid - w * pl.B;
        const uint32_t *bs = bstart + (uint64_t)w * (pl.B + 1);
        uint32_t sb = bs[b], eb = bs[b + 1], t0 = sb / pl.chunk, t1 = (eb - 1) / pl.chunk;
        const g2_xyzz_t *fh = frag_head + (uint64_t)w * cpw, *ft = frag_tail + (uint64_t)w * cpw;
        g2_xyzz_t acc = G2::inf();
        for (uint32_t t = t0 + threadIdx.x; t <= t1; t += 64) acc = G2::add(acc, tk_load((t == t0 && sb != t0 * pl.chunk) ? ft + t : fh + t));
        sh[threadIdx.x] = acc;
        __syncthreads();
        for (uint32_t st = 32; st > 0; st >>= 1) {
            if (threadIdx.x < st) {
                acc = G2::add(acc, sh[threadIdx.x + st]);
                sh[threadIdx.x] = acc;
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) tk_store(buckets + gid, acc);
        __syncthreads();
    }
}
// one lane per (window, segment of G2_SEG buckets): seg_out = sum_{v in segment} v * B_v = tot + v0 * run (running-sum trick)
__global__ __launch_bounds__(64) void k_g2_reduce_segments(const g2_xyzz_t *__restrict__ buckets, g2_xyzz_t *__restrict__ seg_out, g2_plan_t pl, uint32_t segs) {
    uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= pl.W * segs) return;
    uint32_t w = gid / segs, sg = gid - w * segs, v0 = sg * G2_SEG;
    uint32_t len = pl.B - v0 < G2_SEG ? pl.B - v0 : G2_SEG;
    const g2_xyzz_t *bk = buckets + (uint64_t)w * pl.B + v0;
    g2_xyzz_t run = G2::inf(), tot = G2::inf();
    for (int k = (int)len - 1; k >= 0; k--) {
        run = G2::add(run, tk_load(bk + k));
        tot = G2::add(tot, run);
    }
    if (v0) {
        g2_xyzz_t m = G2::inf();
        for (int bit = 31 - __builtin_clz(v0); bit >= 0; bit--) {
            m = G2::dbl(m);
            if ((v0 >> bit) & 1) m = G2::add(m, run);
        }
        tot = G2::add(tot, m);
    }
    tk_store(seg_out + gid, tot);
}
// grid (W, G2_PARTS), 64 lanes: partial window sums; the host adds the parts before its Horner step
__global__ __launch_bounds__(64) void k_g2_reduce_windows(const g2_xyzz_t *__restrict__ seg_in, g2_xyzz_t *__restrict__ win_out, uint32_t segs) {
    __shared__ g2_xyzz_t sh[64];
    const uint32_t w = blockIdx.x, part = blockIdx.y, t = threadIdx.x;
    const g2_xyzz_t *in = seg_in + (uint64_t)w * segs;
    const uint32_t per = (segs + G2_PARTS - 1) / G2_PARTS, lo = part * per, hi = lo + per < segs ? lo + per : segs;
    g2_xyzz_t acc = G2::inf();
    for (uint32_t k = lo + t; k < hi; k += 64) acc = G2::add(acc, tk_load(in + k));
    sh[t] = acc;
    __syncthreads();
    for (uint32_t s = 32; s > 0; s >>= 1) {
        if (t < s) {
            acc = G2::add(acc, sh[t + s]);
            sh[t] = acc;
        }
        __syncthreads();
    }
    if (t == 0) tk_store(win_out + (uint64_t)w * G2_PARTS + part, acc);
}

static void g2_store_canonical(tkmk_g2_projective *o, const g2_xyzz_t &r) {
    std::memset(o, 0, sizeof *o);
    if (G2::is_inf(r)) {
        o->y.c0.limbs[0] = 1;   // (0, 1, 0)
        return;
    }
    g2_affine_t a = G2::to_affine(r);
    fq_t c[4] = {Fq::from_mont(a.x.c0), Fq::from_mont(a.x.c1), Fq::from_mont(a.y.c0), Fq::from_mont(a.y.c1)};
    tkmk_fq *dst[4] = {&o->x.c0, &o->x.c1, &o->y.c0, &o->y.c1};
    for (int k = 0; k < 4; k++)
        for (int i = 0; i < Fq::N; i++) dst[k]->limbs[i] = c[k].l[i];
    o->z.c0.limbs[0] = 1;
}

static uint32_t g2_choose_c(uint32_t n) {   // minimise W * (n + 4 B) over c <= 12
    uint32_t best = 2;
    double best_cost = 1e300;
    for (uint32_t c = 2; c <= 12; c++) {
        double cost = (double)(255 / c + 1) * ((double)n + 4.0 * (double)(1u << (c - 1)));
        if (cost < best_cost) best_cost = cost, best = c;
    }
    return best;
}

// one MSM of n points on stream s: device scalars / Montgomery bases in, XYZZ result (Montgomery) out on the host
static tkmk_error g2_msm_one(const fr_t *scalars, const g2_affine_t *bases_mont, uint32_t n, uint32_t c_req, uint32_t bits, bool scalars_mont,
                             hipStream_t s, g2_xyzz_t *result) {
    g2_plan_t pl;
    pl.n = n, pl.bits = bits;
    pl.c = c_req ? c_req : g2_choose_c(n);
    if (pl.c < 2) pl.c = 2;
    if (pl.c > 12) pl.c = 12;
    pl.W = bits / pl.c + 1;
    pl.B = 1u << (pl.c - 1);
    pl.chunk = 32;   // Fp2 additions are ~3.5 x a G1 addition: shorter chains per lane, more lanes
    tk_frame frame(s);
    tk_scratch d_dig, d_sorted, d_counts, d_cursor, d_bstart, d_buckets, d_fh, d_ft, d_big, d_segs, d_wins;
    const uint32_t cpw = (n + pl.chunk - 1) / pl.chunk, segs = (pl.B + G2_SEG - 1) / G2_SEG, big_cap = 1u << 14;
    TK_TRY(d_dig.alloc((size_t)pl.W * n * 4, s));
    TK_TRY(d_sorted.alloc((size_t)pl.W * n * 4, s));
    TK_TRY(d_counts.alloc((size_t)pl.W * pl.B * 4, s));
    TK_TRY(d_cursor.alloc((size_t)pl.W * pl.B * 4, s));
    TK_TRY(d_bstart.alloc((size_t)pl.W * (pl.B + 1) * 4, s));
    TK_TRY(d_buckets.alloc((size_t)pl.W * pl.B * sizeof(g2_xyzz_t), s));
    TK_TRY(d_fh.alloc((size_t)pl.W * cpw * sizeof(g2_xyzz_t), s));
    TK_TRY(d_ft.alloc((size_t)pl.W * cpw * sizeof(g2_xyzz_t), s));
    TK_TRY(d_big.alloc((size_t)(big_cap + 1) * 4, s));
    TK_TRY(d_segs.alloc((size_t)pl.W * segs * sizeof(g2_xyzz_t), s));
    TK_TRY(d_wins.alloc((size_t)pl.W * G2_PARTS * sizeof(g2_xyzz_t), s));
    TK_HIP(hipMemsetAsync(d_counts.p, 0, (size_t)pl.W * pl.B * 4, s));
    TK_HIP(hipMemsetAsync(d_buckets.p, 0, (size_t)pl.W * pl.B * sizeof(g2_xyzz_t), s));   // all-zero record = infinity (zz = 0)
    TK_HIP(hipMemsetAsync(d_big.p, 0, 4, s));
    hipLaunchKernelGGL(k_g2_digits, tk_div_up(n, 256), 256, 0, s, scalars, d_dig.as<uint32_t>(), d_counts.as<uint32_t>(), pl, scalars_mont ? 1 : 0);
    hipLaunchKernelGGL(k_g2_scan, pl.W, 1024, 0, s, (const uint32_t *)d_counts.p, d_bstart.as<uint32_t>(), d_cursor.as<uint32_t>(), pl);
    hipLaunchKernelGGL(k_g2_fill, tk_div_up((uint64_t)pl.W * n, 256), 256, 0, s, (const uint32_t *)d_dig.p, d_cursor.as<uint32_t>(), d_sorted.as<uint32_t>(), pl);
    hipLaunchKernelGGL(k_g2_accumulate, tk_div_up((uint64_t)pl.W * cpw, 64), 64, 0, s, bases_mont, (const uint32_t *)d_sorted.p, (const uint32_t *)d_bstart.p,
                       d_buckets.as<g2_xyzz_t>(), d_fh.as<g2_xyzz_t>(), d_ft.as<g2_xyzz_t>(), pl, cpw);
    hipLaunchKernelGGL(k_g2_combine, tk_div_up((uint64_t)pl.W * pl.B, 64), 64, 0, s, (const uint32_t *)d_bstart.p, d_buckets.as<g2_xyzz_t>(),
                       (const g2_xyzz_t *)d_fh.p, (const g2_xyzz_t *)d_ft.p, pl, cpw, d_big.as<uint32_t>(), big_cap);
    hipLaunchKernelGGL(k_g2_combine_big, 512, 64, 0, s, (const uint32_t *)d_bstart.p, d_buckets.as<g2_xyzz_t>(), (const g2_xyzz_t *)d_fh.p,
                       (const g2_xyzz_t *)d_ft.p, pl, cpw, (const uint32_t *)d_big.p, big_cap);
    hipLaunchKernelGGL(k_g2_reduce_segments, tk_div_up((uint64_t)pl.W * segs, 64), 64, 0, s, (const g2_xyzz_t *)d_buckets.p, d_segs.as<g2_xyzz_t>(), pl, segs);
    hipLaunchKernelGGL(k_g2_reduce_windows, dim3(pl.W, G2_PARTS), 64, 0, s, (const g2_xyzz_t *)d_segs.p, d_wins.as<g2_xyzz_t>(), segs);
    TK_HIP(hipGetLastError());
    std::vector<g2_xyzz_t> wins((size_t)pl.W * G2_PARTS);
    TK_HIP(hipMemcpyAsync(wins.data(), d_wins.p, wins.size() * sizeof(g2_xyzz_t), hipMemcpyDeviceToHost, s));
    TK_HIP(hipStreamSynchronize(s));
    g2_xyzz_t acc = G2::inf();   // Horner over the window sums on the host
    for (int w = (int)pl.W - 1; w >= 0; w--) {
        for (uint32_t k = 0; k < pl.c; k++) acc = G2::dbl(acc);
        for (int part = 0; part < G2_PARTS; part++) acc = G2::add(acc, wins[(size_t)w * G2_PARTS + part]);
    }
    *result = acc;
    return TKMK_SUCCESS;
}

// results[b] = sum_i scalars[b * msm_size + i] * bases[(shared ? 0 : b * msm_size) + i] on G2; config fields as in bls12_381_msm
// (precompute_factor must be 1, c in 0 | 2..12).  Results: canonical projective (x_affine, y_affine, 1) or (0, 1, 0).
TK_API tkmk_error bls12_381_g2_msm(const tkmk_fr *scalars, const tkmk_g2_affine *bases, int msm_size, const tkmk_msm_config *cfg,
                                   tkmk_g2_projective *results) {
    if (!cfg || cfg->ext || cfg->precompute_factor > 1) return TKMK_ERR_INVALID_ARGUMENT;
    if (msm_size < 0 || cfg->batch_size < 1 || cfg->bitsize < 0 || cfg->bitsize > 255 || cfg->c < 0 || cfg->c > 12 || cfg->c == 1) return TKMK_ERR_INVALID_ARGUMENT;
    if (!results) return TKMK_ERR_INVALID_POINTER;
    TK_TRY(tk_require_device());
    const uint32_t n = (uint32_t)msm_size, batch = (uint32_t)cfg->batch_size;
    const uint32_t bits = cfg->bitsize ? (uint32_t)cfg->bitsize : 255u;
    hipStream_t s = tk_stream(cfg->stream_handle);
    tk_frame frame(s);
    std::vector<tkmk_g2_projective> host_res(batch);
    if (n == 0) {
        for (uint32_t b = 0; b < batch; b++) g2_store_canonical(&host_res[b], G2::inf());
    } else {
        if (!scalars || !bases) return TKMK_ERR_INVALID_POINTER;
        const size_t n_bases = cfg->are_points_shared_in_batch ? (size_t)n : (size_t)n * batch;
        tk_staged S, P;
        TK_TRY(S.in(scalars, (size_t)n * batch * sizeof(fr_t), cfg->are_scalars_on_device, s));
        TK_TRY(P.in(bases, n_bases * sizeof(g2_affine_t), cfg->are_points_on_device, s));
        tk_scratch d_bm;
        TK_TRY(d_bm.alloc(n_bases * sizeof(g2_affine_t), s));
        hipLaunchKernelGGL(k_g2_prepare, tk_div_up(n_bases, 128), 128, 0, s, (const g2_affine_t *)P.dev, d_bm.as<g2_affine_t>(), (uint64_t)n_bases,
                           cfg->are_points_montgomery_form ? 1 : 0);
        TK_HIP(hipGetLastError());
        for (uint32_t b = 0; b < batch; b++) {
            g2_xyzz_t r;
            TK_TRY(g2_msm_one((const fr_t *)S.dev + (size_t)b * n, d_bm.as<g2_affine_t>() + (cfg->are_points_shared_in_batch ? 0 : (size_t)b * n), n, (uint32_t)cfg->c,
                              bits, cfg->are_scalars_montgomery_form, s, &r));
            g2_store_canonical(&host_res[b], r);
        }
    }
    if (cfg->are_results_on_device) {
        TK_HIP(hipMemcpyAsync(results, host_res.data(), (size_t)batch * sizeof(tkmk_g2_projective), hipMemcpyHostToDevice, s));
        TK_HIP(hipStreamSynchronize(s));
    } else {
        for (uint32_t b = 0; b < batch; b++) results[b] = host_res[b];
    }
    return TKMK_SUCCESS;
}
