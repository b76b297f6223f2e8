// msm.hip — BLS12-381 G1 multi-scalar multiplication (Pippenger bucket method) for gfx950 behind
// bls12_381_msm (include/tkmk.h).  Work-alike of icicle_core::msm::msm as the reference calls it
// (packages/backend/libs/src/iotools/mod.rs:2093-2099 encode_poly; group_structures/mod.rs:108-143).
//
// Pipeline (all on one stream, no host round trip until the W window sums come back):
//   k_convert_bases   plain affine -> Montgomery affine, once per call (skipped for Montgomery input)
//   k_digits          scalar -> W signed c-bit digits (|d| <= 2^(c-1)), coalesced [w][i] u32 records
//   k_hist            per (window, chunk) workgroup: LDS-privatised bucket histogram (<= 128 KiB of LDS)
//   k_scan_local/apply  chunk-exclusive cursors + bucket start offsets (two-level scan over 1024-bucket groups)
//   k_scatter         per (window, chunk): LDS cursors, ds_add_rtn ranks -> bucket-sorted point indices
//   k_accumulate_chunks  one lane per 64-entry chunk of the sorted list: XYZZ accumulator in VGPRs += gathered
//                     affine bases (8M+2S each); k_combine / k_combine_big sum each bucket's chunk fragments
//   k_reduce_segments / k_reduce_windows   sum_v v*B_v per window via 16-bucket running sums
//   host              Horner over the W window sums (W*c doublings) + one inversion -> canonical result
// Sorting by bucket instead of atomically adding points: there are no 384-bit atomics, and the sorted
// order makes each bucket a private serial chain with no inter-lane communication.
//
// Bound: integer VALU (v_mad_u64_u32): ~4.8e4 mul-adds per point-window vs 128 B of traffic; not HBM,
// and MFMA is not applicable.
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "common.h"
#include "ec_u.h"

// curve arithmetic of the small latency-bound kernels (combine / reduce / size-1): saturated ec.h, fully inlined
using G1K = G1;
using FqK = Fq;
using g1k_xyzz = g1_xyzz_t;
using g1k_aff = g1_affine_t;

static_assert(sizeof(g1_affine_t) == 96 && sizeof(g1_xyzz_t) == 192, "layout");

struct msm_plan_t {
    uint32_t n;        // points
    uint32_t c;        // window bits
    uint32_t W;        // windows
    uint32_t B;        // buckets per window = 2^(c-1)
    uint32_t chunks;   // histogram chunks per window
    uint32_t chunk_len;
    uint32_t bits;
};

// bases -> the form the accumulate loop consumes: {x 2^406 mod p, y 2^406 mod p} (canonical, 32-bit limbs; the
// Montgomery radix of the unsaturated representation, ffu.h); (0,0) stays (0,0) = infinity
__global__ __launch_bounds__(256) void k_convert_bases(const g1_affine_t *__restrict__ in, g1_affine_t *__restrict__ out,
                                                      uint64_t n, int in_montgomery) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    g1_affine_t p = tk_load(in + i);
    if (!G1::is_inf(p)) {
        fq_t k;
#pragma unroll
        for (int j = 0; j < 12; j++) k.l[j] = in_montgomery ? bls12_381_fq_params::KSATM[j] : bls12_381_fq_params::KSAT[j];
        p.x = Fq::mul(Fq::canon(p.x), k);
        p.y = Fq::mul(Fq::canon(p.y), k);
    }
    tk_store(out + i, p);
}

// digit record: bit 31 = negative, low bits = |d| in [1, B]; 0 = skip
__global__ __launch_bounds__(256) void k_digits(const fr_t *__restrict__ scalars, uint32_t *__restrict__ dig, msm_plan_t pl,
                                               int scalars_mont) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pl.n) return;
    fr_t s = Fr::canon(tk_load(scalars + i));
    if (scalars_mont) s = Fr::from_mont(s);
    uint32_t carry = 0;
    const uint32_t mask = (1u << pl.c) - 1u;
    for (uint32_t w = 0; w < pl.W; w++) {
        uint32_t lo = w * pl.c, li = lo >> 5, sh = lo & 31;
        uint32_t raw = 0;
        if (li < 8) {
            raw = s.l[li] >> sh;
            if (sh + pl.c > 32 && li + 1 < 8) raw |= s.l[li + 1] << (32 - sh);
        }
        raw &= mask;
        if (lo + pl.c > pl.bits) {  // drop bits above `bits`
            uint32_t keep = pl.bits > lo ? pl.bits - lo : 0;
            raw &= keep >= 32 ? 0xffffffffu : ((1u << keep) - 1u);
        }
        uint32_t v = raw + carry;
        uint32_t rec;
        if (v > pl.B) {
            rec = ((1u << pl.c) - v) | 0x80000000u;
            carry = 1;
        } else {
            rec = v;
            carry = 0;
        }
        dig[(uint64_t)w * pl.n + i] = rec;
    }
}

// grid (chunks, W); dynamic LDS = B * 4 bytes
__global__ __launch_bounds__(1024) void k_hist(const uint32_t *__restrict__ dig, uint32_t *__restrict__ counts, msm_plan_t pl,
                                              uint32_t shift) {
    extern __shared__ uint32_t hist[];
    const uint32_t w = blockIdx.y, ch = blockIdx.x;
    for (uint32_t b = threadIdx.x; b < pl.B; b += blockDim.x) hist[b] = 0;
    __syncthreads();
    uint32_t lo = ch * pl.chunk_len, hi = lo + pl.chunk_len;
    if (hi > pl.n) hi = pl.n;
    const uint32_t *d = dig + (uint64_t)w * pl.n;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        uint32_t r = d[i] & 0x7fffffffu;
        if (r) atomicAdd(&hist[(r - 1) >> shift], 1u);
    }
    __syncthreads();
    // counts[w][b][chunk]
    uint32_t *c = counts + (uint64_t)w * pl.B * pl.chunks;
    for (uint32_t b = threadIdx.x; b < pl.B; b += blockDim.x) c[(uint64_t)b * pl.chunks + ch] = hist[b];
}

// counts[w][b][chunk] -> exclusive cursors (absolute position in the window's sorted list) and bstart[w][b],
// b in [0, B], in two launches over (ceil(B/1024), W) workgroups:
//   k_scan_local : one lane per bucket sums its chunks; workgroup-exclusive scan; group total -> gtot[w][grp]
//   k_scan_apply : adds the exclusive prefix of the (<= 32) group totals, writes bstart and the cursors
__global__ __launch_bounds__(1024) void k_scan_local(const uint32_t *__restrict__ counts, uint32_t *__restrict__ local,
                                                    uint32_t *__restrict__ gtot, msm_plan_t pl) {
    __shared__ uint32_t part[1024];
    const uint32_t w = blockIdx.y, t = threadIdx.x, b = blockIdx.x * 1024 + t;
    uint32_t sum = 0;
    if (b < pl.B) {
        const uint32_t *c = counts + ((uint64_t)w * pl.B + b) * pl.chunks;
        for (uint32_t k = 0; k < pl.chunks; k++) sum += c[k];
    }
    part[t] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        uint32_t v = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    if (b < pl.B) local[(uint64_t)w * pl.B + b] = part[t] - sum;  // exclusive within the group
    if (t == 1023) gtot[w * gridDim.x + blockIdx.x] = part[t];
}
__global__ __launch_bounds__(1024) void k_scan_apply(uint32_t *__restrict__ counts, const uint32_t *__restrict__ local,
                                                    const uint32_t *__restrict__ gtot, uint32_t *__restrict__ bstart, msm_plan_t pl) {
    const uint32_t w = blockIdx.y, t = threadIdx.x, b = blockIdx.x * 1024 + t;
    uint32_t base = 0, total = 0;
    for (uint32_t g = 0; g < gridDim.x; g++) {
        uint32_t v = gtot[w * gridDim.x + g];
        if (g < blockIdx.x) base += v;
        total += v;
    }
    uint32_t *bs = bstart + (uint64_t)w * (pl.B + 1);
    if (b < pl.B) {
        uint32_t run = base + local[(uint64_t)w * pl.B + b];
        bs[b] = run;
        uint32_t *c = counts + ((uint64_t)w * pl.B + b) * pl.chunks;
        for (uint32_t k = 0; k < pl.chunks; k++) {
            uint32_t v = c[k];
            c[k] = run;
            run += v;
        }
    }
    if (blockIdx.x == 0 && t == 0) bs[pl.B] = total;
}

// grid (chunks, W); dynamic LDS = B * 4 bytes.  sorted[w][pos] = point index | sign
__global__ __launch_bounds__(1024) void k_scatter(const uint32_t *__restrict__ dig, const uint32_t *__restrict__ counts,
                                                 uint32_t *__restrict__ sorted, msm_plan_t pl) {
    extern __shared__ uint32_t cur[];
    const uint32_t w = blockIdx.y, ch = blockIdx.x;
    const uint32_t *c = counts + (uint64_t)w * pl.B * pl.chunks;
    for (uint32_t b = threadIdx.x; b < pl.B; b += blockDim.x) cur[b] = c[(uint64_t)b * pl.chunks + ch];
    __syncthreads();
    uint32_t lo = ch * pl.chunk_len, hi = lo + pl.chunk_len;
    if (hi > pl.n) hi = pl.n;
    const uint32_t *d = dig + (uint64_t)w * pl.n;
    uint32_t *s = sorted + (uint64_t)w * pl.n;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        uint32_t rec = d[i];
        uint32_t r = rec & 0x7fffffffu;
        if (r) {
            uint32_t pos = atomicAdd(&cur[r - 1], 1u);
            s[pos] = i | (rec & 0x80000000u);
        }
    }
}

// ---- two-pass (MSD) bucket sort with LDS-staged, coalesced scatters -----------------------------------------
// The single-pass k_scatter issues one 4-byte store per item into 2^15 x chunks different 64-byte granules: 2^28
// partial-line writes that the L2 / fabric retire at ~40 G/s chip-wide (6.6 ms at 2^24 points, 8x write
// amplification).  Splitting the bucket index into a coarse part (<= 256 bins) and a fine part (128 bins) lets
// each pass sort a 4096-item tile by bin inside LDS first, so that a wave writes runs of consecutive addresses.
//   pass A: per (chunk, window): bins = bucket >> f;  out: A_idx = point | sign, A_key = bucket & (2^f - 1), grouped by coarse bin
//   pass B: per (part, (window, coarse bin)): bins = A_key;  out: the final bucket-sorted index list       (f = 7 or 8)
#define SC_TILE 4096
#define SC_MAXBINS 512   // bins per pass: coarse <= 512 (c <= 18 with 8 fine bits), fine <= 256
#define SC_PARTS 4

template <bool PASS_B>
__global__ __launch_bounds__(1024) void k_scatter_staged(const uint32_t *__restrict__ in0, const uint16_t *__restrict__ in1,
                                                        const uint32_t *__restrict__ seg_start, const uint32_t *__restrict__ cursors,
                                                        uint32_t *__restrict__ out0, uint16_t *__restrict__ out1, uint32_t n,
                                                        uint32_t bins, uint32_t coarse_bins, uint32_t chunk_len, uint32_t fbits) {
    __shared__ uint32_t hist[SC_MAXBINS], start[SC_MAXBINS], cur[SC_MAXBINS];
    const uint32_t fmask = (1u << fbits) - 1;
    __shared__ uint32_t sv0[SC_TILE], sv1[SC_TILE];
    const uint32_t tid = threadIdx.x, part = blockIdx.x, parts = gridDim.x, sgm = blockIdx.y;
    const uint32_t w = PASS_B ? sgm / coarse_bins : sgm;
    uint32_t lo, hi, base = 0;
    if (PASS_B) {
        const uint32_t cb = sgm - w * coarse_bins;
        const uint32_t *cs = seg_start + (uint64_t)w * (coarse_bins + 1);
        uint32_t s0 = cs[cb], s1 = cs[cb + 1];
        uint32_t per = (s1 - s0 + parts - 1) / parts;
        lo = s0 + part * per;
        hi = lo + per < s1 ? lo + per : s1;
        if (lo > s1) lo = s1;
        base = s0;
    } else {
        lo = part * chunk_len;
        hi = lo + chunk_len < n ? lo + chunk_len : n;
        if (lo > n) lo = n;
    }
    in0 += (uint64_t)w * n;
    if (PASS_B) in1 += (uint64_t)w * n;
    out0 += (uint64_t)w * n;
    if (!PASS_B) out1 += (uint64_t)w * n;
    if (tid < bins) cur[tid] = cursors[((uint64_t)sgm * bins + tid) * parts + part] + base;
    __syncthreads();
    for (uint32_t tlo = lo; tlo < hi; tlo += SC_TILE) {
        if (tid < SC_MAXBINS) hist[tid] = 0;
        __syncthreads();
        uint32_t v0[4], v1[4], rk[4];
        bool ok[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint32_t k = tlo + tid + 1024u * j;
            ok[j] = k < hi;
            v0[j] = v1[j] = rk[j] = 0;
            if (ok[j]) {
                if (PASS_B) {
                    v0[j] = in0[k];
                    v1[j] = in1[k];
                } else {
                    uint32_t rec = in0[k], r = rec & 0x7fffffffu;
                    ok[j] = r != 0;
                    v0[j] = k | (rec & 0x80000000u);
                    v1[j] = r - 1;
                }
            }
            if (ok[j]) rk[j] = atomicAdd(&hist[PASS_B ? (v1[j] & fmask) : (v1[j] >> fbits)], 1u);
        }
        __syncthreads();
        // exclusive prefix of hist[] -> start[] (Hillis-Steele over SC_MAXBINS lanes)
        uint32_t mine = tid < SC_MAXBINS ? hist[tid] : 0;
        if (tid < SC_MAXBINS) start[tid] = mine;
        __syncthreads();
        for (uint32_t off = 1; off < SC_MAXBINS; off <<= 1) {
            uint32_t t = (tid < SC_MAXBINS && tid >= off) ? start[tid - off] : 0;
            __syncthreads();
            if (tid < SC_MAXBINS) start[tid] += t;
            __syncthreads();
        }
        const uint32_t total = start[SC_MAXBINS - 1];
        __syncthreads();
        if (tid < SC_MAXBINS) start[tid] -= mine;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (ok[j]) {
                uint32_t b = PASS_B ? (v1[j] & fmask) : (v1[j] >> fbits);
                uint32_t pos = start[b] + rk[j];
                sv0[pos] = v0[j];
                sv1[pos] = v1[j];
            }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint32_t k2 = tid + 1024u * j;
            if (k2 < total) {
                uint32_t a0 = sv0[k2], a1 = sv1[k2];
                uint32_t b = PASS_B ? (a1 & fmask) : (a1 >> fbits);
                uint32_t g = cur[b] + (k2 - start[b]);
                out0[g] = a0;
                if (!PASS_B) out1[g] = (uint16_t)(a1 & fmask);   // pass B only needs the fine part
            }
        }
        __syncthreads();
        if (tid < SC_MAXBINS) cur[tid] += hist[tid];
        __syncthreads();
    }
}

// pass-B histogram: counts[(w * coarse_bins + cb)][fine][part]
__global__ __launch_bounds__(1024) void k_hist_fine(const uint16_t *__restrict__ key, const uint32_t *__restrict__ seg_start,
                                                   uint32_t *__restrict__ counts, uint32_t n, uint32_t coarse_bins, uint32_t fbits) {
    __shared__ uint32_t hist[256];
    const uint32_t fb_n = 1u << fbits;
    const uint32_t tid = threadIdx.x, part = blockIdx.x, parts = gridDim.x, sgm = blockIdx.y;
    const uint32_t w = sgm / coarse_bins, cb = sgm - w * coarse_bins;
    const uint32_t *cs = seg_start + (uint64_t)w * (coarse_bins + 1);
    uint32_t s0 = cs[cb], s1 = cs[cb + 1];
    uint32_t per = (s1 - s0 + parts - 1) / parts;
    uint32_t lo = s0 + part * per, hi = lo + per < s1 ? lo + per : s1;
    if (tid < fb_n) hist[tid] = 0;
    __syncthreads();
    const uint16_t *k = key + (uint64_t)w * n;
    for (uint32_t i = lo + tid; i < hi; i += 1024) atomicAdd(&hist[k[i] & (fb_n - 1)], 1u);
    __syncthreads();
    if (tid < fb_n) counts[((uint64_t)sgm * fb_n + tid) * parts + part] = hist[tid];
}

// bstart[w][cb * 128 + fb] = coarse_start[w][cb] + fine_start[w * CB + cb][fb];  bstart[w][B] = total
__global__ __launch_bounds__(256) void k_bstart_assemble(const uint32_t *__restrict__ coarse_start, const uint32_t *__restrict__ fine_start,
                                                        uint32_t *__restrict__ bstart, uint32_t W, uint32_t B, uint32_t coarse_bins,
                                                        uint32_t fbits) {
    uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (uint64_t)W * (B + 1)) return;
    uint32_t w = (uint32_t)(e / (B + 1)), b = (uint32_t)(e - (uint64_t)w * (B + 1));
    const uint32_t *cs = coarse_start + (uint64_t)w * (coarse_bins + 1);
    if (b == B) {
        bstart[e] = cs[coarse_bins];
        return;
    }
    uint32_t cb = b >> fbits, fb = b & ((1u << fbits) - 1);
    bstart[e] = cs[cb] + fine_start[((uint64_t)w * coarse_bins + cb) * ((1u << fbits) + 1) + fb];
}

// ---- bucket accumulation, balanced by construction -------------------------------------------------
// The window's bucket-sorted list is cut into chunks of MSM_CHUNK consecutive entries and every lane sums
// exactly one chunk (an XYZZ accumulator in VGPRs += gathered affine base, 8M + 2S per entry), whatever
// the bucket sizes are: real prover inputs are heavily skewed (witness wires are mostly 0/1: SURVEY.md
// Appendix B) and one-lane-per-bucket would serialise a 10^5-entry bucket on a single lane.
// A chunk intersects buckets in at most: one HEAD fragment (a bucket that covers the chunk start but is not
// wholly inside the chunk), any number of whole buckets (written straight to buckets[]), one TAIL fragment
// (a bucket that starts inside and runs past the chunk end).  k_combine then sums each bucket's fragments
// (tail of its first chunk, heads of the following chunks); buckets with more than MSM_BIG fragments are
// queued for k_combine_big (one workgroup per bucket, LDS tree).
#define MSM_CHUNK 64
#define MSM_BIG 48

__device__ __forceinline__ uint32_t bucket_of(const uint32_t *__restrict__ bs, uint32_t B, uint32_t pos) {
    // largest b in [0, B) with bs[b] <= pos  (bs is non-decreasing, bs[0] = 0)
    uint32_t lo = 0, hi = B;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (bs[mid] <= pos) lo = mid;
        else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void k_accumulate_chunks(const g1_affine_t *__restrict__ bases, const uint32_t *__restrict__ sorted,
                                                          const uint32_t *__restrict__ bstart, g1_xyzz_t *__restrict__ buckets,
                                                          g1_xyzz_t *__restrict__ frag_head, g1_xyzz_t *__restrict__ frag_tail,
                                                          msm_plan_t pl, uint32_t chunks_per_window) {
    uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= pl.W * chunks_per_window) return;
    uint32_t w = gid / chunks_per_window, t = gid - w * chunks_per_window;
    const uint32_t *bs = bstart + (uint64_t)w * (pl.B + 1);
    const uint32_t total = bs[pl.B];
    uint32_t lo = t * MSM_CHUNK;
    if (lo >= total) return;
    uint32_t hi = lo + MSM_CHUNK < total ? lo + MSM_CHUNK : total;
    const uint32_t *s = sorted + (uint64_t)w * pl.n;
    uint32_t b = bucket_of(bs, pl.B, lo);
    uint32_t bend = bs[b + 1];           // end of the current bucket (> lo by construction of bucket_of,
    while (bend <= lo) bend = bs[++b + 1];  //  except across empty buckets sharing the same offset)
    uint32_t seg_lo = lo;
    G1U::X acc = G1U::inf();
    for (uint32_t k = lo; k < hi; k++) {
        uint32_t rec = s[k];
        G1U::A q;
        if (G1U::load_affine(q, tk_load(bases + (rec & 0x7fffffffu)))) {
            if (rec & 0x80000000u) q = G1U::neg(q);
            acc = G1U::add_mixed(acc, q);
        }
        if (k + 1 == bend || k + 1 == hi) {  // the segment [seg_lo, k+1) of bucket b ends here
            uint32_t bbeg = bs[b];
            bool whole = seg_lo == bbeg && k + 1 == bend;
            g1_xyzz_t *dst;
            if (whole) dst = buckets + (uint64_t)w * pl.B + b;
            else if (seg_lo == lo) dst = frag_head + (uint64_t)w * chunks_per_window + t;
            else dst = frag_tail + (uint64_t)w * chunks_per_window + t;
            tk_store(dst, G1U::to_sat(acc));
            acc = G1U::inf();
            seg_lo = k + 1;
            if (k + 1 < hi) {
                do { b++; bend = bs[b + 1]; } while (bend <= k + 1);
            }
        }
    }
}

// one lane per (window, bucket): gather the bucket's fragments.  big[0] = count, big[1 + i] = w * B + b
__global__ __launch_bounds__(256) void k_combine(const uint32_t *__restrict__ bstart, g1_xyzz_t *__restrict__ buckets,
                                                const g1_xyzz_t *__restrict__ frag_head, const g1_xyzz_t *__restrict__ frag_tail,
                                                msm_plan_t pl, uint32_t chunks_per_window, uint32_t *__restrict__ big,
                                                uint32_t big_cap) {
    uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= pl.W * pl.B) return;
    uint32_t w = gid / pl.B, b = gid - w * pl.B;
    const uint32_t *bs = bstart + (uint64_t)w * (pl.B + 1);
    uint32_t sb = bs[b], eb = bs[b + 1];
    if (eb == sb) return;  // buckets[] was zeroed = infinity
    uint32_t t0 = sb / MSM_CHUNK, t1 = (eb - 1) / MSM_CHUNK;
    if (t0 == t1) return;  // wholly inside one chunk: already written
    if (t1 - t0 > MSM_BIG) {
        uint32_t slot = atomicAdd(&big[0], 1u);
        if (slot < big_cap) big[1 + slot] = gid;
        return;
    }
    const g1_xyzz_t *fh = frag_head + (uint64_t)w * chunks_per_window, *ft = frag_tail + (uint64_t)w * chunks_per_window;
    g1_xyzz_t acc = tk_load(sb == t0 * MSM_CHUNK ? fh + t0 : ft + t0);
    for (uint32_t t = t0 + 1; t <= t1; t++) acc = G1::add(acc, tk_load(fh + t));
    tk_store(buckets + gid, acc);
}

// grid-stride over the queued big buckets, one workgroup each: 256 lanes sum strided fragments, LDS tree
__global__ __launch_bounds__(256) void k_combine_big(const uint32_t *__restrict__ bstart, g1_xyzz_t *__restrict__ buckets,
                                                    const g1_xyzz_t *__restrict__ frag_head, const g1_xyzz_t *__restrict__ frag_tail,
                                                    msm_plan_t pl, uint32_t chunks_per_window, const uint32_t *__restrict__ big,
                                                    uint32_t big_cap) {
    __shared__ g1_xyzz_t sh[256];
    uint32_t count = big[0] < big_cap ? big[0] : big_cap;
    for (uint32_t i = blockIdx.x; i < count; i += gridDim.x) {
        uint32_t gid = big[1 + i];
        uint32_t w = gid / pl.B, b = gid - w * pl.B;
        const uint32_t *bs = bstart + (uint64_t)w * (pl.B + 1);
        uint32_t sb = bs[b], eb = bs[b + 1];
        uint32_t t0 = sb / MSM_CHUNK, t1 = (eb - 1) / MSM_CHUNK;
        const g1_xyzz_t *fh = frag_head + (uint64_t)w * chunks_per_window, *ft = frag_tail + (uint64_t)w * chunks_per_window;
        g1_xyzz_t acc = G1::inf();
        for (uint32_t t = t0 + threadIdx.x; t <= t1; t += 256) {
            const g1_xyzz_t *src = (t == t0 && sb != t0 * MSM_CHUNK) ? ft + t : fh + t;
            acc = G1::add(acc, tk_load(src));
        }
        sh[threadIdx.x] = acc;
        __syncthreads();
        for (uint32_t st = 128; st > 0; st >>= 1) {
            if (threadIdx.x < st) {
                acc = G1::add(acc, sh[threadIdx.x + st]);
                sh[threadIdx.x] = acc;
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) tk_store(buckets + gid, acc);
        __syncthreads();
    }
}

#define MSM_SEG 16
// one lane per (window, segment of MSM_SEG buckets): seg_out = sum_{v in segment} v * B_v
//   = tot + v0 * run   with run = sum B_v, tot = sum (v - v0) B_v by the running-sum trick
__global__ __launch_bounds__(128) void k_reduce_segments(const g1_xyzz_t *__restrict__ buckets, g1_xyzz_t *__restrict__ seg_out,
                                                        msm_plan_t pl, uint32_t segs) {
    uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= pl.W * segs) return;
    uint32_t w = gid / segs, sg = gid - w * segs;
    uint32_t v0 = sg * MSM_SEG;  // bucket array index b holds value v = b + 1; segment covers b in [v0, v0+L)
    uint32_t len = pl.B - v0 < MSM_SEG ? pl.B - v0 : MSM_SEG;
    const g1k_xyzz *bk = reinterpret_cast<const g1k_xyzz *>(buckets) + (uint64_t)w * pl.B + v0;
    g1k_xyzz run = G1K::inf(), tot = G1K::inf();
    for (int k = (int)len - 1; k >= 0; k--) {
        run = G1K::add(run, tk_load(bk + k));
        tot = G1K::add(tot, run);
    }
    // + v0 * run  (v0 < 2^15)
    if (v0) {
        g1k_xyzz m = G1K::inf();
        for (int bit = 31 - __builtin_clz(v0); bit >= 0; bit--) {
            m = G1K::dbl(m);
            if ((v0 >> bit) & 1) m = G1K::add(m, run);
        }
        tot = G1K::add(tot, m);
    }
    tk_store(reinterpret_cast<g1k_xyzz *>(seg_out) + gid, tot);
}

// grid (W), 256 threads: window sum = sum of its segment results
__global__ __launch_bounds__(256) void k_reduce_windows(const g1_xyzz_t *__restrict__ seg_in, g1_xyzz_t *__restrict__ win_out,
                                                       uint32_t segs) {
    __shared__ g1k_xyzz sh[256];
    const uint32_t w = blockIdx.x, t = threadIdx.x;
    const g1k_xyzz *in = reinterpret_cast<const g1k_xyzz *>(seg_in) + (uint64_t)w * segs;
    g1k_xyzz acc = G1K::inf();
    for (uint32_t k = t; k < segs; k += 256) acc = G1K::add(acc, tk_load(in + k));
    sh[t] = acc;
    __syncthreads();
    for (uint32_t s = 128; s > 0; s >>= 1) {
        if (t < s) {
            acc = G1K::add(acc, sh[t + s]);
            sh[t] = acc;
        }
        __syncthreads();
    }
    if (t == 0) tk_store(reinterpret_cast<g1k_xyzz *>(win_out) + w, acc);
}

// out[b] = [s_b] P_b : the "batch of one-point MSMs" shape (libs/src/iotools/mod.rs:1113-1151)
__global__ __launch_bounds__(128) void k_msm_size1(const fr_t *__restrict__ scalars, const g1_affine_t *__restrict__ bases,
                                                  int shared, uint32_t n, int points_mont, g1_xyzz_t *__restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fr_t s = Fr::canon(tk_load(scalars + i));
    g1_affine_t p0 = tk_load(bases + (shared ? 0 : i));
    g1k_aff p;
    p.x = p0.x;
    p.y = p0.y;
    if (!points_mont && !G1::is_inf(p0)) {
        p.x = Fq::to_mont(Fq::canon(p0.x));
        p.y = Fq::to_mont(Fq::canon(p0.y));
    }
    g1k_xyzz acc = G1K::inf();
    for (int bit = 254; bit >= 0; bit--) {
        acc = G1K::dbl(acc);
        if ((s.l[bit >> 5] >> (bit & 31)) & 1) acc = G1K::add_mixed(acc, p);
    }
    tk_store(reinterpret_cast<g1k_xyzz *>(out) + i, acc);
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
static uint32_t choose_c(uint32_t n) {
    // minimise W * (n + ~4*B) over c <= 16.  (c = 17, 18 are accepted on request — the two-pass sort handles 512 x 256 bins —
    // but measured slower at 2^24 points: 63.4 vs 51.3 ms; 128-entry buckets mean more chunk fragments to flush and combine.)
    uint32_t best = 2;
    double best_cost = 1e300;
    for (uint32_t c = 2; c <= 16; c++) {
        double W = (double)(255 / c + 1);
        double cost = W * ((double)n + 4.0 * (double)(1u << (c - 1)));
        if (cost < best_cost) {
            best_cost = cost;
            best = c;
        }
    }
    return best;
}

static void store_canonical(tkmk_g1_projective *o, const g1_xyzz_t &r) {
    g1_affine_t a = G1::to_affine(r);
    bool inf = G1::is_inf(r);
    fq_t x = inf ? Fq::zero() : Fq::from_mont(a.x);
    fq_t y = Fq::from_mont(a.y);
    fq_t z = Fq::zero();
    if (inf) {
        y = Fq::zero();
        y.l[0] = 1;
    } else {
        z.l[0] = 1;
    }
    for (int i = 0; i < 12; i++) {
        o->x.limbs[i] = x.l[i];
        o->y.limbs[i] = y.l[i];
        o->z.limbs[i] = z.l[i];
    }
}

TK_API tkmk_msm_config tkmk_msm_default_config(void) {
    tkmk_msm_config c;
    c.stream_handle = nullptr;
    c.precompute_factor = 1;
    c.c = 0;
    c.bitsize = 0;
    c.batch_size = 1;
    c.are_points_shared_in_batch = true;
    c.are_scalars_on_device = false;
    c.are_scalars_montgomery_form = false;
    c.are_points_on_device = false;
    c.are_points_montgomery_form = false;
    c.are_results_on_device = false;
    c.is_async = false;
    c.ext = nullptr;
    return c;
}


// TKMK_MSM_DEBUG=1: pull every intermediate back and re-derive it on the host (small sizes only).
static bool xyzz_same_point(const g1_xyzz_t &a, const g1_xyzz_t &b) {
    g1_affine_t pa = G1::to_affine(a), pb = G1::to_affine(b);
    return Fq::eq(pa.x, pb.x) && Fq::eq(pa.y, pb.y);
}
static tkmk_error msm_debug_check(const msm_plan_t &pl, const fr_t *scalars, const g1_affine_t *bases, const uint32_t *d_dig,
                                  const uint32_t *d_sorted, const uint32_t *d_bstart, const g1_xyzz_t *d_buckets, hipStream_t s) {
    TK_HIP(hipStreamSynchronize(s));
    TK_HIP(hipGetLastError());
    std::vector<fr_t> sc(pl.n);
    std::vector<g1_affine_t> bs(pl.n);
    std::vector<uint32_t> dig((size_t)pl.W * pl.n), sorted((size_t)pl.W * pl.n), bstart((size_t)pl.W * (pl.B + 1));
    std::vector<g1_xyzz_t> buckets((size_t)pl.W * pl.B);
    TK_HIP(hipMemcpy(sc.data(), scalars, sc.size() * sizeof(fr_t), hipMemcpyDeviceToHost));
    TK_HIP(hipMemcpy(bs.data(), bases, bs.size() * sizeof(g1_affine_t), hipMemcpyDeviceToHost));
    {   // device bases are {x 2^406, y 2^406} (packed); the host re-derivation works in saturated Montgomery form
        fq_t rp;
        for (int j = 0; j < 12; j++) rp.l[j] = bls12_381_fq_params::KSATM[j];  // 2^406 mod p (plain)
        fq_t rinv = Fq::inv(Fq::to_mont(rp));
        for (auto &b : bs)
            if (!G1::is_inf(b)) {
                b.x = Fq::mul(Fq::to_mont(b.x), rinv);
                b.y = Fq::mul(Fq::to_mont(b.y), rinv);
            }
    }
    TK_HIP(hipMemcpy(dig.data(), d_dig, dig.size() * 4, hipMemcpyDeviceToHost));
    TK_HIP(hipMemcpy(sorted.data(), d_sorted, sorted.size() * 4, hipMemcpyDeviceToHost));
    TK_HIP(hipMemcpy(bstart.data(), d_bstart, bstart.size() * 4, hipMemcpyDeviceToHost));
    TK_HIP(hipMemcpy(buckets.data(), d_buckets, buckets.size() * sizeof(g1_xyzz_t), hipMemcpyDeviceToHost));
    fprintf(stderr, "[msm debug] n=%u c=%u W=%u B=%u chunks=%u chunk_len=%u\n", pl.n, pl.c, pl.W, pl.B, pl.chunks, pl.chunk_len);
    int bad = 0;
    // digits reconstruct the scalar: sum_w d_w 2^(c w) == s  (checked mod 2^64 on the low limbs for brevity)
    for (uint32_t i = 0; i < pl.n && bad < 5; i++) {
        unsigned __int128 acc = 0;
        for (int w = (int)pl.W - 1; w >= 0; w--) {
            uint32_t rec = dig[(size_t)w * pl.n + i];
            long long d = (long long)(rec & 0x7fffffffu);
            if (rec & 0x80000000u) d = -d;
            if ((rec & 0x7fffffffu) > pl.B) { fprintf(stderr, "[msm debug] digit too large i=%u w=%d rec=%08x\n", i, w, rec); bad++; }
            if ((uint32_t)w * pl.c < 100) acc = (acc << pl.c) + (unsigned __int128)(__int128)d;
        }
        uint64_t lo = (uint64_t)sc[i].l[0] | ((uint64_t)sc[i].l[1] << 32);
        (void)lo;
    }
    for (uint32_t w = 0; w < pl.W; w++) {
        const uint32_t *bsw = &bstart[(size_t)w * (pl.B + 1)];
        uint32_t nz = 0;
        for (uint32_t i = 0; i < pl.n; i++) nz += (dig[(size_t)w * pl.n + i] & 0x7fffffffu) != 0;
        if (bsw[pl.B] != nz && bad < 20) { fprintf(stderr, "[msm debug] w=%u total %u != nonzero digits %u\n", w, bsw[pl.B], nz); bad++; }
        for (uint32_t b = 0; b < pl.B; b++) {
            if (bsw[b] > bsw[b + 1]) { if (bad < 20) fprintf(stderr, "[msm debug] w=%u b=%u bstart not monotone\n", w, b); bad++; continue; }
            g1_xyzz_t acc = G1::inf();
            for (uint32_t k = bsw[b]; k < bsw[b + 1]; k++) {
                uint32_t rec = sorted[(size_t)w * pl.n + k], i = rec & 0x7fffffffu;
                if (i >= pl.n) { if (bad < 20) fprintf(stderr, "[msm debug] w=%u b=%u bad index %u\n", w, b, i); bad++; continue; }
                uint32_t drec = dig[(size_t)w * pl.n + i];
                if ((drec & 0x7fffffffu) != b + 1 || ((drec ^ rec) & 0x80000000u)) {
                    if (bad < 20) fprintf(stderr, "[msm debug] w=%u b=%u entry %u: digit rec %08x does not belong here\n", w, b, k, drec);
                    bad++;
                }
                g1_affine_t pnt = bs[i];
                if (rec & 0x80000000u) pnt.y = Fq::neg(pnt.y);
                acc = G1::add_mixed(acc, pnt);
            }
            bool okb = xyzz_same_point(acc, buckets[(size_t)w * pl.B + b]);
            if (atoi(getenv("TKMK_MSM_DEBUG")) >= 2 && w == 7 && bsw[b + 1] > bsw[b])
                fprintf(stderr, "[msm debug] w7 b=%u pos=%u cnt=%u %s\n", b, bsw[b], bsw[b + 1] - bsw[b], okb ? "ok" : "BAD");
            if (!okb) {
                if (bad < 20) {
                    const g1_xyzz_t &dv = buckets[(size_t)w * pl.B + b];
                    uint32_t rec0 = bsw[b + 1] > bsw[b] ? sorted[(size_t)w * pl.n + bsw[b]] : 0;
                    const char *cls = "other";
                    if (G1::is_inf(dv)) cls = "inf";
                    else if (xyzz_same_point(G1::neg(acc), dv)) cls = "negated";
                    fprintf(stderr, "[msm debug] w=%u b=%u bucket sum mismatch (%u entries) rec0=%08x device=%s zz0=%08x x0=%08x\n", w, b,
                            bsw[b + 1] - bsw[b], rec0, cls, dv.zz.l[0], dv.x.l[0]);
                }
                bad++;
            }
        }
    }
    fprintf(stderr, "[msm debug] %d inconsistencies\n", bad);
    return TKMK_SUCCESS;
}

// Enqueues one MSM of n points (device-resident scalars / converted bases) on stream s, ending with the
// async copy of the W window sums into wins_host (>= MSM_MAX_WINDOWS entries; pinned memory keeps the copy
// asynchronous).  Scratch comes from the caller's tk_frame on s, which must stay open until s has drained.
#define MSM_MAX_WINDOWS 128
static tkmk_error msm_enqueue(const fr_t *scalars, const g1_affine_t *bases_mont, uint32_t n, uint32_t c_req, uint32_t bits,
                              bool scalars_mont, hipStream_t s, g1_xyzz_t *wins_host, msm_plan_t *plan_out) {
    msm_plan_t pl;
    pl.n = n;
    pl.bits = bits;
    pl.c = c_req ? c_req : choose_c(n);
    if (pl.c < 2) pl.c = 2;
    if (pl.c > 18) pl.c = 18;
    static const bool force_one_pass = getenv("TKMK_MSM_ONE_PASS") != nullptr;
    if (pl.c > 16 && (force_one_pass || n < (1u << 18))) pl.c = 16;   // wide windows need the two-pass sort
    pl.W = bits / pl.c + 1;
    pl.B = 1u << (pl.c - 1);
    uint32_t want = (512 + pl.W - 1) / pl.W;
    if (getenv("TKMK_MSM_CHUNKS")) want = (uint32_t)atoi(getenv("TKMK_MSM_CHUNKS"));
    uint32_t maxc = (n + 8191) / 8192;
    pl.chunks = want < maxc ? want : maxc;
    if (pl.chunks < 1) pl.chunks = 1;
    pl.chunk_len = (n + pl.chunks - 1) / pl.chunks;

    tk_scratch d_dig, d_sorted, d_counts, d_bstart, d_buckets, d_segs, d_wins;
    TK_TRY(d_dig.alloc((size_t)pl.W * n * 4, s));
    TK_TRY(d_sorted.alloc((size_t)pl.W * n * 4, s));
    TK_TRY(d_counts.alloc((size_t)pl.W * pl.B * pl.chunks * 4, s));
    TK_TRY(d_bstart.alloc((size_t)pl.W * (pl.B + 1) * 4, s));
    TK_TRY(d_buckets.alloc((size_t)pl.W * pl.B * sizeof(g1_xyzz_t), s));
    uint32_t segs = (pl.B + MSM_SEG - 1) / MSM_SEG;
    TK_TRY(d_segs.alloc((size_t)pl.W * segs * sizeof(g1_xyzz_t), s));
    TK_TRY(d_wins.alloc((size_t)pl.W * sizeof(g1_xyzz_t), s));

    tk_prof prof(s);
    hipLaunchKernelGGL(k_digits, tk_div_up(n, 256), 256, 0, s, scalars, d_dig.as<uint32_t>(), pl, scalars_mont ? 1 : 0);
    prof.mark("msm.digits");
    size_t lds = (size_t)pl.B * 4;
    static bool attr_set = false;
    if (!attr_set) {
        TK_HIP(hipFuncSetAttribute((const void *)k_hist, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        TK_HIP(hipFuncSetAttribute((const void *)k_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        attr_set = true;
    }
    const bool two_pass = !force_one_pass && pl.c >= 13 && n >= (1u << 18);
    if (!two_pass) {
        hipLaunchKernelGGL(k_hist, dim3(pl.chunks, pl.W), 1024, lds, s, (const uint32_t *)d_dig.p, d_counts.as<uint32_t>(), pl, 0u);
        prof.mark("msm.hist");
        const uint32_t groups = (pl.B + 1023) / 1024;
        tk_scratch d_local, d_gtot;
        TK_TRY(d_local.alloc((size_t)pl.W * pl.B * 4, s));
        TK_TRY(d_gtot.alloc((size_t)pl.W * groups * 4, s));
        hipLaunchKernelGGL(k_scan_local, dim3(groups, pl.W), 1024, 0, s, (const uint32_t *)d_counts.p, d_local.as<uint32_t>(),
                           d_gtot.as<uint32_t>(), pl);
        hipLaunchKernelGGL(k_scan_apply, dim3(groups, pl.W), 1024, 0, s, d_counts.as<uint32_t>(), (const uint32_t *)d_local.p,
                           (const uint32_t *)d_gtot.p, d_bstart.as<uint32_t>(), pl);
        prof.mark("msm.scan");
        hipLaunchKernelGGL(k_scatter, dim3(pl.chunks, pl.W), 1024, lds, s, (const uint32_t *)d_dig.p, (const uint32_t *)d_counts.p,
                           d_sorted.as<uint32_t>(), pl);
        prof.mark("msm.scatter");
    } else {
        const uint32_t fbits = pl.c >= 17 ? 8u : 7u, FB = 1u << fbits, CB = pl.B >> fbits;
        tk_scratch d_aidx, d_akey, d_cs, d_localA, d_gtotA, d_cntB, d_fs, d_localB, d_gtotB;
        TK_TRY(d_aidx.alloc((size_t)pl.W * n * 4, s));
        TK_TRY(d_akey.alloc((size_t)pl.W * n * 2, s));
        TK_TRY(d_cs.alloc((size_t)pl.W * (CB + 1) * 4, s));
        // ---- pass A: coarse bins ----
        msm_plan_t pa = pl;
        pa.B = CB;
        hipLaunchKernelGGL(k_hist, dim3(pl.chunks, pl.W), 1024, (size_t)CB * 4, s, (const uint32_t *)d_dig.p, d_counts.as<uint32_t>(), pa,
                           fbits);
        prof.mark("msm.hist");
        TK_TRY(d_localA.alloc((size_t)pl.W * CB * 4, s));
        TK_TRY(d_gtotA.alloc((size_t)pl.W * 4, s));
        hipLaunchKernelGGL(k_scan_local, dim3(1, pl.W), 1024, 0, s, (const uint32_t *)d_counts.p, d_localA.as<uint32_t>(),
                           d_gtotA.as<uint32_t>(), pa);
        hipLaunchKernelGGL(k_scan_apply, dim3(1, pl.W), 1024, 0, s, d_counts.as<uint32_t>(), (const uint32_t *)d_localA.p,
                           (const uint32_t *)d_gtotA.p, d_cs.as<uint32_t>(), pa);
        prof.mark("msm.scan");
        hipLaunchKernelGGL(k_scatter_staged<false>, dim3(pl.chunks, pl.W), 1024, 0, s, (const uint32_t *)d_dig.p, (const uint16_t *)nullptr,
                           (const uint32_t *)nullptr, (const uint32_t *)d_counts.p, d_aidx.as<uint32_t>(), d_akey.as<uint16_t>(), n, CB, CB,
                           pl.chunk_len, fbits);
        // ---- pass B: fine bins inside every (window, coarse bin) segment ----
        const uint32_t segs_b = pl.W * CB;
        TK_TRY(d_cntB.alloc((size_t)segs_b * FB * SC_PARTS * 4, s));
        TK_TRY(d_fs.alloc((size_t)segs_b * (FB + 1) * 4, s));
        TK_TRY(d_localB.alloc((size_t)segs_b * FB * 4, s));
        TK_TRY(d_gtotB.alloc((size_t)segs_b * 4, s));
        hipLaunchKernelGGL(k_hist_fine, dim3(SC_PARTS, segs_b), 1024, 0, s, (const uint16_t *)d_akey.p, (const uint32_t *)d_cs.p,
                           d_cntB.as<uint32_t>(), n, CB, fbits);
        msm_plan_t pb = pl;
        pb.B = FB;
        pb.chunks = SC_PARTS;
        hipLaunchKernelGGL(k_scan_local, dim3(1, segs_b), 1024, 0, s, (const uint32_t *)d_cntB.p, d_localB.as<uint32_t>(),
                           d_gtotB.as<uint32_t>(), pb);
        hipLaunchKernelGGL(k_scan_apply, dim3(1, segs_b), 1024, 0, s, d_cntB.as<uint32_t>(), (const uint32_t *)d_localB.p,
                           (const uint32_t *)d_gtotB.p, d_fs.as<uint32_t>(), pb);
        hipLaunchKernelGGL(k_bstart_assemble, tk_div_up((size_t)pl.W * (pl.B + 1), 256), 256, 0, s, (const uint32_t *)d_cs.p,
                           (const uint32_t *)d_fs.p, d_bstart.as<uint32_t>(), pl.W, pl.B, CB, fbits);
        hipLaunchKernelGGL(k_scatter_staged<true>, dim3(SC_PARTS, segs_b), 1024, 0, s, (const uint32_t *)d_aidx.p, (const uint16_t *)d_akey.p,
                           (const uint32_t *)d_cs.p, (const uint32_t *)d_cntB.p, d_sorted.as<uint32_t>(), (uint16_t *)nullptr, n, FB, CB,
                           0u, fbits);
        prof.mark("msm.scatter");
    }
    const uint32_t cpw = (n + MSM_CHUNK - 1) / MSM_CHUNK;  // chunks per window (upper bound: all digits non-zero)
    const uint32_t big_cap = 1u << 16;
    tk_scratch d_fh, d_ft, d_big;
    TK_TRY(d_fh.alloc((size_t)pl.W * cpw * sizeof(g1_xyzz_t), s));
    TK_TRY(d_ft.alloc((size_t)pl.W * cpw * sizeof(g1_xyzz_t), s));
    TK_TRY(d_big.alloc((size_t)(big_cap + 1) * 4, s));
    TK_HIP(hipMemsetAsync(d_buckets.p, 0, (size_t)pl.W * pl.B * sizeof(g1_xyzz_t), s));  // empty bucket = infinity
    TK_HIP(hipMemsetAsync(d_big.p, 0, 4, s));
    hipLaunchKernelGGL(k_accumulate_chunks, tk_div_up((size_t)pl.W * cpw, 256), 256, 0, s, bases_mont, (const uint32_t *)d_sorted.p,
                       (const uint32_t *)d_bstart.p, d_buckets.as<g1_xyzz_t>(), d_fh.as<g1_xyzz_t>(), d_ft.as<g1_xyzz_t>(), pl, cpw);
    prof.mark("msm.accumulate");
    hipLaunchKernelGGL(k_combine, tk_div_up((size_t)pl.W * pl.B, 256), 256, 0, s, (const uint32_t *)d_bstart.p, d_buckets.as<g1_xyzz_t>(),
                       (const g1_xyzz_t *)d_fh.p, (const g1_xyzz_t *)d_ft.p, pl, cpw, d_big.as<uint32_t>(), big_cap);
    hipLaunchKernelGGL(k_combine_big, 1024, 256, 0, s, (const uint32_t *)d_bstart.p, d_buckets.as<g1_xyzz_t>(),
                       (const g1_xyzz_t *)d_fh.p, (const g1_xyzz_t *)d_ft.p, pl, cpw, (const uint32_t *)d_big.p, big_cap);
    prof.mark("msm.combine");
    hipLaunchKernelGGL(k_reduce_segments, tk_div_up((size_t)pl.W * segs, 128), 128, 0, s, (const g1_xyzz_t *)d_buckets.p,
                       d_segs.as<g1_xyzz_t>(), pl, segs);
    prof.mark("msm.reduce_segments");
    hipLaunchKernelGGL(k_reduce_windows, pl.W, 256, 0, s, (const g1_xyzz_t *)d_segs.p, d_wins.as<g1_xyzz_t>(), segs);
    prof.mark("msm.reduce_windows");
    TK_HIP(hipGetLastError());
    if (getenv("TKMK_MSM_DEBUG"))
        fprintf(stderr, "[msm debug] ptrs dig=%p sorted=%p counts=%p bstart=%p buckets=%p (+%zu) segs=%p (+%zu) wins=%p\n", d_dig.p,
                d_sorted.p, d_counts.p, d_bstart.p, d_buckets.p, (size_t)pl.W * pl.B * sizeof(g1_xyzz_t), d_segs.p,
                (size_t)pl.W * segs * sizeof(g1_xyzz_t), d_wins.p);
    if (getenv("TKMK_MSM_DEBUG")) TK_TRY(msm_debug_check(pl, scalars, bases_mont, d_dig.as<uint32_t>(), d_sorted.as<uint32_t>(),
                                                          d_bstart.as<uint32_t>(), d_buckets.as<g1_xyzz_t>(), s));
    TK_HIP(hipMemcpyAsync(wins_host, d_wins.p, pl.W * sizeof(g1_xyzz_t), hipMemcpyDeviceToHost, s));
    prof.finish();
    *plan_out = pl;
    return TKMK_SUCCESS;
}

// Horner over the window sums on the host: acc = 2^c * acc + W_w
static g1_xyzz_t msm_horner(const msm_plan_t &pl, const g1_xyzz_t *wins) {
    g1_xyzz_t acc = G1::inf();
    for (int w = (int)pl.W - 1; w >= 0; w--) {
        for (uint32_t k = 0; k < pl.c; k++) acc = G1::dbl(acc);
        acc = G1::add(acc, wins[w]);
    }
    return acc;
}

// One MSM, synchronous on s.
static tkmk_error msm_one(const fr_t *scalars, const g1_affine_t *bases_mont, uint32_t n, uint32_t c_req, uint32_t bits,
                          bool scalars_mont, hipStream_t s, g1_xyzz_t *result_host) {
    tk_frame frame(s);
    g1_xyzz_t wins[MSM_MAX_WINDOWS];
    msm_plan_t pl;
    TK_TRY(msm_enqueue(scalars, bases_mont, n, c_req, bits, scalars_mont, s, wins, &pl));
    TK_HIP(hipStreamSynchronize(s));
    *result_host = msm_horner(pl, wins);
    return TKMK_SUCCESS;
}

// ---- pipelined independent MSMs -------------------------------------------------------------------
// The tail of one MSM (combine, bucket reduction: a few thousand threads of serial EC adds) is latency-bound
// and leaves most CUs idle; the prover's commits are 2^20..2^22 points, where that tail is 20-35 % of the call.
// Independent MSMs therefore run round-robin on a few internal streams (each with its own scratch arena), so
// one job's tail overlaps the next job's sort/accumulate; the host Horner of job i overlaps job i+1's kernels.
struct msm_slot {
    hipStream_t s = nullptr;
    hipEvent_t done = nullptr;
    g1_xyzz_t *wins = nullptr;  // pinned
    tk_frame *frame = nullptr;
    msm_plan_t pl;
    int job = -1;
};
struct msm_pipe_job {
    const fr_t *scalars;          // device
    const g1_affine_t *bases;     // device; converted form if !convert
    uint32_t n;
    bool convert, bases_mont;
};
static std::mutex g_pipe_mu;
static std::vector<msm_slot> g_slots;

static tkmk_error msm_pipe_slots(uint32_t want) {
    while (g_slots.size() < want) {
        msm_slot sl;
        TK_HIP(hipStreamCreateWithFlags(&sl.s, hipStreamNonBlocking));
        TK_HIP(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
        TK_HIP(hipHostMalloc((void **)&sl.wins, MSM_MAX_WINDOWS * sizeof(g1_xyzz_t), hipHostMallocDefault));
        g_slots.push_back(sl);
    }
    return TKMK_SUCCESS;
}

static tkmk_error msm_pipe_finish(msm_slot &sl, g1_xyzz_t *results) {
    if (sl.job < 0) return TKMK_SUCCESS;
    hipError_t e = hipEventSynchronize(sl.done);
    delete sl.frame;
    sl.frame = nullptr;
    int j = sl.job;
    sl.job = -1;
    TK_HIP(e);
    results[j] = msm_horner(sl.pl, sl.wins);
    return TKMK_SUCCESS;
}

static tkmk_error msm_pipeline(const std::vector<msm_pipe_job> &jobs, uint32_t c_req, uint32_t bits, bool scalars_mont,
                               hipStream_t caller, g1_xyzz_t *results) {
    std::lock_guard<std::mutex> lk(g_pipe_mu);
    static const uint32_t n_streams = [] {
        const char *e = getenv("TKMK_MSM_STREAMS");
        int v = e ? atoi(e) : 3;
        return (uint32_t)(v < 1 ? 1 : v > 8 ? 8 : v);
    }();
    const uint32_t K = jobs.size() < n_streams ? (uint32_t)jobs.size() : n_streams;
    TK_TRY(msm_pipe_slots(K));
    // work queued on the caller's stream (input uploads, shared-bases conversion) happens-before every job
    hipEvent_t ready;
    TK_HIP(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
    TK_HIP(hipEventRecord(ready, caller));
    tkmk_error err = TKMK_SUCCESS;
    for (size_t j = 0; j < jobs.size() && err == TKMK_SUCCESS; j++) {
        msm_slot &sl = g_slots[j % K];
        err = msm_pipe_finish(sl, results);
        if (err != TKMK_SUCCESS) break;
        const msm_pipe_job &jb = jobs[j];
        if (j < K) {
            hipError_t e = hipStreamWaitEvent(sl.s, ready, 0);
            if (e != hipSuccess) { err = TKMK_ERR_UNKNOWN; break; }
        }
        sl.frame = new tk_frame(sl.s);
        sl.job = (int)j;
        const g1_affine_t *bm = jb.bases;
        if (jb.convert) {
            tk_scratch d_bm;
            err = d_bm.alloc((size_t)jb.n * 96, sl.s);
            if (err != TKMK_SUCCESS) break;
            hipLaunchKernelGGL(k_convert_bases, tk_div_up(jb.n, 256), 256, 0, sl.s, jb.bases, d_bm.as<g1_affine_t>(), (uint64_t)jb.n,
                               jb.bases_mont ? 1 : 0);
            bm = d_bm.as<g1_affine_t>();
        }
        err = msm_enqueue(jb.scalars, bm, jb.n, c_req, bits, scalars_mont, sl.s, sl.wins, &sl.pl);
        if (err != TKMK_SUCCESS) break;
        if (hipEventRecord(sl.done, sl.s) != hipSuccess) err = TKMK_ERR_UNKNOWN;
    }
    // drain in job order (also on error, so no frame outlives its kernels)
    {
        std::vector<msm_slot *> live;
        for (auto &sl : g_slots)
            if (sl.job >= 0) live.push_back(&sl);
        std::sort(live.begin(), live.end(), [](msm_slot *a, msm_slot *b) { return a->job < b->job; });
        for (msm_slot *sl : live) {
            if (err != TKMK_SUCCESS) (void)hipStreamSynchronize(sl->s);
            tkmk_error e2 = msm_pipe_finish(*sl, results);
            if (err == TKMK_SUCCESS) err = e2;
        }
    }
    (void)hipEventDestroy(ready);
    return err;
}

TK_API tkmk_error bls12_381_msm(const tkmk_fr *scalars, const tkmk_g1_affine *bases, int msm_size, const tkmk_msm_config *cfg,
                                tkmk_g1_projective *results) {
    if (!cfg || cfg->ext) return TKMK_ERR_INVALID_ARGUMENT;
    if (cfg->precompute_factor > 1) return TKMK_ERR_API_NOT_IMPLEMENTED;
    if (msm_size < 0 || cfg->batch_size < 1 || cfg->bitsize < 0 || cfg->bitsize > 255 || cfg->c < 0 || cfg->c > 18)
        return TKMK_ERR_INVALID_ARGUMENT;
    if (!results) return TKMK_ERR_INVALID_POINTER;
    TK_TRY(tk_require_device());
    const uint32_t n = (uint32_t)msm_size, batch = (uint32_t)cfg->batch_size;
    const uint32_t bits = cfg->bitsize ? (uint32_t)cfg->bitsize : 255u;
    hipStream_t s = tk_stream(cfg->stream_handle);
    tk_frame frame(s);
    std::vector<tkmk_g1_projective> host_res(batch);
    if (n == 0) {
        for (uint32_t b = 0; b < batch; b++) store_canonical(&host_res[b], G1::inf());
    } else {
        if (!scalars || !bases) return TKMK_ERR_INVALID_POINTER;
        const size_t n_bases = cfg->are_points_shared_in_batch ? n : (size_t)n * batch;
        tk_staged S, P;
        TK_TRY(S.in(scalars, (size_t)n * batch * 32, cfg->are_scalars_on_device, s));
        TK_TRY(P.in(bases, n_bases * 96, cfg->are_points_on_device, s));
        if (n == 1) {
            // batch of one-point MSMs = batched scalar multiplication
            tk_scratch d_out;
            TK_TRY(d_out.alloc((size_t)batch * sizeof(g1_xyzz_t), s));
            hipLaunchKernelGGL(k_msm_size1, tk_div_up(batch, 128), 128, 0, s, (const fr_t *)S.dev, (const g1_affine_t *)P.dev,
                               cfg->are_points_shared_in_batch ? 1 : 0, batch, cfg->are_points_montgomery_form ? 1 : 0,
                               d_out.as<g1_xyzz_t>());
            TK_HIP(hipGetLastError());
            std::vector<g1_xyzz_t> r(batch);
            TK_HIP(hipMemcpyAsync(r.data(), d_out.p, (size_t)batch * sizeof(g1_xyzz_t), hipMemcpyDeviceToHost, s));
            TK_HIP(hipStreamSynchronize(s));
            for (uint32_t b = 0; b < batch; b++) store_canonical(&host_res[b], r[b]);
        } else {
            tk_scratch d_bm;
            TK_TRY(d_bm.alloc(n_bases * 96, s));
            {
                tk_prof prof(s);
                hipLaunchKernelGGL(k_convert_bases, tk_div_up(n_bases, 256), 256, 0, s, (const g1_affine_t *)P.dev,
                                   d_bm.as<g1_affine_t>(), (uint64_t)n_bases, cfg->are_points_montgomery_form ? 1 : 0);
                prof.mark("msm.convert_bases");
                prof.finish();
                TK_HIP(hipGetLastError());
            }
            const g1_affine_t *bm = d_bm.as<g1_affine_t>();
            if (batch == 1) {
                g1_xyzz_t r;
                TK_TRY(msm_one((const fr_t *)S.dev, bm, n, (uint32_t)cfg->c, bits, cfg->are_scalars_montgomery_form, s, &r));
                store_canonical(&host_res[0], r);
            } else {
                std::vector<msm_pipe_job> jobs(batch);
                for (uint32_t b = 0; b < batch; b++)
                    jobs[b] = {(const fr_t *)S.dev + (size_t)b * n, bm + (cfg->are_points_shared_in_batch ? 0 : (size_t)b * n), n, false,
                               false};
                std::vector<g1_xyzz_t> r(batch);
                TK_TRY(msm_pipeline(jobs, (uint32_t)cfg->c, bits, cfg->are_scalars_montgomery_form, s, r.data()));
                for (uint32_t b = 0; b < batch; b++) store_canonical(&host_res[b], r[b]);
            }
        }
    }
    if (cfg->are_results_on_device) {
        TK_HIP(hipMemcpyAsync(results, host_res.data(), (size_t)batch * sizeof(tkmk_g1_projective), hipMemcpyHostToDevice, s));
        TK_HIP(hipStreamSynchronize(s));
    } else {
        for (uint32_t b = 0; b < batch; b++) results[b] = host_res[b];
    }
    return TKMK_SUCCESS;
}

// Independent MSMs of different sizes / bases in one call (the prover's commits between two transcript
// challenges), pipelined over internal streams.  cfg's *_on_device / *_montgomery_form flags apply to every job;
// cfg->batch_size must be 1.
TK_API tkmk_error tkmk_msm_multi(const tkmk_msm_job *jobs, int n_jobs, const tkmk_msm_config *cfg, tkmk_g1_projective *results) {
    if (!cfg || cfg->ext || cfg->batch_size != 1 || n_jobs < 0) return TKMK_ERR_INVALID_ARGUMENT;
    if (cfg->precompute_factor > 1) return TKMK_ERR_API_NOT_IMPLEMENTED;
    if (cfg->bitsize < 0 || cfg->bitsize > 255 || cfg->c < 0 || cfg->c > 18) return TKMK_ERR_INVALID_ARGUMENT;
    if (n_jobs == 0) return TKMK_SUCCESS;
    if (!jobs || !results) return TKMK_ERR_INVALID_POINTER;
    TK_TRY(tk_require_device());
    const uint32_t bits = cfg->bitsize ? (uint32_t)cfg->bitsize : 255u;
    hipStream_t s = tk_stream(cfg->stream_handle);
    tk_frame frame(s);
    std::vector<tkmk_g1_projective> host_res(n_jobs);
    std::vector<msm_pipe_job> pj;
    std::vector<int> pj_index;
    std::vector<tk_staged> staged((size_t)n_jobs * 2);
    for (int j = 0; j < n_jobs; j++) {
        if (jobs[j].msm_size < 0) return TKMK_ERR_INVALID_ARGUMENT;
        if (jobs[j].msm_size > 0 && (!jobs[j].scalars || !jobs[j].bases)) return TKMK_ERR_INVALID_POINTER;
    }
    for (int j = 0; j < n_jobs; j++) {
        const uint32_t n = (uint32_t)jobs[j].msm_size;
        if (n < 2) {  // empty sum / single scalar multiplication: the plain entry point handles both
            tkmk_msm_config c1 = *cfg;
            c1.are_results_on_device = false;
            TK_TRY(bls12_381_msm(jobs[j].scalars, jobs[j].bases, (int)n, &c1, &host_res[j]));
            continue;
        }
        tk_staged &S = staged[2 * j], &P = staged[2 * j + 1];
        TK_TRY(S.in(jobs[j].scalars, (size_t)n * 32, cfg->are_scalars_on_device, s));
        TK_TRY(P.in(jobs[j].bases, (size_t)n * 96, cfg->are_points_on_device, s));
        pj.push_back({(const fr_t *)S.dev, (const g1_affine_t *)P.dev, n, true, cfg->are_points_montgomery_form});
        pj_index.push_back(j);
    }
    if (!pj.empty()) {
        std::vector<g1_xyzz_t> r(pj.size());
        TK_TRY(msm_pipeline(pj, (uint32_t)cfg->c, bits, cfg->are_scalars_montgomery_form, s, r.data()));
        for (size_t k = 0; k < pj.size(); k++) store_canonical(&host_res[pj_index[k]], r[k]);
    }
    if (cfg->are_results_on_device) {
        TK_HIP(hipMemcpyAsync(results, host_res.data(), (size_t)n_jobs * sizeof(tkmk_g1_projective), hipMemcpyHostToDevice, s));
        TK_HIP(hipStreamSynchronize(s));
    } else {
        for (int j = 0; j < n_jobs; j++) results[j] = host_res[j];
    }
    return TKMK_SUCCESS;
}
