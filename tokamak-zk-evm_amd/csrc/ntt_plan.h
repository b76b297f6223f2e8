// ntt_plan.h — index arithmetic of the multi-pass Stockham NTT, shared by the HIP kernel (ntt.hip)
// and the host-side emulation used by the not-gpu tests (tests/hostcheck).
//
// Transform: `batch` vectors of length n = 2^logn in ICICLE's two layouts
// (packages/backend/libs/src/bivariate_polynomial/mod.rs:1467-1476):
//   rows    (columns_batch = false): element p of vector b at  b*n + p
//   columns (columns_batch = true ): element p of vector b at  p*batch + b
// Natural order in and out.  The transform is split into passes of radix R_k = 2^logR_k (Stockham
// autosort, out of place).  With Ns = R_0 * ... * R_{k-1} and J = n / R, pass k computes for every
// j in [0, J):
//     v[r]   = in[j + r*J] * w_{Ns*R}^{(j mod Ns) * r}            r in [0, R)
//     V      = DFT_R(v)                                            natural order
//     out[(j / Ns)*Ns*R + (j mod Ns) + r*Ns] = V[r]
// A workgroup owns a TILE of T = TILE_ELEMS / R "lines" (a line = one (vector, j) pair) staged in LDS:
//   rows   : line L = tile*T + l  ->  vector b = L / J, j = L mod J     (lines are consecutive j)
//   columns: tile -> (j, group of T consecutive vectors b)               (lines are consecutive b)
// so that consecutive lines are consecutive in memory: every global access is a run of >= T*32 bytes,
// or, when J == 1 / Ns == 1 in the row layout, a fully contiguous block of the tile.
#pragma once
#include <stdint.h>

#ifndef FF_HD
#define FF_HD __host__ __device__ __forceinline__
#endif

struct ntt_pass_t {
    uint32_t logn;     // vector length
    uint32_t logR;     // radix of this pass
    uint32_t logNs;    // product of the radices of the previous passes
    uint32_t logT;     // lines per tile
    uint32_t columns;  // 1 = columns layout
    uint32_t inverse;
    uint32_t first;    // apply `pre` on load
    uint32_t last;     // apply `post` / post_const on store
    uint64_t batch;
    uint32_t logN;     // twiddle domain
    uint32_t groups;   // columns layout: ceil(batch / T)
    uint64_t tiles;    // grid size
    // zero-padded input (first pass of a job only; 0 = the input is a full vector): positions >= in_len read as zero
    // without touching memory, and with in_compact the vectors are stored 2^in_logn apart instead of 2^logn (row layout)
    uint64_t in_len;
    uint32_t in_logn;
    uint32_t in_compact;
};

struct ntt_line_t {
    uint64_t b, j;
    bool valid;
};

FF_HD uint32_t ntt_bitrev(uint32_t x, uint32_t bits) { return bits ? (__builtin_bitreverse32(x) >> (32 - bits)) : 0; }

FF_HD ntt_line_t ntt_line(const ntt_pass_t &p, uint64_t tile, uint32_t l) {
    ntt_line_t r;
    uint32_t logJ = p.logn - p.logR;
    if (p.columns) {
        r.j = tile / p.groups;
        r.b = (tile % p.groups) * ((uint64_t)1 << p.logT) + l;
    } else {
        uint64_t L = (tile << p.logT) + l;
        r.b = L >> logJ;
        r.j = L & (((uint64_t)1 << logJ) - 1);
    }
    r.valid = r.b < p.batch;
    return r;
}
// position (within the vector) of input r / output r of line (b, j)
FF_HD uint64_t ntt_pos_in(const ntt_pass_t &p, uint64_t j, uint32_t r) { return j + ((uint64_t)r << (p.logn - p.logR)); }
FF_HD uint64_t ntt_pos_out(const ntt_pass_t &p, uint64_t j, uint32_t r) {
    uint64_t lo = j & (((uint64_t)1 << p.logNs) - 1);
    uint64_t hi = j >> p.logNs;
    return (hi << (p.logNs + p.logR)) + lo + ((uint64_t)r << p.logNs);
}
FF_HD uint64_t ntt_addr(const ntt_pass_t &p, uint64_t b, uint64_t pos) {
    return p.columns ? pos * p.batch + b : (b << p.logn) + pos;
}
// index into tw[] (tw[i] = w_N^i, i < N) of w_{2^logm}^e for the pass direction; e < 2^logm
FF_HD uint64_t ntt_tw_index(const ntt_pass_t &p, uint32_t logm, uint64_t e) {
    uint64_t idx = e << (p.logN - logm);
    if (p.inverse && idx) idx = ((uint64_t)1 << p.logN) - idx;
    return idx;
}
// exponent of the Stockham twiddle applied to input r of line j: (j mod Ns) * r  (< Ns*R)
FF_HD uint64_t ntt_stockham_exp(const ntt_pass_t &p, uint64_t j, uint32_t r) {
    return (j & (((uint64_t)1 << p.logNs) - 1)) * r;
}
// iteration order of the tile's (l, r) pairs for global loads / stores: r fastest when the tile is a
// contiguous block in r (row layout with J == 1 on load, Ns == 1 on store), else l fastest.
FF_HD bool ntt_in_rfast(const ntt_pass_t &p) { return !p.columns && p.logn == p.logR; }
FF_HD bool ntt_out_rfast(const ntt_pass_t &p) { return !p.columns && p.logNs == 0; }
// LDS slot of element (l, r): XOR swizzle keeps both iteration orders (nearly) bank-conflict free
FF_HD uint32_t ntt_slot(uint32_t logT, uint32_t l, uint32_t r) { return (r << logT) + (l ^ (r & ((1u << logT) - 1))); }

// Split logn into passes of at most max_logR bits, as evenly as possible; returns the pass count.
static inline int ntt_split(uint32_t logn, uint32_t max_logR, uint32_t *logR_out) {
    if (logn == 0) {
        logR_out[0] = 0;
        return 1;
    }
    int passes = (int)((logn + max_logR - 1) / max_logR);
    uint32_t base = logn / passes, extra = logn % passes;
    for (int k = 0; k < passes; k++) logR_out[k] = base + ((uint32_t)k < extra ? 1 : 0);
    return passes;
}

// Parameters of pass k (0-based) of a `passes`-pass transform with radices logR[].
static inline ntt_pass_t ntt_make_pass(uint32_t logn, uint64_t batch, bool columns, bool inverse, uint32_t logN,
                                       const uint32_t *logR, int passes, int k, uint32_t log_tile) {
    ntt_pass_t p;
    p.logn = logn;
    p.logR = logR[k];
    p.logNs = 0;
    for (int i = 0; i < k; i++) p.logNs += logR[i];
    p.logT = log_tile - logR[k];
    p.columns = columns ? 1 : 0;
    p.inverse = inverse ? 1 : 0;
    p.first = k == 0;
    p.last = k == passes - 1;
    p.batch = batch;
    p.logN = logN;
    p.in_len = 0;
    p.in_logn = logn;
    p.in_compact = 0;
    uint64_t T = 1ull << p.logT, J = 1ull << (logn - logR[k]);
    if (columns) {
        p.groups = (uint32_t)((batch + T - 1) / T);
        p.tiles = J * p.groups;
    } else {
        p.groups = 1;
        p.tiles = (batch * J + T - 1) / T;
    }
    return p;
}
