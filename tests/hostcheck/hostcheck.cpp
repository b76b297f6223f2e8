// hostcheck.cpp — TEST-ONLY: runs the product's __host__ __device__ field / curve templates
// (tokamak-zk-evm_amd/csrc/ff.h, ec.h) on the CPU so the not-gpu test tier can compare the exact
// code the kernels inline against the oracle.  Never linked into libtkmk_hip.so.
#include "hostcheck_common.h"

extern "C" {
void hc_fr_op(int op, const uint8_t *a, const uint8_t *b, uint8_t *o, size_t n) { binop<Fr>(op, a, b, o, n); }
void hc_fq_op(int op, const uint8_t *a, const uint8_t *b, uint8_t *o, size_t n) { binop<Fq>(op, a, b, o, n); }

static g1_affine_t load_aff(const uint8_t *p) {
    g1_affine_t a;
    load<Fq>(a.x, p);
    load<Fq>(a.y, p + 48);
    if (!G1::is_inf(a)) {
        a.x = Fq::to_mont(a.x);
        a.y = Fq::to_mont(a.y);
    }
    return a;
}
static void store_aff(uint8_t *p, const g1_affine_t &a) {
    store<Fq>(p, Fq::from_mont(a.x));
    store<Fq>(p + 48, Fq::from_mont(a.y));
}
// out = P (+) Q through the mixed add, the full add, and doubling paths
void hc_g1_add(int mode, const uint8_t *p, const uint8_t *q, uint8_t *o) {
    g1_affine_t P = load_aff(p), Q = load_aff(q);
    g1_xyzz_t r;
    if (mode == 0) r = G1::add_mixed(G1::from_affine(P), Q);
    else if (mode == 1) {
        // give both operands non-trivial ZZ/ZZZ: (P+Q) + (Q+Q) - ... keep simple: add(from(P), from(Q))
        r = G1::add(G1::from_affine(P), G1::from_affine(Q));
    } else if (mode == 2) {
        // (P + Q) + (P + Q) via full add with projective operands -> doubling branch with ZZ != 1
        g1_xyzz_t s = G1::add_mixed(G1::from_affine(P), Q);
        r = G1::add(s, s);
    } else {
        // ((P+Q) + Q) with a projective lhs
        g1_xyzz_t s = G1::add_mixed(G1::from_affine(P), Q);
        r = G1::add_mixed(s, Q);
    }
    store_aff(o, G1::to_affine(r));
}
void hc_g1_scalar_mul(const uint8_t *k, const uint8_t *p, uint8_t *o) {
    uint32_t kk[8];
    memcpy(kk, k, 32);
    store_aff(o, G1::to_affine(G1::scalar_mul(kk, 8, G1::from_affine(load_aff(p)))));
}
}

// ---------------------------------------------------------------------------------------------------
// Host emulation of k_ntt_pass (csrc/ntt.hip): same plan (ntt_plan.h), same per-tile data flow (load with
// pre-scale / Stockham twiddle into bit-reversed swizzled slots, radix-2 DIT stages, post-scaled store),
// executed sequentially.  max_logR / log_tile are parameters so small sizes exercise multi-pass plans.
// ---------------------------------------------------------------------------------------------------
#include <vector>

#include "ntt_plan.h"

extern "C" int hc_ntt(const uint8_t *in, uint32_t logn, uint64_t batch, int columns, int inverse, const uint8_t *coset,
                      uint8_t *out, uint32_t max_logR, uint32_t log_tile, uint32_t logN, const uint8_t *root32) {
    if (logN < logn || max_logR > log_tile) return -1;
    uint64_t N = 1ull << logN, n = 1ull << logn, total = n * batch;
    // domain
    fr_t w;
    load<Fr>(w, root32);   // w_{2^32}, plain: the caller's (declared) root of unity, as bls12_381_ntt_init_domain takes it
    w = Fr::to_mont(w);
    for (uint32_t i = logN; i < 32; i++) w = Fr::sqr(w);
    std::vector<fr_t> tw(N);
    tw[0] = Fr::one();
    for (uint64_t i = 1; i < N; i++) tw[i] = Fr::mul(tw[i - 1], w);
    // coset tables
    fr_t one_plain = Fr::zero();
    one_plain.l[0] = 1;
    fr_t g = one_plain;
    if (coset) load<Fr>(g, coset);
    bool has_coset = !Fr::eq(g, one_plain);
    fr_t nm = Fr::zero();
    nm.l[0] = (uint32_t)n;
    fr_t ninv = Fr::inv(Fr::to_mont(nm));
    std::vector<fr_t> table;
    if (has_coset) {
        table.resize(n);
        fr_t gm = Fr::to_mont(g), scale = Fr::one();
        if (inverse) {
            gm = Fr::inv(gm);
            scale = ninv;
        }
        for (uint64_t i = 0; i < n; i++) table[i] = Fr::mul(Fr::pow_u64(gm, i), scale);
    }
    uint32_t logR[16];
    int passes = ntt_split(logn, max_logR, logR);
    std::vector<fr_t> bufA(total), bufB(total);
    for (uint64_t i = 0; i < total; i++) load<Fr>(bufA[i], in + 32 * i);
    std::vector<fr_t> *src = &bufA, *dst = &bufB;
    uint32_t TILE = 1u << log_tile;
    std::vector<fr_t> lds(TILE);
    for (int k = 0; k < passes; k++) {
        ntt_pass_t p = ntt_make_pass(logn, batch, columns != 0, inverse != 0, logN, logR, passes, k, log_tile);
        uint32_t lR = p.logR, lT = p.logT, R = 1u << lR, T = 1u << lT;
        const fr_t *pre = (!inverse && has_coset && p.first) ? table.data() : nullptr;
        int post_mode = (inverse && p.last) ? (has_coset ? 2 : 1) : 0;
        bool in_rfast = ntt_in_rfast(p), out_rfast = ntt_out_rfast(p);
        for (uint64_t tile = 0; tile < p.tiles; tile++) {
            for (uint32_t e = 0; e < TILE; e++) {
                uint32_t l = in_rfast ? e >> lR : e & (T - 1);
                uint32_t r = in_rfast ? e & (R - 1) : e >> lT;
                ntt_line_t ln = ntt_line(p, tile, l);
                fr_t x = Fr::zero();
                if (ln.valid) {
                    uint64_t pos = ntt_pos_in(p, ln.j, r);
                    x = (*src)[ntt_addr(p, ln.b, pos)];
                    if (p.first) {
                        x = Fr::canon(x);
                        if (pre) x = Fr::mul(x, pre[pos]);
                    }
                    if (p.logNs) x = Fr::mul(x, tw[ntt_tw_index(p, p.logNs + lR, ntt_stockham_exp(p, ln.j, r))]);
                }
                lds[ntt_slot(lT, l, ntt_bitrev(r, lR))] = x;
            }
            for (uint32_t s = 0; s < lR; s++) {
                uint32_t h = 1u << s;
                for (uint32_t q = 0; q < (TILE >> 1); q++) {
                    uint32_t l = q & (T - 1), qq = q >> lT;
                    uint32_t pos = qq & (h - 1), grp = qq >> s;
                    uint32_t r0 = (grp << (s + 1)) + pos, r1 = r0 + h;
                    uint32_t s0 = ntt_slot(lT, l, r0), s1 = ntt_slot(lT, l, r1);
                    fr_t a = lds[s0], b = lds[s1];
                    if (s) b = Fr::mul(b, tw[ntt_tw_index(p, lR, pos << (lR - 1 - s))]);
                    lds[s0] = Fr::add(a, b);
                    lds[s1] = Fr::sub(a, b);
                }
            }
            for (uint32_t e = 0; e < TILE; e++) {
                uint32_t l = out_rfast ? e >> lR : e & (T - 1);
                uint32_t r = out_rfast ? e & (R - 1) : e >> lT;
                ntt_line_t ln = ntt_line(p, tile, l);
                if (!ln.valid) continue;
                fr_t x = lds[ntt_slot(lT, l, r)];
                uint64_t pos = ntt_pos_out(p, ln.j, r);
                if (p.last) {
                    if (post_mode == 1) x = Fr::mul(x, ninv);
                    if (post_mode == 2) x = Fr::mul(x, table[pos]);
                }
                (*dst)[ntt_addr(p, ln.b, pos)] = x;
            }
        }
        std::swap(src, dst);
    }
    for (uint64_t i = 0; i < total; i++) store<Fr>(out + 32 * i, (*src)[i]);
    return passes;
}

