// ntt.hip — scalar-field (Fr) NTT / iNTT for gfx950 behind bls12_381_ntt*, tkmk_bintt (include/tkmk.h).
// Work-alike of icicle_core::ntt::{ntt, initialize_domain, release_domain, get_root_of_unity} with the
// semantics the reference relies on (packages/backend/libs/src/bivariate_polynomial/mod.rs:33-55,1422-1478;
// ordering / coset behaviour pinned by libs/src/tests.rs:107-180,1075-1087).
//
// Design (see ntt_plan.h for the index algebra): multi-pass Stockham, each pass a radix-2^logR
// (logR <= 9) transform of a 2048-element tile held in LDS as two 16-byte planes (XOR-swizzled), with
// the pass's R/2 butterfly twiddles staged in an LDS twiddle tile.  Global accesses are runs of
// >= T*32 B (T = 2048/R lines per tile) in both the row and the strided column layout, so the column
// pass of the reference's _biNTT needs no transpose.  Data stays in PLAIN form in HBM; twiddles are in
// Montgomery form, so mont_mul(plain, twiddle) is again plain: no conversion passes.  Coset scaling and
// the 1/n of the inverse are fused into the first load / last store.
//
// Arithmetic intensity is ~20 integer mul-adds per byte: the kernel is bound by v_mad_u64_u32 issue,
// not by HBM; no MFMA (integer modular arithmetic, not a dense contraction).
#include <stdio.h>

#include <map>
#include <mutex>

#include "common.h"
#include "ntt_plan.h"

#define NTT_LOG_TILE 11
#define NTT_TILE (1 << NTT_LOG_TILE)
#define NTT_THREADS 512
#define NTT_MAX_LOGR 9

struct ntt_domain_t {
    fr_t *tw = nullptr;  // tw[i] = w_N^i (Montgomery), i < N
    uint32_t logN = 0;
    fr_t root_plain;
    // inter-pass (Stockham) twiddle tables, built on first use and kept until the domain is released:
    // key (logNs, logR, inverse) -> table[r << logNs | jm] = w_{Ns R}^{+-jm r}.  Gathering these from tw[] costs more
    // than the pass's butterflies (each lane hits its own cache line: 7 ms of an 11 ms pass at 256 x 2^20); in this
    // layout consecutive lines read consecutive entries, like the data itself.
    std::map<uint32_t, fr_t *> stockham;
};
static ntt_domain_t g_dom;
static std::mutex g_dom_mu;

// tw[i] = prod over set bits b of i of pw[b],  pw[b] = w^(2^b) (Montgomery)
__global__ __launch_bounds__(256) void k_build_twiddles(fr_t *__restrict__ tw, const fr_t *__restrict__ pw, uint64_t n,
                                                        uint32_t logn) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fr_t acc = Fr::one();
    for (uint32_t b = 0; b < logn; b++)
        if ((i >> b) & 1) acc = Fr::mul(acc, tk_load(pw + b));
    tk_store(tw + i, acc);
}
// table[(r << logNs) + jm] = w_{Ns R}^{jm r} in the pass direction (copied out of tw[])
__global__ __launch_bounds__(256) void k_build_stockham(fr_t *__restrict__ out, const fr_t *__restrict__ tw, ntt_pass_t p) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >> (p.logNs + p.logR)) return;
    uint64_t r = i >> p.logNs, jm = i & (((uint64_t)1 << p.logNs) - 1);
    tk_store(out + i, tk_load(tw + ntt_tw_index(p, p.logNs + p.logR, jm * r)));
}
// out[i] = scale * g^i (Montgomery): the coset pre-/post-scale table of one NTT call
__global__ __launch_bounds__(256) void k_build_powers(fr_t *__restrict__ out, fr_t g, fr_t scale, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    tk_store(out + i, Fr::mul(Fr::pow_u64(g, i), scale));
}

struct lds_elem {
    static __device__ __forceinline__ void put(uint4 *lo, uint4 *hi, uint32_t slot, const fr_t &x) {
        const uint4 *s = reinterpret_cast<const uint4 *>(&x);
        lo[slot] = s[0];
        hi[slot] = s[1];
    }
    static __device__ __forceinline__ fr_t get(const uint4 *lo, const uint4 *hi, uint32_t slot) {
        fr_t x;
        uint4 *d = reinterpret_cast<uint4 *>(&x);
        d[0] = lo[slot];
        d[1] = hi[slot];
        return x;
    }
};

// One Stockham pass over one tile per workgroup.  post_mode: 0 none, 1 multiply by post_const, 2 by post[pos].
__global__ __launch_bounds__(NTT_THREADS) void k_ntt_pass(ntt_pass_t p, const fr_t *__restrict__ in, fr_t *__restrict__ out,
                                                         const fr_t *__restrict__ tw, const fr_t *__restrict__ stw,
                                                         const fr_t *__restrict__ pre, const fr_t *__restrict__ post, fr_t post_const,
                                                         int post_mode) {
    __shared__ uint4 lo[NTT_TILE], hi[NTT_TILE];
    __shared__ fr_t twl[1 << (NTT_MAX_LOGR - 1)];
    const uint32_t tid = threadIdx.x;
    const uint32_t logR = p.logR, logT = p.logT;
    const uint32_t R = 1u << logR, T = 1u << logT;
    const uint64_t tile = blockIdx.x;

    for (uint32_t e = tid; e < (R >> 1); e += NTT_THREADS) twl[e] = tk_load(tw + ntt_tw_index(p, logR, e));

    const bool in_rfast = ntt_in_rfast(p), out_rfast = ntt_out_rfast(p);
    for (uint32_t e = tid; e < NTT_TILE; e += NTT_THREADS) {
        uint32_t l = in_rfast ? e >> logR : e & (T - 1);
        uint32_t r = in_rfast ? e & (R - 1) : e >> logT;
        ntt_line_t ln = ntt_line(p, tile, l);
        fr_t x = Fr::zero();
        if (ln.valid) {
            uint64_t pos = ntt_pos_in(p, ln.j, r);
            x = tk_load(in + ntt_addr(p, ln.b, pos));
            if (p.first) {
                x = Fr::canon(x);
                if (pre) x = Fr::mul(x, tk_load(pre + pos));
            }
            if (p.logNs) {
                const fr_t *t = stw ? stw + ((uint64_t)r << p.logNs) + (ln.j & (((uint64_t)1 << p.logNs) - 1))
                                    : tw + ntt_tw_index(p, p.logNs + logR, ntt_stockham_exp(p, ln.j, r));
                x = Fr::mul(x, tk_load(t));
            }
        }
        lds_elem::put(lo, hi, ntt_slot(logT, l, ntt_bitrev(r, logR)), x);
    }
    __syncthreads();

    // DIT stages on bit-reversed rows, two at a time: a lane holds the 4 elements r0 + {0, h, 2h, 3h} in registers
    // and does both stages' 4 butterflies between one LDS read and one LDS write (half the LDS traffic, address
    // arithmetic and barriers of a stage-by-stage loop).  l is the fastest lane index so a 16-lane LDS group touches
    // consecutive 16-byte slots; the group index runs faster than the twiddle index while there are enough groups, so
    // a whole wave shares its twiddles and the w = 1 products (pos == 0: all of stage 0, 1/2 of stage 1, 1/4 of stage
    // 2, ...) are skipped through a wave-uniform branch: ~0.5 of the ~3.5 products per element per pass.
    uint32_t s = 0;
    if (logR & 1) {  // odd radix: stage 0 alone (h = 1: every twiddle is 1)
        for (uint32_t q = tid; q < (NTT_TILE >> 1); q += NTT_THREADS) {
            uint32_t l = q & (T - 1), r0 = (q >> logT) << 1;
            uint32_t s0 = ntt_slot(logT, l, r0), s1 = ntt_slot(logT, l, r0 + 1);
            fr_t a = lds_elem::get(lo, hi, s0), b = lds_elem::get(lo, hi, s1);
            lds_elem::put(lo, hi, s0, Fr::add(a, b));
            lds_elem::put(lo, hi, s1, Fr::sub(a, b));
        }
        __syncthreads();
        s = 1;
    }
    for (; s + 1 < logR; s += 2) {
        const uint32_t h = 1u << s;
        const uint32_t lg = logR - 1 - s;              // stage s: twiddle of pos is twl[pos << lg]
        const uint32_t lG = logR - 2 - s;              // log2(number of 4-element groups per line)
        const bool grp_fast = logT + lG >= 6;          // 64 consecutive lanes then share pos
        for (uint32_t q = tid; q < (NTT_TILE >> 2); q += NTT_THREADS) {
            uint32_t l = q & (T - 1), qq = q >> logT;
            uint32_t pos = grp_fast ? qq >> lG : qq & (h - 1);
            uint32_t grp = grp_fast ? qq & ((1u << lG) - 1) : qq >> s;
            uint32_t r0 = (grp << (s + 2)) + pos;
            uint32_t a0 = ntt_slot(logT, l, r0), a1 = ntt_slot(logT, l, r0 + h), a2 = ntt_slot(logT, l, r0 + 2 * h),
                     a3 = ntt_slot(logT, l, r0 + 3 * h);
            fr_t e0 = lds_elem::get(lo, hi, a0), e1 = lds_elem::get(lo, hi, a1), e2 = lds_elem::get(lo, hi, a2),
                 e3 = lds_elem::get(lo, hi, a3);
            const bool unit = pos == 0 && (grp_fast || h == 1);   // wave-uniform: t1 = t2 = 1
            if (!unit) {
                fr_t t1 = twl[pos << lg];
                e1 = Fr::mul(e1, t1);
                e3 = Fr::mul(e3, t1);
            }
            fr_t f0 = Fr::add(e0, e1), f1 = Fr::sub(e0, e1), f2 = Fr::add(e2, e3), f3 = Fr::sub(e2, e3);
            if (!unit) f2 = Fr::mul(f2, twl[pos << (lg - 1)]);
            f3 = Fr::mul(f3, twl[(pos + h) << (lg - 1)]);
            lds_elem::put(lo, hi, a0, Fr::add(f0, f2));
            lds_elem::put(lo, hi, a2, Fr::sub(f0, f2));
            lds_elem::put(lo, hi, a1, Fr::add(f1, f3));
            lds_elem::put(lo, hi, a3, Fr::sub(f1, f3));
        }
        __syncthreads();
    }

    for (uint32_t e = tid; e < NTT_TILE; e += NTT_THREADS) {
        uint32_t l = out_rfast ? e >> logR : e & (T - 1);
        uint32_t r = out_rfast ? e & (R - 1) : e >> logT;
        ntt_line_t ln = ntt_line(p, tile, l);
        if (!ln.valid) continue;
        fr_t x = lds_elem::get(lo, hi, ntt_slot(logT, l, r));
        uint64_t pos = ntt_pos_out(p, ln.j, r);
        if (p.last) {
            if (post_mode == 1) x = Fr::mul(x, post_const);
            if (post_mode == 2) x = Fr::mul(x, tk_load(post + pos));
        }
        tk_store(out + ntt_addr(p, ln.b, pos), x);
    }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
static fr_t fr_from_api(const tkmk_fr *x) {
    fr_t r;
    for (int i = 0; i < 8; i++) r.l[i] = x->limbs[i];
    return Fr::canon(r);
}
static void fr_to_api(tkmk_fr *o, const fr_t &x) {
    for (int i = 0; i < 8; i++) o->limbs[i] = x.l[i];
}
static fr_t fr_root_plain(uint32_t logn) {  // w_{2^logn}, Montgomery
    fr_t w;
    for (int i = 0; i < 8; i++) w.l[i] = bls12_381_fr_params::ROOT[i];
    w = Fr::to_mont(w);
    for (uint32_t i = logn; i < (uint32_t)bls12_381_fr_params::TWO_ADICITY; i++) w = Fr::sqr(w);
    return w;
}

TK_API tkmk_error bls12_381_get_root_of_unity(uint64_t max_size, tkmk_fr *rou_out) {
    if (!rou_out) return TKMK_ERR_INVALID_POINTER;
    uint32_t l = 0;
    while (l < 32 && (1ull << l) < max_size) l++;
    if ((1ull << l) < max_size) return TKMK_ERR_INVALID_ARGUMENT;
    fr_to_api(rou_out, Fr::from_mont(fr_root_plain(l)));
    return TKMK_SUCCESS;
}

TK_API tkmk_error bls12_381_ntt_init_domain(const tkmk_fr *primitive_root, const tkmk_ntt_init_domain_config *cfg) {
    if (!primitive_root) return TKMK_ERR_INVALID_POINTER;
    TK_TRY(tk_require_device());
    std::lock_guard<std::mutex> lk(g_dom_mu);
    if (g_dom.tw) return TKMK_ERR_INVALID_ARGUMENT;  // one global domain; release first (reference does: mod.rs:41-45)
    fr_t w = Fr::to_mont(fr_from_api(primitive_root));
    // order of the root must be a power of two <= 2^32
    fr_t pw[33];
    uint32_t logN = 0;
    pw[0] = w;
    while (logN < 32 && !Fr::eq(pw[logN], Fr::one())) {
        pw[logN + 1] = Fr::sqr(pw[logN]);
        logN++;
    }
    if (!Fr::eq(pw[logN], Fr::one())) return TKMK_ERR_INVALID_ARGUMENT;
    // primitive: w^(N/2) == -1 unless N == 1
    if (logN > 0 && !Fr::eq(pw[logN - 1], Fr::neg(Fr::one()))) return TKMK_ERR_INVALID_ARGUMENT;
    hipStream_t s = tk_stream(cfg ? cfg->stream_handle : nullptr);
    tk_frame frame(s);
    uint64_t N = 1ull << logN;
    fr_t *tw = nullptr;
    hipError_t e = hipMalloc((void **)&tw, N * sizeof(fr_t));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return e == hipErrorOutOfMemory ? TKMK_ERR_OUT_OF_MEMORY : TKMK_ERR_ALLOCATION_FAILED;
    }
    tk_scratch dpw;
    tkmk_error t = dpw.alloc(sizeof(fr_t) * 33, s);
    if (t != TKMK_SUCCESS) {
        (void)hipFree(tw);
        return t;
    }
    if (hipMemcpyAsync(dpw.p, pw, sizeof(fr_t) * 33, hipMemcpyHostToDevice, s) != hipSuccess) {
        (void)hipFree(tw);
        return TKMK_ERR_COPY_FAILED;
    }
    hipLaunchKernelGGL(k_build_twiddles, tk_div_up(N, 256), 256, 0, s, tw, (const fr_t *)dpw.p, N, logN);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
        (void)hipFree(tw);
        return TKMK_ERR_UNKNOWN;
    }
    g_dom.tw = tw;
    g_dom.logN = logN;
    g_dom.root_plain = fr_from_api(primitive_root);
    return TKMK_SUCCESS;
}

TK_API tkmk_error bls12_381_ntt_release_domain(void) {
    std::lock_guard<std::mutex> lk(g_dom_mu);
    if (g_dom.tw) {
        (void)hipDeviceSynchronize();
        (void)hipFree(g_dom.tw);
    }
    for (auto &kv : g_dom.stockham) (void)hipFree(kv.second);
    g_dom.stockham.clear();
    g_dom.tw = nullptr;
    g_dom.logN = 0;
    return TKMK_SUCCESS;
}

TK_API tkmk_ntt_config tkmk_ntt_default_config(void) {
    tkmk_ntt_config c;
    c.stream_handle = nullptr;
    for (int i = 0; i < 8; i++) c.coset_gen.limbs[i] = 0;
    c.coset_gen.limbs[0] = 1;
    c.batch_size = 1;
    c.columns_batch = false;
    c.ordering = TKMK_ORDER_NN;
    c.are_inputs_on_device = false;
    c.are_outputs_on_device = false;
    c.is_async = false;
    c.ext = nullptr;
    return c;
}

// One 1-D batched transform = a list of passes; a _biNTT = row passes followed by column passes.
struct ntt_job_t {
    uint32_t logn;
    uint64_t batch;
    bool columns;
    fr_t coset;  // plain; 1 = none
};

struct pass_launch_t {
    ntt_pass_t p;
    const fr_t *stw;  // Stockham table of this pass (nullptr: first pass, or too large -> gathered from tw[])
    const fr_t *pre;
    const fr_t *post;
    fr_t post_const;
    int post_mode;
};

// Stockham table of pass p from the domain's cache (g_dom_mu held by the caller); built once, synchronously.
#define NTT_STOCKHAM_MAX_LOG 25  // 1 GiB per table; beyond that the pass gathers from tw[]
static tkmk_error stockham_table(const ntt_pass_t &p, hipStream_t s, const fr_t **out) {
    *out = nullptr;
    if (p.logNs == 0 || p.logNs + p.logR > NTT_STOCKHAM_MAX_LOG) return TKMK_SUCCESS;
    const uint32_t key = (p.logNs << 8) | (p.logR << 1) | p.inverse;
    auto it = g_dom.stockham.find(key);
    if (it != g_dom.stockham.end()) {
        *out = it->second;
        return TKMK_SUCCESS;
    }
    const uint64_t entries = 1ull << (p.logNs + p.logR);
    fr_t *t = nullptr;
    if (hipMalloc((void **)&t, entries * sizeof(fr_t)) != hipSuccess) {
        (void)hipGetLastError();
        return TKMK_SUCCESS;  // no room for the table: fall back to the gather
    }
    hipLaunchKernelGGL(k_build_stockham, tk_div_up(entries, 256), 256, 0, s, t, (const fr_t *)g_dom.tw, p);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
        (void)hipFree(t);
        return TKMK_ERR_UNKNOWN;
    }
    g_dom.stockham[key] = t;
    *out = t;
    return TKMK_SUCCESS;
}

// Appends the passes of one job; allocates the coset tables it needs from `tables`.
static tkmk_error plan_job(const ntt_job_t &job, bool inverse, uint32_t logN, hipStream_t s, tk_scratch *tables, int &n_tables,
                           pass_launch_t *out, int &n_out) {
    uint32_t logR[8];
    int passes = ntt_split(job.logn, NTT_MAX_LOGR, logR);
    fr_t one_plain = Fr::zero();
    one_plain.l[0] = 1;
    bool has_coset = !Fr::eq(job.coset, one_plain);
    if (has_coset && Fr::is_zero(job.coset)) return TKMK_ERR_INVALID_ARGUMENT;
    uint64_t n = 1ull << job.logn;
    fr_t *table = nullptr;
    fr_t ninv;  // Montgomery form of 1/n
    {
        fr_t nm = Fr::zero();
        nm.l[0] = (uint32_t)n;
        nm.l[1] = (uint32_t)(n >> 32);
        ninv = Fr::inv(Fr::to_mont(nm));
    }
    if (has_coset) {
        TK_TRY(tables[n_tables].alloc(n * sizeof(fr_t), s));
        table = tables[n_tables].as<fr_t>();
        n_tables++;
        fr_t g = Fr::to_mont(job.coset);
        fr_t scale = Fr::one();
        if (inverse) {
            g = Fr::inv(g);
            scale = ninv;
        }
        hipLaunchKernelGGL(k_build_powers, tk_div_up(n, 256), 256, 0, s, table, g, scale, n);
        TK_HIP(hipGetLastError());
    }
    for (int k = 0; k < passes; k++) {
        pass_launch_t &L = out[n_out++];
        ntt_pass_t &p = L.p;
        p = ntt_make_pass(job.logn, job.batch, job.columns, inverse, logN, logR, passes, k, NTT_LOG_TILE);
        TK_TRY(stockham_table(p, s, &L.stw));
        L.pre = (!inverse && has_coset && p.first) ? table : nullptr;
        L.post = nullptr;
        L.post_mode = 0;
        L.post_const = ninv;
        if (inverse && p.last) {
            L.post_mode = has_coset ? 2 : 1;
            L.post = has_coset ? table : nullptr;
        }
    }
    return TKMK_SUCCESS;
}

// Runs the pass list in -> ... -> out, ping-ponging through scratch; in may alias out.
static tkmk_error run_passes(pass_launch_t *L, int n, const fr_t *in, fr_t *out, uint64_t total, const fr_t *tw, hipStream_t s) {
    // Destination of pass k: `out` when (n-1-k) is even, else scratch A, so the last pass lands in `out`.
    // When in aliases out and n is odd (>1) pass 0 would overwrite its own input: route it through B.
    // A single pass is tile-local (reads and writes the same positions) and may run in place.
    tk_scratch tmpA, tmpB;
    fr_t *A = nullptr, *B = nullptr;
    bool alias = (const void *)in == (const void *)out;
    if (n > 1) {
        TK_TRY(tmpA.alloc(total * sizeof(fr_t), s));
        A = tmpA.as<fr_t>();
        if (alias && (n & 1)) {
            TK_TRY(tmpB.alloc(total * sizeof(fr_t), s));
            B = tmpB.as<fr_t>();
        }
    }
    const fr_t *src = in;
    tk_prof prof(s);
    for (int k = 0; k < n; k++) {
        fr_t *dst = ((n - 1 - k) % 2 == 0) ? out : A;
        if (k == 0 && B) dst = B;
        if (L[k].p.tiles == 0 || L[k].p.tiles > 0x7fffffffull) return TKMK_ERR_INVALID_ARGUMENT;
        hipLaunchKernelGGL(k_ntt_pass, (unsigned)L[k].p.tiles, NTT_THREADS, 0, s, L[k].p, src, dst, tw, L[k].stw, L[k].pre, L[k].post,
                           L[k].post_const, L[k].post_mode);
        TK_HIP(hipGetLastError());
        char name[32];
        snprintf(name, sizeof name, "ntt.pass%d", k);
        prof.mark(name);
        src = dst;
    }
    prof.finish();
    return TKMK_SUCCESS;
}

static tkmk_error ntt_run(const tkmk_fr *input, tkmk_fr *output, const ntt_job_t *jobs, int n_jobs, bool inverse, bool in_dev,
                          bool out_dev, bool is_async, hipStream_t s) {
    TK_TRY(tk_require_device());
    tk_frame frame(s);
    std::lock_guard<std::mutex> lk(g_dom_mu);
    if (!g_dom.tw) return TKMK_ERR_INVALID_ARGUMENT;  // domain not initialised (reference panics: mod.rs:1434-1436)
    uint64_t total = (1ull << jobs[0].logn) * jobs[0].batch;
    for (int i = 0; i < n_jobs; i++)
        if (jobs[i].logn > g_dom.logN) return TKMK_ERR_INVALID_ARGUMENT;
    tk_staged I, O;
    TK_TRY(I.in(input, total * 32, in_dev, s));
    if (!in_dev && !out_dev) {
        O.dev = I.dev;  // transform in place in the staging buffer
    } else {
        TK_TRY(O.out(output, total * 32, out_dev, s));
    }
    pass_launch_t L[16];
    tk_scratch tables[2];
    int nL = 0, nT = 0;
    for (int i = 0; i < n_jobs; i++) TK_TRY(plan_job(jobs[i], inverse, g_dom.logN, s, tables, nT, L, nL));
    // with two jobs (biNTT) the aliasing rules are the same: treat the concatenated list as one chain
    TK_TRY(run_passes(L, nL, (const fr_t *)I.dev, (fr_t *)O.dev, total, g_dom.tw, s));
    if (!out_dev) TK_HIP(hipMemcpyAsync(output, O.dev, total * 32, hipMemcpyDeviceToHost, s));
    if (!is_async || !out_dev) TK_HIP(hipStreamSynchronize(s));
    return TKMK_SUCCESS;
}

TK_API tkmk_error bls12_381_ntt(const tkmk_fr *input, int size, tkmk_ntt_dir dir, const tkmk_ntt_config *cfg, tkmk_fr *output) {
    if (!cfg || cfg->ext || cfg->ordering != TKMK_ORDER_NN) return TKMK_ERR_INVALID_ARGUMENT;
    if (!input || !output) return TKMK_ERR_INVALID_POINTER;
    int logn = size > 0 ? tk_log2_exact((uint64_t)size) : -1;
    if (logn < 0 || cfg->batch_size < 1) return TKMK_ERR_INVALID_ARGUMENT;
    ntt_job_t job;
    job.logn = (uint32_t)logn;
    job.batch = (uint64_t)cfg->batch_size;
    job.columns = cfg->columns_batch && cfg->batch_size > 1;
    job.coset = fr_from_api(&cfg->coset_gen);
    return ntt_run(input, output, &job, 1, dir == TKMK_NTT_INVERSE, cfg->are_inputs_on_device, cfg->are_outputs_on_device,
                   cfg->is_async, tk_stream(cfg->stream_handle));
}

TK_API tkmk_error tkmk_bintt(const tkmk_fr *input, size_t x_size, size_t y_size, tkmk_ntt_dir dir, const tkmk_fr *coset_x,
                             const tkmk_fr *coset_y, bool on_device, tkmk_stream stream, tkmk_fr *output) {
    if (!input || !output) return TKMK_ERR_INVALID_POINTER;
    int lx = tk_log2_exact(x_size), ly = tk_log2_exact(y_size);
    if (lx < 0 || ly < 0) return TKMK_ERR_INVALID_ARGUMENT;
    fr_t one = Fr::zero();
    one.l[0] = 1;
    fr_t cx = coset_x ? fr_from_api(coset_x) : one, cy = coset_y ? fr_from_api(coset_y) : one;
    ntt_job_t jobs[2];
    int n = 0;
    // libs/src/bivariate_polynomial/mod.rs:1449-1476
    if (x_size == 1) {
        jobs[n++] = ntt_job_t{(uint32_t)ly, 1, false, cy};
    } else if (y_size == 1) {
        jobs[n++] = ntt_job_t{(uint32_t)lx, 1, false, cx};
    } else {
        jobs[n++] = ntt_job_t{(uint32_t)ly, (uint64_t)x_size, false, cy};  // rows of length y_size
        jobs[n++] = ntt_job_t{(uint32_t)lx, (uint64_t)y_size, true, cx};   // strided columns of length x_size
    }
    return ntt_run(input, output, jobs, n, dir == TKMK_NTT_INVERSE, on_device, on_device, false, tk_stream(stream));
}
