// g1ntt.hip — the bivariate NTT over G1 POINTS ("in the exponent") behind tkmk_g1_ntt (include/tkmk.h):
//     out[i][j] = sum_{a < x_size, b < y_size} w_x^(+-i a) w_y^(+-j b) in[a][b]
// No reference counterpart: the reference commits every polynomial in the coefficient basis (encode_poly, libs/src/iotools/mod.rs:2041-2113).
// Applied once per circuit to the sub-grid [tau_x^a tau_y^b] G of the CRS (inverse direction) it yields the LAGRANGE-basis points
//     Lambda_ij = sum_ab w_x^(-i a) w_y^(-j b) [tau_x^a tau_y^b] G = [N L_i(tau_x) L_j(tau_y)] G,
// so that the commitment of a polynomial given by its evaluations e_ij on the roots of unity is (1/N) sum_ij e_ij Lambda_ij — the same
// point as encode_poly of its coefficients.  The prover's u, v, w and b ARE evaluations (read_R1CS_gen_uvwXY, gen_bXY:
// libs/src/iotools/mod.rs:1287-1420, libs/src/polynomial_structures/mod.rs:132-162), mostly zeros and small numbers, which an MSM skips or
// passes in one window; their inverse transforms are dense 255-bit scalars.
//
// Stockham decimation-in-frequency stages over XYZZ points, one lane per butterfly:  (a, b) -> (a + b, [w](a - b)).  The scalar
// multiplication is double-and-add over the bits of the twiddle; lanes are ordered with the batch index fastest, so a wave shares its
// twiddle and only executes the additions of set bits.  A one-time cost per circuit (seconds); the per-proof kernels are elsewhere.
#include <vector>

#include "common.h"

namespace {

struct g1ntt_stage_t {
    uint32_t m, s;                     // current transform length, stride (Stockham: m halves, s doubles)
    uint32_t seq_stride, batch_stride; // element (k, c) of sequence c at k * seq_stride + c * batch_stride
    uint32_t batch;                    // sequences
    uint32_t tw_step;                  // twiddle of butterfly row p: table[p * tw_step]
};

// converted / plain / Montgomery affine records -> XYZZ in saturated Montgomery form (the strided sub-grid view is resolved here)
__global__ __launch_bounds__(256) void k_g1ntt_load(const g1_affine_t *__restrict__ in, uint32_t in_stride, uint32_t y_size, uint64_t n, int form,
                                                   fq_t conv, g1_xyzz_t *__restrict__ out) {
    uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const uint32_t a = (uint32_t)(e / y_size), b = (uint32_t)(e - (uint64_t)a * y_size);
    g1_affine_t p = tk_load(in + (uint64_t)a * in_stride + b);
    if (!G1::is_inf(p)) {
        p.x = Fq::canon(p.x), p.y = Fq::canon(p.y);
        if (form == TKMK_BASES_PLAIN) p.x = Fq::to_mont(p.x), p.y = Fq::to_mont(p.y);
        else if (form == TKMK_BASES_CONVERTED) p.x = Fq::mul(p.x, conv), p.y = Fq::mul(p.y, conv);   // x R' * (R^2 / R') / R = x R
    }
    tk_store(out + e, G1::from_affine(p));
}

__global__ __launch_bounds__(64) void k_g1ntt_stage(const g1_xyzz_t *__restrict__ x, g1_xyzz_t *__restrict__ y, g1ntt_stage_t st, const fr_t *__restrict__ tw,
                                                   uint64_t lanes) {
    uint64_t L = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (L >= lanes) return;
    const uint32_t c = (uint32_t)(L % st.batch), t = (uint32_t)(L / st.batch);   // batch index fastest: the wave shares p
    const uint32_t h = st.m >> 1, p = t / st.s, q = t - p * st.s;
    const uint64_t base = (uint64_t)c * st.batch_stride;
    const g1_xyzz_t a = tk_load(x + base + (uint64_t)(q + st.s * p) * st.seq_stride);
    const g1_xyzz_t b = tk_load(x + base + (uint64_t)(q + st.s * (p + h)) * st.seq_stride);
    tk_store(y + base + (uint64_t)(q + st.s * (2 * p)) * st.seq_stride, G1::add(a, b));
    g1_xyzz_t d = G1::add(a, G1::neg(b));
    if (p) {   // [w] d, w = table[p * tw_step] (plain integer below r), most significant set bit first
        const fr_t w = tk_load(tw + (uint64_t)p * st.tw_step);
        int top = 254;
        while (top > 0 && !((w.l[top >> 5] >> (top & 31)) & 1u)) top--;
        g1_xyzz_t acc = d;
        for (int bit = top - 1; bit >= 0; bit--) {
            acc = G1::dbl(acc);
            if ((w.l[bit >> 5] >> (bit & 31)) & 1u) acc = G1::add(acc, d);
        }
        d = acc;
    }
    tk_store(y + base + (uint64_t)(q + st.s * (2 * p + 1)) * st.seq_stride, d);
}

// XYZZ -> plain affine records (one inversion per point)
__global__ __launch_bounds__(128) void k_g1ntt_store(const g1_xyzz_t *__restrict__ in, uint64_t n, g1_affine_t *__restrict__ out) {
    uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    g1_xyzz_t p = tk_load(in + e);
    g1_affine_t a;
    if (G1::is_inf(p)) {
        a.x = Fq::zero(), a.y = Fq::zero();
    } else {
        a = G1::to_affine(p);
        a.x = Fq::from_mont(a.x), a.y = Fq::from_mont(a.y);
    }
    tk_store(out + e, a);
}

// out[i] = [w] in[i]: plain affine records in and out, one scalar for every point (the wave runs the same double-and-add)
__global__ __launch_bounds__(128) void k_g1_scale(const g1_affine_t *__restrict__ in, uint64_t n, fr_t w, int top, g1_affine_t *__restrict__ out) {
    uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    g1_affine_t p = tk_load(in + e);
    g1_affine_t a;
    a.x = Fq::zero(), a.y = Fq::zero();
    if (!G1::is_inf(p) && top >= 0) {
        p.x = Fq::to_mont(Fq::canon(p.x)), p.y = Fq::to_mont(Fq::canon(p.y));
        g1_xyzz_t acc = G1::from_affine(p);
        for (int bit = top - 1; bit >= 0; bit--) {
            acc = G1::dbl(acc);
            if ((w.l[bit >> 5] >> (bit & 31)) & 1u) acc = G1::add_mixed(acc, p);
        }
        if (!G1::is_inf(acc)) {
            a = G1::to_affine(acc);
            a.x = Fq::from_mont(a.x), a.y = Fq::from_mont(a.y);
        }
    }
    tk_store(out + e, a);
}

// ---- prefix sums of points: out[j] = sum_{j' <= j} in[idx(j')] ---------------------------------------------------------------
// three phases: runs of PS_RUN elements per lane, a one-workgroup scan of the run totals, the runs again with their offsets
#define PS_RUN 64
__device__ __forceinline__ uint64_t ps_index(uint64_t j, uint32_t rows, uint32_t cols, int transposed) {
    return transposed ? (j % rows) * cols + j / rows : j;
}
__device__ __forceinline__ g1_affine_t ps_load(const g1_affine_t *in, uint64_t at, int form, const fq_t &conv) {
    g1_affine_t p = tk_load(in + at);
    if (!G1::is_inf(p)) {
        p.x = Fq::canon(p.x), p.y = Fq::canon(p.y);
        if (form == TKMK_BASES_PLAIN) p.x = Fq::to_mont(p.x), p.y = Fq::to_mont(p.y);
        else if (form == TKMK_BASES_CONVERTED) p.x = Fq::mul(p.x, conv), p.y = Fq::mul(p.y, conv);
    }
    return p;
}
__global__ __launch_bounds__(64) void k_g1ps_runs(const g1_affine_t *__restrict__ in, uint64_t n, uint32_t rows, uint32_t cols, int transposed, int form,
                                                 fq_t conv, g1_xyzz_t *__restrict__ totals) {
    uint64_t L = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, lo = L * PS_RUN;
    if (lo >= n) return;
    uint64_t hi = lo + PS_RUN < n ? lo + PS_RUN : n;
    g1_xyzz_t acc = G1::inf();
    for (uint64_t j = lo; j < hi; j++) acc = G1::add_mixed(acc, ps_load(in, ps_index(j, rows, cols, transposed), form, conv));
    tk_store(totals + L, acc);
}
// exclusive scan of `count` run totals in place, one workgroup of PS_SCAN lanes (count / PS_SCAN totals per lane, an LDS scan of the
// lane sums in between: 96 KB of the CU's 160)
#define PS_SCAN 512
__global__ __launch_bounds__(PS_SCAN) void k_g1ps_scan(g1_xyzz_t *__restrict__ totals, uint64_t count) {
    extern __shared__ g1_xyzz_t ps_sh[];   // PS_SCAN records
    const uint32_t t = threadIdx.x;
    const uint64_t per = (count + PS_SCAN - 1) / PS_SCAN, lo = (uint64_t)t * per < count ? (uint64_t)t * per : count, hi = lo + per < count ? lo + per : count;
    g1_xyzz_t sum = G1::inf();
    for (uint64_t k = lo; k < hi; k++) sum = G1::add(sum, tk_load(totals + k));
    ps_sh[t] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < PS_SCAN; off <<= 1) {   // inclusive Hillis-Steele over the lane sums
        g1_xyzz_t v = G1::inf();
        if (t >= off) v = ps_sh[t - off];
        __syncthreads();
        if (t >= off) ps_sh[t] = G1::add(ps_sh[t], v);
        __syncthreads();
    }
    g1_xyzz_t run = t ? ps_sh[t - 1] : G1::inf();
    for (uint64_t k = lo; k < hi; k++) {
        g1_xyzz_t v = tk_load(totals + k);
        tk_store(totals + k, run);
        run = G1::add(run, v);
    }
}
__global__ __launch_bounds__(64) void k_g1ps_apply(const g1_affine_t *__restrict__ in, uint64_t n, uint32_t rows, uint32_t cols, int transposed, int form,
                                                  fq_t conv, const g1_xyzz_t *__restrict__ offsets, g1_xyzz_t *__restrict__ out) {
    uint64_t L = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, lo = L * PS_RUN;
    if (lo >= n) return;
    uint64_t hi = lo + PS_RUN < n ? lo + PS_RUN : n;
    g1_xyzz_t acc = tk_load(offsets + L);
    for (uint64_t j = lo; j < hi; j++) {
        acc = G1::add_mixed(acc, ps_load(in, ps_index(j, rows, cols, transposed), form, conv));
        tk_store(out + j, acc);
    }
}

fr_t fr_from_abi(const tkmk_fr &v) {
    fr_t r;
    for (int i = 0; i < 8; i++) r.l[i] = v.limbs[i];
    return r;
}

}  // namespace

TK_API tkmk_error tkmk_g1_ntt(const tkmk_g1_affine *in_dev, int bases_form, uint32_t in_stride, uint32_t x_size, uint32_t y_size, tkmk_ntt_dir dir,
                              tkmk_g1_affine *out_dev, tkmk_stream stream) {
    return tkmk_g1_ntt_axes(in_dev, bases_form, in_stride, x_size, y_size, dir, TKMK_G1_NTT_AXIS_X | TKMK_G1_NTT_AXIS_Y, out_dev, stream);
}

// the same with either pass switched off: AXIS_Y alone = x_size independent transforms of length y_size along the rows, AXIS_X alone =
// y_size transforms of length x_size along the columns.  A sharded context runs the X pass over its own columns of the grid and the Y
// pass over its own rows, with a change of layout in between (host/tkmk_service.hpp): 1 / G of the group transform per rank.
TK_API tkmk_error tkmk_g1_ntt_axes(const tkmk_g1_affine *in_dev, int bases_form, uint32_t in_stride, uint32_t x_size, uint32_t y_size, tkmk_ntt_dir dir,
                                   int axes, tkmk_g1_affine *out_dev, tkmk_stream stream) {
    if (!in_dev || !out_dev) return TKMK_ERR_INVALID_POINTER;
    if (axes & ~(TKMK_G1_NTT_AXIS_X | TKMK_G1_NTT_AXIS_Y)) return TKMK_ERR_INVALID_ARGUMENT;
    if (!x_size || !y_size || (x_size & (x_size - 1)) || (y_size & (y_size - 1)) || in_stride < y_size) return TKMK_ERR_INVALID_ARGUMENT;
    if (bases_form != TKMK_BASES_PLAIN && bases_form != TKMK_BASES_MONTGOMERY && bases_form != TKMK_BASES_CONVERTED) return TKMK_ERR_INVALID_ARGUMENT;
    if ((uint64_t)x_size * y_size >= (1ull << 31)) return TKMK_ERR_INVALID_ARGUMENT;
    TK_TRY(tk_require_device());
    hipStream_t s = tk_stream(stream);
    tk_frame frame(s);
    const uint64_t n = (uint64_t)x_size * y_size;
    // twiddle tables on the host: w^k or w^-k for k < size, plain integers (the declared root-of-unity convention)
    auto table = [&](uint32_t size, std::vector<fr_t> &out) -> tkmk_error {
        out.assign(size, Fr::zero());
        out[0].l[0] = 1;
        if (size == 1) return TKMK_SUCCESS;
        tkmk_fr w_abi;
        TK_TRY(bls12_381_get_root_of_unity(size, &w_abi));
        fr_t w = Fr::to_mont(fr_from_abi(w_abi));
        if (dir == TKMK_NTT_INVERSE) w = Fr::inv(w);
        fr_t run = Fr::one();
        for (uint32_t k = 1; k < size; k++) {
            run = Fr::mul(run, w);
            out[k] = Fr::from_mont(run);
        }
        return TKMK_SUCCESS;
    };
    std::vector<fr_t> twx, twy;
    TK_TRY(table(x_size, twx));
    TK_TRY(table(y_size, twy));
    tk_scratch d_twx, d_twy, d_a, d_b;
    TK_TRY(d_twx.alloc((size_t)x_size * sizeof(fr_t), s));
    TK_TRY(d_twy.alloc((size_t)y_size * sizeof(fr_t), s));
    TK_TRY(d_a.alloc(n * sizeof(g1_xyzz_t), s));
    TK_TRY(d_b.alloc(n * sizeof(g1_xyzz_t), s));
    TK_HIP(hipMemcpyAsync(d_twx.p, twx.data(), (size_t)x_size * sizeof(fr_t), hipMemcpyHostToDevice, s));
    TK_HIP(hipMemcpyAsync(d_twy.p, twy.data(), (size_t)y_size * sizeof(fr_t), hipMemcpyHostToDevice, s));
    // converted records hold x R' (R' = 2^(29 * 14)); the constant that takes them to the saturated Montgomery form is R^2 / R'
    fq_t rp;
    for (int j = 0; j < Fq::N; j++) rp.l[j] = bls12_381_fq_params::KSATM[j];   // R' mod p, a plain integer
    const fq_t conv = Fq::mul(Fq::inv(Fq::to_mont(rp)), Fq::r2());
    hipLaunchKernelGGL(k_g1ntt_load, tk_div_up(n, 256), 256, 0, s, (const g1_affine_t *)in_dev, in_stride, y_size, n, bases_form, conv, d_a.as<g1_xyzz_t>());
    g1_xyzz_t *cur = d_a.as<g1_xyzz_t>(), *nxt = d_b.as<g1_xyzz_t>();
    auto axis = [&](uint32_t len, uint32_t seq_stride, uint32_t batch_stride, uint32_t batch, const fr_t *tw) {
        for (uint32_t m = len, st = 1; m > 1; m >>= 1, st <<= 1) {
            g1ntt_stage_t g{m, st, seq_stride, batch_stride, batch, len / m};
            const uint64_t lanes = (uint64_t)batch * (len / 2);
            hipLaunchKernelGGL(k_g1ntt_stage, tk_div_up(lanes, 64), 64, 0, s, (const g1_xyzz_t *)cur, nxt, g, tw, lanes);
            std::swap(cur, nxt);
        }
    };
    if (axes & TKMK_G1_NTT_AXIS_Y) axis(y_size, 1, y_size, x_size, d_twy.as<fr_t>());   // along Y inside every row
    if (axes & TKMK_G1_NTT_AXIS_X) axis(x_size, y_size, 1, y_size, d_twx.as<fr_t>());   // along X inside every column
    hipLaunchKernelGGL(k_g1ntt_store, tk_div_up(n, 128), 128, 0, s, (const g1_xyzz_t *)cur, n, (g1_affine_t *)out_dev);
    TK_HIP(hipGetLastError());
    TK_HIP(hipStreamSynchronize(s));   // the host twiddle vectors and the frame's scratch end with this call
    return TKMK_SUCCESS;
}

// out[i] = [scalar] in[i] for n plain affine records on the device (in place allowed); scalar: host, plain integer below r.
// The prover's Lagrange-basis tables are the UNSCALED inverse transform of the monomial grid, N [L_i L_j] G; scaled once by 1 / N here, a
// commitment from evaluations is the MSM itself and no per-proof scalar multiplication by 1 / N remains (tkmk_service.hpp).
TK_API tkmk_error tkmk_g1_scale(const tkmk_g1_affine *in_dev, uint64_t n, const tkmk_fr *scalar, tkmk_g1_affine *out_dev, tkmk_stream stream) {
    if (!in_dev || !out_dev || !scalar) return TKMK_ERR_INVALID_POINTER;
    if (!n) return TKMK_ERR_INVALID_ARGUMENT;
    TK_TRY(tk_require_device());
    fr_t w;
    for (int i = 0; i < 8; i++) w.l[i] = scalar->limbs[i];
    w = Fr::canon(w);
    int top = 254;
    while (top >= 0 && !((w.l[top >> 5] >> (top & 31)) & 1u)) top--;   // -1: the zero scalar (every point becomes infinity)
    hipStream_t s = tk_stream(stream);
    hipLaunchKernelGGL(k_g1_scale, tk_div_up(n, 128), 128, 0, s, (const g1_affine_t *)in_dev, n, w, top, (g1_affine_t *)out_dev);
    TK_HIP(hipGetLastError());
    return TKMK_SUCCESS;
}

// out[j] = sum_{j' <= j} in[idx(j')], idx(j) = j, or (j % rows) * cols + j / rows with `transposed` (a rows x cols row-major table walked
// column by column: the order of prove1's running product, lib.rs:1858-1866).  in: device, rows * cols records of form TKMK_BASES_*;
// out: device, rows * cols plain affine records.
TK_API tkmk_error tkmk_g1_prefix_sums(const tkmk_g1_affine *in_dev, int bases_form, uint32_t rows, uint32_t cols, int transposed,
                                      tkmk_g1_affine *out_dev, tkmk_stream stream) {
    if (!in_dev || !out_dev) return TKMK_ERR_INVALID_POINTER;
    if (!rows || !cols || (uint64_t)rows * cols >= (1ull << 31)) return TKMK_ERR_INVALID_ARGUMENT;
    if (bases_form != TKMK_BASES_PLAIN && bases_form != TKMK_BASES_MONTGOMERY && bases_form != TKMK_BASES_CONVERTED) return TKMK_ERR_INVALID_ARGUMENT;
    if ((const void *)in_dev == (const void *)out_dev) return TKMK_ERR_INVALID_ARGUMENT;
    TK_TRY(tk_require_device());
    hipStream_t s = tk_stream(stream);
    tk_frame frame(s);
    const uint64_t n = (uint64_t)rows * cols, runs = (n + PS_RUN - 1) / PS_RUN;
    fq_t rp;
    for (int j = 0; j < Fq::N; j++) rp.l[j] = bls12_381_fq_params::KSATM[j];
    const fq_t conv = Fq::mul(Fq::inv(Fq::to_mont(rp)), Fq::r2());
    tk_scratch d_tot, d_out;
    TK_TRY(d_tot.alloc(runs * sizeof(g1_xyzz_t), s));
    TK_TRY(d_out.alloc(n * sizeof(g1_xyzz_t), s));
    static hipError_t attr = hipFuncSetAttribute((const void *)k_g1ps_scan, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(PS_SCAN * sizeof(g1_xyzz_t)));
    TK_HIP(attr);
    hipLaunchKernelGGL(k_g1ps_runs, tk_div_up(runs, 64), 64, 0, s, (const g1_affine_t *)in_dev, n, rows, cols, transposed, bases_form, conv, d_tot.as<g1_xyzz_t>());
    hipLaunchKernelGGL(k_g1ps_scan, 1, PS_SCAN, PS_SCAN * sizeof(g1_xyzz_t), s, d_tot.as<g1_xyzz_t>(), runs);
    hipLaunchKernelGGL(k_g1ps_apply, tk_div_up(runs, 64), 64, 0, s, (const g1_affine_t *)in_dev, n, rows, cols, transposed, bases_form, conv,
                       (const g1_xyzz_t *)d_tot.p, d_out.as<g1_xyzz_t>());
    hipLaunchKernelGGL(k_g1ntt_store, tk_div_up(n, 128), 128, 0, s, (const g1_xyzz_t *)d_out.p, n, (g1_affine_t *)out_dev);
    TK_HIP(hipGetLastError());
    TK_HIP(hipStreamSynchronize(s));
    return TKMK_SUCCESS;
}
