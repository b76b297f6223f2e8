// poly.hip — device-resident bookkeeping of the bivariate coefficient matrix behind the tkmk_poly_* entries
// (include/tkmk.h).  The reference does all of this on the HOST after copying the whole matrix D->H
// ("resize operands" alone was 6.6 s of a 21 s GPU prove before their fix):
//   find_degree              packages/backend/libs/src/bivariate_polynomial/mod.rs:1480-1515
//   resize / mul_monomial    :1784-1806 / :1820-1844   (both are "place src at an offset into a zero matrix")
//   _scale_coeffs            :1567-1613
//   eval_x / eval_y / eval   :1719-1750
//   div_by_vanishing_opt     :2284-2410
//   div_by_ruffini           :2412-2477
// Matrix layout: element (ix, iy) at ix*y_size + iy, plain little-endian Fr.  All kernels are HBM-streaming
// except the two synthetic divisions, which are first-order linear recurrences along one axis.
#include "common.h"

// ---- find_degree: largest row / column index holding a non-zero coefficient (-1 if none) ----
__global__ __launch_bounds__(256) void k_find_degree(const fr_t *__restrict__ c, uint32_t xs, uint32_t ys, int *__restrict__ out) {
    __shared__ int sx[256], sy[256];
    int mx = -1, my = -1;
    uint64_t total = (uint64_t)xs * ys;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (uint64_t)gridDim.x * blockDim.x) {
        if (!Fr::is_zero(tk_load(c + e))) {
            int i = (int)(e / ys), j = (int)(e - (uint64_t)i * ys);
            mx = i > mx ? i : mx;
            my = j > my ? j : my;
        }
    }
    sx[threadIdx.x] = mx;
    sy[threadIdx.x] = my;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            sx[threadIdx.x] = max(sx[threadIdx.x], sx[threadIdx.x + s]);
            sy[threadIdx.x] = max(sy[threadIdx.x], sy[threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (sx[0] >= 0) atomicMax(out, sx[0]);
        if (sy[0] >= 0) atomicMax(out + 1, sy[0]);
    }
}

// dst (dx x dy) = zero matrix with src (sx x sy) placed at (ox, oy); source elements falling outside are dropped
__global__ __launch_bounds__(256) void k_place(const fr_t *__restrict__ src, uint32_t sx, uint32_t sy, fr_t *__restrict__ dst,
                                              uint32_t dx, uint32_t dy, uint32_t ox, uint32_t oy) {
    uint64_t total = (uint64_t)dx * dy;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t i = (uint32_t)(e / dy), j = (uint32_t)(e - (uint64_t)i * dy);
        fr_t v = Fr::zero();
        if (i >= ox && j >= oy && i - ox < sx && j - oy < sy) v = tk_load(src + (uint64_t)(i - ox) * sy + (j - oy));
        tk_store(dst + e, v);
    }
}

// ---- fused linear combination: out = sum_t c_t * shift(p_t, ox_t, oy_t), every operand read once, out written once ----
// The reference's poly_comb! (prove/src/lib.rs:30-38) and the operator chains around it (&a * &s, &a + &b, mul_monomial:
// libs/src/bivariate_polynomial/mod.rs:532-1281, 1820-1844) cost one full pass and one temporary of the OUTPUT size per operator;
// a 6-term combination is ~35 passes over 2^22..2^25 elements.  Here it is one pass.
#define LC_MAX_TERMS 24
struct lincomb_args_t {
    const fr_t *p[LC_MAX_TERMS];
    fr_t c[LC_MAX_TERMS];                     // Montgomery form (plain * Montgomery -> plain)
    uint32_t xs[LC_MAX_TERMS], ys[LC_MAX_TERMS], ox[LC_MAX_TERMS], oy[LC_MAX_TERMS];
    uint8_t kind[LC_MAX_TERMS];               // 0: c * p, 1: + p, 2: - p  (unit coefficients skip the product)
    uint32_t n;
};
// Two adjacent elements of a row per lane and step (dy even: 64 contiguous bytes per operand and lane in flight, the row / column of the
// pair computed once — by shift and mask when dy is a power of two, which every matrix of the polynomial layer is); a lane of an odd-width
// matrix takes one element.  HBM-streaming: FETCH x 2 + WRITE over the launch time is quoted in DESIGN.md section 5.
template <int V>
__global__ __launch_bounds__(256) void k_lincomb(lincomb_args_t a, fr_t *__restrict__ out, uint32_t dx, uint32_t dy, int dy_shift) {
    const uint64_t total = (uint64_t)dx * dy / V;   // V-element groups
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t e = g * V;
        uint32_t i, j;
        if (dy_shift >= 0) i = (uint32_t)(e >> dy_shift), j = (uint32_t)e & (dy - 1);
        else i = (uint32_t)(e / dy), j = (uint32_t)(e - (uint64_t)i * dy);
        fr_t acc[V];
#pragma unroll
        for (int v = 0; v < V; v++) acc[v] = Fr::zero();
        for (uint32_t t = 0; t < a.n; t++) {
            const uint32_t ii = i - a.ox[t];      // unsigned wrap makes "below the offset" fail the range test too
            if (ii >= a.xs[t]) continue;
            const fr_t *row = a.p[t] + (uint64_t)ii * a.ys[t];
            const uint32_t kind = a.kind[t];
#pragma unroll
            for (int v = 0; v < V; v++) {
                const uint32_t jj = j + v - a.oy[t];
                if (jj < a.ys[t]) {
                    fr_t x = Fr::canon(tk_load(row + jj));
                    if (kind == 1) acc[v] = Fr::add(acc[v], x);
                    else if (kind == 2) acc[v] = Fr::sub(acc[v], x);
                    else acc[v] = Fr::add(acc[v], Fr::mul(x, a.c[t]));
                }
            }
        }
#pragma unroll
        for (int v = 0; v < V; v++) tk_store(out + e + v, acc[v]);
    }
}

// pw[i] = g^i (Montgomery), i < n
__global__ __launch_bounds__(256) void k_powers(fr_t *__restrict__ out, fr_t g, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) tk_store(out + i, Fr::pow_u64(g, i));
}

// dst[i][j] = src[i][j] * px[i] * py[j]   (px / py may be null = all ones); x_minus_one: use (px[i] - 1) instead
__global__ __launch_bounds__(256) void k_scale(const fr_t *__restrict__ src, fr_t *__restrict__ dst, uint32_t xs, uint32_t ys,
                                              const fr_t *__restrict__ px, const fr_t *__restrict__ py, int x_minus_one) {
    uint64_t total = (uint64_t)xs * ys;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t i = (uint32_t)(e / ys), j = (uint32_t)(e - (uint64_t)i * ys);
        fr_t v = Fr::canon(tk_load(src + e));
        if (px) v = Fr::mul(v, x_minus_one ? Fr::sub(tk_load(px + i), Fr::one()) : tk_load(px + i));
        if (py) v = Fr::mul(v, tk_load(py + j));
        tk_store(dst + e, v);
    }
}

// out[i] = sum_j m[i][j] * w[j]   (one workgroup per row; w Montgomery, result plain)
__global__ __launch_bounds__(256) void k_row_dot(const fr_t *__restrict__ m, uint32_t ys, const fr_t *__restrict__ w,
                                                fr_t *__restrict__ out) {
    __shared__ fr_t sh[256];
    const fr_t *row = m + (uint64_t)blockIdx.x * ys;
    fr_t acc = Fr::zero();
    for (uint32_t j = threadIdx.x; j < ys; j += 256) acc = Fr::add(acc, Fr::mul(Fr::canon(tk_load(row + j)), tk_load(w + j)));
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            acc = Fr::add(acc, sh[threadIdx.x + s]);
            sh[threadIdx.x] = acc;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) tk_store(out + blockIdx.x, acc);
}
// part[chunk][j] = sum_{i in chunk} m[i][j] * w[i]   (grid: (ceil(ys/256), chunks))
__global__ __launch_bounds__(256) void k_col_dot_partial(const fr_t *__restrict__ m, uint32_t xs, uint32_t ys,
                                                        const fr_t *__restrict__ w, fr_t *__restrict__ part, uint32_t rows_per_chunk) {
    uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= ys) return;
    uint32_t i0 = blockIdx.y * rows_per_chunk, i1 = i0 + rows_per_chunk < xs ? i0 + rows_per_chunk : xs;
    fr_t acc = Fr::zero();
    for (uint32_t i = i0; i < i1; i++) acc = Fr::add(acc, Fr::mul(Fr::canon(tk_load(m + (uint64_t)i * ys + j)), tk_load(w + i)));
    tk_store(part + (uint64_t)blockIdx.y * ys + j, acc);
}
__global__ __launch_bounds__(256) void k_col_sum(const fr_t *__restrict__ part, uint32_t chunks, uint32_t ys, fr_t *__restrict__ out) {
    uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= ys) return;
    fr_t acc = Fr::zero();
    for (uint32_t k = 0; k < chunks; k++) acc = Fr::add(acc, tk_load(part + (uint64_t)k * ys + j));
    tk_store(out + j, acc);
}

// ---- div_by_vanishing_opt: P = Q_X (X^c - 1) + Q_Y (Y^d - 1), coefficient recurrences (mod.rs:2308-2367) ----
// acc[lx][y] = sum_b p[b*c + lx][y]
__global__ __launch_bounds__(256) void k_dvo_fold(const fr_t *__restrict__ p, uint32_t xs, uint32_t ys, uint32_t c,
                                                 fr_t *__restrict__ acc) {
    uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (uint64_t)c * ys) return;
    uint32_t lx = (uint32_t)(e / ys), y = (uint32_t)(e - (uint64_t)lx * ys);
    fr_t s = Fr::zero();
    for (uint32_t bx = 0; bx * c < xs; bx++) s = Fr::add(s, Fr::canon(tk_load(p + ((uint64_t)bx * c + lx) * ys + y)));
    tk_store(acc + e, s);
}
// quo_y[x][y] = quo_y[x][y-d] - acc[x][y] for y < ys - d (0 elsewhere): one lane per (x, y mod d), serial over y/d
__global__ __launch_bounds__(256) void k_dvo_quo_y(const fr_t *__restrict__ acc, uint32_t c, uint32_t ys, uint32_t d,
                                                  fr_t *__restrict__ qy) {
    uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (uint64_t)c * d) return;
    uint32_t x = (uint32_t)(e / d), r = (uint32_t)(e - (uint64_t)x * d);
    fr_t prev = Fr::zero();
    for (uint32_t y = r; y < ys; y += d) {
        fr_t v = Fr::zero();
        if (y + d < ys) {  // y < ys - d
            v = Fr::sub(prev, tk_load(acc + (uint64_t)x * ys + y));
            prev = v;
        }
        tk_store(qy + (uint64_t)x * ys + y, v);
    }
}
// b = p with, for x < c:  b[x][y] += q[x][y] (y < ys-d)  and  b[x][y] -= q[x][y-d] (y >= d, y-d < ys-d)
// quo_x[x][y] = quo_x[x-c][y] - b[x][y] for x < xs - c (0 elsewhere): one lane per (x mod c, y), serial over x/c
__global__ __launch_bounds__(256) void k_dvo_quo_x(const fr_t *__restrict__ p, const fr_t *__restrict__ qy, uint32_t xs, uint32_t ys,
                                                  uint32_t c, uint32_t d, fr_t *__restrict__ qx) {
    uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (uint64_t)c * ys) return;
    uint32_t lx = (uint32_t)(e / ys), y = (uint32_t)(e - (uint64_t)lx * ys);
    fr_t prev = Fr::zero();
    for (uint32_t x = lx; x < xs; x += c) {
        fr_t v = Fr::zero();
        if (x + c < xs) {  // x < xs - c
            fr_t b = Fr::canon(tk_load(p + (uint64_t)x * ys + y));
            if (x < c && ys > d) {
                if (y + d < ys) b = Fr::add(b, tk_load(qy + (uint64_t)x * ys + y));
                if (y >= d) b = Fr::sub(b, tk_load(qy + (uint64_t)x * ys + y - d));  // q[y-d] is defined for y-d < ys-d
            }
            v = Fr::sub(prev, b);
            prev = v;
        }
        tk_store(qx + (uint64_t)x * ys + y, v);
    }
}

// ---- div_by_ruffini: P = Q_X (X - x) + R_X(Y),  R_X = Q_Y (Y - y) + R_Y  (mod.rs:2412-2477) ----
// Along X every column is the Horner recurrence B_k = c_k + x B_{k+1} (B_len = 0), with q[k-1] = B_k and remainder B_0
// (_div_uni_coeffs_by_ruffini, mod.rs:2460-2477).  The reference runs it serially per column (rayon over columns);
// here the rows are cut into segments of L: (1) local Horner per (segment, column) with zero carry-in,
// (2) a short serial pass over the segments per column propagates the carries with x^L, (3) each segment reruns
// its recurrence from the true carry-in and writes q.  Critical path 2L + len/L steps instead of len.
__global__ __launch_bounds__(256) void k_ruffini_local(const fr_t *__restrict__ p, uint32_t xs, uint32_t ys, fr_t xm, uint32_t L,
                                                      fr_t *__restrict__ H) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x, sgm = blockIdx.y;
    if (j >= ys) return;
    uint32_t lo = sgm * L, hi = lo + L < xs ? lo + L : xs;
    fr_t h = Fr::zero();
    for (uint32_t k = hi; k-- > lo;) h = Fr::add(Fr::canon(tk_load(p + (uint64_t)k * ys + j)), Fr::mul(h, xm));
    tk_store(H + (uint64_t)sgm * ys + j, h);
}
// carry[s][j] = B at the first row above segment s = sum_{u > s} H[u][j] * (x^L)^(u - s - 1).  One workgroup per column j,
// one lane per segment (S <= 256): geometric suffix scan in LDS, log2(S) product steps instead of a serial walk over the
// segments (which, with only y_size lanes alive, was most of the division's run time).  xLm = x^L (Montgomery)
__global__ __launch_bounds__(256) void k_ruffini_carry(const fr_t *__restrict__ H, uint32_t S, uint32_t ys, fr_t xLm,
                                                      fr_t *__restrict__ carry) {
    __shared__ fr_t sh[256];
    const uint32_t j = blockIdx.x, t = threadIdx.x;
    fr_t v = t < S ? tk_load(H + (uint64_t)t * ys + j) : Fr::zero();
    sh[t] = v;
    fr_t pw = xLm;
    for (uint32_t st = 1; st < 256; st <<= 1) {
        __syncthreads();
        fr_t o = t + st < 256 ? sh[t + st] : Fr::zero();
        __syncthreads();
        v = Fr::add(v, Fr::mul(o, pw));   // plain + plain * mont
        sh[t] = v;
        pw = Fr::sqr(pw);
    }
    __syncthreads();
    if (t < S) tk_store(carry + (uint64_t)t * ys + j, t + 1 < 256 ? sh[t + 1] : Fr::zero());
}
__global__ __launch_bounds__(256) void k_ruffini_apply(const fr_t *__restrict__ p, const fr_t *__restrict__ carry, uint32_t xs, uint32_t ys,
                                                      fr_t xm, uint32_t L, fr_t *__restrict__ qx, fr_t *__restrict__ rx) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x, sgm = blockIdx.y;
    if (j >= ys) return;
    uint32_t lo = sgm * L, hi = lo + L < xs ? lo + L : xs;
    fr_t b = tk_load(carry + (uint64_t)sgm * ys + j);
    if (hi == xs) tk_store(qx + (uint64_t)(xs - 1) * ys + j, Fr::zero());  // top quotient row
    for (uint32_t k = hi; k-- > lo;) {
        b = Fr::add(Fr::canon(tk_load(p + (uint64_t)k * ys + j)), Fr::mul(b, xm));
        if (k >= 1) tk_store(qx + (uint64_t)(k - 1) * ys + j, b);
        else tk_store(rx + j, b);
    }
}
// univariate division of r (length n) by (Y - y): b_k = r[k] + y b_{k+1}, b_n = 0; q[k-1] = b_k (k >= 1), q[n-1] = 0, rem = b_0.
// One workgroup: every lane takes a run of Ls coefficients (local Horner), the runs are chained by a geometric suffix
// scan in LDS, then every lane replays its run with the incoming carry.  (A single lane walking all n coefficients was
// a 512-step dependent chain of field products on an otherwise idle GPU.)
__global__ __launch_bounds__(256) void k_ruffini_y(const fr_t *__restrict__ r, uint32_t n, fr_t ym, fr_t *__restrict__ q, fr_t *__restrict__ rem) {
    __shared__ fr_t sh[256];
    const uint32_t t = threadIdx.x;
    const uint32_t Ls = (n + 255) / 256;
    const uint32_t lo = t * Ls < n ? t * Ls : n, hi = lo + Ls < n ? lo + Ls : n;
    fr_t h = Fr::zero();
    for (uint32_t k = hi; k-- > lo;) h = Fr::add(Fr::canon(tk_load(r + k)), Fr::mul(h, ym));
    sh[t] = h;
    fr_t pw = Fr::pow_u64(ym, Ls), v = h;
    for (uint32_t st = 1; st < 256; st <<= 1) {
        __syncthreads();
        fr_t o = t + st < 256 ? sh[t + st] : Fr::zero();
        __syncthreads();
        v = Fr::add(v, Fr::mul(o, pw));
        sh[t] = v;
        pw = Fr::sqr(pw);
    }
    __syncthreads();
    fr_t b = t + 1 < 256 ? sh[t + 1] : Fr::zero();   // value of the recurrence just above this lane's run
    if (t == 0) tk_store(q + n - 1, Fr::zero());
    for (uint32_t k = hi; k-- > lo;) {
        b = Fr::add(Fr::canon(tk_load(r + k)), Fr::mul(b, ym));
        if (k >= 1) tk_store(q + k - 1, b);
        else tk_store(rem, b);
    }
}

// ---------------------------------------------------------------------------------------------------
static unsigned stream_grid(uint64_t n) {
    uint64_t g = (n + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    return (unsigned)(g ? g : 1);
}
static fr_t fr_in(const tkmk_fr *x) {
    fr_t r;
    for (int i = 0; i < 8; i++) r.l[i] = x->limbs[i];
    return Fr::canon(r);
}
static bool fr_is_one(const fr_t &x) {
    fr_t o = Fr::zero();
    o.l[0] = 1;
    return Fr::eq(x, o);
}

TK_API tkmk_error tkmk_poly_find_degree(const tkmk_fr *coeffs_dev, uint32_t x_size, uint32_t y_size, int64_t *x_degree,
                                        int64_t *y_degree, tkmk_stream stream) {
    if (!coeffs_dev || !x_degree || !y_degree) return TKMK_ERR_INVALID_POINTER;
    if (!x_size || !y_size) return TKMK_ERR_INVALID_ARGUMENT;
    TK_TRY(tk_require_device());
    tk_stat_add(TK_STAT_POLY_ELEMENTS, (uint64_t)x_size * y_size);   // elements this streaming pass touches (bench.py / the sharded prover's per-rank accounting)
    hipStream_t s = tk_stream(stream);
    tk_frame frame(s);
    tk_scratch d;
    TK_TRY(d.alloc(8, s));
    int init[2] = {-1, -1}, res[2];
    TK_HIP(hipMemcpyAsync(d.p, init, 8, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_find_degree, stream_grid((uint64_t)x_size * y_size), 256, 0, s, (const fr_t *)coeffs_dev, x_size, y_size,
                       d.as<int>());
    TK_HIP(hipGetLastError());
    TK_HIP(hipMemcpyAsync(res, d.p, 8, hipMemcpyDeviceToHost, s));
    TK_HIP(hipStreamSynchronize(s));
    *x_degree = res[0];
    *y_degree = res[1];
    return TKMK_SUCCESS;
}

TK_API tkmk_error tkmk_poly_place(const tkmk_fr *src_dev, uint32_t sx, uint32_t sy, tkmk_fr *dst_dev, uint32_t dx, uint32_t dy,
                                  uint32_t off_x, uint32_t off_y, tkmk_stream stream) {
    if (!src_dev || !dst_dev) return TKMK_ERR_INVALID_POINTER;
    if (!sx || !sy || !dx || !dy || (const void *)src_dev == (void *)dst_dev) return TKMK_ERR_INVALID_ARGUMENT;
    TK_TRY(tk_require_device());
    tk_stat_add(TK_STAT_POLY_ELEMENTS, (uint64_t)dx * dy);   // elements this streaming pass touches (bench.py / the sharded prover's per-rank accounting)
    hipStream_t s = tk_stream(stream);
    hipLaunchKernelGGL(k_place, stream_grid((uint64_t)dx * dy), 256, 0, s, (const fr_t *)src_dev, sx, sy, (fr_t *)dst_dev, dx, dy, off_x,
                       off_y);
    TK_HIP(hipGetLastError());
    return TKMK_SUCCESS;
}

// out (out_xs x out_ys, fully written) = sum_t coeffs[t] * X^off_x[t] Y^off_y[t] * polys[t]; polys[t] is an x_sizes[t] x y_sizes[t]
// coefficient matrix; a shifted operand must fit: x_sizes[t] + off_x[t] <= out_xs (same for y).  coeffs are HOST scalars (plain);
// off_x / off_y may be NULL (no shifts).  `out` must not alias an operand.  More than LC_MAX_TERMS terms are folded in groups.
TK_API tkmk_error tkmk_poly_lincomb(uint32_t n_terms, const tkmk_fr *coeffs_host, const tkmk_fr *const *polys_dev, const uint32_t *x_sizes,
                                    const uint32_t *y_sizes, const uint32_t *off_x, const uint32_t *off_y, tkmk_fr *out_dev, uint32_t out_xs,
                                    uint32_t out_ys, tkmk_stream stream) {
    if (!out_dev || (n_terms && (!coeffs_host || !polys_dev || !x_sizes || !y_sizes))) return TKMK_ERR_INVALID_POINTER;
    if (!out_xs || !out_ys) return TKMK_ERR_INVALID_ARGUMENT;
    for (uint32_t t = 0; t < n_terms; t++) {
        if (!polys_dev[t]) return TKMK_ERR_INVALID_POINTER;
        if (polys_dev[t] == out_dev) return TKMK_ERR_INVALID_ARGUMENT;
        uint64_t ox = off_x ? off_x[t] : 0, oy = off_y ? off_y[t] : 0;
        if (!x_sizes[t] || !y_sizes[t] || x_sizes[t] + ox > out_xs || y_sizes[t] + oy > out_ys) return TKMK_ERR_INVALID_ARGUMENT;
    }
    TK_TRY(tk_require_device());
    tk_stat_add(TK_STAT_POLY_ELEMENTS, (uint64_t)out_xs * out_ys);   // elements this streaming pass touches (bench.py / the sharded prover's per-rank accounting)
    hipStream_t s = tk_stream(stream);
    if (n_terms == 0) {
        TK_HIP(hipMemsetAsync(out_dev, 0, (size_t)out_xs * out_ys * sizeof(fr_t), s));
        return TKMK_SUCCESS;
    }
    fr_t one = Fr::zero(), minus_one;
    one.l[0] = 1;
    minus_one = Fr::from_mont(Fr::neg(Fr::to_mont(one)));
    tk_frame frame(s);
    tk_scratch partial;   // only when the terms do not fit one launch
    const fr_t *carry = nullptr;
    for (uint32_t first = 0; first < n_terms;) {
        lincomb_args_t a;
        a.n = 0;
        if (carry) {      // the sum so far re-enters as a unit-coefficient term
            a.p[0] = carry, a.xs[0] = out_xs, a.ys[0] = out_ys, a.ox[0] = a.oy[0] = 0, a.kind[0] = 1, a.c[0] = Fr::zero();
            a.n = 1;
        }
        while (first < n_terms && a.n < LC_MAX_TERMS) {
            fr_t c = Fr::canon(fr_in(coeffs_host + first));
            uint32_t k = a.n++;
            a.p[k] = (const fr_t *)polys_dev[first];
            a.xs[k] = x_sizes[first], a.ys[k] = y_sizes[first];
            a.ox[k] = off_x ? off_x[first] : 0, a.oy[k] = off_y ? off_y[first] : 0;
            a.kind[k] = Fr::eq(c, one) ? 1 : Fr::eq(c, minus_one) ? 2 : 0;
            a.c[k] = Fr::to_mont(c);
            first++;
        }
        fr_t *dst = (fr_t *)out_dev;
        if (first < n_terms) {   // more groups follow: accumulate in scratch, alternate so that no launch reads what it writes
            tk_scratch nxt;
            TK_TRY(nxt.alloc((size_t)out_xs * out_ys * sizeof(fr_t), s));
            dst = nxt.as<fr_t>();
            partial = nxt;
        }
        int dy_shift = -1;
        if ((out_ys & (out_ys - 1)) == 0)
            for (dy_shift = 0; (1u << dy_shift) < out_ys; dy_shift++) {}
        if (out_ys % 2 == 0) hipLaunchKernelGGL(k_lincomb<2>, stream_grid((uint64_t)out_xs * out_ys / 2), 256, 0, s, a, dst, out_xs, out_ys, dy_shift);
        else hipLaunchKernelGGL(k_lincomb<1>, stream_grid((uint64_t)out_xs * out_ys), 256, 0, s, a, dst, out_xs, out_ys, dy_shift);
        TK_HIP(hipGetLastError());
        carry = dst;
    }
    return TKMK_SUCCESS;
}

static tkmk_error powers_table(tk_scratch &t, const fr_t &g_plain, uint64_t n, hipStream_t s) {
    TK_TRY(t.alloc(n * sizeof(fr_t), s));
    hipLaunchKernelGGL(k_powers, tk_div_up(n, 256), 256, 0, s, t.as<fr_t>(), Fr::to_mont(g_plain), n);
    TK_HIP(hipGetLastError());
    return TKMK_SUCCESS;
}

TK_API tkmk_error tkmk_poly_scale_coeffs(const tkmk_fr *src_dev, uint32_t x_size, uint32_t y_size, const tkmk_fr *factor_x,
                                         const tkmk_fr *factor_y, tkmk_fr *dst_dev, tkmk_stream stream) {
    if (!src_dev || !dst_dev) return TKMK_ERR_INVALID_POINTER;
    if (!x_size || !y_size) return TKMK_ERR_INVALID_ARGUMENT;
    TK_TRY(tk_require_device());
    tk_stat_add(TK_STAT_POLY_ELEMENTS, (uint64_t)x_size * y_size);   // elements this streaming pass touches (bench.py / the sharded prover's per-rank accounting)
    hipStream_t s = tk_stream(stream);
    tk_frame frame(s);
    tk_scratch tx, ty;
    const fr_t *px = nullptr, *py = nullptr;
    if (factor_x && !fr_is_one(fr_in(factor_x))) {
        TK_TRY(powers_table(tx, fr_in(factor_x), x_size, s));
        px = tx.as<fr_t>();
    }
    if (factor_y && !fr_is_one(fr_in(factor_y))) {
        TK_TRY(powers_table(ty, fr_in(factor_y), y_size, s));
        py = ty.as<fr_t>();
    }
    hipLaunchKernelGGL(k_scale, stream_grid((uint64_t)x_size * y_size), 256, 0, s, (const fr_t *)src_dev, (fr_t *)dst_dev, x_size, y_size,
                       px, py, 0);
    TK_HIP(hipGetLastError());
    return TKMK_SUCCESS;
}

// ---- product with scale * (1 + X + ... + X^(m-1)) in coefficient space: out[k][j] = scale * sum_{i = k-m+1 .. k} p[i][j] ----
// The Lagrange polynomial of index 0 on the m-th roots of unity is exactly that (all coefficients 1/m): prove2 and prove4 multiply
// 2^23..2^24-coefficient polynomials by it (K0 * ..., lib.rs:2238-2246, 3012-3040), which the reference does with three bivariate
// NTTs per product.  A sliding-window sum is two running sums over X: block sums, their scan, then one pass.
// T[b][j] = sum of rows [b * WS_BLOCK, (b + 1) * WS_BLOCK) of column j (rows >= x_size are zero)
__global__ __launch_bounds__(256) void k_ws_block_sums(const fr_t *__restrict__ p, uint32_t xs, uint32_t ys, uint32_t nb, uint32_t WS_BLOCK, fr_t *__restrict__ T) {
    uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (uint64_t)nb * ys) return;
    uint32_t b = (uint32_t)(id / ys), j = (uint32_t)(id - (uint64_t)b * ys);
    fr_t acc = Fr::zero();
    for (uint32_t r = 0; r < WS_BLOCK; r++) {
        uint32_t i = b * WS_BLOCK + r;
        if (i < xs) acc = Fr::add(acc, Fr::canon(tk_load(p + (uint64_t)i * ys + j)));
    }
    tk_store(T + id, acc);
}
// exclusive scan of T along b, per column (nb is a few hundred at most)
__global__ __launch_bounds__(256) void k_ws_scan(fr_t *__restrict__ T, uint32_t nb, uint32_t ys) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ys) return;
    fr_t run = Fr::zero();
    for (uint32_t b = 0; b < nb; b++) {
        fr_t v = tk_load(T + (uint64_t)b * ys + j);
        tk_store(T + (uint64_t)b * ys + j, run);
        run = Fr::add(run, v);
    }
}
// out rows of block b: leading running sum S[k] minus the trailing one S[k - m]; m is a multiple of WS_BLOCK, so the trailing
// sum starts at a block boundary too
__global__ __launch_bounds__(256) void k_ws_apply(const fr_t *__restrict__ p, const fr_t *__restrict__ T, uint32_t xs, uint32_t ys, uint32_t nb,
                                                 uint32_t WS_BLOCK, uint32_t m, fr_t scale_mont, uint32_t out_xs, fr_t *__restrict__ out) {
    uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t ob = (out_xs + WS_BLOCK - 1) / WS_BLOCK;
    if (id >= (uint64_t)ob * ys) return;
    uint32_t b = (uint32_t)(id / ys), j = (uint32_t)(id - (uint64_t)b * ys);
    const uint32_t mb = m / WS_BLOCK;
    // sums of all rows below the two blocks (past the last block of p: the total = prefix of the last block + that block)
    auto prefix = [&](uint32_t blk) {
        if (blk < nb) return tk_load(T + (uint64_t)blk * ys + j);
        fr_t t = tk_load(T + (uint64_t)(nb - 1) * ys + j);
        for (uint32_t r = 0; r < WS_BLOCK; r++) {
            uint32_t i = (nb - 1) * WS_BLOCK + r;
            if (i < xs) t = Fr::add(t, Fr::canon(tk_load(p + (uint64_t)i * ys + j)));
        }
        return t;
    };
    fr_t lead = prefix(b), trail = b >= mb ? prefix(b - mb) : Fr::zero();
    for (uint32_t r = 0; r < WS_BLOCK; r++) {
        uint32_t k = b * WS_BLOCK + r;
        if (k >= out_xs) break;
        if (k < xs) lead = Fr::add(lead, Fr::canon(tk_load(p + (uint64_t)k * ys + j)));
        if (k >= m && k - m < xs) trail = Fr::add(trail, Fr::canon(tk_load(p + (uint64_t)(k - m) * ys + j)));
        tk_store(out + (uint64_t)k * ys + j, Fr::mul(Fr::sub(lead, trail), scale_mont));   // plain * Montgomery = plain
    }
}
TK_API tkmk_error tkmk_poly_mul_ones_x(const tkmk_fr *p_dev, uint32_t x_size, uint32_t y_size, uint32_t m, const tkmk_fr *scale,
                                       uint32_t out_x_size, tkmk_fr *out_dev, tkmk_stream stream) {
    if (!p_dev || !out_dev || !scale) return TKMK_ERR_INVALID_POINTER;
    if (!x_size || !y_size || !out_x_size || !m || p_dev == out_dev) return TKMK_ERR_INVALID_ARGUMENT;
    uint32_t WS_BLOCK = 64;   // rows per running-sum block: the largest power of two <= 64 that divides m
    while (m % WS_BLOCK) WS_BLOCK >>= 1;
    TK_TRY(tk_require_device());
    tk_stat_add(TK_STAT_POLY_ELEMENTS, (uint64_t)out_x_size * y_size);   // elements this streaming pass touches (bench.py / the sharded prover's per-rank accounting)
    hipStream_t s = tk_stream(stream);
    tk_frame frame(s);
    const uint32_t nb = (x_size + WS_BLOCK - 1) / WS_BLOCK, ob = (out_x_size + WS_BLOCK - 1) / WS_BLOCK;
    tk_scratch T;
    TK_TRY(T.alloc((size_t)nb * y_size * sizeof(fr_t), s));
    hipLaunchKernelGGL(k_ws_block_sums, tk_div_up((uint64_t)nb * y_size, 256), 256, 0, s, (const fr_t *)p_dev, x_size, y_size, nb, WS_BLOCK, T.as<fr_t>());
    hipLaunchKernelGGL(k_ws_scan, tk_div_up(y_size, 256), 256, 0, s, T.as<fr_t>(), nb, y_size);
    hipLaunchKernelGGL(k_ws_apply, tk_div_up((uint64_t)ob * y_size, 256), 256, 0, s, (const fr_t *)p_dev, (const fr_t *)T.p, x_size, y_size, nb, WS_BLOCK, m,
                       Fr::to_mont(Fr::canon(fr_in(scale))), out_x_size, (fr_t *)out_dev);
    TK_HIP(hipGetLastError());
    return TKMK_SUCCESS;
}

// dst[i][j] = evals[i][j] * (w_x^i - 1): the evaluation-domain form of "multiply by (X - 1)" used by
// PolyExpr::MulXMinusOne (mod.rs:372-378, x_minus_one_evals :504-518); w_x = root of unity of order x_size
TK_API tkmk_error tkmk_poly_mul_x_minus_one_evals(const tkmk_fr *evals_dev, uint32_t x_size, uint32_t y_size, tkmk_fr *dst_dev,
                                                  tkmk_stream stream) {
    if (!evals_dev || !dst_dev) return TKMK_ERR_INVALID_POINTER;
    if (!x_size || !y_size || (x_size & (x_size - 1))) return TKMK_ERR_INVALID_ARGUMENT;
    TK_TRY(tk_require_device());
    hipStream_t s = tk_stream(stream);
    tk_frame frame(s);
    tkmk_fr w;
    TK_TRY(bls12_381_get_root_of_unity(x_size, &w));
    tk_scratch tx;
    TK_TRY(powers_table(tx, fr_in(&w), x_size, s));
    hipLaunchKernelGGL(k_scale, stream_grid((uint64_t)x_size * y_size), 256, 0, s, (const fr_t *)evals_dev, (fr_t *)dst_dev, x_size, y_size,
                       (const fr_t *)tx.p, (const fr_t *)nullptr, 1);
    TK_HIP(hipGetLastError());
    return TKMK_SUCCESS;
}

// out_dev[j] = P(x, Y) coefficients (y_size values)
TK_API tkmk_error tkmk_poly_eval_x(const tkmk_fr *coeffs_dev, uint32_t x_size, uint32_t y_size, const tkmk_fr *x, tkmk_fr *out_dev,
                                   tkmk_stream stream) {
    if (!coeffs_dev || !x || !out_dev) return TKMK_ERR_INVALID_POINTER;
    if (!x_size || !y_size) return TKMK_ERR_INVALID_ARGUMENT;
    TK_TRY(tk_require_device());
    tk_stat_add(TK_STAT_POLY_ELEMENTS, (uint64_t)x_size * y_size);   // elements this streaming pass touches (bench.py / the sharded prover's per-rank accounting)
    hipStream_t s = tk_stream(stream);
    tk_frame frame(s);
    tk_scratch tx, part;
    TK_TRY(powers_table(tx, fr_in(x), x_size, s));
    uint32_t rows_per_chunk = 64, chunks = (x_size + rows_per_chunk - 1) / rows_per_chunk;
    TK_TRY(part.alloc((size_t)chunks * y_size * sizeof(fr_t), s));
    hipLaunchKernelGGL(k_col_dot_partial, dim3(tk_div_up(y_size, 256), chunks), 256, 0, s, (const fr_t *)coeffs_dev, x_size, y_size,
                       (const fr_t *)tx.p, part.as<fr_t>(), rows_per_chunk);
    hipLaunchKernelGGL(k_col_sum, tk_div_up(y_size, 256), 256, 0, s, (const fr_t *)part.p, chunks, y_size, (fr_t *)out_dev);
    TK_HIP(hipGetLastError());
    return TKMK_SUCCESS;
}
// out_dev[i] = P(X, y) coefficients (x_size values)
TK_API tkmk_error tkmk_poly_eval_y(const tkmk_fr *coeffs_dev, uint32_t x_size, uint32_t y_size, const tkmk_fr *y, tkmk_fr *out_dev,
                                   tkmk_stream stream) {
    if (!coeffs_dev || !y || !out_dev) return TKMK_ERR_INVALID_POINTER;
    if (!x_size || !y_size) return TKMK_ERR_INVALID_ARGUMENT;
    TK_TRY(tk_require_device());
    hipStream_t s = tk_stream(stream);
    tk_frame frame(s);
    tk_scratch ty;
    TK_TRY(powers_table(ty, fr_in(y), y_size, s));
    hipLaunchKernelGGL(k_row_dot, x_size, 256, 0, s, (const fr_t *)coeffs_dev, y_size, (const fr_t *)ty.p, (fr_t *)out_dev);
    TK_HIP(hipGetLastError());
    return TKMK_SUCCESS;
}
// *out (host) = P(x, y)
TK_API tkmk_error tkmk_poly_eval(const tkmk_fr *coeffs_dev, uint32_t x_size, uint32_t y_size, const tkmk_fr *x, const tkmk_fr *y,
                                 tkmk_fr *out_host, tkmk_stream stream) {
    if (!coeffs_dev || !x || !y || !out_host) return TKMK_ERR_INVALID_POINTER;
    if (!x_size || !y_size) return TKMK_ERR_INVALID_ARGUMENT;
    TK_TRY(tk_require_device());
    tk_stat_add(TK_STAT_POLY_ELEMENTS, (uint64_t)x_size * y_size);   // elements this streaming pass touches (bench.py / the sharded prover's per-rank accounting)
    hipStream_t s = tk_stream(stream);
    tk_frame frame(s);
    tk_scratch tx, ty, rows, res;
    TK_TRY(powers_table(tx, fr_in(x), x_size, s));
    TK_TRY(powers_table(ty, fr_in(y), y_size, s));
    TK_TRY(rows.alloc((size_t)x_size * sizeof(fr_t), s));
    TK_TRY(res.alloc(sizeof(fr_t), s));
    hipLaunchKernelGGL(k_row_dot, x_size, 256, 0, s, (const fr_t *)coeffs_dev, y_size, (const fr_t *)ty.p, rows.as<fr_t>());
    hipLaunchKernelGGL(k_row_dot, 1, 256, 0, s, (const fr_t *)rows.p, x_size, (const fr_t *)tx.p, res.as<fr_t>());
    TK_HIP(hipGetLastError());
    TK_HIP(hipMemcpyAsync(out_host, res.p, sizeof(fr_t), hipMemcpyDeviceToHost, s));
    TK_HIP(hipStreamSynchronize(s));
    return TKMK_SUCCESS;
}

// p: x_size x y_size with c | x_size, d | y_size (the reference optimize_size()s first).  quo_x: x_size x y_size,
// quo_y: c x y_size.  In place is not allowed.
TK_API tkmk_error tkmk_poly_div_by_vanishing_opt(const tkmk_fr *p_dev, uint32_t x_size, uint32_t y_size, uint32_t c, uint32_t d,
                                                 tkmk_fr *quo_x_dev, tkmk_fr *quo_y_dev, tkmk_stream stream) {
    if (!p_dev || !quo_x_dev || !quo_y_dev) return TKMK_ERR_INVALID_POINTER;
    if (!c || !d || (c & (c - 1)) || (d & (d - 1)) || !x_size || !y_size || x_size % c || y_size % d) return TKMK_ERR_INVALID_ARGUMENT;
    TK_TRY(tk_require_device());
    tk_stat_add(TK_STAT_POLY_ELEMENTS, (uint64_t)x_size * y_size);   // elements this streaming pass touches (bench.py / the sharded prover's per-rank accounting)
    hipStream_t s = tk_stream(stream);
    tk_frame frame(s);
    tk_scratch acc;
    TK_TRY(acc.alloc((size_t)c * y_size * sizeof(fr_t), s));
    hipLaunchKernelGGL(k_dvo_fold, tk_div_up((uint64_t)c * y_size, 256), 256, 0, s, (const fr_t *)p_dev, x_size, y_size, c, acc.as<fr_t>());
    hipLaunchKernelGGL(k_dvo_quo_y, tk_div_up((uint64_t)c * d, 256), 256, 0, s, (const fr_t *)acc.p, c, y_size, d, (fr_t *)quo_y_dev);
    hipLaunchKernelGGL(k_dvo_quo_x, tk_div_up((uint64_t)c * y_size, 256), 256, 0, s, (const fr_t *)p_dev, (const fr_t *)quo_y_dev, x_size,
                       y_size, c, d, (fr_t *)quo_x_dev);
    TK_HIP(hipGetLastError());
    return TKMK_SUCCESS;
}

// q_x: x_size x y_size, q_y: y_size values (1 x y_size), *r_host = P(x, y)
TK_API tkmk_error tkmk_poly_div_by_ruffini(const tkmk_fr *p_dev, uint32_t x_size, uint32_t y_size, const tkmk_fr *x, const tkmk_fr *y,
                                           tkmk_fr *q_x_dev, tkmk_fr *q_y_dev, tkmk_fr *r_host, tkmk_stream stream) {
    if (!p_dev || !x || !y || !q_x_dev || !q_y_dev || !r_host) return TKMK_ERR_INVALID_POINTER;
    if (!x_size || !y_size) return TKMK_ERR_INVALID_ARGUMENT;
    TK_TRY(tk_require_device());
    tk_stat_add(TK_STAT_POLY_ELEMENTS, (uint64_t)x_size * y_size);   // elements this streaming pass touches (bench.py / the sharded prover's per-rank accounting)
    hipStream_t s = tk_stream(stream);
    tk_frame frame(s);
    tk_scratch rx, rem;
    TK_TRY(rx.alloc((size_t)y_size * sizeof(fr_t), s));
    TK_TRY(rem.alloc(sizeof(fr_t), s));
    const fr_t xm = Fr::to_mont(fr_in(x));
    if (x_size < 2) {
        // _div_uni_coeffs_by_ruffini with len < 2: quotient (0), remainder = the coefficient
        TK_HIP(hipMemsetAsync(q_x_dev, 0, (size_t)y_size * sizeof(fr_t), s));
        TK_HIP(hipMemcpyAsync(rx.p, p_dev, (size_t)y_size * sizeof(fr_t), hipMemcpyDeviceToDevice, s));
    } else {
        // S <= 256 segments (one lane each in k_ruffini_carry) of L rows: 32-row runs up to 8192 rows, longer above
        uint32_t L = x_size >= 64 ? 32 : x_size;
        while ((x_size + L - 1) / L > 256) L *= 2;
        const uint32_t S = (x_size + L - 1) / L;
        tk_scratch H, carry;
        TK_TRY(H.alloc((size_t)S * y_size * sizeof(fr_t), s));
        TK_TRY(carry.alloc((size_t)S * y_size * sizeof(fr_t), s));
        dim3 grid(tk_div_up(y_size, 256), S);
        hipLaunchKernelGGL(k_ruffini_local, grid, 256, 0, s, (const fr_t *)p_dev, x_size, y_size, xm, L, H.as<fr_t>());
        hipLaunchKernelGGL(k_ruffini_carry, y_size, 256, 0, s, (const fr_t *)H.p, S, y_size, Fr::pow_u64(xm, L),
                           carry.as<fr_t>());
        hipLaunchKernelGGL(k_ruffini_apply, grid, 256, 0, s, (const fr_t *)p_dev, (const fr_t *)carry.p, x_size, y_size, xm, L,
                           (fr_t *)q_x_dev, rx.as<fr_t>());
    }
    hipLaunchKernelGGL(k_ruffini_y, 1, 256, 0, s, (const fr_t *)rx.p, y_size, Fr::to_mont(fr_in(y)), (fr_t *)q_y_dev, rem.as<fr_t>());
    TK_HIP(hipGetLastError());
    TK_HIP(hipMemcpyAsync(r_host, rem.p, sizeof(fr_t), hipMemcpyDeviceToHost, s));
    TK_HIP(hipStreamSynchronize(s));
    return TKMK_SUCCESS;
}
