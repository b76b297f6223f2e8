// vecops.hip — element-wise Fr kernels behind the bls12_381_vector_* / scalar_*_vec / matrix_transpose
// entries (include/tkmk.h).  Work-alike of icicle_core::vec_ops::VecOps<ScalarField> as the reference
// uses it (packages/backend/libs/src/vector_operations/mod.rs:34-139; bivariate_polynomial/mod.rs:332-435).
//
// All of these are HBM-streaming kernels: 32 B in (x2) / 32 B out per element, one element per lane,
// two dwordx4 accesses per operand so a wave touches 2 KiB of contiguous memory per operand.
// Data stays in plain (non-Montgomery) form at rest; mul needs two Montgomery products (a*R^2/R*b/R).
#include "common.h"

enum { OP_ADD = 0, OP_SUB, OP_MUL, OP_DIV, OP_INV, OP_SADD, OP_SSUB, OP_SMUL };

template <int OP>
__global__ __launch_bounds__(256) void k_vec(const fr_t *__restrict__ a, const fr_t *__restrict__ b,
                                            fr_t *__restrict__ out, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    fr_t s;
    if (OP == OP_SADD || OP == OP_SSUB || OP == OP_SMUL) {
        s = Fr::canon(tk_load(a));
        if (OP == OP_SMUL) s = Fr::to_mont(s);
    }
    for (; i < n; i += stride) {
        fr_t r;
        if (OP == OP_ADD) r = Fr::add(Fr::canon(tk_load(a + i)), Fr::canon(tk_load(b + i)));
        if (OP == OP_SUB) r = Fr::sub(Fr::canon(tk_load(a + i)), Fr::canon(tk_load(b + i)));
        if (OP == OP_MUL) r = Fr::mul(Fr::to_mont(tk_load(a + i)), tk_load(b + i));
        if (OP == OP_SADD) r = Fr::add(s, Fr::canon(tk_load(b + i)));
        if (OP == OP_SSUB) r = Fr::sub(s, Fr::canon(tk_load(b + i)));
        if (OP == OP_SMUL) r = Fr::mul(s, tk_load(b + i));
                tk_store(out + i, r);
    }
}
// Division / inversion: K elements per lane share one Fermat inversion (Montgomery's trick: 1 inversion + 6 products
// per element instead of ~380).  K = 16 for long vectors: the K prefix products stay in registers and the operands are
// read again on the way back (a second coalesced read is far cheaper than 16 more live field elements per lane).
template <bool DIV, int K>
__global__ __launch_bounds__(256) void k_vec_inv(const fr_t *__restrict__ a, const fr_t *__restrict__ b,
                                                fr_t *__restrict__ out, uint64_t n) {
    // lane handles K stride-S elements: i, i+S, ..., i+(K-1)S (S = total threads) so that every load/store instruction
    // stays coalesced.  Denominators in Montgomery form, zero-safe: inv(0) = 0 (a zero contributes 1 to the products).
    uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t S = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t base = tid; base < n; base += (uint64_t)K * S) {
        fr_t pre[K];
        fr_t acc = Fr::one();
#pragma unroll
        for (int k = 0; k < K; k++) {
            uint64_t i = base + k * S;
            pre[k] = acc;
            if (i < n) {
                fr_t v = Fr::canon(tk_load((DIV ? b : a) + i));
                if (!Fr::is_zero(v)) acc = Fr::mul(acc, Fr::to_mont(v));
            }
        }
        fr_t ia = Fr::inv(acc);  // (prod of the non-zero x_k)^-1, Montgomery
#pragma unroll
        for (int k = K - 1; k >= 0; k--) {
            uint64_t i = base + k * S;
            if (i >= n) continue;
            fr_t v = Fr::canon(tk_load((DIV ? b : a) + i));   // read again instead of keeping K operands live
            fr_t r = Fr::zero();
            if (!Fr::is_zero(v)) {
                fr_t xi = Fr::mul(ia, pre[k]);  // x_k^-1 (Montgomery)
                ia = Fr::mul(ia, Fr::to_mont(v));
                r = DIV ? Fr::mul(xi, Fr::canon(tk_load(a + i))) : Fr::from_mont(xi);  // Mont * plain -> plain
            }
            tk_store(out + i, r);
        }
    }
}

// block-wide reduction of one vector chunk; OP 0 = sum, 1 = product (Montgomery domain for product)
template <int OP>
__global__ __launch_bounds__(256) void k_reduce(const fr_t *__restrict__ in, fr_t *__restrict__ out, uint64_t n,
                                               uint64_t vec_stride, uint64_t elem_stride, int in_plain) {
    // grid: (chunks, batch). in element j of vector b at b*vec_stride + j*elem_stride
    __shared__ fr_t sh[256];
    const fr_t *v = in + (uint64_t)blockIdx.y * vec_stride;
    fr_t acc = OP == 0 ? Fr::zero() : Fr::one();
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x) {
        fr_t x = Fr::canon(tk_load(v + j * elem_stride));
        if (OP == 0) acc = Fr::add(acc, x);
        else acc = Fr::mul(acc, in_plain ? Fr::to_mont(x) : x);
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            fr_t y = sh[threadIdx.x + s];
            acc = OP == 0 ? Fr::add(acc, y) : Fr::mul(acc, y);
            sh[threadIdx.x] = acc;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (OP == 1 && gridDim.x == 1) acc = Fr::from_mont(acc);  // last stage: Montgomery -> plain
        tk_store(out + (uint64_t)blockIdx.y * gridDim.x + blockIdx.x, acc);
    }
}

// rows x cols -> cols x rows through a 16 x 16 LDS tile of 32-byte elements (+1 column padding)
__global__ __launch_bounds__(256) void k_transpose(const fr_t *__restrict__ in, fr_t *__restrict__ out, uint32_t rows,
                                                  uint32_t cols) {
    __shared__ uint4 lo[16][17], hi[16][17];
    uint32_t tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    uint32_t c = blockIdx.x * 16 + tx, r = blockIdx.y * 16 + ty;
    if (r < rows && c < cols) {
        const uint4 *s = reinterpret_cast<const uint4 *>(in + (uint64_t)r * cols + c);
        lo[ty][tx] = s[0];
        hi[ty][tx] = s[1];
    }
    __syncthreads();
    uint32_t orow = blockIdx.x * 16 + ty, ocol = blockIdx.y * 16 + tx;  // out is cols x rows
    if (orow < cols && ocol < rows) {
        uint4 *d = reinterpret_cast<uint4 *>(out + (uint64_t)orow * rows + ocol);
        d[0] = lo[tx][ty];
        d[1] = hi[tx][ty];
    }
}

static unsigned grid_for(uint64_t n, unsigned per_block) {
    uint64_t g = (n + per_block - 1) / per_block;
    if (g > 256 * 16) g = 256 * 16;  // grid-stride the rest (256 CUs x 16 blocks)
    if (g == 0) g = 1;
    return (unsigned)g;
}

static tkmk_error vec_entry(int op, const tkmk_fr *a, const tkmk_fr *b, uint64_t n, const tkmk_vecops_config *cfg,
                            tkmk_fr *out) {
    if (!cfg || cfg->ext) return TKMK_ERR_INVALID_ARGUMENT;
    TK_TRY(tk_require_device());
    if (n == 0) return TKMK_SUCCESS;
    uint64_t total = n * (uint64_t)(cfg->batch_size > 0 ? cfg->batch_size : 1);
    hipStream_t s = tk_stream(cfg->stream_handle);
    tk_frame frame(s);
    bool scalar_a = op == OP_SADD || op == OP_SSUB || op == OP_SMUL;
    bool unary = op == OP_INV;
    tk_staged A, B, O;
    TK_TRY(A.in(a, scalar_a ? 32 : total * 32, cfg->is_a_on_device, s));
    if (!unary) TK_TRY(B.in(b, total * 32, cfg->is_b_on_device, s));
    TK_TRY(O.out(out, total * 32, cfg->is_result_on_device, s));
    const fr_t *pa = (const fr_t *)A.dev, *pb = (const fr_t *)B.dev;
    fr_t *po = (fr_t *)O.dev;
    unsigned g = grid_for(total, 256);
    switch (op) {
        case OP_ADD: hipLaunchKernelGGL(k_vec<OP_ADD>, g, 256, 0, s, pa, pb, po, total); break;
        case OP_SUB: hipLaunchKernelGGL(k_vec<OP_SUB>, g, 256, 0, s, pa, pb, po, total); break;
        case OP_MUL: hipLaunchKernelGGL(k_vec<OP_MUL>, g, 256, 0, s, pa, pb, po, total); break;
        case OP_SADD: hipLaunchKernelGGL(k_vec<OP_SADD>, g, 256, 0, s, pa, pb, po, total); break;
        case OP_SSUB: hipLaunchKernelGGL(k_vec<OP_SSUB>, g, 256, 0, s, pa, pb, po, total); break;
        case OP_SMUL: hipLaunchKernelGGL(k_vec<OP_SMUL>, g, 256, 0, s, pa, pb, po, total); break;
        case OP_DIV:
            if (total >= (1u << 16)) hipLaunchKernelGGL((k_vec_inv<true, 16>), grid_for(total, 4096), 256, 0, s, pa, pb, po, total);
            else hipLaunchKernelGGL((k_vec_inv<true, 4>), grid_for(total, 1024), 256, 0, s, pa, pb, po, total);
            break;
        case OP_INV:
            if (total >= (1u << 16)) hipLaunchKernelGGL((k_vec_inv<false, 16>), grid_for(total, 4096), 256, 0, s, pa, pa, po, total);
            else hipLaunchKernelGGL((k_vec_inv<false, 4>), grid_for(total, 1024), 256, 0, s, pa, pa, po, total);
            break;
        default: return TKMK_ERR_INVALID_ARGUMENT;
    }
    TK_HIP(hipGetLastError());
    TK_TRY(O.copy_back(out, total * 32, cfg->is_result_on_device, s));
    if (!cfg->is_async || !cfg->is_result_on_device) TK_HIP(hipStreamSynchronize(s));
    return TKMK_SUCCESS;
}

TK_API tkmk_vecops_config tkmk_vecops_default_config(void) {
    tkmk_vecops_config c;
    c.stream_handle = nullptr;
    c.is_a_on_device = false;
    c.is_b_on_device = false;
    c.is_result_on_device = false;
    c.is_async = false;
    c.batch_size = 1;
    c.columns_batch = false;
    c.ext = nullptr;
    return c;
}
TK_API tkmk_error bls12_381_vector_add(const tkmk_fr *a, const tkmk_fr *b, uint64_t n, const tkmk_vecops_config *c, tkmk_fr *o) {
    return vec_entry(OP_ADD, a, b, n, c, o);
}
// accumulate (icicle_core::vec_ops::accumulate_scalars): a[i] += b[i], in place in `a` (is_result_on_device is taken from is_a_on_device)
TK_API tkmk_error bls12_381_vector_accumulate(tkmk_fr *a, const tkmk_fr *b, uint64_t n, const tkmk_vecops_config *c) {
    if (!c) return TKMK_ERR_INVALID_ARGUMENT;
    tkmk_vecops_config cc = *c;
    cc.is_result_on_device = c->is_a_on_device;
    return vec_entry(OP_ADD, a, b, n, &cc, a);
}
TK_API tkmk_error bls12_381_vector_sub(const tkmk_fr *a, const tkmk_fr *b, uint64_t n, const tkmk_vecops_config *c, tkmk_fr *o) {
    return vec_entry(OP_SUB, a, b, n, c, o);
}
TK_API tkmk_error bls12_381_vector_mul(const tkmk_fr *a, const tkmk_fr *b, uint64_t n, const tkmk_vecops_config *c, tkmk_fr *o) {
    return vec_entry(OP_MUL, a, b, n, c, o);
}
TK_API tkmk_error bls12_381_vector_div(const tkmk_fr *a, const tkmk_fr *b, uint64_t n, const tkmk_vecops_config *c, tkmk_fr *o) {
    return vec_entry(OP_DIV, a, b, n, c, o);
}
TK_API tkmk_error bls12_381_vector_inv(const tkmk_fr *a, uint64_t n, const tkmk_vecops_config *c, tkmk_fr *o) {
    return vec_entry(OP_INV, a, nullptr, n, c, o);
}
TK_API tkmk_error bls12_381_scalar_add_vec(const tkmk_fr *a, const tkmk_fr *b, uint64_t n, const tkmk_vecops_config *c, tkmk_fr *o) {
    return vec_entry(OP_SADD, a, b, n, c, o);
}
TK_API tkmk_error bls12_381_scalar_sub_vec(const tkmk_fr *a, const tkmk_fr *b, uint64_t n, const tkmk_vecops_config *c, tkmk_fr *o) {
    return vec_entry(OP_SSUB, a, b, n, c, o);
}
TK_API tkmk_error bls12_381_scalar_mul_vec(const tkmk_fr *a, const tkmk_fr *b, uint64_t n, const tkmk_vecops_config *c, tkmk_fr *o) {
    return vec_entry(OP_SMUL, a, b, n, c, o);
}

static tkmk_error reduce_entry(int op, const tkmk_fr *a, uint64_t n, const tkmk_vecops_config *cfg, tkmk_fr *out) {
    if (!cfg || cfg->ext) return TKMK_ERR_INVALID_ARGUMENT;
    TK_TRY(tk_require_device());
    uint64_t batch = cfg->batch_size > 0 ? cfg->batch_size : 1;
    if (n == 0) return TKMK_ERR_INVALID_ARGUMENT;
    hipStream_t s = tk_stream(cfg->stream_handle);
    tk_frame frame(s);
    tk_staged A, O;
    TK_TRY(A.in(a, n * batch * 32, cfg->is_a_on_device, s));
    TK_TRY(O.out(out, batch * 32, cfg->is_result_on_device, s));
    uint64_t vs = cfg->columns_batch ? 1 : n, es = cfg->columns_batch ? batch : 1;
    unsigned chunks = (unsigned)((n + 4095) / 4096);
    if (chunks > 64) chunks = 64;
    const fr_t *pa = (const fr_t *)A.dev;
    fr_t *po = (fr_t *)O.dev;
    if (chunks == 1) {
        if (op == 0) hipLaunchKernelGGL(k_reduce<0>, dim3(1, (unsigned)batch), 256, 0, s, pa, po, n, vs, es, 1);
        else hipLaunchKernelGGL(k_reduce<1>, dim3(1, (unsigned)batch), 256, 0, s, pa, po, n, vs, es, 1);
    } else {
        tk_scratch part;
        TK_TRY(part.alloc((size_t)chunks * batch * 32, s));
        fr_t *pp = part.as<fr_t>();
        // stage 1: partials (products stay in Montgomery form); stage 2 combines them
        if (op == 0) {
            hipLaunchKernelGGL(k_reduce<0>, dim3(chunks, (unsigned)batch), 256, 0, s, pa, pp, n, vs, es, 1);
            hipLaunchKernelGGL(k_reduce<0>, dim3(1, (unsigned)batch), 256, 0, s, (const fr_t *)pp, po, (uint64_t)chunks, (uint64_t)chunks,
                               (uint64_t)1, 1);
        } else {
            hipLaunchKernelGGL(k_reduce<1>, dim3(chunks, (unsigned)batch), 256, 0, s, pa, pp, n, vs, es, 1);
            hipLaunchKernelGGL(k_reduce<1>, dim3(1, (unsigned)batch), 256, 0, s, (const fr_t *)pp, po, (uint64_t)chunks, (uint64_t)chunks,
                               (uint64_t)1, 0);
        }
    }
    TK_HIP(hipGetLastError());
    TK_TRY(O.copy_back(out, batch * 32, cfg->is_result_on_device, s));
    if (!cfg->is_async || !cfg->is_result_on_device) TK_HIP(hipStreamSynchronize(s));
    return TKMK_SUCCESS;
}
TK_API tkmk_error bls12_381_vector_sum(const tkmk_fr *a, uint64_t n, const tkmk_vecops_config *c, tkmk_fr *o) {
    return reduce_entry(0, a, n, c, o);
}
TK_API tkmk_error bls12_381_vector_product(const tkmk_fr *a, uint64_t n, const tkmk_vecops_config *c, tkmk_fr *o) {
    return reduce_entry(1, a, n, c, o);
}

TK_API tkmk_error bls12_381_matrix_transpose(const tkmk_fr *in, uint32_t rows, uint32_t cols, const tkmk_vecops_config *cfg,
                                             tkmk_fr *out) {
    if (!cfg || cfg->ext) return TKMK_ERR_INVALID_ARGUMENT;
    TK_TRY(tk_require_device());
    uint64_t total = (uint64_t)rows * cols;
    if (total == 0) return TKMK_SUCCESS;
    if (in == out) return TKMK_ERR_INVALID_ARGUMENT;
    hipStream_t s = tk_stream(cfg->stream_handle);
    tk_frame frame(s);
    tk_staged A, O;
    TK_TRY(A.in(in, total * 32, cfg->is_a_on_device, s));
    TK_TRY(O.out(out, total * 32, cfg->is_result_on_device, s));
    hipLaunchKernelGGL(k_transpose, dim3(tk_div_up(cols, 16), tk_div_up(rows, 16)), 256, 0, s, (const fr_t *)A.dev, (fr_t *)O.dev,
                       rows, cols);
    TK_HIP(hipGetLastError());
    TK_TRY(O.copy_back(out, total * 32, cfg->is_result_on_device, s));
    if (!cfg->is_async || !cfg->is_result_on_device) TK_HIP(hipStreamSynchronize(s));
    return TKMK_SUCCESS;
}

// ---------------------------------------------------------------------------------------------------
// Exclusive suffix product: out[i] = prod_{j > i} a[j], out[n-1] = 1 — the copy-constraint running product of
// prove1, which the reference computes with a serial host loop over 2^20 elements
// (packages/backend/prove/src/lib.rs:1858-1862: r[idx] = r[idx+1] * scalers[idx+1]).  Three launches: per-workgroup
// products (16 consecutive elements per lane, LDS suffix scan over the 256 lanes), a scan of the workgroup products,
// and the apply pass.  Accumulators are kept in Montgomery form.
// ---------------------------------------------------------------------------------------------------
#define SP_RUN 16
#define SP_BLOCK (256 * SP_RUN)
// lane-level exclusive suffix scan of Montgomery values over 256 lanes: sh[t] <- prod_{u > t} v[u]; returns the
// inclusive block product in *total
__device__ __forceinline__ fr_t sp_suffix_scan(fr_t v, fr_t *sh, fr_t *total) {
    const uint32_t t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    fr_t incl = v;  // inclusive suffix product
    for (uint32_t off = 1; off < 256; off <<= 1) {
        fr_t other = t + off < 256 ? sh[t + off] : Fr::one();
        __syncthreads();
        incl = Fr::mul(incl, other);
        sh[t] = incl;
        __syncthreads();
    }
    fr_t excl = t + 1 < 256 ? sh[t + 1] : Fr::one();
    *total = sh[0];
    __syncthreads();
    return excl;
}
__global__ __launch_bounds__(256) void k_sp_block_products(const fr_t *__restrict__ a, uint64_t n, fr_t *__restrict__ bprod) {
    __shared__ fr_t sh[256];
    uint64_t base = (uint64_t)blockIdx.x * SP_BLOCK + (uint64_t)threadIdx.x * SP_RUN;
    fr_t p = Fr::one();
    for (int k = 0; k < SP_RUN; k++)
        if (base + k < n) p = Fr::mul(p, Fr::to_mont(Fr::canon(tk_load(a + base + k))));
    fr_t total;
    (void)sp_suffix_scan(p, sh, &total);
    if (threadIdx.x == 0) tk_store(bprod + blockIdx.x, total);
}
// single workgroup: carry[b] = prod_{c > b} bprod[c]  (nb <= 2^24 / 4096; serial chunks of 256)
__global__ __launch_bounds__(256) void k_sp_block_scan(const fr_t *__restrict__ bprod, uint32_t nb, fr_t *__restrict__ carry) {
    __shared__ fr_t sh[256];
    fr_t tail = Fr::one();  // product of all blocks beyond the current chunk
    for (int64_t hi = nb; hi > 0; hi -= 256) {
        int64_t lo = hi - 256 < 0 ? 0 : hi - 256;
        int64_t idx = lo + threadIdx.x;
        fr_t v = idx < hi ? tk_load(bprod + idx) : Fr::one();
        fr_t total;
        fr_t excl = sp_suffix_scan(v, sh, &total);
        if (idx < hi) tk_store(carry + idx, Fr::mul(excl, tail));
        tail = Fr::mul(tail, total);
    }
}
__global__ __launch_bounds__(256) void k_sp_apply(const fr_t *__restrict__ a, uint64_t n, const fr_t *__restrict__ carry,
                                                 fr_t *__restrict__ out) {
    __shared__ fr_t sh[256];
    uint64_t base = (uint64_t)blockIdx.x * SP_BLOCK + (uint64_t)threadIdx.x * SP_RUN;
    fr_t x[SP_RUN];
    fr_t p = Fr::one();
#pragma unroll
    for (int k = 0; k < SP_RUN; k++) {
        x[k] = base + k < n ? Fr::to_mont(Fr::canon(tk_load(a + base + k))) : Fr::one();
        p = Fr::mul(p, x[k]);
    }
    fr_t total;
    fr_t run = Fr::mul(sp_suffix_scan(p, sh, &total), tk_load(carry + blockIdx.x));  // product of everything beyond this lane's run
#pragma unroll
    for (int k = SP_RUN - 1; k >= 0; k--) {
        if (base + k < n) tk_store(out + base + k, Fr::from_mont(run));
        run = Fr::mul(run, x[k]);
    }
}

TK_API tkmk_error tkmk_vec_suffix_product(const tkmk_fr *a_dev, uint64_t n, tkmk_fr *out_dev, tkmk_stream stream) {
    if ((!a_dev || !out_dev) && n) return TKMK_ERR_INVALID_POINTER;
    TK_TRY(tk_require_device());
    if (n == 0) return TKMK_SUCCESS;
    if ((const void *)a_dev == (void *)out_dev) return TKMK_ERR_INVALID_ARGUMENT;
    hipStream_t s = tk_stream(stream);
    tk_frame frame(s);
    uint32_t nb = (uint32_t)((n + SP_BLOCK - 1) / SP_BLOCK);
    tk_scratch bprod, carry;
    TK_TRY(bprod.alloc((size_t)nb * sizeof(fr_t), s));
    TK_TRY(carry.alloc((size_t)nb * sizeof(fr_t), s));
    hipLaunchKernelGGL(k_sp_block_products, nb, 256, 0, s, (const fr_t *)a_dev, n, bprod.as<fr_t>());
    hipLaunchKernelGGL(k_sp_block_scan, 1, 256, 0, s, (const fr_t *)bprod.p, nb, carry.as<fr_t>());
    hipLaunchKernelGGL(k_sp_apply, nb, 256, 0, s, (const fr_t *)a_dev, n, (const fr_t *)carry.p, (fr_t *)out_dev);
    TK_HIP(hipGetLastError());
    return TKMK_SUCCESS;
}
