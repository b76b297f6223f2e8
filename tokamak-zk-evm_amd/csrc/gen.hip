// gen.hip — deterministic device-side input generation and the batched fixed-base
// scalar multiplication that the reference's setup expresses as "n one-point MSMs"
// (packages/backend/libs/src/iotools/mod.rs:1113-1151).  Lets tests / bench.py build 2^24-point inputs in
// HBM from a seed instead of shipping multi-GiB fixtures.
#include "common.h"

__device__ __forceinline__ uint64_t splitmix64_at(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// element i = outputs 4i..4i+3 of the stream as little-endian 64-bit limbs, reduced mod r
template <class Fr>
__global__ __launch_bounds__(256) void k_fr_random(uint64_t seed, uint64_t first, uint64_t n, typename Fr::E *__restrict__ out) {
    using fr_t = typename Fr::E;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fr_t x;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint64_t z = splitmix64_at(seed, 4 * (first + i) + k);
        x.l[2 * k] = (uint32_t)z;
        x.l[2 * k + 1] = (uint32_t)(z >> 32);
    }
    tk_store(out + i, Fr::canon(x));
}

// out[i] = [s_i] P through a 4-bit fixed-window table of P held in LDS (15 multiples, XYZZ)
template <class Fr, class Fq>
__global__ __launch_bounds__(128) void k_g1_batch_scalar_mul(const typename Fr::E *__restrict__ scalars, affine_t<Fq> base, uint64_t n,
                                                            affine_t<Fq> *__restrict__ out) {
    using fr_t = typename Fr::E;
    using G1 = ec<Fq>;
    using g1_affine_t = affine_t<Fq>;
    using g1_xyzz_t = xyzz_t<Fq>;
    __shared__ g1_xyzz_t table[16];
    if (threadIdx.x == 0) {
        table[0] = G1::inf();
        g1_affine_t b = base;
        if (!G1::is_inf(b)) {
            b.x = Fq::to_mont(Fq::canon(b.x));
            b.y = Fq::to_mont(Fq::canon(b.y));
        }
        table[1] = G1::from_affine(b);
        for (int k = 2; k < 16; k++) table[k] = G1::add_mixed(table[k - 1], b);
    }
    __syncthreads();
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fr_t s = Fr::canon(tk_load(scalars + i));
    g1_xyzz_t acc = G1::inf();
    for (int w = 63; w >= 0; w--) {
        for (int d = 0; d < 4; d++) acc = G1::dbl(acc);
        uint32_t nib = (s.l[w >> 3] >> ((w & 7) * 4)) & 15u;
        if (nib) acc = G1::add(acc, table[nib]);
    }
    g1_affine_t r = G1::to_affine(acc);
    r.x = Fq::from_mont(r.x);
    r.y = Fq::from_mont(r.y);
    tk_store(out + i, r);
}

// ---- fixed-base path for large batches (CRS generation: 2^22 .. 2^24 multiples of one generator, Sigma1::gen) ----
// table[j * 256 + d] = [d * 2^(8 j)] P as affine Montgomery points ((0,0) for d = 0), j < ceil(bits / 8): a scalar is then
// one mixed addition per byte instead of 256 doublings + 64 general additions, and the affine conversion shares one inversion
// between FB_K results (Montgomery's trick, prefix products staged in LDS).
constexpr int FB_K = 16;

template <class Fq>
__global__ __launch_bounds__(64) void k_fb_window_bases(affine_t<Fq> base, int windows, affine_t<Fq> *__restrict__ table) {
    using G1 = ec<Fq>;
    int j = threadIdx.x;
    if (j >= windows) return;
    affine_t<Fq> b = base;
    if (!G1::is_inf(b)) {
        b.x = Fq::to_mont(Fq::canon(b.x));
        b.y = Fq::to_mont(Fq::canon(b.y));
    }
    xyzz_t<Fq> acc = G1::from_affine(b);
    for (int k = 0; k < 8 * j; k++) acc = G1::dbl(acc);
    table[j * 256 + 1] = G1::to_affine(acc);   // Montgomery form kept
}
template <class Fq>
__global__ __launch_bounds__(256) void k_fb_table_rows(int windows, affine_t<Fq> *__restrict__ table) {
    using G1 = ec<Fq>;
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= windows * 256) return;
    int j = e >> 8, d = e & 255;
    if (d == 1) return;
    affine_t<Fq> out;
    out.x = Fq::zero();
    out.y = Fq::zero();
    if (d) {
        affine_t<Fq> b = table[j * 256 + 1];
        xyzz_t<Fq> acc = G1::inf();
        for (int bit = 7; bit >= 0; bit--) {
            acc = G1::dbl(acc);
            if ((d >> bit) & 1) acc = G1::add_mixed(acc, b);
        }
        out = G1::to_affine(acc);
    }
    table[e] = out;
}
// rows d = 1 are read by k_fb_table_rows while other lanes write rows d != 1 of the same table: disjoint entries
template <class Fr, class Fq>
__global__ __launch_bounds__(128) void k_fb_accumulate(const typename Fr::E *__restrict__ scalars, const affine_t<Fq> *__restrict__ table,
                                                      int windows, uint64_t n, xyzz_t<Fq> *__restrict__ acc_out) {
    using G1 = ec<Fq>;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    typename Fr::E s = Fr::canon(tk_load(scalars + i));
    xyzz_t<Fq> acc = G1::inf();
    for (int j = 0; j < windows; j++) {
        uint32_t d = (s.l[j >> 2] >> ((j & 3) * 8)) & 255u;
        if (d) acc = G1::add_mixed(acc, table[j * 256 + d]);
    }
    acc_out[i] = acc;
}
template <class Fq>
__global__ __launch_bounds__(64) void k_fb_to_affine(const xyzz_t<Fq> *__restrict__ acc, uint64_t n, affine_t<Fq> *__restrict__ out) {
    using E = typename Fq::E;
    __shared__ E pref[FB_K][64];
    uint64_t first = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * FB_K;
    if (first >= n) return;
    int cnt = (int)((n - first) < (uint64_t)FB_K ? (n - first) : (uint64_t)FB_K);
    E run = Fq::one();
    for (int k = 0; k < cnt; k++) {   // pref[k] = product of the non-zero zzz before k
        pref[k][threadIdx.x] = run;
        E z = acc[first + k].zzz;
        if (!Fq::is_zero(z)) run = Fq::mul(run, z);
    }
    E inv = Fq::inv(run);
    for (int k = cnt - 1; k >= 0; k--) {
        xyzz_t<Fq> p = acc[first + k];
        affine_t<Fq> r;
        if (Fq::is_zero(p.zz)) {
            r.x = Fq::zero();
            r.y = Fq::zero();
        } else {
            E izzz = Fq::mul(inv, pref[k][threadIdx.x]);
            inv = Fq::mul(inv, p.zzz);
            E izz = Fq::sqr(Fq::mul(p.zz, izzz));   // (ZZ / ZZZ)^2 = 1 / ZZ
            r.x = Fq::from_mont(Fq::mul(p.x, izz));
            r.y = Fq::from_mont(Fq::mul(p.y, izzz));
        }
        tk_store(out + first + k, r);
    }
}

template <class Fr>
static tkmk_error fr_random_device(uint64_t seed, uint64_t first, uint64_t n, void *out_dev, tkmk_stream s) {
    if (!out_dev && n) return TKMK_ERR_INVALID_POINTER;
    TK_TRY(tk_require_device());
    if (n == 0) return TKMK_SUCCESS;
    hipLaunchKernelGGL((k_fr_random<Fr>), tk_div_up(n, 256), 256, 0, tk_stream(s), seed, first, n, (typename Fr::E *)out_dev);
    TK_HIP(hipGetLastError());
    TK_HIP(hipStreamSynchronize(tk_stream(s)));
    return TKMK_SUCCESS;
}
TK_API tkmk_error tkmk_fr_random_device(uint64_t seed, uint64_t first, uint64_t n, tkmk_fr *out_dev, tkmk_stream s) {
    return fr_random_device<Fr>(seed, first, n, out_dev, s);
}
TK_API tkmk_error tkmk_bn254_fr_random_device(uint64_t seed, uint64_t first, uint64_t n, tkmk_bn254_fr *out_dev, tkmk_stream s) {
    return fr_random_device<ff<bn254_fr_params>>(seed, first, n, out_dev, s);
}

template <class Fr, class Fq>
static tkmk_error g1_batch_scalar_mul_device(const void *scalars_dev, const uint32_t *base_x, const uint32_t *base_y, uint64_t n,
                                             void *out_dev, tkmk_stream s) {
    if ((!scalars_dev || !out_dev) && n) return TKMK_ERR_INVALID_POINTER;
    TK_TRY(tk_require_device());
    if (n == 0) return TKMK_SUCCESS;
    affine_t<Fq> b;
    for (int i = 0; i < Fq::N; i++) {
        b.x.l[i] = base_x[i];
        b.y.l[i] = base_y[i];
    }
    hipStream_t st = tk_stream(s);
    if (n < 8192) {   // small batches (single CRS points, test vectors): the table would cost more than it saves
        hipLaunchKernelGGL((k_g1_batch_scalar_mul<Fr, Fq>), tk_div_up(n, 128), 128, 0, st, (const typename Fr::E *)scalars_dev, b, n,
                           (affine_t<Fq> *)out_dev);
        TK_HIP(hipGetLastError());
        TK_HIP(hipStreamSynchronize(st));
        return TKMK_SUCCESS;
    }
    constexpr int windows = (int)sizeof(typename Fr::E) / 4 * 4;   // one byte per window over the whole limb array (32 for both curves)
    static_assert(windows <= 64, "k_fb_window_bases runs one lane per window");
    const uint64_t tile = (uint64_t)1 << 22;                        // 768 MiB of XYZZ scratch per tile
    tk_frame frame(st);
    tk_scratch tbl, acc;
    TK_TRY(tbl.alloc((size_t)windows * 256 * sizeof(affine_t<Fq>), st));
    TK_TRY(acc.alloc((size_t)(n < tile ? n : tile) * sizeof(xyzz_t<Fq>), st));
    hipLaunchKernelGGL((k_fb_window_bases<Fq>), 1, 64, 0, st, b, windows, tbl.as<affine_t<Fq>>());
    hipLaunchKernelGGL((k_fb_table_rows<Fq>), tk_div_up((uint64_t)windows * 256, 256), 256, 0, st, windows, tbl.as<affine_t<Fq>>());
    for (uint64_t off = 0; off < n; off += tile) {
        uint64_t m = n - off < tile ? n - off : tile;
        hipLaunchKernelGGL((k_fb_accumulate<Fr, Fq>), tk_div_up(m, 128), 128, 0, st, (const typename Fr::E *)scalars_dev + off,
                           (const affine_t<Fq> *)tbl.p, windows, m, acc.as<xyzz_t<Fq>>());
        hipLaunchKernelGGL((k_fb_to_affine<Fq>), tk_div_up(tk_div_up(m, FB_K), 64), 64, 0, st, (const xyzz_t<Fq> *)acc.p, m,
                           (affine_t<Fq> *)out_dev + off);
    }
    TK_HIP(hipGetLastError());
    TK_HIP(hipStreamSynchronize(st));
    return TKMK_SUCCESS;
}
TK_API tkmk_error tkmk_g1_batch_scalar_mul_device(const tkmk_fr *scalars_dev, const tkmk_g1_affine *base_host, uint64_t n,
                                                  tkmk_g1_affine *out_dev, tkmk_stream s) {
    if (!base_host) return TKMK_ERR_INVALID_POINTER;
    return g1_batch_scalar_mul_device<Fr, Fq>(scalars_dev, base_host->x.limbs, base_host->y.limbs, n, out_dev, s);
}
TK_API tkmk_error tkmk_bn254_g1_batch_scalar_mul_device(const tkmk_bn254_fr *scalars_dev, const tkmk_bn254_g1_affine *base_host,
                                                        uint64_t n, tkmk_bn254_g1_affine *out_dev, tkmk_stream s) {
    if (!base_host) return TKMK_ERR_INVALID_POINTER;
    return g1_batch_scalar_mul_device<ff<bn254_fr_params>, ff<bn254_fq_params>>(scalars_dev, base_host->x.limbs, base_host->y.limbs, n,
                                                                               out_dev, s);
}

// G1Affine::generate_random (icicle_core::traits::GenerateRandom, used by the reference's tests): n points [k_i]G with fresh uniform
// k_i, G the standard generator; HOST output.  The multiples are computed on the device (fixed-base path above).
extern "C" tkmk_error bls12_381_generate_scalars(tkmk_fr *out_host, size_t n);
TK_API tkmk_error bls12_381_generate_random_affine_points(tkmk_g1_affine *out_host, size_t n) {
    if (!out_host && n) return TKMK_ERR_INVALID_POINTER;
    if (n == 0) return TKMK_SUCCESS;
    TK_TRY(tk_require_device());
    static const uint32_t GX[12] = {0xdb22c6bbu, 0xfb3af00au, 0xf97a1aefu, 0x6c55e83fu, 0x171bac58u, 0xa14e3a3fu,
                                    0x9774b905u, 0xc3688c4fu, 0x4fa9ac0fu, 0x2695638cu, 0x3197d794u, 0x17f1d3a7u};   // setup/mpc-setup/src/conversions.rs:68-79
    static const uint32_t GY[12] = {0x46c5e7e1u, 0x0caa2329u, 0xa2888ae4u, 0xd03cc744u, 0x2c04b3edu, 0x00db18cbu,
                                    0xd5d00af6u, 0xfcf5e095u, 0x741d8ae4u, 0xa09e30edu, 0xe3aaa0f1u, 0x08b3f481u};
    std::vector<tkmk_fr> k(n);
    TK_TRY(bls12_381_generate_scalars(k.data(), n));
    void *d_k = nullptr, *d_out = nullptr;
    TK_HIP(hipMalloc(&d_k, n * sizeof(tkmk_fr)));
    hipError_t e = hipMalloc(&d_out, n * sizeof(tkmk_g1_affine));
    if (e != hipSuccess) {
        (void)hipFree(d_k);
        return tk_map_hip_error(e);
    }
    tkmk_error r = tk_map_hip_error(hipMemcpy(d_k, k.data(), n * sizeof(tkmk_fr), hipMemcpyHostToDevice));
    if (r == TKMK_SUCCESS) r = g1_batch_scalar_mul_device<Fr, Fq>((const tkmk_fr *)d_k, GX, GY, n, (tkmk_g1_affine *)d_out, nullptr);
    if (r == TKMK_SUCCESS) r = tk_map_hip_error(hipMemcpy(out_host, d_out, n * sizeof(tkmk_g1_affine), hipMemcpyDeviceToHost));
    (void)hipFree(d_k), (void)hipFree(d_out);
    return r;
}

// dst[i] = src[idx[i]] for rows of row_bytes (multiple of 16): the device-side gather behind the binding
// commitments, which the reference builds on the host by walking nested CRS tables
// (packages/backend/libs/src/group_structures/mod.rs:145-300 encode_*_common -> msm_g1_bases :127-143)
__global__ __launch_bounds__(256) void k_gather_rows(const uint4 *__restrict__ src, const uint32_t *__restrict__ idx, uint64_t n,
                                                    uint32_t vec_per_row, uint4 *__restrict__ dst) {
    uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * vec_per_row) return;
    uint64_t i = e / vec_per_row;
    uint32_t k = (uint32_t)(e - i * vec_per_row);
    dst[e] = src[(uint64_t)idx[i] * vec_per_row + k];
}
TK_API tkmk_error tkmk_gather_rows_device(const void *src_dev, uint32_t row_bytes, const uint32_t *idx_dev, uint64_t n, void *dst_dev,
                                          tkmk_stream s) {
    if ((!src_dev || !idx_dev || !dst_dev) && n) return TKMK_ERR_INVALID_POINTER;
    if (row_bytes == 0 || row_bytes % 16) return TKMK_ERR_INVALID_ARGUMENT;
    TK_TRY(tk_require_device());
    if (n == 0) return TKMK_SUCCESS;
    hipLaunchKernelGGL(k_gather_rows, tk_div_up(n * (row_bytes / 16), 256), 256, 0, tk_stream(s), (const uint4 *)src_dev, idx_dev, n,
                       row_bytes / 16, (uint4 *)dst_dev);
    TK_HIP(hipGetLastError());
    return TKMK_SUCCESS;
}
