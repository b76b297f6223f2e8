/*
 * tk_oracle.h — CPU ORACLE (test infrastructure, NOT the product).
 *
 * A plain-C restatement of the arithmetic the Tokamak zk-EVM prover's polynomial-commitment
 * hot path performs through ICICLE v3.8.0 (git tag pinned at packages/backend/Cargo.toml:20-23;
 * ICICLE itself is NOT in /root/reference, so its published algorithms are restated here and
 * parity is anchored on the reference's own call sites and tests):
 *
 *   - BLS12-381 scalar field Fr / base field Fq arithmetic
 *   - G1 group law, scalar multiplication, MSM         (call sites: libs/src/iotools/mod.rs:2093-2099,
 *                                                        libs/src/group_structures/mod.rs:108-114,127-143)
 *   - radix-2 NTT / iNTT with ICICLE's batch / columns_batch / coset semantics, natural order in & out
 *                                                       (call sites: libs/src/bivariate_polynomial/mod.rs:1422-1478)
 *   - the 2-D "_biNTT" (rows of length y_size, then strided columns of length x_size)
 *   - element-wise vector ops and transpose            (libs/src/vector_operations/mod.rs)
 *
 * PARITY STATUS: the reference holds NO known-answer vectors for raw MSM / NTT outputs
 * (libs/src/tests.rs is self-consistency only) => raw outputs are "parity unpinned".
 * What IS pinned, and checked by tests/test_oracle_pins.py:
 *   - r (scalar modulus) = prime in every committed .r1cs header (qap-compiler/subcircuits/library/r1cs)
 *   - in-memory encoding = little-endian u32 limbs of the plain (non-Montgomery) integer, and the
 *     standard G1 generator limbs            (setup/mpc-setup/src/conversions.rs:43-95)
 *   - fixed-tau G1 generator is on y^2 = x^3 + 4 (setup/trusted-setup/src/main.rs:68-80)
 *   - NTT ordering / coset semantics           (libs/src/tests.rs:107-180, 1075-1087)
 *   - commit identity encode_poly(P) = [P(tau_x, tau_y)]G   (setup/trusted-setup/src/main.rs:236-246)
 *   - root of unity: w_{2^32} = 5^((r-1)/2^32) (ICICLE's / ffjavascript's convention; inferred — SURVEY.md §8c)
 * plus an independent Python big-int restatement used on small cases.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * All API buffers hold PLAIN little-endian integers: Fr = 32 bytes, Fq = 48 bytes,
 * G1 affine = 96 bytes {x, y}, point at infinity = all-zero (libs/src/iotools/mod.rs:1785-1816).
 */
#ifndef TK_ORACLE_H
#define TK_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- fields (n elements, plain LE) ---- */
void tko_fr_add(const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n);
void tko_fr_sub(const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n);
void tko_fr_mul(const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n);
void tko_fr_inv(const uint8_t *a, uint8_t *out, size_t n);           /* inv(0) = 0 */
void tko_fr_scalar_mul(const uint8_t *s, const uint8_t *a, uint8_t *out, size_t n);
void tko_fr_scalar_add(const uint8_t *s, const uint8_t *a, uint8_t *out, size_t n);
void tko_fr_scalar_sub(const uint8_t *s, const uint8_t *a, uint8_t *out, size_t n); /* s - a[i] (ICICLE v3 scalar_sub_vec) */
void tko_fr_pow_u64(const uint8_t *a, uint64_t e, uint8_t *out);
void tko_fr_transpose(const uint8_t *in, size_t rows, size_t cols, uint8_t *out);
void tko_fr_suffix_product(const uint8_t *s, size_t n, uint8_t *out); /* prove/src/lib.rs:1858-1862 */
void tko_fq_mul(const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n);
void tko_fq_add(const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n);
void tko_fq_sub(const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n);
void tko_fq_inv(const uint8_t *a, uint8_t *out, size_t n);

/* deterministic inputs: splitmix64(seed) stream -> 256-bit -> mod r */
void tko_fr_random(uint64_t seed, size_t first, size_t n, uint8_t *out);

/* ---- NTT ---- */
/* w_N for the smallest power of two N >= max_size (ICICLE ntt::get_root_of_unity) */
int tko_get_root_of_unity(uint64_t max_size, uint8_t *out);
/* ICICLE ntt::ntt semantics: `batch` vectors of length n; columns_batch=0 -> vector b occupies
 * [b*n, (b+1)*n); columns_batch=1 -> element i of vector b is at i*batch + b.
 * inverse=0: out[k] = sum_j (g^j in[j]) w^{jk};  inverse=1: the exact inverse of that map.
 * coset_gen may be NULL (= 1). in may alias out. Returns 0 on success. */
int tko_ntt(const uint8_t *in, size_t n, size_t batch, int columns_batch, int inverse,
            const uint8_t *coset_gen, uint8_t *out);
/* _biNTT (libs/src/bivariate_polynomial/mod.rs:1422-1478): element (ix,iy) at ix*y_size+iy */
int tko_bintt(const uint8_t *in, size_t x_size, size_t y_size, int inverse,
              const uint8_t *coset_x, const uint8_t *coset_y, uint8_t *out);
/* O(n^2) definition, single vector, forward only, no coset — pins tko_ntt on small n */
int tko_dft_naive(const uint8_t *in, size_t n, uint8_t *out);

/* ---- G1 ---- */
void tko_g1_generator(uint8_t *out96);
int  tko_g1_on_curve(const uint8_t *p96);                 /* infinity counts as on curve */
void tko_g1_add(const uint8_t *p96, const uint8_t *q96, uint8_t *out96);
void tko_g1_neg(const uint8_t *p96, uint8_t *out96);
void tko_g1_scalar_mul(const uint8_t *s32, const uint8_t *p96, uint8_t *out96);
/* out[i] = [s_i] P (n results), threads */
void tko_g1_batch_scalar_mul(const uint8_t *s, const uint8_t *p96, size_t n, uint8_t *out);
/* bases P_i = [h_i]G, h_i = tko_fr_random(seed) stream */
void tko_g1_random_bases(uint64_t seed, size_t first, size_t n, uint8_t *out);
/* sum_i [s_i] P_i by n independent double-and-add scalar multiplications (slow, trusted) */
void tko_g1_msm_naive(const uint8_t *s, const uint8_t *p, size_t n, uint8_t *out96);
/* bucketed Pippenger, OpenMP over windows; threads<=0 -> all cores. Same result as _naive. */
void tko_g1_msm(const uint8_t *s, const uint8_t *p, size_t n, int threads, uint8_t *out96);
/* homogeneous projective (X/Z, Y/Z) -> affine, 144 B -> 96 B (ICICLE G1Projective -> G1Affine) */
void tko_g1_proj_to_affine(const uint8_t *p144, uint8_t *out96);

/* ---- BN254 (alt_bn128) instantiation of the same field / G1 code (tk_g1.inc): Fr, Fq 32 B, affine 64 B, projective 96 B.
 * Not used by the reference (BLS12-381 only); named by BASELINE.json's MSM configs. ---- */
void tko_bn254_fr_add(const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n);
void tko_bn254_fr_sub(const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n);
void tko_bn254_fr_mul(const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n);
void tko_bn254_fr_inv(const uint8_t *a, uint8_t *out, size_t n);
void tko_bn254_fq_add(const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n);
void tko_bn254_fq_sub(const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n);
void tko_bn254_fq_mul(const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n);
void tko_bn254_fq_inv(const uint8_t *a, uint8_t *out, size_t n);
void tko_bn254_fr_random(uint64_t seed, size_t first, size_t n, uint8_t *out);
/* NTT over the BN254 scalar field: same definitions as tko_ntt / tko_bintt / tko_dft_naive; w_{2^28} = 5^((r-1)/2^28) */
int tko_bn254_get_root_of_unity(uint64_t max_size, uint8_t *out);
int tko_bn254_ntt(const uint8_t *in, size_t n, size_t batch, int columns_batch, int inverse, const uint8_t *coset_gen, uint8_t *out);
int tko_bn254_bintt(const uint8_t *in, size_t x_size, size_t y_size, int inverse, const uint8_t *coset_x, const uint8_t *coset_y,
                    uint8_t *out);
int tko_bn254_dft_naive(const uint8_t *in, size_t n, uint8_t *out);
void tko_bn254_g1_generator(uint8_t *out64);
int  tko_bn254_g1_on_curve(const uint8_t *p64);
void tko_bn254_g1_add(const uint8_t *p64, const uint8_t *q64, uint8_t *out64);
void tko_bn254_g1_neg(const uint8_t *p64, uint8_t *out64);
void tko_bn254_g1_scalar_mul(const uint8_t *s32, const uint8_t *p64, uint8_t *out64);
void tko_bn254_g1_batch_scalar_mul(const uint8_t *s, const uint8_t *p64, size_t n, uint8_t *out);
void tko_bn254_g1_random_bases(uint64_t seed, size_t first, size_t n, uint8_t *out);
void tko_bn254_g1_msm_naive(const uint8_t *s, const uint8_t *p, size_t n, uint8_t *out64);
void tko_bn254_g1_msm(const uint8_t *s, const uint8_t *p, size_t n, int threads, uint8_t *out64);
void tko_bn254_g1_proj_to_affine(const uint8_t *p96, uint8_t *out64);

/* ---- bivariate coefficient-matrix routines (DensePolynomialExt host loops; element (ix,iy) at ix*ys+iy) ---- */
void tko_poly_find_degree(const uint8_t *c, size_t xs, size_t ys, int64_t *xd, int64_t *yd);
void tko_poly_resized_dims(size_t tx, size_t ty, size_t *nx, size_t *ny);
void tko_poly_resize(const uint8_t *c, size_t xs, size_t ys, size_t nx, size_t ny, uint8_t *out);
int tko_poly_mul_monomial(const uint8_t *c, size_t xs, size_t ys, size_t ex, size_t ey, size_t nx, size_t ny, uint8_t *out);
void tko_poly_scale_coeffs(const uint8_t *c, size_t xs, size_t ys, const uint8_t *fx, const uint8_t *fy, uint8_t *out);
void tko_poly_eval(const uint8_t *c, size_t xs, size_t ys, const uint8_t *x, const uint8_t *y, uint8_t *out);
void tko_poly_eval_x(const uint8_t *c, size_t xs, size_t ys, const uint8_t *x, uint8_t *out);
void tko_poly_eval_y(const uint8_t *c, size_t xs, size_t ys, const uint8_t *y, uint8_t *out);
int tko_poly_div_by_vanishing_opt(const uint8_t *p, size_t xs, size_t ys, size_t c, size_t d, uint8_t *quo_x, uint8_t *quo_y);
void tko_poly_div_by_ruffini(const uint8_t *p, size_t xs, size_t ys, const uint8_t *x, const uint8_t *y, uint8_t *q_x,
                             uint8_t *q_y, uint8_t *r);

/* eval_sparse_rows (libs/src/iotools/mod.rs:1590-1608) on CSR input, one placement */
void tko_r1cs_eval_rows(const uint32_t *row_ptr, const uint32_t *wire, const uint8_t *coeff, size_t n_rows,
                        const uint8_t *variables, uint8_t *out, size_t out_len);

int tko_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
