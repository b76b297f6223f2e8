/*
 * tk_oracle.c — CPU ORACLE (test infrastructure, NOT the product). See tk_oracle.h for the
 * scope, the reference call sites each routine restates and the parity status.
 *
 * Build: gcc -O2 -fopenmp -fPIC -shared (oracle/Makefile).
 */
#include "tk_oracle.h"
#include "../include/tkmk.h" /* only for TKMK_BLS12_381_FR_ROOT_GENERATOR: the one declared convention both sides must share */
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
typedef uint64_t u64;

/* ------------------------------------------------------------------------------------------
 * Generic Montgomery field over N 64-bit limbs, instantiated for Fr (N=4) and Fq (N=6).
 * Constants other than the modulus are derived at first use, so the only hard-coded numbers
 * are r and p themselves (r is pinned by the .r1cs headers; p by the on-curve pins).
 * ------------------------------------------------------------------------------------------ */
#define DEFINE_FIELD(F, N)                                                                     \
    typedef struct { u64 l[N]; } F##_t;                                                        \
    static F##_t F##_P, F##_R1, F##_R2;                                                        \
    static u64 F##_INV;                                                                        \
    static int F##_ready = 0;                                                                  \
    static inline int F##_geq_p(const F##_t *a) {                                              \
        for (int i = N - 1; i >= 0; i--) {                                                     \
            if (a->l[i] > F##_P.l[i]) return 1;                                                \
            if (a->l[i] < F##_P.l[i]) return 0;                                                \
        }                                                                                      \
        return 1;                                                                              \
    }                                                                                          \
    static inline u64 F##_sub_p(F##_t *a) {                                                    \
        u64 br = 0;                                                                            \
        for (int i = 0; i < N; i++) {                                                          \
            u128 d = (u128)a->l[i] - F##_P.l[i] - br;                                          \
            a->l[i] = (u64)d;                                                                  \
            br = (u64)(d >> 64) & 1;                                                           \
        }                                                                                      \
        return br;                                                                             \
    }                                                                                          \
    static inline void F##_add(F##_t *o, const F##_t *a, const F##_t *b) {                     \
        u64 c = 0;                                                                             \
        F##_t t;                                                                               \
        for (int i = 0; i < N; i++) {                                                          \
            u128 s = (u128)a->l[i] + b->l[i] + c;                                              \
            t.l[i] = (u64)s;                                                                   \
            c = (u64)(s >> 64);                                                                \
        }                                                                                      \
        if (c || F##_geq_p(&t)) F##_sub_p(&t);                                                 \
        *o = t;                                                                                \
    }                                                                                          \
    static inline void F##_sub(F##_t *o, const F##_t *a, const F##_t *b) {                     \
        u64 br = 0;                                                                            \
        F##_t t;                                                                               \
        for (int i = 0; i < N; i++) {                                                          \
            u128 d = (u128)a->l[i] - b->l[i] - br;                                             \
            t.l[i] = (u64)d;                                                                   \
            br = (u64)(d >> 64) & 1;                                                           \
        }                                                                                      \
        if (br) {                                                                              \
            u64 c = 0;                                                                         \
            for (int i = 0; i < N; i++) {                                                      \
                u128 s = (u128)t.l[i] + F##_P.l[i] + c;                                        \
                t.l[i] = (u64)s;                                                               \
                c = (u64)(s >> 64);                                                            \
            }                                                                                  \
        }                                                                                      \
        *o = t;                                                                                \
    }                                                                                          \
    static inline void F##_neg(F##_t *o, const F##_t *a) {                                     \
        F##_t z;                                                                               \
        memset(&z, 0, sizeof z);                                                               \
        F##_sub(o, &z, a);                                                                     \
    }                                                                                          \
    static inline int F##_is_zero(const F##_t *a) {                                            \
        u64 x = 0;                                                                             \
        for (int i = 0; i < N; i++) x |= a->l[i];                                              \
        return x == 0;                                                                         \
    }                                                                                          \
    static inline int F##_eq(const F##_t *a, const F##_t *b) {                                 \
        u64 x = 0;                                                                             \
        for (int i = 0; i < N; i++) x |= a->l[i] ^ b->l[i];                                    \
        return x == 0;                                                                         \
    }                                                                                          \
    /* CIOS Montgomery product: o = a*b/2^(64N) mod p */                                       \
    static inline void F##_mul(F##_t *o, const F##_t *a, const F##_t *b) {                     \
        u64 t[N + 2];                                                                          \
        memset(t, 0, sizeof t);                                                                \
        for (int i = 0; i < N; i++) {                                                          \
            u64 c = 0;                                                                         \
            for (int j = 0; j < N; j++) {                                                      \
                u128 s = (u128)a->l[j] * b->l[i] + t[j] + c;                                   \
                t[j] = (u64)s;                                                                 \
                c = (u64)(s >> 64);                                                            \
            }                                                                                  \
            u128 s = (u128)t[N] + c;                                                           \
            t[N] = (u64)s;                                                                     \
            t[N + 1] = (u64)(s >> 64);                                                         \
            u64 m = t[0] * F##_INV;                                                            \
            s = (u128)m * F##_P.l[0] + t[0];                                                   \
            c = (u64)(s >> 64);                                                                \
            for (int j = 1; j < N; j++) {                                                      \
                s = (u128)m * F##_P.l[j] + t[j] + c;                                           \
                t[j - 1] = (u64)s;                                                             \
                c = (u64)(s >> 64);                                                            \
            }                                                                                  \
            s = (u128)t[N] + c;                                                                \
            t[N - 1] = (u64)s;                                                                 \
            t[N] = t[N + 1] + (u64)(s >> 64);                                                  \
        }                                                                                      \
        F##_t r;                                                                               \
        memcpy(r.l, t, sizeof r.l);                                                            \
        if (t[N] || F##_geq_p(&r)) F##_sub_p(&r);                                              \
        *o = r;                                                                                \
    }                                                                                          \
    static inline void F##_sqr(F##_t *o, const F##_t *a) { F##_mul(o, a, a); }                 \
    static void F##_init(const u64 *p) {                                                       \
        memcpy(F##_P.l, p, sizeof F##_P.l);                                                    \
        u64 inv = 1;                                                                           \
        for (int i = 0; i < 6; i++) inv *= 2 - p[0] * inv; /* p^-1 mod 2^64 (Newton) */        \
        F##_INV = (u64)0 - inv;                                                                \
        /* R mod p by 64N doublings of 1, R^2 mod p by 64N more */                             \
        F##_t x;                                                                               \
        memset(&x, 0, sizeof x);                                                               \
        x.l[0] = 1;                                                                            \
        for (int i = 0; i < 64 * N; i++) F##_add(&x, &x, &x);                                  \
        F##_R1 = x;                                                                            \
        for (int i = 0; i < 64 * N; i++) F##_add(&x, &x, &x);                                  \
        F##_R2 = x;                                                                            \
        F##_ready = 1;                                                                         \
    }                                                                                          \
    static inline void F##_to_mont(F##_t *o, const F##_t *a) { F##_mul(o, a, &F##_R2); }       \
    static inline void F##_from_mont(F##_t *o, const F##_t *a) {                               \
        F##_t one;                                                                             \
        memset(&one, 0, sizeof one);                                                           \
        one.l[0] = 1;                                                                          \
        F##_mul(o, a, &one);                                                                   \
    }                                                                                          \
    /* plain LE bytes (value reduced mod p if needed) -> Montgomery */                         \
    static inline void F##_load(F##_t *o, const uint8_t *b) {                                  \
        F##_t t;                                                                               \
        memcpy(t.l, b, 8 * N);                                                                 \
        while (F##_geq_p(&t)) F##_sub_p(&t);                                                   \
        F##_to_mont(o, &t);                                                                    \
    }                                                                                          \
    static inline void F##_store(uint8_t *b, const F##_t *a) {                                 \
        F##_t t;                                                                               \
        F##_from_mont(&t, a);                                                                  \
        memcpy(b, t.l, 8 * N);                                                                 \
    }                                                                                          \
    /* o = a^e, e given as nl 64-bit limbs (plain) */                                          \
    static void F##_pow(F##_t *o, const F##_t *a, const u64 *e, int nl) {                      \
        F##_t acc = F##_R1, base = *a;                                                         \
        for (int i = 0; i < nl; i++)                                                           \
            for (int b = 0; b < 64; b++) {                                                     \
                if ((e[i] >> b) & 1) F##_mul(&acc, &acc, &base);                               \
                F##_sqr(&base, &base);                                                         \
            }                                                                                  \
        *o = acc;                                                                              \
    }                                                                                          \
    /* Fermat inverse, inv(0)=0 */                                                             \
    static void F##_inv(F##_t *o, const F##_t *a) {                                            \
        u64 e[N];                                                                              \
        memcpy(e, F##_P.l, sizeof e);                                                          \
        e[0] -= 2; /* p is odd and > 2: no borrow */                                           \
        F##_pow(o, a, e, N);                                                                   \
    }

DEFINE_FIELD(fr, 4)
DEFINE_FIELD(fq, 6)
DEFINE_FIELD(bnr, 4)
DEFINE_FIELD(bnq, 4)

/* r: prime field of every committed .r1cs (qap-compiler/subcircuits/library/r1cs, header prime) */
static const u64 FR_MOD[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL,
                              0x73eda753299d7d48ULL};
/* p: BLS12-381 base field */
static const u64 FQ_MOD[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                              0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};

/* BN254 scalar and base field (EIP-196 / alt_bn128) */
static const u64 BNR_MOD[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const u64 BNQ_MOD[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};

static fq_t g1_CURVE_B;  /* curve constant 4 (Montgomery) */
static bnq_t bn_CURVE_B; /* curve constant 3 (Montgomery) */
static fr_t FR_ROOT32;   /* w_{2^32} = g^((r-1)/2^32) (Montgomery), g = TKMK_BLS12_381_FR_ROOT_GENERATOR (include/tkmk.h) */
static bnr_t BNR_ROOT28; /* BN254: w_{2^28} = 5^((r-1)/2^28) (Montgomery) */

static void g2_init(void);
static void tko_init(void) {
    if (fr_ready && fq_ready) return;
#pragma omp critical(tko_init_lock)
    {
        if (!(fr_ready && fq_ready)) {
            fq_init(FQ_MOD);
            fq_t four;
            memset(&four, 0, sizeof four);
            four.l[0] = 4;
            fq_to_mont(&g1_CURVE_B, &four);
            g2_init();
            bnq_init(BNQ_MOD);
            bnr_init(BNR_MOD);
            bnq_t three;
            memset(&three, 0, sizeof three);
            three.l[0] = 3;
            bnq_to_mont(&bn_CURVE_B, &three);
            {
                bnr_t five_b;
                memset(&five_b, 0, sizeof five_b);
                five_b.l[0] = 5;
                bnr_to_mont(&five_b, &five_b);
                u64 eb[4];
                for (int i = 0; i < 4; i++) eb[i] = BNR_MOD[i];
                eb[0] -= 1; /* r - 1, then >> 28 */
                for (int i = 0; i < 4; i++) eb[i] = (eb[i] >> 28) | (i < 3 ? eb[i + 1] << 36 : 0);
                bnr_pow(&BNR_ROOT28, &five_b, eb, 4);
            }
            /* root of unity before publishing fr_ready */
            memcpy(fr_P.l, FR_MOD, sizeof fr_P.l);
            fr_init(FR_MOD);
            /* the generator of the two-adic subgroup is the ONE declared convention of include/tkmk.h (an inference, not a pin:
             * see the note there); the same environment variable as in the product selects another non-residue for a process */
            fr_t five;
            memset(&five, 0, sizeof five);
            five.l[0] = TKMK_BLS12_381_FR_ROOT_GENERATOR;
            {
                const char *env = getenv("TKMK_FR_ROOT_GENERATOR");
                int v = env ? atoi(env) : 0;
                if (v >= 2 && v < 65536) five.l[0] = (u64)v;
            }
            fr_to_mont(&five, &five);
            /* (r-1) >> 32 */
            u64 e[4];
            for (int i = 0; i < 4; i++) e[i] = FR_MOD[i];
            e[0] -= 1;
            for (int i = 0; i < 4; i++) e[i] = (e[i] >> 32) | (i < 3 ? e[i + 1] << 32 : 0);
            fr_pow(&FR_ROOT32, &five, e, 4);
        }
    }
}

int tko_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
/* the parallel regions of the calling thread use at most n threads from here on (OpenMP's default counts every hardware thread of the
 * host; a process held to a CPU quota is throttled for a scheduler period each time that many threads wake up: oracle/__init__.py) */
int tko_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------------
 * Element-wise Fr / Fq (ICICLE VecOps as used by libs/src/vector_operations/mod.rs:34-139)
 * ------------------------------------------------------------------------------------------ */
#define VEC_BINOP(NAME, F, SZ, OP)                                                             \
    void NAME(const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n) {                    \
        tko_init();                                                                            \
        _Pragma("omp parallel for schedule(static)") for (size_t i = 0; i < n; i++) {          \
            F##_t x, y, z;                                                                     \
            F##_load(&x, a + SZ * i);                                                          \
            F##_load(&y, b + SZ * i);                                                          \
            OP(&z, &x, &y);                                                                    \
            F##_store(out + SZ * i, &z);                                                       \
        }                                                                                      \
    }
VEC_BINOP(tko_fr_add, fr, 32, fr_add)
VEC_BINOP(tko_fr_sub, fr, 32, fr_sub)
VEC_BINOP(tko_fr_mul, fr, 32, fr_mul)
VEC_BINOP(tko_fq_add, fq, 48, fq_add)
VEC_BINOP(tko_fq_sub, fq, 48, fq_sub)
VEC_BINOP(tko_fq_mul, fq, 48, fq_mul)
VEC_BINOP(tko_bn254_fr_add, bnr, 32, bnr_add)
VEC_BINOP(tko_bn254_fr_sub, bnr, 32, bnr_sub)
VEC_BINOP(tko_bn254_fr_mul, bnr, 32, bnr_mul)
VEC_BINOP(tko_bn254_fq_add, bnq, 32, bnq_add)
VEC_BINOP(tko_bn254_fq_sub, bnq, 32, bnq_sub)
VEC_BINOP(tko_bn254_fq_mul, bnq, 32, bnq_mul)
#define VEC_INV(NAME, F, SZ)                                                                   \
    void NAME(const uint8_t *a, uint8_t *out, size_t n) {                                      \
        tko_init();                                                                            \
        _Pragma("omp parallel for schedule(static)") for (size_t i = 0; i < n; i++) {          \
            F##_t x, y;                                                                        \
            F##_load(&x, a + SZ * i);                                                          \
            F##_inv(&y, &x);                                                                   \
            F##_store(out + SZ * i, &y);                                                       \
        }                                                                                      \
    }
VEC_INV(tko_bn254_fr_inv, bnr, 32)
VEC_INV(tko_bn254_fq_inv, bnq, 32)

void tko_fr_inv(const uint8_t *a, uint8_t *out, size_t n) {
    tko_init();
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        fr_t x, y;
        fr_load(&x, a + 32 * i);
        fr_inv(&y, &x);
        fr_store(out + 32 * i, &y);
    }
}
void tko_fq_inv(const uint8_t *a, uint8_t *out, size_t n) {
    tko_init();
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        fq_t x, y;
        fq_load(&x, a + 48 * i);
        fq_inv(&y, &x);
        fq_store(out + 48 * i, &y);
    }
}
void tko_fr_scalar_mul(const uint8_t *s, const uint8_t *a, uint8_t *out, size_t n) {
    tko_init();
    fr_t k;
    fr_load(&k, s);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        fr_t x;
        fr_load(&x, a + 32 * i);
        fr_mul(&x, &x, &k);
        fr_store(out + 32 * i, &x);
    }
}
void tko_fr_scalar_add(const uint8_t *s, const uint8_t *a, uint8_t *out, size_t n) {
    tko_init();
    fr_t k;
    fr_load(&k, s);
    for (size_t i = 0; i < n; i++) {
        fr_t x;
        fr_load(&x, a + 32 * i);
        fr_add(&x, &x, &k);
        fr_store(out + 32 * i, &x);
    }
}
void tko_fr_scalar_sub(const uint8_t *s, const uint8_t *a, uint8_t *out, size_t n) {
    tko_init();
    fr_t k;
    fr_load(&k, s);
    for (size_t i = 0; i < n; i++) {
        fr_t x;
        fr_load(&x, a + 32 * i);
        fr_sub(&x, &k, &x); /* ICICLE v3 scalar_sub_vec: res[i] = scalar - vec[i] */
        fr_store(out + 32 * i, &x);
    }
}
void tko_fr_pow_u64(const uint8_t *a, uint64_t e, uint8_t *out) {
    tko_init();
    fr_t x;
    fr_load(&x, a);
    fr_pow(&x, &x, &e, 1);
    fr_store(out, &x);
}
void tko_fr_transpose(const uint8_t *in, size_t rows, size_t cols, uint8_t *out) {
    for (size_t i = 0; i < rows; i++)
        for (size_t j = 0; j < cols; j++) memcpy(out + 32 * (j * rows + i), in + 32 * (i * cols + j), 32);
}

/* splitmix64 stream: element i consumes outputs 4i..4i+3 (little-endian limbs), reduced mod r */
static inline u64 splitmix64_at(u64 seed, u64 idx) {
    u64 z = seed + (idx + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
#define DEFINE_RANDOM(F)                                                                       \
    static inline void F##_random_plain(u64 seed, u64 i, F##_t *o) {                           \
        for (int k = 0; k < 4; k++) o->l[k] = splitmix64_at(seed, 4 * i + k);                  \
        while (F##_geq_p(o)) F##_sub_p(o);                                                     \
    }
DEFINE_RANDOM(fr)
DEFINE_RANDOM(bnr)
void tko_bn254_fr_random(uint64_t seed, size_t first, size_t n, uint8_t *out) {
    tko_init();
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        bnr_t x;
        bnr_random_plain(seed, first + i, &x);
        memcpy(out + 32 * i, x.l, 32);
    }
}
void tko_fr_random(uint64_t seed, size_t first, size_t n, uint8_t *out) {
    tko_init();
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        fr_t x;
        fr_random_plain(seed, first + i, &x);
        memcpy(out + 32 * i, x.l, 32);
    }
}

/* ------------------------------------------------------------------------------------------
 * NTT. Definition restated from ICICLE v3.8.0's documented ntt::ntt (natural order "kNN",
 * forward X[k] = sum_j x[j] w^{jk}, inverse scaled by 1/n, coset: forward evaluates on g*<w>,
 * inverse undoes it) as exercised by libs/src/tests.rs:107-180 and 1075-1087.
 * ------------------------------------------------------------------------------------------ */
static int log2_exact(size_t n) {
    int l = 0;
    if (n == 0 || (n & (n - 1))) return -1;
    while (((size_t)1 << l) < n) l++;
    return l;
}
#define FS(x) fr_##x
#define TKN(x) tko_##x
#define NTT_ROOT FR_ROOT32
#define NTT_TWO_ADICITY 32
#include "tk_ntt.inc"
#undef FS
#undef TKN
#undef NTT_ROOT
#undef NTT_TWO_ADICITY

/* BN254 scalar field: two-adicity 28, root 5^((r-1)/2^28) (5 is the smallest quadratic non-residue; the generator both
 * arkworks and ffjavascript use for this field).  No counterpart in the reference. */
#define FS(x) bnr_##x
#define TKN(x) tko_bn254_##x
#define NTT_ROOT BNR_ROOT28
#define NTT_TWO_ADICITY 28
#include "tk_ntt.inc"
#undef FS
#undef TKN
#undef NTT_ROOT
#undef NTT_TWO_ADICITY

/* ------------------------------------------------------------------------------------------
 * G1 groups: the curve-generic code lives in tk_g1.inc and is instantiated per curve.
 * ------------------------------------------------------------------------------------------ */
/* standard generator, restated from setup/mpc-setup/src/conversions.rs:68-79 (u32 LE limbs) */
static const uint32_t g1_GEN_X[12] = {0xdb22c6bb, 0xfb3af00a, 0xf97a1aef, 0x6c55e83f, 0x171bac58, 0xa14e3a3f,
                                      0x9774b905, 0xc3688c4f, 0x4fa9ac0f, 0x2695638c, 0x3197d794, 0x17f1d3a7};
static const uint32_t g1_GEN_Y[12] = {1187375073u, 212476713u,  2726857444u, 3493644100u, 738505709u,  14358731u,
                                      3587181302u, 4243972245u, 1948093156u, 2694721773u, 3819610353u, 146011265u};
#define FQ(x) fq_##x
#define FS(x) fr_##x
#define G(x) g1_##x
#define TKO(x) tko_g1_##x
#define FQB 48
#include "tk_g1.inc"
#undef FQ
#undef FS
#undef G
#undef TKO
#undef FQB

/* ------------------------------------------------------------------------------------------
 * G2 of BLS12-381: the same curve-generic code over Fp2 = Fq[u] / (u^2 + 1), twist y^2 = x^3 + 4(1 + u).  The reference has no G2
 * MSM call site (G2 appears as nine scalar multiplications in Sigma2::gen, libs/src/group_structures/mod.rs:752-777, and in the
 * verifier's pairings); BASELINE.json's north_star names "G1/G2", so the product has bls12_381_g2_msm and this is its oracle twin.
 * Encoding (libs/src/iotools/mod.rs:1709-1713 G2SerdeRkyv; tests/test_g2.py pins it on the reference's fixed G2 generator): a
 * 96-byte Fp2 element = real part then imaginary part, 48-byte little-endian each; a point = x then y; all zero = infinity.
 * ------------------------------------------------------------------------------------------ */
typedef struct { fq_t c0, c1; } fq2_t;
static fq2_t fq2_R1;    /* 1 */
static fq2_t g2_CURVE_B; /* 4 + 4u (Montgomery) */
static inline void fq2_add(fq2_t *o, const fq2_t *a, const fq2_t *b) { fq_add(&o->c0, &a->c0, &b->c0); fq_add(&o->c1, &a->c1, &b->c1); }
static inline void fq2_sub(fq2_t *o, const fq2_t *a, const fq2_t *b) { fq_sub(&o->c0, &a->c0, &b->c0); fq_sub(&o->c1, &a->c1, &b->c1); }
static inline void fq2_neg(fq2_t *o, const fq2_t *a) { fq_neg(&o->c0, &a->c0); fq_neg(&o->c1, &a->c1); }
static inline int fq2_is_zero(const fq2_t *a) { return fq_is_zero(&a->c0) && fq_is_zero(&a->c1); }
static inline int fq2_eq(const fq2_t *a, const fq2_t *b) { return fq_eq(&a->c0, &b->c0) && fq_eq(&a->c1, &b->c1); }
static inline void fq2_mul(fq2_t *o, const fq2_t *a, const fq2_t *b) { /* schoolbook: (a0 b0 - a1 b1) + (a0 b1 + a1 b0) u */
    fq_t t0, t1, t2, t3;
    fq_mul(&t0, &a->c0, &b->c0);
    fq_mul(&t1, &a->c1, &b->c1);
    fq_mul(&t2, &a->c0, &b->c1);
    fq_mul(&t3, &a->c1, &b->c0);
    fq_sub(&o->c0, &t0, &t1);
    fq_add(&o->c1, &t2, &t3);
}
static inline void fq2_sqr(fq2_t *o, const fq2_t *a) { fq2_mul(o, a, a); }
static void fq2_inv(fq2_t *o, const fq2_t *a) { /* conj(a) / (a0^2 + a1^2) */
    fq_t n, t, ni;
    fq_sqr(&n, &a->c0);
    fq_sqr(&t, &a->c1);
    fq_add(&n, &n, &t);
    fq_inv(&ni, &n);
    fq_mul(&o->c0, &a->c0, &ni);
    fq_neg(&t, &a->c1);
    fq_mul(&o->c1, &t, &ni);
}
static inline void fq2_load(fq2_t *o, const uint8_t *b) { fq_load(&o->c0, b); fq_load(&o->c1, b + 48); }
static inline void fq2_store(uint8_t *b, const fq2_t *a) { fq_store(b, &a->c0); fq_store(b + 48, &a->c1); }
/* the standard G2 generator of BLS12-381 (the constant both arkworks and zkcrypto publish), u32 LE limbs: x.c0, x.c1, y.c0, y.c1 */
static const uint32_t g2_GEN_X[24] = {0xc121bdb8, 0xd48056c8, 0xa805bbef, 0x0bac0326, 0x7ae3d177, 0xb4510b64, 0xfa403b02, 0xc6e47ad4, 0x2dc51051, 0x26080527, 0xf08f0a91, 0x024aa2b2,
                                      0x5d042b7e, 0xe5ac7d05, 0x13945d57, 0x334cf112, 0xdc7f5049, 0xb5da61bb, 0x9920b61a, 0x596bd0d0, 0x88274f65, 0x7dacd3a0, 0x52719f60, 0x13e02b60};
static const uint32_t g2_GEN_Y[24] = {0x08b82801, 0xe1935486, 0x3baca289, 0x923ac9cc, 0x5160d12c, 0x6d429a69, 0x8cbdd3a7, 0xadfd9baa, 0xda2e351a, 0x8cc9cdc6, 0x727d6e11, 0x0ce5d527,
                                      0xf05f79be, 0xaaa9075f, 0x5cec1da1, 0x3f370d27, 0x572e99ab, 0x267492ab, 0x85a763af, 0xcb3e287e, 0x2bc28b99, 0x32acd2b0, 0x2ea734cc, 0x0606c4a0};
static void g2_init(void) {
    memset(&fq2_R1, 0, sizeof fq2_R1);
    fq2_R1.c0 = fq_R1;
    fq_t four;
    memset(&four, 0, sizeof four);
    four.l[0] = 4;
    fq_to_mont(&g2_CURVE_B.c0, &four);
    g2_CURVE_B.c1 = g2_CURVE_B.c0;
}
#define FQ(x) fq2_##x
#define FS(x) fr_##x
#define G(x) g2_##x
#define TKO(x) tko_g2_##x
#define FQB 96
#include "tk_g1.inc"
#undef FQ
#undef FS
#undef G
#undef TKO
#undef FQB

/* BN254 (alt_bn128): y^2 = x^3 + 3, generator (1, 2) (EIP-196).  No counterpart in the reference;
 * present because BASELINE.json's configs name a BN254 MSM. */
static const uint32_t bn_GEN_X[8] = {1, 0, 0, 0, 0, 0, 0, 0};
static const uint32_t bn_GEN_Y[8] = {2, 0, 0, 0, 0, 0, 0, 0};
#define FQ(x) bnq_##x
#define FS(x) bnr_##x
#define G(x) bn_##x
#define TKO(x) tko_bn254_g1_##x
#define FQB 32
#include "tk_g1.inc"
#undef FQ
#undef FS
#undef G
#undef TKO
#undef FQB

/* ------------------------------------------------------------------------------------------
 * Bivariate coefficient-matrix routines: restatements of the HOST loops of DensePolynomialExt
 * (packages/backend/libs/src/bivariate_polynomial/mod.rs).  Element (ix,iy) at ix*ys + iy.
 * ------------------------------------------------------------------------------------------ */
/* find_degree, mod.rs:1480-1515 */
void tko_poly_find_degree(const uint8_t *c, size_t xs, size_t ys, int64_t *xd, int64_t *yd) {
    static const uint8_t zero[32] = {0};
    *xd = -1;
    *yd = -1;
    for (size_t i = 0; i < xs; i++)
        for (size_t j = 0; j < ys; j++)
            if (memcmp(c + 32 * (i * ys + j), zero, 32)) {
                if ((int64_t)i > *xd) *xd = (int64_t)i;
                if ((int64_t)j > *yd) *yd = (int64_t)j;
            }
}
/* _find_size_as_twopower, mod.rs:72-86 */
static size_t pow2_at_least(size_t t) {
    size_t p = 1;
    while (p < t) p <<= 1;
    return p;
}
void tko_poly_resized_dims(size_t tx, size_t ty, size_t *nx, size_t *ny) {
    *nx = pow2_at_least(tx);
    *ny = pow2_at_least(ty);
}
/* resize, mod.rs:1784-1806: copy the common top-left block into a zeroed nx x ny matrix (nx, ny already rounded) */
void tko_poly_resize(const uint8_t *c, size_t xs, size_t ys, size_t nx, size_t ny, uint8_t *out) {
    memset(out, 0, 32 * nx * ny);
    size_t rx = xs < nx ? xs : nx, ry = ys < ny ? ys : ny;
    for (size_t i = 0; i < rx; i++) memcpy(out + 32 * ny * i, c + 32 * ys * i, 32 * ry);
}
/* mul_monomial, mod.rs:1820-1844: out is nx x ny (caller rounds (x_degree+1+ex, y_degree+1+ey) up to powers of two) */
int tko_poly_mul_monomial(const uint8_t *c, size_t xs, size_t ys, size_t ex, size_t ey, size_t nx, size_t ny, uint8_t *out) {
    if (xs + ex > nx || ys + ey > ny) return -1; /* the reference slice copy would panic */
    memset(out, 0, 32 * nx * ny);
    for (size_t i = 0; i < xs; i++) memcpy(out + 32 * (ny * (i + ex) + ey), c + 32 * ys * i, 32 * ys);
    return 0;
}
/* _scale_coeffs (both axes), mod.rs:1567-1613: out[i][j] = c[i][j] * fx^i * fy^j */
void tko_poly_scale_coeffs(const uint8_t *c, size_t xs, size_t ys, const uint8_t *fx, const uint8_t *fy, uint8_t *out) {
    tko_init();
    fr_t gx = fr_R1, gy = fr_R1;
    if (fx) fr_load(&gx, fx);
    if (fy) fr_load(&gy, fy);
    fr_t px = fr_R1;
    for (size_t i = 0; i < xs; i++) {
        fr_t py = px;
        for (size_t j = 0; j < ys; j++) {
            fr_t v;
            fr_load(&v, c + 32 * (i * ys + j));
            fr_mul(&v, &v, &py);
            fr_store(out + 32 * (i * ys + j), &v);
            fr_mul(&py, &py, &gy);
        }
        fr_mul(&px, &px, &gx);
    }
}
/* eval, mod.rs:1719-1750: P(x, y) by Horner over both axes */
void tko_poly_eval(const uint8_t *c, size_t xs, size_t ys, const uint8_t *x, const uint8_t *y, uint8_t *out) {
    tko_init();
    fr_t vx, vy, acc;
    fr_load(&vx, x);
    fr_load(&vy, y);
    memset(&acc, 0, sizeof acc);
    for (size_t i = xs; i-- > 0;) {
        fr_t row, t;
        memset(&row, 0, sizeof row);
        for (size_t j = ys; j-- > 0;) {
            fr_load(&t, c + 32 * (i * ys + j));
            fr_mul(&row, &row, &vy);
            fr_add(&row, &row, &t);
        }
        fr_mul(&acc, &acc, &vx);
        fr_add(&acc, &acc, &row);
    }
    fr_store(out, &acc);
}
/* eval_x (out: ys values) / eval_y (out: xs values), mod.rs:1719-1740 */
void tko_poly_eval_x(const uint8_t *c, size_t xs, size_t ys, const uint8_t *x, uint8_t *out) {
    tko_init();
    fr_t vx;
    fr_load(&vx, x);
    for (size_t j = 0; j < ys; j++) {
        fr_t acc, t;
        memset(&acc, 0, sizeof acc);
        for (size_t i = xs; i-- > 0;) {
            fr_load(&t, c + 32 * (i * ys + j));
            fr_mul(&acc, &acc, &vx);
            fr_add(&acc, &acc, &t);
        }
        fr_store(out + 32 * j, &acc);
    }
}
void tko_poly_eval_y(const uint8_t *c, size_t xs, size_t ys, const uint8_t *y, uint8_t *out) {
    tko_init();
    fr_t vy;
    fr_load(&vy, y);
    for (size_t i = 0; i < xs; i++) {
        fr_t acc, t;
        memset(&acc, 0, sizeof acc);
        for (size_t j = ys; j-- > 0;) {
            fr_load(&t, c + 32 * (i * ys + j));
            fr_mul(&acc, &acc, &vy);
            fr_add(&acc, &acc, &t);
        }
        fr_store(out + 32 * i, &acc);
    }
}
/* div_by_vanishing_opt, mod.rs:2284-2410, statement for statement (after its optimize_size):
 * p is xs x ys with c | xs, d | ys; quo_x is xs x ys, quo_y is c x ys */
int tko_poly_div_by_vanishing_opt(const uint8_t *p, size_t xs, size_t ys, size_t c, size_t d, uint8_t *quo_x, uint8_t *quo_y) {
    tko_init();
    if (!c || !d || xs % c || ys % d) return -1;
    size_t m = xs / c;
    fr_t *P = (fr_t *)malloc(sizeof(fr_t) * xs * ys), *acc = (fr_t *)calloc(c * ys, sizeof(fr_t));
    fr_t *qy = (fr_t *)calloc(c * ys, sizeof(fr_t)), *qx = (fr_t *)calloc(xs * ys, sizeof(fr_t));
    for (size_t i = 0; i < xs * ys; i++) fr_load(&P[i], p + 32 * i);
    for (size_t bx = 0; bx < m; bx++)
        for (size_t lx = 0; lx < c; lx++)
            for (size_t y = 0; y < ys; y++) fr_add(&acc[lx * ys + y], &acc[lx * ys + y], &P[(bx * c + lx) * ys + y]);
    if (ys > d)
        for (size_t x = 0; x < c; x++)
            for (size_t y = 0; y < ys - d; y++) {
                fr_t prev;
                memset(&prev, 0, sizeof prev);
                if (y >= d) prev = qy[x * ys + y - d];
                fr_sub(&qy[x * ys + y], &prev, &acc[x * ys + y]);
            }
    fr_t *b = P; /* b_coeffs_vec = p_coeffs_vec */
    if (ys > d)
        for (size_t x = 0; x < c; x++)
            for (size_t y = 0; y < ys - d; y++) {
                fr_t co = qy[x * ys + y];
                fr_add(&b[x * ys + y], &b[x * ys + y], &co);
                fr_sub(&b[x * ys + y + d], &b[x * ys + y + d], &co);
            }
    if (xs > c)
        for (size_t x = 0; x < xs - c; x++)
            for (size_t y = 0; y < ys; y++) {
                fr_t prev;
                memset(&prev, 0, sizeof prev);
                if (x >= c) prev = qx[(x - c) * ys + y];
                fr_sub(&qx[x * ys + y], &prev, &b[x * ys + y]);
            }
    for (size_t i = 0; i < xs * ys; i++) fr_store(quo_x + 32 * i, &qx[i]);
    for (size_t i = 0; i < c * ys; i++) fr_store(quo_y + 32 * i, &qy[i]);
    free(P);
    free(acc);
    free(qy);
    free(qx);
    return 0;
}
/* _div_uni_coeffs_by_ruffini, mod.rs:2460-2477 (Montgomery values; q has len entries) */
static void ruffini_uni(const fr_t *co, size_t len, const fr_t *x, fr_t *q, fr_t *r) {
    memset(q, 0, sizeof(fr_t) * len);
    if (len < 2) {
        *r = co[0];
        return;
    }
    fr_t b = co[len - 1], t;
    q[len - 2] = b;
    for (size_t i = 3; i < len + 1; i++) {
        fr_mul(&t, &b, x);
        fr_add(&b, &co[len - i + 1], &t);
        q[len - i] = b;
    }
    fr_mul(&t, &b, x);
    fr_add(r, &co[0], &t);
}
/* div_by_ruffini, mod.rs:2412-2458: q_x is xs x ys, q_y has ys entries, r one */
void tko_poly_div_by_ruffini(const uint8_t *p, size_t xs, size_t ys, const uint8_t *x, const uint8_t *y, uint8_t *q_x,
                             uint8_t *q_y, uint8_t *r) {
    tko_init();
    fr_t vx, vy;
    fr_load(&vx, x);
    fr_load(&vy, y);
    fr_t *col = (fr_t *)malloc(sizeof(fr_t) * xs), *q = (fr_t *)malloc(sizeof(fr_t) * (xs > ys ? xs : ys));
    fr_t *rx = (fr_t *)malloc(sizeof(fr_t) * ys);
    for (size_t j = 0; j < ys; j++) {
        for (size_t i = 0; i < xs; i++) fr_load(&col[i], p + 32 * (i * ys + j));
        ruffini_uni(col, xs, &vx, q, &rx[j]);
        for (size_t i = 0; i < xs; i++) fr_store(q_x + 32 * (i * ys + j), &q[i]);
    }
    fr_t rem;
    ruffini_uni(rx, ys, &vy, q, &rem);
    for (size_t j = 0; j < ys; j++) fr_store(q_y + 32 * j, &q[j]);
    fr_store(r, &rem);
    free(col);
    free(q);
    free(rx);
}

/* prove1 running product, packages/backend/prove/src/lib.rs:1858-1862:
 * r[n-1] = 1; for idx in (0..n-1).rev(): r[idx] = r[idx+1] * s[idx+1] */
void tko_fr_suffix_product(const uint8_t *s, size_t n, uint8_t *out) {
    tko_init();
    if (n == 0) return;
    fr_t r = fr_R1, t;
    fr_store(out + 32 * (n - 1), &r);
    for (size_t idx = n - 1; idx-- > 0;) {
        fr_load(&t, s + 32 * (idx + 1));
        fr_mul(&r, &r, &t);
        fr_store(out + 32 * idx, &r);
    }
}

/* eval_sparse_rows, packages/backend/libs/src/iotools/mod.rs:1590-1608, on CSR input: for one placement,
 * out[row] = sum over the row's entries of coeff * variables[wire]  (rows >= out_len dropped; empty rows stay 0) */
void tko_r1cs_eval_rows(const uint32_t *row_ptr, const uint32_t *wire, const uint8_t *coeff, size_t n_rows,
                        const uint8_t *variables, uint8_t *out, size_t out_len) {
    tko_init();
    memset(out, 0, 32 * out_len);
    for (size_t r = 0; r < n_rows; r++) {
        if (row_ptr[r] == row_ptr[r + 1]) continue;
        fr_t acc, c, v, t;
        memset(&acc, 0, sizeof acc);
        for (uint32_t k = row_ptr[r]; k < row_ptr[r + 1]; k++) {
            fr_load(&c, coeff + 32 * (size_t)k);
            fr_load(&v, variables + 32 * (size_t)wire[k]);
            fr_mul(&t, &c, &v);
            fr_add(&acc, &acc, &t);
        }
        if (r < out_len) fr_store(out + 32 * r, &acc);
    }
}
