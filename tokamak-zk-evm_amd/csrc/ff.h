// ff.h — prime-field arithmetic on N x 32-bit limbs for gfx950 (and the host, for the launchers
// and the CPU self-test).  Montgomery form with R = 2^(32N); every modulus has >= 1 spare bit.
//
// This is integer VALU work: the inner product step is v_mad_u64_u32 (32x32+64 -> 64).  No MFMA.
// Values live in VGPRs as little-endian u32 limbs — the same layout ICICLE exposes in memory
// (reference pin: packages/backend/setup/mpc-setup/src/conversions.rs:43-95), so loads/stores are
// straight dwordx4 copies with no repacking.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "field_params.h"

#define FF_HD __host__ __device__ __forceinline__

template <class P>
struct alignas(16) Fp {
    static constexpr int N = P::N;
    uint32_t l[N];
};

template <class P>
struct ff {
    static constexpr int N = P::N;
    using E = Fp<P>;

    // Constants are read through these so that device code gets immediates (constexpr arrays of a
    // host struct are not addressable from device code without relocatable constants).
    static FF_HD constexpr uint32_t mod(int i) { return P::MOD[i]; }

    static FF_HD E zero() {
        E r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = 0;
        return r;
    }
    static FF_HD E one() {  // Montgomery 1
        E r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = P::ONE[i];
        return r;
    }
    static FF_HD E r2() {
        E r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = P::R2[i];
        return r;
    }
    static FF_HD E modulus() {
        E r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = P::MOD[i];
        return r;
    }
    static FF_HD bool is_zero(const E &a) {
        uint32_t x = 0;
#pragma unroll
        for (int i = 0; i < N; i++) x |= a.l[i];
        return x == 0;
    }
    static FF_HD bool eq(const E &a, const E &b) {
        uint32_t x = 0;
#pragma unroll
        for (int i = 0; i < N; i++) x |= a.l[i] ^ b.l[i];
        return x == 0;
    }

    // r = a + b (no reduction); returns carry out
    static FF_HD uint32_t add_raw(E &r, const E &a, const E &b) {
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            uint64_t s = (uint64_t)a.l[i] + b.l[i] + c;
            r.l[i] = (uint32_t)s;
            c = (uint32_t)(s >> 32);
        }
        return c;
    }
    // r = a - b; returns borrow out (1 if a < b)
    static FF_HD uint32_t sub_raw(E &r, const E &a, const E &b) {
        uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            uint64_t d = (uint64_t)a.l[i] - b.l[i] - br;
            r.l[i] = (uint32_t)d;
            br = (uint32_t)(d >> 32) & 1u;
        }
        return br;
    }
    // a in [0, 2p) -> [0, p).  Branch-free WITHOUT v_cndmask: on gfx950 a VCC-masked v_cndmask_b32 issues at
    // ~16 cycles per wave (profiles/r01_valu_issue_probes.json) against ~3 + ~4.7 for v_and + v_addc_co, so the
    // select is done arithmetically: t = a - p; r = t + (p & -borrow).  On the device add / sub / reduce are
    // single inline-asm carry chains with the modulus as VOP2 literals (generated: field_params.h); hipcc
    // lowers the portable u64 form to v_lshl_add_u64 plus a zero-extending v_mov per limb.
    static FF_HD E reduce_once(const E &a) {
#if defined(__HIP_DEVICE_COMPILE__)
        E r = a;
        P::reduce_once_asm(r.l);
        return r;
#else
        E t;
        uint32_t br = sub_raw(t, a, modulus());
        uint32_t mask = 0u - br;
        E r;
        uint32_t c = 0;
        for (int i = 0; i < N; i++) {
            uint64_t s = (uint64_t)t.l[i] + (P::MOD[i] & mask) + c;
            r.l[i] = (uint32_t)s;
            c = (uint32_t)(s >> 32);
        }
        return r;
#endif
    }
    static FF_HD E add(const E &a, const E &b) {  // inputs < p; sum < 2p < 2^(32N)
#if defined(__HIP_DEVICE_COMPILE__)
        E r = a;
        P::add_mod_asm(r.l, b.l);
        return r;
#else
        E s;
        add_raw(s, a, b);
        return reduce_once(s);
#endif
    }
    static FF_HD E sub(const E &a, const E &b) {
#if defined(__HIP_DEVICE_COMPILE__)
        E r = a;
        P::sub_mod_asm(r.l, b.l);
        return r;
#else
        E d;
        uint32_t br = sub_raw(d, a, b);
        uint32_t mask = 0u - br;
        E r;
        uint32_t c = 0;
        for (int i = 0; i < N; i++) {
            uint64_t s = (uint64_t)d.l[i] + (P::MOD[i] & mask) + c;
            r.l[i] = (uint32_t)s;
            c = (uint32_t)(s >> 32);
        }
        return r;
#endif
    }
    static FF_HD E neg(const E &a) {
        E d;
        sub_raw(d, modulus(), a);
        uint32_t mask = is_zero(a) ? 0u : 0xffffffffu;
        E r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = d.l[i] & mask;
        return r;
    }
    static FF_HD E dbl(const E &a) { return add(a, a); }

    // Montgomery product a*b/R mod p; inputs < p, output < p.
    //
    // Device path: product scanning (FIPS) with a 96-bit column accumulator {c2 : acc(64)}.  Every limb
    // product is exactly one v_mad_u64_u32 (64-bit accumulate, carry-out to VCC) plus one v_addc_co_u32
    // into the third word; the modulus limbs ride in SGPRs; one inline-asm statement per column.  hipcc's lowering of the portable CIOS loop
    // below spends ~2.7x the VALU slots of the mads themselves on zero-extension moves and 64-bit adds
    // (measured: 38 % of the raw v_mad_u64_u32 rate); this form is 1 mad + 1 addc per product.
    // Host path (launchers, self-test): portable CIOS over 32-bit limbs.
    static FF_HD E mul(const E &a, const E &b) {
#if defined(__HIP_DEVICE_COMPILE__)
        return mul_ps(a, b);
#elif defined(__SIZEOF_INT128__) && defined(__BYTE_ORDER__) && __BYTE_ORDER__ == __ORDER_LITTLE_ENDIAN__
        return mul_cios64(a, b);
#else
        return mul_cios(a, b);
#endif
    }

#if !defined(__HIP_DEVICE_COMPILE__) && defined(__SIZEOF_INT128__) && defined(__BYTE_ORDER__) && __BYTE_ORDER__ == __ORDER_LITTLE_ENDIAN__
    // Host path: the same Montgomery product (radix 2^(32 N), N even) over N / 2 limbs of 64 bits — a pair of 32-bit limbs in memory IS
    // a 64-bit limb on a little-endian host.  Four times fewer limb products than the 32-bit CIOS: the host side's Horner steps
    // over the window sums of an MSM run a few hundred group operations per call.
    static E mul_cios64(const E &a, const E &b) {
        static_assert(N % 2 == 0, "an even number of 32-bit limbs");
        constexpr int M = N / 2;
        uint64_t x[M], y[M], p[M], t[M + 2];
        for (int i = 0; i < M; i++) {
            x[i] = (uint64_t)a.l[2 * i] | ((uint64_t)a.l[2 * i + 1] << 32);
            y[i] = (uint64_t)b.l[2 * i] | ((uint64_t)b.l[2 * i + 1] << 32);
            p[i] = (uint64_t)P::MOD[2 * i] | ((uint64_t)P::MOD[2 * i + 1] << 32);
        }
        uint64_t inv = P::INV;                      // -p^-1 mod 2^32 -> mod 2^64 by one Newton step on p^-1
        {
            uint64_t pinv = 0 - inv;                 // p^-1 mod 2^32
            pinv *= 2 - p[0] * pinv;                 // mod 2^64
            inv = 0 - pinv;
        }
        for (int i = 0; i < M + 2; i++) t[i] = 0;
        for (int i = 0; i < M; i++) {
            unsigned __int128 c = 0;
            for (int j = 0; j < M; j++) {
                c += (unsigned __int128)x[j] * y[i] + t[j];
                t[j] = (uint64_t)c;
                c >>= 64;
            }
            c += t[M];
            t[M] = (uint64_t)c;
            t[M + 1] = (uint64_t)(c >> 64);
            const uint64_t m = t[0] * inv;
            c = (unsigned __int128)m * p[0] + t[0];
            c >>= 64;
            for (int j = 1; j < M; j++) {
                c += (unsigned __int128)m * p[j] + t[j];
                t[j - 1] = (uint64_t)c;
                c >>= 64;
            }
            c += t[M];
            t[M - 1] = (uint64_t)c;
            t[M] = t[M + 1] + (uint64_t)(c >> 64);
        }
        E r;
        for (int i = 0; i < M; i++) r.l[2 * i] = (uint32_t)t[i], r.l[2 * i + 1] = (uint32_t)(t[i] >> 32);
        return reduce_once(r);   // t < 2p < 2^(32N): t[M] == 0 here
    }
#endif

#if defined(__HIP_DEVICE_COMPILE__)
    static __device__ __forceinline__ E mul_ps(const E &a, const E &b) {
        E r;
        P::mul_ps(a.l, b.l, r.l);  // generated straight-line code (field_params.h)
        return reduce_once(r);
    }
#endif

    // portable CIOS.  With one spare bit in the modulus the running value stays below 2^(32N+32).
    static FF_HD E mul_cios(const E &a, const E &b) {
        uint32_t t[N + 1];
#pragma unroll
        for (int i = 0; i <= N; i++) t[i] = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            uint32_t c = 0;
#pragma unroll
            for (int j = 0; j < N; j++) {
                uint64_t s = (uint64_t)a.l[j] * b.l[i] + t[j] + c;
                t[j] = (uint32_t)s;
                c = (uint32_t)(s >> 32);
            }
            uint32_t tn = t[N] + c;  // cannot overflow (bound above)
            uint32_t m = t[0] * P::INV;
            uint64_t s = (uint64_t)m * P::MOD[0] + t[0];
            c = (uint32_t)(s >> 32);
#pragma unroll
            for (int j = 1; j < N; j++) {
                s = (uint64_t)m * P::MOD[j] + t[j] + c;
                t[j - 1] = (uint32_t)s;
                c = (uint32_t)(s >> 32);
            }
            s = (uint64_t)tn + c;
            t[N - 1] = (uint32_t)s;
            t[N] = (uint32_t)(s >> 32);
        }
        E r;
#pragma unroll
        for (int i = 0; i < N; i++) r.l[i] = t[i];
        return reduce_once(r);  // t < 2p and 2p < 2^(32N) so t[N] == 0 here
    }
    static FF_HD E sqr(const E &a) { return mul(a, a); }

    static FF_HD E to_mont(const E &a) { return mul(a, r2()); }
    static FF_HD E from_mont(const E &a) {
        E o = zero();
        o.l[0] = 1;
        return mul(a, o);
    }
    static FF_HD E from_u32(uint32_t v) {  // Montgomery form of a small integer
        E o = zero();
        o.l[0] = v;
        return to_mont(o);
    }
    // a^e for a 64-bit exponent (Montgomery in/out)
    static FF_HD E pow_u64(const E &a, uint64_t e) {
        E acc = one(), base = a;
        while (e) {
            if (e & 1) acc = mul(acc, base);
            base = sqr(base);
            e >>= 1;
        }
        return acc;
    }
    // Fermat inverse (Montgomery in/out); inv(0) = 0
    static FF_HD E inv(const E &a) {
        E acc = one(), base = a;
        uint32_t e[N];
#pragma unroll
        for (int i = 0; i < N; i++) e[i] = P::MOD[i];
        {  // e = p - 2 with borrow (r's low limb is 0x00000001)
            uint32_t br = 2;
            for (int i = 0; i < N && br; i++) {
                uint32_t v = e[i];
                e[i] = v - br;
                br = v < br ? 1u : 0u;
            }
        }
        for (int i = 0; i < N; i++)
            for (int b = 0; b < 32; b++) {
                if ((e[i] >> b) & 1) acc = mul(acc, base);
                base = sqr(base);
            }
        return acc;
    }
    // plain value possibly >= p (any N-limb integer) -> canonical plain
    static FF_HD E canon(const E &a) {
        E r = a;
        // 2^(32N) / p < 2^... : at most a few subtractions for the fields used here
        for (int k = 0; k < 8; k++) {
            E t;
            uint32_t br = sub_raw(t, r, modulus());
            if (br) break;
            r = t;
        }
        return r;
    }
};

using Fr = ff<bls12_381_fr_params>;
using Fq = ff<bls12_381_fq_params>;
using fr_t = Fr::E;
using fq_t = Fq::E;
