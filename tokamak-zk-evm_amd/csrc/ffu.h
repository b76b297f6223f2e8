// ffu.h — "unsaturated" prime-field arithmetic for the MSM accumulate loop: LU limbs of 29 bits in 32-bit
// registers, Montgomery radix 2^(29 LU).
//
// Why a second representation.  On gfx950 an add-with-carry costs as much issue time as a multiply
// (v_addc_co_u32 ~4.7 vs v_mad_u64_u32 ~5.3 cycles per wave, profiles/r01_valu_issue_probes.json), so in the
// saturated 32-bit-limb product (ff.h) half of the VALU time is carry handling.  With 29-bit limbs a column of
// up to 2*LU partial products (< 2^58 each) fits a 64-bit accumulator: every limb product is ONE v_mad_u64_u32 and
// nothing else; additions and subtractions are limb-wise with one carry sweep and never reduce modulo p.
// Cost: LU^2 / N^2 more products (14 vs 12 limbs for the BLS12-381 base field: 392 mads vs 288 mad + 288 addc).
//
// Invariants ("strict" form): limbs 0..LU-2 < 2^29; the top limb holds the rest.  Values are kept only modulo p
// (redundantly): a Montgomery product of operands below 2^10 p is below 1.03 p (the radix has >= 25 spare bits
// for Fq), sums and differences grow by the stated multiples of p and are bounded by construction in the curve
// formulas (ec_u.h).  In host builds every bound is checked with assert (tests/hostcheck).
#pragma once
#include <stdint.h>

#include "ff.h"

#if !defined(__HIP_DEVICE_COMPILE__)
#include <assert.h>
#define FFU_ASSERT(c) assert(c)
#else
#define FFU_ASSERT(c) ((void)0)
#endif

template <class P>
struct ffu {
    static constexpr int L = P::LU;
    static constexpr int N = P::N;  // saturated limb count
    static constexpr uint32_t W = P::WU, MASK = (1u << P::WU) - 1;  // 29 bits for BLS12-381 Fq, 28 for BN254 Fq
    struct alignas(8) E {
        uint32_t l[L];
    };
    using S = Fp<P>;  // saturated element (ff.h)

    static FF_HD E zero() {
        E r;
#pragma unroll
        for (int i = 0; i < L; i++) r.l[i] = 0;
        return r;
    }
    static FF_HD E one() {  // Montgomery 1 (2^(W L) mod p)
        E r;
#pragma unroll
        for (int i = 0; i < L; i++) r.l[i] = P::ONEU[i];
        return r;
    }
    static FF_HD bool strict(const E &a) {
        for (int i = 0; i < L - 1; i++)
            if (a.l[i] > MASK) return false;
        return true;
    }
    // value < k * p, judged on the top limb (host-side bound check only)
    static FF_HD bool below(const E &a, uint32_t k) { return (uint64_t)a.l[L - 1] * 1024 <= (uint64_t)k * P::TOP_PER_P_X1024 + 1024; }

    // one carry sweep: limbs < 2^32 in, strict out (the value must fit: top limb absorbs the last carry)
    static FF_HD E norm(const E &a) {
        E r;
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < L - 1; i++) {
            uint32_t t = a.l[i] + c;
            r.l[i] = t & MASK;
            c = t >> W;
        }
        r.l[L - 1] = a.l[L - 1] + c;
        return r;
    }
    static FF_HD E add(const E &a, const E &b) {
        FFU_ASSERT(strict(a) && strict(b));
        E t;
#pragma unroll
        for (int i = 0; i < L; i++) t.l[i] = a.l[i] + b.l[i];
        return norm(t);
    }
    static FF_HD E dbl(const E &a) { return add(a, a); }
    // a + 2 b with ONE carry sweep (limbs below 3 * 2^29 before it): the PPP + 2 Q of the addition formulas
    static FF_HD E add_dbl(const E &a, const E &b) {
        FFU_ASSERT(strict(a) && strict(b));
        E t;
#pragma unroll
        for (int i = 0; i < L; i++) t.l[i] = a.l[i] + (b.l[i] << 1);
        return norm(t);
    }
    // a - b + K p for K in {2, 4, 8, 16}; b must be strict with value < K p (so that every limb of K p's
    // dominating form covers b's limb); the result is strict with value < a + K p
    template <int K>
    static FF_HD E sub(const E &a, const E &b) {
        FFU_ASSERT(strict(a) && strict(b) && below(b, K));
        E t;
#pragma unroll
        for (int i = 0; i < L; i++) {
            uint32_t c = K == 2 ? P::SUB2[i] : K == 4 ? P::SUB4[i] : K == 8 ? P::SUB8[i] : P::SUB16[i];
            FFU_ASSERT(c >= b.l[i]);
            t.l[i] = a.l[i] + (c - b.l[i]);
        }
        return norm(t);
    }

    // Montgomery product a*b/2^(W L) mod p (redundant): strict operands, a*b < 2^20 p^2; strict result < 1.03 p.
    // Product scanning; a 64-bit column accumulator cannot overflow: 2L products < 2^58 plus a carry < 2^35.
    static FF_HD E mul(const E &a, const E &b) {
        FFU_ASSERT(strict(a) && strict(b) && a.l[L - 1] <= MASK && b.l[L - 1] <= MASK);
        uint32_t m[L];
        uint64_t acc = 0;
        E r;
#pragma unroll
        for (int k = 0; k < L; k++) {
#pragma unroll
            for (int j = 0; j <= k; j++) acc += (uint64_t)a.l[j] * b.l[k - j];
#pragma unroll
            for (int j = 0; j < k; j++) acc += (uint64_t)m[j] * P::MODU[k - j];
            m[k] = ((uint32_t)acc * P::INVU) & MASK;
            acc += (uint64_t)m[k] * P::MODU[0];
            acc >>= W;
        }
#pragma unroll
        for (int k = L; k < 2 * L - 1; k++) {
#pragma unroll
            for (int j = k - L + 1; j < L; j++) {
                acc += (uint64_t)a.l[j] * b.l[k - j];
                acc += (uint64_t)m[j] * P::MODU[k - j];
            }
            r.l[k - L] = (uint32_t)acc & MASK;
            acc >>= W;
        }
        r.l[L - 1] = (uint32_t)acc;
        return r;
    }
    // (a*b + c*d)/2^(W L) mod p with ONE Montgomery reduction (the reduction is a third of a product's multiply-adds):
    // strict operands with top limbs <= MASK, a*b + c*d < 2^20 p^2; strict result < 1.03 p.  A column holds up to 3L
    // products < 2^(2W): 42 * 2^58 (W = 29, L = 14) and 30 * 2^56 (W = 28, L = 10) are both below 2^64.
    static FF_HD E mul_add(const E &a, const E &b, const E &c, const E &d) {
        static_assert(3 * L < (1 << (64 - 2 * (int)W)), "column accumulator would overflow");
        FFU_ASSERT(strict(a) && strict(b) && strict(c) && strict(d));
        FFU_ASSERT(a.l[L - 1] <= MASK && b.l[L - 1] <= MASK && c.l[L - 1] <= MASK && d.l[L - 1] <= MASK);
        uint32_t m[L];
        uint64_t acc = 0;
        E r;
#pragma unroll
        for (int k = 0; k < L; k++) {
#pragma unroll
            for (int j = 0; j <= k; j++) {
                acc += (uint64_t)a.l[j] * b.l[k - j];
                acc += (uint64_t)c.l[j] * d.l[k - j];
            }
#pragma unroll
            for (int j = 0; j < k; j++) acc += (uint64_t)m[j] * P::MODU[k - j];
            m[k] = ((uint32_t)acc * P::INVU) & MASK;
            acc += (uint64_t)m[k] * P::MODU[0];
            acc >>= W;
        }
#pragma unroll
        for (int k = L; k < 2 * L - 1; k++) {
#pragma unroll
            for (int j = k - L + 1; j < L; j++) {
                acc += (uint64_t)a.l[j] * b.l[k - j];
                acc += (uint64_t)c.l[j] * d.l[k - j];
                acc += (uint64_t)m[j] * P::MODU[k - j];
            }
            r.l[k - L] = (uint32_t)acc & MASK;
            acc >>= W;
        }
        r.l[L - 1] = (uint32_t)acc;
        return r;
    }
    // a^2: off-diagonal products once against 2a (limbs < 2^30; column bound still below 2^63)
    static FF_HD E sqr(const E &a) {
        FFU_ASSERT(strict(a) && a.l[L - 1] <= MASK);
        uint32_t m[L], a2[L];
#pragma unroll
        for (int i = 0; i < L; i++) a2[i] = a.l[i] << 1;
        uint64_t acc = 0;
        E r;
#pragma unroll
        for (int k = 0; k < L; k++) {
#pragma unroll
            for (int j = 0; 2 * j < k; j++) acc += (uint64_t)a2[j] * a.l[k - j];
            if ((k & 1) == 0) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
#pragma unroll
            for (int j = 0; j < k; j++) acc += (uint64_t)m[j] * P::MODU[k - j];
            m[k] = ((uint32_t)acc * P::INVU) & MASK;
            acc += (uint64_t)m[k] * P::MODU[0];
            acc >>= W;
        }
#pragma unroll
        for (int k = L; k < 2 * L - 1; k++) {
#pragma unroll
            for (int j = k - L + 1; 2 * j < k; j++) acc += (uint64_t)a2[j] * a.l[k - j];
            if ((k & 1) == 0) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
#pragma unroll
            for (int j = k - L + 1; j < L; j++) acc += (uint64_t)m[j] * P::MODU[k - j];
            r.l[k - L] = (uint32_t)acc & MASK;
            acc >>= W;
        }
        r.l[L - 1] = (uint32_t)acc;
        return r;
    }

    // quick filter for "a == j p for some 0 <= j <= jmax": the low W bits of j p determine j
    static FF_HD bool maybe_multiple_of_p(const E &a, uint32_t jmax) { return ((a.l[0] * P::PINVU) & MASK) <= jmax; }

    // ---- conversions to / from the saturated layout (32-bit limbs) ----
    // 32-bit-limb integer below 2^(W L) -> strict
    static FF_HD E unpack(const S &s) {
        E r;
#pragma unroll
        for (int i = 0; i < L; i++) {
            int bit = (int)W * i, w = bit >> 5, sh = bit & 31;
            uint32_t lo = w < N ? s.l[w] : 0u, hi = w + 1 < N ? s.l[w + 1] : 0u;
            uint32_t v = sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
            r.l[i] = i < L - 1 ? (v & MASK) : v;
        }
        return r;
    }
    // strict value below 2^(32N) -> 32-bit limbs
    static FF_HD S pack(const E &a) {
        S s;
#pragma unroll
        for (int w = 0; w < N; w++) {
            uint32_t v = 0;
#pragma unroll
            for (int i = 0; i < L; i++) {
                int lo = (int)W * i - 32 * w;  // bit position of limb i relative to word w
                if (lo >= 32 || lo + 32 <= 0) continue;
                v |= lo >= 0 ? a.l[i] << lo : a.l[i] >> (-lo);
            }
            s.l[w] = v;
        }
        return s;
    }
    // strict a with value < 2p  ->  canonical [0, p)
    static FF_HD E cond_sub_p(const E &a) {
        E t;
        uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < L; i++) {
            uint32_t d = a.l[i] - P::MODU[i] - br;
            br = d >> 31;
            t.l[i] = i < L - 1 ? (d & MASK) : d;
        }
        uint32_t keep = 0u - br;  // all ones when a < p
        E r;
#pragma unroll
        for (int i = 0; i < L; i++) r.l[i] = (a.l[i] & keep) | (t.l[i] & ~keep);
        return r;
    }
    // x*2^(W L) (redundant, < 2^10 p)  ->  saturated Montgomery form x*2^(32N), canonical
    static FF_HD S to_sat_mont(const E &a) {
        E c;
#pragma unroll
        for (int i = 0; i < L; i++) c.l[i] = P::RSATU[i];
        return pack(cond_sub_p(mul(a, c)));
    }
    // packed canonical x*2^(W L) (what k_convert_bases stores)  ->  strict
    static FF_HD E from_packed(const S &s) { return unpack(s); }
};

using Fqu = ffu<bls12_381_fq_params>;
using fqu_t = Fqu::E;
