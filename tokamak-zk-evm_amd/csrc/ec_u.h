// ec_u.h — XYZZ mixed addition (madd-2008-s, 8M + 2S) on the unsaturated field representation (ffu.h); the
// hot loop of the MSM bucket accumulation.  Same group law and the same degenerate-case behaviour as ec.h
// (P == Q doubles, P == -Q and infinity handled); the rare degenerate cases leave the fast path and run through
// the saturated implementation (ec.h), so there is a single place where they are decided.
//
// Value bounds (multiples of p; every operand of a product must stay below 2^10 p, every subtrahend below the K of
// its sub<K>).  Products return < 1.03 p.  With the accumulator invariant X1 < 5.03p, Y1, ZZ1, ZZZ1 < 1.03p:
//   U2 = X2 ZZ1, S2 = Y2 ZZZ1                    < 1.03
//   P  = U2 - X1  (sub<8>)   in (2.97, 9.03)     == 0 mod p  <=>  P in {3p..9p}
//   R  = S2 - Y1  (sub<2>)   in (0.97, 3.03)
//   PP = P^2, PPP = P PP, Q = X1 PP              < 1.03
//   X3 = R^2 - (PPP + 2Q)  (sub<4>, t < 3.09)    < 5.03      (invariant restored)
//   Y3 = [R (Q - X3) + (2p - Y1) PPP] / R'       < 1.03      (sub<8>, sub<2>; ONE reduction for both products)
//   ZZ3 = ZZ1 PP, ZZZ3 = ZZZ1 PPP                < 1.03
#pragma once
#include "ec.h"
#include "ffu.h"

template <class P>
struct ecu {
    using FU = ffu<P>;
    using FS = ff<P>;
    using GS = ec<FS>;
    using E = typename FU::E;
    struct A {  // affine base: strict, canonical (< p), Montgomery radix 2^(W L); never infinity on the fast path
        E x, y;
    };
    struct X {
        E x, y, zz, zzz;
        bool inf;
    };

    static FF_HD X inf() {
        X r;
        r.x = FU::zero();
        r.y = FU::zero();
        r.zz = FU::zero();
        r.zzz = FU::zero();
        r.inf = true;
        return r;
    }
    // packed record written by k_convert_bases: {x 2^(W L) mod p, y 2^(W L) mod p} as 32-bit limbs; (0,0) = infinity
    static FF_HD bool load_affine(A &q, const affine_t<FS> &rec) {
        if (FS::is_zero(rec.x) && FS::is_zero(rec.y)) return false;
        q.x = FU::from_packed(rec.x);
        q.y = FU::from_packed(rec.y);
        return true;
    }
    static FF_HD A neg(const A &q) {
        A r;
        r.x = q.x;
        E m;
#pragma unroll
        for (int i = 0; i < FU::L; i++) m.l[i] = P::MODU[i];
        // p - y, limb-wise with one borrow sweep (y canonical, y != 0 on the curve y^2 = x^3 + b with b != 0 ... y = 0
        // cannot occur in the prime-order subgroup; p - 0 = p would still be a valid redundant value)
        E t;
        uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < FU::L; i++) {
            uint32_t d = m.l[i] - q.y.l[i] - br;
            br = d >> 31;
            t.l[i] = i < FU::L - 1 ? (d & FU::MASK) : d;
        }
        r.y = t;
        return r;
    }
    static FF_HD X from_affine(const A &q) {
        X r;
        r.x = q.x;
        r.y = q.y;
        r.zz = FU::one();
        r.zzz = FU::one();
        r.inf = false;
        return r;
    }

    // unsaturated XYZZ -> saturated Montgomery XYZZ (what the combine / reduce kernels and the host consume)
    static FF_HD xyzz_t<FS> to_sat(const X &p) {
        xyzz_t<FS> r;
        if (p.inf) return GS::inf();
        r.x = FU::to_sat_mont(p.x);
        r.y = FU::to_sat_mont(p.y);
        r.zz = FU::to_sat_mont(p.zz);
        r.zzz = FU::to_sat_mont(p.zzz);
        return r;
    }
    // "raw" record: the unsaturated coordinates as they are (x R' mod p, redundant: X < 5.03p, the others < 1.03p — both
    // fit the 32N-bit words), normalised and packed, all-zero = infinity.  Writing it costs no multiplication, which
    // matters inside the accumulate loop: lanes leave a bucket at different iterations, so a flush is executed by the
    // whole wave for one or two active lanes; the four conversion products of to_sat() are done later, convergently,
    // by whoever reads the record (raw_to_sat).
    static FF_HD xyzz_t<FS> to_raw(const X &p) {
        xyzz_t<FS> r;
        if (p.inf) return GS::inf();
        r.x = FU::pack(FU::norm(p.x));
        r.y = FU::pack(FU::norm(p.y));
        r.zz = FU::pack(FU::norm(p.zz));
        r.zzz = FU::pack(FU::norm(p.zzz));
        return r;
    }
    static FF_HD xyzz_t<FS> raw_to_sat(const xyzz_t<FS> &raw) {
        if (FS::is_zero(raw.zz)) return GS::inf();   // zz == 0 exactly only for the infinity record (zz != 0 mod p otherwise)
        xyzz_t<FS> r;
        r.x = FU::to_sat_mont(FU::unpack(raw.x));
        r.y = FU::to_sat_mont(FU::unpack(raw.y));
        r.zz = FU::to_sat_mont(FU::unpack(raw.zz));
        r.zzz = FU::to_sat_mont(FU::unpack(raw.zzz));
        return r;
    }
    // raw record -> unsaturated point: unpack only (the bounds of to_raw's inputs carry over)
    static FF_HD X from_raw(const xyzz_t<FS> &raw) {
        X r;
        r.inf = FS::is_zero(raw.zz);
        if (r.inf) return inf();
        r.x = FU::unpack(raw.x);
        r.y = FU::unpack(raw.y);
        r.zz = FU::unpack(raw.zz);
        r.zzz = FU::unpack(raw.zzz);
        return r;
    }
    static FF_HD E from_sat_mont(const typename FS::E &s) {  // x 2^(32N) canonical -> x 2^(W L) strict canonical
        typename FS::E k;
#pragma unroll
        for (int i = 0; i < FS::N; i++) k.l[i] = P::KSATM[i];
        return FU::unpack(FS::mul(s, k));
    }
    static FF_HD X from_sat(const xyzz_t<FS> &p) {
        X r;
        r.inf = GS::is_inf(p);
        if (r.inf) return inf();
        r.x = from_sat_mont(p.x);
        r.y = from_sat_mont(p.y);
        r.zz = from_sat_mont(p.zz);
        r.zzz = from_sat_mont(p.zzz);
        return r;
    }
    // degenerate cases: decide and compute on the saturated representation (ec.h handles P == Q, P == -Q)
    static FF_HD X add_mixed_slow(const X &p, const A &q) {
        affine_t<FS> qs;
        qs.x = FU::to_sat_mont(q.x);
        qs.y = FU::to_sat_mont(q.y);
        return from_sat(GS::add_mixed(to_sat(p), qs));
    }

    // general addition (add-2008-s, 12M + 2S, one shared reduction) for the combine / bucket-reduction kernels.  With both
    // operands in the accumulator invariant (X < 5.03p, Y, ZZ, ZZZ < 1.03p):
    //   U1 = X1 ZZ2, U2 = X2 ZZ1, S1 = Y1 ZZZ2, S2 = Y2 ZZZ1      < 1.03
    //   P = U2 - U1 (sub<2>), R = S2 - S1 (sub<2>)                 in (0.97, 3.03);  P == 0 mod p <=> P in {p, 2p, 3p}
    //   X3 = R^2 - (PPP + 2Q)  (sub<4>)  < 5.03;  Y3 = [R (Q - X3) + (2p - S1) PPP]/R' < 1.03;  ZZ3, ZZZ3 < 1.03
    static FF_HD X add_slow(const X &p, const X &q) { return from_sat(GS::add(to_sat(p), to_sat(q))); }
    static FF_HD X add(const X &p, const X &q) {
        if (p.inf) return q;
        if (q.inf) return p;
        E u1 = FU::mul(p.x, q.zz), u2 = FU::mul(q.x, p.zz);
        E s1 = FU::mul(p.y, q.zzz), s2 = FU::mul(q.y, p.zzz);
        E pd = FU::template sub<2>(u2, u1);
        if (FU::maybe_multiple_of_p(pd, 3)) return add_slow(p, q);   // same x: doubling or cancellation, decided in ec.h
        E rd = FU::template sub<2>(s2, s1);
        E pp = FU::sqr(pd);
        E ppp = FU::mul(pd, pp);
        E qq = FU::mul(u1, pp);
        X r;
        r.inf = false;
        r.x = FU::template sub<4>(FU::sqr(rd), FU::add_dbl(ppp, qq));
        r.y = FU::mul_add(rd, FU::template sub<8>(qq, r.x), FU::template sub<2>(FU::zero(), s1), ppp);
        r.zz = FU::mul(FU::mul(p.zz, q.zz), pp);
        r.zzz = FU::mul(FU::mul(p.zzz, q.zzz), ppp);
        return r;
    }
    // doubling (dbl-2008-s-1, a = 0): U = 2Y1 < 2.06, V = U^2, W = U V, S = X1 V, M = 3 X1^2 < 3.09,
    //   X3 = M^2 - 2S (sub<4>) < 5.03,  Y3 = [M (S - X3) + (2p - W) Y1]/R' < 1.03,  ZZ3 = V ZZ1, ZZZ3 = W ZZZ1
    static FF_HD X dbl(const X &p) {
        if (p.inf) return p;
        E u = FU::dbl(p.y);
        E v = FU::sqr(u);
        E w = FU::mul(u, v);
        E sx = FU::mul(p.x, v);
        E x2 = FU::sqr(p.x);
        E m = FU::add(FU::dbl(x2), x2);
        X r;
        r.inf = false;
        r.x = FU::template sub<4>(FU::sqr(m), FU::dbl(sx));
        r.y = FU::mul_add(m, FU::template sub<8>(sx, r.x), FU::template sub<2>(FU::zero(), w), p.y);
        r.zz = FU::mul(v, p.zz);
        r.zzz = FU::mul(w, p.zzz);
        return r;
    }

    // p + q, q an affine point (not infinity)
    static FF_HD X add_mixed(const X &p, const A &q) {
        if (p.inf) return from_affine(q);
        E u2 = FU::mul(q.x, p.zz);
        E s2 = FU::mul(q.y, p.zzz);
        E pd = FU::template sub<8>(u2, p.x);
        // P == 0 mod p  <=>  P = j p with 3 <= j <= 9: only then can q share its x with the accumulator
        if (FU::maybe_multiple_of_p(pd, 9)) return add_mixed_slow(p, q);
        E rd = FU::template sub<2>(s2, p.y);
        E pp = FU::sqr(pd);
        E ppp = FU::mul(pd, pp);
        E qq = FU::mul(p.x, pp);
        X r;
        r.inf = false;
        E t = FU::add_dbl(ppp, qq);
        r.x = FU::template sub<4>(FU::sqr(rd), t);
        E d = FU::template sub<8>(qq, r.x);
        r.y = FU::mul_add(rd, d, FU::template sub<2>(FU::zero(), p.y), ppp);  // R (Q - X3) + (2p - Y1) PPP, one reduction
        r.zz = FU::mul(p.zz, pp);
        r.zzz = FU::mul(p.zzz, ppp);
        return r;
    }
};

using G1U = ecu<bls12_381_fq_params>;
