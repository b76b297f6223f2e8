// ec.h — short-Weierstrass a = 0 group arithmetic in extended Jacobian ("XYZZ") coordinates:
// x = X/ZZ, y = Y/ZZZ with ZZ^3 = ZZZ^2; ZZ = 0 <=> point at infinity.  XYZZ makes the bucket
// accumulate of Pippenger (accumulator += affine base) an 8M + 2S mixed addition with no inversion.
// All coordinates are in Montgomery form.  Formulas: EFD "xyzz" add-2008-s, madd-2008-s, dbl-2008-s-1,
// mdbl-2008-s-1.  Every degenerate case (either operand at infinity, P == Q, P == -Q) is handled:
// MSM inputs in the prover contain zero scalars, repeated CRS points and (0,0) = infinity records
// (reference: packages/backend/libs/src/iotools/mod.rs:2075-2088 pushes G1Affine::zero()).
#pragma once
#include "ff.h"

template <class F>
struct affine_t {
    typename F::E x, y;  // (0,0) encodes infinity (ICICLE / archive convention)
};

template <class F>
struct xyzz_t {
    typename F::E x, y, zz, zzz;
};

template <class F>
struct ec {
    using E = typename F::E;
    using A = affine_t<F>;
    using X = xyzz_t<F>;

    static FF_HD X inf() {
        X r;
        r.x = F::zero();
        r.y = F::zero();
        r.zz = F::zero();
        r.zzz = F::zero();
        return r;
    }
    static FF_HD bool is_inf(const X &p) { return F::is_zero(p.zz); }
    static FF_HD bool is_inf(const A &p) { return F::is_zero(p.x) && F::is_zero(p.y); }

    static FF_HD X from_affine(const A &p) {
        if (is_inf(p)) return inf();
        X r;
        r.x = p.x;
        r.y = p.y;
        r.zz = F::one();
        r.zzz = F::one();
        return r;
    }
    static FF_HD A neg(const A &p) {
        A r;
        r.x = p.x;
        r.y = F::neg(p.y);
        return r;
    }
    static FF_HD X neg(const X &p) {
        X r = p;
        r.y = F::neg(p.y);
        return r;
    }

    // 2 * (affine P), P != infinity: mdbl-2008-s-1 (a = 0)
    static FF_HD X dbl_affine(const A &p) {
        E u = F::dbl(p.y);
        E v = F::sqr(u);
        E w = F::mul(u, v);
        E s = F::mul(p.x, v);
        E xx = F::sqr(p.x);
        E m = F::add(F::dbl(xx), xx);
        X r;
        r.x = F::sub(F::sqr(m), F::dbl(s));
        r.y = F::sub(F::mul(m, F::sub(s, r.x)), F::mul(w, p.y));
        r.zz = v;
        r.zzz = w;
        return r;
    }
    // dbl-2008-s-1 (a = 0)
    static FF_HD X dbl(const X &p) {
        if (is_inf(p)) return p;
        E u = F::dbl(p.y);
        E v = F::sqr(u);
        E w = F::mul(u, v);
        E s = F::mul(p.x, v);
        E xx = F::sqr(p.x);
        E m = F::add(F::dbl(xx), xx);
        X r;
        r.x = F::sub(F::sqr(m), F::dbl(s));
        r.y = F::sub(F::mul(m, F::sub(s, r.x)), F::mul(w, p.y));
        r.zz = F::mul(v, p.zz);
        r.zzz = F::mul(w, p.zzz);
        return r;
    }
    // acc + affine q : madd-2008-s
    static FF_HD X add_mixed(const X &p, const A &q) {
        if (is_inf(q)) return p;
        if (is_inf(p)) return from_affine(q);
        E u2 = F::mul(q.x, p.zz);
        E s2 = F::mul(q.y, p.zzz);
        E pp_ = F::sub(u2, p.x);
        E r_ = F::sub(s2, p.y);
        if (F::is_zero(pp_)) {
            if (F::is_zero(r_)) return dbl_affine(q);
            return inf();
        }
        E pp = F::sqr(pp_);
        E ppp = F::mul(pp_, pp);
        E q_ = F::mul(p.x, pp);
        X r;
        r.x = F::sub(F::sub(F::sqr(r_), ppp), F::dbl(q_));
        r.y = F::sub(F::mul(r_, F::sub(q_, r.x)), F::mul(p.y, ppp));
        r.zz = F::mul(p.zz, pp);
        r.zzz = F::mul(p.zzz, ppp);
        return r;
    }
    // add-2008-s
    static FF_HD X add(const X &p, const X &q) {
        if (is_inf(q)) return p;
        if (is_inf(p)) return q;
        E u1 = F::mul(p.x, q.zz);
        E u2 = F::mul(q.x, p.zz);
        E s1 = F::mul(p.y, q.zzz);
        E s2 = F::mul(q.y, p.zzz);
        E pp_ = F::sub(u2, u1);
        E r_ = F::sub(s2, s1);
        if (F::is_zero(pp_)) {
            if (F::is_zero(r_)) return dbl(p);
            return inf();
        }
        E pp = F::sqr(pp_);
        E ppp = F::mul(pp_, pp);
        E q_ = F::mul(u1, pp);
        X r;
        r.x = F::sub(F::sub(F::sqr(r_), ppp), F::dbl(q_));
        r.y = F::sub(F::mul(r_, F::sub(q_, r.x)), F::mul(s1, ppp));
        r.zz = F::mul(F::mul(p.zz, q.zz), pp);
        r.zzz = F::mul(F::mul(p.zzz, q.zzz), ppp);
        return r;
    }
    // XYZZ -> affine (one field inversion); infinity -> (0,0)
    static FF_HD A to_affine(const X &p) {
        A r;
        if (is_inf(p)) {
            r.x = F::zero();
            r.y = F::zero();
            return r;
        }
        E iz3 = F::inv(p.zzz);                      // 1/ZZZ
        E iz2 = F::sqr(F::mul(iz3, p.zz));          // (ZZ/ZZZ)^2 = 1/ZZ   (ZZ^3 = ZZZ^2)
        r.x = F::mul(p.x, iz2);
        r.y = F::mul(p.y, iz3);
        return r;
    }
    // [k]P, k = nl little-endian u32 limbs (plain integer), MSB-first double-and-add
    static FF_HD X scalar_mul(const uint32_t *k, int nl, const X &p) {
        X acc = inf();
        for (int i = nl * 32 - 1; i >= 0; i--) {
            acc = dbl(acc);
            if ((k[i >> 5] >> (i & 31)) & 1) acc = add(acc, p);
        }
        return acc;
    }
};

using G1 = ec<Fq>;
using g1_affine_t = affine_t<Fq>;
using g1_xyzz_t = xyzz_t<Fq>;
