// hostcheck_unsat.h — TEST-ONLY: the unsaturated field / curve path of the MSM (csrc/ffu.h, ec_u.h) on the host with every
// bound assertion enabled; instantiated per curve in hostcheck_unsat_bls12_381.cpp / hostcheck_unsat_bn254.cpp so that the
// two heavy translation units compile in parallel.
#pragma once
#include "hostcheck_common.h"
#include "ec_u.h"

// plain Fq values in / out; op 0 mul, 1 sqr, 2 add, 3 sub<2> (a - b mod p), 4 pack(unpack) round trip
template <class P>
static void fqu_op(int op, const uint8_t *a, const uint8_t *b, uint8_t *o, size_t n) {
    using F = ff<P>;
    using FU = ffu<P>;
    const size_t sz = 4 * F::N;
    typename F::E ksat;
    for (int i = 0; i < F::N; i++) ksat.l[i] = P::KSAT[i];
    for (size_t i = 0; i < n; i++) {
        typename F::E x, y;
        load<F>(x, a + sz * i);
        load<F>(y, b + sz * i);
        // plain -> x 2^(W L) mod p (what k_convert_bases does) -> strict limbs
        typename FU::E ux = FU::from_packed(F::mul(x, ksat)), uy = FU::from_packed(F::mul(y, ksat));
        typename FU::E r;
        switch (op) {
            case 0: r = FU::mul(ux, uy); break;
            case 1: r = FU::sqr(ux); break;
            case 2: r = FU::add(ux, uy); break;
            case 3: r = FU::template sub<2>(ux, uy); break;
            default: r = ux;
        }
        // back: x 2^(W L) (redundant) -> saturated Montgomery -> plain
        store<F>(o + sz * i, F::from_mont(FU::to_sat_mont(r)));
    }
}
// chain: acc = P0; acc += P_i (signed) for every following point, through the unsaturated mixed add; result affine plain
template <class P>
static void g1u_accumulate(const uint8_t *pts, const uint8_t *negate, size_t n, uint8_t *out) {
    using F = ff<P>;
    using GU = ecu<P>;
    using G = ec<F>;
    const size_t sz = 4 * F::N;
    typename F::E ksat;
    for (int i = 0; i < F::N; i++) ksat.l[i] = P::KSAT[i];
    typename GU::X acc = GU::inf();
    for (size_t i = 0; i < n; i++) {
        affine_t<F> rec;
        load<F>(rec.x, pts + 2 * sz * i);
        load<F>(rec.y, pts + 2 * sz * i + sz);
        if (!(F::is_zero(rec.x) && F::is_zero(rec.y))) {
            rec.x = F::mul(rec.x, ksat);
            rec.y = F::mul(rec.y, ksat);
        }
        typename GU::A q;
        if (!GU::load_affine(q, rec)) continue;
        if (negate[i]) q = GU::neg(q);
        acc = GU::add_mixed(acc, q);
    }
    // both ways out of the unsaturated form must agree: direct conversion, and the multiplication-free raw record the
    // accumulate kernel stores + the reader-side conversion (ec_u.h to_raw / raw_to_sat)
    affine_t<F> r = G::to_affine(GU::to_sat(acc));
    affine_t<F> r2 = G::to_affine(GU::raw_to_sat(GU::to_raw(acc)));
    if (!(F::eq(r.x, r2.x) && F::eq(r.y, r2.y))) memset(&r, 0xff, sizeof r);   // poison the output: the test will fail
    store<F>(out, F::from_mont(r.x));
    store<F>(out + sz, F::from_mont(r.y));
}
// general add / doubling on the unsaturated form (the combine / bucket-reduction kernels): out = affine plain of
//   mode 0: A + B      mode 1: 2A      mode 2: A + A' (A' = the same point reached by another addition order: doubling branch)
//   mode 3: A + (-A')  (infinity)      mode 4: [k] A by double-and-add (k = 16-bit)
// where A = p0 + p1 + p2 and B = p3 + p4 are built with the mixed adder so that ZZ, ZZZ != 1; records go through to_raw / from_raw
template <class P>
static void g1u_full(int mode, const uint8_t *pts, uint32_t k, uint8_t *out) {
    using F = ff<P>;
    using GU = ecu<P>;
    using G = ec<F>;
    const size_t sz = 4 * F::N;
    typename F::E ksat;
    for (int i = 0; i < F::N; i++) ksat.l[i] = P::KSAT[i];
    typename GU::A q[5];
    for (int i = 0; i < 5; i++) {
        affine_t<F> rec;
        load<F>(rec.x, pts + 2 * sz * i);
        load<F>(rec.y, pts + 2 * sz * i + sz);
        rec.x = F::mul(rec.x, ksat);
        rec.y = F::mul(rec.y, ksat);
        GU::load_affine(q[i], rec);
    }
    auto chain = [&](int a, int b, int c) {
        typename GU::X x = GU::add_mixed(GU::inf(), q[a]);
        x = GU::add_mixed(x, q[b]);
        if (c >= 0) x = GU::add_mixed(x, q[c]);
        return GU::from_raw(GU::to_raw(x));
    };
    typename GU::X A = chain(0, 1, 2), B = chain(3, 4, -1), r;
    switch (mode) {
        case 0: r = GU::add(A, B); break;
        case 1: r = GU::dbl(A); break;
        case 2: r = GU::add(A, chain(2, 0, 1)); break;
        case 3: {
            typename GU::X n = chain(1, 2, 0);
            typename GU::A ny;
            ny.x = n.y;
            ny.y = n.y;
            n.y = GU::neg(ny).y;    // p - y (strict); same bound as a canonical y
            r = GU::add(A, n);
            break;
        }
        default: {
            r = GU::inf();
            for (int bit = 15; bit >= 0; bit--) {
                r = GU::dbl(r);
                if ((k >> bit) & 1) r = GU::add(r, A);
            }
        }
    }
    affine_t<F> a = G::to_affine(GU::raw_to_sat(GU::to_raw(r)));
    store<F>(out, F::from_mont(a.x));
    store<F>(out + sz, F::from_mont(a.y));
}
