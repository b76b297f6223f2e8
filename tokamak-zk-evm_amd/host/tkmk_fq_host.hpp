// tkmk_fq_host.hpp — host-side checks on single G1 points (the six single points of a CRS archive): is (x, y) on
// y^2 = x^3 + 4 over the BLS12-381 base field (the curve of packages/backend/Cargo.toml:23's icicle-bls12-381), or the
// (0, 0) encoding of infinity (libs/src/iotools/mod.rs:1785-1816).  Base-field arithmetic: g2h's 6 x 64-bit Montgomery code.
#pragma once
#include "tkmk_g2.hpp"

namespace tkmk {
namespace fqh {

inline bool g1_on_curve_or_infinity(const G1Affine &p) {
    using namespace g2h;
    uint64_t x[6], y[6];
    std::memcpy(x, p.x.limbs, 48);
    std::memcpy(y, p.y.limbs, 48);
    bool zero_xy = true;
    for (int i = 0; i < 6; i++) zero_xy &= x[i] == 0 && y[i] == 0;
    if (zero_xy) return true;
    if (geq_raw(x, MODQ) || geq_raw(y, MODQ)) return false;   // not canonical field elements
    Fq X = from_plain(x), Y = from_plain(y);
    return mul(Y, Y) == g2h::add(mul(mul(X, X), X), small(4));
}

// a + b for two affine G1 points in the (0, 0) = infinity encoding (G1serde `+`, libs/src/group_structures/mod.rs:895-903): the chord /
// tangent formulas over the base field, one inversion.  For the handful of single-point sums a proof needs (a commitment plus its
// precomputed blinding point); anything longer goes through the MSM.
inline G1Affine g1_affine_add(const G1Affine &p, const G1Affine &q) {
    using namespace g2h;
    auto is_inf = [](const G1Affine &a) {
        static const unsigned char zeros[96] = {0};
        return std::memcmp(&a, zeros, 96) == 0;
    };
    if (is_inf(p)) return q;
    if (is_inf(q)) return p;
    uint64_t t[6];
    auto load = [&](const void *src) {
        std::memcpy(t, src, 48);
        return from_plain(t);
    };
    const Fq x1 = load(p.x.limbs), y1 = load(p.y.limbs), x2 = load(q.x.limbs), y2 = load(q.y.limbs);
    Fq lambda;
    if (x1 == x2) {
        if (!(y1 == y2) || is_zero(y1)) return G1Affine{};                  // q = -p
        const Fq xx = mul(x1, x1);
        lambda = mul(g2h::add(g2h::add(xx, xx), xx), inv(g2h::add(y1, y1)));   // 3 x^2 / 2 y
    } else {
        lambda = mul(sub(y2, y1), inv(sub(x2, x1)));
    }
    const Fq x3 = sub(sub(mul(lambda, lambda), x1), x2), y3 = sub(mul(lambda, sub(x1, x3)), y1);
    G1Affine r{};
    to_plain(x3, t), std::memcpy(r.x.limbs, t, 48);
    to_plain(y3, t), std::memcpy(r.y.limbs, t, 48);
    return r;
}

}  // namespace fqh
}  // namespace tkmk
