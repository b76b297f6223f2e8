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

}  // namespace fqh
}  // namespace tkmk
