// tkmk_g2.hpp — host-side G2 arithmetic for Sigma2: work-alike of Sigma2::gen (packages/backend/libs/src/group_structures/mod.rs:752-777):
// H and nine multiples of it by the trapdoor scalars, single-point operations that the reference also does on the CPU.  The C++ twin of
// tkmk/g2.py (same formulas, same 192-byte encoding: x then y, each real part then imaginary part, 48-byte little-endian; (0,0) = infinity).
// Base field: 6 x 64-bit Montgomery arithmetic with constants derived at start-up from the modulus (as tkmk_fr.hpp does for Fr).
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "tkmk_fr.hpp"

namespace tkmk {
namespace g2h {

using u64 = uint64_t;
using u128 = unsigned __int128;
constexpr int N = 6;
struct Fq {
    u64 l[N];   // Montgomery form
    bool operator==(const Fq &o) const { return std::memcmp(l, o.l, sizeof l) == 0; }
};
// p = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
static const u64 MODQ[N] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull,
                            0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};

inline bool geq_raw(const u64 *a, const u64 *b) {
    for (int i = N - 1; i >= 0; i--)
        if (a[i] != b[i]) return a[i] > b[i];
    return true;
}
inline void sub_raw(u64 *r, const u64 *a, const u64 *b, u64 *borrow = nullptr) {
    u64 br = 0;
    for (int i = 0; i < N; i++) {
        u128 t = (u128)a[i] - b[i] - br;
        r[i] = (u64)t;
        br = (u64)(t >> 64) & 1;
    }
    if (borrow) *borrow = br;
}
inline Fq add(const Fq &a, const Fq &b) {
    Fq r;
    u64 c = 0;
    for (int i = 0; i < N; i++) {
        u128 t = (u128)a.l[i] + b.l[i] + c;
        r.l[i] = (u64)t;
        c = (u64)(t >> 64);
    }
    if (c || geq_raw(r.l, MODQ)) sub_raw(r.l, r.l, MODQ);
    return r;
}
inline Fq sub(const Fq &a, const Fq &b) {
    Fq r;
    u64 br;
    sub_raw(r.l, a.l, b.l, &br);
    if (br) {
        u64 c = 0;
        for (int i = 0; i < N; i++) {
            u128 t = (u128)r.l[i] + MODQ[i] + c;
            r.l[i] = (u64)t;
            c = (u64)(t >> 64);
        }
    }
    return r;
}
struct Consts {
    u64 n0;
    Fq r1, r2;   // 2^384 mod p (Montgomery one), 2^768 mod p
    Consts() {
        u64 inv = 1;
        for (int i = 0; i < 6; i++) inv *= 2 - MODQ[0] * inv;
        n0 = (u64)0 - inv;
        Fq x{};
        x.l[0] = 1;
        for (int i = 0; i < 384; i++) x = add(x, x);
        r1 = x;
        for (int i = 0; i < 384; i++) x = add(x, x);
        r2 = x;
    }
};
inline const Consts &consts() {
    static const Consts c;
    return c;
}
inline Fq mul(const Fq &a, const Fq &b) {   // Montgomery product (CIOS)
    const u64 n0 = consts().n0;
    u64 t[N + 2] = {};
    for (int i = 0; i < N; i++) {
        u64 c = 0;
        for (int j = 0; j < N; j++) {
            u128 s = (u128)a.l[j] * b.l[i] + t[j] + c;
            t[j] = (u64)s;
            c = (u64)(s >> 64);
        }
        u128 s = (u128)t[N] + c;
        t[N] = (u64)s;
        t[N + 1] = (u64)(s >> 64);
        u64 m = t[0] * n0;
        s = (u128)m * MODQ[0] + t[0];
        c = (u64)(s >> 64);
        for (int j = 1; j < N; j++) {
            s = (u128)m * MODQ[j] + t[j] + c;
            t[j - 1] = (u64)s;
            c = (u64)(s >> 64);
        }
        s = (u128)t[N] + c;
        t[N - 1] = (u64)s;
        t[N] = t[N + 1] + (u64)(s >> 64);
    }
    Fq r;
    std::memcpy(r.l, t, sizeof r.l);
    if (t[N] || geq_raw(r.l, MODQ)) sub_raw(r.l, r.l, MODQ);
    return r;
}
inline Fq zero() { return Fq{}; }
inline Fq one() { return consts().r1; }
inline bool is_zero(const Fq &a) {
    u64 o = 0;
    for (u64 v : a.l) o |= v;
    return o == 0;
}
inline Fq from_plain(const u64 *limbs) {
    Fq a;
    std::memcpy(a.l, limbs, sizeof a.l);
    return mul(a, consts().r2);
}
inline void to_plain(const Fq &a, u64 *limbs) {
    Fq o{};
    o.l[0] = 1;
    Fq r = mul(a, o);
    std::memcpy(limbs, r.l, sizeof r.l);
}
inline Fq small(u64 v) {
    u64 l[N] = {v};
    return from_plain(l);
}
inline Fq inv(const Fq &a) {   // a^(p-2)
    u64 e[N];
    u64 two[N] = {2};
    sub_raw(e, MODQ, two);
    Fq r = one(), b = a;
    for (int i = 0; i < 64 * N; i++) {
        if ((e[i / 64] >> (i % 64)) & 1) r = mul(r, b);
        b = mul(b, b);
    }
    return r;
}

// Fp2 = Fp[u] / (u^2 + 1)
struct F2 {
    Fq c0, c1;
    bool operator==(const F2 &o) const { return c0 == o.c0 && c1 == o.c1; }
};
inline F2 f2_add(const F2 &a, const F2 &b) { return {add(a.c0, b.c0), add(a.c1, b.c1)}; }
inline F2 f2_sub(const F2 &a, const F2 &b) { return {sub(a.c0, b.c0), sub(a.c1, b.c1)}; }
inline F2 f2_mul(const F2 &a, const F2 &b) {
    Fq t0 = mul(a.c0, b.c0), t1 = mul(a.c1, b.c1);
    return {sub(t0, t1), sub(sub(mul(add(a.c0, a.c1), add(b.c0, b.c1)), t0), t1)};
}
inline F2 f2_sqr(const F2 &a) { return f2_mul(a, a); }
inline F2 f2_scale(const F2 &a, u64 k) {
    Fq s = small(k);
    return {mul(a.c0, s), mul(a.c1, s)};
}
inline F2 f2_inv(const F2 &a) {
    Fq n = inv(add(mul(a.c0, a.c0), mul(a.c1, a.c1)));
    return {mul(a.c0, n), sub(zero(), mul(a.c1, n))};
}
inline bool f2_is_zero(const F2 &a) { return is_zero(a.c0) && is_zero(a.c1); }

struct Affine {
    F2 x, y;
    bool inf = false;
};
struct Jac {
    F2 X, Y, Z;   // Z = 0: infinity
};
inline bool on_curve(const Affine &p) {
    if (p.inf) return true;
    F2 b{small(4), small(4)};   // 4 (1 + u)
    return f2_sqr(p.y) == f2_add(f2_mul(f2_sqr(p.x), p.x), b);
}
inline Jac jac_inf() { return {{one(), zero()}, {one(), zero()}, {zero(), zero()}}; }
inline Jac to_jac(const Affine &p) { return p.inf ? jac_inf() : Jac{p.x, p.y, {one(), zero()}}; }
inline Jac dbl(const Jac &p) {   // dbl-2009-l (a = 0)
    if (f2_is_zero(p.Z)) return p;
    F2 A = f2_sqr(p.X), B = f2_sqr(p.Y), C = f2_sqr(B);
    F2 D = f2_scale(f2_sub(f2_sub(f2_sqr(f2_add(p.X, B)), A), C), 2);
    F2 E = f2_scale(A, 3);
    F2 X3 = f2_sub(f2_sqr(E), f2_scale(D, 2));
    return {X3, f2_sub(f2_mul(E, f2_sub(D, X3)), f2_scale(C, 8)), f2_scale(f2_mul(p.Y, p.Z), 2)};
}
inline Jac add(const Jac &p, const Jac &q) {   // add-2007-bl
    if (f2_is_zero(p.Z)) return q;
    if (f2_is_zero(q.Z)) return p;
    F2 Z1Z1 = f2_sqr(p.Z), Z2Z2 = f2_sqr(q.Z);
    F2 U1 = f2_mul(p.X, Z2Z2), U2 = f2_mul(q.X, Z1Z1);
    F2 S1 = f2_mul(f2_mul(p.Y, q.Z), Z2Z2), S2 = f2_mul(f2_mul(q.Y, p.Z), Z1Z1);
    if (U1 == U2) return S1 == S2 ? dbl(p) : jac_inf();
    F2 H = f2_sub(U2, U1), I = f2_sqr(f2_scale(H, 2)), J = f2_mul(H, I), r = f2_scale(f2_sub(S2, S1), 2), V = f2_mul(U1, I);
    F2 X3 = f2_sub(f2_sub(f2_sqr(r), J), f2_scale(V, 2));
    F2 Y3 = f2_sub(f2_mul(r, f2_sub(V, X3)), f2_scale(f2_mul(S1, J), 2));
    F2 Z3 = f2_mul(f2_sub(f2_sub(f2_sqr(f2_add(p.Z, q.Z)), Z1Z1), Z2Z2), H);
    return {X3, Y3, Z3};
}
inline Affine to_affine(const Jac &p) {
    if (f2_is_zero(p.Z)) {
        Affine a;
        a.inf = true;
        return a;
    }
    F2 zi = f2_inv(p.Z), zi2 = f2_sqr(zi);
    return {f2_mul(p.X, zi2), f2_mul(p.Y, f2_mul(zi2, zi)), false};
}
// [k] pt, k a plain scalar-field element
inline Affine scalar_mul(const ScalarField &k, const Affine &pt) {
    Jac acc = jac_inf(), base = to_jac(pt);
    for (int i = 0; i < 256; i++) {
        if ((k.limbs[i / 32] >> (i % 32)) & 1) acc = add(acc, base);
        base = dbl(base);
    }
    return to_affine(acc);
}
inline std::array<uint8_t, 192> encode(const Affine &p) {
    std::array<uint8_t, 192> out{};
    if (p.inf) return out;
    const Fq *c[4] = {&p.x.c0, &p.x.c1, &p.y.c0, &p.y.c1};
    for (int i = 0; i < 4; i++) {
        u64 l[N];
        to_plain(*c[i], l);
        std::memcpy(out.data() + 48 * i, l, 48);
    }
    return out;
}
inline Affine decode(const uint8_t *rec) {
    Affine p;
    bool any = false;
    for (int i = 0; i < 192; i++) any |= rec[i] != 0;
    if (!any) {
        p.inf = true;
        return p;
    }
    Fq *c[4] = {&p.x.c0, &p.x.c1, &p.y.c0, &p.y.c1};
    for (int i = 0; i < 4; i++) {
        u64 l[N];
        std::memcpy(l, rec + 48 * i, 48);
        if (geq_raw(l, MODQ)) throw Error("G2 coordinate not reduced");
        *c[i] = from_plain(l);
    }
    return p;
}
// G2BaseField::from_hex over the whole 96-byte limb array (setup/trusted-setup/src/main.rs:75-78): low 48 bytes = real part
inline F2 f2_from_hex(const std::string &h) {
    size_t off = (h.size() >= 2 && h[0] == '0' && (h[1] == 'x' || h[1] == 'X')) ? 2 : 0;
    size_t nd = h.size() - off;
    if (nd > 192) throw Error("G2 base-field hex longer than 96 bytes");
    uint8_t le[96] = {};
    for (size_t k = 0; k < nd; k++) {
        char c = h[h.size() - 1 - k];
        int v = c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1;
        if (v < 0) throw Error("invalid hex digit");
        le[k / 2] |= (uint8_t)(v << (4 * (k & 1)));
    }
    u64 lo[N], hi[N];
    std::memcpy(lo, le, 48);
    std::memcpy(hi, le + 48, 48);
    return {from_plain(lo), from_plain(hi)};
}

}  // namespace g2h
}  // namespace tkmk
