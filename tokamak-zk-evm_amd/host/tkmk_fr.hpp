// tkmk_fr.hpp — host-side arithmetic on single BLS12-381 scalar-field elements (the handful of scalars the prover combines
// between device calls: mixer products, chi^n - 1, omega^-1, kappa powers; packages/backend/prove/src/lib.rs uses the
// ScalarField operators of icicle_bls12_381 for these).  Values are the ABI's plain little-endian u32 limbs; products go
// through a 4 x 64-bit Montgomery multiplication whose constants (-r^-1 mod 2^64, 2^512 mod r) are derived at start-up from
// the modulus alone.  Vectors never come through here — they stay on the device (bls12_381_vector_*).
#pragma once
#include <cstdint>
#include <cstring>
#include <string>

#include "tkmk_base.hpp"

namespace tkmk {
namespace frh {

using u64 = uint64_t;
using u128 = unsigned __int128;

struct U256 {
    u64 l[4];
};
// r = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001 (prime of every committed .r1cs header)
static const U256 MOD = {{0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull}};

inline U256 load(const ScalarField &a) {
    U256 r;
    std::memcpy(r.l, a.limbs, 32);
    return r;
}
inline ScalarField store(const U256 &a) {
    ScalarField r;
    std::memcpy(r.limbs, a.l, 32);
    return r;
}
inline bool geq(const U256 &a, const U256 &b) {
    for (int i = 3; i >= 0; i--)
        if (a.l[i] != b.l[i]) return a.l[i] > b.l[i];
    return true;
}
inline U256 sub_raw(const U256 &a, const U256 &b, u64 *borrow_out = nullptr) {
    U256 r;
    u64 br = 0;
    for (int i = 0; i < 4; i++) {
        u128 t = (u128)a.l[i] - b.l[i] - br;
        r.l[i] = (u64)t;
        br = (u64)(t >> 64) & 1;
    }
    if (borrow_out) *borrow_out = br;
    return r;
}
inline U256 add_mod(const U256 &a, const U256 &b) {
    U256 r;
    u64 c = 0;
    for (int i = 0; i < 4; i++) {
        u128 t = (u128)a.l[i] + b.l[i] + c;
        r.l[i] = (u64)t;
        c = (u64)(t >> 64);
    }
    if (c || geq(r, MOD)) r = sub_raw(r, MOD);
    return r;
}
inline U256 sub_mod(const U256 &a, const U256 &b) {
    u64 br;
    U256 r = sub_raw(a, b, &br);
    if (br) {
        u64 c = 0;
        for (int i = 0; i < 4; i++) {
            u128 t = (u128)r.l[i] + MOD.l[i] + c;
            r.l[i] = (u64)t;
            c = (u64)(t >> 64);
        }
    }
    return r;
}
struct Consts {
    u64 n0;    // -r^-1 mod 2^64
    U256 r2;   // 2^512 mod r
    Consts() {
        u64 inv = 1;
        for (int i = 0; i < 6; i++) inv *= 2 - MOD.l[0] * inv;   // Newton: r^-1 mod 2^64
        n0 = (u64)0 - inv;
        U256 x = {{1, 0, 0, 0}};
        for (int i = 0; i < 512; i++) x = add_mod(x, x);
        r2 = x;
    }
};
inline const Consts &consts() {
    static const Consts c;
    return c;
}
// a * b / 2^256 mod r (CIOS)
inline U256 mont_mul(const U256 &a, const U256 &b) {
    const u64 n0 = consts().n0;
    u64 t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u64 c = 0;
        for (int j = 0; j < 4; j++) {
            u128 s = (u128)a.l[j] * b.l[i] + t[j] + c;
            t[j] = (u64)s;
            c = (u64)(s >> 64);
        }
        u128 s = (u128)t[4] + c;
        t[4] = (u64)s;
        t[5] = (u64)(s >> 64);
        u64 m = t[0] * n0;
        s = (u128)m * MOD.l[0] + t[0];
        c = (u64)(s >> 64);
        for (int j = 1; j < 4; j++) {
            s = (u128)m * MOD.l[j] + t[j] + c;
            t[j - 1] = (u64)s;
            c = (u64)(s >> 64);
        }
        s = (u128)t[4] + c;
        t[3] = (u64)s;
        t[4] = t[5] + (u64)(s >> 64);
    }
    U256 r = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || geq(r, MOD)) r = sub_raw(r, MOD);
    return r;
}

}  // namespace frh

inline ScalarField fr_add(const ScalarField &a, const ScalarField &b) { return frh::store(frh::add_mod(frh::load(a), frh::load(b))); }
inline ScalarField fr_sub(const ScalarField &a, const ScalarField &b) { return frh::store(frh::sub_mod(frh::load(a), frh::load(b))); }
inline ScalarField fr_neg(const ScalarField &a) { return fr_sub(ScalarField{}, a); }
inline ScalarField fr_mul(const ScalarField &a, const ScalarField &b) {
    return frh::store(frh::mont_mul(frh::mont_mul(frh::load(a), frh::load(b)), frh::consts().r2));
}
inline ScalarField fr_one() { return fr_from_u32(1); }
inline ScalarField fr_pow(const ScalarField &a, uint64_t e) {
    ScalarField r = fr_one(), b = a;
    for (; e; e >>= 1) {
        if (e & 1) r = fr_mul(r, b);
        b = fr_mul(b, b);
    }
    return r;
}
// a^(r-2)
inline ScalarField fr_inv(const ScalarField &a) {
    frh::U256 e = frh::sub_raw(frh::MOD, frh::U256{{2, 0, 0, 0}});
    ScalarField r = fr_one(), b = a;
    for (int i = 0; i < 256; i++) {
        if ((e.l[i / 64] >> (i % 64)) & 1) r = fr_mul(r, b);
        b = fr_mul(b, b);
    }
    return r;
}
// ScalarField::from_hex on a HexString (libs/src/iotools/mod.rs:126-146): optional 0x, big-endian digits, reduced mod r
inline ScalarField fr_from_hex(const char *h, size_t len) {
    size_t off = (len >= 2 && h[0] == '0' && (h[1] == 'x' || h[1] == 'X')) ? 2 : 0;
    size_t nd = len - off;
    if (nd > 64) throw Error("hex scalar longer than 32 bytes");
    uint8_t le[32] = {};
    for (size_t k = 0; k < nd; k++) {   // digit k counted from the least significant end
        char c = h[len - 1 - k];
        int v = c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1;
        if (v < 0) throw Error("invalid hex digit in scalar");
        le[k / 2] |= (uint8_t)(v << (4 * (k & 1)));
    }
    frh::U256 v;
    std::memcpy(v.l, le, 32);
    while (frh::geq(v, frh::MOD)) v = frh::sub_raw(v, frh::MOD);
    return frh::store(v);
}
inline ScalarField fr_from_hex(const std::string &h) { return fr_from_hex(h.data(), h.size()); }

}  // namespace tkmk
