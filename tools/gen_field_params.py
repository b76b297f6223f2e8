#!/usr/bin/env python3
"""Emits tokamak-zk-evm_amd/csrc/field_params.h: Montgomery constants (32-bit limbs) for the fields
on the hot path.  Only the moduli, curve constants and the root-of-unity convention are inputs; every
other number is derived here."""
import os

FIELDS = {
    # BLS12-381 (the reference's only curve: packages/backend/Cargo.toml:23)
    "bls12_381_fr": dict(mod=0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001, two_adicity=32, qnr=5, wu=28),
    "bls12_381_fq": dict(mod=0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB),
    # BN254 (second instantiation named by BASELINE.json configs; no counterpart in the reference)
    # experiment: Fr with the tightest unsaturated form (9 x 29 bits = 261 bits, only 6 spare bits: operands must stay below
    # a few r; csrc/diag.hip kind 6 measures its product against the saturated one)
    "bls12_381_fr9": dict(mod=0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001, wu=29, lu=9),
    "bn254_fr": dict(mod=21888242871839275222246405745257275088548364400416034343698204186575808495617, two_adicity=28, qnr=5, wu=28),
    # 28-bit limbs: with 29 the top limb of 2p would be empty (10 limbs start at bit 261 > 254) and the dominating-limb
    # subtraction constants of ffu::sub<K> need a non-zero top limb
    "bn254_fq": dict(mod=21888242871839275222246405745257275088696311157297823662689037894645226208583, wu=28),
}


def limbs(v, n):
    return ", ".join("0x%08xu" % ((v >> (32 * i)) & 0xFFFFFFFF) for i in range(n))


MAX_MACS = 13  # per asm statement: 2 + 2*13 = 28 operands (clang allows 30)


def gen_mul(n):
    """Straight-line product-scanning Montgomery product for n limbs.  One asm statement per run of limb
    products of a column (each product = v_mad_u64_u32 into {acc} + v_addc_co_u32 into c2), so hipcc's
    one-wait-state pad after every inline-asm statement is paid per column, not per product."""
    L = []

    def emit(macs):
        # macs: list of (kind, x_expr, y_expr); kind 'v' -> y in VGPR, 's' -> y in SGPR
        for i in range(0, len(macs), MAX_MACS):
            chunk = macs[i:i + MAX_MACS]
            lines, ins = [], []
            for t, (kind, x, y) in enumerate(chunk):
                lines.append("v_mad_u64_u32 %%0, vcc, %%%d, %%%d, %%0\\n\\tv_addc_co_u32 %%1, vcc, 0, %%1, vcc" % (2 + 2 * t, 3 + 2 * t))
                ins.append('"v"(%s), "%s"(%s)' % (x, kind, y))
            L.append('        asm("%s" : "+v"(acc), "+v"(c2) : %s : "vcc");' % ("\\n\\t".join(lines), ", ".join(ins)))

    for k in range(2 * n):
        macs = []
        lo_j = max(0, k - n + 1)
        for j in range(lo_j, min(k, n - 1) + 1):
            macs.append(("v", "a[%d]" % j, "b[%d]" % (k - j)))
        for j in range(lo_j, min(k - 1, n - 1) + 1):
            macs.append(("s", "m[%d]" % j, "MOD[%d]" % (k - j)))
        if k == 2 * n - 1:
            macs = []  # top column: only the carried-in words (value < 2p < 2^(32n))
        if macs:
            emit(macs)
        if k < n:
            L.append("        m[%d] = INV == 0xffffffffu ? 0u - (uint32_t)acc : (uint32_t)acc * INV;" % k)
            emit([("s", "m[%d]" % k, "MOD[0]")])
        else:
            L.append("        r[%d] = (uint32_t)acc;" % (k - n))
        if k < 2 * n - 1:
            L.append("        acc = (acc >> 32) | ((uint64_t)c2 << 32);")
            L.append("        c2 = 0;")
    return L


def gen_linear(n, p):
    """Carry-chain add / sub / conditional-subtract mod p as single inline-asm statements with the modulus limbs
    as VOP2 literals.  No v_cndmask (about 16 cycles per wave on gfx950 when masked by VCC): the select is
    r = t + (p & -borrow)."""
    P = [(p >> (32 * i)) & 0xFFFFFFFF for i in range(n)]
    L = []
    rs = ", ".join('"+v"(r[%d])' % i for i in range(n))
    bs = ", ".join('"v"(b[%d])' % i for i in range(n))

    def sub_p(t):  # r -= p, borrow in vcc.  A literal and the VCC carry-in cannot share one instruction
        out = []   # (constant-bus limit 1 on gfx9-family VOP2), so limbs 1.. go through the scratch VGPR t.
        for i in range(n):
            if i == 0:
                out.append("v_subrev_co_u32 %%%d, vcc, 0x%08x, %%%d" % (i, P[i], i))
            else:
                out.append("v_mov_b32 %%%d, 0x%08x" % (t, P[i]))
                out.append("v_subb_co_u32 %%%d, vcc, %%%d, %%%d, vcc" % (i, i, t))
        return out

    def add_masked_p(m, t):  # r += p & m   (m, t = operand numbers of mask and scratch)
        out = []
        for i in range(n):
            out.append("v_and_b32 %%%d, 0x%08x, %%%d" % (t, P[i], m))
            op = "v_add_co_u32 %%%d, vcc, %%%d, %%%d" if i == 0 else "v_addc_co_u32 %%%d, vcc, %%%d, %%%d, vcc"
            out.append(op % (i, t, i))
        return out

    def stmt(name, args, lines, outs, ins):
        body = "\\n\\t".join(lines)
        L.append("    static __device__ __forceinline__ void %s(%s) {" % (name, args))
        L.append("        uint32_t m, t;")
        L.append('        asm("%s" : %s, "=&v"(m), "=&v"(t) : %s : "vcc");' % (body, outs, ins) if ins else
                 '        asm("%s" : %s, "=&v"(m), "=&v"(t) : : "vcc");' % (body, outs))
        L.append("    }")

    m, t = n, n + 1          # operand numbers (no b operands)
    # r in [0, 2p) -> [0, p)
    lines = sub_p(t) + ["v_subb_co_u32 %%%d, vcc, %%%d, %%%d, vcc" % (m, m, m)] + add_masked_p(m, t)
    stmt("reduce_once_asm", "uint32_t *r", lines, rs, "")
    mb, tb, b0 = n, n + 1, n + 2   # with b operands after m, t
    add = []
    for i in range(n):
        op = "v_add_co_u32 %%%d, vcc, %%%d, %%%d" if i == 0 else "v_addc_co_u32 %%%d, vcc, %%%d, %%%d, vcc"
        add.append(op % (i, i, b0 + i))
    lines = add + sub_p(tb) + ["v_subb_co_u32 %%%d, vcc, %%%d, %%%d, vcc" % (mb, mb, mb)] + add_masked_p(mb, tb)
    stmt("add_mod_asm", "uint32_t *r, const uint32_t *b", lines, rs, bs)
    sub = []
    for i in range(n):
        op = "v_sub_co_u32 %%%d, vcc, %%%d, %%%d" if i == 0 else "v_subb_co_u32 %%%d, vcc, %%%d, %%%d, vcc"
        sub.append(op % (i, i, b0 + i))
    lines = sub + ["v_subb_co_u32 %%%d, vcc, %%%d, %%%d, vcc" % (mb, mb, mb)] + add_masked_p(mb, tb)
    stmt("sub_mod_asm", "uint32_t *r, const uint32_t *b", lines, rs, bs)
    return L


WU_DEFAULT = 29  # limb width of the unsaturated (carry-free product) representation


def limbs_u(v, n, WU):
    """WU-bit digits; the top limb takes whatever is left"""
    out = []
    for i in range(n):
        out.append(v & ((1 << WU) - 1) if i < n - 1 else v)
        v >>= WU
    assert out[-1] < (1 << 31)
    return ", ".join("0x%08xu" % x for x in out)


def sub_const(k, p, n, WU):
    """K*p written with limbs that dominate any strict operand: c_0 = d_0 + 2^29, c_i = d_i + 2^29 - 1, c_top = d_top - 1"""
    v = k * p
    d = []
    for i in range(n):
        d.append(v & ((1 << WU) - 1) if i < n - 1 else v)
        v >>= WU
    c = [d[0] + (1 << WU)] + [d[i] + (1 << WU) - 1 for i in range(1, n - 1)] + [d[n - 1] - 1]
    assert sum(ci << (WU * i) for i, ci in enumerate(c)) == k * p and c[-1] > 0
    return c


def gen_unsat(name, p, WU, L_override=None):
    """constants of the radix-2^WU representation (csrc/ffu.h)"""
    nsat = (p.bit_length() + 31) // 32
    L = L_override or -(-(p.bit_length() + 24) // WU)   # radix >= 2^24 p: a product of operands below 2^10 p stays below 1.07 p
    assert (2 * p) >> (WU * (L - 1)) >= 2, "top limb of 2p must be non-zero (sub<K> constants)"
    Ru = 1 << (WU * L)
    Rs = 1 << (32 * nsat)
    o = []
    o.append("    // ---- unsaturated representation: %d limbs of %d bits, Montgomery radix 2^%d (csrc/ffu.h) ----" % (L, WU, WU * L))
    o.append("    static constexpr int LU = %d;" % L)
    o.append("    static constexpr int WU = %d;" % WU)
    o.append("    static constexpr uint32_t INVU = 0x%08xu;   // -p^-1 mod 2^WU" % ((-pow(p, -1, 1 << WU)) % (1 << WU)))
    o.append("    static constexpr uint32_t PINVU = 0x%08xu;  // p^-1 mod 2^WU" % pow(p, -1, 1 << WU))
    o.append("    static constexpr uint32_t MODU[%d] = {%s};" % (L, limbs_u(p, L, WU)))
    o.append("    static constexpr uint32_t ONEU[%d] = {%s};   // 2^%d mod p" % (L, limbs_u(Ru % p, L, WU), WU * L))
    o.append("    static constexpr uint32_t RSATU[%d] = {%s};  // 2^%d mod p: mulU(v, RSATU) turns x*2^%d into x*2^%d" % (L, limbs_u(Rs % p, L, WU), 32 * nsat, WU * L, 32 * nsat))
    o.append("    // saturated-side constants: mont_mul(x_plain, KSAT) = x*2^%d mod p;  mont_mul(x*2^%d, KSATM) = x*2^%d mod p" % (WU * L, 32 * nsat, WU * L))
    o.append("    static constexpr uint32_t KSAT[%d] = {%s};" % (nsat, limbs((Ru * Rs) % p, nsat)))
    o.append("    static constexpr uint32_t KSATM[%d] = {%s};" % (nsat, limbs(Ru % p, nsat)))
    for k in (2, 4, 8, 16):
        c = sub_const(k, p, L, WU)
        o.append("    static constexpr uint32_t SUB%d[%d] = {%s};  // %d*p, dominating limbs; top limb %d" % (k, L, ", ".join("0x%08xu" % x for x in c), k, c[-1]))
    o.append("    static constexpr uint64_t TOP_PER_P_X1024 = %dull;  // floor(1024 * p / 2^%d): top-limb growth per multiple of p" % ((1024 * p) >> (WU * (L - 1)), WU * (L - 1)))
    return o


def main():
    out = ["// GENERATED by tools/gen_field_params.py — do not edit.", "#pragma once", "#include <stdint.h>", ""]
    for name, f in FIELDS.items():
        p = f["mod"]
        n = (p.bit_length() + 31) // 32
        assert p.bit_length() <= 32 * n - 1, "one spare bit required by the CIOS bound"
        R = (1 << (32 * n)) % p
        out.append("struct %s_params {" % name)
        out.append("    static constexpr int N = %d;" % n)
        out.append("    static constexpr int BITS = %d;" % p.bit_length())
        out.append("    static constexpr uint32_t INV = 0x%08xu;  // -p^-1 mod 2^32" % ((-pow(p, -1, 1 << 32)) % (1 << 32)))
        out.append("    static constexpr uint32_t MOD[%d] = {%s};" % (n, limbs(p, n)))
        out.append("    static constexpr uint32_t MOD2[%d] = {%s};  // 2p" % (n, limbs(2 * p, n)))
        out.append("    static constexpr uint32_t ONE[%d] = {%s};  // R mod p" % (n, limbs(R, n)))
        out.append("    static constexpr uint32_t R2[%d] = {%s};  // R^2 mod p" % (n, limbs(R * R % p, n)))
        if "two_adicity" in f:
            s = f["two_adicity"]
            assert (p - 1) % (1 << s) == 0 and pow(f["qnr"], (p - 1) // 2, p) == p - 1
            out.append("    static constexpr int TWO_ADICITY = %d;   // the 2^%d-th root of unity is derived at run time from the declared generator (csrc/ntt_impl.inc)" % (s, s))
        out.extend(gen_unsat(name, p, f.get("wu", WU_DEFAULT), f.get("lu")))
        out.append("#if defined(__HIP_DEVICE_COMPILE__)")
        out.append("    // r = a*b/2^(32N) in [0, 2p): product-scanning Montgomery product, 1 v_mad_u64_u32 + 1 v_addc_co_u32")
        out.append("    // per limb product, modulus limbs in SGPRs (see tools/gen_field_params.py:gen_mul)")
        out.append("    static __device__ __forceinline__ void mul_ps(const uint32_t *a, const uint32_t *b, uint32_t *r) {")
        out.append("        uint32_t m[%d];" % n)
        out.append("        uint64_t acc = 0;")
        out.append("        uint32_t c2 = 0;")
        out.extend(gen_mul(n))
        out.append("    }")
        out.append("    // r = (r + b) mod p, r = (r - b) mod p for r, b < p;  r in [0,2p) -> [0,p)")
        out.extend(gen_linear(n, p))
        out.append("#endif")
        out.append("};")
        out.append("")
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tokamak-zk-evm_amd", "csrc", "field_params.h")
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    main()
