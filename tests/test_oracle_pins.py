"""Pins the CPU oracle (oracle/tk_oracle.c) against (1) every constant the reference fixes for the
path (tests/golden/pins.json, made by tests/golden/make_pins.py from the reference files) and (2) an
independent pure-Python restatement (tests/pyref.py).  The reference holds NO known-answer vectors
for raw MSM / NTT outputs (libs/src/tests.rs is self-consistency only): raw outputs are
"parity unpinned"; the properties the reference itself asserts are asserted here on the oracle.
"""
import json
import os
import random

import numpy as np
import pytest

import pyref

PINS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "pins.json")))


def test_scalar_modulus_is_r1cs_prime(oracle):
    assert int(PINS["r1cs_prime"], 16) == oracle.R_MOD == pyref.R
    assert PINS["r1cs_files"] == 14
    # -1 round-trips, r reduces to 0 (import reduces mod r)
    m1 = oracle.to_bytes([oracle.R_MOD - 1], 32)
    one = oracle.to_bytes([1], 32)
    assert oracle.to_ints(oracle.fr_add(m1, one), 32) == [0]


def test_generator_limbs_match_reference(oracle):
    # setup/mpc-setup/src/conversions.rs:55-79: in-memory encoding = plain LE u32 limbs
    g = oracle.g1_generator()
    limbs = np.frombuffer(g.tobytes(), "<u4")
    assert list(limbs[:12]) == PINS["g1_generator_x_limbs"]
    assert list(limbs[12:]) == PINS["g1_generator_y_limbs"]
    assert oracle.g1_on_curve(g)
    x, y = oracle.to_ints(g, 48)
    assert (y * y - x * x * x - 4) % pyref.P == 0
    # well-known standard generator x coordinate
    assert x == 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB


def test_fixed_tau_generator_on_curve(oracle):
    # setup/trusted-setup/src/main.rs:71-74
    x, y = int(PINS["fixed_tau_g1_x"], 16), int(PINS["fixed_tau_g1_y"], 16)
    p = oracle.to_bytes([x, y], 48)
    assert oracle.g1_on_curve(p)
    # prime-order subgroup: [r]P = O
    assert pyref.ec_mul(pyref.R, (x, y)) is None
    rm1 = oracle.to_bytes([oracle.R_MOD - 1], 32)
    assert (oracle.g1_scalar_mul(rm1, p) == oracle.g1_neg(p)).all()
    for k in ("x", "y", "alpha", "gamma", "delta", "eta"):
        assert int(PINS["tau_" + k], 16) < oracle.R_MOD


def test_root_of_unity_convention(oracle):
    w32 = oracle.to_ints(oracle.root_of_unity(1 << 32), 32)[0]
    # the declared convention (include/tkmk.h: TKMK_BLS12_381_FR_ROOT_GENERATOR, an INFERENCE — "parity unpinned"): with the
    # declared generator 5 this is ffjavascript's root; tests/test_root_convention.py runs the 7-based alternative
    assert w32 == pyref.ROOT32 == pow(pyref.root_generator(), (pyref.R - 1) >> 32, pyref.R)
    if pyref.root_generator() == 5:
        assert w32 == 0x0212D79E5B416B6F0FD56DC8D168D6C0C4024FF270B3E0941B788F500B912F1F
    assert pow(w32, 1 << 32, pyref.R) == 1 and pow(w32, 1 << 31, pyref.R) == pyref.R - 1
    for n in (1, 2, 8, 256, 4096, 1 << 23):
        w = oracle.to_ints(oracle.root_of_unity(n), 32)[0]
        assert w == pyref.root_of_unity(n)
    # non power of two rounds up (ICICLE get_root_of_unity(max_size))
    assert (oracle.root_of_unity(5) == oracle.root_of_unity(8)).all()


def test_field_ops_vs_python(oracle):
    rnd = random.Random(7)
    n = 64
    for mod, width, add, sub, mul, inv in (
        (pyref.R, 32, oracle.fr_add, oracle.fr_sub, oracle.fr_mul, oracle.fr_inv),
        (pyref.P, 48, oracle.fq_add, oracle.fq_sub, oracle.fq_mul, oracle.fq_inv),
    ):
        a = [rnd.randrange(mod) for _ in range(n)]
        b = [rnd.randrange(mod) for _ in range(n)]
        a[:4] = [0, 1, mod - 1, mod - 2]
        b[:4] = [mod - 1, mod - 1, mod - 1, 0]
        A, B = oracle.to_bytes(a, width), oracle.to_bytes(b, width)
        assert oracle.to_ints(add(A, B), width) == [(x + y) % mod for x, y in zip(a, b)]
        assert oracle.to_ints(sub(A, B), width) == [(x - y) % mod for x, y in zip(a, b)]
        assert oracle.to_ints(mul(A, B), width) == [(x * y) % mod for x, y in zip(a, b)]
        assert oracle.to_ints(inv(A), width) == [pow(x, mod - 2, mod) for x in a]


def test_random_stream_is_deterministic(oracle):
    a = oracle.fr_random(0x746F6B616D616B00, 8)
    b = oracle.fr_random(0x746F6B616D616B00, 4, first=4)
    assert (a[128:] == b).all()
    assert all(v < oracle.R_MOD for v in oracle.to_ints(a, 32))
    # splitmix64 (first output for seed 0 is 0xE220A8397B1DCDAF), 4 outputs per element, LE limbs, mod r
    M = (1 << 64) - 1

    def sm(seed, idx):
        z = (seed + (idx + 1) * 0x9E3779B97F4A7C15) & M
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        return z ^ (z >> 31)
    assert sm(0, 0) == 0xE220A8397B1DCDAF
    for seed in (0, 0x746F6B616D616B00):
        want = [sum(sm(seed, 4 * i + k) << (64 * k) for k in range(4)) % oracle.R_MOD for i in range(5)]
        assert oracle.to_ints(oracle.fr_random(seed, 5), 32) == want


def test_group_law_vs_python(oracle):
    rnd = random.Random(11)
    g = oracle.g1_generator()
    gx, gy = oracle.to_ints(g, 48)
    for _ in range(4):
        a, b = rnd.randrange(pyref.R), rnd.randrange(pyref.R)
        pa = oracle.g1_scalar_mul(oracle.to_bytes([a], 32), g)
        pb = oracle.g1_scalar_mul(oracle.to_bytes([b], 32), g)
        assert tuple(oracle.to_ints(pa, 48)) == pyref.ec_mul(a, (gx, gy))
        s = oracle.g1_add(pa, pb)
        assert tuple(oracle.to_ints(s, 48)) == pyref.ec_mul((a + b) % pyref.R, (gx, gy))
        assert oracle.g1_on_curve(s)
    # doubling, inverse and identity branches
    two_g = oracle.g1_add(g, g)
    assert tuple(oracle.to_ints(two_g, 48)) == pyref.ec_add((gx, gy), (gx, gy))
    zero = np.zeros(96, np.uint8)
    assert (oracle.g1_add(g, oracle.g1_neg(g)) == zero).all()
    assert (oracle.g1_add(g, zero) == g).all() and (oracle.g1_add(zero, g) == g).all()
    assert (oracle.g1_scalar_mul(oracle.to_bytes([0], 32), g) == zero).all()


def test_msm_pippenger_equals_naive_and_python(oracle):
    n = 37
    s = oracle.fr_random(101, n)
    p = oracle.g1_random_bases(202, n)
    # edge cases: zero scalar, scalar 1, r-1, infinity base, repeated base
    sv = oracle.to_ints(s, 32)
    sv[0], sv[1], sv[2] = 0, 1, oracle.R_MOD - 1
    s = oracle.to_bytes(sv, 32)
    p[96 * 3:96 * 4] = 0
    p[96 * 5:96 * 6] = p[96 * 4:96 * 5]
    naive = oracle.g1_msm_naive(s, p)
    assert (oracle.g1_msm(s, p) == naive).all()
    assert (oracle.g1_msm(s, p, threads=1) == naive).all()
    pts = oracle.to_ints(p, 48)
    pts = [None if (pts[2 * i], pts[2 * i + 1]) == (0, 0) else (pts[2 * i], pts[2 * i + 1]) for i in range(n)]
    assert tuple(oracle.to_ints(naive, 48)) == pyref.pt_to_int(pyref.msm(sv, pts))
    assert (oracle.g1_msm(s[:0], p[:0]) == 0).all()


def test_msm_larger_pippenger_equals_naive(oracle):
    n = 600
    s = oracle.fr_random(5, n)
    p = oracle.g1_random_bases(6, n)
    assert (oracle.g1_msm(s, p) == oracle.g1_msm_naive(s, p)).all()


def test_commit_identity(oracle):
    # encode_poly(P) == [P(tau_x, tau_y)] G  with xy_powers[i*ys + j] = [tau_x^i tau_y^j] G
    # (setup/trusted-setup/src/main.rs:236-246; libs/src/iotools/mod.rs:2075-2099)
    tx, ty = int(PINS["tau_x"], 16), int(PINS["tau_y"], 16)
    g = oracle.to_bytes([int(PINS["fixed_tau_g1_x"], 16), int(PINS["fixed_tau_g1_y"], 16)], 48)
    xs, ys = 5, 3
    mon = [pow(tx, i, pyref.R) * pow(ty, j, pyref.R) % pyref.R for i in range(xs) for j in range(ys)]
    crs = oracle.g1_batch_scalar_mul(oracle.to_bytes(mon, 32), g)
    coeffs = oracle.fr_random(9, xs * ys)
    cv = oracle.to_ints(coeffs, 32)
    val = sum(c * m for c, m in zip(cv, mon)) % pyref.R
    assert (oracle.g1_msm(coeffs, crs) == oracle.g1_scalar_mul(oracle.to_bytes([val], 32), g)).all()


def test_ntt_matches_definition(oracle):
    for n in (1, 2, 4, 16, 64):
        x = oracle.fr_random(300 + n, n)
        xv = oracle.to_ints(x, 32)
        want = pyref.dft(xv)
        assert oracle.to_ints(oracle.ntt(x, n), 32) == want
        assert oracle.to_ints(oracle.dft_naive(x, n), 32) == want
        assert (oracle.ntt(oracle.ntt(x, n), n, inverse=True) == x).all()
    # natural order: NTT of delta_1 is (w^k)_k  (libs/src/tests.rs:1075-1087 ordering pin)
    n = 8
    e1 = oracle.to_bytes([0, 1] + [0] * (n - 2), 32)
    w = pyref.root_of_unity(n)
    assert oracle.to_ints(oracle.ntt(e1, n), 32) == [pow(w, k, pyref.R) for k in range(n)]


def test_ntt_coset_is_coefficient_scaling(oracle):
    # libs/src/tests.rs:134-180
    n, g = 32, 0x1234567
    x = oracle.fr_random(77, n)
    xv = oracle.to_ints(x, 32)
    G = oracle.to_bytes([g], 32)
    scaled = oracle.to_bytes([v * pow(g, j, pyref.R) % pyref.R for j, v in enumerate(xv)], 32)
    ev = oracle.ntt(x, n, coset_gen=G)
    assert (ev == oracle.ntt(scaled, n)).all()
    assert oracle.to_ints(ev, 32) == pyref.dft(xv, coset=g)
    assert (oracle.ntt(ev, n, inverse=True, coset_gen=G) == x).all()
    assert oracle.to_ints(oracle.ntt(ev, n, inverse=True, coset_gen=G), 32) == pyref.dft(oracle.to_ints(ev, 32), True, g)


def test_ntt_batch_layouts(oracle):
    n, batch = 16, 5
    x = oracle.fr_random(55, n * batch)
    rows = oracle.ntt(x, n, batch=batch)
    for b in range(batch):
        assert (rows[32 * n * b:32 * n * (b + 1)] == oracle.ntt(x[32 * n * b:32 * n * (b + 1)].copy(), n)).all()
    # columns_batch: element i of vector b at i*batch + b  == transpose route (libs/src/tests.rs:519-588)
    cols = oracle.ntt(x, n, batch=batch, columns_batch=True)
    t = oracle.fr_transpose(x, n, batch)           # (n x batch) -> (batch x n)
    via_t = oracle.fr_transpose(oracle.ntt(t, n, batch=batch), batch, n)
    assert (cols == via_t).all()


def test_bintt_vs_python_and_roundtrip(oracle):
    xs, ys = 8, 4
    cx, cy = 0xABCDEF, 0x13579B
    m = oracle.fr_random(88, xs * ys)
    mv = oracle.to_ints(m, 32)
    CX, CY = oracle.to_bytes([cx], 32), oracle.to_bytes([cy], 32)
    ev = oracle.bintt(m, xs, ys, coset_x=CX, coset_y=CY)
    assert oracle.to_ints(ev, 32) == pyref.bintt(mv, xs, ys, False, cx, cy)
    assert (oracle.bintt(ev, xs, ys, inverse=True, coset_x=CX, coset_y=CY) == m).all()
    plain = oracle.bintt(m, xs, ys)
    # evaluation (i,j) = P(wx^i, wy^j)
    wx, wy = pyref.root_of_unity(xs), pyref.root_of_unity(ys)
    i, j = 3, 2
    val = sum(mv[a * ys + b] * pow(wx, i * a, pyref.R) * pow(wy, j * b, pyref.R) for a in range(xs) for b in range(ys)) % pyref.R
    assert oracle.to_ints(plain, 32)[i * ys + j] == val
    # degenerate axes take the 1-D path with that axis' coset
    v = oracle.fr_random(89, 8)
    assert (oracle.bintt(v, 1, 8, coset_x=CX, coset_y=CY) == oracle.ntt(v, 8, coset_gen=CY)).all()
    assert (oracle.bintt(v, 8, 1, coset_x=CX, coset_y=CY) == oracle.ntt(v, 8, coset_gen=CX)).all()


def test_convolution_theorem(oracle):
    # poly mul via NTT == schoolbook (libs/src/tests.rs:1042-1088)
    n = 16
    a = oracle.to_ints(oracle.fr_random(1, n // 2), 32) + [0] * (n // 2)
    b = oracle.to_ints(oracle.fr_random(2, n // 2), 32) + [0] * (n // 2)
    A, B = oracle.to_bytes(a, 32), oracle.to_bytes(b, 32)
    prod = oracle.ntt(oracle.fr_mul(oracle.ntt(A, n), oracle.ntt(B, n)), n, inverse=True)
    want = [0] * n
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            if x and y:
                want[i + j] = (want[i + j] + x * y) % pyref.R
    assert oracle.to_ints(prod, 32) == want


def test_suffix_product_definition(oracle):
    vals = [3, 5, 7, 11, 13]
    out = oracle.to_ints(oracle.fr_suffix_product(oracle.to_bytes(vals, 32)), 32)
    assert out == [5 * 7 * 11 * 13, 7 * 11 * 13, 11 * 13, 13, 1]


def test_bn254_constants(oracle):
    """BN254 (alt_bn128) has no counterpart in the reference; its oracle instantiation is pinned on the published curve
    constants (EIP-196): p, r, generator (1, 2) on y^2 = x^3 + 3, 2G, r G = O, and on plain Python integer arithmetic."""
    bn = oracle.bn254
    p, r = bn.P_MOD, bn.R_MOD
    assert p == 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
    assert r == 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
    # BN parameterisation: p = 36u^4+36u^3+24u^2+6u+1, r = 36u^4+36u^3+18u^2+6u+1 with u = 4965661367192848881
    u = 4965661367192848881
    assert p == 36 * u**4 + 36 * u**3 + 24 * u**2 + 6 * u + 1 and r == 36 * u**4 + 36 * u**3 + 18 * u**2 + 6 * u + 1
    g = bn.g1_generator()
    assert oracle.to_ints(g, 32) == [1, 2] and bn.g1_on_curve(g)
    two = oracle.to_ints(bn.g1_add(g, g), 32)
    assert two == [1368015179489954701390400359078579693043519447331113978918064868415326638035,
                   9918110051302171585080402603319702774565515993150576347155970296011118125764]
    assert (two[1] ** 2 - two[0] ** 3 - 3) % p == 0
    assert (bn.g1_scalar_mul(oracle.to_bytes([r - 1], 32), g) == bn.g1_neg(g)).all()
    assert not bn.g1_scalar_mul(oracle.to_bytes([0], 32), g).any()

    def add(A, B):
        if A is None or B is None:
            return A or B
        (x1, y1), (x2, y2) = A, B
        if x1 == x2:
            if (y1 + y2) % p == 0:
                return None
            lam = 3 * x1 * x1 * pow(2 * y1, -1, p) % p
        else:
            lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
        x3 = (lam * lam - x1 - x2) % p
        return x3, (lam * (x1 - x3) - y1) % p

    def mul(k, A):
        R = None
        while k:
            if k & 1:
                R = add(R, A)
            A = add(A, A)
            k >>= 1
        return R

    s = bn.fr_random(3, 8)
    pts = bn.g1_random_bases(4, 8)
    acc = None
    for i in range(8):
        k = oracle.to_ints(s[32 * i:32 * i + 32], 32)[0]
        P = tuple(oracle.to_ints(pts[64 * i:64 * i + 64], 32))
        assert (P[1] ** 2 - P[0] ** 3 - 3) % p == 0
        acc = add(acc, mul(k, P))
    assert oracle.to_ints(bn.g1_msm(s, pts), 32) == list(acc)
    assert (bn.g1_msm(s, pts) == bn.g1_msm_naive(s, pts)).all()


def test_g2_instantiation_against_independent_bigint_g2(oracle, tkmk):
    """oracle.g2 (the curve-generic C code over Fp2) against tkmk/g2.py's Python big-int Jacobian arithmetic, which tests/test_g2.py
    pins on the reference's fixed G2 generator (setup/trusted-setup/src/main.rs:75-78) and on the standard generator's order."""
    from tkmk import g2 as big
    import random
    G = oracle.g2
    gen = G.generator()
    assert G.on_curve(gen) and (gen == big.encode(big.STD_G2)).all()
    fixed = big.encode(big.from_hex_pair(PINS["fixed_tau_g2_x"], PINS["fixed_tau_g2_y"]))
    assert G.on_curve(fixed)
    bad = fixed.copy()
    bad[0] ^= 1
    assert not G.on_curve(bad)
    rnd = random.Random(17)
    le = lambda v: np.frombuffer(int(v).to_bytes(32, "little"), np.uint8).copy()
    for base in (big.STD_G2, big.decode(fixed)):
        enc = big.encode(base)
        for k in (1, 2, 3, big.R - 1, rnd.randrange(big.R), rnd.randrange(1 << 64)):
            assert (G.scalar_mul(le(k), enc) == big.encode(big.scalar_mul(k, base))).all(), k
        assert not G.scalar_mul(le(0), enc).any()
        a, b = big.scalar_mul(rnd.randrange(big.R), base), big.scalar_mul(rnd.randrange(big.R), base)
        assert (G.add(big.encode(a), big.encode(b)) == big.encode(big.add(a, b))).all()
        assert (G.add(big.encode(a), big.encode(a)) == big.encode(big.scalar_mul(2, a))).all()        # doubling through add
        assert not G.add(big.encode(a), G.neg(big.encode(a))).any()
    # [r - 1] G = -G (the generator has order r), MSM == naive == big-int sum
    assert (G.scalar_mul(le(big.R - 1), gen) == G.neg(gen)).all()
    n = 40
    s, pts = oracle.fr_random(23, n), G.random_bases(24, n)
    acc = None
    for i in range(n):
        p = big.decode(pts[192 * i:192 * i + 192])
        assert big.on_curve(p)
        acc = big.add(acc, big.scalar_mul(int.from_bytes(bytes(s[32 * i:32 * i + 32]), "little"), p))
    assert (G.msm(s, pts) == big.encode(acc)).all() and (G.msm_naive(s, pts) == big.encode(acc)).all()
