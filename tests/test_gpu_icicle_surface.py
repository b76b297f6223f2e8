"""gpu tier: the remaining pieces of the ICICLE surface the reference touches: the opaque polynomial object
(DensePolynomial: libs/src/bivariate_polynomial/mod.rs:112-127,1485-1757,2070), accumulate, and GenerateRandom for scalars and G1
points (prove/src/lib.rs:1040-1080).  Polynomial arithmetic is checked against plain Python integers."""
import ctypes
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class Poly:
    def __init__(self, tk, oracle, coeffs=None, handle=None):
        self.tk, self.o = tk, oracle
        if handle is None:
            handle = ctypes.c_void_p()
            buf = np.asarray(oracle.to_bytes(coeffs, 32)) if coeffs else np.zeros(32, np.uint8)
            tk._check(tk.lib().bls12_381_polynomial_create_from_coefficients(tk._p(buf), ctypes.c_size_t(len(coeffs)), False, ctypes.byref(handle)), "create")
        self.h = handle

    def coeffs(self):
        n = ctypes.c_size_t()
        self.tk._check(self.tk.lib().bls12_381_polynomial_nof_coeffs(self.h, ctypes.byref(n)), "nof")
        out = np.empty(32 * n.value, np.uint8)
        self.tk._check(self.tk.lib().bls12_381_polynomial_copy_coeffs(self.h, ctypes.c_size_t(0), n, self.tk._p(out), False), "copy")
        return self.o.to_ints(out, 32)

    def degree(self):
        d = ctypes.c_int64()
        self.tk._check(self.tk.lib().bls12_381_polynomial_degree(self.h, ctypes.byref(d)), "degree")
        return d.value

    def op(self, name, *others):
        outs = [ctypes.c_void_p() for _ in range(2 if name == "divide" else 1)]
        args = [self.h] + [o.h if isinstance(o, Poly) else o for o in others] + [ctypes.byref(x) for x in outs]
        self.tk._check(getattr(self.tk.lib(), "bls12_381_polynomial_" + name)(*args), name)
        res = [Poly(self.tk, self.o, handle=x) for x in outs]
        return res if len(res) > 1 else res[0]

    def __del__(self):
        try:
            self.tk.lib().bls12_381_polynomial_delete(self.h)
        except Exception:
            pass


def _trim(c):
    c = list(c)
    while c and c[-1] == 0:
        c.pop()
    return c


def test_polynomial_handle(gpu, oracle):
    R = oracle.R_MOD
    rnd = random.Random(3)
    gpu.init_ntt_domain_for_size(1 << 12)
    a = [rnd.randrange(R) for _ in range(300)] + [0, 0]
    b = [rnd.randrange(R) for _ in range(77)]
    A, B = Poly(gpu, oracle, a), Poly(gpu, oracle, b)
    assert A.coeffs() == a and A.degree() == 299 and B.degree() == 76
    assert _trim(A.op("add", B).coeffs()) == _trim([(x + (b[i] if i < len(b) else 0)) % R for i, x in enumerate(a)])
    assert _trim(B.op("subtract", A).coeffs()) == _trim([((b[i] if i < len(b) else 0) - x) % R for i, x in enumerate(a)])
    prod = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            prod[i + j] = (prod[i + j] + x * y) % R
    assert _trim(A.op("multiply", B).coeffs()) == _trim(prod)
    s = rnd.randrange(R)
    sb = np.asarray(oracle.to_bytes([s], 32))
    assert A.op("multiply_by_scalar", gpu._p(sb)).coeffs() == [x * s % R for x in a]
    x = rnd.randrange(R)
    out = np.empty(32, np.uint8)
    gpu._check(gpu.lib().bls12_381_polynomial_evaluate(A.h, gpu._p(np.asarray(oracle.to_bytes([x], 32))), gpu._p(out)), "evaluate")
    assert oracle.to_ints(out, 32)[0] == sum(c * pow(x, i, R) for i, c in enumerate(a)) % R
    assert A.op("slice", ctypes.c_size_t(3), ctypes.c_size_t(7), ctypes.c_size_t(40)).coeffs() == a[3:3 + 7 * 40:7]
    C = A.op("clone")
    assert C.coeffs() == a
    # long division: A = Q * B + Rm with deg Rm < deg B, also through the product back
    Q, Rm = A.op("divide", B)
    q, r = Q.coeffs(), Rm.coeffs()
    assert len(q) == 300 - 77 + 1 and len(r) == 76
    back = [0] * 302
    for i, x in enumerate(q):
        for j, y in enumerate(b):
            back[i + j] = (back[i + j] + x * y) % R
    for i, x in enumerate(r):
        back[i] = (back[i] + x) % R
    assert back == a
    # vanishing-type denominator X^64 - 1 (the legacy div_by_vanishing's shape), numerator of lower degree, constant denominator
    van = [R - 1] + [0] * 63 + [1]
    Qv, Rv = A.op("divide", Poly(gpu, oracle, van))
    qv, rv = Qv.coeffs(), Rv.coeffs()
    chk = [0] * 302
    for i, x in enumerate(qv):
        chk[i + 64] = (chk[i + 64] + x) % R
        chk[i] = (chk[i] - x) % R
    for i, x in enumerate(rv):
        chk[i] = (chk[i] + x) % R
    assert chk == a
    Q0, R0 = B.op("divide", A)
    assert _trim(Q0.coeffs()) == [] and R0.coeffs() == b
    Q1, R1 = A.op("divide", Poly(gpu, oracle, [5]))
    assert Q1.coeffs() == [x * pow(5, R - 2, R) % R for x in a[:300]] and _trim(R1.coeffs()) == []
    with pytest.raises(gpu.TkmkError):
        A.op("divide", Poly(gpu, oracle, [0, 0]))
    # from_rou_evals: inverse NTT
    ev = oracle.fr_random(8, 256)
    h = ctypes.c_void_p()
    gpu._check(gpu.lib().bls12_381_polynomial_create_from_rou_evaluations(gpu._p(np.asarray(ev)), ctypes.c_size_t(256), False, ctypes.byref(h)), "from_rou")
    assert Poly(gpu, oracle, handle=h).coeffs() == oracle.to_ints(oracle.ntt(ev, 256, inverse=True), 32)


def test_accumulate_and_generate_random(gpu, oracle):
    n = 5000
    a, b = oracle.fr_random(1, n), oracle.fr_random(2, n)
    da, db = gpu.DeviceBuffer.from_host(np.asarray(a)), gpu.DeviceBuffer.from_host(np.asarray(b))
    cfg = gpu.lib().tkmk_vecops_default_config()
    cfg.is_a_on_device = cfg.is_b_on_device = True
    gpu._check(gpu.lib().bls12_381_vector_accumulate(gpu._p(da), gpu._p(db), ctypes.c_uint64(n), ctypes.byref(cfg)), "accumulate")
    assert (np.asarray(da.to_host()) == np.asarray(oracle.fr_add(a, b))).all()
    ah = np.asarray(a).copy()
    cfg2 = gpu.lib().tkmk_vecops_default_config()
    gpu._check(gpu.lib().bls12_381_vector_accumulate(gpu._p(ah), gpu._p(np.asarray(b)), ctypes.c_uint64(n), ctypes.byref(cfg2)), "accumulate")
    assert (ah == np.asarray(oracle.fr_add(a, b))).all()
    # GenerateRandom: scalars below r, not all equal; points on the curve and in the prime-order subgroup
    sc = np.empty(32 * 64, np.uint8)
    gpu._check(gpu.lib().bls12_381_generate_scalars(gpu._p(sc), ctypes.c_size_t(64)), "generate_scalars")
    vals = oracle.to_ints(sc, 32)
    assert all(v < oracle.R_MOD for v in vals) and len(set(vals)) == 64
    pts = np.empty(96 * 16, np.uint8)
    gpu._check(gpu.lib().bls12_381_generate_random_affine_points(gpu._p(pts), ctypes.c_size_t(16)), "generate_points")
    rm1 = oracle.to_bytes([oracle.R_MOD - 1], 32)
    for k in range(16):
        p = np.ascontiguousarray(pts[96 * k:96 * (k + 1)])
        assert p.any() and oracle.g1_on_curve(p)
        assert (np.asarray(oracle.g1_scalar_mul(rm1, p)) == np.asarray(oracle.g1_neg(p))).all()      # [r - 1]P = -P
    assert len({bytes(pts[96 * k:96 * (k + 1)]) for k in range(16)}) == 16
