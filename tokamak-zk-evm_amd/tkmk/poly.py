"""Device-resident work-alike of the reference's DensePolynomialExt
(packages/backend/libs/src/bivariate_polynomial/mod.rs:112-118, trait BivariatePolynomial :1283-1416).

Same fields (x_size, y_size, x_degree, y_degree), same method names, argument meaning and error behaviour
(the reference panics; here ValueError).  The difference is where the work happens: the reference copies the
coefficient matrix to the host for find_degree / resize / mul_monomial / scale / divisions; here the matrix
stays in HBM and every method is one or a few kernels behind the C ABI (include/tkmk.h, tkmk_poly_*).
Coefficient (ix, iy) lives at ix*y_size + iy.
"""
import ctypes

import numpy as np

import tkmk


def _is_pow2(n):
    return n > 0 and n & (n - 1) == 0


def _find_size_as_twopower(tx, ty):
    # mod.rs:72-86
    if tx == 0 or ty == 0:
        raise ValueError("Invalid target sizes for resize")
    return 1 << (tx - 1).bit_length(), 1 << (ty - 1).bit_length()


def _dev(a):
    return a if isinstance(a, tkmk.DeviceBuffer) else tkmk.DeviceBuffer.from_host(np.ascontiguousarray(a))


def _fr(x):
    return None if x is None else tkmk._p(np.ascontiguousarray(x))


class DensePolynomialExt:
    def __init__(self, buf, x_size, y_size, x_degree, y_degree):
        self.poly = buf            # tkmk.DeviceBuffer of x_size*y_size Fr
        self.x_size, self.y_size = x_size, y_size
        self.x_degree, self.y_degree = x_degree, y_degree

    # ---- constructors (mod.rs:1517-1551, 1615-1644) ----
    @classmethod
    def zero(cls):
        return cls(tkmk.DeviceBuffer.from_host(np.zeros(32, np.uint8)), 1, 1, -1, -1)

    @classmethod
    def from_coeffs(cls, coeffs, x_size, y_size):
        n = tkmk._len(coeffs)
        if x_size * y_size != n:
            raise ValueError("Mismatch between the coefficient vector and the polynomial size")
        if not _is_pow2(x_size) or not _is_pow2(y_size):
            raise ValueError("The input sizes for from_coeffs must be powers of two.")
        return cls(_dev(coeffs), x_size, y_size, x_size - 1, y_size - 1)

    @classmethod
    def from_rou_evals(cls, evals, x_size, y_size, coset_x=None, coset_y=None):
        if not _is_pow2(x_size) or not _is_pow2(y_size):
            raise ValueError("The input sizes for from_rou_evals must be powers of two.")
        coeffs = tkmk.bintt(_dev(evals), x_size, y_size, inverse=True, coset_x=coset_x, coset_y=coset_y)
        return cls.from_coeffs(coeffs, x_size, y_size)

    def to_rou_evals(self, coset_x=None, coset_y=None, out=None):
        """forward _biNTT of the coefficient matrix; no host round trip (the reference makes one: mod.rs:1657-1662)"""
        return tkmk.bintt(self.poly, self.x_size, self.y_size, coset_x=coset_x, coset_y=coset_y, out=out)

    def clone(self):
        d = tkmk.DeviceBuffer(self.poly.nbytes)
        tkmk._check(tkmk.lib().tkmk_memcpy_d2d(tkmk._p(d), tkmk._p(self.poly), ctypes.c_size_t(self.poly.nbytes)), "tkmk_memcpy_d2d")
        return DensePolynomialExt(d, self.x_size, self.y_size, self.x_degree, self.y_degree)

    def copy_coeffs(self):
        return self.poly.to_host(32 * self.x_size * self.y_size)

    def get_coeff(self, ix, iy):
        if not (ix <= self.x_size and iy <= self.y_size):
            raise ValueError("The index at which to get a coefficient exceeds the coefficient size.")
        return self.poly.to_host(32, offset=32 * (ix * self.y_size + iy))

    def degree(self):
        return self.x_degree, self.y_degree

    # ---- bookkeeping (mod.rs:1480-1515, 1784-1844) ----
    def find_degree(self):
        xd, yd = ctypes.c_int64(), ctypes.c_int64()
        tkmk._check(tkmk.lib().tkmk_poly_find_degree(tkmk._p(self.poly), self.x_size, self.y_size, ctypes.byref(xd),
                                                    ctypes.byref(yd), None), "tkmk_poly_find_degree")
        return xd.value, yd.value

    def _placed(self, nx, ny, ox, oy):
        dst = tkmk.DeviceBuffer(32 * nx * ny)
        tkmk._check(tkmk.lib().tkmk_poly_place(tkmk._p(self.poly), self.x_size, self.y_size, tkmk._p(dst), nx, ny, ox, oy, None),
                    "tkmk_poly_place")
        return dst

    def resize(self, target_x_size, target_y_size):
        nx, ny = _find_size_as_twopower(target_x_size, target_y_size)
        if (self.x_size, self.y_size) == (nx, ny):
            return
        self.poly = self._placed(nx, ny, 0, 0)
        self.x_size, self.y_size = nx, ny

    def optimize_size(self):
        xd, yd = self.find_degree()
        self.x_degree, self.y_degree = xd, yd
        if xd + 1 == 0 or yd + 1 == 0:
            return
        self.resize(xd + 1, yd + 1)

    def mul_monomial(self, x_exponent, y_exponent):
        if x_exponent == 0 and y_exponent == 0:
            return self.clone()
        nx, ny = _find_size_as_twopower(self.x_degree + 1 + x_exponent, self.y_degree + 1 + y_exponent)
        if self.x_size + x_exponent > nx or self.y_size + y_exponent > ny:
            raise ValueError("mul_monomial: coefficient block does not fit the target (the reference slice copy panics)")
        return DensePolynomialExt.from_coeffs(self._placed(nx, ny, x_exponent, y_exponent), nx, ny)

    # ---- scaling / evaluation (mod.rs:1553-1613, 1719-1750) ----
    def _scale(self, fx, fy):
        dst = tkmk.DeviceBuffer(self.poly.nbytes)
        tkmk._check(tkmk.lib().tkmk_poly_scale_coeffs(tkmk._p(self.poly), self.x_size, self.y_size, _fr(fx), _fr(fy), tkmk._p(dst), None),
                    "tkmk_poly_scale_coeffs")
        return DensePolynomialExt.from_coeffs(dst, self.x_size, self.y_size)

    def scale_coeffs_x(self, x_factor):
        return self._scale(x_factor, None)

    def scale_coeffs_y(self, y_factor):
        return self._scale(None, y_factor)

    def eval_x(self, x):
        out = tkmk.DeviceBuffer(32 * self.y_size)
        tkmk._check(tkmk.lib().tkmk_poly_eval_x(tkmk._p(self.poly), self.x_size, self.y_size, _fr(x), tkmk._p(out), None), "tkmk_poly_eval_x")
        return DensePolynomialExt.from_coeffs(out, 1, self.y_size)

    def eval_y(self, y):
        out = tkmk.DeviceBuffer(32 * self.x_size)
        tkmk._check(tkmk.lib().tkmk_poly_eval_y(tkmk._p(self.poly), self.x_size, self.y_size, _fr(y), tkmk._p(out), None), "tkmk_poly_eval_y")
        return DensePolynomialExt.from_coeffs(out, self.x_size, 1)

    def eval(self, x, y):
        out = np.empty(32, np.uint8)
        tkmk._check(tkmk.lib().tkmk_poly_eval(tkmk._p(self.poly), self.x_size, self.y_size, _fr(x), _fr(y), tkmk._p(out), None), "tkmk_poly_eval")
        return out

    # ---- arithmetic (mod.rs:532-1281, 1846-1996) ----
    def _same_shape(self, rhs):
        """operands brought to a common (max) shape like the reference's Add/Sub impls do via resize"""
        nx, ny = max(self.x_size, rhs.x_size), max(self.y_size, rhs.y_size)
        a, b = self, rhs
        if (a.x_size, a.y_size) != (nx, ny):
            a = a.clone()
            a.resize(nx, ny)
        if (b.x_size, b.y_size) != (nx, ny):
            b = b.clone()
            b.resize(nx, ny)
        return a, b, nx, ny

    def __add__(self, rhs):
        a, b, nx, ny = self._same_shape(rhs)
        return DensePolynomialExt.from_coeffs(tkmk.vec_add(a.poly, b.poly), nx, ny)

    def __sub__(self, rhs):
        a, b, nx, ny = self._same_shape(rhs)
        return DensePolynomialExt.from_coeffs(tkmk.vec_sub(a.poly, b.poly), nx, ny)

    def scalar_mul(self, scalar):
        return DensePolynomialExt.from_coeffs(tkmk.scalar_mul(tkmk.DeviceBuffer.from_host(scalar), self.poly), self.x_size, self.y_size)

    def mul_scalar(self, scalar):
        """&poly * &scalar / &scalar * &poly (mod.rs:766-966): a clone when the scalar is one"""
        s = np.ascontiguousarray(scalar)
        if s[0] == 1 and not s[1:].any():
            return self.clone()
        return self.scalar_mul(s)

    def _const_term(self, scalar, sub):
        # mod.rs:1042-1116, 1189-1262: only coefficient (0, 0) changes; here a 32-byte device op on the clone
        out = self.clone()
        s = tkmk.DeviceBuffer.from_host(np.ascontiguousarray(scalar))
        head = tkmk.DeviceBuffer(32)
        lib = tkmk.lib()
        tkmk._check(lib.tkmk_memcpy_d2d(tkmk._p(head), tkmk._p(out.poly), ctypes.c_size_t(32)), "tkmk_memcpy_d2d")
        (tkmk.vec_sub if sub else tkmk.vec_add)(head, s, out=head)
        tkmk._check(lib.tkmk_memcpy_d2d(tkmk._p(out.poly), tkmk._p(head), ctypes.c_size_t(32)), "tkmk_memcpy_d2d")
        return out

    def add_scalar(self, scalar):
        return self._const_term(scalar, False)

    def sub_scalar(self, scalar):
        return self._const_term(scalar, True)

    def __neg__(self):
        zero = tkmk.DeviceBuffer.from_host(np.zeros(32, np.uint8))
        return DensePolynomialExt.from_coeffs(tkmk.scalar_sub(zero, self.poly), self.x_size, self.y_size)

    def _mul(self, rhs):
        # mod.rs:1846-1996: degree scan, scalar fast paths, resize to the product box, 2 forward + 1 inverse _biNTT
        lxd, lyd = self.find_degree()
        rxd, ryd = rhs.find_degree()
        if lxd + lyd == 0 and rxd + ryd > 0:
            return rhs.scalar_mul(self.get_coeff(0, 0))
        if rxd + ryd == 0 and lxd + lyd > 0:
            return self.scalar_mul(rhs.get_coeff(0, 0))
        if rxd + ryd == 0 and lxd + lyd == 0:
            prod = tkmk.vec_mul(self.get_coeff(0, 0), rhs.get_coeff(0, 0))
            return DensePolynomialExt.from_coeffs(prod, 1, 1)
        tx, ty = lxd + rxd + 1, lyd + ryd + 1
        a = self.clone()
        a.resize(tx, ty)
        b = rhs.clone()
        b.resize(tx, ty)
        xs, ys = a.x_size, a.y_size
        ea = tkmk.bintt(a.poly, xs, ys)
        eb = tkmk.bintt(b.poly, xs, ys)
        tkmk.vec_mul(ea, eb, out=ea)
        return DensePolynomialExt.from_rou_evals(ea, xs, ys)

    def __mul__(self, rhs):
        return self._mul(rhs)

    # ---- divisions (mod.rs:2284-2477) ----
    def div_by_vanishing_opt(self, denom_x_degree, denom_y_degree):
        c, d = denom_x_degree, denom_y_degree
        if not (_is_pow2(c) and _is_pow2(d)):
            raise ValueError("The denominators must have degress as powers of two.")
        self.optimize_size()
        if self.x_degree < c or self.y_degree < d:
            raise ValueError("The numerator must have grater degrees than denominators.")
        xs, ys = (self.x_size // c) * c, (self.y_size // d) * d
        qx, qy = tkmk.DeviceBuffer(32 * xs * ys), tkmk.DeviceBuffer(32 * c * ys)
        tkmk._check(tkmk.lib().tkmk_poly_div_by_vanishing_opt(tkmk._p(self.poly), xs, ys, c, d, tkmk._p(qx), tkmk._p(qy), None),
                    "tkmk_poly_div_by_vanishing_opt")
        quo_x = DensePolynomialExt.from_coeffs(qx, xs, ys)
        quo_y = DensePolynomialExt.from_coeffs(qy, c, ys)
        quo_x.x_degree, quo_x.y_degree = (xs - c - 1, ys - 1) if xs > c else (-1, -1)
        quo_y.x_degree, quo_y.y_degree = (c - 1, ys - d - 1) if ys > d else (-1, -1)
        return quo_x, quo_y

    def div_by_ruffini(self, x, y):
        qx, qy = tkmk.DeviceBuffer(32 * self.x_size * self.y_size), tkmk.DeviceBuffer(32 * self.y_size)
        r = np.empty(32, np.uint8)
        tkmk._check(tkmk.lib().tkmk_poly_div_by_ruffini(tkmk._p(self.poly), self.x_size, self.y_size, _fr(x), _fr(y), tkmk._p(qx),
                                                       tkmk._p(qy), tkmk._p(r), None), "tkmk_poly_div_by_ruffini")
        return (DensePolynomialExt.from_coeffs(qx, self.x_size, self.y_size), DensePolynomialExt.from_coeffs(qy, 1, self.y_size), r)


def _next_pow2(n):
    return 1 if n <= 1 else 1 << (n - 1).bit_length()


class PolyExpr:
    """Work-alike of the reference's fused expression evaluator (mod.rs:141-436): NTT every distinct leaf once
    (cache keyed by object identity and domain), evaluate the tree with pointwise device ops, one inverse NTT.
    Temporaries are reused in place instead of allocating a fresh 256 MiB buffer per node."""

    def __init__(self, kind, *args):
        self.kind, self.args = kind, args

    poly = classmethod(lambda cls, p: cls("poly", p))
    scalar = classmethod(lambda cls, s: cls("scalar", np.ascontiguousarray(s)))
    add = classmethod(lambda cls, a, b: cls("add", a, b))
    sub = classmethod(lambda cls, a, b: cls("sub", a, b))
    mul = classmethod(lambda cls, a, b: cls("mul", a, b))
    scale = classmethod(lambda cls, s, e: cls("scale", np.ascontiguousarray(s), e))
    mul_x_minus_one = classmethod(lambda cls, e: cls("xm1", e))

    @classmethod
    def weighted_sum(cls, terms):
        return cls("sum", [cls.scale(s, e) for s, e in terms])

    # -- coefficient route (mod.rs:190-218) --
    def evaluate_coeffs(self):
        k, a = self.kind, self.args
        if k == "poly":
            return a[0].clone()
        if k == "scalar":
            return DensePolynomialExt.from_coeffs(a[0], 1, 1)
        if k == "add":
            return a[0].evaluate_coeffs() + a[1].evaluate_coeffs()
        if k == "sub":
            return a[0].evaluate_coeffs() - a[1].evaluate_coeffs()
        if k == "mul":
            return a[0].evaluate_coeffs() * a[1].evaluate_coeffs()
        if k == "scale":
            return a[1].evaluate_coeffs().scalar_mul(a[0])
        if k == "xm1":
            p = a[0].evaluate_coeffs()
            p.optimize_size()
            return p.mul_monomial(1, 0) - p if p.x_degree >= 0 else p
        terms = a[0]
        if not terms:
            return DensePolynomialExt.zero()
        acc = terms[0].evaluate_coeffs()
        for t in terms[1:]:
            acc = acc + t.evaluate_coeffs()
        return acc

    # -- fused route (mod.rs:220-436) --
    def degree_bound(self):
        k, a = self.kind, self.args
        if k == "poly":
            return a[0].find_degree()
        if k == "scalar":
            return (0, 0) if a[0].any() else (-1, -1)
        if k in ("add", "sub"):
            l, r = a[0].degree_bound(), a[1].degree_bound()
            return max(l[0], r[0]), max(l[1], r[1])
        if k == "mul":
            l, r = a[0].degree_bound(), a[1].degree_bound()
            return (-1, -1) if min(l + r) < 0 else (l[0] + r[0], l[1] + r[1])
        if k == "scale":
            return a[1].degree_bound() if a[0].any() else (-1, -1)
        if k == "xm1":
            d = a[0].degree_bound()
            return (-1, -1) if min(d) < 0 else (d[0] + 1, d[1])
        out = (-1, -1)
        for t in a[0]:
            d = t.degree_bound()
            out = (max(out[0], d[0]), max(out[1], d[1]))
        return out

    def evaluate_fused(self):
        xd, yd = self.degree_bound()
        return self.evaluate_fused_with_domain(_next_pow2(xd + 1) if xd >= 0 else 1, _next_pow2(yd + 1) if yd >= 0 else 1)

    def evaluate_fused_with_domain(self, target_x_size, target_y_size):
        if not _is_pow2(target_x_size) or not _is_pow2(target_y_size):
            raise ValueError("Fused polynomial expression domains must be powers of two.")
        xd, yd = self.degree_bound()
        if (_next_pow2(xd + 1) if xd >= 0 else 1) > target_x_size or (_next_pow2(yd + 1) if yd >= 0 else 1) > target_y_size:
            raise ValueError("Fused polynomial expression domain is too small for the expression degree.")
        cache = {}
        evals = self._one_pass(target_x_size, target_y_size, cache)
        if evals is None:       # too deep / too many operands for the one-pass kernel: node-by-node route
            evals, _ = self._on_domain(target_x_size, target_y_size, cache)
        return DensePolynomialExt.from_rou_evals(evals, target_x_size, target_y_size)

    # -- one-pass route: the whole tree in ONE kernel over the leaves (tkmk_poly_expr_eval, csrc/expr.hip) --
    _OPS = {"add": 2, "sub": 3, "mul": 4}
    MAX_DEPTH, MAX_LEAVES, MAX_CONSTS, MAX_INSTR = 6, 16, 16, 100

    def _compile(self, prog, leaves, consts, depth):
        """appends the postfix program of this node; returns the stack depth it needs on top of `depth`, or None"""
        k, a = self.kind, self.args

        def const_index(c):
            key = bytes(c)
            if key not in consts:
                consts[key] = len(consts)
            return consts[key]

        if k == "poly":
            if id(a[0]) not in leaves:
                leaves[id(a[0])] = (len(leaves), a[0])
            prog.append((0, leaves[id(a[0])][0]))
            return 1
        if k == "scalar":
            prog.append((1, const_index(a[0])))
            return 1
        if k in self._OPS:
            l = a[0]._compile(prog, leaves, consts, depth)
            if l is None:
                return None
            r = a[1]._compile(prog, leaves, consts, depth + 1)
            if r is None:
                return None
            prog.append((self._OPS[k], 0))
            return max(l, 1 + r)
        if k in ("scale", "xm1"):
            d = a[-1]._compile(prog, leaves, consts, depth)
            if d is None:
                return None
            prog.append((5, const_index(a[0])) if k == "scale" else (6, 0))
            return d
        need = 0                          # weighted sum: running total on the stack
        if not a[0]:
            prog.append((1, const_index(np.zeros(32, np.uint8))))
            return 1
        for i, t in enumerate(a[0]):
            d = t._compile(prog, leaves, consts, depth + (1 if i else 0))
            if d is None:
                return None
            need = max(need, d + (1 if i else 0))
            if i:
                prog.append((2, 0))
        return need

    def _one_pass(self, xs, ys, cache):
        prog, leaves, consts = [], {}, {}
        need = self._compile(prog, leaves, consts, 0)
        if (need is None or need > self.MAX_DEPTH or len(leaves) > self.MAX_LEAVES or len(consts) > self.MAX_CONSTS
                or len(prog) > self.MAX_INSTR or not leaves):
            return None
        bufs = [None] * len(leaves)
        for idx, p in leaves.values():
            bufs[idx] = PolyExpr.poly(p)._on_domain(xs, ys, cache)[0]        # leaf NTTs, cached by identity and domain
        cst = np.zeros(32 * max(1, len(consts)), np.uint8)
        for key, i in consts.items():
            cst[32 * i:32 * i + 32] = np.frombuffer(key, np.uint8)
        return tkmk.poly_expr_eval(prog, bufs, cst, len(consts), xs, ys)

    def _on_domain(self, xs, ys, cache):
        """-> (DeviceBuffer of evaluations, owned): owned buffers are temporaries that may be overwritten in place"""
        k, a = self.kind, self.args
        n = xs * ys
        if k == "poly":
            key = (id(a[0]), xs, ys)
            if key not in cache:
                r = a[0].clone()
                r.resize(xs, ys)
                cache[key] = tkmk.bintt(r.poly, xs, ys, out=r.poly)
            return cache[key], False
        if k == "scalar":
            return tkmk.DeviceBuffer.from_host(np.tile(a[0], n)), True
        if k in ("add", "sub", "mul"):
            l, lo = a[0]._on_domain(xs, ys, cache)
            r, ro = a[1]._on_domain(xs, ys, cache)
            fn = {"add": tkmk.vec_add, "sub": tkmk.vec_sub, "mul": tkmk.vec_mul}[k]
            out = l if lo else (r if ro else None)
            return fn(l, r, out=out), True
        if k == "scale":
            e, eo = a[1]._on_domain(xs, ys, cache)
            one = np.zeros(32, np.uint8)
            one[0] = 1
            if (a[0] == one).all():
                return e, eo
            return tkmk.scalar_mul(tkmk.DeviceBuffer.from_host(a[0]), e, out=e if eo else None), True
        if k == "xm1":
            e, eo = a[0]._on_domain(xs, ys, cache)
            out = e if eo else tkmk.DeviceBuffer(32 * n)
            tkmk._check(tkmk.lib().tkmk_poly_mul_x_minus_one_evals(tkmk._p(e), xs, ys, tkmk._p(out), None), "tkmk_poly_mul_x_minus_one_evals")
            return out, True
        acc = None
        for t in a[0]:
            e, eo = t._on_domain(xs, ys, cache)
            if acc is None:
                acc = e if eo else tkmk.vec_add(e, tkmk.DeviceBuffer.from_host(np.zeros(32 * n, np.uint8)))
            else:
                tkmk.vec_add(acc, e, out=acc)
        if acc is None:
            acc = tkmk.DeviceBuffer.from_host(np.zeros(32 * n, np.uint8))
        return acc, True
