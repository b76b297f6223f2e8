"""not-gpu tier: the oracle's restatement of the DensePolynomialExt host loops
(packages/backend/libs/src/bivariate_polynomial/mod.rs) against independent Python big-int checks and the
identities the reference's own tests assert (libs/src/tests.rs:1090-1237 division reconstruction, eval, shifts)."""
import random

import numpy as np
import pytest

import pyref

R = pyref.R


def _rand_mat(rnd, xs, ys, xdeg=None, ydeg=None):
    xdeg = xs - 1 if xdeg is None else xdeg
    ydeg = ys - 1 if ydeg is None else ydeg
    return [rnd.randrange(R) if (i <= xdeg and j <= ydeg) else 0 for i in range(xs) for j in range(ys)]


def test_find_degree_resize_monomial(oracle):
    rnd = random.Random(1)
    xs, ys = 8, 16
    m = _rand_mat(rnd, xs, ys, 5, 9)
    M = oracle.to_bytes(m, 32)
    assert oracle.poly_find_degree(M, xs, ys) == (5, 9)
    assert oracle.poly_find_degree(np.zeros(32 * 4, np.uint8), 2, 2) == (-1, -1)
    small, nx, ny = oracle.poly_resize(M, xs, ys, 6, 10)          # -> 8 x 16 (rounded up)
    assert (nx, ny) == (8, 16) and (small == M).all()
    big, nx, ny = oracle.poly_resize(M, xs, ys, 9, 3)             # grow x to 16, shrink y to 4 (truncates)
    assert (nx, ny) == (16, 4)
    bv = oracle.to_ints(big, 32)
    assert all(bv[i * 4 + j] == (m[i * ys + j] if i < xs else 0) for i in range(16) for j in range(4))
    sh, nx, ny = oracle.poly_mul_monomial(M, xs, ys, 5, 9, 3, 7)  # target (5+1+3, 9+1+7) -> 16 x 32
    assert (nx, ny) == (16, 32)
    sv = oracle.to_ints(sh, 32)
    x, y = rnd.randrange(R), rnd.randrange(R)
    assert pyref.poly_eval(sv, nx, ny, x, y) == pyref.poly_eval(m, xs, ys, x, y) * pow(x, 3, R) * pow(y, 7, R) % R


def test_scale_and_eval(oracle):
    rnd = random.Random(2)
    xs, ys = 4, 8
    m = _rand_mat(rnd, xs, ys)
    M = oracle.to_bytes(m, 32)
    fx, fy, x, y = (rnd.randrange(1, R) for _ in range(4))
    FX, FY, X, Y = (oracle.to_bytes([v], 32) for v in (fx, fy, x, y))
    sc = oracle.to_ints(oracle.poly_scale_coeffs(M, xs, ys, FX, FY), 32)
    assert sc == [m[i * ys + j] * pow(fx, i, R) * pow(fy, j, R) % R for i in range(xs) for j in range(ys)]
    assert oracle.to_ints(oracle.poly_scale_coeffs(M, xs, ys, FX, None), 32) == [m[i * ys + j] * pow(fx, i, R) % R for i in range(xs) for j in range(ys)]
    assert oracle.to_ints(oracle.poly_eval(M, xs, ys, X, Y), 32) == [pyref.poly_eval(m, xs, ys, x, y)]
    ex = oracle.to_ints(oracle.poly_eval_x(M, xs, ys, X), 32)
    assert ex == [sum(m[i * ys + j] * pow(x, i, R) for i in range(xs)) % R for j in range(ys)]
    ey = oracle.to_ints(oracle.poly_eval_y(M, xs, ys, Y), 32)
    assert ey == [sum(m[i * ys + j] * pow(y, j, R) for j in range(ys)) % R for i in range(xs)]
    # scaling coefficients == evaluating at the scaled point (the reference uses this for coset shifts)
    assert pyref.poly_eval(sc, xs, ys, x, y) == pyref.poly_eval(m, xs, ys, x * fx % R, y * fy % R)


@pytest.mark.parametrize("xs,ys,c,d", [(8, 8, 4, 4), (16, 4, 4, 2), (4, 16, 4, 4), (8, 8, 8, 2), (4, 8, 2, 8)])
def test_div_by_vanishing_opt_reconstructs(oracle, xs, ys, c, d):
    # P = Q_X (X^c - 1) + Q_Y (Y^d - 1) with deg_X Q_Y < c has a unique such decomposition (tests.rs:1090-1237)
    rnd = random.Random(xs * 100 + ys)
    qx = _rand_mat(rnd, xs, ys, xs - c - 1, ys - 1) if xs > c else [0] * (xs * ys)
    qy = _rand_mat(rnd, c, ys, c - 1, ys - d - 1) if ys > d else [0] * (c * ys)
    p = [0] * (xs * ys)
    for i in range(xs):
        for j in range(ys):
            v = 0
            if i >= c:
                v += qx[(i - c) * ys + j]
            v -= qx[i * ys + j]
            if i < c:
                if j >= d:
                    v += qy[i * ys + j - d]
                v -= qy[i * ys + j]
            p[i * ys + j] = v % R
    QX, QY = oracle.poly_div_by_vanishing_opt(oracle.to_bytes(p, 32), xs, ys, c, d)
    assert oracle.to_ints(QX, 32) == qx
    assert oracle.to_ints(QY, 32) == qy


@pytest.mark.parametrize("xs,ys", [(1, 1), (1, 8), (8, 1), (2, 2), (8, 16)])
def test_div_by_ruffini_reconstructs(oracle, xs, ys):
    rnd = random.Random(xs * 31 + ys)
    m = _rand_mat(rnd, xs, ys)
    x, y = rnd.randrange(R), rnd.randrange(R)
    QX, QY, r = oracle.poly_div_by_ruffini(oracle.to_bytes(m, 32), xs, ys, oracle.to_bytes([x], 32), oracle.to_bytes([y], 32))
    qx, qy, rv = oracle.to_ints(QX, 32), oracle.to_ints(QY, 32), oracle.to_ints(r, 32)[0]
    assert rv == pyref.poly_eval(m, xs, ys, x, y)
    # P(a,b) = Q_X(a,b)(a - x) + Q_Y(b)(b - y) + r at random points
    for _ in range(3):
        a, b = rnd.randrange(R), rnd.randrange(R)
        lhs = pyref.poly_eval(m, xs, ys, a, b)
        rhs = (pyref.poly_eval(qx, xs, ys, a, b) * (a - x) + pyref.poly_eval(qy, 1, ys, 0, b) * (b - y) + rv) % R
        assert lhs == rhs
    assert qx[(xs - 1) * ys:] == [0] * ys or xs == 1     # top quotient row is zero (degree drops by one)
