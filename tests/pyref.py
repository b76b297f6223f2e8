"""Independent pure-Python (big-int) restatement of the path's arithmetic, for SMALL cases only.

Used to pin the C oracle (oracle/tk_oracle.c): two independently written implementations agreeing
on field ops, the G1 group law, the NTT definition and the MSM definition.
"""
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB


def root_generator():
    """the declared convention of include/tkmk.h (TKMK_BLS12_381_FR_ROOT_GENERATOR: an inference, not a pin), or the process-wide
    override TKMK_FR_ROOT_GENERATOR that the product and the oracle honour as well"""
    import os
    import re
    env = os.environ.get("TKMK_FR_ROOT_GENERATOR")
    if env and 2 <= int(env) < 65536:
        return int(env)
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "tkmk.h")).read()
    return int(re.search(r"#define\s+TKMK_BLS12_381_FR_ROOT_GENERATOR\s+(\d+)", hdr).group(1))


ROOT32 = pow(root_generator(), (R - 1) >> 32, R)


def root_of_unity(n):
    logn = n.bit_length() - 1
    assert 1 << logn == n
    return pow(ROOT32, 1 << (32 - logn), R)


def ec_add(p, q):
    """affine add on y^2 = x^3 + 4 over Fp; None = infinity"""
    if p is None:
        return q
    if q is None:
        return p
    x1, y1 = p
    x2, y2 = q
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    return (x3, (lam * (x1 - x3) - y1) % P)


def ec_mul(k, p):
    acc = None
    while k:
        if k & 1:
            acc = ec_add(acc, p)
        p = ec_add(p, p)
        k >>= 1
    return acc


def msm(scalars, points):
    acc = None
    for k, p in zip(scalars, points):
        acc = ec_add(acc, ec_mul(k % R, p))
    return acc


def dft(x, inverse=False, coset=1):
    """forward: X[k] = sum_j coset^j x[j] w^{jk}; inverse: exact inverse of the forward map"""
    n = len(x)
    w = root_of_unity(n)
    if not inverse:
        y = [v * pow(coset, j, R) % R for j, v in enumerate(x)]
        return [sum(y[j] * pow(w, j * k, R) for j in range(n)) % R for k in range(n)]
    wi = pow(w, -1, R)
    ni = pow(n, -1, R)
    ci = pow(coset, -1, R)
    y = [sum(x[j] * pow(wi, j * k, R) for j in range(n)) * ni % R for k in range(n)]
    return [v * pow(ci, k, R) % R for k, v in enumerate(y)]


def bintt(mat, xs, ys, inverse=False, cx=1, cy=1):
    """mat: list of xs*ys ints, element (ix,iy) at ix*ys+iy (bivariate_polynomial/mod.rs:1465-1477)"""
    rows = [dft(mat[i * ys:(i + 1) * ys], inverse, cy) for i in range(xs)]
    cols = [dft([rows[i][j] for i in range(xs)], inverse, cx) for j in range(ys)]
    return [cols[j][i] for i in range(xs) for j in range(ys)]


def pt_to_int(p):
    return (0, 0) if p is None else p


# ---- bivariate coefficient matrices as {(i, j): coeff} dense lists, element (i, j) at i*ys + j ----
def poly_eval(m, xs, ys, x, y):
    return sum(m[i * ys + j] * pow(x, i, R) * pow(y, j, R) for i in range(xs) for j in range(ys)) % R


def poly_mul_dense(a, axs, ays, b, bxs, bys, oxs, oys):
    out = [0] * (oxs * oys)
    for i in range(axs):
        for j in range(ays):
            v = a[i * ays + j]
            if not v:
                continue
            for k in range(bxs):
                for l in range(bys):
                    w = b[k * bys + l]
                    if w:
                        out[(i + k) * oys + (j + l)] = (out[(i + k) * oys + (j + l)] + v * w) % R
    return out
