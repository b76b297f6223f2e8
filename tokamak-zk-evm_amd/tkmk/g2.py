"""Host-side G2 arithmetic for the trapdoor half of the reference string: work-alike of Sigma2::gen
(packages/backend/libs/src/group_structures/mod.rs:752-777) — nine scalar multiplications of the G2 generator, done on the CPU in
the reference as well (ICICLE's G2 `to_projective() * scalar` on single points).  Not a device path: the MI355X backend has no G2
kernels because nothing on the prover's hot path touches G2 (the consumers are the pairing verifiers).

Curve: the sextic twist E'(Fp2): y^2 = x^3 + 4(1 + u), Fp2 = Fp[u]/(u^2 + 1).  A point is ((x0, x1), (y0, y1)) with x = x0 + x1 u;
None = infinity.  Encoding = ICICLE's G2Affine / the reference's G2SerdeRkyv (libs/src/iotools/mod.rs:1710-1713,1818-1830): 192 bytes,
x then y, each `to_bytes_le` of the 24-limb extension-field element: real part (48 bytes LE) then imaginary part; (0, 0) = infinity.
The fixed generator of the testing recipe (setup/trusted-setup/src/main.rs:75-78) is pinned on this equation with exactly this
component order (tests/test_g2.py)."""
import numpy as np

P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
B2 = (4, 4)                                             # 4 (1 + u)

# the standard generator of the r-torsion of E'(Fp2) (used when the setup draws a random generator [h]H)
STD_G2 = ((0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
           0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E),
          (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
           0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE))


def f2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def f2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def f2_mul(a, b):
    t0, t1 = a[0] * b[0], a[1] * b[1]
    return ((t0 - t1) % P, ((a[0] + a[1]) * (b[0] + b[1]) - t0 - t1) % P)


def f2_sqr(a):
    return ((a[0] + a[1]) * (a[0] - a[1]) % P, 2 * a[0] * a[1] % P)


def f2_scale(a, k):
    return (a[0] * k % P, a[1] * k % P)


def f2_inv(a):
    n = pow((a[0] * a[0] + a[1] * a[1]) % P, P - 2, P)
    return (a[0] * n % P, (-a[1]) * n % P)


def on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    return f2_sqr(y) == f2_add(f2_mul(f2_sqr(x), x), B2)


# Jacobian (X, Y, Z): x = X / Z^2, y = Y / Z^3; a = 0 (dbl-2009-l, add-2007-bl)
def _dbl(p):
    X, Y, Z = p
    if Z == (0, 0):
        return p
    A, B = f2_sqr(X), f2_sqr(Y)
    C = f2_sqr(B)
    D = f2_scale(f2_sub(f2_sub(f2_sqr(f2_add(X, B)), A), C), 2)
    E = f2_scale(A, 3)
    X3 = f2_sub(f2_sqr(E), f2_scale(D, 2))
    return (X3, f2_sub(f2_mul(E, f2_sub(D, X3)), f2_scale(C, 8)), f2_scale(f2_mul(Y, Z), 2))


def _add(p, q):
    if p[2] == (0, 0):
        return q
    if q[2] == (0, 0):
        return p
    Z1Z1, Z2Z2 = f2_sqr(p[2]), f2_sqr(q[2])
    U1, U2 = f2_mul(p[0], Z2Z2), f2_mul(q[0], Z1Z1)
    S1, S2 = f2_mul(f2_mul(p[1], q[2]), Z2Z2), f2_mul(f2_mul(q[1], p[2]), Z1Z1)
    if U1 == U2:
        return _dbl(p) if S1 == S2 else ((1, 0), (1, 0), (0, 0))
    H = f2_sub(U2, U1)
    I = f2_sqr(f2_scale(H, 2))
    J = f2_mul(H, I)
    r = f2_scale(f2_sub(S2, S1), 2)
    V = f2_mul(U1, I)
    X3 = f2_sub(f2_sub(f2_sqr(r), J), f2_scale(V, 2))
    Y3 = f2_sub(f2_mul(r, f2_sub(V, X3)), f2_scale(f2_mul(S1, J), 2))
    Z3 = f2_mul(f2_sub(f2_sub(f2_sqr(f2_add(p[2], q[2])), Z1Z1), Z2Z2), H)
    return (X3, Y3, Z3)


def _to_jac(pt):
    return ((1, 0), (1, 0), (0, 0)) if pt is None else (pt[0], pt[1], (1, 0))


def _to_affine(p):
    if p[2] == (0, 0):
        return None
    zi = f2_inv(p[2])
    zi2 = f2_sqr(zi)
    return (f2_mul(p[0], zi2), f2_mul(p[1], f2_mul(zi2, zi)))


def add(p, q):
    return _to_affine(_add(_to_jac(p), _to_jac(q)))


def scalar_mul(k, pt):
    """[k mod r] pt (pt in the r-torsion; any k works as an integer multiple otherwise)"""
    k = int(k)
    acc, base = _to_jac(None), _to_jac(pt)
    while k:
        if k & 1:
            acc = _add(acc, base)
        base = _dbl(base)
        k >>= 1
    return _to_affine(acc)


def encode(pt):
    """-> 192-byte record (numpy uint8)"""
    if pt is None:
        return np.zeros(192, np.uint8)
    (x0, x1), (y0, y1) = pt
    return np.frombuffer(b"".join(int(v).to_bytes(48, "little") for v in (x0, x1, y0, y1)), np.uint8).copy()


def decode(rec):
    b = bytes(np.asarray(rec, np.uint8))
    v = [int.from_bytes(b[48 * i:48 * i + 48], "little") for i in range(4)]
    if not any(v):
        return None
    if any(c >= P for c in v):
        raise ValueError("G2 coordinate not reduced")
    return ((v[0], v[1]), (v[2], v[3]))


def from_hex_pair(x_hex, y_hex):
    """G2BaseField::from_hex over the whole 96-byte limb array (setup/trusted-setup/src/main.rs:75-78): the low 48 bytes are
    the real part"""
    X, Y = int(x_hex, 16), int(y_hex, 16)
    lo = (1 << 384) - 1
    return ((X & lo, X >> 384), (Y & lo, Y >> 384))


def sigma2_gen(tau, h):
    """Sigma2::gen + H, in the payload's order (tkmk/crs.py G2_POINTS): H, alpha, alpha^2, alpha^3, alpha^4, gamma, delta, eta, x, y"""
    if not on_curve(h) or h is None:
        raise ValueError("the G2 generator is not a point of the twist")
    a = tau["alpha"] % R
    ks = [1, a, a * a % R, pow(a, 3, R), pow(a, 4, R), tau["gamma"] % R, tau["delta"] % R, tau["eta"] % R, tau["x"] % R, tau["y"] % R]
    return [scalar_mul(k, h) for k in ks]
