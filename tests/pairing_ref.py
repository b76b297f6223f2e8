"""Test infrastructure: a small, slow BLS12-381 pairing over Python integers, to check the verifier's equations
(packages/backend/verify-rust/src/lib.rs:248-352, `pairing` of libs/src/group_structures) on real group elements — proof points,
CRS G1 points, Sigma2's G2 points — instead of on discrete logarithms.

Construction (the textbook ate Miller loop in its plainest form): Fp12 = Fp[w] / (w^12 - 2 w^6 + 2), into which Fp2 = Fp[u]/(u^2 + 1)
embeds by u -> w^6 - 1; a point of the twist E'(Fp2): y^2 = x^3 + 4(1 + u) maps to E(Fp12): y^2 = x^3 + 4 by (x, y) -> (x / w^2, y / w^3);
f_{|x|,Q}(P) by affine line functions over Fp12 for the loop count |x| = 0xd201000000010000, then f^((p^12 - 1) / r).  The sign of x and
the usual speed-ups (sparse lines, cyclotomic squaring, Frobenius) are left out: any fixed non-degenerate bilinear map decides a
product-of-pairings equation, and bilinearity / non-degeneracy are checked in tests/test_pairing_ref.py.  ~1 s per pairing; products share
the final exponentiation."""
P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
ATE_LOOP = 0xD201000000010000
FINAL_EXP = (P ** 12 - 1) // R


class F12:
    """element of Fp[w] / (w^12 - 2 w^6 + 2): 12 coefficients, constant term first"""
    __slots__ = ("c",)

    def __init__(self, c):
        self.c = [v % P for v in c]
        assert len(self.c) == 12

    @classmethod
    def of(cls, v):
        return cls([v] + [0] * 11)

    def __add__(self, o):
        return F12([a + b for a, b in zip(self.c, o.c)])

    def __sub__(self, o):
        return F12([a - b for a, b in zip(self.c, o.c)])

    def __neg__(self):
        return F12([-a for a in self.c])

    def __eq__(self, o):
        return self.c == o.c

    def __mul__(self, o):
        if isinstance(o, int):
            return F12([a * o for a in self.c])
        a, b = self.c, o.c
        t = [0] * 23
        for i, ai in enumerate(a):
            if ai:
                for j, bj in enumerate(b):
                    t[i + j] += ai * bj
        for k in range(22, 11, -1):          # w^12 = 2 w^6 - 2
            v = t[k]
            if v:
                t[k - 6] += 2 * v
                t[k - 12] -= 2 * v
        return F12(t[:12])

    def inv(self):
        """extended Euclid in Fp[w] against the modulus polynomial"""
        def deg(p):
            d = len(p) - 1
            while d >= 0 and p[d] == 0:
                d -= 1
            return d

        mod = [2, 0, 0, 0, 0, 0, -2 % P, 0, 0, 0, 0, 0, 1]
        r0, r1 = mod[:], self.c[:] + [0]
        s0, s1 = [0] * 13, [1] + [0] * 12      # s_k * self = r_k  (mod modulus)
        while deg(r1) > 0:
            d0, d1 = deg(r0), deg(r1)
            if d0 < d1:
                r0, r1, s0, s1 = r1, r0, s1, s0
                continue
            q = r0[d0] * pow(r1[d1], P - 2, P) % P
            sh = d0 - d1
            for i in range(d1 + 1):
                r0[i + sh] = (r0[i + sh] - q * r1[i]) % P
            for i in range(13 - sh):
                s0[i + sh] = (s0[i + sh] - q * s1[i]) % P
        if deg(r1) < 0:
            raise ZeroDivisionError("F12 inverse of zero")
        k = pow(r1[0], P - 2, P)
        return F12([v * k for v in s1[:12]])

    def __truediv__(self, o):
        return self * o.inv()

    def __pow__(self, e):
        out, base = F12.of(1), self
        while e:
            if e & 1:
                out = out * base
            base = base * base
            e >>= 1
        return out


W = F12([0, 1] + [0] * 10)
W2_INV, W3_INV = (W * W).inv(), (W * W * W).inv()


def _embed_fp2(a):
    """a0 + a1 u with u = w^6 - 1"""
    return F12([a[0] - a[1], 0, 0, 0, 0, 0, a[1], 0, 0, 0, 0, 0])


def twist(q):
    """point of E'(Fp2) (tkmk.g2 representation) -> point of E(Fp12)"""
    return (_embed_fp2(q[0]) * W2_INV, _embed_fp2(q[1]) * W3_INV)


def cast_g1(p):
    return (F12.of(p[0]), F12.of(p[1]))


def _line(p1, p2, t):
    x1, y1 = p1
    x2, y2 = p2
    xt, yt = t
    if not x1 == x2:
        m = (y2 - y1) / (x2 - x1)
    elif y1 == y2:
        m = (x1 * x1 * 3) / (y1 * 2)
    else:
        return xt - x1
    return m * (xt - x1) - (yt - y1)


def _add(p1, p2):
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2:
        if not y1 == y2:
            return None
        m = (x1 * x1 * 3) / (y1 * 2)
    else:
        m = (y2 - y1) / (x2 - x1)
    x3 = m * m - x1 - x2
    return (x3, m * (x1 - x3) - y1)


def miller_loop(q_twisted, p_cast):
    """f_{|x|, Q}(P) without the final exponentiation; None for either argument = the neutral factor"""
    if q_twisted is None or p_cast is None:
        return F12.of(1)
    r, f = q_twisted, F12.of(1)
    for i in range(ATE_LOOP.bit_length() - 2, -1, -1):
        f = f * f * _line(r, r, p_cast)
        r = _add(r, r)
        if (ATE_LOOP >> i) & 1:
            f = f * _line(r, q_twisted, p_cast)
            r = _add(r, q_twisted)
    return f


def pairing_product(pairs):
    """prod e(P_i, Q_i) for [(P_i, Q_i)], P_i = (x, y) ints or None (G1 affine), Q_i = tkmk.g2 point or None; one final exponentiation"""
    f = F12.of(1)
    for p, q in pairs:
        f = f * miller_loop(None if q is None else twist(q), None if p is None else cast_g1(p))
    return f ** FINAL_EXP


def g1_from_record(rec96):
    b = bytes(rec96)
    x, y = int.from_bytes(b[:48], "little"), int.from_bytes(b[48:96], "little")
    return None if x == 0 and y == 0 else (x, y)
