"""gpu tier: the device-resident DensePolynomialExt work-alike (tkmk/poly.py over tkmk_poly_*) and encode_poly
(tkmk/sigma.py) vs the oracle's restatement of the reference's host loops, bit-exact.
Cases follow libs/src/tests.rs (resize / monomial / scale / eval / mul / div_by_vanishing / div_by_ruffini) and the
prover's shapes (4096x256 ... 8192x512)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P(gpu):
    gpu.init_ntt_domain_for_size(1 << 23)
    from tkmk.poly import DensePolynomialExt
    return DensePolynomialExt


def _sparse_box(oracle, seed, xs, ys, xdeg, ydeg):
    """random coefficients inside the (xdeg, ydeg) box, zero outside"""
    m = oracle.fr_random(seed, xs * ys).reshape(xs, ys, 32).copy()
    m[xdeg + 1:, :, :] = 0
    m[:, ydeg + 1:, :] = 0
    return m.reshape(-1).copy()


@pytest.mark.parametrize("xs,ys,xd,yd", [(8, 16, 5, 9), (1, 64, 0, 10), (64, 1, 33, 0), (512, 256, 300, 129), (4096, 256, 4095, 255)])
def test_find_degree_and_optimize_size(P, gpu, oracle, xs, ys, xd, yd):
    m = _sparse_box(oracle, xs + ys, xs, ys, xd, yd)
    p = P.from_coeffs(m, xs, ys)
    assert p.degree() == (xs - 1, ys - 1)                    # from_coeffs claims the full box (mod.rs:1546-1547)
    assert p.find_degree() == oracle.poly_find_degree(m, xs, ys) == (xd, yd)
    p.optimize_size()
    want, nx, ny = oracle.poly_resize(m, xs, ys, xd + 1, yd + 1)
    assert (p.x_size, p.y_size, p.x_degree, p.y_degree) == (nx, ny, xd, yd)
    assert (p.copy_coeffs() == want).all()
    z = P.from_coeffs(np.zeros(32 * 16, np.uint8), 4, 4)
    assert z.find_degree() == (-1, -1)
    z.optimize_size()                                        # zero polynomial keeps its size (mod.rs:1814-1816)
    assert (z.x_size, z.y_size) == (4, 4)


@pytest.mark.parametrize("xs,ys,cells", [
    (4096, 512, "dense"),                                    # 2^21 elements
    (8192, 512, [(4100, 300)]),                              # half the rows and 40 % of the columns empty
    (2048, 2048, [(0, 0)]),                                  # one coefficient in the far corner: everything else is scanned and found empty
    (2048, 2048, [(2047, 0), (0, 2047)]),                    # the extremes sit in different cells
    (16384, 256, [(9, 255), (16000, 3)]),
    (2048, 1024, [(5, 1000), (1200, 1001), (2000, 31), (7, 32)]),   # the y maximum inside a 32-column slab, reached in a late row
    (1, 1 << 21, [(0, 77)]), ((1 << 21), 1, [(123456, 0)]),  # degenerate shapes
    (4096, 512, [])])                                        # the zero polynomial
def test_find_degree_on_large_matrices(P, gpu, oracle, xs, ys, cells):
    """find_degree on matrices of 2^21 .. 2^22 elements (the grid-stride loop of k_find_degree, several elements per lane): the oracle's
    answer whatever the support looks like.  (An early-exit variant that reads such matrices from the top was measured in round 3 and
    dropped: 231.7-234.1 ms per configs[3] proof against 230.8 with the full scan — at 3 TB/s the scan is already cheap.)"""
    if cells == "dense":
        m = np.asarray(oracle.fr_random(7, xs * ys)).copy()
    else:
        m = np.zeros(32 * xs * ys, np.uint8)
        for k, (i, j) in enumerate(cells):
            m[32 * (i * ys + j)] = 1 + k
    p = P.from_coeffs(m, xs, ys)
    want = oracle.poly_find_degree(m, xs, ys)
    assert p.find_degree() == want
    if cells and cells != "dense":
        assert want == (max(i for i, _ in cells), max(j for _, j in cells))


def test_resize_and_mul_monomial(P, gpu, oracle):
    xs, ys = 32, 64
    m = _sparse_box(oracle, 7, xs, ys, 20, 40)
    p = P.from_coeffs(m, xs, ys)
    q = p.clone()
    q.resize(33, 5)                                          # grow x to 64, truncate y to 8
    want, nx, ny = oracle.poly_resize(m, xs, ys, 33, 5)
    assert (q.x_size, q.y_size) == (nx, ny) == (64, 8) and (q.copy_coeffs() == want).all()
    p.optimize_size()                                        # 32 x 64, degree (20, 40)
    for ex, ey in ((70, 0), (32, 64), (0, 64)):              # shifts the reference's slice copy can hold (mod.rs:1834-1838)
        sh = p.mul_monomial(ex, ey)
        want, nx, ny = oracle.poly_mul_monomial(p.copy_coeffs(), p.x_size, p.y_size, 20, 40, ex, ey)
        assert (sh.x_size, sh.y_size) == (nx, ny) and (sh.copy_coeffs() == want).all()
    with pytest.raises(ValueError):
        p.mul_monomial(70, 3)                                # 64 + 3 columns do not fit the 64-column target: reference panics
    with pytest.raises(ValueError):
        P.from_coeffs(m, xs, ys).resize(0, 4)
    with pytest.raises(ValueError):
        P.from_coeffs(m[:32 * 24], 3, 8)


@pytest.mark.parametrize("xs,ys", [(4, 8), (1, 16), (256, 64), (4096, 256)])
def test_scale_coeffs_and_eval(P, gpu, oracle, xs, ys):
    m = oracle.fr_random(100 + xs, xs * ys)
    p = P.from_coeffs(m, xs, ys)
    fx, fy, x, y = (oracle.fr_random(200 + k, 1) for k in range(4))
    assert (p.scale_coeffs_x(fx).copy_coeffs() == oracle.poly_scale_coeffs(m, xs, ys, fx, None)).all()
    assert (p.scale_coeffs_y(fy).copy_coeffs() == oracle.poly_scale_coeffs(m, xs, ys, None, fy)).all()
    assert (p.eval(x, y) == oracle.poly_eval(m, xs, ys, x, y)).all()
    ex = p.eval_x(x)
    assert (ex.x_size, ex.y_size) == (1, ys) and (ex.copy_coeffs() == oracle.poly_eval_x(m, xs, ys, x)).all()
    ey = p.eval_y(y)
    assert (ey.x_size, ey.y_size) == (xs, 1) and (ey.copy_coeffs() == oracle.poly_eval_y(m, xs, ys, y)).all()


def test_coset_evals_equal_scaled_coefficients(P, gpu, oracle):
    # libs/src/tests.rs:134-180 on the device-resident object
    xs, ys = 16, 8
    p = P.from_coeffs(oracle.fr_random(5, xs * ys), xs, ys)
    cx, cy = oracle.fr_random(6, 1), oracle.fr_random(7, 1)
    a = p.to_rou_evals(cx, cy).to_host()
    b = p.scale_coeffs_x(cx).scale_coeffs_y(cy).to_rou_evals().to_host()
    assert (a == b).all()
    back = P.from_rou_evals(a, xs, ys, cx, cy)
    assert (back.copy_coeffs() == p.copy_coeffs()).all()


@pytest.mark.parametrize("ax,ay,bx,by", [(3, 2, 4, 5), (0, 7, 9, 0), (0, 0, 5, 5), (0, 0, 0, 0), (100, 30, 27, 33)])
def test_mul_vs_oracle(P, gpu, oracle, ax, ay, bx, by):
    # _mul (mod.rs:1846-1996) == inverse NTT of the pointwise product on the oracle
    axs, ays = 1 << max(ax, 1).bit_length(), 1 << max(ay, 1).bit_length()
    bxs, bys = 1 << max(bx, 1).bit_length(), 1 << max(by, 1).bit_length()
    a = _sparse_box(oracle, 11, axs, ays, ax, ay)
    b = _sparse_box(oracle, 12, bxs, bys, bx, by)
    c = P.from_coeffs(a, axs, ays) * P.from_coeffs(b, bxs, bys)
    x, y = oracle.fr_random(13, 1), oracle.fr_random(14, 1)
    want = oracle.fr_mul(oracle.poly_eval(a, axs, ays, x, y), oracle.poly_eval(b, bxs, bys, x, y))
    assert (c.eval(x, y) == want).all()
    cd = c.find_degree()
    assert cd == (ax + bx, ay + by)
    if ax + ay and bx + by:
        nx, ny = oracle.poly_resized_dims(ax + bx + 1, ay + by + 1)
        ra, _, _ = oracle.poly_resize(a, axs, ays, nx, ny)
        rb, _, _ = oracle.poly_resize(b, bxs, bys, nx, ny)
        prod = oracle.bintt(oracle.fr_mul(oracle.bintt(ra, nx, ny), oracle.bintt(rb, nx, ny)), nx, ny, inverse=True)
        assert (c.x_size, c.y_size) == (nx, ny) and (c.copy_coeffs() == prod).all()


@pytest.mark.parametrize("xs,ys,c,d", [(8, 8, 4, 4), (16, 4, 4, 2), (64, 32, 16, 8), (8192, 512, 4096, 256)])
def test_div_by_vanishing_opt(P, gpu, oracle, xs, ys, c, d):
    if xs * ys <= 4096:
        m = oracle.fr_random(300 + xs, xs * ys)
        p = P.from_coeffs(m, xs, ys)
        qx, qy = p.div_by_vanishing_opt(c, d)
        wx, wy = oracle.poly_div_by_vanishing_opt(m, xs, ys, c, d)
        assert (qx.copy_coeffs() == wx).all() and (qy.copy_coeffs() == wy).all()
        assert qx.degree() == (xs - c - 1, ys - 1) and qy.degree() == (c - 1, ys - d - 1)
    else:
        # production shape (prove0: p0 on 8192x512 divided by X^4096-1, Y^256-1): build P from known quotients on the
        # device and recover them.  P = QX*(X^c-1) + QY*(Y^d-1) via shifted copies.
        qxm = _sparse_box(oracle, 31, xs, ys, xs - c - 1, ys - 1)
        qym = _sparse_box(oracle, 32, c, ys, c - 1, ys - d - 1)
        qx, qy = P.from_coeffs(qxm, xs, ys), P.from_coeffs(qym, c, ys)
        qx.optimize_size()                        # 4096 x 512
        qy.optimize_size()                        # 4096 x 256
        p = (qx.mul_monomial(c, 0) - qx) + (qy.mul_monomial(0, d) - qy)
        rx, ry = p.div_by_vanishing_opt(c, d)
        assert (rx.copy_coeffs() == qxm).all() and (ry.copy_coeffs() == qym).all()
    with pytest.raises(ValueError):
        P.from_coeffs(oracle.fr_random(1, 64), 8, 8).div_by_vanishing_opt(3, 4)


@pytest.mark.parametrize("xs,ys", [(1, 1), (1, 8), (8, 1), (2, 2), (64, 32), (4096, 256), (128, 1024), (16384, 4), (8192, 16), (32, 512)])
def test_div_by_ruffini(P, gpu, oracle, xs, ys):
    m = oracle.fr_random(400 + xs + ys, xs * ys)
    x, y = oracle.fr_random(41, 1), oracle.fr_random(42, 1)
    qx, qy, r = P.from_coeffs(m, xs, ys).div_by_ruffini(x, y)
    wx, wy, wr = oracle.poly_div_by_ruffini(m, xs, ys, x, y)
    assert (qx.copy_coeffs() == wx).all() and (qy.copy_coeffs() == wy).all() and (r == wr).all()
    assert (qy.x_size, qy.y_size) == (1, ys)


def test_encode_poly_commit_identity(P, gpu, oracle):
    """Sigma1.encode_poly on a device-resident fixed-tau CRS: == [P(tau_x, tau_y)]G, trimmed to the degree box, zero
    polynomial -> G1serde::zero(), too-large degree -> error (libs/src/iotools/mod.rs:2047-2058, 2112)."""
    from tkmk.sigma import Sigma1
    pins = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "pins.json")))
    R = oracle.R_MOD
    tx, ty = int(pins["tau_x"], 16), int(pins["tau_y"], 16)
    g = oracle.to_bytes([int(pins["fixed_tau_g1_x"], 16), int(pins["fixed_tau_g1_y"], 16)], 48)
    rs_x, rs_y = 32, 16
    mon = [pow(tx, i, R) * pow(ty, j, R) % R for i in range(rs_x) for j in range(rs_y)]
    crs = gpu.g1_batch_scalar_mul_device(gpu.DeviceBuffer.from_host(oracle.to_bytes(mon, 32)), g, rs_x * rs_y)
    sigma = Sigma1(crs, rs_x, rs_y)
    TX, TY = oracle.to_bytes([tx], 32), oracle.to_bytes([ty], 32)
    for xs, ys, xd, yd in ((32, 16, 31, 15), (32, 16, 20, 9), (64, 32, 17, 3), (4, 4, 0, 0)):
        m = _sparse_box(oracle, xs + xd, xs, ys, xd, yd)
        got = sigma.encode_poly(P.from_coeffs(m, xs, ys))
        val = oracle.poly_eval(m, xs, ys, TX, TY)
        assert (got == oracle.g1_scalar_mul(val, g)).all()
    assert (sigma.encode_poly(P.from_coeffs(np.zeros(32 * 16, np.uint8), 4, 4)) == 0).all()
    with pytest.raises(ValueError):
        sigma.encode_poly(P.from_coeffs(oracle.fr_random(3, 64 * 16), 64, 16))
    # several independent commits in one pipelined call (incl. a zero polynomial) == the one-by-one results
    shapes = ((32, 16, 31, 15), (32, 16, 20, 9), (4, 4, -1, -1), (64, 32, 17, 3), (32, 16, 5, 15))
    ms = [np.zeros(32 * xs * ys, np.uint8) if xd < 0 else _sparse_box(oracle, 7 * xs + xd, xs, ys, xd, yd) for xs, ys, xd, yd in shapes]
    many = sigma.encode_polys([P.from_coeffs(m, sh[0], sh[1]) for m, sh in zip(ms, shapes)])
    for m, sh, cm in zip(ms, shapes, many):
        assert (cm == sigma.encode_poly(P.from_coeffs(m, sh[0], sh[1]))).all()
        assert (cm == oracle.g1_scalar_mul(oracle.poly_eval(m, sh[0], sh[1], TX, TY), g)).all() or sh[2] < 0
    assert (many[2] == 0).all()


def test_polyexpr_fused_equals_coefficient_route(P, gpu, oracle):
    """PolyExpr (mod.rs:141-436; reference test libs/src/tests.rs:1240-1276): fused evaluation-domain route ==
    coefficient route == the expression evaluated at a random point with the oracle."""
    from tkmk.poly import PolyExpr
    xs, ys = 16, 8
    A = P.from_coeffs(_sparse_box(oracle, 1, xs, ys, 9, 5), xs, ys)
    B = P.from_coeffs(_sparse_box(oracle, 2, xs, ys, 6, 7), xs, ys)
    C = P.from_coeffs(_sparse_box(oracle, 3, xs, ys, 3, 2), xs, ys)
    s1, s2 = oracle.fr_random(4, 1), oracle.fr_random(5, 1)
    E = PolyExpr
    expr = E.sub(E.add(E.mul(E.poly(A), E.poly(B)), E.scale(s1, E.mul_x_minus_one(E.poly(C)))),
                 E.weighted_sum([(s2, E.mul(E.poly(A), E.poly(C))), (s1, E.poly(B)), (s2, E.scalar(s1))]))
    assert expr.degree_bound() == (15, 12)
    fused = expr.evaluate_fused()
    coeff = expr.evaluate_coeffs()
    assert (fused.x_size, fused.y_size) == (16, 16)
    x, y = oracle.fr_random(6, 1), oracle.fr_random(7, 1)
    ev = lambda p: oracle.poly_eval(p.copy_coeffs(), p.x_size, p.y_size, x, y)
    a, b, c = ev(A), ev(B), ev(C)
    one = oracle.to_bytes([1], 32)
    xm1 = oracle.fr_sub(x, one)
    want = oracle.fr_sub(oracle.fr_add(oracle.fr_mul(a, b), oracle.fr_mul(s1, oracle.fr_mul(c, xm1))),
                         oracle.fr_add(oracle.fr_add(oracle.fr_mul(s2, oracle.fr_mul(a, c)), oracle.fr_mul(s1, b)), oracle.fr_mul(s2, s1)))
    assert (ev(fused) == want).all() and (ev(coeff) == want).all()
    big = expr.evaluate_fused_with_domain(64, 32)
    assert (big.x_size, big.y_size) == (64, 32) and (ev(big) == want).all()
    with pytest.raises(ValueError):
        expr.evaluate_fused_with_domain(8, 16)


def test_read_R1CS_gen_uvwXY(P, gpu, oracle):
    """read_R1CS_gen_uvwXY (libs/src/iotools/mod.rs:1287-1420) on the device: committed subcircuits (data fixtures), synthetic
    placement variables; vs the oracle's eval_sparse_rows + transpose + inverse bivariate NTT"""
    import random
    from tkmk import r1cs
    qap = os.path.join(os.path.dirname(__file__), "golden", "qap")
    infos = json.load(open(os.path.join(qap, "subcircuitInfo.json")))
    params = dict(json.load(open(os.path.join(qap, "setupParams.json"))), n=64, s_max=8)   # small grid for the oracle
    by_id = {e["id"]: e for e in infos}
    rnd = random.Random(8)
    placements = []
    for sid in (1, 12, 2, 1, 12):                               # 5 placements <= s_max, ids repeat
        nw = by_id[sid]["Nwires"]
        placements.append({"subcircuitId": sid, "variables": ["0x%x" % rnd.randrange(oracle.R_MOD) for _ in range(nw)]})
    placements[1]["variables"][0] = "0x01"
    u, v, w = r1cs.read_R1CS_gen_uvwXY(qap, placements, infos, params)
    n, s_max = params["n"], params["s_max"]
    for m, poly in enumerate((u, v, w)):
        ev = np.zeros(32 * s_max * n, np.uint8)
        for slot, pl in enumerate(placements):
            e = by_id[pl["subcircuitId"]]
            s = r1cs.SubcircuitR1CS.from_r1cs_sparse_only(os.path.join(qap, "r1cs", "subcircuit%d.r1cs" % e["id"]), params, e)
            ptr, wires, coeffs = s.csr[m]
            var = oracle.to_bytes([r1cs.hex_to_fr(h) for h in pl["variables"]], 32)
            ev[32 * n * slot:32 * n * (slot + 1)] = oracle.r1cs_eval_rows(ptr, wires, coeffs if coeffs.size else np.zeros(32, np.uint8), var, n)
        want = oracle.bintt(oracle.fr_transpose(ev, s_max, n), n, s_max, inverse=True)
        assert (poly.x_size, poly.y_size) == (n, s_max)
        assert (poly.copy_coeffs() == want).all(), "matrix %d" % m
    with pytest.raises(ValueError):
        r1cs.read_R1CS_gen_uvwXY(qap, placements * 2, infos, params)     # more placements than s_max


def test_witness_and_permutation_polynomials(P, gpu, oracle):
    """gen_bXY / gen_a_free_X (libs/src/polynomial_structures/mod.rs:104-162) and Permutation::to_poly
    (libs/src/iotools/mod.rs:419-455): evaluations of the resulting polynomials on the grid give back the inputs"""
    import random
    from tkmk import witness
    rnd = random.Random(12)
    R = oracle.R_MOD
    params = {"l": 4, "l_D": 20, "s_max": 8, "l_free": 8, "l_user": 5}
    m_i, s_max = params["l_D"] - params["l"], params["s_max"]
    infos = [{"id": 0, "flattenMap": [0, 4, 5, 19, 20, 7]}, {"id": 1, "flattenMap": [3, 6, 18]}]
    placements = [{"subcircuitId": 0, "variables": ["0x1", "0x0", "0xabc", "0x%x" % (R - 1), "0x5", "0x7"]},
                  {"subcircuitId": 1, "variables": ["0x9", "0x%x" % rnd.randrange(R), "0x0"]},
                  {"subcircuitId": 0, "variables": ["0x2", "0x3", "0x0", "0x4", "0x5", "0x6"]}]
    b = witness.gen_bXY(placements, infos, params)
    ev = oracle.to_ints(b.to_rou_evals().to_host(), 32)
    want = [0] * (m_i * s_max)
    for i, pl in enumerate(placements):
        for g, v in zip(infos[pl["subcircuitId"]]["flattenMap"], pl["variables"]):
            if params["l"] <= g < params["l_D"] and v != "0x0":
                want[(g - params["l"]) * s_max + i] = int(v, 16) % R
    assert ev == want
    with pytest.raises(ValueError):
        witness.gen_bXY([{"subcircuitId": 1, "variables": ["0x1"]}], infos, params)
    inst = {"a_pub_user": ["0x%x" % rnd.randrange(R) for _ in range(5)], "a_pub_block": ["0x%x" % rnd.randrange(R) for _ in range(3)]}
    a = witness.gen_a_free_X(inst, params)
    assert oracle.to_ints(a.to_rou_evals().to_host(), 32) == [int(h, 16) % R for h in inst["a_pub_user"] + inst["a_pub_block"]]
    perm = [{"row": 1, "col": 2, "X": 5, "Y": 7}, {"row": 5, "col": 7, "X": 1, "Y": 2}, {"row": 15, "col": 0, "X": 15, "Y": 0}]
    s0, s1 = witness.permutation_to_poly(perm, m_i, s_max)
    wx = oracle.to_ints(oracle.root_of_unity(m_i), 32)[0]
    wy = oracle.to_ints(oracle.root_of_unity(s_max), 32)[0]
    e0, e1 = oracle.to_ints(s0.to_rou_evals().to_host(), 32), oracle.to_ints(s1.to_rou_evals().to_host(), 32)
    tgt = {(p["row"], p["col"]): (p["X"], p["Y"]) for p in perm}
    for r in range(m_i):
        for c in range(s_max):
            X, Y = tgt.get((r, c), (r, c))
            assert e0[r * s_max + c] == pow(wx, X, R) and e1[r * s_max + c] == pow(wy, Y, R)


def test_expr_one_pass_kernel(P, gpu, oracle):
    """tkmk_poly_expr_eval (csrc/expr.hip): the postfix program of a whole expression in ONE pass over the leaves ==
    the node-by-node route, element for element; form tracking (plain / Montgomery) exercised through every operand
    combination; malformed programs are rejected; over-deep trees fall back to the node-by-node route."""
    from tkmk.poly import PolyExpr as E
    xs, ys = 32, 8
    n = xs * ys
    la, lb, lc = (oracle.fr_random(50 + k, n) for k in range(3))
    k1, k2 = oracle.fr_random(60, 1), oracle.fr_random(61, 1)
    da, db, dc = (gpu.DeviceBuffer.from_host(v) for v in (la, lb, lc))
    consts = np.concatenate([k1, k2])
    wx = oracle.to_ints(oracle.root_of_unity(xs), 32)[0]
    R = oracle.R_MOD
    xm1 = oracle.to_bytes([(pow(wx, i, R) - 1) % R for i in range(xs) for _ in range(ys)], 32)
    mul, add, sub = oracle.fr_mul, oracle.fr_add, oracle.fr_sub
    sc = lambda s, v: oracle.fr_scalar_mul(s, v)   # noqa: E731
    LEAF, CONST, ADD, SUB, MUL, SCALE, XM1 = range(7)
    cases = [
        ([(LEAF, 0), (LEAF, 1), (MUL, 0)], mul(la, lb)),                                   # plain * plain
        ([(LEAF, 0), (CONST, 0), (MUL, 0), (LEAF, 1), (ADD, 0)], add(sc(k1, la), lb)),       # plain * mont -> plain
        ([(CONST, 0), (CONST, 1), (MUL, 0), (LEAF, 2), (SUB, 0)], sub(np.tile(mul(k1, k2), n), lc)),   # mont - plain
        ([(LEAF, 2), (CONST, 1), (SUB, 0)], sub(lc, np.tile(k2, n))),                       # plain - mont
        ([(LEAF, 0), (XM1, 0), (SCALE, 1), (LEAF, 1), (LEAF, 2), (MUL, 0), (SUB, 0)], sub(sc(k2, mul(la, xm1)), mul(lb, lc))),
        ([(LEAF, 0), (LEAF, 0), (MUL, 0), (LEAF, 0), (MUL, 0)], mul(mul(la, la), la)),        # a leaf used three times
        ([(CONST, 1)], np.tile(k2, n)),
    ]
    for prog, want in cases:
        got = gpu.poly_expr_eval(prog, [da, db, dc], consts, 2, xs, ys).to_host()
        assert (got == want).all(), prog
    # output may alias a leaf
    tmp = gpu.DeviceBuffer.from_host(la)
    gpu.poly_expr_eval([(LEAF, 0), (LEAF, 1), (MUL, 0), (LEAF, 0), (ADD, 0)], [tmp, db], consts, 2, xs, ys, out=tmp)
    assert (tmp.to_host() == add(mul(la, lb), la)).all()
    for bad in ([(ADD, 0)], [(LEAF, 0), (LEAF, 1)], [(LEAF, 7)], [(CONST, 5)], [(LEAF, 0), (SCALE, 9)], [(9, 0)],
                [(LEAF, 0)] * 7 + [(ADD, 0)] * 6):
        with pytest.raises(gpu.TkmkError):
            gpu.poly_expr_eval(bad, [da, db, dc], consts, 2, xs, ys)
    # PolyExpr: one-pass == node-by-node, and a right-deep tree beyond the stack limit still evaluates (fallback)
    A, B, C = (P.from_rou_evals(gpu.DeviceBuffer.from_host(v), xs, ys) for v in (la, lb, lc))
    expr = E.sub(E.mul(E.add(E.poly(A), E.poly(B)), E.scale(k1, E.mul_x_minus_one(E.poly(C)))), E.weighted_sum([(k2, E.poly(A)), (k1, E.scalar(k2))]))
    one = expr._one_pass(2 * xs, 2 * ys, {})
    assert one is not None
    node, _ = expr._on_domain(2 * xs, 2 * ys, {})
    assert (one.to_host() == node.to_host()).all()
    deep = E.poly(A)
    for _ in range(8):
        deep = E.add(E.poly(B), deep)                       # B + (B + (B + ... A)): needs 9 stack levels left to right
    assert deep._one_pass(xs, ys, {}) is None
    want = A
    for _ in range(8):
        want = B + want
    assert (deep.evaluate_fused_with_domain(xs, ys).copy_coeffs() == want.copy_coeffs()).all()


def test_poly_lincomb_fused_pass(gpu, oracle):
    """tkmk_poly_lincomb == sum of c_t X^ox Y^oy p_t computed with Python integers: operands of different shapes, shifts, the
    unit / minus-one fast paths, zero coefficients, more terms than one launch holds, and the refusals"""
    import random
    rnd = random.Random(12)
    R = oracle.R_MOD
    out_xs, out_ys = 32, 16
    shapes = [(32, 16, 0, 0), (8, 16, 3, 0), (16, 4, 0, 12), (1, 1, 31, 15), (4, 8, 28, 8), (32, 1, 0, 7), (1, 16, 9, 0)]
    for n_terms in (1, 3, 7, 16, 17, 40):
        terms, dense = [], [[0] * out_ys for _ in range(out_xs)]
        for t in range(n_terms):
            xs, ys, ox, oy = shapes[t % len(shapes)]
            vals = [rnd.randrange(R) for _ in range(xs * ys)]
            c = rnd.choice([1, R - 1, 0, rnd.randrange(R), rnd.randrange(R)])
            buf = gpu.DeviceBuffer.from_host(np.asarray(oracle.to_bytes(vals, 32)))
            terms.append((np.asarray(oracle.to_bytes([c], 32)), buf, xs, ys, ox, oy))
            for i in range(xs):
                for j in range(ys):
                    dense[i + ox][j + oy] = (dense[i + ox][j + oy] + c * vals[i * ys + j]) % R
        got = oracle.to_ints(np.asarray(gpu.poly_lincomb(terms, out_xs, out_ys).to_host()), 32)
        assert got == [v for row in dense for v in row], n_terms
    zero = gpu.poly_lincomb([], 4, 4).to_host()
    assert not np.asarray(zero).any()
    buf = gpu.DeviceBuffer.from_host(np.zeros(32 * 64, np.uint8))
    one = np.asarray(oracle.to_bytes([1], 32))
    for bad in ([(one, buf, 8, 8, 1, 0)], [(one, buf, 8, 8, 0, 1)], [(one, buf, 16, 4)]):     # a shifted operand must fit into out
        with pytest.raises(gpu.TkmkError):
            gpu.poly_lincomb(bad, 8, 8)


def test_expr_leaf_views(P, gpu, oracle):
    """tkmk_poly_expr_eval_views: broadcast vectors and rotated matrices as leaves.  (a) index algebra against numpy rolls /
    repeats; (b) the reason the views exist: on a domain that contains the sub-domain's root, the evaluations of
    p(w^-1 X, Y) are p's evaluations rotated by domain/order rows, and an X-only polynomial's evaluations are a column vector
    — both against the transform-everything route on the coefficients"""
    xs, ys = 32, 8
    n = xs * ys
    la, lb = oracle.fr_random(150, n), oracle.fr_random(151, n)
    vx, vy = oracle.fr_random(152, xs), oracle.fr_random(153, ys)
    k1 = oracle.fr_random(160, 1)
    da, db, dx, dy = (gpu.DeviceBuffer.from_host(v) for v in (la, lb, vx, vy))
    mul, add, sub = oracle.fr_mul, oracle.fr_add, oracle.fr_sub
    LEAF, CONST, ADD, SUB, MUL, SCALE, XM1 = range(7)
    A = np.asarray(la).reshape(xs, ys, 32)
    rolled = np.ascontiguousarray(np.roll(A, (5, 3), axis=(0, 1))).reshape(-1)          # [i][j] <- A[i - 5][j - 3]
    bx = np.ascontiguousarray(np.repeat(np.asarray(vx).reshape(xs, 1, 32), ys, axis=1)).reshape(-1)
    by = np.ascontiguousarray(np.tile(np.asarray(vy).reshape(1, ys, 32), (xs, 1, 1))).reshape(-1)
    leaves = [(da, xs, ys, 5, 3), (db, xs, ys, 0, 0), (dx, xs, 1, 0, 0), (dy, 1, ys, 0, 0), (dx, xs, 1, 31, 0), (dy, 1, ys, 0, 7)]
    bx31 = np.ascontiguousarray(np.repeat(np.roll(np.asarray(vx).reshape(xs, 32), 31, axis=0).reshape(xs, 1, 32), ys, axis=1)).reshape(-1)
    by7 = np.ascontiguousarray(np.tile(np.roll(np.asarray(vy).reshape(ys, 32), 7, axis=0).reshape(1, ys, 32), (xs, 1, 1))).reshape(-1)
    cases = [([(LEAF, 0)], rolled),
             ([(LEAF, 0), (LEAF, 1), (MUL, 0)], mul(rolled, lb)),
             ([(LEAF, 2), (LEAF, 3), (MUL, 0), (LEAF, 1), (ADD, 0)], add(mul(bx, by), lb)),               # outer product + matrix
             ([(LEAF, 4), (LEAF, 5), (SUB, 0), (SCALE, 0), (LEAF, 0), (MUL, 0)], mul(oracle.fr_scalar_mul(k1, sub(bx31, by7)), rolled))]
    for prog, want in cases:
        assert (gpu.poly_expr_eval_views(prog, leaves, k1, 1, xs, ys).to_host() == want).all(), prog
    for bad in ([(da, xs, ys, xs, 0)], [(da, xs, ys, 0, ys)], [(dx, xs // 2, 1, 0, 0)], [(dy, 1, 3, 0, 0)]):
        with pytest.raises(gpu.TkmkError):
            gpu.poly_expr_eval_views([(LEAF, 0)], bad, k1, 1, xs, ys)
    with pytest.raises(gpu.TkmkError):
        gpu.poly_expr_eval_views([(LEAF, 0)], [(da, 24, ys, 0, 0)], k1, 1, 24, ys)                   # sizes must be powers of two
    # (b) p(w_8^-1 X, w_4^-1 Y) on the 32 x 8 domain = p's evaluations rotated by (32/8, 8/4); X-only polynomial = column vector
    gpu.init_ntt_domain_for_size(1 << 12)
    R = oracle.R_MOD
    px, py = 8, 4
    coeffs = oracle.fr_random(170, px * py)
    p = P.from_coeffs(gpu.DeviceBuffer.from_host(coeffs), px, py)
    wix = pow(oracle.to_ints(oracle.root_of_unity(px), 32)[0], R - 2, R)
    wiy = pow(oracle.to_ints(oracle.root_of_unity(py), 32)[0], R - 2, R)
    shifted = p.scale_coeffs_x(oracle.to_bytes([wix], 32)).scale_coeffs_y(oracle.to_bytes([wiy], 32))
    q = p.clone()
    q.resize(xs, ys)
    ev_p = q.to_rou_evals()
    q = shifted.clone()
    q.resize(xs, ys)
    want = q.to_rou_evals().to_host()
    got = gpu.poly_expr_eval_views([(LEAF, 0)], [(ev_p, xs, ys, xs // px, ys // py)], k1, 1, xs, ys).to_host()
    assert (got == want).all()
    ux = P.from_coeffs(gpu.DeviceBuffer.from_host(oracle.fr_random(171, px)), px, 1)
    col = ux.clone()
    col.resize(xs, 1)
    col_ev = col.to_rou_evals()
    full = ux.clone()
    full.resize(xs, ys)
    assert (gpu.poly_expr_eval_views([(LEAF, 0)], [(col_ev, xs, 1, 0, 0)], k1, 1, xs, ys).to_host() == full.to_rou_evals().to_host()).all()


@pytest.mark.parametrize("xs,ys,m,rows", [(64, 4, 16, 64), (128, 8, 64, 100), (256, 2, 128, 256), (96 + 32, 16, 32, 97), (1024, 4, 256, 1000), (64, 8, 4, 33), (32, 4, 1, 32)])
def test_mul_ones_x_equals_product_with_K0(P, gpu, oracle, xs, ys, m, rows):
    """tkmk_poly_mul_ones_x: p * (1/m)(1 + X + ... + X^(m-1)) by running sums == the NTT product with K0 = unit evaluations at index 0
    of the m-th roots (how the reference multiplies by K0, lib.rs:2238-2246) == the definition in big integers (small cases)"""
    gpu.init_ntt_domain_for_size(1 << 14)
    R = oracle.R_MOD
    p = np.asarray(oracle.fr_random(400 + xs + m, xs * ys)).copy()
    p[32 * rows * ys:] = 0                                               # degree rows - 1 in X inside an xs-row buffer
    inv_m = oracle.to_bytes([pow(m, R - 2, R)], 32)
    ox = 1 << (rows - 1 + m - 1).bit_length() if rows - 1 + m > 1 else 1
    ox = max(ox, 1)
    while ox < rows - 1 + m:
        ox <<= 1
    got = gpu.poly_mul_ones_x(gpu.DeviceBuffer.from_host(p), xs, ys, m, inv_m, ox).to_host()
    e = np.zeros(32 * m, np.uint8)
    e[0] = 1
    k0 = P.from_rou_evals(gpu.DeviceBuffer.from_host(e), m, 1)                 # unit_evals(m, 0, x-axis)
    assert (k0.copy_coeffs() == np.tile(inv_m, m)).all()                     # all coefficients 1/m
    if m > 1:
        prod = k0 * P.from_coeffs(gpu.DeviceBuffer.from_host(p), xs, ys)
        pc = np.asarray(prod.copy_coeffs()).reshape(prod.x_size, prod.y_size, 32)
        g = np.asarray(got).reshape(ox, ys, 32)
        nx, ny = min(prod.x_size, ox), min(prod.y_size, ys)
        assert (pc[:nx, :ny] == g[:nx, :ny]).all() and not g[nx:].any() and not g[:, ny:].any() and not pc[nx:].any() and not pc[:, ny:].any()
    if xs * ys <= 1024:
        v = oracle.to_ints(p, 32)
        im = pow(m, R - 2, R)
        want = [im * sum(v[i * ys + j] for i in range(max(0, k - m + 1), min(k, xs - 1) + 1)) % R for k in range(ox) for j in range(ys)]
        assert oracle.to_ints(got, 32) == want
    # truncation: fewer output rows than the product has
    short = gpu.poly_mul_ones_x(gpu.DeviceBuffer.from_host(p), xs, ys, m, inv_m, max(1, ox // 2)).to_host()
    assert (short == got[:short.size]).all()


def test_expr_eval_on_row_slabs_equals_the_whole_domain(gpu, oracle):
    """tkmk_poly_expr_eval_views_slab (the sharded prover's evaluator: every rank its ROWS slab): the rows [x_first, x_first + x_rows) of a
    domain evaluated slab by slab — matrix leaves as slabs, an X-only leaf as the slab's piece of the column vector, a Y-only leaf and a Y
    rotation as on the whole domain, (w_x^i - 1) with the GLOBAL row index — and put together equal tkmk_poly_expr_eval_views on the whole
    domain.  A leaf rotated along X is rolled beforehand (what tkmk_dist_rows_rotate hands a rank)."""
    xs, ys, rows = 32, 8, 8
    n = xs * ys
    gpu.init_ntt_domain_for_size(1 << 12)
    la, lb = oracle.fr_random(250, n), oracle.fr_random(251, n)
    vx, vy = oracle.fr_random(252, xs), oracle.fr_random(253, ys)
    k = oracle.fr_random(254, 2)
    LEAF, CONST, ADD, SUB, MUL, SCALE, XM1 = range(7)
    A = np.asarray(la).reshape(xs, ys, 32)
    a_rolled = np.ascontiguousarray(np.roll(A, 2, axis=0))                       # p(w^-2 X, .): rows rotated by 2 over the WHOLE domain
    B = np.asarray(lb).reshape(xs, ys, 32)
    X = np.asarray(vx).reshape(xs, 32)
    dev = lambda a: gpu.DeviceBuffer.from_host(np.ascontiguousarray(a).reshape(-1))          # noqa: E731
    whole_leaves = [(dev(A), xs, ys, 0, 0), (dev(a_rolled), xs, ys, 0, 3), (dev(B), xs, ys, 0, 0), (dev(X), xs, 1, 0, 0), (gpu.DeviceBuffer.from_host(vy), 1, ys, 0, 5)]
    progs = [[(LEAF, 0), (LEAF, 1), (MUL, 0), (XM1, 0), (LEAF, 2), (SUB, 0)],                 # (X - 1)(a * a') - b
             [(LEAF, 3), (LEAF, 4), (MUL, 0), (SCALE, 1), (LEAF, 2), (ADD, 0), (XM1, 0), (CONST, 0), (ADD, 0)],
             [(LEAF, 1)]]
    for prog in progs:
        want = np.asarray(gpu.poly_expr_eval_views(prog, whole_leaves, k, 2, xs, ys).to_host()).reshape(xs, ys, 32)
        got = []
        for first in range(0, xs, rows):
            sl = slice(first, first + rows)
            leaves = [(dev(A[sl]), rows, ys, 0, 0), (dev(a_rolled[sl]), rows, ys, 0, 3), (dev(B[sl]), rows, ys, 0, 0), (dev(X[sl]), rows, 1, 0, 0),
                      (gpu.DeviceBuffer.from_host(vy), 1, ys, 0, 5)]
            got.append(np.asarray(gpu.poly_expr_eval_views_slab(prog, leaves, k, 2, xs, first, rows, ys).to_host()).reshape(rows, ys, 32))
        assert (np.concatenate(got) == want).all(), prog
    one = [(dev(A[:rows]), rows, ys, 0, 0)]
    for bad in (dict(x_global=xs, x_first=4, x_rows=rows), dict(x_global=xs, x_first=xs, x_rows=rows), dict(x_global=24, x_first=0, x_rows=rows)):
        with pytest.raises(gpu.TkmkError):
            gpu.poly_expr_eval_views_slab([(LEAF, 0)], one, k, 2, bad["x_global"], bad["x_first"], bad["x_rows"], ys)
    with pytest.raises(gpu.TkmkError):                                           # a rotation along X crosses slabs: refused here
        gpu.poly_expr_eval_views_slab([(LEAF, 0)], [(dev(A[:rows]), rows, ys, 1, 0)], k, 2, xs, 0, rows, ys)
