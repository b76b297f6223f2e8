"""The prover: work-alike of Prover::init / prove0 .. prove4 (packages/backend/prove/src/lib.rs:675-1206, 1446-3206) and of the
driver in prove/src/main.rs:27-97, over the device-resident polynomial layer (tkmk/poly.py), the commitment encoder
(tkmk/sigma.py) and the binding commitments (tkmk/binding.py).  Same structure names (Mixer, InstancePolynomials, Witness,
Quotients, Binding, Proof0..Proof4) and the same statement order as the reference; what differs is where the work runs:

  * every polynomial stays in HBM from `init` to the last commitment (the reference copies coefficient matrices to the host
    for find_degree / resize / scale / the two divisions and back);
  * the independent commitments of one round go to the device as ONE pipelined tkmk_msm_multi call
    (Sigma1.encode_polys): 6 in prove0, 2 in prove2, 9 in prove4;
  * prove1's running product is a device scan (tkmk_vec_suffix_product) instead of a serial host loop (lib.rs:1858-1866);
  * prove2's p_comb is one pass of the fused expression evaluator (PolyExpr.evaluate_fused_with_domain, lib.rs:2110-2146).

Host scalars (mixer values, challenges, evaluations) are Python ints mod r; at the device boundary they are the 32-byte
little-endian records of the C ABI.  `testing_mode=True` runs the reference's `testing-mode` feature checks (R1CS
satisfaction, quotient identities at a random point, zero Ruffini remainders) and raises AssertionError where it panics."""
import json
import os
import secrets
import time

import numpy as np

import tkmk
from tkmk import binding as _binding
from tkmk import proofio
from tkmk.poly import DensePolynomialExt, PolyExpr
from tkmk.r1cs import R_MOD, PlacementValues, read_R1CS_gen_uvwXY
from tkmk.transcript import TranscriptManager
from tkmk.witness import gen_a_free_X, gen_bXY, permutation_to_poly

R = R_MOD


def fr(v):
    """int -> 32-byte little-endian Fr record"""
    return np.frombuffer((int(v) % R).to_bytes(32, "little"), np.uint8).copy()


def fr_int(b):
    return int.from_bytes(bytes(np.asarray(b, np.uint8)[:32]), "little")


def _coeffs(vals, x_size, y_size):
    return DensePolynomialExt.from_coeffs(np.concatenate([fr(v) for v in vals]), x_size, y_size)


def _sparse(entries, x_size, y_size):
    """coefficient vector of x_size * y_size zeros with a few {index: value} entries"""
    buf = np.zeros((x_size * y_size, 32), np.uint8)
    for i, v in entries.items():
        buf[i] = fr(v)
    return DensePolynomialExt.from_coeffs(buf.reshape(-1), x_size, y_size)


def poly_comb(*terms):
    """poly_comb! (lib.rs:30-38): sum of c_i * p_i"""
    acc = None
    for c, p in terms:
        t = p.mul_scalar(fr(c))
        acc = t if acc is None else acc + t
    return acc


def low_degree_x_times_vanishing(coeffs, exponent):
    """lib.rs:48-57: (sum c_i X^i) * (X^exponent - 1)"""
    assert exponent > 0
    x_size = 1 << (exponent + len(coeffs) - 1).bit_length()
    entries = {}
    for i, c in enumerate(coeffs):                       # accumulated: the two copies overlap when exponent < len(coeffs)
        entries[i] = (entries.get(i, 0) - c) % R
        entries[i + exponent] = (entries.get(i + exponent, 0) + c) % R
    return _sparse(entries, x_size, 1)


def low_degree_y_times_vanishing(coeffs, exponent):
    """lib.rs:59-68"""
    assert exponent > 0
    y_size = 1 << (exponent + len(coeffs) - 1).bit_length()
    entries = {}
    for i, c in enumerate(coeffs):                       # accumulated: the two copies overlap when exponent < len(coeffs)
        entries[i] = (entries.get(i, 0) - c) % R
        entries[i + exponent] = (entries.get(i + exponent, 0) + c) % R
    return _sparse(entries, 1, y_size)


def mul_by_x_minus_one(poly):
    return poly.mul_monomial(1, 0) - poly


def mul_by_one_minus_x(poly):
    return poly - poly.mul_monomial(1, 0)


def mul_by_linear_x(poly, coeffs):
    assert len(coeffs) == 2
    return poly.mul_scalar(fr(coeffs[0])) + poly.mul_monomial(1, 0).mul_scalar(fr(coeffs[1]))


def mul_by_linear_y(poly, coeffs):
    assert len(coeffs) == 2
    return poly.mul_scalar(fr(coeffs[0])) + poly.mul_monomial(0, 1).mul_scalar(fr(coeffs[1]))


def mul_by_sparse_const_x_y(poly, constant, x_coeff, y_coeff):
    """lib.rs:96-109"""
    partial = poly.mul_scalar(fr(constant)) + poly.mul_monomial(1, 0).mul_scalar(fr(x_coeff))
    return partial + poly.mul_monomial(0, 1).mul_scalar(fr(y_coeff))


def mul_by_term9(poly, rB_X, rB_Y, t_mi_eval, t_smax_eval):
    """lib.rs:111-124"""
    assert len(rB_X) == 2 and len(rB_Y) == 2
    constant = (t_mi_eval * rB_X[0] + t_smax_eval * rB_Y[0]) % R
    return mul_by_sparse_const_x_y(poly, constant, t_mi_eval * rB_X[1] % R, t_smax_eval * rB_Y[1] % R)


def _vanishing(size, x_axis):
    """t(X) = X^size - 1 on 2*size coefficients (lib.rs:849-894)"""
    return _sparse({0: R - 1, size: 1}, 2 * size, 1) if x_axis else _sparse({0: R - 1, size: 1}, 1, 2 * size)


def _unit_evals(size, index, x_axis):
    """from_rou_evals of the indicator of `index` (the Lagrange polynomials K, L, K0 of lib.rs:2018-2100)"""
    e = np.zeros(32 * size, np.uint8)
    e[32 * index] = 1
    return DensePolynomialExt.from_rou_evals(e, size, 1) if x_axis else DensePolynomialExt.from_rou_evals(e, 1, size)


def g1_lincomb(terms):
    """sum of scalar * point over 96-byte affine records through the MSM entry (the reference's G1serde `+`, `-`, `* scalar`:
    libs/src/group_structures/mod.rs:888-1009); the result is the canonical affine point, (0,0) for infinity"""
    sc = np.concatenate([fr(s) for s, _ in terms])
    pts = np.concatenate([np.asarray(p, np.uint8).reshape(96) for _, p in terms])
    return tkmk.projective_to_affine_bytes(tkmk.msm(sc, pts))


def g1_lincombs(rows):
    """several equally long linear combinations in ONE batched MSM call (independent bases per row) -> [96-byte affine, ...]"""
    k = len(rows[0])
    assert all(len(r) == k for r in rows)
    sc = np.concatenate([fr(s) for r in rows for s, _ in r])
    pts = np.concatenate([np.asarray(p, np.uint8).reshape(96) for r in rows for _, p in r])
    res = tkmk.projective_to_affine_bytes(tkmk.msm(sc, pts, msm_size=k, batch=len(rows), shared_points=False))
    return [np.array(res[96 * i:96 * (i + 1)]) for i in range(len(rows))]


def random_mixer(rng=None):
    """Mixer (lib.rs:1040-1080); rW_X / rW_Y are 3 random values resized to 4 with a zero"""
    rnd = (lambda: secrets.randbelow(R)) if rng is None else (lambda: rng.randrange(R))
    return {"rU_X": rnd(), "rU_Y": rnd(), "rV_X": rnd(), "rV_Y": rnd(),
            "rW_X": [rnd(), rnd(), rnd(), 0], "rW_Y": [rnd(), rnd(), rnd(), 0],
            "rB_X": [rnd(), rnd()], "rB_Y": [rnd(), rnd()], "rO_mid": rnd(), "rR_X": rnd(), "rR_Y": rnd()}


def validate_setup_shape(sp):
    """setup_shape + validate_setup_shape (libs/src/utils/mod.rs:21-46), with the reference's panic messages; -> m_I"""
    if sp["l_D"] < sp["l"]:
        raise ValueError("Invalid setup params: l_D must be >= l.")
    m_i = sp["l_D"] - sp["l"]
    for name, v in (("n", sp["n"]), ("s_max", sp["s_max"]), ("m_I", m_i)):
        if v <= 0 or v & (v - 1):
            raise ValueError("%s is not a power of two." % name)
    return m_i


class Prover:
    def __init__(self):
        self.testing_mode = False
        self.timing = {}

    # ------------------------------------------------------------------ init (lib.rs:675-1206)
    @classmethod
    def init(cls, qap_path, synthesizer_path, setup_path, mixer=None, testing_mode=False, sigma=None, rng=None):
        """-> (Prover, binding dict).  `sigma` = (Sigma1, tables, singles) skips reading <setup_path>/combined_sigma.tkcrs (tests
        and the bench stage the CRS in HBM directly); `mixer` fixes the blinding scalars (ScalarCfg::generate_random in the reference)"""
        t0 = time.perf_counter()
        load = lambda d, name: json.load(open(os.path.join(d, name)))                 # noqa: E731
        inputs = {"setup_params": load(qap_path, "setupParams.json"), "subcircuit_infos": load(qap_path, "subcircuitInfo.json"),
                  "placement_variables": load(synthesizer_path, "placementVariables.json"),
                  "permutation": load(synthesizer_path, "permutation.json"), "instance": load(synthesizer_path, "instance.json")}
        return cls.init_from(inputs, qap_path, setup_path, mixer=mixer, testing_mode=testing_mode, sigma=sigma, rng=rng, _t0=t0)

    @classmethod
    def init_from(cls, inputs, qap_path, setup_path=None, mixer=None, testing_mode=False, sigma=None, rng=None, _t0=None):
        """Prover::init after the JSON files are parsed: `inputs` holds the five parsed documents (the bench hands them over
        in memory; a production-shape placementVariables.json is hundreds of MB of hex text)"""
        from tkmk import crs
        self = cls()
        self.testing_mode = testing_mode
        t0 = time.perf_counter() if _t0 is None else _t0
        sp = inputs["setup_params"]
        self.setup_params = sp
        m_i = validate_setup_shape(sp)
        n, s_max = sp["n"], sp["s_max"]
        self.m_i = m_i
        tkmk.init_ntt_domain_for_size(4 * max(m_i, n) * 2 * s_max)          # prover_verifier_ntt_domain_size (libs/src/utils/mod.rs:51-58)
        subcircuit_infos, placement_variables = inputs["subcircuit_infos"], inputs["placement_variables"]
        self.timing["init.load"] = time.perf_counter() - t0

        t1 = time.perf_counter()
        values = PlacementValues(placement_variables)                          # every hex string parsed once
        self.bXY = gen_bXY(placement_variables, subcircuit_infos, sp, values)
        self.uXY, self.vXY, self.wXY = read_R1CS_gen_uvwXY(qap_path, placement_variables, subcircuit_infos, sp, values)
        self.rXY = DensePolynomialExt.zero()
        self.q = {}
        self.cache = {}
        permutation_raw, instance = inputs["permutation"], inputs["instance"]
        self.a_free_X = gen_a_free_X(instance, sp)
        self.t_n = _vanishing(n, True)
        self.t_mi = _vanishing(m_i, True)
        self.t_smax = _vanishing(s_max, False)
        self.s0XY, self.s1XY = permutation_to_poly(permutation_raw, m_i, s_max)
        self.timing["init.build"] = time.perf_counter() - t1

        if testing_mode:
            self._check_lemma3(permutation_raw)

        if sigma is None:
            path = os.path.join(setup_path, "combined_sigma.tkcrs")
            if not os.path.exists(path):
                raise FileNotFoundError("No reference string is found. Run the Setup first (expected %s)." % path)
            sections = crs.read_payload(path)
            sigma1, tables = crs.load_sigma1(sections, sp)
            singles = {k: np.array(crs.single_g1(sections, k)) for k in ("delta", "eta")}
        else:
            sigma1, tables, singles = sigma
        self.sigma1, self.tables = sigma1, tables
        self.mixer = mx = random_mixer(rng) if mixer is None else mixer

        t2 = time.perf_counter()
        host = lambda name: tables[name].to_host().reshape(-1, 96) if isinstance(tables[name], tkmk.DeviceBuffer) else np.asarray(tables[name], np.uint8).reshape(-1, 96)   # noqa: E731
        A_free = sigma1.encode_poly(self.a_free_X)
        O_pub_free = _binding.encode_O_pub_free(tables["gamma_inv_o_inst"], placement_variables, subcircuit_infos, sp, values)
        O_mid_core = _binding.encode_O_mid_no_zk(tables["eta_inv_li_o_inter_alpha4_kj"], placement_variables, subcircuit_infos, sp, values)
        O_mid = g1_lincomb([(1, O_mid_core), (mx["rO_mid"], singles["delta"])])
        O_prv_core = _binding.encode_O_prv_no_zk(tables["delta_inv_li_o_prv"], placement_variables, subcircuit_infos, sp, values)
        xh, xj, yi = host("delta_inv_alphak_xh_tx"), host("delta_inv_alpha4_xj_tx"), host("delta_inv_alphak_yi_ty")
        O_prv = g1_lincomb([                                                  # lib.rs:1146-1160
            (1, O_prv_core), (R - mx["rO_mid"], singles["eta"]),
            (mx["rU_X"], xh[0 * 3 + 0]), (mx["rV_X"], xh[1 * 3 + 0]),
            (mx["rW_X"][0], xh[2 * 3 + 0]), (mx["rW_X"][1], xh[2 * 3 + 1]), (mx["rW_X"][2], xh[2 * 3 + 2]),
            (mx["rB_X"][0], xj[0]), (mx["rB_X"][1], xj[1]),
            (mx["rU_Y"], yi[0 * 3 + 0]), (mx["rV_Y"], yi[1 * 3 + 0]),
            (mx["rW_Y"][0], yi[2 * 3 + 0]), (mx["rW_Y"][1], yi[2 * 3 + 1]), (mx["rW_Y"][2], yi[2 * 3 + 2]),
            (mx["rB_Y"][0], yi[3 * 3 + 0]), (mx["rB_Y"][1], yi[3 * 3 + 1])])
        self.timing["init.binding"] = time.perf_counter() - t2
        self.timing["init.total"] = time.perf_counter() - t0
        return self, {"A_free": A_free, "O_pub_free": O_pub_free, "O_mid": O_mid, "O_prv": O_prv}

    def _check_lemma3(self, permutation_raw):
        """testing-mode block of init (lib.rs:916-1019): copy constraints on b, well-formed s0 / s1, and the grand-product
        identity prod f = prod g for random thetas"""
        m_i, s_max = self.m_i, self.setup_params["s_max"]
        rows = lambda p: np.asarray(p.to_rou_evals().to_host()).reshape(-1, 32)       # noqa: E731
        b = self.bXY.clone()
        b.resize(m_i, s_max)
        b_ev = rows(b)
        for e in permutation_raw:
            assert (b_ev[e["row"] * s_max + e["col"]] == b_ev[e["X"] * s_max + e["Y"]]).all(), "b(X,Y) violates a copy constraint"
        thetas = [secrets.randbelow(R) for _ in range(3)]
        f, g = self._fg(thetas)
        f.resize(m_i, s_max)
        g.resize(m_i, s_max)
        lhs = tkmk.vec_product(f.to_rou_evals(), m_i * s_max).to_host()
        rhs = tkmk.vec_product(g.to_rou_evals(), m_i * s_max).to_host()
        assert (np.asarray(lhs) == np.asarray(rhs)).all(), "Lemma 3 fails: prod f != prod g"

    def _fg(self, thetas):
        """f = b + th0 s0 + th1 s1 + th2,  g = b + th0 X + th1 Y + th2 (lib.rs:1807-1811)"""
        X_mono = _coeffs([0, 1], 2, 1)
        Y_mono = _coeffs([0, 1], 1, 2)
        f = ((self.bXY + self.s0XY.mul_scalar(fr(thetas[0]))) + self.s1XY.mul_scalar(fr(thetas[1]))).add_scalar(fr(thetas[2]))
        g = ((self.bXY + X_mono.mul_scalar(fr(thetas[0]))) + Y_mono.mul_scalar(fr(thetas[1]))).add_scalar(fr(thetas[2]))
        return f, g

    def _rand_point_check(self, lhs_poly, parts, what):
        """testing-mode identity lhs(x,y) == sum_k q_k(x,y) * d_k(x,y) at a random point"""
        x, y = secrets.randbelow(R), secrets.randbelow(R)
        lhs = fr_int(lhs_poly.eval(fr(x), fr(y)))
        rhs = sum(fr_int(q.eval(fr(x), fr(y))) * d(x, y) for q, d in parts) % R
        assert lhs == rhs, what

    # ------------------------------------------------------------------ prove0 (lib.rs:1446-1782)
    def prove0(self):
        sp, mx = self.setup_params, self.mixer
        n, s_max = sp["n"], sp["s_max"]
        p0XY = self.uXY * self.vXY - self.wXY
        if self.testing_mode:
            u = self.uXY.to_rou_evals()
            uv = tkmk.vec_mul(u, self.vXY.to_rou_evals())
            if not (np.asarray(uv.to_host()) == np.asarray(self.wXY.to_rou_evals().to_host())).all():
                raise AssertionError("Evaluations of u(X,Y), v(X,Y), and w(X,Y) do not satisfy R1CS.")
        self.q[0], self.q[1] = p0XY.div_by_vanishing_opt(n, s_max)
        if self.testing_mode:
            self._rand_point_check(p0XY, [(self.q[0], lambda x, y: pow(x, n, R) - 1), (self.q[1], lambda x, y: pow(y, s_max, R) - 1)],
                                   "u, v, w do not satisfy the arithmetic constraints")
        rW_X = _coeffs(mx["rW_X"], len(mx["rW_X"]), 1)
        rW_Y = _coeffs(mx["rW_Y"], 1, len(mx["rW_Y"]))
        UXY = poly_comb((1, self.uXY), (mx["rU_X"], self.t_n), (mx["rU_Y"], self.t_smax))
        VXY = poly_comb((1, self.vXY), (mx["rV_X"], self.t_n), (mx["rV_Y"], self.t_smax))
        W_zk = low_degree_x_times_vanishing(mx["rW_X"], n) + low_degree_y_times_vanishing(mx["rW_Y"], s_max)
        self.cache["w_zk"] = W_zk
        WXY = self.wXY + W_zk
        Q_AX_XY = poly_comb((1, self.q[0]), (mx["rU_X"], self.vXY), (mx["rV_X"], self.uXY), (R - 1, rW_X),
                            (mx["rU_X"] * mx["rV_X"] % R, self.t_n), (mx["rU_Y"] * mx["rV_X"] % R, self.t_smax))
        Q_AY_XY = poly_comb((1, self.q[1]), (mx["rU_Y"], self.vXY), (mx["rV_Y"], self.uXY), (R - 1, rW_Y),
                            (mx["rU_X"] * mx["rV_Y"] % R, self.t_n), (mx["rU_Y"] * mx["rV_Y"] % R, self.t_smax))
        term_B_zk = low_degree_x_times_vanishing(mx["rB_X"], self.m_i) + low_degree_y_times_vanishing(mx["rB_Y"], s_max)
        self.cache["term_b_zk"] = term_B_zk
        BXY = self.bXY + term_B_zk
        U, V, W, Q_AX, Q_AY, B = self.sigma1.encode_polys([UXY, VXY, WXY, Q_AX_XY, Q_AY_XY, BXY])
        return {"U": U, "V": V, "W": W, "Q_AX": Q_AX, "Q_AY": Q_AY, "B": B}

    # ------------------------------------------------------------------ prove1 (lib.rs:1784-1956)
    def prove1(self, thetas):
        sp, mx = self.setup_params, self.mixer
        m_i, s_max = self.m_i, sp["s_max"]
        fXY, gXY = self._fg(thetas)
        fXY.resize(m_i, s_max)
        gXY.resize(m_i, s_max)
        f_ev, g_ev = fXY.to_rou_evals(), gXY.to_rou_evals()
        # r[last] = 1, r[idx] = r[idx + 1] * (g / f)[idx + 1] over the TRANSPOSED (s_max x m_i) order (lib.rs:1858-1866)
        scalers_tr = tkmk.transpose(tkmk.vec_div(g_ev, f_ev), m_i, s_max)
        r_ev = tkmk.transpose(tkmk.vec_suffix_product(scalers_tr), s_max, m_i)
        self.rXY = DensePolynomialExt.from_rou_evals(r_ev, m_i, s_max)
        if self.testing_mode:
            r, g, f = (np.asarray(b.to_host()).reshape(m_i, s_max, 32) for b in (r_ev, g_ev, f_ev))
            ints = lambda a: int.from_bytes(bytes(a), "little")                       # noqa: E731
            for row in range(1, m_i - 1):                                             # lib.rs:1897-1909 (a strided sample of columns)
                for col in range(0, s_max - 1, max(1, s_max // 8)):
                    assert ints(r[row, col]) * ints(g[row, col]) % R == ints(r[row - 1, col]) * ints(f[row, col]) % R, "r(X,Y) recursion"
        RXY = self.rXY + (self.t_mi.mul_scalar(fr(mx["rR_X"])) + self.t_smax.mul_scalar(fr(mx["rR_Y"])))
        return {"R": self.sigma1.encode_poly(RXY)}

    # ------------------------------------------------------------------ prove2 (lib.rs:1958-2270)
    def prove2(self, thetas, kappa0):
        sp, mx = self.setup_params, self.mixer
        m_i, s_max = self.m_i, sp["s_max"]
        kappa0_sq = kappa0 * kappa0 % R
        w_inv_x = pow(fr_int(tkmk.get_root_of_unity(m_i)), R - 2, R)
        w_inv_y = pow(fr_int(tkmk.get_root_of_unity(s_max)), R - 2, R)
        r_omegaX = self.rXY.scale_coeffs_x(fr(w_inv_x))
        r_omegaX_omegaY = r_omegaX.scale_coeffs_y(fr(w_inv_y))
        fXY, gXY = self._fg(thetas)
        lagrange_KL_XY = _unit_evals(m_i, m_i - 1, True) * _unit_evals(s_max, s_max - 1, False)
        self.cache["lagrange_kl_xy"] = lagrange_KL_XY
        lagrange_K0_XY = _unit_evals(m_i, 0, True)

        P = PolyExpr
        r_gXY = P.mul(P.poly(self.rXY), P.poly(gXY))
        p1XY = P.mul(P.sub(P.poly(self.rXY), P.scalar(fr(1))), P.poly(lagrange_KL_XY))
        p2XY = P.mul_x_minus_one(P.sub(r_gXY, P.mul(P.poly(r_omegaX), P.poly(fXY))))
        p3XY = P.mul(P.poly(lagrange_K0_XY), P.sub(r_gXY, P.mul(P.poly(r_omegaX_omegaY), P.poly(fXY))))
        expr = P.weighted_sum([(fr(1), p1XY), (fr(kappa0), p2XY), (fr(kappa0_sq), p3XY)])
        p_comb = expr.evaluate_fused_with_domain(4 * m_i, 2 * s_max)
        self.q[2], self.q[3] = p_comb.div_by_vanishing_opt(m_i, s_max)
        if self.testing_mode:
            self._rand_point_check(p_comb, [(self.q[2], lambda x, y: pow(x, m_i, R) - 1), (self.q[3], lambda x, y: pow(y, s_max, R) - 1)],
                                   "combined copy-constraint quotient relation")
        r_D1, r_D2, g_D = self.rXY - r_omegaX, self.rXY - r_omegaX_omegaY, gXY - fXY

        def q_c(quot, rB, rR, linear):
            d1_comb = linear(r_D1, rB) + g_D.mul_scalar(fr(rR))
            d2_comb = linear(r_D2, rB) + g_D.mul_scalar(fr(rR))
            return poly_comb((1, quot), (rR, lagrange_KL_XY), (kappa0, mul_by_x_minus_one(d1_comb)), (kappa0_sq, lagrange_K0_XY * d2_comb))

        Q_CX_XY = q_c(self.q[2], mx["rB_X"], mx["rR_X"], mul_by_linear_x)
        Q_CY_XY = q_c(self.q[3], mx["rB_Y"], mx["rR_Y"], mul_by_linear_y)
        Q_CX, Q_CY = self.sigma1.encode_polys([Q_CX_XY, Q_CY_XY])
        return {"Q_CX": Q_CX, "Q_CY": Q_CY}

    # ------------------------------------------------------------------ prove3 (lib.rs:2272-2354)
    def prove3(self, chi, zeta):
        sp, mx = self.setup_params, self.mixer
        m_i, s_max = self.m_i, sp["s_max"]
        c, z = fr(chi), fr(zeta)
        VXY = poly_comb((1, self.vXY), (mx["rV_X"], self.t_n), (mx["rV_Y"], self.t_smax))
        V_eval = fr_int(VXY.eval(c, z))
        RXY = self.rXY + (self.t_mi.mul_scalar(fr(mx["rR_X"])) + self.t_smax.mul_scalar(fr(mx["rR_Y"])))
        R_eval = fr_int(RXY.eval(c, z))
        w_inv_x = pow(fr_int(tkmk.get_root_of_unity(m_i)), R - 2, R)
        w_inv_y = pow(fr_int(tkmk.get_root_of_unity(s_max)), R - 2, R)
        R_omegaX_XY = RXY.scale_coeffs_x(fr(w_inv_x))
        R_omegaX_eval = fr_int(R_omegaX_XY.eval(c, z))
        R_omegaX_omegaY_eval = fr_int(R_omegaX_XY.scale_coeffs_y(fr(w_inv_y)).eval(c, z))
        return {"V_eval": V_eval, "R_eval": R_eval, "R_omegaX_eval": R_omegaX_eval, "R_omegaX_omegaY_eval": R_omegaX_omegaY_eval}

    # ------------------------------------------------------------------ prove4 (lib.rs:2356-3206)
    def prove4(self, proof3, thetas, kappa0, chi, zeta, kappa1):
        sp, mx = self.setup_params, self.mixer
        m_i, s_max, n = self.m_i, sp["s_max"], sp["n"]
        c, z = fr(chi), fr(zeta)
        ev = lambda p: fr_int(p.eval(c, z))                                           # noqa: E731
        neg = lambda v: (R - v % R) % R                                               # noqa: E731

        # --- Pi_A: arithmetic constraints + KZG opening of V (lib.rs:2383-2532)
        t_n_eval = fr_int(self.t_n.eval(c, fr(1)))
        t_smax_eval = fr_int(self.t_smax.eval(fr(1), z))
        small_v_eval = ev(self.vXY)
        rW_X = _coeffs(mx["rW_X"], len(mx["rW_X"]), 1)
        rW_Y = _coeffs(mx["rW_Y"], 1, len(mx["rW_Y"]))
        W_zk = self.cache.get("w_zk")
        if W_zk is None:
            W_zk = low_degree_x_times_vanishing(mx["rW_X"], n) + low_degree_y_times_vanishing(mx["rW_Y"], s_max)
        VXY = poly_comb((1, self.vXY), (mx["rV_X"], self.t_n), (mx["rV_Y"], self.t_smax))
        pA_XY = poly_comb(
            (kappa1, VXY.sub_scalar(fr(proof3["V_eval"]))),
            (small_v_eval, self.uXY), (R - 1, self.wXY),
            (neg(t_n_eval), self.q[0]), (neg(t_smax_eval), self.q[1]),
            (small_v_eval * mx["rU_X"] % R, self.t_n), (small_v_eval * mx["rU_Y"] % R, self.t_smax),
            (neg(mx["rU_X"] * t_n_eval + mx["rU_Y"] * t_smax_eval), self.vXY),
            (t_n_eval, rW_X), (t_smax_eval, rW_Y), (R - 1, W_zk))
        Pi_AX_XY, Pi_AY_XY, rem_A = pA_XY.div_by_ruffini(c, z)

        # --- M, N: openings of R at (chi / w_x, zeta) and (chi / w_x, zeta / w_y) (lib.rs:2534-2701)
        w_inv_x = pow(fr_int(tkmk.get_root_of_unity(m_i)), R - 2, R)
        w_inv_y = pow(fr_int(tkmk.get_root_of_unity(s_max)), R - 2, R)
        RXY = self.rXY + (self.t_mi.mul_scalar(fr(mx["rR_X"])) + self.t_smax.mul_scalar(fr(mx["rR_Y"])))
        M_numerator = RXY.sub_scalar(fr(proof3["R_omegaX_eval"]))
        M_X_XY, M_Y_XY, rem_M = M_numerator.div_by_ruffini(fr(w_inv_x * chi), z)
        N_numerator = RXY.sub_scalar(fr(proof3["R_omegaX_omegaY_eval"]))
        N_X_XY, N_Y_XY, rem_N = N_numerator.div_by_ruffini(fr(w_inv_x * chi), fr(w_inv_y * zeta))

        # --- Pi_C: copy constraints (lib.rs:2703-3130)
        r_omegaX = self.rXY.scale_coeffs_x(fr(w_inv_x))
        r_omegaX_omegaY = r_omegaX.scale_coeffs_y(fr(w_inv_y))
        fXY, gXY = self._fg(thetas)
        t_mi_eval = (pow(chi, m_i, R) - 1) % R
        t_s_max_eval = (pow(zeta, s_max, R) - 1) % R
        lagrange_K0_XY = _unit_evals(m_i, 0, True)
        lagrange_K0_eval = ev(lagrange_K0_XY)
        small_r_eval, small_r_omegaX_eval, small_r_omegaX_omegaY_eval = ev(self.rXY), ev(r_omegaX), ev(r_omegaX_omegaY)
        lagrange_KL_XY = self.cache.get("lagrange_kl_xy")
        if lagrange_KL_XY is None:
            lagrange_KL_XY = _unit_evals(m_i, m_i - 1, True) * _unit_evals(s_max, s_max - 1, False)
        term5 = poly_comb((small_r_eval, gXY), (neg(small_r_omegaX_eval), fXY))
        term6 = poly_comb((small_r_eval, gXY), (neg(small_r_omegaX_omegaY_eval), fXY))
        pC_XY = poly_comb(((small_r_eval - 1) % R, lagrange_KL_XY), (kappa0 * (chi - 1) % R, term5),
                          (kappa0 * kappa0 % R * lagrange_K0_eval % R, term6), (neg(t_mi_eval), self.q[2]), (neg(t_s_max_eval), self.q[3]))
        r_D1, r_D2 = self.rXY - r_omegaX, self.rXY - r_omegaX_omegaY
        r_D1_eval, r_D2_eval = ev(r_D1), ev(r_D2)
        term_B_zk = self.cache.get("term_b_zk")
        if term_B_zk is None:
            term_B_zk = low_degree_x_times_vanishing(mx["rB_X"], m_i) + low_degree_y_times_vanishing(mx["rB_Y"], s_max)
        g_minus_f = gXY - fXY
        term10 = g_minus_f.mul_scalar(fr(mx["rR_X"] * t_mi_eval + mx["rR_Y"] * t_s_max_eval))
        r_d1_t = mul_by_term9(r_D1, mx["rB_X"], mx["rB_Y"], t_mi_eval, t_s_max_eval) + term10
        LHS_zk1 = poly_comb(((chi - 1) * r_D1_eval % R, term_B_zk), (1, mul_by_one_minus_x(r_d1_t)), ((chi - 1) % R, term10))
        r_d2_t = mul_by_term9(r_D2, mx["rB_X"], mx["rB_Y"], t_mi_eval, t_s_max_eval) + term10
        LHS_zk2 = poly_comb((lagrange_K0_eval * r_D2_eval % R, term_B_zk), (lagrange_K0_eval, term10), (R - 1, lagrange_K0_XY * r_d2_t))
        R_minus_eval = RXY.sub_scalar(fr(proof3["R_eval"]))
        k1_2 = kappa1 * kappa1 % R
        LHS_for_copy = poly_comb((k1_2, pC_XY), (k1_2 * kappa0 % R, LHS_zk1), (k1_2 * kappa0 % R * kappa0 % R, LHS_zk2),
                                 (k1_2 * kappa1 % R, R_minus_eval))
        Pi_CX_XY, Pi_CY_XY, rem_C = LHS_for_copy.div_by_ruffini(c, z)

        # --- Pi_B: opening of a_free (lib.rs:3137-3181)
        A_eval = ev(self.a_free_X)
        pi_B_XY, _, rem_B = self.a_free_X.sub_scalar(fr(A_eval)).div_by_ruffini(c, z)

        if self.testing_mode:                                                         # lib.rs:2591-2600, 2658-2667, 3087-3096
            for name, rem in (("Pi_A", rem_A), ("M", rem_M), ("N", rem_N), ("Pi_C", rem_C), ("Pi_B", rem_B)):
                assert not np.asarray(rem).any(), "non-zero Ruffini remainder for " + name
            self._rand_point_check(LHS_for_copy, [(Pi_CX_XY, lambda x, y: x - chi), (Pi_CY_XY, lambda x, y: y - zeta)], "Pi_C quotient identity")

        Pi_AX, Pi_AY, M_X, M_Y, N_X, N_Y, Pi_CX, Pi_CY, Pi_B0 = self.sigma1.encode_polys(
            [Pi_AX_XY, Pi_AY_XY, M_X_XY, M_Y_XY, N_X_XY, N_Y_XY, Pi_CX_XY, Pi_CY_XY, pi_B_XY])
        k1_4 = k1_2 * k1_2 % R
        Pi_B, Pi_X, Pi_Y = g1_lincombs([[(k1_4, Pi_B0), (0, Pi_B0), (0, Pi_B0)],     # encode(pi_B) * kappa1^4 (lib.rs:3180)
                                        [(1, Pi_AX), (1, Pi_CX), (k1_4, Pi_B0)],      # lib.rs:3183-3184
                                        [(1, Pi_AY), (1, Pi_CY), (0, Pi_AY)]])
        proof4 = {"Pi_X": Pi_X, "Pi_Y": Pi_Y, "M_X": M_X, "M_Y": M_Y, "N_X": N_X, "N_Y": N_Y}
        proof4_test = {"Pi_CX": Pi_CX, "Pi_CY": Pi_CY, "Pi_AX": Pi_AX, "Pi_AY": Pi_AY, "Pi_B": Pi_B, "M_X": M_X, "M_Y": M_Y, "N_X": N_X, "N_Y": N_Y}
        return proof4, proof4_test


def run_rounds(prover, binding):
    """the round loop of prove/src/main.rs:47-76 -> (points, scalars, challenges, proof4_test, seconds per round)"""
    manager = TranscriptManager()
    times = {}

    def timed(name, fn):
        tkmk.synchronize()
        t = time.perf_counter()
        out = fn()
        tkmk.synchronize()
        times[name] = time.perf_counter() - t
        return out

    proof0 = timed("prove0", prover.prove0)
    manager.add_proof0(*(proof0[k] for k in ("U", "V", "W", "Q_AX", "Q_AY", "B")))
    thetas = manager.get_thetas()
    proof1 = timed("prove1", lambda: prover.prove1(thetas))
    manager.add_proof1(proof1["R"])
    kappa0 = manager.get_kappa0()
    proof2 = timed("prove2", lambda: prover.prove2(thetas, kappa0))
    manager.add_proof2(proof2["Q_CX"], proof2["Q_CY"])
    chi, zeta = manager.get_chi_zeta()
    proof3 = timed("prove3", lambda: prover.prove3(chi, zeta))
    manager.add_proof3(proof3["V_eval"], proof3["R_eval"], proof3["R_omegaX_eval"], proof3["R_omegaX_omegaY_eval"])
    kappa1 = manager.get_kappa1()
    proof4, proof4_test = timed("prove4", lambda: prover.prove4(proof3, thetas, kappa0, chi, zeta, kappa1))
    points = dict(binding)
    for part in (proof0, proof1, proof2, proof4):
        points.update(part)
    challenges = {"thetas": thetas, "kappa0": kappa0, "chi": chi, "zeta": zeta, "kappa1": kappa1}
    return points, dict(proof3), challenges, proof4_test, times


def prove(qap_path, synthesizer_path, setup_path, output_path, **kw):
    """main() of prove/src/main.rs: init, five rounds, proof.json in the Solidity-verifier format"""
    prover, binding = Prover.init(qap_path, synthesizer_path, setup_path, **kw)
    points, scalars, challenges, proof4_test, times = run_rounds(prover, binding)
    os.makedirs(output_path, exist_ok=True)
    proofio.write_json(os.path.join(output_path, "proof.json"), proofio.format_proof(points, scalars))
    return points, scalars, challenges, proof4_test, dict(prover.timing, **times)
