"""CRS generation on the device: work-alike of the trusted setup's evaluation phase and Sigma::gen / Sigma1::gen
(packages/backend/setup/trusted-setup/src/main.rs:99-196, packages/backend/libs/src/group_structures/mod.rs:313-551,
packages/backend/libs/src/field_structures/mod.rs:67-165) for the G1 side of the reference string — everything `prove`,
`preprocess` and the G1 half of `verify` read.

Where the work runs:
  * Lagrange bases at tau (gen_evaled_lagrange_bases, libs/src/vector_operations/mod.rs:19-28): power vector by
    tkmk_poly_scale_coeffs, one inverse NTT;
  * the QAP mixture o_j(tau_x) = alpha u_j + alpha^2 v_j + alpha^3 w_j with u_j(tau_x) = sum_rows A[row][j] L_row(tau_x):
    the transposed sparse matrices times the Lagrange vector through the same kernel that evaluates R1CS rows
    (tkmk_r1cs_eval_rows on the CSC form);
  * the table scalars (outer products with L_i(tau_y), scalings by gamma^-1 / eta^-1 / delta^-1) with the Fr vector ops;
  * every point as one fixed-base batched scalar multiplication (tkmk_g1_batch_scalar_mul_device): 2^22 .. 2^24 of them
    for xy_powers, (m_D - l_D) * s_max for delta_inv_li_o_prv.
Sigma2 (H and its nine multiples by the trapdoor scalars, consumed only by the pairing verifiers) is ten single-point operations;
they run as one batch of one-point G2 MSMs (bls12_381_g2_msm; tkmk/g2.py's big-int version is the check in the tests); without a G2 generator the payload's G2 section is left zero.
"""
import json
import os

import numpy as np

import tkmk
from tkmk import crs as crsmod
from tkmk.r1cs import R_MOD, R1csBinary
from tkmk.sigma import Sigma1

R = R_MOD
TAU_FIELDS = ("x", "y", "alpha", "gamma", "delta", "eta")


def _fr(v):
    return np.frombuffer((int(v) % R).to_bytes(32, "little"), np.uint8).copy()


def _ints(buf):
    b = np.asarray(buf, np.uint8).tobytes()
    return [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]


def _frs(vals):
    return np.frombuffer(b"".join((int(v) % R).to_bytes(32, "little") for v in vals), np.uint8).copy()


def _inv(a):
    return pow(a % R, R - 2, R)


def gen_evaled_lagrange_bases(val, size):
    """-> DeviceBuffer of L_i(val), i < size (libs/src/vector_operations/mod.rs:19-28): inverse NTT of (val^k)_k"""
    one = np.zeros(32, np.uint8)
    one[0] = 1
    ones = tkmk.DeviceBuffer.from_host(np.tile(one, size))
    pows = tkmk.DeviceBuffer(32 * size)
    tkmk._check(tkmk.lib().tkmk_poly_scale_coeffs(tkmk._p(ones), size, 1, tkmk._p(_fr(val)), None, tkmk._p(pows), None), "tkmk_poly_scale_coeffs")
    return tkmk.bintt(pows, size, 1, inverse=True)


def _transposed(csr_m, n_wires):
    """CSR over constraints (row_ptr, wire, coeff) -> the same matrix as CSR over wires (ptr, constraint row, coeff)"""
    ptr, wires, coeff = csr_m
    nnz = int(wires.size)
    rows = np.repeat(np.arange(ptr.size - 1, dtype=np.uint32), np.diff(ptr.astype(np.int64)))
    order = np.argsort(wires, kind="stable")
    t_ptr = np.zeros(n_wires + 1, np.int64)
    np.cumsum(np.bincount(wires, minlength=n_wires), out=t_ptr[1:])
    c = np.asarray(coeff, np.uint8).reshape(-1, 32)[order].reshape(-1) if nnz else np.zeros(0, np.uint8)
    return t_ptr.astype(np.uint32), rows[order].astype(np.uint32), c


def evaled_qap_mixture(qap_path, subcircuit_infos, setup_params, tau):
    """o_evaled_vec of the setup (main.rs:129-160, from_r1cs_to_evaled_qap_mixture): m_D values, as a list of ints"""
    n, m_d = setup_params["n"], setup_params["m_D"]
    x_lag = gen_evaled_lagrange_bases(tau["x"], n)                    # L_row(tau_x)
    alpha = [pow(tau["alpha"], k, R) for k in (1, 2, 3)]
    slot = tkmk.DeviceBuffer.from_host(np.zeros(1, np.uint32).view(np.uint8))
    o_vec = [0] * m_d
    lib = tkmk.lib()
    for info in subcircuit_infos:
        b = R1csBinary.read(os.path.join(qap_path, "r1cs", "subcircuit%d.r1cs" % info["id"]))
        if b.n_wires != info["Nwires"] or b.n_constraints != info["Nconsts"] or b.n_constraints > n:
            raise ValueError("R1CS shape does not match subcircuitInfo / n for subcircuit %d" % info["id"])
        o = [0] * b.n_wires
        for m, csr_m in enumerate(b.csr()):
            t_ptr, t_rows, t_coeff = _transposed(csr_m, b.n_wires)
            if t_rows.size == 0:
                continue
            out = tkmk.DeviceBuffer.from_host(np.zeros(32 * b.n_wires, np.uint8))
            d = [tkmk.DeviceBuffer.from_host(a.view(np.uint8)) for a in (t_ptr, t_rows)] + [tkmk.DeviceBuffer.from_host(t_coeff)]
            # "constraints" = wires, "wires" = constraint rows, "variables" = the Lagrange vector (n entries), one slot
            tkmk._check(lib.tkmk_r1cs_eval_rows(tkmk._p(d[0]), tkmk._p(d[1]), tkmk._p(d[2]), b.n_wires, int(t_rows.size), tkmk._p(x_lag), n, 1,
                                                tkmk._p(slot), b.n_wires, tkmk._p(out), None), "tkmk_r1cs_eval_rows")
            for j, v in enumerate(_ints(out.to_host())):
                if v:
                    o[j] = (o[j] + alpha[m] * v) % R
        fm = info["flattenMap"]
        for j in range(b.n_wires):
            if o[j]:
                o_vec[fm[j]] = o[j]
    return o_vec


def _outer_scaled(col, row_dev, n_row, scale):
    """DeviceBuffer of scale * col[j] * row[i] at [j * n_row + i] (type_scaled_outer_product_2d!)"""
    k = len(col)
    if k == 0:
        return tkmk.DeviceBuffer(32)
    colv = tkmk.DeviceBuffer.from_host(_frs([c * scale % R for c in col]))
    a = tkmk.gather_rows_device(colv, 32, np.repeat(np.arange(k, dtype=np.uint32), n_row))
    b = tkmk.gather_rows_device(row_dev, 32, np.tile(np.arange(n_row, dtype=np.uint32), k))
    return tkmk.vec_mul(a, b, out=a)


class Sigma:
    """the G1 side of the reference string, resident in HBM: `sigma1` (Sigma1 with xy_powers), `tables` (the six G1 tables of
    tkmk/crs.py), `singles` (G, x, y, delta, eta, lagrange_KL as 96-byte records)"""

    def __init__(self, sigma1, tables, singles, setup_params, g2_points=None):
        self.sigma1, self.tables, self.singles, self.setup_params = sigma1, tables, singles, setup_params
        self.g2_points = g2_points          # [H, alpha, alpha2, alpha3, alpha4, gamma, delta, eta, x, y] as tkmk.g2 points, or None

    @classmethod
    def gen(cls, setup_params, tau, qap_path, subcircuit_infos, g1_gen, g2_gen=None):
        sp = setup_params
        n, s_max, l, l_free, l_user, l_user_out, l_d, m_d = (sp[k] for k in ("n", "s_max", "l", "l_free", "l_user", "l_user_out", "l_D", "m_D"))
        from tkmk.prove import validate_setup_shape
        m_i = validate_setup_shape(sp)                    # main.rs:93-94
        if l_free != 0 and l_free & (l_free - 1):        # validate_public_wire_size (libs/src/utils/mod.rs:48-52)
            raise ValueError("l is not a power of two.")
        tkmk.init_ntt_domain_for_size(max(n, l_free, m_i, s_max))      # trusted_setup_ntt_domain_size (libs/src/utils/mod.rs:60-66)
        g = np.ascontiguousarray(g1_gen, np.uint8)
        pts = lambda scal_dev, count: tkmk.g1_batch_scalar_mul_device(scal_dev, g, count)   # noqa: E731
        x, y, a = tau["x"], tau["y"], tau["alpha"]
        gi, di, ei = _inv(tau["gamma"]), _inv(tau["delta"]), _inv(tau["eta"])

        k_dev, l_dev, m_dev = gen_evaled_lagrange_bases(x, m_i), gen_evaled_lagrange_bases(y, s_max), gen_evaled_lagrange_bases(x, l_free)
        k_vec, l_vec, m_vec = _ints(k_dev.to_host()), _ints(l_dev.to_host()), _ints(m_dev.to_host())
        o_vec = evaled_qap_mixture(qap_path, subcircuit_infos, sp, tau)

        # xy_powers[h * 2 s_max + i] = [x^h y^i]G, h < max(2n, 2 m_I)
        h_max, rs_y = max(2 * n, 2 * m_i), 2 * s_max
        one = np.zeros(32, np.uint8)
        one[0] = 1
        mon = tkmk.DeviceBuffer(32 * h_max * rs_y)
        ones = tkmk.DeviceBuffer.from_host(np.tile(one, h_max * rs_y))
        tkmk._check(tkmk.lib().tkmk_poly_scale_coeffs(tkmk._p(ones), h_max, rs_y, tkmk._p(_fr(x)), tkmk._p(_fr(y)), tkmk._p(mon), None),
                    "tkmk_poly_scale_coeffs")
        sigma1 = Sigma1(pts(mon, h_max * rs_y), h_max, rs_y)
        del ones, mon

        # gamma_inv_o_inst (:405-440): the public wires sit in placements 0..3 (L_0 .. L_3 of tau_y); the free ones also carry M_j(tau_x)
        user_vec = [l_vec[0]] * l_user_out + [l_vec[1]] * (l_user - l_user_out) + [l_vec[2]] * (l_free - l_user) + [l_vec[3]] * (l - l_free)
        if len(user_vec) != l:
            raise ValueError("user_vec length mismatch: expected l")
        gamma_scal = [gi * (user_vec[j] * o_vec[j] + (m_vec[j] if j < l_free else 0)) % R for j in range(l)]
        a4 = pow(a, 4, R)
        inter = [(o_vec[l + j] + a4 * k_vec[j]) % R for j in range(m_i)]
        tables = {
            "gamma_inv_o_inst": pts(tkmk.DeviceBuffer.from_host(_frs(gamma_scal)), l),
            "eta_inv_li_o_inter_alpha4_kj": pts(_outer_scaled(inter, l_dev, s_max, ei), m_i * s_max),
            "delta_inv_li_o_prv": pts(_outer_scaled(o_vec[l_d:m_d], l_dev, s_max, di), (m_d - l_d) * s_max),
        }
        t_n, t_mi, t_s = (pow(x, n, R) - 1) % R, (pow(x, m_i, R) - 1) % R, (pow(y, s_max, R) - 1) % R
        small = {"delta_inv_alphak_xh_tx": [di * pow(a, k, R) * pow(x, h, R) * t_n % R for k in (1, 2, 3) for h in range(3)],
                 "delta_inv_alpha4_xj_tx": [di * a4 * pow(x, j, R) * t_mi % R for j in range(2)],
                 "delta_inv_alphak_yi_ty": [di * pow(a, k, R) * pow(y, i, R) * t_s % R for k in (1, 2, 3, 4) for i in range(3)]}
        for name, sc in small.items():                   # read on the host only (crsmod.SMALL_TABLES)
            tables[name] = pts(tkmk.DeviceBuffer.from_host(_frs(sc)), len(sc)).to_host()
        single_scalars = [1, x, y, tau["delta"], tau["eta"], l_vec[s_max - 1] * k_vec[m_i - 1] % R]
        sing = pts(tkmk.DeviceBuffer.from_host(_frs(single_scalars)), 6).to_host().reshape(6, 96)
        singles = {name: sing[i].copy() for i, name in enumerate(crsmod.G1_SINGLES)}
        g2_points = None
        if g2_gen is not None:               # Sigma2::gen (:752-777) and H
            from tkmk import g2
            if g2_gen is None or not g2.on_curve(g2_gen):
                raise ValueError("the G2 generator is not a point of the twist")
            ks = [1, a % R, a * a % R, pow(a, 3, R), pow(a, 4, R), tau["gamma"] % R, tau["delta"] % R, tau["eta"] % R, x % R, y % R]
            res = tkmk.msm_g2(_frs(ks), g2.encode(g2_gen), msm_size=1, batch=10, shared_points=True).reshape(10, 288)
            g2_points = [g2.decode(r[:192]) if r[192:].any() else None for r in res]      # (x, y, 1) or (0, 1, 0)
        return cls(sigma1, tables, singles, sp, g2_points)

    def prover_view(self):
        """the (sigma1, tables, singles) triple Prover.init(sigma=...) takes"""
        return self.sigma1, self.tables, self.singles

    def payload(self):
        """TKCRS001 bytes (tkmk/crs.py); the G2 section is zero (see the module docstring)"""
        if self.g2_points is None:
            g2_section = np.zeros(len(crsmod.G2_POINTS) * crsmod.G2_BYTES, np.uint8)
        else:
            from tkmk import g2
            g2_section = np.concatenate([g2.encode(p) for p in self.g2_points])
        sections = {"g1": np.concatenate([self.singles[k] for k in crsmod.G1_SINGLES]), "xy_powers": self.sigma1.xy_powers.to_host(),
                    "g2": g2_section}
        sections.update({k: (v.to_host() if isinstance(v, tkmk.DeviceBuffer) else np.asarray(v, np.uint8)) for k, v in self.tables.items()})
        return crsmod.build_payload(sections)

    def write(self, out_dir):
        os.makedirs(out_dir, exist_ok=True)
        path = os.path.join(out_dir, "combined_sigma.tkcrs")
        with open(path, "wb") as f:
            f.write(self.payload())
        return path


def trusted_setup(qap_path, out_dir, tau, g1_gen, g2_gen=None):
    """the file-level surface of the trusted setup: <qap_path>/{setupParams.json, subcircuitInfo.json, r1cs/*}
    in, <out_dir>/combined_sigma.tkcrs out"""
    with open(os.path.join(qap_path, "setupParams.json")) as f:
        sp = json.load(f)
    with open(os.path.join(qap_path, "subcircuitInfo.json")) as f:
        infos = json.load(f)
    sigma = Sigma.gen(sp, tau, qap_path, infos, g1_gen, g2_gen)
    return sigma, sigma.write(out_dir)
