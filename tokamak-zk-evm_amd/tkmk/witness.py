"""Witness / instance / permutation polynomials: work-alikes of gen_bXY and Instance::gen_a_free_X
(packages/backend/libs/src/polynomial_structures/mod.rs:104-162) and Permutation::to_poly
(packages/backend/libs/src/iotools/mod.rs:419-455).  The evaluation matrices are assembled on the host exactly as the
reference does (they are index shuffles of JSON input); the inverse bivariate NTTs run on the device.
"""
import numpy as np

import tkmk
from tkmk.poly import DensePolynomialExt
from tkmk.r1cs import R_MOD, hex_to_fr


def _fr_bytes(vals):
    return np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in vals), np.uint8).copy()


def gen_bXY(placement_variables, subcircuit_infos, setup_params, values=None):
    from tkmk.r1cs import PlacementValues
    l, l_d, s_max = setup_params["l"], setup_params["l_D"], setup_params["s_max"]
    m_i = l_d - l
    infos = {e["id"]: e for e in subcircuit_infos}
    values = PlacementValues(placement_variables) if values is None else values
    w = np.zeros((m_i * s_max, 32), np.uint8)
    sel = {}                                             # per subcircuit: local wires on the interface range and their rows
    for i, pl in enumerate(placement_variables):
        sid = pl["subcircuitId"]
        if sid not in sel:
            fm = np.asarray(infos[sid]["flattenMap"], np.int64)
            loc = np.nonzero((fm >= l) & (fm < l_d))[0]
            sel[sid] = (len(fm), loc, (fm[loc] - l) * s_max)
        n_wires, loc, rows = sel[sid]
        if len(pl["variables"]) != n_wires:
            raise ValueError("Corrupted placement variables.")
        w[rows + i] = values[i][loc]                     # a "0x0" entry is a zero record either way (mod.rs:150-152)
    return DensePolynomialExt.from_rou_evals(w.reshape(-1), m_i, s_max)


def gen_a_free_X(instance, setup_params):
    l_free, l_user = setup_params["l_free"], setup_params["l_user"]
    m_block = l_free - l_user
    vals = [hex_to_fr(instance["a_pub_user"][i]) for i in range(l_user)] + [hex_to_fr(instance["a_pub_block"][i]) for i in range(m_block)]
    return DensePolynomialExt.from_rou_evals(_fr_bytes(vals), l_free, 1)


def permutation_to_poly(perm_raw, m_i, s_max):
    """perm_raw: list of {"row", "col", "X", "Y"}; returns (s0XY, s1XY)"""
    wx = int.from_bytes(tkmk.get_root_of_unity(m_i).tobytes(), "little")
    wy = int.from_bytes(tkmk.get_root_of_unity(s_max).tobytes(), "little")
    xp, yp = [1] * m_i, [1] * s_max
    for i in range(1, m_i):
        xp[i] = xp[i - 1] * wx % R_MOD
    for j in range(1, s_max):
        yp[j] = yp[j - 1] * wy % R_MOD
    xb = _fr_bytes(xp).reshape(m_i, 32)
    yb = _fr_bytes(yp).reshape(s_max, 32)
    s0 = np.repeat(xb[:, None, :], s_max, axis=1).copy()        # s0[row][col] = wx^row
    s1 = np.repeat(yb[None, :, :], m_i, axis=0).copy()          # s1[row][col] = wy^col
    if perm_raw:                                          # one scatter for all entries (mod.rs:440-449 loops over them)
        k = len(perm_raw)
        col = lambda key: np.fromiter((p[key] for p in perm_raw), np.int64, k)          # noqa: E731
        rows, cols = col("row"), col("col")
        s0[rows, cols] = xb[col("X")]
        s1[rows, cols] = yb[col("Y")]
    return (DensePolynomialExt.from_rou_evals(s0.reshape(-1), m_i, s_max), DensePolynomialExt.from_rou_evals(s1.reshape(-1), m_i, s_max))
