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


# ---- thin bindings of the device-side entries of Prover::init (include/tkmk.h "Witness side of the path"); the native host side
# (host/tkmk_service.hpp) is their product caller, tests/test_gpu_witness.py checks each on its own ----
import ctypes


class R1csLibrary:
    """tkmk_r1cs_library_create / _eval / _destroy: csr_per_sub = [(A, B, C)] with each matrix a (row_ptr u32, wire u32, coeff bytes)
    triple (tkmk.r1cs.R1csBinary.csr()); n_rows / n_wires per subcircuit kind"""

    def __init__(self, csr_per_sub, n_rows, n_wires):
        n_sub = len(csr_per_sub)
        self._keep = []
        rp = (ctypes.c_void_p * (3 * n_sub))()
        wi = (ctypes.c_void_p * (3 * n_sub))()
        co = (ctypes.c_void_p * (3 * n_sub))()
        for s, mats in enumerate(csr_per_sub):
            for m, (ptr, wires, coeff) in enumerate(mats):
                a = np.ascontiguousarray(ptr, np.uint32)
                b = np.ascontiguousarray(wires, np.uint32)
                c = np.ascontiguousarray(coeff, np.uint8)
                self._keep += [a, b, c]
                rp[3 * s + m] = a.ctypes.data
                wi[3 * s + m] = b.ctypes.data if b.size else None
                co[3 * s + m] = c.ctypes.data if c.size else None
        nr = np.ascontiguousarray(n_rows, np.uint32)
        nw = np.ascontiguousarray(n_wires, np.uint32)
        self._h = ctypes.c_void_p()
        tkmk._check(tkmk.lib().tkmk_r1cs_library_create(ctypes.c_uint32(n_sub), tkmk._p(nr.view(np.uint8)), tkmk._p(nw.view(np.uint8)), rp, wi, co,
                                                        ctypes.byref(self._h)), "tkmk_r1cs_library_create")

    def eval(self, vars_dev, placement_ids, placement_var_offsets, n, s_max):
        """-> (u, v, w) host bytes, n x s_max evaluation matrices, element (row, placement)"""
        ids = np.ascontiguousarray(placement_ids, np.uint32)
        off = np.ascontiguousarray(placement_var_offsets, np.uint64)
        d_ids = tkmk.DeviceBuffer.from_host(ids.view(np.uint8)) if ids.size else None
        d_off = tkmk.DeviceBuffer.from_host(off.view(np.uint8)) if off.size else None
        outs = [tkmk.DeviceBuffer(32 * n * s_max) for _ in range(3)]
        tkmk._check(tkmk.lib().tkmk_r1cs_library_eval(self._h, tkmk._p(vars_dev), None if d_ids is None else tkmk._p(d_ids), None if d_off is None else tkmk._p(d_off),
                                                      ctypes.c_uint32(ids.size), ctypes.c_uint32(n), ctypes.c_uint32(s_max), tkmk._p(outs[0]), tkmk._p(outs[1]),
                                                      tkmk._p(outs[2]), None), "tkmk_r1cs_library_eval")
        tkmk.synchronize()
        return tuple(o.to_host() for o in outs)

    def close(self):
        if self._h:
            tkmk.lib().tkmk_r1cs_library_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def witness_route(vars_dev, var_offsets, slots, list_wire, list_row, matrix_dev=None, matrix_stride=0, want_lists=False, index_inner=1,
                  index_add_slot=False):
    """tkmk_witness_route for the placements of one subcircuit kind -> (scalars bytes, index u32) when want_lists, else None"""
    off = np.ascontiguousarray(var_offsets, np.uint64)
    sl = np.ascontiguousarray(slots, np.uint32)
    lw = np.ascontiguousarray(list_wire, np.uint32)
    lr = np.ascontiguousarray(list_row, np.uint32)
    dev = lambda a: tkmk.DeviceBuffer.from_host(a.view(np.uint8)) if a.size else None     # noqa: E731
    d_off, d_sl, d_lw, d_lr = dev(off), dev(sl), dev(lw), dev(lr)
    total = off.size * lw.size
    d_sc = tkmk.DeviceBuffer(32 * max(total, 1)) if want_lists else None
    d_ix = tkmk.DeviceBuffer(4 * max(total, 1)) if want_lists else None
    ptr = lambda b: None if b is None else tkmk._p(b)                                       # noqa: E731
    tkmk._check(tkmk.lib().tkmk_witness_route(tkmk._p(vars_dev), ptr(d_off), ptr(d_sl), ctypes.c_uint32(off.size), ptr(d_lw), ptr(d_lr),
                                              ctypes.c_uint32(lw.size), ptr(matrix_dev), ctypes.c_uint32(matrix_stride), ptr(d_sc), ptr(d_ix),
                                              ctypes.c_uint32(index_inner), ctypes.c_int(1 if index_add_slot else 0), None), "tkmk_witness_route")
    tkmk.synchronize()
    if not want_lists:
        return None
    return d_sc.to_host()[:32 * total], d_ix.to_host()[:4 * total].view(np.uint32).copy()


def fr_scatter_table(table_dev, table_len, src_idx, dst_idx, out_dev, out_len):
    """tkmk_fr_scatter_table: out[dst[i]] = table[src[i]]"""
    src = np.ascontiguousarray(src_idx, np.uint32)
    dst = np.ascontiguousarray(dst_idx, np.uint32)
    d_src = tkmk.DeviceBuffer.from_host(src.view(np.uint8)) if src.size else None
    d_dst = tkmk.DeviceBuffer.from_host(dst.view(np.uint8)) if dst.size else None
    tkmk._check(tkmk.lib().tkmk_fr_scatter_table(tkmk._p(table_dev), ctypes.c_uint64(table_len), None if d_src is None else tkmk._p(d_src),
                                                 None if d_dst is None else tkmk._p(d_dst), ctypes.c_uint64(src.size), tkmk._p(out_dev),
                                                 ctypes.c_uint64(out_len), None), "tkmk_fr_scatter_table")
