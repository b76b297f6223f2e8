"""Binding commitments over witness wires: the work-alikes of encode_O_pub_free / encode_O_mid_no_zk /
encode_O_prv_no_zk (packages/backend/libs/src/group_structures/mod.rs:184-300, 607-700).  The reference walks the nested
CRS tables on the host and copies one G1Affine per wire into a fresh Vec before every MSM; here the tables stay in HBM
(flattened row-major, [global_idx][placement]), the host only builds the index / scalar lists from the JSON inputs (hex
parsing as in libs/src/iotools/mod.rs:126-146) and the bases are gathered on the device (tkmk_gather_rows_device)."""
import numpy as np

import tkmk

_PUB_RANGES = {"bufferPubOut": "Out_idx", "bufferPubIn": "In_idx", "bufferBlockIn": "In_idx"}


def _msm_gathered(table, indices, scalars):
    """msm_g1_bases (:127-143) over table rows `indices`; empty input -> G1serde::zero()"""
    if len(indices) != len(scalars):
        raise ValueError("msm input length mismatch")
    if not len(indices):
        return np.zeros(96, np.uint8)
    if isinstance(scalars, np.ndarray):                  # (K, 32) little-endian records straight from PlacementValues
        sc = np.ascontiguousarray(scalars, np.uint8).reshape(-1)
    else:
        sc = np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in scalars), np.uint8).copy()
    if isinstance(table, tkmk.DeviceBuffer):
        bases = tkmk.gather_rows_device(table, 96, np.asarray(indices, np.uint32))
        return tkmk.projective_to_affine_bytes(tkmk.msm(tkmk.DeviceBuffer.from_host(sc), bases))
    t = np.asarray(table, np.uint8).reshape(-1, 96)
    return tkmk.projective_to_affine_bytes(tkmk.msm(sc, np.ascontiguousarray(t[np.asarray(indices, np.int64)]).reshape(-1)))


def _values(placement_variables, values):
    from tkmk.r1cs import PlacementValues
    return PlacementValues(placement_variables) if values is None else values


def _cat(parts, width):
    if not parts:
        return np.zeros((0, width), np.uint8) if width else np.zeros(0, np.int64)
    return np.concatenate(parts)


def encode_O_pub_free(gamma_inv_o_inst, placement_variables, subcircuit_infos, setup_params, values=None):
    """:184-229: interface wires of the public buffers, bases gamma_inv_o_inst[flattenMap[j]]"""
    values = _values(placement_variables, values)
    idx, wt = [], []
    for i, pl in enumerate(placement_variables):
        info = subcircuit_infos[pl["subcircuitId"]]
        key = _PUB_RANGES.get(info["name"])
        if key is None:                                  # bufferEVMIn and every non-buffer subcircuit are skipped
            continue
        start, cnt = info[key][0], info[key][1]
        wt.append(values[i][start:start + cnt])
        idx.append(np.asarray(info["flattenMap"][start:start + cnt], np.int64))
    return _msm_gathered(gamma_inv_o_inst, _cat(idx, 0), _cat(wt, 32))


def count_o_mid_nvar(placement_variables, subcircuit_infos):
    """:231-251"""
    n = 0
    for pl in placement_variables:
        info = subcircuit_infos[pl["subcircuitId"]]
        if info["name"] == "bufferPubOut":
            n += info["In_idx"][1]
        elif info["name"] in ("bufferPubIn", "bufferBlockIn", "bufferEVMIn"):
            n += info["Out_idx"][1]
        else:
            n += info["Out_idx"][1] + info["In_idx"][1]
        n += 1                                           # each constant wire
    return n


def count_o_prv_nvar(placement_variables, subcircuit_infos):
    """:253-264"""
    n = 0
    for pl in placement_variables:
        info = subcircuit_infos[pl["subcircuitId"]]
        n += info["Nwires"] - info["In_idx"][1] - info["Out_idx"][1] - 1
    return n


def encode_statement(offset, end, n_var, placement_variables, subcircuit_infos, table, inner, values=None):
    """encode_statement_common (:266-300): every wire j of placement i whose global index lies in [offset, end) contributes
    variables[j] * table[global - offset][i]; table is flattened with `inner` entries per global index"""
    values = _values(placement_variables, values)
    idx, wt, sel = [], [], {}
    for i, pl in enumerate(placement_variables):
        sid = pl["subcircuitId"]
        if sid not in sel:
            info = subcircuit_infos[sid]
            fm = np.asarray(info["flattenMap"][:info["Nwires"]], np.int64)
            loc = np.nonzero((fm >= offset) & (fm < end))[0]
            sel[sid] = (loc, (fm[loc] - offset) * inner)
        loc, rows = sel[sid]
        wt.append(values[i][loc])
        idx.append(rows + i)
    idx, wt = _cat(idx, 0), _cat(wt, 32)
    if len(idx) != n_var:
        raise ValueError("nVar mismatch while encoding statement: aligned_rs.len()=%d, nVar=%d" % (len(idx), n_var))
    return _msm_gathered(table, idx, wt)


def encode_O_mid_no_zk(eta_inv_li_o_inter_alpha4_kj, placement_variables, subcircuit_infos, setup_params, values=None):
    """:634-648: intermediate wires [l, l_D)"""
    return encode_statement(setup_params["l"], setup_params["l_D"], count_o_mid_nvar(placement_variables, subcircuit_infos),
                            placement_variables, subcircuit_infos, eta_inv_li_o_inter_alpha4_kj, setup_params["s_max"], values)


def encode_O_prv_no_zk(delta_inv_li_o_prv, placement_variables, subcircuit_infos, setup_params, values=None):
    """:693-707: private wires [l_D, m_D)"""
    return encode_statement(setup_params["l_D"], setup_params["m_D"], count_o_prv_nvar(placement_variables, subcircuit_infos),
                            placement_variables, subcircuit_infos, delta_inv_li_o_prv, setup_params["s_max"], values)
