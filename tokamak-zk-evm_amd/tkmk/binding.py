"""Binding commitments over witness wires: the work-alikes of encode_O_pub_free / encode_O_mid_no_zk /
encode_O_prv_no_zk (packages/backend/libs/src/group_structures/mod.rs:184-300, 607-700).  The reference walks the nested
CRS tables on the host and copies one G1Affine per wire into a fresh Vec before every MSM; here the tables stay in HBM
(flattened row-major, [global_idx][placement]), the host only builds the index / scalar lists from the JSON inputs (hex
parsing as in libs/src/iotools/mod.rs:126-146) and the bases are gathered on the device (tkmk_gather_rows_device)."""
import numpy as np

import tkmk
from tkmk.r1cs import hex_to_fr

_PUB_RANGES = {"bufferPubOut": "Out_idx", "bufferPubIn": "In_idx", "bufferBlockIn": "In_idx"}


def _msm_gathered(table, indices, scalars):
    """msm_g1_bases (:127-143) over table rows `indices`; empty input -> G1serde::zero()"""
    if len(indices) != len(scalars):
        raise ValueError("msm input length mismatch")
    if not indices:
        return np.zeros(96, np.uint8)
    sc = np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in scalars), np.uint8).copy()
    if isinstance(table, tkmk.DeviceBuffer):
        bases = tkmk.gather_rows_device(table, 96, np.asarray(indices, np.uint32))
        return tkmk.projective_to_affine_bytes(tkmk.msm(tkmk.DeviceBuffer.from_host(sc), bases))
    t = np.asarray(table, np.uint8).reshape(-1, 96)
    return tkmk.projective_to_affine_bytes(tkmk.msm(sc, np.ascontiguousarray(t[np.asarray(indices, np.int64)]).reshape(-1)))


def encode_O_pub_free(gamma_inv_o_inst, placement_variables, subcircuit_infos, setup_params):
    """:184-229: interface wires of the public buffers, bases gamma_inv_o_inst[flattenMap[j]]"""
    idx, wt = [], []
    for pl in placement_variables:
        info = subcircuit_infos[pl["subcircuitId"]]
        key = _PUB_RANGES.get(info["name"])
        if key is None:                                  # bufferEVMIn and every non-buffer subcircuit are skipped
            continue
        start, cnt = info[key][0], info[key][1]
        for j in range(start, start + cnt):
            wt.append(hex_to_fr(pl["variables"][j]))
            idx.append(info["flattenMap"][j])
    return _msm_gathered(gamma_inv_o_inst, idx, wt)


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


def encode_statement(offset, end, n_var, placement_variables, subcircuit_infos, table, inner):
    """encode_statement_common (:266-300): every wire j of placement i whose global index lies in [offset, end) contributes
    variables[j] * table[global - offset][i]; table is flattened with `inner` entries per global index"""
    idx, wt = [], []
    for i, pl in enumerate(placement_variables):
        info = subcircuit_infos[pl["subcircuitId"]]
        fm = info["flattenMap"]
        for j in range(info["Nwires"]):
            if offset <= fm[j] < end:
                wt.append(hex_to_fr(pl["variables"][j]))
                idx.append((fm[j] - offset) * inner + i)
    if len(idx) != n_var:
        raise ValueError("nVar mismatch while encoding statement: aligned_rs.len()=%d, nVar=%d" % (len(idx), n_var))
    return _msm_gathered(table, idx, wt)


def encode_O_mid_no_zk(eta_inv_li_o_inter_alpha4_kj, placement_variables, subcircuit_infos, setup_params):
    """:634-648: intermediate wires [l, l_D)"""
    return encode_statement(setup_params["l"], setup_params["l_D"], count_o_mid_nvar(placement_variables, subcircuit_infos),
                            placement_variables, subcircuit_infos, eta_inv_li_o_inter_alpha4_kj, setup_params["s_max"])


def encode_O_prv_no_zk(delta_inv_li_o_prv, placement_variables, subcircuit_infos, setup_params):
    """:693-707: private wires [l_D, m_D)"""
    return encode_statement(setup_params["l_D"], setup_params["m_D"], count_o_prv_nvar(placement_variables, subcircuit_infos),
                            placement_variables, subcircuit_infos, delta_inv_li_o_prv, setup_params["s_max"])
