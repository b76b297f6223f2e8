"""gpu tier: binding commitments over witness wires (tkmk/binding.py; reference libs/src/group_structures/mod.rs:184-300,
607-707) on synthetic subcircuit descriptions and CRS tables [k]G with known k: each commitment must equal
[sum_j w_j k_j]G computed with plain integers, for device-resident and host-resident tables alike."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(gpu, oracle):
    rnd = random.Random(12)
    R = oracle.R_MOD
    sp = {"l": 6, "l_free": 4, "l_user": 3, "l_user_out": 1, "l_D": 14, "m_D": 22, "n": 8, "s_max": 4, "s_D": 5}
    # local wire j of subcircuit k -> global wire flattenMap[j]; Out_idx / In_idx = [start, count] over local wires
    infos = [
        {"id": 0, "name": "bufferPubOut", "Nwires": 4, "Nconsts": 1, "Out_idx": [1, 2], "In_idx": [3, 1], "flattenMap": [6, 0, 1, 7]},
        {"id": 1, "name": "bufferPubIn", "Nwires": 4, "Nconsts": 1, "Out_idx": [1, 1], "In_idx": [2, 2], "flattenMap": [6, 8, 2, 3]},
        {"id": 2, "name": "bufferBlockIn", "Nwires": 3, "Nconsts": 1, "Out_idx": [1, 1], "In_idx": [2, 1], "flattenMap": [6, 9, 4]},
        {"id": 3, "name": "bufferEVMIn", "Nwires": 3, "Nconsts": 1, "Out_idx": [1, 1], "In_idx": [2, 1], "flattenMap": [6, 10, 5]},
        {"id": 4, "name": "ADD", "Nwires": 7, "Nconsts": 1, "Out_idx": [1, 1], "In_idx": [2, 2], "flattenMap": [6, 11, 12, 13, 14, 15, 16]},
    ]
    placements = []
    for sid in (0, 1, 2, 3, 4, 4):
        n = infos[sid]["Nwires"]
        vals = [rnd.randrange(R) for _ in range(n)]
        vals[0] = 1
        if sid == 4:
            vals[4] = 0
        placements.append({"subcircuitId": sid, "variables": ["0x%x" % v for v in vals]})
    g = oracle.g1_generator()

    def table(n):
        ks = [rnd.randrange(1, R) for _ in range(n)]
        return ks, gpu.g1_batch_scalar_mul_device(gpu.DeviceBuffer.from_host(oracle.to_bytes(ks, 32)), g, n)

    return sp, infos, placements, g, table, R


def _pt(oracle, g, dot):
    return oracle.g1_scalar_mul(oracle.to_bytes([dot % oracle.R_MOD], 32), g.copy())


def test_binding_commitments(gpu, oracle):
    from tkmk import binding
    sp, infos, pls, g, table, R = _setup(gpu, oracle)
    val = lambda pl, j: int(pl["variables"][j], 16)   # noqa: E731
    # O_pub_free: interface ranges of the three public buffers, bases gamma[flattenMap[j]]
    kg, gamma = table(sp["l"])
    want = 0
    for pl in pls:
        info = infos[pl["subcircuitId"]]
        rng = {"bufferPubOut": info["Out_idx"], "bufferPubIn": info["In_idx"], "bufferBlockIn": info["In_idx"]}.get(info["name"])
        if rng:
            for j in range(rng[0], rng[0] + rng[1]):
                want += val(pl, j) * kg[info["flattenMap"][j]]
    got = binding.encode_O_pub_free(gamma, pls, infos, sp)
    assert (got == _pt(oracle, g, want)).all()
    assert (binding.encode_O_pub_free(gamma.to_host(), pls, infos, sp) == got).all()
    assert not binding.encode_O_pub_free(gamma, pls[3:], infos, sp).any()          # no public buffers -> G1serde::zero()
    # O_mid: wires in [l, l_D) with table[(global - l)][placement]; O_prv: wires in [l_D, m_D)
    # (placements exceed s_max = 4 in the full list: use the first four for the statement commitments)
    sub = pls[:2] + pls[4:]
    for fn, lo, hi, cnt in ((binding.encode_O_mid_no_zk, sp["l"], sp["l_D"], binding.count_o_mid_nvar),
                            (binding.encode_O_prv_no_zk, sp["l_D"], sp["m_D"], binding.count_o_prv_nvar)):
        kt, tab = table((hi - lo) * sp["s_max"])
        want, n = 0, 0
        for i, pl in enumerate(sub):
            info = infos[pl["subcircuitId"]]
            for j in range(info["Nwires"]):
                gidx = info["flattenMap"][j]
                if lo <= gidx < hi:
                    want += val(pl, j) * kt[(gidx - lo) * sp["s_max"] + i]
                    n += 1
        assert cnt(sub, infos) == n
        got = fn(tab, sub, infos, sp)
        assert (got == _pt(oracle, g, want)).all()
        assert (fn(tab.to_host(), sub, infos, sp) == got).all()
    # nVar mismatch is an error, as in the reference (panic at :291-297)
    bad = [dict(i) for i in infos]
    bad[4] = dict(bad[4], In_idx=[2, 3])
    with pytest.raises(ValueError):
        binding.encode_O_mid_no_zk(tab, sub, bad, sp)
