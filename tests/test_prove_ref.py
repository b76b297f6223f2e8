"""CPU tier: the reference restatement of the prover used by tests/test_gpu_prove.py (tests/prove_ref.py) is itself
checked before it is trusted: on random satisfying circuits (tests/synth_circuit.py) the five Ruffini remainders are zero
and the verifier's arithmetic and copy equations (packages/backend/verify-rust/src/lib.rs:154-196, 225-246, 291-317) hold
on the discrete logarithms; tampering with any proof element or evaluation breaks them.  Also the file formats the
generator writes are read back by the product's host-side readers (no device call)."""
import json
import os
import random

import pytest


def _crs(oracle, sp, rnd):
    pins = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "pins.json")))
    R = oracle.R_MOD
    m_i, s_max = sp["l_D"] - sp["l"], sp["s_max"]
    rand = lambda k: [rnd.randrange(1, R) for _ in range(k)]                          # noqa: E731
    crs = {"tau_x": int(pins["tau_x"], 16), "tau_y": int(pins["tau_y"], 16), "delta": rnd.randrange(1, R), "eta": rnd.randrange(1, R),
           "gamma_inv_o_inst": rand(sp["l"]), "eta_inv_li_o_inter_alpha4_kj": [rand(s_max) for _ in range(m_i)],
           "delta_inv_li_o_prv": [rand(s_max) for _ in range(sp["m_D"] - sp["l_D"])],
           "delta_inv_alphak_xh_tx": [rand(3) for _ in range(3)], "delta_inv_alpha4_xj_tx": rand(2),
           "delta_inv_alphak_yi_ty": [rand(3) for _ in range(4)]}
    g = oracle.to_bytes([int(pins["fixed_tau_g1_x"], 16), int(pins["fixed_tau_g1_y"], 16)], 48)
    return crs, g


@pytest.mark.parametrize("seed,shape", [(1, dict(s_max=8)), (2, dict(s_max=4, n_gate_kinds=3, n_out=1, n_in=2, n_prv=11, used_placements=3)),
                                        (3, dict(s_max=16, n_gate_kinds=1, n_out=3, n_in=2, n_prv=2, k_pub=3))])
def test_restated_prover_verifies(oracle, tmp_path, seed, shape):
    import prove_ref
    import synth_circuit
    from tkmk.prove import random_mixer
    R = oracle.R_MOD
    rnd = random.Random(seed)
    inst = synth_circuit.build(str(tmp_path), rnd, **shape)
    sp = inst["setup_params"]
    crs, g = _crs(oracle, sp, rnd)
    d, s, ch, p4t, rp = prove_ref.run(inst, crs, random_mixer(random.Random(seed)), g)
    assert rp.r1cs_satisfied()
    assert all(v == 0 for v in rp.remainders.values())
    s0c, s1c, klc = rp.commit(rp.s0), rp.commit(rp.s1), rp.commit(rp.KL)
    k2 = rnd.randrange(1, R)
    assert prove_ref.verify_arith(d, s, ch, p4t, crs, sp)
    assert prove_ref.verify_copy(d, s, ch, p4t, crs, sp, s0c, s1c, klc, k2)
    # Pi_X / Pi_Y are the sums the verifier's snark_aux expects (prove/src/lib.rs:3183-3184)
    assert d["Pi_X"] == (p4t["Pi_AX"] + p4t["Pi_CX"] + p4t["Pi_B"]) % R and d["Pi_Y"] == (p4t["Pi_AY"] + p4t["Pi_CY"]) % R
    for name in ("U", "V", "W", "Q_AX", "Q_AY"):
        assert not prove_ref.verify_arith(dict(d, **{name: (d[name] + 1) % R}), s, ch, p4t, crs, sp), name
    for name in ("B", "R", "Q_CX", "Q_CY"):
        assert not prove_ref.verify_copy(dict(d, **{name: (d[name] + 1) % R}), s, ch, p4t, crs, sp, s0c, s1c, klc, k2), name
    for name in ("R_eval", "R_omegaX_eval", "R_omegaX_omegaY_eval"):
        assert not prove_ref.verify_copy(d, dict(s, **{name: (s[name] + 1) % R}), ch, p4t, crs, sp, s0c, s1c, klc, k2), name
    assert not prove_ref.verify_arith(d, dict(s, V_eval=(s["V_eval"] + 1) % R), ch, p4t, crs, sp)


def test_unsatisfying_witness_leaves_a_remainder(oracle, tmp_path):
    """a witness that violates R1CS must not produce a proof that verifies"""
    import prove_ref
    import synth_circuit
    from tkmk.prove import random_mixer
    rnd = random.Random(5)
    inst = synth_circuit.build(str(tmp_path), rnd, s_max=4, n_gate_kinds=1)
    sub = inst["subs"][inst["placement_variables"][1]["subcircuitId"]]
    inst["placement_variables"][1]["variables"][list(sub.prvs())[0]] = "0x5"
    crs, g = _crs(oracle, inst["setup_params"], rnd)
    d, s, ch, p4t, rp = prove_ref.run(inst, crs, random_mixer(random.Random(5)), g)
    assert not rp.r1cs_satisfied()
    assert rp.remainders["Pi_A"] != 0
    assert not prove_ref.verify_arith(d, s, ch, p4t, crs, inst["setup_params"])


def test_generated_files_are_read_by_the_host_readers(tmp_path):
    import synth_circuit
    from tkmk.r1cs import R_MOD, R1csBinary, SubcircuitR1CS
    inst = synth_circuit.build(str(tmp_path), random.Random(9), s_max=4)
    sp = inst["setup_params"]
    for info, sub in zip(inst["infos"], inst["subs"]):
        path = os.path.join(inst["qap"], "r1cs", "subcircuit%d.r1cs" % info["id"])
        assert R1csBinary.read(path).prime() == R_MOD
        r = SubcircuitR1CS.from_r1cs_sparse_only(path, sp, info)
        assert (r.n_wires, r.n_constraints) == (sub.n_wires, len(sub.rows))
    m_i = sp["l_D"] - sp["l"]
    assert m_i & (m_i - 1) == 0 and all(0 <= e["row"] < m_i and 0 <= e["X"] < m_i for e in inst["permutation"])
