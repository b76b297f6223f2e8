"""CPU tier: the reference restatement of the prover used by tests/test_gpu_prove.py (tests/prove_ref.py) is itself
checked before it is trusted: on random satisfying circuits (tests/synth_circuit.py) the five Ruffini remainders are zero
and ALL of the verifier's equations — arithmetic, copy, binding and the combined verify_snark (packages/backend/verify-rust/
src/lib.rs:154-352) — hold on the discrete logarithms, over a CRS from the restated trusted setup (Sigma::gen with the fixed tau);
tampering with any proof element, evaluation or public input breaks them.  Also the file formats the
generator writes are read back by the product's host-side readers (no device call)."""
import json
import os
import random

import pytest


def _crs(oracle, inst):
    """the restated fixed-tau setup (prove_ref.sigma_gen): discrete logarithms of every CRS entry"""
    import prove_ref
    pins = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "pins.json")))
    tau = {k: int(pins["tau_" + k], 16) for k in ("x", "y", "alpha", "gamma", "delta", "eta")}
    g = oracle.to_bytes([int(pins["fixed_tau_g1_x"], 16), int(pins["fixed_tau_g1_y"], 16)], 48)
    return prove_ref.sigma_gen(inst, tau), g


@pytest.mark.parametrize("seed,shape", [(1, dict(s_max=8)), (2, dict(s_max=8, n_gate_kinds=3, n_out=1, n_in=2, n_prv=11, used_placements=6, bit_fraction=0.6)),
                                        (3, dict(s_max=16, n_gate_kinds=1, n_out=3, n_in=2, n_prv=2, k_pub=3)),
                                        (4, dict(s_max=8, n_gate_kinds=2, used_placements=6, k_out=0, k_pub=3)),   # l_user_out = 0: an empty bufferPubOut
                                        (5, dict(s_max=8, n_gate_kinds=2, n_out=1, n_in=2, n_prv=1, k_out=0, k_pub=2, l_extra=1, used_placements=7))])   # n = 2
def test_restated_prover_verifies(oracle, tmp_path, seed, shape):
    import prove_ref
    import synth_circuit
    from tkmk.prove import random_mixer
    R = oracle.R_MOD
    rnd = random.Random(seed)
    inst = synth_circuit.build(str(tmp_path), rnd, **shape)
    sp = inst["setup_params"]
    crs, g = _crs(oracle, inst)
    d, s, ch, p4t, rp = prove_ref.run(inst, crs, random_mixer(random.Random(seed)), g)
    assert rp.r1cs_satisfied()
    assert all(v == 0 for v in rp.remainders.values())
    pre = prove_ref.preprocess(rp, inst, crs)
    s0c, s1c, klc = pre["s0"], pre["s1"], crs["lagrange_KL"]
    assert klc == rp.commit(rp.KL)                       # Sigma::gen's lagrange_KL is the commitment of K_{m_I-1}(X) L_{s_max-1}(Y)
    k2 = rnd.randrange(1, R)
    assert prove_ref.verify_arith(d, s, ch, p4t, crs, sp)
    assert prove_ref.verify_copy(d, s, ch, p4t, crs, sp, s0c, s1c, klc, k2)
    assert prove_ref.verify_binding(d, s, ch, p4t, crs, pre, rp.a_free, k2)
    assert prove_ref.verify_snark(d, s, ch, crs, sp, pre, rp.a_free, k2)
    for name in ("A_free", "O_pub_free", "O_mid", "O_prv", "B", "U", "V", "W"):
        assert not prove_ref.verify_binding(dict(d, **{name: (d[name] + 1) % R}), s, ch, p4t, crs, pre, rp.a_free, k2), name
    for name in ("Pi_X", "Pi_Y", "M_X", "M_Y", "N_X", "N_Y", "O_prv", "Q_CX"):
        assert not prove_ref.verify_snark(dict(d, **{name: (d[name] + 1) % R}), s, ch, crs, sp, pre, rp.a_free, k2), name
    # a different public input than the one proven does not verify
    other = prove_ref.P(rp.a_free.c.copy())
    other.c[0, 0] = (other.c[0, 0] + 1) % R
    assert not prove_ref.verify_binding(d, s, ch, p4t, crs, pre, other, k2)
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
    inst = synth_circuit.build(str(tmp_path), rnd, s_max=8, n_gate_kinds=1, used_placements=6)
    sub = inst["subs"][inst["placement_variables"][4]["subcircuitId"]]
    inst["placement_variables"][4]["variables"] = list(inst["placement_variables"][4]["variables"])
    inst["placement_variables"][4]["variables"][list(sub.prvs())[0]] = "0x5"
    crs, g = _crs(oracle, inst)
    d, s, ch, p4t, rp = prove_ref.run(inst, crs, random_mixer(random.Random(5)), g)
    assert not rp.r1cs_satisfied()
    assert rp.remainders["Pi_A"] != 0
    assert not prove_ref.verify_arith(d, s, ch, p4t, crs, inst["setup_params"])


def test_generated_files_are_read_by_the_host_readers(tmp_path):
    import synth_circuit
    from tkmk.r1cs import R_MOD, R1csBinary, SubcircuitR1CS
    inst = synth_circuit.build(str(tmp_path), random.Random(9), s_max=8)
    sp = inst["setup_params"]
    for info, sub in zip(inst["infos"], inst["subs"]):
        path = os.path.join(inst["qap"], "r1cs", "subcircuit%d.r1cs" % info["id"])
        assert R1csBinary.read(path).prime() == R_MOD
        r = SubcircuitR1CS.from_r1cs_sparse_only(path, sp, info)
        assert (r.n_wires, r.n_constraints) == (sub.n_wires, len(sub.rows))
    m_i = sp["l_D"] - sp["l"]
    assert m_i & (m_i - 1) == 0 and all(0 <= e["row"] < m_i and 0 <= e["X"] < m_i for e in inst["permutation"])


def test_validate_setup_shape_messages():
    """the reference's panics of setup_shape / validate_setup_shape (libs/src/utils/mod.rs:21-46)"""
    import pytest
    from tkmk.prove import validate_setup_shape
    good = {"l": 8, "l_D": 40, "n": 16, "s_max": 8, "l_free": 4}
    assert validate_setup_shape(good) == 32
    for change, msg in (({"n": 12}, "n is not a power of two."), ({"s_max": 6}, "s_max is not a power of two."),
                        ({"l_D": 41}, "m_I is not a power of two."), ({"l_D": 7}, "Invalid setup params: l_D must be >= l.")):
        with pytest.raises(ValueError, match=msg.replace(".", r"\.").replace(">", r"\>")):
            validate_setup_shape(dict(good, **change))
