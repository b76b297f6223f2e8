"""gpu tier: the five prover rounds (tkmk/prove.py, work-alike of Prover::init / prove0..prove4 and the round loop of
packages/backend/prove/src/{lib.rs,main.rs}) end to end over the device path, against tests/prove_ref.py — a big-int
restatement of the same rounds from the reference's un-refactored expressions, in the exponent of a fixed-tau CRS.

What is compared: all 19 proof points (as [dlog]G from the oracle's scalar multiplication), the 4 evaluations, every
Fiat-Shamir challenge, and the Proof4Test points; then the verifier's arithmetic and copy equations
(verify-rust/src/lib.rs verify_arith / verify_copy) are checked on the discrete logarithms, and the reference's
testing-mode assertions (R1CS satisfaction, Lemma 3, quotient identities, zero Ruffini remainders) run inside the product.
Inputs: tests/synth_circuit.py (random satisfying subcircuits in the reference's file formats).  The reference ships no
proof / witness fixtures and cannot be built here, so the proof bytes themselves are "parity unpinned"; what pins them is
the commit identity (trusted-setup/src/main.rs:236-246), the transcript known answers (tests/test_transcript.py) and the
verifier equations.  Not covered: verify_binding (needs Sigma::gen's QAP-derived tables; the binding tables here are random
multiples of G, so O_mid / O_prv / O_pub_free are checked as the linear combinations the prover must form, nothing more)."""
import json
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _pins():
    return json.load(open(os.path.join(HERE, "golden", "pins.json")))


def stage_crs(gpu, oracle, sp, rnd):
    """fixed-tau CRS resident in HBM + the discrete logarithms of every entry"""
    from tkmk.sigma import Sigma1
    pins = _pins()
    R = oracle.R_MOD
    tx, ty = int(pins["tau_x"], 16), int(pins["tau_y"], 16)
    g = oracle.to_bytes([int(pins["fixed_tau_g1_x"], 16), int(pins["fixed_tau_g1_y"], 16)], 48)
    m_i, s_max = sp["l_D"] - sp["l"], sp["s_max"]
    rs_x, rs_y = max(2 * sp["n"], 2 * m_i), 2 * s_max

    def pts(scalars):
        return gpu.g1_batch_scalar_mul_device(gpu.DeviceBuffer.from_host(oracle.to_bytes(scalars, 32)), g, len(scalars))

    rand = lambda k: [rnd.randrange(1, R) for _ in range(k)]                          # noqa: E731
    crs = {"tau_x": tx, "tau_y": ty, "delta": rnd.randrange(1, R), "eta": rnd.randrange(1, R),
           "gamma_inv_o_inst": rand(sp["l"]),
           "eta_inv_li_o_inter_alpha4_kj": [rand(s_max) for _ in range(m_i)],
           "delta_inv_li_o_prv": [rand(s_max) for _ in range(sp["m_D"] - sp["l_D"])],
           "delta_inv_alphak_xh_tx": [rand(3) for _ in range(3)], "delta_inv_alpha4_xj_tx": rand(2),
           "delta_inv_alphak_yi_ty": [rand(3) for _ in range(4)]}
    flat = lambda t: [v for row in t for v in row]                                    # noqa: E731
    sigma1 = Sigma1(pts([pow(tx, i, R) * pow(ty, j, R) % R for i in range(rs_x) for j in range(rs_y)]), rs_x, rs_y)
    tables = {"gamma_inv_o_inst": pts(crs["gamma_inv_o_inst"]),
              "eta_inv_li_o_inter_alpha4_kj": pts(flat(crs["eta_inv_li_o_inter_alpha4_kj"])),
              "delta_inv_li_o_prv": pts(flat(crs["delta_inv_li_o_prv"])),
              "delta_inv_alphak_xh_tx": pts(flat(crs["delta_inv_alphak_xh_tx"])),
              "delta_inv_alpha4_xj_tx": pts(crs["delta_inv_alpha4_xj_tx"]),
              "delta_inv_alphak_yi_ty": pts(flat(crs["delta_inv_alphak_yi_ty"]))}
    singles = {"delta": np.asarray(oracle.g1_scalar_mul(oracle.to_bytes([crs["delta"]], 32), g)),
               "eta": np.asarray(oracle.g1_scalar_mul(oracle.to_bytes([crs["eta"]], 32), g))}
    return (sigma1, tables, singles), crs, g


def seeded_mixer(seed):
    from tkmk.prove import random_mixer
    return random_mixer(random.Random(seed))


@pytest.mark.parametrize("seed,shape", [(11, dict(s_max=8, n_gate_kinds=2, n_out=2, n_in=3, n_prv=6)),
                                        (12, dict(s_max=4, n_gate_kinds=3, n_out=1, n_in=2, n_prv=11, used_placements=3)),
                                        (13, dict(s_max=16, n_gate_kinds=1, n_out=3, n_in=2, n_prv=2, k_pub=3))])
def test_prove_equals_reference_restatement_and_verifies(gpu, oracle, tmp_path, seed, shape):
    import prove_ref
    import synth_circuit
    from tkmk import proofio
    from tkmk.prove import Prover, run_rounds
    rnd = random.Random(seed)
    inst = synth_circuit.build(str(tmp_path), rnd, **shape)
    sp = inst["setup_params"]
    sigma, crs, g = stage_crs(gpu, oracle, sp, rnd)
    mixer = seeded_mixer(seed)

    prover, binding = Prover.init(inst["qap"], inst["synth"], None, mixer=mixer, testing_mode=True, sigma=sigma)
    points, scalars, challenges, p4t, _ = run_rounds(prover, binding)

    dlogs, ref_scalars, ref_ch, ref_p4t, rp = prove_ref.run(inst, crs, mixer, g)
    assert rp.r1cs_satisfied()
    assert challenges == ref_ch
    assert scalars == ref_scalars
    for name in proofio.PROOF_POINT_ORDER:
        assert (np.asarray(points[name]) == np.asarray(prove_ref.g1_of(dlogs[name], g))).all(), name
    for name, d in ref_p4t.items():
        assert (np.asarray(p4t[name]) == np.asarray(prove_ref.g1_of(d, g))).all(), name
    assert all(v == 0 for v in rp.remainders.values()), rp.remainders
    # the verifier's equations on the discrete logarithms (s0, s1, lagrange_KL commitments are CRS / preprocess data)
    assert prove_ref.verify_arith(dlogs, ref_scalars, ref_ch, ref_p4t, crs, sp)
    assert prove_ref.verify_copy(dlogs, ref_scalars, ref_ch, ref_p4t, crs, sp, rp.commit(rp.s0), rp.commit(rp.s1), rp.commit(rp.KL),
                                 kappa2=rnd.randrange(1, oracle.R_MOD))
    # a tampered evaluation must fail them
    bad = dict(ref_scalars, V_eval=(ref_scalars["V_eval"] + 1) % oracle.R_MOD)
    assert not prove_ref.verify_arith(dlogs, bad, ref_ch, ref_p4t, crs, sp)
    # proof.json round trip in the Solidity-verifier format
    fmt = json.loads(json.dumps(proofio.format_proof(points, scalars)))
    assert len(fmt["proof_entries_part1"]) == 38 and len(fmt["proof_entries_part2"]) == 42
    back_points, back_scalars = proofio.recover_proof(fmt)
    assert back_scalars == scalars and all((np.asarray(back_points[k]) == np.asarray(points[k])).all() for k in points)


def test_testing_mode_rejects_bad_witness_and_bad_copy(gpu, oracle, tmp_path):
    """the reference panics in testing-mode when R1CS or the copy constraints fail (lib.rs:1513-1517, 980-993)"""
    import synth_circuit
    from tkmk.prove import Prover
    rnd = random.Random(21)
    inst = synth_circuit.build(str(tmp_path), rnd, s_max=4, n_gate_kinds=1)
    sigma, _, _ = stage_crs(gpu, oracle, inst["setup_params"], rnd)
    pv_path = os.path.join(inst["synth"], "placementVariables.json")
    good = json.load(open(pv_path))
    # (a) break one private wire of a gate placement: b and the copy constraints stay fine, R1CS does not
    bad = json.loads(json.dumps(good))
    sub = inst["subs"][bad[1]["subcircuitId"]]
    bad[1]["variables"][list(sub.prvs())[0]] = "0x5"
    json.dump(bad, open(pv_path, "w"))
    prover, _ = Prover.init(inst["qap"], inst["synth"], None, mixer=seeded_mixer(1), testing_mode=True, sigma=sigma)
    with pytest.raises(AssertionError, match="do not satisfy R1CS"):
        prover.prove0()
    # (b) break an input wire that a copy constraint ties to its source: caught in init
    bad = json.loads(json.dumps(good))
    bad[1]["variables"][list(sub.ins())[0]] = "0x7"
    json.dump(bad, open(pv_path, "w"))
    with pytest.raises(AssertionError, match="copy constraint"):
        Prover.init(inst["qap"], inst["synth"], None, mixer=seeded_mixer(1), testing_mode=True, sigma=sigma)
    json.dump(good, open(pv_path, "w"))


def _stage_crs_file(gpu, oracle, sp, rnd, crs_dir):
    from tkmk import crs as crsmod
    (sigma1, tables, singles), crs, g = stage_crs(gpu, oracle, sp, rnd)
    zero_g1 = np.zeros(96, np.uint8)
    g_aff = np.frombuffer(bytes(g), np.uint8)
    sections = {"g1": np.concatenate([g_aff, zero_g1, zero_g1, singles["delta"], singles["eta"], zero_g1]),
                "xy_powers": sigma1.xy_powers.to_host(), "g2": np.zeros(10 * 192, np.uint8)}
    sections.update({k: v.to_host() for k, v in tables.items()})
    os.makedirs(crs_dir, exist_ok=True)
    with open(os.path.join(crs_dir, "combined_sigma.tkcrs"), "wb") as f:
        f.write(crsmod.build_payload(sections))
    return (sigma1, tables, singles), crs, g


@pytest.mark.parametrize("seed,shape", [(41, dict(s_max=8, n_gate_kinds=2)), (42, dict(s_max=4, n_gate_kinds=3, n_out=1, n_in=2, n_prv=11, used_placements=3))])
def test_native_prove_binary(gpu, oracle, tmp_path, seed, shape):
    """tokamak-zk-evm_amd/bin/prove (host/prove_main.cpp over host/tkmk_prover.hpp, the C++ host side) writes the same proof.json as
    the Python prover for the same blinding scalars, and both equal the exponent restatement"""
    import subprocess
    import prove_ref
    import synth_circuit
    from tkmk import proofio
    from tkmk.prove import Prover, run_rounds
    binary = os.path.join(os.path.dirname(HERE), "tokamak-zk-evm_amd", "bin", "prove")
    assert os.path.exists(binary), "bin/prove is not built (run __graft_entry__.build())"
    rnd = random.Random(seed)
    inst = synth_circuit.build(str(tmp_path), rnd, **shape)
    sp = inst["setup_params"]
    crs_dir, out_dir = str(tmp_path / "crs"), str(tmp_path / "out")
    sigma, crs, g = _stage_crs_file(gpu, oracle, sp, rnd, crs_dir)
    os.makedirs(out_dir)
    mixer = seeded_mixer(seed)
    hx = lambda v: [hx(e) for e in v] if isinstance(v, list) else "0x%x" % v          # noqa: E731
    mixer_path = str(tmp_path / "mixer.json")
    json.dump({k: hx(v) for k, v in mixer.items()}, open(mixer_path, "w"))
    cmd = [binary, "--crs", crs_dir, "--synthesizer-stat", inst["synth"], "--output", out_dir, "--subcircuit-library", inst["qap"]]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, TKMK_PROVE_MIXER=mixer_path))
    assert r.returncode == 0, r.stderr
    native_points, native_scalars = proofio.recover_proof(json.load(open(os.path.join(out_dir, "proof.json"))))

    prover, binding = Prover.init(inst["qap"], inst["synth"], None, mixer=mixer, sigma=sigma)
    points, scalars, _, _, _ = run_rounds(prover, binding)
    assert native_scalars == scalars
    for name in proofio.PROOF_POINT_ORDER:
        assert (np.asarray(native_points[name]) == np.asarray(points[name])).all(), name
    dlogs, ref_scalars, _, _, _ = prove_ref.run(inst, crs, mixer, g)
    assert native_scalars == ref_scalars
    for name in proofio.PROOF_POINT_ORDER:
        assert (np.asarray(native_points[name]) == np.asarray(prove_ref.g1_of(dlogs[name], g))).all(), name
    # fresh blinding without the hook; missing CRS -> the reference's message and a non-zero exit
    r2 = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stderr
    p2, _ = proofio.recover_proof(json.load(open(os.path.join(out_dir, "proof.json"))))
    assert (np.asarray(p2["A_free"]) == np.asarray(points["A_free"])).all() and not (np.asarray(p2["U"]) == np.asarray(points["U"])).all()
    os.remove(os.path.join(crs_dir, "combined_sigma.tkcrs"))
    r3 = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r3.returncode != 0 and "No reference string is found" in r3.stderr


def test_prove_cli_files_in_files_out(gpu, oracle, tmp_path):
    """process-level surface of `prove` (prove/src/main.rs:8-25): directories in, proof.json out; CRS staged as TKCRS001"""
    import subprocess
    import sys
    import synth_circuit
    from tkmk import crs as crsmod
    from tkmk import proofio
    rnd = random.Random(31)
    inst = synth_circuit.build(str(tmp_path), rnd, s_max=4, n_gate_kinds=2)
    sp = inst["setup_params"]
    (sigma1, tables, singles), crs, g = stage_crs(gpu, oracle, sp, rnd)
    zero_g1 = np.zeros(96, np.uint8)
    g_aff = np.frombuffer(bytes(g), np.uint8)
    sections = {"g1": np.concatenate([g_aff, zero_g1, zero_g1, singles["delta"], singles["eta"], zero_g1]),
                "xy_powers": sigma1.xy_powers.to_host(), "g2": np.zeros(10 * 192, np.uint8)}
    sections.update({k: v.to_host() for k, v in tables.items()})
    crs_dir, out_dir = tmp_path / "crs", tmp_path / "out"
    crs_dir.mkdir()
    (crs_dir / "combined_sigma.tkcrs").write_bytes(crsmod.build_payload(sections))
    pkg = os.path.join(os.path.dirname(HERE), "tokamak-zk-evm_amd")
    cmd = [sys.executable, "-m", "tkmk.cli", "prove", "--crs", str(crs_dir), "--synthesizer-stat", inst["synth"], "--output", str(out_dir),
           "--subcircuit-library", inst["qap"]]
    runs = []
    for _ in range(2):
        r = subprocess.run(cmd, cwd=pkg, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        runs.append(proofio.recover_proof(json.load(open(out_dir / "proof.json"))))
    (p_a, s_a), (p_b, s_b) = runs
    import prove_ref
    rp = prove_ref.RefProver(inst, crs, seeded_mixer(0))
    bind = rp.binding()
    for k in ("A_free", "O_pub_free"):                       # the parts of the proof that carry no blinding
        assert (np.asarray(p_a[k]) == np.asarray(prove_ref.g1_of(bind[k], g))).all()
        assert (np.asarray(p_a[k]) == np.asarray(p_b[k])).all()
    assert not (np.asarray(p_a["U"]) == np.asarray(p_b["U"])).all()       # fresh mixer every run (lib.rs:1040-1080)
    # missing CRS -> the reference's message and a non-zero exit
    os.remove(crs_dir / "combined_sigma.tkcrs")
    r = subprocess.run(cmd, cwd=pkg, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "No reference string is found" in (r.stderr + r.stdout)
