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
multiples of G, so O_mid / O_prv / O_pub_free are checked as the linear combinations the prover must form, nothing more).
Update: the CRS now comes from the product's own Sigma.gen (tkmk/setup.py, checked entry by entry against the restated
setup), so verify_binding and the combined verify_snark equation are checked too — a proof from the GPU path verifies."""
import json
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _pins():
    return json.load(open(os.path.join(HERE, "golden", "pins.json")))


def _tau():
    pins = _pins()
    return {k: int(pins["tau_" + k], 16) for k in ("x", "y", "alpha", "gamma", "delta", "eta")}


def stage_crs(gpu, oracle, inst):
    """the CRS of the fixed-tau trusted setup (setup/trusted-setup/src/main.rs:68-80), generated on the device by the product's
    Sigma.gen (tkmk/setup.py) from the circuit's .r1cs files, + the discrete logarithm of every entry from the restatement
    (prove_ref.sigma_gen).  -> ((sigma1, tables, singles), dlogs, g)"""
    import prove_ref
    from tkmk.setup import Sigma
    pins = _pins()
    g = oracle.to_bytes([int(pins["fixed_tau_g1_x"], 16), int(pins["fixed_tau_g1_y"], 16)], 48)
    from tkmk import g2
    sigma = Sigma.gen(inst["setup_params"], _tau(), inst["qap"], inst["infos"], np.frombuffer(bytes(g), np.uint8),
                      g2.from_hex_pair(pins["fixed_tau_g2_x"], pins["fixed_tau_g2_y"]))
    return sigma, prove_ref.sigma_gen(inst, _tau()), g


def test_sigma_gen_equals_restated_setup(gpu, oracle, tmp_path):
    """every G1 entry of the generated reference string is [dlog]G for the dlog the restated setup computes; the TKCRS001
    payload written by Sigma.write reads back to the same sections"""
    import prove_ref
    import synth_circuit
    from tkmk import crs as crsmod
    inst = synth_circuit.build(str(tmp_path), random.Random(5), s_max=8, n_gate_kinds=2, used_placements=7)
    sp = inst["setup_params"]
    sigma, crs, g = stage_crs(gpu, oracle, inst)
    R = oracle.R_MOD
    m_i, s_max = sp["l_D"] - sp["l"], sp["s_max"]
    flat = lambda t: [v for row in t for v in row]                                    # noqa: E731
    want = {"gamma_inv_o_inst": crs["gamma_inv_o_inst"], "eta_inv_li_o_inter_alpha4_kj": flat(crs["eta_inv_li_o_inter_alpha4_kj"]),
            "delta_inv_li_o_prv": flat(crs["delta_inv_li_o_prv"]), "delta_inv_alphak_xh_tx": flat(crs["delta_inv_alphak_xh_tx"]),
            "delta_inv_alpha4_xj_tx": crs["delta_inv_alpha4_xj_tx"], "delta_inv_alphak_yi_ty": flat(crs["delta_inv_alphak_yi_ty"])}
    assert any(crs["o_vec"]) and any(want["delta_inv_li_o_prv"])
    for name, dlogs in want.items():
        t = sigma.tables[name]
        got = (t.to_host() if hasattr(t, "to_host") else np.asarray(t)).reshape(-1, 96)
        exp = np.asarray(oracle.g1_batch_scalar_mul(oracle.to_bytes(dlogs, 32), g)).reshape(-1, 96)
        assert got.shape == exp.shape and (got == exp).all(), name
    rs_x, rs_y = max(2 * sp["n"], 2 * m_i), 2 * s_max
    mon = [pow(crs["tau_x"], i, R) * pow(crs["tau_y"], j, R) % R for i in range(rs_x) for j in range(rs_y)]
    assert (sigma.sigma1.xy_powers.to_host().reshape(-1, 96) == np.asarray(oracle.g1_batch_scalar_mul(oracle.to_bytes(mon, 32), g)).reshape(-1, 96)).all()
    for name, d in (("G", 1), ("x", crs["tau_x"]), ("y", crs["tau_y"]), ("delta", crs["delta"]), ("eta", crs["eta"]), ("lagrange_KL", crs["lagrange_KL"])):
        assert (sigma.singles[name] == np.asarray(prove_ref.g1_of(d, g))).all(), name
    path = sigma.write(str(tmp_path / "crs"))
    sections = crsmod.read_payload(path)
    crsmod.check_shapes(sections, sp)
    assert (np.asarray(sections["eta_inv_li_o_inter_alpha4_kj"]) == sigma.tables["eta_inv_li_o_inter_alpha4_kj"].to_host()).all()
    assert (np.asarray(crsmod.single_g1(sections, "delta")) == sigma.singles["delta"]).all()
    # Sigma2 in the payload: ten 192-byte records, H first, [alpha]H second (tkmk/g2.py; relations in tests/test_g2.py)
    from tkmk import g2
    recs = np.asarray(sections["g2"]).reshape(10, 192)
    h = g2.from_hex_pair(_pins()["fixed_tau_g2_x"], _pins()["fixed_tau_g2_y"])
    assert g2.decode(recs[0]) == h and g2.decode(recs[1]) == g2.scalar_mul(crs["alpha"], h) and g2.decode(recs[9]) == g2.scalar_mul(crs["tau_y"], h)


def seeded_mixer(seed):
    from tkmk.prove import random_mixer
    return random_mixer(random.Random(seed))


@pytest.mark.parametrize("seed,shape", [(11, dict(s_max=8, n_gate_kinds=2, n_out=2, n_in=3, n_prv=6)),
                                        (12, dict(s_max=8, n_gate_kinds=3, n_out=1, n_in=2, n_prv=11, used_placements=6, bit_fraction=0.6)),
                                        (13, dict(s_max=16, n_gate_kinds=1, n_out=3, n_in=2, n_prv=2, k_pub=3)),
                                        (14, dict(s_max=8, n_gate_kinds=2, used_placements=6, k_out=0, k_pub=3)),   # l_user_out = 0: an empty bufferPubOut
                                        # n = 2: the 4-coefficient blinding polynomial overlaps its own shift by n (found by tools/prove_fuzz.py)
                                        (15, dict(s_max=8, n_gate_kinds=2, n_out=1, n_in=2, n_prv=1, k_out=0, k_pub=2, l_extra=1, used_placements=7))])
def test_prove_equals_reference_restatement_and_verifies(gpu, oracle, tmp_path, seed, shape):
    import prove_ref
    import synth_circuit
    from tkmk import proofio
    from tkmk.prove import Prover, run_rounds
    rnd = random.Random(seed)
    inst = synth_circuit.build(str(tmp_path), rnd, **shape)
    sp = inst["setup_params"]
    sigma_obj, crs, g = stage_crs(gpu, oracle, inst)
    sigma = sigma_obj.prover_view()
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
    # all four verifier equations on the discrete logarithms, with the preprocess round run by the product on the same CRS
    from tkmk.preprocess import Preprocess
    pre_pts = Preprocess.gen(sigma[0], sigma[1]["gamma_inv_o_inst"], inst["permutation"], inst["instance"], sp)
    pre = prove_ref.preprocess(rp, inst, crs)
    for name in ("s0", "s1", "O_pub_fix"):
        assert (np.asarray(getattr(pre_pts, name)) == np.asarray(prove_ref.g1_of(pre[name], g))).all(), name
    k2 = rnd.randrange(1, oracle.R_MOD)
    assert prove_ref.verify_arith(dlogs, ref_scalars, ref_ch, ref_p4t, crs, sp)
    assert prove_ref.verify_copy(dlogs, ref_scalars, ref_ch, ref_p4t, crs, sp, pre["s0"], pre["s1"], crs["lagrange_KL"], kappa2=k2)
    assert prove_ref.verify_binding(dlogs, ref_scalars, ref_ch, ref_p4t, crs, pre, rp.a_free, k2)
    assert prove_ref.verify_snark(dlogs, ref_scalars, ref_ch, crs, sp, pre, rp.a_free, k2)
    if seed == 11:
        # the same combined equation with ACTUAL pairings (tests/pairing_ref.py) on nothing but what the product produced: the GPU
        # prover's points and evaluations, the preprocess points, and G / [x]G / [y]G / lagrange_KL / Sigma2 read back from the
        # TKCRS001 payload the product's setup wrote
        from tkmk import crs as crsmod
        from tkmk import g2
        from tkmk.prove import fr, fr_int
        sections = crsmod.parse_payload(sigma_obj.payload())
        crs_g1 = {k: np.asarray(crsmod.single_g1(sections, k)) for k in ("G", "x", "y", "lagrange_KL")}
        recs = np.asarray(sections["g2"]).reshape(10, 192)
        sigma2 = {name: g2.decode(recs[i]) for i, name in enumerate(crsmod.G2_POINTS)}
        pre_points = {name: np.asarray(getattr(pre_pts, name)) for name in ("s0", "s1", "O_pub_fix")}
        a_eval = fr_int(prover.a_free_X.eval(fr(challenges["chi"]), fr(challenges["zeta"])))
        assert prove_ref.verify_snark_pairing(points, scalars, challenges, sp, crs_g1, pre_points, sigma2, a_eval, k2)
        tampered = dict(points, M_Y=np.asarray(points["N_Y"]))
        assert not prove_ref.verify_snark_pairing(tampered, scalars, challenges, sp, crs_g1, pre_points, sigma2, a_eval, k2)
    # a tampered evaluation must fail them
    bad = dict(ref_scalars, V_eval=(ref_scalars["V_eval"] + 1) % oracle.R_MOD)
    assert not prove_ref.verify_arith(dlogs, bad, ref_ch, ref_p4t, crs, sp)
    # proof.json round trip in the Solidity-verifier format
    fmt = json.loads(json.dumps(proofio.format_proof(points, scalars)))
    assert len(fmt["proof_entries_part1"]) == 38 and len(fmt["proof_entries_part2"]) == 42
    back_points, back_scalars = proofio.recover_proof(fmt)
    assert back_scalars == scalars and all((np.asarray(back_points[k]) == np.asarray(points[k])).all() for k in points)


def test_prove_equals_restatement_at_a_larger_shape(gpu, oracle, tmp_path):
    """m_I = 128, n = 32, s_max = 32: the p_comb domain is 512 x 64, the commitment boxes up to 256 x 63 — the prover's NTTs and MSMs leave
    the single-pass / tiny sizes of the other shapes (the big-int restatement needs ~15 s here; m_I = 256 was run once by hand)"""
    import prove_ref
    import synth_circuit
    from tkmk import proofio
    from tkmk.prove import Prover, run_rounds
    rnd = random.Random(1)
    inst = synth_circuit.build(str(tmp_path), rnd, s_max=32, n_gate_kinds=6, n_out=6, n_in=10, n_prv=24, k_out=2, k_pub=3, l_free=8, l_extra=3,
                               used_placements=27, bit_fraction=0.5)
    sp = inst["setup_params"]
    assert (sp["l_D"] - sp["l"], sp["n"], sp["s_max"]) == (128, 32, 32)
    sigma_obj, crs, g = stage_crs(gpu, oracle, inst)
    mixer = seeded_mixer(1)
    prover, binding = Prover.init(inst["qap"], inst["synth"], None, mixer=mixer, testing_mode=True, sigma=sigma_obj.prover_view())
    points, scalars, challenges, _, _ = run_rounds(prover, binding)
    dlogs, ref_scalars, ref_ch, _, rp = prove_ref.run(inst, crs, mixer, g)
    assert challenges == ref_ch and scalars == ref_scalars
    for name in proofio.PROOF_POINT_ORDER:
        assert (np.asarray(points[name]) == np.asarray(prove_ref.g1_of(dlogs[name], g))).all(), name
    assert prove_ref.verify_snark(dlogs, ref_scalars, ref_ch, crs, sp, prove_ref.preprocess(rp, inst, crs), rp.a_free, 5)


def test_testing_mode_rejects_bad_witness_and_bad_copy(gpu, oracle, tmp_path):
    """the reference panics in testing-mode when R1CS or the copy constraints fail (lib.rs:1513-1517, 980-993)"""
    import synth_circuit
    from tkmk.prove import Prover
    rnd = random.Random(21)
    inst = synth_circuit.build(str(tmp_path), rnd, s_max=8, n_gate_kinds=1, used_placements=6)
    sigma = stage_crs(gpu, oracle, inst)[0].prover_view()
    pv_path = os.path.join(inst["synth"], "placementVariables.json")
    good = json.load(open(pv_path))
    # (a) break one private wire of a gate placement: b and the copy constraints stay fine, R1CS does not
    bad = json.loads(json.dumps(good))
    sub = inst["subs"][bad[4]["subcircuitId"]]
    bad[4]["variables"][list(sub.prvs())[0]] = "0x5"
    json.dump(bad, open(pv_path, "w"))
    prover, _ = Prover.init(inst["qap"], inst["synth"], None, mixer=seeded_mixer(1), testing_mode=True, sigma=sigma)
    with pytest.raises(AssertionError, match="do not satisfy R1CS"):
        prover.prove0()
    # (b) break an input wire that a copy constraint ties to its source: caught in init
    bad = json.loads(json.dumps(good))
    bad[4]["variables"][list(sub.ins())[0]] = "0x7"
    json.dump(bad, open(pv_path, "w"))
    with pytest.raises(AssertionError, match="copy constraint"):
        Prover.init(inst["qap"], inst["synth"], None, mixer=seeded_mixer(1), testing_mode=True, sigma=sigma)
    json.dump(good, open(pv_path, "w"))


def _stage_crs_file(gpu, oracle, inst, crs_dir):
    sigma, crs, g = stage_crs(gpu, oracle, inst)
    sigma.write(crs_dir)
    return sigma.prover_view(), crs, g


@pytest.mark.parametrize("seed,shape", [(41, dict(s_max=8, n_gate_kinds=2)), (42, dict(s_max=8, n_gate_kinds=3, n_out=1, n_in=2, n_prv=11, used_placements=6, bit_fraction=0.6)),
                                        (43, dict(s_max=8, n_gate_kinds=2, used_placements=6, k_out=0, k_pub=3)),
                                        (44, dict(s_max=8, n_gate_kinds=2, n_out=1, n_in=2, n_prv=1, k_out=0, k_pub=2, l_extra=1, used_placements=7))])
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
    sigma, crs, g = _stage_crs_file(gpu, oracle, inst, crs_dir)
    os.makedirs(out_dir)
    mixer = seeded_mixer(seed)
    hx = lambda v: [hx(e) for e in v] if isinstance(v, list) else "0x%x" % v          # noqa: E731
    mixer_path = str(tmp_path / "mixer.json")
    json.dump({k: hx(v) for k, v in mixer.items()}, open(mixer_path, "w"))
    if seed == 42:       # the binary scans placementVariables.json itself: any key order / whitespace / extra scalar keys must do
        pv_path = os.path.join(inst["synth"], "placementVariables.json")
        docs = json.load(open(pv_path))
        json.dump([{"variables": d["variables"], "note": "x", "subcircuitId": d["subcircuitId"], "k": 7} for d in docs], open(pv_path, "w"), indent=1)
    cmd = [binary, "--crs", crs_dir, "--synthesizer-stat", inst["synth"], "--output", out_dir, "--subcircuit-library", inst["qap"]]
    # fixed blinding scalars exist only in the testing-mode build (the reference's compile-time `testing-mode` feature):
    # the production binary refuses the flag before touching the GPU
    r = subprocess.run(cmd + ["--testing-mixer", mixer_path], capture_output=True, text=True, timeout=600)
    assert r.returncode == 2 and "testing-mode build" in r.stderr
    r = subprocess.run([binary + "-testing"] + cmd[1:] + ["--testing-mixer", mixer_path], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert "NOT zero-knowledge" in r.stderr                       # the testing hook announces itself
    native_points, native_scalars = proofio.recover_proof(json.load(open(os.path.join(out_dir, "proof.json"))))
    # an inherited environment variable must not fix the blinding scalars (round-1 hook, removed): same command, fresh proof
    r_env = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, TKMK_PROVE_MIXER=mixer_path))
    assert r_env.returncode == 0 and "NOT zero-knowledge" not in r_env.stderr
    env_points, _ = proofio.recover_proof(json.load(open(os.path.join(out_dir, "proof.json"))))
    assert not (np.asarray(env_points["U"]) == np.asarray(native_points["U"])).all()

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
    # a CRS generated for another shape: refused with the section named (Rust would index out of bounds / panic)
    sp_path = os.path.join(inst["qap"], "setupParams.json")
    good_sp = open(sp_path).read()
    json.dump(dict(sp, s_max=2 * sp["s_max"]), open(sp_path, "w"))
    r4 = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r4.returncode != 0 and "does not match setupParams.json" in r4.stderr
    open(sp_path, "w").write(good_sp)
    # corrupted witness document: an error, not a crash
    pv_path = os.path.join(inst["synth"], "placementVariables.json")
    good_pv = open(pv_path).read()
    open(pv_path, "w").write(good_pv[:len(good_pv) // 2])
    r5 = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r5.returncode == 1 and "placementVariables.json" in r5.stderr
    open(pv_path, "w").write(good_pv)
    os.remove(os.path.join(crs_dir, "combined_sigma.tkcrs"))
    r3 = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r3.returncode != 0 and "No reference string is found" in r3.stderr


def test_native_setup_binary(gpu, oracle, tmp_path):
    """tokamak-zk-evm_amd/bin/trusted-setup (host/setup_main.cpp over host/tkmk_setup.hpp + tkmk_g2.hpp) with --fixed-tau writes the very
    payload tkmk/setup.py produces — every G1 table and the ten G2 points (so everything checked on that payload elsewhere holds for it)"""
    import subprocess
    import synth_circuit
    binary = os.path.join(os.path.dirname(HERE), "tokamak-zk-evm_amd", "bin", "trusted-setup")
    assert os.path.exists(binary), "bin/trusted-setup is not built (run __graft_entry__.build())"
    inst = synth_circuit.build(str(tmp_path), random.Random(61), s_max=8, n_gate_kinds=2, used_placements=7, bit_fraction=0.3)
    crs_dir = tmp_path / "crs"
    crs_dir.mkdir()
    r = subprocess.run([binary, "--fixed-tau", "--subcircuit-library", inst["qap"], "--output", str(crs_dir)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    sigma_obj, _, _ = stage_crs(gpu, oracle, inst)
    assert open(crs_dir / "combined_sigma.tkcrs", "rb").read() == sigma_obj.payload()
    # without --fixed-tau: a fresh tau every run, a well-formed payload of the same shape
    from tkmk import crs as crsmod
    outs = []
    for k in range(2):
        d = tmp_path / ("rnd%d" % k)
        d.mkdir()
        r = subprocess.run([binary, "--subcircuit-library", inst["qap"], "--output", str(d)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        sections = crsmod.read_payload(str(d / "combined_sigma.tkcrs"))
        crsmod.check_shapes(sections, inst["setup_params"])
        outs.append(bytes(np.asarray(sections["g1"])) + bytes(np.asarray(sections["g2"])))
    assert outs[0] != outs[1] and any(outs[0][-1920:])
    r = subprocess.run([binary, "--subcircuit-library", str(tmp_path / "nowhere"), "--output", str(crs_dir)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "trusted-setup:" in r.stderr


def test_native_pipeline_setup_preprocess_prove_then_verify_with_pairings(gpu, oracle, tmp_path):
    """the user's flow with the three native binaries — bin/trusted-setup, bin/preprocess, bin/prove: directories in, files out — and then
    what a verifier does with the files alone (verify-rust/src/lib.rs:54-117,248-289): recover the points, replay the transcript for the
    challenges, interpolate a_pub from instance.json, and evaluate the combined equation with actual pairings against the CRS file's
    G1 singles and Sigma2 (tests/pairing_ref.py).  No prover state, no discrete logarithm, fresh random blinding."""
    import subprocess
    import prove_ref
    import synth_circuit
    from tkmk import crs as crsmod
    from tkmk import g2, proofio
    from tkmk.transcript import TranscriptManager
    bin_dir = os.path.join(os.path.dirname(HERE), "tokamak-zk-evm_amd", "bin")
    rnd = random.Random(71)
    inst = synth_circuit.build(str(tmp_path), rnd, s_max=8, n_gate_kinds=2, used_placements=8, bit_fraction=0.4)
    sp = inst["setup_params"]
    crs_dir, out_dir = str(tmp_path / "crs"), str(tmp_path / "out")
    os.makedirs(crs_dir)
    os.makedirs(out_dir)
    common = ["--synthesizer-stat", inst["synth"], "--output", out_dir, "--subcircuit-library", inst["qap"]]
    for cmd in ([os.path.join(bin_dir, "trusted-setup"), "--fixed-tau", "--subcircuit-library", inst["qap"], "--output", crs_dir],
                [os.path.join(bin_dir, "preprocess"), "--crs", crs_dir] + common, [os.path.join(bin_dir, "prove"), "--crs", crs_dir] + common):
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (cmd[0], r.stderr)
    points, scalars = proofio.recover_proof(json.load(open(os.path.join(out_dir, "proof.json"))))
    pre_points = proofio.recover_preprocess(json.load(open(os.path.join(out_dir, "preprocess.json"))))
    m = TranscriptManager()                                  # Verifier::collect_challenges
    m.add_proof0(*(points[k] for k in ("U", "V", "W", "Q_AX", "Q_AY", "B")))
    thetas = m.get_thetas()
    m.add_proof1(points["R"])
    kappa0 = m.get_kappa0()
    m.add_proof2(points["Q_CX"], points["Q_CY"])
    chi, zeta = m.get_chi_zeta()
    m.add_proof3(scalars["V_eval"], scalars["R_eval"], scalars["R_omegaX_eval"], scalars["R_omegaX_omegaY_eval"])
    ch = {"thetas": thetas, "kappa0": kappa0, "chi": chi, "zeta": zeta, "kappa1": m.get_kappa1()}
    ins = json.load(open(os.path.join(inst["synth"], "instance.json")))
    a = [int(h, 16) for h in ins["a_pub_user"][:sp["l_user"]]] + [int(h, 16) for h in ins["a_pub_block"][:sp["l_free"] - sp["l_user"]]]
    a_eval = prove_ref.interpolate([[v] for v in a], sp["l_free"], 1).eval(chi, zeta)          # Instance::gen_a_free_X, then eval
    sections = crsmod.read_payload(os.path.join(crs_dir, "combined_sigma.tkcrs"))
    crs_g1 = {k: np.asarray(crsmod.single_g1(sections, k)) for k in ("G", "x", "y", "lagrange_KL")}
    recs = np.asarray(sections["g2"]).reshape(10, 192)
    sigma2 = {name: g2.decode(recs[i]) for i, name in enumerate(crsmod.G2_POINTS)}
    kappa2 = rnd.randrange(1, oracle.R_MOD)
    assert prove_ref.verify_snark_pairing(points, scalars, ch, sp, crs_g1, pre_points, sigma2, a_eval, kappa2)
    # the same files with one public input changed do not verify
    assert not prove_ref.verify_snark_pairing(points, scalars, ch, sp, crs_g1, pre_points, sigma2, (a_eval + 1) % oracle.R_MOD, kappa2)


def test_binaries_under_the_cli_argv(gpu, tmp_path):
    """tokamak-cli's runtime layout and EXACTLY the argv it sends (packages/cli/src/cli.ts:537-546 `backendOutputArgs`,
    runtime.ts:465-485,1840-1848): <runtime>/bin/{trusted-setup,preprocess,prove}, resources under <runtime>/resource, no
    --subcircuit-library anywhere.  The library is found next to the installation (host/tkmk_args.hpp); preprocess.json equals the
    one made with the flag given, proof.json verifies with pairings from the files alone."""
    import shutil
    import subprocess
    import synth_circuit
    import verify_files
    pkg = os.path.join(os.path.dirname(HERE), "tokamak-zk-evm_amd")
    rt = tmp_path / "runtime"
    (rt / "bin").mkdir(parents=True)
    for name in ("trusted-setup", "preprocess", "prove"):
        shutil.copy(os.path.join(pkg, "bin", name), rt / "bin" / name)
    os.symlink(os.path.join(pkg, "libtkmk_hip.so"), rt / "libtkmk_hip.so")          # the binaries' rpath is $ORIGIN/..
    inst = synth_circuit.build(str(tmp_path / "work"), random.Random(83), s_max=8, n_gate_kinds=2, used_placements=8, bit_fraction=0.4)
    res = rt / "resource"
    shutil.copytree(inst["qap"], res / "qap-compiler" / "library")
    dirs = {k: res / k / "output" for k in ("setup", "synthesizer", "preprocess", "prove")}
    shutil.copytree(inst["synth"], dirs["synthesizer"])
    for k in ("setup", "preprocess", "prove"):
        dirs[k].mkdir(parents=True)
    env = {k: v for k, v in os.environ.items() if k != "TKMK_SUBCIRCUIT_LIBRARY"}
    env["HOME"] = str(tmp_path / "home")                                            # no cache of a release binary to fall back on

    def backend_output_args(out):                                                   # cli.ts:537-546, verbatim
        return ["--crs", str(dirs["setup"]), "--synthesizer-stat", str(dirs["synthesizer"]), "--output", str(out)]
    for cmd in ([str(rt / "bin" / "trusted-setup"), "--output", str(dirs["setup"]), "--fixed-tau"],
                [str(rt / "bin" / "preprocess")] + backend_output_args(dirs["preprocess"]),
                [str(rt / "bin" / "prove")] + backend_output_args(dirs["prove"])):
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, (cmd, r.stdout, r.stderr)
    # the files tokamak-cli checks for before it spawns the next stage (PREPROCESS_ / PROVE_ / VERIFY_REQUIRED_FILES, cli.ts:91-109)
    for required in ("sigma_preprocess.rkyv", "combined_sigma.rkyv", "sigma_verify.json"):
        assert os.path.exists(dirs["setup"] / required), required
    assert os.path.exists(dirs["preprocess"] / "preprocess.json") and os.path.exists(dirs["prove"] / "proof.json")
    # the same preprocess.json as with the library named explicitly (the round-2 invocation)
    explicit = tmp_path / "explicit"
    explicit.mkdir()
    r = subprocess.run([os.path.join(pkg, "bin", "preprocess")] + backend_output_args(explicit) + ["--subcircuit-library=" + inst["qap"]],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr
    assert json.load(open(explicit / "preprocess.json")) == json.load(open(dirs["preprocess"] / "preprocess.json"))
    shutil.copy(dirs["preprocess"] / "preprocess.json", dirs["prove"] / "preprocess.json")
    assert verify_files.verify(inst["qap"], str(dirs["synthesizer"]), str(dirs["setup"]), str(dirs["prove"]))
    assert not verify_files.verify(inst["qap"], str(dirs["synthesizer"]), str(dirs["setup"]), str(dirs["prove"]), tamper_public_input=True)


def test_prove_cli_files_in_files_out(gpu, oracle, tmp_path):
    """process-level surface: `setup --fixed-tau` (setup/trusted-setup/src/main.rs:27-46) writes the CRS, `preprocess` and `prove`
    (preprocess/src/main.rs, prove/src/main.rs:8-25) read it: directories in, preprocess.json / proof.json out"""
    import subprocess
    import sys
    import synth_circuit
    from tkmk import crs as crsmod
    from tkmk import proofio
    rnd = random.Random(31)
    inst = synth_circuit.build(str(tmp_path), rnd, s_max=8, n_gate_kinds=2, used_placements=7)
    sp = inst["setup_params"]
    crs_dir, out_dir = tmp_path / "crs", tmp_path / "out"
    pkg = os.path.join(os.path.dirname(HERE), "tokamak-zk-evm_amd")
    r = subprocess.run([sys.executable, "-m", "tkmk.cli", "setup", "--fixed-tau", "--subcircuit-library", inst["qap"], "--output", str(crs_dir)],
                       cwd=pkg, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    sigma_obj, crs, g = stage_crs(gpu, oracle, inst)
    assert open(crs_dir / "combined_sigma.tkcrs", "rb").read() == sigma_obj.payload()      # the command is Sigma.gen with the fixed tau
    r = subprocess.run([sys.executable, "-m", "tkmk.cli", "preprocess", "--crs", str(crs_dir), "--synthesizer-stat", inst["synth"], "--output",
                        str(out_dir), "--subcircuit-library", inst["qap"]], cwd=pkg, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
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
    pre_pts = proofio.recover_preprocess(json.load(open(out_dir / "preprocess.json")))
    pre = prove_ref.preprocess(rp, inst, crs)
    for k in ("s0", "s1", "O_pub_fix"):
        assert (np.asarray(pre_pts[k]) == np.asarray(prove_ref.g1_of(pre[k], g))).all(), k
    # missing CRS -> the reference's message and a non-zero exit
    os.remove(crs_dir / "combined_sigma.tkcrs")
    r = subprocess.run(cmd, cwd=pkg, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "No reference string is found" in (r.stderr + r.stdout)


def _dist_prove_worker(rank, world, port, tmp, seed, q):
    import sys
    root = os.path.dirname(HERE)
    sys.path[:0] = [root, os.path.join(root, "tokamak-zk-evm_amd"), os.path.join(root, "tools"), HERE]
    import torch.distributed as dist
    import oracle
    import synth_circuit
    import tkmk
    from tkmk.prove import Prover, run_rounds
    from tkmk.setup import Sigma
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tkmk.set_device(0)                               # rehearsal: both ranks on GPU 0 (one-GPU box); RCCL ranks use their own GPU
        inst = synth_circuit.build(os.path.join(tmp, "rank%d" % rank), random.Random(seed), s_max=8, n_gate_kinds=2, used_placements=7)
        pins = _pins()
        g = oracle.to_bytes([int(pins["fixed_tau_g1_x"], 16), int(pins["fixed_tau_g1_y"], 16)], 48)
        sigma = Sigma.gen(inst["setup_params"], _tau(), inst["qap"], inst["infos"], np.frombuffer(bytes(g), np.uint8))
        sigma.sigma1.dist, sigma.sigma1.comm_device = dist, "cpu"
        prover, binding = Prover.init(inst["qap"], inst["synth"], None, mixer=seeded_mixer(seed), sigma=sigma.prover_view())
        points, scalars, _, _, _ = run_rounds(prover, binding)
        q.put((rank, {k: bytes(np.asarray(v)) for k, v in points.items()}, scalars))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_prove_with_round_commits_spread_over_two_ranks(gpu, oracle, tmp_path):
    """SURVEY.md §8e row 4 in the real prover: two ranks (gloo; both on GPU 0 here) replicate the polynomial work, each runs the
    commitments it owns (sharding.balanced_assignment), one all_gather per round — both end with the single-process proof"""
    import socket
    import synth_circuit
    import torch.multiprocessing as mp
    from tkmk.prove import Prover, run_rounds
    seed = 51
    inst = synth_circuit.build(str(tmp_path / "single"), random.Random(seed), s_max=8, n_gate_kinds=2, used_placements=7)
    sigma = stage_crs(gpu, oracle, inst)[0]
    prover, binding = Prover.init(inst["qap"], inst["synth"], None, mixer=seeded_mixer(seed), sigma=sigma.prover_view())
    points, scalars, _, _, _ = run_rounds(prover, binding)
    want = {k: bytes(np.asarray(v)) for k, v in points.items()}
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dist_prove_worker, args=(r, 2, port, str(tmp_path), seed, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=400) for _ in range(2)]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(r for r, _, _ in got) == [0, 1]
    for _, pts, sc in got:
        assert pts == want and sc == scalars
