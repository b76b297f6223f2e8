"""Test infrastructure: what a verifier does with the prover's FILES alone (packages/backend/verify/verify-rust/src/lib.rs:54-117,
248-289): recover the 19 points and 4 evaluations from proof.json, the three preprocess points from preprocess.json, replay the
Fiat-Shamir transcript for the challenges, interpolate a_pub from instance.json, and evaluate the combined verification
equation with actual pairings (tests/pairing_ref.py) against the G1 singles and Sigma2 of the CRS container.  No prover state, no
discrete logarithm; independent of the circuit size except for the 128-point interpolation of the public input.
Used by tests/test_gpu_prove.py (small shapes), tests/test_gpu_fullsize.py (production shape and BASELINE configs[3]) and
tools/prove_bench.py --check."""
import json
import os
import random
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def crs_sections(crs_dir, setup_params=None):
    from tkmk import crs as crsmod
    from tkmk import rkyv
    flat = os.path.join(crs_dir, "combined_sigma.tkcrs")
    if os.path.exists(flat):
        return crsmod.read_payload(flat)
    buf = np.fromfile(os.path.join(crs_dir, "combined_sigma.rkyv"), np.uint8)
    return rkyv.decode_combined_sigma(buf, expect=None if setup_params is None else rkyv.expect_for(setup_params))


def sigma_verify_points(crs_dir):
    """the verifier's reference string as the reference's `verify` reads it (verify-rust/src/lib.rs:68-71): <crs_dir>/sigma_verify.json,
    SigmaVerify { G, H, sigma_1 { x, y }, sigma_2 { alpha .. y }, lagrange_KL } with every coordinate one big-endian hex number
    (G1serde / G2serde, libs/src/iotools/mod.rs:986-1059) -> (crs_g1 dict of 96-byte affine records, sigma2 dict of decoded G2 points)"""
    from tkmk import g2
    sv = json.load(open(os.path.join(crs_dir, "sigma_verify.json")))
    g1 = lambda pt: np.frombuffer(int(pt["x"], 16).to_bytes(48, "little") + int(pt["y"], 16).to_bytes(48, "little"), np.uint8).copy()     # noqa: E731
    crs_g1 = {"G": g1(sv["G"]), "x": g1(sv["sigma_1"]["x"]), "y": g1(sv["sigma_1"]["y"]), "lagrange_KL": g1(sv["lagrange_KL"])}
    sigma2 = {"H": g2.from_hex_pair(sv["H"]["x"], sv["H"]["y"])}
    for name, pt in sv["sigma_2"].items():
        sigma2[name] = g2.from_hex_pair(pt["x"], pt["y"])
    return crs_g1, sigma2


def verify(qap_dir, synth_dir, crs_dir, out_dir, tamper_public_input=False, seed=5, from_sigma_verify=False):
    """-> True iff <out_dir>/proof.json verifies against <out_dir>/preprocess.json (made with bin/preprocess when absent), the public
    inputs of <synth_dir>/instance.json and the CRS in <crs_dir>"""
    import prove_ref
    from tkmk import crs as crsmod
    from tkmk import g2, proofio
    from tkmk.transcript import TranscriptManager
    sp = json.load(open(os.path.join(qap_dir, "setupParams.json")))
    if not os.path.exists(os.path.join(out_dir, "preprocess.json")):
        r = subprocess.run([os.path.join(ROOT, "tokamak-zk-evm_amd", "bin", "preprocess"), "--crs", crs_dir, "--synthesizer-stat", synth_dir, "--output", out_dir,
                            "--subcircuit-library", qap_dir], capture_output=True, text=True, timeout=900)
        if r.returncode != 0:
            raise RuntimeError("bin/preprocess failed: " + r.stderr)
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
    ins = json.load(open(os.path.join(synth_dir, "instance.json")))
    a = [int(h, 16) for h in ins["a_pub_user"][:sp["l_user"]]] + [int(h, 16) for h in ins["a_pub_block"][:sp["l_free"] - sp["l_user"]]]
    a_eval = prove_ref.interpolate([[v] for v in a], sp["l_free"], 1).eval(chi, zeta)          # Instance::gen_a_free_X, then eval
    if tamper_public_input:
        a_eval = (a_eval + 1) % prove_ref.R
    if from_sigma_verify:      # the files the reference's verifier reads, and nothing else of the CRS
        crs_g1, sigma2 = sigma_verify_points(crs_dir)
        if set(sigma2) != set(crsmod.G2_POINTS):
            raise ValueError("sigma_verify.json: unexpected Sigma2 entries")
    else:
        sections = crs_sections(crs_dir, sp)
        crs_g1 = {k: np.asarray(crsmod.single_g1(sections, k)) for k in ("G", "x", "y", "lagrange_KL")}
        recs = np.asarray(sections["g2"]).reshape(10, 192)
        if not recs.any():
            raise ValueError("the CRS holds no Sigma2 (all-zero G2 section): every pairing would be 1 and the check vacuous")
        sigma2 = {name: g2.decode(recs[i]) for i, name in enumerate(crsmod.G2_POINTS)}
    kappa2 = random.Random(seed).randrange(1, prove_ref.R)
    return bool(prove_ref.verify_snark_pairing(points, scalars, ch, sp, crs_g1, pre_points, sigma2, a_eval, kappa2))
