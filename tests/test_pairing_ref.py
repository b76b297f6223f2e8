"""CPU tier: (1) the reference pairing of tests/pairing_ref.py is a non-degenerate bilinear map of order r (so it decides
product-of-pairings equations); (2) the verifier's combined equation verify_snark (packages/backend/verify-rust/src/lib.rs:248-289) holds
with ACTUAL pairings — proof points as group elements, the G1 side of the CRS, Sigma2's G2 points from tkmk/g2.py — for the proofs of
the restated prover, and fails for a tampered proof or another public input.  No discrete logarithm enters the check itself."""
import json
import os
import random

import numpy as np
import pytest

PINS = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pins.json")))


@pytest.fixture(scope="module")
def ctx(oracle, tkmk):
    import pairing_ref
    from tkmk import g2
    g = oracle.to_bytes([int(PINS["fixed_tau_g1_x"], 16), int(PINS["fixed_tau_g1_y"], 16)], 48)
    h = g2.from_hex_pair(PINS["fixed_tau_g2_x"], PINS["fixed_tau_g2_y"])
    return pairing_ref, g2, g, h


def test_pairing_is_bilinear_and_nondegenerate(oracle, ctx):
    pr, g2, g, h = ctx
    G = pr.g1_from_record(g)
    mul = lambda k: pr.g1_from_record(oracle.g1_scalar_mul(oracle.to_bytes([k % oracle.R_MOD], 32), g))      # noqa: E731
    one = pr.F12.of(1)
    a = pr.F12(list(range(5, 17)))
    assert a * a.inv() == one
    x, y = pr.twist(h)
    assert y * y == x * x * x + pr.F12.of(4)
    e = pr.pairing_product([(G, h)])
    assert not e == one and e ** pr.R == one
    ka, kb = 0x1234567, 0x7654321
    assert pr.pairing_product([(mul(ka), g2.scalar_mul(kb, h))]) == e ** (ka * kb)
    assert pr.pairing_product([(mul(ka), h)]) == pr.pairing_product([(G, g2.scalar_mul(ka, h))])
    assert pr.pairing_product([(mul(ka), h), (mul(pr.R - ka), h)]) == one
    assert pr.pairing_product([(None, h), (G, None)]) == one


def test_verify_snark_with_pairings(oracle, ctx, tmp_path):
    import prove_ref
    import synth_circuit
    from tkmk.prove import random_mixer
    pr, g2, g, h = ctx
    R = oracle.R_MOD
    rnd = random.Random(17)
    inst = synth_circuit.build(str(tmp_path), rnd, s_max=8, n_gate_kinds=2, used_placements=7)
    sp = inst["setup_params"]
    tau = {k: int(PINS["tau_" + k], 16) for k in ("x", "y", "alpha", "gamma", "delta", "eta")}
    crs = prove_ref.sigma_gen(inst, tau)
    d, s, ch, p4t, rp = prove_ref.run(inst, crs, random_mixer(random.Random(17)), g)
    pre = prove_ref.preprocess(rp, inst, crs)
    rec = lambda dlog: np.asarray(prove_ref.g1_of(dlog, g))                           # noqa: E731
    points = {k: rec(v) for k, v in d.items()}
    crs_g1 = {"G": rec(1), "x": rec(crs["tau_x"]), "y": rec(crs["tau_y"]), "lagrange_KL": rec(crs["lagrange_KL"])}
    pre_points = {k: rec(v) for k, v in pre.items()}
    names = ("H", "alpha", "alpha2", "alpha3", "alpha4", "gamma", "delta", "eta", "x", "y")
    sigma2 = dict(zip(names, g2.sigma2_gen(tau, h)))
    a_eval = rp.a_free.eval(ch["chi"], ch["zeta"])
    k2 = rnd.randrange(1, R)
    assert prove_ref.verify_snark_pairing(points, s, ch, sp, crs_g1, pre_points, sigma2, a_eval, k2)
    bad = dict(points, Pi_X=rec(d["Pi_X"] + 1))
    assert not prove_ref.verify_snark_pairing(bad, s, ch, sp, crs_g1, pre_points, sigma2, a_eval, k2)
    assert not prove_ref.verify_snark_pairing(points, s, ch, sp, crs_g1, pre_points, sigma2, (a_eval + 1) % R, k2)
    wrong_sigma2 = dict(sigma2, delta=g2.scalar_mul(tau["delta"] + 1, h))
    assert not prove_ref.verify_snark_pairing(points, s, ch, sp, crs_g1, pre_points, wrong_sigma2, a_eval, k2)
