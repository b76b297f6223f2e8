"""CPU tier: host-side G2 arithmetic behind Sigma2 (tokamak-zk-evm_amd/tkmk/g2.py; Sigma2::gen, packages/backend/libs/src/group_structures/
mod.rs:752-777).  Pins: the fixed G2 generator of the testing recipe (setup/trusted-setup/src/main.rs:75-78, tests/golden/pins.json) is a
point of the twist y^2 = x^3 + 4(1 + u) with real part = low 48 bytes — and of order r; group laws on it; the ten Sigma2 points
satisfy the relations their definitions imply; the 192-byte record round-trips."""
import json
import os
import random

import numpy as np
import pytest

PINS = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pins.json")))


@pytest.fixture(scope="module")
def g2(tkmk):
    from tkmk import g2 as m
    return m


def test_fixed_generator_pins_the_encoding(g2):
    h = g2.from_hex_pair(PINS["fixed_tau_g2_x"], PINS["fixed_tau_g2_y"])
    assert g2.on_curve(h)
    swapped = ((h[0][1], h[0][0]), (h[1][1], h[1][0]))           # the other component order is NOT on the twist
    assert not g2.on_curve(swapped)
    assert g2.scalar_mul(g2.R, h) is None and g2.scalar_mul(g2.R - 1, h) is not None
    assert g2.on_curve(g2.STD_G2) and g2.scalar_mul(g2.R, g2.STD_G2) is None
    from tkmk import cli
    assert cli.FIXED_G2 == (PINS["fixed_tau_g2_x"], PINS["fixed_tau_g2_y"])
    assert cli.FIXED_TAU == {k: int(PINS["tau_" + k], 16) for k in ("x", "y", "alpha", "gamma", "delta", "eta")}
    assert "0x%096x" % cli.FIXED_G1[0] == PINS["fixed_tau_g1_x"] and "0x%096x" % cli.FIXED_G1[1] == PINS["fixed_tau_g1_y"]


def test_group_laws(g2):
    h = g2.from_hex_pair(PINS["fixed_tau_g2_x"], PINS["fixed_tau_g2_y"])
    rnd = random.Random(1)
    a, b = rnd.randrange(g2.R), rnd.randrange(g2.R)
    pa, pb = g2.scalar_mul(a, h), g2.scalar_mul(b, h)
    assert g2.on_curve(pa) and g2.on_curve(pb)
    assert g2.add(pa, pb) == g2.scalar_mul((a + b) % g2.R, h) == g2.add(pb, pa)
    assert g2.scalar_mul(a, pb) == g2.scalar_mul(a * b % g2.R, h)
    assert g2.add(pa, pa) == g2.scalar_mul(2 * a % g2.R, h)                       # doubling through add
    assert g2.add(pa, g2.scalar_mul(g2.R - a, h)) is None and g2.add(pa, None) == pa and g2.add(None, pb) == pb
    assert g2.scalar_mul(0, h) is None and g2.scalar_mul(1, h) == h
    for p in (h, pa, None):
        assert g2.decode(g2.encode(p)) == p and g2.encode(p).size == 192
    with pytest.raises(ValueError):
        g2.decode(bytes([255] * 192))


def test_sigma2_relations(g2):
    tau = {k: int(PINS["tau_" + k], 16) for k in ("x", "y", "alpha", "gamma", "delta", "eta")}
    h = g2.from_hex_pair(PINS["fixed_tau_g2_x"], PINS["fixed_tau_g2_y"])
    pts = g2.sigma2_gen(tau, h)
    names = ("H", "alpha", "alpha2", "alpha3", "alpha4", "gamma", "delta", "eta", "x", "y")
    s2 = dict(zip(names, pts))
    assert s2["H"] == h and all(g2.on_curve(p) for p in pts)
    a = tau["alpha"]
    assert s2["alpha2"] == g2.scalar_mul(a, s2["alpha"]) and s2["alpha3"] == g2.scalar_mul(a, s2["alpha2"]) and s2["alpha4"] == g2.scalar_mul(a, s2["alpha3"])
    assert s2["x"] == s2["alpha"]                                  # Tau::gen_fixed has alpha == x (libs/src/field_structures/mod.rs:43-64)
    assert g2.scalar_mul(tau["delta"], s2["gamma"]) == g2.scalar_mul(tau["gamma"], s2["delta"])
    with pytest.raises(ValueError):
        g2.sigma2_gen(tau, ((1, 2), (3, 4)))


def test_cpp_g2_equals_python(g2):
    """host/tkmk_g2.hpp (6 x 64-bit Montgomery Fq, Fp2, Jacobian G2) through tests/host_cpp/g2_driver against tkmk/g2.py"""
    import subprocess
    driver = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_cpp", "g2_driver")
    assert os.path.exists(driver), "tests/host_cpp/g2_driver is not built (run __graft_entry__.build())"
    h = g2.from_hex_pair(PINS["fixed_tau_g2_x"], PINS["fixed_tau_g2_y"])
    rnd = random.Random(5)
    ks = [0, 1, 2, g2.R - 1, g2.R - 2, 1 << 255, (1 << 64) - 1] + [rnd.randrange(g2.R) for _ in range(12)]
    lines = ["oncurve 0x0"] + ["mul 0x%x" % k for k in ks] + ["addmul 0x%x" % k for k in ks[:6]]
    r = subprocess.run([driver, PINS["fixed_tau_g2_x"], PINS["fixed_tau_g2_y"]], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    out = r.stdout.split()
    assert out[0] == "1"
    want = [g2.scalar_mul(k % g2.R, h) for k in ks] + [g2.add(g2.scalar_mul(k % g2.R, h), h) for k in ks[:6]]
    for got, w in zip(out[1:], want):
        assert bytes.fromhex(got) == bytes(g2.encode(w))
    # a point that is not on the twist is reported as such
    r = subprocess.run([driver, PINS["fixed_tau_g2_y"], PINS["fixed_tau_g2_x"]], input="oncurve 0x0\n", capture_output=True, text=True, timeout=60)
    assert r.stdout.split() == ["0"]


def test_cpp_g1_affine_add_equals_oracle(oracle):
    """fqh::g1_affine_add (host/tkmk_fq_host.hpp: a commitment plus its precomputed blinding point, G1serde `+` of
    libs/src/group_structures/mod.rs:895-903) against the oracle's addition: generic pairs, p + p, p + (-p), either operand at infinity"""
    import subprocess
    driver = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_cpp", "g2_driver")
    assert os.path.exists(driver), "tests/host_cpp/g2_driver is not built (run __graft_entry__.build())"
    pts = np.asarray(oracle.g1_random_bases(77, 8)).reshape(8, 96)
    inf = np.zeros(96, np.uint8)
    pairs = [(pts[i], pts[i + 1]) for i in range(7)] + [(pts[0], pts[0]), (pts[1], np.asarray(oracle.g1_neg(np.ascontiguousarray(pts[1])))), (inf, pts[2]), (pts[3], inf), (inf, inf)]
    lines = ["g1add %s %s" % (bytes(p).hex(), bytes(q).hex()) for p, q in pairs]
    r = subprocess.run([driver, PINS["fixed_tau_g2_x"], PINS["fixed_tau_g2_y"]], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    out = r.stdout.split()
    assert len(out) == len(pairs)
    for got, (p, q) in zip(out, pairs):
        assert bytes.fromhex(got) == bytes(np.asarray(oracle.g1_add(np.ascontiguousarray(p), np.ascontiguousarray(q)))), (bytes(p).hex()[:16], bytes(q).hex()[:16])
