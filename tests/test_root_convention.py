"""The one convention of the path that is an inference and not a pin: the generator g of the scalar field's two-adic subgroup,
omega_{2^32} = g^((r-1)/2^32) (include/tkmk.h: TKMK_BLS12_381_FR_ROOT_GENERATOR; "parity unpinned").  The
reference takes the root from ICICLE (libs/src/bivariate_polynomial/mod.rs:47-52), which is not in the tree.  5 (ffjavascript's
rule, what the reference's browser verifier uses on native proofs) is declared; 7 (arkworks / zkcrypto) is the alternative.

These tests make the switch a proven one-line change: the product, the oracle and the Python restatements all derive the root
from the declared constant or from the process-wide override TKMK_FR_ROOT_GENERATOR, and with EITHER generator
  * (not gpu) the C-ABI root, the oracle's and pyref's agree, differ between the two generators, and the NTT pins of
    tests/test_oracle_pins.py hold (run again in a child process under the override);
  * (gpu) the NTT / polynomial / prover parity tiers pass, and the native pipeline trusted-setup -> preprocess -> prove produces a
    proof that verifies from its files with real pairings — while a proof made under one generator is REJECTED when checked
    against a preprocess made under the other (the silent incompatibility the declaration warns about)."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def _child(code, gen, timeout=1800):
    env = dict(os.environ, TKMK_FR_ROOT_GENERATOR=str(gen), PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "tokamak-zk-evm_amd"), HERE, os.path.join(ROOT, "tools")]))
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


ROOTS_CODE = """
import json, numpy as np
import oracle, pyref, tkmk
out = {}
for n in (1 << 32, 1 << 23, 4096, 2):
    a = int.from_bytes(bytes(tkmk.get_root_of_unity(n)), "little")
    b = oracle.to_ints(oracle.root_of_unity(n), 32)[0]
    assert a == b == pyref.root_of_unity(n), n
    out[str(n)] = a
print(json.dumps(out))
"""


@pytest.mark.parametrize("gen", [5, 7])
def test_product_oracle_and_pyref_share_the_root(gen):
    r = _child(ROOTS_CODE, gen)
    assert r.returncode == 0, r.stderr
    roots = json.loads(r.stdout.strip().splitlines()[-1])
    w32 = roots[str(1 << 32)]
    assert w32 == pow(gen, (R - 1) >> 32, R) and pow(w32, 1 << 31, R) == R - 1
    assert roots["2"] == R - 1 and pow(roots["4096"], 4096, R) == 1 and pow(roots["4096"], 2048, R) != 1


def test_the_two_generators_order_the_domain_differently():
    a = json.loads(_child(ROOTS_CODE, 5).stdout.strip().splitlines()[-1])
    b = json.loads(_child(ROOTS_CODE, 7).stdout.strip().splitlines()[-1])
    assert a[str(1 << 32)] != b[str(1 << 32)] and a["4096"] != b["4096"] and a["2"] == b["2"]


def test_a_residue_is_refused():
    # 4 is a square: 4^((r-1)/2^32) has order below 2^32 — the C ABI reports it instead of building a broken domain
    r = _child("import tkmk\ntry:\n    tkmk.get_root_of_unity(1 << 20)\n    print('accepted')\nexcept tkmk.TkmkError as e:\n    print('refused', e.code)\n", 4)
    assert r.returncode == 0 and r.stdout.split() == ["refused", "11"], r.stdout + r.stderr


@pytest.mark.parametrize("gen", [5, 7])
def test_cpu_pins_hold_under_either_generator(gen):
    env = dict(os.environ, TKMK_FR_ROOT_GENERATOR=str(gen))
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "tests/test_oracle_pins.py", "tests/test_oracle_poly.py",
                        "tests/test_prove_ref.py", "tests/test_hostcheck.py", "-p", "no:cacheprovider"], capture_output=True, text=True, timeout=1800, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_gpu_parity_tiers_under_the_alternative_generator():
    """NTT vs oracle, polynomial layer vs oracle, prover vs the exponent restatement + verifier equations: all green with g = 7 (the
    default tier runs them with the declared g)"""
    env = dict(os.environ, TKMK_FR_ROOT_GENERATOR="7")
    # the tests of the three tiers whose result depends on the root: the transforms against the oracle (bivariate = both 1-D passes, cosets,
    # in place), products, the fused evaluator and its leaf views, and the prover end to end.  Left to the default tier alone: the 1-D
    # shape sweeps, the large padded-transform shapes and the root-free polynomial bookkeeping (find_degree, resize, lincomb).
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "tests/test_gpu_ntt.py", "tests/test_gpu_poly.py",
                        "tests/test_gpu_prove.py::test_prove_equals_reference_restatement_and_verifies", "tests/test_gpu_service.py::test_service_equals_python_prover_and_restatement",
                        "-k", "bintt_vs_oracle or row_then_column or coset or mul_vs_oracle or polyexpr_fused or restatement",
                        "-p", "no:cacheprovider", "-o", "addopts="], capture_output=True, text=True, timeout=3000, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


PIPELINE_CODE = """
import os, random, subprocess, sys
import synth_circuit, verify_files
tmp, mode = sys.argv[1], sys.argv[2]
bins = os.path.join(%r, "tokamak-zk-evm_amd", "bin")
qap, synth, crs, out = (os.path.join(tmp, d) for d in ("qap", "synth", "crs", "out"))
if mode == "make":
    inst = synth_circuit.build(tmp, random.Random(81), s_max=8, n_gate_kinds=2, used_placements=8, bit_fraction=0.4)
    os.makedirs(crs, exist_ok=True); os.makedirs(out, exist_ok=True)
    common = ["--synthesizer-stat", synth, "--output", out, "--subcircuit-library", qap]
    for cmd in ([os.path.join(bins, "trusted-setup"), "--fixed-tau", "--subcircuit-library", qap, "--output", crs],
                [os.path.join(bins, "preprocess"), "--crs", crs] + common, [os.path.join(bins, "prove"), "--crs", crs] + common):
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (cmd[0], r.stderr)
print("VERIFIES" if verify_files.verify(qap, synth, crs, out) else "REJECTED")
""" % ROOT


@pytest.mark.gpu
def test_native_pipeline_verifies_under_either_generator_but_not_across(tmp_path):
    dirs = {}
    for gen in (5, 7):
        d = tmp_path / ("g%d" % gen)
        d.mkdir()
        r = subprocess.run([sys.executable, "-c", PIPELINE_CODE, str(d), "make"], capture_output=True, text=True, timeout=1800, cwd=ROOT,
                           env=dict(os.environ, TKMK_FR_ROOT_GENERATOR=str(gen), PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "tokamak-zk-evm_amd"), HERE, os.path.join(ROOT, "tools")])))
        assert r.returncode == 0, r.stderr[-3000:]
        assert r.stdout.strip().splitlines()[-1] == "VERIFIES", (gen, r.stdout)
        dirs[gen] = d
    # the proof made under g = 7 against the preprocess (s0 / s1 commitments, built from omega powers) made under g = 5: rejected
    import shutil
    shutil.copy(dirs[7] / "out" / "proof.json", dirs[5] / "out" / "proof.json")
    r = subprocess.run([sys.executable, "-c", PIPELINE_CODE, str(dirs[5]), "check"], capture_output=True, text=True, timeout=1800, cwd=ROOT,
                       env=dict(os.environ, TKMK_FR_ROOT_GENERATOR="5", PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "tokamak-zk-evm_amd"), HERE, os.path.join(ROOT, "tools")])))
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.strip().splitlines()[-1] == "REJECTED"
