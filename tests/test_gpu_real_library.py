"""gpu tier: the product opens the reference's REAL subcircuit library (all fourteen committed .r1cs + the library's own subcircuitInfo.json
and setupParams.json: s_D = 14, m_D = 26591, l_D = 4824, n = 4096, s_max = 256) and takes it through the three binaries under the argv
tokamak-cli sends (packages/cli/src/cli.ts:537-546; trusted-setup: runtime.ts:1840-1848), in tokamak-cli's runtime layout with the library
at <runtime>/resource/qap-compiler/library — then through the resident prover on the same directories.

Inputs: tests/real_library.py (which says what the manufactured assignment is and is not: the buffers and DecToBit carry real values, the
other eight kinds are placed with the all-zero assignment; the library's wasm witness calculators are never run).
Checks: every file tokamak-cli requires exists; proof.json verifies with real pairings from the files alone (tests/verify_files.py) and is
rejected for a changed public input; the CRS sections have the sizes the library's parameters dictate; the resident prover
(libtkmk_prover.so) opens the same library + CRS and its proof verifies too; the Python prover with the reference's testing-mode assertions
(R1CS satisfaction, Lemma 3 / copy constraints, quotient identities: prove/src/lib.rs:916-1019,1472-1545) passes on it and equals the
native proof for the same blinding scalars.  The proof BYTES stay unpinned (the reference holds no proof for any input)."""
import json
import os
import random
import shutil
import subprocess

import numpy as np
import pytest

import real_library

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(os.path.dirname(HERE), "tokamak-zk-evm_amd")


@pytest.fixture(scope="module")
def runtime(gpu, tmp_path_factory):
    """tokamak-cli's runtime tree with the real library, after trusted-setup -> preprocess -> prove under its argv"""
    tmp = tmp_path_factory.mktemp("real_library_runtime")
    rt = tmp / "runtime"
    (rt / "bin").mkdir(parents=True)
    for name in ("trusted-setup", "preprocess", "prove"):
        shutil.copy(os.path.join(PKG, "bin", name), rt / "bin" / name)
    os.symlink(os.path.join(PKG, "libtkmk_hip.so"), rt / "libtkmk_hip.so")          # the binaries' rpath is $ORIGIN/..
    res = rt / "resource"
    library = real_library.assemble(str(res / "qap-compiler" / "library"))
    dirs = {k: res / k / "output" for k in ("setup", "synthesizer", "preprocess", "prove")}
    for d in dirs.values():
        d.mkdir(parents=True)
    made = real_library.make_inputs(str(dirs["synthesizer"]), random.Random(2028))
    env = {k: v for k, v in os.environ.items() if k != "TKMK_SUBCIRCUIT_LIBRARY"}
    env["HOME"] = str(tmp / "home")

    def backend_output_args(out):                                                   # cli.ts:537-546, verbatim
        return ["--crs", str(dirs["setup"]), "--synthesizer-stat", str(dirs["synthesizer"]), "--output", str(out)]
    logs = {}
    for cmd in ([str(rt / "bin" / "trusted-setup"), "--output", str(dirs["setup"]), "--fixed-tau"],
                [str(rt / "bin" / "preprocess")] + backend_output_args(dirs["preprocess"]),
                [str(rt / "bin" / "prove")] + backend_output_args(dirs["prove"])):
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, (cmd, r.stdout[-2000:], r.stderr[-2000:])
        logs[os.path.basename(cmd[0])] = r.stdout + r.stderr
    shutil.copy(dirs["preprocess"] / "preprocess.json", dirs["prove"] / "preprocess.json")
    yield {"library": library, "dirs": {k: str(v) for k, v in dirs.items()}, "made": made, "logs": logs, "tmp": str(tmp)}
    shutil.rmtree(tmp, ignore_errors=True)
    gpu.release_scratch()


def test_binaries_open_the_real_library_and_the_proof_verifies(gpu, runtime):
    import verify_files
    d = runtime["dirs"]
    for required in ("sigma_preprocess.rkyv", "combined_sigma.rkyv", "sigma_verify.json"):      # cli.ts:91-109
        assert os.path.exists(os.path.join(d["setup"], required)), required
    assert os.path.exists(os.path.join(d["preprocess"], "preprocess.json")) and os.path.exists(os.path.join(d["prove"], "proof.json"))
    # every binary says which library it resolved (host/tkmk_args.hpp): the one next to the installation
    for name in ("trusted-setup", "preprocess", "prove"):
        assert runtime["library"] in runtime["logs"][name], name
    assert verify_files.verify(runtime["library"], d["synthesizer"], d["setup"], d["prove"])
    assert not verify_files.verify(runtime["library"], d["synthesizer"], d["setup"], d["prove"], tamper_public_input=True)
    # and from the verifier's own files alone, as tokamak-cli's verify stage has them (sigma_verify.json instead of the prover's CRS)
    assert verify_files.verify(runtime["library"], d["synthesizer"], d["setup"], d["prove"], from_sigma_verify=True)
    assert not verify_files.verify(runtime["library"], d["synthesizer"], d["setup"], d["prove"], tamper_public_input=True, from_sigma_verify=True)


def test_crs_sections_have_the_sizes_the_real_parameters_dictate(gpu, runtime):
    """Sigma1's tables for s_D = 14, m_D = 26591 (libs/src/group_structures/mod.rs:313-551): gamma_inv_o_inst l entries, eta_inv_li_o_inter_alpha4_kj
    m_I x s_max, delta_inv_li_o_prv (m_D - l_D) x s_max, xy_powers max(2n, 2 m_I) x 2 s_max — read back from the archive the binary wrote"""
    from tkmk import rkyv
    sp = real_library.setup_params()
    buf = np.fromfile(os.path.join(runtime["dirs"]["setup"], "combined_sigma.rkyv"), np.uint8)
    sections = rkyv.decode_combined_sigma(buf, expect=rkyv.expect_for(sp))
    m_i = sp["l_D"] - sp["l"]
    count = lambda k: np.asarray(sections[k]).size // 96                                    # noqa: E731
    assert count("gamma_inv_o_inst") == sp["l"]
    assert count("eta_inv_li_o_inter_alpha4_kj") == m_i * sp["s_max"]
    assert count("delta_inv_li_o_prv") == (sp["m_D"] - sp["l_D"]) * sp["s_max"] == 21767 * 256
    assert count("xy_powers") == max(2 * sp["n"], 2 * m_i) * 2 * sp["s_max"]


def test_resident_and_python_provers_on_the_real_library(gpu, runtime):
    """libtkmk_prover.so opens the same directories (fresh blinding: verifies; fixed blinding: equals the Python prover, which runs the
    reference's testing-mode assertions on this assignment)"""
    import verify_files
    from tkmk import crs as crsmod
    from tkmk import proofio, service
    from tkmk.prove import Prover, random_mixer, run_rounds
    d = runtime["dirs"]
    sp = real_library.setup_params()
    out = os.path.join(runtime["tmp"], "resident_out")
    os.makedirs(out)
    shutil.copy(os.path.join(d["preprocess"], "preprocess.json"), os.path.join(out, "preprocess.json"))
    mixer = random_mixer(random.Random(14))
    mixer_path = os.path.join(runtime["tmp"], "mixer.json")
    hx = lambda v: [hx(e) for e in v] if isinstance(v, list) else "0x%x" % v          # noqa: E731
    json.dump({k: hx(v) for k, v in mixer.items()}, open(mixer_path, "w"))
    with service.Prover(runtime["library"], d["setup"], testing=True) as p:
        _, tm = p.prove(d["synthesizer"], out, want_json=False)
        assert verify_files.verify(runtime["library"], d["synthesizer"], d["setup"], out)
        doc, _ = p.prove(d["synthesizer"], None, testing_mixer_json=mixer_path)
    sections = verify_files.crs_sections(d["setup"], sp)
    sigma1, tables = crsmod.load_sigma1(sections, sp)
    singles = {k: np.array(crsmod.single_g1(sections, k)) for k in ("delta", "eta")}
    prover, binding = Prover.init(runtime["library"], d["synthesizer"], None, mixer=mixer, testing_mode=True, sigma=(sigma1, tables, singles))
    points, scalars, _, _, _ = run_rounds(prover, binding)
    assert proofio.format_proof(points, scalars) == doc
