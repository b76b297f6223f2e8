"""not-gpu tier: the reference's WHOLE production subcircuit library (all fourteen .r1cs files + its own subcircuitInfo.json /
setupParams.json, committed as data by tests/golden/make_pins.py, assembled by tests/real_library.py) through the readers
(tkmk/r1cs.py = libs/src/iotools/mod.rs:505-760; SubcircuitInfo :458-469), and the manufactured assignment of
real_library.make_inputs checked row by row: (A.w)(B.w) = C.w with the oracle's sparse-row evaluation."""
import json
import os
import random

import numpy as np
import pytest

import real_library


@pytest.fixture(scope="module")
def library(tmp_path_factory):
    return real_library.assemble(str(tmp_path_factory.mktemp("real_library")))


def test_the_assembled_directory_is_the_reference_library(library):
    sp = json.load(open(os.path.join(library, "setupParams.json")))
    pins = json.load(open(os.path.join(real_library.GOLDEN, "pins.json")))
    assert sp == pins["setup_params"] and (sp["s_D"], sp["m_D"], sp["l_D"], sp["l"], sp["n"], sp["s_max"]) == (14, 26591, 4824, 728, 4096, 256)
    infos = json.load(open(os.path.join(library, "subcircuitInfo.json")))
    assert [e["id"] for e in infos] == list(range(14)) == sorted(int(f[10:-5]) for f in os.listdir(os.path.join(library, "r1cs")))
    # the global wire numbering the CRS is built for: public wires [0, l), interface [l, l_D), private [l_D, m_D), every index used
    # exactly once across the library except the shared ones inside [l, l_D) — none here: the flattenMaps partition [0, m_D)
    seen = [g for e in infos for g in e["flattenMap"]]
    used_public = {g for g in seen if g < sp["l"]}
    assert max(seen) == sp["m_D"] - 1 and len(used_public) == 65 + 20 + 24 + 600           # l_free - l_user - 24 = 19 public slots are padding
    assert sum(1 for g in seen if g >= sp["l_D"]) == sp["m_D"] - sp["l_D"] == len({g for g in seen if g >= sp["l_D"]})
    # the two kinds with no outputs (Out_idx = [1, 0]) exist in the library: the readers must take an empty range
    assert [e["name"] for e in infos if e["Out_idx"][1] == 0] == ["EdDsaVerify", "VerifyMerkleProof"]


def test_reader_on_all_fourteen_subcircuits(library, oracle):
    """header fields equal subcircuitInfo.json, n >= Nconsts, the CSR form is consistent, and the oracle's row evaluation equals big-int
    arithmetic on the raw file walk (scan_constraints, independent of the numpy path csr() takes) — for every kind, the seven largest
    (270-540 KB, up to 5093 wires) included"""
    from tkmk import r1cs
    sp = json.load(open(os.path.join(library, "setupParams.json")))
    rnd = random.Random(2027)
    total_rows = 0
    for e in json.load(open(os.path.join(library, "subcircuitInfo.json"))):
        path = os.path.join(library, "r1cs", "subcircuit%d.r1cs" % e["id"])
        b = r1cs.R1csBinary.read(path)
        assert (b.n_wires, b.n_constraints, b.field_size) == (e["Nwires"], e["Nconsts"], 32) and b.prime() == oracle.R_MOD
        assert len(e["flattenMap"]) == e["Nwires"] and e["Nconsts"] <= sp["n"]
        s = r1cs.SubcircuitR1CS.from_r1cs_sparse_only(path, sp, e)
        var = [rnd.choice((0, 1, oracle.R_MOD - 1, rnd.randrange(oracle.R_MOD))) for _ in range(e["Nwires"])]
        want = [[0] * e["Nconsts"] for _ in range(3)]
        for m, wire, coeff, row in b.scan_constraints():
            want[m][row] = (want[m][row] + int.from_bytes(coeff, "little") * var[wire]) % oracle.R_MOD
        V = oracle.to_bytes(var, 32)
        for m, (ptr, wires, coeffs) in enumerate(s.csr):
            assert len(ptr) == e["Nconsts"] + 1 and ptr[-1] == wires.size == coeffs.size // 32
            assert wires.size == 0 or int(wires.max()) < e["Nwires"]
            got = oracle.to_ints(oracle.r1cs_eval_rows(ptr, wires, coeffs if coeffs.size else np.zeros(32, np.uint8), V, e["Nconsts"]), 32)
            assert got == want[m], (e["id"], m)
        total_rows += e["Nconsts"]
    assert total_rows == 24275                      # sum of Nconsts over the committed library


def test_manufactured_assignment_satisfies_every_row_and_every_copy_constraint(library, oracle, tmp_path):
    from tkmk import r1cs
    sp = json.load(open(os.path.join(library, "setupParams.json")))
    infos = {e["id"]: e for e in json.load(open(os.path.join(library, "subcircuitInfo.json")))}
    made = real_library.make_inputs(str(tmp_path), random.Random(7))
    assert set(made["order"]) == set(range(14)) and made["order"][:4] == [0, 1, 2, 3]
    csr = {sid: r1cs.R1csBinary.read(os.path.join(library, "r1cs", "subcircuit%d.r1cs" % sid)).csr() for sid in infos}
    nonzero_rows = 0
    for sid, w in zip(made["order"], made["values"]):
        V = oracle.to_bytes(w, 32)
        a, b, c = (oracle.to_ints(oracle.r1cs_eval_rows(p, wi, co if co.size else np.zeros(32, np.uint8), V, infos[sid]["Nconsts"]), 32)
                   for p, wi, co in csr[sid])
        assert all((x * y - z) % oracle.R_MOD == 0 for x, y, z in zip(a, b, c)), infos[sid]["name"]
        nonzero_rows += sum(1 for x, y, z in zip(a, b, c) if x or y or z)
    assert nonzero_rows > 200                       # the DecToBit rows carry non-zero values ((b - 1) b with b = 1 has A = 0, B = 1)
    # copy constraints: both ends of every entry are interface cells holding one value
    l, m_i = sp["l"], sp["l_D"] - sp["l"]
    cell_value = {}
    for p, (sid, w) in enumerate(zip(made["order"], made["values"])):
        for wire, g in enumerate(infos[sid]["flattenMap"]):
            if l <= g < sp["l_D"]:
                cell_value[(g - l, p)] = w[wire]
    assert len(made["permutation"]) > 40
    for e in made["permutation"]:
        assert 0 <= e["row"] < m_i and 0 <= e["X"] < m_i
        assert cell_value[(e["row"], e["col"])] == cell_value[(e["X"], e["Y"])]
    assert any(cell_value[(e["row"], e["col"])] > 1 for e in made["permutation"])
    # the documents parse with the readers the Python prover uses, and the public inputs have the lengths Instance::gen_a_free_X and
    # encode_o_pub_fix_common index (libs/src/polynomial_structures/mod.rs:103-128, group_structures/mod.rs:145-182)
    ins = json.load(open(tmp_path / "instance.json"))
    assert len(ins["a_pub_user"]) == sp["l_user"] and len(ins["a_pub_block"]) == sp["l_free"] - sp["l_user"] and len(ins["a_pub_function"]) == sp["l"] - sp["l_free"]
