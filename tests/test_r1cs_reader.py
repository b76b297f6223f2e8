"""not-gpu tier: the `.r1cs` reader (tkmk/r1cs.py, mirror of libs/src/iotools/mod.rs:505-760) on seven compiled
subcircuits taken as DATA from the reference's committed library (tests/golden/qap: three small ones; tests/golden/qap_more: four
mid-size ones, up to 1201 wires; copied by tests/golden/make_pins.py), plus the oracle's eval_sparse_rows restatement against
Python big ints."""
import json
import os
import random
import struct
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
QAP = os.path.join(HERE, "golden", "qap")
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tokamak-zk-evm_amd"))


@pytest.fixture(scope="module")
def r1cs_mod():
    from tkmk import r1cs
    return r1cs


@pytest.fixture(scope="module")
def qap():
    infos = json.load(open(os.path.join(QAP, "subcircuitInfo.json")))
    params = json.load(open(os.path.join(QAP, "setupParams.json")))
    return infos, params


QAP_MORE = os.path.join(HERE, "golden", "qap_more")


def test_reader_and_oracle_rows_on_the_mid_size_subcircuits(r1cs_mod, qap, oracle):
    """bufferPubOut, bufferEVMIn (1201 wires), DecToBit, Accumulator of the production library: header fields equal subcircuitInfo.json,
    the CSR form is consistent, and the oracle's sparse-row evaluation equals big-int arithmetic on the raw file walk
    (scan_constraints: the reader's slow path, independent of the numpy fast path csr() takes)"""
    _, params = qap
    rnd = random.Random(14)
    for e in json.load(open(os.path.join(QAP_MORE, "subcircuitInfo.json"))):
        path = os.path.join(QAP_MORE, "r1cs", "subcircuit%d.r1cs" % e["id"])
        b = r1cs_mod.R1csBinary.read(path)
        assert (b.n_wires, b.n_constraints, b.field_size) == (e["Nwires"], e["Nconsts"], 32) and b.prime() == oracle.R_MOD
        var = [rnd.choice((0, 1, rnd.randrange(oracle.R_MOD))) for _ in range(e["Nwires"])]
        want = [[0] * e["Nconsts"] for _ in range(3)]
        for m, wire, coeff, row in b.scan_constraints():
            want[m][row] = (want[m][row] + int.from_bytes(coeff, "little") * var[wire]) % oracle.R_MOD
        V = oracle.to_bytes(var, 32)
        for m, (ptr, wires, coeffs) in enumerate(b.csr()):
            assert len(ptr) == e["Nconsts"] + 1 and ptr[-1] == wires.size == coeffs.size // 32
            got = oracle.to_ints(oracle.r1cs_eval_rows(ptr, wires, coeffs if coeffs.size else np.zeros(32, np.uint8), V, e["Nconsts"]), 32)
            assert got == want[m], (e["id"], m)


def test_reader_matches_subcircuit_info(r1cs_mod, qap, oracle):
    infos, params = qap
    for e in infos:
        b = r1cs_mod.R1csBinary.read(os.path.join(QAP, "r1cs", "subcircuit%d.r1cs" % e["id"]))
        assert (b.n_wires, b.n_constraints, b.field_size) == (e["Nwires"], e["Nconsts"], 32)
        assert b.prime() == oracle.R_MOD                       # every circuit is over the BLS12-381 scalar field
        s = r1cs_mod.SubcircuitR1CS.from_r1cs_sparse_only(os.path.join(QAP, "r1cs", "subcircuit%d.r1cs" % e["id"]), params, e)
        for m in range(3):
            ptr, wires, coeffs = s.csr[m]
            assert len(ptr) == e["Nconsts"] + 1 and ptr[-1] == wires.size == coeffs.size // 32
            assert all(w < e["Nwires"] for w in wires)
            assert s.active_wires(m) == sorted(set(wires.tolist()))


def test_reader_rejects_corrupt_files(r1cs_mod, qap, tmp_path):
    infos, params = qap
    good = open(os.path.join(QAP, "r1cs", "subcircuit12.r1cs"), "rb").read()
    info = [e for e in infos if e["id"] == 12][0]

    def write(b):
        p = tmp_path / "x.r1cs"
        p.write_bytes(b)
        return str(p)

    with pytest.raises(r1cs_mod.R1csError, match="magic"):
        r1cs_mod.R1csBinary.read(write(b"R1CS" + good[4:]))
    with pytest.raises(r1cs_mod.R1csError, match="version"):
        r1cs_mod.R1csBinary.read(write(good[:4] + struct.pack("<I", 2) + good[8:]))
    with pytest.raises(r1cs_mod.R1csError):
        r1cs_mod.R1csBinary.read(write(good[:-5]))                          # truncated section
    with pytest.raises(r1cs_mod.R1csError, match="nWires mismatch"):
        r1cs_mod.SubcircuitR1CS.from_r1cs_sparse_only(write(good), params, dict(info, Nwires=info["Nwires"] + 1))
    with pytest.raises(r1cs_mod.R1csError, match="nConstraints mismatch"):
        r1cs_mod.SubcircuitR1CS.from_r1cs_sparse_only(write(good), params, dict(info, Nconsts=info["Nconsts"] + 1))
    with pytest.raises(r1cs_mod.R1csError, match="smaller"):
        r1cs_mod.SubcircuitR1CS.from_r1cs_sparse_only(write(good), dict(params, n=4), info)


def test_hex_scalars(r1cs_mod, oracle):
    assert r1cs_mod.hex_to_fr("0x01") == 1 and r1cs_mod.hex_to_fr("ff") == 255 and r1cs_mod.hex_to_fr("0x") == 0
    assert r1cs_mod.hex_to_fr("0x" + "f" * 64) == (2 ** 256 - 1) % oracle.R_MOD


def test_oracle_row_evaluation_vs_python(r1cs_mod, qap, oracle):
    infos, params = qap
    rnd = random.Random(4)
    for e in infos:
        s = r1cs_mod.SubcircuitR1CS.from_r1cs_sparse_only(os.path.join(QAP, "r1cs", "subcircuit%d.r1cs" % e["id"]), params, e)
        var = [rnd.randrange(oracle.R_MOD) for _ in range(e["Nwires"])]
        V = oracle.to_bytes(var, 32)
        for m in range(3):
            ptr, wires, coeffs = s.csr[m]
            cv = oracle.to_ints(coeffs, 32) if coeffs.size else []
            want = [sum(cv[k] * var[wires[k]] for k in range(ptr[r], ptr[r + 1])) % oracle.R_MOD for r in range(e["Nconsts"])]
            got = oracle.to_ints(oracle.r1cs_eval_rows(ptr, wires, coeffs if coeffs.size else np.zeros(32, np.uint8), V, params["n"]), 32)
            assert got[:e["Nconsts"]] == want and not any(got[e["Nconsts"]:])
