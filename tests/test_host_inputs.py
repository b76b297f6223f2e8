"""CPU tier: the direct scanner of placementVariables.json in the C++ host side (tokamak-zk-evm_amd/host/tkmk_inputs.hpp; the
reference deserialises the same document with serde, libs/src/iotools/mod.rs:366-372) through tests/host_cpp/inputs_driver:
any key order / whitespace / extra scalar keys parse to the values Python's json gives, malformed documents are errors."""
import json
import os
import random
import subprocess

import pytest

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
DRIVER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_cpp", "inputs_driver")


def _run(path):
    r = subprocess.run([DRIVER, str(path)], capture_output=True, text=True, timeout=60)
    return r.returncode, r.stdout.strip().splitlines()


def _expected(docs):
    lines = ["ok %d %d" % (len(docs), sum(len(d["variables"]) for d in docs))]
    for d in docs:
        x = 0
        for h in d["variables"]:
            x ^= int(h, 16) % R if h not in ("0x", "") else 0
        lines.append("%d %d 0x%064x" % (d["subcircuitId"], len(d["variables"]), x))
    return lines


def test_scanner_equals_json(tkmk, tmp_path):
    assert os.path.exists(DRIVER), "tests/host_cpp/inputs_driver is not built (run __graft_entry__.build())"
    rnd = random.Random(2)
    docs = [{"subcircuitId": rnd.randrange(20), "variables": ["0x%x" % rnd.choice([0, 1, rnd.randrange(R), R - 1, R + 3, rnd.randrange(1 << 64)])
                                                              for _ in range(rnd.randrange(0, 40))]} for _ in range(25)]
    docs.append({"subcircuitId": 3, "variables": []})
    variants = [json.dumps(docs), json.dumps(docs, indent=2), json.dumps(docs, separators=(",", ":")),
                json.dumps([{"variables": d["variables"], "note": "x y", "k": 12, "subcircuitId": d["subcircuitId"]} for d in docs], indent=1),
                "\n \t" + json.dumps(docs) + "\n\n"]
    for k, text in enumerate(variants):
        p = tmp_path / ("pv%d.json" % k)
        p.write_text(text)
        rc, lines = _run(p)
        assert rc == 0 and lines == _expected(docs), k
    p = tmp_path / "empty.json"
    p.write_text("[]")
    assert _run(p) == (0, ["ok 0 0"])


@pytest.mark.parametrize("text", [
    "", "[", "[{", '[{"subcircuitId": 1}]', '[{"variables": ["0x1"]}]', '[{"subcircuitId": 1, "variables": ["0x1"', '[{"subcircuitId": 1, "variables": ["0xzz"]}]',
    '[{"subcircuitId": 1, "variables": ["0x1" "0x2"]}]', '[{"subcircuitId": -1, "variables": []}]', '[{"subcircuitId": 1, "variables": ["0x' + "f" * 65 + '"]}]',
    '[{"subcircuitId": 1, "variables": ["a\\\\u0041"]}]', '{"subcircuitId": 1, "variables": []}', '[{"subcircuitId": 1, "variables": []} {"subcircuitId": 2, "variables": []}]',
    '[{"subcircuitId": 1, "variables": [], }]', '[{"subcircuitId": 1, "variables": ["0x1",]}]'])
def test_scanner_rejects_malformed_documents(tkmk, tmp_path, text):
    p = tmp_path / "bad.json"
    p.write_text(text)
    rc, lines = _run(p)
    assert rc == 1 and lines and lines[0].startswith("error: "), (text, lines)
