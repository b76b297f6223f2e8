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


# ---- the multi-threaded readers of host/tkmk_fastparse.hpp (what the resident prover uses per proof) ----
def _run_args(*args):
    r = subprocess.run([DRIVER] + [str(a) for a in args], capture_output=True, text=True, timeout=60)
    return r.returncode, r.stdout.strip().splitlines()


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_fast_reader_equals_serial_scanner_and_json(tkmk, tmp_path, threads):
    rnd = random.Random(5)
    n_wires = [rnd.randrange(0, 30) for _ in range(6)]
    n_wires[2] = 0
    docs = [{"subcircuitId": k, "variables": ["0x%x" % rnd.choice([0, 1, rnd.randrange(R), R - 1, R + 3, rnd.randrange(1 << 64), rnd.randrange(1 << 255)])
                                              for _ in range(n_wires[k])]} for k in [rnd.randrange(6) for _ in range(300)]]
    nw = ",".join(str(v) for v in n_wires)
    variants = [json.dumps(docs), json.dumps(docs, indent=2), json.dumps(docs, separators=(",", ":")),
                json.dumps([{"variables": d["variables"], "note": "x y", "k": 12, "subcircuitId": d["subcircuitId"]} for d in docs], indent=1),
                "\n \t" + json.dumps(docs) + "\n\n", json.dumps(docs[:1]), json.dumps(docs[:2])]
    for k, text in enumerate(variants):
        p = tmp_path / ("pv%d.json" % k)
        p.write_text(text)
        want = _expected(json.loads(text))
        assert _run_args("fast", p, threads, nw) == (0, want), k
        assert _run(p) == (0, want), k                                   # the serial scanner agrees
    p = tmp_path / "empty.json"
    p.write_text(" [ ] ")
    assert _run_args("fast", p, threads, nw) == (0, ["ok 0 0"])
    # a placement with the wrong number of variables for its kind, or an unknown kind: the reference's messages
    for bad, msg in (([{"subcircuitId": 0, "variables": ["0x1"] * (n_wires[0] + 1)}], "Corrupted placement variables."),
                     ([{"subcircuitId": 1, "variables": ["0x1"] * max(0, n_wires[1] - 1)}] if n_wires[1] else [{"subcircuitId": 0, "variables": []}] if n_wires[0] else None,
                      "Corrupted placement variables."),
                     ([{"subcircuitId": 6, "variables": []}], "Invalid subcircuit id in placement_variables.")):
        if bad is None:
            continue
        p = tmp_path / "bad.json"
        p.write_text(json.dumps(docs[:5] + bad + docs[5:9]))
        rc, lines = _run_args("fast", p, threads, nw)
        assert rc == 1 and msg in lines[0], (bad, lines)


@pytest.mark.parametrize("text", [
    "", "[", "[{", '[{"subcircuitId": 1}]', '[{"variables": ["0x1"]}]', '[{"subcircuitId": 1, "variables": ["0x1"', '[{"subcircuitId": 1, "variables": ["0xzz"]}]',
    '[{"subcircuitId": 1, "variables": ["0x1" "0x2"]}]', '[{"subcircuitId": -1, "variables": []}]', '[{"subcircuitId": 1, "variables": ["0x' + "f" * 65 + '"]}]',
    '[{"subcircuitId": 1, "variables": ["a\\\\u0041"]}]', '{"subcircuitId": 1, "variables": []}', '[{"subcircuitId": 1, "variables": []} {"subcircuitId": 2, "variables": []}]',
    '[{"subcircuitId": 1, "variables": [], }]', '[{"subcircuitId": 1, "variables": ["0x1",]}]', '[{"subcircuitId": 1, "variables": []}] x', 'x [{"subcircuitId": 1, "variables": []}]',
    '[{"subcircuitId": 1, "variables": []},]', '[{"subcircuitId": 1, "subcircuitId": 1, "variables": []}]', '[{"subcircuitId": 1, "variables": [], "z": {"a": 1}}]'])
def test_fast_reader_rejects_malformed_documents(tkmk, tmp_path, text):
    p = tmp_path / "bad.json"
    p.write_text(text)
    for threads in (1, 4):
        rc, lines = _run_args("fast", p, threads, "1,1,2")
        assert rc == 1 and lines and lines[0].startswith("error: "), (text, lines)


def test_fast_permutation_reader(tkmk, tmp_path):
    rnd = random.Random(6)
    perm = [{"row": rnd.randrange(4096), "col": rnd.randrange(1024), "X": rnd.randrange(4096), "Y": rnd.randrange(1024)} for _ in range(5000)]
    want = ["ok %d" % len(perm)] + ["%d %d %d %d" % (e["row"], e["col"], e["X"], e["Y"]) for e in perm]
    shuffled = [{"Y": e["Y"], "note": "n", "row": e["row"], "X": e["X"], "col": e["col"]} for e in perm]
    for k, text in enumerate((json.dumps(perm), json.dumps(perm, indent=1), json.dumps(shuffled), json.dumps(perm, separators=(",", ":")))):
        p = tmp_path / ("perm%d.json" % k)
        p.write_text(text)
        for threads in (1, 5):
            assert _run_args("perm", p, threads) == (0, want), (k, threads)
    p = tmp_path / "e.json"
    p.write_text("[]")
    assert _run_args("perm", p, 2) == (0, ["ok 0"])
    for bad in ('[{"row": 1, "col": 2, "X": 3}]', '[{"row": 1, "col": 2, "X": 3, "Y": 4}', '[{"row": 1, "col": 2, "X": 3, "Y": -4}]', '[{"row": 1, "row": 1, "col": 2, "X": 3, "Y": 4}]',
                '[{"row": 1, "col": 2, "X": 3, "Y": 4} {"row": 1, "col": 2, "X": 3, "Y": 4}]', '[{"row": 1, "col": 2, "X": 3, "Y": 99999999999}]'):
        p.write_text(bad)
        rc, lines = _run_args("perm", p, 2)
        assert rc == 1 and lines[0].startswith("error: "), (bad, lines)


def test_fast_reader_every_digit_count_and_case(tkmk, tmp_path):
    """the 16-digits-at-a-time decoder of the bulk path: every length 1..64, upper / lower / mixed case, with and without the 0x
    prefix, values at and above the modulus; a bad character in every position of a 64-digit value is an error"""
    rnd = random.Random(11)
    vals = []
    for nd in range(1, 65):
        v = rnd.getrandbits(4 * nd) | (1 << (4 * nd - 1)) if nd < 64 else rnd.getrandbits(256)
        h = "%x" % v
        h = h.rjust(nd, "0")[:nd] if len(h) <= nd else h[-nd:]
        for form in (h, h.upper(), "".join(c.upper() if rnd.random() < 0.5 else c for c in h)):
            for prefix in ("0x", "0X", ""):
                vals.append(prefix + form)
    vals += ["0x" + "f" * 64, "0x%x" % R, "0x%x" % (R - 1), "0x" + "0" * 64, "0x", ""]
    docs = [{"subcircuitId": 0, "variables": vals}]
    p = tmp_path / "pv.json"
    p.write_text(json.dumps(docs))
    x = 0
    for h in vals:
        body = h[2:] if h[:2] in ("0x", "0X") else h
        x ^= (int(body, 16) if body else 0) % R
    for threads in (1, 4):
        assert _run_args("fast", p, threads, str(len(vals))) == (0, ["ok 1 %d" % len(vals), "0 %d 0x%064x" % (len(vals), x)])
    good = "ab" * 32
    for pos in range(64):
        for bad_char in ("g", "/", ":", "@", "`", " ", "\\\\"):
            broken = good[:pos] + bad_char + good[pos + 1:]
            if len(bad_char) > 1:
                broken = good[:pos] + bad_char + good[pos + 2:] if pos < 63 else good[:62] + bad_char
            p.write_text('[{"subcircuitId": 0, "variables": ["0x%s"]}]' % broken)
            rc, lines = _run_args("fast", p, 1, "1")
            assert rc == 1 and lines[0].startswith("error: "), (pos, bad_char, lines)
