"""not-gpu tier: the host-only parts of the C++ host side (the multi-threaded JSON readers, the rkyv archive reader / writer, host Fr
and Fq arithmetic behind them) rebuilt with AddressSanitizer + UndefinedBehaviorSanitizer and run on valid, malformed and truncated
inputs.  GPU AddressSanitizer is not available on this pool, so this is where memory errors in the host code would surface
(SURVEY.md section 5).  The sanitized drivers are built on first use (tests/host_cpp/*_san, git-ignored)."""
import json
import os
import random
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PKG = os.path.join(ROOT, "tokamak-zk-evm_amd")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


@pytest.fixture(scope="module")
def san(tkmk):
    out = {}
    for name in ("inputs_driver", "rkyv_driver", "args_driver"):
        src = os.path.join(HERE, "host_cpp", name + ".cpp")
        exe = os.path.join(HERE, "host_cpp", name + "_san")
        deps = [src] + [os.path.join(PKG, "host", f) for f in os.listdir(os.path.join(PKG, "host")) if f.endswith(".hpp")]
        if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
            r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
                                "-I" + os.path.join(PKG, "host"), src, "-o", exe, "-L" + PKG, "-ltkmk_hip", "-pthread", "-Wl,-rpath," + PKG],
                               capture_output=True, text=True, timeout=900)
            assert r.returncode == 0, r.stderr[-3000:]
        out[name] = exe
    return out


def _run(exe, *args):
    r = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=300, env=ENV)
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    return r


def test_json_readers_under_asan_ubsan(san, tmp_path):
    rnd = random.Random(9)
    n_wires = [40, 7, 0, 300]
    docs = [{"subcircuitId": k, "variables": ["0x%x" % rnd.choice([0, 1, rnd.randrange(R), R - 1, rnd.randrange(1 << 256)]) for _ in range(n_wires[k])]}
            for k in [rnd.randrange(4) for _ in range(500)]]
    text = json.dumps(docs)
    (tmp_path / "pv.json").write_text(text)
    nw = ",".join(map(str, n_wires))
    ok = _run(san["inputs_driver"], "fast", tmp_path / "pv.json", 8, nw)
    assert ok.returncode == 0 and ok.stdout.startswith("ok 500 ")
    assert _run(san["inputs_driver"], tmp_path / "pv.json").stdout == ok.stdout
    # every prefix class of a truncated / damaged document is an error, never a bad access
    for cut in sorted({rnd.randrange(len(text)) for _ in range(60)} | {0, 1, 2, len(text) - 1}):
        (tmp_path / "cut.json").write_text(text[:cut])
        assert _run(san["inputs_driver"], "fast", tmp_path / "cut.json", 5, nw).returncode == 1
        assert _run(san["inputs_driver"], tmp_path / "cut.json").returncode == 1
    for _ in range(40):
        b = bytearray(text.encode())
        for _ in range(3):
            b[rnd.randrange(len(b))] = rnd.choice(b'{}[]",:x0 \\')
        (tmp_path / "fz.json").write_bytes(bytes(b))
        assert _run(san["inputs_driver"], "fast", tmp_path / "fz.json", 4, nw).returncode in (0, 1)
    perm = [{"row": rnd.randrange(64), "col": rnd.randrange(8), "X": rnd.randrange(64), "Y": rnd.randrange(8)} for _ in range(2000)]
    ptext = json.dumps(perm)
    (tmp_path / "perm.json").write_text(ptext)
    assert _run(san["inputs_driver"], "perm", tmp_path / "perm.json", 6).stdout.startswith("ok 2000")
    for cut in (0, 1, 17, len(ptext) // 2, len(ptext) - 1):
        (tmp_path / "pc.json").write_text(ptext[:cut])
        assert _run(san["inputs_driver"], "perm", tmp_path / "pc.json", 3).returncode == 1


def test_rkyv_reader_and_writer_under_asan_ubsan(san, oracle, tmp_path):
    from test_rkyv import SP, _curve_sections
    from tkmk import crs, rkyv
    sec, rows, ex = _curve_sections(oracle, SP)
    (tmp_path / "p.tkcrs").write_bytes(crs.build_payload(sec))
    rnd = random.Random(4)
    for order in rkyv.ORDERS:
        archive = rkyv.encode_combined_sigma(sec, rows, order)
        (tmp_path / "a.rkyv").write_bytes(archive)
        args = [ex["xy_powers"], ex["gamma_inv_o_inst"], ex["eta_inv_li_o_inter_alpha4_kj"], ex["delta_inv_li_o_prv"], ex["rs_y"], "auto"]
        r = _run(san["rkyv_driver"], "decode", tmp_path / "a.rkyv", tmp_path / "o.tkcrs", *args)
        assert r.returncode == 0 and r.stdout.strip() == order
        r = _run(san["rkyv_driver"], "encode", tmp_path / "p.tkcrs", tmp_path / "b.rkyv", order, len(rows["eta_inv_li_o_inter_alpha4_kj"]), len(rows["delta_inv_li_o_prv"]))
        assert r.returncode == 0 and (tmp_path / "b.rkyv").read_bytes() == archive
        # damaged archives: flipped bytes in the root object and the row headers, truncations — refused or (if the damage is in
        # point data only) decoded, never a bad access
        for _ in range(60):
            b = bytearray(archive)
            lo = rnd.choice([len(b) - 2552, len(b) - 4000, 0])
            b[rnd.randrange(max(lo, 0), len(b))] ^= 1 << rnd.randrange(8)
            (tmp_path / "d.rkyv").write_bytes(bytes(b))
            assert _run(san["rkyv_driver"], "decode", tmp_path / "d.rkyv", tmp_path / "x", *args).returncode in (0, 1)
        for cut in (0, 5, 96, len(archive) - 2552, len(archive) - 1):
            (tmp_path / "t.rkyv").write_bytes(archive[:cut])
            assert _run(san["rkyv_driver"], "decode", tmp_path / "t.rkyv", tmp_path / "x", *args).returncode == 1
    pre = rkyv.encode_sigma_preprocess(sec["xy_powers"], sec["gamma_inv_o_inst"])
    for cut in (0, 8, 16, len(pre) - 3, len(pre)):
        (tmp_path / "pre.rkyv").write_bytes(pre[:cut])
        assert _run(san["rkyv_driver"], "pre-decode", tmp_path / "pre.rkyv", tmp_path / "x").returncode == (0 if cut == len(pre) else 1)


def test_argument_parser_and_library_resolution_under_asan_ubsan(san, tmp_path):
    """host/tkmk_args.hpp (the argv tokamak-cli sends, clap's flag forms, the library search) on ordinary, odd and hostile argument
    vectors: a result or an error, never a bad access"""
    lib = tmp_path / "lib"
    lib.mkdir()
    (lib / "setupParams.json").write_text("{}")
    exe = san["args_driver"]
    env = dict(ENV, HOME=str(tmp_path / "home"))
    env.pop("TKMK_SUBCIRCUIT_LIBRARY", None)
    env.pop("XDG_CACHE_HOME", None)

    def run(argv, extra=None):
        r = subprocess.run([exe] + argv, capture_output=True, text=True, timeout=60, env=dict(env, **(extra or {})))
        assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
        return r
    r = run(["--crs", "a", "--synthesizer-stat=b", "--output", "c", "--subcircuit-library", str(lib)])
    assert r.returncode == 0 and r.stdout.strip() == "ok a|b|c|%s|0" % os.path.realpath(lib)
    r = run(["--crs=a=b", "--fixed-tau"], {"TKMK_SUBCIRCUIT_LIBRARY": str(lib)})
    assert r.returncode == 0 and r.stdout.startswith("ok a=b|||") and r.stdout.strip().endswith("|1")
    assert run(["--crs", "a"]).returncode == 1                                      # nothing to resolve from
    assert run(["--crs"]).returncode == 2 and run(["--fixed-tau=1"]).returncode == 2 and run(["--crs", "a", "--crs", "b"]).returncode == 2
    rnd = random.Random(3)
    atoms = ["--crs", "--crs=", "--output", "--subcircuit-library", "--subcircuit-library=" + str(lib), "--fixed-tau", "--", "-", "", "=", "--=x", "x" * 5000,
             "--synthesizer-stat=" + "y" * 3000, str(lib), "--help", "-h", "\xff\xfe", "--crs=\n"]
    for _ in range(150):
        argv = [rnd.choice(atoms) for _ in range(rnd.randrange(0, 9))]
        assert run(argv, rnd.choice([None, {"TKMK_SUBCIRCUIT_LIBRARY": str(lib)}, {"XDG_CACHE_HOME": str(tmp_path)}, {"HOME": ""}])).returncode in (0, 1, 2)
    # a cache directory full of odd entries
    snaps = tmp_path / "tokamak-zk-evm" / "subcircuit-library"
    for name in ("staging-1-2", "a", "b" * 200, ".hidden"):
        (snaps / name / "library").mkdir(parents=True)
    (snaps / "a" / "library" / "setupParams.json").write_text("{}")
    (snaps / "plainfile").write_text("x")
    r = run(["--crs", "a"], {"XDG_CACHE_HOME": str(tmp_path)})
    assert r.returncode == 0 and r.stdout.strip().endswith("/a/library|0")
