"""The three backend binaries under the argv `tokamak-cli` really sends (packages/cli/src/cli.ts:537-546 `backendOutputArgs`:
`--crs D --synthesizer-stat D --output D`; runtime.ts:1840-1848: `trusted-setup --output D --fixed-tau`) — no
--subcircuit-library, which only the reference's non-release builds take (libs/src/subcircuit_library.rs:17-58).
CPU tier: argument parsing and library resolution up to the point where the binary asks for the device (there is no CPU
fallback, so on a box without a GPU a correctly started binary ends with "no HIP device").  The GPU tier runs the same argv to
proof.json / preprocess.json (tests/test_gpu_prove.py::test_binaries_under_the_cli_argv)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tokamak-zk-evm_amd", "bin")
NO_DEVICE = "no HIP device"


def _has_gpu():
    import tkmk
    try:
        return tkmk.device_count() > 0
    except Exception:
        return False


def _run(name, argv, env_extra=None, home=None):
    env = {k: v for k, v in os.environ.items() if k not in ("TKMK_SUBCIRCUIT_LIBRARY", "XDG_CACHE_HOME")}
    if home is not None:
        env["HOME"] = str(home)
    env.update(env_extra or {})
    return subprocess.run([os.path.join(BIN, name)] + argv, capture_output=True, text=True, timeout=120, env=env)


def _library(tmp_path, name="library"):
    d = tmp_path / name
    d.mkdir(parents=True)
    (d / "setupParams.json").write_text('{"l":4,"l_user_out":1,"l_user":2,"l_free":4,"l_D":8,"m_D":16,"n":8,"s_D":2,"s_max":4}')
    return d


CLI_ARGV = {"prove": lambda t: ["--crs", str(t / "crs"), "--synthesizer-stat", str(t / "synth"), "--output", str(t / "out")],
            "preprocess": lambda t: ["--crs", str(t / "crs"), "--synthesizer-stat", str(t / "synth"), "--output", str(t / "out")],
            "trusted-setup": lambda t: ["--output", str(t / "crs"), "--fixed-tau"]}


@pytest.mark.parametrize("name", sorted(CLI_ARGV))
def test_cli_argv_without_any_library_fails_with_the_reason(tkmk, tmp_path, name):
    r = _run(name, CLI_ARGV[name](tmp_path), home=tmp_path / "home")
    assert r.returncode == 1, (r.stdout, r.stderr)                     # not the usage error (2) of round 2
    assert "--subcircuit-library is required" in r.stderr and "TKMK_SUBCIRCUIT_LIBRARY" in r.stderr
    assert "Usage" not in r.stderr


@pytest.mark.parametrize("name", sorted(CLI_ARGV))
def test_cli_argv_finds_the_library_of_the_installation(tkmk, tmp_path, name):
    """env, then <exe>/../resource/qap-compiler/library, then the release binaries' cache directory
    (<cache>/tokamak-zk-evm/subcircuit-library/<snapshot>/library, libs/src/subcircuit_library.rs:66-72)"""
    if _has_gpu():
        pytest.skip("CPU-tier check: with a device the binaries go on into the (dummy) inputs")
    lib = _library(tmp_path)
    r = _run(name, CLI_ARGV[name](tmp_path), {"TKMK_SUBCIRCUIT_LIBRARY": str(lib)}, home=tmp_path / "home")
    assert r.returncode == 1 and NO_DEVICE in r.stderr, r.stderr      # started, resolved, asked for the GPU
    # a wrong env value is an error, not a silent fall-through to another library
    r = _run(name, CLI_ARGV[name](tmp_path), {"TKMK_SUBCIRCUIT_LIBRARY": str(tmp_path / "nowhere")}, home=tmp_path / "home")
    assert r.returncode == 1 and "holds no setupParams.json" in r.stderr
    # the cache of a reference release binary, XDG_CACHE_HOME and HOME/.cache
    snap = tmp_path / "xdg" / "tokamak-zk-evm" / "subcircuit-library" / "2.0.6-abcdef012345"
    _library(snap)
    (tmp_path / "xdg" / "tokamak-zk-evm" / "subcircuit-library" / "staging-1-2").mkdir()
    r = _run(name, CLI_ARGV[name](tmp_path), {"XDG_CACHE_HOME": str(tmp_path / "xdg")}, home=tmp_path / "home")
    assert r.returncode == 1 and NO_DEVICE in r.stderr, r.stderr
    home = tmp_path / "home2"
    _library(home / ".cache" / "tokamak-zk-evm" / "subcircuit-library" / "snap")
    r = _run(name, CLI_ARGV[name](tmp_path), home=home)
    assert r.returncode == 1 and NO_DEVICE in r.stderr, r.stderr
    # tokamak-cli's runtime layout: <runtime>/bin/<binary>, <runtime>/resource/... (runtime.ts:465-485)
    rt = tmp_path / "runtime"
    (rt / "bin").mkdir(parents=True)
    shutil.copy(os.path.join(BIN, name), rt / "bin" / name)
    os.symlink(os.path.join(ROOT, "tokamak-zk-evm_amd", "libtkmk_hip.so"), rt / "libtkmk_hip.so")
    _library(rt / "resource" / "qap-compiler")
    env = {k: v for k, v in os.environ.items() if k not in ("TKMK_SUBCIRCUIT_LIBRARY", "XDG_CACHE_HOME")}
    env["HOME"] = str(tmp_path / "home")
    r = subprocess.run([str(rt / "bin" / name)] + CLI_ARGV[name](tmp_path), capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 1 and NO_DEVICE in r.stderr, r.stderr


@pytest.mark.parametrize("name", sorted(CLI_ARGV))
def test_two_cached_snapshots_are_refused_and_tmp_is_never_searched(tkmk, tmp_path, name):
    """the reference opens the snapshot its embedded integrity hash names; this resolver has no hash, so two candidates are an error that
    lists both (no newest-mtime guess), and without HOME / XDG_CACHE_HOME nothing under $TMPDIR or /tmp is ever taken"""
    root = tmp_path / "xdg" / "tokamak-zk-evm" / "subcircuit-library"
    a, b = _library(root / "2.0.6-aaaaaaaaaaaa"), _library(root / "2.0.7-bbbbbbbbbbbb")
    r = _run(name, CLI_ARGV[name](tmp_path), {"XDG_CACHE_HOME": str(tmp_path / "xdg")}, home=tmp_path / "home")
    assert r.returncode == 1 and "more than one subcircuit-library snapshot" in r.stderr and str(a) in r.stderr and str(b) in r.stderr
    # the explicit choices still work with both present
    if not _has_gpu():
        r = _run(name, CLI_ARGV[name](tmp_path), {"XDG_CACHE_HOME": str(tmp_path / "xdg"), "TKMK_SUBCIRCUIT_LIBRARY": str(b)}, home=tmp_path / "home")
        assert r.returncode == 1 and NO_DEVICE in r.stderr and ("Subcircuit library: " + str(b)) in r.stdout
    # a snapshot under $TMPDIR is not a candidate when there is no per-user cache directory
    _library(tmp_path / "tmpdir" / "tokamak-zk-evm" / "subcircuit-library" / "planted")
    env = {k: v for k, v in os.environ.items() if k not in ("TKMK_SUBCIRCUIT_LIBRARY", "XDG_CACHE_HOME", "HOME")}
    env["TMPDIR"] = str(tmp_path / "tmpdir")
    r = subprocess.run([os.path.join(BIN, name)] + CLI_ARGV[name](tmp_path), capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 1 and "--subcircuit-library is required" in r.stderr and "planted" not in r.stdout


def test_flag_forms_clap_accepts(tkmk, tmp_path):
    """`--flag value` and `--flag=value`, any order; usage errors exit 2 like clap's"""
    lib = _library(tmp_path)
    if not _has_gpu():
        r = _run("prove", ["--output=" + str(tmp_path / "o"), "--subcircuit-library=" + str(lib), "--crs", str(tmp_path / "c"), "--synthesizer-stat=" + str(tmp_path / "s")])
        assert r.returncode == 1 and NO_DEVICE in r.stderr, r.stderr
        r = _run("trusted-setup", ["--fixed-tau", "--output=" + str(tmp_path / "o"), "--subcircuit-library", str(lib)])
        assert r.returncode == 1 and NO_DEVICE in r.stderr, r.stderr
    r = _run("prove", ["--crs", "a", "--synthesizer-stat", "b"])
    assert r.returncode == 2 and "--output <PATH>" in r.stderr and "Usage: prove" in r.stderr
    r = _run("preprocess", ["--crs", "a", "--synthesizer-stat", "b", "--output", "c", "--bogus"])
    assert r.returncode == 2 and "unexpected argument '--bogus'" in r.stderr
    r = _run("preprocess", ["--crs", "a", "--crs", "b", "--synthesizer-stat", "b", "--output", "c"])
    assert r.returncode == 2 and "cannot be used multiple times" in r.stderr
    r = _run("prove", ["--crs", "a", "--synthesizer-stat", "b", "--output"])
    assert r.returncode == 2 and "a value is required" in r.stderr
    r = _run("trusted-setup", ["--fixed-tau=1", "--output", "x"])
    assert r.returncode == 2 and "unexpected value" in r.stderr
    r = _run("prove", ["--subcircuit-library", str(tmp_path / "nowhere"), "--crs", "a", "--synthesizer-stat", "b", "--output", "c"])
    assert r.returncode == 1 and "cannot resolve subcircuit library path" in r.stderr      # subcircuit_library.rs:42-45
    import re
    for name in CLI_ARGV:
        r = _run(name, ["--help"])
        assert r.returncode == 0 and r.stdout.startswith("Usage: " + name)
        # `tokamak-cli doctor` (packages/cli/src/cli.ts:390-404, 655-664): `<binary> --version`, first x.y.z of a line that starts with the name
        for flag in ("--version", "-V"):
            r = _run(name, [flag])
            assert r.returncode == 0 and r.stdout.startswith(name + " ")
            assert re.search(r"\b\d+\.\d+\.\d+(?:[-+][0-9A-Za-z.-]+)?\b", r.stdout).group(0).startswith("2.1.3")


def test_production_prove_refuses_the_fixed_blinding_hook(tkmk, tmp_path):
    """the reference gates fixed blinding scalars behind the compile-time feature `testing-mode`; bin/prove and
    libtkmk_prover.so must not be drivable into a non-zero-knowledge proof at run time"""
    lib = _library(tmp_path)
    argv = ["--crs", "a", "--synthesizer-stat", "b", "--output", "c", "--subcircuit-library", str(lib), "--testing-mixer", str(tmp_path / "m.json")]
    r = _run("prove", argv)
    assert r.returncode == 2 and "testing-mode build" in r.stderr
    assert os.path.exists(os.path.join(BIN, "prove-testing"))
    import re
    src = open(os.path.join(ROOT, "tokamak-zk-evm_amd", "host", "prover_abi.cpp")).read()
    assert re.search(r"#ifdef TKMK_TESTING_MODE\s+Mixer mixer = testing_mixer_json \?", src)
