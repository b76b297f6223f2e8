"""not-gpu tier: the rkyv archive reader / writer (tkmk/rkyv.py and its C++ twin host/tkmk_rkyv.hpp behind tests/host_cpp/rkyv_driver).

No archive ships with the reference ("parity unpinned" for the container), so the pins are:
  * the shape test of the reference's own decoder (backend-wasm/tools/rkyv-decoder-wasm/src/lib.rs:209-272): its sample
    SigmaRkyv, serialized by the restated rkyv::to_bytes, must decode into the nine sections with the sizes and first / last
    records that test asserts;
  * the archive layout facts of rkyv 0.7 restated in host/tkmk_rkyv.hpp (root last, relative pointers from the field's own
    position, row headers after rows), checked on the raw bytes here;
  * round trips under every field order the reader knows, with auto-detection by validation on real curve points;
  * byte equality between the Python and the C++ implementations, both directions."""
import os
import struct
import subprocess

import numpy as np
import pytest

from tkmk import crs, rkyv

HERE = os.path.dirname(os.path.abspath(__file__))
DRIVER = os.path.join(HERE, "host_cpp", "rkyv_driver")


def g1(seed):
    return bytes((seed + i) % 256 for i in range(48)) + bytes((seed + 48 + i) % 256 for i in range(48))


def g2(seed):
    return bytes((seed + i) % 256 for i in range(96)) + bytes((seed + 96 + i) % 256 for i in range(96))


def sample_sigma():
    """sample_sigma() of the decoder's test (lib.rs:234-272) as decoder sections + row structure"""
    sections = {"g1": g1(1) + g1(5) + g1(6) + g1(7) + g1(8) + g1(29), "xy_powers": g1(3) + g1(4), "gamma_inv_o_inst": g1(9),
                "eta_inv_li_o_inter_alpha4_kj": g1(10) + g1(11) + g1(12), "delta_inv_li_o_prv": g1(13),
                "delta_inv_alphak_xh_tx": g1(14) + g1(15) + g1(16), "delta_inv_alpha4_xj_tx": g1(17), "delta_inv_alphak_yi_ty": g1(18) + g1(19),
                "g2": b"".join(g2(k) for k in [2] + list(range(20, 29)))}
    rows = {"eta_inv_li_o_inter_alpha4_kj": [2, 1], "delta_inv_li_o_prv": [1], "delta_inv_alphak_xh_tx": [1, 2], "delta_inv_alphak_yi_ty": [2]}
    return sections, rows


@pytest.mark.parametrize("order", rkyv.ORDERS)
def test_decoder_shape_test_of_the_reference(order):
    sections, rows = sample_sigma()
    archive = rkyv.encode_combined_sigma(sections, rows, order)
    sec = rkyv.decode_combined_sigma(archive, orders=(order,), check_points=False)
    want = [6, 2, 1, 3, 1, 3, 1, 2]                                   # lib.rs:218-226, in G1 points
    assert [sec[n].size // 96 for n in crs.SECTION_NAMES[:8]] == want and sec["g2"].size == 10 * 192
    assert bytes(sec["g1"][:96]) == g1(1) and bytes(sec["g1"][96:192]) == g1(5) and bytes(sec["g1"][480:576]) == g1(29)   # lib.rs:228-230
    assert bytes(sec["g2"][:192]) == g2(2) and bytes(sec["g2"][9 * 192:]) == g2(28)                                         # lib.rs:231-232
    for n in crs.SECTION_NAMES:
        assert bytes(sec[n]) == sections[n], n
    with pytest.raises(rkyv.RkyvFormatError):
        rkyv.decode_combined_sigma(b"not an archive")                 # lib.rs:236-239


def test_archive_layout_facts():
    """root object last; vector headers = (i32 offset from the header's own position, u32 length); nested rows precede their headers"""
    sections, rows = sample_sigma()
    a = rkyv.encode_combined_sigma(sections, rows, "rustc_size_groups")
    s1f, rf, root_size, root_align = rkyv.sigma_layout("rustc_size_groups")
    assert root_size == 2552 and root_align == 4 and len(a) % 4 == 0
    assert (rf["H"], rf["sigma_2"], rf["G"], rf["lagrange_KL"], rf["sigma_1"]) == (0, 192, 1920, 2016, 2112)
    assert [s1f[f] for f in ("x", "y", "delta", "eta", "xy_powers", "gamma_inv_o_inst")] == [0, 96, 192, 288, 384, 392]
    root = len(a) - root_size
    assert a[:192] == g1(3) + g1(4) and a[192:288] == g1(9)           # serialization starts with xy_powers, then gamma_inv_o_inst
    at = root + rf["sigma_1"] + s1f["xy_powers"]
    rel, ln = struct.unpack("<iI", a[at:at + 8])
    assert at + rel == 0 and ln == 2
    at = root + rf["sigma_1"] + s1f["eta_inv_li_o_inter_alpha4_kj"]
    rel, ln = struct.unpack("<iI", a[at:at + 8])
    heads = at + rel
    assert ln == 2 and heads % 4 == 0 and heads == 288 + 3 * 96       # two rows (2 + 1 points) written first, headers right after (576 is 4-aligned)
    r0, l0 = struct.unpack("<iI", a[heads:heads + 8])
    r1, l1 = struct.unpack("<iI", a[heads + 8:heads + 16])
    assert (heads + r0, l0, heads + 8 + r1, l1) == (288, 2, 288 + 192, 1)
    # the other two orders place the fields differently but describe the same data
    assert rkyv.sigma_layout("rustc_align_only")[1]["sigma_1"] == 0 and rkyv.sigma_layout("declared")[1]["G"] == 0
    assert rkyv.sigma_layout("declared")[1]["sigma_1"] == 288        # G 96 + H 192


def _curve_sections(oracle, sp):
    """a structurally faithful CRS for setup params sp with real curve points ([k]G for distinct k), rows as Sigma1::gen makes them"""
    ex = rkyv.expect_for(sp)
    rows = rkyv.rows_for(sp)
    total = 6 + ex["xy_powers"] + ex["gamma_inv_o_inst"] + ex["eta_inv_li_o_inter_alpha4_kj"] + ex["delta_inv_li_o_prv"] + 9 + 2 + 12
    pts = np.asarray(oracle.g1_random_bases(77, total)).reshape(-1, 96)
    it = iter(range(total))
    take = lambda k: np.concatenate([pts[next(it)] for _ in range(k)]) if k else np.zeros(0, np.uint8)   # noqa: E731
    sec = {"g1": take(6)}
    xy = take(ex["xy_powers"]).copy()
    xy[:96] = sec["g1"][:96]                                          # xy_powers[0] = G, [1] = y, [rs_y] = x
    xy[96:192] = sec["g1"][192:288]
    xy[96 * ex["rs_y"]:96 * (ex["rs_y"] + 1)] = sec["g1"][96:192]
    sec["xy_powers"] = xy
    sec["gamma_inv_o_inst"] = take(ex["gamma_inv_o_inst"])
    sec["eta_inv_li_o_inter_alpha4_kj"] = take(ex["eta_inv_li_o_inter_alpha4_kj"])
    sec["delta_inv_li_o_prv"] = take(ex["delta_inv_li_o_prv"])
    sec["delta_inv_alphak_xh_tx"], sec["delta_inv_alpha4_xj_tx"], sec["delta_inv_alphak_yi_ty"] = take(9), take(2), take(12)
    sec["g2"] = np.frombuffer(b"".join(g2(k) for k in range(10)), np.uint8)
    return sec, rows, ex


SP = {"l": 8, "l_user_out": 1, "l_user": 3, "l_free": 4, "l_D": 16, "m_D": 21, "n": 8, "s_D": 6, "s_max": 4}


@pytest.mark.parametrize("order", rkyv.ORDERS)
def test_round_trip_and_order_detection(oracle, order):
    sec, rows, ex = _curve_sections(oracle, SP)
    archive = rkyv.encode_combined_sigma(sec, rows, order)
    got, got_rows, got_order = rkyv.decode_combined_sigma(archive, expect=ex, want_details=True)
    assert got_order == order                                         # the wrong orders fail validation, the right one passes
    assert got_rows == rows
    for n in crs.SECTION_NAMES:
        assert bytes(got[n]) == bytes(sec[n]), n
    assert crs.check_shapes(got, SP) == 8
    # a different circuit's archive is refused, so is a damaged one
    with pytest.raises(rkyv.RkyvFormatError):
        rkyv.decode_combined_sigma(archive, expect=dict(ex, xy_powers=ex["xy_powers"] * 2))
    with pytest.raises(rkyv.RkyvFormatError):
        rkyv.decode_combined_sigma(archive[:-8], expect=ex)
    s1f, rf, root_size, _ = rkyv.sigma_layout(order)
    bad = bytearray(archive)
    bad[len(bad) - root_size + rf["G"] + 5] ^= 0x40                   # G no longer on the curve / no longer xy_powers[0]
    with pytest.raises(rkyv.RkyvFormatError):
        rkyv.decode_combined_sigma(bytes(bad), expect=ex)


def test_sigma_preprocess_round_trip(oracle):
    sec, _, _ = _curve_sections(oracle, SP)
    a = rkyv.encode_sigma_preprocess(sec["xy_powers"], sec["gamma_inv_o_inst"])
    assert len(a) == sec["xy_powers"].size + sec["gamma_inv_o_inst"].size + 16
    got = rkyv.decode_sigma_preprocess(a)
    assert bytes(got["xy_powers"]) == bytes(sec["xy_powers"]) and bytes(got["gamma_inv_o_inst"]) == bytes(sec["gamma_inv_o_inst"])
    with pytest.raises(rkyv.RkyvFormatError):
        rkyv.decode_sigma_preprocess(a[:40])


@pytest.mark.parametrize("order", rkyv.ORDERS)
def test_cpp_reader_and_writer_equal_python(oracle, tmp_path, order):
    assert os.path.exists(DRIVER), "tests/host_cpp/rkyv_driver is not built (run __graft_entry__.build())"
    sec, rows, ex = _curve_sections(oracle, SP)
    payload = crs.build_payload(sec)
    archive = rkyv.encode_combined_sigma(sec, rows, order)
    (tmp_path / "a.rkyv").write_bytes(archive)
    (tmp_path / "p.tkcrs").write_bytes(payload)
    run = lambda *a: subprocess.run([DRIVER] + [str(x) for x in a], capture_output=True, text=True, timeout=120)   # noqa: E731
    # C++ reader on the Python-written archive: same order found, same nine sections
    r = run("decode", tmp_path / "a.rkyv", tmp_path / "out.tkcrs", ex["xy_powers"], ex["gamma_inv_o_inst"], ex["eta_inv_li_o_inter_alpha4_kj"],
            ex["delta_inv_li_o_prv"], ex["rs_y"], "auto")
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip() == order
    assert (tmp_path / "out.tkcrs").read_bytes() == payload
    # C++ writer: the same bytes as the Python writer
    r = run("encode", tmp_path / "p.tkcrs", tmp_path / "b.rkyv", order, len(rows["eta_inv_li_o_inter_alpha4_kj"]), len(rows["delta_inv_li_o_prv"]))
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "b.rkyv").read_bytes() == archive
    # sigma_preprocess.rkyv both ways
    r = run("pre-encode", tmp_path / "p.tkcrs", tmp_path / "pre.rkyv")
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "pre.rkyv").read_bytes() == rkyv.encode_sigma_preprocess(sec["xy_powers"], sec["gamma_inv_o_inst"])
    r = run("pre-decode", tmp_path / "pre.rkyv", tmp_path / "pre.bin")
    assert r.returncode == 0 and r.stdout.split() == [str(ex["xy_powers"]), str(ex["gamma_inv_o_inst"])]
    assert (tmp_path / "pre.bin").read_bytes() == bytes(sec["xy_powers"]) + bytes(sec["gamma_inv_o_inst"])
    # refusals: garbage, and an archive of another circuit
    (tmp_path / "junk").write_bytes(b"not an archive")
    r = run("decode", tmp_path / "junk", tmp_path / "x", "-", "auto")
    assert r.returncode == 1 and "Invalid sigma archive" in r.stderr
    r = run("decode", tmp_path / "a.rkyv", tmp_path / "x", 2 * ex["xy_powers"], ex["gamma_inv_o_inst"], ex["eta_inv_li_o_inter_alpha4_kj"],
            ex["delta_inv_li_o_prv"], ex["rs_y"], "auto")
    assert r.returncode == 1 and "setupParams.json" in r.stderr


def test_cpp_reader_passes_the_decoder_shape_test(tmp_path):
    sections, rows = sample_sigma()
    for order in rkyv.ORDERS:
        (tmp_path / "s.rkyv").write_bytes(rkyv.encode_combined_sigma(sections, rows, order))
        r = subprocess.run([DRIVER, "decode", str(tmp_path / "s.rkyv"), str(tmp_path / "s.tkcrs"), "-", order, "nocheck"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        assert (tmp_path / "s.tkcrs").read_bytes() == crs.build_payload(sections)
