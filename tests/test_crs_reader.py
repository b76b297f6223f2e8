"""not-gpu tier: reader of the reference's flat CRS payload "TKCRS001" (tkmk/crs.py).  Golden shape: the sample sigma of the
reference's own decoder test (packages/backend-wasm/tools/rkyv-decoder-wasm/src/lib.rs:209-293): points are the byte
patterns g1(seed): x[i] = seed + i, y[i] = seed + 48 + i (g2: 96-byte coordinates), section sizes 6,2,1,3,1,3,1,2 G1
points and 10 G2 points, first / last entries of sections 0 and 8 as asserted there."""
import os
import struct
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tokamak-zk-evm_amd"))
from tkmk import crs  # noqa: E402


def g1(seed):
    return bytes((seed + i) & 255 for i in range(48)) + bytes((seed + 48 + i) & 255 for i in range(48))


def g2(seed):
    return bytes((seed + i) & 255 for i in range(96)) + bytes((seed + 96 + i) & 255 for i in range(96))


def sample_payload():
    # sample_sigma() of the reference test, laid out by encode_combined_sigma_payload (lib.rs:118-160)
    sections = [
        g1(1) + g1(5) + g1(6) + g1(7) + g1(8) + g1(29),            # G, x, y, delta, eta, lagrange_KL
        g1(3) + g1(4),                                             # xy_powers
        g1(9),                                                     # gamma_inv_o_inst
        g1(10) + g1(11) + g1(12),                                  # nested rows flattened
        g1(13),
        g1(14) + g1(15) + g1(16),
        g1(17),
        g1(18) + g1(19),
        b"".join(g2(s) for s in (2, 20, 21, 22, 23, 24, 25, 26, 27, 28)),   # H, alpha .. y
    ]
    head = b"TKCRS001" + struct.pack("<I", 9) + b"".join(struct.pack("<I", len(s)) for s in sections)
    return head + b"".join(sections), sections


def test_reference_decoder_sample_shape():
    payload, raw = sample_payload()
    sec = crs.parse_payload(payload)
    sizes = [sec[n].size for n in crs.SECTION_NAMES]
    assert sizes == [6 * 96, 2 * 96, 96, 3 * 96, 96, 3 * 96, 96, 2 * 96, 10 * 192]          # lib.rs:217-225
    assert bytes(crs.single_g1(sec, "G")) == g1(1) and bytes(crs.single_g1(sec, "x")) == g1(5)   # lib.rs:227-229
    assert bytes(crs.single_g1(sec, "lagrange_KL")) == g1(29)
    assert bytes(sec["g2"][:192]) == g2(2) and bytes(sec["g2"][9 * 192:]) == g2(28)          # lib.rs:230-231
    assert bytes(sec["eta_inv_li_o_inter_alpha4_kj"]) == g1(10) + g1(11) + g1(12)
    assert crs.build_payload({n: r for n, r in zip(crs.SECTION_NAMES, raw)}) == payload


def test_malformed_payloads_are_rejected(tmp_path):
    payload, _ = sample_payload()
    for bad in (b"not an archive", b"TKCRS002" + payload[8:], payload[:-1], payload + b"\0", payload[:20],
                payload[:8] + struct.pack("<I", 8) + payload[12:]):
        with pytest.raises(crs.CrsFormatError):
            crs.parse_payload(bad)
    lens = list(struct.unpack("<9I", payload[12:48]))
    lens[1] -= 1
    lens[2] += 1                                                    # sums still match, but sections are not whole points
    with pytest.raises(crs.CrsFormatError):
        crs.parse_payload(payload[:12] + struct.pack("<9I", *lens) + payload[48:])
    p = tmp_path / "crs.bin"
    p.write_bytes(payload)
    assert bytes(crs.read_payload(str(p))["gamma_inv_o_inst"]) == g1(9)                      # mmap route


def test_shape_check_against_setup_params():
    sp = {"n": 4, "l": 3, "l_D": 7, "m_D": 12, "s_max": 2}          # m_i = 4, private wires = 5
    pts = {"xy_powers": 8 * 4, "gamma_inv_o_inst": 3, "eta_inv_li_o_inter_alpha4_kj": 8, "delta_inv_li_o_prv": 10,
           "delta_inv_alphak_xh_tx": 9, "delta_inv_alpha4_xj_tx": 2, "delta_inv_alphak_yi_ty": 12}
    sections = {"g1": bytes(6 * 96), "g2": bytes(10 * 192)}
    sections.update({k: bytes(96 * v) for k, v in pts.items()})
    sec = crs.parse_payload(crs.build_payload(sections))
    assert crs.check_shapes(sec, sp) == 4
    with pytest.raises(crs.CrsFormatError):
        crs.check_shapes(sec, dict(sp, s_max=4))
