"""CRS container reader: the flat section payload the reference itself derives from combined_sigma.rkyv for its second
prover (packages/backend-wasm/tools/rkyv-decoder-wasm/src/lib.rs:118-204): magic "TKCRS001", u32 section count (9),
nine u32 little-endian section lengths, then the sections

  0 six single G1 points   G, sigma_1.x, sigma_1.y, sigma_1.delta, sigma_1.eta, lagrange_KL
  1 sigma_1.xy_powers                       [i * 2 s_max + j] = [tau_x^i tau_y^j] G
  2 sigma_1.gamma_inv_o_inst                l points
  3 sigma_1.eta_inv_li_o_inter_alpha4_kj    nested rows, flattened in order
  4 sigma_1.delta_inv_li_o_prv              nested rows, flattened in order
  5 sigma_1.delta_inv_alphak_xh_tx          3 x 3
  6 sigma_1.delta_inv_alpha4_xj_tx          2
  7 sigma_1.delta_inv_alphak_yi_ty          4 x 3
  8 ten G2 points          H, alpha, alpha2, alpha3, alpha4, gamma, delta, eta, x, y

G1 = 96 bytes (48-byte little-endian x, y; (0,0) = infinity: libs/src/iotools/mod.rs:1701-1706,1785-1816) — exactly the record
the MSM consumes, so a section is uploaded as is (or used straight from an mmap).  The rkyv archive itself (relative-pointer
layout of rkyv 0.7) is not parsed here: no archive ships with the reference to check a parser against, while this payload
has a byte-exact specification and a shape test in the reference (lib.rs:209-232), restated in tests/test_crs_reader.py."""
import mmap
import struct

import numpy as np

MAGIC = b"TKCRS001"
SECTION_COUNT = 9
G1_BYTES, G2_BYTES = 96, 192
G1_SINGLES = ("G", "x", "y", "delta", "eta", "lagrange_KL")
G2_POINTS = ("H", "alpha", "alpha2", "alpha3", "alpha4", "gamma", "delta", "eta", "x", "y")
SMALL_TABLES = ("delta_inv_alphak_xh_tx", "delta_inv_alpha4_xj_tx", "delta_inv_alphak_yi_ty")
SECTION_NAMES = ("g1", "xy_powers", "gamma_inv_o_inst", "eta_inv_li_o_inter_alpha4_kj", "delta_inv_li_o_prv",
                 "delta_inv_alphak_xh_tx", "delta_inv_alpha4_xj_tx", "delta_inv_alphak_yi_ty", "g2")


class CrsFormatError(ValueError):
    pass


def parse_payload(buf):
    """buf: bytes-like (bytes, mmap, numpy uint8).  -> dict section name -> numpy uint8 view (no copy)"""
    a = np.frombuffer(buf, np.uint8) if not isinstance(buf, np.ndarray) else buf
    if a.size < 12 or bytes(a[:8]) != MAGIC:
        raise CrsFormatError("not a TKCRS001 payload")
    count = struct.unpack("<I", bytes(a[8:12]))[0]
    if count != SECTION_COUNT:
        raise CrsFormatError("expected %d sections, found %d" % (SECTION_COUNT, count))
    head = 12 + 4 * count
    if a.size < head:
        raise CrsFormatError("truncated section table")
    lens = struct.unpack("<%dI" % count, bytes(a[12:head]))
    if head + sum(lens) != a.size:
        raise CrsFormatError("section lengths (%d) do not add up to the payload size (%d)" % (head + sum(lens), a.size))
    out, off = {}, head
    for name, n in zip(SECTION_NAMES, lens):
        unit = G2_BYTES if name == "g2" else G1_BYTES
        if n % unit:
            raise CrsFormatError("section %s is not a whole number of %d-byte points" % (name, unit))
        out[name] = a[off:off + n]
        off += n
    if out["g1"].size != len(G1_SINGLES) * G1_BYTES or out["g2"].size != len(G2_POINTS) * G2_BYTES:
        raise CrsFormatError("unexpected size of the single-point sections")
    return out


def read_payload(path):
    """memory-maps the file (xy_powers alone is 384 MiB at the production shape) and parses it"""
    f = open(path, "rb")
    m = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
    return parse_payload(m)


def build_payload(sections):
    """inverse of parse_payload (used by tests and by tools that stage a generated CRS): sections = dict name -> bytes-like"""
    parts = [bytes(np.asarray(sections[n], np.uint8)) if not isinstance(sections[n], (bytes, bytearray)) else bytes(sections[n])
             for n in SECTION_NAMES]
    return MAGIC + struct.pack("<I", SECTION_COUNT) + struct.pack("<%dI" % SECTION_COUNT, *[len(p) for p in parts]) + b"".join(parts)


def single_g1(sections, name):
    i = G1_SINGLES.index(name)
    return sections["g1"][G1_BYTES * i:G1_BYTES * (i + 1)]


def check_shapes(sections, setup_params):
    """section sizes implied by SetupParams (Sigma1::gen, libs/src/group_structures/mod.rs:345-551); raises on mismatch"""
    n, l, l_d, m_d, s_max = (setup_params[k] for k in ("n", "l", "l_D", "m_D", "s_max"))
    m_i = l_d - l
    want = {"xy_powers": max(2 * n, 2 * m_i) * 2 * s_max, "gamma_inv_o_inst": l,
            "eta_inv_li_o_inter_alpha4_kj": m_i * s_max, "delta_inv_li_o_prv": (m_d - (l + m_i)) * s_max,
            "delta_inv_alphak_xh_tx": 9, "delta_inv_alpha4_xj_tx": 2, "delta_inv_alphak_yi_ty": 12}
    for name, pts in want.items():
        if sections[name].size != pts * G1_BYTES:
            raise CrsFormatError("%s holds %d points, SetupParams imply %d" % (name, sections[name].size // G1_BYTES, pts))
    return m_i


def load_sigma1(sections, setup_params):
    """-> (Sigma1 with xy_powers resident in HBM, dict of the other G1 tables as DeviceBuffers)"""
    import tkmk
    from tkmk.sigma import Sigma1
    m_i = check_shapes(sections, setup_params)
    n, s_max = setup_params["n"], setup_params["s_max"]
    up = lambda name: tkmk.DeviceBuffer.from_host(np.ascontiguousarray(sections[name]))   # noqa: E731
    sigma1 = Sigma1(up("xy_powers"), max(2 * n, 2 * m_i), 2 * s_max)
    # the three 2..12-point tables are only ever read on the host (O_prv's blinding terms, prove/src/lib.rs:1146-1160)
    tables = {name: (np.array(sections[name]) if name in SMALL_TABLES else up(name)) for name in SECTION_NAMES[2:8] if sections[name].size}
    return sigma1, tables
