"""Reader / writer for the reference's CRS archives, Python mirror of host/tkmk_rkyv.hpp (see its header for the restated
rkyv 0.7 format and for why the field order inside the archived structs is resolved by validation, not assumed):

  combined_sigma.rkyv    rkyv::to_bytes::<_, 256>(&SigmaRkyv)            packages/backend/libs/src/iotools/mod.rs:280-285, types :1701-1783
  sigma_preprocess.rkyv  rkyv::to_bytes::<_, 256>(&SigmaPreprocessRkyv)  :287-294

decode_combined_sigma -> the nine sections of the reference's own decoder (backend-wasm/tools/rkyv-decoder-wasm/src/lib.rs:118-140),
as a dict with tkmk.crs.SECTION_NAMES keys; encode_combined_sigma is its inverse for a chosen field order (test fixtures and the
`setup` command's output).  Pure numpy / struct: no device is touched."""
import struct

import numpy as np

from tkmk.crs import G1_BYTES, G2_BYTES, SECTION_NAMES

ORDERS = ("rustc_size_groups", "rustc_align_only", "declared")
VEC = 8
P_MOD = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab


class RkyvFormatError(ValueError):
    pass


def _tz(v):
    return (v & -v).bit_length() - 1


def layout(fields, order):
    """fields: [(size, align)] in declaration order -> ([offset], struct size, struct align)"""
    idx = list(range(len(fields)))
    if order == "rustc_size_groups":        # current rustc: group by log2(max(align, size)), larger groups first, stable
        idx.sort(key=lambda i: -_tz(max(fields[i][1], fields[i][0])))
    elif order == "rustc_align_only":       # older rustc: by alignment, larger first, stable
        idx.sort(key=lambda i: -fields[i][1])
    elif order != "declared":
        raise ValueError(order)
    offs, off, align = [0] * len(fields), 0, 1
    for i in idx:
        size, al = fields[i]
        off = (off + al - 1) // al * al
        offs[i] = off
        off += size
        align = max(align, al)
    return offs, (off + align - 1) // align * align, align


S1_FIELDS = ("xy_powers", "x", "y", "delta", "eta", "gamma_inv_o_inst", "eta_inv_li_o_inter_alpha4_kj", "delta_inv_li_o_prv",
             "delta_inv_alphak_xh_tx", "delta_inv_alpha4_xj_tx", "delta_inv_alphak_yi_ty")            # libs/src/iotools/mod.rs:1727-1741
ROOT_FIELDS = ("G", "H", "sigma_1", "sigma_2", "lagrange_KL")                                           # :1716-1724
_S1_VECS = {"xy_powers", "gamma_inv_o_inst", "eta_inv_li_o_inter_alpha4_kj", "delta_inv_li_o_prv", "delta_inv_alphak_xh_tx",
            "delta_inv_alpha4_xj_tx", "delta_inv_alphak_yi_ty"}
NESTED = ("eta_inv_li_o_inter_alpha4_kj", "delta_inv_li_o_prv", "delta_inv_alphak_xh_tx", "delta_inv_alphak_yi_ty")


def sigma_layout(order):
    s1_offs, s1_size, s1_align = layout([(VEC, 4) if f in _S1_VECS else (G1_BYTES, 1) for f in S1_FIELDS], order)
    root_offs, root_size, root_align = layout([(G1_BYTES, 1), (G2_BYTES, 1), (s1_size, s1_align), (9 * G2_BYTES, 1), (G1_BYTES, 1)], order)
    return dict(zip(S1_FIELDS, s1_offs)), dict(zip(ROOT_FIELDS, root_offs)), root_size, root_align


def _vec(a, at, elem, what):
    if at + 8 > a.size:
        raise RkyvFormatError("vector header of %s outside the archive" % what)
    rel, ln = struct.unpack("<iI", bytes(a[at:at + 8]))
    pos = at + rel
    if pos < 0 or pos + ln * elem > a.size:
        raise RkyvFormatError("%s points outside the archive" % what)
    return pos, ln


def _nested(a, at, what):
    pos, rows = _vec(a, at, VEC, what)
    if pos % 4:
        raise RkyvFormatError("misaligned row headers of " + what)
    parts, lens = [], []
    for r in range(rows):
        p, ln = _vec(a, pos + VEC * r, G1_BYTES, what)
        parts.append((p, ln))
        lens.append(ln)
    if parts and all(parts[k][0] + parts[k][1] * G1_BYTES == parts[k + 1][0] for k in range(len(parts) - 1)):
        return a[parts[0][0]:parts[0][0] + sum(lens) * G1_BYTES], lens          # adjacent rows: one view, no copy
    return (np.concatenate([a[p:p + ln * G1_BYTES] for p, ln in parts]) if parts else a[:0]), lens


def g1_on_curve_or_infinity(rec):
    x, y = int.from_bytes(bytes(rec[:48]), "little"), int.from_bytes(bytes(rec[48:96]), "little")
    if x == 0 and y == 0:
        return True
    return x < P_MOD and y < P_MOD and (y * y - x * x * x - 4) % P_MOD == 0


def _try(a, order, expect, check_points=True):
    s1f, rf, root_size, root_align = sigma_layout(order)
    if a.size < root_size:
        raise RkyvFormatError("archive shorter than its root object")
    root = a.size - root_size
    if root % root_align:
        raise RkyvFormatError("misaligned root object")
    s1, s2 = root + rf["sigma_1"], root + rf["sigma_2"]
    xy_pos, xy_len = _vec(a, s1 + s1f["xy_powers"], G1_BYTES, "xy_powers")
    gm_pos, gm_len = _vec(a, s1 + s1f["gamma_inv_o_inst"], G1_BYTES, "gamma_inv_o_inst")
    xj_pos, xj_len = _vec(a, s1 + s1f["delta_inv_alpha4_xj_tx"], G1_BYTES, "delta_inv_alpha4_xj_tx")
    nested = {f: _nested(a, s1 + s1f[f], f) for f in NESTED}
    got = {"xy_powers": xy_len, "gamma_inv_o_inst": gm_len, "eta_inv_li_o_inter_alpha4_kj": sum(nested["eta_inv_li_o_inter_alpha4_kj"][1]),
           "delta_inv_li_o_prv": sum(nested["delta_inv_li_o_prv"][1])}
    for k, v in (expect or {}).items():
        if k in got and got[k] != v:
            raise RkyvFormatError("%s holds %d points, SetupParams imply %d" % (k, got[k], v))
    if xy_pos + xy_len * G1_BYTES > root or gm_pos < xy_pos + xy_len * G1_BYTES:
        raise RkyvFormatError("blocks out of serialization order")
    g1 = lambda at: a[at:at + G1_BYTES]                                               # noqa: E731
    singles = [g1(root + rf["G"]), g1(s1 + s1f["x"]), g1(s1 + s1f["y"]), g1(s1 + s1f["delta"]), g1(s1 + s1f["eta"]), g1(root + rf["lagrange_KL"])]
    xy = a[xy_pos:xy_pos + xy_len * G1_BYTES]
    if check_points:
        if not all(g1_on_curve_or_infinity(p) for p in singles):
            raise RkyvFormatError("a single G1 point is not on the curve")
        if xy_len >= 1 and not (xy[:96] == singles[0]).all():
            raise RkyvFormatError("xy_powers[0] != G")
        if xy_len >= 2 and not (xy[96:192] == singles[2]).all():
            raise RkyvFormatError("xy_powers[1] != sigma_1.y")
        rs_y = (expect or {}).get("rs_y")
        if rs_y and xy_len > rs_y and not (xy[96 * rs_y:96 * (rs_y + 1)] == singles[1]).all():
            raise RkyvFormatError("xy_powers[rs_y] != sigma_1.x")
    sec = {"g1": np.concatenate(singles), "xy_powers": xy, "gamma_inv_o_inst": a[gm_pos:gm_pos + gm_len * G1_BYTES],
           "delta_inv_alpha4_xj_tx": a[xj_pos:xj_pos + xj_len * G1_BYTES],
           "g2": np.concatenate([a[root + rf["H"]:root + rf["H"] + G2_BYTES], a[s2:s2 + 9 * G2_BYTES]])}
    rows = {}
    for f in NESTED:
        sec[f], rows[f] = nested[f]
    return sec, rows


def expect_for(setup_params):
    n, l, l_d, m_d, s_max = (setup_params[k] for k in ("n", "l", "l_D", "m_D", "s_max"))
    m_i = l_d - l
    return {"xy_powers": max(2 * n, 2 * m_i) * 2 * s_max, "gamma_inv_o_inst": l, "eta_inv_li_o_inter_alpha4_kj": m_i * s_max,
            "delta_inv_li_o_prv": (m_d - l_d) * s_max, "rs_y": 2 * s_max}


def decode_combined_sigma(buf, expect=None, want_details=False, orders=ORDERS, check_points=True):
    """buf: bytes-like of the archive -> dict of the nine decoder sections (numpy uint8 views where the archive holds them
    contiguously).  expect = expect_for(setup_params) adds the circuit's size checks.  Tries the known field orders and returns
    the first under which the archive validates.  check_points=False keeps only the structural checks (the shape test of the
    reference's decoder uses byte patterns, not curve points); then name the one order to read with in `orders`."""
    a = np.frombuffer(buf, np.uint8) if not isinstance(buf, np.ndarray) else buf
    why = []
    for order in orders:
        try:
            sec, rows = _try(a, order, expect, check_points)
        except RkyvFormatError as e:
            why.append("%s: %s" % (order, e))
            continue
        return (sec, rows, order) if want_details else sec
    raise RkyvFormatError("Invalid sigma archive: combined_sigma.rkyv validates under none of the known field orders (%s)" % "; ".join(why))


def decode_sigma_preprocess(buf):
    a = np.frombuffer(buf, np.uint8) if not isinstance(buf, np.ndarray) else buf
    if a.size < 16 or (a.size - 16) % 4:
        raise RkyvFormatError("Invalid sigma_preprocess archive")
    xy_pos, xy_len = _vec(a, a.size - 16, G1_BYTES, "xy_powers")
    gm_pos, gm_len = _vec(a, a.size - 8, G1_BYTES, "gamma_inv_o_inst")
    if xy_pos + xy_len * G1_BYTES > a.size - 16 or gm_pos < xy_pos + xy_len * G1_BYTES:
        raise RkyvFormatError("Invalid sigma_preprocess archive: blocks out of serialization order")
    return {"xy_powers": a[xy_pos:xy_pos + xy_len * G1_BYTES], "gamma_inv_o_inst": a[gm_pos:gm_pos + gm_len * G1_BYTES]}


# ---- writer: the bytes rkyv::to_bytes produces for these types ----
def _b(x):
    return bytes(np.ascontiguousarray(np.asarray(x, np.uint8))) if not isinstance(x, (bytes, bytearray)) else bytes(x)


def encode_combined_sigma(sections, rows, order="rustc_size_groups"):
    """sections: dict with tkmk.crs.SECTION_NAMES keys; rows: {nested table name: [points per row]} (e.g. eta: [s_max] * m_I).
    Serialization order = declaration order of SigmaRkyv / Sigma1Rkyv; nested vectors write their rows, then (aligned to 4)
    the row headers; the root object comes last."""
    s1f, rf, root_size, root_align = sigma_layout(order)
    out = bytearray()
    at = {}

    def block(data):
        p = len(out)
        out.extend(data)
        return p

    def nested(name):
        data = _b(sections[name])
        assert sum(rows[name]) * G1_BYTES == len(data), name
        starts, off = [], 0
        for r in rows[name]:
            starts.append(block(data[off:off + r * G1_BYTES]))
            off += r * G1_BYTES
        while len(out) % 4:
            out.append(0)
        heads = len(out)
        for p, r in zip(starts, rows[name]):
            out.extend(struct.pack("<iI", p - len(out), r))
        return heads, len(rows[name])

    xy = _b(sections["xy_powers"])
    at["xy_powers"] = (block(xy), len(xy) // G1_BYTES)
    gm = _b(sections["gamma_inv_o_inst"])
    at["gamma_inv_o_inst"] = (block(gm), len(gm) // G1_BYTES)
    at["eta_inv_li_o_inter_alpha4_kj"] = nested("eta_inv_li_o_inter_alpha4_kj")
    at["delta_inv_li_o_prv"] = nested("delta_inv_li_o_prv")
    at["delta_inv_alphak_xh_tx"] = nested("delta_inv_alphak_xh_tx")
    xj = _b(sections["delta_inv_alpha4_xj_tx"])
    at["delta_inv_alpha4_xj_tx"] = (block(xj), len(xj) // G1_BYTES)
    at["delta_inv_alphak_yi_ty"] = nested("delta_inv_alphak_yi_ty")
    while len(out) % root_align:
        out.append(0)
    root = len(out)
    out.extend(bytes(root_size))
    g1s, g2s = _b(sections["g1"]), _b(sections["g2"])
    s1, s2 = root + rf["sigma_1"], root + rf["sigma_2"]
    put = lambda pos, data: out.__setitem__(slice(pos, pos + len(data)), data)          # noqa: E731
    put(root + rf["G"], g1s[0:96])
    for k, f in enumerate(("x", "y", "delta", "eta")):
        put(s1 + s1f[f], g1s[96 * (k + 1):96 * (k + 2)])
    put(root + rf["lagrange_KL"], g1s[480:576])
    put(root + rf["H"], g2s[:G2_BYTES])
    put(s2, g2s[G2_BYTES:])
    for f, (pos, ln) in at.items():
        fp = s1 + s1f[f]
        put(fp, struct.pack("<iI", pos - fp, ln))
    return bytes(out)


def encode_sigma_preprocess(xy_powers, gamma_inv_o_inst):
    xy, gm = _b(xy_powers), _b(gamma_inv_o_inst)
    out = bytearray(xy + gm)
    while len(out) % 4:
        out.append(0)
    out += struct.pack("<iI", 0 - len(out), len(xy) // G1_BYTES)
    out += struct.pack("<iI", len(xy) - len(out), len(gm) // G1_BYTES)
    return bytes(out)


def rows_for(setup_params):
    """row structure of the nested tables as Sigma1::gen builds them (libs/src/group_structures/mod.rs:345-551)"""
    m_i = setup_params["l_D"] - setup_params["l"]
    s_max = setup_params["s_max"]
    return {"eta_inv_li_o_inter_alpha4_kj": [s_max] * m_i, "delta_inv_li_o_prv": [s_max] * (setup_params["m_D"] - setup_params["l_D"]),
            "delta_inv_alphak_xh_tx": [3, 3, 3], "delta_inv_alphak_yi_ty": [3, 3, 3, 3]}
