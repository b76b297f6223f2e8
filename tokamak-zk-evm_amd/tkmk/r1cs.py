"""iden3 `.r1cs` reader and device-side u/v/w generation: work-alikes of R1csBinary / SubcircuitR1CS::from_r1cs_sparse_only
(packages/backend/libs/src/iotools/mod.rs:505-760) and read_R1CS_gen_uvwXY (:1287-1420).

Same validations and error conditions as the reference reader (magic, version, section bounds, field size, wire index
range, trailing bytes, nWires / nConstraints vs subcircuitInfo, n >= Nconsts).  The evaluation itself — a rayon loop over
placements on the host in the reference — runs as one kernel launch per (used subcircuit, matrix) behind
tkmk_r1cs_eval_rows; the transposes and the three inverse bivariate NTTs follow on the device.
"""
import ctypes
import os
import struct
from itertools import repeat as _repeat

import numpy as np

import tkmk
from tkmk.poly import DensePolynomialExt

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


class R1csError(ValueError):
    pass


class R1csBinary:
    def __init__(self, data, constraints_offset, constraints_size, field_size, n_wires, n_constraints):
        self.data, self.constraints_offset, self.constraints_size = data, constraints_offset, constraints_size
        self.field_size, self.n_wires, self.n_constraints = field_size, n_wires, n_constraints

    @classmethod
    def read(cls, path):
        data = open(path, "rb").read()

        def need(off, n):
            if off + n > len(data):
                raise R1csError("unexpected end of R1CS file")

        off = 0
        need(off, 4)
        if data[:4] != b"r1cs":
            raise R1csError("invalid R1CS magic")
        off = 4
        need(off, 8)
        version, section_count = struct.unpack_from("<II", data, off)
        off += 8
        if version != 1:
            raise R1csError("unsupported R1CS version %d" % version)
        header = constraints = None
        for _ in range(section_count):
            need(off, 12)
            stype, ssize = struct.unpack_from("<IQ", data, off)
            off += 12
            if off + ssize > len(data):
                raise R1csError("R1CS section extends past end of file")
            if stype == 1:
                header = (off, ssize)
            elif stype == 2:
                constraints = (off, ssize)
            off += ssize
        if header is None:
            raise R1csError("missing R1CS header section")
        if constraints is None:
            raise R1csError("missing R1CS constraints section")
        cur = header[0]
        need(cur, 4)
        field_size = struct.unpack_from("<I", data, cur)[0]
        cur += 4
        if field_size == 0 or field_size % 8 != 0:
            raise R1csError("invalid R1CS field size %d" % field_size)
        need(cur, field_size + 28)
        cur += field_size                                  # prime (checked by tests against the scalar modulus)
        n_wires, _pub_out, _pub_in, _prv_in = struct.unpack_from("<IIII", data, cur)
        cur += 16
        cur += 8                                           # nLabels
        n_constraints = struct.unpack_from("<I", data, cur)[0]
        cur += 4
        if cur > header[0] + header[1]:
            raise R1csError("R1CS header extends past section end")
        return cls(data, constraints[0], constraints[1], field_size, n_wires, n_constraints)

    def prime(self):
        """the header's field modulus (little-endian)"""
        off = 12
        while True:
            stype, ssize = struct.unpack_from("<IQ", self.data, off)
            off += 12
            if stype == 1:
                return int.from_bytes(self.data[off + 4:off + 4 + self.field_size], "little")
            off += ssize

    def scan_constraints(self):
        """yields (matrix_idx, wire_idx, coeff_bytes, row_idx) in file order (mod.rs:613-650)"""
        data, off = self.data, self.constraints_offset
        end = self.constraints_offset + self.constraints_size
        fs = self.field_size
        for row in range(self.n_constraints):
            for m in range(3):
                if off + 4 > len(data):
                    raise R1csError("unexpected end of R1CS file")
                cnt = struct.unpack_from("<I", data, off)[0]
                off += 4
                for _ in range(cnt):
                    if off + 4 + fs > len(data):
                        raise R1csError("unexpected end of R1CS file")
                    wire = struct.unpack_from("<I", data, off)[0]
                    off += 4
                    if wire >= self.n_wires:
                        raise R1csError("R1CS wire index %d exceeds nWires %d" % (wire, self.n_wires))
                    yield m, wire, data[off:off + fs], row
                    off += fs
        if off != end:
            raise R1csError("R1CS constraints section has %d trailing bytes" % max(0, end - off))


    def csr(self):
        """A, B, C as CSR triples (row_ptr u32, wire u32, coeff bytes), coefficients reduced mod r: the same walk and the same
        errors as scan_constraints, with each linear combination read as one numpy record array"""
        data = self.data
        fs = self.field_size
        if fs != 32:
            return self._csr_slow()
        # pass 1 (sequential by format): the count word of every linear combination -> its start offset and length, by the
        # library's host-only walker (a Python loop over 3 * n_constraints count words costs 3 ms per production subcircuit)
        nlc = 3 * self.n_constraints
        st = np.zeros(nlc, np.uint64)
        ct = np.zeros(nlc, np.uint32)
        used = ctypes.c_size_t(0)
        section = (np.frombuffer(data, np.uint8, self.constraints_size, self.constraints_offset) if self.constraints_size
                   else np.zeros(1, np.uint8))
        rc = tkmk.lib().tkmk_r1cs_index(tkmk._p(section), ctypes.c_size_t(self.constraints_size), ctypes.c_uint32(self.n_constraints),
                                        ctypes.c_uint32(fs), tkmk._p(st.view(np.uint8)), tkmk._p(ct.view(np.uint8)), ctypes.byref(used))
        if rc != 0:
            raise R1csError("unexpected end of R1CS file")
        if used.value != self.constraints_size:
            raise R1csError("R1CS constraints section has %d trailing bytes" % max(0, self.constraints_size - used.value))
        st = st.astype(np.int64) + self.constraints_offset
        starts = [st[m::3] for m in range(3)]
        counts = [ct[m::3] for m in range(3)]
        # pass 2: all (wire, coefficient) records of one matrix in a single gather
        raw = np.frombuffer(data, np.uint8)
        out = []
        for m in range(3):
            cnt = np.asarray(counts[m], np.int64)
            ptr = np.zeros(cnt.size + 1, np.int64)
            np.cumsum(cnt, out=ptr[1:])
            total = int(ptr[-1])
            entry = np.repeat(np.asarray(starts[m], np.int64), cnt) + 36 * (np.arange(total) - np.repeat(ptr[:-1], cnt))
            rec = raw[entry[:, None] + np.arange(36)] if total else np.zeros((0, 36), np.uint8)
            wires = np.ascontiguousarray(rec[:, :4]).view("<u4").reshape(-1)
            if total and int(wires.max()) >= self.n_wires:
                raise R1csError("R1CS wire index %d exceeds nWires %d" % (int(wires.max()), self.n_wires))
            coeff = _reduce_le32(np.ascontiguousarray(rec[:, 4:])).reshape(-1)
            out.append((ptr.astype(np.uint32), wires.astype(np.uint32), coeff))
        return out

    def _csr_slow(self):
        rows = [[[] for _ in range(self.n_constraints)] for _ in range(3)]
        for m, wire, coeff, row in self.scan_constraints():
            rows[m][row].append((wire, int.from_bytes(coeff, "little") % R_MOD))
        csr = []
        for m in range(3):
            ptr, wires, coeffs = [0], [], []
            for r in rows[m]:
                for wire, v in r:
                    wires.append(wire)
                    coeffs.append(v)
                ptr.append(len(wires))
            cb = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in coeffs), np.uint8).copy() if coeffs else np.zeros(0, np.uint8)
            csr.append((np.array(ptr, np.uint32), np.array(wires, np.uint32), cb))
        return csr


def _reduce_le32(rows):
    """(N, 32) little-endian values -> the same reduced mod r (ScalarField::from_bytes_le); only the rare rows >= r are touched"""
    if rows.shape[0] == 0:
        return rows
    limbs = np.ascontiguousarray(rows).view("<u8").reshape(-1, 4)
    r4 = [(R_MOD >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)]
    gt = np.zeros(limbs.shape[0], bool)
    eq = np.ones(limbs.shape[0], bool)
    for k in (3, 2, 1, 0):
        gt |= eq & (limbs[:, k] > r4[k])
        eq &= limbs[:, k] == r4[k]
    for i in np.nonzero(gt | eq)[0]:
        rows[i] = np.frombuffer((int.from_bytes(rows[i].tobytes(), "little") % R_MOD).to_bytes(32, "little"), np.uint8)
    return rows


def hex_list_to_fr32(hexes):
    """[HexString, ...] -> (N, 32) uint8 little-endian Fr records, each ScalarField::from_hex (mod.rs:126-146) of its entry;
    one bytes.fromhex over the whole list instead of one big-int conversion per element"""
    try:
        # right-align every entry in 66 characters with C-level map/join; the 'x' of a "0x" prefix becomes one more zero digit
        txt = "".join(map(str.rjust, hexes, _repeat(66), _repeat("0"))).replace("x", "0").replace("X", "0")
        if len(txt) != 66 * len(hexes):
            raise ValueError
        be = np.frombuffer(bytes.fromhex(txt), np.uint8).reshape(-1, 33)
        if be[:, 0].any():                                # more than 256 bits of digits
            raise ValueError
        be = be[:, 1:]
    except (ValueError, TypeError):                      # an entry longer than 256 bits (or odd text): element-wise route
        return np.frombuffer(b"".join(hex_to_fr(h).to_bytes(32, "little") for h in hexes), np.uint8).reshape(-1, 32).copy()
    return _reduce_le32(np.ascontiguousarray(be[:, ::-1]))


class SubcircuitR1CS:
    """sparse rows of A, B, C as CSR (row_ptr, wire, coeff) — from_r1cs_sparse_only (mod.rs:685-760)"""

    def __init__(self, n_wires, n_constraints, csr):
        self.n_wires, self.n_constraints, self.csr = n_wires, n_constraints, csr
        self._dev = None

    @classmethod
    def from_r1cs_sparse_only(cls, path, setup_params, info):
        b = R1csBinary.read(path)
        if b.n_wires != info["Nwires"]:
            raise R1csError("R1CS nWires mismatch for subcircuit %d: binary=%d, info=%d" % (info["id"], b.n_wires, info["Nwires"]))
        if b.n_constraints != info["Nconsts"]:
            raise R1csError("R1CS nConstraints mismatch for subcircuit %d: binary=%d, info=%d" % (info["id"], b.n_constraints, info["Nconsts"]))
        if setup_params["n"] < info["Nconsts"]:
            raise R1csError("n is smaller than the actual number of constraints.")
        return cls(b.n_wires, b.n_constraints, b.csr())

    def active_wires(self, m):
        return sorted(set(int(w) for w in self.csr[m][1]))

    def device(self):
        if self._dev is None:
            self._dev = []
            for ptr, wires, coeffs in self.csr:
                d = [tkmk.DeviceBuffer.from_host(ptr.view(np.uint8)),
                     tkmk.DeviceBuffer.from_host(wires.view(np.uint8)) if wires.size else tkmk.DeviceBuffer(16),
                     tkmk.DeviceBuffer.from_host(coeffs) if coeffs.size else tkmk.DeviceBuffer(32)]
                self._dev.append(d)
        return self._dev


def hex_to_fr(h):
    """ScalarField::from_hex on a HexString (mod.rs:126-146): optional 0x, any length, big-endian digits"""
    h = h[2:] if h.startswith(("0x", "0X")) else h
    return (int(h, 16) if h else 0) % R_MOD


class PlacementValues:
    """the `variables` of every placement parsed once into (Nwires, 32) little-endian Fr records; gen_bXY, the u/v/w
    evaluation and the three binding commitments all read the same hex strings (the reference parses them in each:
    libs/src/polynomial_structures/mod.rs:150, libs/src/iotools/mod.rs:1373, libs/src/group_structures/mod.rs:213,285)"""

    def __init__(self, placement_variables):
        self.rows = [hex_list_to_fr32(pl["variables"]) for pl in placement_variables]

    def __getitem__(self, i):
        return self.rows[i]


def read_R1CS_gen_uvwXY(qap_path, placement_variables, subcircuit_infos, setup_params, values=None):
    """placement_variables: list of {"subcircuitId": id, "variables": [hex, ...]} (synthesizer output);
    returns (uXY, vXY, wXY) as device-resident DensePolynomialExt of size n x s_max"""
    values = PlacementValues(placement_variables) if values is None else values
    n, s_max = setup_params["n"], setup_params["s_max"]
    if len(placement_variables) > s_max:
        raise ValueError("placement_variables length exceeds s_max.")
    infos = {e["id"]: e for e in subcircuit_infos}
    by_id = {}
    for i, pl in enumerate(placement_variables):
        sid = pl["subcircuitId"]
        if sid not in infos:
            raise ValueError("Invalid subcircuit id in placement_variables.")
        by_id.setdefault(sid, []).append(i)
    evals = [tkmk.DeviceBuffer.from_host(np.zeros(32 * s_max * n, np.uint8)) for _ in range(3)]   # s_max x n, zero filled
    lib = tkmk.lib()
    for sid, slots in by_id.items():
        r1cs = SubcircuitR1CS.from_r1cs_sparse_only(os.path.join(qap_path, "r1cs", "subcircuit%d.r1cs" % sid), setup_params, infos[sid])
        var = np.concatenate([values[i] for i in slots]).reshape(-1)
        if var.size != 32 * len(slots) * r1cs.n_wires:
            raise ValueError("placement variable count does not match nWires of subcircuit %d" % sid)
        d_var = tkmk.DeviceBuffer.from_host(var)
        d_slot = tkmk.DeviceBuffer.from_host(np.array(slots, np.uint32).view(np.uint8))
        for m, (d_ptr, d_wire, d_coeff) in enumerate(r1cs.device()):
            nnz = int(r1cs.csr[m][1].size)
            tkmk._check(lib.tkmk_r1cs_eval_rows(tkmk._p(d_ptr), tkmk._p(d_wire), tkmk._p(d_coeff), r1cs.n_constraints, nnz, tkmk._p(d_var),
                                                r1cs.n_wires, len(slots), tkmk._p(d_slot), n, tkmk._p(evals[m]), None), "tkmk_r1cs_eval_rows")
    out = []
    for e in evals:
        t = tkmk.transpose(e, s_max, n)                     # transpose_inplace(&mut u_eval, s_max, n): -> n x s_max
        out.append(DensePolynomialExt.from_rou_evals(t, n, s_max))
    return tuple(out)
