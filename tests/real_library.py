"""Test infrastructure: the reference's REAL subcircuit library as a directory the product can open, and synthesizer documents for it.

The library (packages/frontend/qap-compiler/subcircuits/library/{setupParams.json, subcircuitInfo.json, r1cs/subcircuit0..13.r1cs}) is
committed as DATA in three pieces by tests/golden/make_pins.py (qap: 1, 2, 12; qap_more: 0, 3, 7, 9; qap_rest: 4, 5, 6, 8, 10, 11, 13 + the
library's own subcircuitInfo.json / setupParams.json); assemble() puts them back into the one directory layout the readers expect
(libs/src/iotools/mod.rs:458-469 SubcircuitInfo, :505-650 the .r1cs reader, libs/src/subcircuit_library.rs:17-58 the directory).

The reference ships no synthesizer output, and the witness calculators of the library are prebuilt wasm (never run here), so
make_inputs() manufactures an assignment that satisfies every row of every placed subcircuit WITHOUT them:
  * the five buffers (bufferPubOut / PubIn / BlockIn / EVMIn / PrvIn: rows (in - out)^2 = 0 and in - out = 0) carry random field
    elements, out_i = in_i, constant wire 1;
  * DecToBit (rows (b_i - 1) b_i = 0, in_lo = sum 2^i b_i, in_hi likewise) carries the bit decomposition of two random 128-bit inputs,
    constant wire 1;
  * every other kind (ALU1, ALU2, SubExpBatch, Accumulator, Poseidon, JubjubExpBatch, EdDsaVerify, VerifyMerkleProof) is placed with the
    all-zero assignment, constant wire included: (A.0)(B.0) = 0 = C.0 row by row.  That is a satisfying assignment of the R1CS the
    prover is given (it never reads the constant wire as 1), not an execution trace of the circuit the subcircuit was compiled from.
Copy constraints (permutation.json, libs/src/iotools/mod.rs:408-455) join cells that carry equal values: bufferPrvIn outputs -> the
DecToBit inputs, bufferPubIn outputs -> bufferPrvIn inputs of a second bufferPrvIn placement, and rings of zero cells across the zero
placements."""
import json
import os
import shutil

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PIECES = {"qap": (1, 2, 12), "qap_more": (0, 3, 7, 9), "qap_rest": (4, 5, 6, 8, 10, 11, 13)}
BUFFERS = (0, 1, 2, 3, 4)
DEC_TO_BIT = 7


def r1cs_path(sid):
    for d, ids in PIECES.items():
        if sid in ids:
            return os.path.join(GOLDEN, d, "r1cs", "subcircuit%d.r1cs" % sid)
    raise KeyError(sid)


def infos():
    return json.load(open(os.path.join(GOLDEN, "qap_rest", "subcircuitInfo.json")))


def setup_params():
    return json.load(open(os.path.join(GOLDEN, "qap_rest", "setupParams.json")))


def assemble(dst):
    """-> dst, holding setupParams.json, subcircuitInfo.json and r1cs/subcircuit{0..13}.r1cs exactly as the reference's library does"""
    os.makedirs(os.path.join(dst, "r1cs"), exist_ok=True)
    for sid in range(14):
        shutil.copy(r1cs_path(sid), os.path.join(dst, "r1cs", "subcircuit%d.r1cs" % sid))
    for name in ("subcircuitInfo.json", "setupParams.json"):
        shutil.copy(os.path.join(GOLDEN, "qap_rest", name), os.path.join(dst, name))
    return dst


def default_order():
    """placements 0..3 are the four public buffers in library order (Sigma1::gen builds gamma_inv_o_inst for exactly that:
    libs/src/group_structures/mod.rs:405-440); then every other kind at least twice, mixed"""
    return [0, 1, 2, 3, 4, 7, 5, 6, 8, 9, 10, 11, 12, 13, 4, 7, 13, 12, 11, 10, 9, 8, 6, 5, 7, 12, 9]


def make_inputs(synth_dir, rnd, order=None):
    """writes placementVariables.json / instance.json / permutation.json for the real library into synth_dir.
    -> dict(order, values (per placement: list of ints), permutation)"""
    info = {e["id"]: e for e in infos()}
    sp = setup_params()
    l, l_D = sp["l"], sp["l_D"]
    order = default_order() if order is None else order
    assert order[:4] == [0, 1, 2, 3] and len(order) <= sp["s_max"]
    values = []
    for sid in order:
        e = info[sid]
        w = [0] * e["Nwires"]
        (o0, no), (i0, ni) = e["Out_idx"], e["In_idx"]
        if sid in BUFFERS:
            assert no == ni
            w[0] = 1
            for j in range(ni):
                w[i0 + j] = w[o0 + j] = rnd.randrange(R)
        elif sid == DEC_TO_BIT:
            w[0] = 1
            lo, hi = rnd.randrange(1 << 128), rnd.randrange(1 << 128)
            w[i0], w[i0 + 1] = lo, hi
            for j in range(128):
                w[o0 + j] = (lo >> j) & 1
                w[o0 + 128 + j] = (hi >> j) & 1
        values.append(w)
    # copy constraints: each ring joins cells that must carry one value; the value is then written into every cell of the ring
    cell = lambda p, wire: (info[order[p]]["flattenMap"][wire] - l, p)                        # noqa: E731   (row, col)
    where = {}
    for p, sid in enumerate(order):
        where.setdefault(sid, []).append(p)
    rings = []
    prv, dec = where.get(4, []), where.get(DEC_TO_BIT, [])
    for k, d in enumerate(dec):                     # bufferPrvIn output 2k, 2k+1 of the first bufferPrvIn feed DecToBit placement d
        if not prv:
            break
        e4, e7 = info[4], info[DEC_TO_BIT]
        for j in range(2):
            src_out, src_in = e4["Out_idx"][0] + 2 * k + j, e4["In_idx"][0] + 2 * k + j
            v = values[d][e7["In_idx"][0] + j]
            values[prv[0]][src_out] = values[prv[0]][src_in] = v
            rings.append([(prv[0], src_out), (d, e7["In_idx"][0] + j)])
    if len(prv) > 1:                                # bufferPubIn's interface outputs feed the first inputs of the second bufferPrvIn
        e1, e4 = info[1], info[4]
        for j in range(e1["Out_idx"][1]):
            v = values[1][e1["Out_idx"][0] + j]
            tgt = e4["In_idx"][0] + 100 + j
            values[prv[1]][tgt] = values[prv[1]][e4["Out_idx"][0] + 100 + j] = v
            rings.append([(1, e1["Out_idx"][0] + j), (prv[1], tgt)])
    zeros = [p for p, sid in enumerate(order) if sid not in BUFFERS and sid != DEC_TO_BIT]
    for j in range(1, 4):                           # rings of zero cells: interface wire j of every zero placement that has one
        ring = []
        for p in zeros:
            e = info[order[p]]
            if j < e["Nwires"] and l <= e["flattenMap"][j] < l_D:
                ring.append((p, j))
        if len(ring) > 1:
            rings.append(ring)
    perm = []
    for ring in rings:
        cells = [cell(p, wire) for p, wire in ring]
        assert all(0 <= r < l_D - l for r, _ in cells), "copy constraints join interface wires only"
        assert len({values[p][wire] for p, wire in ring}) == 1
        for a, b in zip(cells, cells[1:] + cells[:1]):
            perm.append({"row": a[0], "col": a[1], "X": b[0], "Y": b[1]})
    hx = lambda v: "0x%x" % v                                                                # noqa: E731
    pv = [{"subcircuitId": sid, "variables": [hx(v) for v in w]} for sid, w in zip(order, values)]
    pub = lambda p, idx: [hx(values[p][idx[0] + j]) for j in range(idx[1])]                  # noqa: E731
    block = pub(2, info[2]["In_idx"])
    block += ["0x0"] * (sp["l_free"] - sp["l_user"] - len(block))         # the unused public slots up to l_free (a power of two) are zero
    instance = {"a_pub_user": pub(0, info[0]["Out_idx"]) + pub(1, info[1]["In_idx"]), "a_pub_block": block,
                "a_pub_function": pub(3, info[3]["In_idx"])}
    os.makedirs(synth_dir, exist_ok=True)
    json.dump(pv, open(os.path.join(synth_dir, "placementVariables.json"), "w"))
    json.dump(instance, open(os.path.join(synth_dir, "instance.json"), "w"))
    json.dump(perm, open(os.path.join(synth_dir, "permutation.json"), "w"))
    return {"order": order, "values": values, "permutation": perm, "instance": instance}
