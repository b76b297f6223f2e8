"""Test / bench infrastructure (not product code): a synthetic subcircuit library + synthesizer output with the file formats `prove` reads.

The reference ships no witness / proof fixtures (and its witness calculators are prebuilt wasm, which is never run here),
so satisfying inputs are manufactured: random arithmetic subcircuits written as iden3 `.r1cs` v1 files (the format of
packages/frontend/qap-compiler/subcircuits/library/r1cs/*.r1cs, reader packages/backend/libs/src/iotools/mod.rs:505-650),
witnesses by forward evaluation, placements chained through copy constraints (permutation.json entries {row, col, X, Y}:
libs/src/iotools/mod.rs:408-455), setupParams.json / subcircuitInfo.json / placementVariables.json / instance.json with
the field names of libs/src/iotools/mod.rs:166-177,366-372,399-406,458-469.

Layout of one subcircuit's local wires: [0] = the constant 1, then outputs, then inputs, then private wires
(Out_idx = [1, n_out], In_idx = [1 + n_out, n_in]).  Global wire ranges: [0, l) public, [l, l_D) interface, [l_D, m_D) private.
Subcircuits 0..3 are the four public buffers of the library (bufferPubOut / PubIn / BlockIn / EVMIn) and occupy placements
0..3, as the CRS assumes (gamma_inv_o_inst is built with L_0..L_3(tau_y) for them: libs/src/group_structures/mod.rs:405-440).
"""
import json
import os
import struct

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def write_r1cs(path, n_wires, rows):
    """rows: list of (A, B, C), each a list of (wire, coeff int)"""
    cons = bytearray()
    for abc in rows:
        for lc in abc:
            cons += struct.pack("<I", len(lc))
            for wire, coeff in lc:
                cons += struct.pack("<I", wire) + (coeff % R).to_bytes(32, "little")
    header = struct.pack("<I", 32) + R.to_bytes(32, "little") + struct.pack("<IIIIQI", n_wires, 0, 0, 0, n_wires, len(rows))
    labels = b"".join(struct.pack("<Q", i) for i in range(n_wires))
    with open(path, "wb") as f:
        f.write(b"r1cs" + struct.pack("<II", 1, 3))
        for stype, body in ((1, header), (2, bytes(cons)), (3, labels)):
            f.write(struct.pack("<IQ", stype, len(body)) + body)


class Subcircuit:
    def __init__(self, sid, name, n_out, n_in, n_prv):
        self.id, self.name, self.n_out, self.n_in, self.n_prv = sid, name, n_out, n_in, n_prv
        self.n_wires = 1 + n_out + n_in + n_prv
        self.rows = []          # (A, B, C)
        self.order = []         # (target wire, A, B, lin): target = (A.w)(B.w) + lin.w, in evaluation order
        self.flatten_map = None

    def outs(self):
        return range(1, 1 + self.n_out)

    def ins(self):
        return range(1 + self.n_out, 1 + self.n_out + self.n_in)

    def prvs(self):
        return range(1 + self.n_out + self.n_in, self.n_wires)

    def define(self, target, A, B, lin):
        self.order.append((target, A, B, lin))
        self.rows.append((A, B, [(target, 1)] + [(w, -c) for w, c in lin]))

    def witness(self, inputs):
        w = [0] * self.n_wires
        w[0] = 1
        for wire, v in zip(self.ins(), inputs):
            w[wire] = v % R
        dot = lambda lc: sum(c * w[k] for k, c in lc) % R                             # noqa: E731
        for target, A, B, lin in self.order:
            w[target] = (dot(A) * dot(B) + dot(lin)) % R
        return w


BUFFER_NAMES = ("bufferPubOut", "bufferPubIn", "bufferBlockIn", "bufferEVMIn")


def buffer(sid, name, k):
    """the shape of the library's four public buffers: out_i * 1 = in_i.  bufferPubOut's OUTPUTS are public wires, the other
    three have public INPUTS (libs/src/group_structures/mod.rs:184-229 picks Out_idx / In_idx accordingly)"""
    s = Subcircuit(sid, name, k, k, 0)
    for o, i in zip(s.outs(), s.ins()):
        s.define(o, [(i, 1)], [(0, 1)], [])
    return s


def random_gates(sid, rnd, n_out, n_in, n_prv, fan=3, bit_fraction=0.0):
    """bit_fraction of the private wires are constant bits / small values (w = c * 1 with c in {0, 1, small}): real witnesses are
    dominated by 0/1 wires and small constants, which is what skews the buckets of the binding MSMs"""
    s = Subcircuit(sid, "synthGate%d" % sid, n_out, n_in, n_prv)
    known = [0] + list(s.ins())
    lc = lambda: [(w, rnd.randrange(1, R)) for w in rnd.sample(known, min(fan, len(known)))]   # noqa: E731
    for t in list(s.prvs()) + list(s.outs()):
        if t in s.prvs() and rnd.random() < bit_fraction:
            c = rnd.choice([0, 0, 0, 1, 1, 1, 1, rnd.randrange(2, 256)])
            s.define(t, [(0, c)], [(0, 1)], [])
        else:
            s.define(t, lc(), lc(), lc() if rnd.random() < 0.5 else [])
        known.append(t)
    return s


def generate(rnd, s_max=8, n_gate_kinds=2, n_out=2, n_in=3, n_prv=6, k_pub=2, used_placements=None, l_free=4, l_extra=4, n=None,
             m_i=None, pool=None, k_out=1, bit_fraction=0.0):
    """-> the instance in memory (setup params, subcircuit infos, placements with witnesses, permutation, instance).

    Public wires follow the layout the CRS is built for (Sigma1::gen, libs/src/group_structures/mod.rs:405-440): placement 0 is
    bufferPubOut (k_out = l_user_out public outputs), 1 bufferPubIn (k_pub user inputs), 2 bufferBlockIn (l_free - l_user block
    inputs), 3 bufferEVMIn (l_extra = l - l_free function inputs); gate placements follow, fed by the buffers' interface outputs
    and by earlier gates, and bufferPubOut takes its inputs from gate outputs.  Every feed is a copy constraint.
    pool=K (production shapes): only K distinct gate witnesses are computed — every gate placement past the first K re-uses one
    of them together with its input wiring, so generation stays at K forward evaluations however many placements there are."""
    l_user_out, l_user = k_out, k_out + k_pub
    k_block, k_fn = l_free - l_user, l_extra
    assert k_block >= 0 and l_free & (l_free - 1) == 0
    l = l_free + k_fn
    subs = [buffer(i, name, k) for i, (name, k) in enumerate(zip(BUFFER_NAMES, (k_out, k_pub, k_block, k_fn)))]
    # n_prv: one count for every gate kind, or a list with one count per kind (the production shape: private wires summing to the
    # real library's m_D - l_D = 21767)
    prv_counts = list(n_prv) if isinstance(n_prv, (list, tuple)) else [n_prv] * n_gate_kinds
    assert len(prv_counts) == n_gate_kinds
    subs += [random_gates(4 + g, rnd, n_out, n_in, prv_counts[g], bit_fraction=bit_fraction) for g in range(n_gate_kinds)]
    pub_base = (0, l_user_out, l_user, l_free)        # first public wire of each buffer
    # interface wires: constant + the non-public side of the buffers, constant + outputs + inputs of the gates
    need_iface = sum(1 + s.n_out for s in subs[:4]) + sum(1 + s.n_out + s.n_in for s in subs[4:])
    if m_i is None:
        m_i = 1 << (need_iface - 1).bit_length()
    assert need_iface <= m_i and m_i & (m_i - 1) == 0
    l_D = l + m_i
    nxt_iface, nxt_prv = l, l_D
    for s in subs:
        fm = [0] * s.n_wires
        fm[0] = nxt_iface
        nxt_iface += 1
        public = s.outs() if s.id == 0 else (s.ins() if s.id < 4 else ())
        for w in list(s.outs()) + list(s.ins()):
            if w in public:
                continue
            fm[w] = nxt_iface
            nxt_iface += 1
        for j, w in enumerate(public):
            fm[w] = pub_base[s.id] + j
        for w in s.prvs():
            fm[w] = nxt_prv
            nxt_prv += 1
        s.flatten_map = fm
    m_D = nxt_prv
    n_min = 1 << max(0, max(len(s.rows) for s in subs) - 1).bit_length()
    n = n_min if n is None else n
    assert n >= n_min
    sp = {"l_free": l_free, "l_user_out": l_user_out, "l_user": l_user, "l": l, "l_D": l_D, "m_D": m_D, "n": n, "s_D": len(subs), "s_max": s_max}
    infos = [{"id": s.id, "name": s.name, "Nwires": s.n_wires, "Nconsts": len(s.rows), "Out_idx": [1, s.n_out],
              "In_idx": [1 + s.n_out, s.n_in], "flattenMap": s.flatten_map} for s in subs]

    used = s_max if used_placements is None else used_placements
    assert 5 <= used <= s_max, "four buffers and at least one gate placement"
    hx = lambda v: "0x%x" % v                                                        # noqa: E731
    placements, hexes, sources = [None] * 4, [None] * 4, [None] * 4
    produced = []                                     # (placement, local wire) whose value may be consumed
    for b in (1, 2, 3):
        w = subs[b].witness([rnd.randrange(R) for _ in subs[b].ins()])
        placements[b], hexes[b] = (subs[b], w), [hx(v) for v in w]
        produced += [(b, o) for o in subs[b].outs()]
    consumers = {}                                    # source (placement, wire) -> [(placement, wire), ...]
    gate_outs = []
    for p in range(4, used):
        if pool is not None and p >= 4 + pool:
            q = 4 + rnd.randrange(pool)               # same subcircuit, same sources, same witness as placement q
            s, w, srcs = placements[q][0], placements[q][1], sources[q]
            hexes.append(hexes[q])
        else:
            s = subs[4 + rnd.randrange(n_gate_kinds)]
            srcs = [rnd.choice(produced) for _ in s.ins()]
            w = s.witness([placements[sp_][1][sw] for sp_, sw in srcs])
            hexes.append([hx(v) for v in w])
            produced += [(p, o) for o in s.outs()]
            gate_outs += [(p, o) for o in s.outs()]
        placements.append((s, w))
        sources.append(srcs)
        for src, dst in zip(srcs, s.ins()):
            consumers.setdefault(src, []).append((p, dst))
    srcs0 = [rnd.choice(gate_outs) for _ in subs[0].ins()]
    w0 = subs[0].witness([placements[sp_][1][sw] for sp_, sw in srcs0])
    placements[0], hexes[0] = (subs[0], w0), [hx(v) for v in w0]
    for src, dst in zip(srcs0, subs[0].ins()):
        consumers.setdefault(src, []).append((0, dst))
    cell = lambda pl, wire: (placements[pl][0].flatten_map[wire] - l, pl)              # noqa: E731   (row, col)
    perm = []
    for src, dsts in consumers.items():
        cyc = [cell(*src)] + [cell(*d) for d in dsts]
        for a, b in zip(cyc, cyc[1:] + cyc[:1]):
            perm.append({"row": a[0], "col": a[1], "X": b[0], "Y": b[1]})
    pv = [{"subcircuitId": s.id, "variables": h} for (s, _), h in zip(placements, hexes)]
    pub = lambda b, wires: [hx(placements[b][1][w]) for w in wires]                    # noqa: E731
    instance = {"a_pub_user": pub(0, subs[0].outs()) + pub(1, subs[1].ins()), "a_pub_block": pub(2, subs[2].ins()),
                "a_pub_function": pub(3, subs[3].ins())}
    return {"setup_params": sp, "infos": infos, "subs": subs, "placements": placements, "placement_variables": pv,
            "instance": instance, "permutation": perm, "m_i": m_i,
            "r1cs_rows": sum(len(s.rows) for s, _ in placements)}


def write(inst, out_dir, synth_files=True):
    """<out_dir>/qap/{setupParams.json, subcircuitInfo.json, r1cs/subcircuit{id}.r1cs} and, unless synth_files is False,
    <out_dir>/synth/{placementVariables.json, instance.json, permutation.json}"""
    qap, synth = os.path.join(out_dir, "qap"), os.path.join(out_dir, "synth")
    os.makedirs(os.path.join(qap, "r1cs"), exist_ok=True)
    for s in inst["subs"]:
        write_r1cs(os.path.join(qap, "r1cs", "subcircuit%d.r1cs" % s.id), s.n_wires, s.rows)
    json.dump(inst["setup_params"], open(os.path.join(qap, "setupParams.json"), "w"))
    json.dump(inst["infos"], open(os.path.join(qap, "subcircuitInfo.json"), "w"))
    inst["qap"] = qap
    if synth_files:
        os.makedirs(synth, exist_ok=True)
        json.dump(inst["placement_variables"], open(os.path.join(synth, "placementVariables.json"), "w"))
        json.dump(inst["instance"], open(os.path.join(synth, "instance.json"), "w"))
        json.dump(inst["permutation"], open(os.path.join(synth, "permutation.json"), "w"))
        inst["synth"] = synth
    return inst


def build(out_dir, rnd, **shape):
    return write(generate(rnd, **shape), out_dir)
