"""Direct parity tests of the witness-side entries of Prover::init (include/tkmk.h "Witness side of the path"), each on its own
through the C ABI — in whole proofs they only ever see the wire lists tools/synth_circuit.py emits.

  tkmk_r1cs_library_create / _eval   vs the oracle's sparse-row evaluation (eval_uvwxy_sparse_rows / eval_sparse_rows,
                                     libs/src/iotools/mod.rs:1426-1523,1590-1608) on the three committed reference `.r1cs` fixtures
  tkmk_witness_route                 vs the reference's loops restated in a few lines of Python: gen_bXY
                                     (libs/src/polynomial_structures/mod.rs:132-162) and encode_statement_common /
                                     encode_o_pub_free_common (libs/src/group_structures/mod.rs:184-300), on hand-made wire lists
  tkmk_fr_scatter_table              vs numpy, Permutation::to_poly's redirects (libs/src/iotools/mod.rs:438-448), incl. bad indices
"""
import json
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
# committed reference data (tests/golden/make_pins.py): ALL fourteen kinds of the production library — three small subcircuits, four
# mid-size ones (bufferPubOut, bufferEVMIn, DecToBit, Accumulator) and the seven largest (bufferPrvIn, ALU1, ALU2, SubExpBatch with 5093
# wires, Poseidon, JubjubExpBatch, VerifyMerkleProof: 270-540 KB each)
FIXTURES = [("qap", 1), ("qap", 2), ("qap", 12), ("qap_more", 0), ("qap_more", 3), ("qap_more", 7), ("qap_more", 9)] + [
    ("qap_rest", i) for i in (4, 5, 6, 8, 10, 11, 13)]


def _library(oracle):
    from tkmk import r1cs
    bins = []
    for d, sid in FIXTURES:
        infos = {e["id"]: e for e in json.load(open(os.path.join(GOLDEN, d, "subcircuitInfo.json")))}      # qap_rest holds the library's whole file
        b = r1cs.R1csBinary.read(os.path.join(GOLDEN, d, "r1cs", "subcircuit%d.r1cs" % sid))
        assert b.n_wires == infos[sid]["Nwires"]
        bins.append(b)
    return bins, [b.csr() for b in bins]


def _fr(oracle, vals):
    return oracle.to_bytes([v % oracle.R_MOD for v in vals], 32)


def test_r1cs_library_eval_vs_oracle_sparse_rows(gpu, oracle):
    """all fourteen kinds of the reference's production library (up to 5093 wires, 3936 rows), every kind placed 3 times in mixed order, random witnesses with zeros / ones / r-1 mixed in, placements fewer than s_max:
    u / v / w (n x s_max, element (row, placement)) must equal the oracle's row evaluation of each placement, zeros elsewhere"""
    from tkmk import witness
    tk = gpu
    bins, csrs = _library(oracle)
    n_rows = [b.n_constraints for b in bins]
    n_wires = [b.n_wires for b in bins]
    n = 1 << max(6, (max(n_rows) - 1).bit_length())
    s_max = 64
    rnd = random.Random(2026)
    order = list(range(len(bins))) * 3                                 # 42 placements <= s_max, every one of the 14 kinds 3 times
    rnd.shuffle(order)
    assert n == 4096                                                   # the production library's n (setupParams.json)
    lib = witness.R1csLibrary(csrs, n_rows, n_wires)
    offsets, chunks, pos = [], [], 5                                   # variables do not start at element 0 and are not back to back
    for k, kind in enumerate(order):
        vals = [rnd.choice((0, 1, oracle.R_MOD - 1, rnd.randrange(oracle.R_MOD), rnd.randrange(1 << 128))) for _ in range(n_wires[kind])]
        if k == 3:
            vals = [0] * n_wires[kind]                                 # an all-zero placement
        offsets.append(pos)
        chunks.append((pos, vals))
        pos += n_wires[kind] + rnd.randrange(0, 7)
    flat = [rnd.randrange(oracle.R_MOD) for _ in range(pos + 3)]       # garbage between the placements must not be read into results
    for p, vals in chunks:
        flat[p:p + len(vals)] = vals
    d_vars = tk.DeviceBuffer.from_host(_fr(oracle, flat))
    got = lib.eval(d_vars, order, offsets, n, s_max)
    for m in range(3):
        want = np.zeros((n, s_max, 32), np.uint8)
        for slot, kind in enumerate(order):
            ptr, wires, coeff = csrs[kind][m]
            var = _fr(oracle, flat[offsets[slot]:offsets[slot] + n_wires[kind]])
            rows = oracle.r1cs_eval_rows(ptr, wires, coeff if coeff.size else np.zeros(32, np.uint8), var, n)
            want[:, slot, :] = np.asarray(rows).reshape(n, 32)
        assert (got[m].reshape(n, s_max, 32) == want).all(), "matrix %d" % m
        assert want[:n_rows[order[0]], 0].any()                        # the comparison is not zeros against zeros
    # no placements at all: three zero matrices; more placements than s_max, or a library taller than n: refused
    z = lib.eval(d_vars, [], [], n, s_max)
    assert not any(a.any() for a in z)
    with pytest.raises(tk.TkmkError):
        lib.eval(d_vars, order * 2, offsets * 2, n, s_max)
    with pytest.raises(tk.TkmkError):
        lib.eval(d_vars, order, offsets, 8, s_max)
    lib.close()


def test_r1cs_library_create_validates(gpu, oracle):
    """a wire index past n_wires, a decreasing row_ptr, row_ptr[0] != 0: refused at create (so _eval cannot read outside a placement)"""
    from tkmk import witness
    tk = gpu
    one = _fr(oracle, [1])
    ok = (np.array([0, 1, 1], np.uint32), np.array([2], np.uint32), one)
    empty = (np.array([0, 0, 0], np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.uint8))
    witness.R1csLibrary([(ok, empty, empty)], [2], [3]).close()
    for bad in ((np.array([0, 1, 1], np.uint32), np.array([3], np.uint32), one),          # wire 3 of 3
                (np.array([0, 1, 0], np.uint32), np.array([2], np.uint32), one),          # decreasing
                (np.array([1, 1, 1], np.uint32), np.array([2], np.uint32), one)):         # does not start at 0
        with pytest.raises(tk.TkmkError):
            witness.R1csLibrary([(bad, empty, empty)], [2], [3])


def _route_reference(flat, var_off, slots, list_wire, list_row, m_rows, stride, index_inner, add_slot):
    """the reference's loops: for placement i (global index slots[i]) and list entry e (local wire j, row g):
    gen_bXY:                 interface_witness[g * s_max + i] = variables[j]          (polynomial_structures/mod.rs:146-153)
    encode_statement_common: aligned_variable.push(variables[j]); aligned_rs.push(base_at(g, i))   (group_structures/mod.rs:281-289)
    with base_at(g, i) = table[g * index_inner + i] (eta / delta tables) or gamma_at(g) = table[g] (encode_o_pub_free_common :224-228)"""
    matrix = [[0] * stride for _ in range(m_rows)]
    scalars, index = [], []
    for i, off in enumerate(var_off):
        for j, g in zip(list_wire, list_row):
            v = flat[off + j]
            matrix[g][slots[i]] = v
            scalars.append(v)
            index.append(g * index_inner + (slots[i] if add_slot else 0))
    return matrix, scalars, index


@pytest.mark.parametrize("case", ["empty_kind", "single_wire", "max_index_wire", "repeated_rows", "many"])
def test_witness_route_vs_reference_loops(gpu, oracle, case):
    from tkmk import witness
    tk = gpu
    rnd = random.Random(sum(map(ord, case)))
    n_wires, m_rows, s_max = 37, 11, 8
    if case == "empty_kind":            # a kind with no wire on the range (bufferEVMIn in encode_o_pub_free_common) / no placement of it
        variants = [([], [], [0, 1]), ([3], [2], [])]
    elif case == "single_wire":
        variants = [([5], [7], [4])]
    elif case == "max_index_wire":      # last local wire, last row, last slot
        variants = [([n_wires - 1, 0], [m_rows - 1, 0], [s_max - 1, 0])]
    elif case == "repeated_rows":       # two local wires mapped to the same global row: the later one wins in the matrix (serial loop)
        variants = [([1, 9, 4], [3, 3, 6], [2, 5])]
    else:
        m_rows = 40
        variants = [(rnd.sample(range(n_wires), 20), rnd.sample(range(m_rows), 20), rnd.sample(range(s_max), 6))]
    for list_wire, list_row, slots in variants:
        var_off = [3 + 50 * k for k in range(len(slots))]
        flat = [rnd.choice((0, 1, rnd.randrange(oracle.R_MOD))) for _ in range(3 + 50 * max(len(slots), 1) + n_wires)]
        d_vars = tk.DeviceBuffer.from_host(_fr(oracle, flat))
        for index_inner, add_slot in ((s_max, True), (1, False)):
            d_matrix = tk.DeviceBuffer.from_host(np.zeros(32 * m_rows * s_max, np.uint8))
            got = witness.witness_route(d_vars, var_off, slots, list_wire, list_row, matrix_dev=d_matrix, matrix_stride=s_max, want_lists=True,
                                        index_inner=index_inner, index_add_slot=add_slot)
            matrix, scalars, index = _route_reference(flat, var_off, slots, list_wire, list_row, m_rows, s_max, index_inner, add_slot)
            if case == "repeated_rows":
                # the kernel writes the cells of one placement concurrently: which of two wires on ONE row lands is unordered there,
                # so the service never passes such a list to the matrix output (flattenMap is injective: subcircuitInfo.json); the
                # lists keep every entry in order
                for i, sl in enumerate(slots):
                    cell = int.from_bytes(d_matrix.to_host()[32 * (3 * s_max + sl):32 * (3 * s_max + sl + 1)].tobytes(), "little")
                    assert cell in (flat[var_off[i] + 1], flat[var_off[i] + 9])
            else:
                assert (d_matrix.to_host() == _fr(oracle, [v for row in matrix for v in row])).all()
            total = len(slots) * len(list_wire)
            if total:
                assert (got[0] == _fr(oracle, scalars)).all()
                assert got[1].tolist() == index
            else:
                assert got[0].size == 0 and got[1].size == 0
        # matrix only / lists only
        d_matrix = tk.DeviceBuffer.from_host(np.zeros(32 * m_rows * s_max, np.uint8))
        assert witness.witness_route(d_vars, var_off, slots, list_wire, list_row, matrix_dev=d_matrix, matrix_stride=s_max) is None
        if case != "repeated_rows":
            assert (d_matrix.to_host() == _fr(oracle, [v for row in _route_reference(flat, var_off, slots, list_wire, list_row, m_rows, s_max, 1, False)[0] for v in row])).all()


def test_witness_route_lists_feed_the_binding_msm(gpu, oracle):
    """the (scalar, CRS row) lists are what tkmk_msm_multi_ex consumes as base_index: MSM over the routed lists = the oracle's MSM over
    the operands gathered the reference's way (encode_statement_common: base_at(global_idx, i) = table[global_idx * s_max + i])"""
    from tkmk import witness
    tk = gpu
    rnd = random.Random(77)
    n_wires, m_rows, s_max = 24, 6, 4
    table = np.asarray(oracle.g1_random_bases(5, m_rows * s_max)).reshape(-1, 96)
    list_wire, list_row, slots = [2, 5, 11, 23], [0, 5, 3, 1], [1, 3, 0]
    var_off = [0, n_wires, 2 * n_wires]
    flat = [rnd.choice((0, 1, rnd.randrange(oracle.R_MOD))) for _ in range(3 * n_wires)]
    d_vars = tk.DeviceBuffer.from_host(_fr(oracle, flat))
    sc, ix = witness.witness_route(d_vars, var_off, slots, list_wire, list_row, want_lists=True, index_inner=s_max, index_add_slot=True)
    d_table = tk.DeviceBuffer.from_host(np.ascontiguousarray(table.reshape(-1)))
    got = tk.projective_to_affine_bytes(tk.msm_multi_ex([dict(scalars=tk.DeviceBuffer.from_host(sc), bases=d_table, n=ix.size,
                                                              base_index=tk.DeviceBuffer.from_host(ix.view(np.uint8)), table_len=m_rows * s_max)]))
    _, scalars, index = _route_reference(flat, var_off, slots, list_wire, list_row, m_rows, s_max, s_max, True)
    want = oracle.g1_msm(_fr(oracle, scalars), np.ascontiguousarray(table[index].reshape(-1)))
    assert (got == np.asarray(want)).all()


def test_fr_scatter_table(gpu, oracle):
    """out[dst[i]] = table[src[i]]: Permutation::to_poly's s0[row][col] = w_x^X (iotools/mod.rs:438-448); untouched cells keep their
    value; an index past either array is an error, never a stray access, and the valid entries of that call still land"""
    from tkmk import witness
    tk = gpu
    rnd = np.random.default_rng(9)
    t_len, o_len, k = 64, 4096, 1500
    table = np.asarray(oracle.fr_random(3, t_len)).reshape(t_len, 32)
    base = np.asarray(oracle.fr_random(4, o_len)).reshape(o_len, 32)
    src = rnd.integers(0, t_len, k, dtype=np.uint32)
    dst = rnd.permutation(o_len)[:k].astype(np.uint32)                  # distinct destinations
    src[0], dst[0] = t_len - 1, o_len - 1                               # the last element of both arrays
    d_table = tk.DeviceBuffer.from_host(np.ascontiguousarray(table.reshape(-1)))
    d_out = tk.DeviceBuffer.from_host(np.ascontiguousarray(base.reshape(-1)))
    witness.fr_scatter_table(d_table, t_len, src, dst, d_out, o_len)
    want = base.copy()
    want[dst] = table[src]
    assert (d_out.to_host().reshape(o_len, 32) == want).all()
    witness.fr_scatter_table(d_table, t_len, [], [], d_out, o_len)      # n = 0: nothing happens
    assert (d_out.to_host().reshape(o_len, 32) == want).all()
    for bad_src, bad_dst in (([1, t_len, 2], [10, 11, 12]), ([1, 2, 3], [10, o_len, 12]), ([0xffffffff], [0]), ([0], [0xffffffff])):
        d_o = tk.DeviceBuffer.from_host(np.ascontiguousarray(base.reshape(-1)))
        with pytest.raises(tk.TkmkError) as e:
            witness.fr_scatter_table(d_table, t_len, bad_src, bad_dst, d_o, o_len)
        assert e.value.code == 11                                  # TKMK_ERR_INVALID_ARGUMENT (include/tkmk.h)
        w2 = base.copy()
        for s_i, d_i in zip(bad_src, bad_dst):
            if s_i < t_len and d_i < o_len:
                w2[d_i] = table[s_i]
        assert (d_o.to_host().reshape(o_len, 32) == w2).all()
