"""gpu tier: the resident prover (libtkmk_prover.so, include/tkmk_prover.h, host/tkmk_service.hpp) and the MSM over views of
resident tables (tkmk_msm_multi_ex) it is built on.

  * tkmk_msm_multi_ex == bls12_381_msm on the gathered operands (oracle-checked), for strided boxes, index lists, all three
    base forms, tiny sizes and the out-of-range refusal;
  * the service's proof == the Python prover's (tkmk/prove.py) == the exponent restatement (tests/prove_ref.py), bit for bit, for
    the same blinding scalars — i.e. the device-side init (CSR library, witness routing, table scatter) reproduces
    Prover::init (packages/backend/prove/src/lib.rs:675-1206);
  * the same from the reference's own CRS container, combined_sigma.rkyv (prove/src/sigma_source.rs:22-32), under every field
    order the reader knows; preprocess from sigma_preprocess.rkyv (preprocess/src/main.rs:47-53)."""
import json
import os
import random
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


# ------------------------------------------------------------------------------------------------ MSM over views
def _aff(tkmk, proj):
    return tkmk.projective_to_affine_bytes(proj)


@pytest.mark.parametrize("form", ["plain", "montgomery", "converted"])
def test_msm_multi_ex_strided_boxes(gpu, oracle, form):
    """a tx x ty coefficient box of an xs x ys matrix against the matching sub-grid of a rows x cols table (encode_poly's shape)"""
    tk = gpu
    rows, cols = 40, 24                                             # the "CRS"
    table = np.asarray(oracle.g1_random_bases(5, rows * cols))
    table[96 * 7:96 * 8] = 0                                        # an infinity record inside the grid
    xs, ys = 32, 16                                                 # the coefficient matrix
    coeffs = np.asarray(oracle.fr_random(6, xs * ys))
    coeffs[32 * 3:32 * 4] = 0
    d_coeffs = tk.DeviceBuffer.from_host(coeffs)
    if form == "montgomery":                                        # ICICLE's Montgomery form: x * 2^384 mod p, (0,0) stays infinity
        mont = oracle.to_bytes([(x << 384) % oracle.P_MOD for x in oracle.to_ints(table, 48)], 48)
        d_table = tk.DeviceBuffer.from_host(np.asarray(mont))
        bf = tk.BASES_MONTGOMERY
    elif form == "converted":
        d_table = tk.msm_convert_bases(table)
        bf = tk.BASES_CONVERTED
    else:
        d_table = tk.DeviceBuffer.from_host(table)
        bf = tk.BASES_PLAIN
    jobs, want = [], []
    for tx, ty in ((32, 16), (17, 9), (1, 16), (5, 1), (2, 1), (1, 1)):
        jobs.append(dict(scalars=d_coeffs, bases=d_table, n=tx * ty, scalar_view=(ty, ys), base_view=(ty, cols), table_len=rows * cols))
        c = coeffs.reshape(xs, ys, 32)[:tx, :ty].reshape(-1)
        b = table.reshape(rows, cols, 96)[:tx, :ty].reshape(-1)
        want.append(np.asarray(oracle.g1_msm(np.ascontiguousarray(c), np.ascontiguousarray(b))))
    got = _aff(tk, tk.msm_multi_ex(jobs, bases_form=bf))
    for k, w in enumerate(want):
        assert (got[96 * k:96 * (k + 1)] == w).all(), (form, k)
    # a box that does not fit the table is refused before anything is launched
    with pytest.raises(tk.TkmkError):
        tk.msm_multi_ex([dict(scalars=d_coeffs, bases=d_table, n=rows * cols, scalar_view=None, base_view=(cols, cols + 1), table_len=rows * cols)], bases_form=bf)


def test_msm_multi_ex_index_lists_and_sizes(gpu, oracle):
    """gathered rows of a binding table (msm_g1_bases' shape): repeated rows, witness-like scalars, sizes across the sort variants"""
    tk = gpu
    rnd = np.random.default_rng(9)
    table_pts = 5000
    table = np.asarray(oracle.g1_random_bases(11, table_pts))
    d_conv = tk.msm_convert_bases(table)
    d_plain = tk.DeviceBuffer.from_host(table)
    for n in (2, 3, 64, 1000, 70000, (1 << 18) + 5):
        idx = rnd.integers(0, table_pts, n, dtype=np.uint32)
        sc = np.asarray(oracle.fr_random(100 + n, n)).reshape(n, 32).copy()
        sc[rnd.random(n) < 0.5] = 0
        sc[rnd.random(n) < 0.3, 1:] = 0                              # small values
        sc = sc.reshape(-1)
        want = np.asarray(oracle.g1_msm(sc, np.ascontiguousarray(table.reshape(-1, 96)[idx].reshape(-1))))
        d_sc, d_idx = tk.DeviceBuffer.from_host(sc), tk.DeviceBuffer.from_host(idx.view(np.uint8))
        for bases, bf in ((d_conv, tk.BASES_CONVERTED), (d_plain, tk.BASES_PLAIN)):
            got = _aff(tk, tk.msm_multi_ex([dict(scalars=d_sc, bases=bases, n=n, base_index=d_idx, table_len=table_pts)], bases_form=bf))
            assert (got == want).all(), (n, bf)
    # empty and one-point jobs next to a real one
    sc = np.asarray(oracle.fr_random(3, 4))
    d_sc = tk.DeviceBuffer.from_host(sc)
    d_idx = tk.DeviceBuffer.from_host(np.array([4999, 0, 17, 17], np.uint32).view(np.uint8))
    got = _aff(tk, tk.msm_multi_ex([dict(scalars=d_sc, bases=d_conv, n=0, table_len=table_pts),
                                    dict(scalars=d_sc, bases=d_conv, n=1, base_index=d_idx, table_len=table_pts),
                                    dict(scalars=d_sc, bases=d_conv, n=4, base_index=d_idx, table_len=table_pts)], bases_form=tk.BASES_CONVERTED))
    rows = table.reshape(-1, 96)
    assert not got[:96].any()
    assert (got[96:192] == np.asarray(oracle.g1_msm(sc[:32], np.ascontiguousarray(rows[4999])))).all()
    assert (got[192:] == np.asarray(oracle.g1_msm(sc, np.ascontiguousarray(rows[[4999, 0, 17, 17]].reshape(-1))))).all()
    # an index past the table: an error, not a fault (the read is clamped on the device)
    d_bad = tk.DeviceBuffer.from_host(np.array([1, table_pts, 2, 3], np.uint32).view(np.uint8))
    with pytest.raises(tk.TkmkError):
        tk.msm_multi_ex([dict(scalars=d_sc, bases=d_conv, n=4, base_index=d_bad, table_len=table_pts)], bases_form=tk.BASES_CONVERTED)
    # an index list without the table's length would be gathered unchecked: refused (1 and 4 points: host-resolved and pipeline paths)
    for n_pts in (1, 4):
        with pytest.raises(tk.TkmkError):
            tk.msm_multi_ex([dict(scalars=d_sc, bases=d_conv, n=n_pts, base_index=d_idx, table_len=0)], bases_form=tk.BASES_CONVERTED)


# ------------------------------------------------------------------------------------------------ the resident prover
def _mixer_file(tmp_path, mixer):
    hx = lambda v: [hx(e) for e in v] if isinstance(v, list) else "0x%x" % v          # noqa: E731
    path = str(tmp_path / "mixer.json")
    json.dump({k: hx(v) for k, v in mixer.items()}, open(path, "w"))
    return path


SHAPES = [(51, dict(s_max=8, n_gate_kinds=2)),
          (52, dict(s_max=8, n_gate_kinds=3, n_out=1, n_in=2, n_prv=11, used_placements=6, bit_fraction=0.6)),
          (53, dict(s_max=16, n_gate_kinds=2, n_out=2, n_in=3, n_prv=9, used_placements=11, k_out=0, k_pub=3)),
          (54, dict(s_max=8, n_gate_kinds=2, n_out=1, n_in=2, n_prv=1, k_out=0, k_pub=2, l_extra=1, used_placements=7))]


@pytest.mark.parametrize("seed,shape", SHAPES)
def test_service_equals_python_prover_and_restatement(gpu, oracle, tmp_path, seed, shape):
    import prove_ref
    import synth_circuit
    from test_gpu_prove import _stage_crs_file, seeded_mixer
    from tkmk import proofio, service
    from tkmk.prove import Prover, run_rounds
    inst = synth_circuit.build(str(tmp_path), random.Random(seed), **shape)
    crs_dir, out_dir = str(tmp_path / "crs"), str(tmp_path / "out")
    sigma, crs, g = _stage_crs_file(gpu, oracle, inst, crs_dir)
    mixer = seeded_mixer(seed)
    mixer_path = _mixer_file(tmp_path, mixer)
    prover, binding = Prover.init(inst["qap"], inst["synth"], None, mixer=mixer, sigma=sigma)
    points, scalars, _, _, _ = run_rounds(prover, binding)
    want_doc = proofio.format_proof(points, scalars)
    with service.Prover(inst["qap"], crs_dir, testing=True) as p:
        assert p.crs_source == "combined_sigma.tkcrs"
        for rep in range(2):                                          # the context is reusable: same inputs, same proof
            doc, tm = p.prove(inst["synth"], out_dir, testing_mixer_json=mixer_path)
            assert doc == want_doc
            assert json.load(open(os.path.join(out_dir, "proof.json"))) == want_doc
            assert tm["init_s"] > 0 and tm["rounds_s"] > 0 and abs(tm["total_s"] - (tm["init_s"] + tm["rounds_s"] + tm["write_s"])) < 0.05
        fresh, _ = p.prove(inst["synth"], None)                       # production path: blinding from getrandom()
        fp, _ = proofio.recover_proof(fresh)
        assert (np.asarray(fp["A_free"]) == np.asarray(points["A_free"])).all() and not (np.asarray(fp["U"]) == np.asarray(points["U"])).all()
        # a corrupted witness document is an error of that call; the context stays usable
        pv = os.path.join(inst["synth"], "placementVariables.json")
        good = open(pv).read()
        open(pv, "w").write(good[:len(good) // 2])
        with pytest.raises(service.ProverError) as e:
            p.prove(inst["synth"], None)
        assert "placementVariables.json" in str(e.value) or "Corrupted placement variables" in str(e.value)
        open(pv, "w").write(good)
        doc, _ = p.prove(inst["synth"], None, testing_mixer_json=mixer_path)
        assert doc == want_doc
    dlogs, ref_scalars, _, _, _ = prove_ref.run(inst, crs, mixer, g)
    native_points, native_scalars = proofio.recover_proof(want_doc)
    assert native_scalars == ref_scalars
    for name in proofio.PROOF_POINT_ORDER:
        assert (np.asarray(native_points[name]) == np.asarray(prove_ref.g1_of(dlogs[name], g))).all(), name


@pytest.mark.parametrize("seed,shape", SHAPES[1:3])
def test_service_with_lagrange_tables_equals_python_prover(gpu, oracle, tmp_path, monkeypatch, seed, shape):
    """the evaluation-basis commitments of prove0 (U, V, W, B as (1/N) MSM(evaluations, Lagrange-basis table) + blinding terms): forced
    on at a small shape by asking for a commit table; the proof must be the Python prover's (coefficient route), byte for byte, and
    TKMK_PROVER_LAGRANGE=0 must give the same document through the coefficient route of the native side"""
    import synth_circuit
    from test_gpu_prove import _stage_crs_file, seeded_mixer
    from tkmk import proofio, service
    from tkmk.prove import Prover, run_rounds
    inst = synth_circuit.build(str(tmp_path), random.Random(seed), **shape)
    crs_dir, out_dir = str(tmp_path / "crs"), str(tmp_path / "out")
    sigma, crs, g = _stage_crs_file(gpu, oracle, inst, crs_dir)
    mixer = seeded_mixer(seed)
    mixer_path = _mixer_file(tmp_path, mixer)
    prover, binding = Prover.init(inst["qap"], inst["synth"], None, mixer=mixer, sigma=sigma)
    points, scalars, _, _, _ = run_rounds(prover, binding)
    want_doc = proofio.format_proof(points, scalars)
    monkeypatch.setenv("TKMK_PROVER_TABLE_C", "12")
    for lagrange in ("1", "0"):
        monkeypatch.setenv("TKMK_PROVER_LAGRANGE", lagrange)
        with service.Prover(inst["qap"], crs_dir, testing=True) as p:
            for rep in range(2):
                doc, _ = p.prove(inst["synth"], out_dir, testing_mixer_json=mixer_path)
                assert doc == want_doc, (lagrange, rep)


@pytest.mark.parametrize("order", ["rustc_size_groups", "rustc_align_only", "declared"])
def test_prove_and_preprocess_from_the_reference_archives(gpu, oracle, tmp_path, order):
    """<crs>/combined_sigma.rkyv and <crs>/sigma_preprocess.rkyv instead of the flat payload: same proof.json / preprocess.json"""
    import synth_circuit
    from test_gpu_prove import _stage_crs_file, seeded_mixer
    from tkmk import crs as crsmod
    from tkmk import rkyv, service
    inst = synth_circuit.build(str(tmp_path), random.Random(61), s_max=8, n_gate_kinds=2, used_placements=7)
    sp = inst["setup_params"]
    flat_dir, arch_dir, out_dir = str(tmp_path / "crs"), str(tmp_path / "crs_rkyv"), str(tmp_path / "out")
    _stage_crs_file(gpu, oracle, inst, flat_dir)
    sections = crsmod.read_payload(os.path.join(flat_dir, "combined_sigma.tkcrs"))
    os.makedirs(arch_dir)
    open(os.path.join(arch_dir, "combined_sigma.rkyv"), "wb").write(rkyv.encode_combined_sigma(sections, rkyv.rows_for(sp), order))
    open(os.path.join(arch_dir, "sigma_preprocess.rkyv"), "wb").write(rkyv.encode_sigma_preprocess(sections["xy_powers"], sections["gamma_inv_o_inst"]))
    mixer_path = _mixer_file(tmp_path, seeded_mixer(61))
    with service.Prover(inst["qap"], flat_dir, testing=True) as p:
        want, _ = p.prove(inst["synth"], None, testing_mixer_json=mixer_path)
    with service.Prover(inst["qap"], arch_dir, testing=True) as p:
        assert p.crs_source == "combined_sigma.rkyv"
        got, _ = p.prove(inst["synth"], None, testing_mixer_json=mixer_path)
    assert got == want
    # the binaries on the archive directory
    os.makedirs(out_dir)
    bins = os.path.join(ROOT, "tokamak-zk-evm_amd", "bin")
    args = ["--crs", arch_dir, "--synthesizer-stat", inst["synth"], "--output", out_dir, "--subcircuit-library", inst["qap"]]
    r = subprocess.run([os.path.join(bins, "prove-testing")] + args + ["--testing-mixer", mixer_path], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert "combined_sigma.rkyv" in r.stdout and json.load(open(os.path.join(out_dir, "proof.json"))) == want
    r = subprocess.run([os.path.join(bins, "preprocess")] + args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    pre_arch = json.load(open(os.path.join(out_dir, "preprocess.json")))
    r = subprocess.run([os.path.join(bins, "preprocess")] + ["--crs", flat_dir] + args[2:], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert json.load(open(os.path.join(out_dir, "preprocess.json"))) == pre_arch
    # an archive of another circuit is refused with the reason
    open(os.path.join(arch_dir, "combined_sigma.rkyv"), "wb").write(rkyv.encode_combined_sigma(sections, rkyv.rows_for(sp), order)[:-16])
    with pytest.raises(service.ProverError) as e:
        service.Prover(inst["qap"], arch_dir)
    assert "Invalid sigma archive" in str(e.value)


def test_native_setup_writes_archives_the_prover_reads(gpu, tmp_path):
    """bin/trusted-setup --fixed-tau writes combined_sigma.rkyv, sigma_preprocess.rkyv, sigma_verify.json and combined_sigma.tkcrs; they hold
    the same sections, and the service proves from either container"""
    import synth_circuit
    from tkmk import crs as crsmod
    from tkmk import rkyv
    inst = synth_circuit.build(str(tmp_path), random.Random(71), s_max=8, n_gate_kinds=2, used_placements=6)
    out = str(tmp_path / "crs")
    os.makedirs(out)
    r = subprocess.run([os.path.join(ROOT, "tokamak-zk-evm_amd", "bin", "trusted-setup"), "--subcircuit-library", inst["qap"], "--output", out, "--fixed-tau"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    flat = crsmod.read_payload(os.path.join(out, "combined_sigma.tkcrs"))
    arch, rows, order = rkyv.decode_combined_sigma(open(os.path.join(out, "combined_sigma.rkyv"), "rb").read(), expect=rkyv.expect_for(inst["setup_params"]),
                                                   want_details=True)
    assert order == "rustc_size_groups" and rows == rkyv.rows_for(inst["setup_params"])
    for name in crsmod.SECTION_NAMES:
        assert bytes(arch[name]) == bytes(flat[name]), name
    pre = rkyv.decode_sigma_preprocess(open(os.path.join(out, "sigma_preprocess.rkyv"), "rb").read())
    assert bytes(pre["xy_powers"]) == bytes(flat["xy_powers"]) and bytes(pre["gamma_inv_o_inst"]) == bytes(flat["gamma_inv_o_inst"])
    # sigma_verify.json, the verifier's part (SigmaVerify: libs/src/group_structures/mod.rs:849-860; points as {"x": hex, "y": hex} read
    # back with from_hex: iotools/mod.rs:999-1059): the same points as the payload's singles and G2 section, the fixed generators of the
    # reference's recipe among them (tests/golden/pins.json)
    from tkmk import g2
    sv = json.load(open(os.path.join(out, "sigma_verify.json")))
    assert list(sv) == ["G", "H", "sigma_1", "sigma_2", "lagrange_KL"] and list(sv["sigma_2"]) == list(crsmod.G2_POINTS[1:])
    g1_of = lambda name: bytes(crsmod.single_g1(flat, name))                                                        # noqa: E731
    g1_json = lambda pt: int(pt["x"], 16).to_bytes(48, "little") + int(pt["y"], 16).to_bytes(48, "little")          # noqa: E731
    assert g1_json(sv["G"]) == g1_of("G") and g1_json(sv["sigma_1"]["x"]) == g1_of("x") and g1_json(sv["sigma_1"]["y"]) == g1_of("y")
    assert g1_json(sv["lagrange_KL"]) == g1_of("lagrange_KL")
    recs = np.asarray(flat["g2"]).reshape(10, 192)
    for i, name in enumerate(crsmod.G2_POINTS):
        pt = sv["H"] if name == "H" else sv["sigma_2"][name]
        assert len(pt["x"]) == 2 + 192 and g2.from_hex_pair(pt["x"], pt["y"]) == g2.decode(recs[i]), name
    pins = json.load(open(os.path.join(ROOT, "tests", "golden", "pins.json")))
    assert int(sv["G"]["x"], 16) == int(pins["fixed_tau_g1_x"], 16) and int(sv["H"]["x"], 16) == int(pins["fixed_tau_g2_x"], 16)
    assert int(sv["H"]["y"], 16) == int(pins["fixed_tau_g2_y"], 16)


@pytest.mark.parametrize("c", [12, 16, 18, 20])
def test_msm_table_jobs_wide_windows_single_bucket_set(gpu, oracle, c):
    """precomputed TABLE jobs of tkmk_msm_multi_ex: the whole base table expanded once (bls12_381_msm_precompute_bases, factor =
    window count: one bucket set), then MSMs over strided boxes and index lists of level 0 — bit-identical to the oracle's MSM on
    the gathered operands, for window widths up to 20 bits (1024 x 512 bins in the two-pass sort)"""
    tk = gpu
    rows, cols = 600, 512                                           # table of 307200 points; boxes of >= 2^18 / windows entries
    table = np.asarray(oracle.g1_random_bases(50 + c, 4096))        # 4096 distinct points, tiled (the arithmetic does not care)
    table = np.ascontiguousarray(np.tile(table.reshape(-1, 96), (rows * cols // 4096, 1)).reshape(-1))
    table[96 * 11:96 * 12] = 0
    windows = 255 // c + 1
    d_table = tk.msm_precompute_bases(table, rows * cols, windows, c=c)
    assert d_table.nbytes == 96 * rows * cols * windows
    xs, ys = 1024, 512
    coeffs = np.asarray(oracle.fr_random(60 + c, 1 << 16))
    coeffs = np.ascontiguousarray(np.tile(coeffs.reshape(-1, 32), (xs * ys // (1 << 16), 1)).reshape(-1))
    coeffs[32 * 5:32 * 6] = 0
    d_coeffs = tk.DeviceBuffer.from_host(coeffs)
    rnd = np.random.default_rng(c)
    n_idx = 70000
    idx = rnd.integers(0, rows * cols, n_idx, dtype=np.uint32)
    d_idx = tk.DeviceBuffer.from_host(idx.view(np.uint8))
    jobs, want = [], []
    for tx, ty in ((600, 512), (513, 300), (37, 512)):
        if c > 16 and tx * ty * windows < (1 << 18):
            continue
        jobs.append(dict(scalars=d_coeffs, bases=d_table, n=tx * ty, scalar_view=(ty, ys), base_view=(ty, cols), table_len=rows * cols, table=(c, windows)))
        cc = coeffs.reshape(xs, ys, 32)[:tx, :ty].reshape(-1)
        bb = table.reshape(rows, cols, 96)[:tx, :ty].reshape(-1)
        want.append(np.asarray(oracle.g1_msm(np.ascontiguousarray(cc), np.ascontiguousarray(bb))))
    jobs.append(dict(scalars=d_coeffs, bases=d_table, n=n_idx, base_index=d_idx, table_len=rows * cols, table=(c, windows)))
    want.append(np.asarray(oracle.g1_msm(np.ascontiguousarray(coeffs[:32 * n_idx]), np.ascontiguousarray(table.reshape(-1, 96)[idx].reshape(-1)))))
    # a plain job over level 0 of the same buffer next to them (small commits keep the multi-window path)
    jobs.append(dict(scalars=d_coeffs, bases=d_table, n=300, base_view=(10, cols), scalar_view=(10, ys), table_len=rows * cols))
    want.append(np.asarray(oracle.g1_msm(np.ascontiguousarray(coeffs.reshape(xs, ys, 32)[:30, :10].reshape(-1)),
                                         np.ascontiguousarray(table.reshape(rows, cols, 96)[:30, :10].reshape(-1)))))
    got = _aff(tk, tk.msm_multi_ex(jobs, bases_form=tk.BASES_CONVERTED))
    for k, w in enumerate(want):
        assert (got[96 * k:96 * (k + 1)] == w).all(), (c, k)
    if c > 16:      # too small for a wide window: refused, not silently narrowed
        with pytest.raises(tk.TkmkError):
            tk.msm_multi_ex([dict(scalars=d_coeffs, bases=d_table, n=1000, table_len=rows * cols, table=(c, windows))], bases_form=tk.BASES_CONVERTED)


def test_two_contexts_prove_concurrently_in_one_process(gpu, oracle, tmp_path):
    """include/tkmk.h THREADING: two UNSHARDED contexts of one process take turns on the default stream behind the prover library's lock —
    two threads proving two different circuits at the same time get, byte for byte, the proofs they get one after the other"""
    import threading
    import synth_circuit
    from test_gpu_prove import _stage_crs_file, seeded_mixer
    from tkmk import service
    jobs = []
    for k, (seed, shape) in enumerate([(91, dict(s_max=8, n_gate_kinds=2)), (92, dict(s_max=16, n_gate_kinds=3, n_out=2, n_in=3, n_prv=9, used_placements=11))]):
        d = tmp_path / ("c%d" % k)
        d.mkdir()
        inst = synth_circuit.build(str(d), random.Random(seed), **shape)
        _stage_crs_file(gpu, oracle, inst, str(d / "crs"))
        jobs.append((inst, str(d / "crs"), _mixer_file(d, seeded_mixer(seed))))
    provers = [service.Prover(inst["qap"], crs, testing=True) for inst, crs, _ in jobs]
    try:
        alone = [p.prove(inst["synth"], None, testing_mixer_json=m)[0] for p, (inst, _, m) in zip(provers, jobs)]
        together, errors = [None, None], []

        def body(k):
            try:
                for _ in range(3):
                    together[k] = provers[k].prove(jobs[k][0]["synth"], None, testing_mixer_json=jobs[k][2])[0]
            except BaseException as e:      # noqa: BLE001
                errors.append(e)
        threads = [threading.Thread(target=body, args=(k,)) for k in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        assert together == alone
    finally:
        for p in provers:
            p.close()


def test_helper_thread_pipeline_width_and_profiling_do_not_change_a_byte(gpu, oracle, tmp_path, monkeypatch):
    """the binding batch on its helper thread (TKMK_PROVER_ASYNC_BINDING, default on) against in line, three pipeline streams against one,
    the event profiler on against off: one and the same proof.json for the same blinding scalars"""
    import synth_circuit
    from test_gpu_prove import _stage_crs_file, seeded_mixer
    from tkmk import service
    inst = synth_circuit.build(str(tmp_path), random.Random(93), s_max=16, n_gate_kinds=3, n_out=2, n_in=3, n_prv=9, used_placements=13, bit_fraction=0.5)
    crs_dir = str(tmp_path / "crs")
    _stage_crs_file(gpu, oracle, inst, crs_dir)
    mixer_path = _mixer_file(tmp_path, seeded_mixer(93))
    monkeypatch.setenv("TKMK_PROVER_TABLE_C", "12")
    docs = {}
    with service.Prover(inst["qap"], crs_dir, testing=True) as p:
        docs["default"] = p.prove(inst["synth"], None, testing_mixer_json=mixer_path)[0]
        gpu.profile_enable(True)
        docs["profiling"] = p.prove(inst["synth"], None, testing_mixer_json=mixer_path)[0]
        gpu.profile_enable(False)
        gpu.msm_set_pipeline_streams(1)               # also turns the helper thread off for this proof (the serialised profiling pass)
        docs["one stream"] = p.prove(inst["synth"], None, testing_mixer_json=mixer_path)[0]
        gpu.msm_set_pipeline_streams(0)
    # the switch is read once per process: a second process with the helper thread off
    import subprocess
    import sys
    code = ("import sys, json; sys.path[:0] = %r\nfrom tkmk import service\nimport tkmk\ntkmk.set_device(0)\n"
            "p = service.Prover(%r, %r, testing=True)\nprint(json.dumps(p.prove(%r, None, testing_mixer_json=%r)[0]))\n") % (
        [os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "tokamak-zk-evm_amd")], inst["qap"], crs_dir, inst["synth"], mixer_path)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, TKMK_PROVER_ASYNC_BINDING="0"))
    assert r.returncode == 0, r.stderr[-2000:]
    docs["binding in line"] = json.loads(r.stdout.strip().splitlines()[-1])
    for k, d in docs.items():
        assert d == docs["default"], k
