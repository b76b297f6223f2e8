"""gpu tier: ONE proof over G GPUs through the native prover (tkmk_prover_open_sharded, include/tkmk_prover.h; host/tkmk_service.hpp
ShardLink) — executed with G = 2 and 4 VIRTUAL ranks over the loopback transport of libtkmk_dist.so
(one host thread per rank on the one GPU of the test box; the RCCL transport differs below `transport_all_gather` only).

Every rank holds the grid rows ix = r mod G of the commit tables (xy_powers, Lagrange-basis tables, their prefix sums) and commits its
rows of every polynomial; the partial commitments meet in one all-gather per round.  Checked:
  * every rank's proof.json is BYTE-EQUAL to the single-GPU resident prover's for the same blinding scalars — with the precomputed
    commit table and the evaluation-basis commitments forced on at a small shape (TKMK_PROVER_TABLE_C), with them off, and on the
    reference's commit list (TEST_PARTS | COEFFICIENT_BASIS);
  * without fixed blinding scalars rank 0's getrandom() draw is broadcast: all ranks return one and the same proof, it verifies from
    the files with real pairings and is rejected for a changed public input;
  * a rank's failure (corrupted witness document) is an error on every rank, and the contexts stay usable.
Unmeasured on multi-GPU hardware (no SCALE run so far); `bench.py --gpus N --one-proof` is the entry for it."""
import json
import os
import random
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SHAPES = [(61, dict(s_max=8, n_gate_kinds=2)),
          (62, dict(s_max=16, n_gate_kinds=3, n_out=2, n_in=3, n_prv=9, used_placements=11, k_out=0, k_pub=3, bit_fraction=0.5))]


def _mixer_file(tmp_path, mixer):
    hx = lambda v: [hx(e) for e in v] if isinstance(v, list) else "0x%x" % v          # noqa: E731
    path = str(tmp_path / "mixer.json")
    json.dump({k: hx(v) for k, v in mixer.items()}, open(path, "w"))
    return path


def _open_ranks(dist, service, comms, qap, crs_dir):
    return dist.run_ranks(comms, lambda c: service.Prover(qap, crs_dir, testing=True, comm=c))


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("seed,shape", SHAPES)
@pytest.mark.parametrize("table_c", ["12", "0"])
def test_sharded_proof_equals_the_single_gpu_proof(gpu, oracle, tmp_path, monkeypatch, world, seed, shape, table_c):
    import synth_circuit
    from test_gpu_prove import _stage_crs_file, seeded_mixer
    from tkmk import dist, service
    inst = synth_circuit.build(str(tmp_path), random.Random(seed), **shape)
    sp = inst["setup_params"]
    if world > min(sp["n"], sp["l_D"] - sp["l"], sp["s_max"]):
        pytest.skip("more ranks than the circuit has rows / columns")
    crs_dir = str(tmp_path / "crs")
    _stage_crs_file(gpu, oracle, inst, crs_dir)
    mixer_path = _mixer_file(tmp_path, seeded_mixer(seed))
    monkeypatch.setenv("TKMK_PROVER_TABLE_C", table_c)          # 12: commit table + Lagrange-basis tables at this small shape; 0: neither
    with service.Prover(inst["qap"], crs_dir, testing=True) as single:
        want, _ = single.prove(inst["synth"], None, testing_mixer_json=mixer_path)
    comms = dist.loopback_comms(world)
    provers = _open_ranks(dist, service, comms, inst["qap"], crs_dir)
    try:
        assert [service.lib(True).tkmk_prover_world_size(p._h) for p in provers] == [world] * world
        by_rank = {p.comm.rank: p for p in provers}
        for rep in range(2):                                    # the contexts are reusable
            docs = dist.run_ranks(comms, lambda c: by_rank[c.rank].prove(inst["synth"], str(tmp_path / ("out%d" % c.rank)), testing_mixer_json=mixer_path)[0])
            for r in range(world):
                assert docs[r] == want, (r, rep)
            assert json.load(open(tmp_path / "out0" / "proof.json")) == want
        # the reference's commit list (parts of Pi one by one, coefficient-basis commitments): same bytes, and the boxes every rank
        # reports are the whole polynomial's, not its share
        res = dist.run_ranks(comms, lambda c: by_rank[c.rank].prove(inst["synth"], None, testing_mixer_json=mixer_path, test_parts=True,
                                                                    coefficient_basis=True, want_boxes=True))
        for r in range(world):
            assert res[r][0] == want and res[r][2] == res[0][2]
        assert {b["name"] for b in res[0][2]} >= {"U", "Pi_AX", "Pi_CX", "Pi_B", "M_X", "N_X"}
    finally:
        for p in provers:
            p.close()
        for c in comms:
            c.close()


def test_sharded_proof_with_fresh_blinding_verifies_and_failures_reach_every_rank(gpu, oracle, tmp_path, monkeypatch):
    import synth_circuit
    import verify_files
    from test_gpu_prove import _stage_crs_file
    from tkmk import dist, service
    world = 4
    inst = synth_circuit.build(str(tmp_path), random.Random(67), s_max=8, n_gate_kinds=2, used_placements=7, bit_fraction=0.4)
    crs_dir = str(tmp_path / "crs")
    _stage_crs_file(gpu, oracle, inst, crs_dir)
    monkeypatch.setenv("TKMK_PROVER_TABLE_C", "12")
    comms = dist.loopback_comms(world)
    provers = _open_ranks(dist, service, comms, inst["qap"], crs_dir)
    by_rank = {p.comm.rank: p for p in provers}
    try:
        out = str(tmp_path / "out")
        docs = dist.run_ranks(comms, lambda c: by_rank[c.rank].prove(inst["synth"], out)[0])          # every rank names the directory: rank 0 writes
        assert docs[1] == docs[0] and docs[2] == docs[0] and docs[3] == docs[0]        # rank 0's blinding scalars reached every rank
        again = dist.run_ranks(comms, lambda c: by_rank[c.rank].prove(inst["synth"], None)[0])
        assert again[0] != docs[0]                              # and they are fresh per proof
        assert verify_files.verify(inst["qap"], inst["synth"], crs_dir, out)
        assert not verify_files.verify(inst["qap"], inst["synth"], crs_dir, out, tamper_public_input=True)
        # every rank reads the same files: a corrupted document fails all of them before any collective, and the contexts live on
        pv = os.path.join(inst["synth"], "placementVariables.json")
        good = open(pv).read()
        open(pv, "w").write(good[:len(good) // 2])

        def failing(c):
            with pytest.raises(service.ProverError):
                by_rank[c.rank].prove(inst["synth"], None)
            return True
        assert dist.run_ranks(comms, failing) == [True] * world
        open(pv, "w").write(good)
        docs = dist.run_ranks(comms, lambda c: by_rank[c.rank].prove(inst["synth"], out if c.rank == 0 else None)[0])
        assert docs[1] == docs[0]
        assert verify_files.verify(inst["qap"], inst["synth"], crs_dir, out)
    finally:
        for p in provers:
            p.close()
        for c in comms:
            c.close()


def test_shard_column_arithmetic():
    """Shard::cols_of (host/tkmk_host.hpp) restated: |{iy < total : iy = r mod G}|; the shares of any box add up to the box"""
    for world in (1, 2, 3, 4, 8):
        for total in (0, 1, 2, 5, 8, 4097, 8191):
            shares = [len(range(r, total, world)) for r in range(world)]
            assert sum(shares) == total
            assert shares == [(total - r + world - 1) // world if total > r else 0 for r in range(world)]


def test_production_shape_sharded_over_two_virtual_ranks(gpu):
    """the production shape (n = 4096, m_I = 4096, s_max = 256, 166 placements) with G = 2: the shares are large enough for the
    precomputed commit table (20-bit windows) and the evaluation-basis commits to run on every rank, as on real hardware; the two ranks'
    proofs equal the single-GPU context's byte for byte, and each context's resident tables are half the single one's"""
    import shutil
    import prove_bench
    from tkmk import dist, service
    from tkmk.prove import random_mixer
    files = prove_bench.stage_files(s_max=256, placements=166)
    try:
        mixer = random_mixer(random.Random(2027))
        hx = lambda v: [hx(e) for e in v] if isinstance(v, list) else "0x%x" % v          # noqa: E731
        mixer_path = os.path.join(files["tmp"], "mixer.json")
        json.dump({k: hx(v) for k, v in mixer.items()}, open(mixer_path, "w"))
        free0 = gpu.available_memory()[1]
        with service.Prover(files["qap"], files["crs"], testing=True) as single:
            used_single = free0 - gpu.available_memory()[1]
            want, _, boxes = single.prove(files["synth"], None, testing_mixer_json=mixer_path, want_boxes=True)
        gpu.release_scratch()
        world = 2
        comms = dist.loopback_comms(world)
        free1 = gpu.available_memory()[1]
        provers = _open_ranks(dist, service, comms, files["qap"], files["crs"])
        used_sharded_total = free1 - gpu.available_memory()[1]
        by_rank = {p.comm.rank: p for p in provers}
        try:
            res = dist.run_ranks(comms, lambda c: by_rank[c.rank].prove(files["synth"], None, testing_mixer_json=mixer_path, want_boxes=True))
            for r in range(world):
                assert res[r][0] == want, r
                assert res[r][2] == boxes                      # same commit structure (evaluation-basis U, V, W, B, R; Pi_X once)
            # both ranks' resident state together stays near the single context's (tables halved; the binding tables, the subcircuit
            # library and the allocator's cached blocks are per context)
            assert used_sharded_total < 1.6 * used_single, (used_sharded_total, used_single)
        finally:
            for p in provers:
                p.close()
            for c in comms:
                c.close()
    finally:
        shutil.rmtree(files["tmp"], ignore_errors=True)
        gpu.release_scratch()


def test_configs3_sharded_over_four_and_eight_virtual_ranks(gpu):
    """BASELINE.json configs[3] (n = 4096, m_I = 4096, s_max = 1024, 1024 placements: the bench's workload) through the sharded prover with G = 4
    (20-bit table windows) and G = 8 (16-bit windows): every rank's proof equals the single-GPU context's byte for byte
    for the same blinding scalars — at the size the headline is measured on, not only at the small shapes above"""
    import shutil
    import prove_bench
    from tkmk import dist, service
    from tkmk.prove import random_mixer
    files = prove_bench.stage_files(s_max=1024)
    try:
        mixer = random_mixer(random.Random(2028))
        hx = lambda v: [hx(e) for e in v] if isinstance(v, list) else "0x%x" % v          # noqa: E731
        mixer_path = os.path.join(files["tmp"], "mixer.json")
        json.dump({k: hx(v) for k, v in mixer.items()}, open(mixer_path, "w"))
        with service.Prover(files["qap"], files["crs"], testing=True) as single:
            want, _, boxes = single.prove(files["synth"], None, testing_mixer_json=mixer_path, want_boxes=True)
        gpu.release_scratch()
        for world in (4, 8):
            comms = dist.loopback_comms(world)
            provers = _open_ranks(dist, service, comms, files["qap"], files["crs"])
            by_rank = {p.comm.rank: p for p in provers}
            try:
                res = dist.run_ranks(comms, lambda c: by_rank[c.rank].prove(files["synth"], None, testing_mixer_json=mixer_path, want_boxes=True))
                for r in range(world):
                    assert res[r][0] == want, (world, r)
                    assert res[r][2] == boxes, (world, r)
            finally:
                for p in provers:
                    p.close()
                for c in comms:
                    c.close()
                gpu.release_scratch()
    finally:
        shutil.rmtree(files["tmp"], ignore_errors=True)
        gpu.release_scratch()


def test_an_input_error_only_one_rank_can_see_is_every_ranks_error(gpu, oracle, tmp_path, monkeypatch):
    """a rank converts only its own placements' values, so a bad hex digit inside placement 1 is seen by rank 1 alone — the ranks agree on
    the outcome of the input phase before any of them goes on: EVERY rank reports the error (none proves on with a peer missing), the
    communicator is not torn down, and the next proof on the repaired document is the single-GPU proof again"""
    import synth_circuit
    from test_gpu_prove import _stage_crs_file, seeded_mixer
    from tkmk import dist, service
    world = 4
    inst = synth_circuit.build(str(tmp_path), random.Random(68), s_max=8, n_gate_kinds=2, used_placements=8)
    crs_dir = str(tmp_path / "crs")
    _stage_crs_file(gpu, oracle, inst, crs_dir)
    mixer_path = _mixer_file(tmp_path, seeded_mixer(68))
    with service.Prover(inst["qap"], crs_dir, testing=True) as single:
        want, _ = single.prove(inst["synth"], None, testing_mixer_json=mixer_path)
    comms = dist.loopback_comms(world)
    provers = _open_ranks(dist, service, comms, inst["qap"], crs_dir)
    by_rank = {p.comm.rank: p for p in provers}
    pv_path = os.path.join(inst["synth"], "placementVariables.json")
    pv = json.load(open(pv_path))
    try:
        bad = json.loads(json.dumps(pv))
        bad[1]["variables"][2] = "0xzz"                         # placement 1 belongs to rank 1 of 4
        json.dump(bad, open(pv_path, "w"))

        def attempt(c):
            try:
                by_rank[c.rank].prove(inst["synth"], None, testing_mixer_json=mixer_path)
                return "proved"
            except service.ProverError as e:
                return str(e)
        res = dist.run_ranks(comms, attempt)
        assert all(r != "proved" for r in res), res
        assert "invalid hex digit" in res[1] and all("another rank" in res[r] for r in (0, 2, 3)), res
        json.dump(pv, open(pv_path, "w"))
        docs = dist.run_ranks(comms, lambda c: by_rank[c.rank].prove(inst["synth"], None, testing_mixer_json=mixer_path)[0])
        assert all(d == want for d in docs)
    finally:
        for p in provers:
            p.close()
        for c in comms:
            c.close()


def test_the_ranks_divide_the_polynomial_work(gpu, oracle, tmp_path, monkeypatch):
    """the library's own work counters over one proof: ALL G virtual ranks together (they share this process's counters) transform and
    stream about what the single-GPU context does alone — each rank 1 / G of it — where the round-3 prover replicated both on every rank
    (G times the single figure).  Counting conventions: a bivariate call counts its elements once, the sharded transform's two 1-D
    passes once each (so 2 x for the transforms); the streaming passes (poly.elements) count the elements they touch."""
    import synth_circuit
    from test_gpu_prove import _stage_crs_file, seeded_mixer
    from tkmk import dist, service
    world = 4
    inst = synth_circuit.build(str(tmp_path), random.Random(69), s_max=32, n_gate_kinds=4, n_out=6, n_in=10, n_prv=40, k_out=2, k_pub=3, l_free=8, l_extra=3)
    crs_dir = str(tmp_path / "crs")
    _stage_crs_file(gpu, oracle, inst, crs_dir)
    mixer_path = _mixer_file(tmp_path, seeded_mixer(69))
    monkeypatch.setenv("TKMK_PROVER_TABLE_C", "0")
    with service.Prover(inst["qap"], crs_dir, testing=True) as single:
        single.prove(inst["synth"], None, testing_mixer_json=mixer_path)
        gpu.native_stats_reset()
        want, _ = single.prove(inst["synth"], None, testing_mixer_json=mixer_path)
        one = gpu.native_stats()
    comms = dist.loopback_comms(world)
    provers = _open_ranks(dist, service, comms, inst["qap"], crs_dir)
    by_rank = {p.comm.rank: p for p in provers}
    try:
        dist.run_ranks(comms, lambda c: by_rank[c.rank].prove(inst["synth"], None, testing_mixer_json=mixer_path))
        gpu.native_stats_reset()
        docs = dist.run_ranks(comms, lambda c: by_rank[c.rank].prove(inst["synth"], None, testing_mixer_json=mixer_path)[0])
        allr = gpu.native_stats()
        assert all(d == want for d in docs)
        per_rank = {k: v / world for k, v in allr.items()}
        assert per_rank["ntt.elements"] <= 2.3 * one["ntt.elements"] / world, (per_rank, one)
        assert per_rank["poly.elements"] <= 1.6 * one["poly.elements"] / world, (per_rank, one)
        assert per_rank["msm.points"] <= 1.3 * one["msm.points"] / world + 4096, (per_rank, one)
    finally:
        for p in provers:
            p.close()
        for c in comms:
            c.close()


def test_fuzz_shapes_native_and_sharded_against_the_restatement():
    """tools/prove_fuzz.py over 16 seeded random circuit shapes (s_max 8 / 16, 0-2 output and 1-3 public buffers, 2-6 gate kinds, m_I 16-64):
    the Python prover and the native binary against the big-int restatement (every point, evaluation and challenge; the verifier equation on
    the discrete logarithms), and the same files through tkmk_prover_open_sharded with 2, 4 or 8 virtual ranks, with and without the commit /
    Lagrange tables — every rank's proof.json the binary's byte for byte.  (The tool's longer runs are recorded under profiles/.)"""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prove_fuzz.py"), "16", "8100", "sharded"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.strip().splitlines()[-1] == "all 16 cases equal the restatement and verify", r.stdout[-1500:]
    assert r.stdout.count(" ok ") == 16
