"""gpu tier: libtkmk_dist.so (include/tkmk_dist.h) — the sharded MSM (one MSM, and the commit batch of a prover round over row-sharded
tables) and the slab-sharded bivariate NTT.
RCCL refuses two ranks on one device, so a one-GPU box forms a ONE-rank RCCL communicator: that drives every RCCL call of the path
(ncclCommInitRank, ncclAllGather, ncclAllToAll on device buffers).  The G >= 2 code of the same entry points — pack / place of the
transpose, column batches of width y_size / G, gather + device-side sum of partials, empty and infinite partials — runs through the
LOOPBACK transport: G = 2, 4, 8 virtual ranks in this process, one host thread each, collectives as device copies.  Multi-GPU
hardware itself: bench.py --gpus N --msm-sharded (unmeasured so far: no SCALE run)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_one_rank_communicator_msm_and_bintt(gpu, oracle):
    from tkmk import dist
    comm = dist.Comm(dist.unique_id(), 1, 0)
    n = 3000
    s, p = oracle.fr_random(21, n), oracle.g1_random_bases(22, n)
    want = np.asarray(oracle.g1_msm(s, p))
    assert (gpu.projective_to_affine_bytes(comm.msm_sharded(s, p)) == want).all()
    ds, dp = gpu.DeviceBuffer.from_host(np.asarray(s)), gpu.DeviceBuffer.from_host(np.asarray(p))
    assert (gpu.projective_to_affine_bytes(comm.msm_sharded(ds, dp)) == want).all()
    xs, ys = 64, 32
    gpu.init_ntt_domain_for_size(xs * ys)
    m = oracle.fr_random(23, xs * ys)
    cx, cy = oracle.fr_random(24, 1), oracle.fr_random(25, 1)
    for inverse in (False, True):
        slab = gpu.DeviceBuffer.from_host(np.asarray(m))
        got = np.asarray(comm.bintt_sharded(slab, xs, ys, inverse=inverse, coset_x=cx, coset_y=cy).to_host())
        assert (got == np.asarray(oracle.bintt(m, xs, ys, inverse=inverse, coset_x=cx, coset_y=cy))).all(), inverse
    with pytest.raises(dist.DistError):
        comm.bintt_sharded(gpu.DeviceBuffer(32 * 24), 6, 4)          # not powers of two
    comm.close()


# ---- G >= 2 through the LOOPBACK transport: the same entry points, virtual ranks on one GPU (include/tkmk_dist.h) ----
@pytest.mark.parametrize("world", [2, 4, 8])
def test_loopback_msm_sharded(gpu, oracle, world):
    """tkmk_msm_sharded with G = 2, 4, 8: ragged shards incl. a rank with msm_size = 0 and a rank whose partial result is the point at
    infinity; every rank must return the MSM over ALL points (the oracle's), from host operands and from device operands"""
    from tkmk import dist
    n = 5000
    s, p = np.asarray(oracle.fr_random(31, n)).reshape(n, 32).copy(), np.asarray(oracle.g1_random_bases(32, n)).reshape(n, 96).copy()
    cuts = sorted(np.random.default_rng(world).choice(np.arange(1, n), world - 1, replace=False).tolist())
    bounds = [0] + cuts + [n]
    bounds[1] = bounds[0]                                   # rank 0: no points at all
    lo, hi = bounds[-2], bounds[-1]                         # last rank: points that cancel -> its partial is infinity
    if world > 2:
        half = (hi - lo) // 2
        hi = lo + 2 * half
        p[lo + half:hi] = p[lo:lo + half]
        s[lo + half:hi] = np.asarray(oracle.to_bytes([(oracle.R_MOD - v) % oracle.R_MOD for v in oracle.to_ints(s[lo:lo + half].reshape(-1), 32)], 32)).reshape(half, 32)
        s, p = s[:hi], p[:hi]
        bounds[-1] = hi
    want = np.asarray(oracle.g1_msm(np.ascontiguousarray(s.reshape(-1)), np.ascontiguousarray(p.reshape(-1))))
    shards = [(np.ascontiguousarray(s[bounds[r]:bounds[r + 1]].reshape(-1)), np.ascontiguousarray(p[bounds[r]:bounds[r + 1]].reshape(-1))) for r in range(world)]
    comms = dist.loopback_comms(world)
    assert [c.rank for c in comms] == list(range(world)) and all(dist.lib().tkmk_comm_is_loopback(c.handle) == 1 for c in comms)
    got = dist.run_ranks(comms, lambda c: c.msm_sharded(*shards[c.rank], n=shards[c.rank][0].size // 32))
    for r in range(world):
        assert (gpu.projective_to_affine_bytes(got[r]) == want).all(), r
    dev = [(gpu.DeviceBuffer.from_host(a) if a.size else gpu.DeviceBuffer(32), gpu.DeviceBuffer.from_host(b) if b.size else gpu.DeviceBuffer(96)) for a, b in shards]
    got = dist.run_ranks(comms, lambda c: c.msm_sharded(*dev[c.rank], n=shards[c.rank][0].size // 32))
    for r in range(world):
        assert (gpu.projective_to_affine_bytes(got[r]) == want).all(), r
    for c in comms:
        c.close()


@pytest.mark.parametrize("world,xs,ys", [(2, 64, 32), (4, 32, 64), (8, 128, 16)])
def test_loopback_bintt_sharded(gpu, oracle, world, xs, ys):
    """tkmk_bintt_sharded with G = 2, 4, 8: x-slabs in, y-slabs out; the stitched y-slabs must be tkmk_bintt / the oracle's bivariate
    NTT of the whole matrix — forward and inverse, with cosets: the pack loop, the all-to-all's receive placement and the column batch
    of width y_size / G"""
    from tkmk import dist
    gpu.init_ntt_domain_for_size(xs * ys)
    m = np.asarray(oracle.fr_random(40 + world, xs * ys))
    cx, cy = oracle.fr_random(41, 1), oracle.fr_random(42, 1)
    comms = dist.loopback_comms(world)
    rows, cols = xs // world, ys // world
    for inverse in (False, True):
        for cosets in ((None, None), (cx, cy)):
            slabs = [gpu.DeviceBuffer.from_host(np.ascontiguousarray(m[32 * ys * rows * r:32 * ys * rows * (r + 1)])) for r in range(world)]
            outs = [gpu.DeviceBuffer(32 * xs * cols) for _ in range(world)]
            dist.run_ranks(comms, lambda c: c.bintt_sharded(slabs[c.rank], xs, ys, inverse=inverse, coset_x=cosets[0], coset_y=cosets[1], out=outs[c.rank]))
            stitched = np.empty((xs, ys, 32), np.uint8)
            for r in range(world):
                stitched[:, cols * r:cols * (r + 1), :] = np.asarray(outs[r].to_host()).reshape(xs, cols, 32)
            want = np.asarray(oracle.bintt(m, xs, ys, inverse=inverse, coset_x=cosets[0], coset_y=cosets[1]))
            assert (stitched.reshape(-1) == want).all(), (inverse, cosets[0] is not None)
            single = np.asarray(gpu.bintt(m, xs, ys, inverse=inverse, coset_x=cosets[0], coset_y=cosets[1]))
            assert (stitched.reshape(-1) == single).all()
    for c in comms:
        c.close()


@pytest.mark.parametrize("world", [2, 4])
def test_loopback_msm_multi_ex_sharded_row_interleaved_tables(gpu, oracle, world):
    """tkmk_msm_multi_ex_sharded: the commit batch of a prover round over a ROW-INTERLEAVED table (rank r holds grid rows ix = r mod G):
    every job is a box of the replicated coefficient matrix against the matching box of the CRS grid, a rank's share = its rows of the
    box through strided views.  One all-gather for the whole batch; every rank gets every commitment = the oracle's MSM over the box.
    Jobs: a wide box, a box with fewer rows than ranks (empty on some ranks), a one-row box, an index-list job present on rank 0 only."""
    from tkmk import dist
    rs_x, rs_y = 24, 16                                      # CRS grid; coefficient matrices are 32 x 16 with stride 16
    table = np.asarray(oracle.g1_random_bases(70 + world, rs_x * rs_y)).reshape(rs_x, rs_y, 96)
    cx, cy = 32, 16
    coeffs = np.asarray(oracle.fr_random(71, cx * cy)).reshape(cx, cy, 32)
    d_coeffs = gpu.DeviceBuffer.from_host(np.ascontiguousarray(coeffs.reshape(-1)))      # replicated on every rank
    local_tables = [gpu.msm_convert_bases(gpu.DeviceBuffer.from_host(np.ascontiguousarray(table[r::world].reshape(-1)))) for r in range(world)]
    local_rows = [len(range(r, rs_x, world)) for r in range(world)]
    boxes = [(21, 13), (world - 1, 16), (1, 7)]
    idx = np.array([5, 0, 17, 17, 3], np.uint32)
    d_idx = gpu.DeviceBuffer.from_host(idx.view(np.uint8))
    full_conv = gpu.msm_convert_bases(gpu.DeviceBuffer.from_host(np.ascontiguousarray(table.reshape(-1))))

    def jobs_of(r):
        jobs = []
        for tx, ty in boxes:
            mine = len(range(r, tx, world))                  # grid rows r, r + G, ... < tx
            jobs.append(dict(scalars=d_coeffs, scalar_offset=32 * cy * r, bases=local_tables[r], n=mine * ty, scalar_view=(ty, cy * world),
                             base_view=(ty, rs_y), table_len=local_rows[r] * rs_y))
        jobs.append(dict(scalars=d_coeffs, bases=full_conv, n=idx.size if r == 0 else 0, base_index=d_idx, table_len=rs_x * rs_y))
        return jobs
    comms = dist.loopback_comms(world)
    got = dist.run_ranks(comms, lambda c: c.msm_multi_ex_sharded(jobs_of(c.rank)))
    want = [np.asarray(oracle.g1_msm(np.ascontiguousarray(coeffs[:tx, :ty].reshape(-1)), np.ascontiguousarray(table[:tx, :ty].reshape(-1)))) for tx, ty in boxes]
    want.append(np.asarray(oracle.g1_msm(np.ascontiguousarray(coeffs.reshape(-1, 32)[:idx.size].reshape(-1)), np.ascontiguousarray(table.reshape(-1, 96)[idx].reshape(-1)))))
    for r in range(world):
        aff = gpu.projective_to_affine_bytes(got[r])
        for k, w in enumerate(want):
            assert (aff[96 * k:96 * (k + 1)] == w).all(), (r, k)
    for c in comms:
        c.close()


def test_loopback_rendezvous_times_out_instead_of_hanging(gpu, oracle, monkeypatch):
    """a rank that never reaches the collective (it failed earlier) must not hang its peers: the loopback rendezvous gives up after
    TKMK_LOOPBACK_TIMEOUT_S and the waiting rank gets an error; the group is then unusable and says so at once"""
    import time
    from tkmk import dist
    monkeypatch.setenv("TKMK_LOOPBACK_TIMEOUT_S", "2")
    comms = dist.loopback_comms(2)
    s, p = oracle.fr_random(91, 8), oracle.g1_random_bases(92, 8)

    def body(c):
        if c.rank == 1:
            return "absent"                                  # never calls the collective entry
        t0 = time.time()
        with pytest.raises(dist.DistError) as e:
            c.msm_sharded(s, p)
        assert "did not reach the collective" in str(e.value) and 1.5 < time.time() - t0 < 30
        t0 = time.time()
        with pytest.raises(dist.DistError):
            c.msm_sharded(s, p)                              # broken group: no second wait
        assert time.time() - t0 < 1.5
        return "timed out"
    assert dist.run_ranks(comms, body) == ["timed out", "absent"]
    for c in comms:
        c.close()
