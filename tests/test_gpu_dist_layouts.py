"""gpu tier: the sharded prover's two matrix layouts and what moves between them (include/tkmk_dist.h: COLS = rank r holds the columns
iy = r mod G, ROWS = contiguous row slabs), through the LOOPBACK transport with G = 2, 4, 8 virtual ranks on the one GPU of the test box:

  tkmk_dist_fwd_cols_to_rows / tkmk_dist_inv_rows_to_cols   vs the oracle's bivariate transform of the WHOLE matrix (zero-padded forward,
                                                            inverse, either pass skipped, both skipped = a pure change of layout)
  tkmk_dist_rows_rotate                                     vs numpy.roll of the whole matrix
  tkmk_comm_ring_shift, tkmk_comm_all_gather_host           vs the obvious
  tkmk_comm_agree and the status block of the sharded MSM    ONE rank's refused job is an error on EVERY rank — no rank returns a sum
                                                            that lacks a share (the one-sided case; all-ranks-fail is covered elsewhere)
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cols_of(m, G, r):               # m: (x, y, 32) -> rank r's COLS part, contiguous
    return np.ascontiguousarray(m[:, r::G, :])


def _rows_of(m, G, r):
    h = m.shape[0] // G
    return np.ascontiguousarray(m[r * h:(r + 1) * h])


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("shape", [(16, 8, 16, 8), (8, 8, 32, 16), (32, 16, 32, 64), (64, 32, 64, 32)])   # in_x, in_y, x_size, y_size
def test_forward_and_inverse_transforms_between_the_layouts(gpu, oracle, world, shape):
    from tkmk import dist
    in_x, in_y, xs, ys = shape
    if in_y < world or xs < world:
        pytest.skip("fewer columns / rows than ranks")
    gpu.init_ntt_domain_for_size(max(xs, ys) * 4)
    coeff = np.asarray(oracle.fr_random(100 + in_x + ys, in_x * in_y)).reshape(in_x, in_y, 32)
    padded = np.zeros((xs, ys, 32), np.uint8)
    padded[:in_x, :in_y] = coeff
    flat = lambda a: np.ascontiguousarray(a.reshape(-1))                                    # noqa: E731
    want_fwd = np.asarray(oracle.bintt(flat(padded), xs, ys)).reshape(xs, ys, 32)
    comms = dist.loopback_comms(world)
    try:
        def forward(c, flags=0):
            d = gpu.DeviceBuffer.from_host(flat(_cols_of(coeff, world, c.rank)))
            return np.asarray(c.fwd_cols_to_rows(d, in_x, in_y, xs, ys, flags).to_host()).reshape(xs // world, ys, 32)
        got = dist.run_ranks(comms, forward)
        for r in range(world):
            assert (got[r] == _rows_of(want_fwd, world, r)).all(), ("forward", r)
        # both passes skipped: the padded matrix itself, rows instead of columns
        got = dist.run_ranks(comms, lambda c: forward(c, dist.SKIP_X_PASS | dist.SKIP_Y_PASS))
        for r in range(world):
            assert (got[r] == _rows_of(padded, world, r)).all(), ("relayout", r)
        # one pass only, then the other on the result through the inverse entry's skip flags: X forward, then (rows -> cols, nothing) and
        # a full inverse of the forward result must give the padded coefficients back in the COLS layout
        def back(c):
            d = gpu.DeviceBuffer.from_host(flat(_rows_of(want_fwd, world, c.rank)))
            return np.asarray(c.inv_rows_to_cols(d, xs, ys).to_host()).reshape(xs, ys // world, 32)
        got = dist.run_ranks(comms, back)
        for r in range(world):
            assert (got[r] == _cols_of(padded, world, r)).all(), ("inverse", r)
        def relayout_back(c):
            d = gpu.DeviceBuffer.from_host(flat(_rows_of(want_fwd, world, c.rank)))
            return np.asarray(c.inv_rows_to_cols(d, xs, ys, dist.SKIP_X_PASS | dist.SKIP_Y_PASS).to_host()).reshape(xs, ys // world, 32)
        got = dist.run_ranks(comms, relayout_back)
        for r in range(world):
            assert (got[r] == _cols_of(want_fwd, world, r)).all(), ("relayout back", r)
        # Y pass only (forward) == the oracle's row transforms of the padded matrix
        rows_only = np.asarray(oracle.ntt(flat(padded), ys, batch=xs)).reshape(xs, ys, 32)
        got = dist.run_ranks(comms, lambda c: forward(c, dist.SKIP_X_PASS))
        for r in range(world):
            assert (got[r] == _rows_of(rows_only, world, r)).all(), ("Y pass only", r)
    finally:
        for c in comms:
            c.close()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_rows_rotate_ring_shift_and_host_gather(gpu, oracle, world):
    from tkmk import dist
    xs, ys = 8 * world, 16
    m = np.asarray(oracle.fr_random(7, xs * ys)).reshape(xs, ys, 32)
    h = xs // world
    comms = dist.loopback_comms(world)
    try:
        for rot in (0, 1, 2, h):
            got = dist.run_ranks(comms, lambda c: np.asarray(c.rows_rotate(gpu.DeviceBuffer.from_host(np.ascontiguousarray(_rows_of(m, world, c.rank).reshape(-1))), h, ys, rot).to_host()))
            want = np.roll(m, rot, axis=0)
            for r in range(world):
                assert (got[r].reshape(h, ys, 32) == _rows_of(want, world, r)).all(), (rot, r)
        with pytest.raises(dist.DistError):
            dist.run_ranks(comms, lambda c: c.rows_rotate(gpu.DeviceBuffer(32 * h * ys), h, ys, h + 1))

        def shift(c, d):
            import ctypes
            send = gpu.DeviceBuffer.from_host(np.full(64, c.rank, np.uint8))           # (kept alive across the call)
            out = gpu.DeviceBuffer(64)
            dist._check(dist.lib().tkmk_comm_ring_shift(c.handle, gpu._p(send), ctypes.c_size_t(64), d, gpu._p(out)), "tkmk_comm_ring_shift")
            return int(np.asarray(out.to_host())[0])
        for d in (1, 3, world, world + 1):
            assert dist.run_ranks(comms, lambda c: shift(c, d)) == [(r - d) % world for r in range(world)]
        got = dist.run_ranks(comms, lambda c: c.all_gather_host(bytes([c.rank, 7, c.rank * 2])))
        assert all(g == [bytes([q, 7, q * 2]) for q in range(world)] for g in got)
        # a payload past the staging pair kept with the communicator
        big = dist.run_ranks(comms, lambda c: c.all_gather_host(bytes([c.rank]) * 10000))
        assert all(g == [bytes([q]) * 10000 for q in range(world)] for g in big)
    finally:
        for c in comms:
            c.close()


@pytest.mark.parametrize("world", [2, 4])
def test_one_ranks_failure_is_an_error_on_every_rank(gpu, oracle, world):
    """tkmk_comm_agree, and the status block inside the sharded MSM's gathered record: rank 1 alone gets a job the single-GPU entry
    refuses (an index list against a table of length 0) — every rank must report an error; afterwards the communicators still work"""
    from tkmk import dist
    n = 64
    s, p = np.asarray(oracle.fr_random(41, n)), np.asarray(oracle.g1_random_bases(42, n))
    want = np.asarray(oracle.g1_msm(s, p))
    comms = dist.loopback_comms(world)
    try:
        def agree(c):
            try:
                c.agree(11 if c.rank == 1 else 0)
                return "ok"
            except dist.DistError as e:
                return "error:%d" % e.code
        res = dist.run_ranks(comms, agree)
        assert all(r.startswith("error") for r in res), res
        assert res[1] == "error:11"                                 # the failing rank keeps its own code
        assert dist.run_ranks(comms, lambda c: (c.agree(0), "ok")[1]) == ["ok"] * world

        ds, dp = gpu.DeviceBuffer.from_host(s), gpu.DeviceBuffer.from_host(p)
        bounds = [(n * r // world, n * (r + 1) // world) for r in range(world)]
        idx = [gpu.DeviceBuffer.from_host(np.arange(lo, hi, dtype=np.uint32).view(np.uint8)) for lo, hi in bounds]

        def batch(c, poison):
            lo, hi = bounds[c.rank]
            job = dict(scalars=ds, scalar_offset=32 * lo, bases=dp, n=hi - lo, base_index=idx[c.rank], table_len=n)
            if poison and c.rank == 1:
                job["table_len"] = 0                                # refused by tkmk_msm_multi_ex: every index is past the table
            try:
                return gpu.projective_to_affine_bytes(c.msm_multi_ex_sharded([job], bases_form=gpu.BASES_PLAIN))
            except dist.DistError as e:
                return "error:%d" % e.code
        res = dist.run_ranks(comms, lambda c: batch(c, True))
        assert all(isinstance(r, str) for r in res), "a rank returned a sum although rank 1's share failed"
        res = dist.run_ranks(comms, lambda c: batch(c, False))
        for r in res:
            assert (np.asarray(r) == want).all()
    finally:
        for c in comms:
            c.close()


@pytest.mark.parametrize("world", [2, 4])
def test_relayout_of_point_records(gpu, oracle, world):
    """tkmk_dist_relayout_cols_to_rows / _rows_to_cols on 96-byte records (G1 affine points: what the group transforms behind the
    Lagrange-basis tables move at a sharded open) and on 32-byte ones, against numpy slicing of the whole matrix; a record size that is
    not a multiple of 16 is refused"""
    from tkmk import dist
    xs, ys = 16, 8
    comms = dist.loopback_comms(world)
    try:
        for rec in (96, 32, 48):
            m = np.random.default_rng(rec).integers(0, 256, (xs, ys, rec), dtype=np.uint8)
            to_rows = dist.run_ranks(comms, lambda c: np.asarray(c.relayout_cols_to_rows(gpu.DeviceBuffer.from_host(np.ascontiguousarray(m[:, c.rank::world]).reshape(-1)),
                                                                                         xs, ys, rec).to_host()))
            h = xs // world
            for r in range(world):
                assert (to_rows[r].reshape(h, ys, rec) == m[r * h:(r + 1) * h]).all(), (rec, r)
            back = dist.run_ranks(comms, lambda c: np.asarray(c.relayout_rows_to_cols(gpu.DeviceBuffer.from_host(np.ascontiguousarray(m[c.rank * h:(c.rank + 1) * h]).reshape(-1)),
                                                                                      xs, ys, rec).to_host()))
            for r in range(world):
                assert (back[r].reshape(xs, ys // world, rec) == m[:, r::world]).all(), (rec, r)

        def bad(c):
            with pytest.raises(dist.DistError):
                c.relayout_cols_to_rows(gpu.DeviceBuffer(40 * xs * ys // world), xs, ys, 40)
            return True
        assert dist.run_ranks(comms, bad) == [True] * world
    finally:
        for c in comms:
            c.close()
