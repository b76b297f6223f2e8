"""not-gpu tier: the N > 1 path (point-sharded MSM + one all_gather of 144-byte partials) with world_size 2 over
gloo on the CPU.  The per-rank MSM and the final combine are played by the ORACLE here (test-only stand-ins: the
product's msm needs a GPU); what is under test is the product's sharding / gather plumbing
(tokamak-zk-evm_amd/tkmk/sharding.py) and that shard sums recombine to the full MSM."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tokamak-zk-evm_amd"))
    import torch.distributed as dist
    import oracle
    from tkmk import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s = oracle.fr_random(5, n)
        p = oracle.g1_random_bases(6, n)
        lo, hi = sharding.shard_range(n, rank, world)
        part_aff = oracle.g1_msm(s[32 * lo:32 * hi].copy(), p[96 * lo:96 * hi].copy(), threads=1)
        part = np.zeros(144, np.uint8)           # canonical projective as the C ABI returns it
        if part_aff.any():
            part[:96] = part_aff
            part[96] = 1
        else:
            part[48] = 1
        gathered = sharding.gather_partials(dist, part, device="cpu")
        assert gathered.shape == (world, 144)
        assert (gathered[rank] == part).all()
        total = np.zeros(96, np.uint8)
        for g in gathered:
            aff = g[:96].copy() if g[96:].any() else np.zeros(96, np.uint8)
            total = oracle.g1_add(total, aff)
        q.put((rank, bool((total == oracle.g1_msm(s, p, threads=1)).all())))
    finally:
        dist.destroy_process_group()


def test_shard_range_covers_everything():
    sys.path.insert(0, os.path.join(ROOT, "tokamak-zk-evm_amd"))
    from tkmk import sharding
    for n in (0, 1, 7, 8, 1000, (1 << 28) + 3):
        for world in (1, 2, 3, 8):
            rs = [sharding.shard_range(n, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in rs]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_point_sharded_msm_world2_gloo():
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world, n = 2, 301      # odd size: ragged shards
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    res = dict(q.get(timeout=10) for _ in range(world))
    assert res == {0: True, 1: True}


class _OracleNttOps:
    """test-only stand-in for the per-rank GPU transforms"""

    def __init__(self, oracle):
        self.o = oracle

    def ntt_rows(self, buf, n, batch, coset, inverse):
        return self.o.ntt(buf, n, batch=batch, inverse=inverse, coset_gen=coset)

    def ntt_cols(self, buf, n, batch, coset, inverse):
        return self.o.ntt(buf, n, batch=batch, columns_batch=True, inverse=inverse, coset_gen=coset)


def _ntt_worker(rank, world, port, xs, ys, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tokamak-zk-evm_amd"))
    import torch.distributed as dist
    import oracle
    from tkmk import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = oracle.fr_random(9, xs * ys)
        cx, cy = oracle.fr_random(10, 1), oracle.fr_random(11, 1)
        rows, cols = xs // world, ys // world
        ok = True
        for inverse in (False, True):
            want = oracle.bintt(m, xs, ys, inverse=inverse, coset_x=cx, coset_y=cy).reshape(xs, ys, 32)
            slab = m.reshape(xs, ys * 32)[rank * rows:(rank + 1) * rows].reshape(-1).copy()
            got = sharding.bintt_sharded(_OracleNttOps(oracle), dist, slab, xs, ys, inverse=inverse, coset_x=cx, coset_y=cy)
            ok &= bool((got.reshape(xs, cols, 32) == want[:, rank * cols:(rank + 1) * cols, :]).all())
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_bintt_all_to_all_world2_gloo():
    """the single exchange step of the path: x-slabs -> row NTTs -> all_to_all -> column NTTs on y-slabs"""
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 2
    procs = [ctx.Process(target=_ntt_worker, args=(r, world, port, 16, 8, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    assert dict(q.get(timeout=10) for _ in range(world)) == {0: True, 1: True}


def _commit_worker(rank, world, port, sizes, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tokamak-zk-evm_amd"))
    import torch.distributed as dist
    import oracle
    from tkmk import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        jobs = [(oracle.fr_random(20 + k, n), oracle.g1_random_bases(40 + k, n), n) for k, n in enumerate(sizes)]
        touched = []

        def oracle_multi(js):                    # test-only stand-in for tkmk.msm_multi: 96-byte affine results
            touched.extend(j[2] for j in js)
            return np.concatenate([oracle.g1_msm(s, p, threads=1) for s, p, _ in js])

        got = sharding.commits_sharded(oracle_multi, dist, jobs, result_bytes=96, device="cpu")
        want = np.stack([oracle.g1_msm(s, p, threads=1) for s, p, _ in jobs])
        mine = [sizes[j] for j in sharding.jobs_of_rank(len(sizes), rank, world)]
        q.put((rank, bool((got == want).all()) and touched == mine))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_round_commits_sharded_world2_gloo():
    """independent commits of one round: job j on rank j mod G, one all_gather, results in job order on every rank;
    an odd job count leaves the last slot of rank 1 empty"""
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 2
    procs = [ctx.Process(target=_commit_worker, args=(r, world, port, [40, 7, 19, 64, 3], q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    assert dict(q.get(timeout=10) for _ in range(world)) == {0: True, 1: True}


def test_balanced_assignment_is_deterministic_and_balanced():
    from tkmk import sharding
    # one prove4 round at the production shape: two 4 M-point commits, five ~1 M, two tiny (prove/src/lib.rs prove4)
    sizes = [1052929, 257, 1052929, 258, 1052929, 258, 4194304, 512, 127]
    for world in (1, 2, 4, 8):
        owner = sharding.balanced_assignment(sizes, world)
        assert owner == sharding.balanced_assignment(list(sizes), world) and set(owner) <= set(range(world))
        load = [sum(s for s, o in zip(sizes, owner) if o == q) for q in range(world)]
        assert max(load) <= max(max(sizes), -(-sum(sizes) // world) + max(s for s in sizes if s < max(sizes)))
    assert sharding.balanced_assignment(sizes, 2).count(0) + sharding.balanced_assignment(sizes, 2).count(1) == len(sizes)
    assert sharding.balanced_assignment([], 4) == []


def _balanced_worker(rank, world, port, sizes, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tokamak-zk-evm_amd"))
    import torch.distributed as dist
    import oracle
    from tkmk import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        jobs = [(oracle.fr_random(60 + k, n), oracle.g1_random_bases(80 + k, n), n) for k, n in enumerate(sizes)]
        touched = []

        def oracle_multi(js):                    # test-only stand-in for tkmk.msm_multi: 96-byte affine results
            touched.extend(j[2] for j in js)
            return np.concatenate([oracle.g1_msm(s, p, threads=1) for s, p, _ in js]) if js else np.zeros(0, np.uint8)

        got = sharding.commits_balanced(oracle_multi, dist, jobs, sizes, result_bytes=96, device="cpu")
        want = np.stack([oracle.g1_msm(s, p, threads=1) for s, p, _ in jobs])
        owner = sharding.balanced_assignment(sizes, world)
        mine = [sizes[j] for j in range(len(sizes)) if owner[j] == rank]
        q.put((rank, bool((got == want).all()) and touched == mine))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_round_commits_balanced_world2_gloo():
    """the commits of one round spread by size (what Sigma1.encode_polys does when a process group is set): every rank ends with
    all commitments in job order, each MSM ran on exactly one rank, the big job sits alone"""
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 2
    procs = [ctx.Process(target=_balanced_worker, args=(r, world, port, [9, 70, 11, 8, 10, 12], q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    assert dict(q.get(timeout=10) for _ in range(world)) == {0: True, 1: True}
