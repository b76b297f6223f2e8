"""gpu tier: libtkmk_dist.so (include/tkmk_dist.h) — the sharded MSM and the slab-sharded bivariate NTT over RCCL on device buffers.
A one-GPU box can only form a ONE-rank communicator (RCCL refuses two ranks on one device), which still drives every RCCL call
of the path (ncclCommInitRank, ncclAllGather, ncclAllToAll on device buffers, the pack / place index algebra with G = 1); the
G = 2 partitioning itself is covered on the CPU by tests/test_sharding_gloo.py, and bench.py --gpus N --msm-sharded runs this
entry on N GPUs."""
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
