"""gpu tier: the BASELINE.json configurations at their FULL sizes, through size-independent properties (the oracle
cannot produce 2^24-point references in seconds): configs[1] = 2^24-point G1 MSM (BLS12-381 and the BN254 twin),
configs[2] = 256 independent length-2^20 NTTs.  Everything stays in HBM; comparisons are single-point results or
device-side differences reduced to a scalar."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _zero_everywhere(gpu, buf, n):
    """all n field elements of a device buffer are zero  <=>  no non-zero coefficient in the n x 1 'matrix'"""
    from tkmk.poly import DensePolynomialExt as P
    return P.from_coeffs(buf, n, 1).find_degree() == (-1, -1)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_msm_2_24_linearity_and_additivity(gpu, oracle, curve):
    o = oracle if curve == "bls12_381" else oracle.bn254
    aff = 96 if curve == "bls12_381" else 64
    n = 1 << 24
    h = gpu.fr_random_device(81, n, curve=curve)
    p = gpu.g1_batch_scalar_mul_device(h, o.g1_generator(), n, curve=curve)
    a, b = gpu.fr_random_device(82, n, curve=curve), gpu.fr_random_device(83, n, curve=curve)

    def msm(s, bases, **kw):
        return gpu.projective_to_affine_bytes(gpu.msm(s, bases, curve=curve, **kw), curve=curve)

    ma, mb = msm(a, p), msm(b, p)
    if curve == "bls12_381":
        # absolute check through the discrete logarithms: P_i = h_i G, so MSM(a, P) = (sum_i a_i h_i) G — the inner product on
        # the Fr vector ops (tested against the oracle on their own), one scalar multiplication in the oracle
        dot = gpu.vec_sum(gpu.vec_mul(a, h))
        dot = np.asarray(dot.to_host() if hasattr(dot, "to_host") else dot, np.uint8)
        assert (ma == o.g1_scalar_mul(dot, o.g1_generator())).all()
    h.free()
    # generator spot check of the 2^24 device-generated bases against the oracle
    assert (p.to_host(aff * 8, aff * 5000) == o.g1_random_bases(81, 8, first=5000)).all()
    if curve == "bls12_381":
        mab = msm(gpu.vec_add(a, b), p)                      # MSM(a + b) = MSM(a) + MSM(b)   (Fr vector ops are BLS12-381's)
        assert (mab == o.g1_add(ma, mb)).all()
    # additivity over a split of the points: MSM(whole) = MSM(first part) + MSM(rest), odd split
    cut = (n // 3) | 1
    a1 = gpu.DeviceBuffer(32 * cut)
    a2 = gpu.DeviceBuffer(32 * (n - cut))
    p1 = gpu.DeviceBuffer(aff * cut)
    p2 = gpu.DeviceBuffer(aff * (n - cut))
    import ctypes
    lib = gpu.lib()
    for dst, src, off, nb in ((a1, a, 0, 32 * cut), (a2, a, 32 * cut, 32 * (n - cut)), (p1, p, 0, aff * cut), (p2, p, aff * cut, aff * (n - cut))):
        gpu._check(lib.tkmk_memcpy_d2d(ctypes.c_void_p(dst.ptr), ctypes.c_void_p(src.ptr + off), ctypes.c_size_t(nb)), "tkmk_memcpy_d2d")
    assert (o.g1_add(msm(a1, p1), msm(a2, p2)) == ma).all()
    # the pipelined multi entry returns the same two points
    both = gpu.projective_to_affine_bytes(gpu.msm_multi([(a, p, n), (b, p, n)], curve=curve), curve=curve)
    assert (both[:aff] == ma).all() and (both[aff:] == mb).all()


def test_ntt_256_x_2_20_roundtrip_and_linearity(gpu, oracle):
    gpu.init_ntt_domain_for_size(1 << 23)
    n, batch = 1 << 20, 256
    total = n * batch
    a = gpu.fr_random_device(91, total)
    ea = gpu.ntt(a, n, batch=batch)
    back = gpu.ntt(ea, n, batch=batch, inverse=True)
    assert _zero_everywhere(gpu, gpu.vec_sub(back, a, out=back), total)              # iNTT(NTT(a)) == a, all 2^28 elements
    b = gpu.fr_random_device(92, total)
    eb = gpu.ntt(b, n, batch=batch)
    eab = gpu.ntt(gpu.vec_add(a, b, out=b), n, batch=batch)                          # b <- a + b
    gpu.vec_add(ea, eb, out=ea)
    assert _zero_everywhere(gpu, gpu.vec_sub(ea, eab, out=ea), total)                # NTT(a + b) == NTT(a) + NTT(b)
    # one vector of the batch against the definition on a sparse input: x = delta_3 in vector 200 -> X[k] = w^{3k}
    x = gpu.DeviceBuffer(32 * total)
    gpu._check(gpu.lib().tkmk_memset(gpu._p(x), 0, gpu.ctypes.c_size_t(32 * total)), "tkmk_memset")
    one = np.zeros(32, np.uint8)
    one[0] = 1
    import ctypes
    gpu._check(gpu.lib().tkmk_memcpy_h2d(ctypes.c_void_p(x.ptr + 32 * (200 * n + 3)), gpu._p(one), ctypes.c_size_t(32)), "tkmk_memcpy_h2d")
    X = gpu.ntt(x, n, batch=batch)
    w = oracle.to_ints(oracle.root_of_unity(n), 32)[0]
    for k in (0, 1, 77777, n - 1):
        got = oracle.to_ints(X.to_host(32, 32 * (200 * n + k)), 32)[0]
        assert got == pow(w, 3 * k, oracle.R_MOD)
    assert not X.to_host(32 * 16, 32 * (199 * n + 5)).any()                          # neighbours stay zero


def test_configs4_2_28_points_whole_and_in_eight_shards(gpu, oracle):
    """BASELINE.json configs[4] — a 2^28-point G1 MSM split over 8 GPUs — at its FULL size on the one GPU of the box: the eight
    2^25-point shards one after another through the sharded entry (libtkmk_dist.so, one-rank communicator: the RCCL all-gather of
    partial results runs, the per-rank work is exactly a rank's), their sum, and the whole 2^28-point MSM in one call; all against
    the discrete-log checksum (sum_i a_i h_i) G.  34 GB of operands stay in HBM."""
    from tkmk import dist
    import ctypes
    lib = gpu.lib()
    shards, n_shard = 8, 1 << 25
    n = shards * n_shard
    G = oracle.g1_generator()
    a = gpu.fr_random_device(301, n)
    h = gpu.fr_random_device(302, n)
    p = gpu.DeviceBuffer(96 * n)
    for j in range(shards):                                # bases P_i = h_i G, one slab at a time (bounded scratch)
        hj = gpu.DeviceBuffer(32 * n_shard)
        gpu._check(lib.tkmk_memcpy_d2d(ctypes.c_void_p(hj.ptr), ctypes.c_void_p(h.ptr + 32 * n_shard * j), ctypes.c_size_t(32 * n_shard)), "tkmk_memcpy_d2d")
        pj = gpu.g1_batch_scalar_mul_device(hj, G, n_shard)
        gpu._check(lib.tkmk_memcpy_d2d(ctypes.c_void_p(p.ptr + 96 * n_shard * j), ctypes.c_void_p(pj.ptr), ctypes.c_size_t(96 * n_shard)), "tkmk_memcpy_d2d")
        hj.free()
        pj.free()
    assert (p.to_host(96 * 4, 96 * (n - 4)) == oracle.g1_random_bases(302, 4, first=n - 4)).all()     # the last four bases, oracle's own [h_i] G
    prod = gpu.vec_mul(a, h)
    h.free()
    dot = gpu.vec_sum(prod)
    prod.free()
    dot = np.asarray(dot.to_host() if hasattr(dot, "to_host") else dot, np.uint8)
    want = oracle.g1_scalar_mul(dot, G)
    comm = dist.Comm(dist.unique_id(), 1, 0)
    acc = np.zeros(96, np.uint8)
    for j in range(shards):
        aj, pj = gpu.DeviceBuffer(32 * n_shard), gpu.DeviceBuffer(96 * n_shard)
        gpu._check(lib.tkmk_memcpy_d2d(ctypes.c_void_p(aj.ptr), ctypes.c_void_p(a.ptr + 32 * n_shard * j), ctypes.c_size_t(32 * n_shard)), "tkmk_memcpy_d2d")
        gpu._check(lib.tkmk_memcpy_d2d(ctypes.c_void_p(pj.ptr), ctypes.c_void_p(p.ptr + 96 * n_shard * j), ctypes.c_size_t(96 * n_shard)), "tkmk_memcpy_d2d")
        acc = oracle.g1_add(acc, gpu.projective_to_affine_bytes(comm.msm_sharded(aj, pj)))
        aj.free()
        pj.free()
    comm.close()
    assert (acc == want).all()
    whole = gpu.projective_to_affine_bytes(gpu.msm(a, p))
    assert (whole == want).all()
