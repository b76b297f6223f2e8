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
    h.free()
    a, b = gpu.fr_random_device(82, n, curve=curve), gpu.fr_random_device(83, n, curve=curve)

    def msm(s, bases, **kw):
        return gpu.projective_to_affine_bytes(gpu.msm(s, bases, curve=curve, **kw), curve=curve)

    ma, mb = msm(a, p), msm(b, p)
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
