"""gpu tier: NTT / iNTT / _biNTT through the C ABI vs the oracle, bit-exact.
Cases follow the reference's tests (libs/src/tests.rs:107-180 round trip + coset; :519-646 row/column routes;
:1042-1088 multiplication via NTT) and its production shapes (4096x256 ... 16384x512)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dom(gpu):
    gpu.init_ntt_domain_for_size(1 << 23)   # production domain: 4*max(m_I,n)*2*s_max (libs/src/utils/mod.rs:51-58)
    return gpu


@pytest.mark.parametrize("logn,batch", [(0, 3), (1, 1), (3, 5), (8, 1), (8, 9), (9, 16), (10, 3), (12, 2), (13, 1), (16, 1)])
def test_ntt_rows_vs_oracle(dom, oracle, logn, batch):
    n = 1 << logn
    x = oracle.fr_random(100 + logn, n * batch)
    g = oracle.fr_random(7, 1)
    for inverse in (False, True):
        for coset in (None, g):
            want = oracle.ntt(x, n, batch=batch, inverse=inverse, coset_gen=coset)
            assert (dom.ntt(x, n, batch=batch, inverse=inverse, coset_gen=coset) == want).all(), (inverse, coset is not None)


@pytest.mark.parametrize("logn,batch", [(3, 5), (6, 33), (12, 16), (12, 256), (13, 40)])
def test_ntt_columns_vs_oracle(dom, oracle, logn, batch):
    n = 1 << logn
    x = oracle.fr_random(200 + logn, n * batch)
    g = oracle.fr_random(8, 1)
    for inverse in (False, True):
        for coset in (None, g):
            want = oracle.ntt(x, n, batch=batch, columns_batch=True, inverse=inverse, coset_gen=coset)
            got = dom.ntt(x, n, batch=batch, columns_batch=True, inverse=inverse, coset_gen=coset)
            assert (got == want).all(), (inverse, coset is not None)


def test_ntt_in_place_and_device_resident(dom, oracle):
    n, batch = 1 << 12, 4
    x = oracle.fr_random(31, n * batch)
    d = dom.DeviceBuffer.from_host(x)
    dom.ntt(d, n, batch=batch, out=d)
    assert (d.to_host() == oracle.ntt(x, n, batch=batch)).all()
    dom.ntt(d, n, batch=batch, inverse=True, out=d)
    assert (d.to_host() == x).all()
    # odd number of passes, in place (2^19 -> 3 passes)
    n = 1 << 19
    x = oracle.fr_random(32, n)
    d = dom.DeviceBuffer.from_host(x)
    dom.ntt(d, n, out=d)
    assert (d.to_host() == oracle.ntt(x, n)).all()


@pytest.mark.parametrize("xs,ys", [(1, 64), (64, 1), (16, 8), (64, 32), (4096, 256)])
def test_bintt_vs_oracle(dom, oracle, xs, ys):
    m = oracle.fr_random(300 + xs, xs * ys)
    cx, cy = oracle.fr_random(9, 1), oracle.fr_random(10, 1)
    for inverse in (False, True):
        for kx, ky in ((None, None), (cx, cy)):
            want = oracle.bintt(m, xs, ys, inverse=inverse, coset_x=kx, coset_y=ky)
            assert (dom.bintt(m, xs, ys, inverse=inverse, coset_x=kx, coset_y=ky) == want).all()


def test_bintt_equals_row_then_column_calls(dom, oracle):
    # the reference issues two ntt calls (bivariate_polynomial/mod.rs:1465-1476); tkmk_bintt fuses them
    xs, ys = 512, 128
    m = oracle.fr_random(41, xs * ys)
    cx, cy = oracle.fr_random(42, 1), oracle.fr_random(43, 1)
    rows = dom.ntt(m, ys, batch=xs, coset_gen=cy)
    two_calls = dom.ntt(rows, xs, batch=ys, columns_batch=True, coset_gen=cx)
    assert (dom.bintt(m, xs, ys, coset_x=cx, coset_y=cy) == two_calls).all()


def test_production_shape_roundtrip_and_convolution(dom, oracle):
    # 8192 x 512 (2^22): iNTT(NTT(a)) == a and the convolution theorem on a sparse pair (tests.rs:1042-1088)
    xs, ys = 8192, 512
    a = dom.fr_random_device(51, xs * ys)
    ev = dom.bintt(a, xs, ys)
    back = dom.bintt(ev, xs, ys, inverse=True)
    assert (back.to_host() == a.to_host()).all()
    # (1 + X)(1 + Y) = 1 + X + Y + XY
    p = np.zeros(32 * xs * ys, np.uint8)
    q = np.zeros(32 * xs * ys, np.uint8)
    p[0] = 1
    p[32 * ys] = 1          # X^1
    q[0] = 1
    q[32] = 1               # Y^1
    prod = dom.bintt(dom.vec_mul(dom.bintt(p, xs, ys), dom.bintt(q, xs, ys)), xs, ys, inverse=True)
    want = np.zeros_like(p)
    for idx in (0, 1, ys, ys + 1):
        want[32 * idx] = 1
    assert (prod == want).all()


def test_ntt_errors(dom, oracle):
    x = oracle.fr_random(1, 8)
    with pytest.raises(dom.TkmkError):
        dom.ntt(x[:32 * 3], 3)               # not a power of two
    # larger than the domain (reference panics: mod.rs:1437-1445); the domain is process-wide and grow-only, so start from a known one
    dom.release_ntt_domain()
    dom.init_ntt_domain_for_size(1 << 12)
    try:
        with pytest.raises(dom.TkmkError):
            dom.ntt(np.zeros(32 << 13, np.uint8), 1 << 13)
        assert (dom.ntt(dom.ntt(x, 8), 8, inverse=True) == x).all()      # the failed call left the library usable
    finally:
        dom.init_ntt_domain_for_size(1 << 23)


def test_full_domain_sizes(dom, oracle):
    """largest shapes the production domain allows (2^23 elements): the p_comb domain 16384 x 512 and one 2^23-point
    vector — round trips, and linearity NTT(a + b) == NTT(a) + NTT(b)"""
    xs, ys = 16384, 512
    a, b = dom.fr_random_device(71, xs * ys), dom.fr_random_device(72, xs * ys)
    ea, eb = dom.bintt(a, xs, ys), dom.bintt(b, xs, ys)
    eab = dom.bintt(dom.vec_add(a, b), xs, ys)
    assert (dom.vec_add(ea, eb).to_host() == eab.to_host()).all()
    assert (dom.bintt(ea, xs, ys, inverse=True).to_host() == a.to_host()).all()
    n = 1 << 23
    ev = dom.ntt(a, n)
    assert (dom.ntt(ev, n, inverse=True).to_host() == a.to_host()).all()
    # spot-check 2^23-point outputs against the definition: X[k] = sum_j x[j] w^{jk} for a sparse x
    x = np.zeros(32 * n, np.uint8)
    x[32 * 1] = 1          # x = delta_1  ->  X[k] = w^k
    x[32 * 5 + 0] = 2      # + 2 delta_5 ->  X[k] = w^k + 2 w^{5k}
    X = dom.ntt(x, n)
    w = oracle.to_ints(oracle.root_of_unity(n), 32)[0]
    for k in (0, 1, 12345, n - 1):
        want = (pow(w, k, oracle.R_MOD) + 2 * pow(w, 5 * k, oracle.R_MOD)) % oracle.R_MOD
        assert oracle.to_ints(X[32 * k:32 * k + 32].copy(), 32)[0] == want


@pytest.mark.parametrize("in_x,in_y,xs,ys", [(4, 8, 16, 16), (1, 16, 64, 16), (16, 1, 16, 128), (32, 32, 32, 32), (8, 512, 4096, 1024), (256, 64, 1024, 64),
                                             (64, 256, 64, 1024), (1, 1, 1, 1), (1, 4, 1, 4096), (8, 1, 2048, 1), (2, 2, 4096, 2048), (512, 128, 2048, 512)])
def test_bintt_padded_equals_bintt_of_the_resized_matrix(gpu, oracle, in_x, in_y, xs, ys):
    """tkmk_bintt_padded (row pass over the existing rows only, absent rows / columns read as zeros) == the oracle's _biNTT of the
    explicitly zero-padded matrix — what resize + _biNTT of the reference computes (bivariate_polynomial/mod.rs:1646-1674)"""
    gpu.init_ntt_domain_for_size(max(xs * ys, 4))
    a = np.asarray(oracle.fr_random(4000 + in_x * 7 + in_y, in_x * in_y)).reshape(in_x, in_y, 32)
    padded = np.zeros((xs, ys, 32), np.uint8)
    padded[:in_x, :in_y] = a
    d = gpu.DeviceBuffer.from_host(np.ascontiguousarray(a.reshape(-1)))
    for cx, cy in ((None, None), (oracle.fr_random(9, 1), oracle.fr_random(10, 1))):
        want = np.asarray(oracle.bintt(np.ascontiguousarray(padded.reshape(-1)), xs, ys, coset_x=cx, coset_y=cy))
        got = np.asarray(gpu.bintt_padded(d, in_x, in_y, xs, ys, coset_x=cx, coset_y=cy).to_host())
        assert (got == want).all(), (cx is not None)
    # the output buffer may hold anything beforehand (rows past in_x of the intermediate are never written and never read)
    out = gpu.DeviceBuffer.from_host(np.full(32 * xs * ys, 0xAB, np.uint8))
    got = np.asarray(gpu.bintt_padded(d, in_x, in_y, xs, ys, out=out).to_host())
    assert (got == np.asarray(oracle.bintt(np.ascontiguousarray(padded.reshape(-1)), xs, ys))).all()


def test_bintt_padded_refuses_bad_shapes(gpu):
    gpu.init_ntt_domain_for_size(1 << 10)
    d = gpu.DeviceBuffer(32 * 64)
    for args in ((8, 8, 4, 8), (8, 8, 8, 4), (3, 8, 8, 8), (8, 8, 8, 24)):
        with pytest.raises(gpu.TkmkError):
            gpu.bintt_padded(d, *args)
    with pytest.raises(gpu.TkmkError):
        gpu.bintt_padded(d, 8, 8, 8, 8, out=d)      # in place is not possible: the operand is compact
