"""gpu tier: BN254 scalar-field NTT (bn254_ntt*, tkmk_bn254_bintt: the second instantiation of csrc/ntt_impl.inc) vs the oracle's
BN254 instantiation.  The reference has no BN254 path; BASELINE.json's configs[0] names "2^16-point BN254 G1 MSM + 2^16
scalar-field NTT", which is reproduced here as a parity case (GPU vs oracle).  Root convention: w_{2^28} = 5^((r-1)/2^28),
pinned in the oracle against Python integers."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
C = "bn254"


@pytest.fixture(scope="module")
def dom(gpu):
    gpu.init_ntt_domain_for_size(1 << 20, curve=C)
    return gpu


def test_root_of_unity_convention(dom, oracle):
    bn = oracle.bn254
    for size in (2, 1 << 10, 1 << 28):
        assert (dom.get_root_of_unity(size, curve=C) == bn.root_of_unity(size)).all()
    w = oracle.to_ints(dom.get_root_of_unity(1 << 28, curve=C), 32)[0]
    assert w == pow(5, (bn.R_MOD - 1) >> 28, bn.R_MOD)
    with pytest.raises(dom.TkmkError):
        dom.get_root_of_unity(1 << 29, curve=C)             # two-adicity of the BN254 scalar field is 28


@pytest.mark.parametrize("logn,batch", [(0, 3), (1, 1), (5, 7), (9, 4), (11, 3), (16, 1), (18, 2)])
def test_bn254_ntt_rows_and_columns(dom, oracle, logn, batch):
    bn = oracle.bn254
    n = 1 << logn
    x = bn.fr_random(100 + logn, n * batch)
    g = bn.fr_random(7, 1)
    for columns in (False, True):
        for inverse in (False, True):
            for coset in (None, g):
                want = bn.ntt(x, n, batch=batch, columns_batch=columns, inverse=inverse, coset_gen=coset)
                got = dom.ntt(x, n, batch=batch, columns_batch=columns, inverse=inverse, coset_gen=coset, curve=C)
                assert (got == want).all(), (columns, inverse, coset is not None)


def test_bn254_bintt_and_device_buffers(dom, oracle):
    bn = oracle.bn254
    xs, ys = 256, 64
    m = bn.fr_random(9, xs * ys)
    cx, cy = bn.fr_random(10, 1), bn.fr_random(11, 1)
    want = bn.bintt(m, xs, ys, coset_x=cx, coset_y=cy)
    assert (dom.bintt(m, xs, ys, coset_x=cx, coset_y=cy, curve=C) == want).all()
    d = dom.DeviceBuffer.from_host(m)
    e = dom.bintt(d, xs, ys, coset_x=cx, coset_y=cy, curve=C)
    assert (e.to_host() == want).all()
    dom.bintt(e, xs, ys, inverse=True, coset_x=cx, coset_y=cy, out=e, curve=C)      # in place, inverse
    assert (e.to_host() == m).all()


def test_baseline_config0_shape(dom, oracle):
    """BASELINE.json configs[0]: 2^16-point BN254 G1 MSM + 2^16 scalar-field NTT, GPU vs the CPU oracle"""
    bn = oracle.bn254
    n = 1 << 16
    s = bn.fr_random(21, n)
    p = bn.g1_random_bases(22, n)
    got = dom.projective_to_affine_bytes(dom.msm(s, p, curve=C), curve=C)
    assert (got == bn.g1_msm(s, p)).all()
    assert (dom.ntt(s, n, curve=C) == bn.ntt(s, n)).all()


def test_domains_of_the_two_fields_coexist(dom, oracle):
    dom.init_ntt_domain_for_size(1 << 12)                      # BLS12-381 domain next to the BN254 one
    x = oracle.fr_random(31, 1 << 10)
    assert (dom.ntt(x, 1 << 10) == oracle.ntt(x, 1 << 10)).all()
    y = oracle.bn254.fr_random(32, 1 << 10)
    assert (dom.ntt(y, 1 << 10, curve=C) == oracle.bn254.ntt(y, 1 << 10)).all()
