"""gpu tier: BN254 G1 MSM (bn254_msm, the second instantiation of csrc/msm_impl.inc) through the C ABI vs the oracle's
BN254 instantiation, bit-exact on the affine result.  The reference has no BN254 path, so there is
nothing of the reference's to pin these outputs on: parity here is oracle-vs-kernel plus the group-law identities below;
the oracle's BN254 code is itself pinned on the EIP-196 generator / 2G constants and on plain Python integer arithmetic
(tests/test_oracle_pins.py::test_bn254_constants)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
C = "bn254"


def _msm_affine(gpu, s, p, **kw):
    return gpu.projective_to_affine_bytes(gpu.msm(s, p, curve=C, **kw), curve=C)


@pytest.mark.parametrize("n", [1, 2, 3, 17, 128, 257, 1000, 4099])
def test_bn254_msm_vs_oracle_sizes(gpu, oracle, n):
    bn = oracle.bn254
    s = bn.fr_random(1000 + n, n)
    p = bn.g1_random_bases(2000 + n, n)
    assert (_msm_affine(gpu, s, p) == bn.g1_msm(s, p)).all()


@pytest.mark.parametrize("c", [2, 3, 5, 8, 11, 15, 16, 18])
def test_bn254_msm_every_window_width(gpu, oracle, c):
    bn = oracle.bn254
    n = 300
    s = bn.fr_random(77, n)
    p = bn.g1_random_bases(78, n)
    assert (_msm_affine(gpu, s, p, c=c) == bn.g1_msm(s, p)).all()


def test_bn254_msm_edge_scalars_and_bases(gpu, oracle):
    bn = oracle.bn254
    n = 64
    sv = oracle.to_ints(bn.fr_random(5, n), 32)
    sv[0:6] = [0, 1, bn.R_MOD - 1, 2, (1 << 253), 0x8000]
    sv[10:20] = [0] * 10
    p = bn.g1_random_bases(6, n)
    p[64 * 7:64 * 8] = 0                                   # infinity base
    p[64 * 9:64 * 10] = p[64 * 8:64 * 9]                   # repeated base (doubling branch)
    sv[9] = sv[8]
    p[64 * 12:64 * 13] = bn.g1_neg(p[64 * 11:64 * 12].copy())   # P and -P in one bucket
    sv[12] = sv[11]
    s = oracle.to_bytes(sv, 32)
    assert (_msm_affine(gpu, s, p) == bn.g1_msm(s, p)).all()
    z = np.zeros(32 * n, np.uint8)
    assert (_msm_affine(gpu, z, p) == 0).all()
    ones = oracle.to_bytes([1] * n, 32)
    same = np.tile(p[:64], n)
    assert (_msm_affine(gpu, ones, same) == bn.g1_scalar_mul(oracle.to_bytes([n], 32), p[:64].copy())).all()
    assert (_msm_affine(gpu, s[:0], p[:0], msm_size=0) == 0).all()
    # r * G = infinity, (r-1) * G = -G  (scalars are taken as given, below r)
    g = bn.g1_generator()
    assert (_msm_affine(gpu, oracle.to_bytes([bn.R_MOD - 1], 32), g) == bn.g1_neg(g)).all()


def test_bn254_msm_batch_and_multi(gpu, oracle):
    bn = oracle.bn254
    n, batch = 50, 3
    s = bn.fr_random(8, n * batch)
    p = bn.g1_random_bases(9, n * batch)
    shared = _msm_affine(gpu, s, p[:64 * n].copy(), msm_size=n, batch=batch, shared_points=True)
    per = _msm_affine(gpu, s, p, msm_size=n, batch=batch, shared_points=False)
    for b in range(batch):
        sb = s[32 * n * b:32 * n * (b + 1)].copy()
        assert (shared[64 * b:64 * (b + 1)] == bn.g1_msm(sb, p[:64 * n].copy())).all()
        assert (per[64 * b:64 * (b + 1)] == bn.g1_msm(sb, p[64 * n * b:64 * n * (b + 1)].copy())).all()
    g = bn.g1_generator()
    sc = bn.fr_random(10, 20)
    ones = _msm_affine(gpu, sc, g, msm_size=1, batch=20, shared_points=True)
    assert (ones == bn.g1_batch_scalar_mul(sc, g)).all()
    sizes = [700, 0, 1, 33, 3000, 2]
    jobs, want = [], []
    for k, m in enumerate(sizes):
        sj = bn.fr_random(300 + k, m) if m else np.zeros(0, np.uint8)
        pj = bn.g1_random_bases(400 + k, m) if m else np.zeros(0, np.uint8)
        jobs.append((sj, pj, m))
        want.append(bn.g1_msm(sj, pj) if m else np.zeros(64, np.uint8))
    got = gpu.projective_to_affine_bytes(gpu.msm_multi(jobs, curve=C), curve=C)
    for k in range(len(sizes)):
        assert (got[64 * k:64 * (k + 1)] == want[k]).all(), k


def test_bn254_device_generators_match_oracle(gpu, oracle):
    bn = oracle.bn254
    n = 3000
    assert (gpu.fr_random_device(21, n, first=5, curve=C).to_host() == bn.fr_random(21, n, first=5)).all()
    k = gpu.fr_random_device(22, n, curve=C)
    pts = gpu.g1_batch_scalar_mul_device(k, bn.g1_generator(), n, curve=C).to_host()
    assert (pts == bn.g1_random_bases(22, n)).all()


def test_bn254_msm_large_identity(gpu, oracle):
    # 2^20 points (two-pass sort, c = 16): sum s_i (k_i G) == (sum s_i k_i) G
    bn = oracle.bn254
    n = 1 << 20
    sd = gpu.fr_random_device(91, n, curve=C)
    k = gpu.fr_random_device(92, n, curve=C)
    g = bn.g1_generator()
    pd = gpu.g1_batch_scalar_mul_device(k, g, n, curve=C)
    got = _msm_affine(gpu, sd, pd)
    sv = oracle.to_ints(sd.to_host(), 32)
    kv = oracle.to_ints(k.to_host(), 32)
    dot = sum(x * y for x, y in zip(sv, kv)) % bn.R_MOD
    assert (got == bn.g1_scalar_mul(oracle.to_bytes([dot], 32), g.copy())).all()
    # sample cross-check of the generated bases against the oracle
    assert (pd.to_host(64 * 64, 64 * 1000) == bn.g1_random_bases(92, 64, first=1000)).all()
