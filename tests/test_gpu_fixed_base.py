"""gpu tier: the fixed-base path of tkmk_g1_batch_scalar_mul_device (csrc/gen.hip: byte-window table of the base, one mixed addition
per scalar byte, shared inversions) — what Sigma::gen's 2^22 .. 2^24 multiples of the generator run through
(packages/backend/libs/src/group_structures/mod.rs:384-393, type_scaled_monomials_1d!) — against the oracle's scalar multiplication."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _special_scalars(oracle, n, seed, R):
    vals = oracle.to_ints(oracle.fr_random(seed, n), 32)
    special = [0, 1, 2, R - 1, R - 2, 255, 256, 1 << 8, 1 << 248, (1 << 248) + 1, 0xFF << 120, (R - 1) & ~(0xFF << 64)]
    for k, v in enumerate(special):
        vals[k * 7] = v % R
    vals[-1] = 0
    vals[-2] = R - 1
    return vals


@pytest.mark.parametrize("n", [8192, 8192 + 37, 20000 + 5])
def test_fixed_base_equals_oracle(gpu, oracle, n):
    R = oracle.R_MOD
    vals = _special_scalars(oracle, n, 77, R)
    sc = oracle.to_bytes(vals, 32)
    g = oracle.g1_random_bases(5, 1)                               # an arbitrary base, not the generator
    got = gpu.g1_batch_scalar_mul_device(gpu.DeviceBuffer.from_host(sc), g, n).to_host()
    want = np.asarray(oracle.g1_batch_scalar_mul(sc, g))
    assert (got.reshape(-1, 96) == want.reshape(-1, 96)).all()
    assert not got.reshape(-1, 96)[0].any() and not got.reshape(-1, 96)[-1].any()      # zero scalars give (0,0)


def test_fixed_base_infinity_base_and_small_path_agree(gpu, oracle):
    n = 9000
    sc = oracle.fr_random(3, n)
    inf = np.zeros(96, np.uint8)
    assert not gpu.g1_batch_scalar_mul_device(gpu.DeviceBuffer.from_host(sc), inf, n).to_host().any()
    # the same scalars through the small-batch kernel (n < 8192 per call) give the same points
    g = oracle.g1_generator()
    big = gpu.g1_batch_scalar_mul_device(gpu.DeviceBuffer.from_host(sc), g, n).to_host().reshape(-1, 96)
    half = n // 2
    small = np.concatenate([gpu.g1_batch_scalar_mul_device(gpu.DeviceBuffer.from_host(np.ascontiguousarray(sc[32 * a:32 * b])), g, b - a).to_host()
                            for a, b in ((0, half), (half, n))]).reshape(-1, 96)
    assert (big == small).all()


def test_fixed_base_across_the_tile_boundary(gpu, oracle):
    """n > 2^22 runs in tiles; sampled entries on both sides of the boundary (and the last one) against the oracle"""
    n = (1 << 22) + 1000 + 3
    d = gpu.fr_random_device(123, n)
    g = oracle.g1_generator()
    out = gpu.g1_batch_scalar_mul_device(d, g, n)
    idx = [0, 1, (1 << 22) - 1, 1 << 22, (1 << 22) + 1, n - 17, n - 2, n - 1]
    for i in idx:
        s = d.to_host(32, offset=32 * i)
        assert (out.to_host(96, offset=96 * i) == np.asarray(oracle.g1_scalar_mul(s, g))).all(), i


def test_fixed_base_bn254(gpu, oracle):
    o = oracle.bn254 if hasattr(oracle, "bn254") else None
    if o is None:
        pytest.skip("oracle has no bn254 instantiation")
    n = 8192 + 11
    sc = o.fr_random(9, n)
    g = o.g1_generator()
    got = gpu.g1_batch_scalar_mul_device(gpu.DeviceBuffer.from_host(sc), g, n, curve="bn254").to_host()
    assert (got.reshape(-1, 64) == np.asarray(o.g1_batch_scalar_mul(sc, g)).reshape(-1, 64)).all()
