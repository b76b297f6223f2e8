"""gpu tier: Fr vector ops through the C ABI vs the oracle, bit-exact (integer work).
Mirrors the reference's use at libs/src/vector_operations/mod.rs:30-139."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _edge_vec(oracle, seed, n):
    v = oracle.to_ints(oracle.fr_random(seed, n), 32)
    edge = [0, 1, oracle.R_MOD - 1, 2][:n]
    v[:len(edge)] = edge
    return oracle.to_bytes(v, 32)


@pytest.mark.parametrize("n", [1, 5, 255, 4097])
def test_binary_and_scalar_ops(gpu, oracle, n):
    a = _edge_vec(oracle, 1, n)
    b = oracle.fr_random(2, n)
    s = oracle.fr_random(3, 1)
    assert (gpu.vec_add(a, b) == oracle.fr_add(a, b)).all()
    assert (gpu.vec_sub(a, b) == oracle.fr_sub(a, b)).all()
    assert (gpu.vec_mul(a, b) == oracle.fr_mul(a, b)).all()
    assert (gpu.scalar_mul(s, a) == oracle.fr_scalar_mul(s, a)).all()
    assert (gpu.scalar_add(s, a) == oracle.fr_scalar_add(s, a)).all()
    assert (gpu.scalar_sub(s, a) == oracle.fr_scalar_sub(s, a)).all()
    assert (gpu.vec_inv(a) == oracle.fr_inv(a)).all()          # inv(0) = 0
    assert (gpu.vec_div(b, a) == oracle.fr_mul(b, oracle.fr_inv(a))).all()


def test_device_resident_operands(gpu, oracle):
    n = 3000
    a, b = oracle.fr_random(5, n), oracle.fr_random(6, n)
    da, db = gpu.DeviceBuffer.from_host(a), gpu.DeviceBuffer.from_host(b)
    out = gpu.vec_mul(da, db)
    assert isinstance(out, gpu.DeviceBuffer)
    assert (out.to_host() == oracle.fr_mul(a, b)).all()
    assert (gpu.vec_add(da, b).to_host() == oracle.fr_add(a, b)).all()   # mixed host/device like the reference
    gpu.vec_sub(da, db, out=da)                                          # in place
    assert (da.to_host() == oracle.fr_sub(a, b)).all()


@pytest.mark.parametrize("rows,cols", [(1, 7), (16, 16), (33, 5), (256, 64), (100, 257)])
def test_transpose(gpu, oracle, rows, cols):
    a = oracle.fr_random(rows * 1000 + cols, rows * cols)
    assert (gpu.transpose(a, rows, cols) == oracle.fr_transpose(a, rows, cols)).all()


@pytest.mark.parametrize("n,batch", [(1, 1), (300, 1), (10000, 1), (64, 5)])
def test_sum_and_product(gpu, oracle, n, batch):
    a = oracle.fr_random(n + batch, n * batch)
    vals = oracle.to_ints(a, 32)
    want_sum = [sum(vals[b * n:(b + 1) * n]) % oracle.R_MOD for b in range(batch)]
    assert oracle.to_ints(gpu.vec_sum(a, n, batch), 32) == want_sum
    want_prod = []
    for b in range(batch):
        p = 1
        for v in vals[b * n:(b + 1) * n]:
            p = p * v % oracle.R_MOD
        want_prod.append(p)
    assert oracle.to_ints(gpu.vec_product(a, n, batch), 32) == want_prod


def test_large_mul_linearity(gpu, oracle):
    # size-independent property at 2^22 elements: (a + b) * c == a*c + b*c
    n = 1 << 22
    a, b, c = (gpu.fr_random_device(s, n) for s in (21, 22, 23))
    lhs = gpu.vec_mul(gpu.vec_add(a, b), c)
    rhs = gpu.vec_add(gpu.vec_mul(a, c), gpu.vec_mul(b, c))
    assert (lhs.to_host() == rhs.to_host()).all()
    # and the device generator equals the oracle stream
    assert (a.to_host(32 * 1000) == oracle.fr_random(21, 1000)).all()


@pytest.mark.parametrize("field", [0, 1])
def test_device_montgomery_product_edges(gpu, oracle, field):
    """the hand-scheduled v_mad_u64_u32 product (csrc/ff.h mul_ps) on carry-heavy operands"""
    import random
    rnd = random.Random(99)
    width, mod, omul = (32, oracle.R_MOD, oracle.fr_mul) if field == 0 else (48, oracle.P_MOD, oracle.fq_mul)
    edge = [0, 1, 2, mod - 1, mod - 2, (mod - 1) // 2, (1 << (mod.bit_length() - 1)), (1 << 32) - 1, (1 << 64) - 1,
            ((1 << (mod.bit_length() - 1)) - 1), mod >> 1, 0xFFFFFFFF00000000FFFFFFFF % mod]
    a = [x for x in edge for _ in edge] + [rnd.randrange(mod) for _ in range(5000)]
    b = [y for _ in edge for y in edge] + [rnd.randrange(mod) for _ in range(5000)]
    A, B = oracle.to_bytes(a, width), oracle.to_bytes(b, width)
    assert (gpu.diag_field_mul(field, A, B) == omul(A, B)).all()


@pytest.mark.parametrize("n", [1, 2, 17, 4096, 4097, 70001, 1 << 20])
def test_suffix_product(gpu, oracle, n):
    """prove1's running product (prove/src/lib.rs:1858-1862) as a device scan vs the oracle's serial loop"""
    a = oracle.fr_random(900 + n % 1000, n)
    if n > 17:
        a[32 * 5:32 * 6] = 0            # a zero factor kills every product to its left
    got = gpu.vec_suffix_product(a).to_host()
    assert (got == oracle.fr_suffix_product(a)).all()


def test_release_scratch_returns_memory(gpu, oracle):
    n = 1 << 18
    s = gpu.fr_random_device(1, n)
    h = gpu.fr_random_device(2, n)
    p = gpu.g1_batch_scalar_mul_device(h, oracle.g1_generator(), n)
    r1 = gpu.msm(s, p)
    _, free_before = gpu.available_memory()
    gpu.release_scratch()
    _, free_after = gpu.available_memory()
    assert free_after > free_before
    assert (gpu.msm(s, p) == r1).all()          # arenas regrow transparently


def test_long_vector_inverse_and_division(gpu, oracle):
    # >= 2^16 elements take the 16-per-lane batched inversion (csrc/vecops.hip k_vec_inv<., 16>): ragged length, zeros
    # sprinkled through every lane's batch (a zero must not poison its neighbours), in-place output over either operand
    n = (1 << 17) + 12345
    av = oracle.to_ints(oracle.fr_random(31, n), 32)
    for i in list(range(0, n, 97)) + [1, 2, n - 1, n - 2]:
        av[i] = 0
    a = oracle.to_bytes(av, 32)
    b = oracle.fr_random(32, n)
    inv = oracle.fr_inv(a)
    assert (gpu.vec_inv(a) == inv).all()
    quot = oracle.fr_mul(b, inv)
    assert (gpu.vec_div(b, a) == quot).all()
    da, db = gpu.DeviceBuffer.from_host(a), gpu.DeviceBuffer.from_host(b)
    gpu.vec_div(db, da, out=db)                       # numerator overwritten
    assert (db.to_host() == quot).all()
    db = gpu.DeviceBuffer.from_host(b)
    gpu.vec_div(db, da, out=da)                       # denominator overwritten
    assert (da.to_host() == quot).all()
    da = gpu.DeviceBuffer.from_host(a)
    gpu.vec_inv(da, out=da)
    assert (da.to_host() == inv).all()
