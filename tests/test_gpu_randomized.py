"""gpu tier: seeded randomized differential runs of the two hot kernels against the oracle: random sizes, window widths, bit
sizes, batch layouts, scalar patterns and operand residency for the MSM; random lengths, batch shapes, directions, cosets
and in-place / host / device operands for the NTT.  Complements the hand-picked edge cases of the other gpu tests."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scalars(oracle, rnd, n, pattern, mod):
    if pattern == "uniform":
        return [rnd.randrange(mod) for _ in range(n)]
    if pattern == "small":
        return [rnd.choice((0, 1, 2, 255, 1 << 16, (1 << 128) - 1)) for _ in range(n)]
    if pattern == "edges":
        return [rnd.choice((0, 1, mod - 1, mod - 2, 1 << 254, (1 << 255) % mod, rnd.randrange(mod))) for _ in range(n)]
    v = rnd.randrange(1, mod)                          # one repeated scalar: every point in the same buckets
    return [v] * n


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_msm_randomized(gpu, oracle, curve):
    o = oracle if curve == "bls12_381" else oracle.bn254
    aff = 96 if curve == "bls12_381" else 64
    mod = o.R_MOD
    rnd = random.Random(20260101 if curve == "bls12_381" else 20260202)
    pool_n = 6000
    pool = o.g1_random_bases(900, pool_n)
    for it in range(36):
        n = rnd.choice((2, 3, 5, 31, 64, 65, 200, 777, 1500, 4096, 6000))
        c = rnd.choice((0, 0, 2, 4, 7, 9, 12, 13, 14, 16, 17))
        pattern = rnd.choice(("uniform", "uniform", "small", "edges", "same"))
        start = rnd.randrange(0, pool_n - n + 1)
        p = pool[aff * start:aff * (start + n)].copy()
        for _ in range(rnd.choice((0, 0, 1, 3))):       # sprinkle infinity bases and duplicates
            k = rnd.randrange(n)
            p[aff * k:aff * (k + 1)] = 0
        if n > 4 and rnd.random() < 0.4:
            k = rnd.randrange(1, n)
            p[aff * k:aff * (k + 1)] = p[aff * (k - 1):aff * k]
        sv = _scalars(oracle, rnd, n, pattern, mod)
        s = oracle.to_bytes(sv, 32)
        want = o.g1_msm(s, p)
        dev_s, dev_p = rnd.random() < 0.5, rnd.random() < 0.5
        ss = gpu.DeviceBuffer.from_host(s) if dev_s else s
        pp = gpu.DeviceBuffer.from_host(p) if dev_p else p
        got = gpu.projective_to_affine_bytes(gpu.msm(ss, pp, msm_size=n, c=c, curve=curve), curve=curve)
        assert (got == want).all(), (curve, it, n, c, pattern, dev_s, dev_p)
        if rnd.random() < 0.3:                          # reduced bit size: scalars truncated to their low bits
            bits = rnd.choice((17, 64, 130, 200))
            tv = [x & ((1 << bits) - 1) for x in sv]
            want_t = o.g1_msm(oracle.to_bytes(tv, 32), p)
            got_t = gpu.projective_to_affine_bytes(gpu.msm(ss, pp, msm_size=n, c=c, bitsize=bits, curve=curve), curve=curve)
            assert (got_t == want_t).all(), (curve, it, n, c, bits)
        if rnd.random() < 0.3 and n >= 8:               # the same points as a batch of two half-size MSMs
            h = n // 2
            res = gpu.projective_to_affine_bytes(gpu.msm(s[:32 * 2 * h].copy(), p[:aff * 2 * h].copy(), msm_size=h, batch=2, shared_points=False,
                                                         c=c, curve=curve), curve=curve)
            assert (res[:aff] == o.g1_msm(s[:32 * h].copy(), p[:aff * h].copy())).all()
            assert (res[aff:] == o.g1_msm(s[32 * h:32 * 2 * h].copy(), p[aff * h:aff * 2 * h].copy())).all()


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_ntt_randomized(gpu, oracle, curve):
    o = oracle if curve == "bls12_381" else oracle.bn254
    gpu.init_ntt_domain_for_size(1 << 20, curve=curve)
    rnd = random.Random(77 if curve == "bls12_381" else 78)
    for it in range(40):
        logn = rnd.randrange(0, 15)
        n = 1 << logn
        batch = rnd.choice((1, 1, 2, 3, 5, 8, 17, 64))
        if n * batch > (1 << 17):
            batch = max(1, (1 << 17) // n)
        columns = rnd.random() < 0.5
        inverse = rnd.random() < 0.5
        coset = o.fr_random(1000 + it, 1) if rnd.random() < 0.5 else None
        x = o.fr_random(2000 + it, n * batch)
        want = o.ntt(x, n, batch=batch, columns_batch=columns, inverse=inverse, coset_gen=coset)
        mode = rnd.choice(("host", "device", "inplace"))
        if mode == "host":
            got = gpu.ntt(x, n, batch=batch, columns_batch=columns, inverse=inverse, coset_gen=coset, curve=curve)
        else:
            d = gpu.DeviceBuffer.from_host(x)
            out = gpu.ntt(d, n, batch=batch, columns_batch=columns, inverse=inverse, coset_gen=coset, curve=curve,
                          out=d if mode == "inplace" else None)
            got = out.to_host()
        assert (got == want).all(), (curve, it, logn, batch, columns, inverse, coset is not None, mode)
    for it in range(12):                                  # bivariate shapes incl. degenerate axes
        xs, ys = 1 << rnd.randrange(0, 9), 1 << rnd.randrange(0, 9)
        m = o.fr_random(3000 + it, xs * ys)
        cx = o.fr_random(3100 + it, 1) if rnd.random() < 0.5 else None
        cy = o.fr_random(3200 + it, 1) if rnd.random() < 0.5 else None
        inv = rnd.random() < 0.5
        got = gpu.bintt(m, xs, ys, inverse=inv, coset_x=cx, coset_y=cy, curve=curve)
        assert (got == o.bintt(m, xs, ys, inverse=inv, coset_x=cx, coset_y=cy)).all(), (curve, xs, ys, inv)
