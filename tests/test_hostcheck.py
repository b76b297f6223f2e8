"""not-gpu tier: the product's __host__ __device__ field / curve templates (csrc/ff.h, csrc/ec.h), compiled
for the HOST into a test-only library (tests/hostcheck), agree with the oracle bit for bit.  This is the
same source the HIP kernels inline, so arithmetic bugs are caught before a GPU is involved."""
import ctypes
import os
import random
import shutil
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "tokamak-zk-evm_amd", "csrc")


@pytest.fixture(scope="module")
def hc():
    import importlib.util
    spec = importlib.util.spec_from_file_location("hostcheck_build", os.path.join(HERE, "hostcheck", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    so = mod.build()
    if so is None:
        pytest.skip("hipcc not available")
    return ctypes.CDLL(so)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _edge(mod, n, rnd):
    v = [rnd.randrange(mod) for _ in range(n)]
    v[:6] = [0, 1, mod - 1, mod - 2, 2, (1 << (mod.bit_length() - 1))]
    return v


@pytest.mark.parametrize("field", ["fr", "fq"])
def test_field_templates(hc, oracle, field):
    rnd = random.Random(3)
    width, mod = (32, oracle.R_MOD) if field == "fr" else (48, oracle.P_MOD)
    fn = getattr(hc, "hc_%s_op" % field)
    n = 200
    a, b = _edge(mod, n, rnd), _edge(mod, n, rnd)
    rnd.shuffle(b)
    A, B = oracle.to_bytes(a, width), oracle.to_bytes(b, width)
    out = np.empty_like(A)
    o_add, o_sub, o_mul, o_inv = [getattr(oracle, "%s_%s" % (field, k)) for k in ("add", "sub", "mul", "inv")]
    for op, want in ((0, o_add(A, B)), (1, o_sub(A, B)), (2, o_mul(A, B)), (3, o_mul(A, B)), (4, o_inv(A)),
                     (5, o_sub(np.zeros_like(A), A))):
        fn(op, _p(A), _p(B), _p(out), ctypes.c_size_t(n))
        assert (out == want).all(), "op %d" % op


def test_g1_templates(hc, oracle):
    g = oracle.g1_generator()
    pts = oracle.g1_random_bases(42, 6)
    P = [pts[96 * i:96 * (i + 1)].copy() for i in range(6)]
    zero = np.zeros(96, np.uint8)
    out = np.empty(96, np.uint8)
    cases = [(P[0], P[1]), (P[2], P[2]), (P[3], oracle.g1_neg(P[3])), (zero, P[4]), (P[4], zero), (zero, zero), (g, g)]
    for p, q in cases:
        s = oracle.g1_add(p, q)
        for mode in (0, 1):
            hc.hc_g1_add(mode, _p(p), _p(q), _p(out))
            assert (out == s).all(), mode
        hc.hc_g1_add(2, _p(p), _p(q), _p(out))
        assert (out == oracle.g1_add(s, s)).all()
        hc.hc_g1_add(3, _p(p), _p(q), _p(out))
        assert (out == oracle.g1_add(s, q)).all()
    for k in (0, 1, 2, oracle.R_MOD - 1, 0x1234567890ABCDEF1234567890ABCDEF):
        K = oracle.to_bytes([k], 32)
        hc.hc_g1_scalar_mul(_p(K), _p(P[5]), _p(out))
        assert (out == oracle.g1_scalar_mul(K, P[5])).all()


def _hc_ntt(hc, x, logn, batch, columns, inverse, coset, max_logR, log_tile, logN):
    out = np.empty_like(x)
    import oracle
    rc = hc.hc_ntt(_p(x), logn, ctypes.c_uint64(batch), int(columns), int(inverse), _p(coset) if coset is not None else None,
                   _p(out), max_logR, log_tile, logN, _p(oracle.root_of_unity(1 << 32)))
    assert rc > 0
    return out, rc


@pytest.mark.parametrize("logn,batch,max_logR,log_tile", [
    (0, 3, 3, 5), (1, 5, 3, 5), (3, 1, 3, 5), (3, 7, 3, 5), (5, 4, 3, 5), (6, 3, 3, 5), (7, 2, 3, 5), (9, 1, 3, 5),
    (8, 3, 4, 6), (10, 2, 9, 11), (11, 1, 9, 11), (12, 2, 9, 11),
])
def test_ntt_pass_plan_emulation(hc, oracle, logn, batch, max_logR, log_tile):
    """The kernel's plan / index algebra (csrc/ntt_plan.h) reproduces the oracle NTT for 1..3-pass plans, both
    layouts, both directions, with and without coset, with a domain larger than the transform."""
    n = 1 << logn
    x = oracle.fr_random(1000 + logn, n * batch)
    g = oracle.to_bytes([0x9E3779B97F4A7C15F39CC0605CEDC834], 32)
    for columns in (False, True):
        for inverse in (False, True):
            for coset in (None, g):
                want = oracle.ntt(x, n, batch=batch, columns_batch=columns, inverse=inverse, coset_gen=coset)
                got, passes = _hc_ntt(hc, x, logn, batch, columns, inverse, coset, max_logR, log_tile, logn + 2)
                assert passes == max(1, -(-logn // max_logR))
                assert (got == want).all(), (columns, inverse, coset is not None)


def test_unsaturated_field_matches_oracle(hc, oracle):
    """csrc/ffu.h (29-bit limbs, carry-free products) with all bound assertions on: mul / sqr / add / sub vs the oracle"""
    rnd = random.Random(17)
    mod = oracle.P_MOD
    n = 400
    a, b = _edge(mod, n, rnd), _edge(mod, n, rnd)
    rnd.shuffle(b)
    a += [mod - 1] * 4 + [(1 << 380) - 1, (1 << 377) - 1]
    b += [mod - 1, 1, 0, mod - 2, (1 << 380) - 1, (1 << 29) - 1]
    n = len(a)
    A, B = oracle.to_bytes(a, 48), oracle.to_bytes(b, 48)
    out = np.empty_like(A)
    for op, want in ((0, oracle.fq_mul(A, B)), (1, oracle.fq_mul(A, A)), (2, oracle.fq_add(A, B)), (3, oracle.fq_sub(A, B)), (4, A)):
        hc.hc_fqu_op(op, _p(A), _p(B), _p(out), ctypes.c_size_t(n))
        assert (out == want).all(), "op %d" % op


def test_unsaturated_mixed_add_chain_matches_oracle(hc, oracle):
    """csrc/ec_u.h: a bucket-like accumulation chain incl. infinity records, a repeated point (doubling branch), P then -P
    (cancels to infinity, then continues) — same result as the oracle's group law"""
    pts = oracle.g1_random_bases(77, 40)
    P = [pts[96 * i:96 * (i + 1)].copy() for i in range(40)]
    seq = P[:10] + [np.zeros(96, np.uint8)] + [P[10], P[10]] + P[11:20] + [P[3]] + P[20:]
    neg = [0] * len(seq)
    neg[4] = 1
    neg[15] = 1
    # make the running sum hit infinity: append the negation of everything so far, then continue
    acc = np.zeros(96, np.uint8)
    for q, s in zip(seq, neg):
        acc = oracle.g1_add(acc, oracle.g1_neg(q) if s else q)
    seq2 = seq + [acc] + [P[0], P[0], P[1]]
    neg2 = neg + [1] + [0, 1, 0]
    want = np.zeros(96, np.uint8)
    for q, s in zip(seq2, neg2):
        want = oracle.g1_add(want, oracle.g1_neg(q) if s else q)
    buf = np.concatenate(seq2)
    flags = np.array(neg2, np.uint8)
    out = np.empty(96, np.uint8)
    hc.hc_g1u_accumulate(_p(buf), _p(flags), ctypes.c_size_t(len(seq2)), _p(out))
    assert (out == want).all()
    assert (want == P[1]).all()


# ---- BN254 instantiation of the same templates (8 saturated limbs, 10 x 28-bit unsaturated limbs) ----
@pytest.mark.parametrize("field", ["fr", "fq"])
def test_bn254_field_templates(hc, oracle, field):
    bn = oracle.bn254
    rnd = random.Random(5)
    mod = bn.R_MOD if field == "fr" else bn.P_MOD
    fn = getattr(hc, "hc_bn254_%s_op" % field)
    n = 200
    a, b = _edge(mod, n, rnd), _edge(mod, n, rnd)
    rnd.shuffle(b)
    A, B = oracle.to_bytes(a, 32), oracle.to_bytes(b, 32)
    out = np.empty_like(A)
    o_add, o_sub, o_mul, o_inv = [getattr(bn, "%s_%s" % (field, k)) for k in ("add", "sub", "mul", "inv")]
    for op, want in ((0, o_add(A, B)), (1, o_sub(A, B)), (2, o_mul(A, B)), (3, o_mul(A, B)), (4, o_inv(A)),
                     (5, o_sub(np.zeros_like(A), A))):
        fn(op, _p(A), _p(B), _p(out), ctypes.c_size_t(n))
        assert (out == want).all(), "op %d" % op
    # and against plain Python integers (the oracle's BN254 fields are new code too)
    assert oracle.to_ints(o_mul(A, B), 32) == [x * y % mod for x, y in zip(a, b)]
    assert oracle.to_ints(o_inv(A), 32) == [pow(x, -1, mod) if x else 0 for x in a]


def test_bn254_unsaturated_field_and_chain(hc, oracle):
    bn = oracle.bn254
    rnd = random.Random(19)
    mod = bn.P_MOD
    a, b = _edge(mod, 400, rnd), _edge(mod, 400, rnd)
    rnd.shuffle(b)
    a += [mod - 1] * 4 + [(1 << 253) - 1, (1 << 252) - 1]
    b += [mod - 1, 1, 0, mod - 2, (1 << 253) - 1, (1 << 28) - 1]
    n = len(a)
    A, B = oracle.to_bytes(a, 32), oracle.to_bytes(b, 32)
    out = np.empty_like(A)
    for op, want in ((0, bn.fq_mul(A, B)), (1, bn.fq_mul(A, A)), (2, bn.fq_add(A, B)), (3, bn.fq_sub(A, B)), (4, A)):
        hc.hc_bn254_fqu_op(op, _p(A), _p(B), _p(out), ctypes.c_size_t(n))
        assert (out == want).all(), "op %d" % op
    # accumulation chain with infinity records, a doubling, a cancellation to infinity and a restart
    pts = bn.g1_random_bases(78, 40)
    P = [pts[64 * i:64 * (i + 1)].copy() for i in range(40)]
    seq = P[:10] + [np.zeros(64, np.uint8)] + [P[10], P[10]] + P[11:20] + [P[3]] + P[20:]
    neg = [0] * len(seq)
    neg[4] = neg[15] = 1
    acc = np.zeros(64, np.uint8)
    for q, s in zip(seq, neg):
        acc = bn.g1_add(acc, bn.g1_neg(q) if s else q)
    seq2 = seq + [acc] + [P[0], P[0], P[1]]
    neg2 = neg + [1] + [0, 1, 0]
    flags = np.array(neg2, np.uint8)
    buf = np.concatenate(seq2)
    out = np.empty(64, np.uint8)
    hc.hc_bn254_g1u_accumulate(_p(buf), _p(flags), ctypes.c_size_t(len(seq2)), _p(out))
    assert (out == P[1]).all()
    # a long random chain keeps every bound assertion of ffu / ecu satisfied
    pts = bn.g1_random_bases(79, 300)
    flags = np.array([rnd.randrange(2) for _ in range(300)], np.uint8)
    want = np.zeros(64, np.uint8)
    for i in range(300):
        q = pts[64 * i:64 * (i + 1)].copy()
        want = bn.g1_add(want, bn.g1_neg(q) if flags[i] else q)
    hc.hc_bn254_g1u_accumulate(_p(pts), _p(flags), ctypes.c_size_t(300), _p(out))
    assert (out == want).all()


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_unsaturated_full_add_and_doubling(hc, oracle, curve):
    """csrc/ec_u.h add / dbl (tail kernels of the MSM) with every bound assertion on, incl. the doubling and cancellation
    branches reached through projectively different representations of the same point, and a double-and-add chain"""
    o = oracle if curve == "bls12_381" else oracle.bn254
    fn = hc.hc_g1u_full if curve == "bls12_381" else hc.hc_bn254_g1u_full
    sz = 96 if curve == "bls12_381" else 64
    for seed in (5, 6, 7):
        pts = o.g1_random_bases(seed, 5)
        P = [pts[sz * i:sz * (i + 1)].copy() for i in range(5)]
        A = o.g1_add(o.g1_add(P[0], P[1]), P[2])
        B = o.g1_add(P[3], P[4])
        out = np.empty(sz, np.uint8)
        k = 0xB5C3 + seed
        want = {0: o.g1_add(A, B), 1: o.g1_add(A, A), 2: o.g1_add(A, A), 3: np.zeros(sz, np.uint8),
                4: o.g1_scalar_mul(oracle.to_bytes([k], 32), A)}
        for mode, w in want.items():
            fn(mode, _p(pts), ctypes.c_uint32(k), _p(out))
            assert (out == w).all(), (curve, seed, mode)
