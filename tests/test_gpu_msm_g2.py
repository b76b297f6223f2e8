"""GPU parity tier: G2 MSM (bls12_381_g2_msm, tokamak-zk-evm_amd/csrc/msm_g2.hip) against the oracle's G2 instantiation (oracle.g2,
itself pinned on the independent big-int G2 of tkmk/g2.py and the reference's fixed G2 generator in tests/test_oracle_pins.py).
The reference has no G2 MSM call site (Sigma2::gen, libs/src/group_structures/mod.rs:752-777, multiplies the generator nine times);
the last test runs exactly that shape through the entry.  Bit-exact: results are canonical affine coordinates."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def _affine(proj):
    """288-byte canonical projective results -> 192-byte affine records (all zero = infinity)"""
    proj = np.asarray(proj, np.uint8).reshape(-1, 288)
    out = np.zeros((proj.shape[0], 192), np.uint8)
    for k, rec in enumerate(proj):
        if rec[192:].any():
            assert rec[192] == 1 and not rec[193:].any(), "z must be 1"
            out[k] = rec[:192]
        else:
            assert rec[96] == 1 and not rec[:96].any() and not rec[97:192].any(), "infinity must be (0, 1, 0)"
    return out


def _scalars(vals):
    return np.frombuffer(b"".join(int(v % (1 << 256)).to_bytes(32, "little") for v in vals), np.uint8).copy()


@pytest.mark.parametrize("n", [1, 2, 3, 31, 32, 33, 100, 1000, 4097, 1 << 14])
def test_g2_msm_vs_oracle(gpu, oracle, n):
    s = oracle.fr_random(900 + n, n)
    p = oracle.g2.random_bases(40 + n, n)
    want = oracle.g2.msm(s, p)
    assert oracle.g2.on_curve(want)
    got = _affine(gpu.msm_g2(s, p))[0]
    assert (got == want).all()
    # device-resident operands, explicit window widths (signed digits at the top window, the carry out of it)
    ds, dp = gpu.DeviceBuffer.from_host(s), gpu.DeviceBuffer.from_host(p)
    for c in (2, 5, 12) if n <= 4097 else (11,):
        assert (_affine(gpu.msm_g2(ds, dp, c=c))[0] == want).all(), c


def test_g2_msm_edge_cases(gpu, oracle):
    n = 72                                          # (the oracle's naive G2 MSM is what this test's time goes to)
    p = oracle.g2.random_bases(7, n).reshape(n, 192).copy()
    vals = [0, 1, R - 1, R - 2, 2, (1 << 255) % R, (1 << 254), R >> 1] + [int.from_bytes(bytes(oracle.fr_random(5, n)[32 * i:32 * i + 32]), "little") for i in range(8, n)]
    p[3] = 0                                        # an infinity record among the bases
    p[10] = p[11]                                   # the same point twice: doubling inside a bucket when the digits agree
    vals[10] = vals[11] = 12345
    p[20] = oracle.g2.neg(p[21].copy())             # P and -P with the same scalar: cancels to infinity inside a bucket
    vals[20] = vals[21] = 777
    s = _scalars(vals)
    want = oracle.g2.msm_naive(s, p.reshape(-1))
    assert (_affine(gpu.msm_g2(s, p.reshape(-1).copy()))[0] == want).all()
    # all-zero scalars, all-infinity bases, empty MSM -> (0, 1, 0)
    zero = np.zeros(192, np.uint8)
    assert (_affine(gpu.msm_g2(np.zeros(32 * n, np.uint8), p.reshape(-1).copy()))[0] == zero).all()
    assert (_affine(gpu.msm_g2(s, np.zeros(192 * n, np.uint8)))[0] == zero).all()
    assert (_affine(gpu.msm_g2(np.zeros(0, np.uint8), np.zeros(0, np.uint8), msm_size=0))[0] == zero).all()
    # one heavily repeated scalar: one giant bucket per window (the queued big-bucket path)
    n2 = 1500
    p2 = oracle.g2.random_bases(8, n2)
    s2 = np.tile(_scalars([0x0123456789ABCDEF0123456789ABCDEF0123456789ABCDEF0123456789ABCDEF % R]), n2)
    assert (_affine(gpu.msm_g2(s2, p2))[0] == oracle.g2.msm(s2, p2)).all()
    # bitsize: only the low bits count when the caller promises short scalars
    s3 = _scalars([v & 0xFFFF for v in vals])
    assert (_affine(gpu.msm_g2(s3, p.reshape(-1).copy(), bitsize=16))[0] == oracle.g2.msm_naive(s3, p.reshape(-1))).all()


def test_g2_msm_batches_and_montgomery_flags(gpu, oracle):
    n, batch = 64, 3
    s = oracle.fr_random(61, n * batch)
    p = oracle.g2.random_bases(62, n * batch)
    shared = _affine(gpu.msm_g2(s, p[:192 * n].copy(), msm_size=n, batch=batch, shared_points=True))
    own = _affine(gpu.msm_g2(s, p, msm_size=n, batch=batch, shared_points=False))
    for b in range(batch):
        sb = s[32 * n * b:32 * n * (b + 1)]
        assert (shared[b] == oracle.g2.msm(sb, p[:192 * n])).all()
        assert (own[b] == oracle.g2.msm(sb, p[192 * n * b:192 * n * (b + 1)])).all()
    # Montgomery-form inputs: x * 2^256 mod r for scalars, c * 2^384 mod p for every base-field coordinate
    P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
    sm = _scalars([int.from_bytes(bytes(s[32 * i:32 * i + 32]), "little") * (1 << 256) % R for i in range(n)])
    pm = np.frombuffer(b"".join((int.from_bytes(bytes(p[48 * i:48 * i + 48]), "little") * (1 << 384) % P).to_bytes(48, "little") for i in range(4 * n)), np.uint8).copy()
    want = oracle.g2.msm(s[:32 * n], p[:192 * n])
    assert (_affine(gpu.msm_g2(sm, pm, scalars_montgomery=True, points_montgomery=True))[0] == want).all()


def test_g2_msm_rejects_bad_configs(gpu, oracle):
    s, p = oracle.fr_random(1, 4), oracle.g2.random_bases(1, 4)
    for kw in ({"c": 13}, {"c": 1}, {"bitsize": 256}, {"batch": 0}):
        with pytest.raises(gpu.TkmkError):
            gpu.msm_g2(s, p, msm_size=4, **kw)


def test_sigma2_gen_shape_through_the_entry(gpu, oracle, tkmk):
    """Sigma2::gen (group_structures/mod.rs:752-777): nine multiples of H, here as a batch of one-point MSMs with shared points,
    against the host-side big-int G2 the setup uses"""
    from tkmk import g2
    import json, os
    pins = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "pins.json")))
    h = g2.from_hex_pair(pins["fixed_tau_g2_x"], pins["fixed_tau_g2_y"])
    import random
    rnd = random.Random(9)
    ks = [rnd.randrange(R) for _ in range(9)]
    got = _affine(gpu.msm_g2(_scalars(ks), np.asarray(g2.encode(h), np.uint8).copy(), msm_size=1, batch=9, shared_points=True))
    for k, rec in zip(ks, got):
        assert (rec == np.asarray(g2.encode(g2.scalar_mul(k, h)), np.uint8)).all()
