"""gpu tier: BLS12-381 G1 MSM through the C ABI vs the oracle, bit-exact on the affine result.
Edge cases follow what the prover feeds encode_poly: zero scalars, repeated bases,
(0,0)=infinity bases, tiny sizes, every window width."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _msm_affine(gpu, s, p, **kw):
    return gpu.projective_to_affine_bytes(gpu.msm(s, p, **kw))


@pytest.mark.parametrize("n", [1, 2, 3, 17, 128, 257, 1000, 4099])
def test_msm_vs_oracle_sizes(gpu, oracle, n):
    s = oracle.fr_random(1000 + n, n)
    p = oracle.g1_random_bases(2000 + n, n)
    assert (_msm_affine(gpu, s, p) == oracle.g1_msm(s, p)).all()


@pytest.mark.parametrize("c", [2, 3, 5, 8, 11, 15, 16, 18])
def test_msm_every_window_width(gpu, oracle, c):
    n = 300
    s = oracle.fr_random(77, n)
    p = oracle.g1_random_bases(78, n)
    assert (_msm_affine(gpu, s, p, c=c) == oracle.g1_msm(s, p)).all()


def test_msm_edge_scalars_and_bases(gpu, oracle):
    n = 64
    sv = oracle.to_ints(oracle.fr_random(5, n), 32)
    sv[0:6] = [0, 1, oracle.R_MOD - 1, 2, (1 << 255) % oracle.R_MOD, 0x8000]
    sv[10:20] = [0] * 10
    s = oracle.to_bytes(sv, 32)
    p = oracle.g1_random_bases(6, n)
    p[96 * 7:96 * 8] = 0                               # infinity base
    p[96 * 9:96 * 10] = p[96 * 8:96 * 9]               # repeated base (same bucket -> doubling branch)
    sv2 = list(sv)
    sv2[9] = sv2[8]
    s2 = oracle.to_bytes(sv2, 32)
    p[96 * 12:96 * 13] = oracle.g1_neg(p[96 * 11:96 * 12].copy())   # P and -P
    sv2[12] = sv2[11]
    s2 = oracle.to_bytes(sv2, 32)
    for sc in (s, s2):
        assert (_msm_affine(gpu, sc, p) == oracle.g1_msm(sc, p)).all()
    # all-zero scalars -> infinity; all-same base and scalar 1 -> [n]P
    z = np.zeros(32 * n, np.uint8)
    assert (_msm_affine(gpu, z, p) == 0).all()
    ones = oracle.to_bytes([1] * n, 32)
    same = np.tile(p[:96], n)
    assert (_msm_affine(gpu, ones, same) == oracle.g1_scalar_mul(oracle.to_bytes([n], 32), p[:96].copy())).all()
    assert (_msm_affine(gpu, s[:0], p[:0], msm_size=0) == 0).all()


def test_msm_batch_shapes(gpu, oracle):
    # batch with shared / per-batch bases (libs/src/iotools/mod.rs:1239-1252) and n one-point MSMs (:1113-1151)
    n, batch = 50, 3
    s = oracle.fr_random(8, n * batch)
    p = oracle.g1_random_bases(9, n * batch)
    shared = _msm_affine(gpu, s, p[:96 * n].copy(), msm_size=n, batch=batch, shared_points=True)
    per = _msm_affine(gpu, s, p, msm_size=n, batch=batch, shared_points=False)
    for b in range(batch):
        sb = s[32 * n * b:32 * n * (b + 1)].copy()
        assert (shared[96 * b:96 * (b + 1)] == oracle.g1_msm(sb, p[:96 * n].copy())).all()
        assert (per[96 * b:96 * (b + 1)] == oracle.g1_msm(sb, p[96 * n * b:96 * n * (b + 1)].copy())).all()
    g = oracle.g1_generator()
    k = 20
    sc = oracle.fr_random(10, k)
    ones = _msm_affine(gpu, sc, g, msm_size=1, batch=k, shared_points=True)
    assert (ones == oracle.g1_batch_scalar_mul(sc, g)).all()


def test_commit_identity_on_fixed_tau_crs(gpu, oracle):
    # encode_poly(P) == [P(tau_x, tau_y)] G on a CRS sub-grid built like Sigma1.xy_powers
    # (setup/trusted-setup/src/main.rs:236-246; libs/src/iotools/mod.rs:2075-2099), CRS made on the GPU
    import json
    import os
    pins = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "pins.json")))
    R = oracle.R_MOD
    tx, ty = int(pins["tau_x"], 16), int(pins["tau_y"], 16)
    g = oracle.to_bytes([int(pins["fixed_tau_g1_x"], 16), int(pins["fixed_tau_g1_y"], 16)], 48)
    xs, ys = 33, 17
    mon = [pow(tx, i, R) * pow(ty, j, R) % R for i in range(xs) for j in range(ys)]
    d_mon = gpu.DeviceBuffer.from_host(oracle.to_bytes(mon, 32))
    crs = gpu.g1_batch_scalar_mul_device(d_mon, g, xs * ys)
    crs_host = crs.to_host()
    assert (crs_host[:96] == g).all()
    assert (crs_host[96 * ys:96 * (ys + 1)] == oracle.g1_scalar_mul(oracle.to_bytes([tx], 32), g)).all()
    coeffs = oracle.fr_random(12, xs * ys)
    val = sum(c * m for c, m in zip(oracle.to_ints(coeffs, 32), mon)) % R
    got = _msm_affine(gpu, gpu.DeviceBuffer.from_host(coeffs), crs)
    assert (got == oracle.g1_scalar_mul(oracle.to_bytes([val], 32), g)).all()


def test_msm_2_16_vs_oracle(gpu, oracle):
    # BASELINE.json configs[0] size: 2^16 points, oracle Pippenger on the host cores
    n = 1 << 16
    s = gpu.fr_random_device(0x746F6B616D616B01, n)
    h = gpu.fr_random_device(0x746F6B616D616B02, n)
    p = gpu.g1_batch_scalar_mul_device(h, oracle.g1_generator(), n)
    sh, ph = s.to_host(), p.to_host()
    assert (ph[:96 * 8] == oracle.g1_random_bases(0x746F6B616D616B02, 8)).all()
    assert (_msm_affine(gpu, s, p) == oracle.g1_msm(sh, ph)).all()


def test_msm_linearity_2_20(gpu, oracle):
    # size-independent properties at 2^20: MSM(a+b) == MSM(a) + MSM(b); MSM(k*a) == k*MSM(a)
    n = 1 << 20
    a, b = gpu.fr_random_device(61, n), gpu.fr_random_device(62, n)
    h = gpu.fr_random_device(63, n)
    p = gpu.g1_batch_scalar_mul_device(h, oracle.g1_generator(), n)
    ma, mb = _msm_affine(gpu, a, p), _msm_affine(gpu, b, p)
    mab = _msm_affine(gpu, gpu.vec_add(a, b), p)
    assert (mab == oracle.g1_add(ma, mb)).all()
    k = oracle.fr_random(64, 1)
    mka = _msm_affine(gpu, gpu.scalar_mul(k, a), p)
    assert (mka == oracle.g1_scalar_mul(k, ma)).all()


def test_msm_skewed_witness_like_scalars(gpu, oracle):
    """Real binding MSMs take raw witness values: mostly 0 / 1 / small -> one giant bucket in
    the low window, empty high windows.  Exercises the chunk head / tail fragments and the big-bucket combine."""
    import random
    rnd = random.Random(5)
    n = 30000
    vals = []
    for i in range(n):
        u = rnd.random()
        if u < 0.55:
            vals.append(1)
        elif u < 0.75:
            vals.append(0)
        elif u < 0.9:
            vals.append(rnd.randrange(1 << 16))
        elif u < 0.97:
            vals.append(rnd.randrange(1 << 128))
        else:
            vals.append(rnd.randrange(oracle.R_MOD))
    s = oracle.to_bytes(vals, 32)
    h = gpu.fr_random_device(91, n)
    p = gpu.g1_batch_scalar_mul_device(h, oracle.g1_generator(), n)
    ph = p.to_host()
    assert (_msm_affine(gpu, s, p) == oracle.g1_msm(s, ph)).all()
    # every scalar equal: a single bucket per window holds all n entries (n / 64 fragments -> big path)
    same = oracle.to_bytes([0xABCDEF] * n, 32)
    assert (_msm_affine(gpu, same, p) == oracle.g1_msm(same, ph)).all()
    # bucket boundaries exactly on chunk boundaries: 64 / 128 / 192 copies of three digits
    vals = [3] * 64 + [5] * 128 + [9] * 192 + [7] * 1
    s2 = oracle.to_bytes(vals, 32)
    assert (_msm_affine(gpu, s2, gpu.DeviceBuffer.from_host(ph[:96 * len(vals)].copy()), c=4)
            == oracle.g1_msm(s2, ph[:96 * len(vals)].copy())).all()


def test_msm_2_20_vs_oracle(gpu, oracle):
    # a production-size commit (4096 x 256 coefficients) against the oracle's Pippenger on the host cores
    n = 1 << 20
    s = gpu.fr_random_device(0x746F6B616D616B03, n)
    h = gpu.fr_random_device(0x746F6B616D616B04, n)
    p = gpu.g1_batch_scalar_mul_device(h, oracle.g1_generator(), n)
    assert (_msm_affine(gpu, s, p) == oracle.g1_msm(s.to_host(), p.to_host())).all()


def test_gathered_bases_msm(gpu, oracle):
    """binding commitments (encode_O_* -> msm_g1_bases, libs/src/group_structures/mod.rs:127-300): bases picked out of a
    resident CRS table by an index list, gathered on the device, then one MSM"""
    import random
    rnd = random.Random(3)
    table_n, n = 5000, 1500
    h = gpu.fr_random_device(81, table_n)
    table = gpu.g1_batch_scalar_mul_device(h, oracle.g1_generator(), table_n)
    idx = np.array([rnd.randrange(table_n) for _ in range(n)], np.uint32)
    idx[10] = idx[11]                                   # the same wire can appear twice
    bases = gpu.gather_rows_device(table, 96, idx)
    th = table.to_host().reshape(table_n, 96)
    want_bases = th[idx].reshape(-1).copy()
    assert (bases.to_host() == want_bases).all()
    s = oracle.to_bytes([rnd.choice([0, 1, 1, 1, rnd.randrange(1 << 128), rnd.randrange(oracle.R_MOD)]) for _ in range(n)], 32)
    assert (_msm_affine(gpu, s, bases) == oracle.g1_msm(s, want_bases)).all()


@pytest.mark.parametrize("c", [13, 16, 17, 18])
def test_msm_two_pass_sort_wide_windows(gpu, oracle, c):
    """n >= 2^18 takes the two-pass (coarse / fine) LDS-staged bucket sort; c = 17, 18 exist only there"""
    n = 1 << 18
    s = gpu.fr_random_device(0x1234 + c, n)
    h = gpu.fr_random_device(0x5678, n)
    p = gpu.g1_batch_scalar_mul_device(h, oracle.g1_generator(), n)
    want = oracle.g1_msm(s.to_host(), p.to_host())
    assert (_msm_affine(gpu, s, p, c=c) == want).all()


def test_msm_multi_matches_single_calls(gpu, oracle):
    # tkmk_msm_multi: independent jobs of ragged sizes (incl. 0 and 1) pipelined over internal streams; every
    # result must equal the oracle's and the single-call result; more jobs than streams exercises slot reuse
    sizes = [700, 0, 1, 33, 5000, 2, 1200, 64, 3000]
    jobs, want = [], []
    for k, n in enumerate(sizes):
        s = oracle.fr_random(300 + k, n) if n else np.zeros(0, np.uint8)
        p = oracle.g1_random_bases(400 + k, n) if n else np.zeros(0, np.uint8)
        jobs.append((s, p, n))
        want.append(oracle.g1_msm(s, p) if n else np.zeros(96, np.uint8))
    got = gpu.projective_to_affine_bytes(gpu.msm_multi(jobs))
    for k in range(len(sizes)):
        assert (got[96 * k:96 * (k + 1)] == want[k]).all(), f"job {k} (n={sizes[k]})"
    # device-resident inputs, twice in a row (arena reuse on the internal streams)
    dev_jobs = [(gpu.DeviceBuffer.from_host(s), gpu.DeviceBuffer.from_host(p), n) for s, p, n in jobs if n >= 2]
    ref = [w for w, n in zip(want, sizes) if n >= 2]
    for _ in range(2):
        got = gpu.projective_to_affine_bytes(gpu.msm_multi(dev_jobs))
        for k in range(len(dev_jobs)):
            assert (got[96 * k:96 * (k + 1)] == ref[k]).all()


def test_msm_batch_pipeline_large(gpu, oracle):
    # batch_size > 1 goes through the same pipeline; 2^18 points takes the two-pass sort on every stream
    n, batch = 1 << 18, 4
    sd = gpu.fr_random_device(91, n * batch)
    base = oracle.g1_generator()
    k = gpu.fr_random_device(92, n)
    pd = gpu.g1_batch_scalar_mul_device(k, base, n)
    got = gpu.projective_to_affine_bytes(gpu.msm(sd, pd, msm_size=n, batch=batch, shared_points=True))
    # Σ s_i·(k_i·G) = (Σ s_i·k_i)·G: check against the device dot product
    kk = oracle.to_ints(k.to_host(), 32)
    sh = sd.to_host()
    for b in range(batch):
        sv = oracle.to_ints(sh[32 * n * b:32 * n * (b + 1)], 32)
        dot = sum(x * y for x, y in zip(sv, kk)) % oracle.R_MOD
        assert (got[96 * b:96 * (b + 1)] == oracle.g1_scalar_mul(oracle.to_bytes([dot], 32), base.copy())).all()


def test_msm_giant_bucket_multi_workgroup_path(gpu, oracle):
    # witness MSMs put millions of "wire = 1" scalars into ONE bucket: above 2048 chunk fragments
    # (131072 entries) a bucket is summed by 32 workgroups + a final wave (k_combine_giant_*), below by one workgroup
    n = 300000
    k = gpu.fr_random_device(71, n)
    g = oracle.g1_generator()
    bases = gpu.g1_batch_scalar_mul_device(k, g, n)
    kv = oracle.to_ints(k.to_host(), 32)
    R = oracle.R_MOD
    sv = [1] * n
    for i in range(0, n, 7):
        sv[i] = 0
    for i in range(3, n, 1000):
        sv[i] = (i * 0x9E3779B97F4A7C15 + 12345) % R             # a sprinkle of full-width scalars
    for i in range(5, n, 50):
        sv[i] = 1 << 16                                           # a second giant-ish bucket in window 1 (c = 16)
    s = gpu.DeviceBuffer.from_host(oracle.to_bytes(sv, 32))
    got = _msm_affine(gpu, s, bases)
    dot = sum(a * b for a, b in zip(sv, kv)) % R
    assert (got == oracle.g1_scalar_mul(oracle.to_bytes([dot], 32), g.copy())).all()
    # same inputs with an explicit narrow window (more windows, same giant bucket in window 0)
    assert (_msm_affine(gpu, s, bases, c=13) == got).all()


@pytest.mark.parametrize("factor,c", [(2, 0), (3, 7), (8, 0), (16, 16), (100, 0)])
def test_msm_precomputed_bases(gpu, oracle, factor, c):
    """precompute_factor (ICICLE msm_precompute_bases): the expanded table 2^(c W' j) P_i gives the same result as the
    plain MSM for factors that divide the window count, that do not (padding windows), and that exceed it (clamped)"""
    n = 700
    s = oracle.fr_random(500 + factor, n)
    sv = oracle.to_ints(s, 32)
    sv[:4] = [0, 1, oracle.R_MOD - 1, 1 << 254]
    s = oracle.to_bytes(sv, 32)
    p = oracle.g1_random_bases(600 + factor, n)
    p[96 * 5:96 * 6] = 0                                   # an infinity base stays infinity in every table row
    want = oracle.g1_msm(s, p)
    table = gpu.msm_precompute_bases(p, n, factor, c=c)
    rows = min(factor, gpu.msm_windows(n, c))
    assert table.nbytes == 96 * n * rows
    got = _msm_affine(gpu, s, table, msm_size=n, c=c, precompute_factor=factor)
    assert (got == want).all()
    # device-resident scalars, twice (table reuse), and through the pipelined multi entry with two jobs
    sd = gpu.DeviceBuffer.from_host(s)
    for _ in range(2):
        assert (_msm_affine(gpu, sd, table, msm_size=n, c=c, precompute_factor=factor) == want).all()
    s2 = oracle.fr_random(700 + factor, n)
    res = gpu.projective_to_affine_bytes(gpu.msm_multi([(sd, table, n), (gpu.DeviceBuffer.from_host(s2), table, n)], c=c,
                                                       precompute_factor=factor))
    assert (res[:96] == want).all() and (res[96:] == oracle.g1_msm(s2, p)).all()


def test_msm_precomputed_bases_large_identity(gpu, oracle):
    # 2^18 points, full precompute (one window group): sum s_i (k_i G) == (sum s_i k_i) G
    n = 1 << 18
    k = gpu.fr_random_device(93, n)
    g = oracle.g1_generator()
    bases = gpu.g1_batch_scalar_mul_device(k, g, n)
    sd = gpu.fr_random_device(94, n)
    f = gpu.msm_windows(n)
    table = gpu.msm_precompute_bases(bases, n, f)
    got = _msm_affine(gpu, sd, table, msm_size=n, precompute_factor=f)
    assert (got == _msm_affine(gpu, sd, bases)).all()
    kv, sv = oracle.to_ints(k.to_host(), 32), oracle.to_ints(sd.to_host(), 32)
    dot = sum(a * b for a, b in zip(kv, sv)) % oracle.R_MOD
    assert (got == oracle.g1_scalar_mul(oracle.to_bytes([dot], 32), g.copy())).all()


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_msm_montgomery_form_inputs(gpu, oracle, curve):
    # MSMConfig::are_scalars_montgomery_form / are_points_montgomery_form (ICICLE: x * 2^(32 limbs) mod modulus): the same
    # MSM with either or both operands handed over in Montgomery form, also through the precompute entry
    o = oracle if curve == "bls12_381" else oracle.bn254
    fq_bytes = 48 if curve == "bls12_381" else 32
    aff = 2 * fq_bytes
    n = 500
    s = o.fr_random(801, n)
    p = o.g1_random_bases(802, n)
    p[aff * 3:aff * 4] = 0                                                   # infinity stays (0, 0) in either form
    want = o.g1_msm(s, p)
    R_MOD, P_MOD = o.R_MOD, o.P_MOD
    sm = oracle.to_bytes([(x << 256) % R_MOD for x in oracle.to_ints(s, 32)], 32)
    pm = oracle.to_bytes([(x << (8 * fq_bytes)) % P_MOD for x in oracle.to_ints(p, fq_bytes)], fq_bytes)

    def run(sc, pt, **kw):
        return gpu.projective_to_affine_bytes(gpu.msm(sc, pt, curve=curve, **kw), curve=curve)

    assert (run(sm, p, scalars_montgomery=True) == want).all()
    assert (run(s, pm, points_montgomery=True) == want).all()
    assert (run(sm, pm, scalars_montgomery=True, points_montgomery=True) == want).all()
    table = gpu.msm_precompute_bases(pm, n, 4, curve=curve, points_montgomery=True)   # Montgomery bases into the precompute table
    assert (run(sm, table, msm_size=n, precompute_factor=4, scalars_montgomery=True) == want).all()
    one = oracle.to_bytes([(1 << 256) % R_MOD], 32)                              # Montgomery 1: [1]P through the size-1 path
    got1 = run(one, pm[:aff].copy(), msm_size=1, scalars_montgomery=True, points_montgomery=True)
    assert (got1 == p[:aff]).all()
