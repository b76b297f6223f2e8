"""GPU parity tier: the NTT over G1 points (tkmk_g1_ntt, csrc/g1ntt.hip) and what it is for — the Lagrange-basis CRS.  No reference
counterpart (the reference commits coefficients only); checked "in the exponent": the points are [h_ab] G with known h, so the transform
of the points must be [NTT(h)] G with the scalar NTT of the oracle, and the commitment of a polynomial from its evaluations over the
transformed CRS grid must be the point encode_poly gives from its coefficients (libs/src/iotools/mod.rs:2041-2113)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _points(gpu, oracle, seed, n):
    h = gpu.fr_random_device(seed, n)
    pts = gpu.g1_batch_scalar_mul_device(h, oracle.g1_generator(), n)
    return np.asarray(h.to_host()), pts


@pytest.mark.parametrize("xs,ys", [(1, 1), (2, 1), (1, 4), (4, 2), (8, 16), (64, 32)])
def test_g1_ntt_in_the_exponent(gpu, oracle, xs, ys):
    n = xs * ys
    gpu.init_ntt_domain_for_size(1 << 12)
    h, pts = _points(gpu, oracle, 500 + n, n)
    R = oracle.R_MOD
    for inverse in (False, True):
        got = np.asarray(gpu.g1_ntt(pts, xs, ys, inverse=inverse).to_host())
        want_scalars = oracle.bintt(h, xs, ys, inverse=inverse)
        if inverse:      # the group transform is not scaled by 1/N
            want_scalars = oracle.fr_scalar_mul(oracle.to_bytes([n % R], 32), want_scalars)
        want = np.asarray(oracle.g1_batch_scalar_mul(want_scalars, np.tile(oracle.g1_generator(), n)))
        assert (got == want).all(), (xs, ys, inverse)
    # forward then inverse = N * identity
    back = gpu.g1_ntt(gpu.g1_ntt(pts, xs, ys), xs, ys, inverse=True).to_host()
    P = np.asarray(pts.to_host()).reshape(n, 96)
    nb = oracle.to_bytes([n % R], 32)
    step = max(1, n // 64)                                  # every point for small grids, a sample of 64 for the large one
    for i in range(0, n, step):
        assert (np.asarray(back)[96 * i:96 * i + 96] == np.asarray(oracle.g1_scalar_mul(nb, P[i]))).all(), i


def test_g1_ntt_forms_strides_and_errors(gpu, oracle):
    xs, ys, stride = 8, 4, 7
    gpu.init_ntt_domain_for_size(1 << 12)
    h, pts = _points(gpu, oracle, 77, xs * stride)
    full = np.asarray(pts.to_host()).reshape(xs, stride, 96)
    sub = gpu.DeviceBuffer.from_host(np.ascontiguousarray(full[:, :ys]).reshape(-1))
    want = np.asarray(gpu.g1_ntt(sub, xs, ys, inverse=True).to_host())
    assert (np.asarray(gpu.g1_ntt(pts, xs, ys, inverse=True, in_stride=stride).to_host()) == want).all()            # strided sub-grid
    conv = gpu.msm_convert_bases(pts, xs * stride)
    assert (np.asarray(gpu.g1_ntt(conv, xs, ys, inverse=True, in_stride=stride, bases_form=gpu.BASES_CONVERTED).to_host()) == want).all()
    z = np.asarray(pts.to_host()).copy()
    z[96 * 3:96 * 4] = 0                                                                                                 # a point at infinity among the inputs
    hz = h.copy()
    hz[32 * 3:32 * 4] = 0
    got = np.asarray(gpu.g1_ntt(gpu.DeviceBuffer.from_host(z), xs, stride // 7 * 4, in_stride=stride).to_host())
    hs = np.ascontiguousarray(hz.reshape(xs, stride, 32)[:, :4]).reshape(-1)
    want2 = oracle.g1_batch_scalar_mul(oracle.bintt(hs, xs, 4), np.tile(oracle.g1_generator(), xs * 4))
    assert (got == np.asarray(want2)).all()
    for bad in (dict(x_size=6, y_size=4), dict(x_size=8, y_size=4, in_stride=3), dict(x_size=8, y_size=4, bases_form=5)):
        with pytest.raises(gpu.TkmkError):
            gpu.g1_ntt(pts, bad["x_size"], bad["y_size"], in_stride=bad.get("in_stride", stride), bases_form=bad.get("bases_form", 0))


def test_commit_from_evaluations_equals_commit_from_coefficients(gpu, oracle):
    """(1/N) MSM(evaluations, inverse group NTT of the CRS grid) == MSM(coefficients, CRS grid) — with sparse / small evaluations,
    the case the Lagrange-basis CRS exists for"""
    xs, ys = 32, 16
    n = xs * ys
    gpu.init_ntt_domain_for_size(1 << 12)
    R = oracle.R_MOD
    _, crs = _points(gpu, oracle, 901, n)
    lam = gpu.g1_ntt(crs, xs, ys, inverse=True)
    rng = np.random.default_rng(5)
    vals = [0] * n
    for k in rng.choice(n, n // 3, replace=False):
        vals[int(k)] = int(rng.integers(0, 1 << 20)) if rng.random() < 0.6 else int.from_bytes(rng.bytes(31), "little")
    ev = oracle.to_bytes(vals, 32)
    coeffs = oracle.bintt(ev, xs, ys, inverse=True)
    from_coeffs = gpu.projective_to_affine_bytes(gpu.msm(coeffs, crs))
    s_ev = gpu.projective_to_affine_bytes(gpu.msm(ev, lam))
    inv_n = oracle.to_bytes([pow(n, R - 2, R)], 32)
    assert (np.asarray(oracle.g1_scalar_mul(inv_n, s_ev)) == from_coeffs).all()


@pytest.mark.parametrize("rows,cols,transposed", [(1, 1, False), (5, 3, False), (5, 3, True), (64, 33, True), (300, 70, False), (128, 64, True)])
def test_g1_prefix_sums_in_the_exponent(gpu, oracle, rows, cols, transposed):
    n = rows * cols
    h, pts = _points(gpu, oracle, 700 + n, n)
    R = oracle.R_MOD
    hv = oracle.to_ints(h, 32)
    order = [(j % rows) * cols + j // rows for j in range(n)] if transposed else list(range(n))
    run, want = 0, []
    for j in range(n):
        run = (run + hv[order[j]]) % R
        want.append(run)
    want_pts = np.asarray(oracle.g1_batch_scalar_mul(oracle.to_bytes(want, 32), np.tile(oracle.g1_generator(), n)))
    got = np.asarray(gpu.g1_prefix_sums(pts, rows, cols, transposed=transposed).to_host())
    assert (got == want_pts).all()
    conv = gpu.msm_convert_bases(pts, n)
    assert (np.asarray(gpu.g1_prefix_sums(conv, rows, cols, transposed=transposed, bases_form=gpu.BASES_CONVERTED).to_host()) == want_pts).all()


def test_piecewise_constant_vector_commits_through_its_jumps(gpu, oracle):
    """sum_j r_j L_j == sum_j (r_j - r_{j+1}) S_j with S = prefix sums of L: the identity prove1's commitment of R uses"""
    n = 4096
    _, lam = _points(gpu, oracle, 811, n)
    pre = gpu.g1_prefix_sums(lam, n, 1)
    R = oracle.R_MOD
    rng = np.random.default_rng(9)
    jumps = sorted(int(v) for v in rng.choice(n - 1, 40, replace=False))
    vals, cur = [], int.from_bytes(rng.bytes(31), "little")
    for j in range(n):
        vals.append(cur)
        if j in jumps:
            cur = int.from_bytes(rng.bytes(31), "little")
    diffs = [(vals[j] - (vals[j + 1] if j + 1 < n else 0)) % R for j in range(n)]
    assert sum(1 for d in diffs if d) <= len(jumps) + 1
    direct = gpu.projective_to_affine_bytes(gpu.msm(oracle.to_bytes(vals, 32), lam))
    through = gpu.projective_to_affine_bytes(gpu.msm(oracle.to_bytes(diffs, 32), pre))
    assert (direct == through).all()


@pytest.mark.parametrize("xs,ys", [(8, 4), (16, 16)])
def test_g1_ntt_one_axis_at_a_time(gpu, oracle, xs, ys):
    """tkmk_g1_ntt_axes: the X pass alone is ys independent transforms along the columns, the Y pass alone xs along the rows (in the
    exponent against the oracle's 1-D batches); one after the other, in either order, they are the bivariate transform — what a sharded
    context runs over its own columns and its own rows with a change of layout in between"""
    n = xs * ys
    gpu.init_ntt_domain_for_size(1 << 12)
    h, pts = _points(gpu, oracle, 900 + n, n)
    G = np.tile(oracle.g1_generator(), n)
    R = oracle.R_MOD
    both = np.asarray(gpu.g1_ntt(pts, xs, ys, inverse=True).to_host())
    y_only = gpu.g1_ntt(pts, xs, ys, inverse=True, axes=gpu.G1_NTT_AXIS_Y)
    x_only = gpu.g1_ntt(pts, xs, ys, inverse=True, axes=gpu.G1_NTT_AXIS_X)
    want_y = oracle.fr_scalar_mul(oracle.to_bytes([ys % R], 32), oracle.ntt(h, ys, batch=xs, inverse=True))                       # unscaled inverse
    want_x = oracle.fr_scalar_mul(oracle.to_bytes([xs % R], 32), oracle.ntt(h, xs, batch=ys, columns_batch=True, inverse=True))
    assert (np.asarray(y_only.to_host()) == np.asarray(oracle.g1_batch_scalar_mul(want_y, G))).all()
    assert (np.asarray(x_only.to_host()) == np.asarray(oracle.g1_batch_scalar_mul(want_x, G))).all()
    assert (np.asarray(gpu.g1_ntt(y_only, xs, ys, inverse=True, axes=gpu.G1_NTT_AXIS_X).to_host()) == both).all()
    assert (np.asarray(gpu.g1_ntt(x_only, xs, ys, inverse=True, axes=gpu.G1_NTT_AXIS_Y).to_host()) == both).all()
    with pytest.raises(gpu.TkmkError):
        gpu.g1_ntt(pts, xs, ys, axes=4)


def test_g1_scale_every_point_by_one_scalar(gpu, oracle):
    """tkmk_g1_scale: out[i] = [s] in[i] against the oracle's scalar multiplication — a full-size scalar, 1, 0, the 1 / N a Lagrange table
    is scaled by, a point at infinity in the input, and in place"""
    n, R = 67, oracle.R_MOD
    h, pts = _points(gpu, oracle, 91, n)
    P = np.asarray(pts.to_host()).copy().reshape(n, 96)
    P[5] = 0                                               # (0, 0) = infinity stays infinity
    src = gpu.DeviceBuffer.from_host(P.reshape(-1))
    for s in (0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF % R, 1, 0, pow(4096 * 1024, -1, R), R - 1):
        sb = oracle.to_bytes([s], 32)
        got = np.asarray(gpu.g1_scale(src, n, sb).to_host()).reshape(n, 96)
        for i in range(n):
            want = np.zeros(96, np.uint8) if (i == 5 or s == 0) else np.asarray(oracle.g1_scalar_mul(sb, P[i]))
            assert (got[i] == want).all(), (hex(s), i)
    sb = oracle.to_bytes([7], 32)
    inplace = gpu.DeviceBuffer.from_host(P.reshape(-1))
    gpu.g1_scale(inplace, n, sb, out=inplace)
    assert (np.asarray(inplace.to_host()).reshape(n, 96)[9] == np.asarray(oracle.g1_scalar_mul(sb, P[9]))).all()
