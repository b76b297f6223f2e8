"""CPU ORACLE loader (test infrastructure, NOT the product).

ctypes binding of oracle/libtk_oracle.so (built by oracle/Makefile from tk_oracle.c).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product library
(libtkmk_hip.so) never links or calls it.  Buffers are numpy uint8 arrays holding plain little-endian
field elements (Fr 32 B, Fq 48 B, G1 affine 96 B) — see tk_oracle.h for the parity status.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.run(["make", "-s", "-C", _HERE, "libtk_oracle.so"], check=True)


def usable_cpus(limit=None):
    """CPUs this process may really use: the affinity mask and the cgroup CPU quota, whichever is smaller (at most `limit`)"""
    n = limit if limit else (os.cpu_count() or 1)
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                n = min(n, max(1, int(quota) // int(period)))
        except (OSError, ValueError):
            pass
    try:       # cgroup v1
        quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if quota > 0 and period > 0:
            n = min(n, max(1, quota // period))
    except (OSError, ValueError):
        pass
    return max(1, n)


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libtk_oracle.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
        _LIB.tko_ntt.restype = ctypes.c_int
        _LIB.tko_bintt.restype = ctypes.c_int
        _LIB.tko_dft_naive.restype = ctypes.c_int
        _LIB.tko_get_root_of_unity.restype = ctypes.c_int
        _LIB.tko_g1_on_curve.restype = ctypes.c_int
        _LIB.tko_bn254_g1_on_curve.restype = ctypes.c_int
        for f in ("tko_bn254_ntt", "tko_bn254_bintt", "tko_bn254_dft_naive", "tko_bn254_get_root_of_unity"):
            getattr(_LIB, f).restype = ctypes.c_int
        _LIB.tko_num_threads.restype = ctypes.c_int
        _LIB.tko_set_num_threads.restype = ctypes.c_int
        if "OMP_NUM_THREADS" not in os.environ:
            # OpenMP's default is every hardware thread of the host (256 on a GPU box); a process that only has a share of them — an
            # affinity mask or a cgroup CPU quota, 16 for one GPU — is throttled for a scheduler period every time 256 threads wake up:
            # measured 0.1-0.2 s per oracle call, whatever its size (tools/scratch: a 2-point NTT took as long as a 2^12-point one)
            _LIB.tko_set_num_threads(usable_cpus(_LIB.tko_num_threads()))
        _LIB.tko_poly_mul_monomial.restype = ctypes.c_int
        _LIB.tko_poly_div_by_vanishing_opt.restype = ctypes.c_int
    return _LIB


def _p(a):
    if a is None:
        return None
    assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.c_void_p)


def _sz(n):
    return ctypes.c_size_t(n)


def _u64(n):
    return ctypes.c_uint64(n)


def _binop(name, width):
    def f(a, b):
        out = np.empty_like(a)
        getattr(lib(), name)(_p(a), _p(b), _p(out), _sz(a.size // width))
        return out
    return f


fr_add = _binop("tko_fr_add", 32)
fr_sub = _binop("tko_fr_sub", 32)
fr_mul = _binop("tko_fr_mul", 32)
fq_add = _binop("tko_fq_add", 48)
fq_sub = _binop("tko_fq_sub", 48)
fq_mul = _binop("tko_fq_mul", 48)


def fr_inv(a):
    out = np.empty_like(a)
    lib().tko_fr_inv(_p(a), _p(out), _sz(a.size // 32))
    return out


def fq_inv(a):
    out = np.empty_like(a)
    lib().tko_fq_inv(_p(a), _p(out), _sz(a.size // 48))
    return out


def fr_scalar_mul(s, a):
    out = np.empty_like(a)
    lib().tko_fr_scalar_mul(_p(s), _p(a), _p(out), _sz(a.size // 32))
    return out


def fr_scalar_add(s, a):
    out = np.empty_like(a)
    lib().tko_fr_scalar_add(_p(s), _p(a), _p(out), _sz(a.size // 32))
    return out


def fr_scalar_sub(s, a):
    out = np.empty_like(a)
    lib().tko_fr_scalar_sub(_p(s), _p(a), _p(out), _sz(a.size // 32))
    return out


def fr_pow(a, e):
    out = np.empty(32, np.uint8)
    lib().tko_fr_pow_u64(_p(a), _u64(e), _p(out))
    return out


def fr_transpose(a, rows, cols):
    out = np.empty_like(a)
    lib().tko_fr_transpose(_p(a), _sz(rows), _sz(cols), _p(out))
    return out


def fr_suffix_product(a):
    out = np.empty_like(a)
    lib().tko_fr_suffix_product(_p(a), _sz(a.size // 32), _p(out))
    return out


def fr_random(seed, n, first=0):
    out = np.empty(32 * n, np.uint8)
    lib().tko_fr_random(_u64(seed), _sz(first), _sz(n), _p(out))
    return out


def root_of_unity(max_size):
    out = np.empty(32, np.uint8)
    if lib().tko_get_root_of_unity(_u64(max_size), _p(out)) != 0:
        raise ValueError("no root of unity of that order")
    return out


def ntt(a, n, batch=1, columns_batch=False, inverse=False, coset_gen=None):
    out = np.empty_like(a)
    rc = lib().tko_ntt(_p(a), _sz(n), _sz(batch), int(columns_batch), int(inverse), _p(coset_gen), _p(out))
    if rc != 0:
        raise ValueError("tko_ntt failed: %d" % rc)
    return out


def bintt(a, x_size, y_size, inverse=False, coset_x=None, coset_y=None):
    out = np.empty_like(a)
    rc = lib().tko_bintt(_p(a), _sz(x_size), _sz(y_size), int(inverse), _p(coset_x), _p(coset_y), _p(out))
    if rc != 0:
        raise ValueError("tko_bintt failed: %d" % rc)
    return out


def dft_naive(a, n):
    out = np.empty_like(a)
    rc = lib().tko_dft_naive(_p(a), _sz(n), _p(out))
    if rc != 0:
        raise ValueError("tko_dft_naive failed: %d" % rc)
    return out


def g1_generator():
    out = np.empty(96, np.uint8)
    lib().tko_g1_generator(_p(out))
    return out


def g1_on_curve(p):
    return bool(lib().tko_g1_on_curve(_p(p)))


def g1_add(p, q):
    out = np.empty(96, np.uint8)
    lib().tko_g1_add(_p(p), _p(q), _p(out))
    return out


def g1_neg(p):
    out = np.empty(96, np.uint8)
    lib().tko_g1_neg(_p(p), _p(out))
    return out


def g1_scalar_mul(s, p):
    out = np.empty(96, np.uint8)
    lib().tko_g1_scalar_mul(_p(s), _p(p), _p(out))
    return out


def g1_batch_scalar_mul(s, p):
    n = s.size // 32
    out = np.empty(96 * n, np.uint8)
    lib().tko_g1_batch_scalar_mul(_p(s), _p(p), _sz(n), _p(out))
    return out


def g1_random_bases(seed, n, first=0):
    out = np.empty(96 * n, np.uint8)
    lib().tko_g1_random_bases(_u64(seed), _sz(first), _sz(n), _p(out))
    return out


def g1_msm_naive(s, p):
    out = np.empty(96, np.uint8)
    lib().tko_g1_msm_naive(_p(s), _p(p), _sz(s.size // 32), _p(out))
    return out


def g1_msm(s, p, threads=0):
    out = np.empty(96, np.uint8)
    lib().tko_g1_msm(_p(s), _p(p), _sz(s.size // 32), int(threads), _p(out))
    return out


def g1_proj_to_affine(p144):
    out = np.empty(96, np.uint8)
    lib().tko_g1_proj_to_affine(_p(p144), _p(out))
    return out


class _G2:
    """G2 of BLS12-381 (twist over Fp2): the curve-generic oracle code instantiated over Fp2.  Affine 192 B (x then y, each real then
    imaginary part, 48-byte little-endian; all zero = infinity), projective 288 B.  Same method names as the module-level G1 functions."""
    AFF_BYTES = 192

    def generator(self):
        out = np.empty(192, np.uint8)
        lib().tko_g2_generator(_p(out))
        return out

    def on_curve(self, p):
        return bool(lib().tko_g2_on_curve(_p(p)))

    def add(self, p, q):
        out = np.empty(192, np.uint8)
        lib().tko_g2_add(_p(p), _p(q), _p(out))
        return out

    def neg(self, p):
        out = np.empty(192, np.uint8)
        lib().tko_g2_neg(_p(p), _p(out))
        return out

    def scalar_mul(self, s, p):
        out = np.empty(192, np.uint8)
        lib().tko_g2_scalar_mul(_p(s), _p(p), _p(out))
        return out

    def random_bases(self, seed, n, first=0):
        out = np.empty(192 * n, np.uint8)
        lib().tko_g2_random_bases(_u64(seed), _sz(first), _sz(n), _p(out))
        return out

    def msm_naive(self, s, p):
        out = np.empty(192, np.uint8)
        lib().tko_g2_msm_naive(_p(s), _p(p), _sz(s.size // 32), _p(out))
        return out

    def msm(self, s, p, threads=0):
        out = np.empty(192, np.uint8)
        lib().tko_g2_msm(_p(s), _p(p), _sz(s.size // 32), int(threads), _p(out))
        return out


g2 = _G2()


class _Bn254:
    """BN254 (alt_bn128) instantiation of the oracle's field / G1 code: Fr and Fq 32 B, affine 64 B, projective 96 B.
    Same method names as the module-level BLS12-381 functions."""
    R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    P_MOD = 21888242871839275222246405745257275088696311157297823662689037894645226208583
    FQ_BYTES, AFF_BYTES = 32, 64

    def __init__(self):
        for f in ("fr", "fq"):
            for op in ("add", "sub", "mul"):
                setattr(self, "%s_%s" % (f, op), _binop("tko_bn254_%s_%s" % (f, op), 32))

    @staticmethod
    def _inv(name, a):
        out = np.empty_like(a)
        getattr(lib(), name)(_p(a), _p(out), _sz(a.size // 32))
        return out

    def fr_inv(self, a):
        return self._inv("tko_bn254_fr_inv", a)

    def fq_inv(self, a):
        return self._inv("tko_bn254_fq_inv", a)

    def fr_random(self, seed, n, first=0):
        out = np.empty(32 * n, np.uint8)
        lib().tko_bn254_fr_random(_u64(seed), _sz(first), _sz(n), _p(out))
        return out

    def root_of_unity(self, max_size):
        out = np.empty(32, np.uint8)
        if lib().tko_bn254_get_root_of_unity(_u64(max_size), _p(out)) != 0:
            raise ValueError("no root of unity of that order")
        return out

    def ntt(self, a, n, batch=1, columns_batch=False, inverse=False, coset_gen=None):
        out = np.empty_like(a)
        if lib().tko_bn254_ntt(_p(a), _sz(n), _sz(batch), int(columns_batch), int(inverse), _p(coset_gen), _p(out)) != 0:
            raise ValueError("tko_bn254_ntt failed")
        return out

    def bintt(self, a, x_size, y_size, inverse=False, coset_x=None, coset_y=None):
        out = np.empty_like(a)
        if lib().tko_bn254_bintt(_p(a), _sz(x_size), _sz(y_size), int(inverse), _p(coset_x), _p(coset_y), _p(out)) != 0:
            raise ValueError("tko_bn254_bintt failed")
        return out

    def dft_naive(self, a, n):
        out = np.empty_like(a)
        if lib().tko_bn254_dft_naive(_p(a), _sz(n), _p(out)) != 0:
            raise ValueError("tko_bn254_dft_naive failed")
        return out

    def g1_generator(self):
        out = np.empty(64, np.uint8)
        lib().tko_bn254_g1_generator(_p(out))
        return out

    def g1_on_curve(self, p):
        return bool(lib().tko_bn254_g1_on_curve(_p(p)))

    def g1_add(self, p, q):
        out = np.empty(64, np.uint8)
        lib().tko_bn254_g1_add(_p(p), _p(q), _p(out))
        return out

    def g1_neg(self, p):
        out = np.empty(64, np.uint8)
        lib().tko_bn254_g1_neg(_p(p), _p(out))
        return out

    def g1_scalar_mul(self, s, p):
        out = np.empty(64, np.uint8)
        lib().tko_bn254_g1_scalar_mul(_p(s), _p(p), _p(out))
        return out

    def g1_batch_scalar_mul(self, s, p):
        n = s.size // 32
        out = np.empty(64 * n, np.uint8)
        lib().tko_bn254_g1_batch_scalar_mul(_p(s), _p(p), _sz(n), _p(out))
        return out

    def g1_random_bases(self, seed, n, first=0):
        out = np.empty(64 * n, np.uint8)
        lib().tko_bn254_g1_random_bases(_u64(seed), _sz(first), _sz(n), _p(out))
        return out

    def g1_msm_naive(self, s, p):
        out = np.empty(64, np.uint8)
        lib().tko_bn254_g1_msm_naive(_p(s), _p(p), _sz(s.size // 32), _p(out))
        return out

    def g1_msm(self, s, p, threads=0):
        out = np.empty(64, np.uint8)
        lib().tko_bn254_g1_msm(_p(s), _p(p), _sz(s.size // 32), int(threads), _p(out))
        return out

    def g1_proj_to_affine(self, p96):
        out = np.empty(64, np.uint8)
        lib().tko_bn254_g1_proj_to_affine(_p(p96), _p(out))
        return out


bn254 = _Bn254()


def poly_find_degree(c, xs, ys):
    xd, yd = ctypes.c_int64(), ctypes.c_int64()
    lib().tko_poly_find_degree(_p(c), _sz(xs), _sz(ys), ctypes.byref(xd), ctypes.byref(yd))
    return xd.value, yd.value


def poly_resized_dims(tx, ty):
    nx, ny = ctypes.c_size_t(), ctypes.c_size_t()
    lib().tko_poly_resized_dims(_sz(tx), _sz(ty), ctypes.byref(nx), ctypes.byref(ny))
    return nx.value, ny.value


def poly_resize(c, xs, ys, tx, ty):
    nx, ny = poly_resized_dims(tx, ty)
    out = np.empty(32 * nx * ny, np.uint8)
    lib().tko_poly_resize(_p(c), _sz(xs), _sz(ys), _sz(nx), _sz(ny), _p(out))
    return out, nx, ny


def poly_mul_monomial(c, xs, ys, x_degree, y_degree, ex, ey):
    nx, ny = poly_resized_dims(x_degree + 1 + ex, y_degree + 1 + ey)
    out = np.empty(32 * nx * ny, np.uint8)
    if lib().tko_poly_mul_monomial(_p(c), _sz(xs), _sz(ys), _sz(ex), _sz(ey), _sz(nx), _sz(ny), _p(out)) != 0:
        raise ValueError("mul_monomial: source does not fit (the reference would panic)")
    return out, nx, ny


def poly_scale_coeffs(c, xs, ys, fx=None, fy=None):
    out = np.empty_like(c)
    lib().tko_poly_scale_coeffs(_p(c), _sz(xs), _sz(ys), _p(fx), _p(fy), _p(out))
    return out


def poly_eval(c, xs, ys, x, y):
    out = np.empty(32, np.uint8)
    lib().tko_poly_eval(_p(c), _sz(xs), _sz(ys), _p(x), _p(y), _p(out))
    return out


def poly_eval_x(c, xs, ys, x):
    out = np.empty(32 * ys, np.uint8)
    lib().tko_poly_eval_x(_p(c), _sz(xs), _sz(ys), _p(x), _p(out))
    return out


def poly_eval_y(c, xs, ys, y):
    out = np.empty(32 * xs, np.uint8)
    lib().tko_poly_eval_y(_p(c), _sz(xs), _sz(ys), _p(y), _p(out))
    return out


def poly_div_by_vanishing_opt(p, xs, ys, c, d):
    qx = np.empty(32 * xs * ys, np.uint8)
    qy = np.empty(32 * c * ys, np.uint8)
    if lib().tko_poly_div_by_vanishing_opt(_p(p), _sz(xs), _sz(ys), _sz(c), _sz(d), _p(qx), _p(qy)) != 0:
        raise ValueError("div_by_vanishing_opt: bad shape")
    return qx, qy


def poly_div_by_ruffini(p, xs, ys, x, y):
    qx = np.empty(32 * xs * ys, np.uint8)
    qy = np.empty(32 * ys, np.uint8)
    r = np.empty(32, np.uint8)
    lib().tko_poly_div_by_ruffini(_p(p), _sz(xs), _sz(ys), _p(x), _p(y), _p(qx), _p(qy), _p(r))
    return qx, qy, r


def r1cs_eval_rows(row_ptr, wire, coeff, variables, out_len):
    out = np.empty(32 * out_len, np.uint8)
    rp = np.ascontiguousarray(row_ptr, np.uint32)
    wi = np.ascontiguousarray(wire, np.uint32)
    lib().tko_r1cs_eval_rows(rp.ctypes.data_as(ctypes.c_void_p), wi.ctypes.data_as(ctypes.c_void_p), _p(coeff), _sz(len(rp) - 1),
                             _p(variables), _p(out), _sz(out_len))
    return out


def num_threads():
    return lib().tko_num_threads()


# ---- helpers shared by tests: python int <-> plain LE bytes ----
R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
P_MOD = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB


def to_bytes(vals, width):
    return np.frombuffer(b"".join(int(v).to_bytes(width, "little") for v in vals), np.uint8).copy()


def to_ints(buf, width):
    raw = buf.tobytes()
    return [int.from_bytes(raw[i:i + width], "little") for i in range(0, len(raw), width)]
