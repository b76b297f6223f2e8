"""tkmk — Python (ctypes) binding of libtkmk_hip.so, the MI355X backend's C ABI (include/tkmk.h).

This is plumbing for tests and bench.py: numpy arrays of plain little-endian field elements in, the HIP
library does the work.  It mirrors the reference's thin wrappers over ICICLE
(packages/backend/libs/src/vector_operations/mod.rs, bivariate_polynomial/mod.rs:_biNTT,
iotools/mod.rs:encode_poly) — same argument meaning, errors surface as TkmkError.

There is NO CPU fallback: if the shared library is missing or no gfx950 device is visible, calls raise.
"""
import ctypes
import os

import numpy as np

# the library's own default (csrc/runtime.hip: streams share 4 hardware queues otherwise), made here too because a Python process may
# initialise HIP through another module before libtkmk_hip.so is loaded
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("TKMK_HIP_LIBRARY") or os.path.join(_PKG, "libtkmk_hip.so")   # the override is for kernel experiments (tools/)
_lib = None


class TkmkError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = lib().tkmk_error_string(code).decode() if _lib is not None else str(code)
        super().__init__("%s failed: %s (tkmk_error %d)" % (where, msg, code))


class Fr(ctypes.Structure):
    _fields_ = [("limbs", ctypes.c_uint32 * 8)]


class MSMConfig(ctypes.Structure):
    _fields_ = [
        ("stream_handle", ctypes.c_void_p),
        ("precompute_factor", ctypes.c_int),
        ("c", ctypes.c_int),
        ("bitsize", ctypes.c_int),
        ("batch_size", ctypes.c_int),
        ("are_points_shared_in_batch", ctypes.c_bool),
        ("are_scalars_on_device", ctypes.c_bool),
        ("are_scalars_montgomery_form", ctypes.c_bool),
        ("are_points_on_device", ctypes.c_bool),
        ("are_points_montgomery_form", ctypes.c_bool),
        ("are_results_on_device", ctypes.c_bool),
        ("is_async", ctypes.c_bool),
        ("ext", ctypes.c_void_p),
    ]


class NTTConfig(ctypes.Structure):
    _fields_ = [
        ("stream_handle", ctypes.c_void_p),
        ("coset_gen", Fr),
        ("batch_size", ctypes.c_int),
        ("columns_batch", ctypes.c_bool),
        ("ordering", ctypes.c_int),
        ("are_inputs_on_device", ctypes.c_bool),
        ("are_outputs_on_device", ctypes.c_bool),
        ("is_async", ctypes.c_bool),
        ("ext", ctypes.c_void_p),
    ]


class NTTInitDomainConfig(ctypes.Structure):
    _fields_ = [("stream_handle", ctypes.c_void_p), ("is_async", ctypes.c_bool), ("ext", ctypes.c_void_p)]


class VecOpsConfig(ctypes.Structure):
    _fields_ = [
        ("stream_handle", ctypes.c_void_p),
        ("is_a_on_device", ctypes.c_bool),
        ("is_b_on_device", ctypes.c_bool),
        ("is_result_on_device", ctypes.c_bool),
        ("is_async", ctypes.c_bool),
        ("batch_size", ctypes.c_int),
        ("columns_batch", ctypes.c_bool),
        ("ext", ctypes.c_void_p),
    ]


# every symbol include/tkmk.h declares (tests/test_abi.py checks the library exports all of them)
SYMBOLS = [
    "tkmk_device_count", "tkmk_set_device", "tkmk_get_available_memory", "tkmk_malloc", "tkmk_malloc_async", "tkmk_free",
    "tkmk_free_async", "tkmk_memcpy_h2d", "tkmk_memcpy_d2h", "tkmk_memcpy_d2d", "tkmk_memcpy_h2d_async",
    "tkmk_memcpy_d2h_async", "tkmk_memcpy_2d_d2d", "tkmk_memset", "tkmk_stream_create", "tkmk_stream_synchronize", "tkmk_stream_destroy", "tkmk_stream_set_background",
    "tkmk_device_synchronize", "tkmk_release_scratch", "tkmk_error_string", "tkmk_is_hip_build", "tkmk_keccak256", "tkmk_r1cs_index", "tkmk_msm_default_config", "bls12_381_msm", "bls12_381_g2_msm", "tkmk_g1_ntt", "tkmk_g1_ntt_axes", "tkmk_g1_prefix_sums", "tkmk_g1_scale", "bls12_381_msm_precompute_bases", "bn254_msm_precompute_bases", "tkmk_msm_multi", "bn254_msm", "tkmk_bn254_msm_multi", "bn254_get_root_of_unity", "bn254_ntt_init_domain", "bn254_ntt_release_domain",
    "bn254_ntt", "tkmk_bn254_bintt", "tkmk_bn254_fr_random_device",
    "tkmk_bn254_g1_batch_scalar_mul_device",
    "tkmk_ntt_default_config", "bls12_381_get_root_of_unity", "bls12_381_ntt_init_domain", "bls12_381_ntt_release_domain",
    "bls12_381_ntt", "tkmk_bintt", "tkmk_vecops_default_config", "bls12_381_vector_add", "bls12_381_vector_sub",
    "bls12_381_vector_mul", "bls12_381_vector_div", "bls12_381_vector_inv", "bls12_381_scalar_add_vec",
    "bls12_381_scalar_sub_vec", "bls12_381_scalar_mul_vec", "bls12_381_vector_sum", "bls12_381_vector_product",
    "bls12_381_matrix_transpose", "tkmk_vec_suffix_product", "tkmk_fr_random_device", "tkmk_gather_rows_device", "tkmk_g1_batch_scalar_mul_device", "tkmk_profile_enable",
    "tkmk_profile_reset", "tkmk_profile_get", "tkmk_diag_bench", "tkmk_diag_field_mul", "tkmk_poly_find_degree",
    "tkmk_poly_place", "tkmk_poly_scale_coeffs", "tkmk_poly_mul_x_minus_one_evals", "tkmk_poly_mul_ones_x", "tkmk_poly_expr_eval", "tkmk_poly_expr_eval_views", "tkmk_poly_expr_eval_views_slab", "tkmk_poly_eval_x", "tkmk_poly_eval_y", "tkmk_poly_eval",
    "tkmk_poly_div_by_vanishing_opt", "tkmk_poly_div_by_ruffini", "tkmk_r1cs_eval_rows",
    "tkmk_msm_multi_ex", "bls12_381_msm_convert_bases", "tkmk_r1cs_library_create", "tkmk_r1cs_library_destroy", "tkmk_r1cs_library_eval",
    "tkmk_witness_route", "tkmk_fr_scatter_table", "tkmk_msm_set_pipeline_streams", "tkmk_msm_get_pipeline_streams", "tkmk_host_malloc", "tkmk_host_free", "tkmk_stats_reset", "tkmk_stats_get",
    "bls12_381_ntt_domain_size", "bn254_ntt_domain_size", "tkmk_poly_lincomb", "tkmk_bintt_padded", "tkmk_diag_device_switch", "tkmk_diag_gather_probe",
    "bls12_381_generate_random_affine_points", "bls12_381_generate_scalars", "bls12_381_polynomial_add", "bls12_381_polynomial_clone", "bls12_381_polynomial_coeffs_device_ptr", "bls12_381_polynomial_copy_coeffs", "bls12_381_polynomial_create_from_coefficients", "bls12_381_polynomial_create_from_rou_evaluations", "bls12_381_polynomial_degree", "bls12_381_polynomial_delete", "bls12_381_polynomial_divide", "bls12_381_polynomial_evaluate", "bls12_381_polynomial_get_coeff", "bls12_381_polynomial_multiply", "bls12_381_polynomial_multiply_by_scalar", "bls12_381_polynomial_nof_coeffs", "bls12_381_polynomial_slice", "bls12_381_polynomial_subtract", "bls12_381_vector_accumulate",
]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libtkmk_hip.so is not built (run __graft_entry__.build()); there is no CPU fallback")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.tkmk_error_string.restype = ctypes.c_char_p
        _lib.tkmk_msm_default_config.restype = MSMConfig
        _lib.tkmk_ntt_default_config.restype = NTTConfig
        _lib.tkmk_vecops_default_config.restype = VecOpsConfig
    return _lib


def _check(code, where):
    if code != 0:
        raise TkmkError(code, where)


def _p(a):
    if a is None:
        return None
    if isinstance(a, DeviceBuffer):
        return ctypes.c_void_p(a.ptr)
    assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.c_void_p)


def _on_dev(a):
    return isinstance(a, DeviceBuffer)


def _fr_struct(buf32):
    f = Fr()
    ctypes.memmove(ctypes.byref(f), buf32.ctypes.data, 32)
    return f


def _out_like(a, nbytes, out):
    if out is not None:
        return out
    if _on_dev(a):
        return DeviceBuffer(nbytes)
    return np.empty(nbytes, np.uint8)


class DeviceBuffer:
    """RAII device allocation — the work-alike of icicle_runtime::memory::DeviceVec."""

    def __init__(self, nbytes):
        p = ctypes.c_void_p()
        _check(lib().tkmk_malloc(ctypes.byref(p), ctypes.c_size_t(nbytes)), "tkmk_malloc")
        self.ptr = p.value
        self.nbytes = nbytes

    @classmethod
    def from_host(cls, arr):
        d = cls(arr.size)
        _check(lib().tkmk_memcpy_h2d(ctypes.c_void_p(d.ptr), _p(arr), ctypes.c_size_t(arr.size)), "tkmk_memcpy_h2d")
        return d

    def to_host(self, nbytes=None, offset=0):
        n = self.nbytes - offset if nbytes is None else nbytes
        out = np.empty(n, np.uint8)
        _check(lib().tkmk_memcpy_d2h(_p(out), ctypes.c_void_p(self.ptr + offset), ctypes.c_size_t(n)), "tkmk_memcpy_d2h")
        return out

    def free(self):
        if self.ptr:
            lib().tkmk_free(ctypes.c_void_p(self.ptr))
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def device_count():
    n = ctypes.c_int()
    _check(lib().tkmk_device_count(ctypes.byref(n)), "tkmk_device_count")
    return n.value


def set_device(i):
    _check(lib().tkmk_set_device(int(i)), "tkmk_set_device")


def synchronize():
    _check(lib().tkmk_device_synchronize(), "tkmk_device_synchronize")


def available_memory():
    t, f = ctypes.c_size_t(), ctypes.c_size_t()
    _check(lib().tkmk_get_available_memory(ctypes.byref(t), ctypes.byref(f)), "tkmk_get_available_memory")
    return t.value, f.value


# ---- NTT (reference: libs/src/bivariate_polynomial/mod.rs:33-55, 1422-1478) ----
# curve -> (get_root_of_unity, init_domain, release_domain, ntt, bintt) symbols
_NTT_SYMS = {
    "bls12_381": ("bls12_381_get_root_of_unity", "bls12_381_ntt_init_domain", "bls12_381_ntt_release_domain", "bls12_381_ntt", "tkmk_bintt"),
    "bn254": ("bn254_get_root_of_unity", "bn254_ntt_init_domain", "bn254_ntt_release_domain", "bn254_ntt", "tkmk_bn254_bintt"),
}


def get_root_of_unity(max_size, curve="bls12_381"):
    sym = _NTT_SYMS[curve][0]
    out = np.empty(32, np.uint8)
    _check(getattr(lib(), sym)(ctypes.c_uint64(max_size), _p(out)), sym)
    return out


_domain_size = {}


def init_ntt_domain_for_size(size, curve="bls12_381"):
    """grow-only global domain (one per scalar field), like init_ntt_domain_for_size (bivariate_polynomial/mod.rs:33-55)"""
    if size <= 0 or size & (size - 1):
        raise ValueError("NTT domain size must be a non-zero power of two")
    have = ntt_domain_size(curve)              # the library's own record: another host side in this process may have grown it
    if have >= size:
        return
    if have:
        release_ntt_domain(curve)
    root = get_root_of_unity(size, curve)
    cfg = NTTInitDomainConfig(None, False, None)
    sym = _NTT_SYMS[curve][1]
    _check(getattr(lib(), sym)(_p(root), ctypes.byref(cfg)), sym)
    _domain_size[curve] = size


def ntt_domain_size(curve="bls12_381"):
    v = ctypes.c_uint64()
    sym = "bls12_381_ntt_domain_size" if curve == "bls12_381" else "bn254_ntt_domain_size"
    _check(getattr(lib(), sym)(ctypes.byref(v)), sym)
    return v.value


def bintt_padded(a, in_x, in_y, x_size, y_size, coset_x=None, coset_y=None, out=None):
    """forward _biNTT of the zero-padded extension of the compact in_x x in_y matrix `a` (DeviceBuffer) to x_size x y_size"""
    STATS["ntt_elements"] += int(x_size) * int(y_size)
    out = DeviceBuffer(32 * x_size * y_size) if out is None else out
    _check(lib().tkmk_bintt_padded(_p(a), ctypes.c_size_t(in_x), ctypes.c_size_t(in_y), ctypes.c_size_t(x_size), ctypes.c_size_t(y_size),
                                   _p(coset_x), _p(coset_y), None, _p(out)), "tkmk_bintt_padded")
    return out


def release_ntt_domain(curve="bls12_381"):
    sym = _NTT_SYMS[curve][2]
    _check(getattr(lib(), sym)(), sym)
    _domain_size.pop(curve, None)


# algorithmic work issued through this binding since the last reset: points committed by MSM calls and elements transformed by
# (bi)NTT calls (tools/prove_bench.py turns them into SURVEY.md §8d's algorithmic bytes: 128 B per point, 64 B per element)
STATS = {"msm_points": 0, "ntt_elements": 0}


def stats_reset():
    STATS["msm_points"] = STATS["ntt_elements"] = 0


def ntt(a, n, batch=1, columns_batch=False, inverse=False, coset_gen=None, out=None, stream=None, curve="bls12_381"):
    STATS["ntt_elements"] += int(n) * int(batch)
    cfg = lib().tkmk_ntt_default_config()
    cfg.batch_size = batch
    cfg.columns_batch = columns_batch
    if coset_gen is not None:
        cfg.coset_gen = _fr_struct(coset_gen)
    cfg.stream_handle = stream
    out = _out_like(a, 32 * n * batch, out)
    cfg.are_inputs_on_device = _on_dev(a)
    cfg.are_outputs_on_device = _on_dev(out)
    sym = _NTT_SYMS[curve][3]
    _check(getattr(lib(), sym)(_p(a), int(n), 1 if inverse else 0, ctypes.byref(cfg), _p(out)), sym)
    return out


def bintt(a, x_size, y_size, inverse=False, coset_x=None, coset_y=None, out=None, stream=None, curve="bls12_381"):
    STATS["ntt_elements"] += int(x_size) * int(y_size)
    out = _out_like(a, 32 * x_size * y_size, out)
    if _on_dev(a) != _on_dev(out):
        raise ValueError("tkmk_bintt takes both buffers on the same side")
    sym = _NTT_SYMS[curve][4]
    _check(getattr(lib(), sym)(_p(a), ctypes.c_size_t(x_size), ctypes.c_size_t(y_size), 1 if inverse else 0, _p(coset_x),
                               _p(coset_y), _on_dev(a), ctypes.c_void_p(stream), _p(out)), sym)
    return out


# ---- vector ops (reference: libs/src/vector_operations/mod.rs:30-139) ----
def _vec(name, a, b, n, out, batch=1, columns_batch=False, out_elems=None):
    cfg = lib().tkmk_vecops_default_config()
    cfg.is_a_on_device = _on_dev(a)
    cfg.is_b_on_device = _on_dev(b) if b is not None else False
    cfg.batch_size = batch
    cfg.columns_batch = columns_batch
    nbytes = 32 * (out_elems if out_elems is not None else n * batch)
    if out is None:
        out = DeviceBuffer(nbytes) if (_on_dev(a) or _on_dev(b)) else np.empty(nbytes, np.uint8)
    cfg.is_result_on_device = _on_dev(out)
    fn = getattr(lib(), name)
    if b is None:
        _check(fn(_p(a), ctypes.c_uint64(n), ctypes.byref(cfg), _p(out)), name)
    else:
        _check(fn(_p(a), _p(b), ctypes.c_uint64(n), ctypes.byref(cfg), _p(out)), name)
    return out


def vec_add(a, b, out=None):
    return _vec("bls12_381_vector_add", a, b, _len(a), out)


def vec_sub(a, b, out=None):
    return _vec("bls12_381_vector_sub", a, b, _len(a), out)


def vec_mul(a, b, out=None):
    return _vec("bls12_381_vector_mul", a, b, _len(a), out)


def vec_div(a, b, out=None):
    return _vec("bls12_381_vector_div", a, b, _len(a), out)


def vec_inv(a, out=None):
    return _vec("bls12_381_vector_inv", a, None, _len(a), out)


def scalar_add(s, v, out=None):
    return _vec("bls12_381_scalar_add_vec", s, v, _len(v), out)


def scalar_sub(s, v, out=None):
    return _vec("bls12_381_scalar_sub_vec", s, v, _len(v), out)


def scalar_mul(s, v, out=None):
    return _vec("bls12_381_scalar_mul_vec", s, v, _len(v), out)


def vec_sum(a, n=None, batch=1, columns_batch=False):
    n = _len(a) // batch if n is None else n
    return _vec("bls12_381_vector_sum", a, None, n, None, batch, columns_batch, out_elems=batch)


def vec_product(a, n=None, batch=1, columns_batch=False):
    n = _len(a) // batch if n is None else n
    return _vec("bls12_381_vector_product", a, None, n, None, batch, columns_batch, out_elems=batch)


def transpose(a, rows, cols, out=None):
    cfg = lib().tkmk_vecops_default_config()
    cfg.is_a_on_device = _on_dev(a)
    out = _out_like(a, 32 * rows * cols, out)
    cfg.is_result_on_device = _on_dev(out)
    _check(lib().bls12_381_matrix_transpose(_p(a), ctypes.c_uint32(rows), ctypes.c_uint32(cols), ctypes.byref(cfg), _p(out)),
           "bls12_381_matrix_transpose")
    return out


def _len(a):
    return (a.nbytes if _on_dev(a) else a.size) // 32


# ---- MSM (reference: libs/src/iotools/mod.rs:2093-2099, group_structures/mod.rs:108-143) ----
# curve -> (msm symbol, multi symbol, fr_random symbol, batch scalar-mul symbol, affine bytes)
_CURVES = {
    "bls12_381": ("bls12_381_msm", "tkmk_msm_multi", "tkmk_fr_random_device", "tkmk_g1_batch_scalar_mul_device", 96,
                  "bls12_381_msm_precompute_bases"),
    "bn254": ("bn254_msm", "tkmk_bn254_msm_multi", "tkmk_bn254_fr_random_device", "tkmk_bn254_g1_batch_scalar_mul_device", 64,
              "bn254_msm_precompute_bases"),
}


def msm_windows(n, c=0, bitsize=0, curve="bls12_381", explicit_c=False):
    """number of scalar windows the library uses for an n-point MSM (c = 0: its own choice), i.e. the largest useful
    precompute factor; mirrors choose_c / msm_resolve_c in csrc/msm_impl.inc"""
    bits = bitsize or (255 if curve == "bls12_381" else 254)
    if not c:
        best, cost = 2, None
        for cc in range(2, 17):
            v = (255 // cc + 1) * (n + 4.0 * (1 << (cc - 1)))
            if cost is None or v < cost:
                best, cost = cc, v
        c = best
    if explicit_c:                                   # msm_precompute_bases takes an explicit width as given (tables for table jobs)
        return bits // min(max(c, 2), 20) + 1
    c = min(max(c, 2), 20)
    if c > 16 and n < (1 << 18):
        c = 16
    return bits // c + 1


def msm_precompute_bases(bases, n, factor, c=0, bitsize=0, curve="bls12_381", points_montgomery=False):
    """-> DeviceBuffer with the n * min(factor, windows) expanded points (converted form) for msm(..., precompute_factor=factor)"""
    sym, aff = _CURVES[curve][5], _CURVES[curve][4]
    cfg = lib().tkmk_msm_default_config()
    cfg.precompute_factor = factor
    cfg.c, cfg.bitsize = c, bitsize
    cfg.are_points_on_device = _on_dev(bases)
    cfg.are_points_montgomery_form = points_montgomery
    cfg.are_results_on_device = True
    f = min(factor, msm_windows(n, c, bitsize, curve, explicit_c=bool(c)))
    out = DeviceBuffer(aff * n * f)
    _check(getattr(lib(), sym)(_p(bases), int(n), ctypes.byref(cfg), _p(out)), sym)
    return out


def msm(scalars, bases, msm_size=None, batch=1, shared_points=True, c=0, bitsize=0, stream=None, curve="bls12_381",
        precompute_factor=1, scalars_montgomery=False, points_montgomery=False):
    """returns `batch` projective results (144 B each; 96 B for bn254) on the host"""
    sym, aff = _CURVES[curve][0], _CURVES[curve][4]
    cfg = lib().tkmk_msm_default_config()
    n = _len(scalars) // batch if msm_size is None else msm_size
    STATS["msm_points"] += int(n) * int(batch)
    cfg.batch_size = batch
    cfg.are_points_shared_in_batch = shared_points
    cfg.are_scalars_on_device = _on_dev(scalars)
    cfg.are_points_on_device = _on_dev(bases)
    cfg.c = c
    cfg.bitsize = bitsize
    cfg.stream_handle = stream
    cfg.precompute_factor = precompute_factor
    cfg.are_scalars_montgomery_form = scalars_montgomery
    cfg.are_points_montgomery_form = points_montgomery
    out = np.empty(aff // 2 * 3 * batch, np.uint8)
    _check(getattr(lib(), sym)(_p(scalars), _p(bases), int(n), ctypes.byref(cfg), _p(out)), sym)
    return out


G1_NTT_AXIS_X, G1_NTT_AXIS_Y = 1, 2        # TKMK_G1_NTT_AXIS_* of include/tkmk.h


def g1_ntt(points, x_size, y_size, inverse=False, bases_form=0, in_stride=None, out=None, axes=None):
    """bivariate NTT over G1 points (tkmk_g1_ntt): points = DeviceBuffer of affine records, rows of in_stride (default y_size);
    returns a DeviceBuffer of x_size * y_size plain affine records; the inverse is not scaled by 1 / (x_size * y_size).
    axes (tkmk_g1_ntt_axes): G1_NTT_AXIS_Y alone transforms the rows, G1_NTT_AXIS_X alone the columns"""
    out = DeviceBuffer(96 * x_size * y_size) if out is None else out
    stride = ctypes.c_uint32(in_stride if in_stride is not None else y_size)
    if axes is None:
        _check(lib().tkmk_g1_ntt(_p(points), int(bases_form), stride, ctypes.c_uint32(x_size), ctypes.c_uint32(y_size), int(1 if inverse else 0), _p(out), None), "tkmk_g1_ntt")
    else:
        _check(lib().tkmk_g1_ntt_axes(_p(points), int(bases_form), stride, ctypes.c_uint32(x_size), ctypes.c_uint32(y_size), int(1 if inverse else 0), int(axes), _p(out),
                                      None), "tkmk_g1_ntt_axes")
    return out


def g1_prefix_sums(points, rows, cols, transposed=False, bases_form=0, out=None):
    """out[j] = sum_{j' <= j} points[idx(j')] (tkmk_g1_prefix_sums); idx(j) = (j % rows) * cols + j // rows when transposed"""
    out = DeviceBuffer(96 * rows * cols) if out is None else out
    _check(lib().tkmk_g1_prefix_sums(_p(points), int(bases_form), ctypes.c_uint32(rows), ctypes.c_uint32(cols), int(bool(transposed)), _p(out), None),
           "tkmk_g1_prefix_sums")
    return out


def g1_scale(points, n, scalar, out=None):
    """out[i] = [scalar] points[i] (tkmk_g1_scale): plain affine device records, one 32-byte little-endian scalar"""
    out = DeviceBuffer(96 * n) if out is None else out
    sc = np.ascontiguousarray(np.frombuffer(bytes(scalar), np.uint8))
    _check(lib().tkmk_g1_scale(_p(points), ctypes.c_uint64(n), _p(sc), _p(out), None), "tkmk_g1_scale")
    return out


def msm_g2(scalars, bases, msm_size=None, batch=1, shared_points=True, c=0, bitsize=0, stream=None, scalars_montgomery=False,
           points_montgomery=False):
    """G2 MSM (bls12_381_g2_msm): bases are 192-byte affine records (x.c0, x.c1, y.c0, y.c1; all zeros = infinity); returns `batch`
    canonical projective results (288 B each) on the host"""
    cfg = lib().tkmk_msm_default_config()
    n = _len(scalars) // batch if msm_size is None else msm_size
    cfg.batch_size = batch
    cfg.are_points_shared_in_batch = shared_points
    cfg.are_scalars_on_device = _on_dev(scalars)
    cfg.are_points_on_device = _on_dev(bases)
    cfg.c = c
    cfg.bitsize = bitsize
    cfg.stream_handle = stream
    cfg.are_scalars_montgomery_form = scalars_montgomery
    cfg.are_points_montgomery_form = points_montgomery
    out = np.empty(288 * batch, np.uint8)
    _check(lib().bls12_381_g2_msm(_p(scalars), _p(bases), int(n), ctypes.byref(cfg), _p(out)), "bls12_381_g2_msm")
    return out


class MsmJob(ctypes.Structure):
    _fields_ = [("scalars", ctypes.c_void_p), ("bases", ctypes.c_void_p), ("msm_size", ctypes.c_int)]


def msm_multi(jobs, c=0, bitsize=0, stream=None, curve="bls12_381", precompute_factor=1):
    """jobs = [(scalars, bases[, msm_size])...], all host buffers or all DeviceBuffers; returns len(jobs) projective
    results on the host.  Independent MSMs are pipelined over internal streams (tkmk_msm_multi)."""
    sym, aff = _CURVES[curve][1], _CURVES[curve][4]
    cfg = lib().tkmk_msm_default_config()
    cfg.precompute_factor = precompute_factor
    if not jobs:
        return np.empty(0, np.uint8)
    on_dev_s = {_on_dev(j[0]) for j in jobs}
    on_dev_p = {_on_dev(j[1]) for j in jobs}
    if len(on_dev_s) != 1 or len(on_dev_p) != 1:
        raise ValueError("msm_multi: every job's scalars (and every job's bases) must live on the same side")
    cfg.are_scalars_on_device = on_dev_s.pop()
    cfg.are_points_on_device = on_dev_p.pop()
    cfg.c = c
    cfg.bitsize = bitsize
    cfg.stream_handle = stream
    arr = (MsmJob * len(jobs))()
    for k, j in enumerate(jobs):
        n = j[2] if len(j) > 2 else _len(j[0])
        STATS["msm_points"] += int(n)
        arr[k] = MsmJob(_p(j[0]).value, _p(j[1]).value, int(n))
    out = np.empty(aff // 2 * 3 * len(jobs), np.uint8)
    _check(getattr(lib(), sym)(arr, len(jobs), ctypes.byref(cfg), _p(out)), sym)
    return out


class MsmJobEx(ctypes.Structure):
    _fields_ = [("scalars", ctypes.c_void_p), ("bases", ctypes.c_void_p), ("msm_size", ctypes.c_int),
                ("scalar_cols", ctypes.c_uint32), ("scalar_stride", ctypes.c_uint32), ("base_cols", ctypes.c_uint32),
                ("base_stride", ctypes.c_uint32), ("base_index", ctypes.c_void_p), ("base_table_len", ctypes.c_uint64),
                ("table_c", ctypes.c_uint32), ("table_factor", ctypes.c_uint32)]


BASES_PLAIN, BASES_MONTGOMERY, BASES_CONVERTED = 0, 1, 2


def msm_convert_bases(bases, n=None, points_montgomery=False, out=None):
    """bases (host array or DeviceBuffer of 96-byte affine records) -> DeviceBuffer in the MSM's resident form (BASES_CONVERTED);
    out=bases converts a DeviceBuffer in place"""
    n = _len(bases) * 32 // 96 if n is None else n
    cfg = lib().tkmk_msm_default_config()
    cfg.are_points_on_device = _on_dev(bases)
    cfg.are_points_montgomery_form = points_montgomery
    cfg.are_results_on_device = True
    out = DeviceBuffer(96 * n) if out is None else out
    _check(lib().bls12_381_msm_convert_bases(_p(bases), ctypes.c_uint64(n), ctypes.byref(cfg), _p(out)), "bls12_381_msm_convert_bases")
    return out


def msm_job_ex_array(jobs):
    """the tkmk_msm_job_ex array of a job list (see msm_multi_ex); "scalars" / "bases" may carry a byte offset into their buffer as
    "scalar_offset" / "base_offset" (a rank's share of a view starts inside the table)"""
    arr = (MsmJobEx * max(len(jobs), 1))()
    for k, j in enumerate(jobs):
        sv, bv, ix = j.get("scalar_view") or (0, 0), j.get("base_view") or (0, 0), j.get("base_index")
        STATS["msm_points"] += int(j["n"])
        tc, tf = j.get("table", (0, 0))                  # (c, factor) of a precomputed table (msm_precompute_bases over the whole table)
        arr[k] = MsmJobEx(_p(j["scalars"]).value + int(j.get("scalar_offset", 0)), _p(j["bases"]).value + int(j.get("base_offset", 0)), int(j["n"]),
                          sv[0], sv[1], bv[0], bv[1], None if ix is None else _p(ix).value, int(j.get("table_len", 0)), int(tc), int(tf))
    return arr


def msm_multi_ex(jobs, bases_form=BASES_PLAIN, c=0, bitsize=0, stream=None):
    """jobs = [dict(scalars=DeviceBuffer, bases=DeviceBuffer, n=points, scalar_view=(cols, stride) | None,
    base_view=(cols, stride) | None, base_index=DeviceBuffer of u32 | None, table_len=records behind bases)];
    returns len(jobs) projective results on the host (tkmk_msm_multi_ex: MSMs over views of resident tables)"""
    cfg = lib().tkmk_msm_default_config()
    cfg.are_scalars_on_device = cfg.are_points_on_device = True
    cfg.c, cfg.bitsize, cfg.stream_handle = c, bitsize, stream
    if not jobs:
        return np.empty(0, np.uint8)
    arr = msm_job_ex_array(jobs)
    out = np.empty(144 * len(jobs), np.uint8)
    _check(lib().tkmk_msm_multi_ex(arr, len(jobs), ctypes.byref(cfg), int(bases_form), _p(out)), "tkmk_msm_multi_ex")
    return out


def projective_to_affine_bytes(proj, curve="bls12_381"):
    """ABI results are canonical (x_aff, y_aff, 1) / (0,1,0): dropping z is the affine conversion"""
    aff = _CURVES[curve][4]
    pb = aff // 2 * 3
    out = []
    for i in range(0, proj.size, pb):
        z = proj[i + aff:i + pb]
        out.append(np.zeros(aff, np.uint8) if not z.any() else proj[i:i + aff].copy())
    return np.concatenate(out)


# ---- deterministic device-side input generation ----
def fr_random_device(seed, n, first=0, out=None, curve="bls12_381"):
    sym = _CURVES[curve][2]
    out = DeviceBuffer(32 * n) if out is None else out
    _check(getattr(lib(), sym)(ctypes.c_uint64(seed), ctypes.c_uint64(first), ctypes.c_uint64(n), _p(out), None), sym)
    return out


def g1_batch_scalar_mul_device(scalars_dev, base_host, n, out=None, curve="bls12_381"):
    sym, aff = _CURVES[curve][3], _CURVES[curve][4]
    out = DeviceBuffer(aff * n) if out is None else out
    _check(getattr(lib(), sym)(_p(scalars_dev), _p(base_host), ctypes.c_uint64(n), _p(out), None), sym)
    return out


def diag_bench(kind, iters, blocks, reps=3):
    ms = ctypes.c_float()
    _check(lib().tkmk_diag_bench(int(kind), ctypes.c_uint32(iters), ctypes.c_uint32(blocks), int(reps), ctypes.byref(ms)),
           "tkmk_diag_bench")
    return ms.value


def profile_enable(on=True):
    _check(lib().tkmk_profile_enable(1 if on else 0), "tkmk_profile_enable")


def profile_reset():
    _check(lib().tkmk_profile_reset(), "tkmk_profile_reset")


def msm_set_pipeline_streams(n):
    """internal streams of the multi-MSM entries (1 = serialised: every kernel alone on the device; 0 = default)"""
    _check(lib().tkmk_msm_set_pipeline_streams(int(n)), "tkmk_msm_set_pipeline_streams")


def msm_get_pipeline_streams():
    return int(lib().tkmk_msm_get_pipeline_streams())


def profile_get(name):
    """(sum of ms, launches) of one kernel section since the last reset"""
    ms, cnt = ctypes.c_double(), ctypes.c_int()
    _check(lib().tkmk_profile_get(name.encode(), ctypes.byref(ms), ctypes.byref(cnt)), "tkmk_profile_get")
    return ms.value, cnt.value


def poly_lincomb(terms, out_xs, out_ys):
    """terms = [(coefficient (32-byte array), DeviceBuffer, x_size, y_size[, off_x, off_y])] -> DeviceBuffer holding
    sum_t c_t X^ox Y^oy p_t as an out_xs x out_ys coefficient matrix (tkmk_poly_lincomb: one fused pass)"""
    n = len(terms)
    coeffs = np.ascontiguousarray(np.concatenate([np.asarray(t[0], np.uint8).reshape(32) for t in terms])) if n else np.zeros(32, np.uint8)
    ptrs = (ctypes.c_void_p * max(n, 1))(*[_p(t[1]).value for t in terms])
    arr = lambda k, d: (ctypes.c_uint32 * max(n, 1))(*[(t[k] if len(t) > k else d) for t in terms])   # noqa: E731
    out = DeviceBuffer(32 * out_xs * out_ys)
    _check(lib().tkmk_poly_lincomb(n, _p(coeffs), ptrs, arr(2, 1), arr(3, 1), arr(4, 0), arr(5, 0), _p(out), int(out_xs), int(out_ys), None),
           "tkmk_poly_lincomb")
    return out


def native_stats_reset():
    _check(lib().tkmk_stats_reset(), "tkmk_stats_reset")


def native_stats():
    """work the library was asked to do since the last reset, counted inside the library (whichever host side called it)"""
    out = {}
    for name in ("msm.points", "msm.calls", "msm.bucket_additions", "ntt.elements", "ntt.calls", "poly.elements"):
        v = ctypes.c_uint64()
        _check(lib().tkmk_stats_get(name.encode(), ctypes.byref(v)), "tkmk_stats_get")
        out[name] = v.value
    return out


def diag_field_mul(field, a, b):
    """element-wise a*b through the device Montgomery product (field 0 = Fr, 1 = Fq); host arrays in/out"""
    width = 32 if field == 0 else 48
    da, db = DeviceBuffer.from_host(a), DeviceBuffer.from_host(b)
    do = DeviceBuffer(a.size)
    _check(lib().tkmk_diag_field_mul(int(field), _p(da), _p(db), _p(do), ctypes.c_uint64(a.size // width)), "tkmk_diag_field_mul")
    return do.to_host()


def gather_rows_device(src, row_bytes, idx, out=None):
    """out[i] = src[idx[i]] (rows of row_bytes); idx: numpy uint32 array or DeviceBuffer of u32"""
    if not isinstance(idx, DeviceBuffer):
        idx = DeviceBuffer.from_host(np.ascontiguousarray(idx, dtype=np.uint32).view(np.uint8))
    n = idx.nbytes // 4
    out = DeviceBuffer(row_bytes * n) if out is None else out
    _check(lib().tkmk_gather_rows_device(_p(src), ctypes.c_uint32(row_bytes), _p(idx), ctypes.c_uint64(n), _p(out), None),
           "tkmk_gather_rows_device")
    return out


class ExprInstr(ctypes.Structure):
    _fields_ = [("op", ctypes.c_uint8), ("arg", ctypes.c_uint8)]


def poly_expr_eval(prog, leaves, consts, n_consts, x_size, y_size, out=None):
    """prog: [(opcode, arg), ...] in postfix order (tkmk_expr_opcode); leaves: DeviceBuffers of x_size*y_size evaluations;
    consts: numpy uint8 (32 bytes per constant).  One kernel pass; returns the DeviceBuffer of result evaluations."""
    arr = (ExprInstr * len(prog))(*[ExprInstr(int(o), int(a)) for o, a in prog])
    ptrs = (ctypes.c_void_p * max(1, len(leaves)))(*[b.ptr for b in leaves])
    out = DeviceBuffer(32 * x_size * y_size) if out is None else out
    _check(lib().tkmk_poly_expr_eval(arr, ctypes.c_uint32(len(prog)), ptrs, ctypes.c_uint32(len(leaves)), _p(consts),
                                     ctypes.c_uint32(n_consts), ctypes.c_uint32(x_size), ctypes.c_uint32(y_size), _p(out), None),
           "tkmk_poly_expr_eval")
    return out


def poly_mul_ones_x(p, x_size, y_size, m, scale, out_x_size, out=None):
    """out = p * scale * (1 + X + ... + X^(m-1)), out_x_size rows (tkmk_poly_mul_ones_x); p: DeviceBuffer of x_size * y_size coefficients"""
    out = DeviceBuffer(32 * out_x_size * y_size) if out is None else out
    _check(lib().tkmk_poly_mul_ones_x(_p(p), ctypes.c_uint32(x_size), ctypes.c_uint32(y_size), ctypes.c_uint32(m), _p(scale),
                                      ctypes.c_uint32(out_x_size), _p(out), None), "tkmk_poly_mul_ones_x")
    return out


class ExprLeaf(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("x_len", ctypes.c_uint32), ("y_len", ctypes.c_uint32), ("rot_x", ctypes.c_uint32),
                ("rot_y", ctypes.c_uint32)]


def poly_expr_eval_views(prog, leaves, consts, n_consts, x_size, y_size, out=None):
    """as poly_expr_eval with leaf views: leaves = [(DeviceBuffer, x_len, y_len, rot_x, rot_y), ...] (tkmk_expr_leaf)"""
    arr = (ExprInstr * len(prog))(*[ExprInstr(int(o), int(a)) for o, a in prog])
    lv = (ExprLeaf * max(1, len(leaves)))(*[ExprLeaf(b.ptr, int(xl), int(yl), int(rx), int(ry)) for b, xl, yl, rx, ry in leaves])
    out = DeviceBuffer(32 * x_size * y_size) if out is None else out
    _check(lib().tkmk_poly_expr_eval_views(arr, ctypes.c_uint32(len(prog)), lv, ctypes.c_uint32(len(leaves)), _p(consts),
                                           ctypes.c_uint32(n_consts), ctypes.c_uint32(x_size), ctypes.c_uint32(y_size), _p(out), None),
           "tkmk_poly_expr_eval_views")
    return out


def poly_expr_eval_views_slab(prog, leaves, consts, n_consts, x_global, x_first, x_rows, y_size, out=None):
    """tkmk_poly_expr_eval_views_slab: the evaluator on the rows [x_first, x_first + x_rows) of a domain with x_global rows; leaves as
    poly_expr_eval_views with x_len in {x_rows, 1} and rot_x = 0"""
    arr = (ExprInstr * len(prog))(*[ExprInstr(int(o), int(a)) for o, a in prog])
    lv = (ExprLeaf * max(1, len(leaves)))(*[ExprLeaf(b.ptr if not isinstance(b, int) else b, int(xl), int(yl), int(rx), int(ry)) for b, xl, yl, rx, ry in leaves])
    out = DeviceBuffer(32 * x_rows * y_size) if out is None else out
    _check(lib().tkmk_poly_expr_eval_views_slab(arr, ctypes.c_uint32(len(prog)), lv, ctypes.c_uint32(len(leaves)), _p(consts), ctypes.c_uint32(n_consts),
                                                ctypes.c_uint32(x_global), ctypes.c_uint32(x_first), ctypes.c_uint32(x_rows), ctypes.c_uint32(y_size), _p(out), None),
           "tkmk_poly_expr_eval_views_slab")
    return out


def vec_suffix_product(a, out=None):
    """out[i] = prod_{j > i} a[j], out[n-1] = 1 (prove1's running product: prove/src/lib.rs:1858-1862)"""
    a = a if isinstance(a, DeviceBuffer) else DeviceBuffer.from_host(a)
    out = DeviceBuffer(a.nbytes) if out is None else out
    _check(lib().tkmk_vec_suffix_product(_p(a), ctypes.c_uint64(a.nbytes // 32), _p(out), None), "tkmk_vec_suffix_product")
    return out


def release_scratch():
    _check(lib().tkmk_release_scratch(), "tkmk_release_scratch")
