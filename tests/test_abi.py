"""not-gpu tier: the C-ABI library loads, exports every symbol include/tkmk.h declares, and fails loudly
(TKMK_ERR_NO_DEVICE) instead of falling back to a CPU path when no GPU is present."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "tkmk.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = re.findall(r"\b((?:tkmk|bls12_381|bn254)_[a-z0-9_]+)\s*\(", hdr)
    return sorted(set(names))


def test_header_and_binding_agree(tkmk):
    assert _declared_symbols() == sorted(tkmk.SYMBOLS)


def test_library_exports_every_declared_symbol(tkmk):
    lib = tkmk.lib()
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    assert lib.tkmk_is_hip_build() == 1


def test_prover_and_dist_libraries_export_their_headers(tkmk):
    """libtkmk_prover.so / libtkmk_dist.so load on a box without a GPU and export every symbol include/tkmk_prover.h /
    include/tkmk_dist.h declare; without a device the entry points fail loudly (TKMK_ERR_NO_DEVICE), they do not fall back"""
    import ctypes
    from tkmk import dist, service
    for mod, header in ((service, "tkmk_prover.h"), (dist, "tkmk_dist.h")):
        hdr = open(os.path.join(ROOT, "include", header)).read()
        hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
        declared = sorted(set(re.findall(r"\b(tkmk_[a-z0-9_]+)\s*\(", hdr)))
        assert declared == sorted(mod.SYMBOLS), header
        for name in declared:
            assert hasattr(mod.lib(), name), name
    if tkmk.device_count() == 0:
        h = ctypes.c_void_p()
        assert service.lib().tkmk_prover_open(b"/nonexistent", b"/nonexistent", ctypes.byref(h)) == 12
        assert dist.lib().tkmk_comm_init((ctypes.c_uint8 * 128)(), 1, 0, ctypes.byref(h)) == 12


def test_library_state_is_bound_to_one_device(tkmk):
    """tkmk_set_device's rule (csrc/runtime.hip): free choice before the first entry point has bound the library, the bound device
    again afterwards, TKMK_ERR_INVALID_DEVICE for any other — the arenas / allocation cache / NTT domain are per process, not per
    device (a host that hops devices like ICICLE's set_device allows would get device-0 scratch on device 1)"""
    v = tkmk.lib().tkmk_diag_device_switch
    assert v(-1, 0, 8) == 0 and v(-1, 5, 8) == 0          # nothing bound yet: any present device
    assert v(3, 3, 8) == 0                                 # re-selecting the bound device
    assert v(0, 1, 8) == 1 and v(3, 0, 8) == 1             # TKMK_ERR_INVALID_DEVICE: state lives on another device
    assert v(-1, 8, 8) == 1 and v(-1, -1, 8) == 1          # not a device
    assert v(-1, 0, 0) == 12 and v(0, 0, 0) == 12          # TKMK_ERR_NO_DEVICE


def test_config_defaults_match_icicle(tkmk):
    m = tkmk.lib().tkmk_msm_default_config()
    assert (m.precompute_factor, m.c, m.bitsize, m.batch_size, m.are_points_shared_in_batch) == (1, 0, 0, 1, True)
    assert not (m.are_scalars_on_device or m.are_points_on_device or m.are_results_on_device or m.is_async)
    n = tkmk.lib().tkmk_ntt_default_config()
    assert list(n.coset_gen.limbs) == [1, 0, 0, 0, 0, 0, 0, 0] and n.batch_size == 1 and not n.columns_batch
    v = tkmk.lib().tkmk_vecops_default_config()
    assert v.batch_size == 1 and not v.is_a_on_device


def test_root_of_unity_is_host_side_and_matches_oracle(tkmk, oracle):
    for n in (1, 2, 256, 1 << 23, 5):
        assert (tkmk.get_root_of_unity(n) == oracle.root_of_unity(n)).all()


def test_no_silent_cpu_fallback(tkmk):
    if tkmk.device_count() > 0:
        pytest.skip("a GPU is visible")
    a = np.zeros(64, np.uint8)
    for call in (lambda: tkmk.vec_add(a, a), lambda: tkmk.ntt(a, 2), lambda: tkmk.msm(a[:32], np.zeros(96, np.uint8)),
                 lambda: tkmk.init_ntt_domain_for_size(4), lambda: tkmk.DeviceBuffer(64)):
        with pytest.raises(tkmk.TkmkError) as e:
            call()
        assert e.value.code == 12  # TKMK_ERR_NO_DEVICE
