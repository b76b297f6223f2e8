"""gpu tier: the device-runtime part of the C ABI (include/tkmk.h: streams, async copies, device results, the allocation
cache) as the reference uses it through icicle_runtime (IcicleStream, DeviceVec, `are_results_on_device`, `is_async` in
setup/mpc-setup/src/lib.rs:91-122)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_streams_async_copies_and_device_results(gpu, oracle):
    lib = gpu.lib()
    st = ctypes.c_void_p()
    gpu._check(lib.tkmk_stream_create(ctypes.byref(st)), "tkmk_stream_create")
    try:
        n = 3000
        s = oracle.fr_random(1, n)
        p = oracle.g1_random_bases(2, n)
        ds, dp = gpu.DeviceBuffer(s.size), gpu.DeviceBuffer(p.size)
        gpu._check(lib.tkmk_memcpy_h2d_async(gpu._p(ds), gpu._p(s), ctypes.c_size_t(s.size), st), "h2d_async")
        gpu._check(lib.tkmk_memcpy_h2d_async(gpu._p(dp), gpu._p(p), ctypes.c_size_t(p.size), st), "h2d_async")
        # MSM on the user's stream with the result left on the device, then an async copy back on the same stream
        cfg = lib.tkmk_msm_default_config()
        cfg.stream_handle = st
        cfg.are_scalars_on_device = cfg.are_points_on_device = cfg.are_results_on_device = True
        dres = gpu.DeviceBuffer(144)
        gpu._check(lib.bls12_381_msm(gpu._p(ds), gpu._p(dp), n, ctypes.byref(cfg), gpu._p(dres)), "bls12_381_msm")
        res = np.empty(144, np.uint8)
        gpu._check(lib.tkmk_memcpy_d2h_async(gpu._p(res), gpu._p(dres), ctypes.c_size_t(144), st), "d2h_async")
        gpu._check(lib.tkmk_stream_synchronize(st), "tkmk_stream_synchronize")
        assert (gpu.projective_to_affine_bytes(res) == oracle.g1_msm(s, p)).all()
        # NTT on the stream, asynchronous (device in / out), then a vector op on the same stream consuming it
        gpu.init_ntt_domain_for_size(1 << 12)
        x = oracle.fr_random(3, 1 << 12)
        dx = gpu.DeviceBuffer(x.size)
        gpu._check(lib.tkmk_memcpy_h2d_async(gpu._p(dx), gpu._p(x), ctypes.c_size_t(x.size), st), "h2d_async")
        ncfg = lib.tkmk_ntt_default_config()
        ncfg.stream_handle = st
        ncfg.are_inputs_on_device = ncfg.are_outputs_on_device = True
        ncfg.is_async = True
        dy = gpu.DeviceBuffer(x.size)
        gpu._check(lib.bls12_381_ntt(gpu._p(dx), 1 << 12, 0, ctypes.byref(ncfg), gpu._p(dy)), "bls12_381_ntt")
        vcfg = lib.tkmk_vecops_default_config()
        vcfg.stream_handle = st
        vcfg.is_a_on_device = vcfg.is_b_on_device = vcfg.is_result_on_device = True
        vcfg.is_async = True
        dz = gpu.DeviceBuffer(x.size)
        gpu._check(lib.bls12_381_vector_mul(gpu._p(dy), gpu._p(dy), ctypes.c_uint64(1 << 12), ctypes.byref(vcfg), gpu._p(dz)), "vector_mul")
        gpu._check(lib.tkmk_stream_synchronize(st), "tkmk_stream_synchronize")
        ev = oracle.ntt(x, 1 << 12)
        assert (dz.to_host() == oracle.fr_mul(ev, ev)).all()
    finally:
        gpu._check(lib.tkmk_stream_destroy(st), "tkmk_stream_destroy")


def test_strided_copy_and_allocation_cache(gpu, oracle):
    lib = gpu.lib()
    rows, width, spitch, dpitch = 37, 96 * 5, 96 * 9, 96 * 6
    src = np.arange(rows * spitch, dtype=np.uint32).astype(np.uint8)
    d_src = gpu.DeviceBuffer.from_host(src)
    d_dst = gpu.DeviceBuffer(rows * dpitch)
    gpu._check(lib.tkmk_memset(gpu._p(d_dst), 0xAB, ctypes.c_size_t(rows * dpitch)), "tkmk_memset")
    gpu._check(lib.tkmk_memcpy_2d_d2d(gpu._p(d_dst), ctypes.c_size_t(dpitch), gpu._p(d_src), ctypes.c_size_t(spitch), ctypes.c_size_t(width),
                                      ctypes.c_size_t(rows)), "tkmk_memcpy_2d_d2d")
    got = d_dst.to_host().reshape(rows, dpitch)
    assert (got[:, :width] == src.reshape(rows, spitch)[:, :width]).all() and (got[:, width:] == 0xAB).all()
    # freed blocks are handed out again (same size class) and arrive clean of pending work; release returns them
    total, free0 = gpu.available_memory()
    a = gpu.DeviceBuffer(64 << 20)
    ptr = a.ptr
    gpu._check(lib.tkmk_memset(gpu._p(a), 0x5A, ctypes.c_size_t(64 << 20)), "tkmk_memset")
    a.free()
    b = gpu.DeviceBuffer(64 << 20)
    assert b.ptr == ptr                                       # reused from the cache
    assert (b.to_host(4096) == 0x5A).all()                    # ordinary memory: the earlier fill is still there, nothing stale or torn
    b.free()
    _, free1 = gpu.available_memory()
    assert free1 >= free0 - (1 << 20)                         # cached bytes count as available
    gpu.release_scratch()
    c = gpu.DeviceBuffer(1 << 20)
    assert c.to_host(16).size == 16
    with pytest.raises(gpu.TkmkError):
        gpu.DeviceBuffer(1 << 50)                             # absurd size: OUT_OF_MEMORY / ALLOCATION_FAILED, not a crash


def test_free_then_reuse_waits_for_work_on_caller_streams(gpu, oracle):
    """tkmk_free names no stream (Drop for DeviceVec): a block released while a kernel queued on a tkmk_stream_create stream is still
    using it must not be handed to the next tkmk_malloc before that kernel is done — the reuse cache orders the hand-out behind
    every caller stream (csrc/runtime.hip).  Also the regression test of the allocator's own free-list reuse for the
    stale-data failure characterised in tools/coherence_test.hip: what the new owner reads is what the last writer wrote.
    One pass, no repetition."""
    lib = gpu.lib()
    st = ctypes.c_void_p()
    gpu._check(lib.tkmk_stream_create(ctypes.byref(st)), "tkmk_stream_create")
    try:
        n = 1 << 22                                              # 128 MiB operands: the queued work takes a while
        gpu.init_ntt_domain_for_size(n)
        x = oracle.fr_random(31, 1 << 12)
        tile = np.tile(np.asarray(x), n >> 12)
        src = gpu.DeviceBuffer.from_host(tile)
        ncfg = lib.tkmk_ntt_default_config()
        ncfg.stream_handle = st
        ncfg.are_inputs_on_device = ncfg.are_outputs_on_device = True
        ncfg.is_async = True
        ncfg.batch_size = n >> 12
        vcfg = lib.tkmk_vecops_default_config()
        vcfg.stream_handle = st
        vcfg.is_a_on_device = vcfg.is_b_on_device = vcfg.is_result_on_device = True
        vcfg.is_async = True
        victim = gpu.DeviceBuffer(32 * n)
        result = gpu.DeviceBuffer(32 * n)
        # stream st: victim = NTT rows(src); a few dependent passes over victim; result = victim * victim   (all asynchronous)
        gpu._check(lib.bls12_381_ntt(gpu._p(src), 1 << 12, 0, ctypes.byref(ncfg), gpu._p(victim)), "bls12_381_ntt")
        for _ in range(6):
            gpu._check(lib.bls12_381_vector_add(gpu._p(victim), gpu._p(src), ctypes.c_uint64(n), ctypes.byref(vcfg), gpu._p(victim)), "vector_add")
        for _ in range(6):
            gpu._check(lib.bls12_381_vector_sub(gpu._p(victim), gpu._p(src), ctypes.c_uint64(n), ctypes.byref(vcfg), gpu._p(victim)), "vector_sub")
        gpu._check(lib.bls12_381_vector_mul(gpu._p(victim), gpu._p(victim), ctypes.c_uint64(n), ctypes.byref(vcfg), gpu._p(result)), "vector_mul")
        # drop the victim while all of that is still queued, and immediately ask for a block of the same class
        ptr = victim.ptr
        victim.free()
        thief = gpu.DeviceBuffer(32 * n)
        assert thief.ptr == ptr                                  # the cache did hand the same block out ...
        gpu._check(lib.tkmk_memset(gpu._p(thief), 0xEE, ctypes.c_size_t(32 * n)), "tkmk_memset")   # ... and the new owner scribbles over it at once
        gpu._check(lib.tkmk_stream_synchronize(st), "tkmk_stream_synchronize")
        ev = np.asarray(oracle.ntt(np.asarray(x), 1 << 12))
        want = np.asarray(oracle.fr_mul(ev, ev))
        got = np.asarray(result.to_host()).reshape(n >> 12, -1)
        assert (got == want).all()                               # the queued kernels saw their own data, not the scribble
        assert (np.asarray(thief.to_host(1 << 16)) == 0xEE).all()
    finally:
        gpu._check(lib.tkmk_stream_destroy(st), "tkmk_stream_destroy")


def test_msm_batches_on_two_streams_from_two_threads(gpu, oracle):
    """include/tkmk.h THREADING: MSM batches that name different caller streams run concurrently, each over its own pipeline set (streams,
    events, pinned result buffers); batches on one stream take turns.  Two threads each issue batches of four 2^15-point MSMs on their own
    stream (one of them marked background: tkmk_stream_set_background), several times, while the main thread does the same on the default
    stream: every result equals the oracle's, and a destroyed stream's set is gone with it (a fresh stream afterwards works)."""
    import threading
    lib = gpu.lib()
    n, batch, rounds = 1 << 15, 4, 3
    bases = oracle.g1_random_bases(71, n)
    d_bases = gpu.DeviceBuffer.from_host(bases)
    scal = {k: oracle.fr_random(900 + k, n * batch) for k in range(3)}
    want = {k: [np.asarray(oracle.g1_msm(scal[k][32 * n * j:32 * n * (j + 1)].copy(), bases)) for j in range(batch)] for k in range(3)}
    streams = []
    for _ in range(2):
        st = ctypes.c_void_p()
        gpu._check(lib.tkmk_stream_create(ctypes.byref(st)), "tkmk_stream_create")
        streams.append(st)
    # one of the two is a BACKGROUND stream (its batches' accumulate kernels take one workgroup per CU): same results
    gpu._check(lib.tkmk_stream_set_background(streams[1], 1), "tkmk_stream_set_background")
    assert lib.tkmk_stream_set_background(None, 1) != 0        # the default stream is the foreground by definition
    errors = []

    def work(k, st):
        try:
            d_s = gpu.DeviceBuffer.from_host(scal[k])
            for _ in range(rounds):
                res = gpu.projective_to_affine_bytes(gpu.msm(d_s, d_bases, msm_size=n, batch=batch, stream=st))
                for j in range(batch):
                    if not (res[96 * j:96 * (j + 1)] == want[k][j]).all():
                        errors.append((k, j))
        except Exception as e:          # noqa: BLE001 — reported by the main thread
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=work, args=(k, streams[k])) for k in range(2)]
    for t in threads:
        t.start()
    work(2, None)
    for t in threads:
        t.join()
    assert not errors, errors
    for st in streams:
        gpu._check(lib.tkmk_stream_destroy(st), "tkmk_stream_destroy")
    st = ctypes.c_void_p()
    gpu._check(lib.tkmk_stream_create(ctypes.byref(st)), "tkmk_stream_create")
    try:
        work(0, st)
        assert not errors, errors
    finally:
        gpu._check(lib.tkmk_stream_destroy(st), "tkmk_stream_destroy")
