#!/usr/bin/env python3
"""NTT shapes of the configs[3] prover (2^10 / 2^11-point rows, 2^14-point columns, the bivariate domains) timed through the C ABI;
run once per TKMK_NTT_MAX_LOGR value to compare pass plans (profiles/r03_ntt_max_radix_9_vs_10.txt)."""
import json, os, sys, time
sys.path.insert(0, "tokamak-zk-evm_amd")
import tkmk
tkmk.set_device(0)
tkmk.init_ntt_domain_for_size(1 << 25)
def timed(fn, reps=5):
    fn(); tkmk.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    tkmk.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for name, n, batch, cols in (("rows 16384 x 2^10", 1 << 10, 16384, False), ("rows 8192 x 2^10", 1 << 10, 8192, False), ("cols 1024 x 2^14", 1 << 14, 1024, True),
                             ("rows 4096 x 2^11", 1 << 11, 4096, False), ("rows 64 x 2^19", 1 << 19, 64, False)):
    a = tkmk.fr_random_device(7, n * batch); out = tkmk.DeviceBuffer(32 * n * batch)
    print("%-22s %.3f ms" % (name, timed(lambda: tkmk.ntt(a, n, batch=batch, columns_batch=cols, out=out))), flush=True)
    a.free(); out.free()
for xs, ys in ((16384, 1024), (8192, 1024), (16384, 2048)):
    a = tkmk.fr_random_device(9, xs * ys); out = tkmk.DeviceBuffer(32 * xs * ys)
    print("bintt %dx%d %.3f ms" % (xs, ys, timed(lambda: tkmk.bintt(a, xs, ys, out=out))), flush=True)
    a.free(); out.free()
