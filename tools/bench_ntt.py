#!/usr/bin/env python3
"""NTT measurements on the GPU box: BASELINE.json configs[2] (256 x 2^20 row batch) and the reference's
production _biNTT shapes.  Prints one JSON line per case."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tokamak-zk-evm_amd"))
import tkmk  # noqa: E402

tkmk.set_device(0)
tkmk.init_ntt_domain_for_size(1 << 23)
reps = int(os.environ.get("REPS", "5"))


def timed(fn):
    fn()
    tkmk.synchronize()
    tkmk.profile_enable(True)
    tkmk.profile_reset()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    tkmk.synchronize()
    dt = (time.perf_counter() - t0) / reps
    tkmk.profile_enable(False)
    passes = {}
    for k in range(8):
        ms, cnt = tkmk.profile_get("ntt.pass%d" % k)
        if cnt:
            passes["pass%d" % k] = round(ms / cnt, 4)
    return dt, passes


cases = []
if "--small" not in sys.argv:
    cases.append(("rows 256 x 2^20 (cfg3)", 1 << 20, 256, False))
cases += [("rows 2^13 x 512", 512, 1 << 13, False), ("cols 512 x 2^13", 1 << 13, 512, True), ("1-D 2^22", 1 << 22, 1, False)]
for name, n, batch, cols in cases:
    a = tkmk.fr_random_device(7, n * batch)
    out = tkmk.DeviceBuffer(32 * n * batch)
    dt, passes = timed(lambda: tkmk.ntt(a, n, batch=batch, columns_batch=cols, out=out))
    el = n * batch
    print(json.dumps({"case": name, "ms": dt * 1e3, "elements_per_s": el / dt, "alg_GBps": 64 * el / dt / 1e9,
                      "butterflies_per_s": el * (n.bit_length() - 1) / 2 / dt, "passes_ms": passes}), flush=True)
    a.free()
    out.free()
for xs, ys in ((4096, 256), (8192, 512), (16384, 512)):
    a = tkmk.fr_random_device(9, xs * ys)
    out = tkmk.DeviceBuffer(32 * xs * ys)
    dt, passes = timed(lambda: tkmk.bintt(a, xs, ys, out=out))
    el = xs * ys
    print(json.dumps({"case": "bintt %dx%d" % (xs, ys), "ms": dt * 1e3, "elements_per_s": el / dt, "alg_GBps": 64 * el / dt / 1e9,
                      "passes_ms": passes}), flush=True)
    a.free()
    out.free()
