"""HBM-streaming kernels of the polynomial layer: achieved bandwidth at the production size (2^23 Fr elements = 256 MiB).
usage: python tools/vecops_bench.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tokamak-zk-evm_amd"))
import numpy as np  # noqa: E402
import tkmk  # noqa: E402
from tkmk.poly import DensePolynomialExt as P  # noqa: E402

tkmk.set_device(0)
n = 1 << 23
a, b, o = tkmk.fr_random_device(1, n), tkmk.fr_random_device(2, n), tkmk.DeviceBuffer(32 * n)
s = np.frombuffer(tkmk.fr_random_device(3, 1).to_host(), np.uint8).copy()


def timed(fn, reps=10):
    fn()
    tkmk.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    tkmk.synchronize()
    return (time.perf_counter() - t0) / reps


GB = 32 * n / 1e9
cases = [
    ("vec_add", lambda: tkmk.vec_add(a, b, out=o), 3 * GB),
    ("vec_mul", lambda: tkmk.vec_mul(a, b, out=o), 3 * GB),
    ("scalar_mul", lambda: tkmk.scalar_mul(s, a, out=o), 2 * GB),
    ("vec_inv", lambda: tkmk.vec_inv(a, out=o), 2 * GB),
    ("transpose 16384x512", lambda: tkmk.transpose(a, 16384, 512, out=o), 2 * GB),
]
out = {}
for name, fn, traffic in cases:
    try:
        dt = timed(fn)
        out[name] = {"ms": round(dt * 1e3, 4), "GBps": round(traffic / dt, 1), "hbm_frac": round(traffic / dt / 8000, 3)}
    except Exception as e:   # signature drift in the binding should not hide the other lines
        out[name] = {"error": str(e)}
p = P.from_coeffs(tkmk.fr_random_device(5, n), 16384, 512)
for name, fn, traffic in (("find_degree", lambda: p.find_degree(), GB), ("eval(x,y)", lambda: p.eval(s, s), GB),
                          ("scale_coeffs_x", lambda: p.scale_coeffs_x(s), 2 * GB),
                          ("div_by_ruffini", lambda: p.div_by_ruffini(s, s), 4 * GB),
                          ("div_by_vanishing_opt(4096,256)", lambda: p.div_by_vanishing_opt(4096, 256), 3 * GB)):
    try:
        dt = timed(fn, 5)
        out[name] = {"ms": round(dt * 1e3, 4), "GBps_min": round(traffic / dt, 1)}
    except Exception as e:
        out[name] = {"error": str(e)}
print(json.dumps(out, indent=1))
