"""MSM with a resident precomputed base table (precompute_factor) vs the plain call.  usage: python tools/msm_precompute_bench.py [logn ...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tokamak-zk-evm_amd"))
import numpy as np  # noqa: E402
import tkmk  # noqa: E402

tkmk.set_device(0)
gen = np.zeros(96, np.uint8)
gen[:48] = np.frombuffer(int("17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb", 16).to_bytes(48, "little"), np.uint8)
gen[48:] = np.frombuffer(int("08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1", 16).to_bytes(48, "little"), np.uint8)
for logn in [int(a) for a in sys.argv[1:]] or [18, 20, 22]:
    n = 1 << logn
    bases = tkmk.g1_batch_scalar_mul_device(tkmk.fr_random_device(1, n), gen, n)
    sc = tkmk.fr_random_device(2, n)
    row = {"logn": logn}
    ref = tkmk.msm(sc, bases)

    def timed(fn, reps=5):
        fn()
        tkmk.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        tkmk.synchronize()
        return round((time.perf_counter() - t0) / reps * 1e3, 3)

    row["plain_ms"] = timed(lambda: tkmk.msm(sc, bases))
    for f in (2, 4, 16):
        t0 = time.perf_counter()
        table = tkmk.msm_precompute_bases(bases, n, f)
        tkmk.synchronize()
        row["table_f%d_build_s" % f] = round(time.perf_counter() - t0, 3)
        assert (tkmk.msm(sc, table, msm_size=n, precompute_factor=f) == ref).all()
        row["f%d_ms" % f] = timed(lambda: tkmk.msm(sc, table, msm_size=n, precompute_factor=f))
        table.free()
    print(json.dumps(row), flush=True)
