"""Window width sweep at the prover's commit sizes: one MSM and six pipelined MSMs (tkmk_msm_multi) per (n, c).
usage: python tools/msm_c_sweep.py  -> JSON lines"""
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
sizes = [int(a) for a in sys.argv[1:]] or [1 << 18, 4097 * 257, 1 << 21, 8192 * 511]
for n in sizes:
    bases = tkmk.g1_batch_scalar_mul_device(tkmk.fr_random_device(1, n), gen, n)
    scal = [tkmk.fr_random_device(10 + k, n) for k in range(6)]
    rec = {"n": n, "auto_c": None, "single_ms": {}, "multi6_ms_each": {}}
    ref = None
    for c in (0, 11, 12, 13, 14, 15, 16):
        best1 = best6 = 1e9
        for rep in range(4):
            tkmk.synchronize()
            t = time.perf_counter()
            r = tkmk.msm(scal[0], bases, c=c)
            best1 = min(best1, time.perf_counter() - t)
        for rep in range(3):
            tkmk.synchronize()
            t = time.perf_counter()
            tkmk.msm_multi([(s, bases, n) for s in scal], c=c)
            best6 = min(best6, time.perf_counter() - t)
        if ref is None:
            ref = bytes(r)
        assert bytes(r) == ref, "result differs between window widths"
        key = "auto" if c == 0 else str(c)
        rec["single_ms"][key] = round(1e3 * best1, 3)
        rec["multi6_ms_each"][key] = round(1e3 * best6 / 6, 3)
    print(json.dumps(rec), flush=True)
