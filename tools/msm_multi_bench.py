"""Sequential bls12_381_msm calls vs one tkmk_msm_multi call over the same independent jobs (resident inputs).
usage: python tools/msm_multi_bench.py [logn ...]   (TKMK_MSM_STREAMS selects the internal stream count)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tokamak-zk-evm_amd"))
import numpy as np  # noqa: E402
import tkmk  # noqa: E402

tkmk.set_device(0)
logns = [int(a) for a in sys.argv[1:]] or [20, 22]
gen = np.zeros(96, np.uint8)
gx = 0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb
gy = 0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1
gen[:48] = np.frombuffer(gx.to_bytes(48, "little"), np.uint8)
gen[48:] = np.frombuffer(gy.to_bytes(48, "little"), np.uint8)
for logn in logns:
    n, jobs_n = 1 << logn, 8
    k = tkmk.fr_random_device(1, n)
    bases = tkmk.g1_batch_scalar_mul_device(k, gen, n)
    scal = [tkmk.fr_random_device(10 + j, n) for j in range(jobs_n)]
    jobs = [(s, bases, n) for s in scal]
    for _ in range(2):
        seq = [tkmk.msm(s, bases) for s in scal]
        mul = tkmk.msm_multi(jobs)
    assert all((mul[144 * j:144 * (j + 1)] == seq[j]).all() for j in range(jobs_n))
    tkmk.synchronize()
    t0 = time.perf_counter()
    for s in scal:
        tkmk.msm(s, bases)
    t_seq = time.perf_counter() - t0
    t0 = time.perf_counter()
    tkmk.msm_multi(jobs)
    t_mul = time.perf_counter() - t0
    print(json.dumps({"logn": logn, "jobs": jobs_n, "streams": os.environ.get("TKMK_MSM_STREAMS", "default"),
                      "sequential_ms_per_msm": round(1e3 * t_seq / jobs_n, 3), "multi_ms_per_msm": round(1e3 * t_mul / jobs_n, 3),
                      "speedup": round(t_seq / t_mul, 3)}), flush=True)
