"""MSM timing on witness-like scalar distributions (wires are overwhelmingly 0/1 bits, 128-bit
limbs and small constants) against uniform scalars of the same size.  usage: python tools/msm_skew_bench.py [logn]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tokamak-zk-evm_amd"))
import numpy as np  # noqa: E402
import tkmk  # noqa: E402

tkmk.set_device(0)
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n = 1 << logn
gen = np.zeros(96, np.uint8)
gen[:48] = np.frombuffer(int("17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb", 16).to_bytes(48, "little"), np.uint8)
gen[48:] = np.frombuffer(int("08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1", 16).to_bytes(48, "little"), np.uint8)
bases = tkmk.g1_batch_scalar_mul_device(tkmk.fr_random_device(1, n), gen, n)
uni = tkmk.fr_random_device(2, n)
rng = np.random.default_rng(3)
cls = rng.random(n)
s = np.zeros((n, 32), np.uint8)
s[(cls >= 0.45) & (cls < 0.90), 0] = 1                               # 45 % zeros, 45 % ones
small = (cls >= 0.90) & (cls < 0.94)
s[small, 0] = rng.integers(2, 256, small.sum(), dtype=np.uint8)      # small constants
mid = (cls >= 0.94) & (cls < 0.99)
s[mid, :16] = rng.integers(0, 256, (mid.sum(), 16), dtype=np.uint8)  # 128-bit limbs
full = cls >= 0.99
s[full] = uni.to_host().reshape(n, 32)[full]
cases = {"uniform": uni, "witness_like": tkmk.DeviceBuffer.from_host(s.reshape(-1)), "all_ones": None, "all_zero": tkmk.DeviceBuffer.from_host(np.zeros(32 * n, np.uint8))}
ones = np.zeros((n, 32), np.uint8)
ones[:, 0] = 1
cases["all_ones"] = tkmk.DeviceBuffer.from_host(ones.reshape(-1))
out = {"logn": logn}
for name, sc in cases.items():
    tkmk.msm(sc, bases)
    tkmk.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        tkmk.msm(sc, bases)
    tkmk.synchronize()
    out[name + "_ms"] = round((time.perf_counter() - t0) / 3 * 1e3, 3)
print(json.dumps(out))
