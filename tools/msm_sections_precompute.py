"""Per-section times of an MSM with precomputed bases.  usage: python tools/msm_sections_precompute.py LOGN FACTOR"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tokamak-zk-evm_amd"))
import numpy as np  # noqa: E402
import tkmk  # noqa: E402

tkmk.set_device(0)
logn, f = int(sys.argv[1]), int(sys.argv[2])
n = 1 << logn
gen = np.zeros(96, np.uint8)
gen[:48] = np.frombuffer(int("17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb", 16).to_bytes(48, "little"), np.uint8)
gen[48:] = np.frombuffer(int("08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1", 16).to_bytes(48, "little"), np.uint8)
bases = tkmk.g1_batch_scalar_mul_device(tkmk.fr_random_device(1, n), gen, n)
sc = tkmk.fr_random_device(2, n)
table = tkmk.msm_precompute_bases(bases, n, f) if f > 1 else bases
run = lambda: tkmk.msm(sc, table, msm_size=n, precompute_factor=f)   # noqa: E731
run()
tkmk.profile_enable(True)
tkmk.profile_reset()
for _ in range(3):
    run()
tkmk.profile_enable(False)
out = {"logn": logn, "factor": f}
for name in ("convert_bases", "digits", "hist", "scan", "scatter", "accumulate", "combine", "reduce_segments", "reduce_windows"):
    ms, cnt = tkmk.profile_get("msm." + name)
    if cnt:
        out[name] = round(ms / cnt, 4)
print(json.dumps(out))
