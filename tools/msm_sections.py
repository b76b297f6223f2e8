"""Per-section kernel times (tkmk_profile_*) of one MSM for a chosen scalar pattern: uniform | ones | witness.
usage: python tools/msm_sections.py LOGN PATTERN"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tokamak-zk-evm_amd"))
import numpy as np  # noqa: E402
import tkmk  # noqa: E402

tkmk.set_device(0)
logn, pat = int(sys.argv[1]), sys.argv[2]
n = 1 << logn
gen = np.zeros(96, np.uint8)
gen[:48] = np.frombuffer(int("17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb", 16).to_bytes(48, "little"), np.uint8)
gen[48:] = np.frombuffer(int("08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1", 16).to_bytes(48, "little"), np.uint8)
bases = tkmk.g1_batch_scalar_mul_device(tkmk.fr_random_device(1, n), gen, n)
if pat == "uniform":
    sc = tkmk.fr_random_device(2, n)
else:
    s = np.zeros((n, 32), np.uint8)
    if pat == "ones":
        s[:, 0] = 1
    else:
        rng = np.random.default_rng(3)
        cls = rng.random(n)
        s[(cls >= 0.45) & (cls < 0.90), 0] = 1
        small = (cls >= 0.90) & (cls < 0.94)
        s[small, 0] = rng.integers(2, 256, small.sum(), dtype=np.uint8)
        mid = cls >= 0.94
        s[mid, :16] = rng.integers(0, 256, (mid.sum(), 16), dtype=np.uint8)
    sc = tkmk.DeviceBuffer.from_host(s.reshape(-1))
tkmk.msm(sc, bases)
tkmk.profile_enable(True)
tkmk.profile_reset()
for _ in range(3):
    tkmk.msm(sc, bases)
tkmk.profile_enable(False)
out = {"logn": logn, "pattern": pat}
for name in ("convert_bases", "digits", "hist", "scan", "scatter", "accumulate", "combine", "reduce_segments", "reduce_windows"):
    ms, cnt = tkmk.profile_get("msm." + name)
    if cnt:
        out[name] = round(ms / cnt, 4)
print(json.dumps(out))
