"""Timing of table-mode MSMs (tkmk_msm_multi_ex with a precomputed table: one bucket set, wide windows) against the plain
multi-window path, on synthetic resident data: per size, ms per MSM for c = 16 plain and table c in {16, 18, 20}; single jobs and
6 jobs per call (a prover round).  Prints one JSON line per measurement."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tokamak-zk-evm_amd"))
SEED = 0x746F6B616D616B00


def main():
    import tkmk
    tkmk.set_device(0)
    logs = [int(a) for a in sys.argv[1:]] or [20, 22, 24]
    nmax = 1 << max(logs)
    g = np.zeros(96, np.uint8)
    gx = [0xdb22c6bb, 0xfb3af00a, 0xf97a1aef, 0x6c55e83f, 0x171bac58, 0xa14e3a3f, 0x9774b905, 0xc3688c4f, 0x4fa9ac0f, 0x2695638c, 0x3197d794, 0x17f1d3a7]
    gy = [1187375073, 212476713, 2726857444, 3493644100, 738505709, 14358731, 3587181302, 4243972245, 1948093156, 2694721773, 3819610353, 146011265]
    g = np.frombuffer(b"".join(int(v).to_bytes(4, "little") for v in gx + gy), np.uint8).copy()
    s = tkmk.fr_random_device(SEED + 2, nmax)
    h = tkmk.fr_random_device(SEED + 3, nmax)
    plain = tkmk.g1_batch_scalar_mul_device(h, g, nmax)
    h.free()
    conv = tkmk.msm_convert_bases(plain, nmax)
    ref = {}
    modes = [int(v) for v in os.environ.get("TKMK_SWEEP_MODES", "0,16,18,20").split(",")]   # 0 = plain, else the table's window width
    for c in modes:
        table = conv
        windows = 0
        if c:
            windows = 255 // c + 1
            t0 = time.perf_counter()
            table = tkmk.msm_precompute_bases(plain, nmax, windows, c=c)
            tkmk.synchronize()
            build_s = time.perf_counter() - t0
        for logn in logs:
            n = 1 << logn
            for jobs_per_call in (1, 6):
                job = dict(scalars=s, bases=table, n=n, table_len=nmax)
                if c:
                    job["table"] = (c, windows)
                jobs = [job] * jobs_per_call
                res = tkmk.msm_multi_ex(jobs, bases_form=tkmk.BASES_CONVERTED)
                key = logn
                if c == 0:
                    ref[key] = bytes(res[:144])
                ok = bytes(res[:144]) == ref[key] if key in ref else None   # no plain run in this sweep
                tkmk.synchronize()
                reps = 3
                t0 = time.perf_counter()
                for _ in range(reps):
                    tkmk.msm_multi_ex(jobs, bases_form=tkmk.BASES_CONVERTED)
                tkmk.synchronize()
                ms = (time.perf_counter() - t0) / reps / jobs_per_call * 1e3
                sections = None
                if jobs_per_call == 1:
                    tkmk.profile_enable(True)
                    tkmk.profile_reset()
                    tkmk.msm_multi_ex(jobs, bases_form=tkmk.BASES_CONVERTED)
                    tkmk.profile_enable(False)
                    sections = {k: round(tkmk.profile_get("msm." + k)[0], 3) for k in ("digits", "hist", "scan", "scatter", "accumulate", "combine",
                                                                                       "reduce_segments", "reduce_windows")}
                print(json.dumps({"logn": logn, "mode": "table c=%d F=%d" % (c, windows) if c else "plain (auto c)", "jobs_per_call": jobs_per_call,
                                  "ms_per_msm": round(ms, 3), "Mpoints_per_s": round(n / ms / 1e3, 1), "equals_plain": ok,
                                  "table_build_s": round(build_s, 2) if c else None, "sections_ms": sections}), flush=True)
        if c:
            table.free()
        tkmk.release_scratch()


if __name__ == "__main__":
    main()
