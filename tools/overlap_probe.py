"""Does polynomial work overlap a commit batch?  A thread runs four 2^22-point table commits (tkmk_msm_multi_ex) while the main thread
runs the transform / pointwise / linear-combination passes of prove0's quotient; wall of both together against each alone."""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tokamak-zk-evm_amd"))


def main():
    import tkmk
    tkmk.set_device(0)
    n = 1 << 22
    g = np.frombuffer(b"".join(int(v).to_bytes(4, "little") for v in
                               [0xdb22c6bb, 0xfb3af00a, 0xf97a1aef, 0x6c55e83f, 0x171bac58, 0xa14e3a3f, 0x9774b905, 0xc3688c4f, 0x4fa9ac0f, 0x2695638c, 0x3197d794, 0x17f1d3a7,
                                1187375073, 212476713, 2726857444, 3493644100, 738505709, 14358731, 3587181302, 4243972245, 1948093156, 2694721773, 3819610353, 146011265]), np.uint8).copy()
    s = tkmk.fr_random_device(11, n)
    h = tkmk.fr_random_device(12, n)
    plain = tkmk.g1_batch_scalar_mul_device(h, g, n)
    table = tkmk.msm_precompute_bases(plain, n, 13, c=20)
    jobs = [dict(scalars=s, bases=table, n=n, table_len=n, table=(20, 13))] * 4
    tkmk.init_ntt_domain_for_size(1 << 25)
    a, b = tkmk.fr_random_device(13, n), tkmk.fr_random_device(14, n)
    big = tkmk.DeviceBuffer(32 << 24)

    def msm():
        tkmk.msm_multi_ex(jobs, bases_form=tkmk.BASES_CONVERTED)

    def poly():
        ea = tkmk.bintt_padded(a, 4096, 1024, 8192, 2048)
        eb = tkmk.bintt_padded(b, 4096, 1024, 8192, 2048)
        tkmk.vec_mul(ea, eb, out=ea)
        tkmk.bintt(ea, 8192, 2048, inverse=True, out=big)
        for _ in range(6):     # the division / linear-combination passes: streaming over 2^24 elements
            tkmk.vec_add(big, eb, out=eb)
        ea.free()
        eb.free()

    def timed(fn, reps=3):
        fn()
        tkmk.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        tkmk.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    def both():
        t = threading.Thread(target=msm)
        t.start()
        poly()
        t.join()

    print(json.dumps({"msm_4x2p22_ms": round(timed(msm), 2), "poly_ms": round(timed(poly), 2), "both_concurrently_ms": round(timed(both), 2)}))


if __name__ == "__main__":
    main()
