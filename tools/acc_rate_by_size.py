#!/usr/bin/env python3
"""Why do the prover's accumulate launches run at 6.1 x 10^9 additions/s when the 2^24-point MSM runs at 6.5 x 10^9?
Measured per launch, not guessed: every commit shape of a configs[3] proof as ONE table-mode MSM (20-bit windows over the 13-level table of
a 8192 x 2048 grid, as the resident prover holds it), alone on the device (one pipeline stream), with
  * the PLAN of the launch restated from csrc/msm_impl.inc:msm_choose_chunk — entries, chunk length, lanes, rounds of 131 072 lanes, and
    how full the LAST round is (the first suspect named by the verdict);
  * the measured k_accumulate_chunks time (HIP events on the launch stream) and the additions the library counted on the device.
Scalars: dense uniform 255-bit (the quotient polynomials), 'small' (values below 2^20 with 30 % zeros: the A.w / B.w / C.w evaluations) and
'sparse' (6 % non-zero: b's evaluations).  Prints one JSON line per shape and a summary.
usage (GPU box): python tools/acc_rate_by_size.py [--reps 3]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tokamak-zk-evm_amd"), ROOT]

SLOTS = 256 * 4 * 2 * 64          # CUs x SIMDs x waves per SIMD at this kernel's register count x lanes (msm_choose_chunk)


def plan(entries_upper, n_expanded):
    """msm_choose_chunk(entries, W = 1, n) of csrc/msm_impl.inc, then the launch geometry of msm_enqueue"""
    target = min(max(entries_upper // (8 * SLOTS), 48), 224)
    r = max((entries_upper + target * SLOTS // 2) // (target * SLOTS), 1)
    lanes = max(r * SLOTS, 1)
    chunk = min(max(-(-n_expanded // lanes), 32), 256)
    return chunk, r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import tkmk
    tkmk.set_device(0)
    tkmk.msm_set_pipeline_streams(1)
    rs_x, rs_y, windows, c = 8192, 2048, 13, 20
    n_table = rs_x * rs_y
    gen = np.zeros(96, np.uint8)
    gx = 0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb
    gy = 0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1
    gen[:48] = np.frombuffer(gx.to_bytes(48, "little"), np.uint8)
    gen[48:] = np.frombuffer(gy.to_bytes(48, "little"), np.uint8)
    h = tkmk.fr_random_device(0x746F6B01, n_table)
    bases = tkmk.g1_batch_scalar_mul_device(h, gen, n_table)
    h.free()
    table = tkmk.msm_precompute_bases(bases, n_table, windows, c=c)
    bases.free()
    dense = tkmk.fr_random_device(0x746F6B02, n_table)
    host = dense.to_host(32 * rs_y * 4097).reshape(-1, 32).copy()       # the 4097 x 2048 corner is enough for the small / sparse kinds
    rnd = np.random.default_rng(7)
    small = host.copy()
    small[:, 3:] = 0
    small[:, 2] &= 0x0f                                                  # < 2^20
    small[rnd.random(small.shape[0]) < 0.3] = 0
    sparse = host.copy()
    sparse[rnd.random(sparse.shape[0]) >= 0.06] = 0
    d_small = tkmk.DeviceBuffer.from_host(np.ascontiguousarray(small.reshape(-1)))
    d_sparse = tkmk.DeviceBuffer.from_host(np.ascontiguousarray(sparse.reshape(-1)))
    shapes = [("Q_AX", 4097, 2047, dense), ("Q_AY", 4097, 1025, dense), ("Q_CX", 8192, 2047, dense), ("Q_CY", 8191, 1025, dense), ("Pi_X", 8191, 2047, dense),
              ("M_X", 4096, 1024, dense), ("whole 2^24", 8192, 2048, dense), ("u/v/w evaluations (small)", 4096, 1024, d_small), ("b evaluations (sparse)", 4096, 1024, d_sparse)]
    out = []
    for name, tx, ty, sc in shapes:
        job = dict(scalars=sc, bases=table, n=tx * ty, scalar_view=(ty, rs_y), base_view=(ty, rs_y), table_len=n_table, table=(c, windows))
        tkmk.msm_multi_ex([job], bases_form=tkmk.BASES_CONVERTED)
        tkmk.synchronize()
        tkmk.profile_enable(True)
        tkmk.profile_reset()
        tkmk.native_stats_reset()
        for _ in range(args.reps):
            tkmk.msm_multi_ex([job], bases_form=tkmk.BASES_CONVERTED)
        tkmk.synchronize()
        tkmk.profile_enable(False)
        ms, cnt = tkmk.profile_get("msm.accumulate")
        adds = tkmk.native_stats()["msm.bucket_additions"] / args.reps
        n_exp = tx * ty * windows
        chunk, rounds_planned = plan(n_exp, n_exp)
        lanes = -(-n_exp // chunk)                                       # launched (upper bound: all digits non-zero)
        busy_lanes = -(-int(adds) // chunk)                              # lanes whose chunk holds entries
        rounds = -(-busy_lanes // SLOTS)
        last = busy_lanes - (rounds - 1) * SLOTS
        rec = {"commit": name, "box": [tx, ty], "points": tx * ty, "entries": int(adds), "entries_upper_bound": n_exp, "chunk": chunk,
               "lanes_launched": lanes, "lanes_with_work": busy_lanes, "rounds": rounds, "last_round_fill": round(last / SLOTS, 4),
               "accumulate_ms": round(ms / cnt, 3), "additions_per_s": round(adds / (ms / cnt * 1e-3) / 1e9, 3)}
        out.append(rec)
        print(json.dumps(rec), flush=True)
    tot_adds = sum(r["entries"] for r in out[:6])
    tot_ms = sum(r["accumulate_ms"] for r in out[:6])
    print(json.dumps({"summary": "dense quotient commits of one proof", "additions": tot_adds, "accumulate_ms": round(tot_ms, 2),
                      "additions_per_s_e9": round(tot_adds / tot_ms / 1e6, 3)}))


if __name__ == "__main__":
    main()
