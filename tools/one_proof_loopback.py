#!/usr/bin/env python3
"""How much of a sharded proof does NOT divide?  Measured on ONE GPU over the loopback transport (include/tkmk_dist.h): G virtual ranks
of tkmk_prover_open_sharded take turns on the device, so the wall of one proof is the SUM of the ranks' device work — what divides by G
stays constant as G grows, what is replicated grows by itself for every added rank:
    wall(G) ~ divided + G * replicated (+ the loopback's own copies in place of xGMI transfers)
so replicated ~ (wall(G) - wall(1)) / (G - 1), and a rank on real hardware spends about wall(G) / G + its collectives.  Also reports the
library's work counters (all ranks together) against the single-GPU context's.  Not a scaling measurement: no second GPU is involved.
usage (GPU box): python tools/one_proof_loopback.py [--s-max 1024] [--worlds 1,2,4,8] [--repeat 3]"""
import argparse
import json
import os
import shutil
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tokamak-zk-evm_amd"), os.path.join(ROOT, "tools")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--s-max", type=int, default=1024)
    ap.add_argument("--placements", type=int, default=None)
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--repeat", type=int, default=3)
    args = ap.parse_args()
    import prove_bench
    import tkmk
    from tkmk import dist, service
    tkmk.set_device(0)
    files = prove_bench.stage_files(s_max=args.s_max, placements=args.placements)
    out = {"workload": files["workload"], "note": "loopback transport: virtual ranks take turns on ONE GPU; wall = sum of the ranks' device work"}
    try:
        base = None
        for world in [int(w) for w in args.worlds.split(",")]:
            tkmk.release_scratch()
            t = time.perf_counter()
            if world == 1:
                provers, comms = [service.Prover(files["qap"], files["crs"])], [None]
                run = lambda: [provers[0].prove(files["synth"], None, want_json=False)[1]]          # noqa: E731
            else:
                comms = dist.loopback_comms(world)
                provers = dist.run_ranks(comms, lambda c: service.Prover(files["qap"], files["crs"], comm=c))
                by_rank = {p.comm.rank: p for p in provers}
                run = lambda: dist.run_ranks(comms, lambda c: by_rank[c.rank].prove(files["synth"], None, want_json=False)[1])   # noqa: E731
            open_s = time.perf_counter() - t
            run()
            walls, stats = [], None
            for _ in range(args.repeat):
                tkmk.native_stats_reset()
                t = time.perf_counter()
                tms = run()
                tkmk.synchronize()
                walls.append(time.perf_counter() - t)
                stats = tkmk.native_stats()
            wall = statistics.median(walls)
            rec = {"world": world, "open_s": round(open_s, 2), "wall_ms_per_proof_all_ranks_in_turn": round(wall * 1e3, 2), "wall_over_world_ms": round(wall * 1e3 / world, 2),
                   "counters_all_ranks": stats, "parse_s_per_rank": round(statistics.median(t_["parse_s"] for t_ in tms), 5)}
            if base is None and world == 1:
                base = wall
            elif base is not None:
                rec["replicated_ms_estimate"] = round((wall - base) * 1e3 / (world - 1), 2)
            out.setdefault("runs", []).append(rec)
            print(json.dumps(rec), file=sys.stderr, flush=True)
            for p in provers:
                p.close()
            for c in comms:
                if c is not None:
                    c.close()
    finally:
        shutil.rmtree(files["tmp"], ignore_errors=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
